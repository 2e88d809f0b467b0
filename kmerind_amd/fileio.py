"""Host-side input partitioning: the contract of the reference's partitioned_file<...,FASTQParser>
(src/io/file.hpp:1216-1430) for one node -- split a FASTQ buffer into `n` byte ranges that each
start at a true record start, using the 4-line rule of FASTQParser::find_first_record
(src/io/fastq_loader.hpp:269-364): after the first EOL at or after the nominal split point look at
the first characters of the next four lines; a record starts at the line where '@' is followed two
lines later by '+'."""
import numpy as np

_EOL = (10, 13)


def _is_eol(b):
    return b == 10 or b == 13


def find_first_record(data, pos):
    """first record start at or after byte `pos` (len(data) if none)"""
    n = len(data)
    if pos <= 0:
        return 0
    i = pos
    # not at an EOL: skip the rest of this (partial) line
    while i < n and not _is_eol(data[i]):
        i += 1
    starts, firsts = [], []
    for _ in range(4):
        while i < n and _is_eol(data[i]):
            i += 1
        if i >= n:
            return n
        starts.append(i)
        firsts.append(data[i])
        while i < n and not _is_eol(data[i]):
            i += 1
    at, plus = ord("@"), ord("+")
    if firsts[0] == at and firsts[2] == plus:
        return starts[0]
    if firsts[1] == at and firsts[3] == plus:
        return starts[1]
    if firsts[0] == plus and firsts[2] == at:
        return starts[2]
    if firsts[1] == plus and firsts[3] == at:
        return starts[3]
    return n


def partition_fastq(data, n_parts):
    """record-aligned [begin, end) byte ranges, one per rank; they tile the buffer exactly"""
    data = memoryview(data) if not isinstance(data, np.ndarray) else data
    n = len(data)
    cuts = [find_first_record(data, (n * r) // n_parts) for r in range(n_parts)] + [n]
    for r in range(1, n_parts + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(n_parts)]


def partition_fastq_device(ctx, dptr, nbytes, n_parts):
    """partition_fastq for a buffer resident in HBM: the four-line rule runs on the device (kmi_fastq_partition_dev),
    one thread per split point; same ranges as the host version"""
    import ctypes as C
    from . import _lib as L
    cuts = np.zeros(n_parts + 1, dtype=np.uint64)
    ctx.check(L.lib.kmi_fastq_partition_dev(ctx.h, C.c_void_p(dptr), nbytes, n_parts, cuts.ctypes.data_as(C.c_void_p)))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(n_parts)]


# ---------------------------------------------------------------------------
# FASTA: fixed byte blocks with k - 1 characters of overlap and the header bookkeeping that
# FASTAParser::init_parser obtains from the neighbouring ranks (src/io/fasta_loader.hpp:232-456, 485-604;
# block + overlap rule src/io/file.hpp:1564-1600, src/io/kmer_parser.hpp:112-157, kmer_file_helper.hpp:563)
# ---------------------------------------------------------------------------
FA_OUTSIDE, FA_HEADER, FA_SEQUENCE = 0, 1, 2


def fasta_line_kinds(data):
    """per byte: kind of the line it sits on (FA_*), 1 where a line starts, and the number of records
    (header group -> sequence group transitions) that have started at or before it. A line starts at byte 0 and
    after every '\n'; a line beginning with '>' or ';' is a header line."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.asarray(data, dtype=np.uint8)
    n = buf.size
    kind = np.zeros(n, dtype=np.uint8)
    recs = np.zeros(n, dtype=np.int64)
    starts = np.zeros(n, dtype=np.uint8)
    state, ev = FA_OUTSIDE, 0
    line_starts = np.concatenate(([0], np.flatnonzero(buf[:-1] == 10) + 1)) if n else np.zeros(0, dtype=np.int64)
    bounds = np.concatenate((line_starts, [n]))
    for a, b in zip(bounds[:-1], bounds[1:]):
        if buf[a] in (ord(">"), ord(";")):
            state = FA_HEADER
        else:
            if state == FA_HEADER:
                ev += 1
            state = FA_OUTSIDE if state == FA_OUTSIDE else FA_SEQUENCE
        kind[a:b] = state
        recs[a:b] = ev
        starts[a] = 1
    return kind, starts, recs


def partition_fasta(data, n_parts, k):
    """Byte blocks for `n_parts` ranks. Each entry: dict(begin, end, valid_bytes, start_state, at_line_start,
    records_before, index_shift): rank r parses bytes [begin, end), produces the k-mers whose first base lies in its
    first valid_bytes bytes; [begin + valid_bytes, end) is the overlap holding the k - 1 further sequence characters
    (or everything up to the end of the file / the next header)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.asarray(data, dtype=np.uint8)
    n = buf.size
    kind, starts, recs = fasta_line_kinds(buf)
    is_seq_char = (kind == FA_SEQUENCE) & (buf != 10) & (buf != 13)
    seq_rank = np.concatenate(([0], np.cumsum(is_seq_char)))      # sequence characters before byte i
    index_shift = 0 if (n and buf[0] in (ord(">"), ord(";"))) else 1
    cuts = [(n * r) // n_parts for r in range(n_parts)] + [n]
    out = []
    for r in range(n_parts):
        b, e = cuts[r], cuts[r + 1]
        # overlap: up to k - 1 more sequence characters
        need = seq_rank[e] + (k - 1)
        end = int(np.searchsorted(seq_rank, need, side="left")) if need <= seq_rank[n] else n
        end = min(max(end, e), n)
        if b >= n:
            out.append(dict(begin=n, end=n, valid_bytes=0, start_state=FA_OUTSIDE, at_line_start=1, records_before=0,
                            index_shift=index_shift))
            continue
        records_before = int(recs[b - 1]) if b > 0 else 0
        # state of the line byte b sits on; a line that STARTS at b is classified by the machine itself from the
        # state of the previous line
        if b == 0:
            st, ls = FA_OUTSIDE, 1
        elif starts[b]:
            st, ls = int(kind[b - 1]), 1
        else:
            st, ls = int(kind[b]), 0
        out.append(dict(begin=b, end=end, valid_bytes=e - b, start_state=st, at_line_start=ls, records_before=records_before,
                        index_shift=index_shift))
    return out


def partition_fasta_device(ctx, dptr, nbytes, n_parts, k):
    """partition_fasta for a buffer resident in HBM: the bookkeeping runs on the device (kmi_fasta_partition_dev: tile summaries of
    the line-kind machine, prefix sums, one short walk per block); same dicts as partition_fasta"""
    import ctypes as C
    from . import _lib as L
    be = np.zeros(2 * n_parts, dtype=np.uint64)
    parts = (L.FastaPartition * n_parts)()
    ctx.check(L.lib.kmi_fasta_partition_dev(ctx.h, C.c_void_p(dptr), nbytes, n_parts, k, be.ctypes.data_as(C.c_void_p), parts))
    return [dict(begin=int(be[2 * r]), end=int(be[2 * r + 1]), valid_bytes=int(parts[r].valid_bytes), start_state=int(parts[r].start_state),
                 at_line_start=int(parts[r].at_line_start), records_before=int(parts[r].records_before), index_shift=int(parts[r].index_shift))
            for r in range(n_parts)]
