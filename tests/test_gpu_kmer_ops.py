"""GPU parity (through the C ABI) of the array-level k-mer ops and of FASTQ extraction
against the CPU oracle and the committed golden fixtures."""
import json
import os

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SHAPES = [(31, "DNA"), (21, "DNA"), (32, "DNA"), (1, "DNA"), (63, "DNA"), (96, "DNA"),
          (21, "DNA5"), (35, "DNA5"), (63, "DNA5"), (13, "DNA5")]
ALPHA = {"DNA": orc.DNA, "DNA5": orc.DNA5}


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _random_kmers(s, n, seed):
    rng = np.random.default_rng(seed)
    km = rng.integers(0, 1 << 63, size=(n, s.n_words), dtype=np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=(n, s.n_words), dtype=np.uint64)
    pad = s.n_words * 64 - s.n_bits
    km[:, -1] &= np.uint64((1 << (64 - pad)) - 1)
    return km


@pytest.mark.parametrize("k,alpha", SHAPES)
def test_revcomp_canonical_hash_rank(ctx, k, alpha):
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    km = _random_kmers(s, 5000, k)
    for strand in ("canonical", "bimolecule"):
        cfg = K.make_config(k, alpha, strand=strand)
        assert (ctx.revcomp(cfg, km) == orc.revcomp(s, km)).all()
        assert (ctx.canonical(cfg, km) == orc.canonical(s, km)).all()
        for which, w in (("murmur", orc.MURMUR), ("farm", orc.FARM), ("identity", orc.IDENTITY), ("std", orc.STD)):
            for prefix in (True, False):
                assert (ctx.hash(cfg, which, prefix, km) == orc.kmer_hash(s, w, prefix, km)).all(), (which, prefix)
        for p in (1, 2, 3, 8):
            st = orc.BIMOLECULE if strand == "bimolecule" else orc.CANONICAL
            assert (ctx.key_to_rank(cfg, km, p) == orc.key_to_rank(s, orc.MURMUR, st, km, p)).all()
    for dh, w in (("identity", orc.IDENTITY), ("std", orc.STD)):   # KeyToRank hands ceilLog2(p) prefix bits to these
        cfgh = K.make_config(k, alpha, dist_hash=dh)
        for p in (2, 3, 8, 200):
            assert (ctx.key_to_rank(cfgh, km, p) == orc.key_to_rank(s, w, orc.CANONICAL, km, p)).all(), (dh, p)
    cfgf = K.make_config(k, alpha, dist_hash="farm", farm_ndebug=True)
    orc.lib.orc_set_farm_ndebug(1)
    try:
        assert (ctx.hash(cfgf, "farm", True, km) == orc.kmer_hash(s, orc.FARM, True, km)).all()
        assert (ctx.key_to_rank(cfgf, km, 5) == orc.key_to_rank(s, orc.FARM, orc.CANONICAL, km, 5)).all()
    finally:
        orc.lib.orc_set_farm_ndebug(0)


def test_known_answers_on_device(ctx):
    import kmerind_amd as K
    ka = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))["k31_dna"]
    cfg = K.make_config(31, "DNA")
    km = np.array([[int(ka["first_kmer"], 16)]], dtype=np.uint64)
    assert int(ctx.revcomp(cfg, km)[0, 0]) == int(ka["revcomp"], 16)
    assert int(ctx.canonical(cfg, km)[0, 0]) == int(ka["revcomp"], 16)
    assert int(ctx.hash(cfg, "murmur", True, km)[0]) == int(ka["murmur_prefix"], 16)
    assert int(ctx.hash(cfg, "murmur", False, km)[0]) == int(ka["murmur_store"], 16)
    assert int(ctx.hash(cfg, "farm", True, km)[0]) == int(ka["farm_prefix"], 16)
    assert int(ctx.hash(cfg, "farm", False, km)[0]) == int(ka["farm_store"], 16)


FASTQ_FILES = ["test.small.fastq", "test.medium.fastq", "natural.fastq", "natural.withN.fastq",
               "test.debruijn.tiny.fastq", "test.unitiq1.fastq", "test.unitiqs.fastq", "test.unitiq1.short2.fastq"]


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (21, "DNA"), (35, "DNA5"), (63, "DNA"), (63, "DNA5"), (15, "DNA5")])
def test_extract_golden_files(ctx, k, alpha):
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha)
    for name in FASTQ_FILES:
        data = open(os.path.join(GOLD, "data", name), "rb").read()
        ex = orc.extract(s, data, orc.FASTQ)
        kmers, nseq = ctx.read_file(cfg, data)
        assert nseq == ex["n_seqs"], name
        assert kmers.shape == ex["kmers"].shape, name
        assert (kmers == ex["kmers"]).all(), name


def test_extract_reference_table_counts(ctx):
    """TestFileInfo table of mpi_test_fastq_seq_parse.cpp:446-459 (K=35, DNA5) on the device"""
    import kmerind_amd as K
    pg = json.load(open(os.path.join(GOLD, "parse_golden.json")))["fastq"]
    cfg = K.make_config(pg["k"], "DNA5")
    for e in pg["files"]:
        path = os.path.join(GOLD, "data", e["file"])
        if not os.path.exists(path):
            continue
        kmers, nseq = ctx.read_file(cfg, open(path, "rb").read())
        assert (nseq, kmers.shape[0]) == (e["records"], e["kmers"]), e


def _ragged_fastq(rng, n_reads, eol=b"\n", max_len=400):
    recs = []
    for i in range(n_reads):
        ln = int(rng.integers(0, max_len)) if i % 7 else int(rng.integers(0, 40))
        seq = bytes(rng.choice(list(b"ACGTNacgtn"), size=ln, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]).tolist()) if ln else b""
        qual = bytes(rng.integers(33, 74, size=ln, dtype=np.uint8).tolist())
        if ln == 0:
            continue  # an empty sequence line would collapse the 4-line structure
        extra = eol * int(rng.integers(1, 3))
        recs.append(b"@r%d" % i + extra + seq + eol + b"+" + eol + qual + extra)
    return b"".join(recs)


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (63, "DNA5"), (5, "DNA"), (32, "DNA")])
def test_extract_ragged_reads(ctx, k, alpha, eol):
    import kmerind_amd as K
    rng = np.random.default_rng(k)
    data = _ragged_fastq(rng, 900, eol)
    s = orc.kspec(k, ALPHA[alpha])
    ex = orc.extract(s, data, orc.FASTQ)
    kmers, nseq = ctx.read_file(K.make_config(k, alpha), data)
    assert nseq == ex["n_seqs"]
    assert kmers.shape == ex["kmers"].shape
    assert (kmers == ex["kmers"]).all()


def test_extract_edge_cases(ctx):
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    cfg = K.make_config(31, "DNA")
    kmers, nseq = ctx.read_file(cfg, b"")
    assert kmers.shape[0] == 0 and nseq == 0
    # reads shorter than k emit nothing but still count as sequences
    data = b"@a\nACGT\n+\nIIII\n@b\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 40 + b"\n"
    s = orc.kspec(31, orc.DNA)
    ex = orc.extract(s, data, orc.FASTQ)
    kmers, nseq = ctx.read_file(cfg, data)
    assert nseq == ex["n_seqs"] == 2 and kmers.shape[0] == ex["kmers"].shape[0] == 10
    assert (kmers == ex["kmers"]).all()
    # truncated last record (no quality line): the reference still emits its k-mers
    data = b"@a\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 40 + b"\n@b\n" + b"TTGCA" * 9 + b"\n"
    ex = orc.extract(s, data, orc.FASTQ)
    kmers, nseq = ctx.read_file(cfg, data)
    assert (kmers == ex["kmers"]).all() and nseq == ex["n_seqs"]
    # malformed: header without '@' / third line without '+'
    for bad in (b"a\nACGT\n+\nIIII\n", b"@a\nACGT\n-\nIIII\n", b"\n@a\nACGT\n+\nIIII\n"):
        with pytest.raises(ValueError):
            orc.extract(s, bad, orc.FASTQ)
        with pytest.raises(L.KmiError) as ei:
            ctx.read_file(cfg, bad)
        assert ei.value.status == L.ERR_PARSE


def test_fastq_seq_and_qual_must_have_equal_length(ctx):
    """fastq_loader.hpp:454-463: both lines present but of different length -> the parser throws; a record
    without a quality line (truncated input) only warns. Checked wherever the bad record sits (tile interior,
    tile seams, first / last record) and for CRLF line ends."""
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    cfg = K.make_config(21, "DNA")
    s = orc.kspec(21, orc.DNA)
    good = bytes(np.asarray(K.synth_fastq(seed=5, genome_len=50_000, n_reads=400)).tobytes())   # 315-byte records, 126 KB = 15+ tiles
    recs = [good[i:i + 315] for i in range(0, len(good), 315)]
    kmers, nseq = ctx.read_file(cfg, good)
    assert nseq == 400
    for victim in (0, 1, 25, 26, 51, 52, 77, 200, 398, 399):              # 26 records = 8190 bytes: seams nearby
        for delta in (-1, +1, -100, +3):
            r = recs[victim]
            lines = r.split(b"\n")
            q = lines[3]
            lines[3] = q[:delta] if delta < 0 else q + b"I" * delta
            bad = b"".join(recs[:victim]) + b"\n".join(lines) + b"".join(recs[victim + 1:])
            with pytest.raises(ValueError):
                orc.extract(s, bad, orc.FASTQ)
            with pytest.raises(L.KmiError) as ei:
                ctx.read_file(cfg, bad)
            assert ei.value.status == L.ERR_PARSE, (victim, delta)
            idx = K.CountIndex(ctx, cfg)
            with pytest.raises(L.KmiError):
                idx.build(bad)
            idx.close()
    crlf = good.replace(b"\n", b"\r\n")
    kmers2, nseq2 = ctx.read_file(cfg, crlf)
    assert nseq2 == 400 and (kmers2 == kmers).all()
    bad = crlf[:630 + 170] + b"I" + crlf[630 + 170:]                       # one extra byte inside record 2's lines
    try:
        orc.extract(s, bad, orc.FASTQ)
        oracle_ok = True
    except ValueError:
        oracle_ok = False
    try:
        ctx.read_file(cfg, bad)
        dev_ok = True
    except L.KmiError:
        dev_ok = False
    assert oracle_ok == dev_ok
    # very short reads: thousands of lines per tile (the crowded-window path of the check)
    tiny = b"".join(b"@r\n" + bytes([b"ACGT"[i % 4]]) * (1 + i % 3) + b"\n+\n" + b"I" * (1 + i % 3) + b"\n" for i in range(6000))
    c3 = K.make_config(3, "DNA")
    s3 = orc.kspec(3, orc.DNA)
    ex = orc.extract(s3, tiny, orc.FASTQ)
    km, ns = ctx.read_file(c3, tiny)
    assert ns == ex["n_seqs"] == 6000 and (km == ex["kmers"]).all()
    for at in (5, 3000, 5999):
        recs_t = tiny.split(b"@r\n")[1:]
        recs_t[at] = recs_t[at][:-1] + b"I\n"          # quality one longer than the sequence
        broken = b"".join(b"@r\n" + r for r in recs_t)
        with pytest.raises(ValueError):
            orc.extract(s3, broken, orc.FASTQ)
        with pytest.raises(L.KmiError):
            ctx.read_file(c3, broken)
    # quality line missing altogether at the end: warning only
    cut = good[: len(good) - 151]
    ex = orc.extract(s, cut, orc.FASTQ)
    kmers3, nseq3 = ctx.read_file(cfg, cut)
    assert nseq3 == ex["n_seqs"] and (kmers3 == ex["kmers"]).all()


def test_extract_synthetic_reads_multi_tile(ctx):
    import kmerind_amd as K
    data = K.synth_fastq(seed=2, genome_len=200_000, n_reads=3000)
    for k, alpha in ((31, "DNA"), (63, "DNA5")):
        s = orc.kspec(k, ALPHA[alpha])
        ex = orc.extract(s, data, orc.FASTQ)
        kmers, nseq = ctx.read_file(K.make_config(k, alpha), data)
        assert nseq == 3000 and kmers.shape[0] == 3000 * (150 - k + 1)
        assert (kmers == ex["kmers"]).all()


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (21, "DNA"), (63, "DNA5"), (3, "DNA")])
def test_extract_position_ids(ctx, k, alpha):
    """KmerPositionTupleParser on FASTQ: ShortSequenceKmerId of every tuple (kmer_parser.hpp:303-569)"""
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha, index_kind="position")
    rng = np.random.default_rng(k)
    inputs = [open(os.path.join(GOLD, "data", n), "rb").read() for n in ("test.small.fastq", "natural.fastq", "test.unitiq1.fastq")]
    inputs.append(_ragged_fastq(rng, 700, b"\n"))
    inputs.append(_ragged_fastq(rng, 300, b"\r\n", max_len=40))
    inputs.append(bytes(K.synth_fastq(seed=4, genome_len=100_000, n_reads=2500)))
    for data in inputs:
        for off in (0, 123_456_789_012):
            ex = orc.extract(s, data, orc.FASTQ, file_offset=off, want_ids=True)
            kmers, ids, nseq = ctx.read_file(cfg, data, file_offset=off, with_ids=True)
            assert nseq == ex["n_seqs"] and kmers.shape == ex["kmers"].shape
            assert (kmers == ex["kmers"]).all()
            assert (ids == ex["ids"]).all()


def test_position_id_overflow_is_reported(ctx):
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    cfg = K.make_config(21, "DNA", index_kind="position")
    long_read = b"@r\n" + b"ACGT" * 20000 + b"\n+\n" + b"I" * 80000 + b"\n"
    with pytest.raises(L.KmiError) as ei:
        ctx.read_file(cfg, long_read, with_ids=True)
    assert ei.value.status == L.ERR_OVERFLOW


def test_fastq_partition_on_the_device_matches_the_host_rule(ctx):
    """kmi_fastq_partition_dev = FASTQParser::find_first_record per split point (fastq_loader.hpp:269-364): same record-aligned
    ranges as the oracle's restatement of that rule (orc_fastq_align) and as kmerind_amd.fileio.partition_fastq, also when quality lines begin with '@' or '+' and for more parts than records"""
    import kmerind_amd as K
    from kmerind_amd import fileio
    files = [open(os.path.join(GOLD, "data", n), "rb").read() for n in ("natural.fastq", "test.small.fastq", "test.medium.fastq")]
    tricky = b"".join(b"@r%d\nACGTACGTAC\n+\n%s\n" % (i, q) for i, q in enumerate([b"@@@@@@@@@@", b"+IIIIIIIII", b"@+@+@+@+@+", b"IIIIIIIIII"] * 40))
    files.append(tricky)
    files.append(bytes(K.synth_fastq(seed=3, genome_len=5000, n_reads=300)))
    for data in files:
        buf = np.frombuffer(data, dtype=np.uint8)
        d = ctx.alloc(buf.size + 64)
        ctx.to_device(d, buf)
        for parts in (1, 2, 3, 5, 8, 16, 257):
            got = fileio.partition_fastq_device(ctx, d, buf.size, parts)
            # the oracle's restatement of the rule (oracle/kmerind_oracle.c fastq_align), split point by split point
            cuts = [orc.fastq_align(data, buf.size * r // parts) for r in range(parts)] + [buf.size]
            for r in range(1, parts + 1):
                cuts[r] = max(cuts[r], cuts[r - 1])
            assert got == [(cuts[r], cuts[r + 1]) for r in range(parts)], parts
            assert got == fileio.partition_fastq(data, parts), parts          # (and the host-side helper agrees)
        ctx.free(d)


def test_parts_of_a_device_buffer_parse_like_the_whole(ctx):
    """a record-aligned part of a larger device buffer starts at an arbitrary address (the byte kernels load 16 bytes per
    lane: the library aligns such an input itself): the parts' tuples add up to the whole buffer's, ids included"""
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    from kmerind_amd import fileio
    k = 31
    s = orc.kspec(k, orc.DNA)
    data = K.synth_fastq(seed=8, genome_len=4000, n_reads=1000)          # 315-byte records: odd part offsets
    cfg = K.make_config(k, "DNA", strand="canonical", index_kind="position")
    d = ctx.alloc(data.size + 64)
    ctx.to_device(d, data)
    ex = orc.extract(s, bytes(data), orc.FASTQ, want_ids=True)
    got_k, got_i = [], []
    for b, e in fileio.partition_fastq_device(ctx, d, data.size, 7):
        assert b % 315 == 0
        nt, ns = C.c_uint64(), C.c_uint64()
        ctx.check(L.lib.kmi_extract_count_dev(ctx.h, C.byref(cfg), C.c_void_p(d + b), e - b, C.byref(nt), C.byref(ns)))
        assert nt.value == (e - b) // 315 * (150 - k + 1) and ns.value == (e - b) // 315
        dk, di = ctx.alloc(nt.value * 8 + 64), ctx.alloc(nt.value * 8 + 64)
        ctx.check(L.lib.kmi_extract_dev(ctx.h, C.byref(cfg), C.c_void_p(d + b), e - b, b, C.c_void_p(dk), C.c_void_p(di), nt.value, C.byref(nt), C.byref(ns)))
        hk, hi = np.zeros((nt.value, 1), np.uint64), np.zeros(nt.value, np.uint64)
        ctx.to_host(hk, dk); ctx.to_host(hi, di)
        ctx.free(dk); ctx.free(di)
        got_k.append(hk); got_i.append(hi)
    ctx.free(d)
    assert (np.concatenate(got_k) == ex["kmers"]).all() and (np.concatenate(got_i) == ex["ids"]).all()
