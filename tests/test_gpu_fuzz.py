"""Differential fuzz of the fused FASTQ count build and the extract path against the oracle: ragged read lengths (reads shorter
than k, single bases, reads longer than a scan tile), LF / CRLF line ends, lower case and non-ACGT characters, '+' lines that
repeat the name, missing final newline -- the record shapes FASTQParser::get_next_record accepts (fastq_loader.hpp:389-467)."""
import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _random_fastq(seed):
    rng = np.random.default_rng(seed)
    eol = b"\r\n" if seed % 3 == 1 else b"\n"
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTNacgtnRY", dtype=np.uint8)
    recs = []
    n = int(rng.integers(1, 400))
    for i in range(n):
        kind = rng.random()
        length = int(rng.integers(1, 12)) if kind < 0.1 else (int(rng.integers(9000, 20000)) if kind > 0.985 else int(rng.integers(12, 400)))
        seq = alphabet[rng.integers(0, alphabet.size, size=length)].tobytes()
        qual = rng.integers(33, 74, size=length, dtype=np.uint8).tobytes()
        name = b"r%d len=%d" % (i, length)
        plus = b"+" + (name if rng.random() < 0.3 else b"")
        recs.append(b"@" + name + eol + seq + eol + plus + eol + qual + eol)
    data = b"".join(recs)
    if seed % 4 == 2:
        data = data[:-len(eol)]                     # no newline at the end of the file
    return data


@pytest.mark.parametrize("seed", list(range(16)))
def test_fused_build_and_extract_follow_the_oracle_on_ragged_fastq(ctx, seed):
    import kmerind_amd as K
    data = _random_fastq(seed)
    k, alpha, oa = [(31, "DNA", orc.DNA), (15, "DNA", orc.DNA), (21, "DNA5", orc.DNA5), (33, "DNA", orc.DNA), (12, "DNA16", orc.DNA16)][seed % 5]
    flt = ["all", "n_split", "n_filter"][seed % 3]
    fl = {"all": orc.SEQ_ALL, "n_split": orc.SEQ_N_SPLIT, "n_filter": orc.SEQ_N_FILTER}[flt]
    s = orc.kspec(k, oa)
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, seq_filter=fl)
    gk, gi, gn = ctx.read_file(K.make_config(k, alpha, strand="single", index_kind="position", seq_filter=flt), data, with_ids=True)
    assert gn == ex["n_seqs"] and gk.shape == ex["kmers"].shape and (gk == ex["kmers"]).all() and (gi == ex["ids"]).all()
    idx = K.CountIndex(ctx, K.make_config(k, alpha, strand="canonical", seq_filter=flt))
    idx.build(data)
    m = orc.CountMap(s, orc.CANONICAL)
    m.insert(ex["kmers"])
    a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*m.export())
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.close()


def _random_fasta(seed):
    rng = np.random.default_rng(seed)
    eol = b"\r\n" if seed % 3 == 1 else b"\n"
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTNacgtnRY-", dtype=np.uint8)
    out = []
    if seed % 5 == 0:                                   # sequence lines before the first header
        out.append(alphabet[rng.integers(0, alphabet.size, size=int(rng.integers(1, 90)))].tobytes() + eol)
    for r in range(int(rng.integers(1, 60))):
        for h in range(int(rng.integers(1, 3))):       # one or two header / comment lines
            out.append((b">" if rng.random() < 0.8 else b";") + b"rec%d.%d some text" % (r, h) + eol)
        total = int(rng.integers(0, 3000)) if rng.random() < 0.9 else int(rng.integers(9000, 30000))
        width = int(rng.integers(1, 120))
        seq = alphabet[rng.integers(0, alphabet.size, size=total)].tobytes()
        for i in range(0, total, width):
            out.append(seq[i:i + width] + eol)
    data = b"".join(out)
    if seed % 4 == 2 and data.endswith(eol):
        data = data[:-len(eol)]
    return data


@pytest.mark.parametrize("seed", list(range(12)))
def test_fasta_extract_and_build_follow_the_oracle_on_ragged_fasta(ctx, seed):
    """multi-line records of any width, records without sequence, '>' and ';' header lines, lines before the first header,
    CRLF, no final newline (fasta_loader.hpp:485-723), with and without the N filters"""
    import kmerind_amd as K
    data = _random_fasta(seed)
    k, alpha, oa = [(31, "DNA", orc.DNA), (15, "DNA", orc.DNA), (21, "DNA5", orc.DNA5), (64, "DNA", orc.DNA), (12, "DNA16", orc.DNA16)][seed % 5]
    flt = ["all", "n_split", "n_filter"][seed % 3]
    fl = {"all": orc.SEQ_ALL, "n_split": orc.SEQ_N_SPLIT, "n_filter": orc.SEQ_N_FILTER}[flt]
    s = orc.kspec(k, oa)
    ex = orc.extract(s, data, orc.FASTA, want_ids=True, seq_filter=fl)
    gk, gi, gn = ctx.read_file(K.make_config(k, alpha, strand="single", index_kind="position", seq_format="fasta", seq_filter=flt), data,
                               with_ids=True)
    assert gk.shape == ex["kmers"].shape and (gk == ex["kmers"]).all() and (gi == ex["ids"]).all()
    if flt != "n_split":                                 # (pieces are not counted for FASTA: kmerind_hip.h, seq_filter)
        assert gn == ex["n_seqs"]
    idx = K.CountIndex(ctx, K.make_config(k, alpha, strand="canonical", seq_format="fasta", seq_filter=flt))
    idx.build(data)
    m = orc.CountMap(s, orc.CANONICAL)
    m.insert(ex["kmers"])
    a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*m.export())
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.close()


@pytest.fixture(scope="module")
def small_range_ctx():
    """a context whose one-pass front end cuts its input into byte ranges of 2 KB (KMI_FRONT_MIN_RANGE is read when the context is
    made): range boundaries then fall everywhere -- inside headers, sequence lines, '+' lines, quality lines, between CR and LF"""
    import os
    import kmerind_amd as K
    old = os.environ.get("KMI_FRONT_MIN_RANGE")
    os.environ["KMI_FRONT_MIN_RANGE"] = "2048"
    c = K.Context(0)
    if old is None:
        del os.environ["KMI_FRONT_MIN_RANGE"]
    else:
        os.environ["KMI_FRONT_MIN_RANGE"] = old
    yield c
    c.close()


@pytest.mark.parametrize("k", [17, 21, 25, 31])
@pytest.mark.parametrize("kind", ["fixed", "ragged", "crlf", "blank", "lowerN", "noeol"])
def test_one_pass_front_end_fuzz_with_small_ranges(small_range_ctx, kind, k):
    """kmi_front.h (the whole FASTQ front end in one pass, every wavefront on a byte range of its own with an INFERRED line index)
    against the oracle: 32 seeds per (text kind, k), ranges of 2 KB, both strand models, 200 - 4000 reads. Whatever the path decides
    -- take the input or hand it to the general front end -- the map must be the oracle's; over the seeds of a case both must happen
    for the kinds that allow it (this was tools/exp/front_fuzz.py, an untracked script, in round 3)."""
    import kmerind_amd as K
    from tests.test_gpu_index import _fastq_variant, _same_map, STRAND
    ctx = small_range_ctx
    took = 0
    for seed in range(32):
        rng = np.random.default_rng(100_000 * k + 1000 * ["fixed", "ragged", "crlf", "blank", "lowerN", "noeol"].index(kind) + seed)
        strand = "canonical" if seed % 2 == 0 else "single"
        data = _fastq_variant(rng, int(rng.integers(200, 4000)), kind)
        s = orc.kspec(k, orc.DNA)
        idx = K.CountIndex(ctx, K.make_config(k, "DNA", strand=strand))
        ctx.profile(True)
        ctx.profile_reset()
        idx.build(data)
        names = {p["name"] for p in ctx.profile_get() if p["launches"]}
        ctx.profile(False)
        took += "sk_front" in names and "fastq_scan_tiles" not in names
        om = orc.CountMap(s, STRAND[strand])
        om.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
        _same_map(idx, om)
        idx.close()
    assert took > 0 or kind in ("blank",), (kind, k, took)      # the one-pass path really ran (blank lines may always send it away)


@pytest.mark.parametrize("seed", list(range(12)))
def test_position_builds_from_the_parse_follow_the_oracle_on_ragged_fastq(ctx, seed):
    """kmi_tuples.h: the position and position + quality builds partition their tuples straight from the packed input (histogram from
    the entry list, read descriptors from the list pass, ids and quality values made inside the scatter pass). Ragged FASTQ -- reads
    shorter than k, reads longer than a scan tile, CRLF, no final newline -- against the oracle's multimap: k-mers, ShortSequenceKmerId
    at a file offset, quality floats bit for bit; a second build lands in the existing entries."""
    import kmerind_amd as K
    data = _random_fastq(100 + seed)
    k = [31, 21, 15, 28][seed % 4]
    kind = "posqual" if seed % 2 else "position"
    vw = 2 if kind == "posqual" else 1
    off = 0 if seed % 3 else 123_456_789
    s = orc.kspec(k, orc.DNA)
    ex = orc.extract(s, data, orc.FASTQ, file_offset=off, want_ids=True, want_quals=(vw == 2))
    vals = ex["ids"].reshape(-1, 1)
    if vw == 2:
        vals = np.concatenate([vals, ex["quals"].view(np.uint32).astype(np.uint64).reshape(-1, 1)], axis=1)
    ref = orc.MultiMap(s, orc.CANONICAL, vw)
    ref.insert(ex["kmers"], vals)
    idx = K.PositionIndex(ctx, K.make_config(k, "DNA", strand="canonical", index_kind=kind))
    idx.build(data, file_offset=off)

    def canon(keys, v):
        rows = np.concatenate([np.asarray(keys).reshape(len(keys), -1), np.asarray(v).reshape(len(keys), -1)], axis=1)
        return rows[np.lexsort([rows[:, c] for c in range(rows.shape[1] - 1, -1, -1)])] if len(rows) else rows

    gk, gv = idx.to_vector()
    rk, rv = ref.export()
    assert canon(gk, gv[:, :vw] if vw > 1 else gv).shape == canon(rk, rv).shape and (canon(gk, gv[:, :vw] if vw > 1 else gv) == canon(rk, rv)).all()
    idx.build(data, file_offset=off)                               # the same tuples once more: every entry twice
    ref.insert(ex["kmers"], vals)
    gk, gv = idx.to_vector()
    rk, rv = ref.export()
    assert (canon(gk, gv[:, :vw] if vw > 1 else gv) == canon(rk, rv)).all()
    idx.close()
