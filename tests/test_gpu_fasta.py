"""GPU parity of the FASTA path: init_parser record rules, k-mer windows across line breaks,
LongSequenceKmerId, count / position index builds -- against the oracle and the reference's
TestFileInfo table (src/io/test/mpi_test_fasta_seq_parse.cpp:398-403)."""
import json
import os

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALPHA = {"DNA": orc.DNA, "DNA5": orc.DNA5}
FILES = ["test.fasta", "test2.fasta", "natural.fasta", "natural.withN.fasta", "test.unitiqs.fasta"]


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def test_reference_table_counts(ctx):
    import kmerind_amd as K
    pg = json.load(open(os.path.join(GOLD, "parse_golden.json")))["fasta"]
    cfg = K.make_config(pg["k"], "DNA5", seq_format="fasta")
    n = 0
    for e in pg["files"]:
        path = os.path.join(GOLD, "data", e["file"])
        if not os.path.exists(path):
            continue
        kmers, nseq = ctx.read_file(cfg, open(path, "rb").read())
        assert (nseq, kmers.shape[0]) == (e["records"], e["kmers"]), e
        n += 1
    assert n >= 4


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (35, "DNA5"), (63, "DNA5"), (63, "DNA"), (4, "DNA"), (1, "DNA5")])
def test_extract_matches_oracle_on_reference_files(ctx, k, alpha):
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha, seq_format="fasta", index_kind="position")
    for name in FILES:
        data = open(os.path.join(GOLD, "data", name), "rb").read()
        for off in (0, 1 << 35):
            ex = orc.extract(s, data, orc.FASTA, file_offset=off, want_ids=True)
            kmers, ids, nseq = ctx.read_file(cfg, data, file_offset=off, with_ids=True)
            assert nseq == ex["n_seqs"], name
            assert kmers.shape == ex["kmers"].shape, name
            assert (kmers == ex["kmers"]).all(), name
            assert (ids == ex["ids"]).all(), name


def _synthetic_fasta(rng, n_rec, line=80, eol=b"\n", orphan=False, max_len=3000):
    out = []
    if orphan:
        out.append(b"ACGTACGT" + eol + b"TTTT" + eol)
    for i in range(n_rec):
        out.append(b">chr%d some description" % i + eol)
        if i % 11 == 3:
            out.append(b";a comment line that belongs to the header group" + eol)
        ln = int(rng.integers(0, max_len)) if i % 5 else int(rng.integers(0, 70))
        seq = bytes(rng.choice(list(b"ACGTNacgt-"), size=ln, p=[.23, .23, .23, .23, .02, .015, .015, .015, .01, .005]).tolist())
        w = line if i % 3 else int(rng.integers(1, 30))
        for j in range(0, ln, w):
            out.append(seq[j:j + w] + eol)
        if i % 7 == 0:
            out.append(eol)              # blank line inside the sequence group
    return b"".join(out)


@pytest.mark.parametrize("eol,orphan", [(b"\n", False), (b"\r\n", False), (b"\n", True)])
@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (63, "DNA5"), (21, "DNA5")])
def test_extract_synthetic_multiline(ctx, k, alpha, eol, orphan):
    import kmerind_amd as K
    rng = np.random.default_rng(k + len(eol))
    data = _synthetic_fasta(rng, 120, eol=eol, orphan=orphan)
    s = orc.kspec(k, ALPHA[alpha])
    ex = orc.extract(s, data, orc.FASTA, want_ids=True)
    cfg = K.make_config(k, alpha, seq_format="fasta", index_kind="position")
    kmers, ids, nseq = ctx.read_file(cfg, data, with_ids=True)
    assert nseq == ex["n_seqs"] and kmers.shape == ex["kmers"].shape
    assert (kmers == ex["kmers"]).all()
    assert (ids == ex["ids"]).all()


def test_fasta_edge_cases(ctx):
    import kmerind_amd as K
    s = orc.kspec(5, orc.DNA)
    cfg = K.make_config(5, "DNA", seq_format="fasta", index_kind="position")
    cases = [b">only a header\n", b"no header at all\nACGTACGT\n", b">a\nACG\n>b\nACGTACGTAC\n>c\n\n>d\nAAAAAA",
             b">a\rACGTACGT\r>b\rTTTTTTTT\r",            # '\r' alone does not end a line
             b">a\n" + b"ACGTTGCA" * 3000 + b"\n",        # one long line, many tiles
             b">a\n" + b"A\n" * 5000]                     # one base per line
    for data in cases:
        ex = orc.extract(s, data, orc.FASTA, want_ids=True)
        kmers, ids, nseq = ctx.read_file(cfg, data, with_ids=True)
        assert nseq == ex["n_seqs"], data[:30]
        assert kmers.shape == ex["kmers"].shape, data[:30]
        assert (kmers == ex["kmers"]).all() and (ids == ex["ids"]).all(), data[:30]


def test_fasta_count_and_position_index(ctx):
    import kmerind_amd as K
    rng = np.random.default_rng(8)
    base = _synthetic_fasta(rng, 40, max_len=1500)
    data = base + base.replace(b">chr", b">dup")          # every sequence twice
    for k, alpha in ((31, "DNA"), (63, "DNA5")):
        s = orc.kspec(k, ALPHA[alpha])
        ex = orc.extract(s, data, orc.FASTA, want_ids=True)
        om = orc.CountMap(s, orc.CANONICAL)
        om.insert(ex["kmers"])
        idx = K.CountIndex(ctx, K.make_config(k, alpha, seq_format="fasta"))
        idx.build(data)
        gk, gc = idx.to_vector()
        ok, oc = om.export()
        a, b = orc.sorted_pairs(gk, gc), orc.sorted_pairs(ok, oc)
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
        idx.close()
        mm = orc.MultiMap(s, orc.CANONICAL)
        mm.insert(ex["kmers"], ex["ids"])
        pidx = K.PositionIndex(ctx, K.make_config(k, alpha, seq_format="fasta", index_kind="position"))
        pidx.build(data)
        pk, pv = pidx.to_vector()
        mk, mv = mm.export()
        assert (orc.sorted_rows(pk, pv) == orc.sorted_rows(mk, mv)).all()
        fk, fv = pidx.find(ex["kmers"][:200])
        ek, ev = mm.find(ex["kmers"][:200])
        assert (orc.sorted_rows(fk, fv) == orc.sorted_rows(ek, ev)).all()
        pidx.close()


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (63, "DNA5"), (5, "DNA")])
def test_fasta_split_over_ranks_equals_whole_file(ctx, k, alpha):
    """FASTA block partition with k - 1 overlap and the header bookkeeping of init_parser
    (fasta_loader.hpp:232-456,485-604; kmer_parser.hpp:112-157): every rank parses its own byte block, the
    concatenation of what the ranks produce (k-mers and LongSequenceKmerIds, rank order) is the whole-file result.
    Cuts fall wherever n * r / p lands: inside header lines, inside sequence lines, on EOLs, in orphan lines."""
    import kmerind_amd as K
    from kmerind_amd import fileio
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha, seq_format="fasta", index_kind="position")
    rng = np.random.default_rng(k)
    inputs = [open(os.path.join(GOLD, "data", name), "rb").read() for name in FILES]
    inputs.append(_synthetic_fasta(rng, 40, line=60, eol=b"\n", orphan=True))
    inputs.append(_synthetic_fasta(rng, 25, line=70, eol=b"\r\n", orphan=False))
    for data in inputs:
        whole = orc.extract(s, data, orc.FASTA, file_offset=1000, want_ids=True)
        buf = np.frombuffer(data, dtype=np.uint8)
        d = ctx.alloc(buf.size + 64)
        ctx.to_device(d, buf)
        for p in (2, 3, 5, 7, 8, 16):
            # the bookkeeping of every block computed ON THE DEVICE (kmi_fasta_partition_dev); the host-side helper must agree,
            # and the blocks' tuples must add up to the oracle's parse of the whole file (below)
            parts = fileio.partition_fasta_device(ctx, d, buf.size, p, k)
            assert parts == fileio.partition_fasta(data, p, k), (p, len(data))
            got_k, got_i = [], []
            for part in parts:
                block = data[part["begin"]:part["end"]]
                if not block:
                    continue
                ctx.set_fasta_partition(part)
                try:
                    km, ids, _ = ctx.read_file(cfg, block, file_offset=1000 + part["begin"], with_ids=True)
                finally:
                    ctx.set_fasta_partition(None)
                got_k.append(km); got_i.append(ids)
            gk = np.concatenate(got_k) if got_k else np.zeros((0, s.n_words), dtype=np.uint64)
            gi = np.concatenate(got_i) if got_i else np.zeros(0, dtype=np.uint64)
            assert gk.shape == whole["kmers"].shape, (p, len(data))
            assert (gk == whole["kmers"]).all(), (p, len(data))
            assert (gi == whole["ids"]).all(), (p, len(data))
        ctx.free(d)
    # whole-file behaviour is back once the partition is cleared
    km, ids, _ = ctx.read_file(cfg, inputs[0], file_offset=1000, with_ids=True)
    w0 = orc.extract(s, inputs[0], orc.FASTA, file_offset=1000, want_ids=True)
    assert (km == w0["kmers"]).all() and (ids == w0["ids"]).all()


@pytest.mark.parametrize("k,strand,seq_filter", [(31, "canonical", "all"), (25, "single", "all"), (19, "canonical", "all"), (31, "canonical", "n_split"),
                                                 (31, "canonical", "n_filter")])
def test_fasta_count_index_goes_through_super_kmers(ctx, k, strand, seq_filter):
    """A FASTA count index of one-word DNA k-mers (k >= 17) is built from super-k-mers cut out of the compacted character stream
    (fasta_runs -> sk_minimizer -> ... -> sk_reduce): those kernels must have run, and the index must be the oracle's -- multi-line
    records, short and empty ones, comment lines, lower case, N (an A inside a 2-bit k-mer; a cut under the sequence filters);
    a second build merges into the entries of the first; queries find what was built."""
    import kmerind_amd as K
    rng = np.random.default_rng(k)
    data = _synthetic_fasta(rng, 60, max_len=4000) + b">long\n" + bytes(rng.choice(list(b"ACGT"), size=30_000).tolist()) + b"\n"
    data2 = _synthetic_fasta(rng, 25, max_len=2500)
    s = orc.kspec(k, orc.DNA)
    filt = {"all": orc.SEQ_ALL, "n_split": orc.SEQ_N_SPLIT, "n_filter": orc.SEQ_N_FILTER}[seq_filter]
    om = orc.CountMap(s, orc.CANONICAL if strand == "canonical" else orc.SINGLE)
    idx = K.CountIndex(ctx, K.make_config(k, "DNA", strand=strand, seq_format="fasta", seq_filter=seq_filter))
    ctx.profile(True)
    ctx.profile_reset()
    idx.build(data)
    names = {p["name"] for p in ctx.profile_get() if p["launches"]}
    ctx.profile(False)
    assert {"fasta_runs", "sk_minimizer", "sk_scatter", "sk_reduce"} <= names and "fasta_extract" not in names, names
    ex = orc.extract(s, data, orc.FASTA, seq_filter=filt)["kmers"]
    om.insert(ex)
    a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*om.export())
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.build(data2)
    om.insert(orc.extract(s, data2, orc.FASTA, seq_filter=filt)["kmers"])
    a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*om.export())
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    q = np.concatenate([ex[::17], rng.integers(0, 1 << (2 * k - 1), size=(300, 1), dtype=np.uint64)])
    fa, fb = orc.sorted_pairs(*idx.find(q)), orc.sorted_pairs(*[x.astype(np.uint64) if x.dtype != np.uint64 else x for x in om.find(q)])
    assert fa[0].shape == fb[0].shape and (fa[0] == fb[0]).all() and (fa[1] == fb[1]).all()
    idx.close()


def test_fasta_with_more_runs_than_a_tile_has_slots_takes_the_kmer_path(ctx):
    """thousands of 40-base records: a tile of the compacted stream then holds more window runs than the run list has slots
    for, the super-k-mer front end says so and the build takes the k-mer pipeline -- same index"""
    import kmerind_amd as K
    k = 31
    rng = np.random.default_rng(3)
    recs = [b">r%d\n" % i + bytes(rng.choice(list(b"ACGT"), size=40).tolist()) + b"\n" for i in range(6000)]
    data = b"".join(recs)
    s = orc.kspec(k, orc.DNA)
    om = orc.CountMap(s, orc.CANONICAL)
    om.insert(orc.extract(s, data, orc.FASTA)["kmers"])
    idx = K.CountIndex(ctx, K.make_config(k, "DNA", seq_format="fasta"))
    ctx.profile(True)
    ctx.profile_reset()
    idx.build(data)
    names = {p["name"] for p in ctx.profile_get() if p["launches"]}
    ctx.profile(False)
    assert "fasta_runs" in names and "fasta_extract" in names and "sk_reduce" not in names, names
    a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*om.export())
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.close()


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (63, "DNA5")])
def test_fasta_block_of_one_rank_of_several(ctx, k, alpha):
    """read_file_* on one rank of several, FASTA (kmi_extract_fasta_block_host): every rank holds the whole file and parses block
    `rank` of an equal split, the bookkeeping computed on the device; the blocks in rank order are the oracle's parse of the file."""
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha, seq_format="fasta", index_kind="position")
    rng = np.random.default_rng(7 * k)
    inputs = [open(os.path.join(GOLD, "data", name), "rb").read() for name in FILES]
    inputs.append(_synthetic_fasta(rng, 30, line=60, eol=b"\n", orphan=True))
    for data in inputs:
        whole = orc.extract(s, data, orc.FASTA, file_offset=0, want_ids=True)
        for p in (1, 2, 3, 5):
            got = [ctx.read_file(cfg, data, with_ids=True, fasta_block=(r, p)) for r in range(p)]
            gk = np.concatenate([g[0] for g in got])
            gi = np.concatenate([g[1] for g in got])
            assert gk.shape == whole["kmers"].shape and (gk == whole["kmers"]).all() and (gi == whole["ids"]).all(), (p, len(data))
