"""GPU parity of the PositionIndex (multimap) path: KmerPositionTupleParser tuples -> insert ->
count / find / erase, against the oracle's unordered_multimap restatement."""
import os

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALPHA = {"DNA": orc.DNA, "DNA5": orc.DNA5}
STRAND = {"single": orc.SINGLE, "canonical": orc.CANONICAL, "bimolecule": orc.BIMOLECULE}


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _same(idx, om):
    gk, gv = idx.to_vector()
    ok, ov = om.export()
    assert gk.shape == ok.shape
    assert (orc.sorted_rows(gk, gv) == orc.sorted_rows(ok, ov)).all()


@pytest.mark.parametrize("k,alpha,strand", [(31, "DNA", "canonical"), (21, "DNA", "single"), (63, "DNA5", "canonical"),
                                            (63, "DNA", "bimolecule"), (15, "DNA5", "single")])
def test_position_index_build_insert_query(ctx, k, alpha, strand):
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha, strand=strand, index_kind="position")
    data = bytes(K.synth_fastq(seed=k, genome_len=3000, n_reads=1200))
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
    om = orc.MultiMap(s, STRAND[strand])
    om.insert(ex["kmers"], ex["ids"])
    idx = K.PositionIndex(ctx, cfg)
    idx.build(data)                                     # KmerPositionTupleParser + multimap insert on the device
    assert idx.local_size() == ex["kmers"].shape[0]
    _same(idx, om)
    # a second batch through insert(tuples) at another file offset
    data2 = bytes(K.synth_fastq(seed=k, genome_len=3000, n_reads=500, first_read=5000))
    ex2 = orc.extract(s, data2, orc.FASTQ, file_offset=1 << 33, want_ids=True)
    om.insert(ex2["kmers"], ex2["ids"])
    idx.insert(ex2["kmers"], ex2["ids"])
    _same(idx, om)
    # queries: present (with repeats), absent
    rng = np.random.default_rng(k)
    present = ex["kmers"][rng.integers(0, ex["kmers"].shape[0], size=3000)]
    absent = orc.extract(s, bytes(K.synth_fastq(seed=99, genome_len=50000, n_reads=30)), orc.FASTQ)["kmers"]
    q = np.concatenate([present, absent, present[:50]])
    gk, gc = idx.count(q)
    ok, oc = om.count(q)
    assert (orc.sorted_rows(gk, gc) == orc.sorted_rows(ok, oc)).all()
    fk, fv = idx.find(q)
    ek, ev = om.find(q)
    assert fk.shape == ek.shape and fk.shape[0] > q.shape[0]      # many positions per k-mer
    assert (orc.sorted_rows(fk, fv) == orc.sorted_rows(ek, ev)).all()
    n_gpu, n_cpu = idx.erase(present[:1000]), om.erase(present[:1000])
    assert n_gpu == n_cpu and idx.local_size() == om.size()
    _same(idx, om)
    idx.close()


def test_position_index_on_reference_files(ctx):
    import kmerind_amd as K
    s = orc.kspec(21, orc.DNA)
    cfg = K.make_config(21, "DNA", strand="canonical", index_kind="position")
    for name in ("test.small.fastq", "natural.fastq"):
        data = open(os.path.join(GOLD, "data", name), "rb").read()
        ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
        om = orc.MultiMap(s, orc.CANONICAL)
        om.insert(ex["kmers"], ex["ids"])
        idx = K.PositionIndex(ctx, cfg)
        idx.build(data)
        _same(idx, om)
        # every stored id points at the k-mer's text in the file (mpi_test_fastq_seq_parse.cpp:235-330)
        keys, ids = idx.to_vector()
        for key, i in list(zip(keys, ids[:, 0]))[:300]:
            pos = ((int(i) >> 16) & 0xFFFFFFFFFF) + (int(i) & 0xFFFF)
            fwd = orc.kmers_from_string(s, data[pos:pos + 21])
            assert orc.canonical(s, fwd)[0].tolist() == key.tolist()
        idx.close()


def test_position_index_api_guards(ctx):
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    cfg = K.make_config(31, "DNA", index_kind="position")
    idx = K.PositionIndex(ctx, cfg)
    with pytest.raises(L.KmiError):
        K.CountIndex.insert(idx, np.zeros((3, 1), dtype=np.uint64))      # key-only insert is for counting maps
    k, v = idx.find(np.array([[7]], dtype=np.uint64))
    assert k.shape[0] == 0
    k, c = idx.count(np.array([[7], [7]], dtype=np.uint64))
    assert k.shape[0] == 1 and c.tolist() == [0]
    idx.close()


def test_route_tuples_matches_key_to_rank(ctx):
    """imxx::distribute on (k-mer, id) records: grouped by KeyToRank, payload carried along"""
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    k, p = 31, 5
    s = orc.kspec(k, orc.DNA)
    cfg = K.make_config(k, "DNA", strand="canonical", index_kind="position")
    ex = orc.extract(s, bytes(K.synth_fastq(seed=2, genome_len=30000, n_reads=900)), orc.FASTQ, want_ids=True)
    rec = np.ascontiguousarray(np.concatenate([ex["kmers"], ex["ids"][:, None]], axis=1))
    n = rec.shape[0]
    din, dout = ctx.alloc(rec.nbytes), ctx.alloc(rec.nbytes)
    ctx.to_device(din, rec)
    counts = np.zeros(p, dtype=np.uint64)
    ctx.check(L.lib.kmi_route_tuples_dev(ctx.h, C.byref(cfg), C.c_void_p(din), n, p, 1, C.c_void_p(dout),
                                         counts.ctypes.data_as(C.c_void_p)))
    out = np.zeros_like(rec)
    ctx.to_host(out, dout)
    ctx.free(din); ctx.free(dout)
    canon = orc.canonical(s, ex["kmers"])
    ranks = orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, canon, p)
    assert counts.tolist() == np.bincount(ranks, minlength=p).tolist()
    off = 0
    for r in range(p):
        seg = out[off:off + int(counts[r])]
        exp = np.concatenate([canon[ranks == r], ex["ids"][ranks == r][:, None]], axis=1)
        assert (orc.sorted_rows(seg) == orc.sorted_rows(exp)).all()
        off += int(counts[r])


@pytest.mark.parametrize("kind", ["position", "posqual"])
def test_extract_route_records_matches_key_to_rank(ctx, kind):
    """kmi_extract_route_records_dev = kmi_extract_records_dev + kmi_route_tuples_dev: the (k-mer, id[, quality]) tuples of a
    FASTQ buffer at a file offset, canonical k-mers grouped by KeyToRank with their values carried along"""
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    k, p, off0 = 31, 3, 7_000_000
    s = orc.kspec(k, orc.DNA)
    cfg = K.make_config(k, "DNA", strand="canonical", index_kind=kind)
    data = bytes(K.synth_fastq(seed=12, genome_len=30000, n_reads=700))
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=(kind == "posqual"), file_offset=off0)
    cols = [ex["kmers"], ex["ids"][:, None]]
    if kind == "posqual":
        cols.append(ex["quals"].astype(np.float32).view(np.uint32).astype(np.uint64)[:, None])
    n, rw = ex["kmers"].shape[0], len(cols) + ex["kmers"].shape[1] - 1
    raw = np.frombuffer(data, dtype=np.uint8)
    din, dout = ctx.alloc(raw.nbytes + 64), ctx.alloc((n + 8) * rw * 8)
    ctx.to_device(din, raw)
    counts = np.zeros(p, dtype=np.uint64)
    nt, ns = C.c_uint64(0), C.c_uint64(0)
    ctx.check(L.lib.kmi_extract_route_records_dev(ctx.h, C.byref(cfg), C.c_void_p(din), raw.nbytes, off0, p, C.c_void_p(dout), n,
                                                  C.byref(nt), C.byref(ns), counts.ctypes.data_as(C.c_void_p)))
    assert nt.value == n and ns.value == 700
    out = np.zeros((n, rw), dtype=np.uint64)
    ctx.to_host(out, dout)
    with pytest.raises(L.KmiError):      # capacity
        ctx.check(L.lib.kmi_extract_route_records_dev(ctx.h, C.byref(cfg), C.c_void_p(din), raw.nbytes, off0, p, C.c_void_p(dout), n - 1,
                                                      C.byref(nt), C.byref(ns), counts.copy().ctypes.data_as(C.c_void_p)))
    ctx.free(din); ctx.free(dout)
    canon = orc.canonical(s, ex["kmers"])
    ranks = orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, canon, p)
    assert counts.tolist() == np.bincount(ranks, minlength=p).tolist()
    exp_all = np.concatenate([canon] + cols[1:], axis=1)
    off = 0
    for r in range(p):
        seg = out[off:off + int(counts[r])]
        assert (orc.sorted_rows(seg) == orc.sorted_rows(exp_all[ranks == r])).all()
        off += int(counts[r])
