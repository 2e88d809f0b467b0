"""The RCCL layer of the C ABI (kmi_comm_*, include/kmerind_hip.h) and the collectives built on it
(kmi_index_*_dist_host). One GPU is all a test box has and RCCL refuses two ranks on one device, so the
communicator here has ONE rank and KMI_FORCE_DIST=1 makes the entry points run their exchange anyway: keys are
routed on the device, go through grouped ncclSend / ncclRecv to the same rank and come back -- every RCCL call of
the multi-rank path executes, the first exchange with its checksums. The multi-rank logic (counts, offsets, order)
is covered on CPU by tests/test_dist_gloo.py; results are checked against the oracle's single map."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dist_ctx(monkeypatch):
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    monkeypatch.setenv("KMI_FORCE_DIST", "1")
    ctx = K.Context(0, rank=0, nranks=1)
    comm = C.c_void_p()
    ctx.check(L.lib.kmi_comm_create(ctx.h, None, C.byref(comm)))
    yield ctx, comm
    L.lib.kmi_comm_destroy(comm)
    ctx.close()


def test_unique_id_and_self_exchange(dist_ctx):
    from kmerind_amd import _lib as L
    ctx, comm = dist_ctx
    uid = (C.c_char * 128)()
    assert L.lib.kmi_comm_unique_id(uid) == L.OK and any(bytes(uid))
    n = 100_000
    a = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    d_a, d_b = ctx.alloc(a.nbytes), ctx.alloc(a.nbytes)
    ctx.to_device(d_a, a)
    sc, rc = np.array([n], np.uint64), np.zeros(1, np.uint64)
    ctx.check(L.lib.kmi_comm_all_to_all_counts(comm, sc.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p)))
    assert rc.tolist() == [n]
    for _ in range(2):                                 # the first exchange carries checksums, the second does not
        ctx.check(L.lib.kmi_comm_all_to_all_v(comm, C.c_void_p(d_a), sc.ctypes.data_as(C.c_void_p), C.c_void_p(d_b),
                                              rc.ctypes.data_as(C.c_void_p), 8))
        b = np.zeros_like(a)
        ctx.to_host(b, d_b)
        assert (a == b).all()
    v = C.c_uint64(41)
    ctx.check(L.lib.kmi_comm_allreduce_sum_u64(comm, C.byref(v)))
    assert v.value == 41
    ctx.free(d_a); ctx.free(d_b)


def test_peer_message_in_pieces(dist_ctx, monkeypatch):
    """a peer message above the piece size (1 GiB less a page in production: the RCCL build of this image was seen to corrupt larger
    ones) travels as several grouped send / receive rounds whose number every rank derives from the largest message anywhere.
    KMI_COMM_PIECE shrinks the piece to 4 KB, so an 800 KB message goes in 196 rounds -- with checksums, twice, and then through
    a whole build whose records travel the same way."""
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    ctx, comm = dist_ctx
    monkeypatch.setenv("KMI_COMM_PIECE", "4096")
    n = 100_003
    a = np.arange(n, dtype=np.uint64) * np.uint64(0xD6E8FEB86659FD93) + np.uint64(7)
    d_a, d_b = ctx.alloc(a.nbytes), ctx.alloc(a.nbytes)
    ctx.to_device(d_a, a)
    sc, rc = np.array([n], np.uint64), np.array([n], np.uint64)
    for _ in range(2):
        ctx.check(L.lib.kmi_comm_all_to_all_v(comm, C.c_void_p(d_a), sc.ctypes.data_as(C.c_void_p), C.c_void_p(d_b), rc.ctypes.data_as(C.c_void_p), 8))
        b = np.zeros_like(a)
        ctx.to_host(b, d_b)
        assert (a == b).all()
    ctx.free(d_a); ctx.free(d_b)
    s = orc.kspec(31, orc.DNA)
    data = K.synth_fastq(seed=5, genome_len=20_000, n_reads=2_000)
    idx = K.CountIndex(ctx, K.make_config(31, "DNA", strand="canonical"))
    ctx.check(L.lib.kmi_index_build_dist_host(idx.h, comm, data.ctypes.data_as(C.c_void_p), data.size, 0))
    om = orc.CountMap(s, orc.CANONICAL)
    om.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
    a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*om.export())
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.close()


@pytest.mark.parametrize("strand", ["canonical", "single"])
def test_count_index_collectives_over_rccl(dist_ctx, strand):
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    ctx, comm = dist_ctx
    k = 31
    s = orc.kspec(k, orc.DNA)
    st = {"canonical": orc.CANONICAL, "single": orc.SINGLE}[strand]
    data = K.synth_fastq(seed=21, genome_len=20_000, n_reads=2_000)
    ex = orc.extract(s, data, orc.FASTQ)["kmers"]
    om = orc.CountMap(s, st)
    idx = K.CountIndex(ctx, K.make_config(k, "DNA", strand=strand))
    # build: the fused build's super-k-mer records grouped by owner rank -> RCCL -> the back end (kmi_index_sk_produce / consume)
    ctx.profile(True)
    ctx.profile_reset()
    ctx.check(L.lib.kmi_index_build_dist_host(idx.h, comm, data.ctypes.data_as(C.c_void_p), data.size, 0))
    names = {p["name"] for p in ctx.profile_get() if p["launches"]}
    ctx.profile(False)
    assert {"sk_scatter", "sk_recv_scatter", "sk_reduce"} <= names, names
    om.insert(ex)
    # insert of k-mers: route -> RCCL -> insert
    extra = np.ascontiguousarray(ex[::7])
    ctx.check(L.lib.kmi_index_insert_dist_host(idx.h, comm, extra.ctypes.data_as(C.c_void_p), extra.shape[0]))
    om.insert(extra)
    keys, counts = idx.to_vector()
    ok, oc = om.export()
    a, b = orc.sorted_pairs(keys, counts), orc.sorted_pairs(ok, oc)
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    n = C.c_uint64()
    ctx.check(L.lib.kmi_index_size_dist(idx.h, comm, C.byref(n)))
    assert n.value == om.size()
    q = np.ascontiguousarray(np.concatenate([ex[:3000], np.random.default_rng(1).integers(0, 1 << 62, (500, 1), dtype=np.uint64)]))
    for fn, ofn in ((L.lib.kmi_index_count_dist_host, om.count), (L.lib.kmi_index_find_dist_host, om.find)):
        r = L.Results()
        ctx.check(fn(idx.h, comm, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        gk = np.ctypeslib.as_array(r.keys, shape=(r.n,)).copy().reshape(-1, 1) if r.n else np.zeros((0, 1), np.uint64)
        gv = np.ctypeslib.as_array(r.values, shape=(r.n,)).copy() if r.n else np.zeros(0, np.uint64)
        L.lib.kmi_results_free(C.byref(r))
        ek, ev = ofn(q)
        a, b = orc.sorted_pairs(gk, gv), orc.sorted_pairs(ek, np.asarray(ev).astype(np.uint64))
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    ne = C.c_uint64()
    ctx.check(L.lib.kmi_index_erase_dist_host(idx.h, comm, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(ne)))
    assert ne.value == om.erase(q) and idx.local_size() == om.size()
    idx.close()


def test_position_quality_index_collectives_over_rccl(dist_ctx):
    """config 5's index type: (k-mer, (id, quality)) records parsed on the device, routed, exchanged over RCCL, inserted;
    then tuples inserted through the tuple collective; find answers carry both value words"""
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    ctx, comm = dist_ctx
    k = 31
    s = orc.kspec(k, orc.DNA)
    data = K.synth_fastq(seed=22, genome_len=10_000, n_reads=800)
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=True)
    vals = np.ascontiguousarray(np.stack([ex["ids"], ex["quals"].view(np.uint32).astype(np.uint64)], axis=1))
    om = orc.MultiMap(s, orc.CANONICAL, vw=2)
    idx = K.PositionIndex(ctx, K.make_config(k, "DNA", strand="canonical", index_kind="posqual"))
    ctx.check(L.lib.kmi_index_build_dist_host(idx.h, comm, data.ctypes.data_as(C.c_void_p), data.size, 0))
    om.insert(ex["kmers"], vals)
    kk, vv = np.ascontiguousarray(ex["kmers"][:500]), np.ascontiguousarray(vals[:500] + np.uint64(1 << 40))
    ctx.check(L.lib.kmi_index_insert_tuples_dist_host(idx.h, comm, kk.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), 500))
    om.insert(kk, vv)
    gk, gv = idx.to_vector()
    ek, ev = om.export()
    assert (orc.sorted_rows(gk, gv) == orc.sorted_rows(ek, ev)).all()
    q = np.ascontiguousarray(ex["kmers"][::9])
    r = L.Results()
    ctx.check(L.lib.kmi_index_find_dist_host(idx.h, comm, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
    fk = np.ctypeslib.as_array(r.keys, shape=(r.n,)).copy().reshape(-1, 1)
    fv = np.ctypeslib.as_array(r.values, shape=(r.n * 2,)).copy().reshape(-1, 2)
    L.lib.kmi_results_free(C.byref(r))
    ek, ev = om.find(q)
    assert (orc.sorted_rows(fk, fv) == orc.sorted_rows(ek, ev)).all()
    idx.close()
