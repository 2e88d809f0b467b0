"""The library's OWN multi-rank code with more than one rank: kmi_index_build_dist_dev (the chunked, overlapped build through
exchanged super-k-mer records), kmi_index_{insert, insert_pairs, count, find, erase, size}_dist*, the position index
and the de Bruijn node map over ranks -- everything in C above kmi_comm. RCCL refuses two ranks on one device and a test box has
one GPU, so the communicator here is kmi_comm_create_transport over a gloo group (kmerind_amd/transport.py): two and four
processes share the GPU, the library stages its device buffers through pinned memory around the two callbacks, and every line
of the per-peer logic (offsets, counts with riders, the verdict of a chunk, the receive pool, checksums of a PEER's message) runs
with r != 0. What the reference does here: imxx::distribute, incremental_mxx.hpp:1039-1109, under every collective of
distributed_unordered_map.hpp (:880-983 count, :564-687 find, :719-779 erase, :1697-1745 insert). Checked against the oracle's
single map: the union of the ranks' maps must be that map whatever the rank count."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import oracle as orc

pytestmark = pytest.mark.gpu
STRAND = {"canonical": orc.CANONICAL, "single": orc.SINGLE}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _results(L, r, vw=1):
    n = r.n
    k = np.ctypeslib.as_array(r.keys, shape=(n,)).copy().reshape(-1, 1) if n else np.zeros((0, 1), np.uint64)
    v = np.ctypeslib.as_array(r.values, shape=(n * vw,)).copy().reshape(n, vw) if n else np.zeros((0, vw), np.uint64)
    L.lib.kmi_results_free(C.byref(r))
    return k, (v[:, 0] if vw == 1 else v)


def _shares(data, world, case):
    """record-aligned byte ranges of the ranks; the cases skew them"""
    from kmerind_amd import fileio
    if case == "empty-rank0":                      # rank 0 holds nothing and still enters every collective
        parts = fileio.partition_fastq(data, world - 1) if world > 1 else [(0, len(data))]
        return [(0, 0)] + list(parts)
    if case == "regrow":                           # nearly everything on the last rank: the receive pools' estimate is far off
        cut = fileio.find_first_record(data, len(data) // 20)
        small = fileio.partition_fastq(data[:cut], world - 1)
        return list(small) + [(cut, len(data))]
    return fileio.partition_fastq(data, world)


def _count_worker(rank, world, port, data, k, strand, case, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), KMI_DIST_CHUNKS="3")
    if case == "fallback-rank1" and rank == 1:
        os.environ["KMI_SK_DBG"] = "7"             # rank 1's front end declines every chunk: all ranks send those as k-mers
    if case == "error-rank1":
        os.environ["KMI_SK_DBG"] = "9"             # rank 1 fails (as a parse error would) on its second chunk
    if case == "regrow":
        os.environ.update(KMI_DIST_POOL_SLACK="64", KMI_DIST_POOL_PCT="10")   # a pool for a tenth of what will arrive
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmerind_amd as K
        from kmerind_amd import _lib as L
        from kmerind_amd.transport import GroupComm
        ctx = K.Context(0, rank=rank, nranks=world)
        comm = GroupComm(ctx)
        idx = K.CountIndex(ctx, K.make_config(k, "DNA", strand=strand))
        b, e = _shares(data, world, case)[rank]
        buf = np.frombuffer(data[b:e], dtype=np.uint8).copy()
        d = ctx.alloc(max(buf.size, 16) + 64)
        if buf.size:
            ctx.to_device(d, buf)
        st = L.lib.kmi_index_build_dist_dev(idx.h, comm.h, C.c_void_p(d), buf.size, b)
        if case == "error-rank1":
            # every rank comes back (nobody waits in an exchange for rank 1), with an error, and the index is untouched
            ret[rank] = (st, (L.lib.kmi_last_error(ctx.h) or b"").decode(), idx.local_size())
            return
        ctx.check(st)
        regrows = C.c_uint64()
        ctx.check(L.lib.kmi_ctx_debug_counter(ctx.h, 0, C.byref(regrows)))
        keys0, cnts0 = idx.to_vector()
        s = orc.kspec(k)
        allk = orc.extract(s, data, orc.FASTQ)["kmers"]
        # ---- insert of k-mers, weighted pairs, update: every rank brings its own slice
        mine = np.ascontiguousarray(allk[rank::world][::11])
        ctx.check(L.lib.kmi_index_insert_dist_host(idx.h, comm.h, mine.ctypes.data_as(C.c_void_p), mine.shape[0]))
        pk = np.ascontiguousarray(allk[rank::world][::13])
        pairs = np.ascontiguousarray(np.concatenate([pk, np.full((pk.shape[0], 1), 3 + rank, np.uint64)], axis=1))
        ctx.check(L.lib.kmi_index_insert_pairs_dist_host(idx.h, comm.h, pairs.ctypes.data_as(C.c_void_p), pairs.shape[0]))
        n = C.c_uint64()
        ctx.check(L.lib.kmi_index_size_dist(idx.h, comm.h, C.byref(n)))
        # ---- queries: every rank asks for its own mix of present and absent keys
        rng = np.random.default_rng(100 + rank)
        q = np.ascontiguousarray(np.concatenate([allk[rng.integers(0, allk.shape[0], 400)], rng.integers(0, 1 << (2 * k), (150, 1), dtype=np.uint64)]))
        r = L.Results()
        ctx.check(L.lib.kmi_index_count_dist_host(idx.h, comm.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        ck, cv = _results(L, r)
        r = L.Results()
        ctx.check(L.lib.kmi_index_find_dist_host(idx.h, comm.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        fk, fv = _results(L, r)
        keys1, cnts1 = idx.to_vector()
        # ---- the routing half on its own (what update() with a host functor stands on): pairs to their keys' owners' HOSTS
        rp = np.ascontiguousarray(np.concatenate([q[:300], (np.arange(300, dtype=np.uint64) + np.uint64(1000 * rank)).reshape(-1, 1)], axis=1))
        r = L.Results()
        ctx.check(L.lib.kmi_index_route_pairs_dist_host(idx.h, comm.h, rp.ctypes.data_as(C.c_void_p), rp.shape[0], C.byref(r)))
        rk, rv = _results(L, r)
        er = np.ascontiguousarray(q[:60])
        ne = C.c_uint64()
        ctx.check(L.lib.kmi_index_erase_dist_host(idx.h, comm.h, er.ctypes.data_as(C.c_void_p), er.shape[0], C.byref(ne)))
        n2 = C.c_uint64()
        ctx.check(L.lib.kmi_index_size_dist(idx.h, comm.h, C.byref(n2)))
        ret[rank] = dict(keys0=keys0.copy(), cnts0=cnts0.copy(), keys1=keys1.copy(), cnts1=cnts1.copy(), size=n.value, q=q.copy(), ck=ck, cv=cv, fk=fk, fv=fv,
                         size_after=n2.value, rp=rp.copy(), rk=rk, rv=rv, calls=dict(comm.calls), regrows=regrows.value, mine=mine.copy(), pairs=pairs.copy())
        idx.close()
        comm.close()
        ctx.free(d)
        ctx.close()
    finally:
        dist.destroy_process_group()


def _same(a_keys, a_cnts, om):
    ok, oc = om.export()
    a, b = orc.sorted_pairs(a_keys, a_cnts), orc.sorted_pairs(ok, oc)
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()


@pytest.mark.parametrize("world,k,strand,case", [(2, 31, "canonical", "plain"), (4, 31, "canonical", "plain"), (2, 21, "single", "plain"),
                                                   (2, 31, "canonical", "fallback-rank1"), (4, 25, "single", "empty-rank0"),
                                                   (2, 31, "canonical", "regrow"), (4, 31, "canonical", "regrow")])
def test_count_index_c_layer_over_ranks(world, k, strand, case):
    import kmerind_amd as K
    data = bytes(K.synth_fastq(seed=31 + world, genome_len=40_000, n_reads=4_000))
    ret = mp.Manager().dict()
    mp.spawn(_count_worker, args=(world, _free_port(), data, k, strand, case, ret), nprocs=world, join=True)
    s = orc.kspec(k)
    st = STRAND[strand]
    allk = orc.extract(s, data, orc.FASTQ)["kmers"]
    om = orc.CountMap(s, st)
    om.insert(allk)
    R = [ret[r] for r in range(world)]
    # the build: the union of the ranks' maps is the single map, no key on two ranks
    keys = np.concatenate([x["keys0"] for x in R])
    _same(keys, np.concatenate([x["cnts0"] for x in R]), om)
    assert np.unique(keys, axis=0).shape[0] == keys.shape[0]
    assert all(x["keys0"].shape[0] > 0 for x in R)                                  # every rank owns a part of the bucket space
    assert all(x["calls"]["all_to_all_v"] >= 4 and x["calls"]["bytes"] > 0 for x in R)   # the messenger really carried the exchange
    if case == "regrow":
        assert sum(x["regrows"] for x in R) >= 1
    # inserts and weighted pairs of every rank
    for x in R:
        om.insert(x["mine"])
        for w in sorted(set(x["pairs"][:, 1].tolist())):
            sel = x["pairs"][x["pairs"][:, 1] == w][:, :1]
            for _ in range(int(w)):
                om.insert(np.ascontiguousarray(sel))
    _same(np.concatenate([x["keys1"] for x in R]), np.concatenate([x["cnts1"] for x in R]), om)
    assert all(x["size"] == om.size() for x in R)
    erased = set()
    for x in R:
        ek, ec = om.count(x["q"])
        a, b = orc.sorted_pairs(x["ck"], x["cv"]), orc.sorted_pairs(ek, ec)
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
        ek, ev = om.find(x["q"])
        a, b = orc.sorted_pairs(x["fk"], x["fv"]), orc.sorted_pairs(ek, np.asarray(ev).astype(np.uint64))
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    # routed pairs: all of them arrive somewhere, with the key the map stores; a stored key arrives at the rank that stores it
    sent = np.concatenate([np.concatenate([(orc.canonical(s, x["rp"][:, :1]) if strand == "canonical" else x["rp"][:, :1]), x["rp"][:, 1:]], axis=1) for x in R])
    got = np.concatenate([np.concatenate([x["rk"], x["rv"].reshape(-1, 1)], axis=1) for x in R])
    assert sent.shape == got.shape and (sent[np.lexsort(sent.T[::-1])] == got[np.lexsort(got.T[::-1])]).all()
    for x in R:
        have = set(x["keys1"][:, 0].tolist())
        others = set(np.concatenate([y["keys1"][:, 0] for y in R if y is not x]).tolist())
        assert not (set(x["rk"][:, 0].tolist()) & others)
        assert len(set(x["rk"][:, 0].tolist()) & have) > 0
    for x in R:
        om.erase(x["q"][:60])
    assert all(x["size_after"] == om.size() for x in R)


def test_a_rank_that_fails_ends_the_build_on_every_rank():
    """ADVICE (round 3): a rank whose front end fails used to return before the chunk's count exchange and leave its peers in
    ncclSend / ncclRecv for ever. Now the failure rides on that exchange: rank 1 returns its own error, the others KMI_ERR_PEER,
    nobody hangs, and no rank's index has changed."""
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    world = 4
    data = bytes(K.synth_fastq(seed=77, genome_len=30_000, n_reads=3_000))
    ret = mp.Manager().dict()
    mp.spawn(_count_worker, args=(world, _free_port(), data, 31, "canonical", "error-rank1", ret), nprocs=world, join=True)
    assert ret[1][0] == L.ERR_PARSE and "KMI_SK_DBG=9" in ret[1][1]
    for r in (0, 2, 3):
        assert ret[r][0] == L.ERR_PEER and "another rank" in ret[r][1], ret[r]
    assert all(ret[r][2] == 0 for r in range(world))


def _pos_worker(rank, world, port, data, k, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmerind_amd as K
        from kmerind_amd import _lib as L
        from kmerind_amd import fileio
        from kmerind_amd.transport import GroupComm
        ctx = K.Context(0, rank=rank, nranks=world)
        comm = GroupComm(ctx)
        idx = K.PositionIndex(ctx, K.make_config(k, "DNA", strand="canonical", index_kind="posqual"))
        b, e = fileio.partition_fastq(data, world)[rank]
        buf = np.frombuffer(data[b:e], dtype=np.uint8).copy()
        ctx.check(L.lib.kmi_index_build_dist_host(idx.h, comm.h, buf.ctypes.data_as(C.c_void_p), buf.size, b))
        keys, vals = idx.to_vector()
        s = orc.kspec(k)
        q = np.ascontiguousarray(orc.extract(s, data, orc.FASTQ)["kmers"][np.random.default_rng(rank).integers(0, 3000, size=200)])
        r = L.Results()
        ctx.check(L.lib.kmi_index_find_dist_host(idx.h, comm.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        fk, fv = _results(L, r, 2)
        n = C.c_uint64()
        ctx.check(L.lib.kmi_index_size_dist(idx.h, comm.h, C.byref(n)))
        ret[rank] = (keys.copy(), vals.copy(), n.value, q.copy(), fk, fv)
        idx.close()
        comm.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_position_quality_index_c_layer_two_ranks():
    """config 5's index over ranks through the C layer: (k-mer, (id, quality)) records parsed on the device with their file
    offsets, grouped by KeyToRank, exchanged, inserted; find answers come back with both value words"""
    import kmerind_amd as K
    world, k = 2, 31
    data = bytes(K.synth_fastq(seed=19, genome_len=15_000, n_reads=1_500))
    ret = mp.Manager().dict()
    mp.spawn(_pos_worker, args=(world, _free_port(), data, k, ret), nprocs=world, join=True)
    s = orc.kspec(k)
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=True)
    vals = np.concatenate([ex["ids"].reshape(-1, 1), ex["quals"].view(np.uint32).astype(np.uint64).reshape(-1, 1)], axis=1)
    ref = orc.MultiMap(s, orc.CANONICAL, 2)
    ref.insert(ex["kmers"], vals)
    rk, rv = ref.export()

    def canon(keys, v):
        rows = np.concatenate([keys, v.reshape(keys.shape[0], -1)], axis=1)
        return rows[np.lexsort([rows[:, c] for c in range(rows.shape[1] - 1, -1, -1)])]

    got = canon(np.concatenate([ret[r][0] for r in range(world)]), np.concatenate([ret[r][1] for r in range(world)]))
    assert got.shape == canon(rk, rv).shape and (got == canon(rk, rv)).all()
    for r in range(world):
        assert ret[r][2] == ref.size()
        assert (orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, ret[r][0], world) == r).all()
        ek, ev = ref.find(ret[r][3])
        assert (canon(ret[r][4], ret[r][5]) == canon(ek, ev)).all()


def _dbg_worker(rank, world, port, data, k, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmerind_amd as K
        from kmerind_amd import _lib as L
        from kmerind_amd import fileio
        from kmerind_amd.transport import GroupComm
        ctx = K.Context(0, rank=rank, nranks=world)
        comm = GroupComm(ctx)
        g = K.DeBruijnNodes(ctx, K.make_config(k))
        b, e = fileio.partition_fastq(data, world)[rank]
        buf = np.frombuffer(data[b:e], dtype=np.uint8).copy()
        ctx.check(L.lib.kmi_dbg_build_dist_host(g.h, comm.h, buf.ctypes.data_as(C.c_void_p), buf.size))
        n = C.c_uint64()
        ctx.check(L.lib.kmi_dbg_size_dist(g.h, comm.h, C.byref(n)))
        keys, cnt = g.to_vector()
        s = orc.kspec(k)
        q = np.ascontiguousarray(np.concatenate([orc.dbg_parse(s, data)[0][rank::7][:300], np.random.default_rng(rank).integers(0, 1 << 62, (100, 1), dtype=np.uint64)]))
        r = L.Results()
        ctx.check(L.lib.kmi_dbg_find_dist_host(g.h, comm.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        fk = np.ctypeslib.as_array(r.keys, shape=(r.n,)).copy().reshape(-1, 1)
        fv = np.ctypeslib.as_array(r.values, shape=(r.n * 5,)).copy().view(np.uint32).reshape(r.n, 10)[:, :9].copy()
        L.lib.kmi_results_free(C.byref(r))
        ret[rank] = (keys.copy(), cnt.copy(), n.value, q.copy(), fk, fv)
        g.close()
        comm.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_de_bruijn_nodes_c_layer_over_ranks():
    """the de Bruijn engine's build over ranks (de_bruijn_construct_engine.hpp:90-158 over the distributed map): the union of the
    ranks' node maps -- k-mers with their eight edge counters and occurrence counts -- is the oracle's single map; find() over
    the communicator answers every rank's own queries"""
    import kmerind_amd as K
    world, k = 2, 31
    data = bytes(K.synth_fastq(seed=41, genome_len=10_000, n_reads=1_500))
    ret = mp.Manager().dict()
    mp.spawn(_dbg_worker, args=(world, _free_port(), data, k, ret), nprocs=world, join=True)
    s = orc.kspec(k)
    om = orc.DbgMap(s)
    om.insert(*orc.dbg_parse(s, data))

    def nodes(keys, cnt):
        rows = np.concatenate([np.asarray(keys, np.uint64).reshape(-1, 1), np.asarray(cnt).astype(np.uint64).reshape(len(keys), -1)], axis=1)
        return rows[np.argsort(rows[:, 0], kind="stable")]

    got = nodes(np.concatenate([ret[r][0] for r in range(world)]), np.concatenate([ret[r][1] for r in range(world)]))
    exp = nodes(*om.export(canonical=True))
    assert got.shape == exp.shape and (got == exp).all()
    for r in range(world):
        assert ret[r][2] == om.size()
        assert (nodes(ret[r][4], ret[r][5]) == nodes(*om.find(ret[r][3], canonical=True))).all()


def _fasta_worker(rank, world, port, data, k, alpha, look, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmerind_amd as K
        from kmerind_amd import _lib as L
        from kmerind_amd.transport import GroupComm
        ctx = K.Context(0, rank=rank, nranks=world)
        comm = GroupComm(ctx)
        idx = K.PositionIndex(ctx, K.make_config(k, alpha, strand="canonical", seq_format="fasta", index_kind="position"))
        n = len(data)
        lo = n // world * rank + (n % world) * rank // world
        hi = n if rank + 1 == world else n // world * (rank + 1) + (n % world) * (rank + 1) // world
        rounds = 0
        while True:                                  # what the facade's build_posix does: the block plus look-ahead, more when asked
            end = min(n, hi + look)
            buf = np.frombuffer(data[lo:end], dtype=np.uint8).copy()
            need = C.c_int(0)
            ptr = buf.ctypes.data_as(C.c_void_p) if buf.size else None
            ctx.check(L.lib.kmi_index_build_fasta_range_dist_host(idx.h, comm.h, ptr, buf.size, lo, hi - lo, 1 if end == n else 0,
                                                                  data[lo - 1] if lo > 0 else -1, C.byref(need)))
            rounds += 1
            if not need.value:
                break
            look *= 16
        keys, vals = idx.to_vector()
        # read_file_* of the same block (kmi_extract_fasta_range_dist_host): the tuples themselves, in file order
        cfg = K.make_config(k, alpha, seq_format="fasta", index_kind="position")
        look2 = 16
        while True:
            end = min(n, hi + look2)
            buf = np.frombuffer(data[lo:end], dtype=np.uint8).copy()
            need = C.c_int(0)
            t = L.Tuples()
            ptr = buf.ctypes.data_as(C.c_void_p) if buf.size else None
            ctx.check(L.lib.kmi_extract_fasta_range_dist_host(ctx.h, C.byref(cfg), comm.h, ptr, buf.size, lo, hi - lo, 1 if end == n else 0,
                                                              data[lo - 1] if lo > 0 else -1, C.byref(need), C.byref(t)))
            if not need.value:
                break
            look2 *= 16
        nw = (k * (3 if alpha == "DNA5" else 2) + 63) // 64
        tk = np.ctypeslib.as_array(t.kmers, shape=(t.n_tuples * nw,)).copy().reshape(-1, nw) if t.n_tuples else np.zeros((0, nw), np.uint64)
        ti = np.ctypeslib.as_array(t.ids, shape=(t.n_tuples,)).copy() if t.n_tuples else np.zeros(0, np.uint64)
        L.lib.kmi_tuples_free(C.byref(t))
        ret[rank] = (keys.copy(), vals.copy(), rounds, comm.calls["bytes"], tk, ti)
        idx.close()
        comm.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,alpha,which", [(2, 31, "DNA", "test.fasta"), (4, 21, "DNA5", "test2.fasta"), (3, 31, "DNA", "synthetic"),
                                                 (4, 63, "DNA5", "synthetic"), (4, 15, "DNA", "tiny")])
def test_fasta_position_index_by_byte_range_over_ranks(world, k, alpha, which):
    """FASTA over ranks the way the reference shards it (file.hpp:1436-1610): every rank holds 1/p of the file's BYTES plus
    look-ahead, wherever that cuts -- inside a header, inside a sequence line, before the first header -- and the block
    bookkeeping comes from the ranks' block summaries (kmi_fasta_block_summary_dev) gathered inside
    kmi_index_build_fasta_range_dist_host. The union of the ranks' (k-mer, LongSequenceKmerId) tuples is the oracle's parse of the
    whole file: same k-mers, same sequence indices, same file offsets."""
    from tests.test_gpu_fasta import _synthetic_fasta, GOLD
    alpha_id = {"DNA": orc.DNA, "DNA5": orc.DNA5}[alpha]
    if which == "synthetic":
        data = _synthetic_fasta(np.random.default_rng(5 * k + world), 40, line=60, eol=b"\n", orphan=True)
    elif which == "tiny":
        data = b"ACGTACGTACGTACGTAAC\n>r1 x\nACGTTGCATGCATGCATGCAAGT\nTTGACCA\n;c\n>r2\n\nGGGTACGATCGATCGATGCATGCAC\n"
    else:
        data = open(os.path.join(GOLD, "data", which), "rb").read()
    ret = mp.Manager().dict()
    mp.spawn(_fasta_worker, args=(world, _free_port(), data, k, alpha, 16, ret), nprocs=world, join=True)   # 16 bytes of look-ahead: every rank has to ask for more
    s = orc.kspec(k, alpha_id)
    ex = orc.extract(s, data, orc.FASTA, want_ids=True)
    ref = orc.MultiMap(s, orc.CANONICAL, 1)
    ref.insert(ex["kmers"], ex["ids"].reshape(-1, 1))
    rk, rv = ref.export()

    def canon(keys, v):
        rows = np.concatenate([np.asarray(keys).reshape(len(v), -1), np.asarray(v).reshape(len(v), -1)], axis=1)
        return rows[np.lexsort([rows[:, c] for c in range(rows.shape[1] - 1, -1, -1)])]

    got = canon(np.concatenate([ret[r][0] for r in range(world)]), np.concatenate([ret[r][1] for r in range(world)]))
    exp = canon(rk, rv)
    assert got.shape == exp.shape and (got == exp).all()
    assert any(ret[r][2] > 1 for r in range(world)) or k <= 17     # the look-ahead had to grow on some rank
    # read_file_*: the ranks' tuples, rank after rank, are the file's tuples in file order
    tk = np.concatenate([ret[r][4] for r in range(world)])
    ti = np.concatenate([ret[r][5] for r in range(world)])
    assert tk.shape == ex["kmers"].shape and (tk == ex["kmers"]).all() and (ti == ex["ids"]).all()
