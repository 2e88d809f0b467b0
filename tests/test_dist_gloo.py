"""world_size-2 rehearsal of the N>1 path on CPU tensors (gloo): record-aligned sharding,
KeyToRank routing and the all-to-all exchange wrapper that bench.py runs over RCCL.
The GPU compute is replaced by the oracle here -- as the checker's stand-in only -- so what
is exercised is the product's host logic: kmerind_amd.fileio and kmerind_amd.dist."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, k, ret, msg_max=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmerind_amd import dist as kdist
        from kmerind_amd import fileio
        if msg_max:
            kdist.MSG_MAX_WORDS = msg_max      # force the piecewise exchange
        s = orc.kspec(k)
        b, e = fileio.partition_fastq(data, world)[rank]
        ex = orc.extract(s, data[b:e], orc.FASTQ, file_offset=b)
        keys = orc.canonical(s, ex["kmers"])                       # transform_input
        ranks = orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, keys, world)
        order = np.argsort(ranks, kind="stable")                    # stable bucket permutation
        send = torch.from_numpy(keys[order].view(np.int64))
        counts = np.bincount(ranks, minlength=world).tolist()
        recv, recv_counts = kdist.exchange_keys(send, counts)
        m = orc.CountMap(s, orc.SINGLE)
        m.insert(recv.numpy().view(np.uint64))
        total = kdist.global_size(m.size())
        mine_k, mine_c = m.export()
        ret[rank] = (mine_k.copy(), mine_c.copy(), total, recv_counts, counts, ex["kmers"].shape[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,msg_max", [(2, None), (3, None), (2, 5000)])
def test_sharded_build_over_gloo_matches_single_rank(world, msg_max):
    """msg_max = 5000: peer messages of ~36 k keys go in 8 pieces (the path that keeps RCCL messages below 1 GiB)"""
    import kmerind_amd as K
    k = 31
    data = bytes(K.synth_fastq(seed=3, genome_len=20_000, n_reads=600))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), data, k, ret, msg_max), nprocs=world, join=True)
    s = orc.kspec(k)
    ex = orc.extract(s, data, orc.FASTQ)
    ref = orc.CountMap(s, orc.CANONICAL)
    ref.insert(ex["kmers"])
    rk, rc = ref.export()
    keys = np.concatenate([ret[r][0] for r in range(world)])
    cnts = np.concatenate([ret[r][1] for r in range(world)])
    assert sum(ret[r][5] for r in range(world)) == ex["kmers"].shape[0]
    a, b = orc.sorted_pairs(keys, cnts), orc.sorted_pairs(rk, rc)
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    for r in range(world):
        assert ret[r][2] == ref.size()                              # MapType::size() = allreduce
        # every key a rank owns hashes to that rank (KeyToRank), as in the reference
        assert (orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, ret[r][0], world) == r).all()
        # all2all(counts): what r receives from src is what src sent to r
        assert ret[r][3] == [ret[src][4][r] for src in range(world)]


def _pairs_worker(rank, world, port, data, k, nb, ret, msg_max=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmerind_amd import dist as kdist
        from kmerind_amd import fileio
        if msg_max:
            kdist.MSG_MAX_WORDS = msg_max      # force the piecewise exchange
        s = orc.kspec(k)
        b, e = fileio.partition_fastq(data, world)[rank]
        m = orc.CountMap(s, orc.CANONICAL)                          # the rank's local reduction
        m.insert(orc.extract(s, data[b:e], orc.FASTQ, file_offset=b)["kmers"])
        keys, cnts = m.export()
        ranks = orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, keys, world)
        bucket = (keys[:, 0] % np.uint64(nb)).astype(np.int64)      # stand-in for the device's placement buckets
        order = np.lexsort([bucket, ranks])                         # by rank, then by bucket: the split's output order
        bc = np.zeros((world, nb), dtype=np.int32)
        np.add.at(bc, (ranks.astype(np.int64), bucket), 1)
        rk, rv, rb = kdist.exchange_pairs(torch.from_numpy(keys[order].view(np.int64)), torch.from_numpy(cnts[order].astype(np.int32)),
                                          torch.from_numpy(bc))
        ret[rank] = (rk.numpy().view(np.uint64).copy(), rv.numpy().copy(), rb.numpy().copy(), bc)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,msg_max", [(2, None), (3, None), (2, 700)])
def test_combine_first_exchange_over_gloo(world, msg_max):
    """exchange_pairs: the (k-mer, count) messages and the bucket-count matrix of the combine-first count insert. Row s of what
    rank r receives is row r of what rank s sent; every part is still ordered by bucket; summing the received counts per key
    gives the single-rank map."""
    import kmerind_amd as K
    k, nb = 31, 16
    data = bytes(K.synth_fastq(seed=5, genome_len=5_000, n_reads=900))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pairs_worker, args=(world, _free_port(), data, k, nb, ret, msg_max), nprocs=world, join=True)   # msg_max: several pieces per message
    s = orc.kspec(k)
    ref = orc.CountMap(s, orc.CANONICAL)
    ref.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
    merged = {}
    for r in range(world):
        rk, rv, rb, _ = ret[r]
        assert rb.tolist() == [ret[src][3][r].tolist() for src in range(world)]
        off = 0
        for src in range(world):
            n = int(rb[src].sum())
            part = rk[off:off + n, 0]
            assert (np.diff((part % np.uint64(nb)).astype(np.int64)) >= 0).all()      # bucket order survives the exchange
            assert ((part % np.uint64(nb)).astype(np.int64) == np.repeat(np.arange(nb), rb[src])).all()
            off += n
        assert off == rk.shape[0] == rv.shape[0]
        assert (orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, rk, world) == r).all()
        for key, c in zip(rk[:, 0].tolist(), rv.tolist()):
            merged[key] = merged.get(key, 0) + c
    ek, ec = ref.export()
    assert len(merged) == ek.shape[0]
    assert all(merged[int(key)] == int(c) for key, c in zip(ek[:, 0], ec))


def _order_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmerind_amd import dist as kdist
        kdist.MSG_MAX_WORDS = 64                      # every peer message of a few hundred rows goes in several pieces
        counts = [100 + 37 * ((rank + d) % world) for d in range(world)]
        rows = []
        for d, c in enumerate(counts):                # row value = (source, destination, position inside the message)
            rows.append(np.stack([np.full(c, rank), np.full(c, d), np.arange(c)], axis=1))
        send = torch.from_numpy(np.concatenate(rows).astype(np.int64))
        recv, rc = kdist.exchange_keys(send, counts)
        assert kdist.verify_exchange(send, counts, recv, rc)
        bad = recv.clone()
        bad[0, 2] += 1                                # a corrupted payload must be noticed
        try:
            kdist.verify_exchange(send, counts, bad, rc)
            noticed = False
        except RuntimeError:
            noticed = True
        ret[rank] = (recv.numpy().copy(), rc, noticed)
    finally:
        dist.destroy_process_group()


def test_exchange_keys_keeps_source_order_when_messages_travel_in_pieces():
    """imxx::distribute's receive buffer is the concatenation by source rank ascending (incremental_mxx.hpp:1098); the routed
    query / answer exchanges slice it per source, so the order must hold for any number of pieces"""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_order_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        recv, rc, noticed = ret[r]
        assert noticed
        off = 0
        for src in range(world):
            c = 100 + 37 * ((src + r) % world)
            assert rc[src] == c
            seg = recv[off:off + c]
            assert (seg[:, 0] == src).all() and (seg[:, 1] == r).all() and (seg[:, 2] == np.arange(c)).all()
            off += c
        assert off == recv.shape[0]


def test_bench_launcher_reports_a_failed_rank():
    """python bench.py --gpus 2 without a GPU: both children fail at context creation; the launcher must come back non-zero
    (not hang, not exit 0)"""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--reads", "1000", "--genome",
                          "100000", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
