"""Two ranks on one GPU (gloo, payload staged through the host): kmerind_amd.dist.DistributedCountIndex, the combine-first
N > 1 count build, end to end through the C ABI, against the oracle's single map."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import oracle as orc

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, k, strand, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        import kmerind_amd as K
        from kmerind_amd import dist as kdist
        from kmerind_amd import fileio
        dev = torch.device("cuda", 0)
        ctx = K.Context(0, rank=rank, nranks=world)
        cfg = K.make_config(k, "DNA", strand=strand)
        didx = kdist.DistributedCountIndex(ctx, cfg, stage_through_host=True, device=dev)
        parts = fileio.partition_fastq(data, world * 2)             # two build calls per rank: the index grows incrementally
        for j in range(2):
            b, e = parts[rank * 2 + j]
            buf = np.frombuffer(data[b:e], dtype=np.uint8)
            pad = (-buf.size) % 16
            d = torch.from_numpy(np.concatenate([buf, np.zeros(pad, np.uint8)])).to(dev)
            didx.build_device(d.data_ptr(), buf.size, dev)
        keys, cnts = didx.index.to_vector()
        ret[rank] = (keys.copy(), cnts.copy(), didx.size())
        didx.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,strand", [(31, "canonical"), (21, "single")])
def test_distributed_count_index_two_ranks_one_gpu(k, strand):
    import kmerind_amd as K
    world = 2
    data = bytes(K.synth_fastq(seed=9, genome_len=30_000, n_reads=2_000))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), data, k, strand, ret), nprocs=world, join=True)
    s = orc.kspec(k)
    st = orc.CANONICAL if strand == "canonical" else orc.SINGLE
    ref = orc.CountMap(s, st)
    ref.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
    rk, rc = ref.export()
    keys = np.concatenate([ret[r][0] for r in range(world)])
    cnts = np.concatenate([ret[r][1] for r in range(world)])
    a, b = orc.sorted_pairs(keys, cnts), orc.sorted_pairs(rk, rc)
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    for r in range(world):
        assert ret[r][2] == ref.size()
        assert (orc.key_to_rank(s, orc.MURMUR, st, ret[r][0], world) == r).all()
