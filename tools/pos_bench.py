"""Position / position+quality index build rate (not the headline metric).
  python tools/pos_bench.py [reads] [kind]                 one rank, fused device build
  python tools/pos_bench.py [reads] [kind] --ranks 2       rehearsal of config 5's flow with two ranks sharing the GPU (gloo):
      kmerind_amd.dist.DistributedPositionIndex -- records parsed, routed and inserted on the device in record-aligned
      batches, only the exchanged records staged through the host (RCCL moves them device to device); then 100 k routed
      find queries per rank."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K


def one_rank(n_reads, kind):
    host = np.asarray(K.synth_fastq(seed=5, genome_len=20_000_000, n_reads=n_reads))
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(host).to(dev)
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = K.make_config(31, "DNA", strand="canonical", index_kind=kind)
    idx = K.PositionIndex(ctx, cfg)
    for _ in range(2):
        idx.clear(); idx.build_device(d.data_ptr(), host.size)
    torch.cuda.synchronize()
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    steps = 3
    for _ in range(steps):
        idx.clear(); idx.build_device(d.data_ptr(), host.size)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nk = n_reads * 120
    prof = sorted(ctx.profile_get(), key=lambda p: -p["total_ms"])
    print("%s index, %d reads: %.2f ms per build, %.1f G tuples/s, entries %d" % (kind, n_reads, dt * 1e3, nk / dt / 1e9, idx.local_size()))
    print({p["name"]: round(p["total_ms"] / steps, 3) for p in prof if p["launches"]})


def rank_main(rank, world, port, n_reads, kind):
    import torch.distributed as dist
    from kmerind_amd import dist as kdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    per = n_reads // world
    host = np.asarray(K.synth_fastq(seed=5, genome_len=20_000_000, n_reads=per, first_read=rank * per))
    d = torch.from_numpy(host).to(dev)
    ctx = K.Context(0, rank=rank, nranks=world, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = K.make_config(31, "DNA", strand="canonical", index_kind=kind)
    didx = kdist.DistributedPositionIndex(ctx, cfg, stage_through_host=True, device=dev)
    dist.barrier()
    t0 = time.perf_counter()
    didx.build_device(d.data_ptr(), host.size, file_offset=rank * host.size, batch_bytes=64 << 20)
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    q = np.random.default_rng(rank).integers(0, 1 << 62, size=(100_000, 1), dtype=np.uint64)
    t1 = time.perf_counter()
    fk, fv = didx.find(q)
    dq = time.perf_counter() - t1
    total = didx.size()
    if rank == 0:
        print("%s index over %d ranks on one GPU (gloo rehearsal, batches of 64 MiB): %d reads, %.1f ms build = %.2f G tuples/s, %d entries; "
              "100 k routed find queries per rank %.1f ms" % (kind, world, per * world, dt * 1e3, per * world * 120 / dt / 1e9, total, dq * 1e3))
    didx.close(); ctx.close()
    dist.destroy_process_group()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_reads = int(args[0]) if len(args) > 0 else 2_000_000
    kind = args[1] if len(args) > 1 else "position"
    if "--ranks" in sys.argv:
        import socket
        import torch.multiprocessing as mp
        world = int(sys.argv[sys.argv.index("--ranks") + 1])
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(rank_main, args=(world, port, n_reads, kind), nprocs=world, join=True)
    else:
        one_rank(n_reads, kind)


if __name__ == "__main__":
    main()
