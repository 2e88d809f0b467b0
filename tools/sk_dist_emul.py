"""What ONE rank of an N-GPU super-k-mer build does after the exchange, measured on one GPU: the reads of every rank of the
weak-scaling family (12.5 M reads per rank of an N x 125 Mbp genome) go through the front end here, one rank's input after the
other, owner 0's share of each is kept -- together exactly what rank 0 receives -- and kmi_index_sk_consume_dev is timed on it.
Also prints the front end's time for one rank's input. (No exchange: a one-GPU box has no peer.)
  python tools/sk_dist_emul.py [N] [reads_per_rank]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K
from kmerind_amd import _lib as L


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 12_500_000
    genome = 125_000_000 * world * n_reads // 12_500_000
    dev = torch.device("cuda", 0)
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = K.make_config(31)
    idx = K.CountIndex(ctx, cfg)
    parts, t_front = [], None
    for r in range(world):
        host = K.synth_fastq(3, genome, n_reads, 150, first_read=r * n_reads)
        d = torch.from_numpy(np.asarray(host)).to(dev)
        recs, n, produced = C.c_void_p(), C.c_uint64(), C.c_int()
        sc = np.zeros(world, dtype=np.uint64)
        for rep in range(2 if r == 0 else 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.check(L.lib.kmi_index_sk_produce_dev(idx.h, C.c_void_p(d.data_ptr()), host.size, world, None, 0, C.byref(recs), C.byref(n),
                                                     sc.ctypes.data_as(C.c_void_p), C.byref(produced)))
            torch.cuda.synchronize()
            if r == 0:
                t_front = time.perf_counter() - t0
        assert produced.value
        mine = torch.empty((int(sc[0]), 2), dtype=torch.int64, device=dev)     # owner 0's records come first
        ctx.check(L.lib.kmi_copy_on_device(ctx.h, C.c_void_p(mine.data_ptr()), recs, int(sc[0]) * 16))
        parts.append(mine)
        print("rank %d: %d records, %d for owner 0 (max/mean over owners %.4f)" % (r, n.value, int(sc[0]), float(sc.max()) / float(sc.mean())), flush=True)
        del d, host
    recv = torch.cat(parts)
    del parts
    for _ in range(2):
        idx.clear()
        ctx.check(L.lib.kmi_index_sk_consume_dev(idx.h, C.c_void_p(recv.data_ptr()), recv.shape[0], world))
    torch.cuda.synchronize()
    ctx.profile(True); ctx.profile_reset()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.clear()
        ctx.check(L.lib.kmi_index_sk_consume_dev(idx.h, C.c_void_p(recv.data_ptr()), recv.shape[0], world))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = sorted(ctx.profile_get(), key=lambda p: -p["total_ms"])
    print("N = %d: front end of one rank's reads %.2f ms; consume of %d received records (%.2f GB) %.2f ms; %d distinct k-mers on this rank" %
          (world, t_front * 1e3, recv.shape[0], recv.shape[0] * 16 / 1e9, dt * 1e3, idx.local_size()))
    print({p["name"]: round(p["total_ms"] / steps, 3) for p in prof if p["launches"]})


if __name__ == "__main__":
    main()
