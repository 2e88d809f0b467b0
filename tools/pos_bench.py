"""Position / position+quality index build rate (not the headline metric). usage: python tools/pos_bench.py [reads] [kind]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kmerind_amd as K


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    kind = sys.argv[2] if len(sys.argv) > 2 else "position"
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


main()
