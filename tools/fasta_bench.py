"""FASTA count-index build rate (not the headline metric): G random bases, 80 per line, one header per 50 Mbp.
usage: python tools/fasta_bench.py [G_Mbp] [k] [alphabet]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kmerind_amd as K


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    alpha = sys.argv[3] if len(sys.argv) > 3 else "DNA"
    rng = np.random.default_rng(4)
    parts = []
    for r in range(max(1, g // 50)):
        n = min(50, g) * 1_000_000
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n, dtype=np.uint8)]
        lines = np.empty((n // 80, 81), dtype=np.uint8)
        lines[:, :80] = seq[: n // 80 * 80].reshape(-1, 80)
        lines[:, 80] = 10
        parts.append(np.frombuffer(b">chr%d\n" % r, dtype=np.uint8))
        parts.append(lines.reshape(-1))
    host = np.concatenate(parts)
    pad = (-host.size) % 16
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(np.concatenate([host, np.zeros(pad, np.uint8)])).to(dev)
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = K.make_config(k, alpha, strand="canonical", seq_format="fasta")
    idx = K.CountIndex(ctx, cfg)
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
    prof = sorted(ctx.profile_get(), key=lambda p: -p["total_ms"])
    nk = host.size * 80 // 81
    print("FASTA %d Mbp k=%d %s: %.2f ms per build, ~%.1f G k-mers/s, distinct %d" % (g, k, alpha, dt * 1e3, nk / dt / 1e9, idx.local_size()))
    print({p["name"]: round(p["total_ms"] / steps, 3) for p in prof if p["launches"]})


main()
