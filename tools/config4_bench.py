"""Config 4: k = 63 DNA5 PositionIndex over a synthetic FASTA genome with N runs -- build rate and per-kernel times
(not the headline metric).   python tools/config4_bench.py [Mbp]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K
from tests.test_gpu_fullsize import _synth_fasta


def main():
    mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    data, recs = _synth_fasta(mbp * 1_000_000, min(100_000_000, mbp * 1_000_000), seed=4)
    dev = torch.device("cuda", 0)
    pad = (-data.size) % 16
    d = torch.from_numpy(np.concatenate([data, np.zeros(pad, np.uint8)])).to(dev)
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = K.make_config(63, "DNA5", strand="canonical", index_kind="position", seq_format="fasta")
    idx = K.PositionIndex(ctx, cfg)
    idx.build_device(d.data_ptr(), data.size)
    torch.cuda.synchronize()
    ctx.profile(True); ctx.profile_reset()
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.clear(); idx.build_device(d.data_ptr(), data.size)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = idx.local_size()
    prof = sorted(ctx.profile_get(), key=lambda p: -p["total_ms"])
    print("config 4, %d Mbp: %.2f ms per build, %.1f G tuples/s (%d tuples of 32 B), 33.01 B/tuple -> %.0f GB/s of the contract bytes" %
          (mbp, dt * 1e3, n / dt / 1e9, n, n * 33.01 / dt / 1e9))
    print({p["name"]: round(p["total_ms"] / steps, 3) for p in prof if p["launches"]})


if __name__ == "__main__":
    main()
