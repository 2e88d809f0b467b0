"""De Bruijn node build rate (not the headline metric): kmi_dbg_build_dev over a resident FASTQ buffer, per-kernel times,
and the checker's CPU restatement on a bounded sample for scale.
  python tools/dbg_bench.py [reads] [genome] [k]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    genome = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
    host = np.asarray(K.synth_fastq(seed=5, genome_len=genome, n_reads=n_reads))
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(host).to(dev)
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    g = K.DeBruijnNodes(ctx, K.make_config(k))
    for _ in range(2):
        g.clear(); g.build_device(d.data_ptr(), host.size)
    torch.cuda.synchronize()
    ctx.profile(True); ctx.profile_reset()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        g.clear(); g.build_device(d.data_ptr(), host.size)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nk = n_reads * (150 - k + 1)
    prof = sorted(ctx.profile_get(), key=lambda p: -p["total_ms"])
    print("de Bruijn nodes, k=%d, %d reads over %d bp: %.2f ms per build, %.2f G k-mers/s, %d nodes" % (k, n_reads, genome, dt * 1e3, nk / dt / 1e9, g.local_size()))
    print({p["name"]: round(p["total_ms"] / steps, 3) for p in prof if p["launches"]})
    # the checker's restatement on the first 20 000 reads (one core)
    from tests import oracle as orc
    s = orc.kspec(k)
    head = bytes(host[: 315 * 20_000])
    t0 = time.perf_counter()
    kk, ee = orc.dbg_parse(s, head)
    m = orc.DbgMap(s)
    m.insert(kk, ee)
    dt_cpu = time.perf_counter() - t0
    print("CPU restatement (1 core, 20 000 reads): %.2f M k-mers/s" % (kk.shape[0] / dt_cpu / 1e6))


if __name__ == "__main__":
    main()
