"""first build of a fresh context vs the following ones (allocations included in the first), with and without another context
alive beside it:  python tools/cold_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K
host = np.asarray(K.synth_fastq(seed=2, genome_len=100_000_000, n_reads=10_000_000))
dev = torch.device("cuda", 0)
d = torch.from_numpy(host).to(dev)
keep = []
for rep in range(4):
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    idx = K.CountIndex(ctx, K.make_config(31))
    ts = []
    for i in range(3):
        idx.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.build_device(d.data_ptr(), host.size); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print("context", rep, "builds (ms):", ts, "(earlier contexts alive: %d)" % len(keep), "free GB: %.1f" % (torch.cuda.mem_get_info(dev)[0] / 1e9))
    if rep < 2:
        keep.append((ctx, idx))
    else:
        idx.close(); ctx.close()
