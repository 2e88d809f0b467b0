"""The first build of a FRESH PROCESS, attributed: process start -> import -> context -> first build (of which: time inside hipMalloc /
hipFree, bytes and calls that reached hipMalloc) -> second build; then a second context in the same process (its blocks come from the
process-wide cache).   python tools/cold_fresh.py [reads] [genome]
(bench.py's extra.cold times a second context in a warm process; the driver saw 805 ms for it in round 3 where the build box saw 8.)"""
import ctypes as C
import os
import sys
import time

t_start = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmerind_amd as K
from kmerind_amd import _lib as L
t_import = time.perf_counter()
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
genome = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
host = np.asarray(K.synth_fastq(seed=2, genome_len=genome, n_reads=reads))
dev = torch.device("cuda", 0)
d = torch.from_numpy(host).to(dev)
torch.cuda.synchronize()
t_data = time.perf_counter()


def counters(ctx):
    v = C.c_uint64()
    out = []
    for w in (1, 2, 3, 4):
        ctx.check(L.lib.kmi_ctx_debug_counter(ctx.h, w, C.byref(v)))
        out.append(v.value)
    return out


def one_context(tag):
    t0 = time.perf_counter()
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    idx = K.CountIndex(ctx, K.make_config(31))
    torch.cuda.synchronize()
    t_ctx = time.perf_counter() - t0
    res = []
    prev = counters(ctx)
    for i in range(3):
        idx.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.build_device(d.data_ptr(), host.size); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        cur = counters(ctx)
        res.append("%.2f ms (hipMalloc/hipFree %.2f ms, %.2f GB in %d calls, %d cached blocks reused)" % (
            ms, (cur[0] - prev[0]) / 1e3, (cur[1] - prev[1]) / 1e9, cur[2] - prev[2], cur[3] - prev[3]))
        prev = cur
    print("%s: context %.1f ms; builds: %s" % (tag, t_ctx * 1e3, " | ".join(res)), flush=True)
    idx.close(); ctx.close()


print("process start -> import %.0f ms; synthetic input + upload %.0f ms; free %.1f GB" % ((t_import - t_start) * 1e3, (t_data - t_import) * 1e3, torch.cuda.mem_get_info(dev)[0] / 1e9), flush=True)
one_context("first context of the process")
one_context("second context (after the first was destroyed)")
freed = C.c_uint64()
L.lib.kmi_release_cached_memory(-1, C.byref(freed))
print("kmi_release_cached_memory freed %.1f GB" % (freed.value / 1e9))
one_context("third context (cache released)")
