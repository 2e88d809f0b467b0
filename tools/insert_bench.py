"""Per-kernel times of extract-only and insert-only (config 2 batch) on one GPU."""
import ctypes as C
import time
import torch
import kmerind_amd as K
from kmerind_amd import _lib as L

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
ctx = K.Context(device=0, rank=0, nranks=1, stream=stream.cuda_stream)
cfg = K.make_config(31, "DNA", strand="canonical")
n_reads = 10_000_000
host = K.synth_fastq(2, 100_000_000, n_reads, 150)
d_bytes = torch.from_numpy(host).to(dev)
n_kmers = n_reads * 120
d_keys = torch.empty((n_kmers + 64, 1), dtype=torch.int64, device=dev)
nt, ns = C.c_uint64(), C.c_uint64()
idx = K.CountIndex(ctx, cfg)
for it in range(3):
    if it == 1:
        ctx.profile(True); ctx.profile_reset()
    ctx.check(L.lib.kmi_extract_dev(ctx.h, C.byref(cfg), C.c_void_p(d_bytes.data_ptr()), host.nbytes, 0, C.c_void_p(d_keys.data_ptr()),
                                    None, n_kmers, C.byref(nt), C.byref(ns)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx.clear()
    idx.insert_device(d_keys.data_ptr(), nt.value)
    torch.cuda.synchronize()
    print("insert wall %.2f ms" % ((time.perf_counter() - t0) * 1e3))
for q in sorted(ctx.profile_get(), key=lambda q: -q["total_ms"]):
    if q["launches"]:
        print("%-22s %8.3f ms x %d" % (q["name"], q["total_ms"] / q["launches"], q["launches"]))
