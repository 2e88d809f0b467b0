"""Times the device side of the N > 1 build path on ONE rank (no exchange): kmi_extract_route_dev for p ranks on the
config-2 batch, then kmi_index_insert_dev of a same-size key array."""
import ctypes as C
import sys
import numpy as np
import torch
import kmerind_amd as K
from kmerind_amd import _lib as L

p = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
ctx = K.Context(device=0, rank=0, nranks=1, stream=stream.cuda_stream)
cfg = K.make_config(31, "DNA", strand="canonical")
host = K.synth_fastq(3, 100_000_000 * p, n_reads, 150)
d_bytes = torch.from_numpy(host).to(dev)
n_kmers = n_reads * 120
d_send = torch.empty((n_kmers + 64, 1), dtype=torch.int64, device=dev)
counts = np.zeros(p, dtype=np.uint64)
nt, ns = C.c_uint64(), C.c_uint64()
idx = K.CountIndex(ctx, cfg)
for it in range(3):
    if it == 1:
        ctx.profile(True); ctx.profile_reset()
    idx.clear()
    ctx.check(L.lib.kmi_extract_route_dev(ctx.h, C.byref(cfg), C.c_void_p(d_bytes.data_ptr()), host.nbytes, p,
                                          C.c_void_p(d_send.data_ptr()), d_send.shape[0], C.byref(nt), C.byref(ns),
                                          counts.ctypes.data_as(C.c_void_p)))
    idx.insert_device(d_send.data_ptr(), nt.value)
    torch.cuda.synchronize()
prof = ctx.profile_get()
tot = 0.0
for q in sorted(prof, key=lambda q: -q["total_ms"]):
    if q["launches"]:
        print("%-22s %8.3f ms" % (q["name"], q["total_ms"] / q["launches"]))
        tot += q["total_ms"] / q["launches"]
print("sum %.3f ms; counts balance max/mean %.4f" % (tot, counts.max() / counts.mean()))
