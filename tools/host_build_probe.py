"""kmi_index_build_host with and without the chunked copy (KMI_HOST_OVERLAP), against the same build from device memory:
  python tools/host_build_probe.py [reads]"""
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
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    host = np.asarray(K.synth_fastq(seed=2, genome_len=n_reads * 10, n_reads=n_reads))
    dev = torch.device("cuda", 0)
    pinned = torch.from_numpy(host).pin_memory()
    d = torch.empty(host.size, dtype=torch.uint8, device=dev)
    ctx = K.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = K.make_config(31, "DNA", strand="canonical")
    idx = K.CountIndex(ctx, cfg)
    ref = K.CountIndex(ctx, cfg)
    d.copy_(pinned); ref.build_device(d.data_ptr(), host.size)
    rk, rc = ref.to_vector()

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps * 1e3

    def fed():
        idx.clear(); ctx.check(L.lib.kmi_index_build_host(idx.h, C.c_void_p(pinned.data_ptr()), host.size, 0))

    def serial():
        idx.clear(); d.copy_(pinned, non_blocking=True); idx.build_device(d.data_ptr(), host.size)

    def copy():
        d.copy_(pinned, non_blocking=True)

    t_fed = timed(fed)
    k, c = idx.to_vector()
    o1, o2 = np.argsort(k[:, 0], kind="stable"), np.argsort(rk[:, 0], kind="stable")
    same = k.shape == rk.shape and bool((k[o1] == rk[o2]).all()) and bool((c[o1] == rc[o2]).all())
    print("kmi_index_build_host %.2f ms; copy alone %.2f ms; copy then build %.2f ms; same index as the device build: %s (%d entries)"
          % (t_fed, timed(copy), timed(serial), same, k.shape[0]))
    print({p["name"]: round(p["total_ms"], 3) for p in ctx.profile_get() if p["launches"]} if False else "")


main()
