"""Do the build's kernels leave room for each other? Two contexts on two streams build the SAME index of config 2 at the same time
(two host threads; the library releases the GIL inside a call): if a pair of concurrent builds takes clearly less than two builds
in a row, a chunked build that runs the front end of chunk c + 1 beside the scatter passes of chunk c would gain that much.
  python tools/overlap_probe.py [reads]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    host = np.asarray(K.synth_fastq(seed=1, genome_len=n_reads * 10, n_reads=n_reads))
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(host).to(dev)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    ctxs = [K.Context(0, stream=s.cuda_stream) for s in streams]
    idxs = [K.CountIndex(c, K.make_config(31, "DNA", strand="canonical")) for c in ctxs]

    def builds(i, n):
        for _ in range(n):
            idxs[i].clear(); idxs[i].build_device(d.data_ptr(), host.size)
        streams[i].synchronize()

    for i in (0, 1):
        builds(i, 3)
    steps = 10
    t0 = time.perf_counter(); builds(0, steps); t1 = time.perf_counter()
    one = (t1 - t0) / steps
    th = [threading.Thread(target=builds, args=(i, steps)) for i in (0, 1)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    pair = (time.perf_counter() - t0) / steps
    print("one build %.2f ms; two at the same time %.2f ms per pair (%.2f per build): overlap gains %.0f %%" % (one * 1e3, pair * 1e3, pair * 500, 100 * (1 - pair / (2 * one))))


main()
