"""Builds that could hide a performance cliff (skewed or low-duplication inputs, shapes off the headline path), timed once each:
  python tools/cliff_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K


def timed(name, fn):
    t0 = time.perf_counter()
    r = fn()
    print("%-58s %9.1f ms   %s" % (name, (time.perf_counter() - t0) * 1e3, r), flush=True)


def main():
    ctx = K.Context(0)
    reads = 1_000_000
    poly = np.frombuffer((b"@p\n" + b"A" * 150 + b"\n+\n" + b"I" * 150 + b"\n") * reads, dtype=np.uint8)
    rep = np.frombuffer((b"@p\n" + b"ACGTTGCA" * 18 + b"ACGTTG\n+\n" + b"I" * 150 + b"\n") * reads, dtype=np.uint8)
    norm = np.asarray(K.synth_fastq(seed=3, genome_len=10_000_000, n_reads=reads))
    low = np.asarray(K.synth_fastq(seed=4, genome_len=400_000_000, n_reads=4 * reads))
    for label, data in (("poly-A", poly), ("8-periodic", rep), ("12x coverage", norm), ("low duplication (4 M reads over 400 Mbp)", low)):
        d = ctx.alloc(data.nbytes); ctx.to_device(d, data)
        for k, alpha, kind in ((31, "DNA", "count"), (63, "DNA", "count"), (21, "DNA5", "count"), (31, "DNA", "position"), (31, "DNA", "dbg")):
            if kind == "count":
                idx = K.CountIndex(ctx, K.make_config(k, alpha))
            elif kind == "position":
                idx = K.PositionIndex(ctx, K.make_config(k, alpha, index_kind="position"))
            else:
                idx = K.DeBruijnNodes(ctx, K.make_config(k, alpha))
            idx.build_device(d, data.nbytes)   # warm-up (allocations)
            idx.clear()
            timed("%s, %s k=%d %s" % (label, kind, k, alpha), lambda: (idx.build_device(d, data.nbytes), idx.local_size())[1])
            idx.close()
        ctx.free(d)
    ctx.close()


main()
