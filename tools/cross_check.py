"""The super-k-mer build against the k-mer pipeline on the same inputs, larger than the oracle tests afford: random read counts,
genome sizes (duplication from 1 x to 60 x) and k; both count maps exported, sorted and compared entry by entry.
  python tools/cross_check.py [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K


def export_sorted(idx):
    k, c = idx.to_vector()
    o = np.argsort(k[:, 0], kind="stable")
    return k[o, 0], np.asarray(c)[o]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    ctx = K.Context(0)
    os.environ["KMI_FUSED_PATH"] = "kmer"
    ctx_k = K.Context(0)
    del os.environ["KMI_FUSED_PATH"]
    rng = np.random.default_rng(20260101)
    for r in range(rounds):
        n_reads = int(rng.integers(20_000, 1_500_000))
        genome = int(n_reads * 150 / rng.choice([1, 2, 6, 12, 30, 60]))
        k = int(rng.integers(17, 33))
        strand = "canonical" if rng.random() < 0.7 else "single"
        data = np.asarray(K.synth_fastq(seed=1000 + r, genome_len=max(genome, 1000), n_reads=n_reads))
        a = K.CountIndex(ctx, K.make_config(k, "DNA", strand=strand))
        b = K.CountIndex(ctx_k, K.make_config(k, "DNA", strand=strand))
        a.build(data); b.build(data)
        ka, ca = export_sorted(a); kb, cb = export_sorted(b)
        same = ka.shape == kb.shape and bool((ka == kb).all()) and bool((ca == cb).all())
        # a second build into the filled index (the scratch-index path)
        more = np.asarray(K.synth_fastq(seed=5000 + r, genome_len=max(genome, 1000), n_reads=max(n_reads // 5, 1000)))
        a.build(more); b.build(more)
        ka, ca = export_sorted(a); kb, cb = export_sorted(b)
        same2 = ka.shape == kb.shape and bool((ka == kb).all()) and bool((ca == cb).all())
        print("round %2d: %8d reads over %10d bp, k = %2d %-9s  %9d entries  %s %s" % (r, n_reads, genome, k, strand, ka.shape[0], "same" if same else "DIFFERENT", "same" if same2 else "DIFFERENT"), flush=True)
        a.close(); b.close()
        if not (same and same2):
            sys.exit(1)
    # the de Bruijn node build through super-k-mer records against the tuple path
    os.environ["KMI_DBG_SUPERKMER"] = "0"
    ctx_t = K.Context(0)
    del os.environ["KMI_DBG_SUPERKMER"]
    for r in range(max(rounds // 2, 1)):
        n_reads = int(rng.integers(20_000, 800_000))
        genome = int(n_reads * 150 / rng.choice([1, 6, 12, 40]))
        k = int(rng.integers(17, 33))
        data = np.asarray(K.synth_fastq(seed=9000 + r, genome_len=max(genome, 1000), n_reads=n_reads))
        a = K.DeBruijnNodes(ctx, K.make_config(k)); b = K.DeBruijnNodes(ctx_t, K.make_config(k))
        a.build(data); b.build(data)
        ka, va = a.to_vector(); kb, vb = b.to_vector()
        oa, ob = np.argsort(ka[:, 0], kind="stable"), np.argsort(kb[:, 0], kind="stable")
        same = ka.shape == kb.shape and bool((ka[oa] == kb[ob]).all()) and bool((np.asarray(va)[oa] == np.asarray(vb)[ob]).all())
        print("de Bruijn round %2d: %8d reads over %10d bp, k = %2d  %9d nodes  %s" % (r, n_reads, genome, k, ka.shape[0], "same" if same else "DIFFERENT"), flush=True)
        a.close(); b.close()
        if not same:
            sys.exit(1)
    # the whole-line fine pass of the record partitions against the plain form (position, position + quality, three-word keys)
    os.environ["KMI_LINES_P2"] = "0"
    ctx_p = K.Context(0)
    del os.environ["KMI_LINES_P2"]
    for r in range(max(rounds // 2, 1)):
        n_reads = int(rng.integers(5_000, 300_000))
        k, alpha, kind = [(31, "DNA", "position"), (31, "DNA", "posqual"), (63, "DNA5", "position"), (21, "DNA", "posqual"), (40, "DNA", "position")][r % 5]
        data = np.asarray(K.synth_fastq(seed=12000 + r, genome_len=int(n_reads * 150 / rng.choice([1, 12])), n_reads=n_reads))
        a = K.PositionIndex(ctx, K.make_config(k, alpha, index_kind=kind)); b = K.PositionIndex(ctx_p, K.make_config(k, alpha, index_kind=kind))
        a.build(data); b.build(data)
        ka, va = a.to_vector(); kb, vb = b.to_vector()
        ra = np.concatenate([ka, va.reshape(ka.shape[0], -1)], axis=1); rb = np.concatenate([kb, vb.reshape(kb.shape[0], -1)], axis=1)
        ra = ra[np.lexsort(ra.T[::-1])]; rb = rb[np.lexsort(rb.T[::-1])]
        same = ra.shape == rb.shape and bool((ra == rb).all())
        print("position round %2d: %7d reads, k = %2d %-4s %-8s %9d tuples  %s" % (r, n_reads, k, alpha, kind, ka.shape[0], "same" if same else "DIFFERENT"), flush=True)
        a.close(); b.close()
        if not same:
            sys.exit(1)
    print("all rounds agree")


main()
