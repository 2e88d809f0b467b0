"""BASELINE.json configs[1] at full size (10 M synthetic 150-bp reads, 1.2e9 k-mers) checked
through size-independent properties: conservation of mass, canonical/unique keys, linearity of
insert, agreement with the oracle on a sampled sub-range, erase-all."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu

N_READS = 10_000_000
GENOME = 100_000_000
K = 31


def test_config2_full_size_properties():
    import kmerind_amd as Kx
    ctx = Kx.Context(0)
    cfg = Kx.make_config(K, "DNA", strand="canonical")
    data = Kx.synth_fastq(seed=2, genome_len=GENOME, n_reads=N_READS)
    n_kmers = N_READS * (150 - K + 1)
    dbytes = ctx.alloc(data.nbytes)
    ctx.to_device(dbytes, data)
    idx = Kx.CountIndex(ctx, cfg)
    idx.build_device(dbytes, data.nbytes)
    keys, counts = idx.to_vector()
    # conservation: every parsed k-mer is counted exactly once
    assert int(counts.astype(np.uint64).sum()) == n_kmers
    # keys are unique and canonical
    assert np.unique(keys[:, 0]).size == keys.shape[0]
    sample = keys[:: max(1, keys.shape[0] // 200_000)]
    assert (ctx.canonical(cfg, sample) == sample).all()
    # checksum of checksums stays the same when the same reads arrive as two insert() batches
    chk = int((keys[:, 0] * counts.astype(np.uint64)).sum(dtype=np.uint64))
    n_distinct = keys.shape[0]
    del keys, counts
    # the oracle on the first 20k reads: all of its keys are present with counts >= its counts
    s = orc.kspec(K)
    head = bytes(data[: 20_000 * 315])
    ex = orc.extract(s, head, orc.FASTQ)
    om = orc.CountMap(s, orc.CANONICAL)
    om.insert(ex["kmers"])
    ok, oc = om.export()
    fk, fv = idx.find(ok)
    assert fk.shape[0] == ok.shape[0]
    a, b = orc.sorted_pairs(fk, fv), orc.sorted_pairs(ok, oc.astype(np.uint64))
    assert (a[0] == b[0]).all() and (a[1] >= b[1]).all()
    ck, cv = idx.count(ex["kmers"][:500_000])
    assert int(cv.sum()) == ck.shape[0]
    # linearity: inserting the same reads again doubles every count
    half = (N_READS // 2) * 315
    idx2 = Kx.CountIndex(ctx, cfg)
    idx2.build_device(dbytes, half)
    idx2.build_device(dbytes + half, data.nbytes - half)
    k2, c2 = idx2.to_vector()
    assert k2.shape[0] == n_distinct
    assert int((k2[:, 0] * c2.astype(np.uint64)).sum(dtype=np.uint64)) == chk
    idx2.build_device(dbytes, data.nbytes)
    k3, c3 = idx2.to_vector()
    assert k3.shape[0] == n_distinct and int(c3.astype(np.uint64).sum()) == 2 * n_kmers
    # erase everything
    assert idx2.erase(k3) == n_distinct
    assert idx2.local_size() == 0
    idx.close(); idx2.close()
    ctx.free(dbytes)
    ctx.close()
