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


def _synth_fasta(total_bp, chrom_bp, seed):
    """one '>chr<i>' header per chrom_bp bases, 80 bases per line, bases i.i.d. uniform over ACGT,
    0.1 % of the positions inside N runs of length 100. Returns (bytes as uint8 array, [(first byte of the sequence text,
    bases) per record])."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts, recs, off = [], [], 0
    for c in range(total_bp // chrom_bp):
        seq = lut[rng.integers(0, 4, size=chrom_bp, dtype=np.uint8)]
        for st in rng.integers(0, chrom_bp - 100, size=chrom_bp // 100_000):      # 0.1 % of positions in runs of 100
            seq[st:st + 100] = ord("N")
        hdr = np.frombuffer((">chr%d\n" % c).encode(), dtype=np.uint8)
        lines = np.empty((chrom_bp // 80, 81), dtype=np.uint8)
        lines[:, :80] = seq.reshape(-1, 80)
        lines[:, 80] = 10
        parts += [hdr, lines.reshape(-1)]
        recs.append((off + hdr.size, seq))
        off += hdr.size + lines.size
    return np.concatenate(parts), recs


def test_config4_full_size_fasta_position_index_properties():
    """BASELINE.json configs[3] (SURVEY config 4): k = 63 DNA5 (three-word Kmer, N is a character of its own) PositionIndex with
    LongSequenceKmerId values on a 1 Gbp synthetic FASTA, ten records, 80-column lines, N runs. Size-independent properties:
    every window of every record is indexed exactly once (conservation); sampled windows -- plain ones, ones across an N
    run, the first and the last of a record -- are found with exactly the id the reference assigns (file offset of the
    window's first base | record index << 40, sequence.hpp:231-296), via the oracle's k-mer for that text; erase takes them out."""
    import os
    import kmerind_amd as Kx
    total_bp = int(os.environ.get("KMI_TEST_FASTA_BP", 1_000_000_000))
    chrom_bp, k = 100_000_000, 63
    data, recs = _synth_fasta(total_bp, min(chrom_bp, total_bp), seed=4)
    ctx = Kx.Context(0)
    cfg = Kx.make_config(k, "DNA5", strand="canonical", index_kind="position", seq_format="fasta")
    d = ctx.alloc(data.size + 64)
    ctx.to_device(d, data)
    idx = Kx.PositionIndex(ctx, cfg)
    idx.build_device(d, data.size)
    ctx.free(d)
    n_windows = sum(seq.size - k + 1 for _, seq in recs)
    assert idx.local_size() == n_windows                               # conservation: one tuple per window, N included
    s = orc.kspec(k, orc.DNA5)
    rng = np.random.default_rng(44)
    texts, ids = [], []
    for ri, (start, seq) in enumerate(recs):
        n_pos = np.flatnonzero(seq == ord("N"))
        picks = list(rng.integers(0, seq.size - k + 1, size=40)) + [0, seq.size - k]
        picks += [int(n_pos[0]) - 30, int(n_pos[-1]) - 5]              # windows that hold part of an N run
        for o in picks:
            o = int(min(max(o, 0), seq.size - k))
            texts.append(seq[o:o + k].tobytes())
            ids.append((ri << 40) | (start + o + o // 80))             # LongSequenceKmerId: record index, file offset of the first base
    q = np.concatenate([orc.kmers_from_string(s, t) for t in texts])
    fk, fv = idx.find(q)
    canon = orc.canonical(s, q)
    got = {}
    for key, v in zip(fk.tolist(), fv[:, 0].tolist()):
        got.setdefault(tuple(key), set()).add(int(v))
    for key, want in zip(canon.tolist(), ids):
        assert want in got.get(tuple(key), ()), (key, want)
    ck, cv = idx.count(q)
    assert ck.shape[0] == np.unique(canon, axis=0).shape[0] and (cv >= 1).all()
    n_hit = sum(len(v) for v in got.values())
    assert idx.erase(q) == n_hit and idx.local_size() == n_windows - n_hit
    idx.close()
    ctx.close()
