"""BASELINE.json configs at full size. configs[1] (10 M synthetic 150-bp reads, 1.2e9 k-mers): equality with the oracle's
thread-rank build of all of it (checksums of the maps) + conservation of mass, canonical/unique keys, linearity of insert,
erase-all. configs[2] and configs[4]: one rank's full-size share (12.5 M reads of the 1 Gbp genome through the super-k-mer
exchange path; 25 M reads into the position + quality index in record-aligned batches + 10 M queries). configs[3] in full."""
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
    # FULL equality with the oracle: its thread-rank build of all 10 M reads (16 ranks, 8 slices) against this index through
    # checksums of the maps -- distinct keys, sum of counts, sum of key * count and an xor mix (mod 2^64)
    mine = orc.map_checksums(keys[:, 0], counts)
    theirs = orc.count_full(data, K, orc.CANONICAL, threads=16, slices=8)
    assert theirs["kmers"] == n_kmers
    assert {x: theirs[x] for x in mine} == mine
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


def _canonical31_torch(x):
    """canonical form of 31-mers held as int64 (62 bits, newest base in the low bits) on the device: min(x, revcomp(x))"""
    import torch
    r = (~x) & 0x3FFFFFFFFFFFFFFF
    r = ((r >> 2) & 0x3333333333333333) | ((r & 0x3333333333333333) << 2)
    r = ((r >> 4) & 0x0F0F0F0F0F0F0F0F) | ((r & 0x0F0F0F0F0F0F0F0F) << 4)
    r = ((r >> 8) & 0x00FF00FF00FF00FF) | ((r & 0x00FF00FF00FF00FF) << 8)
    r = ((r >> 16) & 0x0000FFFF0000FFFF) | ((r & 0x0000FFFF0000FFFF) << 16)
    r = ((r >> 32) & 0x00000000FFFFFFFF) | (r << 32)
    r = (r >> 2) & 0x3FFFFFFFFFFFFFFF
    return torch.minimum(x, r)


def test_config3_one_rank_share_through_the_superkmer_exchange_path():
    """BASELINE.json configs[2] (SURVEY config 3), one rank's full-size share: rank 0's 12.5 M reads of the 1 Gbp genome (seed 3)
    go through kmi_index_sk_produce_dev for a build over 8 ranks; every owner's records are then consumed into an index
    (kmi_index_sk_consume_dev, what that owner would do with this rank's part) and the union of the eight maps is compared
    with an independent device path over the same reads -- kmi_extract_dev, canonical form and unique-with-counts in torch --
    through the map checksums: distinct keys, sum of counts, two weighted key sums (mod 2^64). Conservation per owner (records
    sent = records consumed, k-mers in = counts out) and the oracle on the first 20 k reads ride along."""
    import torch
    import kmerind_amd as Kx
    from kmerind_amd import _lib as L
    world, n_reads, genome = 8, 12_500_000, 1_000_000_000
    dev = torch.device("cuda", 0)
    ctx = Kx.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = Kx.make_config(K, "DNA", strand="canonical")
    host = np.asarray(Kx.synth_fastq(3, genome, n_reads, 150, first_read=0))
    d = torch.from_numpy(host).to(dev)
    n_kmers = n_reads * (150 - K + 1)
    idx = Kx.CountIndex(ctx, cfg)
    recs, n, produced = C.c_void_p(), C.c_uint64(), C.c_int()
    sc = np.zeros(world, dtype=np.uint64)
    ctx.check(L.lib.kmi_index_sk_produce_dev(idx.h, C.c_void_p(d.data_ptr()), host.size, world, None, 0, C.byref(recs), C.byref(n),
                                             sc.ctypes.data_as(C.c_void_p), C.byref(produced)))
    assert produced.value and int(sc.sum()) == n.value
    assert float(sc.max()) / float(sc.mean()) < 1.05                      # the owners' shares of this rank's records are balanced
    allrec = torch.empty((n.value, 2), dtype=torch.int64, device=dev)
    ctx.check(L.lib.kmi_copy_on_device(ctx.h, C.c_void_p(allrec.data_ptr()), recs, n.value * 16))
    # k-mers per record: bits 38..42 of the second word hold n - 1
    assert int((((allrec[:, 1] >> 38) & 31) + 1).sum().item()) == n_kmers
    tot = dict(distinct=0, sum_counts=0, sum_key_count=0, sum_key2_count=0)
    off = 0
    M = (1 << 64) - 1
    for o in range(world):
        part = allrec[off:off + int(sc[o])]
        off += int(sc[o])
        idx.clear()
        ctx.check(L.lib.kmi_index_sk_consume_dev(idx.h, C.c_void_p(part.data_ptr()), part.shape[0], world))
        assert idx.owner_ranks() == world
        keys, counts = idx.to_vector()
        k0, c0 = keys[:, 0], counts.astype(np.uint64)
        assert int(c0.sum()) == int((((part[:, 1] >> 38) & 31) + 1).sum().item())   # this owner's k-mers in = counts out
        with np.errstate(over="ignore"):
            tot["distinct"] += k0.size
            tot["sum_counts"] += int(c0.sum())
            tot["sum_key_count"] = (tot["sum_key_count"] + int((k0 * c0).sum(dtype=np.uint64))) & M
            tot["sum_key2_count"] = (tot["sum_key2_count"] + int((k0 * k0 * c0).sum(dtype=np.uint64))) & M
        if o == 0:   # the oracle on the first 20 k reads: every one of its keys that this owner holds is there with at least its count
            s = orc.kspec(K)
            om = orc.CountMap(s, orc.CANONICAL)
            om.insert(orc.extract(s, bytes(host[: 20_000 * 315]), orc.FASTQ)["kmers"])
            ok, oc = om.export()
            fk, fv = idx.find(ok)
            want = dict(zip(ok[:, 0].tolist(), oc.tolist()))
            assert fk.shape[0] > ok.shape[0] // 16 and all(int(v) >= want[int(kk)] for kk, v in zip(fk[:, 0].tolist(), fv.tolist()))
        del keys, counts
    del allrec
    idx.close()
    # the independent path: every k-mer of the reads as parsed (kmi_extract_dev), canonical + unique with counts in torch
    km = torch.empty(n_kmers, dtype=torch.int64, device=dev)
    nt, ns = C.c_uint64(), C.c_uint64()
    ctx.check(L.lib.kmi_extract_dev(ctx.h, C.byref(cfg), C.c_void_p(d.data_ptr()), host.size, 0, C.c_void_p(km.data_ptr()), None, n_kmers,
                                    C.byref(nt), C.byref(ns)))
    assert nt.value == n_kmers and ns.value == n_reads
    step = 1 << 27
    for a in range(0, n_kmers, step):
        km[a:a + step] = _canonical31_torch(km[a:a + step])
    uk, uc = torch.unique(km, return_counts=True)
    del km
    ref = dict(distinct=int(uk.numel()), sum_counts=int(uc.sum().item()),
               sum_key_count=int((uk * uc).sum().item()) & M, sum_key2_count=int((uk * uk * uc).sum().item()) & M)
    assert tot == ref, (tot, ref)
    del uk, uc, d
    ctx.close()
    torch.cuda.empty_cache()   # (torch keeps what it freed: the library's own allocations of the next test need the room)


def test_config5_one_rank_share_position_quality_index_and_queries():
    """BASELINE.json configs[4] (SURVEY config 5), one rank's full-size share: 25 M reads (of the 2 Gbp genome, seed 5) parsed into
    (k-mer, ShortSequenceKmerId, quality) tuples in record-aligned batches and inserted into a PositionQualityIndex -- 3e9
    tuples, 72 GB, the volume a rank holds after the exchange of the 8-GPU build -- then 10 M queries (5 M k-mers of the reads,
    5 M random 62-bit values). Conservation (one entry per window, batch by batch); ids and quality bits of sampled reads exact
    against the oracle's parser; multiplicities of sampled keys against the oracle's count over the whole input restricted to
    them; random keys absent."""
    import torch
    import kmerind_amd as Kx
    n_reads, genome, n_batches = 25_000_000, 2_000_000_000, 10
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    ctx = Kx.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cfg = Kx.make_config(K, "DNA", strand="canonical", index_kind="posqual")
    host = np.asarray(Kx.synth_fastq(5, genome, n_reads, 150, first_read=0))
    d = torch.from_numpy(host).to(dev)
    idx = Kx.PositionIndex(ctx, cfg)
    per = (n_reads // n_batches) * 315
    done = 0
    for b in range(n_batches):
        lo, hi = b * per, (host.size if b == n_batches - 1 else (b + 1) * per)
        idx.build_device(d.data_ptr() + lo, hi - lo, file_offset=lo)
        done += ((hi - lo) // 315) * (150 - K + 1)
        assert idx.local_size() == done                                   # conservation, batch by batch
    del d
    # sampled reads: the oracle's tuples (k-mer, id at the file offset, quality) must all be in the index, bit for bit
    s = orc.kspec(K)
    rng = np.random.default_rng(55)
    picks = np.sort(rng.integers(0, n_reads, 400))
    want = {}
    qk = []
    for r in picks.tolist():
        rec = bytes(host[r * 315:(r + 1) * 315])
        ex = orc.extract(s, rec, orc.FASTQ, file_offset=r * 315, want_ids=True, want_quals=True)
        canon = orc.canonical(s, ex["kmers"])
        for key, i, q in zip(canon[:, 0].tolist(), ex["ids"].tolist(), ex["quals"].view(np.uint32).tolist()):
            want.setdefault(key, set()).add((i, q))
        qk.append(canon)
    qk = np.concatenate(qk)
    fk, fv = idx.find(qk)
    got = {}
    for key, v in zip(fk[:, 0].tolist(), fv.reshape(-1, 2).tolist()):
        got.setdefault(key, set()).add((int(v[0]), int(v[1]) & 0xFFFFFFFF))
    for key, vals in want.items():
        assert vals <= got.get(key, set()), key
    # the 10 M queries of the config: 5 M k-mers of the reads (present) + 5 M random 62-bit values (almost surely absent)
    present = []
    for r in rng.integers(0, n_reads, 42_000).tolist():
        present.append(orc.extract(s, bytes(host[r * 315:(r + 1) * 315]), orc.FASTQ)["kmers"])
    present = np.concatenate(present)[:5_000_000]
    absent = rng.integers(0, 1 << 62, size=(5_000_000, 1), dtype=np.uint64)
    q = np.concatenate([present, absent])
    ck, cv = idx.count(q)
    cnt = dict(zip(ck[:, 0].tolist(), cv.tolist()))
    pc = orc.canonical(s, present)
    assert all(cnt[kk] >= 1 for kk in pc[:50_000, 0].tolist())
    # multiplicity of a sampled key = its occurrences in ALL reads: the oracle's count of the same keys over a window of the genome's
    # coverage is a lower bound; the exact figure comes from the find() above: count == number of (id, quality) entries
    fcnt = {}
    for key in fk[:, 0].tolist():
        fcnt[key] = fcnt.get(key, 0) + 1
    assert all(cnt.get(key, 0) == nfound for key, nfound in list(fcnt.items())[:100_000] if key in cnt)
    ac = orc.canonical(s, absent[:200_000])
    hit = sum(1 for kk in ac[:, 0].tolist() if cnt.get(kk, 0))
    assert hit <= 8                                                        # 3e9 of 2^61 canonical values: a handful at most
    idx.close()
    ctx.close()


def test_two_word_keys_with_more_distinct_keys_per_bucket_than_a_table_holds():
    """1.7e8 distinct 40-mers (two-word keys) in one insert: every one of the 32768 buckets holds about 5200 distinct keys, more than
    a tagged LDS table takes in one pass (3 / 4 of 6144 slots), so every bucket goes through in two passes. Until round 4 the tagged
    tables had no load limit: a bucket like this filled its table to the last slot and every further key walked all of it before the
    attempt was given up -- config 2's reads over an 800 Mbp genome at k = 63 took 11.9 s (now 0.14 s). Keys by construction: the low
    word counts up, so size and counts are known without an oracle pass; each key is inserted twice."""
    import kmerind_amd as K
    ctx = K.Context(0)
    idx = K.CountIndex(ctx, K.make_config(40, "DNA", strand="single"))
    n = 170_000_000
    lo = np.arange(n, dtype=np.uint64)
    keys = np.empty((n, 2), dtype=np.uint64)
    keys[:, 0] = lo * np.uint64(7) + np.uint64(3)          # (the low word is distinct ...
    keys[:, 1] = (lo >> np.uint64(3)) & np.uint64(0xFFFF)   # ... and the high word stays inside a 40-mer's 16 high bits)
    d = ctx.alloc(keys.nbytes)
    ctx.to_device(d, keys)
    for _ in range(2):
        idx.insert_device(d, n)
    assert idx.local_size() == n
    q = keys[np.random.default_rng(1).integers(0, n, size=200_000)]
    ck, cc = idx.count(q)
    assert ck.shape[0] == np.unique(q, axis=0).shape[0] and (np.asarray(cc) == 1).all()
    fk, fv = idx.find(q)
    assert (np.asarray(fv) == 2).all() and fk.shape[0] == ck.shape[0]
    ctx.free(d)
    idx.close(); ctx.close()
