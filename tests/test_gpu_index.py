"""GPU parity of the count index (insert / build / count / find / erase / to_vector) through
the C ABI against the CPU oracle's restatement of counting_unordered_map."""
import json
import os

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALPHA = {"DNA": orc.DNA, "DNA5": orc.DNA5, "DNA16": orc.DNA16}
STRAND = {"single": orc.SINGLE, "canonical": orc.CANONICAL, "bimolecule": orc.BIMOLECULE}


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _same_map(idx, omap):
    keys, counts = idx.to_vector()
    ok, oc = omap.export()
    assert keys.shape == ok.shape
    a = orc.sorted_pairs(keys, counts)
    b = orc.sorted_pairs(ok, oc)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()


def test_known_answers_count_index(ctx):
    import kmerind_amd as K
    ka = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))
    for case in ka["count_index"]:
        data = open(os.path.join(GOLD, "data", case["file"]), "rb").read()
        for strand, key in (("single", "min_single"), ("canonical", "min_canonical")):
            idx = K.CountIndex(ctx, K.make_config(case["k"], "DNA", strand=strand))
            idx.build(data)
            keys, counts = idx.to_vector()
            assert idx.local_size() == case["distinct"] == keys.shape[0]
            assert set(counts.tolist()) == {case["each_count"]}
            if key in case:
                assert int(keys[:, 0].min()) == int(case[key], 16)
            idx.close()


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (21, "DNA"), (32, "DNA"), (63, "DNA"), (63, "DNA5"), (21, "DNA5"), (96, "DNA")])
@pytest.mark.parametrize("strand", ["single", "canonical", "bimolecule"])
def test_build_matches_oracle(ctx, k, alpha, strand):
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    data = K.synth_fastq(seed=k, genome_len=4000, n_reads=1500)   # ~50x coverage: many repeats
    cfg = K.make_config(k, alpha, strand=strand)
    ex = orc.extract(s, data, orc.FASTQ)
    om = orc.CountMap(s, STRAND[strand])
    om.insert(ex["kmers"])
    idx = K.CountIndex(ctx, cfg)
    idx.build(data)
    _same_map(idx, om)
    # second batch through insert(): counts accumulate, new keys appear
    data2 = K.synth_fastq(seed=k, genome_len=4000, n_reads=700, first_read=1000)
    ex2 = orc.extract(s, data2, orc.FASTQ)
    om.insert(ex2["kmers"])
    idx.insert(ex2["kmers"])
    _same_map(idx, om)
    idx.close()


def test_queries_match_oracle(ctx):
    import kmerind_amd as K
    for k, alpha, strand in ((31, "DNA", "canonical"), (63, "DNA5", "canonical"), (31, "DNA", "single"), (63, "DNA", "bimolecule")):
        s = orc.kspec(k, ALPHA[alpha])
        rng = np.random.default_rng(k)
        data = K.synth_fastq(seed=9, genome_len=20000, n_reads=2000)
        ex = orc.extract(s, data, orc.FASTQ)
        om = orc.CountMap(s, STRAND[strand])
        om.insert(ex["kmers"])
        idx = K.CountIndex(ctx, K.make_config(k, alpha, strand=strand))
        idx.insert(ex["kmers"])
        present = ex["kmers"][rng.integers(0, ex["kmers"].shape[0], size=5000)]
        absent = orc.extract(s, K.synth_fastq(seed=77, genome_len=20000, n_reads=40), orc.FASTQ)["kmers"]
        q = np.concatenate([present, absent, present[:100]])
        for fn in ("count", "find"):
            gk, gv = getattr(idx, fn)(q)
            ok, ov = getattr(om, fn)(q)
            a, b = orc.sorted_pairs(gk, gv), orc.sorted_pairs(ok, ov.astype(np.uint64))
            assert a[0].shape == b[0].shape, (fn, k)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), (fn, k)
        # erase a subset, then the maps must still agree
        er = present[:2000]
        n_gpu = idx.erase(er)
        n_cpu = om.erase(er)
        assert n_gpu == n_cpu
        _same_map(idx, om)
        gk, gv = idx.count(q)
        ok, ov = om.count(q)
        a, b = orc.sorted_pairs(gk, gv), orc.sorted_pairs(ok, ov)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
        idx.close()


def test_empty_and_tiny_inputs(ctx):
    import kmerind_amd as K
    cfg = K.make_config(31, "DNA")
    idx = K.CountIndex(ctx, cfg)
    assert idx.local_size() == 0
    idx.insert(np.zeros((0, 1), dtype=np.uint64))
    idx.build(b"")
    k, v = idx.count(np.array([[5]], dtype=np.uint64))
    assert k.shape[0] == 1 and v.tolist() == [0]
    k, v = idx.find(np.array([[5]], dtype=np.uint64))
    assert k.shape[0] == 0
    assert idx.erase(np.array([[5]], dtype=np.uint64)) == 0
    idx.insert(np.array([[5], [5], [7]], dtype=np.uint64))
    keys, counts = idx.to_vector()
    s = orc.kspec(31)
    exp = orc.canonical(s, np.array([[5], [7]], dtype=np.uint64))
    a = orc.sorted_pairs(keys, counts)
    b = orc.sorted_pairs(exp, np.array([2, 1], dtype=np.uint32))
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.close()


def test_all_ones_key_and_heavy_hitter(ctx):
    """k=32 single strand: the all-T k-mer is the LDS table's empty sentinel; one key repeated 200k times"""
    import kmerind_amd as K
    cfg = K.make_config(32, "DNA", strand="single")
    idx = K.CountIndex(ctx, cfg)
    allT = np.uint64(0xFFFFFFFFFFFFFFFF)
    rng = np.random.default_rng(1)
    other = rng.integers(0, 1 << 62, size=5000, dtype=np.uint64)
    keys = np.concatenate([np.full(200_000, allT, dtype=np.uint64), other, np.full(3, np.uint64(12345))]).reshape(-1, 1)
    idx.insert(keys)
    s = orc.kspec(32)
    om = orc.CountMap(s, orc.SINGLE)
    om.insert(keys)
    _same_map(idx, om)
    gk, gv = idx.find(np.array([[allT], [12345], [999]], dtype=np.uint64))
    got = dict(zip(gk[:, 0].tolist(), gv.tolist()))
    assert got == {int(allT): 200_000, 12345: 3}
    idx.close()


def test_many_distinct_keys_multi_pass_buckets(ctx):
    """3M distinct keys -> ~92 per fine bucket is easy; force table overflow with 40M? no: use a
    skewed set where one fine bucket gets > table capacity distinct keys."""
    import kmerind_amd as K
    from kmerind_amd import core
    cfg = K.make_config(31, "DNA", strand="single")
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 62, size=3_000_000, dtype=np.uint64).reshape(-1, 1)
    idx = K.CountIndex(ctx, cfg)
    idx.insert(keys)
    uk, uc = np.unique(keys[:, 0], return_counts=True)
    gk, gc = idx.to_vector()
    order = np.argsort(gk[:, 0])
    assert (gk[order, 0] == uk).all() and (gc[order] == uc).all()
    idx.close()


def test_route_matches_key_to_rank(ctx):
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    for k, alpha, strand, dh in ((31, "DNA", "canonical", "murmur"), (63, "DNA5", "bimolecule", "farm"), (21, "DNA", "single", "murmur")):
        cfg = K.make_config(k, alpha, strand=strand, dist_hash=dh)
        s = orc.kspec(k, ALPHA[alpha])
        ex = orc.extract(s, K.synth_fastq(seed=3, genome_len=50000, n_reads=800), orc.FASTQ)["kmers"]
        n, nw = ex.shape
        for p in (1, 2, 5, 8):
            din, dout = ctx.alloc(ex.nbytes), ctx.alloc(ex.nbytes)
            ctx.to_device(din, ex)
            counts = np.zeros(p, dtype=np.uint64)
            ctx.check(L.lib.kmi_route_dev(ctx.h, C.byref(cfg), C.c_void_p(din), n, p, C.c_void_p(dout),
                                          counts.ctypes.data_as(C.c_void_p)))
            out = np.zeros_like(ex)
            ctx.to_host(out, dout)
            ctx.free(din); ctx.free(dout)
            # reference: transform_input, then rank = DistHash(DistTrans(key)) % p
            tk = ex if strand == "single" else orc.canonical(s, ex)
            ranks = orc.key_to_rank(s, orc.MURMUR if dh == "murmur" else orc.FARM, STRAND[strand], tk, p)
            assert counts.tolist() == np.bincount(ranks, minlength=p).tolist()
            off = 0
            for r in range(p):
                seg = out[off:off + int(counts[r])]
                exp = tk[ranks == r]
                a = seg[np.lexsort([seg[:, w] for w in range(nw)])]
                b = exp[np.lexsort([exp[:, w] for w in range(nw)])]
                assert (a == b).all()
                off += int(counts[r])


def test_extract_route_fused_matches_extract_then_key_to_rank(ctx):
    """kmi_extract_route_dev = read_file + the bucketing half of imxx::distribute: per destination rank the same
    multiset of transformed keys as oracle extract -> transform -> KeyToRank; then the whole multi-rank build replayed
    on one device (every rank's reads routed, every destination's index built) equals the oracle's single map."""
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    for k, alpha, strand, dh, p in ((31, "DNA", "canonical", "murmur", 8), (21, "DNA", "single", "farm", 3),
                                    (63, "DNA5", "canonical", "murmur", 2), (31, "DNA", "bimolecule", "identity", 5)):
        cfg = K.make_config(k, alpha, strand=strand, dist_hash=dh)
        s = orc.kspec(k, ALPHA[alpha])
        hashes = {"murmur": orc.MURMUR, "farm": orc.FARM, "identity": orc.IDENTITY}
        per_dest = [[] for _ in range(p)]
        all_kmers = []
        for r in range(p):                                    # rank r parses its own reads
            data = np.asarray(K.synth_fastq(seed=7, genome_len=60000, n_reads=500, first_read=r * 500))
            ex = orc.extract(s, data.tobytes(), orc.FASTQ)["kmers"]
            all_kmers.append(ex)
            n, nw = ex.shape
            dbytes, dout = ctx.alloc(data.nbytes), ctx.alloc(ex.nbytes + 64)
            ctx.to_device(dbytes, data)
            counts = np.zeros(p, dtype=np.uint64)
            nt, ns = C.c_uint64(), C.c_uint64()
            ctx.check(L.lib.kmi_extract_route_dev(ctx.h, C.byref(cfg), C.c_void_p(dbytes), data.nbytes, p, C.c_void_p(dout), n,
                                                  C.byref(nt), C.byref(ns), counts.ctypes.data_as(C.c_void_p)))
            assert nt.value == n and ns.value == 500
            out = np.zeros_like(ex)
            ctx.to_host(out, dout)
            ctx.free(dbytes); ctx.free(dout)
            tk = ex if strand == "single" else orc.canonical(s, ex)
            ranks = orc.key_to_rank(s, hashes[dh], STRAND[strand], tk, p)
            assert counts.tolist() == np.bincount(ranks, minlength=p).tolist()
            off = 0
            for d in range(p):
                seg = out[off:off + int(counts[d])]
                exp = tk[ranks == d]
                assert (seg[np.lexsort([seg[:, w] for w in range(nw)])] == exp[np.lexsort([exp[:, w] for w in range(nw)])]).all()
                per_dest[d].append(seg)
                off += int(counts[d])
        m = orc.CountMap(s, STRAND[strand])
        m.insert(np.concatenate(all_kmers))
        got_k, got_c = [], []
        for d in range(p):                                    # destination d inserts what it received
            idx = K.CountIndex(ctx, cfg)
            idx.insert(np.concatenate(per_dest[d]))
            kk, cc = idx.to_vector()
            got_k.append(kk); got_c.append(cc)
            idx.close()
        gk, gc = np.concatenate(got_k), np.concatenate(got_c)
        ek, ec = m.export()
        assert gk.shape == ek.shape
        go, eo = np.lexsort([gk[:, w] for w in range(gk.shape[1])]), np.lexsort([ek[:, w] for w in range(ek.shape[1])])
        assert (gk[go] == ek[eo]).all() and (gc[go] == ec[eo]).all()


def test_combine_first_split_and_merge_replayed_on_one_device(ctx):
    """The N > 1 count build, combine-first (kmi_index_split_by_rank_dev / kmi_index_merge_parts_dev): every rank's local
    index split by KeyToRank -- message d holds exactly the entries whose key maps to rank d, ordered by placement
    bucket with the advertised per-bucket counts -- and every destination's merge of the p parts it receives, replayed
    on one device, equal the oracle's single map restricted to the keys of that rank. A second round merges into the
    non-empty index (the index grows incrementally, like repeated Index::insert)."""
    import kmerind_amd as K
    nb = K.core.num_buckets()
    for k, alpha, strand, dh, p in ((31, "DNA", "canonical", "murmur", 8), (21, "DNA", "single", "farm", 3),
                                    (63, "DNA5", "canonical", "murmur", 2), (31, "DNA", "bimolecule", "identity", 5),
                                    (15, "DNA", "canonical", "murmur", 1)):
        cfg = K.make_config(k, alpha, strand=strand, dist_hash=dh)
        s = orc.kspec(k, ALPHA[alpha])
        hashes = {"murmur": orc.MURMUR, "farm": orc.FARM, "identity": orc.IDENTITY}
        dest = [K.CountIndex(ctx, cfg) for _ in range(p)]
        all_kmers = []
        for rnd in range(2):
            msgs = [[None] * p for _ in range(p)]                 # msgs[d][src] = (keys, counts, bucket counts)
            for r in range(p):                                    # rank r reduces its own reads, then splits
                data = np.asarray(K.synth_fastq(seed=11, genome_len=3000, n_reads=300, first_read=(rnd * p + r) * 300))
                all_kmers.append(orc.extract(s, data.tobytes(), orc.FASTQ)["kmers"])
                loc = K.CountIndex(ctx, cfg)
                loc.build(data)
                n, nw = loc.local_size(), loc.n_words
                lk, lc = loc.to_vector()
                dk, dc, db = ctx.alloc(n * nw * 8 + 64), ctx.alloc(n * 4 + 64), ctx.alloc(p * nb * 4)
                sc = loc.split_by_rank_device(p, dk, dc, n, db)
                ok, oc, ob = np.zeros((n, nw), np.uint64), np.zeros(n, np.uint32), np.zeros((p, nb), np.uint32)
                ctx.to_host(ok, dk); ctx.to_host(oc, dc); ctx.to_host(ob, db)
                ctx.free(dk); ctx.free(dc); ctx.free(db)
                assert loc.local_size() == n                      # the split leaves the index as it was
                loc.close()
                ranks = orc.key_to_rank(s, hashes[dh], STRAND[strand], lk, p)
                assert sc.tolist() == np.bincount(ranks, minlength=p).tolist() == ob.sum(axis=1).tolist()
                off = 0
                for d in range(p):
                    seg_k, seg_c = ok[off:off + int(sc[d])], oc[off:off + int(sc[d])]
                    a, b = orc.sorted_pairs(seg_k, seg_c), orc.sorted_pairs(lk[ranks == d], lc[ranks == d])
                    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
                    msgs[d][r] = (seg_k, seg_c, ob[d])
                    off += int(sc[d])
            for d in range(p):                                    # destination d merges the p parts it received
                mk = np.concatenate([m[0] for m in msgs[d]]); mc = np.concatenate([m[1] for m in msgs[d]])
                mb = np.stack([m[2] for m in msgs[d]])
                dk, dc, db = ctx.alloc(mk.nbytes + 64), ctx.alloc(mc.nbytes + 64), ctx.alloc(mb.nbytes)
                if mk.size:
                    ctx.to_device(dk, mk); ctx.to_device(dc, mc)
                ctx.to_device(db, np.ascontiguousarray(mb))
                dest[d].merge_parts_device(p, dk, dc, db)
                ctx.free(dk); ctx.free(dc); ctx.free(db)
        m = orc.CountMap(s, STRAND[strand])
        m.insert(np.concatenate(all_kmers))
        ek, ec = m.export()
        eranks = orc.key_to_rank(s, hashes[dh], STRAND[strand], ek, p)
        for d in range(p):
            gk, gc = dest[d].to_vector()
            a, b = orc.sorted_pairs(gk, gc), orc.sorted_pairs(ek[eranks == d], ec[eranks == d])
            assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
            dest[d].close()


def _keys_in_one_placement_bucket(n, bucket, hi_words=(1, 77, 1 << 20)):
    """white-box helper: one-word keys whose placement hash (kmi_device.h place_hash: three 32-bit multiply / xorshift
    rounds, a bijection of the low word for a fixed high word) falls into fine bucket `bucket` -- found by inverting it."""
    M = np.uint64(0xffffffff)

    def inv_mul(c):
        return np.uint64(pow(c, -1, 1 << 32))

    def unshift(h, s):
        x = h.copy()
        for _ in range(4):
            x = h ^ (x >> np.uint64(s))
        return x

    per = -(-n // len(hi_words))
    out = []
    for hi in hi_words:
        h = (np.uint64(bucket) << np.uint64(17)) | np.arange(per, dtype=np.uint64)      # final hash values in the bucket
        h = unshift(h, 16)
        h = (h * inv_mul(0x27D4EB2F)) & M
        h = unshift(h, 13)
        h = ((h * inv_mul(0xC2B2AE35)) & M) ^ np.uint64(hi)
        h = unshift(h, 15)
        lo = ((h * inv_mul(0x85EBCA6B)) & M) ^ np.uint64(0x9E3779B9)
        out.append((np.uint64(hi) << np.uint64(32)) | lo)
    return np.concatenate(out)[:n]


@pytest.mark.parametrize("n_hot", [30_000, 250_000])
def test_bucket_with_more_distinct_keys_than_the_lds_table_takes_more_passes(monkeypatch, n_hot):
    """n_hot distinct keys in ONE placement bucket (the one-word LDS table holds 9600 per pass): the reduce and the query
    kernels split the bucket into passes over disjoint key subsets (4 and about 35 of them; the pass count is estimated
    from how far the stream got when the table filled up); insert twice so the second merge meets the big bucket."""
    import kmerind_amd as K
    # (the per-bucket counts of the split below show PLACEMENT buckets only while the index keeps the placement-hash layout:
    # with the super-k-mer build enabled a split re-partitions a one-word DNA index by minimizer bucket first)
    monkeypatch.setenv("KMI_FUSED_PATH", "kmer")
    ctx = K.Context(0)
    cfg = K.make_config(31, "DNA", strand="single")
    rng = np.random.default_rng(8)
    hot = _keys_in_one_placement_bucket(n_hot, bucket=4242)
    assert np.unique(hot).size == hot.size and int(hot.max()) < (1 << 62)
    idx = K.CountIndex(ctx, cfg)
    ref = {}
    for rnd in range(2):
        reps = rng.integers(1, 4, size=hot.size)
        keys = np.concatenate([np.repeat(hot, reps), rng.integers(0, 1 << 62, size=200_000, dtype=np.uint64)])
        rng.shuffle(keys)
        idx.insert(keys.reshape(-1, 1))
        uk, uc = np.unique(keys, return_counts=True)
        for a, b in zip(uk.tolist(), uc.tolist()):
            ref[a] = ref.get(a, 0) + b
    gk, gc = idx.to_vector()
    assert gk.shape[0] == len(ref)
    order = np.argsort(gk[:, 0])
    rk = np.array(sorted(ref), dtype=np.uint64)
    assert (gk[order, 0] == rk).all() and (gc[order] == np.array([ref[int(a)] for a in rk], dtype=np.uint32)).all()
    # the hot keys really share a bucket: the per-bucket counts of a one-rank split show it
    n, nb = idx.local_size(), K.core.num_buckets()
    dk, dc, db = ctx.alloc(n * 8 + 64), ctx.alloc(n * 4 + 64), ctx.alloc(nb * 4)
    idx.split_by_rank_device(1, dk, dc, n, db)
    per_bucket = np.zeros(nb, np.uint32)
    ctx.to_host(per_bucket, db)
    ctx.free(dk); ctx.free(dc); ctx.free(db)
    assert int(per_bucket.sum()) == n and int(per_bucket[4242]) >= hot.size and int(np.delete(per_bucket, 4242).max()) < 200
    q = np.concatenate([hot[::3], rng.integers(0, 1 << 62, size=1000, dtype=np.uint64)]).reshape(-1, 1)
    fk, fc = idx.find(q)
    assert fk.shape[0] == hot[::3].size and all(ref[int(a)] == int(b) for a, b in zip(fk[:, 0], fc))
    idx.close()
    ctx.close()


def test_fit_check_of_the_first_reduce_attempt(ctx):
    """bucket_reduce, once a bucket of the launch has overflowed, lets later buckets decide after their first 8192 keys
    whether they fit one LDS table. Four large buckets in launch order: 30 k distinct keys (overflows, switches the
    check on), 12 k distinct keys twice over (the check ends the first attempt early), 5 k distinct keys four times
    over (passes the check), and 9 k distinct keys whose first 9 k stream positions are all different (the check
    errs on the safe side and takes passes where one table would have done). Counts stay exact in every case."""
    import kmerind_amd as K
    cfg = K.make_config(31, "DNA", strand="single")
    rng = np.random.default_rng(33)
    a = _keys_in_one_placement_bucket(30_000, bucket=64)
    b = _keys_in_one_placement_bucket(12_000, bucket=20_000)
    c = _keys_in_one_placement_bucket(5_000, bucket=26_000)
    d = _keys_in_one_placement_bucket(9_000, bucket=31_000)
    parts = [np.repeat(a, rng.integers(1, 3, size=a.size)), rng.permutation(np.tile(b, 2)), rng.permutation(np.tile(c, 4)),
             np.concatenate([d, rng.permutation(np.tile(d, 2))]), rng.integers(0, 1 << 62, size=100_000, dtype=np.uint64)]
    keys = np.concatenate(parts)                       # not shuffled: the stream order inside a bucket follows the input
    uk, uc = np.unique(keys, return_counts=True)
    for rnd in range(2):                               # the second insert merges into the filled buckets (no check there)
        idx = K.CountIndex(ctx, cfg) if rnd == 0 else idx
        idx.insert(keys.reshape(-1, 1))
        gk, gc = idx.to_vector()
        order = np.argsort(gk[:, 0])
        assert gk.shape[0] == uk.size and (gk[order, 0] == uk).all() and (gc[order] == uc * (rnd + 1)).all()
    idx.close()


@pytest.mark.parametrize("n_hot,p", [(30_000, 1), (30_000, 3)])
def test_merge_of_parts_with_more_distinct_keys_per_bucket_than_the_lds_table(ctx, n_hot, p):
    """combine-first at low coverage: a placement bucket of the parts holds more distinct keys than one LDS table takes, so
    bucket_merge runs in passes. Two source ranks' indexes (overlapping hot keys) are split for p destinations and every
    destination merges what it is sent, twice (the second merge meets the big bucket in the index)."""
    import kmerind_amd as K
    cfg = K.make_config(31, "DNA", strand="single")
    s = orc.kspec(31, ALPHA["DNA"])
    nb = K.core.num_buckets()
    rng = np.random.default_rng(21)
    hot = _keys_in_one_placement_bucket(n_hot, bucket=777)
    dest = [K.CountIndex(ctx, cfg) for _ in range(p)]
    ref = {}
    for rnd in range(2):
        msgs = [[None, None] for _ in range(p)]
        for src in range(2):
            mine = hot[rng.random(hot.size) < 0.8]
            keys = np.concatenate([np.repeat(mine, rng.integers(1, 4, size=mine.size)), rng.integers(0, 1 << 62, size=50_000, dtype=np.uint64)])
            uk, uc = np.unique(keys, return_counts=True)
            for a, b in zip(uk.tolist(), uc.tolist()):
                ref[a] = ref.get(a, 0) + b
            loc = K.CountIndex(ctx, cfg)
            loc.insert(keys.reshape(-1, 1))
            n = loc.local_size()
            dk, dc, db = ctx.alloc(n * 8 + 64), ctx.alloc(n * 4 + 64), ctx.alloc(p * nb * 4)
            sc = loc.split_by_rank_device(p, dk, dc, n, db)
            ok, oc, ob = np.zeros((n, 1), np.uint64), np.zeros(n, np.uint32), np.zeros((p, nb), np.uint32)
            ctx.to_host(ok, dk); ctx.to_host(oc, dc); ctx.to_host(ob, db)
            ctx.free(dk); ctx.free(dc); ctx.free(db)
            loc.close()
            off = 0
            for d in range(p):
                msgs[d][src] = (ok[off:off + int(sc[d])], oc[off:off + int(sc[d])], ob[d])
                off += int(sc[d])
        for d in range(p):
            mk = np.concatenate([m[0] for m in msgs[d]]); mc = np.concatenate([m[1] for m in msgs[d]])
            mb = np.ascontiguousarray(np.stack([m[2] for m in msgs[d]]))
            dk, dc, db = ctx.alloc(mk.nbytes + 64), ctx.alloc(mc.nbytes + 64), ctx.alloc(mb.nbytes)
            ctx.to_device(dk, mk); ctx.to_device(dc, mc); ctx.to_device(db, mb)
            dest[d].merge_parts_device(2, dk, dc, db)
            ctx.free(dk); ctx.free(dc); ctx.free(db)
    rk = np.array(sorted(ref), dtype=np.uint64)
    rc = np.array([ref[int(a)] for a in rk], dtype=np.uint32)
    ranks = orc.key_to_rank(s, orc.MURMUR, STRAND["single"], rk.reshape(-1, 1), p)
    for d in range(p):
        gk, gc = dest[d].to_vector()
        order = np.argsort(gk[:, 0])
        assert gk.shape[0] == int((ranks == d).sum())
        assert (gk[order, 0] == rk[ranks == d]).all() and (gc[order] == rc[ranks == d]).all()
        dest[d].close()


@pytest.mark.parametrize("dist_trans,code", [("lex_less", 1), ("xor_rev_comp", 2)])
def test_single_strand_dist_transforms(ctx, dist_trans, code):
    """DistTrans of SingleStrandHashMapParams (kmer_index.hpp:436-450; pDistTrans of BenchmarkKmerIndex.cpp:150-161): the key is
    stored as parsed but ranked by DistHash(lex_less(key)) or DistHash(key ^ revcomp(key)). KeyToRank on the host-facing op,
    the key router, the fused extract + route and the index split all follow the oracle; both strands of a k-mer land on
    one rank; the other strand models refuse the option."""
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    for k, alpha, dh, p in ((31, "DNA", "murmur", 8), (21, "DNA5", "farm", 3), (40, "DNA", "murmur", 5)):
        s = orc.kspec(k, ALPHA[alpha])
        cfg = K.make_config(k, alpha, strand="single", dist_hash=dh, dist_trans=dist_trans)
        data = np.asarray(K.synth_fastq(seed=13, genome_len=30000, n_reads=700))
        ex = orc.extract(s, data.tobytes(), orc.FASTQ)["kmers"]
        n, nw = ex.shape
        exp = orc.key_to_rank(s, orc.MURMUR if dh == "murmur" else orc.FARM, orc.SINGLE, ex, p, dist_trans=code)
        assert (ctx.key_to_rank(cfg, ex, p) == exp).all()
        assert (orc.key_to_rank(s, orc.MURMUR if dh == "murmur" else orc.FARM, orc.SINGLE, orc.revcomp(s, ex), p, dist_trans=code) == exp).all()
        # router on an existing key array, and the fused extract + route
        for fused in (False, True):
            counts = np.zeros(p, dtype=np.uint64)
            dout = ctx.alloc(ex.nbytes + 64)
            if fused:
                dbytes = ctx.alloc(data.nbytes)
                ctx.to_device(dbytes, data)
                nt, ns = C.c_uint64(), C.c_uint64()
                ctx.check(L.lib.kmi_extract_route_dev(ctx.h, C.byref(cfg), C.c_void_p(dbytes), data.nbytes, p, C.c_void_p(dout), n,
                                                      C.byref(nt), C.byref(ns), counts.ctypes.data_as(C.c_void_p)))
                ctx.free(dbytes)
            else:
                din = ctx.alloc(ex.nbytes)
                ctx.to_device(din, ex)
                ctx.check(L.lib.kmi_route_dev(ctx.h, C.byref(cfg), C.c_void_p(din), n, p, C.c_void_p(dout), counts.ctypes.data_as(C.c_void_p)))
                ctx.free(din)
            out = np.zeros_like(ex)
            ctx.to_host(out, dout)
            ctx.free(dout)
            assert counts.tolist() == np.bincount(exp, minlength=p).tolist()
            off = 0
            for r in range(p):
                seg, want = out[off:off + int(counts[r])], ex[exp == r]
                assert (seg[np.lexsort([seg[:, w] for w in range(nw)])] == want[np.lexsort([want[:, w] for w in range(nw)])]).all()
                off += int(counts[r])
        # combine-first split of a local index
        idx = K.CountIndex(ctx, cfg)
        idx.build(data)
        m, nb = idx.local_size(), K.core.num_buckets()
        dk, dc, db = ctx.alloc(m * nw * 8 + 64), ctx.alloc(m * 4 + 64), ctx.alloc(p * nb * 4)
        sc = idx.split_by_rank_device(p, dk, dc, m, db)
        ok = np.zeros((m, nw), np.uint64)
        ctx.to_host(ok, dk)
        ctx.free(dk); ctx.free(dc); ctx.free(db)
        lk, _ = idx.to_vector()
        lr = orc.key_to_rank(s, orc.MURMUR if dh == "murmur" else orc.FARM, orc.SINGLE, lk, p, dist_trans=code)
        assert sc.tolist() == np.bincount(lr, minlength=p).tolist()
        off = 0
        for r in range(p):
            seg = ok[off:off + int(sc[r])]
            assert (orc.key_to_rank(s, orc.MURMUR if dh == "murmur" else orc.FARM, orc.SINGLE, seg, p, dist_trans=code) == r).all()
            off += int(sc[r])
        idx.close()
    with pytest.raises((L.KmiError, ValueError)):
        ctx.key_to_rank(K.make_config(31, "DNA", strand="canonical", dist_trans=dist_trans), np.zeros((4, 1), np.uint64), 4)


def test_exists_is_one_byte_per_input_key_in_input_order(ctx):
    """densehash exists() (distributed_densehash_map.hpp:1465-1560)"""
    import kmerind_amd as K
    for strand in ("canonical", "single"):
        s = orc.kspec(21)
        cfg = K.make_config(21, "DNA", strand=strand)
        data = K.synth_fastq(seed=2, genome_len=3000, n_reads=400)
        ex = orc.extract(s, data, orc.FASTQ)["kmers"]
        idx = K.CountIndex(ctx, cfg)
        idx.build(data)
        rng = np.random.default_rng(3)
        q = np.concatenate([ex[rng.integers(0, ex.shape[0], size=50)], rng.integers(0, 1 << 42, size=(50, 1), dtype=np.uint64),
                            orc.revcomp(s, ex[:20])])
        q = q[rng.permutation(q.shape[0])]
        stored = set(int(x) for x in (ex[:, 0] if strand == "single" else orc.canonical(s, ex)[:, 0]))
        tq = q[:, 0] if strand == "single" else orc.canonical(s, q)[:, 0]
        want = np.array([int(x) in stored for x in tq], dtype=np.uint8)
        assert (idx.exists(q) == want).all()
        idx.close()


def test_insert_of_already_transformed_keys(ctx):
    """kmi_index_insert_transformed_dev: the local_insert half alone -- routed keys after the exchange are canonical already"""
    import kmerind_amd as K
    s = orc.kspec(31)
    cfg = K.make_config(31, "DNA", strand="canonical")
    ex = orc.extract(s, K.synth_fastq(seed=4, genome_len=5000, n_reads=800), orc.FASTQ)["kmers"]
    tk = orc.canonical(s, ex)
    d = ctx.alloc(tk.nbytes)
    ctx.to_device(d, tk)
    a, b = K.CountIndex(ctx, cfg), K.CountIndex(ctx, cfg)
    a.insert_device(d, tk.shape[0], transformed=True)
    b.insert(ex)
    ctx.free(d)
    pa, pb = orc.sorted_pairs(*a.to_vector()), orc.sorted_pairs(*b.to_vector())
    assert pa[0].shape == pb[0].shape and (pa[0] == pb[0]).all() and (pa[1] == pb[1]).all()
    a.close(); b.close()


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (21, "DNA"), (63, "DNA"), (21, "DNA5")])
@pytest.mark.parametrize("strand", ["single", "canonical"])
def test_weighted_pairs_insert_adds_the_values(ctx, k, alpha, strand):
    """Index::insert(vector<pair<Kmer, count>>): reduction_unordered_map::local_insert adds .second to the stored count
    (distributed_unordered_map.hpp:1603-1618, std::plus<uint32_t>, wrapping). Weights != 1, keys repeated inside the batch,
    keys already in the index, both strands of a k-mer in one batch (canonical model: one entry)."""
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    data = K.synth_fastq(seed=100 + k, genome_len=3000, n_reads=400)
    ex = orc.extract(s, data, orc.FASTQ)["kmers"]
    rng = np.random.default_rng(k)
    first = ex[: ex.shape[0] // 2]
    idx = K.CountIndex(ctx, K.make_config(k, alpha, strand=strand))
    idx.insert(first)                                               # weight 1 each
    pick = rng.integers(0, ex.shape[0], 5000)
    pk = np.concatenate([ex[pick], orc.revcomp(s, ex[pick[:500]])])
    pw = rng.integers(0, 1000, pk.shape[0]).astype(np.uint64)
    pw[:3] = np.uint64(0xFFFFFFFF)                                  # wraps like uint32_t addition
    idx.insert_pairs(pk, pw)
    # expectation: the oracle's map for the weight-1 part, plus the weights summed per transformed key (mod 2^32)
    om = orc.CountMap(s, STRAND[strand])
    om.insert(first)
    ok, oc = om.export()
    tk = pk if strand == "single" else orc.canonical(s, pk)
    exp = {tuple(kk): int(c) for kk, c in zip(ok.tolist(), oc.tolist())}
    for kk, w in zip(tk.tolist(), pw.tolist()):
        exp[tuple(kk)] = (exp.get(tuple(kk), 0) + int(w)) & 0xFFFFFFFF
    keys, counts = idx.to_vector()
    got = {tuple(kk): int(c) for kk, c in zip(keys.tolist(), counts.tolist())}
    assert len(got) == keys.shape[0]                                # distinct entries
    assert got == exp
    # pairs into an EMPTY index
    idx2 = K.CountIndex(ctx, K.make_config(k, alpha, strand=strand))
    idx2.insert_pairs(pk, pw)
    exp2 = {}
    for kk, w in zip(tk.tolist(), pw.tolist()):
        exp2[tuple(kk)] = (exp2.get(tuple(kk), 0) + int(w)) & 0xFFFFFFFF
    keys, counts = idx2.to_vector()
    assert {tuple(kk): int(c) for kk, c in zip(keys.tolist(), counts.tolist())} == exp2
    idx.close()
    idx2.close()


@pytest.mark.parametrize("k,strand", [(31, "canonical"), (32, "single"), (29, "canonical"), (28, "canonical"), (23, "single"), (21, "canonical"), (22, "single"), (20, "canonical"), (17, "canonical")])
def test_superkmer_build_ran_and_matches_oracle(ctx, k, strand):
    """The fused FASTQ build of one-word DNA k-mers (k >= 17) goes through super-k-mers (kmi_superkmer.h): the kernels must
    actually have run (no silent fall-back to the k-mer pipeline) and the index must be the oracle's, also when the build
    lands in an index that already holds entries."""
    import kmerind_amd as K
    s = orc.kspec(k, orc.DNA)
    data = K.synth_fastq(seed=7 * k, genome_len=60_000, n_reads=6_000)
    half = (data.shape[0] // 315 // 2) * 315
    idx = K.CountIndex(ctx, K.make_config(k, "DNA", strand=strand))
    ctx.profile(True)
    ctx.profile_reset()
    idx.build(data[:half])
    names = {p["name"] for p in ctx.profile_get() if p["launches"]}
    ctx.profile(False)
    assert {"sk_front", "sk_scatter", "sk_reduce"} <= names and "fastq_scatter" not in names and "fastq_scan_tiles" not in names, names
    om = orc.CountMap(s, STRAND[strand])
    om.insert(orc.extract(s, data[:half], orc.FASTQ)["kmers"])
    _same_map(idx, om)
    idx.build(data[half:])                                          # second build: merges into the existing entries
    om.insert(orc.extract(s, data[half:], orc.FASTQ)["kmers"])
    _same_map(idx, om)
    idx.close()


@pytest.mark.parametrize("k,strand,win", [(31, "canonical", 0), (32, "single", 16), (21, "canonical", 64), (27, "single", 0)])
def test_reduce2_opt_in_matches_oracle(monkeypatch, k, strand, win):
    """KMI_SK_REDUCE=2 puts sk_reduce2 (kmi_reduce2.h: a bucket's records counting-sorted by further minimizer-hash bits in LDS,
    wavefront-private tables over whole bins) ahead of sk_reduce, which then only redoes what sk_reduce2 declined. It is not the
    default (measured slower), but it must stay the oracle's map: buckets of a few thousand records, both strand models, the
    k = 32 key that equals the empty marker, several window sizes, and a second build into the existing entries."""
    import kmerind_amd as K
    monkeypatch.setenv("KMI_SK_REDUCE", "2")
    if win:
        monkeypatch.setenv("KMI_R2_WIN", str(win))
    c2 = K.Context(0)
    s = orc.kspec(k, orc.DNA)
    data = K.synth_fastq(seed=11 * k, genome_len=150_000, n_reads=30_000)
    if k == 32:   # reads of T only: the single-strand 32-mer of all ones is the table's empty marker
        rec = np.frombuffer(b"@x\n" + b"T" * 150 + b"\n+\n" + b"I" * 150 + b"\n", dtype=np.uint8)
        data = np.concatenate([data] + [rec] * 40)
    idx = K.CountIndex(c2, K.make_config(k, "DNA", strand=strand))
    c2.profile(True)
    c2.profile_reset()
    idx.build(data)
    names = {p["name"] for p in c2.profile_get() if p["launches"]}
    c2.profile(False)
    assert {"sk_reduce", "sk_reduce_redo"} <= names, names
    om = orc.CountMap(s, STRAND[strand])
    om.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
    _same_map(idx, om)
    more = K.synth_fastq(seed=13 * k, genome_len=150_000, n_reads=4_000)
    idx.build(more)
    om.insert(orc.extract(s, more, orc.FASTQ)["kmers"])
    _same_map(idx, om)
    idx.close()
    c2.close()


@pytest.mark.parametrize("k,strand", [(31, "canonical"), (23, "single")])
def test_sparse_index_after_superkmer_build(monkeypatch, k, strand):
    """A large super-k-mer build leaves the reduce's output buffers as the index (per-bucket start + count, unused slots behind
    every bucket: no compaction pass); count and find read that form, everything else makes it dense first. KMI_SPARSE_MIN=1
    forces the form on a small input: same answers before and after, same map, and the dense paths (insert, erase) still work."""
    import kmerind_amd as K
    monkeypatch.setenv("KMI_SPARSE_MIN", "1")
    c2 = K.Context(0)
    s = orc.kspec(k, orc.DNA)
    data = K.synth_fastq(seed=5 * k, genome_len=40_000, n_reads=5_000)
    kmers = orc.extract(s, data, orc.FASTQ)["kmers"]
    om = orc.CountMap(s, STRAND[strand])
    om.insert(kmers)
    idx = K.CountIndex(c2, K.make_config(k, "DNA", strand=strand))
    c2.profile(True)
    c2.profile_reset()
    idx.build(data)
    names = {p["name"] for p in c2.profile_get() if p["launches"]}
    assert "sk_reduce" in names and "bucket_compact" not in names, names      # no compaction inside the build
    assert idx.local_size() == om.export()[0].shape[0]
    rng = np.random.default_rng(k)
    q = np.concatenate([kmers[rng.integers(0, kmers.shape[0], 3000)], rng.integers(0, 1 << 62, (500, 1), dtype=np.uint64) & np.uint64((1 << (2 * k)) - 1 if k < 32 else 0xFFFFFFFFFFFFFFFF)])
    for fn, ofn in ((idx.count, om.count), (idx.find, om.find)):
        gk, gv = fn(q)
        ek, ev = ofn(q)
        a, b = orc.sorted_pairs(gk, gv), orc.sorted_pairs(ek, ev.astype(np.uint64))
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    _same_map(idx, om)                                                         # to_vector: the index is made dense here
    gk, gv = idx.count(q)                                                      # and answers the same afterwards
    ek, ev = om.count(q)
    a, b = orc.sorted_pairs(gk, gv), orc.sorted_pairs(ek, ev.astype(np.uint64))
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.clear()
    idx.build(data)                                                            # sparse again ...
    idx.insert(kmers[:1000])                                                   # ... and an insert into it
    om.insert(kmers[:1000])
    _same_map(idx, om)
    idx.clear()
    idx.build(data)
    idx.erase(kmers[:500])                                                     # erase from the sparse form
    om2 = orc.CountMap(s, STRAND[strand])
    om2.insert(kmers)
    om2.erase(kmers[:500])
    _same_map(idx, om2)
    c2.profile(False)
    idx.close()
    c2.close()


def test_fine_buckets_with_room_and_the_counted_fallback(ctx):
    """A fresh index's build gives every fine bucket of the super-k-mer records room (1.75 - 3.25 x its coarse bucket's mean share, by minimizer length, + 64)
    instead of counting the records first; input whose minimizers crowd a few buckets outgrows the room, which voids that attempt
    and repeats the back end with counted, exact offsets (sk_fine_count). Both outcomes are the oracle's map."""
    import kmerind_amd as K
    s = orc.kspec(31, orc.DNA)
    even = K.synth_fastq(seed=91, genome_len=80_000, n_reads=6_000)
    one = even[:315]
    crowded = np.concatenate([even[: 3_000 * 315]] + [one] * 3_000)           # 3000 copies of one read: ~10 buckets get them all
    for data, counted in ((even, False), (crowded, True)):
        idx = K.CountIndex(ctx, K.make_config(31, "DNA", strand="canonical"))
        ctx.profile(True)
        ctx.profile_reset()
        idx.build(data)
        names = {p["name"] for p in ctx.profile_get() if p["launches"]}
        ctx.profile(False)
        assert "sk_reduce" in names and ("sk_fine_count" in names) == counted, names
        om = orc.CountMap(s, STRAND["canonical"])
        om.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
        _same_map(idx, om)
        idx.close()


def _fastq_variant(rng, n_reads, kind):
    """FASTQ text the one-pass front end has to get right or hand over: ragged read lengths, long headers, CRLF, blank lines, lower
    case and N bases, quality lines that start with '@' or '+', no EOL at the end"""
    out = []
    for i in range(n_reads):
        L = int(rng.integers(20, 260)) if kind != "fixed" else 150
        seq = "".join("ACGT"[c] for c in rng.integers(0, 4, L))
        if kind == "lowerN" and i % 7 == 0:
            seq = seq[:L // 3] + "N" * 5 + seq[L // 3 + 5:].lower()
        qual = "".join(chr(int(c)) for c in rng.integers(35, 74, L))
        if i % 5 == 0: qual = "@" + qual[1:]
        if i % 11 == 0: qual = "+" + qual[1:]
        hdr = "@r%d %s" % (i, "x" * int(rng.integers(0, 40)))
        eol = "\r\n" if kind == "crlf" else "\n"
        rec = hdr + eol + seq + eol + "+" + (hdr[1:] if i % 3 == 0 else "") + eol + qual + eol
        if kind == "blank" and i % 13 == 0: rec += eol
        out.append(rec)
    txt = "".join(out)
    if kind == "noeol": txt = txt.rstrip("\r\n")
    return np.frombuffer(txt.encode(), dtype=np.uint8).copy()


@pytest.mark.parametrize("kind", ["fixed", "ragged", "crlf", "blank", "lowerN", "noeol"])
@pytest.mark.parametrize("k,strand", [(31, "canonical"), (21, "single")])
def test_one_pass_front_end_matches_general_path(monkeypatch, kind, k, strand):
    """kmi_front.h: EOL scan, line roles, packing and the minimizer walk in one pass, every wavefront on a byte range of its own
    whose first line index is inferred and chained afterwards. Ranges of 2 KB make hundreds of inferences on a small input;
    the index must be the oracle's whatever the text looks like (and the kernels must be the one-pass ones)."""
    import kmerind_amd as K
    monkeypatch.setenv("KMI_FRONT_MIN_RANGE", "2048")
    c2 = K.Context(0)
    rng = np.random.default_rng(hash((kind, k)) & 0xffff)
    data = _fastq_variant(rng, 1500, kind)
    s = orc.kspec(k, orc.DNA)
    idx = K.CountIndex(c2, K.make_config(k, "DNA", strand=strand))
    c2.profile(True)
    c2.profile_reset()
    idx.build(data)
    names = {p["name"] for p in c2.profile_get() if p["launches"]}
    c2.profile(False)
    assert "sk_front" in names and "fastq_scan_tiles" not in names, names
    om = orc.CountMap(s, STRAND[strand])
    om.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
    _same_map(idx, om)
    idx.close()
    c2.close()


@pytest.mark.parametrize("damage", ["no_at", "no_plus", "qual_short", "first_byte"])
def test_one_pass_front_end_hands_malformed_input_to_the_general_path(monkeypatch, damage):
    """Whatever the one-pass front end is not sure about goes to the general path, which words the error exactly as before."""
    import kmerind_amd as K
    monkeypatch.setenv("KMI_FRONT_MIN_RANGE", "2048")
    c2 = K.Context(0)
    rng = np.random.default_rng(3)
    txt = bytes(_fastq_variant(rng, 400, "fixed")).decode()
    lines = txt.split("\n")                      # four lines per record, no blank ones in this variant
    i = 4 * 250
    if damage == "no_at": lines[i] = "r" + lines[i][1:]
    if damage == "no_plus": lines[i + 2] = "-" + lines[i + 2][1:]
    if damage == "qual_short": lines[i + 3] = lines[i + 3][:-3]
    bad = "\n".join(lines)
    if damage == "first_byte": bad = "\n" + txt
    data = np.frombuffer(bad.encode(), dtype=np.uint8).copy()
    idx = K.CountIndex(c2, K.make_config(31, "DNA", strand="canonical"))
    with pytest.raises(K._lib.KmiError) as e1:
        idx.build(data)
    monkeypatch.setenv("KMI_FRONT", "general")
    c3 = K.Context(0)
    idx3 = K.CountIndex(c3, K.make_config(31, "DNA", strand="canonical"))
    with pytest.raises(K._lib.KmiError) as e3:
        idx3.build(data)
    assert str(e1.value) == str(e3.value)
    assert idx.local_size() == 0
    idx.close(); idx3.close()
    c2.close(); c3.close()


@pytest.mark.parametrize("k", [31, 40])
def test_saturating_counts_stop_at_the_ceiling(ctx, k):
    """sat_plus<uint32_t> (distributed_densehash_map.hpp:2903-2912) on the device: weighted pairs that push a key past 2^32 - 1 --
    within one call, and on top of a stored count -- leave it AT 2^32 - 1; the default reducer (std::plus) wraps"""
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    s = orc.kspec(k, orc.DNA)
    rng = np.random.default_rng(k)
    keys = orc.canonical(s, rng.integers(0, 1 << 62, (200, s.n_words), dtype=np.uint64) & np.uint64((1 << 62) - 1 if s.n_words == 1 else 0xFFFF))
    keys = np.unique(keys, axis=0)
    big = np.full(keys.shape[0], 0x80000000, dtype=np.uint32)
    for sat in (0, 1):
        idx = K.CountIndex(ctx, K.make_config(k, "DNA", strand="single"))
        ctx.check(L.lib.kmi_index_set_saturating(idx.h, sat))
        idx.insert_pairs(np.concatenate([keys, keys, keys[:50]]), np.concatenate([big, big, big[:50]]))      # 2 x 2^31 (and 3 x for the first 50) in ONE call
        kk, cc = idx.to_vector()
        want = {tuple(r): ((0xFFFFFFFF if sat else ((3 if i < 50 else 2) * 0x80000000) & 0xFFFFFFFF)) for i, r in enumerate(keys.tolist())}
        assert {tuple(r): int(c) for r, c in zip(kk.tolist(), cc.tolist())} == want
        idx.insert_pairs(keys[:10], np.full(10, 7, dtype=np.uint32))                                        # on top of the stored counts
        kk, cc = idx.to_vector()
        got = {tuple(r): int(c) for r, c in zip(kk.tolist(), cc.tolist())}
        for i, r in enumerate(keys[:10].tolist()):
            assert got[tuple(r)] == (0xFFFFFFFF if sat else (want[tuple(r)] + 7) & 0xFFFFFFFF)
        idx.close()


def test_kmer_pipeline_still_selectable(monkeypatch):
    """KMI_FUSED_PATH=kmer at context creation keeps the fused build on the k-mer pipeline (the fall-back of the super-k-mer
    path when a run or a tile exceeds its item capacity): same index."""
    import kmerind_amd as K
    monkeypatch.setenv("KMI_FUSED_PATH", "kmer")
    c2 = K.Context(0)
    k = 31
    s = orc.kspec(k, orc.DNA)
    data = K.synth_fastq(seed=11, genome_len=30_000, n_reads=3_000)
    idx = K.CountIndex(c2, K.make_config(k, "DNA", strand="canonical"))
    c2.profile(True)
    c2.profile_reset()
    idx.build(data)
    names = {p["name"] for p in c2.profile_get() if p["launches"]}
    assert "fastq_scatter" in names and "sk_reduce" not in names
    om = orc.CountMap(s, orc.CANONICAL)
    om.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
    _same_map(idx, om)
    idx.close()
    c2.close()


@pytest.mark.parametrize("k,alpha,strand", [(31, "DNA", "canonical"), (21, "DNA", "single"), (40, "DNA", "canonical"), (16, "DNA16", "single")])
def test_update_with_device_updaters(k, alpha, strand):
    """update(pairs, op) (distributed_densehash_map.hpp:1975-2003 -> densehash_map.hpp:663-714) with the device-side updaters:
    stored keys only, pairs of one key in input order (assign keeps the last), the return value counts the pairs applied"""
    import kmerind_amd as K
    ctx = K.Context(0)
    s = orc.kspec(k, ALPHA[alpha])
    st = STRAND[strand]
    data = bytes(K.synth_fastq(seed=k, genome_len=5000, n_reads=1500))
    kmers = orc.extract(s, data, orc.FASTQ)["kmers"]
    idx = K.CountIndex(ctx, K.make_config(k, alpha, strand=strand))
    idx.build(data)
    om = orc.CountMap(s, st)
    om.insert(kmers)
    keys, cnt = om.export()
    ref = {tuple(a): int(b) for a, b in zip(keys.tolist(), cnt.tolist())}
    rng = np.random.default_rng(k)
    absent = orc.extract(s, bytes(K.synth_fastq(seed=91, genome_len=40000, n_reads=20)), orc.FASTQ)["kmers"]

    def canon(q):
        return q if strand == "single" else orc.canonical(s, q)

    for op in ("add", "max", "min", "assign", "add"):
        q = np.concatenate([kmers[rng.integers(0, kmers.shape[0], size=6000)], absent, orc.revcomp(s, kmers[:500])])
        v = rng.integers(0, 50, size=q.shape[0]).astype(np.uint32)
        hit = 0
        for kk, x in zip(canon(q).tolist(), v.tolist()):
            t = tuple(kk)
            if t not in ref:
                continue
            hit += 1
            ref[t] = {"add": (ref[t] + x) & 0xFFFFFFFF, "max": max(ref[t], x), "min": min(ref[t], x), "assign": x}[op]
        assert idx.update_pairs(q, v, op) == hit
        gk, gc = idx.to_vector()
        assert gk.shape[0] == len(ref) and all(ref[tuple(a)] == int(b) for a, b in zip(gk.tolist(), gc.tolist()))
    idx.close()
    ctx.close()


@pytest.mark.parametrize("k,strand,kind", [(31, "canonical", "plain"), (21, "single", "plain"), (31, "canonical", "malformed"), (31, "canonical", "fasta"),
                                           (31, "canonical", "position")])
def test_build_from_host_memory_in_chunks(monkeypatch, k, strand, kind):
    """kmi_index_build_host does not copy the input before the build starts: the one-pass front end queues the copy in chunks on a
    stream of its own and takes, behind every chunk, the byte ranges whose bytes (and the bytes a range may scan behind its end)
    have arrived; every other path (another index kind, FASTA, an input the front end declines and hands to the general one) has
    to queue the whole copy before it reads. The knobs make a 6 MB input go through several chunks and hundreds of ranges.
    BenchmarkKmerIndex.cpp:526-533 starts its clock with the bytes in host memory."""
    import kmerind_amd as K
    monkeypatch.setenv("KMI_HOST_OVERLAP_MIN", "1")
    monkeypatch.setenv("KMI_FEED_MIN_CHUNK", "65536")
    monkeypatch.setenv("KMI_FEED_RATIO", "3")
    monkeypatch.setenv("KMI_FRONT_MIN_RANGE", "8192")
    c2 = K.Context(0)
    s = orc.kspec(k, orc.DNA)
    c2.profile(True)
    c2.profile_reset()
    if kind == "fasta":
        from tests.test_gpu_fasta import _synthetic_fasta
        data = np.frombuffer(_synthetic_fasta(np.random.default_rng(3), 60, line=70, eol=b"\n", orphan=False), dtype=np.uint8)
        idx = K.CountIndex(c2, K.make_config(k, "DNA", strand=strand, seq_format="fasta"))
        idx.build(data)
        om = orc.CountMap(s, STRAND[strand])
        om.insert(orc.extract(s, data, orc.FASTA)["kmers"])
        _same_map(idx, om)
    elif kind == "position":
        data = K.synth_fastq(seed=5 * k, genome_len=80_000, n_reads=6_000)
        idx = K.PositionIndex(c2, K.make_config(k, "DNA", strand=strand, index_kind="position"))
        idx.build(data)
        assert idx.local_size() == orc.extract(s, data, orc.FASTQ)["kmers"].shape[0]
    else:
        data = K.synth_fastq(seed=7 * k, genome_len=200_000, n_reads=20_000)
        if kind == "malformed":   # a record whose quality line is short: the front end declines, the general path words the error
            data = np.concatenate([data[:315 * 9000], np.frombuffer(b"@bad\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIII\n", dtype=np.uint8), data[315 * 9000:]])
        idx = K.CountIndex(c2, K.make_config(k, "DNA", strand=strand))
        if kind == "malformed":
            with pytest.raises(Exception):
                idx.build(data)
            assert idx.local_size() == 0
        else:
            idx.build(data)
            names = {p["name"] for p in c2.profile_get() if p["launches"]}
            assert "sk_front_fed" in names, names
            om = orc.CountMap(s, STRAND[strand])
            om.insert(orc.extract(s, data, orc.FASTQ)["kmers"])
            _same_map(idx, om)
            more = K.synth_fastq(seed=9 * k, genome_len=100_000, n_reads=3_000)   # into the existing entries
            idx.build(more)
            om.insert(orc.extract(s, more, orc.FASTQ)["kmers"])
            _same_map(idx, om)
    c2.profile(False)
    idx.close()
    c2.close()
