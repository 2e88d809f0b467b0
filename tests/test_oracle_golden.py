"""Pins the CPU oracle against the reference's own golden vectors:
 * src/common/test/test_kmer.cpp known-answer arrays (tests/golden/kmer_golden.json)
 * vendored MurmurHash3 / FarmHash compiled as-is (oracle/_ref)
 * TestFileInfo parse-count tables on test/data files (tests/golden/parse_golden.json)
 * SURVEY.md 8(c) known answers obtained from reference-compiled code.
CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _words_to_u64(vals, word_bits, n_words):
    big = 0
    for i, v in enumerate(vals):
        big |= int(v) << (i * word_bits)
    return np.array([(big >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(n_words)], dtype=np.uint64)


@pytest.mark.parametrize("case,alpha,per_word", [("char2", orc.DNA, 32), ("char3", orc.DNA5, 21)])
def test_pack_slide_known_answers(case, alpha, per_word):
    g = _load("kmer_golden.json")[case]
    codes = g["codes"]
    ex = [int(x, 16) for x in g["kmer_ex"]]
    bits = g["bits"]
    for k in g["ks"]:
        s = orc.kspec(k, alpha)
        km = np.zeros(s.n_words, dtype=np.uint64)
        shift = (per_word - k) * bits
        for i, c in enumerate(codes):
            orc.lib.orc_kmer_next_from_char(C.byref(s), km, c)
            j = i - (k - 1)
            if 0 <= j < len(ex):
                assert int(km[0]) == ex[j] >> shift, (case, k, j)


def test_compare_known_answers():
    g = _load("kmer_golden.json")["compare"]
    s = orc.kspec(g["k"], orc.DNA)
    w = lambda name: _words_to_u64(g[name], g["word_bits"], s.n_words)
    less = lambda a, b: bool(orc.lib.orc_kmer_less(C.byref(s), a, b))
    kmer, s4, g3, g3s4 = w("kmer"), w("smaller4"), w("greater3"), w("g3s4")
    assert less(s4, kmer) and not less(kmer, s4)
    assert less(kmer, g3)
    assert less(g3s4, kmer)
    assert less(g3s4, g3)
    assert less(s4, g3s4)
    assert not less(kmer, kmer)


def test_reverse_known_answers():
    g = _load("kmer_golden.json")["reverse112"]
    for key, k, alpha in (("dna_k56", 56, orc.DNA), ("dna5_k37", 37, orc.DNA5)):
        s = orc.kspec(k, alpha)
        src = _words_to_u64(g["in"], g["word_bits"], s.n_words)
        # a Kmer constructed from raw words is sanitized to n_bits (kmer.hpp:1454-1460)
        pad = s.n_words * 64 - s.n_bits
        src[-1] &= np.uint64((1 << (64 - pad)) - 1)
        exp = _words_to_u64(g[key], g["word_bits"], s.n_words)
        out = np.zeros_like(src)
        orc.lib.orc_kmer_reverse(C.byref(s), src, out)
        assert out.tolist() == exp.tolist(), key


def test_revcomp_is_bitwise_rule():
    """DNA: rc == ~group_reverse ; DNA5: rc == full bit reversal (kmer.hpp:1723-1742,1807-1847)"""
    rng = np.random.default_rng(7)
    for k, alpha in ((31, orc.DNA), (21, orc.DNA), (63, orc.DNA), (63, orc.DNA5), (21, orc.DNA5), (32, orc.DNA)):
        s = orc.kspec(k, alpha)
        for _ in range(50):
            text = bytes(rng.choice(list(b"ACGT" if alpha == orc.DNA else b"ACGTN"), size=k).tolist())
            km = orc.kmers_from_string(s, text)[0]
            rc = orc.revcomp(s, km)[0]
            big = sum(int(km[w]) << (64 * w) for w in range(s.n_words))
            if alpha == orc.DNA:
                rev = 0
                for i in range(k):
                    rev |= (3 - ((big >> (2 * i)) & 3)) << (2 * (k - 1 - i))
            else:
                rev = int(format(big, "0%db" % s.n_bits)[::-1], 2)
            got = sum(int(rc[w]) << (64 * w) for w in range(s.n_words))
            assert got == rev
            # rc of the text-level reverse complement
            comp = bytes.maketrans(b"ACGTN", b"TGCAN")
            assert orc.kmers_from_string(s, text.translate(comp)[::-1])[0].tolist() == rc.tolist()


@pytest.mark.parametrize("ndebug", [False, True])
def test_hashes_match_reference_build(ndebug):
    ref = orc.ref_hash_lib(ndebug)
    if ref is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    mm, fh = ref
    orc.lib.orc_set_farm_ndebug(int(ndebug))
    try:
        _check_hashes(mm, fh)
    finally:
        orc.lib.orc_set_farm_ndebug(0)


def _check_hashes(mm, fh):
    rng = np.random.default_rng(11)
    out = np.zeros(2, dtype=np.uint64)
    exp = np.zeros(2, dtype=np.uint64)
    for length in list(range(0, 65)) * 3:
        buf = rng.integers(0, 256, size=max(length, 1), dtype=np.uint8)
        for seed in (42, 0, 83):
            orc.lib.orc_murmur3_x64_128(buf.ctypes.data, length, seed, out)
            mm(buf.ctypes.data, length, seed, exp.ctypes.data)
            assert out.tolist() == exp.tolist(), ("murmur", length, seed)
            a = orc.lib.orc_farm_hash64_with_seed(buf.ctypes.data, length, seed)
            b = fh(buf.ctypes.data, length, seed)
            assert a == b, ("farm", length, seed)


def test_survey_known_answers_kmer_and_hash():
    ka = _load("survey_known_answers.json")
    read = ka["read1"]
    s31 = orc.kspec(31, orc.DNA)
    k31 = orc.kmers_from_string(s31, read)
    g = ka["k31_dna"]
    assert int(k31[0, 0]) == int(g["first_kmer"], 16)
    rc = orc.revcomp(s31, k31[:1])
    assert int(rc[0, 0]) == int(g["revcomp"], 16)
    assert int(orc.canonical(s31, k31[:1])[0, 0]) == int(g["revcomp"], 16)
    assert int(orc.kmer_hash(s31, orc.MURMUR, True, k31[:1])[0]) == int(g["murmur_prefix"], 16)
    assert int(orc.kmer_hash(s31, orc.MURMUR, False, k31[:1])[0]) == int(g["murmur_store"], 16)
    assert int(orc.kmer_hash(s31, orc.FARM, True, k31[:1])[0]) == int(g["farm_prefix"], 16)
    assert int(orc.kmer_hash(s31, orc.FARM, False, k31[:1])[0]) == int(g["farm_store"], 16)
    s21 = orc.kspec(21, orc.DNA)
    k21 = orc.kmers_from_string(s21, read + "N")
    assert int(k21[0, 0]) == int(ka["k21_dna"]["first_kmer"], 16)
    assert int(k21[-1, 0]) == int(ka["k21_dna"]["last_kmer_with_trailing_N"], 16)
    assert s31.n_words * 8 == ka["sizes"]["Kmer<31,DNA,u64>"]
    assert orc.kspec(63, orc.DNA).n_words * 8 == ka["sizes"]["Kmer<63,DNA,u64>"]
    assert orc.kspec(63, orc.DNA5).n_words * 8 == ka["sizes"]["Kmer<63,DNA5,u64>"]


def test_survey_known_answers_count_index():
    ka = _load("survey_known_answers.json")
    for case in ka["count_index"]:
        data = open(os.path.join(GOLD, "data", case["file"]), "rb").read()
        s = orc.kspec(case["k"], orc.DNA)
        ex = orc.extract(s, data, orc.FASTQ)
        assert ex["n_seqs"] == case["reads"]
        assert ex["kmers"].shape[0] == case["kmers"]
        for strand, key in ((orc.SINGLE, "min_single"), (orc.CANONICAL, "min_canonical")):
            m = orc.CountMap(s, strand)
            m.insert(ex["kmers"])
            keys, counts = m.export()
            assert m.size() == case["distinct"]
            assert set(counts.tolist()) == {case["each_count"]}
            if key in case:
                assert int(keys[:, 0].min()) == int(case[key], 16)


def test_fastq_parse_counts_match_reference_table():
    pg = _load("parse_golden.json")["fastq"]
    s = orc.kspec(pg["k"], orc.DNA5)
    checked = 0
    for e in pg["files"]:
        path = os.path.join(GOLD, "data", e["file"])
        if not os.path.exists(path):
            continue
        data = open(path, "rb").read()
        assert len(data) == e["bytes"]
        ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
        assert ex["n_seqs"] == e["records"], e
        assert ex["kmers"].shape[0] == e["kmers"], e
        checked += 1
    assert checked >= 8


def test_fasta_parse_counts_match_reference_table():
    pg = _load("parse_golden.json")["fasta"]
    s = orc.kspec(pg["k"], orc.DNA5)
    checked = 0
    for e in pg["files"]:
        path = os.path.join(GOLD, "data", e["file"])
        if not os.path.exists(path):
            continue
        data = open(path, "rb").read()
        assert len(data) == e["bytes"]
        ex = orc.extract(s, data, orc.FASTA)
        assert ex["n_seqs"] == e["records"], e
        assert ex["kmers"].shape[0] == e["kmers"], e
        checked += 1
    assert checked >= 4


@pytest.mark.parametrize("fmt", ["fastq", "fasta"])
def test_filtered_parse_counts_match_reference_tables(fmt):
    """NoN*ParseTest / Split*ParseTest tables (mpi_test_fastq_seq_parse.cpp:834-846,1420-1432;
    mpi_test_fasta_seq_parse.cpp:764-769,1255-1260): sequences the NFilter / NSplit iterators yield and the k-mers parsed
    from them."""
    pg = _load("parse_golden.json")[fmt]
    s = orc.kspec(pg["k"], orc.DNA5)
    checked = with_n = 0
    for key, flt in (("n_filter", orc.SEQ_N_FILTER), ("n_split", orc.SEQ_N_SPLIT)):
        plain = {e["file"]: e for e in pg["files"]}
        for e in pg[key]:
            path = os.path.join(GOLD, "data", e["file"])
            if not os.path.exists(path):
                continue
            data = open(path, "rb").read()
            assert len(data) == e["bytes"]
            ex = orc.extract(s, data, orc.FASTQ if fmt == "fastq" else orc.FASTA, seq_filter=flt)
            assert ex["n_yield"] == e["yielded"], (key, e)
            assert ex["kmers"].shape[0] == e["kmers"], (key, e)
            with_n += e["kmers"] != plain[e["file"]]["kmers"]
            checked += 1
    assert checked >= 8 and with_n >= 2      # natural.withN.* differs from its unfiltered row under both filters


def test_alphabet_tables_match_reference():
    """orc_from_ascii against the reference's FROM_ASCII arrays of DNA, DNA5/6, RNA, RNA5/6 (all 256 bytes each)"""
    t = _load("alphabet_tables.json")
    for name, alpha in (("DNA_T", orc.DNA), ("DNA6_T", orc.DNA5), ("RNA_T", orc.RNA), ("RNA6_T", orc.RNA5), ("DNA16_T", orc.DNA16)):
        assert [orc.lib.orc_from_ascii(alpha, c) for c in range(256)] == t[name], name


def test_kmer_text_roundtrip_positions():
    """mpi_test_fastq_seq_parse.cpp:235-330: every k-mer equals the file bytes at its id"""
    data = open(os.path.join(GOLD, "data", "test.small.fastq"), "rb").read()
    s = orc.kspec(21, orc.DNA)
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
    for km, i in zip(ex["kmers"], ex["ids"]):
        rec = (int(i) >> 16) & 0xFFFFFFFFFF
        pos = rec + (int(i) & 0xFFFF)
        text = data[pos:pos + 21]
        assert orc.kmers_from_string(s, text)[0].tolist() == km.tolist()
        assert data[rec:rec + 1] == b"@"


def test_count_query_semantics():
    s = orc.kspec(21, orc.DNA)
    data = open(os.path.join(GOLD, "data", "test.small.fastq"), "rb").read()
    ex = orc.extract(s, data, orc.FASTQ)
    m = orc.CountMap(s, orc.CANONICAL)
    m.insert(ex["kmers"])
    q = np.concatenate([ex["kmers"][:5], ex["kmers"][:5], np.array([[12345]], dtype=np.uint64)])
    keys, cnt = m.count(q)
    assert keys.shape[0] == 6                       # distinct transformed keys
    assert sorted(cnt.tolist()) == [0, 1, 1, 1, 1, 1]  # db.count(k) is 0/1 for a map
    fk, fc = m.find(q)
    assert fk.shape[0] == 5 and set(fc.tolist()) == {7}
    assert m.erase(ex["kmers"][:5]) == 5
    assert m.size() == 35


def test_stable_bucket_and_rank():
    s = orc.kspec(31, orc.DNA)
    rng = np.random.default_rng(3)
    kmers = rng.integers(0, 1 << 62, size=(1000, 1), dtype=np.uint64)
    ranks = orc.key_to_rank(s, orc.MURMUR, orc.CANONICAL, kmers, 5)
    h = orc.kmer_hash(s, orc.MURMUR, True, kmers)
    assert (ranks == (h % np.uint64(5)).astype(np.uint32)).all()
    sizes = np.zeros(5, dtype=np.uint64)
    i2o = np.zeros(1000, dtype=np.uint64)
    orc.lib.orc_stable_bucket(ranks, 1000, 5, sizes, i2o)
    assert sizes.sum() == 1000
    order = np.argsort(ranks, kind="stable")
    assert (i2o[order] == np.arange(1000, dtype=np.uint64)).all()


def test_cpu_baseline_driver_agrees_across_thread_counts():
    data = open(os.path.join(GOLD, "data", "test.medium.fastq"), "rb").read()
    r1 = orc.bench_count_index(data, 21, orc.CANONICAL, 1)
    r3 = orc.bench_count_index(data, 21, orc.CANONICAL, 3)
    assert r1[1:] == (5600, 40) and r3[1:] == (5600, 40)


def test_quality_lut_matches_reference_literals():
    """Illumina18QualityScoreCodec<float>::DecodeLUT (quality_scores.hpp:113-211)"""
    g = _load("quality_lut.json")["decode_lut"]
    assert len(g) == 96
    for q, v in enumerate(g):
        exp = np.float32(np.finfo(np.float32).min) if v == "lowest" else np.float32(float(v))
        got = np.float32(orc.lib.orc_qual_lut(33 + q))
        assert exp.tobytes() == got.tobytes(), q


def test_quality_window_direct_recomputation():
    """src/index/test/test_quality_score_iterator.cpp: sliding values vs direct recomputation (to fp tolerance)"""
    rng = np.random.default_rng(5)
    k = 21
    s = orc.kspec(k, orc.DNA)
    seq = bytes(rng.choice(list(b"ACGT"), size=120).tolist())
    q = rng.integers(35, 74, size=120, dtype=np.uint8)
    data = b"@r\n" + seq + b"\n+\n" + bytes(q.tolist()) + b"\n"
    ex = orc.extract(s, data, orc.FASTQ, want_quals=True)
    lut = np.array([orc.lib.orc_qual_lut(int(c)) for c in range(33, 129)], dtype=np.float64)
    for j, got in enumerate(ex["quals"]):
        direct = 2.0 ** lut[q[j:j + k].astype(int) - 33].sum()
        assert abs(float(got) - direct) <= 1e-5 * direct


def test_identity_and_std_hashers_follow_the_cited_formulas():
    """identity / cpp_std (kmer_hash.hpp:154-230) are NOT pinned by any reference fixture; this only checks the
    oracle against a direct evaluation of the cited definitions on the SURVEY 8(c) 31-mer and a 3-word k-mer."""
    s = orc.kspec(31, orc.DNA)
    key = 0x23FAAF4092CD8D03
    km = np.array([[key]], dtype=np.uint64)
    M = (1 << 64) - 1
    assert int(orc.kmer_hash(s, orc.IDENTITY, False, km)[0]) == key                      # getSuffix(min(nBits, 64))
    assert int(orc.kmer_hash(s, orc.IDENTITY, True, km)[0]) == key >> (62 - 24)          # getPrefix(24): top 24 of 62 bits
    assert int(orc.kmer_hash(s, orc.STD, False, km)[0]) == (key << 1) & M
    assert int(orc.kmer_hash(s, orc.STD, True, km)[0]) == ((key << 1) & M) >> (62 - 32)  # shift = min(nBits,64) - min(32,nBits)
    # KeyToRank: prefix bits = ceilLog2(p)
    for p, bits in ((2, 1), (3, 2), (8, 3), (9, 4)):
        assert int(orc.key_to_rank(s, orc.IDENTITY, orc.SINGLE, km, p)[0]) == (key >> (62 - bits)) % p
        assert int(orc.key_to_rank(s, orc.STD, orc.SINGLE, km, p)[0]) == ((((key << 1) & M) >> (62 - bits)) % p)
    s3 = orc.kspec(63, orc.DNA5)   # 189 bits in 3 words
    w = [0x0123456789ABCDEF, 0xFEDCBA9876543210, 0x1555555555555555 & ((1 << 61) - 1)]
    km3 = np.array([w], dtype=np.uint64)
    val = w[0] | (w[1] << 64) | (w[2] << 128)
    assert int(orc.kmer_hash(s3, orc.IDENTITY, False, km3)[0]) == w[0]
    assert int(orc.kmer_hash(s3, orc.IDENTITY, True, km3)[0]) == val >> (189 - 24)
    h = ((w[0] << 1) ^ (w[1] << 1) ^ (w[2] << 1)) & M
    assert int(orc.kmer_hash(s3, orc.STD, False, km3)[0]) == h
    assert int(orc.kmer_hash(s3, orc.STD, True, km3)[0]) == h >> (64 - 32)


def test_std_hasher_against_the_standard_library_the_reference_calls():
    """cpp_std (kmer_hash.hpp:154-198) = xor of ::std::hash<size_t> of the words, shifted left by one: evaluated here with the
    toolchain's own std::hash<size_t> (oracle/std_hash_probe.cpp), for one-, two- and three-word k-mers; this pins the
    third-party half of the hasher (the Kmer word layout is row a1's)"""
    import ctypes as C
    import subprocess
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libkmerind_stdprobe.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so), "libkmerind_stdprobe.so"], stdout=subprocess.DEVNULL)
    probe = C.CDLL(so).orc_probe_std_hash_size_t
    probe.argtypes, probe.restype = [C.c_size_t], C.c_size_t
    M = (1 << 64) - 1
    rng = np.random.default_rng(4)
    for k, alpha in ((31, orc.DNA), (21, orc.DNA), (33, orc.DNA), (63, orc.DNA5), (96, orc.DNA), (16, orc.DNA16)):
        s = orc.kspec(k, alpha)
        km = rng.integers(0, 1 << 63, size=(64, s.n_words), dtype=np.uint64)
        km[:, -1] &= np.uint64((1 << (64 - (s.n_words * 64 - s.n_bits))) - 1)           # pad bits are zero in a Kmer
        want = []
        for row in km.tolist():
            h = 0                                   # leftover == 0: the word array is a whole number of size_t
            for w in row:
                h ^= (probe(w) << 1) & M
            want.append(h)
        got = orc.kmer_hash(s, orc.STD, False, km)
        assert [int(x) for x in got] == want
        shift = min(s.n_bits, 64) - min(32, s.n_bits)
        assert [int(x) for x in orc.kmer_hash(s, orc.STD, True, km)] == [h >> shift for h in want]


def test_count_full_checksums_match_the_single_map():
    """orc_count_full (thread ranks, slices, checksums of the maps: what the full-size GPU tests compare with) against the oracle's
    one-map build of the same reads"""
    rng = np.random.default_rng(5)
    genome = "".join("ACGT"[c] for c in rng.integers(0, 4, 3000))
    recs = []
    for i in range(400):
        st = int(rng.integers(0, 3000 - 100)); L = int(rng.integers(40, 100))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, genome[st:st + L], "I" * L))
    data = "".join(recs).encode()
    for k, strand in ((31, orc.CANONICAL), (21, orc.SINGLE)):
        s = orc.kspec(k)
        om = orc.CountMap(s, strand)
        kmers = orc.extract(s, data, orc.FASTQ)["kmers"]
        om.insert(kmers)
        want = orc.map_checksums(*om.export())
        for threads, slices in ((1, 1), (3, 2), (4, 5)):
            got = orc.count_full(data, k, strand, threads, slices)
            assert got["kmers"] == kmers.shape[0]
            assert {x: got[x] for x in want} == want
