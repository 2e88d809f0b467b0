"""GPU parity of the filtering sequence iterators (SeqIterType = NFilterSequencesIterator / NSplitSequencesIterator,
src/io/filtered_sequence_iterator.hpp) through the C ABI: kmi_config.seq_filter on the extract and build paths, against
the oracle and the reference's own NoN / Split parse tables."""
import json
import os

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FILTERS = {"all": orc.SEQ_ALL, "n_filter": orc.SEQ_N_FILTER, "n_split": orc.SEQ_N_SPLIT}


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _fastq_with_n(seed, n_reads, n_marks):
    """synthetic 315-byte records with N / n sprinkled over the sequence lines (starts, ends, runs, neighbours), a few 'N'
    in headers-to-be-ignored positions of the quality lines too"""
    import kmerind_amd as K
    buf = np.array(K.synth_fastq(seed=seed, genome_len=5000, n_reads=n_reads), dtype=np.uint8)
    rng = np.random.default_rng(seed)
    recs = rng.integers(0, n_reads, size=n_marks)
    pos = rng.integers(0, 150, size=n_marks)
    pos[: n_marks // 8] = 0                       # first base of the line
    pos[n_marks // 8: n_marks // 4] = 149         # last base of the line
    ch = np.where(rng.random(n_marks) < 0.7, ord("N"), ord("n")).astype(np.uint8)
    buf[recs * 315 + 11 + pos] = ch
    run_recs = rng.integers(0, n_reads, size=max(1, n_marks // 10))   # runs of N
    for r in run_recs:
        a = int(rng.integers(0, 140))
        buf[r * 315 + 11 + a: r * 315 + 11 + a + int(rng.integers(2, 10))] = ord("N")
    q = rng.integers(0, n_reads, size=n_marks)    # 'N' is also a legal quality character (Phred 45)
    buf[q * 315 + 164 + rng.integers(0, 150, size=n_marks)] = ord("N")
    return buf.tobytes()


def _check_extract(ctx, data, fmt, k, alpha, flt, with_ids=True):
    import kmerind_amd as K
    s = orc.kspec(k, orc.DNA if alpha == "DNA" else orc.DNA5)
    cfg = K.make_config(k, alpha, strand="single", seq_format=fmt, seq_filter=flt,
                        index_kind="position" if with_ids else "count")
    ex = orc.extract(s, data, orc.FASTQ if fmt == "fastq" else orc.FASTA, want_ids=with_ids, seq_filter=FILTERS[flt])
    r = ctx.read_file(cfg, data, with_ids=with_ids)
    got = {"kmers": r[0], "ids": r[1] if with_ids else None, "n_seqs": r[-1]}
    assert got["kmers"].shape == ex["kmers"].shape, (flt, got["kmers"].shape, ex["kmers"].shape)
    assert (got["kmers"] == ex["kmers"]).all()
    if with_ids:
        assert (got["ids"] == ex["ids"]).all()
    return got, ex


@pytest.mark.parametrize("flt", ["all", "n_filter", "n_split"])
def test_fastq_filters_match_oracle_and_reference_tables(ctx, flt):
    pg = json.load(open(os.path.join(GOLD, "parse_golden.json")))["fastq"]
    data = open(os.path.join(GOLD, "data", "natural.withN.fastq"), "rb").read()
    got, ex = _check_extract(ctx, data, "fastq", pg["k"], "DNA5", flt)
    assert got["n_seqs"] == ex["n_seqs"]
    if flt != "all":
        row = [e for e in pg[flt] if e["file"] == "natural.withN.fastq"][0]
        assert got["kmers"].shape[0] == row["kmers"]                       # NoNFASTQParseTest / SplitFASTQParseTest tables
        if flt == "n_filter":
            assert got["n_seqs"] == row["yielded"]
    for seed, k, alpha in ((3, 31, "DNA"), (4, 21, "DNA5"), (5, 63, "DNA"), (6, 5, "DNA")):
        data = _fastq_with_n(seed, 700, 400)
        got, ex = _check_extract(ctx, data, "fastq", k, alpha, flt)
        assert got["n_seqs"] == ex["n_seqs"], (flt, seed)
        _check_extract(ctx, data, "fastq", k, alpha, flt, with_ids=False)


@pytest.mark.parametrize("flt", ["n_filter", "n_split"])
def test_count_index_build_with_filter(ctx, flt):
    import kmerind_amd as K
    data = _fastq_with_n(11, 3000, 900)
    for k, strand in ((31, "canonical"), (15, "single")):
        s = orc.kspec(k)
        cfg = K.make_config(k, "DNA", strand=strand, seq_filter=flt)
        idx = K.CountIndex(ctx, cfg)
        idx.build(data)
        m = orc.CountMap(s, orc.CANONICAL if strand == "canonical" else orc.SINGLE)
        m.insert(orc.extract(s, data, orc.FASTQ, seq_filter=FILTERS[flt])["kmers"])
        gk, gc = idx.to_vector()
        ek, ec = m.export()
        a, b = orc.sorted_pairs(gk, gc), orc.sorted_pairs(ek, ec)
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
        idx.close()
        # the fused extract + route of the occurrence-routing N > 1 path honours the filter too
        import ctypes as C
        from kmerind_amd import _lib as L
        ex = orc.extract(s, data, orc.FASTQ, seq_filter=FILTERS[flt])["kmers"]
        tk = ex if strand == "single" else orc.canonical(s, ex)
        p = 4
        buf = np.frombuffer(data, dtype=np.uint8)
        dbytes, dout = ctx.alloc(buf.size + 64), ctx.alloc(tk.nbytes + 64)
        ctx.to_device(dbytes, buf)
        counts = np.zeros(p, dtype=np.uint64)
        nt, ns = C.c_uint64(), C.c_uint64()
        ctx.check(L.lib.kmi_extract_route_dev(ctx.h, C.byref(cfg), C.c_void_p(dbytes), buf.size, p, C.c_void_p(dout), tk.shape[0],
                                              C.byref(nt), C.byref(ns), counts.ctypes.data_as(C.c_void_p)))
        out = np.zeros_like(tk)
        ctx.to_host(out, dout)
        ctx.free(dbytes); ctx.free(dout)
        assert nt.value == tk.shape[0] and int(counts.sum()) == tk.shape[0]
        assert (np.sort(out[:, 0]) == np.sort(tk[:, 0])).all()


@pytest.mark.parametrize("flt", ["n_split", "n_filter"])
def test_fasta_filters_match_oracle_and_reference_table(ctx, flt):
    pg = json.load(open(os.path.join(GOLD, "parse_golden.json")))["fasta"]
    data = open(os.path.join(GOLD, "data", "natural.withN.fasta"), "rb").read()
    got, ex = _check_extract(ctx, data, "fasta", pg["k"], "DNA5", flt)
    row = [e for e in pg[flt] if e["file"] == "natural.withN.fasta"][0]
    assert got["kmers"].shape[0] == row["kmers"]
    if flt == "n_filter":
        assert got["n_seqs"] == ex["n_seqs"] == row["yielded"]
    # multi-line records with N at line ends, line starts, across EOLs, lower case, and a record of only N
    rng = np.random.default_rng(2)
    recs = []
    for r in range(40):
        seq = rng.choice(list(b"ACGT"), size=int(rng.integers(1, 400))).astype(np.uint8)
        for _ in range(int(rng.integers(0, 6))):
            a = int(rng.integers(0, seq.size))
            seq[a: a + int(rng.integers(1, 5))] = ord("N") if rng.random() < 0.6 else ord("n")
        if r == 7:
            seq[:] = ord("N")
        lines = [bytes(seq[i: i + 60]) for i in range(0, seq.size, 60)]
        recs.append(b">r%d some text N n\n" % r + b"\n".join(lines) + b"\n")
    data = b"".join(recs)
    for k, alpha in ((21, "DNA"), (33, "DNA5"), (3, "DNA")):
        got, ex = _check_extract(ctx, data, "fasta", k, alpha, flt)
        if flt == "n_filter":
            assert got["n_seqs"] == ex["n_seqs"]
        _check_extract(ctx, data, "fasta", k, alpha, flt, with_ids=False)
    # a buffer that begins with sequence lines (no header yet) and holds an N there
    orphan = b"ACGTNACGTACGTACGTACGTACGTA\nACGT\n" + data
    _check_extract(ctx, orphan, "fasta", 5, "DNA", flt)


def test_unsupported_filter_combinations_are_refused(ctx):
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    data = open(os.path.join(GOLD, "data", "natural.withN.fasta"), "rb").read()
    with pytest.raises(L.KmiError):
        ctx.read_file(K.make_config(1, "DNA", seq_format="fasta", seq_filter="n_filter"), data)       # FASTA filters need k >= 2
    fq = open(os.path.join(GOLD, "data", "natural.withN.fastq"), "rb").read()
    with pytest.raises(L.KmiError):
        ctx.read_file(K.make_config(21, "DNA", index_kind="posqual", seq_filter="n_split"), fq, with_ids=True, with_quals=True)


@pytest.mark.parametrize("alpha", ["RNA", "RNA5"])
def test_rna_alphabets_match_oracle(ctx, alpha):
    """RNA_T / RNA6_T (alphabets.hpp:365-445, 448-530): U where DNA has T, and T an unknown character. FASTQ extract with
    ids, the fused count build and FASTA, against the oracle's restatement of the two FROM_ASCII tables."""
    import kmerind_amd as K
    oa = orc.RNA if alpha == "RNA" else orc.RNA5
    raw = np.array(K.synth_fastq(seed=21, genome_len=4000, n_reads=600), dtype=np.uint8)
    rng = np.random.default_rng(21)
    is_t = raw == ord("T")
    raw[is_t & (rng.random(raw.size) < 0.8)] = ord("U")              # mostly U, some T left in (unknown characters in RNA)
    low = rng.random(raw.size) < 0.05
    seq_rows = (np.arange(raw.size) % 315 >= 11) & (np.arange(raw.size) % 315 < 161)
    sel = low & seq_rows
    raw[sel] = np.char.lower(raw[sel].view("S1")).view(np.uint8)     # lower case too
    data = raw.tobytes()
    for k in (31, 21) if alpha == "RNA" else (21, 40):
        s = orc.kspec(k, oa)
        cfg = K.make_config(k, alpha, strand="single", index_kind="position")
        ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
        gk, gi, gn = ctx.read_file(cfg, data, with_ids=True)
        assert gn == ex["n_seqs"] and (gk == ex["kmers"]).all() and (gi == ex["ids"]).all()
        idx = K.CountIndex(ctx, K.make_config(k, alpha, strand="canonical"))      # the fused build
        idx.build(data)
        m = orc.CountMap(s, orc.CANONICAL)
        m.insert(ex["kmers"])
        a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*m.export())
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
        idx.close()
    fa = b">r1\nACGUUGCAUGCAUUGGACUUACGuuacgNNACGUAGCUAGCUTTTTACGUACGU\nUUGCA\n>r2\nUGCAUGCAUGCAUGCAUGCAUGCAGGGGG\n"
    s = orc.kspec(9, oa)
    ex = orc.extract(s, fa, orc.FASTA, want_ids=True)
    gk, gi, gn = ctx.read_file(K.make_config(9, alpha, strand="single", seq_format="fasta", index_kind="position"), fa, with_ids=True)
    assert (gk == ex["kmers"]).all() and (gi == ex["ids"]).all()


def test_dna16_alphabet_matches_oracle(ctx):
    """DNA16 (alphabets.hpp:648-733): IUPAC presence bits, 4 bits per character, complement = bit reversal. Array ops, FASTQ
    extract with ids, the fused count build (canonical), FASTA, against the oracle (whose table is pinned on the
    reference's FROM_ASCII array)."""
    import kmerind_amd as K
    raw = np.array(K.synth_fastq(seed=31, genome_len=3000, n_reads=500), dtype=np.uint8)
    rng = np.random.default_rng(31)
    seq_rows = (np.arange(raw.size) % 315 >= 11) & (np.arange(raw.size) % 315 < 161)
    iupac = np.frombuffer(b"RYSWKMBDHVNUrykn-.X", dtype=np.uint8)
    sel = seq_rows & (rng.random(raw.size) < 0.08)
    raw[sel] = iupac[rng.integers(0, iupac.size, size=int(sel.sum()))]
    data = raw.tobytes()
    for k in (15, 16, 21, 40):                                   # one to three words
        s = orc.kspec(k, orc.DNA16)
        cfg = K.make_config(k, "DNA16", strand="single", index_kind="position")
        ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
        gk, gi, gn = ctx.read_file(cfg, data, with_ids=True)
        assert gn == ex["n_seqs"] and gk.shape == ex["kmers"].shape and (gk == ex["kmers"]).all() and (gi == ex["ids"]).all()
        sub = ex["kmers"][:2000]
        assert (ctx.revcomp(cfg, sub) == orc.revcomp(s, sub)).all()
        assert (ctx.canonical(cfg, sub) == orc.canonical(s, sub)).all()
        idx = K.CountIndex(ctx, K.make_config(k, "DNA16", strand="canonical"))
        idx.build(data)
        m = orc.CountMap(s, orc.CANONICAL)
        m.insert(ex["kmers"])
        a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*m.export())
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
        idx.close()
    fa = b">r1\nACGTRYKMNNACGTUUACGBDHV-.ACGTACGT\nACGTTGCA\n>r2\nTGCATGCARYRYRYTGCATGCAGGGGG\n"
    s = orc.kspec(9, orc.DNA16)
    ex = orc.extract(s, fa, orc.FASTA, want_ids=True)
    gk, gi, gn = ctx.read_file(K.make_config(9, "DNA16", strand="single", seq_format="fasta", index_kind="position"), fa, with_ids=True)
    assert (gk == ex["kmers"]).all() and (gi == ex["ids"]).all()
