"""The de Bruijn node restatement of the oracle (edge_iterator.hpp:84-177, de_bruijn_node_trait.hpp:122-124,200-239,
de_bruijn_nodes_distributed.hpp:91-159) on hand-worked vectors: the reference holds no expected values for this path
(test_de_bruijn_graph_construction.cpp prints sizes), so these pin it."""
import os

import numpy as np

from tests import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data")
A, C_, G, T, N = 1, 2, 4, 8, 15   # DNA16 codes (alphabets.hpp:660-681)


def _enc(s, text):
    return int(orc.kmers_from_string(s, text)[0, 0])


def test_edge_iterator_hand_vector():
    # read ACGTACGTN, k = 5: windows ACGTA CGTAC GTACG TACGT ACGTN (N -> A in the 2-bit alphabet, 0xF as an edge)
    s = orc.kspec(5)
    data = b"@r1\nACGTACGTN\n+\nIIIIIIIII\n@r2\nTTTTT\n+\nIIIII\n@r3\nACG\n+\nIII\n"
    kmers, edges = orc.dbg_parse(s, data)
    assert kmers[:, 0].tolist() == [_enc(s, w) for w in ("ACGTA", "CGTAC", "GTACG", "TACGT", "ACGTA", "TTTTT")]
    # left << 4 | right; the first window of a read has no left base, the last no right base; a read of exactly k bases
    # has neither; a read shorter than k gives nothing
    assert edges.tolist() == [C_, A << 4 | G, C_ << 4 | T, G << 4 | N, T << 4, 0]


def test_reverse_complement_edges():
    # (complement(out) << 4) | complement(in), complement = 4-bit reversal: A <-> T, C <-> G, N stays
    f = orc.lib.orc_dbg_edges_revcomp
    assert f(A << 4 | G) == (C_ << 4 | T)
    assert f(G << 4 | N) == (N << 4 | C_)
    assert f(T << 4) == A
    assert all(f(f(e)) == e for e in range(256))


def test_node_map_hand_vector():
    s = orc.kspec(5)
    kmers, edges = orc.dbg_parse(s, b"@r1\nACGTACGTN\n+\nIIIIIIIII\n")
    m = orc.DbgMap(s)
    m.insert(kmers, edges)
    # ACGTA / TACGT and CGTAC / GTACG are strand pairs: two nodes, each under the strand that came first
    keys, cnt = m.export(canonical=False)
    got = {int(k): c.tolist() for k, c in zip(keys[:, 0], cnt)}
    assert got == {
        # ACGTA: itself with out C; TACGT (in G, out N) turned around = in ACGT, out C; ACGT(N) with in T
        _enc(s, "ACGTA"): [0, 2, 0, 0, 1, 1, 1, 2, 3],
        # CGTAC: in A out G; GTACG (in C, out T) turned around = in A, out G
        _enc(s, "CGTAC"): [0, 0, 2, 0, 2, 0, 0, 0, 2],
    }
    # both happen to be the smaller strand already
    keys2, cnt2 = m.export(canonical=True)
    assert {int(k): c.tolist() for k, c in zip(keys2[:, 0], cnt2)} == got
    # the other arrival order keeps the other strand; the canonical view is the same node
    m2 = orc.DbgMap(s)
    m2.insert(kmers[::-1].copy(), edges[::-1].copy())
    raw = {int(k): c.tolist() for k, c in zip(*[x[:, 0] if x.ndim == 2 and x.shape[1] == 1 else x for x in m2.export(canonical=False)])}
    assert _enc(s, "GTACG") in raw and raw[_enc(s, "GTACG")] == [0, 0, 0, 2, 0, 2, 0, 0, 2]   # out T x2, in C x2
    k3, c3 = m2.export(canonical=True)
    assert {int(k): c.tolist() for k, c in zip(k3[:, 0], c3)} == got
    # find under either strand
    fk, fc = m.find(orc.kmers_from_string(s, "GTACG"))
    assert fk[:, 0].tolist() == [_enc(s, "CGTAC")] and fc.tolist() == [[0, 0, 2, 0, 2, 0, 0, 0, 2]]


def test_exists_nodes_keep_bits_only():
    s = orc.kspec(5)
    kmers, edges = orc.dbg_parse(s, b"@r1\nACGTACGTN\n+\nIIIIIIIII\n")
    m = orc.DbgMap(s, exists_only=True)
    m.insert(kmers, edges)
    keys, cnt = m.export()
    got = {int(k): c.tolist() for k, c in zip(keys[:, 0], cnt)}
    assert got[_enc(s, "ACGTA")] == [0, 1, 0, 0, 1, 1, 1, 1, 0] and got[_enc(s, "CGTAC")] == [0, 0, 1, 0, 1, 0, 0, 0, 0]


def test_reference_fixture_conservation():
    # the input the reference's de Bruijn test reads (test/data/test.debruijn.small.fastq): every k-mer occurrence is one
    # update; the edge totals miss exactly one per read end (a node stored under the other strand swaps in and out, so only
    # their sum is fixed)
    for name, k in (("test.debruijn.small.fastq", 21), ("test.debruijn.tiny.fastq", 21)):
        data = open(os.path.join(GOLD, name), "rb").read()
        s = orc.kspec(k)
        kmers, edges = orc.dbg_parse(s, data)
        ex = orc.extract(s, data, orc.FASTQ)
        assert (kmers == ex["kmers"]).all()
        m = orc.DbgMap(s)
        m.insert(kmers, edges)
        keys, cnt = m.export()
        assert cnt[:, 8].sum() == kmers.shape[0]
        n_reads = ex["n_seqs"]
        ambiguous = sum(bin(int(e) & 0xF).count("1") > 1 or bin(int(e) >> 4).count("1") > 1 for e in edges)
        if ambiguous == 0:
            assert cnt[:, :8].sum() == 2 * (kmers.shape[0] - n_reads)
        assert keys.shape[0] == np.unique(orc.canonical(s, kmers), axis=0).shape[0]


def test_fasta_records_hand_worked():
    """FASTA: a record's characters are its sequence lines without their EOLs (NonEOLIter), so the neighbour of a k-mer may sit on
    another line; no edge crosses a header. Worked by hand: record a = ACGTAC / GTTN -> ACGTACGTTN (six 5-mers), record b = AC (none)."""
    s = orc.kspec(5)
    kmers, edges = orc.dbg_parse(s, b">a\nACGTAC\nGTTN\n>b desc\nAC\n", orc.FASTA)
    A, Cc, G, T, N = 1, 2, 4, 8, 15                                   # DNA16 codes
    assert edges.tolist() == [Cc, (A << 4) | G, (Cc << 4) | T, (G << 4) | T, (T << 4) | N, A << 4]
    expect = orc.extract(s, b">a\nACGTACGTTN\n", orc.FASTA)["kmers"]
    assert (kmers == expect).all()
    # CRLF line ends and a one-character line change nothing
    k2, e2 = orc.dbg_parse(s, b">a\r\nACGTAC\r\nG\r\nTTN\r\n>b desc\r\nAC\r\n", orc.FASTA)
    assert (k2 == kmers).all() and (e2 == edges).all()
