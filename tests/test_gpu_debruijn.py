"""GPU parity of the de Bruijn node build (kmi_dbg_*, test/test/debruijn/ of the reference) against the oracle's
restatement: parser tuples bit for bit; nodes compared in the orientation of the smaller strand (the reference keeps
whichever strand arrived first, the library the smaller one -- include/kmerind_hip.h)."""
import os

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "data")
ALPHA = {"DNA": orc.DNA, "DNA5": orc.DNA5, "DNA16": orc.DNA16}


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _nodes(keys, counts):
    return orc.sorted_rows(keys, counts.astype(np.uint64))


def _with_n(data, seed, rate=0.01):
    """sprinkle N into the sequence lines (an N neighbour counts for all four edges; inside a k-mer it reads as A)"""
    rng = np.random.default_rng(seed)
    lines = data.split(b"\n")
    for i in range(1, len(lines), 4):
        b = bytearray(lines[i])
        for j in np.nonzero(rng.random(len(b)) < rate)[0]:
            b[j] = ord("N")
        lines[i] = bytes(b)
    return b"\n".join(lines)


@pytest.mark.parametrize("name,k", [("test.debruijn.tiny.fastq", 21), ("test.debruijn.small.fastq", 21), ("test.debruijn.small.fastq", 31),
                                    ("natural.withN.fastq", 21), ("test.medium.fastq", 15)])
def test_parser_tuples_on_reference_inputs(ctx, name, k):
    import kmerind_amd as K
    data = open(os.path.join(GOLD, name), "rb").read()
    s = orc.kspec(k)
    g = K.DeBruijnNodes(ctx, K.make_config(k))
    gk, ge = g.parse(data)
    ok, oe = orc.dbg_parse(s, data)
    assert gk.shape == ok.shape and (gk == ok).all() and (ge == oe).all()
    g.build(data)
    om = orc.DbgMap(s)
    om.insert(ok, oe)
    assert g.local_size() == om.size()
    assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()


@pytest.mark.parametrize("k,alpha", [(21, "DNA"), (31, "DNA"), (32, "DNA"), (5, "DNA"), (33, "DNA"), (63, "DNA"), (96, "DNA"), (21, "DNA5"),
                                     (16, "DNA16"), (40, "DNA5")])
def test_nodes_build_insert_find(ctx, k, alpha):
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha)
    data = _with_n(bytes(K.synth_fastq(seed=k, genome_len=4000, n_reads=1500)), k)
    ok, oe = orc.dbg_parse(s, data)
    om = orc.DbgMap(s)
    om.insert(ok, oe)
    g = K.DeBruijnNodes(ctx, cfg)
    g.build(data)
    assert g.local_size() == om.size()
    assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    # a second batch through insert(tuples), reversed so that other strands arrive first in the oracle's map
    data2 = _with_n(bytes(K.synth_fastq(seed=k, genome_len=4000, n_reads=700, first_read=4000)), k + 1)
    k2, e2 = orc.dbg_parse(s, data2)
    om.insert(k2[::-1].copy(), e2[::-1].copy())
    g.insert(k2, e2)
    assert g.local_size() == om.size()
    assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    # find: present under either strand (with repeats), absent
    rng = np.random.default_rng(k)
    present = ok[rng.integers(0, ok.shape[0], size=2000)]
    flipped = orc.revcomp(s, present[:700])
    absent = orc.extract(s, bytes(K.synth_fastq(seed=77, genome_len=60000, n_reads=20)), orc.FASTQ)["kmers"]
    q = np.concatenate([present, flipped, absent, present[:40]])
    assert (_nodes(*g.find(q)) == _nodes(*om.find(q, canonical=True))).all()
    ck, cc = g.count(q)
    canon_q = np.unique(orc.canonical(s, q), axis=0)
    assert ck.shape[0] == canon_q.shape[0]
    have = {tuple(r) for r in om.export(canonical=True)[0].tolist()}
    assert all(int(c) == (tuple(r) in have) for r, c in zip(ck.tolist(), cc))
    g.clear()
    assert g.local_size() == 0 and g.find(q)[0].shape[0] == 0


def test_exists_nodes(ctx):
    import kmerind_amd as K
    k = 21
    s = orc.kspec(k)
    data = _with_n(bytes(K.synth_fastq(seed=9, genome_len=3000, n_reads=1000)), 3)
    ok, oe = orc.dbg_parse(s, data)
    om = orc.DbgMap(s, exists_only=True)
    om.insert(ok, oe)
    g = K.DeBruijnNodes(ctx, K.make_config(k), exists_only=True)
    g.build(data)
    assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    assert (_nodes(*g.find(ok[:500])) == _nodes(*om.find(ok[:500], canonical=True))).all()


def test_bucket_with_more_nodes_than_one_table_chunk(ctx):
    """about 10 000 nodes in ONE placement bucket: the accumulate pass walks the bucket in chunks of 3072 table rows, and a
    second insert moves the counts of the nodes that were there to their new positions"""
    import kmerind_amd as K
    from tests.test_gpu_index import _keys_in_one_placement_bucket
    k = 31
    s = orc.kspec(k)
    rng = np.random.default_rng(5)
    cfg = K.make_config(k)
    cand = _keys_in_one_placement_bucket(20_000, bucket=4242).reshape(-1, 1)
    hot = cand[(orc.canonical(s, cand) == cand).all(axis=1)]       # the ones that are their own smaller strand stay in the bucket
    assert hot.shape[0] > 3 * 3072
    om = orc.DbgMap(s)
    g = K.DeBruijnNodes(ctx, cfg)
    for rnd in range(2):
        sel = hot if rnd == 0 else hot[::2]
        keys = np.concatenate([np.repeat(sel, 3, axis=0), rng.integers(0, 1 << 62, size=(50_000, 1), dtype=np.uint64)])
        rng.shuffle(keys)
        edges = rng.integers(0, 256, size=keys.shape[0]).astype(np.uint8)
        om.insert(keys, edges)
        g.insert(keys, edges)
        assert g.local_size() == om.size()
        assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    fk, fc = g.find(hot[::5])
    assert fk.shape[0] == hot[::5].shape[0] and (_nodes(fk, fc) == _nodes(*om.find(hot[::5], canonical=True))).all()


def test_full_size_properties(ctx):
    """1 M reads (120 M k-mers): occurrences sum to the k-mer count, edge totals miss one per read end, the node keys are the
    count index's keys; a sampled set of nodes agrees with the oracle"""
    import kmerind_amd as K
    k = 31
    s = orc.kspec(k)
    cfg = K.make_config(k)
    n_reads = 1_000_000
    data = K.synth_fastq(seed=2, genome_len=10_000_000, n_reads=n_reads)
    g = K.DeBruijnNodes(ctx, cfg)
    g.build(data)
    keys, cnt = g.to_vector()
    n_kmers = n_reads * 120
    assert int(cnt[:, 8].sum()) == n_kmers
    assert int(cnt[:, :8].astype(np.uint64).sum()) == 2 * (n_kmers - n_reads)
    ci = K.CountIndex(ctx, cfg)
    ci.build(data)
    ck, cc = ci.to_vector()
    assert (orc.sorted_pairs(keys, cnt[:, 8]) [0] == orc.sorted_pairs(ck, cc)[0]).all()
    assert (orc.sorted_pairs(keys, cnt[:, 8]) [1] == orc.sorted_pairs(ck, cc)[1]).all()
    # oracle agreement on the first 3000 reads' k-mers: their nodes in a map built from ALL reads need the whole input, so
    # compare a map built from a prefix with the device's map of the same prefix
    head = bytes(data[: 315 * 3000])
    ok, oe = orc.dbg_parse(s, head)
    om = orc.DbgMap(s)
    om.insert(ok, oe)
    g2 = K.DeBruijnNodes(ctx, cfg)
    g2.build(head)
    assert (_nodes(*g2.to_vector()) == _nodes(*om.export(canonical=True))).all()
