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


@pytest.mark.parametrize("k,alpha", [(31, "DNA"), (21, "DNA5"), (63, "DNA")])
def test_erase_nodes(ctx, k, alpha):
    """nodes.erase(keys) (the erase the node map inherits from the distributed map, distributed_unordered_map.hpp:719-779): the
    nodes of the query k-mers -- given under either strand, with repeats and absent keys among them -- leave the map; every other
    node keeps its counts (the oracle's map without those keys), and later inserts build on what stayed."""
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha)
    data = _with_n(bytes(K.synth_fastq(seed=3 * k, genome_len=5000, n_reads=1800)), k)
    ok, oe = orc.dbg_parse(s, data)
    om = orc.DbgMap(s)
    om.insert(ok, oe)
    g = K.DeBruijnNodes(ctx, cfg)
    g.build(data)
    keys, counts = om.export(canonical=True)
    rng = np.random.default_rng(k)
    pick = rng.permutation(keys.shape[0])[: keys.shape[0] // 3]
    victims = keys[pick]
    absent = orc.extract(s, bytes(K.synth_fastq(seed=91, genome_len=50000, n_reads=15)), orc.FASTQ)["kmers"]
    q = np.concatenate([victims[::2], orc.revcomp(s, victims[1::2]), absent, victims[:25]])
    gone = {tuple(r) for r in orc.canonical(s, q).tolist()} & {tuple(r) for r in keys.tolist()}
    assert g.erase(q) == len(gone)
    keep = np.array([tuple(r) not in gone for r in keys.tolist()])
    assert g.local_size() == int(keep.sum())
    assert (_nodes(*g.to_vector()) == _nodes(keys[keep], counts[keep])).all()
    assert g.find(victims)[0].shape[0] == 0
    assert g.erase(victims) == 0                                    # nothing left to erase
    # the map goes on from what stayed: a second batch meets the surviving nodes' counts
    data2 = _with_n(bytes(K.synth_fastq(seed=3 * k, genome_len=5000, n_reads=600, first_read=2500)), k + 1)
    k2, e2 = orc.dbg_parse(s, data2)
    om2 = orc.DbgMap(s)
    survivors = np.array([tuple(r) not in gone for r in orc.canonical(s, ok).tolist()])
    om2.insert(ok[survivors], oe[survivors])
    om2.insert(k2, e2)
    g.insert(k2, e2)
    assert (_nodes(*g.to_vector()) == _nodes(*om2.export(canonical=True))).all()
    g.erase(om2.export(canonical=True)[0])                          # erase everything
    assert g.local_size() == 0
    g.close()


@pytest.mark.parametrize("k,alpha", [(21, "DNA"), (31, "DNA"), (63, "DNA"), (21, "DNA5")])
def test_fasta_input(ctx, k, alpha):
    """The engine's parser is generic over the sequence type (de_bruijn_construct_engine.hpp:108-158): on FASTA the characters of a
    record are its sequence lines without their EOLs, so a k-mer's neighbours may sit on the line before or after it and no edge leads
    across a header. Parser tuples bit for bit and the node map against the oracle, on the reference's FASTA files and on multi-line
    records with N, CRLF and lines of every length."""
    import kmerind_amd as K
    s = orc.kspec(k, ALPHA[alpha])
    cfg = K.make_config(k, alpha, seq_format="fasta")
    rng = np.random.default_rng(k)

    def synth(n_rec, line, eol):
        out = []
        for r in range(n_rec):
            L = int(rng.integers(k - 3, max(6 * line, 3 * k)))
            seq = "".join("ACGTN"[c] for c in rng.choice(5, size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
            out.append((">r%d some text" % r).encode() + eol)
            for i in range(0, L, line):
                out.append(seq[i:i + line].encode() + eol)
        return b"".join(out)
    inputs = [open(os.path.join(GOLD, name), "rb").read() for name in ("test.fasta", "test2.fasta", "natural.fasta", "natural.withN.fasta")]
    inputs += [synth(40, 60, b"\n"), synth(25, 7, b"\r\n"), synth(30, 1, b"\n")]
    for data in inputs:
        ok, oe = orc.dbg_parse(s, data, orc.FASTA)
        g = K.DeBruijnNodes(ctx, cfg)
        gk, ge = g.parse(data)
        assert gk.shape == ok.shape and (gk == ok).all() and (ge == oe).all(), len(data)
        om = orc.DbgMap(s)
        om.insert(ok, oe)
        g.build(data)
        assert g.local_size() == om.size()
        if om.size():
            assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
        g.close()


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


def test_edge_counts_beyond_16_bits(ctx):
    """the accumulate pass counts in 16-bit halves and sends carries on separately: one k-mer 150 000 times with every edge
    bit set, another 70 000 times with alternating edges, twice (the second insert also carries the earlier counts over)"""
    import kmerind_amd as K
    k = 31
    s = orc.kspec(k)
    rng = np.random.default_rng(11)
    hot = np.array([[0x0123456789ABCDE], [0x0FEDCBA98765432]], dtype=np.uint64)
    om = orc.DbgMap(s)
    g = K.DeBruijnNodes(ctx, K.make_config(k))
    for rnd in range(2):
        keys = np.concatenate([np.repeat(hot[:1], 150_000, axis=0), np.repeat(hot[1:], 70_000, axis=0),
                               rng.integers(0, 1 << 62, size=(30_000, 1), dtype=np.uint64)])
        edges = np.concatenate([np.full(150_000, 0xFF, np.uint8), np.tile(np.array([0x21, 0x12], np.uint8), 35_000),
                                rng.integers(0, 256, size=30_000).astype(np.uint8)])
        perm = rng.permutation(keys.shape[0])
        keys, edges = keys[perm], edges[perm]
        om.insert(keys, edges)
        g.insert(keys, edges)
        assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    fk, fc = g.find(hot)
    assert sorted(int(c.max()) for c in fc) == [140_000, 300_000]


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


def test_build_over_rccl_one_rank_self_exchange(monkeypatch):
    """kmi_dbg_build_dist_host with KMI_FORCE_DIST=1 (one GPU, one-rank communicator): parse -> tuples grouped by KeyToRank of
    the canonical k-mer -> grouped ncclSend / ncclRecv -> insert; twice, so the second build meets the nodes of the first"""
    import ctypes as C
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    monkeypatch.setenv("KMI_FORCE_DIST", "1")
    ctx = K.Context(0, rank=0, nranks=1)
    comm = C.c_void_p()
    ctx.check(L.lib.kmi_comm_create(ctx.h, None, C.byref(comm)))
    try:
        k = 31
        s = orc.kspec(k)
        om = orc.DbgMap(s)
        g = K.DeBruijnNodes(ctx, K.make_config(k))
        for seed in (4, 5):
            data = np.frombuffer(_with_n(bytes(K.synth_fastq(seed=seed, genome_len=8000, n_reads=2000)), seed), dtype=np.uint8).copy()
            ctx.check(L.lib.kmi_dbg_build_dist_host(g.h, comm, data.ctypes.data_as(C.c_void_p), data.size))
            om.insert(*orc.dbg_parse(s, data))
            n = C.c_uint64()
            ctx.check(L.lib.kmi_dbg_size_dist(g.h, comm, C.byref(n)))
            assert n.value == om.size()
            assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
        # find() over the communicator: keys to their owners, answers back
        q = np.ascontiguousarray(np.concatenate([orc.dbg_parse(s, data)[0][::9], np.random.default_rng(2).integers(0, 1 << 62, (300, 1), dtype=np.uint64)]))
        r = L.Results()
        ctx.check(L.lib.kmi_dbg_find_dist_host(g.h, comm, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        fk = np.ctypeslib.as_array(r.keys, shape=(r.n,)).copy().reshape(-1, 1)
        fv = np.ctypeslib.as_array(r.values, shape=(r.n * 5,)).copy().view(np.uint32).reshape(r.n, 10)[:, :9].copy()
        L.lib.kmi_results_free(C.byref(r))
        assert (_nodes(fk, fv) == _nodes(*om.find(q, canonical=True))).all()
        r = L.Results()
        ctx.check(L.lib.kmi_dbg_count_dist_host(g.h, comm, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        ck = np.ctypeslib.as_array(r.keys, shape=(r.n,)).copy()
        cv = np.ctypeslib.as_array(r.values, shape=(r.n,)).copy()
        L.lib.kmi_results_free(C.byref(r))
        assert sorted(ck[cv > 0].tolist()) == sorted(set(fk[:, 0].tolist())) and set(cv.tolist()) <= {0, 1}
        g.close()
    finally:
        L.lib.kmi_comm_destroy(comm)
        ctx.close()


@pytest.mark.parametrize("force_dist", [False, True])
def test_reference_sample_program_through_the_facade(force_dist):
    """examples/de_bruijn_graph_construction.cpp (the reference's sample with kmerind/de_bruijn.hpp): node counts and checksums
    of find(), the neighbours node_utils derives, the whole map, and erase; with KMI_FORCE_DIST=1 every member takes the path of
    size() > 1 (byte range + record-aligned cut on the device, the collectives over a one-rank RCCL communicator)"""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "de_bruijn_graph_construction")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "de_bruijn_graph_construction"])
    path = os.path.join(GOLD, "test.debruijn.small.fastq")
    env = dict(os.environ, KMI_FORCE_DIST="1") if force_dist else None
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    k = 21
    s = orc.kspec(k)
    data = open(path, "rb").read()
    kmers, edges = orc.dbg_parse(s, data)
    q = kmers[: kmers.shape[0] // 2] if kmers.shape[0] > 50 else kmers
    mask = np.uint64((1 << (2 * k)) - 1)
    for tag, exists in (("count", False), ("exist", True)):
        m = re.search(tag + r" nodes (\d+) size (\d+) found (\d+) keysum (\d+) edgesum (\d+) nbrsum (\d+) a_out_t_in (\d+)", out.stdout)
        assert m, out.stdout
        got = tuple(int(x) for x in m.groups())
        om = orc.DbgMap(s, exists_only=exists)
        om.insert(kmers, edges)
        fk, fc = om.find(q, canonical=True)
        w = np.arange(1, 9, dtype=np.uint64)
        nbr = 0
        for key, c in zip(fk[:, 0].tolist(), fc.tolist()):
            for i in range(4):
                if c[i]:
                    nbr += (((key << 2) | i) & int(mask)) % 1000003          # nextFromChar
                if c[4 + i]:
                    nbr += ((key >> 2) | (i << (2 * (k - 1)))) % 1000003     # nextReverseFromChar
        ak, ac = om.export(canonical=True)
        assert got == (om.size(), om.size(), fk.shape[0], int(fk[:, 0].sum()), int((fc[:, :8].astype(np.uint64) * w).sum()), nbr,
                       int(ac[:, 0].astype(np.uint64).sum() + ac[:, 7].astype(np.uint64).sum()))
        m = re.search(tag + r" erased (\d+) left (\d+) keysum (\d+)", out.stdout)
        assert m, out.stdout
        gone = {tuple(r) for r in orc.canonical(s, q[::3]).tolist()} & {tuple(r) for r in ak.tolist()}
        stay = [r for r in ak.tolist() if tuple(r) not in gone]
        assert tuple(int(x) for x in m.groups()) == (len(gone), len(stay), sum(int(r[0]) % 1000003 for r in stay))


def test_degenerate_inputs(ctx):
    """nothing, reads shorter than k, a read of exactly k bases (one node without edges), CRLF line ends"""
    import kmerind_amd as K
    k = 21
    s = orc.kspec(k)
    g = K.DeBruijnNodes(ctx, K.make_config(k))
    g.build(b"")
    assert g.local_size() == 0 and g.find(np.zeros((3, 1), np.uint64))[0].shape[0] == 0
    g.build(b"@a\nACGT\n+\nIIII\n@b\nACGTACGT\n+\nIIIIIIII\n")
    assert g.local_size() == 0
    exact = b"@c\nACGTTGCAACGTTGCAACGTA\n+\n" + b"I" * 21 + b"\n"
    g.build(exact)
    keys, cnt = g.to_vector()
    assert keys.shape[0] == 1 and cnt.tolist() == [[0] * 8 + [1]]
    crlf = b"@d\r\nACGTTGCAACGTTGCAACGTACCGT\r\n+\r\n" + b"I" * 25 + b"\r\n"
    om = orc.DbgMap(s)
    om.insert(*orc.dbg_parse(s, exact))
    om.insert(*orc.dbg_parse(s, crlf))
    g.build(crlf)
    assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    gk, ge = g.parse(crlf)
    ok, oe = orc.dbg_parse(s, crlf)
    assert (gk == ok).all() and (ge == oe).all()


@pytest.mark.parametrize("k,reads,genome", [(31, 6000, 30_000), (21, 4000, 8_000), (17, 3000, 3_000), (32, 3000, 20_000), (27, 5000, 400_000)])
def test_nodes_through_super_kmer_records(k, reads, genome):
    """An empty graph built from clean FASTQ reads (one run per read, A C G T only) goes through the count index's super-k-mer build:
    the front end cuts records that carry the base before their first and behind their last k-mer, the back end makes the node
    index, and sk_edges_accumulate adds the edges up from the records -- a k-mer's neighbours are its record's neighbouring bases
    (kmi_debruijn.h; edge_iterator.hpp:84-177 + de_bruijn_nodes_distributed::local_insert). Same nodes and counts as the oracle's
    tuple-by-tuple map; then tuples inserted into that graph (the node index changes its layout and the edge counts follow), find
    under either strand, erase. Poly-A reads push one node's counters past 16 bits and give the k = 32 key that equals the table's
    empty marker when they are poly-T."""
    import kmerind_amd as K
    c2 = K.Context(0)
    s = orc.kspec(k)
    data = bytes(K.synth_fastq(seed=3 * k, genome_len=genome, n_reads=reads))
    if k in (31, 32):   # 700 reads of one base: 84 000 occurrences of one node, every one with the same two edges
        base = b"A" if k == 31 else b"T"
        data += b"".join(b"@p%d\n" % i + base * 150 + b"\n+\n" + b"I" * 150 + b"\n" for i in range(700))
    c2.profile(True)
    c2.profile_reset()
    g = K.DeBruijnNodes(c2, K.make_config(k))
    g.build(data)
    names = {p["name"] for p in c2.profile_get() if p["launches"]}
    assert "sk_edges_accumulate" in names and "dbg_accumulate" not in names, names
    ok, oe = orc.dbg_parse(s, data)
    om = orc.DbgMap(s)
    om.insert(ok, oe)
    assert g.local_size() == om.size()
    assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    rng = np.random.default_rng(k)
    q = np.concatenate([ok[rng.integers(0, ok.shape[0], size=1500)], orc.revcomp(s, ok[rng.integers(0, ok.shape[0], size=500)]),
                        rng.integers(0, 1 << min(2 * k, 63), size=(300, 1), dtype=np.uint64)])
    assert (_nodes(*g.find(q)) == _nodes(*om.find(q, canonical=True))).all()
    # tuples into the graph: the nodes move to the placement-hash layout, their edge counts with them
    data2 = _with_n(bytes(K.synth_fastq(seed=5 * k, genome_len=genome, n_reads=reads // 3, first_read=reads)), k)
    k2, e2 = orc.dbg_parse(s, data2)
    om.insert(k2, e2)
    g.insert(k2, e2)
    assert g.local_size() == om.size()
    assert (_nodes(*g.to_vector()) == _nodes(*om.export(canonical=True))).all()
    assert (_nodes(*g.find(q)) == _nodes(*om.find(q, canonical=True))).all()
    # a second graph: erase straight after the super-k-mer build (the index keeps its layout), then a build into what stayed
    g2 = K.DeBruijnNodes(c2, K.make_config(k))
    g2.build(data)
    victims = np.ascontiguousarray(ok[rng.integers(0, ok.shape[0], size=800)])
    gone = {tuple(r) for r in orc.canonical(s, victims).tolist()}
    assert g2.erase(victims) == len(gone)
    survivors = np.array([tuple(r) not in gone for r in orc.canonical(s, ok).tolist()])
    om2 = orc.DbgMap(s)
    om2.insert(ok[survivors], oe[survivors])
    assert g2.local_size() == om2.size()
    assert (_nodes(*g2.to_vector()) == _nodes(*om2.export(canonical=True))).all()
    g2.build(data2)
    om2.insert(k2, e2)
    assert (_nodes(*g2.to_vector()) == _nodes(*om2.export(canonical=True))).all()
    c2.profile(False)
    g.close(); g2.close(); c2.close()
