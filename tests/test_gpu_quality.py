"""GPU parity of the k-mer quality channel (PositionQualityIndex): sequential float window sum +
exp2 per read, bit-exact against the oracle's restatement of QualityScoreSlidingWindow."""
import os

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _fastq_with_quals(rng, n, lo=33, hi=74, zero_frac=0.0, max_len=300):
    recs = []
    for i in range(n):
        ln = int(rng.integers(1, max_len))
        seq = bytes(rng.choice(list(b"ACGT"), size=ln).tolist())
        q = rng.integers(lo, hi, size=ln, dtype=np.uint8)
        if zero_frac:
            q[rng.random(ln) < zero_frac] = 33            # Phred 0: zero probability of being correct
        recs.append(b"@r%d\n" % i + seq + b"\n+r%d\n" % i + bytes(q.tolist()) + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("k", [31, 21, 5, 63])
def test_kmer_quality_bit_exact(ctx, k):
    import kmerind_amd as K
    alpha = "DNA"
    s = orc.kspec(k, orc.DNA)
    cfg = K.make_config(k, alpha, index_kind="posqual")
    rng = np.random.default_rng(k)
    inputs = [open(os.path.join(GOLD, "data", n), "rb").read() for n in ("test.small.fastq", "natural.fastq")]
    inputs.append(_fastq_with_quals(rng, 400))
    inputs.append(_fastq_with_quals(rng, 300, zero_frac=0.02))
    inputs.append(_fastq_with_quals(rng, 200, lo=33, hi=129))          # the whole LUT incl. Q94/Q95
    inputs.append(bytes(K.synth_fastq(seed=6, genome_len=50_000, n_reads=1500)))
    for data in inputs:
        ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=True)
        kmers, ids, quals, nseq = ctx.read_file(cfg, data, with_ids=True, with_quals=True)
        assert kmers.shape == ex["kmers"].shape and (kmers == ex["kmers"]).all() and (ids == ex["ids"]).all()
        assert quals.dtype == np.float32
        assert (quals.view(np.uint32) == ex["quals"].view(np.uint32)).all()
        assert (quals >= 0).all() and (quals <= 1.0 + 1e-5).all()   # the running float sum may drift a hair above 0


def test_position_quality_index(ctx):
    import kmerind_amd as K
    k = 31
    s = orc.kspec(k, orc.DNA)
    cfg = K.make_config(k, "DNA", strand="canonical", index_kind="posqual")
    data = bytes(K.synth_fastq(seed=12, genome_len=5000, n_reads=800))
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=True)
    vals = np.stack([ex["ids"], ex["quals"].view(np.uint32).astype(np.uint64)], axis=1)
    mm = orc.MultiMap(s, orc.CANONICAL, vw=2)
    mm.insert(ex["kmers"], vals)
    idx = K.PositionIndex(ctx, cfg)
    idx.build(data)
    gk, gv = idx.to_vector()
    mk, mv = mm.export()
    assert (orc.sorted_rows(gk, gv) == orc.sorted_rows(mk, mv)).all()
    q = ex["kmers"][::7]
    fk, fv = idx.find(q)
    ek, ev = mm.find(q)
    assert (orc.sorted_rows(fk, fv) == orc.sorted_rows(ek, ev)).all()
    ck, cc = idx.count(q)
    ok, oc = mm.count(q)
    assert (orc.sorted_rows(ck, cc) == orc.sorted_rows(ok, oc)).all()
    idx.close()


def test_position_quality_index_many_tiles(ctx):
    """200 k reads (24 M tuples): the build's first partition pass reads the (k-mer, id) records and the dense quality array over
    thousands of tiles; every (canonical k-mer, id, quality bits) row of the index against the oracle's tuples, as sorted mixes"""
    import kmerind_amd as K
    k = 31
    s = orc.kspec(k, orc.DNA)
    cfg = K.make_config(k, "DNA", strand="canonical", index_kind="posqual")
    data = bytes(K.synth_fastq(seed=21, genome_len=2_000_000, n_reads=200_000))
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=True, file_offset=123_456_789)
    idx = K.PositionIndex(ctx, cfg)
    idx.build(data, file_offset=123_456_789)
    assert idx.local_size() == ex["kmers"].shape[0] == 200_000 * 120
    gk, gv = idx.to_vector()
    idx.close()

    def mix(kmers, ids, qbits):
        with np.errstate(over="ignore"):
            m = kmers.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            m ^= (ids.astype(np.uint64) + np.uint64(0x7F4A7C15)) * np.uint64(0xC2B2AE3D27D4EB4F)
            m ^= (qbits.astype(np.uint64) + np.uint64(1)) * np.uint64(0x165667B19E3779F9)
        return np.sort(m)
    exp = mix(orc.canonical(s, ex["kmers"])[:, 0], ex["ids"], ex["quals"].view(np.uint32))
    got = mix(gk[:, 0], gv[:, 0], gv[:, 1])
    assert (exp == got).all()
