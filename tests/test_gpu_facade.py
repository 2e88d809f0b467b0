"""The C++ facade (include/kmerind/kmer_index.hpp) driven by the BenchmarkKmerIndex-shaped
harness in examples/, on the GPU, checked against the oracle and SURVEY.md 8(c) known answers."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")


def _run(binary, fastq, ratio, force_dist=False):
    exe = os.path.join(ROOT, "examples", binary)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    env = dict(os.environ, KMI_FORCE_DIST="1") if force_dist else None
    out = subprocess.run([exe, "-F", fastq, "-S", str(ratio)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    nums = {}
    for key, pat in (("total", r"total size is (\d+)"), ("distinct", r"after insert/rehash is (\d+)"),
                     ("count", r"count results (\d+) present (\d+)"), ("find", r"find results (\d+) sum (\d+)"),
                     ("after_erase", r"after erase is (\d+)")):
        m = re.search(pat, out.stdout)
        assert m, out.stdout
        nums[key] = tuple(int(x) for x in m.groups())
    return nums


@pytest.mark.parametrize("binary,k,strand,fname", [
    ("bench_count_k21_dna", 21, orc.CANONICAL, "test.small.fastq"),
    ("bench_count_k31_dna", 31, orc.CANONICAL, "test.medium.fastq"),
    ("bench_count_k31_single_farm", 31, orc.SINGLE, "natural.fastq"),
    ("bench_count_k21_dna", 21, orc.CANONICAL, "test.unitiqs.fastq"),
])
@pytest.mark.parametrize("force_dist", [False, True])
def test_benchmark_harness_matches_oracle(binary, k, strand, fname, force_dist):
    """the reference's BenchmarkKmerIndex sequence through the facade; force_dist: the SAME binary with KMI_FORCE_DIST=1 -- a one-rank
    RCCL communicator, so insert (weighted pairs to their owners), size, count, find and erase take the collectives of size() > 1"""
    path = os.path.join(DATA, fname)
    ratio = 3
    got = _run(binary, path, ratio, force_dist)
    s = orc.kspec(k)
    ex = orc.extract(s, open(path, "rb").read(), orc.FASTQ)
    m = orc.CountMap(s, strand)
    m.insert(ex["kmers"])
    q = ex["kmers"][: ex["kmers"].shape[0] // ratio]
    ck, cv = m.count(q)
    fk, fv = m.find(q)
    assert got["total"] == (ex["kmers"].shape[0],)
    assert got["distinct"] == (m.size(),)
    assert got["count"] == (ck.shape[0], int(cv.sum()))
    assert got["find"] == (fk.shape[0], int(fv.astype(np.uint64).sum()))
    m.erase(q)
    assert got["after_erase"] == (m.size(),)


def test_known_answer_config1():
    """BASELINE.json configs[0]: CountIndex<Kmer<21,DNA>> on test.small.fastq -> 280 k-mers, 40 distinct"""
    got = _run("bench_count_k21_dna", os.path.join(DATA, "test.small.fastq"), 1)
    assert got["total"] == (280,) and got["distinct"] == (40,)
    assert got["count"] == (40, 40) and got["find"] == (40, 280) and got["after_erase"] == (0,)


def test_bad_extension_is_invalid_argument(tmp_path):
    exe = os.path.join(ROOT, "examples", "bench_count_k21_dna")
    bad = tmp_path / "reads.txt"
    bad.write_bytes(b"x\nACGT\n+\nIIII\n")
    out = subprocess.run([exe, "-F", str(bad)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "missing @" in out.stderr


def test_position_index_harness_matches_oracle():
    """PositionIndex<unordered_multimap<Kmer<21,DNA>, ShortSequenceKmerId>> through the facade"""
    path = os.path.join(DATA, "natural.fastq")
    ratio = 4
    got = _run("bench_pos_k21_dna", path, ratio)
    s = orc.kspec(21)
    data = open(path, "rb").read()
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
    m = orc.MultiMap(s, orc.CANONICAL)
    m.insert(ex["kmers"], ex["ids"])
    q = ex["kmers"][: ex["kmers"].shape[0] // ratio]
    ck, cv = m.count(q)
    fk, fv = m.find(q)
    pos = ((fv[:, 0] >> np.uint64(16)) & np.uint64(0xFFFFFFFFFF)) + (fv[:, 0] & np.uint64(0xFFFF))
    assert got["total"] == (ex["kmers"].shape[0],)
    assert got["distinct"] == (m.size(),)
    assert got["count"] == (ck.shape[0], int(cv.sum()))
    assert got["find"] == (fk.shape[0], int(pos.sum()))
    m.erase(q)
    assert got["after_erase"] == (m.size(),)


def test_predicate_forms_match_an_independent_computation():
    """find_if / count_if / erase_if (kmer_index.hpp:156-194), with and without a query vector, and build_posix, on a
    count index and a position index; expectations computed from the oracle's maps with numpy."""
    exe = os.path.join(ROOT, "examples", "predicates")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    path = os.path.join(DATA, "natural.fastq")
    thr = 2
    fasta = os.path.join(DATA, "natural.fasta")
    out = subprocess.run([exe, "-F", path, "-A", fasta, "-T", str(thr)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr

    def grab(pat):
        m = re.search(pat, out.stdout)
        assert m, out.stdout
        return tuple(int(x) for x in m.groups())

    s = orc.kspec(21)
    data = open(path, "rb").read()
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True)
    kmers = ex["kmers"]
    cm = orc.CountMap(s, orc.CANONICAL)
    cm.insert(kmers)
    keys, cnt = cm.export()
    freq = cnt >= thr
    assert grab(r"count size (\d+)") == (keys.shape[0],)
    assert grab(r"find_if\(pred\) (\d+) sum (\d+)") == (int(freq.sum()), int(cnt[freq].sum()))
    assert grab(r"count_if\(pred\) (\d+)") == (int(freq.sum()),)
    q = kmers[: kmers.shape[0] // 2]
    fk, fv = cm.find(q)
    fsel = fv >= thr
    assert grab(r"find_if\(query,pred\) (\d+) sum (\d+)") == (int(fsel.sum()), int(fv[fsel].astype(np.uint64).sum()))
    ck, _ = cm.count(q)
    assert grab(r"count_if\(query,pred\) (\d+) present (\d+)") == (ck.shape[0], int(fsel.sum()))
    after_q = keys.shape[0] - int(fsel.sum())
    assert grab(r"after erase_if\(query,pred\) (\d+)") == (after_q,)
    assert grab(r"after erase_if\(pred\) (\d+)") == (int((~freq).sum()),)

    mm = orc.MultiMap(s, orc.CANONICAL)
    mm.insert(kmers, ex["ids"])
    mk, mv = mm.export()
    pos = ((mv[:, 0] >> np.uint64(16)) & np.uint64(0xFFFFFFFFFF)) + (mv[:, 0] & np.uint64(0xFFFF))
    odd = (pos & np.uint64(1)) != 0
    assert grab(r"pos size (\d+)") == (mk.shape[0],)
    assert grab(r"pos find_if\(pred\) (\d+)") == (int(odd.sum()),)
    # entries of the query's keys with an odd position go away, every other entry stays
    qk = orc.canonical(s, q)
    qset = set(map(bytes, np.ascontiguousarray(qk)))
    in_q = np.array([bytes(r) in qset for r in np.ascontiguousarray(mk)])
    assert grab(r"pos after erase_if\(query,pred\) (\d+)") == (int(mk.shape[0] - (in_q & odd).sum()),)
    assert grab(r"pos after erase_if\(pred\) (\d+)") == (int((~odd).sum()),)
    # build_posix<FASTAParser>: same index type, FASTA grammar
    fex = orc.extract(s, open(fasta, "rb").read(), orc.FASTA)
    fm = orc.CountMap(s, orc.CANONICAL)
    fm.insert(fex["kmers"])
    assert grab(r"fasta size (\d+) total (\d+)") == (fm.size(), fex["kmers"].shape[0])


def test_reference_type_matrix_through_the_facade(tmp_path):
    """examples/type_matrix.cpp: the map / parameter combinations BenchmarkKmerIndex.cpp instantiates (unordered, densehash with
    SpecialKeys, sorted flavours; the three strand models; DistHash / StoreHash / DistTrans choices; count and position
    indexes; build_posix with NSplitSequencesIterator) compile under the reference's names and agree with one another; the
    canonical count is the oracle's."""
    exe = os.path.join(ROOT, "examples", "type_matrix")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    path = os.path.join(DATA, "natural.withN.fastq")
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "type matrix ok" in out.stdout
    m = re.search(r"canonical (\d+)", out.stdout)
    s = orc.kspec(21)
    om = orc.CountMap(s, orc.CANONICAL)
    om.insert(orc.extract(s, open(path, "rb").read(), orc.FASTQ, seq_filter=orc.SEQ_N_SPLIT)["kmers"])
    assert int(m.group(1)) == om.size()


def test_position_quality_index_harness_matches_oracle():
    """PositionQualityIndex<unordered_multimap<Kmer<31,DNA>, pair<ShortSequenceKmerId, float>>> (kmer_index.hpp:405-406) through
    the facade: read_file with KmerPositionQualityTupleParser tuples, insert, count, find (positions and the quality floats'
    bit patterns), erase -- BASELINE config 5's index type on one rank."""
    path = os.path.join(DATA, "natural.fastq")
    ratio = 4
    exe = os.path.join(ROOT, "examples", "bench_posqual_k31_dna")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    got = _run("bench_posqual_k31_dna", path, ratio)
    s = orc.kspec(31)
    data = open(path, "rb").read()
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=True)
    vals = np.stack([ex["ids"], ex["quals"].view(np.uint32).astype(np.uint64)], axis=1)
    m = orc.MultiMap(s, orc.CANONICAL, vw=2)
    m.insert(ex["kmers"], vals)
    q = ex["kmers"][: ex["kmers"].shape[0] // ratio]
    ck, cv = m.count(q)
    fk, fv = m.find(q)
    pos = ((fv[:, 0] >> np.uint64(16)) & np.uint64(0xFFFFFFFFFF)) + (fv[:, 0] & np.uint64(0xFFFF))
    assert got["total"] == (ex["kmers"].shape[0],)
    assert got["distinct"] == (m.size(),)
    assert got["count"] == (ck.shape[0], int(cv.sum()))
    assert got["find"] == (fk.shape[0], int(pos.sum()) + int(fv[:, 1].sum()))
    m.erase(q)
    assert got["after_erase"] == (m.size(),)


@pytest.mark.parametrize("force_dist", [False, True])
def test_weighted_insert_iterators_and_posqual_through_the_facade(force_dist):
    """examples/facade_extras.cpp: insert(vector<pair<Kmer, count>>) adds the values (weights 1 then 3 per occurrence ->
    4 x occurrences), cbegin()/cend()/get_map() walk this rank's entries, PositionQualityIndex tuples round-trip."""
    exe = os.path.join(ROOT, "examples", "facade_extras")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    path = os.path.join(DATA, "natural.fastq")
    env = dict(os.environ, KMI_FORCE_DIST="1") if force_dist else None
    fasta = os.path.join(DATA, "natural.fasta")
    out = subprocess.run([exe, path, fasta], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr

    def grab(pat):
        m = re.search(pat, out.stdout)
        assert m, out.stdout
        return tuple(int(x) for x in m.groups())

    s = orc.kspec(21)
    data = open(path, "rb").read()
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, want_quals=True)
    cm = orc.CountMap(s, orc.CANONICAL)
    cm.insert(ex["kmers"])
    n = ex["kmers"].shape[0]
    assert grab(r"weighted entries (\d+) sum (\d+) occurrences (\d+)") == (cm.size(), 4 * n, n)
    # update(): every 11th occurrence adds 5 to its key's count (one hit per such pair, the all-A k-mer is not stored and is
    # skipped), then entries with a count >= 20 are halved
    keys, cnt = cm.export()
    val = {tuple(kk): 4 * int(c) for kk, c in zip(keys.tolist(), cnt.tolist())}
    canon = orc.canonical(s, ex["kmers"])
    for kk in canon[::11].tolist():
        val[tuple(kk)] += 5
    halved = sum(1 for v in val.values() if v >= 20)
    val = {kk: (v // 2 if v >= 20 else v) for kk, v in val.items()}
    assert grab(r"update hit (\d+) halved (\d+) sum (\d+) entries (\d+)") == (len(canon[::11]), halved, sum(val.values()), cm.size())
    # device-side updaters: max with 9 on every 7th occurrence; assign (the last pair of a key in input order stays)
    for kk in canon[::7].tolist():
        val[tuple(kk)] = max(val[tuple(kk)], 9)
    s3 = sum(val.values())
    for j, kk in enumerate(canon[::3].tolist()):
        val[tuple(kk)] = j % 1000
    assert grab(r"device max hit (\d+) sum (\d+) assign hit (\d+) sum (\d+)") == (len(canon[::7]), s3, len(canon[::3]), sum(val.values()))
    # build_posix (with KMI_FORCE_DIST=1: the byte-range + record-aligned + collective form of size() > 1) and exists()
    present = {tuple(kk) for kk in keys.tolist()}
    qq = canon[::4].tolist()
    assert grab(r"build_posix entries (\d+) sum (\d+) exists (\d+) of (\d+)") == (cm.size(), n, sum(1 for kk in qq if tuple(kk) in present), len(qq) + 1)
    # the same on a FASTA file (KMI_FORCE_DIST=1: whole file on every rank, this rank's block, the collective build)
    fa = orc.extract(s, open(fasta, "rb").read(), orc.FASTA)
    fm = orc.CountMap(s, orc.CANONICAL)
    fm.insert(fa["kmers"])
    assert grab(r"fasta build_posix entries (\d+) sum (\d+) tuples (\d+)") == (fm.size(), fa["kmers"].shape[0], fa["kmers"].shape[0])
    assert grab(r"get_map local_size (\d+) size (\d+)") == (cm.size(), cm.size())
    vals = np.stack([ex["ids"], ex["quals"].view(np.uint32).astype(np.uint64)], axis=1)
    mm = orc.MultiMap(s, orc.CANONICAL, vw=2)
    mm.insert(ex["kmers"], vals)
    fk, fv = mm.find(ex["kmers"][::5])
    pos = ((fv[:, 0] >> np.uint64(16)) & np.uint64(0xFFFFFFFFFF)) + (fv[:, 0] & np.uint64(0xFFFF))
    assert grab(r"posqual tuples (\d+) entries (\d+) found (\d+) pos (\d+) qbits (\d+)") == (n, mm.size(), fk.shape[0], int(pos.sum()), int(fv[:, 1].sum()))
    assert grab(r"posqual all qbits (\d+)") == (int(vals[:, 1].sum()),)
