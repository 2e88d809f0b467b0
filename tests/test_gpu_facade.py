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


def _run(binary, fastq, ratio):
    exe = os.path.join(ROOT, "examples", binary)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    out = subprocess.run([exe, "-F", fastq, "-S", str(ratio)], capture_output=True, text=True, timeout=300)
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
def test_benchmark_harness_matches_oracle(binary, k, strand, fname):
    path = os.path.join(DATA, fname)
    ratio = 3
    got = _run(binary, path, ratio)
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
