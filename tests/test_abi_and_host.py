"""CPU-only checks of the boundary: the shared library loads, exports every symbol the header
declares, refuses to work without a GPU (no fallback), and the host-side helpers (synthetic
input, FASTQ partitioning) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "kmerind_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from kmerind_amd import _lib as L
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L.lib, s), "libkmerind_hip.so does not export %s" % s
        assert s in L.SIGNATURES, "python binding lacks %s" % s
    assert sorted(L.SIGNATURES) == syms


def test_kmer_shape_matches_reference_sizes():
    import kmerind_amd as K
    assert K.Context.shape(K.make_config(31, "DNA")) == (1, 62, 8)
    assert K.Context.shape(K.make_config(21, "DNA")) == (1, 42, 6)
    assert K.Context.shape(K.make_config(63, "DNA")) == (2, 126, 16)
    assert K.Context.shape(K.make_config(63, "DNA5")) == (3, 189, 24)   # SURVEY: sizeof = 24
    with pytest.raises(ValueError):
        K.Context.shape(K.make_config(200, "DNA"))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    with pytest.raises(L.KmiError) as e:
        K.Context(0)
    assert e.value.status == L.ERR_DEVICE


def test_product_does_not_touch_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "kmerind_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert "kmerind_oracle" not in text and "from tests" not in text and "import oracle" not in text, f
    for f in ("include/kmerind_hip.h", "include/kmerind/kmer_index.hpp", "examples/benchmark_kmer_index.cpp"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


def test_synth_fastq_contract():
    import kmerind_amd as K
    d = K.synth_fastq(seed=2, genome_len=100_000, n_reads=1000)
    assert d.size == 315 * 1000
    rec = bytes(d[:315])
    lines = rec.split(b"\n")
    assert lines[0] == b"@000000000" and len(lines[1]) == 150 and lines[2] == b"+" and len(lines[3]) == 150
    assert set(lines[1]) <= set(b"ACGT") and all(ord("#") <= c <= ord("I") for c in lines[3])
    # counter based: any sub-range reproduces the same bytes, any thread count too
    assert (K.synth_fastq(2, 100_000, 300, first_read=500, threads=3) == d[500 * 315:800 * 315]).all()
    assert (K.synth_fastq(2, 100_000, 1000, threads=1) == d).all()
    # reads are substrings of one genome: high coverage => few distinct k-mers
    s = orc.kspec(31)
    ex = orc.extract(s, d, orc.FASTQ)
    assert ex["n_seqs"] == 1000 and ex["kmers"].shape[0] == 120_000
    m = orc.CountMap(s, orc.CANONICAL)
    m.insert(ex["kmers"])
    assert m.size() < 105_000


def test_partition_fastq_is_record_aligned_and_lossless():
    import kmerind_amd as K
    from kmerind_amd import fileio
    s = orc.kspec(21)
    for data in (bytes(K.synth_fastq(5, 50_000, 777)),
                 open(os.path.join(ROOT, "tests/golden/data/natural.fastq"), "rb").read(),
                 open(os.path.join(ROOT, "tests/golden/data/test.small.fastq"), "rb").read()):
        whole = orc.extract(s, data, orc.FASTQ)
        for parts in (1, 2, 3, 8):
            ranges = fileio.partition_fastq(data, parts)
            assert ranges[0][0] == 0 and ranges[-1][1] == len(data)
            got, nseq = [], 0
            for b, e in ranges:
                assert b <= e
                if e > b:
                    assert data[b:b + 1] == b"@"
                    ex = orc.extract(s, data[b:e], orc.FASTQ, file_offset=b)
                    got.append(ex["kmers"]); nseq += ex["n_seqs"]
            assert nseq == whole["n_seqs"]
            assert (np.concatenate(got) == whole["kmers"]).all()


def test_fasta_partitioner_blocks_tile_the_file_and_carry_the_state():
    """kmerind_amd.fileio.partition_fasta: valid ranges tile the buffer, overlap holds k - 1 sequence characters (or
    runs to the end), and the state handed to each rank is the line-kind machine's state at its first byte."""
    from kmerind_amd import fileio
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "data", "test.fasta")
    data = open(path, "rb").read()
    kind, starts, recs = fileio.fasta_line_kinds(data)
    assert kind[0] == fileio.FA_HEADER and starts[0] == 1 and recs[-1] >= 1
    for p in (1, 2, 5, 13):
        parts = fileio.partition_fasta(data, p, 31)
        pos = 0
        for part in parts:
            if part["begin"] == len(data):
                continue
            assert part["begin"] == pos
            pos += part["valid_bytes"]
            assert part["begin"] + part["valid_bytes"] <= part["end"] <= len(data)
            b = part["begin"]
            assert part["at_line_start"] == (1 if (b == 0 or data[b - 1] == 10) else 0)
            if b and not part["at_line_start"]:
                assert part["start_state"] == kind[b]
            assert part["records_before"] == (recs[b - 1] if b else 0)
        assert pos == len(data)
