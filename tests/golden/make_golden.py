#!/usr/bin/env python3
"""Regenerates tests/golden/*.json and tests/golden/data/* from the reference checkout.

Run in the build container only (needs /root/reference, read as text/data):

    python tests/golden/make_golden.py

What it collects (data only -- expected values and test input files, no code):

* kmer_golden.json : the known-answer arrays of the reference's own unit test
  src/common/test/test_kmer.cpp (TestKmerGenerationChar2 :416-462,
  TestKmerGenerationChar3 :562-648, TestKmerComparison1 :654-693,
  TestKmerReverse112 :699-756).
* parse_golden.json : the TestFileInfo tables (records, k-mers, bytes) of
  src/io/test/mpi_test_fastq_seq_parse.cpp:446-459 (K=35, DNA5) and
  src/io/test/mpi_test_fasta_seq_parse.cpp:398-403.
* data/* : the small input files of test/data those tables refer to.
* survey_known_answers.json is written by hand from SURVEY.md section 8(c)
  (values obtained there by running reference-compiled code) and is not
  touched by this script.
"""
import json
import os
import re
import shutil

REF = os.environ.get("KMERIND_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _block(text, start_marker, end_marker=None):
    i = text.index(start_marker)
    j = text.index(end_marker, i) if end_marker else len(text)
    return text[i:j]


def _array(block, name):
    m = re.search(r"\b" + re.escape(name) + r"\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", block, re.S)
    if not m:
        raise KeyError(name)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return [int(tok, 0) for tok in re.findall(r"0x[0-9a-fA-F]+|\d+", body)]


def kmer_golden():
    src = open(os.path.join(REF, "src/common/test/test_kmer.cpp")).read()
    out = {"source": "src/common/test/test_kmer.cpp"}

    b = _block(src, "TEST(KmerGeneration, TestKmerGenerationChar2)", "template<typename Alphabet, int K>")
    out["char2"] = {
        "alphabet": "DNA", "bits": 2, "ks": [31, 28, 13, 4, 1],
        "codes": _array(b, "kmer_data"),
        "kmer_ex": [hex(v) for v in _array(b, "kmer_ex")],
        "rule": "kmer(K) after consuming codes[0..K-1+i] == kmer_ex[i] >> ((32-K)*2)",
    }
    b = _block(src, "TEST(KmerGeneration, TestKmerGenerationChar3)", "TEST(KmerComparison")
    out["char3"] = {
        "alphabet": "DNA5", "bits": 3, "ks": [21, 20, 13, 9, 1],
        "codes": _array(b, "kmer_data_8"),
        "kmer_ex": [hex(v) for v in _array(b, "kmer_ex")],
        "rule": "kmer(K) after consuming codes[0..K-1+i] == kmer_ex[i] >> ((21-K)*3)",
    }
    b = _block(src, "TEST(KmerComparison, TestKmerComparison1)", "TEST(KmerReverse")
    out["compare"] = {
        "k": 41, "alphabet": "DNA", "word_bits": 16,
        "kmer": _array(b, "kmer_val"), "smaller4": _array(b, "kmer_val_s4"),
        "greater3": _array(b, "kmer_val_g3"), "g3s4": _array(b, "kmer_val_g3s4"),
        "expect": ["kmer > smaller4", "greater3 > kmer", "kmer > g3s4", "greater3 > g3s4", "smaller4 < g3s4"],
    }
    b = _block(src, "TEST(KmerReverse, TestKmerReverse112)")
    out["reverse112"] = {
        "word_bits": 16,
        "in": _array(b, "kmer_val"),
        "dna_k56": _array(b, "kmer_ex"),
        "dna5_k37": _array(b, "kmer_ex_3"),
    }
    return out


def parse_golden():
    out = {"fastq": {"k": 35, "alphabet": "DNA5", "source": "src/io/test/mpi_test_fastq_seq_parse.cpp:446-459", "files": []},
           "fasta": {"k": None, "alphabet": "DNA5", "source": "src/io/test/mpi_test_fasta_seq_parse.cpp:398-403", "files": []}}
    fq = open(os.path.join(REF, "src/io/test/mpi_test_fastq_seq_parse.cpp")).read()
    b = _block(fq, "INSTANTIATE_TEST_CASE_P(Bliss, FASTQParseTest", "));")
    for m in re.finditer(r"^\s*TestFileInfo\((\d+),\s*(\d+),\s*(\d+),\s*std::string\(\"/test/data/([^\"]+)\"\)", b, re.M):
        out["fastq"]["files"].append({"file": m.group(4), "records": int(m.group(1)), "kmers": int(m.group(2)), "bytes": int(m.group(3))})
    fa = open(os.path.join(REF, "src/io/test/mpi_test_fasta_seq_parse.cpp")).read()
    km = re.search(r"class FASTAParseTest.*?kmer_size\s*=\s*(\d+)", fa, re.S)
    out["fasta"]["k"] = int(km.group(1)) if km else None
    b = _block(fa, "INSTANTIATE_TEST_CASE_P(Bliss, FASTAParseTest", "));")
    for m in re.finditer(r"^\s*TestFileInfo\((\d+),\s*(\d+),\s*(\d+),\s*std::string\(\"/test/data/([^\"]+)\"\)", b, re.M):
        out["fasta"]["files"].append({"file": m.group(4), "records": int(m.group(1)), "kmers": int(m.group(2)), "bytes": int(m.group(3))})
    # the same tables behind the filtering sequence iterators: NoN*ParseTest (NFilterSequencesIterator) and Split*ParseTest
    # (NSplitSequencesIterator). "records" is what those tests count: every sequence the iterator yields.
    pat = r"^\s*TestFileInfo\((\d+),\s*(\d+),\s*(\d+),\s*std::string\(\"/test/data/([^\"]+)\"\)"
    for fmt, src, stem in (("fastq", fq, "FASTQParseTest"), ("fasta", fa, "FASTAParseTest")):
        for key, cls in (("n_filter", "NoN" + stem), ("n_split", "Split" + stem)):
            b = _block(src, "INSTANTIATE_TEST_CASE_P(Bliss, " + cls, "));")
            out[fmt][key] = [{"file": m.group(4), "yielded": int(m.group(1)), "kmers": int(m.group(2)), "bytes": int(m.group(3))}
                             for m in re.finditer(pat, b, re.M)]
    return out


def quality_lut():
    """Illumina18QualityScoreCodec<float>::DecodeLUT (src/index/quality_scores.hpp:113-211) as data:
    96 values; entry 0 = lowest(); MinScore = 0 keeps the literals of entries 1 and 2."""
    src = open(os.path.join(REF, "src/index/quality_scores.hpp")).read()
    b = _block(src, "static constexpr LUTType DecodeLUT = {{", "}};")
    vals = []
    for line in b.splitlines()[1:]:
        line = line.split("//")[0].strip().rstrip(",")
        if not line:
            continue
        if "lowest()" in line and "?" not in line:
            vals.append("lowest")
        elif "?" in line:
            vals.append(line.split(":")[-1].strip())
        else:
            vals.append(line)
    assert len(vals) == 96, len(vals)
    return {"source": "src/index/quality_scores.hpp:113-211", "codec": "QualityScoreCodec<float,33,126,0>", "decode_lut": vals}


def alphabet_tables():
    """FROM_ASCII of DNA_T, DNA6_T (= DNA5), RNA_T, RNA6_T (= RNA5), DNA16_T (src/common/alphabets.hpp) as data: 256 codes each"""
    src = open(os.path.join(REF, "src/common/alphabets.hpp")).read()
    out = {"source": "src/common/alphabets.hpp (FROM_ASCII of DNA_T:139-161, DNA6_T:225-248, RNA_T:378-400, RNA6_T:459-480)"}
    for name in ("DNA_T", "DNA6_T", "RNA_T", "RNA6_T", "DNA16_T"):
        at = src.index("struct " + name + " ")
        b = _block(src[at:], "FROM_ASCII =", "}};")
        vals = []
        for line in b.splitlines()[1:]:
            line = line.split("//")[0]
            vals += [int(x, 0) for x in re.findall(r"0x[0-9A-Fa-f]+|\d+", line)]
        assert len(vals) == 256, (name, len(vals))
        out[name] = vals
    return out


def main():
    with open(os.path.join(HERE, "alphabet_tables.json"), "w") as f:
        json.dump(alphabet_tables(), f)
    with open(os.path.join(HERE, "quality_lut.json"), "w") as f:
        json.dump(quality_lut(), f, indent=1)
    with open(os.path.join(HERE, "kmer_golden.json"), "w") as f:
        json.dump(kmer_golden(), f, indent=1)
    pg = parse_golden()
    with open(os.path.join(HERE, "parse_golden.json"), "w") as f:
        json.dump(pg, f, indent=1)
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    names = {e["file"] for sect in pg.values() for e in sect["files"]}
    for name in sorted(names):
        src = os.path.join(REF, "test/data", name)
        if os.path.exists(src) and os.path.getsize(src) < 200_000:
            shutil.copyfile(src, os.path.join(HERE, "data", name))
    print("wrote golden fixtures for", len(names), "files")


if __name__ == "__main__":
    main()
