"""Differential fuzz of the fused FASTQ count build and the extract path against the oracle: ragged read lengths (reads shorter
than k, single bases, reads longer than a scan tile), LF / CRLF line ends, lower case and non-ACGT characters, '+' lines that
repeat the name, missing final newline -- the record shapes FASTQParser::get_next_record accepts (fastq_loader.hpp:389-467)."""
import numpy as np
import pytest

from tests import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kmerind_amd as K
    c = K.Context(0)
    yield c
    c.close()


def _random_fastq(seed):
    rng = np.random.default_rng(seed)
    eol = b"\r\n" if seed % 3 == 1 else b"\n"
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTNacgtnRY", dtype=np.uint8)
    recs = []
    n = int(rng.integers(1, 400))
    for i in range(n):
        kind = rng.random()
        length = int(rng.integers(1, 12)) if kind < 0.1 else (int(rng.integers(9000, 20000)) if kind > 0.985 else int(rng.integers(12, 400)))
        seq = alphabet[rng.integers(0, alphabet.size, size=length)].tobytes()
        qual = rng.integers(33, 74, size=length, dtype=np.uint8).tobytes()
        name = b"r%d len=%d" % (i, length)
        plus = b"+" + (name if rng.random() < 0.3 else b"")
        recs.append(b"@" + name + eol + seq + eol + plus + eol + qual + eol)
    data = b"".join(recs)
    if seed % 4 == 2:
        data = data[:-len(eol)]                     # no newline at the end of the file
    return data


@pytest.mark.parametrize("seed", list(range(16)))
def test_fused_build_and_extract_follow_the_oracle_on_ragged_fastq(ctx, seed):
    import kmerind_amd as K
    data = _random_fastq(seed)
    k, alpha, oa = [(31, "DNA", orc.DNA), (15, "DNA", orc.DNA), (21, "DNA5", orc.DNA5), (33, "DNA", orc.DNA), (12, "DNA16", orc.DNA16)][seed % 5]
    flt = ["all", "n_split", "n_filter"][seed % 3]
    fl = {"all": orc.SEQ_ALL, "n_split": orc.SEQ_N_SPLIT, "n_filter": orc.SEQ_N_FILTER}[flt]
    s = orc.kspec(k, oa)
    ex = orc.extract(s, data, orc.FASTQ, want_ids=True, seq_filter=fl)
    gk, gi, gn = ctx.read_file(K.make_config(k, alpha, strand="single", index_kind="position", seq_filter=flt), data, with_ids=True)
    assert gn == ex["n_seqs"] and gk.shape == ex["kmers"].shape and (gk == ex["kmers"]).all() and (gi == ex["ids"]).all()
    idx = K.CountIndex(ctx, K.make_config(k, alpha, strand="canonical", seq_filter=flt))
    idx.build(data)
    m = orc.CountMap(s, orc.CANONICAL)
    m.insert(ex["kmers"])
    a, b = orc.sorted_pairs(*idx.to_vector()), orc.sorted_pairs(*m.export())
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    idx.close()
