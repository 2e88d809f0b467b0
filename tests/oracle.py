"""ctypes binding of the CPU oracle (oracle/libkmerind_oracle.so) and of the
reference hash library built from the reference's vendored sources
(oracle/_ref/libkmerind_refhash.so).  TEST-SIDE ONLY: nothing under
kmerind_amd/ may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = os.path.join(ORACLE_DIR, "libkmerind_oracle.so")
_REF = os.path.join(ORACLE_DIR, "_ref", "libkmerind_refhash.so")

DNA, DNA5, RNA, RNA5, DNA16 = 0, 1, 2, 3, 4
SINGLE, CANONICAL, BIMOLECULE = 0, 1, 2
MURMUR, FARM, IDENTITY, STD = 0, 1, 2, 3
FASTQ, FASTA = 0, 1


class KSpec(C.Structure):
    _fields_ = [("k", C.c_uint32), ("alphabet", C.c_uint32), ("bits_per_char", C.c_uint32),
                ("n_bits", C.c_uint32), ("n_words", C.c_uint32), ("n_bytes", C.c_uint32)]


class Record(C.Structure):
    _fields_ = [("record_offset", C.c_uint64), ("record_size", C.c_uint64), ("seq_begin", C.c_uint64),
                ("seq_end", C.c_uint64), ("qual_begin", C.c_uint64), ("qual_end", C.c_uint64),
                ("seq_index", C.c_uint64)]


def _build():
    src = os.path.join(ORACLE_DIR, "kmerind_oracle.c")
    if (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libkmerind_oracle.so"], stdout=subprocess.DEVNULL)


_build()
lib = C.CDLL(_LIB)
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_SP = C.POINTER(KSpec)

lib.orc_kspec_init.argtypes = [_SP, C.c_uint32, C.c_uint32]
lib.orc_from_ascii.argtypes = [C.c_uint32, C.c_uint8]
lib.orc_from_ascii.restype = C.c_uint8
lib.orc_kmer_next_from_char.argtypes = [_SP, _u64p, C.c_uint8]
lib.orc_kmer_reverse.argtypes = [_SP, _u64p, _u64p]
lib.orc_kmer_revcomp.argtypes = [_SP, _u64p, _u64p]
lib.orc_kmer_less.argtypes = [_SP, _u64p, _u64p]
lib.orc_kmers_revcomp.argtypes = [_SP, _u64p, C.c_size_t, _u64p]
lib.orc_kmers_canonical.argtypes = [_SP, _u64p, C.c_size_t, _u64p]
lib.orc_murmur3_x64_128.argtypes = [C.c_void_p, C.c_int, C.c_uint32, _u64p]
lib.orc_farm_hash64_with_seed.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
lib.orc_farm_hash64_with_seed.restype = C.c_uint64
lib.orc_set_farm_ndebug.argtypes = [C.c_int]
lib.orc_kmers_hash.argtypes = [_SP, C.c_uint32, C.c_int, _u64p, C.c_size_t, _u64p]
lib.orc_key_to_rank.argtypes = [_SP, C.c_uint32, C.c_uint32, _u64p, C.c_size_t, C.c_uint32, _u32p]
lib.orc_key_to_rank_ex.argtypes = [_SP, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, C.c_size_t, C.c_uint32, _u32p]
lib.orc_fastq_records.argtypes = [_u8p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_size_t]
lib.orc_fastq_records.restype = C.c_long
lib.orc_fasta_records.argtypes = [_u8p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_size_t]
lib.orc_fasta_records.restype = C.c_long
lib.orc_extract.argtypes = [_SP, C.c_uint32, _u8p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
lib.orc_extract.restype = C.c_long
lib.orc_extract_filtered.argtypes = [_SP, C.c_uint32, C.c_uint32, _u8p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
lib.orc_extract_filtered.restype = C.c_long
lib.orc_qual_lut.argtypes = [C.c_uint8]
lib.orc_qual_lut.restype = C.c_float
lib.orc_stable_bucket.argtypes = [_u32p, C.c_size_t, C.c_uint32, _u64p, _u64p]
lib.orc_count_map_create.argtypes = [_SP, C.c_uint32, C.c_uint32]
lib.orc_count_map_create.restype = C.c_void_p
lib.orc_count_map_destroy.argtypes = [C.c_void_p]
lib.orc_count_map_insert.argtypes = [C.c_void_p, _u64p, C.c_size_t]
lib.orc_count_map_size.argtypes = [C.c_void_p]
lib.orc_count_map_size.restype = C.c_size_t
lib.orc_count_map_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_count_map_export.restype = C.c_size_t
lib.orc_count_map_count.argtypes = [C.c_void_p, _u64p, C.c_size_t, C.c_void_p, C.c_void_p]
lib.orc_count_map_count.restype = C.c_size_t
lib.orc_count_map_find.argtypes = [C.c_void_p, _u64p, C.c_size_t, C.c_void_p, C.c_void_p]
lib.orc_count_map_find.restype = C.c_size_t
lib.orc_count_map_erase.argtypes = [C.c_void_p, _u64p, C.c_size_t]
lib.orc_count_map_erase.restype = C.c_size_t
lib.orc_multi_map_create.argtypes = [_SP, C.c_uint32, C.c_uint32, C.c_uint32]
lib.orc_multi_map_create.restype = C.c_void_p
lib.orc_multi_map_destroy.argtypes = [C.c_void_p]
lib.orc_multi_map_insert.argtypes = [C.c_void_p, _u64p, _u64p, C.c_size_t]
lib.orc_multi_map_size.argtypes = [C.c_void_p]
lib.orc_multi_map_size.restype = C.c_size_t
lib.orc_multi_map_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_multi_map_export.restype = C.c_size_t
lib.orc_multi_map_count.argtypes = [C.c_void_p, _u64p, C.c_size_t, C.c_void_p, C.c_void_p]
lib.orc_multi_map_count.restype = C.c_size_t
lib.orc_multi_map_find.argtypes = [C.c_void_p, _u64p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
lib.orc_multi_map_find.restype = C.c_size_t
lib.orc_multi_map_erase.argtypes = [C.c_void_p, _u64p, C.c_size_t]
lib.orc_multi_map_erase.restype = C.c_size_t
lib.orc_dbg_parse.argtypes = [_SP, _u8p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
lib.orc_dbg_parse.restype = C.c_long
lib.orc_dbg_parse_fmt.argtypes = [_SP, C.c_uint32, _u8p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
lib.orc_dbg_parse_fmt.restype = C.c_long
lib.orc_dbg_edges_revcomp.argtypes = [C.c_uint8]
lib.orc_dbg_edges_revcomp.restype = C.c_uint8
lib.orc_dbg_map_create.argtypes = [_SP, C.c_uint32, C.c_int]
lib.orc_dbg_map_create.restype = C.c_void_p
lib.orc_dbg_map_destroy.argtypes = [C.c_void_p]
lib.orc_dbg_map_insert.argtypes = [C.c_void_p, _u64p, _u8p, C.c_size_t]
lib.orc_dbg_map_size.argtypes = [C.c_void_p]
lib.orc_dbg_map_size.restype = C.c_size_t
lib.orc_dbg_map_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.orc_dbg_map_export.restype = C.c_size_t
lib.orc_dbg_map_find.argtypes = [C.c_void_p, _u64p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
lib.orc_dbg_map_find.restype = C.c_size_t
lib.orc_bench_count_index.argtypes = [_u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
lib.orc_bench_count_index.restype = C.c_double
lib.orc_count_full.argtypes = [_u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
lib.orc_count_full.restype = C.c_int
lib.orc_fastq_align.argtypes = [_u8p, C.c_size_t, C.c_size_t]
lib.orc_fastq_align.restype = C.c_size_t


def kspec(k, alphabet=DNA):
    s = KSpec()
    if lib.orc_kspec_init(C.byref(s), k, alphabet) != 0:
        raise ValueError("bad kspec")
    return s


def _as_bytes(data):
    if isinstance(data, (bytes, bytearray)):
        return np.frombuffer(bytes(data), dtype=np.uint8).copy()
    return np.ascontiguousarray(data, dtype=np.uint8)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def kmers_from_string(s, text):
    """all k-mers of an ASCII string (no EOLs), as (n, n_words) uint64"""
    text = text.encode() if isinstance(text, str) else text
    n = max(0, len(text) - s.k + 1)
    out = np.zeros((n, s.n_words), dtype=np.uint64)
    km = np.zeros(s.n_words, dtype=np.uint64)
    for i, c in enumerate(text):
        lib.orc_kmer_next_from_char(C.byref(s), km, lib.orc_from_ascii(s.alphabet, c))
        if i >= s.k - 1:
            out[i - s.k + 1] = km
    return out


def revcomp(s, kmers):
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, s.n_words)
    out = np.empty_like(kmers)
    lib.orc_kmers_revcomp(C.byref(s), kmers, kmers.shape[0], out)
    return out


def canonical(s, kmers):
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, s.n_words)
    out = np.empty_like(kmers)
    lib.orc_kmers_canonical(C.byref(s), kmers, kmers.shape[0], out)
    return out


def kmer_hash(s, which, prefix, kmers):
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, s.n_words)
    out = np.empty(kmers.shape[0], dtype=np.uint64)
    lib.orc_kmers_hash(C.byref(s), which, int(prefix), kmers, kmers.shape[0], out)
    return out


def key_to_rank(s, dist_hash, strand, kmers, p, dist_trans=0):
    """dist_trans: 0 = the strand model's own DistTrans, 1 = lex_less, 2 = xor_rev_comp (single-strand model)"""
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, s.n_words)
    out = np.empty(kmers.shape[0], dtype=np.uint32)
    lib.orc_key_to_rank_ex(C.byref(s), dist_hash, strand, dist_trans, kmers, kmers.shape[0], p, out)
    return out


def records(data, fmt=FASTQ, file_offset=0):
    b = _as_bytes(data)
    fn = lib.orc_fastq_records if fmt == FASTQ else lib.orc_fasta_records
    n = fn(b, b.size, file_offset, None, 0)
    if n < 0:
        raise ValueError("parse error")
    arr = (Record * max(n, 1))()
    fn(b, b.size, file_offset, C.cast(arr, C.c_void_p), n)
    return [arr[i] for i in range(n)]


SEQ_ALL, SEQ_N_FILTER, SEQ_N_SPLIT = 0, 1, 2


def extract(s, data, fmt=FASTQ, file_offset=0, want_ids=False, want_quals=False, seq_filter=SEQ_ALL):
    """reference parser output for a whole buffer: dict(kmers, ids, quals, n_seqs, n_yield); seq_filter = the SeqIterType"""
    b = _as_bytes(data)
    nseq, nyield = C.c_size_t(0), C.c_size_t(0)
    n = lib.orc_extract_filtered(C.byref(s), fmt, seq_filter, b, b.size, file_offset, None, None, None, 0, C.byref(nseq), C.byref(nyield))
    if n < 0:
        raise ValueError("parse error")
    kmers = np.zeros((n, s.n_words), dtype=np.uint64)
    ids = np.zeros(n, dtype=np.uint64) if want_ids else None
    quals = np.zeros(n, dtype=np.float32) if want_quals else None
    lib.orc_extract_filtered(C.byref(s), fmt, seq_filter, b, b.size, file_offset, _ptr(kmers), _ptr(ids), _ptr(quals), n,
                             C.byref(nseq), C.byref(nyield))
    return {"kmers": kmers, "ids": ids, "quals": quals, "n_seqs": nseq.value, "n_yield": nyield.value}


class CountMap:
    def __init__(self, s, strand=CANONICAL, store_hash=MURMUR):
        self.s = s
        self.h = lib.orc_count_map_create(C.byref(s), strand, store_hash)

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_count_map_destroy(self.h)
            self.h = None

    def insert(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.s.n_words)
        lib.orc_count_map_insert(self.h, kmers, kmers.shape[0])

    def size(self):
        return lib.orc_count_map_size(self.h)

    def export(self):
        n = self.size()
        keys = np.zeros((n, self.s.n_words), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint32)
        lib.orc_count_map_export(self.h, _ptr(keys), _ptr(counts))
        return keys, counts

    def count(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.s.n_words)
        keys = np.zeros((q.shape[0], self.s.n_words), dtype=np.uint64)
        cnt = np.zeros(q.shape[0], dtype=np.uint64)
        n = lib.orc_count_map_count(self.h, q, q.shape[0], _ptr(keys), _ptr(cnt))
        return keys[:n], cnt[:n]

    def find(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.s.n_words)
        keys = np.zeros((q.shape[0], self.s.n_words), dtype=np.uint64)
        cnt = np.zeros(q.shape[0], dtype=np.uint32)
        n = lib.orc_count_map_find(self.h, q, q.shape[0], _ptr(keys), _ptr(cnt))
        return keys[:n], cnt[:n]

    def erase(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.s.n_words)
        return lib.orc_count_map_erase(self.h, q, q.shape[0])


def dbg_parse(s, data, fmt=None):
    """de_bruijn_parser over a FASTQ (or, fmt=FASTA, FASTA) buffer: (k-mers as parsed, DNA16 edge bytes: left << 4 | right)"""
    b = _as_bytes(data)
    fmt = FASTQ if fmt is None else fmt
    n = lib.orc_dbg_parse_fmt(C.byref(s), fmt, b, b.size, None, None, 0)
    if n < 0:
        raise ValueError("parse error")
    kmers = np.zeros((n, s.n_words), dtype=np.uint64)
    edges = np.zeros(n, dtype=np.uint8)
    lib.orc_dbg_parse_fmt(C.byref(s), fmt, b, b.size, _ptr(kmers), _ptr(edges), n)
    return kmers, edges


class DbgMap:
    """de_bruijn_nodes_distributed restatement: nodes = (k-mer, [out ACGT, in ACGT, self]); canonical=True gives every node
    in the orientation of its lexicographically smaller strand"""

    def __init__(self, s, exists_only=False, store_hash=MURMUR):
        self.s = s
        self.h = lib.orc_dbg_map_create(C.byref(s), store_hash, int(exists_only))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_dbg_map_destroy(self.h)
            self.h = None

    def insert(self, kmers, edges):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.s.n_words)
        edges = np.ascontiguousarray(edges, dtype=np.uint8)
        assert edges.shape[0] == kmers.shape[0]
        lib.orc_dbg_map_insert(self.h, kmers, edges, kmers.shape[0])

    def size(self):
        return lib.orc_dbg_map_size(self.h)

    def export(self, canonical=True):
        n = self.size()
        keys = np.zeros((n, self.s.n_words), dtype=np.uint64)
        counts = np.zeros((n, 9), dtype=np.uint32)
        lib.orc_dbg_map_export(self.h, _ptr(keys), _ptr(counts), int(canonical))
        return keys, counts

    def find(self, q, canonical=True):
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.s.n_words)
        keys = np.zeros((q.shape[0], self.s.n_words), dtype=np.uint64)
        counts = np.zeros((q.shape[0], 9), dtype=np.uint32)
        n = lib.orc_dbg_map_find(self.h, q, q.shape[0], _ptr(keys), _ptr(counts), int(canonical))
        return keys[:n], counts[:n]


class MultiMap:
    """::dsc::unordered_multimap restatement (PositionIndex): values are `vw` u64 words"""

    def __init__(self, s, strand=CANONICAL, vw=1, store_hash=MURMUR):
        self.s, self.vw = s, vw
        self.h = lib.orc_multi_map_create(C.byref(s), strand, store_hash, vw)

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_multi_map_destroy(self.h)
            self.h = None

    def insert(self, kmers, values):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.s.n_words)
        values = np.ascontiguousarray(values, dtype=np.uint64).reshape(kmers.shape[0], self.vw)
        lib.orc_multi_map_insert(self.h, kmers, values, kmers.shape[0])

    def size(self):
        return lib.orc_multi_map_size(self.h)

    def export(self):
        n = self.size()
        keys = np.zeros((n, self.s.n_words), dtype=np.uint64)
        vals = np.zeros((n, self.vw), dtype=np.uint64)
        lib.orc_multi_map_export(self.h, _ptr(keys), _ptr(vals))
        return keys, vals

    def count(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.s.n_words)
        keys = np.zeros((q.shape[0], self.s.n_words), dtype=np.uint64)
        cnt = np.zeros(q.shape[0], dtype=np.uint64)
        n = lib.orc_multi_map_count(self.h, q, q.shape[0], _ptr(keys), _ptr(cnt))
        return keys[:n], cnt[:n]

    def find(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.s.n_words)
        n = lib.orc_multi_map_find(self.h, q, q.shape[0], None, None, 0)
        keys = np.zeros((n, self.s.n_words), dtype=np.uint64)
        vals = np.zeros((n, self.vw), dtype=np.uint64)
        lib.orc_multi_map_find(self.h, q, q.shape[0], _ptr(keys), _ptr(vals), n)
        return keys, vals

    def erase(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.s.n_words)
        return lib.orc_multi_map_erase(self.h, q, q.shape[0])


def sorted_rows(*cols):
    """rows (concatenated 2-D columns) in lexicographic order, for multiset comparison"""
    m = np.concatenate([np.asarray(c, dtype=np.uint64).reshape(len(cols[0]), -1) for c in cols], axis=1)
    order = np.lexsort([m[:, i] for i in range(m.shape[1] - 1, -1, -1)])
    return m[order]


def sorted_pairs(keys, vals):
    """canonical ordering of a (key, value) multiset for comparisons"""
    if len(vals) == 0:
        return np.zeros((0, 1), dtype=np.uint64), np.asarray(vals)
    keys = np.asarray(keys, dtype=np.uint64).reshape(len(vals), -1)
    cols = [np.asarray(vals)] + [keys[:, w] for w in range(keys.shape[1])]
    order = np.lexsort(cols)
    return keys[order], np.asarray(vals)[order]


def bench_count_index(data, k, strand, threads):
    b = _as_bytes(data)
    nk, nd = C.c_uint64(0), C.c_uint64(0)
    t = lib.orc_bench_count_index(b, b.size, k, strand, threads, C.byref(nk), C.byref(nd))
    return t, nk.value, nd.value


def count_full(data, k, strand, threads, slices):
    """the thread-rank build of a whole FASTQ buffer: dict(kmers, distinct, sum_counts, sum_key_count, xor_mix) (sums mod 2^64)"""
    b = _as_bytes(data)
    out = (C.c_uint64 * 5)()
    rc = lib.orc_count_full(b, b.size, k, strand, threads, slices, out)
    assert rc == 0
    return dict(kmers=out[0], distinct=out[1], sum_counts=out[2], sum_key_count=out[3], xor_mix=out[4])


def map_checksums(keys, counts):
    """the same checksums from (keys, counts) arrays of a one-word count map"""
    k = np.asarray(keys, dtype=np.uint64).reshape(-1)
    c = np.asarray(counts).astype(np.uint64)
    with np.errstate(over="ignore"):
        skc = int((k * c).sum(dtype=np.uint64))
        x = int(np.bitwise_xor.reduce(k * (np.uint64(2) * c + np.uint64(1)))) if k.size else 0
    return dict(distinct=int(k.size), sum_counts=int(c.sum(dtype=np.uint64)), sum_key_count=skc, xor_mix=x)


def fastq_align(data, pos):
    b = _as_bytes(data)
    return int(lib.orc_fastq_align(b, b.size, pos))


# ---- reference hash library (vendored MurmurHash3.cpp / farmhash.cc compiled as-is)
def ref_hash_lib(ndebug=False):
    path = _REF.replace(".so", "_ndebug.so") if ndebug else _REF
    if not os.path.exists(path):
        return None
    r = C.CDLL(path)
    mm = getattr(r, "_Z19MurmurHash3_x64_128PKvijPv")
    mm.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    mm.restype = None
    fh = getattr(r, "_ZN4util14Hash64WithSeedEPKcmm")
    fh.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    fh.restype = C.c_uint64
    return mm, fh
