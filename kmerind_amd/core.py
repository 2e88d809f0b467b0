"""Host-side mirror of the reference's operator interface for the k-mer index path.

Names follow the reference: CountIndex.insert / count / find / erase / size / local_size /
build (bliss::index::kmer::Index, src/index/kmer_index.hpp:98-394) and
KmerFileHelper.read_file (src/io/kmer_file_helper.hpp:550-633) -> Context.read_file."""
import ctypes as C
import os

import numpy as np

from . import _lib as L

lib = L.lib


def make_config(k, alphabet="DNA", strand="canonical", dist_hash="murmur", store_hash="murmur",
                index_kind="count", seq_format="fastq", farm_ndebug=False, seq_filter="all", dist_trans="model"):
    alpha = {"DNA": L.ALPHA_DNA, "DNA5": L.ALPHA_DNA5, "DNA6": L.ALPHA_DNA5,
             "RNA": L.ALPHA_RNA, "RNA5": L.ALPHA_RNA5, "RNA6": L.ALPHA_RNA5, "DNA16": L.ALPHA_DNA16}[alphabet]
    st = {"single": L.STRAND_SINGLE, "canonical": L.STRAND_CANONICAL, "bimolecule": L.STRAND_BIMOLECULE}[strand]
    hs = {"murmur": L.HASH_MURMUR, "farm": L.HASH_FARM, "identity": L.HASH_IDENTITY, "std": L.HASH_STD}
    kind = {"count": L.INDEX_COUNT, "position": L.INDEX_POSITION, "posqual": L.INDEX_POSQUAL}[index_kind]
    fmt = {"fastq": L.FMT_FASTQ, "fasta": L.FMT_FASTA}[seq_format]
    flt = {"all": L.SEQ_ALL, "n_filter": L.SEQ_N_FILTER, "n_split": L.SEQ_N_SPLIT}[seq_filter]
    dt = {"model": L.DIST_MODEL, "lex_less": L.DIST_LEX, "xor_rev_comp": L.DIST_XOR}[dist_trans]
    return L.Config(k, alpha, st, hs[dist_hash], hs[store_hash], kind, fmt, int(bool(farm_ndebug)), flt, dt)


def _u64(a, n_words=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if n_words is not None:
        a = a.reshape(-1, n_words)
    return a


class Context:
    """One GPU = one rank (stands where the reference takes an mxx::comm)."""

    def __init__(self, device=0, rank=0, nranks=1, stream=None):
        h = C.c_void_p()
        st = lib.kmi_ctx_create(device, rank, nranks, C.c_void_p(stream or 0), C.byref(h))
        if st != L.OK:
            raise L.KmiError(st, "kmi_ctx_create failed (no usable HIP device? there is no CPU fallback)")
        self.h = h
        self.rank, self.nranks, self.device = rank, nranks, device
        self.stream = int(stream or 0)

    def close(self):
        if getattr(self, "h", None):
            lib.kmi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st != L.OK:
            raise L.KmiError(st, (lib.kmi_last_error(self.h) or b"").decode())

    @staticmethod
    def shape(cfg):
        nw, nb, nby = C.c_uint32(), C.c_uint32(), C.c_uint32()
        if lib.kmi_kmer_shape(C.byref(cfg), C.byref(nw), C.byref(nb), C.byref(nby)) != L.OK:
            raise ValueError("invalid k-mer configuration")
        return nw.value, nb.value, nby.value

    # ---- device memory helpers (raw pointers; bench.py may pass torch data_ptr() instead)
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(lib.kmi_device_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, dptr):
        self.check(lib.kmi_device_free(self.h, C.c_void_p(dptr)))

    def to_device(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(lib.kmi_copy_to_device(self.h, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def to_host(self, arr, dptr):
        self.check(lib.kmi_copy_to_host(self.h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), arr.nbytes))

    def synchronize(self):
        self.check(lib.kmi_synchronize(self.h))

    # ---- array-level k-mer ops
    def _array_op(self, fn, cfg, kmers, out_dtype, per_key, *extra):
        nw = self.shape(cfg)[0]
        kmers = _u64(kmers, nw)
        n = kmers.shape[0]
        out = np.zeros((n, nw) if per_key else n, dtype=out_dtype)
        self.check(fn(self.h, C.byref(cfg), *extra[:2], kmers.ctypes.data_as(C.c_void_p), n, *extra[2:],
                      out.ctypes.data_as(C.c_void_p)))
        return out

    def revcomp(self, cfg, kmers):
        return self._array_op(lib.kmi_revcomp_host, cfg, kmers, np.uint64, True)

    def canonical(self, cfg, kmers):
        return self._array_op(lib.kmi_canonical_host, cfg, kmers, np.uint64, True)

    def hash(self, cfg, which, prefix, kmers):
        which = {"murmur": L.HASH_MURMUR, "farm": L.HASH_FARM, "identity": L.HASH_IDENTITY, "std": L.HASH_STD}.get(which, which)
        return self._array_op(lib.kmi_hash_host, cfg, kmers, np.uint64, False, which, int(prefix))

    def key_to_rank(self, cfg, kmers, nranks):
        nw = self.shape(cfg)[0]
        kmers = _u64(kmers, nw)
        out = np.zeros(kmers.shape[0], dtype=np.uint32)
        self.check(lib.kmi_key_to_rank_host(self.h, C.byref(cfg), kmers.ctypes.data_as(C.c_void_p), kmers.shape[0],
                                            nranks, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- KmerFileHelper::read_file_* equivalent on an in-memory, record-aligned partition
    def read_file(self, cfg, data, file_offset=0, with_ids=False, with_quals=False, fasta_block=None):
        """returns (kmers[n, n_words], n_seqs) in file order, as parsed (no strand transform);
        with_ids (position index kinds): (kmers, ids, n_seqs), ids = Short/LongSequenceKmerId words.
        fasta_block = (rank, nranks): `data` is a WHOLE FASTA file and the tuples of that rank's block of an equal split come back
        (kmi_extract_fasta_block_host: what read_file_* does on one rank of several)"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else \
            np.ascontiguousarray(data, dtype=np.uint8)
        t = L.Tuples()
        if fasta_block is not None:
            self.check(lib.kmi_extract_fasta_block_host(self.h, C.byref(cfg), buf.ctypes.data_as(C.c_void_p), buf.size,
                                                        int(fasta_block[0]), int(fasta_block[1]), C.byref(t)))
        else:
            self.check(lib.kmi_extract_host(self.h, C.byref(cfg), buf.ctypes.data_as(C.c_void_p), buf.size, file_offset,
                                            C.byref(t)))
        nw = self.shape(cfg)[0]
        n = t.n_tuples
        kmers = np.ctypeslib.as_array(t.kmers, shape=(n * nw,)).copy().reshape(n, nw) if n else \
            np.zeros((0, nw), dtype=np.uint64)
        nseq = t.n_seqs
        ids = None
        if with_ids:
            if not t.ids:
                lib.kmi_tuples_free(C.byref(t))
                raise ValueError("ids need a position index kind in the config")
            ids = np.ctypeslib.as_array(t.ids, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint64)
        if with_quals and not with_ids:
            with_ids = True
        quals = None
        if with_quals:
            if not t.quals:
                lib.kmi_tuples_free(C.byref(t))
                raise ValueError("qualities need index_kind='posqual'")
            quals = np.ctypeslib.as_array(t.quals, shape=(n,)).copy() if n else np.zeros(0, dtype=np.float32)
        lib.kmi_tuples_free(C.byref(t))
        if with_quals:
            return kmers, ids, quals, nseq
        if with_ids:
            return kmers, ids, nseq
        return kmers, nseq

    def set_fasta_partition(self, part=None):
        """what FASTAParser::init_parser learns from the neighbouring ranks, for the next FASTA extract / build calls
        (a dict from kmerind_amd.fileio.partition_fasta); None = whole file"""
        if part is None:
            self.check(lib.kmi_ctx_set_fasta_partition(self.h, None))
            return
        fp = L.FastaPartition(part["valid_bytes"], part["start_state"], part["at_line_start"], part["records_before"],
                              part["index_shift"], 0)
        self.check(lib.kmi_ctx_set_fasta_partition(self.h, C.byref(fp)))

    # ---- profiling
    def profile(self, on=True):
        self.check(lib.kmi_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.check(lib.kmi_profile_reset(self.h))

    def profile_get(self):
        arr = (L.KernelTime * 64)()
        n = C.c_size_t()
        self.check(lib.kmi_profile_get(self.h, arr, 64, C.byref(n)))
        return [{"name": arr[i].name.decode(), "total_ms": arr[i].total_ms, "launches": arr[i].launches,
                 "units": arr[i].units} for i in range(min(n.value, 64))]


def num_buckets():
    return int(lib.kmi_index_num_buckets())


class CountIndex:
    """bliss::index::kmer::CountIndex<counting map> on one rank (kmer_index.hpp:409-410)."""

    def __init__(self, ctx, cfg):
        self.ctx, self.cfg = ctx, cfg
        self.n_words = ctx.shape(cfg)[0]
        h = C.c_void_p()
        ctx.check(lib.kmi_index_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):   # (a context that is gone took the device blocks with it: nothing to hand back through it)
                lib.kmi_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def insert(self, kmers):
        kmers = _u64(kmers, self.n_words)
        self.ctx.check(lib.kmi_index_insert_host(self.h, kmers.ctypes.data_as(C.c_void_p), kmers.shape[0]))

    def insert_pairs(self, kmers, counts):
        """Index::insert(vector<pair<Kmer, count>>): every pair's count is added (distributed_unordered_map.hpp:1603-1618)"""
        kmers = _u64(kmers, self.n_words)
        rec = np.zeros((kmers.shape[0], self.n_words + 1), dtype=np.uint64)
        rec[:, :self.n_words] = kmers
        rec[:, self.n_words] = np.asarray(counts, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
        self.ctx.check(lib.kmi_index_insert_pairs_host(self.h, rec.ctypes.data_as(C.c_void_p), rec.shape[0]))

    def update_pairs(self, kmers, values, op="add"):
        """update(pairs, op) with a device-side updater (add / max / min / assign): stored keys only -> number of pairs applied"""
        kmers = _u64(kmers, self.n_words)
        rec = np.zeros((kmers.shape[0], self.n_words + 1), dtype=np.uint64)
        rec[:, :self.n_words] = kmers
        rec[:, self.n_words] = np.asarray(values, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
        n = C.c_uint64()
        code = {"add": 0, "max": 1, "min": 2, "assign": 3}[op]
        self.ctx.check(lib.kmi_index_update_pairs_host(self.h, rec.ctypes.data_as(C.c_void_p), rec.shape[0], code, C.byref(n)))
        return n.value

    def insert_device(self, dptr, n, transformed=False):
        """transformed=True: the keys already went through the InputTransform (routed keys after the exchange)"""
        fn = lib.kmi_index_insert_transformed_dev if transformed else lib.kmi_index_insert_dev
        self.ctx.check(fn(self.h, C.c_void_p(dptr), n))

    def build(self, data, file_offset=0):
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else \
            np.ascontiguousarray(data, dtype=np.uint8)
        self.ctx.check(lib.kmi_index_build_host(self.h, buf.ctypes.data_as(C.c_void_p), buf.size, file_offset))

    def build_device(self, dptr, nbytes, file_offset=0):
        self.ctx.check(lib.kmi_index_build_dev(self.h, C.c_void_p(dptr), nbytes, file_offset))

    def clear(self):
        self.ctx.check(lib.kmi_index_clear(self.h))

    def local_size(self):
        n = C.c_uint64()
        self.ctx.check(lib.kmi_index_local_size(self.h, C.byref(n)))
        return n.value

    size = local_size  # single-rank view; kmerind_amd.dist adds the all-reduce

    def owner_ranks(self):
        """1, or the rank count of the build through exchanged super-k-mers that filled this index (kmi_index_sk_consume_dev)"""
        n = C.c_uint32()
        self.ctx.check(lib.kmi_index_owner_ranks(self.h, C.byref(n)))
        return n.value

    # combine-first distributed insert (kmerind_hip.h): split the entries by destination rank / merge received parts
    def split_by_rank_device(self, nranks, keys_dptr, counts_dptr, capacity, bucket_counts_dptr):
        """-> send counts per rank (numpy uint64). Device buffers: keys [capacity * n_words] u64, counts [capacity] u32,
        bucket counts [nranks * num_buckets()] u32."""
        sc = np.zeros(nranks, dtype=np.uint64)
        self.ctx.check(lib.kmi_index_split_by_rank_dev(self.h, nranks, C.c_void_p(keys_dptr), C.c_void_p(counts_dptr), capacity,
                                                       C.c_void_p(bucket_counts_dptr), sc.ctypes.data_as(C.c_void_p)))
        return sc

    def merge_parts_device(self, nparts, keys_dptr, counts_dptr, bucket_counts_dptr):
        self.ctx.check(lib.kmi_index_merge_parts_dev(self.h, nparts, C.c_void_p(keys_dptr), C.c_void_p(counts_dptr),
                                                     C.c_void_p(bucket_counts_dptr)))

    def to_vector(self):
        n = self.local_size()
        keys = np.zeros((n, self.n_words), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint32)
        got = C.c_uint64()
        self.ctx.check(lib.kmi_index_export_host(self.h, keys.ctypes.data_as(C.c_void_p),
                                                 counts.ctypes.data_as(C.c_void_p), n, C.byref(got)))
        return keys[:got.value], counts[:got.value]

    def _query(self, fn, q):
        q = _u64(q, self.n_words)
        r = L.Results()
        self.ctx.check(fn(self.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        n = r.n
        if n:
            keys = np.ctypeslib.as_array(r.keys, shape=(n * self.n_words,)).copy().reshape(n, self.n_words)
            vals = np.ctypeslib.as_array(r.values, shape=(n,)).copy()
        else:
            keys = np.zeros((0, self.n_words), dtype=np.uint64)
            vals = np.zeros(0, dtype=np.uint64)
        lib.kmi_results_free(C.byref(r))
        return keys, vals

    def count(self, q):
        return self._query(lib.kmi_index_count_host, q)

    def find(self, q):
        return self._query(lib.kmi_index_find_host, q)

    def exists(self, q):
        """densehash exists() (distributed_densehash_map.hpp:1465-1560): one byte per input key, in input order"""
        q = _u64(q, self.n_words)
        keys, cnt = self.count(q)
        tq = q if self.cfg.strand == L.STRAND_SINGLE else self.ctx.canonical(self.cfg, q)
        have = {tuple(k) for k, c in zip(keys.tolist(), np.asarray(cnt).reshape(len(keys), -1)[:, 0].tolist()) if c}
        return np.fromiter((tuple(k) in have for k in tq.tolist()), dtype=np.uint8, count=tq.shape[0])

    def erase(self, q):
        q = _u64(q, self.n_words)
        n = C.c_uint64()
        self.ctx.check(lib.kmi_index_erase_host(self.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(n)))
        return n.value


class PositionIndex(CountIndex):
    """bliss::index::kmer::PositionIndex<unordered_multimap> on one rank (kmer_index.hpp:402-403):
    every (k-mer, ShortSequenceKmerId) tuple is kept."""

    def __init__(self, ctx, cfg):
        if cfg.index_kind == L.INDEX_COUNT:
            raise ValueError("PositionIndex needs index_kind='position'")
        super().__init__(ctx, cfg)
        self.value_words = 1 if cfg.index_kind == L.INDEX_POSITION else 2

    def insert(self, kmers, values):
        kmers = _u64(kmers, self.n_words)
        values = np.ascontiguousarray(values, dtype=np.uint64).reshape(kmers.shape[0], self.value_words)
        self.ctx.check(lib.kmi_index_insert_tuples_host(self.h, kmers.ctypes.data_as(C.c_void_p),
                                                        values.ctypes.data_as(C.c_void_p), kmers.shape[0]))

    def to_vector(self):
        n = self.local_size()
        keys = np.zeros((n, self.n_words), dtype=np.uint64)
        vals = np.zeros((n, self.value_words), dtype=np.uint64)
        got = C.c_uint64()
        self.ctx.check(lib.kmi_index_export_tuples_host(self.h, keys.ctypes.data_as(C.c_void_p),
                                                        vals.ctypes.data_as(C.c_void_p), n, C.byref(got)))
        return keys[:got.value], vals[:got.value]

    def _query(self, fn, q):
        q = _u64(q, self.n_words)
        r = L.Results()
        self.ctx.check(fn(self.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        n, vw = r.n, self.value_words
        if n:
            keys = np.ctypeslib.as_array(r.keys, shape=(n * self.n_words,)).copy().reshape(n, self.n_words)
            vals = np.ctypeslib.as_array(r.values, shape=(n * vw,)).copy().reshape(n, vw)
        else:
            keys = np.zeros((0, self.n_words), dtype=np.uint64)
            vals = np.zeros((0, vw), dtype=np.uint64)
        lib.kmi_results_free(C.byref(r))
        return keys, vals

    def count(self, q):
        keys, vals = self._query(lib.kmi_index_count_host, q)
        return keys, vals[:, 0]


class DeBruijnNodes:
    """de_bruijn_engine<NodeMap> on one rank (test/test/debruijn/de_bruijn_construct_engine.hpp:241-245): nodes =
    (k-mer, [out A C G T, in A C G T, occurrences]); exists_only = node::edge_exists (0 / 1 per edge, no occurrence count).
    Every node is kept under its lexicographically smaller strand (include/kmerind_hip.h)."""

    def __init__(self, ctx, cfg, exists_only=False):
        self.ctx, self.cfg = ctx, cfg
        self.n_words = ctx.shape(cfg)[0]
        h = C.c_void_p()
        ctx.check(lib.kmi_dbg_create(ctx.h, C.byref(cfg), 1 if exists_only else 0, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib.kmi_dbg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parse(self, data):
        """de_bruijn_parser over a FASTQ buffer -> (k-mers as parsed, edge bytes)"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data, dtype=np.uint8)
        d = self.ctx.alloc(buf.size + 64)
        try:
            self.ctx.to_device(d, buf)
            n = C.c_uint64()
            self.ctx.check(lib.kmi_dbg_parse_dev(self.ctx.h, C.byref(self.cfg), C.c_void_p(d), buf.size, None, 0, C.byref(n)))
            rec = np.zeros((n.value, self.n_words + 1), dtype=np.uint64)
            if n.value:
                dr = self.ctx.alloc(rec.nbytes)
                try:
                    self.ctx.check(lib.kmi_dbg_parse_dev(self.ctx.h, C.byref(self.cfg), C.c_void_p(d), buf.size, C.c_void_p(dr), n.value, C.byref(n)))
                    self.ctx.to_host(rec, dr)
                finally:
                    self.ctx.free(dr)
        finally:
            self.ctx.free(d)
        return rec[:, :self.n_words].copy(), rec[:, self.n_words].astype(np.uint8)

    def build(self, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data, dtype=np.uint8)
        self.ctx.check(lib.kmi_dbg_build_host(self.h, buf.ctypes.data_as(C.c_void_p), buf.size))

    def build_device(self, dptr, nbytes):
        self.ctx.check(lib.kmi_dbg_build_dev(self.h, C.c_void_p(dptr), nbytes))

    def insert(self, kmers, edges):
        """insert(vector<pair<Kmer, uint8_t>>): tuples as the parser emits them, either strand"""
        kmers = _u64(kmers, self.n_words)
        rec = np.zeros((kmers.shape[0], self.n_words + 1), dtype=np.uint64)
        rec[:, :self.n_words] = kmers
        rec[:, self.n_words] = np.asarray(edges, dtype=np.uint64) & np.uint64(0xFF)
        self.ctx.check(lib.kmi_dbg_insert_host(self.h, rec.ctypes.data_as(C.c_void_p), rec.shape[0]))

    def clear(self):
        self.ctx.check(lib.kmi_dbg_clear(self.h))

    def erase(self, q):
        """erase(keys): the nodes of these k-mers (either strand) leave the map; returns how many did"""
        q = _u64(q, self.n_words)
        n = C.c_uint64()
        self.ctx.check(lib.kmi_dbg_erase_host(self.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(n)))
        return n.value

    def local_size(self):
        n = C.c_uint64()
        self.ctx.check(lib.kmi_dbg_local_size(self.h, C.byref(n)))
        return n.value

    size = local_size

    def to_vector(self):
        n = self.local_size()
        keys = np.zeros((n, self.n_words), dtype=np.uint64)
        counts = np.zeros((n, 9), dtype=np.uint32)
        got = C.c_uint64()
        self.ctx.check(lib.kmi_dbg_export_host(self.h, keys.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), n, C.byref(got)))
        return keys[:got.value], counts[:got.value]

    def find(self, q):
        q = _u64(q, self.n_words)
        r = L.Results()
        self.ctx.check(lib.kmi_dbg_find_host(self.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        n = r.n
        if n:
            keys = np.ctypeslib.as_array(r.keys, shape=(n * self.n_words,)).copy().reshape(n, self.n_words)
            vals = np.ctypeslib.as_array(r.values, shape=(n * 5,)).copy().view(np.uint32).reshape(n, 10)[:, :9].copy()
        else:
            keys = np.zeros((0, self.n_words), dtype=np.uint64)
            vals = np.zeros((0, 9), dtype=np.uint32)
        lib.kmi_results_free(C.byref(r))
        return keys, vals

    def count(self, q):
        q = _u64(q, self.n_words)
        r = L.Results()
        self.ctx.check(lib.kmi_dbg_count_host(self.h, q.ctypes.data_as(C.c_void_p), q.shape[0], C.byref(r)))
        n = r.n
        keys = np.ctypeslib.as_array(r.keys, shape=(n * self.n_words,)).copy().reshape(n, self.n_words) if n else np.zeros((0, self.n_words), np.uint64)
        vals = np.ctypeslib.as_array(r.values, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
        lib.kmi_results_free(C.byref(r))
        return keys, vals


def synth_fastq(seed, genome_len, n_reads, read_len=150, first_read=0, threads=None):
    """SURVEY.md 8(d) synthetic FASTQ as a numpy uint8 array (host)."""
    nbytes = lib.kmi_synth_fastq_bytes(n_reads, read_len)
    out = np.empty(nbytes, dtype=np.uint8)
    threads = threads or min(16, os.cpu_count() or 1)
    st = lib.kmi_synth_fastq(seed, genome_len, read_len, first_read, n_reads, out.ctypes.data_as(C.c_void_p), nbytes,
                             threads)
    if st != L.OK:
        raise ValueError("kmi_synth_fastq: bad arguments")
    return out
