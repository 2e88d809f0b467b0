"""ctypes binding of libkmerind_hip.so (the C ABI declared in include/kmerind_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at import,
and every call fails with KMI_ERR_DEVICE when no HIP device is usable."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMERIND_HIP_LIB points at another build of the same library (e.g. an installed copy)
LIB_PATH = os.environ.get("KMERIND_HIP_LIB") or os.path.join(_HERE, "libkmerind_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "kmerind_amd: %s is missing. Build it with `make -C kmerind_amd/csrc` "
        "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

OK, ERR_INVALID, ERR_DEVICE, ERR_PARSE, ERR_NOMEM, ERR_OVERFLOW, ERR_PEER = range(7)
ALPHA_DNA, ALPHA_DNA5, ALPHA_RNA, ALPHA_RNA5, ALPHA_DNA16 = 0, 1, 2, 3, 4
STRAND_SINGLE, STRAND_CANONICAL, STRAND_BIMOLECULE = 0, 1, 2
HASH_MURMUR, HASH_FARM, HASH_IDENTITY, HASH_STD = 0, 1, 2, 3
FMT_FASTQ, FMT_FASTA = 0, 1
INDEX_COUNT, INDEX_POSITION, INDEX_POSQUAL = 0, 1, 2
SEQ_ALL, SEQ_N_FILTER, SEQ_N_SPLIT = 0, 1, 2
DIST_MODEL, DIST_LEX, DIST_XOR = 0, 1, 2


class FastaPartition(C.Structure):
    _fields_ = [("valid_bytes", C.c_uint64), ("start_state", C.c_uint32), ("at_line_start", C.c_uint32),
                ("records_before", C.c_uint64), ("index_shift", C.c_uint32), ("reserved", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("k", C.c_uint32), ("alphabet", C.c_uint32), ("strand", C.c_uint32),
                ("dist_hash", C.c_uint32), ("store_hash", C.c_uint32), ("index_kind", C.c_uint32),
                ("seq_format", C.c_uint32), ("farm_ndebug", C.c_uint32), ("seq_filter", C.c_uint32),
                ("dist_trans", C.c_uint32)]


class Tuples(C.Structure):
    _fields_ = [("n_tuples", C.c_uint64), ("n_seqs", C.c_uint64), ("kmers", C.POINTER(C.c_uint64)),
                ("ids", C.POINTER(C.c_uint64)), ("quals", C.POINTER(C.c_float))]


class Results(C.Structure):
    _fields_ = [("n", C.c_uint64), ("keys", C.POINTER(C.c_uint64)), ("values", C.POINTER(C.c_uint64))]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("total_ms", C.c_double), ("launches", C.c_uint64), ("units", C.c_uint64)]


_P = C.c_void_p
_CFG = C.POINTER(Config)
_sz = C.c_size_t
_u64 = C.c_uint64
_u32 = C.c_uint32

# every symbol include/kmerind_hip.h declares, with its signature
SIGNATURES = {
    "kmi_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "kmi_ctx_destroy": (C.c_int, [_P]),
    "kmi_ctx_set_fasta_partition": (C.c_int, [_P, _P]),
    "kmi_ctx_reset_hints": (C.c_int, [_P]),
    "kmi_fasta_partition_dev": (C.c_int, [_P, _P, _sz, _u32, _u32, _P, _P]),
    "kmi_last_error": (C.c_char_p, [_P]),
    "kmi_kmer_shape": (C.c_int, [_CFG, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "kmi_free_host": (None, [_P]),
    "kmi_device_alloc": (C.c_int, [_P, _sz, C.POINTER(_P)]),
    "kmi_device_free": (C.c_int, [_P, _P]),
    "kmi_copy_to_device": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_copy_to_host": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_copy_on_device": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_synchronize": (C.c_int, [_P]),
    "kmi_revcomp_host": (C.c_int, [_P, _CFG, _P, _sz, _P]),
    "kmi_canonical_host": (C.c_int, [_P, _CFG, _P, _sz, _P]),
    "kmi_hash_host": (C.c_int, [_P, _CFG, _u32, C.c_int, _P, _sz, _P]),
    "kmi_key_to_rank_host": (C.c_int, [_P, _CFG, _P, _sz, _u32, _P]),
    "kmi_extract_host": (C.c_int, [_P, _CFG, _P, _sz, _u64, C.POINTER(Tuples)]),
    "kmi_tuples_free": (None, [C.POINTER(Tuples)]),
    "kmi_extract_count_dev": (C.c_int, [_P, _CFG, _P, _sz, C.POINTER(_u64), C.POINTER(_u64)]),
    "kmi_extract_dev": (C.c_int, [_P, _CFG, _P, _sz, _u64, _P, _P, _sz, C.POINTER(_u64), C.POINTER(_u64)]),
    "kmi_extract_records_dev": (C.c_int, [_P, _CFG, _P, _sz, _u64, _P, _sz, C.POINTER(_u64), C.POINTER(_u64)]),
    "kmi_fastq_partition_dev": (C.c_int, [_P, _P, _sz, _u32, _P]),
    "kmi_route_dev": (C.c_int, [_P, _CFG, _P, _sz, _u32, _P, _P]),
    "kmi_route_tuples_dev": (C.c_int, [_P, _CFG, _P, _sz, _u32, _u32, _P, _P]),
    "kmi_extract_route_records_dev": (C.c_int, [_P, _CFG, _P, _sz, _u64, _u32, _P, _sz, C.POINTER(_u64), C.POINTER(_u64), _P]),
    "kmi_extract_route_dev": (C.c_int, [_P, _P, _P, _sz, C.c_uint32, _P, _sz, C.POINTER(_u64), C.POINTER(_u64), _P]),
    "kmi_index_create": (C.c_int, [_P, _CFG, C.POINTER(_P)]),
    "kmi_index_destroy": (C.c_int, [_P]),
    "kmi_index_insert_host": (C.c_int, [_P, _P, _sz]),
    "kmi_index_insert_dev": (C.c_int, [_P, _P, _sz]),
    "kmi_index_insert_transformed_dev": (C.c_int, [_P, _P, _sz]),
    "kmi_index_insert_pairs_host": (C.c_int, [_P, _P, _sz]),
    "kmi_index_insert_pairs_dev": (C.c_int, [_P, _P, _sz]),
    "kmi_index_build_host": (C.c_int, [_P, _P, _sz, _u64]),
    "kmi_index_build_dev": (C.c_int, [_P, _P, _sz, _u64]),
    "kmi_index_clear": (C.c_int, [_P]),
    "kmi_index_set_seq_format": (C.c_int, [_P, C.c_uint32]),
    "kmi_index_set_seq_filter": (C.c_int, [_P, C.c_uint32]),
    "kmi_index_local_size": (C.c_int, [_P, C.POINTER(_u64)]),
    "kmi_index_export_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(_u64)]),
    "kmi_results_free": (None, [C.POINTER(Results)]),
    "kmi_index_count_host": (C.c_int, [_P, _P, _sz, C.POINTER(Results)]),
    "kmi_index_find_host": (C.c_int, [_P, _P, _sz, C.POINTER(Results)]),
    "kmi_index_erase_host": (C.c_int, [_P, _P, _sz, C.POINTER(_u64)]),
    "kmi_index_count_dev": (C.c_int, [_P, _P, _sz, _P, _P, C.POINTER(_u64)]),
    "kmi_index_find_dev": (C.c_int, [_P, _P, _sz, _P, _P, C.POINTER(_u64)]),
    "kmi_index_insert_tuples_host": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_index_insert_tuples_dev": (C.c_int, [_P, _P, _sz]),
    "kmi_index_export_tuples_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(_u64)]),
    "kmi_index_num_buckets": (C.c_uint32, []),
    "kmi_index_split_by_rank_dev": (C.c_int, [_P, _u32, _P, _P, _sz, _P, _P]),
    "kmi_index_merge_parts_dev": (C.c_int, [_P, _u32, _P, _P, _P]),
    "kmi_comm_unique_id": (C.c_int, [_P]),
    "kmi_comm_create": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "kmi_ctx_debug_counter": (C.c_int, [_P, _u32, C.POINTER(_u64)]),
    "kmi_release_cached_memory": (C.c_int, [C.c_int, C.POINTER(_u64)]),
    "kmi_comm_create_transport": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "kmi_comm_destroy": (C.c_int, [_P]),
    "kmi_comm_all_to_all_counts": (C.c_int, [_P, _P, _P]),
    "kmi_comm_all_to_all_v": (C.c_int, [_P, _P, _P, _P, _P, _sz]),
    "kmi_comm_allreduce_sum_u64": (C.c_int, [_P, C.POINTER(_u64)]),
    "kmi_index_insert_dist_host": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_index_insert_tuples_dist_host": (C.c_int, [_P, _P, _P, _P, _sz]),
    "kmi_index_build_dist_host": (C.c_int, [_P, _P, _P, _sz, _u64]),
    "kmi_index_build_dist_dev": (C.c_int, [_P, _P, _P, _sz, _u64]),
    "kmi_index_build_range_dist_host": (C.c_int, [_P, _P, _P, _sz, _u64, _u64, C.c_int, C.POINTER(C.c_int)]),
    "kmi_index_build_fasta_file_dist_host": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_index_build_fasta_range_dist_host": (C.c_int, [_P, _P, _P, _sz, _u64, _u64, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "kmi_fasta_block_summary_dev": (C.c_int, [_P, _P, _sz, C.c_int, _P]),
    "kmi_extract_fasta_block_host": (C.c_int, [_P, _CFG, _P, _sz, _u32, _u32, C.POINTER(Tuples)]),
    "kmi_extract_fasta_range_dist_host": (C.c_int, [_P, _CFG, _P, _P, _sz, _u64, _u64, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(Tuples)]),
    "kmi_index_insert_pairs_dist_host": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_index_update_pairs_dist_host": (C.c_int, [_P, _P, _P, _sz, _u32, C.POINTER(_u64)]),
    "kmi_index_route_pairs_dist_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(Results)]),
    "kmi_extract_range_host": (C.c_int, [_P, _CFG, _P, _sz, _u64, _u64, C.c_int, C.POINTER(C.c_int), C.POINTER(Tuples)]),
    "kmi_fastq_find_records_dev": (C.c_int, [_P, _P, _sz, C.c_int, _P, _u32, _P]),
    "kmi_index_find_hits_dev": (C.c_int, [_P, _P, _sz, C.POINTER(_u64)]),
    "kmi_index_count_dist_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(Results)]),
    "kmi_index_find_dist_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(Results)]),
    "kmi_index_erase_dist_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(_u64)]),
    "kmi_index_size_dist": (C.c_int, [_P, _P, C.POINTER(_u64)]),
    "kmi_index_update_pairs_host": (C.c_int, [_P, _P, _sz, _u32, C.POINTER(_u64)]),
    "kmi_index_update_pairs_dev": (C.c_int, [_P, _P, _sz, _u32, C.POINTER(_u64)]),
    "kmi_index_sk_produce_dev": (C.c_int, [_P, _P, _sz, _u32, _P, _sz, C.POINTER(_P), C.POINTER(_u64), _P, C.POINTER(C.c_int)]),
    "kmi_index_sk_consume_dev": (C.c_int, [_P, _P, _sz, _u32]),
    "kmi_route_owner_dev": (C.c_int, [_P, _CFG, _P, _sz, _u32, _P, _P]),
    "kmi_index_owner_ranks": (C.c_int, [_P, C.POINTER(_u32)]),
    "kmi_index_sk_width": (C.c_int, [_P, C.POINTER(_u32)]),
    "kmi_index_set_saturating": (C.c_int, [_P, C.c_int]),
    "kmi_index_set_owner_ranks": (C.c_int, [_P, _u32]),
    "kmi_dbg_create": (C.c_int, [_P, _CFG, _u32, C.POINTER(_P)]),
    "kmi_dbg_destroy": (C.c_int, [_P]),
    "kmi_dbg_clear": (C.c_int, [_P]),
    "kmi_dbg_local_size": (C.c_int, [_P, C.POINTER(_u64)]),
    "kmi_dbg_parse_dev": (C.c_int, [_P, _CFG, _P, _sz, _P, _sz, C.POINTER(_u64)]),
    "kmi_dbg_build_dev": (C.c_int, [_P, _P, _sz]),
    "kmi_dbg_build_host": (C.c_int, [_P, _P, _sz]),
    "kmi_dbg_insert_dev": (C.c_int, [_P, _P, _sz]),
    "kmi_dbg_insert_host": (C.c_int, [_P, _P, _sz]),
    "kmi_dbg_find_host": (C.c_int, [_P, _P, _sz, C.POINTER(Results)]),
    "kmi_dbg_find_dev": (C.c_int, [_P, _P, _sz, _P, _P, C.POINTER(_u64)]),
    "kmi_dbg_count_host": (C.c_int, [_P, _P, _sz, C.POINTER(Results)]),
    "kmi_dbg_export_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(_u64)]),
    "kmi_dbg_build_dist_host": (C.c_int, [_P, _P, _P, _sz]),
    "kmi_dbg_find_dist_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(Results)]),
    "kmi_dbg_build_range_dist_host": (C.c_int, [_P, _P, _P, _sz, _u64, _u64, C.c_int, C.POINTER(C.c_int)]),
    "kmi_dbg_set_seq_format": (C.c_int, [_P, _u32]),
    "kmi_dbg_erase_host": (C.c_int, [_P, _P, _sz, C.POINTER(_u64)]),
    "kmi_dbg_erase_dist_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(_u64)]),
    "kmi_dbg_count_dist_host": (C.c_int, [_P, _P, _P, _sz, C.POINTER(Results)]),
    "kmi_dbg_size_dist": (C.c_int, [_P, _P, C.POINTER(_u64)]),
    "kmi_profile_enable": (C.c_int, [_P, C.c_int]),
    "kmi_profile_reset": (C.c_int, [_P]),
    "kmi_profile_get": (C.c_int, [_P, C.POINTER(KernelTime), _sz, C.POINTER(_sz)]),
    "kmi_synth_fastq_bytes": (_sz, [_u64, _u32]),
    "kmi_synth_fastq": (C.c_int, [_u64, _u64, _u32, _u64, _u64, _P, _sz, _u32]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class KmiError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("kmerind_hip status %d: %s" % (status, msg))
        self.status = status
