/*
 * kmerind_hip.h -- C ABI of libkmerind_hip.so, the MI355X (gfx950) k-mer index core.
 *
 * This is the drop-in boundary for ONE path of ParBLiSS/kmerind:
 *   FASTA/FASTQ bytes -> k-mer tuples -> (strand transform) -> hash / rank ->
 *   bucket partition -> per-bucket reduce -> count / find / erase.
 *
 * The reference exposes that path as C++ templates, not as an FFI
 * (bliss::index::kmer::Index<MapType,KmerParser>, src/index/kmer_index.hpp:98-394).
 * Each entry point below names the reference function(s) it replaces; the C++
 * facade in include/kmerind/ re-creates the reference's class / alias names on
 * top of these calls.
 *
 * Conventions
 *   - every function returns kmi_status (0 = OK); kmi_last_error(ctx) gives text.
 *     The facade maps KMI_ERR_INVALID -> std::invalid_argument,
 *     KMI_ERR_PARSE -> std::logic_error (the exceptions the reference throws,
 *     kmer_index.hpp:245-254, fastq_loader.hpp:350-363).
 *   - plain pointers and sizes only; no C++ or torch types cross the ABI.
 *   - "_host" entry points take/return host memory exactly like the reference's
 *     std::vector based API; "_dev" entry points take device pointers that are
 *     already resident in HBM (what bench.py times).
 *   - k-mers are the reference's Kmer<K,Alphabet,uint64_t> object bytes:
 *     n_words little-endian 64-bit words, data[0] least significant, newest base
 *     in the low bits, pad bits zero (src/common/kmer.hpp:116-177).
 *   - one context = one GPU = one "rank" (replaces mxx::comm); all calls on a
 *     context are issued from one host thread, like one MPI rank.
 *   - there is no CPU fallback: without a usable HIP device every entry point
 *     fails with KMI_ERR_DEVICE.
 */
#ifndef KMERIND_HIP_H
#define KMERIND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  KMI_OK = 0,
  KMI_ERR_INVALID = 1, /* bad argument / unsupported configuration */
  KMI_ERR_DEVICE = 2,  /* HIP runtime failure, no device */
  KMI_ERR_PARSE = 3,   /* malformed FASTQ/FASTA (reference throws std::logic_error) */
  KMI_ERR_NOMEM = 4,
  KMI_ERR_OVERFLOW = 5, /* id overflow (sequence.hpp:177-183) or capacity exceeded */
  KMI_ERR_PEER = 6      /* a collective ended because ANOTHER rank reported an error (this rank's own state is unchanged or as documented) */
} kmi_status;

enum { KMI_ALPHA_DNA = 0, KMI_ALPHA_DNA5 = 1,                   /* alphabets.hpp:127-185, 212-285 (DNA5 == DNA6) */
       KMI_ALPHA_RNA = 2, KMI_ALPHA_RNA5 = 3,                   /* alphabets.hpp:365-445, 448-530 (RNA5 == RNA6): U for T */
       KMI_ALPHA_DNA16 = 4 };                                   /* alphabets.hpp:648-733: IUPAC, 4 bits, complement = bit reversal */
enum { KMI_STRAND_SINGLE = 0, KMI_STRAND_CANONICAL = 1, KMI_STRAND_BIMOLECULE = 2 }; /* kmer_index.hpp:436-481 */
enum { KMI_HASH_MURMUR = 0, KMI_HASH_FARM = 1,                  /* kmer_hash.hpp:242-311 */
       KMI_HASH_IDENTITY = 2, KMI_HASH_STD = 3 };               /* kmer_hash.hpp:205-230, 154-198 (cpp_std, libstdc++) */
enum { KMI_FMT_FASTQ = 0, KMI_FMT_FASTA = 1 };
enum { KMI_INDEX_COUNT = 0, KMI_INDEX_POSITION = 1, KMI_INDEX_POSQUAL = 2 }; /* kmer_index.hpp:399-411 */
/* the SeqIterType argument of read_file_* / build_* (kmer_index.hpp:239-372): SequencesIterator (every record),
 * NFilterSequencesIterator (records whose sequence holds an 'N' are skipped, filtered_sequence_iterator.hpp:154-165)
 * or NSplitSequencesIterator (sequences are cut at every 'N' / 'n', so no k-mer spans one, :429-440) */
enum { KMI_SEQ_ALL = 0, KMI_SEQ_N_FILTER = 1, KMI_SEQ_N_SPLIT = 2 };
/* DistTrans of SingleStrandHashMapParams ("could be iden, xor, lex_less", kmer_index.hpp:436-450; the benchmark's
 * pDistTrans, BenchmarkKmerIndex.cpp:150-161): what KeyToRank hashes. MODEL = the strand model's own choice (identity for
 * single strand and canonical, lex_less for bimolecule); LEX / XOR (kmer_transform.hpp:90-116,60-88) need
 * KMI_STRAND_SINGLE and make both strands of a k-mer land on one rank while they stay separate keys. */
enum { KMI_DIST_MODEL = 0, KMI_DIST_LEX = 1, KMI_DIST_XOR = 2 };

/* The compile-time parameters of the reference's Index<Map,Parser> as a runtime struct. */
typedef struct {
  uint32_t k;          /* Kmer<K,...>::size */
  uint32_t alphabet;   /* KMI_ALPHA_* */
  uint32_t strand;     /* KMI_STRAND_*: Single / Canonical / Bimolecule HashMapParams */
  uint32_t dist_hash;  /* KMI_HASH_*: DistHash (rank assignment), Prefix=true variant */
  uint32_t store_hash; /* KMI_HASH_*: StoreHash; accepted for API parity, placement on the
                          GPU is internal and does not change results */
  uint32_t index_kind; /* KMI_INDEX_* */
  uint32_t seq_format; /* KMI_FMT_* */
  uint32_t farm_ndebug;/* 0: farmhash as the reference's default RelWithDebInfo build computes it
                          (DebugTweak active, CMakeLists.txt:26,200-204); 1: -DNDEBUG behaviour */
  uint32_t seq_filter; /* KMI_SEQ_*. With a filter, n_seqs counts what the reference's read_block counts: records
                          that pass (N_FILTER) or non-empty pieces (N_SPLIT; FASTA: records). FASTA with a filter
                          needs k >= 2. Quality values (KMI_INDEX_POSQUAL) need KMI_SEQ_ALL. */
  uint32_t dist_trans; /* KMI_DIST_* */
} kmi_config;

typedef struct kmi_ctx kmi_ctx;     /* replaces mxx::comm + per-rank state */
typedef struct kmi_index kmi_index; /* replaces MapType (dsc::counting_*_map etc.) */

/* ---- context ------------------------------------------------------------ */
/* Index(const mxx::comm&) (kmer_index.hpp:115): device = HIP ordinal, rank/nranks =
 * comm.rank()/comm.size(). stream = hipStream_t (NULL = default stream). */
kmi_status kmi_ctx_create(int device, int rank, int nranks, void *stream, kmi_ctx **out);
kmi_status kmi_ctx_destroy(kmi_ctx *ctx);
/* forget what the context learned from its builds so far (the pass structure and the duplication the per-bucket reduce starts
 * from): the next build runs as the first one of a context would, with the workspace blocks still in place */
kmi_status kmi_ctx_reset_hints(kmi_ctx *ctx);
/* counters the library keeps for its tests and for diagnosis (no reference counterpart). which = 0: times a build over ranks had to
 * enlarge its receive pool (kmi_index_build_dist_dev); 1: microseconds this context has spent inside hipMalloc / hipFree; 2: bytes
 * and 3: calls that reached hipMalloc; 4: blocks taken from the process-wide cache instead */
kmi_status kmi_ctx_debug_counter(const kmi_ctx *ctx, uint32_t which, uint64_t *value);
/* A destroyed context leaves its large device blocks (workspace, spare index arrays: >= 1 MB each, 96 GB / 64 blocks at most) in a
 * process-wide cache per device, where the next context of that device finds them: its first build then does not wait for
 * hipMalloc ("first build"). This frees them (device < 0: of every device). */
kmi_status kmi_release_cached_memory(int device, uint64_t *bytes_released /* may be NULL */);
const char *kmi_last_error(const kmi_ctx *ctx);
/* derived Kmer shape (padding.hpp:67-90): words per k-mer, hashed byte length */
kmi_status kmi_kmer_shape(const kmi_config *cfg, uint32_t *n_words, uint32_t *n_bits, uint32_t *n_bytes);
void kmi_free_host(void *p);
kmi_status kmi_device_alloc(kmi_ctx *ctx, size_t bytes, void **dptr);
kmi_status kmi_device_free(kmi_ctx *ctx, void *dptr);
kmi_status kmi_copy_to_device(kmi_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
kmi_status kmi_copy_on_device(kmi_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes); /* on the context's stream, completed on return */
kmi_status kmi_copy_to_host(kmi_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
kmi_status kmi_synchronize(kmi_ctx *ctx);

/* ---- FASTA input split over ranks ------------------------------------------------------------------
 * What FASTAParser::init_parser learns about its block from the neighbouring ranks (fasta_loader.hpp:232-456,
 * 485-604) and the valid range / overlap rule of the k-mer parsers (kmer_parser.hpp:112-157, overlap = k - 1,
 * kmer_file_helper.hpp:563), as plain numbers the caller supplies for the NEXT FASTA extract / build calls on this
 * context: the buffer is bytes [file_offset, file_offset + n_bytes) of the file, only k-mers whose first base lies
 * in its first `valid_bytes` bytes are produced (the rest of the buffer is the overlap the last windows read), and
 * the line-kind machine starts in `start_state` instead of "file start". NULL restores whole-file behaviour. */
enum { KMI_FA_OUTSIDE = 0, KMI_FA_HEADER = 1, KMI_FA_SEQUENCE = 2 };
typedef struct {
  uint64_t valid_bytes;     /* k-mers start in [0, valid_bytes) of the buffer */
  uint32_t start_state;     /* KMI_FA_*: kind of the line that byte 0 of the buffer sits on (OUTSIDE = before any header) */
  uint32_t at_line_start;   /* 1: byte 0 is the first byte of a line (file start or the byte before it is '\n') */
  uint64_t records_before;  /* records (header group -> sequence group transitions) that start before the buffer */
  uint32_t index_shift;     /* 1 when the FILE begins with non-header lines (init_parser's k/2 rule), else 0 */
  uint32_t reserved;
} kmi_fasta_partition;
kmi_status kmi_ctx_set_fasta_partition(kmi_ctx *ctx, const kmi_fasta_partition *part);
/* the same bookkeeping computed ON THE DEVICE for every block of an n_parts-way split of a FASTA buffer that sits in HBM (the
 * partition negotiation of file.hpp:1436-1610 + fasta_loader.hpp:202-470): block r = bytes [begin_end_host[2 r], begin_end_host
 * [2 r + 1]) of the buffer -- its nominal range [floor(n r / p), floor(n (r + 1) / p)) plus the overlap that holds k - 1 further
 * sequence characters -- and parts_host[r] what kmi_ctx_set_fasta_partition wants for it. Machine state and record count at a byte
 * come from prefix sums over the scan's tile summaries. */
kmi_status kmi_fasta_partition_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, uint32_t n_parts, uint32_t k,
                                   uint64_t *begin_end_host /* 2 * n_parts */, kmi_fasta_partition *parts_host /* n_parts */);

/* ---- L2: k-mer value ops on arrays (parity surface for kmer.hpp / kmer_transform.hpp) */
/* Kmer::reverse_complement (kmer.hpp:1118-1127) on n k-mers */
kmi_status kmi_revcomp_host(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *in, size_t n, uint64_t *out);
/* transform::lex_less (kmer_transform.hpp:108-116) */
kmi_status kmi_canonical_host(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *in, size_t n, uint64_t *out);
/* hash::murmur / farm / identity / cpp_std <Kmer,Prefix> (kmer_hash.hpp:154-311); identity and cpp_std with
 * their default-constructed prefix width (24 / 32 bits); inside KeyToRank they get ceilLog2(nranks) like the reference */
kmi_status kmi_hash_host(kmi_ctx *ctx, const kmi_config *cfg, uint32_t which, int prefix,
                         const uint64_t *in, size_t n, uint64_t *out);
/* KeyToRank (distributed_unordered_map.hpp:148-170): DistHash(DistTrans(k)) % nranks */
kmi_status kmi_key_to_rank_host(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *in, size_t n,
                                uint32_t nranks, uint32_t *ranks);

/* ---- L3: file bytes -> tuples.  KmerFileHelper::read_file_* / parse_file_data_old /
 * read_block_old (kmer_file_helper.hpp:110-186,441-482,588-633) + the tuple parsers
 * (kmer_parser.hpp:85-294,303-569,577-900,909-1083).
 * `bytes` is a record-aligned partition whose first byte sits at `file_offset`;
 * tuples come back in file order, as parsed (no strand transform), like the reference. */
typedef struct {
  uint64_t n_tuples;
  uint64_t n_seqs;
  uint64_t *kmers;  /* n_tuples * n_words */
  uint64_t *ids;    /* Short/LongSequenceKmerId (sequence.hpp:127-296) or NULL */
  float *quals;     /* k-mer quality (quality_score_iterator.hpp:166-173) or NULL */
} kmi_tuples;

kmi_status kmi_extract_host(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes, size_t n_bytes,
                            uint64_t file_offset, kmi_tuples *out /* buffers malloc'd; kmi_tuples_free */);
void kmi_tuples_free(kmi_tuples *t);
/* read_file_* of one rank of several (kmer_file_helper.hpp:550-579 over partitioned_file, file.hpp:1216-1430), FASTQ: the rank read
 * file bytes [buffer_offset, buffer_offset + n_bytes) -- its nominal byte range (nominal_bytes) plus look-ahead -- and parses the
 * records from the first record start at or after its first byte to the first one at or after the nominal end. *need_more = 1
 * (nothing parsed): the partition's end is not decidable inside the buffer and the buffer does not reach the file's end. */
kmi_status kmi_extract_range_host(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                  uint64_t nominal_bytes, int reaches_eof, int *need_more, kmi_tuples *out);

/* the same for FASTA (fasta_loader.hpp:202-470 over file.hpp:1436-1610): every rank holds the WHOLE file; the tuples of block
 * `rank` of an equal nranks-way split come back, the block bookkeeping (valid range, machine state, records before) being computed
 * on the device from the whole buffer (kmi_fasta_partition_dev). The union over the ranks is the file's tuples, each once. */
kmi_status kmi_extract_fasta_block_host(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes, size_t n_bytes, uint32_t rank,
                                        uint32_t nranks, kmi_tuples *out);

/* device form: out_kmers_dev capacity in tuples (use kmi_extract_count_dev first, or pass an upper bound n_bytes). bytes_dev
 * may point anywhere (a record-aligned batch inside a larger buffer): an input that is not 16-byte aligned is copied once,
 * device to device, to an aligned workspace buffer. */
kmi_status kmi_extract_count_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                                 uint64_t *n_tuples, uint64_t *n_seqs);
kmi_status kmi_extract_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                           uint64_t file_offset, uint64_t *out_kmers_dev, uint64_t *out_ids_dev,
                           size_t out_capacity, uint64_t *n_tuples, uint64_t *n_seqs);

/* the tuples of the position parsers as records, device to device: n_words key words followed by the value words (id, or
 * id and the quality's float bits in the low half of a second word) -- the object bytes of std::pair<Kmer, ShortSequenceKmerId>
 * / std::pair<Kmer, std::pair<ShortSequenceKmerId, float>> (kmer_parser.hpp:303-569, 577-900): what kmi_route_tuples_dev
 * and kmi_index_insert_tuples_dev take. out_capacity in records (kmi_extract_count_dev gives the number). */
kmi_status kmi_extract_records_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset,
                                   uint64_t *out_records_dev, size_t out_capacity, uint64_t *n_tuples, uint64_t *n_seqs);

/* record-aligned partition of a FASTQ buffer that already sits in HBM: cuts[r] = first record start at or after byte
 * floor(n_bytes * r / n_parts) by the four-line rule of FASTQParser::find_first_record (fastq_loader.hpp:269-364, as
 * partitioned_file applies it, file.hpp:1216-1430); cuts[n_parts] = n_bytes; the ranges [cuts[r], cuts[r + 1]) tile the
 * buffer. What every rank hands to kmi_index_build_* / kmi_extract_* when one buffer is split between ranks or batches. */
kmi_status kmi_fastq_partition_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, uint32_t n_parts, uint64_t *cuts_host /* n_parts + 1 */);
/* the same rule for explicit positions: starts_host[i] = first record start at or after positions_host[i] in the buffer (n_bytes
 * when the buffer holds none behind it). What a rank that read only ITS byte range of a file (plus some look-ahead) uses to find
 * where its partition begins and ends (file.hpp:1342-1422): both neighbours apply the rule at the same file position. */
kmi_status kmi_fastq_find_records_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, int buffer_starts_file /* byte 0 = the file's first byte */,
                                      const uint64_t *positions_host, uint32_t n_pos, uint64_t *starts_host);

/* ---- L4: the exchange step of imxx::distribute (incremental_mxx.hpp:1039-1109):
 * assign_to_buckets + bucket_to_permutation + permute on the device. Output is the
 * send buffer grouped by destination rank, counts[nranks] on the host. The order INSIDE a rank's message is unspecified (the
 * reference's permutation is stable, incremental_mxx.hpp:324-364, but nothing it feeds -- hash map inserts, query dedup --
 * observes that order). The all-to-all itself is done by the caller (RCCL through
 * torch.distributed in kmerind_amd.dist, or ncclSend/ncclRecv from C++). */
kmi_status kmi_route_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *keys_dev, size_t n,
                         uint32_t nranks, uint64_t *out_keys_dev, uint64_t *send_counts_host);

/* same for (k-mer, value) records of the position indexes: value_words 64-bit words follow each key */
kmi_status kmi_route_tuples_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *records_dev, size_t n,
                                uint32_t nranks, uint32_t value_words, uint64_t *out_records_dev, uint64_t *send_counts_host);

/* read_file_* followed by the bucketing half of imxx::distribute, fused (FASTQ): the k-mers of this rank's reads,
 * transformed (InputTrans) and grouped by destination rank = DistHash(DistTrans(k)) % nranks, written straight from
 * the packed input; the tuple array in file order never exists in HBM. Replaces kmi_extract_dev + kmi_route_dev on
 * the multi-GPU build path (kmer_file_helper.hpp:588-633 + incremental_mxx.hpp:1039-1086). */
kmi_status kmi_extract_route_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint32_t nranks,
                                 uint64_t *out_keys_dev, size_t out_capacity, uint64_t *n_tuples, uint64_t *n_seqs,
                                 uint64_t *send_counts_host);

/* the same for the (k-mer, value) tuples of the position indexes (index_kind POSITION / POSQUAL): kmi_extract_records_dev +
 * kmi_route_tuples_dev in one call, records of n_words + value_words words grouped by destination rank in out_records_dev;
 * the tuples in file order never leave the workspace (read_file_* + imxx::distribute of Index::build_* with a multimap,
 * kmer_index.hpp:148-225, distributed_unordered_map.hpp:1466-1515) */
kmi_status kmi_extract_route_records_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset,
                                         uint32_t nranks, uint64_t *out_records_dev, size_t out_capacity, uint64_t *n_tuples,
                                         uint64_t *n_seqs, uint64_t *send_counts_host);

/* ---- L4/L5: the map behind Index<MapType,Parser> ---------------------------- */
kmi_status kmi_index_create(kmi_ctx *ctx, const kmi_config *cfg, kmi_index **out);
kmi_status kmi_index_destroy(kmi_index *idx);
/* Index::insert(std::vector<Kmer>&) for counting maps on ONE rank: InputTransform +
 * local reduce (distributed_unordered_map.hpp:1697-1745,1826-1884). Keys must already
 * belong to this rank when nranks > 1 (i.e. after the exchange). */
kmi_status kmi_index_insert_host(kmi_index *idx, const uint64_t *kmers, size_t n);
kmi_status kmi_index_insert_dev(kmi_index *idx, const uint64_t *kmers_dev, size_t n);
/* the local_insert half alone (distributed_unordered_map.hpp:1734-1741): keys that already went through the
 * InputTransform -- what kmi_route_dev / kmi_extract_route_dev produce and the exchange delivers -- are reduced as they are */
kmi_status kmi_index_insert_transformed_dev(kmi_index *idx, const uint64_t *kmers_dev, size_t n);
/* Index::insert(std::vector<std::pair<Kmer, T>>&) of the counting / reduction maps (kmer_index.hpp:200-225 ->
 * reduction_unordered_map::local_insert, distributed_unordered_map.hpp:1603-1618: `at() = r(at(), v)` with r = std::plus):
 * every pair's value is ADDED to the key's count. records = n objects of (n_words key words, one value word whose low
 * 32 bits are the count) -- the object bytes of std::pair<Kmer, uint32_t>; the keys go through the InputTransform. */
kmi_status kmi_index_insert_pairs_host(kmi_index *idx, const uint64_t *records, size_t n);
kmi_status kmi_index_insert_pairs_dev(kmi_index *idx, const uint64_t *records_dev, size_t n);
/* Index::build_mmap/build_posix for nranks == 1: read_file + insert fused on the
 * device (kmer_index.hpp:239-372). */
kmi_status kmi_index_build_host(kmi_index *idx, const uint8_t *bytes, size_t n_bytes, uint64_t file_offset);
kmi_status kmi_index_build_dev(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset);
/* MapType::clear() (distributed_map_base.hpp:267-272) */
kmi_status kmi_index_clear(kmi_index *idx);
/* The SeqParser template argument of Index::build_posix / build_mmap / build_mpiio<SeqParser, SeqIterType>
 * (kmer_index.hpp:239-372): which record grammar the next kmi_index_build_* call parses (KMI_FMT_*). */
kmi_status kmi_index_set_seq_format(kmi_index *idx, uint32_t seq_format);
/* the SeqIterType template argument of Index::build_* (kmer_index.hpp:239-372): KMI_SEQ_* for the following builds */
kmi_status kmi_index_set_seq_filter(kmi_index *idx, uint32_t seq_filter);
/* MapType::local_size() / size() on one rank (distributed_map_base.hpp:227-245) */
kmi_status kmi_index_local_size(kmi_index *idx, uint64_t *n);
/* MapType::to_vector() (distributed_map_base.hpp:202-217): keys n*n_words, counts n; order unspecified */
kmi_status kmi_index_export_host(kmi_index *idx, uint64_t *keys, uint32_t *counts, size_t capacity, uint64_t *n);

typedef struct {
  uint64_t n;
  uint64_t *keys;   /* n * n_words: the transformed (e.g. canonical) query keys */
  uint64_t *values; /* count(): 0/1 presence (db.count(k)); find(): stored count */
} kmi_results;
void kmi_results_free(kmi_results *r);
/* Index::count (kmer_index.hpp:142-145 -> distributed_unordered_map.hpp:880-983):
 * one entry per DISTINCT transformed query key, value = number of map entries with
 * that key (0 or 1 for a counting map). */
kmi_status kmi_index_count_host(kmi_index *idx, const uint64_t *queries, size_t nq, kmi_results *out);
/* Index::find (kmer_index.hpp:132-135 -> :564-687): (key, stored value) of the
 * distinct transformed query keys that are present. */
kmi_status kmi_index_find_host(kmi_index *idx, const uint64_t *queries, size_t nq, kmi_results *out);
/* Index::erase (kmer_index.hpp:147-149 -> :719-779) */
kmi_status kmi_index_erase_host(kmi_index *idx, const uint64_t *queries, size_t nq, uint64_t *n_erased);
/* device-resident query forms used by bench.py (results stay on the device):
 * out_keys_dev/out_values_dev capacity nq; *n_out distinct results */
kmi_status kmi_index_count_dev(kmi_index *idx, const uint64_t *queries_dev, size_t nq,
                               uint64_t *out_keys_dev, uint64_t *out_values_dev, uint64_t *n_out);
kmi_status kmi_index_find_dev(kmi_index *idx, const uint64_t *queries_dev, size_t nq,
                              uint64_t *out_keys_dev, uint64_t *out_values_dev, uint64_t *n_out);
/* the number of entries kmi_index_find_dev would return for these queries (a multimap returns every entry of a queried key:
 * size the result buffers with this instead of the index's entry count) */
kmi_status kmi_index_find_hits_dev(kmi_index *idx, const uint64_t *queries_dev, size_t nq, uint64_t *n_hits);

/* ---- multimap maps: PositionIndex / PositionQualityIndex (kmer_index.hpp:402-406) over
 * ::dsc::unordered_multimap (distributed_unordered_map.hpp:1466-1515). An index created with
 * index_kind POSITION stores every (k-mer, value) tuple; value = 1 u64 (Short/LongSequenceKmerId),
 * POSQUAL = 2 u64 (id, quality float in the low 32 bits of the second word).
 * kmi_index_build_* on such an index parses with KmerPositionTupleParser semantics.
 * count() returns the multiplicity, find() every (key, value) whose key is queried, erase() removes
 * all of them; kmi_results.values then holds n * value_words words. */
kmi_status kmi_index_insert_tuples_host(kmi_index *idx, const uint64_t *kmers, const uint64_t *values, size_t n);
/* records_dev: n records of (n_words key words, value words), i.e. std::pair<Kmer, value> objects */
kmi_status kmi_index_insert_tuples_dev(kmi_index *idx, const uint64_t *records_dev, size_t n);
kmi_status kmi_index_export_tuples_host(kmi_index *idx, uint64_t *keys, uint64_t *values, size_t capacity, uint64_t *n);

/* ---- combine-first distributed insert of the count index (N > 1 ranks) ------------------------------------
 * The reference's counting maps send every k-mer occurrence through imxx::distribute and reduce at the receiver
 * (distributed_unordered_map.hpp:1715-1745 / distributed_densehash_map.hpp:2584-2610; the local_reduction before
 * the exchange is there but commented out). Integer counts add up associatively, so a rank may reduce its own
 * reads first: it builds a local count index of its partition (kmi_index_build_*), splits the entries by
 * KeyToRank, exchanges (k-mer, count) pairs, and every rank merges what it receives. The resulting distributed
 * index is the reference's (key on rank DistHash(DistTrans(key)) % p, value = occurrences over all ranks); the
 * exchanged volume shrinks by the local coverage.
 *
 * kmi_index_num_buckets(): the fixed number B of placement buckets every index uses (entries of a bucket are
 * contiguous; the placement hash is the same on all ranks).
 * kmi_index_split_by_rank_dev: the entries of `idx` grouped by destination rank (message r starts at the sum of
 * send_counts_host[0..r)), and inside a message ordered by placement bucket; bucket_counts_dev[r * B + b] = entries
 * of message r in bucket b. out buffers hold `capacity` >= kmi_index_local_size entries. `idx` is not changed.
 * kmi_index_merge_parts_dev: kmers_dev / counts_dev = nparts such messages back to back (part s holds
 * sum_b bucket_counts_dev[s * B + b] pairs); their counts are added into `idx`. */
uint32_t kmi_index_num_buckets(void);
kmi_status kmi_index_split_by_rank_dev(kmi_index *idx, uint32_t nranks, uint64_t *out_kmers_dev, uint32_t *out_counts_dev,
                                       size_t capacity, uint32_t *bucket_counts_dev, uint64_t *send_counts_host);
kmi_status kmi_index_merge_parts_dev(kmi_index *idx, uint32_t nparts, const uint64_t *kmers_dev, const uint32_t *counts_dev,
                                     const uint32_t *bucket_counts_dev);

/* ---- more than one rank: the exchange over RCCL (xGMI inside one node) -------------------------------------------
 * One process per GPU (kmi_ctx_create(device, rank, nranks)). A communicator replaces the mxx::comm the reference's maps
 * hold: rank 0 makes an id (kmi_comm_unique_id = ncclGetUniqueId, 128 bytes) and the application hands it to the other
 * ranks (MPI_Bcast in the reference's world), every rank calls kmi_comm_create (ncclCommInitRank; collective).
 * kmi_comm_all_to_all_counts / _v are mxx::all2all / mxx::all2allv of imxx::distribute (incremental_mxx.hpp:1087, 1098):
 * counts on the host, payload in device buffers grouped by destination, received as the concatenation by source rank
 * ascending; grouped ncclSend / ncclRecv on the context's stream, 64-bit counts, peer messages in pieces below 1 GiB, the
 * first exchange of a communicator verified by per-message checksums. nranks == 1 works (self exchange). */
typedef struct kmi_comm kmi_comm;
enum { KMI_COMM_ID_BYTES = 128 };
kmi_status kmi_comm_unique_id(void *id_out /* KMI_COMM_ID_BYTES */);
kmi_status kmi_comm_create(kmi_ctx *ctx, const void *id /* NULL allowed when nranks == 1 */, kmi_comm **out);
/* A communicator over a messenger the APPLICATION brings instead of RCCL: the two collectives imxx::distribute needs
 * (incremental_mxx.hpp:1087 all2all, :1098 all2allv) and the all-reduce behind size() (distributed_map_base.hpp:227-245), as
 * callbacks over HOST buffers -- an MPI communicator (the reference's own mxx::comm: INTEGRATION.md section 4), gloo, sockets.
 * The library stages device buffers through pinned host memory around each call, so everything built on a communicator --
 * kmi_index_*_dist_*, kmi_dbg_*_dist_* -- runs unchanged where RCCL is not an option (ranks sharing one GPU, a cluster without
 * xGMI). Both callbacks are collective and blocking; they return 0 on success. */
typedef struct kmi_transport {
  void *user;
  /* rank r receives send_bytes[r] bytes from this rank (the messages lie back to back in `send`, rank 0's first) and this
   * rank receives recv_bytes[r] bytes from rank r into `recv`, back to back by source rank */
  int (*all_to_all_v)(void *user, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes);
  /* values[0..n) are replaced by their sum (op 0) or maximum (op 1) over the ranks */
  int (*allreduce_u64)(void *user, uint64_t *values, size_t n, int op);
} kmi_transport;
kmi_status kmi_comm_create_transport(kmi_ctx *ctx, const kmi_transport *transport /* copied */, kmi_comm **out);
kmi_status kmi_comm_destroy(kmi_comm *comm);
kmi_status kmi_comm_all_to_all_counts(kmi_comm *comm, const uint64_t *send_counts_host, uint64_t *recv_counts_host);
kmi_status kmi_comm_all_to_all_v(kmi_comm *comm, const void *send_dev, const uint64_t *send_counts_host, void *recv_dev,
                                 const uint64_t *recv_counts_host, size_t elem_bytes /* multiple of 8 */);
kmi_status kmi_comm_allreduce_sum_u64(kmi_comm *comm, uint64_t *value_host);

/* The collectives of Index<MapType, Parser> with comm.size() > 1, every rank calling with its own (possibly empty) share:
 * insert  (distributed_unordered_map.hpp:1697-1745, :1466-1515): InputTransform, KeyToRank grouping on the device
 *         (kmi_route_dev), all2all(counts) + all2allv(keys), local insert of what arrives;
 * build   (kmer_index.hpp:239-372): this rank's record-aligned partition of the file -> parse -> route -> insert (FASTQ count
 *         index: kmi_extract_route_dev, the tuple array never exists; position indexes: records with their values);
 * count / find / erase (:880-983, :564-687, :719-779): the query keys travel to their owners, every owner answers per source
 *         rank, one return exchange; results are this rank's own queries' answers, as the reference returns them;
 * size    (distributed_map_base.hpp:227-245): allreduce of the local sizes.
 * Device buffers throughout; only the caller's vectors cross PCIe. */
kmi_status kmi_index_insert_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *kmers, size_t n);
kmi_status kmi_index_insert_tuples_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *kmers, const uint64_t *values, size_t n);
kmi_status kmi_index_build_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t file_offset);
/* the same with this rank's share already in HBM. The count index of one-word 2-bit k-mers (k >= 17, FASTQ, 2 / 4 / 8 ranks)
 * travels as super-k-mer records in record-aligned chunks (KMI_DIST_CHUNKS, default 4): the records of chunk c are exchanged on
 * the communicator's own stream while the front end of chunk c + 1 runs, the counts of a chunk carry every sender's largest
 * message (so all ranks cut the transfer into the same pieces without another collective) and an exchange whose largest message
 * exceeds every one that carried checksums before is verified on arrival. */
kmi_status kmi_index_build_dist_dev(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset);
/* Index::build_posix / build_mmap / build_mpiio(filename, comm) with comm.size() > 1 (kmer_index.hpp:239-372 over
 * partitioned_file, file.hpp:1216-1430), FASTQ: `bytes` = what this rank read of the file, file bytes [buffer_offset,
 * buffer_offset + n_bytes) -- its nominal range [buffer_offset, buffer_offset + nominal_bytes) plus look-ahead. The partition
 * begins at the first record start at or after the buffer's first byte (the file's first byte for buffer_offset 0) and ends at the
 * first record start at or after the nominal end (four-line rule, on the device); reaches_eof says that the buffer ends with the
 * file. *need_more = 1 and nothing done when the end cannot be decided inside the buffer: read further and call again (no
 * collective has been entered). Then the collective build of that partition. Works for comm.size() == 1 too. */
kmi_status kmi_index_build_range_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                           uint64_t nominal_bytes, int reaches_eof, int *need_more);
/* the same for a FASTA file: every rank passes the WHOLE file; it keeps block comm.rank() of an equal split (bookkeeping on the
 * device, kmi_fasta_partition_dev) and enters the collective build with it. Works for comm.size() == 1 too. */
kmi_status kmi_index_build_fasta_file_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes);
/* ... and with every rank holding only ITS byte range of the file (file.hpp:1436-1610 hands each rank 1/p of the file): bytes =
 * file bytes [buffer_offset, buffer_offset + n_bytes), of which the first nominal_bytes are the rank's block of the equal split
 * (block r = [n r / p, n (r + 1) / p)) and the rest look-ahead; prev_byte = the file byte before the buffer, -1 at the file
 * start. What a block cannot know from its own bytes -- the kind of line its first byte sits on, the records that start before it,
 * whether the file opens with a header (fasta_loader.hpp:232-456) -- comes from the ranks' block summaries
 * (kmi_fasta_block_summary_dev), gathered once over the communicator and composed left to right. *need_more = 1 (returned BEFORE
 * anything collective): the k - 1 sequence characters the last windows of the block reach into do not end inside the look-ahead;
 * read further and call again. The union over the ranks is the whole file's tuples, each once, with the ids of a whole-file parse. */
kmi_status kmi_index_build_fasta_range_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                                 uint64_t nominal_bytes, int reaches_eof, int prev_byte, int *need_more);
/* read_file_* of a FASTA file on one rank of several by BYTE RANGE (kmi_extract_fasta_block_host with only the block in memory): the
 * rank brings its block of an equal split of the file plus look-ahead (arguments as kmi_index_build_fasta_range_dist_host); the other
 * blocks' summaries come over comm in one small gather (collective). */
kmi_status kmi_extract_fasta_range_dist_host(kmi_ctx *ctx, const kmi_config *cfg, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes,
                                             uint64_t buffer_offset, uint64_t nominal_bytes, int reaches_eof, int prev_byte, int *need_more,
                                             kmi_tuples *out);
/* the line-kind machine over bytes [0, n_bytes) of a FASTA buffer in HBM as a transfer function: out6 = for the incoming states
 * KMI_FA_OUTSIDE, HEADER, SEQUENCE in turn {state behind the bytes, records that start inside}. first_is_line_start: byte 0 opens a line */
kmi_status kmi_fasta_block_summary_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, int first_is_line_start, uint64_t *out6);
/* weighted insert and update() of the counting maps with comm.size() > 1: the pairs travel to the ranks that own their keys
 * (records: n x (n_words key words, one value word); *n_updated = pairs applied on THIS rank) */
kmi_status kmi_index_insert_pairs_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *records, size_t n);
kmi_status kmi_index_update_pairs_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *records, size_t n, uint32_t op, uint64_t *n_updated);
/* the routing half on its own (imxx::distribute of the pairs, incremental_mxx.hpp:1039-1109): out = the pairs whose keys THIS rank owns,
 * keys as the map stores them, grouped by source rank in rank order; the order inside one source's group is unspecified (out->keys:
 * n_words per pair, out->values: one word; kmi_results_free). update() with a host functor over ranks (distributed_densehash_map.hpp:1975-2030) applies it to these. */
kmi_status kmi_index_route_pairs_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *records, size_t n, kmi_results *out);
kmi_status kmi_index_count_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out);
kmi_status kmi_index_find_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out);
kmi_status kmi_index_erase_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *queries, size_t nq, uint64_t *n_erased_local);
kmi_status kmi_index_size_dist(kmi_index *idx, kmi_comm *comm, uint64_t *n);

/* ---- update() of the counting maps with a device-side updater ------------------------
 * distributed_densehash_map.hpp:1975-2003 -> densehash_map.hpp:663-714: for every input pair whose transformed key is
 * stored, op(stored value, pair value); pairs of absent keys are skipped; returns the number of calls. The reference
 * takes any functor (the facade runs those on the host); the arithmetic updaters have a device form: records as for
 * kmi_index_insert_pairs_* (key words + one word whose low 32 bits are the value). Pairs with the same key are applied
 * in input order there: ADD / MAX / MIN do not depend on it, ASSIGN keeps the value of the LAST such pair, as there.
 * One rank's entries only (the caller routes the pairs with kmi_route_tuples_dev first when size() > 1). */
enum { KMI_UPDATE_ADD = 0, KMI_UPDATE_MAX = 1, KMI_UPDATE_MIN = 2, KMI_UPDATE_ASSIGN = 3 };
kmi_status kmi_index_update_pairs_host(kmi_index *idx, const uint64_t *records, size_t n, uint32_t op, uint64_t *n_updated);
kmi_status kmi_index_update_pairs_dev(kmi_index *idx, const uint64_t *records_dev, size_t n, uint32_t op, uint64_t *n_updated);

/* ---- a count index over 2, 4 or 8 ranks through exchanged super-k-mers ---------------
 * The reference's distributed insert sends every k-mer to KeyToRank(k-mer) (8 bytes per k-mer over the wire,
 * distributed_unordered_map.hpp:1697-1745). For FASTQ input and one-word DNA k-mers (17 <= k <= 32) the fused build cuts
 * the reads into super-k-mers first (runs of k-mers that share a minimizer: 16 bytes for about nine k-mers), so the
 * exchange can move those instead: the owner of a k-mer is then the rank that owns its MINIMIZER's bucket (the top
 * log2(nranks) bits of the 18 bucket bits), not hash(k-mer) % nranks. Which rank holds a k-mer is not observable through
 * Index (count / find / erase / size are collectives); the union of the ranks' maps is the reference's map, bit for bit.
 *   produce: this rank's FASTQ share -> records grouped by owner rank (2 words per record), send_counts_host[nranks] in
 *            records. They are written to out_records_dev when that buffer (out_capacity records; may be NULL) holds them
 *            all, else to library workspace valid until the next call on the context; *records_dev says where they are.
 *            They are written on the context's stream and the call does not wait for that: read them on that stream, on a
 *            stream ordered behind it (a collective issued from it), or through kmi_copy_on_device. *produced = 0: this
 *            path does not apply (shape, rank count) or an input exceeded a capacity of the fused front end -- EVERY
 *            rank must then take the k-mer route for this input (agree on min(*produced) over ranks first);
 *   (caller: all-to-all of the records, 16-byte elements)
 *   consume: the records that arrived (flat array, any order of sources) -> this rank's part of the index;
 *   kmi_route_owner_dev: query keys (transformed as the index's strand model says) grouped by owner rank, the
 *            counterpart of kmi_route_dev for an index built this way; kmi_index_owner_ranks tells how an index was built
 *            (1 = by kmi_index_build / insert: route with kmi_route_dev); kmi_index_set_owner_ranks declares it for an
 *            index that is filled with k-mers routed by kmi_route_owner_dev only (an input none of whose parts could
 *            be produced as records). */
kmi_status kmi_index_sk_produce_dev(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes, uint32_t nranks,
                                    uint64_t *out_records_dev, size_t out_capacity, const uint64_t **records_dev,
                                    uint64_t *n_records, uint64_t *send_counts_host, int *produced);
kmi_status kmi_index_sk_consume_dev(kmi_index *idx, const uint64_t *records_dev, size_t n_records, uint32_t nranks);
kmi_status kmi_route_owner_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *keys_dev, size_t n, uint32_t nranks,
                               uint64_t *out_keys_dev, uint64_t *send_counts_host);
kmi_status kmi_index_owner_ranks(kmi_index *idx, uint32_t *nranks);
/* *w = the minimizer window W the super-k-mer paths take for this index's FASTQ builds, 0 where they do not apply (multi-word
 * k-mers, 3- / 4-bit alphabets, k < 17, a multimap, a context created with KMI_FUSED_PATH=kmer): callers that choose between
 * the record exchange and another route decide from this, identically on every rank */
kmi_status kmi_index_sk_width(kmi_index *idx, uint32_t *w);
/* the reducer of a counting map: 0 (default) = std::plus<uint32_t>, counts wrap (distributed_unordered_map.hpp:1603-1618);
 * 1 = sat_plus<uint32_t> of saturating_counting_densehash_map (distributed_densehash_map.hpp:2903-2912): a count that would pass
 * 2^32 - 1 stays there. Applies wherever counts are ADDED -- weighted pairs, merging into an index that holds entries, parts
 * received from other ranks. (A single call is assumed to hold fewer than 2^32 occurrences of one key.) */
kmi_status kmi_index_set_saturating(kmi_index *idx, int on);
kmi_status kmi_index_set_owner_ranks(kmi_index *idx, uint32_t nranks);

/* ---- de Bruijn graph nodes ------------------------------------------------------
 * The reference's in-tree consumer of Index: de_bruijn_engine<NodeMap> = Index<NodeMap, de_bruijn_parser>
 * (test/test/debruijn/de_bruijn_construct_engine.hpp:241-245) with NodeMap = de_bruijn_nodes_distributed<Kmer,
 * node::edge_counts<DNA16, int32_t> | node::edge_exists<DNA16>, BimoleculeHashMapParams>
 * (de_bruijn_nodes_distributed.hpp:57-265, de_bruijn_node_trait.hpp:139-336), as
 * test/test/test_de_bruijn_graph_construction.cpp:124-137,195-206 instantiates it.
 * A node = (k-mer, counts[9]): counts[0..3] = out edges A C G T (the base right of the k-mer in a read),
 * counts[4..7] = in edges A C G T (the base left of it), counts[8] = occurrences of the k-mer. Edge bytes are DNA16
 * presence bits (in << 4 | out), so an 'N' neighbour counts for all four. EDGE_EXISTS keeps 0 / 1 per edge and no
 * occurrence count (counts[8] = 0).
 * Orientation: the reference keeps a node under whichever strand reached the map first (an order MPI decides); this
 * library keeps the lexicographically smaller strand and turns the edges with it (reverse_complement_edges,
 * de_bruijn_node_trait.hpp:122-124). The node set is the same, and so is every node up to that flip.
 * Input is FASTQ without a sequence filter (what the engine is instantiated with, :96,195); kmi_config.strand,
 * index_kind and dist_trans are not consulted. erase is not provided. */
typedef struct kmi_dbg kmi_dbg;
enum { KMI_DBG_EDGE_COUNTS = 0, KMI_DBG_EDGE_EXISTS = 1 };
#define KMI_DBG_VALUE_WORDS 5 /* a node value in kmi_results.values: uint32_t counts[9] + one uint32_t of padding */
kmi_status kmi_dbg_create(kmi_ctx *ctx, const kmi_config *cfg, uint32_t node_kind, kmi_dbg **out);   /* NodeMap(comm) */
/* the SeqParser template argument of the engine's build_* (KMI_FMT_FASTQ / KMI_FMT_FASTA): on FASTA a record's characters are its
 * sequence lines without their EOLs, edges follow the characters across line ends and never across a header */
kmi_status kmi_dbg_set_seq_format(kmi_dbg *g, uint32_t seq_format);
kmi_status kmi_dbg_destroy(kmi_dbg *g);
kmi_status kmi_dbg_clear(kmi_dbg *g);
kmi_status kmi_dbg_local_size(kmi_dbg *g, uint64_t *n);                                               /* local_size() */
/* de_bruijn_parser::operator() over every record of a FASTQ buffer (de_bruijn_construct_engine.hpp:109-157):
 * records of n_words + 1 words, the k-mer as parsed and the edge byte (edge_iterator.hpp:163-177) in the low
 * byte of the last word. out_records_dev == NULL: count only. */
kmi_status kmi_dbg_parse_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                             uint64_t *out_records_dev, size_t out_capacity, uint64_t *n_tuples);
/* build_posix / build_mmap<FASTQParser> on one rank (kmer_index.hpp:239-372 with the parser above): parse + insert */
kmi_status kmi_dbg_build_dev(kmi_dbg *g, const uint8_t *bytes_dev, size_t n_bytes);
kmi_status kmi_dbg_build_host(kmi_dbg *g, const uint8_t *bytes, size_t n_bytes);
/* insert(std::vector<std::pair<Kmer, uint8_t>>&) (de_bruijn_nodes_distributed.hpp:230-264, local_insert :91-159):
 * records as kmi_dbg_parse_dev emits them, either strand */
kmi_status kmi_dbg_insert_dev(kmi_dbg *g, const uint64_t *records_dev, size_t n);
kmi_status kmi_dbg_insert_host(kmi_dbg *g, const uint64_t *records, size_t n);
/* find(): one (stored k-mer, node) per distinct query key that is a node, KMI_DBG_VALUE_WORDS words per value;
 * a query matches under either strand. count(): 0 / 1 per distinct query key (values: one word each). */
kmi_status kmi_dbg_find_host(kmi_dbg *g, const uint64_t *queries, size_t nq, kmi_results *out);
kmi_status kmi_dbg_find_dev(kmi_dbg *g, const uint64_t *queries_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_values_dev,
                            uint64_t *n_out);
kmi_status kmi_dbg_count_host(kmi_dbg *g, const uint64_t *queries, size_t nq, kmi_results *out);
/* the local nodes: keys[n * n_words], counts9[n * 9] (either may be NULL) */
kmi_status kmi_dbg_export_host(kmi_dbg *g, uint64_t *keys, uint32_t *counts9, size_t capacity, uint64_t *n);
/* over the ranks of a communicator: every rank parses its record-aligned share, the tuples travel to the rank
 * KeyToRank gives their canonical k-mer (the distribute step of insert, de_bruijn_nodes_distributed.hpp:243-250) */
kmi_status kmi_dbg_build_dist_host(kmi_dbg *g, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes);
kmi_status kmi_dbg_find_dist_host(kmi_dbg *g, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out); /* find(): collective */
/* the engine's build_posix / build_mmap(filename) with comm.size() > 1: as kmi_index_build_range_dist_host (FASTQ byte range +
 * look-ahead, cut at record starts on the device, then the collective build) */
kmi_status kmi_dbg_build_range_dist_host(kmi_dbg *g, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                         uint64_t nominal_bytes, int reaches_eof, int *need_more);
/* nodes.erase(keys): the nodes of the query keys (either strand) leave the map with their edge counts (the erase the node map
 * inherits from the distributed map, distributed_unordered_map.hpp:719-779); *n_erased = nodes removed (here / on this rank) */
kmi_status kmi_dbg_erase_host(kmi_dbg *g, const uint64_t *queries, size_t nq, uint64_t *n_erased);
kmi_status kmi_dbg_erase_dist_host(kmi_dbg *g, kmi_comm *comm, const uint64_t *queries, size_t nq, uint64_t *n_erased_local);
kmi_status kmi_dbg_count_dist_host(kmi_dbg *g, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out); /* count(): collective */
kmi_status kmi_dbg_size_dist(kmi_dbg *g, kmi_comm *comm, uint64_t *n);                                /* size() */

/* ---- measurement support --------------------------------------------------- */
/* per-kernel HIP-event timing on the context's stream (bench.py roofline leg) */
kmi_status kmi_profile_enable(kmi_ctx *ctx, int on);
kmi_status kmi_profile_reset(kmi_ctx *ctx);
/* writes up to cap records; returns count in *n. name points to static storage. */
typedef struct { const char *name; double total_ms; uint64_t launches; uint64_t units; } kmi_kernel_time;
kmi_status kmi_profile_get(kmi_ctx *ctx, kmi_kernel_time *out, size_t cap, size_t *n);

/* deterministic synthetic inputs of SURVEY.md 8(d) (host side, splitmix64):
 * genome of g bases, r reads of read_len as 4-line FASTQ records with a 9-digit id
 * (315 bytes per 150-bp record). reads [first_read, first_read + n_reads) of the
 * data set defined by (seed, genome_len). Returns bytes written. */
size_t kmi_synth_fastq_bytes(uint64_t n_reads, uint32_t read_len);
kmi_status kmi_synth_fastq(uint64_t seed, uint64_t genome_len, uint32_t read_len, uint64_t first_read,
                           uint64_t n_reads, uint8_t *out, size_t out_capacity, uint32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* KMERIND_HIP_H */
