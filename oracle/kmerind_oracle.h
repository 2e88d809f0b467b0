/*
 * kmerind_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the ParBLiSS/kmerind k-mer index hot path:
 * alphabet -> Kmer pack/slide -> reverse complement / canonical -> Murmur3 /
 * FarmHash -> FASTQ/FASTA record rules -> tuple parsers -> KeyToRank + stable
 * bucket permutation -> counting / multi map insert, count, find, erase.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline.  The product
 * (kmerind_amd/, include/) never links, imports or executes anything here.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - pack/slide, reverse, compare: golden vectors of the reference's own
 *     src/common/test/test_kmer.cpp (tests/golden/kmer_golden.json).
 *   - Murmur3 / FarmHash: bit-compared with oracle/_ref/libkmerind_refhash.so,
 *     which is compiled directly from the reference's vendored
 *     ext/smhasher/MurmurHash3.cpp and ext/farmhash/src/farmhash.cc.
 *   - parse counts: the TestFileInfo tables of
 *     src/io/test/mpi_test_fastq_seq_parse.cpp:446-459 on test/data files.
 *   - index contents: known answers recorded in SURVEY.md section 8(c).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference checkout).
 */
#ifndef KMERIND_ORACLE_H
#define KMERIND_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_WORDS 4 /* up to 256-bit k-mers */

/* RNA / RNA5 (= RNA6): alphabets.hpp:365-445, 448-530 -- the DNA / DNA6 tables with U in the place of T */
enum { ORC_DNA = 0, ORC_DNA5 = 1, ORC_RNA = 2, ORC_RNA5 = 3, ORC_DNA16 = 4 };   /* DNA16: alphabets.hpp:648-733 */
#define ORC_IS_2BIT(a) ((a) == ORC_DNA || (a) == ORC_RNA)
enum { ORC_STRAND_SINGLE = 0, ORC_STRAND_CANONICAL = 1, ORC_STRAND_BIMOLECULE = 2 };
enum { ORC_HASH_MURMUR = 0, ORC_HASH_FARM = 1, ORC_HASH_IDENTITY = 2, ORC_HASH_STD = 3 };
enum { ORC_FMT_FASTQ = 0, ORC_FMT_FASTA = 1 };

/* Kmer<K, Alphabet, uint64_t> shape: src/common/kmer.hpp:116-177, padding.hpp:67-90 */
typedef struct {
  uint32_t k;
  uint32_t alphabet;      /* ORC_DNA (2 bits) or ORC_DNA5 (=DNA6, 3 bits) */
  uint32_t bits_per_char; /* derived */
  uint32_t n_bits;        /* derived: k * bits_per_char */
  uint32_t n_words;       /* derived: ceil(n_bits / 64) */
  uint32_t n_bytes;       /* derived: ceil(n_bits / 8) -- the hashed length */
} orc_kspec;

int orc_kspec_init(orc_kspec *s, uint32_t k, uint32_t alphabet);

/* ---- alphabets: src/common/alphabets.hpp:139-161 (DNA), :225-248 (DNA6=DNA5) */
uint8_t orc_from_ascii(uint32_t alphabet, uint8_t c);
uint8_t orc_complement(uint32_t alphabet, uint8_t code);

/* ---- Kmer value ops (words: data[0] least significant) */
void orc_kmer_clear(const orc_kspec *s, uint64_t *kmer);
/* Kmer::nextFromChar: kmer.hpp:731-741,1418-1430,1454-1460 */
void orc_kmer_next_from_char(const orc_kspec *s, uint64_t *kmer, uint8_t code);
/* Kmer::nextReverseFromChar: kmer.hpp:758-768,1436-1448 */
void orc_kmer_next_reverse_from_char(const orc_kspec *s, uint64_t *kmer, uint8_t code);
/* Kmer::reverse: kmer.hpp:1615-1679 (group reverse, no complement) */
void orc_kmer_reverse(const orc_kspec *s, const uint64_t *in, uint64_t *out);
/* Kmer::reverse_complement: kmer.hpp:1118-1127,1723-1742,1807-1847 */
void orc_kmer_revcomp(const orc_kspec *s, const uint64_t *in, uint64_t *out);
/* operator< / operator==: kmer.hpp:790-794,820-823 */
int orc_kmer_less(const orc_kspec *s, const uint64_t *a, const uint64_t *b);
int orc_kmer_equal(const orc_kspec *s, const uint64_t *a, const uint64_t *b);
/* transform::lex_less: src/common/kmer_transform.hpp:108-116 */
void orc_kmer_canonical(const orc_kspec *s, const uint64_t *in, uint64_t *out);
/* transform::xor_rev_comp: kmer_transform.hpp:131-145 */
void orc_kmer_xor_revcomp(const orc_kspec *s, const uint64_t *in, uint64_t *out);
/* fill from an ASCII string of >= k chars (fillFromChars, kmer.hpp:543-565) */
void orc_kmer_from_ascii(const orc_kspec *s, const uint8_t *chars, uint64_t *out);
/* batch forms over n k-mers stored contiguously (n * n_words words) */
void orc_kmers_revcomp(const orc_kspec *s, const uint64_t *in, size_t n, uint64_t *out);
void orc_kmers_canonical(const orc_kspec *s, const uint64_t *in, size_t n, uint64_t *out);
/* SWAR form the reference executes for DNA (bitgroup_ops.hpp:489-515); baseline driver only */
void orc_kmer_revcomp_fast(const orc_kspec *s, const uint64_t *in, uint64_t *out);
void orc_kmers_revcomp_fast(const orc_kspec *s, const uint64_t *in, size_t n, uint64_t *out);

/* ---- hashes */
/* MurmurHash3_x64_128: ext/smhasher/MurmurHash3.cpp:255-335 */
void orc_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2]);
/* util::Hash64WithSeed -> farmhashna: ext/farmhash/src/farmhash.cc:373-414,
 * 455-466,519-529,1469-1471; len <= 64 only (k-mers are <= 32 bytes). */
uint64_t orc_farm_hash64_with_seed(const void *key, size_t len, uint64_t seed);
/* 0 (default): reference default build (RelWithDebInfo, no NDEBUG => farmhash
 * DebugTweak active); 1: Release/-DNDEBUG behaviour. CMakeLists.txt:26,200-204 */
void orc_set_farm_ndebug(int on);
/* bliss::kmer::hash::murmur / farm <KMER, Prefix>: src/index/kmer_hash.hpp:242-311 */
uint64_t orc_kmer_hash(const orc_kspec *s, uint32_t which, int prefix, const uint64_t *kmer);
void orc_kmers_hash(const orc_kspec *s, uint32_t which, int prefix, const uint64_t *kmers,
                    size_t n, uint64_t *out);
/* KeyToRank: src/containers/distributed_unordered_map.hpp:148-170
 * rank = DistHash(DistTrans(key)) % p ; strand selects DistTrans per
 * src/index/kmer_index.hpp:436-481 (bimolecule -> lex_less, else identity). */
uint64_t orc_kmer_hash_ex(const orc_kspec *s, uint32_t which, int prefix, unsigned prefix_bits, const uint64_t *kmer);
unsigned orc_ceil_log2(unsigned n);
void orc_key_to_rank(const orc_kspec *s, uint32_t dist_hash, uint32_t strand,
                     const uint64_t *kmers, size_t n, uint32_t p, uint32_t *ranks);
void orc_key_to_rank_ex(const orc_kspec *s, uint32_t dist_hash, uint32_t strand, uint32_t dist_trans,
                        const uint64_t *kmers, size_t n, uint32_t p, uint32_t *ranks);

/* ---- sequence records: src/io/fastq_loader.hpp:389-467, fasta_loader.hpp:485-723 */
typedef struct {
  uint64_t record_offset;    /* file offset of the record start ('@' or '>') = id */
  uint64_t record_size;      /* bytes, incl. trailing EOLs consumed */
  uint64_t seq_begin;        /* file offset of first sequence byte */
  uint64_t seq_end;          /* one past last sequence byte (may include EOLs for FASTA) */
  uint64_t qual_begin, qual_end; /* FASTQ only */
  uint64_t seq_index;        /* FASTA: ordinal of the sequence in the file */
} orc_record;

/* Parses records of `bytes[0..n)`; bytes[0] is at file offset `file_offset` and
 * must be a record start. Returns number of records, or -1 on a parse error
 * (the cases where the reference throws). out may be NULL to count only. */
long orc_fastq_records(const uint8_t *bytes, size_t n, uint64_t file_offset,
                       orc_record *out, size_t out_cap);
long orc_fasta_records(const uint8_t *bytes, size_t n, uint64_t file_offset,
                       orc_record *out, size_t out_cap);

/* ---- tuple parsers: src/io/kmer_parser.hpp (KmerParser :85-294, Count :909-1083,
 * Position :303-569, PositionQuality :577-900) driven by
 * KmerFileHelper::read_block_old (src/io/kmer_file_helper.hpp:110-186).
 *
 * Whole-buffer form: valid range == [file_offset, file_offset+n).
 * Outputs (any may be NULL): kmers (n_words words each, as-parsed i.e. no
 * strand transform), ids (Short/LongSequenceKmerId packed u64), quals (float).
 * Returns number of tuples; *n_seqs gets the sequence count; -1 on parse error.
 * Call with all outputs NULL to size. */
long orc_extract(const orc_kspec *s, uint32_t fmt, const uint8_t *bytes, size_t n,
                 uint64_t file_offset, uint64_t *kmers, uint64_t *ids, float *quals,
                 size_t out_cap, size_t *n_seqs);

/* the same behind one of the reference's filtering sequence iterators (filtered_sequence_iterator.hpp:154-165, 166-440) */
enum { ORC_SEQ_ALL = 0, ORC_SEQ_N_FILTER = 1, ORC_SEQ_N_SPLIT = 2 };
long orc_extract_filtered(const orc_kspec *s, uint32_t fmt, uint32_t seq_filter, const uint8_t *bytes, size_t n,
                          uint64_t file_offset, uint64_t *kmers, uint64_t *ids, float *quals,
                          size_t out_cap, size_t *n_seqs, size_t *n_yield);

/* quality: Illumina18 codec LUT + sliding window,
 * src/index/quality_scores.hpp:88-341, quality_score_iterator.hpp:67-173 */
float orc_qual_lut(uint8_t phred_char);

/* ---- distribute: src/io/incremental_mxx.hpp:273-364,595-640 (stable bucket) */
void orc_stable_bucket(const uint32_t *ranks, size_t n, uint32_t p, uint64_t *bucket_sizes,
                       uint64_t *i2o);

/* ---- counting map (reduction_unordered_map / counting_unordered_map semantics,
 * src/containers/distributed_unordered_map.hpp:1603-1618,1826-1884): a chained
 * hash table keyed by StoreTrans(key) with StoreHash = murmur/farm (Prefix=false). */
typedef struct orc_count_map orc_count_map;
orc_count_map *orc_count_map_create(const orc_kspec *s, uint32_t strand, uint32_t store_hash);
void orc_count_map_destroy(orc_count_map *m);
/* insert(vector<Key>): applies InputTransform (canonical strand -> lex_less) then count[key] += 1 */
void orc_count_map_insert(orc_count_map *m, const uint64_t *kmers, size_t n);
size_t orc_count_map_size(const orc_count_map *m);
/* to_vector(): fills keys (n_words each) and counts; order unspecified */
size_t orc_count_map_export(const orc_count_map *m, uint64_t *keys, uint32_t *counts);
/* count(): distributed_unordered_map.hpp:880-983 -- one (key,count) per distinct
 * transformed query key, 0 when absent. Returns number of results. */
size_t orc_count_map_count(const orc_count_map *m, const uint64_t *queries, size_t nq,
                           uint64_t *out_keys, uint64_t *out_counts);
/* find(): :564-687 -- (key,value) of present distinct transformed query keys only */
size_t orc_count_map_find(const orc_count_map *m, const uint64_t *queries, size_t nq,
                          uint64_t *out_keys, uint32_t *out_counts);
/* erase(): :719-779 */
size_t orc_count_map_erase(orc_count_map *m, const uint64_t *queries, size_t nq);

/* ---- multimap (::dsc::unordered_multimap semantics, distributed_unordered_map.hpp:1466-1515,
 * :231-238, :1100-1131, :1292-1328) for PositionIndex / PositionQualityIndex */
typedef struct orc_multi_map orc_multi_map;
orc_multi_map *orc_multi_map_create(const orc_kspec *s, uint32_t strand, uint32_t store_hash, uint32_t value_words);
void orc_multi_map_destroy(orc_multi_map *m);
void orc_multi_map_insert(orc_multi_map *m, const uint64_t *kmers, const uint64_t *values, size_t n);
size_t orc_multi_map_size(const orc_multi_map *m);
size_t orc_multi_map_export(const orc_multi_map *m, uint64_t *keys, uint64_t *values);
size_t orc_multi_map_count(const orc_multi_map *m, const uint64_t *queries, size_t nq, uint64_t *out_keys, uint64_t *out_counts);
size_t orc_multi_map_find(const orc_multi_map *m, const uint64_t *queries, size_t nq, uint64_t *out_keys, uint64_t *out_values, size_t cap);
size_t orc_multi_map_erase(orc_multi_map *m, const uint64_t *queries, size_t nq);

/* ---- de Bruijn graph nodes: test/test/debruijn/ (the reference's only in-tree consumer of Index).
 * No expected values are held by the reference for this path (its test prints sizes only): the functions below restate
 * edge_iterator.hpp:84-177, de_bruijn_node_trait.hpp:119-131,139-336 and de_bruijn_nodes_distributed.hpp:91-159 literally,
 * and tests/ pins them on hand-worked vectors. */
/* de_bruijn_parser::operator() (de_bruijn_construct_engine.hpp:109-157) over every FASTQ record: k-mers as parsed
 * (no strand transform) zipped with edge_iterator<CharIter, DNA16>: high nibble = DNA16 code of the base left of the k-mer
 * in the read, low nibble = base right of it, 0 where the read ends. Returns the tuple count, -1 on a parse error. */
long orc_dbg_parse(const orc_kspec *s, const uint8_t *bytes, size_t n, uint64_t *kmers, uint8_t *edges, size_t out_cap);
/* the same with the input format given (ORC_FMT_FASTQ / ORC_FMT_FASTA) */
long orc_dbg_parse_fmt(const orc_kspec *s, uint32_t fmt, const uint8_t *bytes, size_t n, uint64_t *kmers, uint8_t *edges, size_t out_cap);
/* input_edge_utils::reverse_complement_edges<DNA16> (de_bruijn_node_trait.hpp:122-124) */
uint8_t orc_dbg_edges_revcomp(uint8_t exts);
/* de_bruijn_nodes_distributed<Kmer, edge_counts<DNA16, int32_t> | edge_exists<DNA16>, BimoleculeHashMapParams>:
 * a node is found under either strand and keeps the strand it was created with (the first one inserted);
 * counts[0..3] = out A C G T, [4..7] = in A C G T, [8] = occurrences of the k-mer (edge_counts::update, :200-239).
 * edge_exists keeps the OR of the edge bytes: export gives it as counts[i] = bit i, counts[8] = 0. */
typedef struct orc_dbg_map orc_dbg_map;
orc_dbg_map *orc_dbg_map_create(const orc_kspec *s, uint32_t store_hash, int exists_only);
void orc_dbg_map_destroy(orc_dbg_map *m);
void orc_dbg_map_insert(orc_dbg_map *m, const uint64_t *kmers, const uint8_t *edges, size_t n);
size_t orc_dbg_map_size(const orc_dbg_map *m);
/* canonical_orientation != 0: every node is given in the orientation of the lexicographically smaller strand (key
 * reverse-complemented, in/out counts swapped and complemented) -- the form the device library stores */
size_t orc_dbg_map_export(const orc_dbg_map *m, uint64_t *keys, uint32_t *counts9, int canonical_orientation);
size_t orc_dbg_map_find(orc_dbg_map *m, const uint64_t *queries, size_t nq, uint64_t *out_keys, uint32_t *out_counts9,
                        int canonical_orientation);

/* ---- CPU baseline driver ("port" of the reference MPI path with T thread-ranks):
 * per rank parse (record-aligned byte range) -> KeyToRank (murmur h[1] % T) ->
 * stable bucket -> in-memory exchange -> per-rank counting map insert.
 * Returns elapsed seconds (read+insert sections); outputs totals. */
/* the thread-rank build of a whole FASTQ buffer in `slices` pieces (maps kept across them) with checksums of the result:
 * out[0] k-mers, [1] distinct keys, [2] sum of counts, [3] sum of key * count, [4] xor of key * (2 count + 1) (mod 2^64) */
int orc_count_full(const uint8_t *bytes, size_t n, uint32_t k, uint32_t strand, uint32_t threads, uint32_t slices, uint64_t *out);
/* record-aligned position at or after pos (fastq_loader.hpp:269-364, the 4-line rule) */
size_t orc_fastq_align(const uint8_t *bytes, size_t n, size_t pos);
double orc_bench_count_index(const uint8_t *bytes, size_t n, uint32_t k, uint32_t strand,
                             uint32_t threads, uint64_t *n_kmers, uint64_t *n_distinct);

void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
