// kmi_internal.h -- host-side context, workspace and launch helpers (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/kmerind_hip.h"
#include "kmi_device.h"

namespace kmi {

enum WsSlot {
  WS_TILE_INFO = 0,   // per-tile scan records
  WS_TILE_BASE,       // per-tile line bases
  WS_TILE_OFF,        // per-tile output offsets
  WS_KEYS_A,          // key ping buffer
  WS_KEYS_B,          // key pong buffer
  WS_HIST,            // fine histogram / offsets
  WS_WGHIST,          // per-workgroup coarse histograms
  WS_CURSOR,          // coarse / fine cursors
  WS_TMP_KEYS,        // per-bucket reduce output (keys)
  WS_TMP_VALS,        // per-bucket reduce output (values)
  WS_BUCKET_CNT,      // per-bucket distinct counts
  WS_BUCKET_OFF,      // scanned
  WS_QUERY_A,
  WS_QUERY_B,
  WS_INPUT,           // host-entry staging of input bytes / keys
  WS_OUTPUT,          // host-entry staging of outputs
  WS_OUTPUT2,
  WS_INPUT2,
  WS_MISC,
  WS_TILE_HDR,        // per-tile position of the last record start before the tile
  WS_QUALS,           // k-mer qualities of the extracted tuples
  WS_READS,           // read descriptors for the quality pass
  WS_FA_IDS,          // LongSequenceKmerId of every compacted FASTA character
  WS_PK_EOL,          // EOL bitmap of the scanned input
  WS_PK_STREAM,       // packed complement-code stream of the scanned input
  WS_ENT_BKT,         // rank bucket of the 8 windows of every entry, one byte each (fused extract + route)
  WS_ENT_LIST,        // entry list: runs of <= 8 consecutive windows, per-tile slots (fused build)
  WS_ENT_CNT,         // entries per scan tile
  WS_PK_BRK,          // window-break bitmap (EOL or N) of the scanned input (N_FILTER / N_SPLIT)
  WS_PK_NB,           // N bitmap and pure EOL bitmap of the pre-pass (N_FILTER)
  WS_SPLIT_RANK,      // destination rank of every index entry (split by rank)
  WS_SPLIT_OFF,       // per-part bucket offsets, part totals and bases (split / merge)
  WS_SK_ITEMS,        // super-k-mer items of the minimizer pass (fused build through super-k-mers)
  WS_DIST_A,          // the collectives over ranks (kmi_index_*_dist_*): send / receive / result buffers
  WS_DIST_B,
  WS_DIST_C,
  WS_DIST_D,
  WS_DBG_RECS,        // de Bruijn node build: (k-mer, edge) records of the parsed input
  WS_DBG_OLD,         // ... keys and bucket offsets of the nodes before an insert (their edge counts move to the new order)
  WS_DBG_POS,         // ... entry positions of the nodes a find() hit
  WS_REDO,            // sk_reduce2's redo list: buckets that go through sk_reduce behind it
  WS_ALIGNED,         // 16-byte aligned copy of an input buffer that arrived at an odd address (a batch inside a larger buffer)
  WS_NUM_SLOTS
};

struct ProfRec {
  const char *name;
  hipEvent_t e0, e1;
  uint64_t units;
};

struct ProfAgg {
  const char *name;
  double total_ms;
  uint64_t launches, units;
};

}  // namespace kmi

struct kmi_ctx {
  int device = 0, rank = 0, nranks = 1;
  hipStream_t stream = nullptr;
  std::string err;
  struct Buf { void *p = nullptr; size_t cap = 0; } ws[kmi::WS_NUM_SLOTS];
  uint32_t *d_flags = nullptr;   // [64] error / overflow flags, pass-structure votes (16..33), work-queue words (40..)
  uint32_t n_cus = 0;            // compute units of the device (grids of persistent workgroups)
  uint64_t *d_totals = nullptr;  // [16] small device scalars + [256] coarse-bucket totals of the fine-offset scan
  hipEvent_t ev_mail = nullptr;  // marks "the read-backs queued so far have landed" (waited for instead of the whole stream)
  // a build from HOST bytes whose copy has not been queued yet (kmi_index_build_host): the one-pass front end queues it in chunks
  // on copy_stream and starts the byte ranges of a chunk behind it; every other reader of the input calls feed_flush first
  const uint8_t *feed_host = nullptr; uint8_t *feed_dev = nullptr; size_t feed_bytes = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t feed_ev[17] = {};
  // the de Bruijn node build through super-k-mers (kmi_debruijn.h): the front end makes records that carry their two outside bases, and
  // the back end leaves word where the fine buckets' records lie (they stay in the workspace until the next build) for the edge pass
  bool edge_records = false;
  bool dbg_superkmer = true;     // KMI_DBG_SUPERKMER=0: the node build always takes the tuple path
  struct { const uint64_t *recs = nullptr, *rec_off = nullptr, *region = nullptr; const uint32_t *cap = nullptr, *cnt = nullptr; bool valid = false; } sk_left;
  bool sk_fine_lines = true;     // the super-k-mer build's fine pass writes whole lines + pad records (sk_scatter_fine_slack_lines_kernel); KMI_SK_FINE_LINES=0: the plain form
  bool lines_p2 = true;          // the fine pass of the position builds writes whole lines (scatter_lines_records); KMI_LINES_P2=0: the plain form
  bool host_overlap = true;      // KMI_HOST_OVERLAP=0: one copy on the build's stream, then the build
  size_t host_overlap_min = (size_t)64 << 20;   // ... and inputs below this many bytes always go that way (KMI_HOST_OVERLAP_MIN; tests lower it)
  size_t feed_min_chunk = (size_t)8 << 20;      // a copy chunk is not split further below this (KMI_FEED_MIN_CHUNK)
  uint64_t *h_totals = nullptr;  // pinned mirror: 16 words of totals, then 1024 words for larger read-backs (one synchronisation for all of them)
  bool prof = false;
  std::vector<kmi::ProfRec> prof_pending;
  std::vector<kmi::ProfAgg> prof_agg;
  std::vector<hipEvent_t> event_pool;
  // released index arrays kept for the next build of the same size (a repeated build / clear cycle then never
  // reaches hipMalloc / hipFree, whose cost on a loaded node is unpredictable)
  struct Spare { void *p; size_t bytes; };
  std::vector<Spare> spare;
  bool fused_superkmer = true;   // fused count-index build through super-k-mers (KMI_FUSED_PATH=kmer in the environment: the k-mer pipeline)
  bool force_dist = false;       // KMI_FORCE_DIST=1: the *_dist_* entry points run their exchange even with one rank (RCCL self exchange: tests)
  uint32_t sk_level_hint = 0;    // sk_reduce: filter bits the buckets of the next build start with (majority of the last build)
  float sk_inv_dup = 0.f;        // sk_reduce: distinct k-mers per k-mer occurrence of the last build (a bucket's expected fill; 0: unknown)
  bool sk_reduce2 = false;       // KMI_SK_REDUCE=2: sk_reduce2 (wavefront-private tables over sorted bins, kmi_reduce2.h) ahead of sk_reduce -- measured slower
  uint32_t sk_r2_win = 0;        // records of sk_reduce2's batch window (KMI_R2_WIN: test knob; 0: by the last build's duplication)
  uint64_t alloc_us = 0, alloc_bytes = 0, alloc_calls = 0, alloc_reused = 0;   // time inside hipMalloc / hipFree, bytes and calls that reached hipMalloc, blocks taken from the process-wide cache (kmi_ctx_debug_counter 1..4)
  uint32_t dist_pool_regrows = 0;     // times a build over ranks had to enlarge its receive pool (kmi_ctx_debug_counter: tests)
  uint32_t dist_pool_pct = 100;       // the estimate itself, in percent (KMI_DIST_POOL_PCT: tests make it too small)
  uint64_t dist_pool_slack = 65536;   // records a rank's receive pool holds beyond the estimate of its share (KMI_DIST_POOL_SLACK: tests shrink it so that the pool has to grow)
  uint32_t dist_chunks = 4;      // record-aligned chunks of a rank's share in the build over ranks (exchange of one beside the front end of the next; KMI_DIST_CHUNKS)
  bool tuples_from_parse = true; // position / position + quality builds partition their tuples straight from the parse (kmi_tuples.h); KMI_TUPLES=extract: extract, then partition
  bool sk_slack = true;          // fine buckets with room instead of a counting pass (sk_scatter_fine_slack_kernel); KMI_SK_SLACK=0: always count
  bool front_fused = true;       // FASTQ front end of the super-k-mer build in one pass (kmi_front.h); KMI_FRONT=general: scan + list + minimizer
  uint32_t front_waves = 0;      // resident wavefronts of the front kernel (ranges of a large input); 0: not asked yet
  uint64_t front_min_range = 64ull << 10;   // smallest byte range of a wavefront (KMI_FRONT_MIN_RANGE: tests shrink it)
  uint64_t sparse_min = 1ull << 26;   // output slots from which a super-k-mer build leaves its index in the sparse form (KMI_SPARSE_MIN)
  int sk_dbg = 0;                // KMI_SK_DBG=7 (test knob): the super-k-mer front end reports a capacity as exceeded
  bool fa_part_set = false;      // kmi_ctx_set_fasta_partition
  kmi_fasta_partition fa_part{};
};

namespace kmi {

// hipMalloc / hipFree of the library's large blocks, timed, behind a process-wide cache per device (kmi_api.hip): what a context
// gives back when it is destroyed waits there for the next context of the same device instead of going through hipFree and
// hipMalloc again -- on a loaded node a fresh context's first build otherwise pays hundreds of milliseconds for its workspace.
// dev_malloc takes a cached block of at least `bytes` and at most 1.5 x that; *got receives the block's real size.
hipError_t dev_malloc(kmi_ctx *ctx, void **p, size_t bytes, size_t *got = nullptr);
void dev_free(kmi_ctx *ctx, void *p);                       // hipFree, timed
void dev_retire(kmi_ctx *ctx, void *p, size_t bytes);       // into the process-wide cache (hipFree when that is full)

// device blocks of the index arrays: exact-size reuse from the context's spare list, else hipMalloc
inline hipError_t pool_alloc(kmi_ctx *ctx, void **p, size_t bytes) {
  if (bytes == 0) bytes = 256;
  for (size_t i = 0; i < ctx->spare.size(); ++i)
    if (ctx->spare[i].bytes == bytes) { *p = ctx->spare[i].p; ctx->spare.erase(ctx->spare.begin() + (long)i); return hipSuccess; }
  hipError_t e = dev_malloc(ctx, p, bytes);   // (index arrays are asked for by exact size: a cached block must match it)
  if (e != hipSuccess && !ctx->spare.empty()) {   // give the cached blocks back and retry once
    for (auto &b : ctx->spare) dev_free(ctx, b.p);
    ctx->spare.clear();
    e = dev_malloc(ctx, p, bytes);
  }
  return e;
}
inline void pool_free(kmi_ctx *ctx, void *p, size_t bytes) {
  if (!p) return;
  if (bytes == 0) bytes = 256;
  // a few blocks wait here for the next owner (an index's arrays, a workspace slot: ws_get). When the list is full the SMALLEST one
  // goes -- offset / count tables of a few hundred KB come and go with every build and must not push out the multi-GB buffers a
  // sparse index hands back (giving those to hipFree and asking hipMalloc for them again costs hundreds of milliseconds per build)
  constexpr size_t kMaxSpare = 12;
  if (ctx->spare.size() >= kMaxSpare) {
    size_t m = 0;
    for (size_t i = 1; i < ctx->spare.size(); ++i) if (ctx->spare[i].bytes < ctx->spare[m].bytes) m = i;
    if (ctx->spare[m].bytes >= bytes) { dev_free(ctx, p); return; }   // (the newcomer is the smallest)
    dev_free(ctx, ctx->spare[m].p);
    ctx->spare.erase(ctx->spare.begin() + (long)m);
  }
  ctx->spare.push_back({p, bytes});
  // ... and the list never holds more than 32 GB: the OLDEST blocks go first then (the arrays of a multimap that grows batch by
  // batch come back in sizes nobody asks for again)
  constexpr size_t kMaxSpareBytes = 32ull << 30;
  size_t total = 0;
  for (const auto &b : ctx->spare) total += b.bytes;
  while (total > kMaxSpareBytes && !ctx->spare.empty()) {
    total -= ctx->spare.front().bytes;
    dev_free(ctx, ctx->spare.front().p);
    ctx->spare.erase(ctx->spare.begin());
  }
}

inline kmi_status set_err(kmi_ctx *ctx, kmi_status st, const char *fmt, const char *a = "", const char *b = "") {
  if (ctx) {
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, a, b);
    ctx->err = buf;
  }
  return st;
}

#define KMI_HIP(ctx, call)                                                              \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) return kmi::set_err((ctx), KMI_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

#define KMI_TRY(expr)                       \
  do {                                      \
    kmi_status s__ = (expr);                \
    if (s__ != KMI_OK) return s__;          \
  } while (0)

// workspace slot of at least `bytes` (contents are not preserved when it grows)
kmi_status ws_get(kmi_ctx *ctx, WsSlot slot, size_t bytes, void **out);
void ws_release(kmi_ctx *ctx, WsSlot slot);
bool ws_detach(kmi_ctx *ctx, WsSlot slot, const void *p, size_t *bytes);

// profiling hooks around one kernel launch
void prof_begin(kmi_ctx *ctx, const char *name, uint64_t units);
void prof_end(kmi_ctx *ctx, size_t slot);

struct ProfScope {   // scopes may nest: each closes the record it opened
  kmi_ctx *c;
  size_t slot;
  bool on;
  ProfScope(kmi_ctx *ctx, const char *name, uint64_t units) : c(ctx), slot(0), on(ctx->prof) { if (on) { slot = c->prof_pending.size(); prof_begin(c, name, units); } }
  ~ProfScope() { if (on) prof_end(c, slot); }
};

inline bool valid_config(const kmi_config *cfg, KShape *shape) {
  if (!cfg || cfg->k == 0) return false;
  uint32_t bits = (cfg->alphabet == KMI_ALPHA_DNA || cfg->alphabet == KMI_ALPHA_RNA) ? 2 :
                  ((cfg->alphabet == KMI_ALPHA_DNA5 || cfg->alphabet == KMI_ALPHA_RNA5) ? 3 : (cfg->alphabet == KMI_ALPHA_DNA16 ? 4 : 0));
  if (!bits) return false;
  KShape s = make_shape(cfg->k, bits);
  if (s.n_words > (uint32_t)kMaxWords) return false;
  if (cfg->strand > 2 || cfg->dist_hash > 3 || cfg->store_hash > 3 || cfg->seq_format > 1 || cfg->index_kind > 2 || cfg->seq_filter > 2) return false;
  if (cfg->seq_filter && cfg->seq_format == KMI_FMT_FASTA && cfg->k == 1) return false;   // break bits need k >= 2 there
  if (cfg->seq_filter && cfg->index_kind == KMI_INDEX_POSQUAL) return false;
  if (cfg->dist_trans > 2 || (cfg->dist_trans && cfg->strand != KMI_STRAND_SINGLE)) return false;
  if (shape) *shape = s;
  return true;
}

// RNA alphabets: the byte classifiers see T and U swapped (kmi_device.h swap_tu_dword)
inline bool is_rna(const kmi_config *cfg) { return cfg->alphabet == KMI_ALPHA_RNA || cfg->alphabet == KMI_ALPHA_RNA5; }

// dispatch on (n_words, bits)
#define KMI_DISPATCH(shape, FN, ...)                                                   \
  do {                                                                                 \
    if ((shape).bits == 2) {                                                           \
      switch ((shape).n_words) {                                                       \
        case 1: return FN<1, 2>(__VA_ARGS__);                                          \
        case 2: return FN<2, 2>(__VA_ARGS__);                                          \
        case 3: return FN<3, 2>(__VA_ARGS__);                                          \
        case 4: return FN<4, 2>(__VA_ARGS__);                                          \
      }                                                                                \
    } else if ((shape).bits == 3) {                                                    \
      switch ((shape).n_words) {                                                       \
        case 1: return FN<1, 3>(__VA_ARGS__);                                          \
        case 2: return FN<2, 3>(__VA_ARGS__);                                          \
        case 3: return FN<3, 3>(__VA_ARGS__);                                          \
        case 4: return FN<4, 3>(__VA_ARGS__);                                          \
      }                                                                                \
    } else {                                                                           \
      switch ((shape).n_words) {                                                       \
        case 1: return FN<1, 4>(__VA_ARGS__);                                          \
        case 2: return FN<2, 4>(__VA_ARGS__);                                          \
        case 3: return FN<3, 4>(__VA_ARGS__);                                          \
        case 4: return FN<4, 4>(__VA_ARGS__);                                          \
      }                                                                                \
    }                                                                                  \
    return KMI_ERR_INVALID;                                                            \
  } while (0)

// The byte kernels load 16 bytes per lane: an input that does not start on a 16-byte boundary (a record-aligned batch or
// partition inside a larger device buffer) is copied once to an aligned workspace buffer, device to device.
inline kmi_status align_input(kmi_ctx *ctx, const uint8_t **bytes_dev, size_t n_bytes) {
  if (n_bytes == 0 || (reinterpret_cast<uintptr_t>(*bytes_dev) & 15u) == 0) return KMI_OK;
  void *p;
  KMI_TRY(ws_get(ctx, WS_ALIGNED, n_bytes + 64, &p));
  KMI_HIP(ctx, hipMemcpyAsync(p, *bytes_dev, n_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  *bytes_dev = (const uint8_t *)p;
  return KMI_OK;
}

// ---- the RCCL exchange (kmi_comm.hip)
kmi_status comm_all_to_all_counts(kmi_comm *c, const uint64_t *send_counts, uint64_t *recv_counts);
kmi_status comm_all_to_all_v(kmi_comm *c, const void *send_dev, const uint64_t *send_counts, void *recv_dev, const uint64_t *recv_counts,
                             size_t elem_bytes);
kmi_status comm_allreduce_sum(kmi_comm *c, uint64_t *value);
kmi_status comm_allgather_words(kmi_comm *c, const uint64_t *mine, size_t n, uint64_t *all);
kmi_status comm_all_to_all_counts2(kmi_comm *c, const uint64_t *send_counts, uint64_t my_largest_bytes, uint64_t *recv_counts, uint64_t *largest_bytes);
kmi_status comm_all_to_all_v_async(kmi_comm *c, const void *send_dev, const uint64_t *send_counts, void *recv_dev, const uint64_t *recv_counts,
                                   size_t elem_bytes, uint64_t largest_bytes);
kmi_status comm_exchange_join(kmi_comm *c);
kmi_status comm_exchange_wait(kmi_comm *c);
kmi_ctx *comm_ctx(kmi_comm *c);
int comm_size(kmi_comm *c);
int comm_rank(kmi_comm *c);

// ---- entry points implemented across the .hip files
kmi_status extract_count(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                         uint64_t *n_tuples, uint64_t *n_seqs);
kmi_status extract_run(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                       uint64_t file_offset, uint64_t *out_kmers_dev, uint64_t *out_ids_dev, size_t out_capacity,
                       bool apply_strand, bool scan_done, uint64_t *n_tuples, uint64_t *n_seqs, float *out_quals_dev = nullptr,
                       uint32_t rec_words = 0, bool edges = false);
kmi_status upload_quality_lut(kmi_ctx *ctx);

// tile scan of a FASTQ partition; the packed arrays and per-tile line bases stay in the workspace
struct ReadDesc { uint64_t seq_pos, out_off; };   // slot = sequence index of the read; out_off = ~0: the read has no k-mer
struct FastqScan {
  uint64_t n_tiles, n_tuples, n_seqs, n_bytes, n_cover;
  const uint32_t *line_base;
  const uint64_t *hdr_base;   // [n_tiles] 1 + position of the last record start before the tile (0: none)
  const uint64_t *tile_off;   // [n_tiles + 1] k-mer windows before each scan tile
  const uint8_t *pk_eol, *pk_stream;
  const uint8_t *pk_brk;      // window-break bitmap of a sequence filter, or null
};
// check_lengths = false: the caller runs fastq_list_kernel, which carries the seq/qual length rule, and asks for
// fastq_length_verdict afterwards
kmi_status fastq_scan(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, FastqScan *out, bool check_lengths = true);
kmi_status fastq_length_verdict(kmi_ctx *ctx);
// the quality values of the scanned input's k-mers in file order (out_quals[t] for tuple t), from the read descriptors a pass over
// the windows has filled (slot = sequence index): fastq_quality_kernel. alloc_reads: a cleared descriptor array for sc.n_seqs reads
kmi_status fastq_quality_reads(kmi_ctx *ctx, const FastqScan &sc, ReadDesc **reads);
kmi_status fastq_quality_launch(kmi_ctx *ctx, const uint8_t *bytes_dev, const FastqScan &sc, uint32_t k, const ReadDesc *reads, float *out_quals);

// FASTA: byte-space passes -> compacted character stream (kmi_fasta.hip)
struct FastaScan {
  uint64_t n_chars, n_seqs, n_cover;
  uint64_t n_valid;            // characters whose byte lies in the valid range: k-mer windows start below this rank
  const uint8_t *pk_break;     // bit r: character r is the first of a record
  const uint8_t *pk_stream;    // BITS per character, complement codes
  const uint64_t *ids_by_rank; // or null
};
kmi_status fasta_scan(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset, bool want_ids,
                      FastaScan *out);

}  // namespace kmi
