// kmi_api.hip -- C ABI: context, memory, profiling, array-level k-mer ops, extract entry points
#include <stdlib.h>

#include <algorithm>

#include "kmi_block.h"
#include <chrono>
#include <mutex>

#include "kmi_internal.h"

namespace kmi {

// ---------------------------------------------------------------------------
// workspace + profiling
// ---------------------------------------------------------------------------
// ---- the process-wide block cache (see kmi_internal.h) -------------------------------------------------------------------
namespace {
struct CachedBlock { int device; void *p; size_t bytes; };
std::mutex g_cache_mu;
std::vector<CachedBlock> g_cache;
constexpr size_t kCacheMaxBytes = 96ull << 30, kCacheMaxBlocks = 64, kCacheMinBlock = 1ull << 20;
inline uint64_t now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// hipFree of every cached block of the device (all devices: -1); returns the bytes released
size_t cache_flush(int device) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  size_t freed = 0;
  for (size_t i = 0; i < g_cache.size();) {
    if (device < 0 || g_cache[i].device == device) { (void)hipFree(g_cache[i].p); freed += g_cache[i].bytes; g_cache.erase(g_cache.begin() + (long)i); }
    else ++i;
  }
  return freed;
}
}  // namespace

hipError_t dev_malloc(kmi_ctx *ctx, void **p, size_t bytes, size_t *got) {
  const bool exact = got == nullptr;
  if (bytes >= kCacheMinBlock) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t best = g_cache.size();
    for (size_t i = 0; i < g_cache.size(); ++i) {
      const CachedBlock &b = g_cache[i];
      if (b.device != ctx->device || b.bytes < bytes) continue;
      if (exact ? b.bytes != bytes : b.bytes > bytes + bytes / 2) continue;
      if (best == g_cache.size() || b.bytes < g_cache[best].bytes) best = i;
    }
    if (best != g_cache.size()) {
      *p = g_cache[best].p;
      if (got) *got = g_cache[best].bytes;
      g_cache.erase(g_cache.begin() + (long)best);
      ++ctx->alloc_reused;
      return hipSuccess;
    }
  }
  const uint64_t t0 = now_us();
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess && cache_flush(ctx->device)) { (void)hipGetLastError(); e = hipMalloc(p, bytes); }   // the cache holds what this request needs
  ctx->alloc_us += now_us() - t0; ctx->alloc_bytes += bytes; ++ctx->alloc_calls;
  if (got) *got = bytes;
  return e;
}
void dev_free(kmi_ctx *ctx, void *p) {
  if (!p) return;
  const uint64_t t0 = now_us();
  (void)hipFree(p);
  if (ctx) ctx->alloc_us += now_us() - t0;
}
void dev_retire(kmi_ctx *ctx, void *p, size_t bytes) {
  if (!p) return;
  if (bytes >= kCacheMinBlock) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t total = bytes;
    for (const auto &b : g_cache) total += b.bytes;
    if (g_cache.size() < kCacheMaxBlocks && total <= kCacheMaxBytes) { g_cache.push_back({ctx->device, p, bytes}); return; }
  }
  dev_free(ctx, p);
}

kmi_status ws_get(kmi_ctx *ctx, WsSlot slot, size_t bytes, void **out) {
  kmi_ctx::Buf &b = ctx->ws[slot];
  if (bytes == 0) bytes = 256;
  if (b.cap < bytes) {
    if (b.p) { KMI_HIP(ctx, hipStreamSynchronize(ctx->stream)); dev_free(ctx, b.p); b.p = nullptr; b.cap = 0; }
    // a block an index gave back fits (ws_detach hands workspace buffers to indexes; they return through the spare list)
    for (size_t i = 0; i < ctx->spare.size(); ++i)
      if (ctx->spare[i].bytes >= bytes && ctx->spare[i].bytes / 2 <= bytes) {
        b.p = ctx->spare[i].p; b.cap = ctx->spare[i].bytes;
        ctx->spare.erase(ctx->spare.begin() + (long)i);
        *out = b.p;
        return KMI_OK;
      }
    size_t cap = bytes + bytes / 16 + 4096;
    hipError_t e = dev_malloc(ctx, &b.p, cap, &cap);   // (a block another context of this process retired serves as well)
    if (e != hipSuccess) { b.p = nullptr; return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    b.cap = cap;
  }
  *out = b.p;
  return KMI_OK;
}

// the slot's buffer leaves the workspace (its new owner frees it with pool_free(p, *bytes)); false: the slot holds another buffer
bool ws_detach(kmi_ctx *ctx, WsSlot slot, const void *p, size_t *bytes) {
  kmi_ctx::Buf &b = ctx->ws[slot];
  if (!b.p || b.p != p) return false;
  *bytes = b.cap;
  b.p = nullptr; b.cap = 0;
  return true;
}

void ws_release(kmi_ctx *ctx, WsSlot slot) {
  kmi_ctx::Buf &b = ctx->ws[slot];
  if (b.p) { (void)hipStreamSynchronize(ctx->stream); dev_free(ctx, b.p); b.p = nullptr; b.cap = 0; }
}

static hipEvent_t get_event(kmi_ctx *ctx) {
  if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}

void prof_begin(kmi_ctx *ctx, const char *name, uint64_t units) {
  ProfRec r; r.name = name; r.units = units; r.e0 = get_event(ctx); r.e1 = get_event(ctx);
  (void)hipEventRecord(r.e0, ctx->stream);
  ctx->prof_pending.push_back(r);
}

void prof_end(kmi_ctx *ctx, size_t slot) { if (slot < ctx->prof_pending.size()) (void)hipEventRecord(ctx->prof_pending[slot].e1, ctx->stream); }

static void prof_flush(kmi_ctx *ctx) {
  if (ctx->prof_pending.empty()) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (ProfRec &r : ctx->prof_pending) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    bool found = false;
    for (ProfAgg &a : ctx->prof_agg)
      if (strcmp(a.name, r.name) == 0) { a.total_ms += ms; a.launches += 1; a.units += r.units; found = true; break; }
    if (!found) ctx->prof_agg.push_back(ProfAgg{r.name, (double)ms, 1, r.units});
    ctx->event_pool.push_back(r.e0); ctx->event_pool.push_back(r.e1);
  }
  ctx->prof_pending.clear();
}

// ---------------------------------------------------------------------------
// array-level k-mer ops (parity surface)
// ---------------------------------------------------------------------------
enum ArrayOp { OP_REVCOMP = 0, OP_CANONICAL = 1, OP_HASH = 2, OP_RANK = 3 };

template <int NW, int BITS>
__global__ __launch_bounds__(256) void kmer_array_op_kernel(const uint64_t *__restrict__ in, uint64_t n, KShape shape, int op,
                                                            uint32_t which, bool prefix, bool farm_ndebug, uint32_t strand,
                                                            uint32_t dist_trans, uint32_t nranks, uint64_t *__restrict__ out64,
                                                            uint32_t *__restrict__ out32) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t k[NW], r[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) k[w] = in[i * NW + w];
    if (op == OP_REVCOMP) {
      revcomp_words<NW, BITS>(k, r, shape);
#pragma unroll
      for (int w = 0; w < NW; ++w) out64[i * NW + w] = r[w];
    } else if (op == OP_CANONICAL) {
      canonical_words<NW, BITS>(k, r, shape);
#pragma unroll
      for (int w = 0; w < NW; ++w) out64[i * NW + w] = r[w];
    } else if (op == OP_HASH) {
      out64[i] = kmer_hash<NW>(k, shape, which, prefix, farm_ndebug);
    } else {
      // KeyToRank: DistHash(DistTrans(k)) % p ; DistTrans = lex_less for bimolecule, or the single-strand model's choice
      if (strand == KMI_STRAND_BIMOLECULE || dist_trans == KMI_DIST_LEX) { canonical_words<NW, BITS>(k, r, shape);
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = r[w]; }
      else if (dist_trans == KMI_DIST_XOR) { revcomp_words<NW, BITS>(k, r, shape);   // xor_rev_comp (kmer_transform.hpp:60-88)
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] ^= r[w]; }
      const uint64_t h = kmer_hash<NW>(k, shape, which, true, farm_ndebug, ceil_log2_u32(nranks));
      out32[i] = (nranks & (nranks - 1u)) == 0u ? (uint32_t)h & (nranks - 1u) : (uint32_t)(h % nranks);
    }
  }
}

template <int NW, int BITS>
static kmi_status array_op_impl(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, int op, uint32_t which, bool prefix,
                                uint32_t nranks, const uint64_t *in_dev, size_t n, uint64_t *out64, uint32_t *out32) {
  unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
  if (grid == 0) return KMI_OK;
  ProfScope ps(ctx, "kmer_array_op", n);
  hipLaunchKernelGGL((kmer_array_op_kernel<NW, BITS>), dim3(grid), dim3(256), 0, ctx->stream, in_dev, (uint64_t)n, shape, op,
                     which, prefix, cfg->farm_ndebug != 0, cfg->strand, cfg->dist_trans, nranks, out64, out32);
  KMI_HIP(ctx, hipGetLastError());
  return KMI_OK;
}

static kmi_status array_op_dev(kmi_ctx *ctx, const kmi_config *cfg, int op, uint32_t which, bool prefix, uint32_t nranks,
                               const uint64_t *in_dev, size_t n, uint64_t *out64, uint32_t *out32) {
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  KMI_DISPATCH(shape, array_op_impl, ctx, cfg, shape, op, which, prefix, nranks, in_dev, n, out64, out32);
}

static kmi_status array_op_host(kmi_ctx *ctx, const kmi_config *cfg, int op, uint32_t which, bool prefix, uint32_t nranks,
                                const uint64_t *in, size_t n, void *out) {
  if (!ctx) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (n == 0) return KMI_OK;
  if (!in || !out) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  if (op == OP_RANK && nranks == 0) return set_err(ctx, KMI_ERR_INVALID, "nranks must be > 0");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  const size_t in_bytes = n * shape.n_words * sizeof(uint64_t);
  const size_t out_bytes = (op == OP_RANK) ? n * sizeof(uint32_t) : (op == OP_HASH ? n * sizeof(uint64_t) : in_bytes);
  void *din, *dout;
  KMI_TRY(ws_get(ctx, WS_INPUT, in_bytes, &din));
  KMI_TRY(ws_get(ctx, WS_OUTPUT, out_bytes, &dout));
  KMI_HIP(ctx, hipMemcpyAsync(din, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  KMI_TRY(array_op_dev(ctx, cfg, op, which, prefix, nranks, (const uint64_t *)din, n, (uint64_t *)dout, (uint32_t *)dout));
  KMI_HIP(ctx, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

}  // namespace kmi

using namespace kmi;

extern "C" {

kmi_status kmi_ctx_create(int device, int rank, int nranks, void *stream, kmi_ctx **out) {
  if (!out || nranks <= 0 || rank < 0 || rank >= nranks) return KMI_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return KMI_ERR_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return KMI_ERR_DEVICE;
  kmi_ctx *ctx = new kmi_ctx();
  if (const char *fp = getenv("KMI_FUSED_PATH")) ctx->fused_superkmer = strcmp(fp, "kmer") != 0;
  if (const char *dg = getenv("KMI_SK_DBG")) ctx->sk_dbg = atoi(dg);
  if (const char *dc = getenv("KMI_DIST_CHUNKS")) { ctx->dist_chunks = (uint32_t)atoi(dc); if (ctx->dist_chunks < 1) ctx->dist_chunks = 1; if (ctx->dist_chunks > 64) ctx->dist_chunks = 64; }
  if (const char *sl = getenv("KMI_SK_SLACK")) ctx->sk_slack = atoi(sl) != 0;
  if (const char *tp = getenv("KMI_TUPLES")) ctx->tuples_from_parse = strcmp(tp, "extract") != 0;
  if (const char *ps = getenv("KMI_DIST_POOL_SLACK")) ctx->dist_pool_slack = strtoull(ps, nullptr, 10);
  if (const char *pp = getenv("KMI_DIST_POOL_PCT")) { ctx->dist_pool_pct = (uint32_t)atoi(pp); if (ctx->dist_pool_pct < 1) ctx->dist_pool_pct = 1; }
  if (const char *r2 = getenv("KMI_SK_REDUCE")) ctx->sk_reduce2 = atoi(r2) == 2;
  if (const char *ws = getenv("KMI_R2_WIN")) { ctx->sk_r2_win = (uint32_t)atoi(ws); if (ctx->sk_r2_win && ctx->sk_r2_win < 16) ctx->sk_r2_win = 16; if (ctx->sk_r2_win > 256) ctx->sk_r2_win = 256; }
  if (const char *fr = getenv("KMI_FRONT")) ctx->front_fused = strcmp(fr, "general") != 0;
  if (const char *ho = getenv("KMI_HOST_OVERLAP")) ctx->host_overlap = atoi(ho) != 0;
  if (const char *lp2 = getenv("KMI_LINES_P2")) ctx->lines_p2 = atoi(lp2) != 0;
  if (const char *fl = getenv("KMI_SK_FINE_LINES")) ctx->sk_fine_lines = atoi(fl) != 0;
  if (const char *ds = getenv("KMI_DBG_SUPERKMER")) ctx->dbg_superkmer = atoi(ds) != 0;
  if (const char *hm = getenv("KMI_HOST_OVERLAP_MIN")) ctx->host_overlap_min = strtoull(hm, nullptr, 10);
  if (const char *fc = getenv("KMI_FEED_MIN_CHUNK")) { ctx->feed_min_chunk = strtoull(fc, nullptr, 10); if (ctx->feed_min_chunk < 4096) ctx->feed_min_chunk = 4096; }
  if (const char *mr = getenv("KMI_FRONT_MIN_RANGE")) { ctx->front_min_range = strtoull(mr, nullptr, 10); ctx->front_min_range = (ctx->front_min_range + 4095) / 4096 * 4096; if (!ctx->front_min_range) ctx->front_min_range = 4096; }
  if (const char *sm = getenv("KMI_SPARSE_MIN")) ctx->sparse_min = strtoull(sm, nullptr, 10);
  if (const char *fd = getenv("KMI_FORCE_DIST")) ctx->force_dist = atoi(fd) != 0;
  ctx->device = device; ctx->rank = rank; ctx->nranks = nranks; ctx->stream = (hipStream_t)stream;
  if (hipMalloc((void **)&ctx->d_flags, sizeof(uint32_t) * 64) != hipSuccess ||
      hipMalloc((void **)&ctx->d_totals, sizeof(uint64_t) * (16 + 256)) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_totals, sizeof(uint64_t) * (16 + 1024), hipHostMallocDefault)   /* 16 words of totals + a mailbox for the front end's read-back */ != hipSuccess) {
    delete ctx;
    return KMI_ERR_DEVICE;
  }
  (void)hipMemset(ctx->d_flags, 0, sizeof(uint32_t) * 64);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cus = (uint32_t)prop.multiProcessorCount;
  }
  if (upload_quality_lut(ctx) != KMI_OK) { kmi_ctx_destroy(ctx); return KMI_ERR_DEVICE; }
  (void)hipMemset(ctx->d_totals, 0, sizeof(uint64_t) * (16 + 256));
  *out = ctx;
  return KMI_OK;
}

kmi_status kmi_ctx_reset_hints(kmi_ctx *ctx) {
  if (!ctx) return KMI_ERR_INVALID;
  ctx->sk_level_hint = 0; ctx->sk_inv_dup = 0.f;
  return KMI_OK;
}

kmi_status kmi_ctx_debug_counter(const kmi_ctx *ctx, uint32_t which, uint64_t *value) {
  if (!ctx || !value) return KMI_ERR_INVALID;
  switch (which) {
    case 0: *value = ctx->dist_pool_regrows; return KMI_OK;
    case 1: *value = ctx->alloc_us; return KMI_OK;
    case 2: *value = ctx->alloc_bytes; return KMI_OK;
    case 3: *value = ctx->alloc_calls; return KMI_OK;
    case 4: *value = ctx->alloc_reused; return KMI_OK;
    default: return KMI_ERR_INVALID;
  }
}

kmi_status kmi_release_cached_memory(int device, uint64_t *bytes_released) {
  const size_t n = cache_flush(device);
  if (bytes_released) *bytes_released = n;
  return KMI_OK;
}

kmi_status kmi_ctx_destroy(kmi_ctx *ctx) {
  if (!ctx) return KMI_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  // the large blocks wait in the process-wide cache for the next context of this device (kmi_release_cached_memory frees them)
  for (int s = 0; s < WS_NUM_SLOTS; ++s) if (ctx->ws[s].p) dev_retire(ctx, ctx->ws[s].p, ctx->ws[s].cap);
  for (auto &b : ctx->spare) dev_retire(ctx, b.p, b.bytes);
  ctx->spare.clear();
  for (ProfRec &r : ctx->prof_pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->d_flags) (void)hipFree(ctx->d_flags);
  if (ctx->d_totals) (void)hipFree(ctx->d_totals);
  if (ctx->ev_mail) (void)hipEventDestroy(ctx->ev_mail);
  for (hipEvent_t e : ctx->feed_ev) if (e) (void)hipEventDestroy(e);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->h_totals) (void)hipHostFree(ctx->h_totals);
  delete ctx;
  return KMI_OK;
}

const char *kmi_last_error(const kmi_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

kmi_status kmi_ctx_set_fasta_partition(kmi_ctx *ctx, const kmi_fasta_partition *part) {
  if (!ctx) return KMI_ERR_INVALID;
  if (!part) { ctx->fa_part_set = false; return KMI_OK; }
  if (part->start_state > KMI_FA_SEQUENCE || part->index_shift > 1u) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_fasta_partition");
  ctx->fa_part = *part;
  ctx->fa_part_set = true;
  return KMI_OK;
}

kmi_status kmi_kmer_shape(const kmi_config *cfg, uint32_t *n_words, uint32_t *n_bits, uint32_t *n_bytes) {
  KShape s;
  if (!valid_config(cfg, &s)) return KMI_ERR_INVALID;
  if (n_words) *n_words = s.n_words;
  if (n_bits) *n_bits = s.n_bits;
  if (n_bytes) *n_bytes = s.n_bytes;
  return KMI_OK;
}

void kmi_free_host(void *p) { free(p); }

kmi_status kmi_device_alloc(kmi_ctx *ctx, size_t bytes, void **dptr) {
  if (!ctx || !dptr) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 256);
  if (e != hipSuccess) return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
  return KMI_OK;
}

kmi_status kmi_device_free(kmi_ctx *ctx, void *dptr) {
  if (!ctx) return KMI_ERR_INVALID;
  if (dptr) { KMI_HIP(ctx, hipStreamSynchronize(ctx->stream)); KMI_HIP(ctx, hipFree(dptr)); }
  return KMI_OK;
}

kmi_status kmi_copy_to_device(kmi_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
  if (!ctx) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

kmi_status kmi_copy_to_host(kmi_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
  if (!ctx) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

kmi_status kmi_copy_on_device(kmi_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes) {
  if (!ctx) return KMI_ERR_INVALID;
  if (bytes) KMI_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

kmi_status kmi_synchronize(kmi_ctx *ctx) {
  if (!ctx) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

kmi_status kmi_revcomp_host(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *in, size_t n, uint64_t *out) {
  return array_op_host(ctx, cfg, OP_REVCOMP, 0, false, 1, in, n, out);
}
kmi_status kmi_canonical_host(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *in, size_t n, uint64_t *out) {
  return array_op_host(ctx, cfg, OP_CANONICAL, 0, false, 1, in, n, out);
}
kmi_status kmi_hash_host(kmi_ctx *ctx, const kmi_config *cfg, uint32_t which, int prefix, const uint64_t *in, size_t n,
                         uint64_t *out) {
  if (which > 3) return set_err(ctx, KMI_ERR_INVALID, "unknown hash");
  return array_op_host(ctx, cfg, OP_HASH, which, prefix != 0, 1, in, n, out);
}
kmi_status kmi_key_to_rank_host(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *in, size_t n, uint32_t nranks,
                                uint32_t *ranks) {
  if (!cfg) return KMI_ERR_INVALID;
  return array_op_host(ctx, cfg, OP_RANK, cfg->dist_hash, true, nranks, in, n, ranks);
}

// ---- extract
kmi_status kmi_extract_count_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                                 uint64_t *n_tuples, uint64_t *n_seqs) {
  if (!ctx) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  return extract_count(ctx, cfg, bytes_dev, n_bytes, n_tuples, n_seqs);
}

kmi_status kmi_extract_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                           uint64_t file_offset, uint64_t *out_kmers_dev, uint64_t *out_ids_dev, size_t out_capacity,
                           uint64_t *n_tuples, uint64_t *n_seqs) {
  if (!ctx) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  return extract_run(ctx, cfg, bytes_dev, n_bytes, file_offset, out_kmers_dev, out_ids_dev, out_capacity, false, false, n_tuples,
                     n_seqs);
}

kmi_status kmi_extract_records_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset,
                                   uint64_t *out_records_dev, size_t out_capacity, uint64_t *n_tuples, uint64_t *n_seqs) {
  if (!ctx) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (cfg->index_kind == KMI_INDEX_COUNT) return set_err(ctx, KMI_ERR_INVALID, "records are the tuples of the position indexes (index_kind POSITION / POSQUAL)");
  if (cfg->index_kind == KMI_INDEX_POSQUAL && cfg->seq_format != KMI_FMT_FASTQ) return set_err(ctx, KMI_ERR_INVALID, "quality values need FASTQ input");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_tuples) *n_tuples = 0;
  if (n_seqs) *n_seqs = 0;
  if (n_bytes == 0) return KMI_OK;
  const uint32_t rw = shape.n_words + (cfg->index_kind == KMI_INDEX_POSITION ? 1u : 2u);
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  return extract_run(ctx, cfg, bytes_dev, n_bytes, file_offset, out_records_dev, nullptr, out_capacity, false, false, n_tuples, n_seqs, nullptr, rw);
}

kmi_status kmi_extract_host(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes, size_t n_bytes, uint64_t file_offset,
                            kmi_tuples *out) {
  if (!ctx || !out) return KMI_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (n_bytes == 0) return KMI_OK;
  if (!bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *din;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &din));
  KMI_HIP(ctx, hipMemcpyAsync(din, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  uint64_t nt = 0, ns = 0;
  KMI_TRY(extract_count(ctx, cfg, (const uint8_t *)din, n_bytes, &nt, &ns));
  void *dout, *dids = nullptr, *dq = nullptr;
  const size_t out_bytes = (size_t)nt * shape.n_words * sizeof(uint64_t);
  const bool want_ids = cfg->index_kind != KMI_INDEX_COUNT;
  const bool want_quals = cfg->index_kind == KMI_INDEX_POSQUAL;
  KMI_TRY(ws_get(ctx, WS_OUTPUT, out_bytes, &dout));
  if (want_ids) KMI_TRY(ws_get(ctx, WS_OUTPUT2, (size_t)nt * sizeof(uint64_t), &dids));
  if (want_quals) KMI_TRY(ws_get(ctx, WS_INPUT2, (size_t)nt * sizeof(float) + 16, &dq));
  KMI_TRY(extract_run(ctx, cfg, (const uint8_t *)din, n_bytes, file_offset, (uint64_t *)dout, (uint64_t *)dids, (size_t)nt, false, true,
                      &nt, &ns, (float *)dq));
  out->n_tuples = nt; out->n_seqs = ns;
  out->kmers = (uint64_t *)malloc(out_bytes ? out_bytes : 8);
  if (want_ids) out->ids = (uint64_t *)malloc(nt ? nt * sizeof(uint64_t) : 8);
  if (want_quals) out->quals = (float *)malloc(nt ? nt * sizeof(float) : 8);
  if (!out->kmers || (want_ids && !out->ids) || (want_quals && !out->quals)) return set_err(ctx, KMI_ERR_NOMEM, "host malloc failed");
  if (out_bytes) KMI_HIP(ctx, hipMemcpyAsync(out->kmers, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (want_ids && nt) KMI_HIP(ctx, hipMemcpyAsync(out->ids, dids, nt * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (want_quals && nt) KMI_HIP(ctx, hipMemcpyAsync(out->quals, dq, nt * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

// read_file_* of ONE rank of several (kmer_file_helper.hpp:550-579 over partitioned_file, file.hpp:1216-1430), FASTQ: the rank
// read file bytes [buffer_offset, buffer_offset + n_bytes) = its nominal range of nominal_bytes plus look-ahead; its partition
// runs from the first record start at or after the buffer's first byte to the first one at or after the nominal end
// (kmi_fastq_find_records_dev). *need_more = 1, nothing parsed: the end is not decidable inside the buffer -- read further.
kmi_status kmi_extract_range_host(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                  uint64_t nominal_bytes, int reaches_eof, int *need_more, kmi_tuples *out) {
  if (!ctx || !out || !need_more) return KMI_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  *need_more = 0;
  if (cfg->seq_format != KMI_FMT_FASTQ) return set_err(ctx, KMI_ERR_INVALID, "a byte range of a file is cut at FASTQ record starts here");
  if (n_bytes == 0) return KMI_OK;
  if (!bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (nominal_bytes > n_bytes) nominal_bytes = n_bytes;
  void *din;
  KMI_TRY(ws_get(ctx, WS_INPUT2, n_bytes + 64, &din));
  KMI_HIP(ctx, hipMemcpyAsync(din, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  uint64_t pos[2] = {0, nominal_bytes}, cut[2] = {0, n_bytes};
  KMI_TRY(kmi_fastq_find_records_dev(ctx, (const uint8_t *)din, n_bytes, buffer_offset == 0, pos, 2, cut));
  if (nominal_bytes >= n_bytes) cut[1] = n_bytes;
  if (!reaches_eof && (cut[1] >= n_bytes || (cut[0] >= n_bytes && buffer_offset != 0))) { *need_more = 1; return KMI_OK; }
  if (cut[1] < cut[0]) cut[1] = cut[0];
  return kmi_extract_host(ctx, cfg, bytes + cut[0], (size_t)(cut[1] - cut[0]), buffer_offset + cut[0], out);
}

kmi_status kmi_extract_fasta_block_host(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes, size_t n_bytes, uint32_t rank, uint32_t nranks,
                                        kmi_tuples *out) {
  // the tuples of block `rank` of an equal nranks-way split of a FASTA buffer every rank holds whole (see
  // kmi_index_build_fasta_file_dist_host): the union over the ranks is the file's tuples, each exactly once
  if (!ctx || !cfg || !out || nranks == 0 || rank >= nranks) return KMI_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  if (cfg->seq_format != KMI_FMT_FASTA) return set_err(ctx, KMI_ERR_INVALID, "not FASTA");
  if (n_bytes == 0) return KMI_OK;
  if (!bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *din;
  KMI_TRY(ws_get(ctx, WS_INPUT2, n_bytes + 64, &din));
  KMI_HIP(ctx, hipMemcpyAsync(din, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint64_t> be(2 * (size_t)nranks);
  std::vector<kmi_fasta_partition> parts(nranks);
  KMI_TRY(kmi_fasta_partition_dev(ctx, (const uint8_t *)din, n_bytes, nranks, cfg->k, be.data(), parts.data()));
  KMI_TRY(kmi_ctx_set_fasta_partition(ctx, &parts[rank]));
  const kmi_status st = kmi_extract_host(ctx, cfg, bytes + be[2 * rank], (size_t)(be[2 * rank + 1] - be[2 * rank]), be[2 * rank], out);
  (void)kmi_ctx_set_fasta_partition(ctx, nullptr);
  return st;
}

void kmi_tuples_free(kmi_tuples *t) {
  if (!t) return;
  free(t->kmers); free(t->ids); free(t->quals);
  memset(t, 0, sizeof(*t));
}

// ---- profiling
kmi_status kmi_profile_enable(kmi_ctx *ctx, int on) { if (!ctx) return KMI_ERR_INVALID; prof_flush(ctx); ctx->prof = on != 0; return KMI_OK; }
kmi_status kmi_profile_reset(kmi_ctx *ctx) { if (!ctx) return KMI_ERR_INVALID; prof_flush(ctx); ctx->prof_agg.clear(); return KMI_OK; }
kmi_status kmi_profile_get(kmi_ctx *ctx, kmi_kernel_time *out, size_t cap, size_t *n) {
  if (!ctx || !n) return KMI_ERR_INVALID;
  prof_flush(ctx);
  size_t m = std::min(cap, ctx->prof_agg.size());
  for (size_t i = 0; i < m; ++i) {
    out[i].name = ctx->prof_agg[i].name; out[i].total_ms = ctx->prof_agg[i].total_ms;
    out[i].launches = ctx->prof_agg[i].launches; out[i].units = ctx->prof_agg[i].units;
  }
  *n = ctx->prof_agg.size();
  return KMI_OK;
}

}  // extern "C"
