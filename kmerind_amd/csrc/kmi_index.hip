// kmi_index.hip -- the map behind Index<MapType,Parser>: insert / count / find / erase.
//
// Reference path replaced (one rank):
//   counting_unordered_map::insert(vector<Key>)   distributed_unordered_map.hpp:1826-1884
//   reduction_unordered_map::local_insert         :1603-1618   (count[key] += v)
//   unordered_map_base::count / find / erase      :880-983, :564-687, :719-779
//   fsc::unique (query dedup)                     fsc_container_utils.hpp:306-320
//   imxx::distribute bucketing                    incremental_mxx.hpp:273-364,595-640 (kmi_route_dev)
//
// GPU formulation. A node-based hash table with one malloc per key is the wrong shape for
// HBM; the map is a two-level hash partition followed by an LDS-resident reduce:
//   place_hash(key) (32 bit)  -> coarse bucket = top 8 bits, fine bucket = top 15 bits
//   K1  histogram   : per-workgroup LDS histogram of the 32768 fine buckets (privatised,
//                     flushed once with coalesced atomics) + per-workgroup coarse counts
//   K2  scatter     : keys -> 256 coarse buckets. Each tile is bucket-sorted in LDS so a
//                     bucket's keys leave as one contiguous run (coalesced stores); output
//                     ranges come from the per-workgroup counts, so there are no global
//                     atomics and no collisions between workgroups.
//   P2  scatter     : one workgroup per coarse bucket splits it into its 128 fine buckets.
//   C   reduce      : one workgroup per fine bucket builds an open-addressing table in LDS
//                     (key -> count), merging the bucket of the existing index, and emits the
//                     distinct (key,count) pairs. Buckets whose distinct keys exceed the LDS
//                     table are reduced in several passes over disjoint key subsets.
//   queries use the same machinery: partition the query keys, build the LDS table from the
//   (deduplicated) queries of a bucket, stream the index bucket against it.
// The stored index is (bucket_off[32769], keys[], counts[]): entries of a fine bucket are
// contiguous; order inside a bucket is unspecified, like the reference's unordered_map.
#include <stdlib.h>

#include <algorithm>

#include "kmi_extract.h"
#include "kmi_minimizer.h"
#include <type_traits>

namespace kmi {

constexpr int kFineBits = 15;
constexpr int kNumFine = 1 << kFineBits;     // 32768 fine buckets
constexpr int kCoarseBits = 8;
constexpr int kNumCoarse = 1 << kCoarseBits; // 256
constexpr int kSubPerCoarse = kNumFine / kNumCoarse;  // 128
constexpr int kSlotBits = 32 - kFineBits;    // low 17 bits pick the LDS slot
constexpr int kPartThreads = 1024;           // K1/K2/P2 workgroup size
constexpr int kPartGroups = 512;             // K1/K2 workgroups (two per CU)
constexpr int kFineParts = 2;                // P2 workgroups per coarse bucket; part h = K2 groups [h*256, (h+1)*256)
// workgroups of the list-driven histogram and scatter passes (they own the same tiles)
// workgroups of the fused passes (E1 and E2 cut the scan tiles the same way)
template <int NW> constexpr int fused_groups() { return kPartGroups; }
constexpr int kListGroups = 2048;            // workgroups of the window-list pass (small LDS footprint: eight per CU)
constexpr int kLoadBatch = 8;                // independent key loads kept in flight per thread

__host__ __device__ inline uint32_t fine_of(uint32_t h) { return h >> kSlotBits; }
__host__ __device__ inline uint32_t coarse_of(uint32_t h) { return h >> (32 - kCoarseBits); }

// LDS tile of the partition kernels by record width RW (key words + value words): 64 KB of records
template <int RW> struct PartCfg {
  static constexpr int TILE = (RW == 1) ? 8192 : (RW == 2 ? 4096 : (RW <= 4 ? 2048 : 1024));   // records per tile
  static constexpr int CTILE = (RW == 3) ? 3072 : TILE;   // ... of the passes over input chunks (K1 / K2): 24-byte records over 256 buckets leave in runs of 8 with 2048, 72 KB still fits twice per CU (scatter_coarse 16.0 -> 13.7 ms per 1.2e9; P2 is slower with it)
  static constexpr int PER_THREAD = TILE / kPartThreads;                                         // 8, 4, 2, 1
};

// BUCKET_REC: super-k-mer records (kmi_superkmer.h): the fine bucket sits in bits 46..52 of the record's second word
// BUCKET_REC_COARSE: its coarse bucket, bits 53..60 (records that arrived from other ranks are sorted by it first)
enum BucketMode { BUCKET_COARSE = 0, BUCKET_SUB = 1, BUCKET_RANK = 2, BUCKET_REC = 3, BUCKET_REC_COARSE = 4 };

struct BucketFn {
  int mode; KShape shape; uint32_t dist_hash; bool farm_ndebug; uint32_t nranks;
  uint32_t sub;   // rank mode: sub-buckets per rank (power of two, nranks * sub <= 256)
  uint32_t dist_trans = 0;   // rank mode: KMI_DIST_* applied to the key before DistHash (single-strand model only)
  uint32_t layout_w = 0;     // coarse / sub modes: 0 = the index is laid out by placement hash; W = by minimizer bucket (fine15_of_key)
  uint64_t rank_magic = 0;   // rank mode: floor(2^64 / nranks) for rank counts that are no power of two (rank_of_hash)
  uint32_t owner_w = 0;      // rank mode: W = the rank is the owner of the key's minimizer bucket (the top log2(nranks) bucket bits:
                             // where a build through exchanged super-k-mers put it), not KeyToRank's hash
};
inline uint64_t rank_magic_of(uint32_t nranks) { return nranks > 1u ? (uint64_t)(((unsigned __int128)1 << 64) / nranks) : 0ull; }
// h % p without a division: a mask for 2, 4, 8 ... ranks; else q = mulhi(h, floor(2^64 / p)) is the quotient or one
// less, so one correction step gives the remainder (a 64-bit division by a run-time value is a hundred-odd instructions)
__host__ __device__ inline uint32_t rank_of_hash(uint64_t h, uint32_t nranks, uint64_t magic) {
  if ((nranks & (nranks - 1u)) == 0u) return (uint32_t)h & (nranks - 1u);   // uniform
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t q = __umul64hi(h, magic);
#else
  const uint64_t q = (uint64_t)(((unsigned __int128)h * magic) >> 64);
#endif
  uint64_t r = h - q * nranks;
  r = r >= nranks ? r - nranks : r;
  return (uint32_t)r;
}
// Rank mode spreads every rank over `sub` buckets (bucket = rank * sub + a few high hash bits): the buckets of a rank
// stay adjacent, so the output is still grouped by rank, but the per-tile LDS counters are 256 distinct addresses
// instead of p hot ones (with p = 2..8 the same-address LDS atomics were the bottleneck of both rank kernels).
inline uint32_t rank_sub_buckets(uint32_t nranks) { uint32_t s = 1; while (s * 2u * nranks <= (uint32_t)kNumCoarse) s *= 2u; return s; }

// The 15-bit fine bucket of a stored key. An index is laid out either by the placement hash (layout_w = 0: everything that
// arrives as k-mers) or by minimizer bucket (layout_w = W: what the super-k-mer build leaves, one-word 2-bit k-mers only);
// kmi_index::layout_w says which, and every partition of keys for that index uses the same function.
template <int NW> __device__ __forceinline__ uint32_t fine15_of_key(const uint64_t (&key)[NW], uint32_t layout_w, uint32_t k) {
  if constexpr (NW == 1) {
    if (layout_w) {   // uniform. layout_w = W | lp << 8: an index built over 2^lp ranks dropped the lp top bucket bits (its rank)
      const uint32_t lp = layout_w >> 8;
      return ((sk_key_bucket18(key[0], k, layout_w & 0xffu) << lp) & 0x3ffffu) >> 3;
    }
  }
  return fine_of(place_hash<NW>(key));
}

template <int NW> __device__ __forceinline__ uint32_t bucket_of(const uint64_t (&key)[NW], const BucketFn &f) {
  if (f.mode == BUCKET_RANK) {
    if constexpr (NW == 1) {
      if (f.owner_w) {   // uniform (nranks is a power of two here)
        const uint32_t h18 = sk_key_bucket18(key[0], f.shape.k, f.owner_w);
        const uint32_t lp = 31u - (uint32_t)__builtin_clz(f.nranks);
        return (h18 >> (18u - lp)) * f.sub + ((h18 >> 3) & (f.sub - 1u));
      }
    }
    uint64_t t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = key[w];
    if (f.dist_trans) {   // uniform: DistTrans = lex_less / xor_rev_comp (kmer_transform.hpp:90-116, 60-88)
      uint64_t rc[NW];
      if (f.shape.bits == 2) revcomp_words<NW, 2>(key, rc, f.shape); else revcomp_words<NW, 3>(key, rc, f.shape);   // 3 and 4 bits: plain bit reversal
      const bool use_rc = f.dist_trans == KMI_DIST_LEX && less_words<NW>(rc, key);
#pragma unroll
      for (int w = 0; w < NW; ++w) t[w] = f.dist_trans == KMI_DIST_XOR ? (key[w] ^ rc[w]) : (use_rc ? rc[w] : key[w]);
    }
    const uint64_t h = kmer_hash<NW>(t, f.shape, f.dist_hash, true, f.farm_ndebug, ceil_log2_u32(f.nranks));
    const uint32_t spread = (uint32_t)((h >> 40) ^ (h >> 13)) * 0x9E3779B1u;   // identity/std hashes have few high bits
    const uint32_t rank = rank_of_hash(h, f.nranks, f.rank_magic);
    return rank * f.sub + ((spread >> 16) & (f.sub - 1u));
  }
  const uint32_t fb = fine15_of_key<NW>(key, f.layout_w, f.shape.k);
  return f.mode == BUCKET_COARSE ? (fb >> (kFineBits - kCoarseBits)) : (fb & (kSubPerCoarse - 1));
}

template <int NW, int BITS> __device__ __forceinline__ void load_key(const uint64_t *__restrict__ keys, uint64_t i, const KShape &s,
                                                                    uint32_t strand, bool transform, uint64_t (&k)[NW]) {
  uint64_t raw[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) raw[w] = keys[i * NW + w];
  if (transform) strand_key<NW, BITS>(raw, k, s, strand);
  else {
#pragma unroll
    for (int w = 0; w < NW; ++w) k[w] = raw[w];
  }
}

// chunk of keys owned by workgroup w of `groups` (multiple of the tile size)
__host__ __device__ inline uint64_t part_chunk(uint64_t n, uint32_t groups, uint32_t tile) {
  uint64_t c = (n + groups - 1) / groups;
  return (c + tile - 1) / tile * tile;
}

// ---------------------------------------------------------------------------
// K1: histograms
// ---------------------------------------------------------------------------
template <int NW, int BITS, int VW = 0>
__global__ __launch_bounds__(kPartThreads) void hist_fine_kernel(const uint64_t *__restrict__ keys, uint64_t n, KShape shape,
                                                                uint32_t strand, bool transform,
                                                                uint32_t *__restrict__ fine_hist,     // [kFineParts][kNumFine] global
                                                                uint32_t *__restrict__ wg_hist,       // [groups][256]
                                                                uint32_t layout_w = 0,
                                                                uint32_t in_rw = NW + VW /* words per input record (scatter_range: in_q) */) {
  __shared__ uint32_t s_hist[kNumFine];
  for (int i = threadIdx.x; i < kNumFine; i += kPartThreads) s_hist[i] = 0;
  lds_barrier();
  const uint64_t chunk = part_chunk(n, gridDim.x, PartCfg<NW + VW>::CTILE);
  const uint64_t b = (uint64_t)blockIdx.x * chunk;
  const uint64_t e = (b + chunk < n) ? b + chunk : n;
  constexpr int U = (NW == 1) ? kLoadBatch : (NW == 2 ? 4 : 2);
  for (uint64_t i0 = b; i0 < e; i0 += (uint64_t)kPartThreads * U) {
    uint64_t raw[U][NW];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t i = i0 + (uint64_t)u * kPartThreads + threadIdx.x;
      ok[u] = i < e;
      if (ok[u]) {
#pragma unroll
        for (int w = 0; w < NW; ++w) raw[u][w] = keys[i * in_rw + w];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) {
        uint64_t k[NW];
        if (transform) strand_key<NW, BITS>(raw[u], k, shape, strand);
        else {
#pragma unroll
          for (int w = 0; w < NW; ++w) k[w] = raw[u][w];
        }
        atomicAdd(&s_hist[fine15_of_key<NW>(k, layout_w, shape.k)], 1u);
      }
    }
  }
  lds_barrier();
  uint32_t *part_hist = fine_hist + (uint64_t)(blockIdx.x / (gridDim.x / kFineParts)) * kNumFine;
  for (int i = threadIdx.x; i < kNumFine; i += kPartThreads) {
    uint32_t v = s_hist[i];
    if (v) atomicAdd(&part_hist[i], v);
  }
  if (threadIdx.x < kNumCoarse) {
    uint32_t s = 0;
    for (int i = 0; i < kSubPerCoarse; ++i) s += s_hist[threadIdx.x * kSubPerCoarse + ((i + threadIdx.x) & (kSubPerCoarse - 1))];
    wg_hist[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] = s;
  }
}

template <int NW, int BITS, int VW = 0>
__global__ __launch_bounds__(kPartThreads) void hist_rank_kernel(const uint64_t *__restrict__ keys, uint64_t n, KShape shape,
                                                                uint32_t strand, BucketFn fn, uint32_t *__restrict__ wg_hist,
                                                                uint32_t in_rw = NW + VW /* words per input record (scatter_range: in_q) */) {
  __shared__ uint32_t s_hist[kNumCoarse];
  if (threadIdx.x < kNumCoarse) s_hist[threadIdx.x] = 0;
  lds_barrier();
  const uint64_t chunk = part_chunk(n, gridDim.x, PartCfg<NW + VW>::CTILE);
  const uint64_t b = (uint64_t)blockIdx.x * chunk;
  const uint64_t e = (b + chunk < n) ? b + chunk : n;
  for (uint64_t i = b + threadIdx.x; i < e; i += kPartThreads) {
    uint64_t raw[NW], k[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) raw[w] = keys[i * in_rw + w];
    strand_key<NW, BITS>(raw, k, shape, strand);
    atomicAdd(&s_hist[bucket_of<NW>(k, fn)], 1u);
  }
  lds_barrier();
  if (threadIdx.x < kNumCoarse) wg_hist[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] = s_hist[threadIdx.x];
}

// ---------------------------------------------------------------------------
// offsets: fine_off = exclusive scan of fine_hist (u64, kNumFine+1 entries), part_off[h][f] = where part h of fine bucket f
// starts, coarse_base[c] = fine_off[c * 128]
// ---------------------------------------------------------------------------
// Two small launches spread over the chip (a one-workgroup scan read the 64 K counters with a stride of 128 bytes between
// lanes: 0.1 ms per call, twice per build): one workgroup per coarse bucket, one thread per fine bucket of it. (a) totals of the coarse buckets; (b) every workgroup scans the 256 totals for its base, then its own 128 x
// kFineParts counters.
__global__ __launch_bounds__(kSubPerCoarse) void fine_totals_kernel(const uint32_t *__restrict__ fine_hist, uint64_t *__restrict__ coarse_tot) {
  __shared__ uint64_t s_scan[kSubPerCoarse / 64 + 2];
  const uint32_t f = blockIdx.x * kSubPerCoarse + threadIdx.x;
  uint64_t v = 0;
#pragma unroll
  for (int h = 0; h < kFineParts; ++h) v += fine_hist[(uint64_t)h * kNumFine + f];
  uint64_t total;
  (void)block_exclusive_scan<uint64_t>(v, s_scan, &total);
  if (threadIdx.x == 0) coarse_tot[blockIdx.x] = total;
}
__global__ __launch_bounds__(kSubPerCoarse) void fine_offsets_spread_kernel(const uint32_t *__restrict__ fine_hist, const uint64_t *__restrict__ coarse_tot,
                                                                           uint64_t *__restrict__ fine_off, uint64_t *__restrict__ part_off,
                                                                           uint64_t *__restrict__ coarse_base) {
  static_assert(kNumCoarse == 2 * kSubPerCoarse, "two coarse totals per thread");
  __shared__ uint64_t s_scan[kSubPerCoarse / 64 + 2];
  const uint32_t c = blockIdx.x, f = c * kSubPerCoarse + threadIdx.x;
  // base of this coarse bucket = the totals below it (two per thread)
  const uint32_t c0 = 2u * threadIdx.x;
  const uint64_t t0 = coarse_tot[c0], t1 = coarse_tot[c0 + 1];
  uint64_t grand;
  (void)block_exclusive_scan<uint64_t>(t0 + t1, s_scan, &grand);
  const uint64_t below = (c0 < c ? t0 : 0ull) + (c0 + 1 < c ? t1 : 0ull);
  uint64_t base;
  (void)block_exclusive_scan<uint64_t>(below, s_scan, &base);   // (its total: the sum over all threads)
  uint32_t loc[kFineParts];
  uint64_t v = 0;
#pragma unroll
  for (int h = 0; h < kFineParts; ++h) { loc[h] = fine_hist[(uint64_t)h * kNumFine + f]; v += loc[h]; }
  uint64_t off = base + block_exclusive_scan<uint64_t>(v, s_scan, (uint64_t *)nullptr);
  if (threadIdx.x == 0) coarse_base[c] = base;
  fine_off[f] = off;
#pragma unroll
  for (int h = 0; h < kFineParts; ++h) {
    if (part_off) part_off[(uint64_t)h * kNumFine + f] = off;
    off += loc[h];
  }
  if (c == (uint32_t)kNumCoarse - 1 && threadIdx.x == 0) fine_off[kNumFine] = grand;
}
static void launch_fine_offsets(kmi_ctx *ctx, const uint32_t *fine_hist, uint64_t *fine_off, uint64_t *part_off, uint64_t *coarse_base) {
  uint64_t *tot = ctx->d_totals + 16;
  hipLaunchKernelGGL(fine_totals_kernel, dim3(kNumCoarse), dim3(kSubPerCoarse), 0, ctx->stream, fine_hist, tot);
  hipLaunchKernelGGL(fine_offsets_spread_kernel, dim3(kNumCoarse), dim3(kSubPerCoarse), 0, ctx->stream, fine_hist, (const uint64_t *)tot, fine_off, part_off,
                     coarse_base);
}

// wg_off[w][c] = coarse_base[c] + sum_{w' < w} wg_hist[w'][c]: one wavefront per coarse bucket, spread over the chip
__global__ __launch_bounds__(256) void coarse_cursors_kernel(const uint32_t *__restrict__ wg_hist, uint32_t groups,
                                                            const uint64_t *__restrict__ coarse_base, uint64_t *__restrict__ wg_off) {
  const uint32_t c = blockIdx.x * (blockDim.x >> 6) + wave_id();
  if (c >= (uint32_t)kNumCoarse) return;
  uint64_t carry = coarse_base[c];
  for (uint32_t w0 = 0; w0 < groups; w0 += kWave) {
    const uint32_t w = w0 + lane_id();
    const uint64_t v = (w < groups) ? wg_hist[(uint64_t)w * kNumCoarse + c] : 0ull;
    const uint64_t inc = wave_inclusive_scan(v);
    if (w < groups) wg_off[(uint64_t)w * kNumCoarse + c] = carry + inc - v;
    carry += __shfl(inc, kWave - 1, kWave);
  }
}

// rank mode: bucket_cnt[b] and wg_off[w][b] for b < nbuckets, in three small launches spread over the chip:
// column sums (one wavefront per bucket), scan of the <= 256 sums, per-(workgroup, bucket) cursors
__global__ __launch_bounds__(256) void rank_totals_kernel(const uint32_t *__restrict__ wg_hist, uint32_t groups, uint32_t nbuckets,
                                                         uint64_t *__restrict__ bucket_cnt) {
  const uint32_t c = blockIdx.x * (blockDim.x >> 6) + wave_id();
  if (c >= nbuckets) return;
  uint64_t v = 0;
  for (uint32_t w = lane_id(); w < groups; w += kWave) v += wg_hist[(uint64_t)w * kNumCoarse + c];
  v = wave_reduce_sum(v);
  if (lane_id() == 0) bucket_cnt[c] = v;
}
__global__ __launch_bounds__(256) void rank_bases_kernel(const uint64_t *__restrict__ bucket_cnt, uint32_t nbuckets, uint64_t *__restrict__ base) {
  __shared__ uint64_t s_scan[256 / 64 + 2];
  const uint64_t tot = (threadIdx.x < nbuckets) ? bucket_cnt[threadIdx.x] : 0ull;
  uint64_t total;
  const uint64_t off = block_exclusive_scan<uint64_t>(tot, s_scan, &total);
  base[threadIdx.x] = off;   // entries >= nbuckets hold the grand total: harmless, never consumed as a cursor
}
static void launch_rank_offsets(hipStream_t stream, const uint32_t *wg_hist, uint32_t groups, uint32_t nbuckets, uint64_t *bucket_cnt,
                                uint64_t *base /* [kNumCoarse] */, uint64_t *wg_off) {
  hipLaunchKernelGGL(rank_totals_kernel, dim3(kNumCoarse / 4), dim3(256), 0, stream, wg_hist, groups, nbuckets, bucket_cnt);
  hipLaunchKernelGGL(rank_bases_kernel, dim3(1), dim3(256), 0, stream, (const uint64_t *)bucket_cnt, nbuckets, base);
  hipLaunchKernelGGL(coarse_cursors_kernel, dim3(kNumCoarse / 4), dim3(256), 0, stream, wg_hist, groups, (const uint64_t *)base, wg_off);
}

// ---------------------------------------------------------------------------
// K2 / P2: tile-wise bucket sort in LDS + contiguous runs out.
// Per tile: (S0) hash + rank by LDS atomic, (S1) 256-counter scan by the first four waves,
// (S2) publish local offsets and global bases, (S3) keys into the bucket-sorted stage,
// (S4) contiguous copy-out. Four barriers per tile; the running output cursor of bucket b lives
// in a register of thread b; the next tile's keys are already in flight during S1-S4.
// ---------------------------------------------------------------------------
template <int NW, int BITS, int VW = 0, int TILE_ = PartCfg<NW + VW>::TILE>
__device__ __forceinline__ void scatter_range(const uint64_t *__restrict__ in, uint64_t begin, uint64_t end, uint64_t *__restrict__ out,
                                              const KShape &shape, uint32_t strand, bool transform, const BucketFn &fn,
                                              uint64_t cursor /* of bucket threadIdx.x, threads < 256 */,
                                              uint64_t *s_stage, uint8_t *s_bkt, uint32_t *s_cnt, uint32_t *s_lofs,
                                              uint64_t *s_gbase, uint32_t *s_part, uint64_t *__restrict__ out_vals = nullptr,
                                              const float *__restrict__ in_q = nullptr, const uint64_t *__restrict__ in_v = nullptr) {
  // in_q (records only): the input records are one word short and their last value word is the float in_q[i] (its bits in the
  // low half) -- the quality values of a position + quality build, which leave their kernel as one dense array
  // in_v (records only): the input is split -- `in` holds the key words alone, in_v[i] the first value word of record i (and
  // in_q[i], or zero without in_q, the second): the arrays the tuple parsers write, so the histogram pass reads keys only
  // out_vals (records only): the key words go to out[dst * NW ..], the value words to out_vals[dst * VW ..] -- the arrays of a
  // multimap index -- instead of whole records to out[dst * RW ..]
  constexpr int RW = NW + VW;   // record = key words followed by value words
  constexpr int TILE = TILE_;
  constexpr int PT = TILE_ / kPartThreads;
  if (threadIdx.x < kNumCoarse) s_cnt[threadIdx.x] = 0;
  lds_barrier();
  uint64_t raw[PT][RW];
  auto load_tile = [&](uint64_t t0) {   // unconditional (clamped) loads: nothing forces an early wait
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      uint64_t i = t0 + (uint64_t)j * kPartThreads + threadIdx.x;
      i = (i < end) ? i : end - 1;
      if (VW > 0 && in_v) {   // uniform
#pragma unroll
        for (int w = 0; w < NW; ++w) raw[j][w] = in[i * NW + w];
        raw[j][NW] = in_v[i];
        if (VW > 1) raw[j][RW - 1] = in_q ? (uint64_t)__float_as_uint(in_q[i]) : 0ull;
      } else if (VW > 0 && in_q) {   // uniform
#pragma unroll
        for (int w = 0; w < RW - 1; ++w) raw[j][w] = in[i * (RW - 1) + w];
        raw[j][RW - 1] = (uint64_t)__float_as_uint(in_q[i]);
      } else {
#pragma unroll
        for (int w = 0; w < RW; ++w) raw[j][w] = in[i * RW + w];
      }
    }
  };
  load_tile(begin);
  for (uint64_t t0 = begin; t0 < end; t0 += TILE) {
    const uint32_t nt = (uint32_t)((end - t0 < (uint64_t)TILE) ? (end - t0) : (uint64_t)TILE);
    uint64_t k[PT][NW];
    uint64_t v[PT][VW ? VW : 1];
    uint32_t bk[PT], rk[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const uint32_t li = j * kPartThreads + threadIdx.x;
      bk[j] = 0xffffffffu;
      if (li < nt) {
        uint64_t kr[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) kr[w] = raw[j][w];
#pragma unroll
        for (int w = 0; w < VW; ++w) v[j][w] = raw[j][NW + w];
        if (transform) strand_key<NW, BITS>(kr, k[j], shape, strand);
        else {
#pragma unroll
          for (int w = 0; w < NW; ++w) k[j][w] = kr[w];
        }
        if (VW > 0 && fn.mode == BUCKET_REC) bk[j] = (uint32_t)(v[j][0] >> 46) & (uint32_t)(kSubPerCoarse - 1);
        else if (VW > 0 && fn.mode == BUCKET_REC_COARSE) bk[j] = (uint32_t)(v[j][0] >> 53) & (uint32_t)(kNumCoarse - 1);
        else bk[j] = bucket_of<NW>(k[j], fn);
        rk[j] = atomicAdd(&s_cnt[bk[j]], 1u);
      }
    }
    if (t0 + TILE < end) load_tile(t0 + TILE);   // in flight until the next iteration needs it
    lds_barrier();
    uint32_t c = 0, inc = 0;
    if (threadIdx.x < kNumCoarse) {              // waves 0..3, whole waves
      c = s_cnt[threadIdx.x];
      s_cnt[threadIdx.x] = 0;
      inc = wave_inclusive_scan(c);
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
    }
    lds_barrier();
    if (threadIdx.x < kNumCoarse) {
      uint32_t pre = 0;
#pragma unroll
      for (uint32_t w = 0; w < kNumCoarse / kWave; ++w) pre += (w < wave_id()) ? s_part[w] : 0u;
      const uint32_t lo = pre + inc - c;
      s_lofs[threadIdx.x] = lo;
      s_gbase[threadIdx.x] = cursor - lo;
      cursor += c;
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      if (bk[j] != 0xffffffffu) {
        const uint32_t pos = s_lofs[bk[j]] + rk[j];
#pragma unroll
        for (int w = 0; w < NW; ++w) s_stage[(uint64_t)pos * RW + w] = k[j][w];
#pragma unroll
        for (int w = 0; w < VW; ++w) s_stage[(uint64_t)pos * RW + NW + w] = v[j][w];
        s_bkt[pos] = (uint8_t)bk[j];
      }
    }
    lds_barrier();
    if (VW > 0 && out_vals) {   // uniform
      for (uint32_t s = threadIdx.x; s < nt; s += kPartThreads) {
        const uint64_t dst = s_gbase[s_bkt[s]] + s;
#pragma unroll
        for (int w = 0; w < NW; ++w) out[dst * NW + w] = s_stage[(uint64_t)s * RW + w];
#pragma unroll
        for (int w = 0; w < VW; ++w) out_vals[dst * (VW ? VW : 1) + w] = s_stage[(uint64_t)s * RW + NW + w];
      }
    } else {
      for (uint32_t s = threadIdx.x; s < nt; s += kPartThreads) {
        const uint64_t dst = s_gbase[s_bkt[s]] + s;
#pragma unroll
        for (int w = 0; w < RW; ++w) out[dst * RW + w] = s_stage[(uint64_t)s * RW + w];
      }
    }
    // no barrier here: the next tile's S0 only touches s_cnt (reset in S1 above) and its S2/S3
    // writes are separated from this copy-out by the two barriers in between
  }
}

#define KMI_SCATTER_LDS(RW, TL)                                    \
  __shared__ uint64_t s_stage[(TL) * (RW)];                        \
  __shared__ uint8_t s_bkt[TL];                                    \
  __shared__ uint32_t s_cnt[kNumCoarse];                           \
  __shared__ uint32_t s_lofs[kNumCoarse];                          \
  __shared__ uint64_t s_gbase[kNumCoarse];                         \
  __shared__ uint32_t s_part[kNumCoarse / kWave];

// K2: workgroup w scatters its chunk of the input by coarse bucket (or by rank)
template <int NW, int BITS, int VW = 0>
__global__ __launch_bounds__(kPartThreads) void scatter_chunks_kernel(const uint64_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out,
                                                                     KShape shape, uint32_t strand, bool transform, BucketFn fn,
                                                                     const uint64_t *__restrict__ wg_off, const float *__restrict__ in_q = nullptr,
                                                                     const uint64_t *__restrict__ in_v = nullptr) {
  KMI_SCATTER_LDS(NW + VW, (PartCfg<NW + VW>::CTILE))
  const uint64_t cursor = (threadIdx.x < kNumCoarse) ? wg_off[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] : 0ull;
  const uint64_t chunk = part_chunk(n, gridDim.x, PartCfg<NW + VW>::CTILE);
  const uint64_t b = (uint64_t)blockIdx.x * chunk;
  const uint64_t e = (b + chunk < n) ? b + chunk : n;
  if (b < e) scatter_range<NW, BITS, VW, PartCfg<NW + VW>::CTILE>(in, b, e, out, shape, strand, transform, fn, cursor, s_stage, s_bkt, s_cnt, s_lofs, s_gbase, s_part, nullptr, in_q, in_v);
}

// P2: workgroup (c, h) splits the part of coarse bucket c that K2 groups [h*256,(h+1)*256) wrote
template <int NW, int BITS, int VW = 0>
__global__ __launch_bounds__(kPartThreads) void scatter_fine_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, KShape shape,
                                                                   const uint64_t *__restrict__ fine_off, const uint64_t *__restrict__ part_off,
                                                                   const uint64_t *__restrict__ wg_off, uint32_t groups, int mode = BUCKET_SUB,
                                                                   uint32_t layout_w = 0, uint64_t *__restrict__ out_vals = nullptr) {
  KMI_SCATTER_LDS(NW + VW, (PartCfg<NW + VW>::TILE))
  const uint32_t gpp = groups / kFineParts;   // K2 / E2 workgroups per part
  const uint32_t c = blockIdx.x / kFineParts, h = blockIdx.x % kFineParts;
  const uint64_t cursor = (threadIdx.x < kSubPerCoarse) ? part_off[(uint64_t)h * kNumFine + c * kSubPerCoarse + threadIdx.x] : 0ull;
  const uint64_t b = wg_off[(uint64_t)(h * gpp) * kNumCoarse + c];
  const uint64_t e = (h + 1 < (uint32_t)kFineParts) ? wg_off[(uint64_t)((h + 1) * gpp) * kNumCoarse + c] : fine_off[(c + 1) * kSubPerCoarse];
  BucketFn fn; fn.mode = mode; fn.shape = shape; fn.dist_hash = 0; fn.farm_ndebug = false; fn.nranks = 1; fn.sub = 1; fn.layout_w = layout_w;
  if (b < e) scatter_range<NW, BITS, VW>(in, b, e, out, shape, 0u, false, fn, cursor, s_stage, s_bkt, s_cnt, s_lofs, s_gbase, s_part, out_vals);
}


// P2 for one-word keys, whole-line form (written over the bucket count NB; with NB = 256, i.e. for K2 / E2, the carry
// takes half the stage: a whole-line E2 with one 1024-thread workgroup per CU and a 128 KB stage measured 4.4 ms
// against 4.2 ms for the plain form -- E2 is bound by instruction issue (61 VALU per window, rocprofv3 SQ counters),
// not by its partial lines -- so only P2 uses it). The runs a tile-wise bucket sort emits start and end wherever the
// cursors happen to stand, so every wave store begins and ends inside a 128-byte line, and such partial-line
// writes cost 1.4-1.6x (tools/microbench6.hip). Here a bucket only ever emits whole lines: after the one
// unaligned head of its stream, what it has (carry + the tile's new keys) is cut at the last line boundary, the
// tail of < 16 keys is carried into the next tile in registers (16 slots per bucket spread over the threads), and
// the copy-out walks destination lines (16 lanes = one line), not stage slots. The stream ends with one partial line.
constexpr int kLineKeys = 16;                                          // 128-byte line / 8-byte key
template <int NB> struct LinesCfg {
  static constexpr int SCAP = PartCfg<1>::TILE;                                      // stage slots: tile + carry
  static constexpr int T = (SCAP - (kLineKeys - 1) * NB) / kPartThreads * kPartThreads;   // new keys per tile (6144 / 4096)
  static constexpr int PT = T / kPartThreads;
  static constexpr int MAXG = SCAP / kLineKeys + NB;
  static constexpr int TPBK = kPartThreads / NB;                                     // threads per bucket for the carry
  static constexpr int CPT = kLineKeys / TPBK;                                       // carry slots per thread (2 / 4)
  static_assert(kPartThreads % NB == 0 && kLineKeys % TPBK == 0 && T > 0 && NB <= 256, "whole-line geometry");
};

// keyfn(raw key) -> key to store; bktfn(key) -> bucket in [0, NB)
template <int NB, typename KeyFn, typename BktFn>
__device__ __forceinline__ void scatter_lines_range(const uint64_t *__restrict__ in, uint64_t begin, uint64_t end, uint64_t *__restrict__ out,
                                                    uint64_t cursor /* of bucket threadIdx.x, threads < NB */, KeyFn keyfn, BktFn bktfn) {
  using LC = LinesCfg<NB>;
  constexpr int T = LC::T, PT = LC::PT, SCAP = LC::SCAP, MAXG = LC::MAXG, TPBK = LC::TPBK, CPT = LC::CPT;
  __shared__ uint64_t s_stage[SCAP];
  __shared__ uint64_t s_cur0[NB];      // cursor of the bucket before this tile's emission
  __shared__ uint32_t s_cnt[NB];       // S0: carry + new keys of the tile, by LDS atomic
  __shared__ uint32_t s_lofs[NB];      // first stage slot of the bucket
  __shared__ uint32_t s_emit[NB];      // keys that leave this tile (up to the last line boundary)
  __shared__ uint32_t s_old[NB];       // keys carried into this tile
  __shared__ uint32_t s_rem[NB];       // keys carried out of this tile
  __shared__ uint32_t s_lbase[NB];     // first destination-line group of the bucket
  __shared__ uint32_t s_part[NB / kWave];
  __shared__ uint32_t s_ng;
  __shared__ uint8_t s_linebkt[MAXG];  // bucket of every destination-line group
  if (begin >= end) return;
  if (threadIdx.x < NB) { s_cnt[threadIdx.x] = 0; s_rem[threadIdx.x] = 0; s_emit[threadIdx.x] = 0; s_cur0[threadIdx.x] = cursor; }
  const uint32_t cb = threadIdx.x / TPBK, cj = (threadIdx.x % TPBK) * CPT;   // carry: slots cj .. cj+CPT-1 of bucket cb
  uint64_t carry[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) carry[i] = 0;
  uint32_t my_carry = 0;   // thread b < NB: keys it carries
  lds_barrier();
  uint64_t raw[PT];
  auto load_tile = [&](uint64_t t0) {   // unconditional (clamped) loads: nothing forces an early wait
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      uint64_t i = t0 + (uint64_t)j * kPartThreads + threadIdx.x;
      i = (i < end) ? i : end - 1;
      raw[j] = in[i];
    }
  };
  load_tile(begin);
  for (uint64_t t0 = begin; t0 < end; t0 += T) {
    const uint32_t nt = (uint32_t)((end - t0 < (uint64_t)T) ? (end - t0) : (uint64_t)T);
    uint64_t k[PT];
    uint32_t rnk[PT], bkt[PT];   // the rank stays as the LDS atomic returns it: nothing here waits for the atomic
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const uint32_t li = j * kPartThreads + threadIdx.x;
      rnk[j] = 0xffffffffu;
      k[j] = keyfn(raw[j]);
      bkt[j] = bktfn(k[j]);
      if (li < nt) rnk[j] = atomicAdd(&s_cnt[bkt[j]], 1u);   // rank behind the carried keys: s_cnt starts at the carry count
    }
    if (t0 + T < end) load_tile(t0 + T);   // in flight until the next iteration needs it
    lds_barrier();
    uint32_t cnt = 0, emit = 0, ng = 0, inc = 0;
    if (threadIdx.x < NB) {   // whole waves
      cnt = s_cnt[threadIdx.x];
      const uint64_t aend = (cursor + cnt) & ~(uint64_t)(kLineKeys - 1);
      emit = aend > cursor ? (uint32_t)(aend - cursor) : 0u;
      ng = emit ? (uint32_t)((aend - (cursor & ~(uint64_t)(kLineKeys - 1))) / kLineKeys) : 0u;
      inc = wave_inclusive_scan(cnt | (ng << 16));
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
    }
    lds_barrier();
    if (threadIdx.x < NB) {
      uint32_t pre = 0;
#pragma unroll
      for (uint32_t w = 0; w < NB / kWave; ++w) pre += (w < wave_id()) ? s_part[w] : 0u;
      const uint32_t ex = pre + inc - (cnt | (ng << 16));
      const uint32_t lo = ex & 0xffffu, lb = ex >> 16;
      s_lofs[threadIdx.x] = lo; s_emit[threadIdx.x] = emit; s_old[threadIdx.x] = my_carry; s_cur0[threadIdx.x] = cursor;
      s_lbase[threadIdx.x] = lb; s_rem[threadIdx.x] = cnt - emit;
      for (uint32_t i = 0; i < ng; ++i) s_linebkt[lb + i] = (uint8_t)threadIdx.x;
      if (threadIdx.x == NB - 1) s_ng = lb + ng;
      s_cnt[threadIdx.x] = cnt - emit;   // the next tile ranks behind these
      cursor += emit; my_carry = cnt - emit;
    }
    lds_barrier();
    // stage: carried keys first, then the tile's keys, per bucket
    {
      const uint32_t oc = s_old[cb], lo = s_lofs[cb];
#pragma unroll
      for (int i = 0; i < CPT; ++i) if (cj + i < oc) s_stage[lo + cj + i] = carry[i];
    }
#pragma unroll
    for (int j = 0; j < PT; ++j)
      if (rnk[j] != 0xffffffffu) s_stage[s_lofs[bkt[j]] + rnk[j]] = k[j];
    lds_barrier();
    // copy-out by destination line: 16 lanes = one 128-byte line of one bucket
    {
      const uint32_t groups = s_ng, l16 = threadIdx.x & (kLineKeys - 1);
      for (uint32_t g = threadIdx.x >> 4; g < groups; g += kPartThreads / kLineKeys) {
        const uint32_t b = s_linebkt[g];
        const uint64_t cur0 = s_cur0[b];
        const uint64_t dpos = (cur0 & ~(uint64_t)(kLineKeys - 1)) + (uint64_t)(g - s_lbase[b]) * kLineKeys + l16;
        if (dpos >= cur0 && dpos < cur0 + s_emit[b]) out[dpos] = s_stage[s_lofs[b] + (uint32_t)(dpos - cur0)];
      }
      // what stays behind the last line boundary travels on in registers
      const uint32_t rem = s_rem[cb], base = s_lofs[cb] + s_emit[cb];
#pragma unroll
      for (int i = 0; i < CPT; ++i) if (cj + i < rem) carry[i] = s_stage[base + cj + i];
    }
    lds_barrier();   // the next tile rewrites the stage and the per-bucket tables
  }
  // the streams end with one partial line each
  {
    const uint32_t rem = s_rem[cb];
    const uint64_t fc = s_cur0[cb] + s_emit[cb];
#pragma unroll
    for (int i = 0; i < CPT; ++i) if (cj + i < rem) out[fc + cj + i] = carry[i];
  }
}

// P2, one-word keys: workgroup (c, h) splits its part of coarse bucket c into the 128 sub-buckets
__global__ __launch_bounds__(kPartThreads) void scatter_fine_lines_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                                         const uint64_t *__restrict__ fine_off,
                                                                         const uint64_t *__restrict__ part_off,
                                                                         const uint64_t *__restrict__ wg_off, uint32_t groups,
                                                                         uint32_t layout_w = 0, uint32_t k = 0) {
  constexpr int NB = kSubPerCoarse;
  const uint32_t gpp = groups / kFineParts;   // K2 / E2 workgroups per part
  const uint32_t c = blockIdx.x / kFineParts, h = blockIdx.x % kFineParts;
  const uint64_t cursor = (threadIdx.x < NB) ? part_off[(uint64_t)h * kNumFine + c * NB + threadIdx.x] : 0ull;
  const uint64_t begin = wg_off[(uint64_t)(h * gpp) * kNumCoarse + c];
  const uint64_t end = (h + 1 < (uint32_t)kFineParts) ? wg_off[(uint64_t)((h + 1) * gpp) * kNumCoarse + c] : fine_off[(c + 1) * NB];
  scatter_lines_range<NB>(in, begin, end, out, cursor, [](uint64_t r) { return r; },
                          [=](uint64_t key) { const uint64_t kk[1] = {key}; return fine15_of_key<1>(kk, layout_w, k) & (uint32_t)(NB - 1); });
}

// P2 for RECORDS (key words + value words: the tuples of the position indexes), whole-line form. The plain form (scatter_range) writes
// a tile's records of a fine bucket where the bucket's previous tile ended: runs of 16-32 records that begin and end inside 128-byte
// lines, 3.4 TB/s where whole lines reach 5.7 (tools/write_runs.hip, modes 11 / 12) -- and this pass is a copy. Here, as in
// scatter_lines_range, a bucket only ever emits whole groups of 16 ELEMENTS after the unaligned head of its stream: 16 elements are
// NW whole lines of the key array and VW whole lines of the value array (or NW + VW lines of one record array) whatever the word counts
// are, so one element count cuts both output streams at line boundaries. The tail of < 16 elements is carried into the next tile in
// registers (16 slots per bucket spread over eight threads); the copy-out walks destination groups (16 lanes = one group).
template <int RW> struct RecLinesCfg {
  static constexpr int NB = kSubPerCoarse;
  static constexpr int SCAP = (RW <= 2) ? 4096 : 3072;                                      // stage elements: tile + carry (64 / 72 / 96 KB)
  static constexpr int T = (SCAP - (kLineKeys - 1) * NB) / kPartThreads * kPartThreads;    // new elements per tile (2048 / 1024 / 1024)
  static constexpr int PT = T / kPartThreads;
  static constexpr int MAXG = SCAP / kLineKeys + NB;
  static constexpr int TPBK = kPartThreads / NB;                                           // threads per bucket for the carry (8)
  static constexpr int CPT = kLineKeys / TPBK;                                             // carry slots per thread (2)
  static_assert(kPartThreads % NB == 0 && kLineKeys % TPBK == 0 && T > 0, "whole-line geometry");
};
template <int NW, int VW, typename BktFn>
__device__ __forceinline__ void scatter_lines_records(const uint64_t *__restrict__ in, uint64_t begin, uint64_t end, uint64_t *__restrict__ out,
                                                      uint64_t *__restrict__ out_vals /* null: whole records to `out` */,
                                                      uint64_t cursor /* of bucket threadIdx.x, threads < NB; in elements */, BktFn bktfn) {
  constexpr int RW = NW + VW;
  using LC = RecLinesCfg<RW>;
  constexpr int NB = LC::NB, T = LC::T, PT = LC::PT, SCAP = LC::SCAP, MAXG = LC::MAXG, TPBK = LC::TPBK, CPT = LC::CPT;
  __shared__ uint64_t s_stage[SCAP * RW];
  __shared__ uint64_t s_cur0[NB];      // cursor of the bucket before this tile's emission
  __shared__ uint32_t s_cnt[NB];       // carry + new elements of the tile, by LDS atomic
  __shared__ uint32_t s_lofs[NB];      // first stage slot of the bucket
  __shared__ uint32_t s_emit[NB];      // elements that leave this tile (up to the last group boundary)
  __shared__ uint32_t s_old[NB];       // elements carried into this tile
  __shared__ uint32_t s_rem[NB];       // elements carried out of this tile
  __shared__ uint32_t s_lbase[NB];     // first destination group of the bucket
  __shared__ uint32_t s_part[NB / kWave];
  __shared__ uint32_t s_ng;
  __shared__ uint8_t s_linebkt[MAXG];  // bucket of every destination group
  if (begin >= end) return;
  if (threadIdx.x < NB) { s_cnt[threadIdx.x] = 0; s_rem[threadIdx.x] = 0; s_emit[threadIdx.x] = 0; s_cur0[threadIdx.x] = cursor; }
  const uint32_t cb = threadIdx.x / TPBK, cj = (threadIdx.x % TPBK) * CPT;   // carry: slots cj .. cj + CPT - 1 of bucket cb
  uint64_t carry[CPT][RW];
#pragma unroll
  for (int i = 0; i < CPT; ++i)
#pragma unroll
    for (int w = 0; w < RW; ++w) carry[i][w] = 0;
  uint32_t my_carry = 0;   // thread b < NB: elements it carries
  lds_barrier();
  uint64_t raw[PT][RW];
  auto load_tile = [&](uint64_t t0) {   // unconditional (clamped) loads: nothing forces an early wait
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      uint64_t i = t0 + (uint64_t)j * kPartThreads + threadIdx.x;
      i = (i < end) ? i : end - 1;
#pragma unroll
      for (int w = 0; w < RW; ++w) raw[j][w] = in[i * RW + w];
    }
  };
  auto put = [&](uint64_t dpos, const uint64_t *src) {   // element dpos of the output arrays
    if (out_vals) {
#pragma unroll
      for (int w = 0; w < NW; ++w) out[dpos * NW + w] = src[w];
#pragma unroll
      for (int w = 0; w < VW; ++w) out_vals[dpos * VW + w] = src[NW + w];
    } else {
#pragma unroll
      for (int w = 0; w < RW; ++w) out[dpos * RW + w] = src[w];
    }
  };
  load_tile(begin);
  for (uint64_t t0 = begin; t0 < end; t0 += T) {
    const uint32_t nt = (uint32_t)((end - t0 < (uint64_t)T) ? (end - t0) : (uint64_t)T);
    uint64_t el[PT][RW];
    uint32_t rnk[PT], bkt[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const uint32_t li = j * kPartThreads + threadIdx.x;
      rnk[j] = 0xffffffffu;
      uint64_t key[NW];
#pragma unroll
      for (int w = 0; w < RW; ++w) el[j][w] = raw[j][w];
#pragma unroll
      for (int w = 0; w < NW; ++w) key[w] = raw[j][w];
      bkt[j] = bktfn(key);
      if (li < nt) rnk[j] = atomicAdd(&s_cnt[bkt[j]], 1u);   // rank behind the carried elements: s_cnt starts at the carry count
    }
    if (t0 + T < end) load_tile(t0 + T);   // in flight until the next iteration needs it
    lds_barrier();
    uint32_t cnt = 0, emit = 0, ng = 0, inc = 0;
    if (threadIdx.x < NB) {   // whole waves
      cnt = s_cnt[threadIdx.x];
      const uint64_t aend = (cursor + cnt) & ~(uint64_t)(kLineKeys - 1);
      emit = aend > cursor ? (uint32_t)(aend - cursor) : 0u;
      ng = emit ? (uint32_t)((aend - (cursor & ~(uint64_t)(kLineKeys - 1))) / kLineKeys) : 0u;
      inc = wave_inclusive_scan(cnt | (ng << 16));
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
    }
    lds_barrier();
    if (threadIdx.x < NB) {
      uint32_t pre = 0;
#pragma unroll
      for (uint32_t w = 0; w < NB / kWave; ++w) pre += (w < wave_id()) ? s_part[w] : 0u;
      const uint32_t ex = pre + inc - (cnt | (ng << 16));
      const uint32_t lo = ex & 0xffffu, lb = ex >> 16;
      s_lofs[threadIdx.x] = lo; s_emit[threadIdx.x] = emit; s_old[threadIdx.x] = my_carry; s_cur0[threadIdx.x] = cursor;
      s_lbase[threadIdx.x] = lb; s_rem[threadIdx.x] = cnt - emit;
      for (uint32_t i = 0; i < ng; ++i) s_linebkt[lb + i] = (uint8_t)threadIdx.x;
      if (threadIdx.x == NB - 1) s_ng = lb + ng;
      s_cnt[threadIdx.x] = cnt - emit;   // the next tile ranks behind these
      cursor += emit; my_carry = cnt - emit;
    }
    lds_barrier();
    // stage: carried elements first, then the tile's, per bucket
    {
      const uint32_t oc = s_old[cb], lo = s_lofs[cb];
#pragma unroll
      for (int i = 0; i < CPT; ++i)
        if (cj + i < oc) {
#pragma unroll
          for (int w = 0; w < RW; ++w) s_stage[(uint64_t)(lo + cj + i) * RW + w] = carry[i][w];
        }
    }
#pragma unroll
    for (int j = 0; j < PT; ++j)
      if (rnk[j] != 0xffffffffu) {
        const uint32_t pos = s_lofs[bkt[j]] + rnk[j];
#pragma unroll
        for (int w = 0; w < RW; ++w) s_stage[(uint64_t)pos * RW + w] = el[j][w];
      }
    lds_barrier();
    // copy-out by destination group: 16 lanes = 16 elements of one bucket = whole lines of either output array
    {
      const uint32_t groups = s_ng, l16 = threadIdx.x & (kLineKeys - 1);
      for (uint32_t g = threadIdx.x >> 4; g < groups; g += kPartThreads / kLineKeys) {
        const uint32_t b = s_linebkt[g];
        const uint64_t cur0 = s_cur0[b];
        const uint64_t dpos = (cur0 & ~(uint64_t)(kLineKeys - 1)) + (uint64_t)(g - s_lbase[b]) * kLineKeys + l16;
        if (dpos >= cur0 && dpos < cur0 + s_emit[b]) put(dpos, &s_stage[(uint64_t)(s_lofs[b] + (uint32_t)(dpos - cur0)) * RW]);
      }
      // what stays behind the last group boundary travels on in registers
      const uint32_t rem = s_rem[cb], base = s_lofs[cb] + s_emit[cb];
#pragma unroll
      for (int i = 0; i < CPT; ++i)
        if (cj + i < rem) {
#pragma unroll
          for (int w = 0; w < RW; ++w) carry[i][w] = s_stage[(uint64_t)(base + cj + i) * RW + w];
        }
    }
    lds_barrier();   // the next tile rewrites the stage and the per-bucket tables
  }
  // the streams end with one partial group each
  {
    const uint32_t rem = s_rem[cb];
    const uint64_t fc = s_cur0[cb] + s_emit[cb];
#pragma unroll
    for (int i = 0; i < CPT; ++i) if (cj + i < rem) put(fc + cj + i, carry[i]);
  }
}

// P2 of the position indexes' records: workgroup (c, h) as in scatter_fine_kernel
template <int NW, int BITS, int VW>
__global__ __launch_bounds__(kPartThreads) void scatter_fine_records_lines_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, KShape shape,
                                                                                 const uint64_t *__restrict__ fine_off, const uint64_t *__restrict__ part_off,
                                                                                 const uint64_t *__restrict__ wg_off, uint32_t groups, uint32_t layout_w,
                                                                                 uint64_t *__restrict__ out_vals) {
  constexpr int NB = kSubPerCoarse;
  const uint32_t gpp = groups / kFineParts;
  const uint32_t c = blockIdx.x / kFineParts, h = blockIdx.x % kFineParts;
  const uint64_t cursor = (threadIdx.x < NB) ? part_off[(uint64_t)h * kNumFine + c * NB + threadIdx.x] : 0ull;
  const uint64_t begin = wg_off[(uint64_t)(h * gpp) * kNumCoarse + c];
  const uint64_t end = (h + 1 < (uint32_t)kFineParts) ? wg_off[(uint64_t)((h + 1) * gpp) * kNumCoarse + c] : fine_off[(c + 1) * NB];
  const uint32_t k = shape.k;
  scatter_lines_records<NW, VW>(in, begin, end, out, out_vals, cursor,
                                [=](const uint64_t (&key)[NW]) { return fine15_of_key<NW>(key, layout_w, k) & (uint32_t)(NB - 1); });
}

// ---------------------------------------------------------------------------
// Fused build (Index::build_* on one rank): the k-mers are generated from the FASTQ tiles
// inside the histogram pass (E1) and again inside the coarse scatter pass (E2), so the
// extracted tuple array never exists in HBM: 2 x 2.6 B/k-mer of input reads replace
// 8 W + 8 R + 8 R of key traffic. Workgroup w owns the same contiguous run of tiles in both.
// ---------------------------------------------------------------------------
// L: the entry list. Every run of up to 8 consecutive k-mer windows of a read becomes one 16-bit entry
// (tile position of the first window | (windows - 1) << 13), about 0.27 bytes per k-mer; the entries of scan tile t sit
// in a slot of ent_stride(k) entries with their number in ent_cnt[t]. The histogram and scatter passes start from it
// and have no per-byte work left: they read the first k-mer of an entry from the packed stream and roll the others.
// The pass works per LINE, not per byte, and one WAVEFRONT owns a tile, so there is no
// workgroup barrier in it. The lanes share the words of the tile's EOL bitmap plus 1 KB of context on either
// side; line starts (non-EOL after EOL) and line ends (EOL after non-EOL) are ranked with one wave scan and their
// positions land in two small LDS arrays, so line j of the window is [S[j], E[j + eoff]). A line whose index (from
// the scan's line bases) says "sequence" becomes a run (first window, number of windows) clipped to the tile; a
// line that says "quality" is compared with the line two ranks before it (fastq_loader.hpp:454-463); sixteen lanes
// then cut each run into entries. Windows belong to the tile they start in. A window that holds more than
// CAP lines falls back to per-word bit scans.
constexpr int kListThreads = 256;
template <int NW, int BITS> struct ListPassCfg {
  using Cfg = ExCfg<NW, BITS>;
  static constexpr int TILE = Cfg::TILE;
  static constexpr int WORDS = TILE / 32;
  static constexpr int CTX = 32;                               // bitmap words of context on either side (1 KB >= k - 1)
  static constexpr int WIN = WORDS + 2 * CTX;
  static constexpr int WPL = WIN / kWave;                      // bitmap words per lane
  static constexpr int CAP = 512;                              // lines per window on the dense path
  static_assert(WIN % kWave == 0 && WIN * 32 < 65536 && CTX * 32 >= Cfg::KMAX, "window geometry");
  // a run needs a record of >= k + 7 bytes
  // (with a sequence filter the runs are cut at break bytes: a piece needs k bytes and one break)
  static uint32_t max_runs(uint32_t k, bool split = false) { return (uint32_t)TILE / (k + (split ? 1u : 7u)) + 2u; }
  // entry list: one 16-bit entry per run of up to 8 consecutive windows; per-tile slots of this many entries
  static uint32_t ent_stride(uint32_t k, bool split = false) { return (uint32_t)TILE / 8u + max_runs(k, split); }
  // run list (RUNS): one 32-bit entry per run of up to `seg` windows
  static uint32_t run_stride(uint32_t k, uint32_t seg, bool split = false) { return (uint32_t)TILE / seg + max_runs(k, split); }
  static uint32_t wave_lds_bytes(uint32_t k, bool split = false, bool reads = false) {   // bitmap window, event arrays, run table (two with a filter; + the runs' line numbers for read descriptors)
    return (4u * WIN + 4u * CAP + 4u * max_runs(k, split) * ((split ? 2u : 1u) + (reads ? 1u : 0u)) + 15u) & ~15u;
  }
};

// RUNS = true: the list holds the runs themselves, cut only every kSegWindows windows, as 32-bit entries (tile position of the
// first window | (windows - 1) << 13): what the super-k-mer passes start from (one lane walks one run).
template <int NW, int BITS, bool RUNS = false>
__global__ __launch_bounds__(kListThreads) void fastq_list_kernel(PackedInput in, uint64_t n_tiles, uint32_t k, uint32_t max_runs,
                                                                 uint32_t wave_lds_bytes, const uint32_t *__restrict__ line_base,
                                                                 uint32_t *__restrict__ flags, void *__restrict__ ent_out,
                                                                 uint32_t *__restrict__ ent_cnt, uint32_t ent_stride,
                                                                 uint32_t kSegWindows = 128u /* RUNS: windows per entry at most */,
                                                                 ReadDesc *__restrict__ reads = nullptr, const uint64_t *__restrict__ tile_off = nullptr) {
  // reads / tile_off (position + quality builds, no sequence filter): the descriptor of every read whose first window starts in a
  // tile -- slot = the read's sequence index, {buffer position of its first base, file-order index of its first k-mer} -- which is
  // what fastq_quality_kernel starts from (until round 4 the extract pass wrote them). A crowded window (lines of a few bytes)
  // does not make them: flag word 0, bit 3, and the caller takes the extract pass.
  uint16_t *const ent = reinterpret_cast<uint16_t *>(ent_out);
  uint32_t *const ent32 = reinterpret_cast<uint32_t *>(ent_out);
  // in.brk (sequence filters): the runs are cut where a break bit (an N by the filter's rule) falls inside them
  using P = ListPassCfg<NW, BITS>;
  constexpr int TILE = P::TILE, WORDS = P::WORDS, CTX = P::CTX, WIN = P::WIN, WPL = P::WPL, CAP = P::CAP;
  constexpr uint32_t T0 = CTX * 32u, T1 = T0 + (uint32_t)TILE, NONE = 0xffff0000u;   // tile proper in window positions
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  uint32_t *img = reinterpret_cast<uint32_t *>(s_dyn + (size_t)wave_id() * wave_lds_bytes);   // [WIN] bitmap words of the window
  uint16_t *S = reinterpret_cast<uint16_t *>(img + WIN);                                      // [CAP] line starts
  uint16_t *E = S + CAP;                                                                      // [CAP] line ends
  uint32_t *s_run = reinterpret_cast<uint32_t *>(E + CAP);                                    // [max_runs] first window | windows << 16
  uint32_t *s_line = s_run + max_runs * (in.brk ? 2u : 1u);                                   // [max_runs] (reads) line number of the run | 1 << 31: the run opens its line
  const uint32_t *eolw = reinterpret_cast<const uint32_t *>(in.eol);
  const uint64_t n_words = in.n_cover / 32;
  const uint64_t n_waves = (uint64_t)gridDim.x * (kListThreads / kWave);
  const uint64_t wv = (uint64_t)blockIdx.x * (kListThreads / kWave) + wave_id();
  const uint64_t per = (n_tiles + n_waves - 1) / n_waves;
  const uint64_t tb = wv * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  const uint32_t lane = lane_id();
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  auto load_words = [&](uint64_t tile, uint32_t (&w)[WPL]) {   // window words of `tile`; outside the bitmap = EOL
    const int64_t g0 = (int64_t)(tile * WORDS) - CTX + (int64_t)(lane * WPL);
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      const int64_t g = g0 + i;
      const bool ok = g >= 0 && (uint64_t)g < n_words;
      const uint32_t v = eolw[ok ? g : 0];
      w[i] = ok ? v : 0xffffffffu;
    }
  };
  // first EOL position >= `from` in the window image; NONE when there is none
  auto next_eol = [&](uint32_t from) -> uint32_t {
    uint32_t w = from >> 5;
    uint32_t bits = img[w] & (0xffffffffu << (from & 31u));
    while (bits == 0u && ++w < (uint32_t)WIN) bits = img[w];
    return bits ? (w << 5) + (uint32_t)__builtin_ctz(bits) : NONE;
  };
  // windows of the sequence line [s, e) that start inside the tile: first window | count << 16 (0 = none)
  auto clip_run = [&](uint32_t s, uint32_t e) -> uint32_t {
    const uint32_t first = s > T0 ? s : T0;
    if (e < first + k) return 0u;
    uint32_t last = e - k;
    if (last > T1 - 1u) last = T1 - 1u;
    return (first - T0) | ((last - first + 1u) << 16);
  };
  uint32_t w_cur[WPL], w_nxt[WPL];
#pragma unroll
  for (int i = 0; i < WPL; ++i) w_cur[i] = 0xffffffffu;
  if (tb < te) load_words(tb, w_cur);
  uint32_t lb = (tb < te) ? line_base[tb] : 0u;
  bool bad = false, bad_reads = false;
  for (uint64_t t = tb; t < te; ++t) {
    const int64_t gw0 = (int64_t)(t * WORDS) - CTX;
#pragma unroll
    for (int i = 0; i < WPL; ++i) img[lane * WPL + i] = w_cur[i];
    const uint32_t prev_win = (gw0 > 0) ? (eolw[gw0 - 1] >> 31) : 1u;   // EOL status of the byte before the window
    load_words(t + 1, w_nxt);   // the next tile's inputs are in flight during this tile
    const uint32_t lb_nxt = line_base[(t + 1 < n_tiles) ? t + 1 : t];
    uint32_t prev = __shfl_up(w_cur[WPL - 1] >> 31, 1, kWave);
    if (lane == 0) prev = prev_win;
    uint32_t ls[WPL], le[WPL], ns = 0, ne = 0, nsl = 0, nst = 0;
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      const uint32_t before = (w_cur[i] << 1) | prev;   // EOL status of each position's predecessor
      ls[i] = ~w_cur[i] & before;
      le[i] = w_cur[i] & ~before;
      prev = w_cur[i] >> 31;
      const uint32_t c = (uint32_t)__builtin_popcount(ls[i]);
      const uint32_t ww = lane * WPL + i;
      ns += c; ne += (uint32_t)__builtin_popcount(le[i]);
      nsl += (ww < (uint32_t)CTX) ? c : 0u;
      nst += (ww < (uint32_t)(CTX + WORDS)) ? c : 0u;
    }
    const uint32_t packed = ns | (ne << 16);
    const uint32_t inc = wave_inclusive_scan(packed);
    const uint32_t tot = __shfl(inc, kWave - 1, kWave);
    const uint32_t NS = tot & 0xffffu, NE = tot >> 16;
    const uint32_t cnt2 = wave_reduce_sum(nsl | (nst << 16));
    const uint32_t NSL = cnt2 & 0xffffu, NST = cnt2 >> 16;   // line starts before the tile / before the tile's end
    EolBits bm;
    bm.g = eolw; bm.n_words = n_words; bm.img = img; bm.nw = WIN; bm.w0 = (uint64_t)gw0;
    const uint64_t win0 = (uint64_t)(gw0 * 32);               // buffer position of window position 0 (wraps for the first tile, used additively)
    uint32_t n_runs = 0;
    wave_sync();   // window image complete
    if (NS <= (uint32_t)CAP && NE <= (uint32_t)CAP) {
      uint32_t rs = (inc - packed) & 0xffffu, re = (inc - packed) >> 16;
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const uint32_t base = (lane * WPL + i) * 32u;
        uint32_t r = ls[i];
        while (r) { S[rs++] = (uint16_t)(base + (uint32_t)__builtin_ctz(r)); r &= r - 1u; }
        r = le[i];
        while (r) { E[re++] = (uint16_t)(base + (uint32_t)__builtin_ctz(r)); r &= r - 1u; }
      }
      wave_sync();
      const uint32_t eoff = prev_win ? 0u : 1u;   // a line open at the window start owns the first end event
      // candidate lines: c = 0 is the line open at the tile start (the last start before the tile, or the line that
      // was already open at the window start), c >= 1 are the lines that start inside the tile; line index lb + c - 1
      const uint32_t ncand = 1u + (NST - NSL);
      for (uint32_t c0 = 0; c0 < ncand; c0 += kWave) {
        const uint32_t c = c0 + lane;
        const int32_t j = (int32_t)(NSL + c) - 1;            // rank among the window's line starts; -1 = open at window start
        bool exists = c < ncand && (c >= 1u || (lb >= 1u && (j >= 0 || eoff)));
        const uint32_t role = (lb + c - 1u) & 3u;
        uint32_t run = 0, line_of_run = 0;
        if (exists && (role == 1u || role == 3u)) {
          const uint32_t s0 = (j >= 0) ? (uint32_t)S[j] : 0u;
          const int32_t je = j + (int32_t)eoff;
          const uint32_t e0 = (je >= 0 && (uint32_t)je < NE) ? (uint32_t)E[je] : NONE;   // NONE: no EOL up to k - 1 bytes past the tile
          if (role == 1u) {
            run = clip_run(s0, e0);
            if (reads && run) line_of_run = (lb + c - 1u) | ((j >= 0 && s0 >= T0) ? 0x80000000u : 0u);
          } else if (c >= 1u) {                              // a quality line that starts in this tile
            if (j >= 2 && e0 != NONE) {
              const uint32_t len_seq = (uint32_t)E[je - 2] - (uint32_t)S[j - 2];
              bad = bad || (e0 - s0 != len_seq);
            } else {
              bad = bad || fastq_lengths_differ(bm, win0 + s0);
            }
          }
        }
        const uint32_t mine = run ? 1u : 0u;
        const uint32_t sc = wave_inclusive_scan(mine);
        if (run) s_run[n_runs + sc - 1u] = run;
        if (reads && run) s_line[n_runs + sc - 1u] = line_of_run;
        n_runs += __shfl(sc, kWave - 1, kWave);
      }
    } else {
      if (reads) bad_reads = true;
      // crowded window: the lanes walk the line starts of their own words (tile proper only) with bit scans
      const uint32_t first_idx = lb + ((inc - packed) & 0xffffu) - NSL;   // line index of this lane's first start (meaningful inside the tile)
      const bool carry_in = (lane == 0) && lb >= 1u && ((lb - 1u) & 3u) == 1u && (img[CTX] & 1u) == 0u &&
                            ((img[CTX - 1] >> 31) == 0u);               // a sequence line runs across the tile start
      auto walk = [&](auto f) {
        if (carry_in) { const uint32_t r = clip_run(T0, next_eol(T0)); if (r) f(r); }
        uint32_t idx = first_idx;
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
          const uint32_t ww = lane * WPL + i;
          uint32_t rest = ls[i];
          while (rest) {
            const uint32_t s0 = ww * 32u + (uint32_t)__builtin_ctz(rest);
            if (ww >= (uint32_t)CTX && ww < (uint32_t)(CTX + WORDS)) {
              if ((idx & 3u) == 1u) { const uint32_t r = clip_run(s0, next_eol(s0)); if (r) f(r); }
            }
            ++idx; rest &= rest - 1u;
          }
        }
      };
      uint32_t mine = 0;
      walk([&](uint32_t) { ++mine; });
      {   // the length rule for the quality lines of the tile
        uint32_t idx = first_idx;
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
          const uint32_t ww = lane * WPL + i;
          uint32_t rest = ls[i];
          while (rest) {
            if (ww >= (uint32_t)CTX && ww < (uint32_t)(CTX + WORDS) && (idx & 3u) == 3u)
              bad = bad || fastq_lengths_differ(bm, win0 + ww * 32u + (uint32_t)__builtin_ctz(rest));
            ++idx; rest &= rest - 1u;
          }
        }
      }
      const uint32_t sc = wave_inclusive_scan(mine);
      n_runs = __shfl(sc, kWave - 1, kWave);
      if (mine) {
        uint32_t r = sc - mine;
        walk([&](uint32_t run) { s_run[r++] = run; });
      }
    }
    wave_sync();   // run table complete
    const uint32_t *runs = s_run;
    if (in.brk) {   // uniform
      // every lane cuts its runs into the pieces that hold no break byte; counted, ranked with a wave scan, then written
      const uint32_t *brkw = reinterpret_cast<const uint32_t *>(in.brk);
      uint32_t *s_run2 = s_run + max_runs;
      const uint64_t tile0 = t * (uint64_t)TILE;
      // first position >= p whose break bit equals `set` (past the bitmap everything is a break)
      auto next_bit = [&](uint64_t p, bool set) -> uint64_t {
        uint64_t wi = p >> 5;
        if (wi >= n_words) return set ? p : n_words * 32;
        uint32_t bits = (set ? brkw[wi] : ~brkw[wi]) & (0xffffffffu << (p & 31u));
        while (bits == 0u) { if (++wi >= n_words) return n_words * 32; bits = set ? brkw[wi] : ~brkw[wi]; }
        return wi * 32 + (uint32_t)__builtin_ctz(bits);
      };
      auto pieces = [&](uint32_t run, auto f) {   // f(first window (tile position), windows)
        const uint64_t lo = tile0 + (run & 0xffffu), hi = lo + (run >> 16) + k - 1u;
        for (uint64_t p = lo; p < hi;) {
          const uint64_t a = next_bit(p, false);
          if (a >= hi) break;
          uint64_t b = next_bit(a, true);
          b = b < hi ? b : hi;
          if (b - a >= k) f((uint32_t)(a - tile0), (uint32_t)(b - a - k + 1u));
          p = b;
        }
      };
      uint32_t n2 = 0;
      for (uint32_t r0 = 0; r0 < n_runs; r0 += kWave) {
        const uint32_t run = (r0 + lane < n_runs) ? s_run[r0 + lane] : 0u;
        uint32_t mine = 0;
        if (run) pieces(run, [&](uint32_t, uint32_t) { ++mine; });
        const uint32_t inc = wave_inclusive_scan(mine);
        uint32_t o = n2 + inc - mine;
        if (run) pieces(run, [&](uint32_t first, uint32_t c) { s_run2[o++] = first | (c << 16); });
        n2 += __shfl(inc, kWave - 1, kWave);
      }
      wave_sync();
      runs = s_run2; n_runs = n2;
    }
    uint32_t ebase = 0;   // entries of this tile so far
    uint16_t *etile = ent + (uint64_t)t * ent_stride;
    if (reads && !in.brk && !bad_reads) {   // uniform: descriptors of the reads that open in this tile
      uint32_t wbase = 0;   // windows of the tile's runs so far
      for (uint32_t r0 = 0; r0 < n_runs; r0 += kWave) {
        const bool have = r0 + lane < n_runs;
        const uint32_t my_sc = have ? runs[r0 + lane] : 0u, my_ln = have ? s_line[r0 + lane] : 0u;
        const uint32_t c = my_sc >> 16;
        const uint32_t winc = wave_inclusive_scan(c);
        if (have && (my_ln >> 31)) {
          ReadDesc rd; rd.seq_pos = t * (uint64_t)TILE + (my_sc & 0xffffu); rd.out_off = tile_off[t] + wbase + winc - c;
          reads[((my_ln & 0x7fffffffu) - 1u) >> 2] = rd;
        }
        wbase += __shfl(winc, kWave - 1, kWave);
      }
    }
    if constexpr (RUNS) {
      uint32_t *etile32 = ent32 + (uint64_t)t * ent_stride;
      for (uint32_t r0 = 0; r0 < n_runs; r0 += kWave) {
        const uint32_t my_sc = (r0 + lane < n_runs) ? runs[r0 + lane] : 0u;
        const uint32_t s0 = my_sc & 0xffffu, c = my_sc >> 16;
        const uint32_t my_ne = (c + kSegWindows - 1u) / kSegWindows;
        const uint32_t einc = wave_inclusive_scan(my_ne);
        const uint32_t my_eo = ebase + einc - my_ne;
        for (uint32_t j = 0; j < my_ne; ++j) {
          const uint32_t left = c - kSegWindows * j;
          etile32[my_eo + j] = (s0 + kSegWindows * j) | (((left < kSegWindows ? left : kSegWindows) - 1u) << 13);
        }
        ebase += __shfl(einc, kWave - 1, kWave);
      }
    } else
    for (uint32_t r0 = 0; r0 < n_runs; r0 += kWave) {
      const uint32_t nr = (n_runs - r0 < (uint32_t)kWave) ? n_runs - r0 : (uint32_t)kWave;
      const uint32_t my_sc = (lane < nr) ? runs[r0 + lane] : 0u;
      // entry list: a run of c windows = ceil(c / 8) entries (position | (windows - 1) << 13); 16 lanes per run
      const uint32_t my_ne = ((my_sc >> 16) + 7u) >> 3;
      const uint32_t einc = wave_inclusive_scan(my_ne);
      const uint32_t my_eo = ebase + einc - my_ne;
      for (uint32_t g = 0; g < nr; g += kWave / 16) {
        const int rr = (int)(g + (lane >> 4));
        const uint32_t sc = __shfl(my_sc, rr, kWave), eo = __shfl(my_eo, rr, kWave), ne = __shfl(my_ne, rr, kWave);
        if ((uint32_t)rr < nr) {
          const uint32_t s0 = sc & 0xffffu, c = sc >> 16;
          for (uint32_t j = lane & 15u; j < ne; j += 16u) {
            const uint32_t left = c - 8u * j;
            etile[eo + j] = (uint16_t)((s0 + 8u * j) | (((left < 8u ? left : 8u) - 1u) << 13));
          }
        }
      }
      ebase += __shfl(einc, kWave - 1, kWave);
    }
    if (lane == 0) ent_cnt[t] = ebase;
    wave_sync();   // the next tile overwrites the image, the event arrays and the run table
#pragma unroll
    for (int i = 0; i < WPL; ++i) w_cur[i] = w_nxt[i];
    lb = lb_nxt;
  }
  if (bad) atomicOr(&flags[0], 4u);
  if (bad_reads) atomicOr(&flags[0], 8u);
}

// geometry of the list-driven passes
template <int NW, int BITS> struct ListCfg {
  using Cfg = ExCfg<NW, BITS>;
  static constexpr int NT = Cfg::NT;
  static constexpr int MAXQ = 61440 / (8 * NW * NT);       // windows per thread and round (scatter)
  static constexpr int CAPW = MAXQ * NT;                   // windows per round (stage capacity)
  static constexpr int RMAX = (BITS >= 3) ? 2 : 4;         // scan tiles per round (LDS image of their packed stream)
  static constexpr int UNITS = RMAX * NT + Cfg::HALO_CHUNKS;
  static constexpr int STREAM_DW = (UNITS * Cfg::C * BITS + 31) / 32 + 2 * NW + 2;
  static constexpr int ULOADS = (UNITS + NT - 1) / NT;     // stream units per thread (scatter)
};

// E1: fine histogram + per-workgroup coarse counts, driven by the entry list (one entry = up to 8 consecutive windows
// of a read: the first k-mer is read from the stream image, the others roll). A round = the RMAX scan tiles whose
// packed stream sits in LDS as one image; the image is double-buffered, so there is one barrier per round.
constexpr int kHistThreads = 1024;
template <int NW, int BITS, typename F>
__device__ __forceinline__ void for_entry_windows(const uint32_t *img, uint32_t pos, uint32_t len, const KShape &shape, bool canonical, F f) {
  using Cfg = ExCfg<NW, BITS>;
  if constexpr (NW == 1 && BITS == 2) {
    // uniform dispatch: the rolled loop is specialised on the word the top code lives in, on the strand rule, and on
    // whether every lane of the wavefront holds a full entry of 8 windows (whole reads: nearly always)
    const bool full = __all(len == 8u);
    if (shape.k >= 17u) {
      if (canonical) { if (full) roll_entry_windows<true, true, true>(img, pos, len, shape, f); else roll_entry_windows<true, true, false>(img, pos, len, shape, f); }
      else { if (full) roll_entry_windows<true, false, true>(img, pos, len, shape, f); else roll_entry_windows<true, false, false>(img, pos, len, shape, f); }
    } else {
      if (canonical) roll_entry_windows<false, true, false>(img, pos, len, shape, f);
      else roll_entry_windows<false, false, false>(img, pos, len, shape, f);
    }
  } else {
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) {
      if (j < len) {
        uint64_t rc[NW], fw[NW], key[NW];
        window_at<Cfg>(img, pos + j, shape, rc, fw);
        select_strand<NW>(rc, fw, canonical, key);
        f(j, key);
      }
    }
  }
}

template <int NW, int BITS>
__global__ __launch_bounds__(kHistThreads) void fastq_hist_list_kernel(PackedInput in, uint64_t n_tiles, KShape shape, bool canonical,
                                                                      const uint16_t *__restrict__ ent, const uint32_t *__restrict__ ent_cnt,
                                                                      uint32_t ent_stride, uint32_t *__restrict__ fine_hist,
                                                                      uint32_t *__restrict__ wg_hist) {
  using Cfg = ExCfg<NW, BITS>;
  using L = ListCfg<NW, BITS>;
  constexpr int NT = kHistThreads, RMAX = L::RMAX;
  constexpr int UL = (L::UNITS + NT - 1) / NT;
  constexpr int EB = 2;   // entries in flight per thread
  __shared__ uint32_t s_hist[kNumFine];
  __shared__ uint32_t s_stream[2][L::STREAM_DW];
  for (int i = threadIdx.x; i < kNumFine; i += NT) s_hist[i] = 0;
  const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t tb = (uint64_t)blockIdx.x * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  const uint64_t last_unit = in.n_cover / Cfg::C - 1;
  auto load_image = [&](uint64_t t, uint64_t (&st)[UL]) {
#pragma unroll
    for (int i = 0; i < UL; ++i) {
      uint64_t g = t * Cfg::NT + (uint64_t)i * NT + threadIdx.x;
      g = g < last_unit ? g : last_unit;
      st[i] = read_stream_unit<BITS, Cfg::C>(in.stream, g);
    }
  };
  auto store_image = [&](uint32_t *img, const uint64_t (&st)[UL]) {
#pragma unroll
    for (int i = 0; i < UL; ++i) {
      const int u = i * NT + threadIdx.x;
      if (u < L::UNITS) store_stream_bits<BITS, Cfg::C>(img, u, st[i]);
    }
  };
  uint64_t st[UL];
  if (tb < te) { load_image(tb, st); store_image(s_stream[0], st); }
  lds_barrier();   // histogram cleared, first image in place
  int buf = 0;
  for (uint64_t t = tb; t < te; t += RMAX) {
    const bool more = t + RMAX < te;
    if (more) load_image(t + RMAX, st);   // in flight while this round is processed
    uint32_t eo[RMAX + 1];                // entries of the round's tiles, cumulative
    eo[0] = 0;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) eo[i + 1] = eo[i] + ((t + i < te) ? ent_cnt[t + i] : 0u);
    const uint32_t total = eo[RMAX];
    const uint32_t *img = s_stream[buf];
    for (uint32_t e0 = 0; e0 < total; e0 += NT * EB) {
      uint32_t ev[EB], er[EB];
#pragma unroll
      for (int m = 0; m < EB; ++m) {
        uint32_t e = e0 + m * NT + threadIdx.x;
        e = e < total ? e : total - 1;
        uint32_t r = 0;
#pragma unroll
        for (int i = 1; i < RMAX; ++i) r += (e >= eo[i]) ? 1u : 0u;
        uint32_t base = 0;
#pragma unroll
        for (int i = 1; i < RMAX; ++i) base = (r == (uint32_t)i) ? eo[i] : base;
        er[m] = r;
        ev[m] = ent[(t + r) * ent_stride + (e - base)];
      }
#pragma unroll
      for (int m = 0; m < EB; ++m) {
        if (e0 + m * NT + threadIdx.x < total) {
          for_entry_windows<NW, BITS>(img, er[m] * Cfg::TILE + (ev[m] & 0x1fffu), (ev[m] >> 13) + 1u, shape, canonical,
                                      [&](uint32_t, const uint64_t (&key)[NW]) { atomicAdd(&s_hist[fine_of(place_hash<NW>(key))], 1u); });
        }
      }
    }
    if (more) store_image(s_stream[buf ^ 1], st);   // last read two barriers ago
    lds_barrier();
    buf ^= 1;
  }
  uint32_t *part_hist = fine_hist + (uint64_t)(blockIdx.x / (gridDim.x / kFineParts)) * kNumFine;
  for (int i = threadIdx.x; i < kNumFine; i += NT) {
    uint32_t v = s_hist[i];
    if (v) atomicAdd(&part_hist[i], v);
  }
  for (int c = threadIdx.x; c < kNumCoarse; c += NT) {
    uint32_t sum = 0;
    for (int i = 0; i < kSubPerCoarse; ++i) sum += s_hist[c * kSubPerCoarse + ((i + c) & (kSubPerCoarse - 1))];
    wg_hist[(uint64_t)blockIdx.x * kNumCoarse + c] = sum;
  }
}

// Rank counts from the entry list (the counting half of imxx::distribute fused with read_file): per-workgroup counts
// of the rank buckets, same tile ownership as the scatter that follows. The rank bucket of every window is kept (one
// byte per window, eight per entry) so that the rank hash (Murmur / Farm) is computed once.
template <int NW, int BITS, int TPB>
__global__ __launch_bounds__(TPB) void fastq_rank_hist_list_kernel(PackedInput in, uint64_t n_tiles, KShape shape, bool canonical,
                                                                  const uint16_t *__restrict__ ent, const uint32_t *__restrict__ ent_cnt,
                                                                  uint32_t ent_stride, BucketFn fn, uint32_t *__restrict__ wg_hist,
                                                                  uint64_t *__restrict__ ent_bkt /* rank bucket of window j in byte j */) {
  using Cfg = ExCfg<NW, BITS>;
  using L = ListCfg<NW, BITS>;
  constexpr int NT = TPB, RMAX = L::RMAX;
  constexpr int UL = (L::UNITS + NT - 1) / NT;
  constexpr int EB = 2;
  __shared__ uint32_t s_hist[kNumCoarse];
  __shared__ uint32_t s_stream[2][L::STREAM_DW];
  if (threadIdx.x < kNumCoarse) s_hist[threadIdx.x] = 0;
  const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t tb = (uint64_t)blockIdx.x * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  const uint64_t last_unit = in.n_cover / Cfg::C - 1;
  auto load_image = [&](uint64_t t, uint64_t (&st)[UL]) {
#pragma unroll
    for (int i = 0; i < UL; ++i) {
      uint64_t g = t * Cfg::NT + (uint64_t)i * NT + threadIdx.x;
      g = g < last_unit ? g : last_unit;
      st[i] = read_stream_unit<BITS, Cfg::C>(in.stream, g);
    }
  };
  auto store_image = [&](uint32_t *img, const uint64_t (&st)[UL]) {
#pragma unroll
    for (int i = 0; i < UL; ++i) {
      const int u = i * NT + threadIdx.x;
      if (u < L::UNITS) store_stream_bits<BITS, Cfg::C>(img, u, st[i]);
    }
  };
  uint64_t st[UL];
  if (tb < te) { load_image(tb, st); store_image(s_stream[0], st); }
  lds_barrier();
  int buf = 0;
  for (uint64_t t = tb; t < te; t += RMAX) {
    const bool more = t + RMAX < te;
    if (more) load_image(t + RMAX, st);
    uint32_t eo[RMAX + 1];
    eo[0] = 0;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) eo[i + 1] = eo[i] + ((t + i < te) ? ent_cnt[t + i] : 0u);
    const uint32_t total = eo[RMAX];
    const uint32_t *img = s_stream[buf];
    for (uint32_t e0 = 0; e0 < total; e0 += NT * EB) {
      uint32_t ev[EB], er[EB];
      uint64_t eidx[EB];
#pragma unroll
      for (int m = 0; m < EB; ++m) {
        uint32_t e = e0 + m * NT + threadIdx.x;
        e = e < total ? e : total - 1;
        uint32_t r = 0;
#pragma unroll
        for (int i = 1; i < RMAX; ++i) r += (e >= eo[i]) ? 1u : 0u;
        uint32_t base = 0;
#pragma unroll
        for (int i = 1; i < RMAX; ++i) base = (r == (uint32_t)i) ? eo[i] : base;
        er[m] = r;
        eidx[m] = (t + r) * ent_stride + (e - base);
        ev[m] = ent[eidx[m]];
      }
#pragma unroll
      for (int m = 0; m < EB; ++m) {
        if (e0 + m * NT + threadIdx.x < total) {
          uint64_t bk = 0;
          for_entry_windows<NW, BITS>(img, er[m] * Cfg::TILE + (ev[m] & 0x1fffu), (ev[m] >> 13) + 1u, shape, canonical,
                                      [&](uint32_t j, const uint64_t (&key)[NW]) {
                                        const uint32_t b = bucket_of<NW>(key, fn);
                                        atomicAdd(&s_hist[b], 1u);
                                        bk |= (uint64_t)b << (8u * j);
                                      });
          ent_bkt[eidx[m]] = bk;
        }
      }
    }
    if (more) store_image(s_stream[buf ^ 1], st);
    lds_barrier();
    buf ^= 1;
  }
  lds_barrier();
  if (threadIdx.x < kNumCoarse) wg_hist[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] = s_hist[threadIdx.x];
}

// E2 from the entry list. A round is a run of up to EMAX = CAPW / 8 consecutive entries of this workgroup's tiles (it
// may start and end inside a tile, and spans at most RMAX scan tiles, whose packed stream is one contiguous LDS
// image), so the bucket sort works on a nearly full stage whatever the tile boundaries; an entry is up to 8 consecutive
// windows of a read: the first k-mer is read from the image, the others roll. No per-byte work at all.
// RANK = true: buckets come from the bytes the rank histogram pass left (ent_bkt) instead of the placement hash.
template <int NW, int BITS, bool RANK = false>
__global__ __launch_bounds__((ExCfg<NW, BITS>::NT), (NW == 1 ? 4 : 1)) void fastq_scatter_list_kernel(PackedInput in, uint64_t n_tiles, KShape shape, bool canonical,
                                                                                  const uint16_t *__restrict__ ent,
                                                                                  const uint32_t *__restrict__ ent_cnt, uint32_t ent_stride,
                                                                                  const uint64_t *__restrict__ ent_bkt,
                                                                                  const uint64_t *__restrict__ wg_off, uint64_t *__restrict__ out) {
  using Cfg = ExCfg<NW, BITS>;
  using L = ListCfg<NW, BITS>;
  constexpr int NT = L::NT, CAPW = L::CAPW, RMAX = L::RMAX;
  constexpr int EMAX = CAPW / 8;                     // entries per round: every entry holds at most 8 windows
  constexpr int EPT = (EMAX + NT - 1) / NT;          // entries per thread and round
  constexpr int KPT = EPT * 8;                       // keys per thread and round
  static_assert(NT >= kNumCoarse, "one thread per coarse bucket");
  __shared__ uint64_t s_stage[CAPW * NW];
  __shared__ uint8_t s_bkt[CAPW];
  __shared__ uint32_t s_stream[L::STREAM_DW];
  __shared__ uint32_t s_cnt[kNumCoarse];
  __shared__ uint32_t s_lofs[kNumCoarse];
  __shared__ uint64_t s_gbase[kNumCoarse];
  __shared__ uint32_t s_part[kNumCoarse / kWave];
  __shared__ uint32_t s_total;
  uint64_t cursor = (threadIdx.x < kNumCoarse) ? wg_off[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] : 0ull;
  if (threadIdx.x < kNumCoarse) s_cnt[threadIdx.x] = 0;
  const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t tb = (uint64_t)blockIdx.x * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  if (tb >= te) return;
  const uint64_t last_unit = in.n_cover / Cfg::C - 1;
  // a round: entries ei .. of tile t, then whole tiles, up to EMAX entries / RMAX tiles; eo[i] = rank of tile t+i's first entry
  struct Round { uint64_t t; uint32_t ei, total; uint32_t eo[RMAX + 1]; uint64_t nt; uint32_t nei; };
  auto plan = [&](uint64_t t, uint32_t ei, Round &r) -> bool {
    while (t < te && ei >= ent_cnt[t]) { ++t; ei = 0; }   // tiles that are used up or hold no window are skipped
    if (t >= te) return false;
    r.t = t; r.ei = ei; r.eo[0] = 0;
    r.nt = t + RMAX; r.nei = 0;                            // where the next round starts if every tile is used up
    bool cut = false;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      uint32_t avail = (t + i < te) ? ent_cnt[t + i] : 0u;
      if (i == 0) avail -= ei;
      uint32_t take = avail;
      if (r.eo[i] + take > (uint32_t)EMAX) take = (uint32_t)EMAX - r.eo[i];
      if (cut) take = 0;
      if (!cut && take < avail) { cut = true; r.nt = t + i; r.nei = (i == 0 ? ei : 0u) + take; }
      r.eo[i + 1] = r.eo[i] + take;
    }
    r.total = r.eo[RMAX];
    return true;
  };
  uint64_t st[L::ULOADS];
  uint32_t ev[EPT], er[EPT];
  uint64_t eb[EPT];
  auto fetch = [&](const Round &r) {   // clamped loads, not guarded ones, so nothing forces an early wait
#pragma unroll
    for (int i = 0; i < L::ULOADS; ++i) {
      uint64_t g = r.t * NT + (uint64_t)i * NT + threadIdx.x;
      g = g < last_unit ? g : last_unit;
      st[i] = read_stream_unit<BITS, Cfg::C>(in.stream, g);
    }
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
      uint32_t e = m * NT + threadIdx.x;
      e = e < r.total ? e : r.total - 1;
      uint32_t tr = 0;
#pragma unroll
      for (int i = 1; i < RMAX; ++i) tr += (e >= r.eo[i]) ? 1u : 0u;
      uint32_t base = 0;
#pragma unroll
      for (int i = 1; i < RMAX; ++i) base = (tr == (uint32_t)i) ? r.eo[i] : base;
      const uint64_t idx = (r.t + tr) * ent_stride + (tr == 0u ? r.ei : 0u) + (e - base);
      er[m] = tr;
      ev[m] = ent[idx];
      if (RANK) eb[m] = ent_bkt[idx];
    }
  };
  Round cur, nxt;
  bool have = plan(tb, 0u, cur);
  if (have) fetch(cur);
  while (have) {
#pragma unroll
    for (int i = 0; i < L::ULOADS; ++i) {
      const int u = i * NT + threadIdx.x;
      if (u < L::UNITS) store_stream_bits<BITS, Cfg::C>(s_stream, u, st[i]);
    }
    lds_barrier();
    // S0: keys of this thread's entries, coarse bucket and rank inside (round, bucket)
    uint64_t key[KPT][NW];
    // rank inside (round, bucket) as the LDS atomic returns it -- kept raw, so that nothing waits for the atomic here --
    // and the bucket bytes, four to a register
    uint32_t rnk[KPT], bpk[KPT / 4];
#pragma unroll
    for (int q = 0; q < KPT; ++q) rnk[q] = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < KPT / 4; ++q) bpk[q] = 0;
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
      if ((uint32_t)(m * NT) + threadIdx.x < cur.total) {
        for_entry_windows<NW, BITS>(s_stream, er[m] * Cfg::TILE + (ev[m] & 0x1fffu), (ev[m] >> 13) + 1u, shape, canonical,
                                    [&](uint32_t j, const uint64_t (&kk)[NW]) {
#pragma unroll
                                      for (int w = 0; w < NW; ++w) key[m * 8 + j][w] = kk[w];
                                      const uint32_t b = RANK ? (uint32_t)((eb[m] >> (8u * j)) & 0xffu) : coarse_of(place_hash<NW>(kk));
                                      rnk[m * 8 + j] = atomicAdd(&s_cnt[b], 1u);
                                      bpk[(m * 8 + j) / 4] |= b << (8u * ((m * 8 + j) % 4u));
                                    });
      }
    }
    // the next round's inputs travel while this one is sorted and written out
    have = plan(cur.nt, cur.nei, nxt);
    if (have) fetch(nxt);
    lds_barrier();
    uint32_t c = 0, inc = 0;
    if (threadIdx.x < kNumCoarse) {
      c = s_cnt[threadIdx.x];
      s_cnt[threadIdx.x] = 0;
      inc = wave_inclusive_scan(c);
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
    }
    lds_barrier();
    if (threadIdx.x < kNumCoarse) {
      uint32_t pre = 0;
#pragma unroll
      for (uint32_t w = 0; w < kNumCoarse / kWave; ++w) pre += (w < wave_id()) ? s_part[w] : 0u;
      const uint32_t lo = pre + inc - c;
      s_lofs[threadIdx.x] = lo;
      s_gbase[threadIdx.x] = cursor - lo;
      cursor += c;
      if (threadIdx.x == kNumCoarse - 1) s_total = pre + inc;   // windows of the round
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      if (rnk[q] != 0xffffffffu) {
        const uint32_t b = (bpk[q / 4] >> (8u * (q % 4))) & 0xffu;
        const uint32_t pos = s_lofs[b] + rnk[q];
#pragma unroll
        for (int w = 0; w < NW; ++w) s_stage[(uint64_t)pos * NW + w] = key[q][w];
        s_bkt[pos] = (uint8_t)b;
      }
    }
    lds_barrier();
    const uint32_t total = s_total;
    for (uint32_t s = threadIdx.x; s < total; s += NT) {
      const uint64_t dst = s_gbase[s_bkt[s]] + s;
#pragma unroll
      for (int w = 0; w < NW; ++w) out[dst * NW + w] = s_stage[(uint64_t)s * NW + w];
    }
    lds_barrier();   // the next round overwrites the stream image and the stage
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------
// LDS table
// ---------------------------------------------------------------------------
template <int NW> struct TabCfg {
  // home slots. One-word keys: one 1024-thread workgroup per CU with a table that fills the CU's LDS (154 KB). Two
  // 512-thread workgroups with half the table each ran at the same speed on config 2 (4.25 ms: the kernel is bound by
  // the dependent LDS round trips of the probe walk, not by the load factor), and the large table takes twice the
  // distinct keys per pass (9120 per bucket, 3.0e8 per index) before a bucket needs a second pass.
  // (12160 + 64 slots x 12 B + 16 KB of per-wave miss queues = 160 KB)
  static constexpr int CAP = (NW == 1) ? 12160 : (NW == 2 ? 6144 : (NW == 3 ? 4608 : 3712));   // home slots (tagged tables: 24 / 32 / 40 B per slot, 144 - 147 KB; a
                                                                                              // bucket of config 2 at k = 63 holds 3050 distinct keys: 4608 slots ran it at a load of 0.66)
  // One-word tables probe linearly WITHOUT wrap-around: a probe sequence that starts near the end runs on
  // into PAD extra slots, so a probe step is "next address, read, compare" and nothing else. The very last
  // slot is never filled (an insert that would need it reports overflow), which ends every probe sequence.
  static constexpr int PAD = (NW == 1) ? 64 : 0;
  static constexpr int SLOTS = CAP + PAD;
  static constexpr int LIMIT = CAP * 3 / 4;      // distinct keys per pass before the bucket is split into more passes
  static constexpr int NT = (NW == 1) ? 1024 : 512;
};
constexpr int kMaxProbe = 192;   // longer probe sequences than this mean the LDS table is overloaded
constexpr uint32_t kMaxPasses = 1u << 16;
constexpr uint64_t kEmptyKey = ~0ull;

// one LDS atomic per wavefront: every lane that `want`s a slot gets a distinct index
__device__ __forceinline__ uint32_t wave_alloc(uint32_t *ctr, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return 0u;
  const int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if ((int)lane_id() == leader) base = atomicAdd(ctr, (uint32_t)__popcll(m));
  base = __shfl(base, leader, kWave);
  return base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
}
// count += w for a counting map: std::plus<uint32_t> wraps; sat_plus (distributed_densehash_map.hpp:2903-2912) stops at the type's
// maximum. An add that wraps (old > ~w) puts the ceiling back; every later add to that counter wraps again and does the same, so
// once a key's total has passed 2^32 - 1 the counter ends on the ceiling whatever the interleaving of the lanes.
__device__ __forceinline__ void count_add(uint32_t *p, uint32_t w, bool sat) {
  if (!sat) { atomicAdd(p, w); return; }
  const uint32_t old = atomicAdd(p, w);
  if (old > ~w) atomicMax(p, 0xffffffffu);
}
constexpr uint32_t kTagEmpty = 0u, kTagLock = 1u;

template <int NW> struct LdsTable {
  uint64_t *keys;      // [SLOTS*NW]
  uint32_t *vals;      // [SLOTS]
  uint32_t *tags;      // [SLOTS] (NW > 1 only)
  uint32_t *distinct;  // counter
  uint32_t *overflow;  // flag
  uint32_t *special;   // value of the key that equals the empty sentinel (NW == 1)
  uint32_t *special_set;
  uint32_t *progress;  // keys of the stream the insert had scanned when the table overflowed (pass-count estimate)
  // geometry in use: TabCfg's by default; the query kernel sizes it to the bucket's query count (a table that is cleared
  // and swept once per bucket should not be larger than its keys need)
  uint32_t cap, slots, limit;
};

template <int NW> __device__ __forceinline__ void table_clear(const LdsTable<NW> &t) {
  for (uint32_t i = threadIdx.x; i < t.slots; i += blockDim.x) {
    t.vals[i] = 0;
    if (NW == 1) t.keys[i] = kEmptyKey; else t.tags[i] = kTagEmpty;
  }
  if (threadIdx.x == 0) { *t.distinct = 0; *t.overflow = 0; *t.special = 0; *t.special_set = 0; *t.progress = 0; }
}

__device__ __forceinline__ uint32_t slot_of(uint32_t h, int cap) {
  return (uint32_t)(((uint64_t)(h & ((1u << kSlotBits) - 1u)) * (uint32_t)cap) >> kSlotBits);
}

// find-or-insert; returns slot (or -1 when the table overflowed / special key). `inserted` tells a new key.
template <int NW> __device__ __forceinline__ int table_upsert(const LdsTable<NW> &t, const uint64_t (&key)[NW], uint32_t h) {
  const uint32_t CAP = t.cap;
  uint32_t slot = slot_of(h, (int)CAP);
  if (NW == 1) {
    if (key[0] == kEmptyKey) { *t.special_set = 1; return -2; }
    for (int probes = 0; slot < t.slots - 1u; ++probes, ++slot) {
      unsigned long long old = atomicCAS((unsigned long long *)&t.keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key[0]);
      if (old == kEmptyKey) {
        if (atomicAdd(t.distinct, 1u) >= t.limit || probes >= kMaxProbe) *t.overflow = 1;   // too loaded: more passes
        return (int)slot;
      }
      if (old == key[0]) return (int)slot;
    }
    *t.overflow = 1;
    return -1;
  } else {
    const uint32_t tagv = h | 0x80000000u;
    int probes = 0;
    while (probes < (int)CAP) {
      uint32_t tg = __atomic_load_n(&t.tags[slot], __ATOMIC_RELAXED);
      if (tg == kTagEmpty) {
        if (__atomic_load_n(t.overflow, __ATOMIC_RELAXED)) return -1;   // the pass is lost already: no further claims
        uint32_t old = atomicCAS(&t.tags[slot], kTagEmpty, kTagLock);
        if (old == kTagEmpty) {
#pragma unroll
          for (int w = 0; w < NW; ++w) t.keys[(uint64_t)slot * NW + w] = key[w];
          __threadfence_block();
          atomicExch(&t.tags[slot], tagv);
          // (the load limit holds for tagged tables too: without it a bucket of more distinct keys than slots filled its table to the
          // last slot, and every key of the stream then walked all of it before the pass was given up -- O(keys x slots) per attempt:
          // 11.9 s for config 2's reads over an 800 Mbp genome at k = 63)
          if (atomicAdd(t.distinct, 1u) >= t.limit || probes >= kMaxProbe) *t.overflow = 1;
          return (int)slot;
        }
        continue;  // somebody else took it: look again
      }
      if (tg == kTagLock) continue;  // being written by another lane
      if (tg == tagv) {
        bool eq = true;
#pragma unroll
        for (int w = 0; w < NW; ++w) eq &= (t.keys[(uint64_t)slot * NW + w] == key[w]);
        if (eq) { if (probes >= kMaxProbe) *t.overflow = 1; return (int)slot; }
      }
      slot = (slot + 1 == CAP) ? 0u : slot + 1;
      ++probes;
    }
    *t.overflow = 1;
    return -1;
  }
}

// fit check of a first attempt (CHECK): a bucket of exactly LIMIT = 9120 distinct keys shows 5406 +- 29 distinct ones after
// its first 8192 draws (D (1 - e^(-n/D))); more than that (+ 2.4 sigma) and the bucket will not fit one table
constexpr uint32_t kFitsStep8192 = 5475u;
static_assert(TabCfg<1>::LIMIT == 9120, "kFitsStep8192 belongs to this limit");

// which reduction pass a key belongs to when a bucket needs several
__device__ __forceinline__ uint32_t pass_of(uint32_t h, uint32_t npass) {
  return npass == 1 ? 0u : (uint32_t)(((uint64_t)(h * 0x9E3779B1u) * npass) >> 32);
}

// NW == 1 insert of keys[b, e) with weight 1, U keys per thread and batch. The U home slots are read with
// plain 64-bit loads issued back to back; a key seen before (the common case at sequencing coverage) then
// costs that read plus one 32-bit add. Only a key whose home slot holds another key walks on (next address,
// read, compare), and a compare-and-swap is spent only on an empty slot, i.e. once per distinct key.
// The walk is the expensive part (every step is a full wave instruction sequence for the few lanes that
// still probe), hence no wrap-around, no probe counter and no bounds test inside it (see TabCfg).
// SPECIAL: the keys can equal the empty marker (only k-mers that fill all 64 bits can); otherwise that test is left out
// SPILL (with MULTI): the keys of later passes are written, compacted, to `spill` (spill[0], spill[dir], spill[2 dir] ...;
// *spill_cnt counts them), so the next pass reads only what is left instead of filtering the whole stream again
// CHECK (first attempt at a bucket with nothing to merge): after the first load step the workgroup meets once, and if the
// distinct keys seen so far say the bucket will not fit one table (occupancy: d of D keys after n draws = D (1 - e^(-n/D))),
// the attempt ends there with the pass count to use in *hint, instead of filling the table before it fails.
template <int U, bool MULTI, bool SPECIAL = true, bool SPILL = false, bool CHECK = false>
__device__ __forceinline__ void table_insert_stream1(const LdsTable<1> &t, const uint64_t *__restrict__ keys /* bucket base */, uint32_t n,
                                                     uint32_t npass, uint32_t pass, uint64_t *spill = nullptr, int spill_dir = 1,
                                                     uint32_t *spill_cnt = nullptr, uint32_t *hint = nullptr) {
  constexpr int CAP = TabCfg<1>::CAP;
  constexpr uint32_t LAST = TabCfg<1>::SLOTS - 1;
  constexpr uint32_t NT = TabCfg<1>::NT;
  constexpr uint32_t STEP = NT * U;
  if (n == 0) return;
  const uint32_t last = n - 1;
  // 32-bit indices relative to the bucket, clamped (not guarded) loads, two register sets in ping-pong:
  // the batch after the one being inserted is always in flight and nothing is copied between them
  auto load = [&](uint32_t i0, uint64_t (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t i = i0 + (uint32_t)u * NT + threadIdx.x;
      i = i < last ? i : last;
      r[u] = keys[i];
    }
  };
  auto insert = [&](uint32_t i0, const uint64_t (&k)[U], auto full_tag) {   // full_tag: every key of the batch lies inside the bucket
    constexpr bool FULL = decltype(full_tag)::value;
    uint32_t slot[U], actm = 0, spec = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t kk[1] = {k[u]};
      const uint32_t h = place_hash<1>(kk);
      slot[u] = slot_of(h, CAP);
      bool a = FULL || i0 + (uint32_t)u * NT + threadIdx.x < n;
      if (MULTI) {
        const bool mine = pass_of(h, npass) == pass;
        if (SPILL) {   // wave-uniform call: one LDS add per wavefront
          const bool later = a && !mine;
          const uint32_t pos = wave_alloc(spill_cnt, later);
          if (later) spill[(long long)pos * spill_dir] = k[u];
        }
        a = a && mine;
      }
      const bool sp = SPECIAL && k[u] == kEmptyKey;
      actm |= (a && !sp) ? (1u << u) : 0u;
      spec += (a && sp) ? 1u : 0u;
    }
    if (SPECIAL && spec) { *t.special_set = 1; atomicAdd(t.special, spec); }   // the key that equals the empty marker
    uint64_t cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = __atomic_load_n(&t.keys[slot[u]], __ATOMIC_RELAXED);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (actm & (1u << u)) {
        uint32_t s = slot[u];
        uint64_t c = cur[u];
        for (;;) {
          while (c != k[u] && c != kEmptyKey) { ++s; c = __atomic_load_n(&t.keys[s], __ATOMIC_RELAXED); }   // the walk
          if (c == k[u]) break;
          // empty slot: first sighting of this key (or the table is overloaded)
          if (s >= LAST || __atomic_load_n(t.distinct, __ATOMIC_RELAXED) >= (uint32_t)TabCfg<1>::LIMIT) { *t.overflow = 1; s = LAST; break; }
          const unsigned long long old = atomicCAS((unsigned long long *)&t.keys[s], (unsigned long long)kEmptyKey, (unsigned long long)k[u]);
          if (old == kEmptyKey) {   // count the new key, one LDS add per wavefront
            const unsigned long long m = __ballot(1);
            if ((int)lane_id() == __ffsll((long long)m) - 1) atomicAdd(t.distinct, (uint32_t)__popcll(m));
            break;
          }
          c = old;   // lost the slot to another lane: its key is ours (done) or the walk goes on
        }
        atomicAdd(&t.vals[s], 1u);   // (slot LAST never holds a key: counts parked there on overflow are never read)
      }
    }
  };
  uint64_t ka[U], kb[U];
  load(0u, ka);
  for (uint32_t i0 = 0; i0 < n; i0 += 2 * STEP) {
    // (with CHECK every wave must reach the two barriers of the first step, whatever another wave has flagged by then)
    if (!(CHECK && i0 == 0u) && __atomic_load_n(t.overflow, __ATOMIC_RELAXED)) {   // this pass is lost already: stop filling the table
      if (lane_id() == 0) atomicMax(t.progress, i0);
      break;
    }
    const bool has_b = i0 + STEP < n;
    if (has_b) load(i0 + STEP, kb);
    if (i0 + STEP <= n) insert(i0, ka, std::true_type{}); else insert(i0, ka, std::false_type{});
    if (CHECK && i0 == 0u && n > 2u * STEP) {   // uniform; every wave is here (nothing sets the overflow flag before the first step is in)
      constexpr uint32_t kFits = kFitsStep8192;
      static_assert(STEP == 8192u, "kFits belongs to this step and limit");
      lds_barrier();
      if (threadIdx.x == 0) {
        const uint32_t d1 = *t.distinct;
        if (d1 >= kFits) {
          const float r = (float)d1 / (float)STEP;           // (1 - e^-x) / x with x = STEP / D
          float x = fmaxf(2.f * (1.f - r), 1e-4f);
          for (int it = 0; it < 4; ++it) {
            const float e = __expf(-x), g = (1.f - e) / x - r, dg = (e * (x + 1.f) - 1.f) / (x * x);
            x = fmaxf(x - g / dg, 1e-4f);
          }
          const float D = (float)STEP / x;
          // distinct keys among all n (nearly every key of the step new: the inversion loses its footing, and all n are)
          const float dn = r > 0.99f ? (float)n : D * (1.f - __expf(-(float)n / D));
          const float want = ceilf(dn * 1.15f / (float)TabCfg<1>::LIMIT);
          *hint = want < 2.f ? 2u : (want > 65536.f ? 65536u : (uint32_t)want);
        }
      }
      lds_barrier();
      if (*hint) return;   // written only between the two barriers: the same for every lane
    }
    if (has_b) {
      if (i0 + 2 * STEP < n) load(i0 + 2 * STEP, ka);
      if (i0 + 2 * STEP <= n) insert(i0 + STEP, kb, std::true_type{}); else insert(i0 + STEP, kb, std::false_type{});
    }
  }
}

// ---------------------------------------------------------------------------
// Flat insert of one-word keys (weight 1), the common case of bucket_reduce. table_insert_stream1 above walks, claims and
// counts inside one divergent per-key loop, so every wavefront runs the whole slow path for each of its U keys as soon as
// ONE lane needs it -- and one nearly always does: its exec-mask bookkeeping (67 scalar instructions per key against 59
// vector ones, rocprofv3 SQ_INSTS_*) was what bound the kernel, not the table. Here the per-key code is straight-line:
//   fast path  hash, home slot, ONE 64-bit read, compare; a hit (the key was seen before and sits in its home slot: the
//              rule at sequencing coverage) costs one predicated 32-bit LDS add;
//   misses     (first sighting, or the home slot holds another key) go to a 128-entry queue of the WAVEFRONT in LDS
//              (ballot + mbcnt: no atomics), and whenever 64 are waiting the wavefront pops them, one per lane, and runs
//              the probe walk / claim loop with every lane busy.
// MULTI: only the keys of `pass` (pass_of) are taken. SPECIAL: the keys may equal the empty marker (k-mers of 64 bits).
// CHECK: the fit check of a first attempt (see table_insert_stream1); the queue is emptied before the distinct count is read.
// ---------------------------------------------------------------------------
constexpr int kMissQ = 2 * kWave;   // queue entries per wavefront: fewer than 64 wait, at most 64 join per step

// pass count from the distinct keys d1 among the first `step` of a bucket's n keys (occupancy: d of D after s draws = D (1 - e^(-s/D)))
__device__ __forceinline__ uint32_t passes_from_first_step(uint32_t d1, uint32_t step, uint32_t n) {
  const float r = (float)d1 / (float)step;           // (1 - e^-x) / x with x = step / D
  float x = fmaxf(2.f * (1.f - r), 1e-4f);
  for (int it = 0; it < 4; ++it) {
    const float e = __expf(-x), g = (1.f - e) / x - r, dg = (e * (x + 1.f) - 1.f) / (x * x);
    x = fmaxf(x - g / dg, 1e-4f);
  }
  const float D = (float)step / x;
  // distinct keys among all n (nearly every key of the step new: the inversion loses its footing, and all n are)
  const float dn = r > 0.99f ? (float)n : D * (1.f - __expf(-(float)n / D));
  const float want = ceilf(dn * 1.15f / (float)TabCfg<1>::LIMIT);
  return want < 2.f ? 2u : (want > 65536.f ? 65536u : (uint32_t)want);
}

// The slow path as ONE out-of-line function (LDS address-space pointers, so the table accesses stay ds_ instructions):
// table_insert_flat reaches it from 2 x U places, once per 64 misses -- inlined, those copies made the kernel 50 k lines of ISA.
// Pops queue entries [first, first + cnt), cnt <= 64, one per lane: walk from the home slot, claim an empty slot, count.
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
template <int CAP, int SLOTS, int LIMIT>
__device__ __attribute__((noinline)) void probe_insert_lds_cap(lds_u64_t *tkeys, lds_u32_t *tvals, lds_u32_t *distinct, lds_u32_t *overflow,
                                                              const lds_u64_t *wq, uint32_t first, uint32_t cnt) {
  constexpr uint32_t LAST = SLOTS - 1;
  const uint32_t lane = lane_id();
  if (lane < cnt) {
    const uint64_t key = wq[first + lane];
    const uint64_t kk[1] = {key};
    uint32_t s = slot_of(place_hash<1>(kk), CAP);
    uint64_t c = __atomic_load_n(&tkeys[s], __ATOMIC_RELAXED);
    for (;;) {
      while (c != key && c != kEmptyKey) { ++s; c = __atomic_load_n(&tkeys[s], __ATOMIC_RELAXED); }   // the walk
      if (c == key) break;
      // empty slot: first sighting of this key (or the table is overloaded)
      if (s >= LAST || __atomic_load_n(distinct, __ATOMIC_RELAXED) >= (uint32_t)LIMIT) { *overflow = 1; s = LAST; break; }
      uint64_t expected = kEmptyKey;
      if (__atomic_compare_exchange_n(&tkeys[s], &expected, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
        // count the new key, one LDS add per wavefront
        const unsigned long long m = __ballot(1);
        if ((int)lane == __ffsll((long long)m) - 1) __atomic_fetch_add(distinct, (uint32_t)__popcll(m), __ATOMIC_RELAXED);
        break;
      }
      c = expected;   // lost the slot to another lane: its key is ours (done) or the walk goes on
    }
    __atomic_fetch_add(&tvals[s], 1u, __ATOMIC_RELAXED);   // (slot LAST never holds a key: counts parked there on overflow are never read)
  }
}

__device__ __forceinline__ void probe_insert_lds(lds_u64_t *tkeys, lds_u32_t *tvals, lds_u32_t *distinct, lds_u32_t *overflow,
                                                 const lds_u64_t *wq, uint32_t first, uint32_t cnt) {
  probe_insert_lds_cap<TabCfg<1>::CAP, TabCfg<1>::SLOTS, TabCfg<1>::LIMIT>(tkeys, tvals, distinct, overflow, wq, first, cnt);
}

template <int U, bool MULTI, bool SPECIAL, bool CHECK>
__device__ __forceinline__ void table_insert_flat(const LdsTable<1> &t, uint64_t *wq /* this wavefront's queue [kMissQ] */,
                                                  const uint64_t *__restrict__ keys /* bucket base */, uint32_t n, uint32_t npass,
                                                  uint32_t pass, uint32_t *hint = nullptr) {
  constexpr int CAP = TabCfg<1>::CAP;
  constexpr uint32_t LAST = TabCfg<1>::SLOTS - 1;
  constexpr uint32_t NT = TabCfg<1>::NT;
  constexpr uint32_t STEP = NT * U;
  if (n == 0) return;
  const uint32_t last = n - 1;
  const uint32_t lane = lane_id();
  uint32_t qn = 0;   // keys waiting in the queue (the same in every lane)
  lds_u64_t *const tkeys = (lds_u64_t *)t.keys;
  lds_u32_t *const tvals = (lds_u32_t *)t.vals;
  lds_u32_t *const tdist = (lds_u32_t *)t.distinct;
  lds_u32_t *const tovf = (lds_u32_t *)t.overflow;
  const lds_u64_t *const wql = (const lds_u64_t *)wq;
  auto drain = [&](uint32_t cnt) {   // the top cnt <= 64 entries of the queue, one per lane
    probe_insert_lds(tkeys, tvals, tdist, tovf, wql, qn - cnt, cnt);
    qn -= cnt;
  };
  auto load = [&](uint32_t i0, uint64_t (&r)[U]) {   // clamped (not guarded) loads, indices relative to the bucket
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t i = i0 + (uint32_t)u * NT + threadIdx.x;
      i = i < last ? i : last;
      r[u] = keys[i];
    }
  };
  auto insert = [&](uint32_t i0, const uint64_t (&k)[U], auto full_tag) {   // full_tag: every key of the batch lies inside the bucket
    constexpr bool FULL = decltype(full_tag)::value;
    uint32_t slot[U], valid = 0, spec = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t kk[1] = {k[u]};
      const uint32_t h = place_hash<1>(kk);
      slot[u] = slot_of(h, CAP);
      bool a = FULL || i0 + (uint32_t)u * NT + threadIdx.x < n;
      if (MULTI) a = a && pass_of(h, npass) == pass;
      const bool sp = SPECIAL && k[u] == kEmptyKey;
      valid |= (a && !sp) ? (1u << u) : 0u;
      spec += (a && sp) ? 1u : 0u;
    }
    if (SPECIAL && spec) { *t.special_set = 1; atomicAdd(t.special, spec); }   // the key that equals the empty marker
    uint64_t cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = __atomic_load_n(&t.keys[slot[u]], __ATOMIC_RELAXED);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool v = (valid >> u) & 1u;
      bool hit = v && cur[u] == k[u];
      if (v && cur[u] == kEmptyKey) {   // first sighting with a free home slot: claimed here, in line (with little duplication most keys are)
        const unsigned long long old = atomicCAS((unsigned long long *)&t.keys[slot[u]], (unsigned long long)kEmptyKey, (unsigned long long)k[u]);
        if (old == kEmptyKey) {
          hit = true;
          const unsigned long long cm = __ballot(1);
          if ((int)lane == __ffsll((long long)cm) - 1) atomicAdd(t.distinct, (uint32_t)__popcll(cm));
        } else hit = old == k[u];
      }
      if (hit) atomicAdd(&t.vals[slot[u]], 1u);
      const bool miss = v && !hit;
      const unsigned long long m = __ballot(miss);
      if (m) {   // uniform
        const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (miss) wq[pos] = k[u];
        qn += (uint32_t)__popcll(m);
        if (qn >= (uint32_t)kWave) drain((uint32_t)kWave);
      }
    }
  };
  uint64_t ka[U], kb[U];
  load(0u, ka);
  for (uint32_t i0 = 0; i0 < n; i0 += 2 * STEP) {
    // (with CHECK every wave must reach the two barriers of the first step, whatever another wave has flagged by then)
    if (!(CHECK && i0 == 0u) && __atomic_load_n(t.overflow, __ATOMIC_RELAXED)) {   // this pass is lost already: stop filling the table
      if (lane == 0) atomicMax(t.progress, i0);
      qn = 0;
      break;
    }
    const bool has_b = i0 + STEP < n;
    if (has_b) load(i0 + STEP, kb);
    if (i0 + STEP <= n) insert(i0, ka, std::true_type{}); else insert(i0, ka, std::false_type{});
    if (CHECK && i0 == 0u && n > 2u * STEP) {   // uniform; every wave is here (only this step's keys can have set the overflow flag)
      static_assert(!CHECK || STEP == 8192u, "kFitsStep8192 belongs to this step");
      if (qn) drain(qn);                        // the distinct count has to cover the whole step
      lds_barrier();
      if (threadIdx.x == 0) {
        const uint32_t d1 = *t.distinct;
        if (d1 >= kFitsStep8192) *hint = passes_from_first_step(d1, STEP, n);
      }
      lds_barrier();
      if (*hint) return;   // written only between the two barriers: the same for every lane
    }
    if (has_b) {
      if (i0 + 2 * STEP < n) load(i0 + 2 * STEP, ka);
      if (i0 + 2 * STEP <= n) insert(i0 + STEP, kb, std::true_type{}); else insert(i0 + STEP, kb, std::false_type{});
    }
  }
  if (qn) drain(qn);
}

// lookup only; returns slot or -1
template <int NW> __device__ __forceinline__ int table_find(const LdsTable<NW> &t, const uint64_t (&key)[NW], uint32_t h) {
  const uint32_t CAP = t.cap;
  uint32_t slot = slot_of(h, (int)CAP);
  if (NW == 1) {
    if (key[0] == kEmptyKey) return *t.special_set ? -2 : -1;
    for (;; ++slot) {   // ends at the latest on the never-filled last slot
      uint64_t k = t.keys[slot];
      if (k == key[0]) return (int)slot;
      if (k == kEmptyKey) return -1;
    }
  } else {
    const uint32_t tagv = h | 0x80000000u;
    for (uint32_t probes = 0; probes < CAP; ++probes) {
      uint32_t tg = t.tags[slot];
      if (tg == kTagEmpty) return -1;
      if (tg == tagv) {
        bool eq = true;
#pragma unroll
        for (int w = 0; w < NW; ++w) eq &= (t.keys[(uint64_t)slot * NW + w] == key[w]);
        if (eq) return (int)slot;
      }
      slot = (slot + 1 == CAP) ? 0u : slot + 1;
    }
    return -1;
  }
}

template <int NW> __device__ __forceinline__ bool slot_used(const LdsTable<NW> &t, int slot) {
  return NW == 1 ? (t.keys[slot] != kEmptyKey) : (t.tags[slot] != kTagEmpty);
}

// visit keys[b, e) with U independent loads per thread in flight; f(key words, index). The loads of the NEXT batch are issued before
// the current one is worked on: a batch's work is a chain of LDS round trips per key, and the tagged tables leave a CU eight
// wavefronts -- with the loads issued only when the previous batch was done, every batch began with a full trip to HBM that nothing
// else could hide (the k = 63 reduce of config 2's reads ran at 39 G keys/s)
template <int NW, int U, typename F>
__device__ __forceinline__ void for_each_key(const uint64_t *__restrict__ keys, uint64_t b, uint64_t e, F f) {
  uint64_t nxt[U][NW];
  auto load = [&](uint64_t i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t i = i0 + (uint64_t)u * blockDim.x + threadIdx.x;
      i = i < e ? i : e - 1;   // (clamped, not guarded: nothing forces an early wait; b < e here)
#pragma unroll
      for (int w = 0; w < NW; ++w) nxt[u][w] = keys[i * NW + w];
    }
  };
  if (b >= e) return;
  load(b);
  for (uint64_t i0 = b; i0 < e; i0 += (uint64_t)blockDim.x * U) {
    uint64_t raw[U][NW];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < NW; ++w) raw[u][w] = nxt[u][w];
    if (i0 + (uint64_t)blockDim.x * U < e) load(i0 + (uint64_t)blockDim.x * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t i = i0 + (uint64_t)u * blockDim.x + threadIdx.x;
      if (i < e) f(raw[u], i);
    }
  }
}
template <int NW> struct BatchOf { static constexpr int U = (NW == 1) ? kLoadBatch : (NW == 2 ? 4 : 2); };

// the query kernel's table: it holds a bucket's distinct QUERY keys (a few hundred for the benchmark's 10 M queries), is
// cleared and swept once per bucket, and 32768 short-lived workgroups are launched, so it keeps the smaller geometry
// (two 512-thread workgroups per CU) that the reduce table had before it grew
template <int NW> struct QTabCfg {
  static constexpr int CAP = (NW == 1) ? 6656 : TabCfg<NW>::CAP;
  static constexpr int PAD = TabCfg<NW>::PAD;
  static constexpr int SLOTS = CAP + PAD;
  static constexpr int LIMIT = CAP * 3 / 4;
  static constexpr int NT = 512;
};

#define KMI_TABLE_LDS_CFG(NW, CFG)                                          \
  __shared__ uint64_t s_tk[CFG::SLOTS * NW];                                \
  __shared__ uint32_t s_tv[CFG::SLOTS];                                     \
  __shared__ uint32_t s_tt[(NW == 1) ? 1 : CFG::SLOTS];                    \
  __shared__ uint32_t s_ctl[8];                                             \
  LdsTable<NW> tab;                                                         \
  tab.keys = s_tk; tab.vals = s_tv; tab.tags = s_tt; tab.distinct = &s_ctl[0]; tab.overflow = &s_ctl[1]; \
  tab.special = &s_ctl[2]; tab.special_set = &s_ctl[3]; tab.progress = &s_ctl[5]; \
  tab.cap = CFG::CAP; tab.slots = CFG::SLOTS; tab.limit = CFG::LIMIT;
#define KMI_TABLE_LDS(NW) KMI_TABLE_LDS_CFG(NW, TabCfg<NW>)

// ---------------------------------------------------------------------------
// C: per fine bucket reduce  (new keys weight 1, old entries weight = their count)
// output to tmp arrays at tmp_off[b] = new_off[b] + old_off[b]; out_cnt[b] = distinct
// ---------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__((TabCfg<NW>::NT)) void bucket_reduce_kernel(const uint64_t *__restrict__ new_keys, const uint64_t *__restrict__ new_off,
                                                                        const uint64_t *__restrict__ old_keys, const uint32_t *__restrict__ old_vals,
                                                                        const uint64_t *__restrict__ old_off, uint64_t *__restrict__ tmp_keys,
                                                                        uint32_t *__restrict__ tmp_vals, uint32_t *__restrict__ out_cnt,
                                                                        uint32_t *__restrict__ flags, bool full_word_keys,
                                                                        uint64_t *scratch /* as large as new_keys and free, or null */, bool sat = false) {
  KMI_TABLE_LDS(NW)
  __shared__ uint64_t s_missq[(NW == 1) ? (TabCfg<NW>::NT / kWave) * kMissQ : 1];   // per-wavefront miss queues (table_insert_flat)
  uint64_t *wq = s_missq + ((NW == 1) ? wave_id() * kMissQ : 0);
  const uint32_t b = blockIdx.x;
  const uint64_t nb = new_off[b], ne = new_off[b + 1];
  const uint64_t ob = old_off ? old_off[b] : 0ull, oe = old_off ? old_off[b + 1] : 0ull;
  if (nb == ne && ob == oe) { if (threadIdx.x == 0) out_cnt[b] = 0; return; }
  const uint64_t tmp0 = nb + ob;
  uint32_t *s_out = &s_ctl[4];
  uint32_t *s_spill = &s_ctl[6];
  uint32_t *s_hint = &s_ctl[7];
  uint32_t npass = 1;
  // The fit check of the first attempt (CHECK) costs two workgroup barriers per bucket, 7 % of this kernel on input
  // that never needs it, so it is switched on by the first bucket of the launch that overflows (flags[8], cleared by
  // the host before the launch): sequencing data at coverage never pays it, a genome or a thin sample pays one lost
  // attempt per CU and then checks.
  // Other workgroups flip flags[8] while this one runs, and the waves of a workgroup start at different times: the flag is
  // read ONCE, by one lane, and handed to the others through LDS behind a barrier, because it selects between code paths
  // that hold different numbers of workgroup barriers (a per-wave read could send waves of one workgroup down both).
  if (threadIdx.x == 0) s_ctl[5] = (NW == 1) ? __hip_atomic_load(&flags[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  lds_barrier();
  const bool check_on = s_ctl[5] != 0u;   // (s_ctl[5] is the progress word: table_clear resets it behind the next barrier)
  lds_barrier();
  while (true) {
    if (threadIdx.x == 0) *s_out = 0;
    bool failed = false;
    // several passes (NW == 1): every pass but the last writes the keys of the later passes, compacted, to a list the next
    // pass reads instead of the whole bucket. The lists alternate between the bucket's range of `scratch` (upwards) and the
    // top of its range of tmp_keys (downwards: entries emitted plus keys still to do never exceed the bucket's keys, and a
    // pass emits only after it has read its stream). new_keys stays intact for a restart with more passes.
    const uint64_t *src = new_keys + nb * NW;
    uint32_t n_src = (uint32_t)(ne - nb);
    uint32_t pass_len = n_src;   // keys the pass in work was given (for the pass-count estimate when it overflows)
    for (uint32_t pass = 0; pass < npass && !failed; ++pass) {
      table_clear<NW>(tab);
      if (threadIdx.x == 0) { *s_spill = 0; *s_hint = 0; }
      lds_barrier();
      for_each_key<NW, BatchOf<NW>::U>(old_keys, ob, oe, [&](const uint64_t (&k)[NW], uint64_t i) {
        const uint32_t h = place_hash<NW>(k);
        if (pass_of(h, npass) != pass) return;
        int s = table_upsert<NW>(tab, k, h);
        if (s >= 0) count_add(&tab.vals[s], old_vals[i], sat);
        else if (s == -2) count_add(tab.special, old_vals[i], sat);
      });
      if constexpr (NW == 1) {
        if (npass == 1 && ob == oe && check_on) {   // nothing to merge: the first load step tells whether the bucket fits (CHECK)
          if (full_word_keys) table_insert_flat<kLoadBatch, false, true, true>(tab, wq, new_keys + nb, (uint32_t)(ne - nb), 1u, 0u, s_hint);
          else table_insert_flat<kLoadBatch, false, false, true>(tab, wq, new_keys + nb, (uint32_t)(ne - nb), 1u, 0u, s_hint);
        } else if (npass == 1) {
          if (full_word_keys) table_insert_flat<kLoadBatch, false, true, false>(tab, wq, new_keys + nb, (uint32_t)(ne - nb), 1u, 0u);
          else table_insert_flat<kLoadBatch, false, false, false>(tab, wq, new_keys + nb, (uint32_t)(ne - nb), 1u, 0u);
        } else if (scratch == nullptr) {
          table_insert_flat<kLoadBatch, true, true, false>(tab, wq, new_keys + nb, (uint32_t)(ne - nb), npass, pass);
        } else if (pass + 1u < npass) {
          const bool up = (pass & 1u) == 0u;
          uint64_t *spill = up ? scratch + nb : tmp_keys + tmp0 + (ne - nb) + (oe - ob) - 1u;
          pass_len = n_src;
          table_insert_stream1<kLoadBatch, true, true, true>(tab, src, n_src, npass, pass, spill, up ? 1 : -1, s_spill);
          __syncthreads();   // workgroup-scope release / acquire: the list is read by other lanes of this workgroup in the next
                             // pass (a device-scope fence here would write the XCD's L2 back once per bucket and pass)
          n_src = *s_spill;
          src = up ? spill : spill - (n_src ? n_src - 1u : 0u);
        } else {
          pass_len = n_src;
          table_insert_flat<kLoadBatch, false, true, false>(tab, wq, src, n_src, 1u, 0u);   // what is left belongs to the last pass
        }
      } else {
        for_each_key<NW, BatchOf<NW>::U>(new_keys, nb, ne, [&](const uint64_t (&k)[NW], uint64_t) {
          // (a lost pass ends here, not at the end of the stream. The pass count doubles from attempt to attempt: an estimate from how
          // far the stream had got when the table was full was tried and cost more attempts than it saved -- 163 against 117 ms for
          // 5.3e8 distinct 63-mers. Reading the home slots of a batch's keys back to back before any of them is inserted -- the flat
          // insert of the one-word tables -- did not move this kernel either: 19.9 ms with and without.)
          if (__atomic_load_n(tab.overflow, __ATOMIC_RELAXED)) return;
          const uint32_t h = place_hash<NW>(k);
          if (pass_of(h, npass) != pass) return;
          int s = table_upsert<NW>(tab, k, h);
          if (s >= 0) atomicAdd(&tab.vals[s], 1u);
        });
      }
      lds_barrier();
      if (*tab.overflow || (NW == 1 && *s_hint)) { failed = true; break; }
      for (int s = threadIdx.x; s < TabCfg<NW>::SLOTS; s += blockDim.x) {
        const bool used = slot_used<NW>(tab, s);
        const uint32_t pos = wave_alloc(s_out, used);
        if (used) {
#pragma unroll
          for (int w = 0; w < NW; ++w) tmp_keys[(tmp0 + pos) * NW + w] = tab.keys[(uint64_t)s * NW + w];
          tmp_vals[tmp0 + pos] = tab.vals[s];
        }
      }
      lds_barrier();
      if (NW == 1 && threadIdx.x == 0 && *tab.special_set) {
        const uint32_t pos = atomicAdd(s_out, 1u);
        tmp_keys[(tmp0 + pos) * NW] = kEmptyKey;
        tmp_vals[tmp0 + pos] = *tab.special;
      }
      lds_barrier();
    }
    if (!failed) break;
    {
      // how many passes next: the table held LIMIT distinct keys of this pass after `progress` of the bucket's n stream keys
      // (known to one load step), so the pass's share of the stream needs about n / progress tables; a third on top for
      // the spread between passes. Without that knowledge (the old entries alone overflowed) the count doubles.
      // (a later pass reads a list of the keys of the passes from it on: its share of that list fills the table after
      // `progress` of pass_len keys just as a share of the whole bucket would, so the same ratio scales the pass count)
      const uint32_t n_new = pass_len, prog = *tab.progress, hinted = (NW == 1) ? *s_hint : 0u;
      uint32_t want = npass * 2u;
      if (NW == 1 && npass == 1u && !check_on && threadIdx.x == 0) atomicOr(&flags[8], 1u);   // later buckets of this launch check
      if (hinted) want = hinted;   // the first load step already told (CHECK)
      else if (NW == 1 && prog > 0u && n_new > 0u) {
        const uint32_t seen = prog > 12288u ? prog - 8192u : prog / 2u + 2048u;   // the overflow came somewhere inside the last step
        const uint64_t est = ((uint64_t)npass * n_new * 4u + 3ull * seen - 1ull) / (3ull * seen);
        want = est > (uint64_t)kMaxPasses ? kMaxPasses : (uint32_t)est;
        if (want <= npass) want = npass + 1u;
      }
      lds_barrier();   // everyone has read the progress word before the next attempt clears it
      npass = want;
    }
    if (npass > kMaxPasses) { if (threadIdx.x == 0) { atomicOr(&flags[2], 1u); *s_out = 0; } lds_barrier(); break; }
    lds_barrier();
  }
  if (threadIdx.x == 0) out_cnt[b] = *s_out;
}

// ---------------------------------------------------------------------------
// Combine-first distributed insert of the count index (N > 1). The reference sends every k-mer occurrence through
// imxx::distribute and reduces at the receiver (distributed_unordered_map.hpp:1715-1745; its local_reduction before
// the exchange exists but is commented out). Counts add up associatively, so reducing the rank's own reads first gives
// the same index and cuts the exchanged volume by the local coverage: the rank builds a local count index of its
// reads (the one-rank pipeline), SPLITS its entries by KeyToRank keeping fine-bucket order inside every rank's
// message, exchanges (k-mer, count) pairs plus the per-bucket counts, and MERGES the p bucket-ordered parts it
// receives into its index without partitioning anything again (all ranks use the same placement hash).
// ---------------------------------------------------------------------------
// split 1: destination rank of every entry (kept, one byte) and the (rank, bucket) counts
template <int NW>
__global__ __launch_bounds__(256) void split_count_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ off, BucketFn fn,
                                                         uint8_t *__restrict__ rank_of, uint32_t *__restrict__ cnt /* [nranks][kNumFine] */) {
  __shared__ uint32_t s_cnt[kNumCoarse];
  const uint32_t b = blockIdx.x;
  if (threadIdx.x < fn.nranks) s_cnt[threadIdx.x] = 0;
  lds_barrier();
  const uint64_t e0 = off[b], e1 = off[b + 1];
  for (uint64_t i = e0 + threadIdx.x; i < e1; i += 256) {
    uint64_t k[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) k[w] = keys[i * NW + w];
    const uint32_t r = bucket_of<NW>(k, fn);   // fn.sub == 1: the rank itself (DistTrans and DistHash as the routers apply them)
    rank_of[i] = (uint8_t)r;
    atomicAdd(&s_cnt[r], 1u);
  }
  lds_barrier();
  if (threadIdx.x < fn.nranks) cnt[(uint64_t)threadIdx.x * kNumFine + b] = s_cnt[threadIdx.x];
}

// per part (rank): exclusive scan of its bucket counts -> boff[part][kNumFine + 1]; tot[part]
__global__ __launch_bounds__(1024) void part_offsets_kernel(const uint32_t *__restrict__ cnt, uint64_t *__restrict__ boff, uint64_t *__restrict__ tot) {
  __shared__ uint64_t s_scan[1024 / 64 + 2];
  constexpr int PER = kNumFine / 1024;
  const uint32_t *c = cnt + (uint64_t)blockIdx.x * kNumFine;
  uint64_t *o = boff + (uint64_t)blockIdx.x * (kNumFine + 1);
  uint64_t loc[PER], sum = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) { loc[i] = c[threadIdx.x * PER + i]; sum += loc[i]; }
  uint64_t total;
  uint64_t x = block_exclusive_scan<uint64_t>(sum, s_scan, &total);
#pragma unroll
  for (int i = 0; i < PER; ++i) { o[threadIdx.x * PER + i] = x; x += loc[i]; }
  if (threadIdx.x == 0) { o[kNumFine] = total; tot[blockIdx.x] = total; }
}

// base[r] = entries of the parts before r (nparts <= 256); base[nparts] = all
__global__ __launch_bounds__(256) void part_bases_kernel(const uint64_t *__restrict__ tot, uint32_t nparts, uint64_t *__restrict__ base) {
  __shared__ uint64_t s_scan[256 / 64 + 2];
  const uint64_t v = threadIdx.x < nparts ? tot[threadIdx.x] : 0ull;
  uint64_t total;
  const uint64_t x = block_exclusive_scan<uint64_t>(v, s_scan, &total);
  if (threadIdx.x < nparts) base[threadIdx.x] = x;
  if (threadIdx.x == 0) base[nparts] = total;
}

// split 2: entries of bucket b to their rank's message, at base[r] + boff[r][b] (order inside is free)
template <int NW>
__global__ __launch_bounds__(256) void split_scatter_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                           const uint64_t *__restrict__ off, const uint8_t *__restrict__ rank_of, uint32_t nranks,
                                                           const uint64_t *__restrict__ boff, const uint64_t *__restrict__ base,
                                                           uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_vals) {
  __shared__ uint32_t s_pos[kNumCoarse];
  __shared__ uint64_t s_dst[kNumCoarse];
  const uint32_t b = blockIdx.x;
  if (threadIdx.x < nranks) {
    s_pos[threadIdx.x] = 0;
    s_dst[threadIdx.x] = base[threadIdx.x] + boff[(uint64_t)threadIdx.x * (kNumFine + 1) + b];
  }
  lds_barrier();
  const uint64_t e0 = off[b], e1 = off[b + 1];
  for (uint64_t i = e0 + threadIdx.x; i < e1; i += 256) {
    const uint32_t r = rank_of[i];
    const uint64_t d = s_dst[r] + atomicAdd(&s_pos[r], 1u);
#pragma unroll
    for (int w = 0; w < NW; ++w) out_keys[d * NW + w] = keys[i * NW + w];
    out_vals[d] = vals[i];
  }
}

// comb[b] = sum over the parts of boff[part][b]  (b = 0 .. kNumFine)
__global__ __launch_bounds__(256) void parts_sum_kernel(const uint64_t *__restrict__ boff, uint32_t nparts, uint64_t *__restrict__ comb) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b > (uint32_t)kNumFine) return;
  uint64_t x = 0;
  for (uint32_t s = 0; s < nparts; ++s) x += boff[(uint64_t)s * (kNumFine + 1) + b];
  comb[b] = x;
}

// merge: bucket b of the index and bucket b of every received part -> one table; same output contract as
// bucket_reduce_kernel (tmp arrays at comb[b] + old_off[b], out_cnt[b] = distinct)
template <int NW>
__global__ __launch_bounds__((TabCfg<NW>::NT)) void bucket_merge_kernel(const uint64_t *__restrict__ part_keys, const uint32_t *__restrict__ part_vals,
                                                                       uint32_t nparts, const uint64_t *__restrict__ base,
                                                                       const uint64_t *__restrict__ boff, const uint64_t *__restrict__ comb,
                                                                       const uint64_t *__restrict__ old_keys, const uint32_t *__restrict__ old_vals,
                                                                       const uint64_t *__restrict__ old_off, uint64_t *__restrict__ tmp_keys,
                                                                       uint32_t *__restrict__ tmp_vals, uint32_t *__restrict__ out_cnt,
                                                                       uint32_t *__restrict__ flags, bool sat = false) {
  KMI_TABLE_LDS(NW)
  const uint32_t b = blockIdx.x;
  const uint64_t ob = old_off ? old_off[b] : 0ull, oe = old_off ? old_off[b + 1] : 0ull;
  if (comb[b] == comb[b + 1] && ob == oe) { if (threadIdx.x == 0) out_cnt[b] = 0; return; }
  const uint64_t tmp0 = comb[b] + ob;
  uint32_t *s_out = &s_ctl[4];
  uint32_t npass = 1;
  while (true) {
    if (threadIdx.x == 0) *s_out = 0;
    bool failed = false;
    for (uint32_t pass = 0; pass < npass && !failed; ++pass) {
      table_clear<NW>(tab);
      lds_barrier();
      auto add = [&](const uint64_t *keys, const uint32_t *vals, uint64_t kb, uint64_t ke) {
        for_each_key<NW, BatchOf<NW>::U>(keys, kb, ke, [&](const uint64_t (&k)[NW], uint64_t i) {
          const uint32_t h = place_hash<NW>(k);
          if (pass_of(h, npass) != pass) return;
          int s = table_upsert<NW>(tab, k, h);
          if (s >= 0) count_add(&tab.vals[s], vals[i], sat);
          else if (s == -2) count_add(tab.special, vals[i], sat);
        });
      };
      add(old_keys, old_vals, ob, oe);
      for (uint32_t s = 0; s < nparts; ++s) {
        const uint64_t *bo = boff + (uint64_t)s * (kNumFine + 1);
        add(part_keys, part_vals, base[s] + bo[b], base[s] + bo[b + 1]);
      }
      lds_barrier();
      if (*tab.overflow) { failed = true; break; }
      for (int s = threadIdx.x; s < TabCfg<NW>::SLOTS; s += blockDim.x) {
        const bool used = slot_used<NW>(tab, s);
        const uint32_t pos = wave_alloc(s_out, used);
        if (used) {
#pragma unroll
          for (int w = 0; w < NW; ++w) tmp_keys[(tmp0 + pos) * NW + w] = tab.keys[(uint64_t)s * NW + w];
          tmp_vals[tmp0 + pos] = tab.vals[s];
        }
      }
      lds_barrier();
      if (NW == 1 && threadIdx.x == 0 && *tab.special_set) {
        const uint32_t pos = atomicAdd(s_out, 1u);
        tmp_keys[(tmp0 + pos) * NW] = kEmptyKey;
        tmp_vals[tmp0 + pos] = *tab.special;
      }
      lds_barrier();
    }
    if (!failed) break;
    npass *= 2;
    if (npass > kMaxPasses) { if (threadIdx.x == 0) { atomicOr(&flags[2], 1u); *s_out = 0; } lds_barrier(); break; }
    lds_barrier();
  }
  if (threadIdx.x == 0) out_cnt[b] = *s_out;
}

// Weighted insert: reduction_unordered_map::local_insert with the caller's value (distributed_unordered_map.hpp:1603-1618,
// `at() = r(at(), v)` with r = std::plus): records = key words followed by ONE value word whose low 32 bits are the count
// (the object bytes of std::pair<Kmer, uint32_t>). Bucket b of the index and the records of bucket b go through one table.
// Same output contract as bucket_reduce_kernel.
template <int NW>
__global__ __launch_bounds__((TabCfg<NW>::NT)) void bucket_reduce_pairs_kernel(const uint64_t *__restrict__ recs, const uint64_t *__restrict__ new_off,
                                                                              const uint64_t *__restrict__ old_keys, const uint32_t *__restrict__ old_vals,
                                                                              const uint64_t *__restrict__ old_off, uint64_t *__restrict__ tmp_keys,
                                                                              uint32_t *__restrict__ tmp_vals, uint32_t *__restrict__ out_cnt,
                                                                              uint32_t *__restrict__ flags, bool sat = false) {
  KMI_TABLE_LDS(NW)
  constexpr int RW = NW + 1;
  const uint32_t b = blockIdx.x;
  const uint64_t nb = new_off[b], ne = new_off[b + 1];
  const uint64_t ob = old_off ? old_off[b] : 0ull, oe = old_off ? old_off[b + 1] : 0ull;
  if (nb == ne && ob == oe) { if (threadIdx.x == 0) out_cnt[b] = 0; return; }
  const uint64_t tmp0 = nb + ob;
  uint32_t *s_out = &s_ctl[4];
  uint32_t npass = 1;
  while (true) {
    if (threadIdx.x == 0) *s_out = 0;
    bool failed = false;
    for (uint32_t pass = 0; pass < npass && !failed; ++pass) {
      table_clear<NW>(tab);
      lds_barrier();
      for_each_key<NW, BatchOf<NW>::U>(old_keys, ob, oe, [&](const uint64_t (&k)[NW], uint64_t i) {
        const uint32_t h = place_hash<NW>(k);
        if (pass_of(h, npass) != pass) return;
        int s = table_upsert<NW>(tab, k, h);
        if (s >= 0) count_add(&tab.vals[s], old_vals[i], sat);
        else if (s == -2) count_add(tab.special, old_vals[i], sat);
      });
      for (uint64_t i = nb + threadIdx.x; i < ne; i += blockDim.x) {
        uint64_t k[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = recs[i * RW + w];
        const uint32_t v = (uint32_t)recs[i * RW + NW];
        const uint32_t h = place_hash<NW>(k);
        if (pass_of(h, npass) != pass) continue;
        int s = table_upsert<NW>(tab, k, h);
        if (s >= 0) count_add(&tab.vals[s], v, sat);
        else if (s == -2) count_add(tab.special, v, sat);
      }
      lds_barrier();
      if (*tab.overflow) { failed = true; break; }
      for (int s = threadIdx.x; s < TabCfg<NW>::SLOTS; s += blockDim.x) {
        const bool used = slot_used<NW>(tab, s);
        const uint32_t pos = wave_alloc(s_out, used);
        if (used) {
#pragma unroll
          for (int w = 0; w < NW; ++w) tmp_keys[(tmp0 + pos) * NW + w] = tab.keys[(uint64_t)s * NW + w];
          tmp_vals[tmp0 + pos] = tab.vals[s];
        }
      }
      lds_barrier();
      if (NW == 1 && threadIdx.x == 0 && *tab.special_set) {
        const uint32_t pos = atomicAdd(s_out, 1u);
        tmp_keys[(tmp0 + pos) * NW] = kEmptyKey;
        tmp_vals[tmp0 + pos] = *tab.special;
      }
      lds_barrier();
    }
    if (!failed) break;
    npass *= 2;
    if (npass > kMaxPasses) { if (threadIdx.x == 0) { atomicOr(&flags[2], 1u); *s_out = 0; } lds_barrier(); break; }
    lds_barrier();
  }
  if (threadIdx.x == 0) out_cnt[b] = *s_out;
}

// records (key words, count word) of an index whose keys are already distinct and fine-partitioned -> the index arrays
template <int NW>
__global__ __launch_bounds__(256) void unzip_pairs_kernel(const uint64_t *__restrict__ recs, uint64_t n, uint64_t *__restrict__ keys,
                                                         uint32_t *__restrict__ vals) {
  constexpr int RW = NW + 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int w = 0; w < NW; ++w) keys[i * NW + w] = recs[i * RW + w];
    vals[i] = (uint32_t)recs[i * RW + NW];
  }
}

// scan of per-bucket counts -> offsets (kNumFine+1) ; totals[slot] = total
__global__ __launch_bounds__(1024) void bucket_offsets_kernel(const uint32_t *__restrict__ cnt, uint64_t *__restrict__ off,
                                                             uint64_t *__restrict__ totals, int slot) {
  __shared__ uint64_t s_scan[1024 / 64 + 2];
  constexpr int PER = kNumFine / 1024;
  uint64_t loc[PER], sum = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) { loc[i] = cnt[threadIdx.x * PER + i]; sum += loc[i]; }
  uint64_t total;
  uint64_t o = block_exclusive_scan<uint64_t>(sum, s_scan, &total);
#pragma unroll
  for (int i = 0; i < PER; ++i) { off[threadIdx.x * PER + i] = o; o += loc[i]; }
  if (threadIdx.x == 0) { off[kNumFine] = total; totals[slot] = total; }
}

// compaction copy: bucket b's `cnt[b]` entries from tmp (at src_off_a[b] + src_off_b[b]) to final (dst_off[b])
template <int NW, typename V>
__global__ __launch_bounds__(256) void bucket_compact_kernel(const uint64_t *__restrict__ tmp_keys, const V *__restrict__ tmp_vals,
                                                            const uint64_t *__restrict__ src_off_a, const uint64_t *__restrict__ src_off_b,
                                                            const uint64_t *__restrict__ dst_off, uint64_t *__restrict__ keys,
                                                            V *__restrict__ vals) {
  const uint32_t b = blockIdx.x;
  const uint64_t d0 = dst_off[b], n = dst_off[b + 1] - d0;
  const uint64_t s0 = src_off_a[b] + (src_off_b ? src_off_b[b] : 0ull);
  for (uint64_t i = threadIdx.x; i < n * NW; i += blockDim.x) keys[d0 * NW + i] = tmp_keys[s0 * NW + i];
  for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) vals[d0 + i] = tmp_vals[s0 + i];
}

// ---------------------------------------------------------------------------
// Q: per fine bucket query. mode 0 = count (emit every distinct query key with 0/1),
// 1 = find (emit (key, stored count) of hits), 2 = erase (emit surviving index entries)
// ---------------------------------------------------------------------------
enum QueryMode { Q_COUNT = 0, Q_FIND = 1, Q_ERASE = 2, Q_HITS = 3 /* find without output: hits per bucket (sizes a multimap find) */ };

// VW = 0: counting map (values are u32 counts in idx_vals32); VW > 0: multimap (every entry carries
// VW 64-bit value words in idx_mvals, a key may occur many times). Output value stride OW = max(1, VW).
template <int NW, int VW>
__global__ __launch_bounds__((QTabCfg<NW>::NT)) void bucket_query_kernel(int mode, const uint64_t *__restrict__ q_keys, const uint64_t *__restrict__ q_off,
                                                                       const uint64_t *__restrict__ idx_keys, const uint32_t *__restrict__ idx_vals32,
                                                                       const uint64_t *__restrict__ idx_mvals,
                                                                       const uint64_t *__restrict__ idx_off, uint64_t *__restrict__ tmp_keys,
                                                                       uint64_t *__restrict__ tmp_vals64, uint32_t *__restrict__ tmp_vals32,
                                                                       uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ flags,
                                                                       bool emit_index, const uint32_t *__restrict__ idx_cnt = nullptr,
                                                                       const uint64_t *__restrict__ out_off = nullptr) {
  // idx_cnt (sparse index, count / find of a counting map only): bucket b holds idx_cnt[b] entries from idx_off[b]
  // out_off (find of a multimap after a Q_HITS pass): bucket b's results go to out_off[b] -- compact, no slot per index entry
  // emit_index (find on a counting map): the value of a hit is the entry's position in the index arrays instead of its count
  // (the de Bruijn node map gathers the node's edge counts from there)
  KMI_TABLE_LDS_CFG(NW, QTabCfg<NW>)
  constexpr int OW = VW ? VW : 1;
  const uint32_t b = blockIdx.x;
  const uint64_t qb = q_off[b], qe = q_off[b + 1];
  const uint64_t ib = idx_off ? idx_off[b] : 0ull, ie = idx_off ? (idx_cnt ? ib + idx_cnt[b] : idx_off[b + 1]) : 0ull;
  // output slot base: count results are bounded by the bucket's queries; erase survivors and multimap
  // find hits by the bucket's entries
  const bool by_entries = (mode == Q_ERASE) || (VW > 0 && mode == Q_FIND);
  const uint64_t tmp0 = out_off ? out_off[b] : (by_entries ? ib : qb);
  uint32_t *s_out = &s_ctl[4];
  auto emit_entry = [&](uint32_t pos, const uint64_t (&k)[NW], uint64_t i) {
#pragma unroll
    for (int w = 0; w < NW; ++w) tmp_keys[(tmp0 + pos) * NW + w] = k[w];
    if constexpr (VW == 0) {
      if (mode == Q_ERASE) tmp_vals32[tmp0 + pos] = idx_vals32[i]; else tmp_vals64[tmp0 + pos] = emit_index ? i : (uint64_t)idx_vals32[i];
    } else {
#pragma unroll
      for (int w = 0; w < VW; ++w) tmp_vals64[(tmp0 + pos) * VW + w] = idx_mvals[i * VW + w];
    }
  };
  {
    // the table only ever holds this bucket's distinct query keys: twice their number of home slots, at least 512
    uint32_t want = 2u * (uint32_t)((qe - qb) < (uint64_t)QTabCfg<NW>::CAP ? (qe - qb) : (uint64_t)QTabCfg<NW>::CAP);
    want = (want + 255u) & ~255u;
    want = want < 512u ? 512u : want;
    if (want < (uint32_t)QTabCfg<NW>::CAP) { tab.cap = want; tab.slots = want + QTabCfg<NW>::PAD; tab.limit = want * 3u / 4u; }
  }
  if (qb == qe) {
    if (mode == Q_ERASE) {
      // nothing to erase here: all entries survive
      for (uint64_t i = ib + threadIdx.x; i < ie; i += blockDim.x) {
        uint64_t k[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = idx_keys[i * NW + w];
        emit_entry((uint32_t)(i - ib), k, i);
      }
      if (threadIdx.x == 0) out_cnt[b] = (uint32_t)(ie - ib);
    } else if (threadIdx.x == 0) out_cnt[b] = 0;
    return;
  }
  uint32_t npass = 1;
  while (true) {
    if (threadIdx.x == 0) *s_out = 0;
    bool failed = false;
    for (uint32_t pass = 0; pass < npass && !failed; ++pass) {
      table_clear<NW>(tab);
      lds_barrier();
      for_each_key<NW, BatchOf<NW>::U>(q_keys, qb, qe, [&](const uint64_t (&k)[NW], uint64_t) {
        const uint32_t h = place_hash<NW>(k);
        if (pass_of(h, npass) != pass) return;
        (void)table_upsert<NW>(tab, k, h);
      });
      lds_barrier();
      if (*tab.overflow) { failed = true; break; }
      // stream the index bucket against the query table
      for (uint64_t i = ib + threadIdx.x; i < ie; i += blockDim.x) {
        uint64_t k[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = idx_keys[i * NW + w];
        const uint32_t h = place_hash<NW>(k);
        if (pass_of(h, npass) != pass) continue;
        const int s = table_find<NW>(tab, k, h);
        const bool hit = (s >= 0) || (s == -2);
        if (mode == Q_COUNT) {
          // db.count(k): number of entries with this key (0/1 for a map, multiplicity for a multimap)
          if (s >= 0) atomicAdd(&tab.vals[s], 1u); else if (s == -2) atomicAdd(tab.special, 1u);
        } else if (mode == Q_FIND) {
          const uint32_t pos = wave_alloc(s_out, hit);
          if (hit) emit_entry(pos, k, i);
        } else if (mode == Q_HITS) {
          (void)wave_alloc(s_out, hit);
        } else {
          const uint32_t pos = wave_alloc(s_out, !hit);
          if (!hit) emit_entry(pos, k, i);
        }
      }
      lds_barrier();
      if (mode == Q_COUNT) {
        for (uint32_t s = threadIdx.x; s < ((tab.slots + kWave - 1u) & ~(uint32_t)(kWave - 1)); s += blockDim.x) {   // whole waves: wave_alloc
          const bool used = s < tab.slots && slot_used<NW>(tab, (int)s);
          const uint32_t pos = wave_alloc(s_out, used);
          if (used) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tmp_keys[(tmp0 + pos) * NW + w] = tab.keys[(uint64_t)s * NW + w];
            tmp_vals64[(tmp0 + pos) * OW] = tab.vals[s];
          }
        }
        if (NW == 1 && threadIdx.x == 0 && *tab.special_set) {
          const uint32_t pos = atomicAdd(s_out, 1u);
          tmp_keys[(tmp0 + pos) * NW] = kEmptyKey;
          tmp_vals64[(tmp0 + pos) * OW] = *tab.special;
        }
      }
      lds_barrier();
    }
    if (!failed) break;
    npass *= 2;
    if (npass > kMaxPasses) { if (threadIdx.x == 0) { atomicOr(&flags[2], 1u); *s_out = 0; } lds_barrier(); break; }
    lds_barrier();
  }
  if (threadIdx.x == 0) out_cnt[b] = *s_out;
}

// multimap insert: bucket b of the new index = old entries of b followed by the new records of b
template <int NW, int VW>
__global__ __launch_bounds__(256) void bucket_concat_kernel(const uint64_t *__restrict__ recs, const uint64_t *__restrict__ new_off,
                                                           const uint64_t *__restrict__ old_keys, const uint64_t *__restrict__ old_vals,
                                                           const uint64_t *__restrict__ old_off, uint64_t *__restrict__ dst_off,
                                                           uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
  constexpr int RW = NW + VW;
  const uint32_t b = blockIdx.x;
  const uint64_t nb = new_off[b], nn = new_off[b + 1] - nb;
  const uint64_t ob = old_off ? old_off[b] : 0ull, on = old_off ? old_off[b + 1] - ob : 0ull;
  const uint64_t d0 = nb + ob;
  if (threadIdx.x == 0) {
    dst_off[b] = d0;
    if (b == kNumFine - 1) dst_off[kNumFine] = d0 + nn + on;
  }
  for (uint64_t i = threadIdx.x; i < on * NW; i += blockDim.x) keys[d0 * NW + i] = old_keys[ob * NW + i];
  for (uint64_t i = threadIdx.x; i < on * VW; i += blockDim.x) vals[d0 * VW + i] = old_vals[ob * VW + i];
  const uint64_t d1 = d0 + on;
  for (uint64_t i = threadIdx.x; i < nn; i += blockDim.x) {
#pragma unroll
    for (int w = 0; w < NW; ++w) keys[(d1 + i) * NW + w] = recs[(nb + i) * RW + w];
#pragma unroll
    for (int w = 0; w < VW; ++w) vals[(d1 + i) * VW + w] = recs[(nb + i) * RW + NW + w];
  }
}

// compaction copy with VW value words per entry
template <int NW, int VW>
__global__ __launch_bounds__(256) void bucket_compact_words_kernel(const uint64_t *__restrict__ tmp_keys, const uint64_t *__restrict__ tmp_vals,
                                                                  const uint64_t *__restrict__ src_off, const uint64_t *__restrict__ dst_off,
                                                                  uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
  const uint32_t b = blockIdx.x;
  const uint64_t d0 = dst_off[b], n = dst_off[b + 1] - d0, s0 = src_off[b];
  for (uint64_t i = threadIdx.x; i < n * NW; i += blockDim.x) keys[d0 * NW + i] = tmp_keys[s0 * NW + i];
  for (uint64_t i = threadIdx.x; i < n * VW; i += blockDim.x) vals[d0 * VW + i] = tmp_vals[s0 * VW + i];
}

}  // namespace kmi

#include "kmi_superkmer.h"
#include "kmi_reduce2.h"
#include "kmi_front.h"

// ===========================================================================
// host side
// ===========================================================================
using namespace kmi;

struct kmi_index {
  kmi_ctx *ctx = nullptr;
  kmi_config cfg{};
  KShape shape{};
  uint64_t *keys = nullptr;       // [n_entries * n_words]
  uint32_t *vals = nullptr;       // [n_entries] counts (counting map)
  uint64_t *mvals = nullptr;      // [n_entries * val_words] values (multimap)
  uint32_t val_words = 0;         // 0: counting map; 1: position id; 2: position id + quality
  uint64_t *bucket_off = nullptr; // [kNumFine + 1]
  // SPARSE form (what a large super-k-mer build leaves: the reduce's output buffers as they are, no compaction pass): bucket b
  // holds bucket_cnt[b] entries from bucket_off[b], with unused slots behind them; dense_off = the offsets the compacted
  // arrays will have. Count / find read it as it is; everything else calls ensure_dense first.
  uint32_t *bucket_cnt = nullptr; // [kNumFine]; null = dense (bucket b = [bucket_off[b], bucket_off[b + 1]))
  uint64_t *dense_off = nullptr;  // [kNumFine + 1] (sparse form only)
  uint64_t n_entries = 0;
  bool has_data = false;
  bool saturating = false;        // count += w stops at 2^32 - 1 (sat_plus) instead of wrapping (std::plus): kmi_index_set_saturating
  uint32_t owner_lp = 0;          // the entries are this rank's share of a build over 2^owner_lp ranks by minimizer-bucket owner (sk_consume)
  bool find_emits_index = false;  // find() of a counting map reports entry positions instead of counts (kmi_debruijn.h)
  uint32_t layout_w = 0;          // what the fine buckets mean: 0 = top bits of the placement hash; W = minimizer bucket (fine15_of_key)
  size_t keys_bytes = 0, vals_bytes = 0, mvals_bytes = 0;   // sizes of the blocks above (for the context's spare list)
};
constexpr size_t kOffBytes = sizeof(uint64_t) * ((size_t)1 << 15) + sizeof(uint64_t);

namespace kmi {

struct Partitioned;
static kmi_status index_insert_pairs(kmi_index *idx, const uint64_t *recs_dev, size_t n, bool transform, bool distinct_in,
                                     Partitioned *part_out = nullptr);
static kmi_status ensure_layout(kmi_index *idx, uint32_t target_w);
static kmi_status ensure_dense(kmi_index *idx);
static void free_index_arrays(kmi_index *idx);
template <int NW>
__global__ void zip_pairs_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t n, uint64_t *__restrict__ recs);
// the layout a count index of this configuration is kept in when it has the choice (split / merge between ranks agree on it)
static uint32_t preferred_layout(const kmi_index *idx) {
  return (idx->val_words == 0 && idx->shape.n_words == 1 && idx->shape.bits == 2 && idx->ctx->fused_superkmer) ? sk_window_of(idx->shape.k) : 0u;
}


struct Partitioned {
  uint64_t *keys;      // fine-partitioned keys (WS_KEYS_B or WS_QUERY_B)
  uint64_t *fine_off;  // [kNumFine+1]
  uint64_t *scratch = nullptr;   // the other partition buffer (free once the keys sit in `keys`): pass lists of bucket_reduce
};

struct PartWs {
  uint64_t *buf_a, *buf_b;
  uint32_t *fine_hist;   // [kFineParts][kNumFine]
  uint64_t *fine_off;    // [kNumFine + 1]
  uint64_t *part_off;    // [kFineParts][kNumFine]
  uint64_t *coarse_base; // [kNumCoarse]
  uint32_t *wg_hist;     // [kPartGroups][kNumCoarse]
  uint64_t *wg_off;      // [kPartGroups][kNumCoarse]
};

static kmi_status get_part_ws(kmi_ctx *ctx, size_t n, int nw, WsSlot slot_a, WsSlot slot_b, PartWs *w) {
  void *p;
  const size_t key_bytes = (n ? n : 1) * nw * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, slot_a, key_bytes, &p)); w->buf_a = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, slot_b, key_bytes, &p)); w->buf_b = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_HIST, sizeof(uint32_t) * kNumFine * kFineParts + sizeof(uint64_t) * ((kNumFine + 1) * 2 + kNumFine * kFineParts + kNumCoarse) + 256, &p));
  w->fine_hist = (uint32_t *)p;
  // offset arrays live behind the histogram: [0] for inserts, [1] for queries, then the per-part offsets
  uint64_t *off_base = (uint64_t *)((char *)p + sizeof(uint32_t) * kNumFine * kFineParts);
  w->fine_off = off_base + (slot_b == WS_QUERY_B ? (kNumFine + 1) : 0);
  w->part_off = off_base + 2 * (kNumFine + 1);
  w->coarse_base = w->part_off + (uint64_t)kNumFine * kFineParts;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); w->wg_hist = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_CURSOR, sizeof(uint64_t) * kPartGroups * kNumCoarse, &p)); w->wg_off = (uint64_t *)p;
  return KMI_OK;
}

// K1 + offsets + K2 + P2 on `n` keys; result in slot `slot_b`, scratch in `slot_a`
template <int NW, int BITS, int VW = 0>
static kmi_status partition_impl(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint64_t *keys_dev, size_t n, bool transform,
                                 WsSlot slot_a, WsSlot slot_b, Partitioned *out, uint32_t layout_w = 0, uint64_t *split_keys = nullptr,
                                 uint64_t *split_vals = nullptr, const float *in_q = nullptr, const uint64_t *in_v = nullptr) {
  // in_q (records of two value words): keys_dev holds (key words, first value word) and the second value word of record i is
  // in_q[i]; in_v: keys_dev holds the key words alone, the value words come from in_v / in_q (scatter_range)
  // split_keys / split_vals (records): the last pass writes key words and value words into these two arrays (n entries each)
  // instead of records into the workspace -- the first insert into an empty multimap index needs no further copy
  PartWs w;
  KMI_TRY(get_part_ws(ctx, n, NW + VW, slot_a, slot_b, &w));
  KMI_HIP(ctx, hipMemsetAsync(w.fine_hist, 0, sizeof(uint32_t) * kNumFine * kFineParts, ctx->stream));
  {
    ProfScope ps(ctx, "hist_fine", n);
    hipLaunchKernelGGL((hist_fine_kernel<NW, BITS, VW>), dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, keys_dev, (uint64_t)n, shape,
                       cfg->strand, transform, w.fine_hist, w.wg_hist, layout_w,
                       (VW > 0 && in_v) ? (uint32_t)NW : (uint32_t)(NW + VW) - ((VW > 0 && in_q) ? 1u : 0u));
  }
  {
    ProfScope ps(ctx, "fine_offsets", kNumFine);
    launch_fine_offsets(ctx, w.fine_hist, w.fine_off, w.part_off, w.coarse_base);
    hipLaunchKernelGGL(coarse_cursors_kernel, dim3(kNumCoarse / 4), dim3(256), 0, ctx->stream, (const uint32_t *)w.wg_hist, (uint32_t)kPartGroups,
                       (const uint64_t *)w.coarse_base, w.wg_off);
  }
  BucketFn fn; fn.mode = BUCKET_COARSE; fn.shape = shape; fn.dist_hash = 0; fn.farm_ndebug = false; fn.nranks = 1; fn.sub = 1; fn.layout_w = layout_w;
  {
    ProfScope ps(ctx, "scatter_coarse", n);
    hipLaunchKernelGGL((scatter_chunks_kernel<NW, BITS, VW>), dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, keys_dev, (uint64_t)n, w.buf_a,
                       shape, cfg->strand, transform, fn, w.wg_off, (VW > 0) ? in_q : (const float *)nullptr,
                       (VW > 0) ? in_v : (const uint64_t *)nullptr);
  }
  {
    ProfScope ps(ctx, "scatter_fine", n);
    if (NW == 1 && VW == 0)
      hipLaunchKernelGGL(scatter_fine_lines_kernel, dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, (const uint64_t *)w.buf_a,
                         w.buf_b, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off, (const uint64_t *)w.wg_off, (uint32_t)kPartGroups,
                         layout_w, shape.k);
    else if (NW + VW <= 4 && ctx->lines_p2)   // whole lines (scatter_lines_records)
      hipLaunchKernelGGL((scatter_fine_records_lines_kernel<NW, BITS, VW>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream,
                         (const uint64_t *)w.buf_a, (VW > 0 && split_keys) ? split_keys : w.buf_b, shape, (const uint64_t *)w.fine_off,
                         (const uint64_t *)w.part_off, (const uint64_t *)w.wg_off, (uint32_t)kPartGroups, layout_w,
                         (VW > 0 && split_keys) ? split_vals : (uint64_t *)nullptr);
    else
      hipLaunchKernelGGL((scatter_fine_kernel<NW, BITS, VW>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, w.buf_a,
                         (VW > 0 && split_keys) ? split_keys : w.buf_b, shape, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off,
                         (const uint64_t *)w.wg_off, (uint32_t)kPartGroups, (int)BUCKET_SUB, layout_w, (VW > 0 && split_keys) ? split_vals : (uint64_t *)nullptr);
  }
  KMI_HIP(ctx, hipGetLastError());
  out->keys = (VW > 0 && split_keys) ? split_keys : w.buf_b; out->fine_off = w.fine_off;
  out->scratch = (NW == 1 && VW == 0) ? w.buf_a : nullptr;
  return KMI_OK;
}

static kmi_status read_total(kmi_ctx *ctx, int slot, uint64_t *v) {
  uint32_t flag = 0;
  KMI_HIP(ctx, hipMemcpyAsync(ctx->h_totals + slot, ctx->d_totals + slot, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(&flag, ctx->d_flags + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *v = ctx->h_totals[slot];
  if (flag) {
    KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags + 2, 0, sizeof(uint32_t), ctx->stream));
    return set_err(ctx, KMI_ERR_OVERFLOW, "a bucket could not be reduced within the pass limit");
  }
  return KMI_OK;
}

// replace the index arrays by the compacted content of tmp
template <int NW>
static kmi_status adopt_tmp(kmi_index *idx, const uint64_t *tmp_keys, const uint32_t *tmp_vals, const uint64_t *src_a, const uint64_t *src_b,
                            const uint32_t *out_cnt, bool fastq_verdict = false, uint64_t n_slots = 0, bool allow_sparse = false) {
  // allow_sparse (with n_slots, one-word keys): a large output is adopted as it lies -- bucket b = out_cnt[b] entries from src_a[b] --
  // instead of being compacted now (kmi_index::bucket_cnt; ensure_dense does it when something needs dense arrays)
  // n_slots (the output slots of the reduce, when they start at 0 and src_b is null): a reduce that found every key distinct
  // filled all of them, so the workspace buffers ARE the index arrays -- they change owner instead of being copied
  kmi_ctx *ctx = idx->ctx;
  uint64_t *new_off = nullptr;
  KMI_HIP(ctx, pool_alloc(ctx, (void **)&new_off, kOffBytes));
  {
    ProfScope ps(ctx, "bucket_offsets", kNumFine);
    hipLaunchKernelGGL(bucket_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, out_cnt, new_off, ctx->d_totals, 4);
  }
  uint64_t total = 0;
  kmi_status st_total = read_total(ctx, 4, &total);
  if (st_total == KMI_OK && fastq_verdict) st_total = fastq_length_verdict(ctx);   // the index stays as it was on a parse error
  if (st_total != KMI_OK) { pool_free(ctx, new_off, kOffBytes); return st_total; }
  if (n_slots && total == n_slots && !src_b && ctx->ws[WS_TMP_KEYS].p == (const void *)tmp_keys && ctx->ws[WS_TMP_VALS].p == (const void *)tmp_vals) {
    size_t kb0 = 0, vb0 = 0;
    (void)ws_detach(ctx, WS_TMP_KEYS, tmp_keys, &kb0);
    (void)ws_detach(ctx, WS_TMP_VALS, tmp_vals, &vb0);
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_index_arrays(idx);
    idx->keys = const_cast<uint64_t *>(tmp_keys); idx->vals = const_cast<uint32_t *>(tmp_vals); idx->bucket_off = new_off;
    idx->n_entries = total; idx->has_data = true; idx->keys_bytes = kb0; idx->vals_bytes = vb0;
    return KMI_OK;
  }
  // (the sparse form holds the reduce's output buffers -- 12 bytes per k-mer OCCURRENCE, 14 GB at config 2 against 1.2 GB dense -- for the
  // index's lifetime, and the next build allocates its own: it is taken only while the device has room for that second set. Several
  // indexes in one context, or a smaller GPU, get the compaction pass instead of an out-of-memory error: ADVICE r3)
  bool room = false;
  if (NW == 1 && allow_sparse && n_slots >= ctx->sparse_min) {
    size_t free_b = 0, total_b = 0;
    room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= ctx->ws[WS_TMP_KEYS].cap + ctx->ws[WS_TMP_VALS].cap + ((size_t)1 << 30);
  }
  if (NW == 1 && allow_sparse && room && n_slots >= ctx->sparse_min && !src_b && ctx->ws[WS_TMP_KEYS].p == (const void *)tmp_keys &&
      ctx->ws[WS_TMP_VALS].p == (const void *)tmp_vals) {
    uint64_t *off = nullptr; uint32_t *cnt = nullptr;
    hipError_t e1 = pool_alloc(ctx, (void **)&off, kOffBytes);
    hipError_t e2 = pool_alloc(ctx, (void **)&cnt, sizeof(uint32_t) * kNumFine);
    if (e1 == hipSuccess && e2 == hipSuccess) {
      KMI_HIP(ctx, hipMemcpyAsync(off, src_a, kOffBytes, hipMemcpyDeviceToDevice, ctx->stream));
      KMI_HIP(ctx, hipMemcpyAsync(cnt, out_cnt, sizeof(uint32_t) * kNumFine, hipMemcpyDeviceToDevice, ctx->stream));
      size_t kb0 = 0, vb0 = 0;
      (void)ws_detach(ctx, WS_TMP_KEYS, tmp_keys, &kb0);
      (void)ws_detach(ctx, WS_TMP_VALS, tmp_vals, &vb0);
      free_index_arrays(idx);   // (the two copies above are ordered on the stream like everything that may reuse these blocks)
      idx->keys = const_cast<uint64_t *>(tmp_keys); idx->vals = const_cast<uint32_t *>(tmp_vals); idx->bucket_off = off; idx->bucket_cnt = cnt;
      idx->dense_off = new_off; idx->n_entries = total; idx->has_data = true; idx->keys_bytes = kb0; idx->vals_bytes = vb0;
      return KMI_OK;
    }
    if (off) pool_free(ctx, off, kOffBytes);
    if (cnt) pool_free(ctx, cnt, sizeof(uint32_t) * kNumFine);
  }
  uint64_t *nk = nullptr; uint32_t *nv = nullptr;
  const size_t kb = (total ? total : 1) * NW * sizeof(uint64_t), vb = (total ? total : 1) * sizeof(uint32_t);
  hipError_t e1 = pool_alloc(ctx, (void **)&nk, kb);
  hipError_t e2 = pool_alloc(ctx, (void **)&nv, vb);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    if (nk) pool_free(ctx, nk, kb);
    if (nv) pool_free(ctx, nv, vb);
    pool_free(ctx, new_off, kOffBytes);
    return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the index arrays");
  }
  {
    ProfScope ps(ctx, "bucket_compact", total);
    hipLaunchKernelGGL((bucket_compact_kernel<NW, uint32_t>), dim3(kNumFine), dim3(256), 0, ctx->stream, tmp_keys, tmp_vals, src_a, src_b,
                       (const uint64_t *)new_off, nk, nv);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_index_arrays(idx);
  idx->keys = nk; idx->vals = nv; idx->bucket_off = new_off; idx->n_entries = total; idx->has_data = true;
  idx->keys_bytes = kb; idx->vals_bytes = vb;
  return KMI_OK;
}

// C + compaction: fold the fine-partitioned keys into the index
template <int NW>
static kmi_status reduce_and_adopt(kmi_index *idx, const Partitioned &part, size_t n, bool fastq_verdict = false) {
  kmi_ctx *ctx = idx->ctx;
  void *p;
  const uint64_t cap = n + idx->n_entries;
  KMI_TRY(ws_get(ctx, WS_TMP_KEYS, cap * NW * sizeof(uint64_t), &p)); uint64_t *tmp_keys = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_TMP_VALS, cap * sizeof(uint32_t), &p)); uint32_t *tmp_vals = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_BUCKET_CNT, sizeof(uint32_t) * kNumFine, &p)); uint32_t *out_cnt = (uint32_t *)p;
  KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags + 8, 0, sizeof(uint32_t), ctx->stream));   // "a bucket overflowed" (bucket_reduce_kernel)
  {
    ProfScope ps(ctx, "bucket_reduce", n);
    hipLaunchKernelGGL((bucket_reduce_kernel<NW>), dim3(kNumFine), dim3(TabCfg<NW>::NT), 0, ctx->stream, (const uint64_t *)part.keys,
                       (const uint64_t *)part.fine_off, (const uint64_t *)idx->keys, (const uint32_t *)idx->vals,
                       (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), tmp_keys, tmp_vals, out_cnt, ctx->d_flags,
                       idx->shape.n_bits == 64u * NW, part.scratch, idx->saturating);
  }
  KMI_HIP(ctx, hipGetLastError());
  return adopt_tmp<NW>(idx, tmp_keys, tmp_vals, part.fine_off, idx->has_data ? idx->bucket_off : nullptr, out_cnt, fastq_verdict,
                       idx->has_data ? 0 : (uint64_t)n);
}

template <int NW, int BITS>
static kmi_status insert_impl(kmi_index *idx, const uint64_t *keys_dev, size_t n, bool transform) {
  kmi_ctx *ctx = idx->ctx;
  if (n == 0) return KMI_OK;
  KMI_TRY(ensure_layout(idx, 0u));   // k-mers are partitioned by the placement hash
  Partitioned part;
  KMI_TRY((partition_impl<NW, BITS>(ctx, &idx->cfg, idx->shape, keys_dev, n, transform, WS_KEYS_A, WS_KEYS_B, &part)));
  return reduce_and_adopt<NW>(idx, part, n);
}

// Index::build_* on one rank, fused: FASTQ tiles -> histogram, FASTQ tiles -> coarse buckets
template <int NW, int BITS>
static kmi_status build_fused_impl(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes) {
  kmi_ctx *ctx = idx->ctx;
  KMI_TRY(ensure_layout(idx, 0u));
  FastqScan sc;
  KMI_TRY(fastq_scan(ctx, &idx->cfg, bytes_dev, n_bytes, &sc, false));   // reports malformed FASTQ (the length rule rides on the list pass)
  const uint64_t n = sc.n_tuples, n_tiles = sc.n_tiles;
  if (n == 0) return KMI_OK;
  PackedInput in; in.eol = sc.pk_eol; in.stream = sc.pk_stream; in.n_bytes = sc.n_bytes; in.n_cover = sc.n_cover; in.n_valid = sc.n_bytes; in.brk = sc.pk_brk;
  const bool split = sc.pk_brk != nullptr;   // a sequence filter cuts the runs: more, shorter runs per tile
  const uint32_t *line_base = sc.line_base;
  const bool canonical = idx->cfg.strand != KMI_STRAND_SINGLE;
  PartWs w;
  KMI_TRY(get_part_ws(ctx, n, NW, WS_KEYS_A, WS_KEYS_B, &w));
  void *pl;
  using LPC = ListPassCfg<NW, BITS>;
  KMI_TRY(ws_get(ctx, WS_ENT_LIST, sizeof(uint16_t) * ((size_t)n_tiles * LPC::ent_stride(idx->shape.k, split) + 64), &pl));
  uint16_t *ent = (uint16_t *)pl;
  KMI_TRY(ws_get(ctx, WS_ENT_CNT, sizeof(uint32_t) * (n_tiles + 8), &pl));
  uint32_t *ent_cnt = (uint32_t *)pl;
  KMI_HIP(ctx, hipMemsetAsync(w.fine_hist, 0, sizeof(uint32_t) * kNumFine * kFineParts, ctx->stream));
  {
    ProfScope ps(ctx, "fastq_list", n);
    using LP = ListPassCfg<NW, BITS>;
    const uint32_t wave_lds = LP::wave_lds_bytes(idx->shape.k, split);
    hipLaunchKernelGGL((fastq_list_kernel<NW, BITS>), dim3(kListGroups), dim3(kListThreads), wave_lds * (kListThreads / kWave), ctx->stream, in,
                       n_tiles, idx->shape.k, LP::max_runs(idx->shape.k, split), wave_lds, line_base, ctx->d_flags, ent, ent_cnt,
                       LP::ent_stride(idx->shape.k, split));
  }
  {
    ProfScope ps(ctx, "fastq_hist", n);
    hipLaunchKernelGGL((fastq_hist_list_kernel<NW, BITS>), dim3(fused_groups<NW>()), dim3(kHistThreads), 0, ctx->stream, in, n_tiles, idx->shape,
                       canonical, (const uint16_t *)ent, (const uint32_t *)ent_cnt, LPC::ent_stride(idx->shape.k, split), w.fine_hist, w.wg_hist);
  }
  {
    ProfScope ps(ctx, "fine_offsets", kNumFine);
    launch_fine_offsets(ctx, w.fine_hist, w.fine_off, w.part_off, w.coarse_base);
    hipLaunchKernelGGL(coarse_cursors_kernel, dim3(kNumCoarse / 4), dim3(256), 0, ctx->stream, (const uint32_t *)w.wg_hist, (uint32_t)fused_groups<NW>(),
                       (const uint64_t *)w.coarse_base, w.wg_off);
  }
  {
    ProfScope ps(ctx, "fastq_scatter", n);
    hipLaunchKernelGGL((fastq_scatter_list_kernel<NW, BITS>), dim3(fused_groups<NW>()), dim3(ExCfg<NW, BITS>::NT), 0, ctx->stream, in,
                       n_tiles, idx->shape, canonical, (const uint16_t *)ent, (const uint32_t *)ent_cnt, LPC::ent_stride(idx->shape.k, split),
                       (const uint64_t *)nullptr, (const uint64_t *)w.wg_off, w.buf_a);
  }
  {
    ProfScope ps(ctx, "scatter_fine", n);
    if (NW == 1)
      hipLaunchKernelGGL(scatter_fine_lines_kernel, dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, (const uint64_t *)w.buf_a,
                         w.buf_b, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off, (const uint64_t *)w.wg_off,
                         (uint32_t)fused_groups<NW>());
    else if (NW <= 4 && ctx->lines_p2)
      hipLaunchKernelGGL((scatter_fine_records_lines_kernel<NW, BITS, 0>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream,
                         (const uint64_t *)w.buf_a, w.buf_b, idx->shape, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off,
                         (const uint64_t *)w.wg_off, (uint32_t)fused_groups<NW>(), 0u, (uint64_t *)nullptr);
    else
      hipLaunchKernelGGL((scatter_fine_kernel<NW, BITS>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, w.buf_a, w.buf_b,
                         idx->shape, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off, (const uint64_t *)w.wg_off,
                         (uint32_t)fused_groups<NW>());
  }
  KMI_HIP(ctx, hipGetLastError());
  Partitioned part; part.keys = w.buf_b; part.fine_off = w.fine_off; part.scratch = (NW == 1) ? w.buf_a : nullptr;
  return reduce_and_adopt<NW>(idx, part, (size_t)n, true);
}

// Index::build_* on one rank through super-k-mers (kmi_superkmer.h): FASTQ, one-word 2-bit k-mers, k >= 17.
// returns KMI_OK with *done = false when the input does not fit the item capacities (the k-mer pipeline takes over)
// The super-k-mer build in two halves, so that a build over several ranks can put its exchange between them.
// Front end: FASTQ scan -> window runs -> minimizer items -> 16-byte records grouped by the top 8 bucket bits (WS_KEYS_A),
// the groups of workgroup g at wg_off[g][c]. h_cnt / h_base: records and start of every group.
struct SkFront {
  bool ok = false;                 // false: a run or a tile exceeded its item capacity (the caller takes the k-mer path)
  uint64_t *recs = nullptr;        // WS_KEYS_A
  uint64_t n_records = 0, n_kmers = 0, n_seqs = 0;
  uint64_t h_cnt[kNumCoarse], h_base[kNumCoarse];
  uint64_t *wg_off = nullptr;      // [kPartGroups][kNumCoarse] (WS_CURSOR)
};

template <int W>
static kmi_status sk_front_end(kmi_ctx *ctx, const kmi_config *cfg, const KShape &shape, const FastqScan &sc, uint32_t lp, SkFront *f,
                               uint64_t *out = nullptr, size_t out_cap = 0 /* records: the caller's buffer takes them when they fit */,
                               const FastaScan *fa = nullptr /* FASTA: the compacted character stream instead of the FASTQ scan */) {
  constexpr int NW = 1, BITS = 2;
  f->ok = false;
  uint64_t n = fa ? 0 : sc.n_tuples;   // (FASTA: the run pass counts the windows)
  const uint64_t n_tiles = fa ? (fa->n_chars + 8191) / 8192 : sc.n_tiles;
  const uint32_t k = shape.k;
  PackedInput in;
  if (fa) { in.eol = fa->pk_break; in.stream = fa->pk_stream; in.n_bytes = fa->n_chars; in.n_cover = fa->n_cover; in.n_valid = fa->n_valid; in.brk = nullptr; }
  else { in.eol = sc.pk_eol; in.stream = sc.pk_stream; in.n_bytes = sc.n_bytes; in.n_cover = sc.n_cover; in.n_valid = sc.n_bytes; in.brk = sc.pk_brk; }
  const bool split = !fa && sc.pk_brk != nullptr;
  const bool canonical = cfg->strand != KMI_STRAND_SINGLE;
  using LP = ListPassCfg<NW, BITS>;
  // (a tile of the compacted FASTA stream is all windows: 8192 of them against the 3100 of a FASTQ tile -- twice the item slots)
  const uint32_t seg = sk_segment_of((uint32_t)W), ipt = sk_items_per_tile((uint32_t)W) * (fa ? 2u : 1u);
  const uint32_t stride = fa ? 8192u / seg + 72u : LP::run_stride(k, seg, split);
  void *p;
  KMI_TRY(ws_get(ctx, WS_ENT_LIST, sizeof(uint32_t) * ((size_t)n_tiles * stride + 64), &p)); uint32_t *ent = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_ENT_CNT, sizeof(uint32_t) * (n_tiles + 8), &p)); uint32_t *ent_cnt = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_ENT_BKT, sizeof(uint32_t) * ((size_t)n_tiles * stride + 64), &p)); uint32_t *run_items = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_SK_ITEMS, sizeof(uint32_t) * ((size_t)n_tiles * ipt + 64), &p)); uint32_t *items = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); uint32_t *wg_hist = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_CURSOR, sizeof(uint64_t) * kPartGroups * kNumCoarse, &p)); uint64_t *wg_off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * kNumCoarse * 3, &p)); uint64_t *cnt = (uint64_t *)p, *base = cnt + kNumCoarse;
  KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags + 9, 0, sizeof(uint32_t), ctx->stream));
  if (fa) {
    ProfScope ps(ctx, "fasta_runs", fa->n_chars);
    KMI_HIP(ctx, hipMemsetAsync(ctx->d_totals + 6, 0, sizeof(uint64_t), ctx->stream));
    hipLaunchKernelGGL(fasta_runs_kernel, dim3(2048), dim3(256), 0, ctx->stream, reinterpret_cast<const uint32_t *>(fa->pk_break), fa->n_chars, fa->n_valid,
                       fa->n_cover, n_tiles, k, seg, ent, ent_cnt, stride, (unsigned long long *)(ctx->d_totals + 6), ctx->d_flags);
  } else {
    ProfScope ps(ctx, "fastq_list", n);
    const uint32_t wave_lds = LP::wave_lds_bytes(k, split);
    hipLaunchKernelGGL((fastq_list_kernel<NW, BITS, true>), dim3(kListGroups), dim3(kListThreads), wave_lds * (kListThreads / kWave), ctx->stream, in,
                       n_tiles, k, LP::max_runs(k, split), wave_lds, sc.line_base, ctx->d_flags, (void *)ent, ent_cnt, stride, seg);
  }
  {
    ProfScope ps(ctx, "sk_minimizer", n);
    hipLaunchKernelGGL((sk_minimizer_kernel<W>), dim3(kPartGroups), dim3(kSkThreads), 0, ctx->stream, in, n_tiles, k, (const uint32_t *)ent,
                       (const uint32_t *)ent_cnt, stride, ipt, items, run_items, wg_hist, ctx->d_flags);
  }
  {
    ProfScope ps(ctx, "sk_offsets", kNumCoarse);
    launch_rank_offsets(ctx->stream, (const uint32_t *)wg_hist, (uint32_t)kPartGroups, (uint32_t)kNumCoarse, cnt, base, wg_off);
  }
  KMI_HIP(ctx, hipGetLastError());
  uint64_t h_cnt[2 * kNumCoarse];
  uint32_t h_flags[10] = {0};
  KMI_HIP(ctx, hipMemcpyAsync(h_cnt, cnt, sizeof(uint64_t) * 2 * kNumCoarse, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(h_flags, ctx->d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, ctx->stream));
  if (fa) KMI_HIP(ctx, hipMemcpyAsync(&n, ctx->d_totals + 6, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // the seq / qual length rule rode on the list pass (flag word 0, bit 2: fastq_length_verdict's): the index stays as it was on a parse error
  if (!fa && (h_flags[0] & 4u)) return fastq_length_verdict(ctx);
  const uint32_t h_flag = h_flags[9];
  if (h_flag) return KMI_OK;              // a run with more items than a lane's list holds, or a tile with more than its share
  uint64_t R = 0;
  for (int c = 0; c < kNumCoarse; ++c) { R += h_cnt[c]; f->h_cnt[c] = h_cnt[c]; f->h_base[c] = h_cnt[kNumCoarse + c]; }
  uint64_t *rec_a = out;
  if (!out || R + 64 > out_cap) { KMI_TRY(ws_get(ctx, WS_KEYS_A, (R + 64) * 16, &p)); rec_a = (uint64_t *)p; }   // (64 records of slack behind the last one)
  {
    ProfScope ps(ctx, "sk_scatter", n);
    if (canonical)
      hipLaunchKernelGGL(sk_scatter_kernel<true>, dim3(kPartGroups), dim3(kSkThreads), 0, ctx->stream, in, n_tiles, k, (const uint32_t *)ent,
                         (const uint32_t *)ent_cnt, stride, ipt, (const uint32_t *)items, (const uint32_t *)run_items, (const uint64_t *)wg_off, rec_a, lp);
    else
      hipLaunchKernelGGL(sk_scatter_kernel<false>, dim3(kPartGroups), dim3(kSkThreads), 0, ctx->stream, in, n_tiles, k, (const uint32_t *)ent,
                         (const uint32_t *)ent_cnt, stride, ipt, (const uint32_t *)items, (const uint32_t *)run_items, (const uint64_t *)wg_off, rec_a, lp);
  }
  KMI_HIP(ctx, hipGetLastError());
  f->ok = true; f->recs = rec_a; f->n_records = R; f->n_kmers = n; f->wg_off = wg_off;
  return KMI_OK;
}

// The same front end in one pass over the bytes (kmi_front.h): FASTQ without a sequence filter. *took = false: the fast path
// declined (a shape it does not take, or something in the input it is not sure about) and nothing was changed -- the caller
// runs fastq_scan + sk_front_end, which also words parse errors.
// a build from host memory whose bytes have not been copied yet (kmi_index_build_host): whoever is about to read the input on the
// context's stream and is not the one-pass front end (which feeds itself, chunk by chunk) queues the whole copy first
static kmi_status feed_flush(kmi_ctx *ctx) {
  if (!ctx->feed_host) return KMI_OK;
  const uint8_t *h = ctx->feed_host;
  ctx->feed_host = nullptr;
  KMI_HIP(ctx, hipMemcpyAsync(ctx->feed_dev, h, ctx->feed_bytes, hipMemcpyHostToDevice, ctx->stream));
  return KMI_OK;
}

template <int W>
static kmi_status sk_front_fast(kmi_ctx *ctx, const kmi_config *cfg, const KShape &shape, const uint8_t *bytes_dev, size_t n_bytes, uint32_t lp, SkFront *f,
                                bool *took, uint64_t *out = nullptr, size_t out_cap = 0, bool local_fmt = false) {
  *took = false;
  f->ok = false;
  const bool fed = ctx->feed_host != nullptr && bytes_dev == ctx->feed_dev && n_bytes == ctx->feed_bytes;
  if (!fed) KMI_TRY(feed_flush(ctx));
  if (!ctx->front_fused || cfg->seq_format != KMI_FMT_FASTQ || cfg->seq_filter != KMI_SEQ_ALL || n_bytes < 64) return feed_flush(ctx);
  const uint32_t k = shape.k;
  const bool canonical = cfg->strand != KMI_STRAND_SINGLE;
  // ranges: one per resident wavefront of the front kernel when the input is large; never below the context's minimum
  if (!ctx->front_waves) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sk_front_kernel<19, false>, kFrThreads, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
    ctx->front_waves = (uint32_t)per_cu * (ctx->n_cus ? ctx->n_cus : 256u) * (uint32_t)kFrWaves;
  }
  // (fed from host memory: sixteen times the ranges -- what runs after the last byte has arrived is one range per wavefront slot,
  // and a range is worked through by one wavefront from end to end: 0.35 ms for 256 KB)
  uint64_t range_bytes = (n_bytes + ctx->front_waves * (fed ? 16u : 1u) - 1) / (ctx->front_waves * (fed ? 16u : 1u));
  range_bytes = (range_bytes + kFrStep - 1) / kFrStep * kFrStep;
  if (range_bytes < ctx->front_min_range) range_bytes = ctx->front_min_range;
  if (range_bytes > (8ull << 20)) range_bytes = 8ull << 20;
  const uint64_t n_ranges64 = (n_bytes + range_bytes - 1) / range_bytes;
  const uint32_t rpg = (uint32_t)((n_ranges64 + kPartGroups - 1) / kPartGroups);
  if (rpg > kFrMaxGroupRanges) return feed_flush(ctx);
  const uint32_t n_ranges = (uint32_t)n_ranges64;
  const uint32_t run_cap = (uint32_t)(range_bytes / 64) + 64u, item_cap = (uint32_t)(range_bytes / 8) + 64u;
  void *p;
  KMI_TRY(ws_get(ctx, WS_TILE_INFO, sizeof(FrRange) * (n_ranges + 1), &p)); FrRange *info = (FrRange *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_BASE, sizeof(uint64_t) * (kPartGroups + 1), &p)); uint64_t *group_runs = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_ENT_LIST, sizeof(uint32_t) * ((size_t)n_ranges * run_cap + 64), &p)); uint32_t *run_items = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_PK_STREAM, sizeof(uint32_t) * kFrRowDw * ((size_t)n_ranges * run_cap + 64), &p)); uint32_t *rows = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_SK_ITEMS, sizeof(uint32_t) * ((size_t)n_ranges * item_cap + 64), &p)); uint32_t *items = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); uint32_t *wg_hist = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_CURSOR, sizeof(uint64_t) * kPartGroups * kNumCoarse, &p)); uint64_t *wg_off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * kNumCoarse * 3, &p)); uint64_t *cnt = (uint64_t *)p, *base = cnt + kNumCoarse;
  // (flags 0 .. 15, the k-mer total and the group histograms, in one launch)
  hipLaunchKernelGGL(sk_zero_kernel, dim3(128), dim3(1024), 0, ctx->stream, wg_hist, (uint32_t)(kPartGroups * kNumCoarse),
                     reinterpret_cast<uint32_t *>(ctx->d_totals + 6), 2u, (uint32_t *)nullptr, 0u, ctx->d_flags, 0u, 16u, 0u, 0u);
  if (!fed) {
    ProfScope ps(ctx, "sk_front", n_bytes);
    const uint32_t wgs = (n_ranges + kFrWaves - 1) / kFrWaves;
    if (ctx->edge_records)
      hipLaunchKernelGGL((sk_front_kernel<W, true>), dim3(wgs), dim3(kFrThreads), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, range_bytes, n_ranges, k, is_rna(cfg),
                         run_cap, item_cap, rpg, info, run_items, rows, items, wg_hist, (unsigned long long *)(ctx->d_totals + 6), ctx->d_flags, 0u, 0xffffffffu);
    else
      hipLaunchKernelGGL((sk_front_kernel<W, false>), dim3(wgs), dim3(kFrThreads), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, range_bytes, n_ranges, k, is_rna(cfg),
                         run_cap, item_cap, rpg, info, run_items, rows, items, wg_hist, (unsigned long long *)(ctx->d_totals + 6), ctx->d_flags, 0u, 0xffffffffu);
  } else {
    // The input is still in host memory: its copy goes out in chunks on a stream of its own, all of them queued now, and behind every
    // chunk the front end takes the ranges whose bytes have arrived -- a range reads up to kFrOverrun + three steps behind its own end
    // (the lines it owns end there, and the two lines behind them; the next step's prefetch). Only the last chunk's ranges run after
    // the last byte is here; everything else of the front end is hidden behind the copy (BenchmarkKmerIndex.cpp:526-533 times from
    // bytes in host memory).
    ProfScope ps(ctx, "sk_front_fed", n_bytes);
    const uint8_t *host = ctx->feed_host;
    ctx->feed_host = nullptr;
    // chunk sizes: the front end works through a chunk about twenty times faster than the link delivers the next one, so every chunk
    // is a sixteenth of the one before it -- three or four copies in all (sixteen equal chunks cost 1.2 ms more than one copy: a gap
    // behind every copy, and the copies themselves a little slower)
    constexpr size_t kMaxChunks = sizeof(ctx->feed_ev) / sizeof(ctx->feed_ev[0]) - 1;
    size_t cut[kMaxChunks + 1];   // chunk c = bytes [cut[c], cut[c + 1])
    size_t n_chunks = 0;
    {
      size_t ratio = 16;
      if (const char *e = getenv("KMI_FEED_RATIO")) { const long v = atol(e); if (v >= 2 && v <= 64) ratio = (size_t)v; }
      size_t at = 0;
      cut[0] = 0;
      while (n_chunks + 1 < kMaxChunks && n_bytes - at > ctx->feed_min_chunk) {
        size_t len = (n_bytes - at) - (n_bytes - at) / ratio;
        len = len / 4096 * 4096;
        at += len; cut[++n_chunks] = at;
      }
      cut[++n_chunks] = n_bytes;
    }
    if (!ctx->copy_stream) KMI_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (size_t c = 0; c <= n_chunks; ++c) if (!ctx->feed_ev[c]) KMI_HIP(ctx, hipEventCreateWithFlags(&ctx->feed_ev[c], hipEventDisableTiming));
    // (the copy overwrites the input buffer: it starts behind what the build's stream has queued so far)
    KMI_HIP(ctx, hipEventRecord(ctx->feed_ev[n_chunks], ctx->stream));
    KMI_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->feed_ev[n_chunks], 0));
    for (size_t c = 0; c < n_chunks; ++c) {
      KMI_HIP(ctx, hipMemcpyAsync(const_cast<uint8_t *>(bytes_dev) + cut[c], host + cut[c], cut[c + 1] - cut[c], hipMemcpyHostToDevice, ctx->copy_stream));
      KMI_HIP(ctx, hipEventRecord(ctx->feed_ev[c], ctx->copy_stream));
    }
    const uint64_t margin = (uint64_t)kFrOverrun + 3ull * kFrStep;
    uint32_t r_done = 0;
    for (size_t c = 0; c < n_chunks; ++c) {
      KMI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->feed_ev[c], 0));
      const uint64_t arrived = cut[c + 1];
      uint32_t r_end = n_ranges;
      if (c + 1 < n_chunks) r_end = arrived > margin + range_bytes ? (uint32_t)((arrived - margin) / range_bytes) : 0u;   // ranges [0, r_end) end margin bytes before `arrived`
      if (r_end > n_ranges) r_end = n_ranges;
      if (r_end <= r_done) continue;
      const uint32_t wgs = (r_end - r_done + kFrWaves - 1) / kFrWaves;
      if (ctx->edge_records)
        hipLaunchKernelGGL((sk_front_kernel<W, true>), dim3(wgs), dim3(kFrThreads), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, range_bytes, n_ranges, k, is_rna(cfg),
                           run_cap, item_cap, rpg, info, run_items, rows, items, wg_hist, (unsigned long long *)(ctx->d_totals + 6), ctx->d_flags, r_done, r_end);
      else
        hipLaunchKernelGGL((sk_front_kernel<W, false>), dim3(wgs), dim3(kFrThreads), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, range_bytes, n_ranges, k, is_rna(cfg),
                           run_cap, item_cap, rpg, info, run_items, rows, items, wg_hist, (unsigned long long *)(ctx->d_totals + 6), ctx->d_flags, r_done, r_end);
      r_done = r_end;
    }
  }
  {
    ProfScope ps(ctx, "sk_offsets", kNumCoarse);
#ifdef KMI_FR_TIMING
    {
      unsigned long long t[8];
      hipStreamSynchronize(ctx->stream);
      hipMemcpy(t, ctx->d_flags + 48, sizeof(t), hipMemcpyDeviceToHost);
      fprintf(stderr, "sk_front wave clocks: wait-bytes %llu  eol %llu  lines %llu  load+pack %llu  walk %llu  final %llu  copy-out %llu  rest %llu\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
      hipMemset(ctx->d_flags + 48, 0, sizeof(t));
    }
#endif
    hipLaunchKernelGGL(sk_front_verify_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const FrRange *)info, n_ranges, rpg, group_runs, ctx->d_totals + 12, ctx->d_flags);
    launch_rank_offsets(ctx->stream, (const uint32_t *)wg_hist, (uint32_t)kPartGroups, (uint32_t)kNumCoarse, cnt, base, wg_off);
  }
  KMI_HIP(ctx, hipGetLastError());
  // The scatter pass is queued BEFORE the host looks at the front end's verdict (it checks the flag itself), so the GPU does not
  // idle through the host's round trip: its output then has to be sized by what the front end can produce at most (one item per
  // item slot), which is three times the usual -- done for inputs up to 8 GB, and when the caller did not bring the buffer.
  const uint64_t r_bound = (uint64_t)n_ranges * item_cap;
  bool early = !out && n_bytes <= (8ull << 30);
  uint64_t *rec_a = out;
  if (early) {
    // (the early buffer is sized for the most the front end can produce, three times the usual: where that does not fit -- several
    // indexes in one context, a smaller GPU -- the scatter pass waits for the host to size it by what was produced: ADVICE r3)
    if (ws_get(ctx, WS_KEYS_A, (r_bound + 64) * 16, &p) == KMI_OK) rec_a = (uint64_t *)p; else { early = false; ctx->err.clear(); }
  }
  auto launch_scatter = [&]() {
    ProfScope ps(ctx, "sk_scatter", n_bytes);
    if (canonical)
      hipLaunchKernelGGL(sk_scatter_rows_kernel<true>, dim3(kPartGroups), dim3(kFrScThreads), 0, ctx->stream, (const FrRange *)info, n_ranges, rpg, run_cap, item_cap, k,
                         (const uint32_t *)run_items, (const uint32_t *)rows, (const uint32_t *)items, (const uint64_t *)wg_off, rec_a, lp, local_fmt, (const uint32_t *)ctx->d_flags,
                         ctx->edge_records ? 1u : 0u);
    else
      hipLaunchKernelGGL(sk_scatter_rows_kernel<false>, dim3(kPartGroups), dim3(kFrScThreads), 0, ctx->stream, (const FrRange *)info, n_ranges, rpg, run_cap, item_cap, k,
                         (const uint32_t *)run_items, (const uint32_t *)rows, (const uint32_t *)items, (const uint64_t *)wg_off, rec_a, lp, local_fmt, (const uint32_t *)ctx->d_flags,
                         ctx->edge_records ? 1u : 0u);
  };
  // what the host needs of the front end comes back into PINNED memory: four copies queued back to back and one synchronisation
  // (into pageable memory every copy was a host round trip of its own: 0.1 ms of an idle GPU per build)
  uint64_t *const mail = ctx->h_totals + 16;
  const uint64_t *const h_cnt = mail, *const h_tot = mail + 2 * kNumCoarse + 1;
  KMI_HIP(ctx, hipMemcpyAsync(mail, cnt, sizeof(uint64_t) * 2 * kNumCoarse, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(mail + 2 * kNumCoarse, ctx->d_flags + 9, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(mail + 2 * kNumCoarse + 1, ctx->d_totals + 12, sizeof(uint64_t) * 3, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(mail + 2 * kNumCoarse + 4, ctx->d_totals + 6, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (early) {
    // the host waits for the read-backs only (an event behind them), not for the scatter pass queued behind that: what it queues
    // next -- the back end's tables and kernels -- is on the stream before the scatter pass has finished
    if (!ctx->ev_mail) KMI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_mail, hipEventDisableTiming));
    KMI_HIP(ctx, hipEventRecord(ctx->ev_mail, ctx->stream));
    launch_scatter();
    KMI_HIP(ctx, hipEventSynchronize(ctx->ev_mail));
  } else {
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  const uint32_t h_flag = *reinterpret_cast<const uint32_t *>(mail + 2 * kNumCoarse);
  const uint64_t n = mail[2 * kNumCoarse + 4];
  if (h_flag) {   // not this path's input (or not well-formed): the general path decides
    if (getenv("KMI_FRONT_DEBUG")) {
      uint32_t why = 0;
      (void)hipMemcpy(&why, ctx->d_flags + 10, sizeof(why), hipMemcpyDeviceToHost);
      fprintf(stderr, "sk_front declined: reasons 0x%x (2 first byte, 4 crowded lines, 8 inference, 16 marker / length, 32 run queue, 64 run capacity, 128 items per run, 256 item capacity, 512 / 1024 edge records: a read of several runs / a base that is not A C G T; 0 = the chain of line indices)\n", why);
    }
    KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags + 9, 0, 2 * sizeof(uint32_t), ctx->stream));
    return KMI_OK;
  }
  *took = true;
  uint64_t R = 0;
  for (int c = 0; c < kNumCoarse; ++c) { R += h_cnt[c]; f->h_cnt[c] = h_cnt[c]; f->h_base[c] = h_cnt[kNumCoarse + c]; }
  f->n_records = R; f->n_kmers = n; f->wg_off = wg_off; f->recs = nullptr; f->n_seqs = (h_tot[0] + 2) / 4;
  if (n == 0) { f->ok = true; return KMI_OK; }
  if (!early) {
    if (!out || R + 64 > out_cap) { KMI_TRY(ws_get(ctx, WS_KEYS_A, (R + 64) * 16, &p)); rec_a = (uint64_t *)p; }
    launch_scatter();
  }
  KMI_HIP(ctx, hipGetLastError());
  f->ok = true; f->recs = rec_a;
  return KMI_OK;
}
template <int W>
static kmi_status sk_front_fast_any(kmi_ctx *ctx, const kmi_config *cfg, const KShape &shape, const uint8_t *bytes_dev, size_t n_bytes, uint32_t lp, SkFront *f,
                                    bool *took, uint64_t *out = nullptr, size_t out_cap = 0) {
  return sk_front_fast<W>(ctx, cfg, shape, bytes_dev, n_bytes, lp, f, took, out, out_cap);
}

// Back end: records grouped by coarse bucket (rec_a; group c = [h_base[c], h_base[c] + h_cnt[c]), written by kPartGroups
// workgroups at wg_off[g][c]) -> fine buckets -> sk_reduce -> the index (layout W | lp << 8), or added to what it holds.
template <int W>
static kmi_status sk_back_end(kmi_index *idx, const uint64_t *rec_a, uint64_t R, const uint64_t *h_cnt, const uint64_t *h_base, const uint64_t *wg_off,
                              uint64_t n, uint32_t lp, bool exact = false, bool local_fmt = false) {
  // local_fmt: the records carry nine further hash bits where the coarse bits were (sk_scatter_rows_kernel); else three, above the bucket bits
  // exact = false (an index without entries): the fine buckets get room instead of exact offsets, so the records are not read an
  // extra time to be counted (sk_scatter_fine_slack_kernel); a bucket that outgrows its room sends the build through here again
  // with exact = true
  constexpr int NW = 1;
  kmi_ctx *ctx = idx->ctx;
  const bool slack = !exact && ctx->sk_slack && (!idx->has_data || idx->n_entries == 0) && R >= 4096;   // (room_x64 below: <= 6.1 x the records)
  ctx->sk_left.valid = false;
  const uint32_t k = idx->shape.k;
  const bool canonical = idx->cfg.strand != KMI_STRAND_SINGLE;
  void *p;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); uint32_t *wg_hist = (uint32_t *)p;   // (not read by fine_offsets)
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * kNumCoarse * 3, &p)); uint64_t *cend = (uint64_t *)p + 2 * kNumCoarse;
  KMI_TRY(ws_get(ctx, WS_HIST, sizeof(uint32_t) * kNumFine * kFineParts + sizeof(uint64_t) * ((kNumFine + 1) * 2 + kNumFine * kFineParts + kNumCoarse) + 256, &p));
  uint32_t *fine_hist = (uint32_t *)p;
  uint64_t *fine_off = (uint64_t *)((char *)p + sizeof(uint32_t) * kNumFine * kFineParts);
  uint64_t *part_off = fine_off + 2 * (kNumFine + 1);
  uint64_t *coarse_base = part_off + (uint64_t)kNumFine * kFineParts;
  uint64_t h_end[kNumCoarse];
  for (int c = 0; c < kNumCoarse; ++c) h_end[c] = h_base[c] + h_cnt[c];
  KMI_HIP(ctx, hipMemcpyAsync(cend, h_end, sizeof(h_end), hipMemcpyHostToDevice, ctx->stream));   // (h_end outlives the copy: this function synchronises before it returns)
  // the slack layout: every fine bucket of coarse bucket c has room for room_x64 / 64 of c's mean share + 64, in steps of 8 records
  // How uneven minimizer buckets are depends on the minimizer length m = k - W + 1 (few canonical m-mers: a few of them take most
  // minima) and on the rank bits of a build over ranks (a rank's buckets stand for fewer minimizer classes of a larger genome).
  // Fullest fine bucket / its coarse bucket's mean, measured on synthetic reads: m = 11: 1.9 - 2.6, m = 12: 1.63, m >= 13: 1.36 -
  // 1.54; for m = 13 and lp = 0 / 1 / 2 / 3: 1.45 / < 1.6 / 1.79 / 2.27.
  const uint32_t m_len = idx->shape.k - (uint32_t)W + 1u;
  const uint64_t room_m = m_len <= 11u ? 26u : (m_len == 12u ? 18u : 14u);            // in eighths of the mean
  static const uint64_t room_lp[4] = {8, 10, 12, 15};                                   // in eighths
  const uint64_t room_x64 = room_m * room_lp[lp < 4u ? lp : 3u];
  uint64_t h_region[kNumCoarse], total_b = 0;
  uint32_t h_cap[kNumCoarse];
  for (int c = 0; c < kNumCoarse; ++c) {
    const uint64_t cap = ((h_cnt[c] * room_x64 / 64 + kSubPerCoarse - 1) / kSubPerCoarse + 64 + 7) / 8 * 8;
    h_cap[c] = (uint32_t)cap; h_region[c] = total_b; total_b += cap * kSubPerCoarse;
  }
  KMI_TRY(ws_get(ctx, WS_KEYS_B, ((slack ? total_b : R) + 64) * 16, &p)); uint64_t *rec_b = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_SPLIT_OFF, sizeof(uint32_t) * kNumFine * kFineParts + sizeof(uint64_t) * ((kNumFine + 1) + kNumFine * kFineParts + kNumCoarse) + 256, &p));
  uint32_t *fine_kmers = (uint32_t *)p;
  uint64_t *kmer_off = (uint64_t *)((char *)p + sizeof(uint32_t) * kNumFine * kFineParts);
  uint64_t *k_part = kmer_off + (kNumFine + 1), *k_base = k_part + (uint64_t)kNumFine * kFineParts;
  uint64_t *d_region = nullptr; uint32_t *d_cap = nullptr, *fine_cnt = nullptr;
  if (slack) {
    KMI_TRY(ws_get(ctx, WS_BUCKET_OFF, sizeof(uint64_t) * kNumCoarse + sizeof(uint32_t) * kNumCoarse + 64, &p));
    d_region = (uint64_t *)p; d_cap = (uint32_t *)(d_region + kNumCoarse);
    fine_cnt = fine_hist;   // (the first half of the histogram block: records per fine bucket, counted as they are appended)
    KMI_HIP(ctx, hipMemcpyAsync(d_region, h_region, sizeof(h_region), hipMemcpyHostToDevice, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(d_cap, h_cap, sizeof(h_cap), hipMemcpyHostToDevice, ctx->stream));
    // (the append counters, the k-mer counts, the reduce's votes -- flags 16 .. 33 --, the room flag 34 and the queue word 40)
    hipLaunchKernelGGL(sk_zero_kernel, dim3(96), dim3(1024), 0, ctx->stream, fine_cnt, (uint32_t)kNumFine, fine_kmers, (uint32_t)(kNumFine * kFineParts),
                       (uint32_t *)nullptr, 0u, ctx->d_flags, 16u, 35u, 40u, 48u);
    {
      ProfScope ps(ctx, "sk_scatter_fine", R);
      if (ctx->sk_fine_lines && !ctx->sk_reduce2)   // whole lines + pad records (sk_reduce2 reads a bucket into its stage as it lies: no pads for it)
        hipLaunchKernelGGL(sk_scatter_fine_slack_lines_kernel, dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, (const uint64_t *)rec_a, rec_b,
                           (const uint64_t *)wg_off, (const uint64_t *)cend, (uint32_t)kPartGroups, (const uint64_t *)d_region, (const uint32_t *)d_cap,
                           fine_cnt, fine_kmers, ctx->d_flags);
      else
        hipLaunchKernelGGL(sk_scatter_fine_slack_kernel, dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, (const uint64_t *)rec_a, rec_b,
                           (const uint64_t *)wg_off, (const uint64_t *)cend, (uint32_t)kPartGroups, (const uint64_t *)d_region, (const uint32_t *)d_cap,
                           fine_cnt, fine_kmers, ctx->d_flags);
    }
    {
      ProfScope ps(ctx, "fine_offsets", kNumFine);
      launch_fine_offsets(ctx, fine_kmers, kmer_off, k_part, k_base);
    }
  } else {
    {
      ProfScope ps(ctx, "sk_fine_count", R);
      hipLaunchKernelGGL(sk_fine_count_kernel, dim3(kNumCoarse * kFineParts), dim3(1024), 0, ctx->stream, (const uint64_t *)rec_a,
                         (const uint64_t *)wg_off, (const uint64_t *)cend, (uint32_t)kPartGroups, fine_hist, fine_kmers);
    }
    {
      ProfScope ps(ctx, "fine_offsets", kNumFine);
      launch_fine_offsets(ctx, fine_hist, fine_off, part_off, coarse_base);
      launch_fine_offsets(ctx, fine_kmers, kmer_off, k_part, k_base);
    }
    {
      ProfScope ps(ctx, "sk_scatter_fine", R);
      hipLaunchKernelGGL((scatter_fine_kernel<1, 2, 1>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, (const uint64_t *)rec_a,
                         rec_b, idx->shape, (const uint64_t *)fine_off, (const uint64_t *)part_off, (const uint64_t *)wg_off,
                         (uint32_t)kPartGroups, (int)BUCKET_REC, 0u);
    }
  }
  // per bucket: as many output slots as it has k-mers (bucket_reduce_kernel's contract), compacted by adopt_tmp
  KMI_TRY(ws_get(ctx, WS_TMP_KEYS, (n + 64) * sizeof(uint64_t), &p)); uint64_t *tmp_keys = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_TMP_VALS, (n + 64) * sizeof(uint32_t), &p)); uint32_t *tmp_vals = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_BUCKET_CNT, sizeof(uint32_t) * kNumFine, &p)); uint32_t *out_cnt = (uint32_t *)p;
  if (!slack) KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags + 16, 0, sizeof(uint32_t) * 18, ctx->stream));   // the pass-structure votes of sk_reduce (the slack path has zeroed them)
  {
    const uint32_t nmax = sk_nmax_of(k);
    // persistent workgroups (one per CU: each takes the whole LDS) pull the buckets from a queue word
    uint32_t *queue = ctx->d_flags + 40;
    if (!slack) KMI_HIP(ctx, hipMemsetAsync(queue, 0, sizeof(uint32_t) * 8, ctx->stream));   // (+ the redo pass's queue word, the redo count, a spare)
#ifndef KMI_SK_WGS_PER_CU
#define KMI_SK_WGS_PER_CU 1
#endif
    const uint32_t wgs = (ctx->n_cus ? ctx->n_cus : 256u) * KMI_SK_WGS_PER_CU;
    // KMI_SK_REDUCE=2: sk_reduce2 (wavefront-private tables over the sorted bins of a bucket, kmi_reduce2.h) takes every bucket first;
    // what it puts on its redo list -- a bin that does not fit a private table, a bucket that does not fit the stage -- goes through
    // sk_reduce (shared tables, passes) behind it, which reads the list's length on the device. Built for the round-3 verdict and
    // measured slower than sk_reduce on every input tried: the default is sk_reduce alone.
    const bool two = ctx->sk_reduce2;
    const uint32_t *redo_list = nullptr, *redo_cnt = nullptr;
    if (two) {
      KMI_TRY(ws_get(ctx, WS_REDO, sizeof(uint32_t) * kNumFine, &p)); uint32_t *rl = (uint32_t *)p;
      redo_list = rl; redo_cnt = ctx->d_flags + 42;
      // window of a batch: 64 records while a window's k-mers are mostly copies of one another; smaller where the last build found
      // little duplication (a window's distinct k-mers have to fit a private table)
      const float dupl = ctx->sk_inv_dup;
      const uint32_t win = ctx->sk_r2_win ? ctx->sk_r2_win : (dupl <= 0.2f ? 32u : (dupl <= 0.45f ? 24u : 16u));
      const uint32_t xshift = local_fmt ? 53u : 61u, xbits = local_fmt ? (uint32_t)(kR2BinBitsMax - 3) : 3u;
      ProfScope ps(ctx, "sk_reduce", n);   // (the profile keeps the name of the kernel it replaced)
#define KMI_SK_REDUCE2(CANON, OWN, SPECIAL)                                                                                              \
      hipLaunchKernelGGL((sk_reduce2_kernel<CANON, OWN, SPECIAL>), dim3(wgs * kR2PerCu), dim3(KMI_R2_WAVES * 64), 0, ctx->stream, (const uint64_t *)rec_b, \
                         (const uint64_t *)fine_off, k, (const uint64_t *)kmer_off, tmp_keys, tmp_vals, out_cnt, ctx->d_flags, queue, (uint32_t)kNumFine, \
                         xshift, xbits, win, (const uint64_t *)d_region, (const uint32_t *)d_cap, (const uint32_t *)fine_cnt, rl, ctx->d_flags + 42)
      // (second parameter: bytes of unit marks of a step -- 64 records of (nmax + 1) / 2 units)
      if (k == 32u) { if (canonical) KMI_SK_REDUCE2(true, 64 * 11, true); else KMI_SK_REDUCE2(false, 64 * 11, true); }   // (a 32-mer can equal the empty marker)
      else if (nmax <= 22u) { if (canonical) KMI_SK_REDUCE2(true, 64 * 11, false); else KMI_SK_REDUCE2(false, 64 * 11, false); }
      else { if (canonical) KMI_SK_REDUCE2(true, 64 * 16, false); else KMI_SK_REDUCE2(false, 64 * 16, false); }
#undef KMI_SK_REDUCE2
      queue = ctx->d_flags + 41;
#ifdef KMI_R2_TIMING
      {
        unsigned long long t[8];
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(t, ctx->d_flags + 48, sizeof(t), hipMemcpyDeviceToHost);
        fprintf(stderr, "sk_reduce2 wave clocks: count %llu  scan+place %llu  grab/wait-in-batches %llu  dedupe+expand %llu  emit %llu  barrier-after-place %llu  barrier-after-batches %llu  between %llu\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
        hipMemset(ctx->d_flags + 48, 0, sizeof(t));
        unsigned long long d[128];
        hipMemcpyFromSymbol(d, HIP_SYMBOL(g_r2_dbg), sizeof(d));
        fprintf(stderr, "  per wavefront number (Mcycles in the batch phase / batches / steps / expand iterations):");
        for (int w = 0; w < KMI_R2_WAVES; ++w) fprintf(stderr, "  %d: %.0f/%llu/%llu/%llu", w, d[w] / 1e6, d[32 + w], d[64 + w], d[96 + w]);
        fprintf(stderr, "\n");
        memset(d, 0, sizeof(d));
        hipMemcpyToSymbol(HIP_SYMBOL(g_r2_dbg), d, sizeof(d));
      }
#endif
      if (getenv("KMI_R2_DEBUG")) {
        uint32_t st[8];
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(st, ctx->d_flags + 40, sizeof(st), hipMemcpyDeviceToHost);
        fprintf(stderr, "sk_reduce2: redo list %u buckets (stage: gave up %u, a part too large %u; batches that did not fit their table %u; buckets in parts %u), window %u\n", st[2], st[4], st[6], st[5], st[7], win);
      }
    }
    ProfScope ps(ctx, two ? "sk_reduce_redo" : "sk_reduce", n);
#define KMI_SK_REDUCE(CANON, OWN, SPECIAL)                                                                                               \
    hipLaunchKernelGGL((sk_reduce_kernel<CANON, OWN, SPECIAL>), dim3(wgs), dim3(KMI_SK_NT), 0, ctx->stream, (const uint64_t *)rec_b,        \
                       (const uint64_t *)fine_off, k, (const uint64_t *)kmer_off, tmp_keys, tmp_vals, out_cnt, ctx->d_flags, queue, (uint32_t)kNumFine, \
                       ctx->sk_level_hint, lp, ctx->sk_inv_dup, (const uint64_t *)d_region, (const uint32_t *)d_cap, (const uint32_t *)fine_cnt, redo_list, redo_cnt)
    if (k == 32u) { if (canonical) KMI_SK_REDUCE(true, 64 * 21, true); else KMI_SK_REDUCE(false, 64 * 21, true); }   // (a 32-mer can equal the empty marker)
    else if (nmax <= 21u) { if (canonical) KMI_SK_REDUCE(true, 64 * 21, false); else KMI_SK_REDUCE(false, 64 * 21, false); }
    else if (nmax <= 24u) { if (canonical) KMI_SK_REDUCE(true, 64 * 24, false); else KMI_SK_REDUCE(false, 64 * 24, false); }
    else { if (canonical) KMI_SK_REDUCE(true, 64 * 32, false); else KMI_SK_REDUCE(false, 64 * 32, false); }
#undef KMI_SK_REDUCE
  }
  KMI_HIP(ctx, hipGetLastError());
#ifdef KMI_SK_TIMING
  {
    unsigned long long t[8];
    hipStreamSynchronize(ctx->stream);
    hipMemcpy(t, ctx->d_flags + 48, sizeof(t), hipMemcpyDeviceToHost);
    fprintf(stderr, "sk_reduce wave clocks: A-T1 %llu  A-direct %llu  A-wait %llu  B-work %llu  B-wait %llu  emit %llu  between %llu\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6]);
    hipMemset(ctx->d_flags + 48, 0, sizeof(t));
  }
#endif
  // a fine bucket that outgrew its room? (its records beyond the room were not written: this attempt's result is void; read with
  // the level votes below)
  if (slack) KMI_HIP(ctx, hipMemcpyAsync(ctx->h_totals + 14, ctx->d_flags + 34, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  // the level most buckets ended at: where the buckets of the NEXT build (batch, step) of this context start. The copy is queued
  // here and read after the synchronisation adopt_tmp needs anyway (one host round trip less per build).
  KMI_HIP(ctx, hipMemcpyAsync(ctx->h_totals + 8 /* pinned; words 8..12 */, ctx->d_flags + 16, sizeof(uint32_t) * 9, hipMemcpyDeviceToHost, ctx->stream));
  auto read_levels = [&]() {
    const uint32_t *lv = reinterpret_cast<const uint32_t *>(ctx->h_totals + 8);
    uint64_t seen = 0; uint32_t best = 0;
    for (uint32_t l = 0; l < 9; ++l) { seen += lv[l]; if (lv[l] > lv[best]) best = l; }
    // (level 0 finishes are only recorded once the hint is non-zero: no record at all means "everything fit at level 0")
    if (seen == 0) ctx->sk_level_hint = 0;
    else if (ctx->sk_level_hint == 0) { uint64_t up = seen; ctx->sk_level_hint = (up * 2 > (uint64_t)kNumFine) ? best : 0u; }
    else ctx->sk_level_hint = best;
  };
  const uint32_t layout = (uint32_t)W | (lp << 8);
  if (!idx->has_data || idx->n_entries == 0) {
    // the index IS the reduce output: entries grouped by minimizer bucket. Queries partition their keys by the same function
    // (fine15_of_key); whatever needs the placement-hash layout converts the entries once (ensure_layout).
    KMI_TRY((adopt_tmp<NW>(idx, tmp_keys, tmp_vals, kmer_off, nullptr, out_cnt, false, n, true)));
    if (slack && (*reinterpret_cast<const uint32_t *>(ctx->h_totals + 14) || getenv("KMI_SLACK_DEBUG"))) {
      if (getenv("KMI_SLACK_DEBUG")) {   // how full the fullest fine bucket was
        std::vector<uint32_t> fc(kNumFine);
        (void)hipMemcpy(fc.data(), fine_cnt, sizeof(uint32_t) * kNumFine, hipMemcpyDeviceToHost);
        double worst = 0; uint32_t over = 0;
        for (int f = 0; f < kNumFine; ++f) {
          const double mean = (double)h_cnt[f / kSubPerCoarse] / kSubPerCoarse;
          if (mean > 0 && fc[f] / mean > worst) worst = fc[f] / mean;
          over += fc[f] > h_cap[f / kSubPerCoarse];
        }
        fprintf(stderr, "fine buckets with room: %u of %d outgrew it, the fullest holds %.3f x its coarse bucket's mean (lp %u, %llu records)\n", over,
                (int)kNumFine, worst, lp, (unsigned long long)R);
      }
    }
    if (slack && *reinterpret_cast<const uint32_t *>(ctx->h_totals + 14)) {
      KMI_TRY(kmi_index_clear(idx));
      return sk_back_end<W>(idx, rec_a, R, h_cnt, h_base, wg_off, n, lp, true, local_fmt);
    }
    read_levels();
    idx->layout_w = layout;
    if (n) ctx->sk_inv_dup = (float)((double)idx->n_entries / (double)n);   // where the buckets of the next build start (sk_reduce_kernel)
    ctx->sk_left.recs = rec_b; ctx->sk_left.rec_off = slack ? nullptr : fine_off; ctx->sk_left.region = d_region; ctx->sk_left.cap = d_cap;
    ctx->sk_left.cnt = fine_cnt; ctx->sk_left.valid = true;
    return KMI_OK;
  }
  // the index holds entries already: the new ones become a scratch index, whose pairs are added to the old
  kmi_index scratch;
  scratch.ctx = ctx; scratch.cfg = idx->cfg; scratch.shape = idx->shape; scratch.val_words = 0; scratch.saturating = idx->saturating;
  kmi_status st = adopt_tmp<NW>(&scratch, tmp_keys, tmp_vals, kmer_off, nullptr, out_cnt);
  if (st == KMI_OK) read_levels();
  if (st == KMI_OK && n) ctx->sk_inv_dup = (float)((double)scratch.n_entries / (double)n);
  if (st == KMI_OK && scratch.n_entries) {
    st = ws_get(ctx, WS_OUTPUT, (scratch.n_entries + 64) * 2 * sizeof(uint64_t), &p);   // (WS_INPUT2 is the re-layout's)
    if (st == KMI_OK) {
      hipLaunchKernelGGL((zip_pairs_kernel<NW>), dim3(2048), dim3(256), 0, ctx->stream, (const uint64_t *)scratch.keys, (const uint32_t *)scratch.vals,
                         scratch.n_entries, (uint64_t *)p);
      st = index_insert_pairs(idx, (const uint64_t *)p, (size_t)scratch.n_entries, false, true);
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  free_index_arrays(&scratch);
  return st;
}

template <int W>
static kmi_status build_superkmer_w(kmi_index *idx, const FastqScan &sc, bool *done) {
  *done = false;
  SkFront f;
  KMI_TRY((sk_front_end<W>(idx->ctx, &idx->cfg, idx->shape, sc, 0u, &f)));
  if (!f.ok) return KMI_OK;
  *done = true;
  return sk_back_end<W>(idx, f.recs, f.n_records, f.h_cnt, f.h_base, f.wg_off, f.n_kmers, 0u);
}

// FASTA count index through super-k-mers: the compacted character stream of kmi_fasta.hip is "FASTQ without line roles", its
// run list comes from the break bitmap (fasta_runs_kernel), everything behind it is the FASTQ build's
template <int W>
static kmi_status build_superkmer_fasta_w(kmi_index *idx, const FastaScan &fa, bool *done) {
  *done = false;
  SkFront f;
  FastqScan none{};
  KMI_TRY((sk_front_end<W>(idx->ctx, &idx->cfg, idx->shape, none, 0u, &f, nullptr, 0, &fa)));
  if (!f.ok) return KMI_OK;
  *done = true;
  if (f.n_records == 0) return KMI_OK;
  return sk_back_end<W>(idx, f.recs, f.n_records, f.h_cnt, f.h_base, f.wg_off, f.n_kmers, 0u);
}

// ---- a build over 2^lp ranks through exchanged super-k-mer records ------------------------------------------------------
// produce: the front end; the records leave grouped by owner rank (the owner of a record = the top lp bits of its bucket bits;
// groups of 256 / nranks consecutive coarse buckets), send_counts = records per rank.
template <int W>
static kmi_status sk_produce_w(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes, uint32_t nranks, const uint64_t **recs_out, uint64_t *n_records,
                               uint64_t *send_counts, int *produced, uint64_t *out, size_t out_cap) {
  kmi_ctx *ctx = idx->ctx;
  *produced = 0; *recs_out = nullptr; *n_records = 0;
  for (uint32_t r = 0; r < nranks; ++r) send_counts[r] = 0;
  if (n_bytes == 0) { *produced = 1; return KMI_OK; }
  const uint32_t lp = 31u - (uint32_t)__builtin_clz(nranks);
  SkFront f;
  bool took = false;
  KMI_TRY((sk_front_fast<W>(ctx, &idx->cfg, idx->shape, bytes_dev, n_bytes, lp, &f, &took, out, out_cap)));
  if (took && f.ok && f.n_kmers == 0) { *produced = 1; return KMI_OK; }
  if (!took || !f.ok) {
    FastqScan sc;
    KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));   // (the scan's 16-byte loads)
    KMI_TRY(fastq_scan(ctx, &idx->cfg, bytes_dev, n_bytes, &sc, false));
    if (sc.n_tuples == 0) { KMI_TRY(fastq_length_verdict(ctx)); *produced = 1; return KMI_OK; }
    KMI_TRY((sk_front_end<W>(ctx, &idx->cfg, idx->shape, sc, lp, &f, out, out_cap)));
    if (!f.ok) return KMI_OK;
  }
  const uint32_t per = (uint32_t)kNumCoarse / nranks;
  for (uint32_t c = 0; c < (uint32_t)kNumCoarse; ++c) send_counts[c / per] += f.h_cnt[c];
  *recs_out = f.recs; *n_records = f.n_records; *produced = 1;
  // (the records are being written on the context's stream: whoever reads them is ordered behind it -- a collective on a stream
  // that waits for this one, the library's own exchange, or kmi_copy_on_device)
  return KMI_OK;
}

// consume: what arrived (a flat record array; inside every source's part the records are grouped by the sender's buckets, whose
// low lp bits are not the receiver's coarse bucket bits) is sorted by coarse bucket once, then takes the back end
template <int W>
static kmi_status sk_consume_w(kmi_index *idx, const uint64_t *recs_dev, uint64_t R, uint32_t nranks) {
  kmi_ctx *ctx = idx->ctx;
  if (R == 0) return KMI_OK;
  const uint32_t lp = 31u - (uint32_t)__builtin_clz(nranks);
  void *p;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); uint32_t *wg_hist = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_CURSOR, sizeof(uint64_t) * kPartGroups * kNumCoarse, &p)); uint64_t *wg_off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * kNumCoarse * 3, &p)); uint64_t *cnt = (uint64_t *)p, *base = cnt + kNumCoarse;
  KMI_TRY(ws_get(ctx, WS_KEYS_A, (R + 64) * 16, &p)); uint64_t *rec_a = (uint64_t *)p;
  KMI_HIP(ctx, hipMemsetAsync(ctx->d_totals + 6, 0, sizeof(uint64_t), ctx->stream));
  {
    ProfScope ps(ctx, "sk_recv_hist", R);
    hipLaunchKernelGGL(sk_recv_hist_kernel, dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, recs_dev, R, wg_hist, (unsigned long long *)(ctx->d_totals + 6));
  }
  {
    ProfScope ps(ctx, "sk_offsets", kNumCoarse);
    launch_rank_offsets(ctx->stream, (const uint32_t *)wg_hist, (uint32_t)kPartGroups, (uint32_t)kNumCoarse, cnt, base, wg_off);
  }
  BucketFn fn; fn.mode = BUCKET_REC_COARSE; fn.shape = idx->shape; fn.dist_hash = 0; fn.farm_ndebug = false; fn.nranks = 1; fn.sub = 1;
  {
    ProfScope ps(ctx, "sk_recv_scatter", R);
    hipLaunchKernelGGL((scatter_chunks_kernel<1, 2, 1>), dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, recs_dev, R, rec_a, idx->shape, 0u, false, fn,
                       (const uint64_t *)wg_off);
  }
  KMI_HIP(ctx, hipGetLastError());
  uint64_t h[2 * kNumCoarse], n_kmers = 0;
  KMI_HIP(ctx, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(&n_kmers, ctx->d_totals + 6, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  KMI_TRY((sk_back_end<W>(idx, rec_a, R, h, h + kNumCoarse, wg_off, n_kmers, lp)));
  idx->owner_lp = lp;
  return KMI_OK;
}

// which W the super-k-mer paths take for this index (0: they do not apply)
static uint32_t sk_width_of(const kmi_index *idx) {
  return (idx->val_words == 0 && idx->shape.n_words == 1 && idx->shape.bits == 2 && idx->ctx->fused_superkmer && idx->cfg.seq_format == KMI_FMT_FASTQ)
             ? sk_window_of(idx->shape.k) : 0u;
}

template <int W>
static kmi_status build_superkmer_fast_w(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes, bool *done) {
  *done = false;
  SkFront f;
  bool took = false;
  KMI_TRY((sk_front_fast<W>(idx->ctx, &idx->cfg, idx->shape, bytes_dev, n_bytes, 0u, &f, &took, nullptr, 0, true)));
  if (!took || !f.ok) return KMI_OK;
  *done = true;
  if (f.n_kmers == 0) return KMI_OK;
  return sk_back_end<W>(idx, f.recs, f.n_records, f.h_cnt, f.h_base, f.wg_off, f.n_kmers, 0u, false, true);
}

static kmi_status index_build_fused(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes) {
  const uint32_t w = (idx->shape.n_words == 1 && idx->shape.bits == 2 && idx->ctx->fused_superkmer) ? sk_window_of(idx->shape.k) : 0u;
  if (w) {
    kmi_ctx *ctx = idx->ctx;
    bool done = false;
    // one pass over the bytes when the input allows it (kmi_front.h) ...
    kmi_status st = w == 19u ? build_superkmer_fast_w<19>(idx, bytes_dev, n_bytes, &done) : (w == 13u ? build_superkmer_fast_w<13>(idx, bytes_dev, n_bytes, &done)
                  : (w == 11u ? build_superkmer_fast_w<11>(idx, bytes_dev, n_bytes, &done) : build_superkmer_fast_w<7>(idx, bytes_dev, n_bytes, &done)));
    if (st != KMI_OK || done) return st;
    // ... else the general front end: scan (which words parse errors), list, minimizer
    FastqScan sc;
    KMI_TRY(fastq_scan(ctx, &idx->cfg, bytes_dev, n_bytes, &sc, false));
    if (sc.n_tuples == 0) return KMI_OK;
    st = w == 19u ? build_superkmer_w<19>(idx, sc, &done) : (w == 13u ? build_superkmer_w<13>(idx, sc, &done)
       : (w == 11u ? build_superkmer_w<11>(idx, sc, &done) : build_superkmer_w<7>(idx, sc, &done)));
    if (st != KMI_OK || done) return st;
  }
  KMI_DISPATCH(idx->shape, build_fused_impl, idx, bytes_dev, n_bytes);
}

static kmi_status index_insert(kmi_index *idx, const uint64_t *keys_dev, size_t n, bool transform) {
  KMI_DISPATCH(idx->shape, insert_impl, idx, keys_dev, n, transform);
}

static void free_index_arrays(kmi_index *idx) {
  kmi_ctx *ctx = idx->ctx;
  pool_free(ctx, idx->keys, idx->keys_bytes);
  pool_free(ctx, idx->vals, idx->vals_bytes);
  pool_free(ctx, idx->mvals, idx->mvals_bytes);
  pool_free(ctx, idx->bucket_off, kOffBytes);
  if (idx->bucket_cnt) pool_free(ctx, idx->bucket_cnt, sizeof(uint32_t) * kNumFine);
  if (idx->dense_off) pool_free(ctx, idx->dense_off, kOffBytes);
  idx->bucket_cnt = nullptr; idx->dense_off = nullptr;
  idx->keys = nullptr; idx->vals = nullptr; idx->mvals = nullptr; idx->bucket_off = nullptr;
  idx->keys_bytes = idx->vals_bytes = idx->mvals_bytes = 0;
}

// pairs (key words, count word) out of the index arrays
template <int NW>
__global__ __launch_bounds__(256) void zip_pairs_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t n,
                                                       uint64_t *__restrict__ recs) {
  constexpr int RW = NW + 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int w = 0; w < NW; ++w) recs[i * RW + w] = keys[i * NW + w];
    recs[i * RW + NW] = vals[i];
  }
}

// A count index of one-word 2-bit k-mers is laid out by placement hash or by minimizer bucket (kmi_index::layout_w). The
// super-k-mer build leaves the second, everything that arrives as k-mers or pairs is partitioned by the first; an entry
// point that needs the other layout re-partitions the (distinct) entries once: zip -> partition by the target's bucket
// function -> unzip.
template <int NW, int BITS>
static kmi_status relayout_impl(kmi_index *idx, uint32_t target_w) {
  kmi_ctx *ctx = idx->ctx;
  const uint64_t n = idx->n_entries;
  void *p;
  KMI_TRY(ws_get(ctx, WS_INPUT2, (n + 64) * (NW + 1) * sizeof(uint64_t), &p)); uint64_t *recs = (uint64_t *)p;
  {
    ProfScope ps(ctx, "zip_pairs", n);
    hipLaunchKernelGGL((zip_pairs_kernel<NW>), dim3(2048), dim3(256), 0, ctx->stream, (const uint64_t *)idx->keys, (const uint32_t *)idx->vals, n, recs);
  }
  Partitioned part;
  KMI_TRY((partition_impl<NW, BITS, 1>(ctx, &idx->cfg, idx->shape, recs, (size_t)n, false, WS_KEYS_A, WS_KEYS_B, &part, target_w)));
  {   // same number of entries: the arrays are rewritten in place
    ProfScope ps(ctx, "unzip_pairs", n);
    hipLaunchKernelGGL((unzip_pairs_kernel<NW>), dim3(2048), dim3(256), 0, ctx->stream, (const uint64_t *)part.keys, n, idx->keys, idx->vals);
  }
  KMI_HIP(ctx, hipMemcpyAsync(idx->bucket_off, part.fine_off, kOffBytes, hipMemcpyDeviceToDevice, ctx->stream));
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  idx->layout_w = target_w;
  return KMI_OK;
}
static kmi_status relayout_dispatch(kmi_index *idx, uint32_t target_w) { KMI_DISPATCH(idx->shape, relayout_impl, idx, target_w); }
// sparse -> dense: the compaction pass the build left out (bucket_compact_kernel into arrays of exactly n_entries)
template <int NW, int BITS>
static kmi_status densify_impl(kmi_index *idx) {
  kmi_ctx *ctx = idx->ctx;
  const uint64_t total = idx->n_entries;
  uint64_t *nk = nullptr; uint32_t *nv = nullptr;
  const size_t kb = (total ? total : 1) * NW * sizeof(uint64_t), vb = (total ? total : 1) * sizeof(uint32_t);
  hipError_t e1 = pool_alloc(ctx, (void **)&nk, kb);
  hipError_t e2 = pool_alloc(ctx, (void **)&nv, vb);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    if (nk) pool_free(ctx, nk, kb);
    if (nv) pool_free(ctx, nv, vb);
    return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the index arrays");
  }
  {
    ProfScope ps(ctx, "bucket_compact", total);
    hipLaunchKernelGGL((bucket_compact_kernel<NW, uint32_t>), dim3(kNumFine), dim3(256), 0, ctx->stream, (const uint64_t *)idx->keys,
                       (const uint32_t *)idx->vals, (const uint64_t *)idx->bucket_off, (const uint64_t *)nullptr, (const uint64_t *)idx->dense_off, nk, nv);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pool_free(ctx, idx->keys, idx->keys_bytes);
  pool_free(ctx, idx->vals, idx->vals_bytes);
  pool_free(ctx, idx->bucket_off, kOffBytes);
  pool_free(ctx, idx->bucket_cnt, sizeof(uint32_t) * kNumFine);
  idx->keys = nk; idx->vals = nv; idx->bucket_off = idx->dense_off; idx->dense_off = nullptr; idx->bucket_cnt = nullptr;
  idx->keys_bytes = kb; idx->vals_bytes = vb;
  return KMI_OK;
}
static kmi_status ensure_dense(kmi_index *idx) {
  if (!idx->bucket_cnt) return KMI_OK;
  return densify_impl<1, 2>(idx);   // (only one-word 2-bit builds leave the sparse form)
}
static kmi_status ensure_layout(kmi_index *idx, uint32_t target_w) {
  KMI_TRY(ensure_dense(idx));
  if (idx->layout_w == target_w) return KMI_OK;
  if (!idx->has_data || idx->n_entries == 0 || idx->val_words) { idx->layout_w = idx->val_words ? 0u : target_w; return KMI_OK; }
  return relayout_dispatch(idx, target_w);
}

static kmi_status alloc_mm_arrays(kmi_ctx *ctx, uint64_t total, int nw, int vw, uint64_t **nk, uint64_t **nv, uint64_t **noff, size_t *kb,
                                  size_t *vb) {
  *nk = *nv = *noff = nullptr;
  *kb = (total ? total : 1) * nw * sizeof(uint64_t);
  *vb = (total ? total : 1) * vw * sizeof(uint64_t);
  hipError_t e0 = pool_alloc(ctx, (void **)noff, kOffBytes);
  hipError_t e1 = pool_alloc(ctx, (void **)nk, *kb);
  hipError_t e2 = pool_alloc(ctx, (void **)nv, *vb);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) {
    if (*nk) pool_free(ctx, *nk, *kb);
    if (*nv) pool_free(ctx, *nv, *vb);
    if (*noff) pool_free(ctx, *noff, kOffBytes);
    return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the index arrays");
  }
  return KMI_OK;
}

}  // namespace kmi
#include "kmi_tuples.h"
namespace kmi {

// The partition of a position / position + quality build straight from the parse (kmi_tuples.h): tuple_hist -> offsets ->
// [quality values] -> tuple_scatter (records into the coarse buckets) -> scatter_fine. Same result as extract + partition_impl
// (every (k-mer, id[, quality]) tuple in the fine bucket of its placement hash); the tuple arrays in file order never exist.
// FASTQ: sc; FASTA: fa (with ids_by_rank). *n_out = tuples. split_keys / split_vals: as partition_impl (sized by the caller
// through size_cb once the tuple count is known).
template <int NW, int BITS, int VW, typename AllocFn>
static kmi_status partition_from_parse(kmi_index *idx, const uint8_t *bytes_dev, const FastqScan *sc, const FastaScan *fa, uint64_t file_offset,
                                       Partitioned *out, uint64_t *n_out, AllocFn alloc_split) {
  kmi_ctx *ctx = idx->ctx;
  using Cfg = ExCfg<NW, BITS>;
  const bool fasta = fa != nullptr;
  const bool canonical = idx->cfg.strand != KMI_STRAND_SINGLE;
  const bool quality = VW == 2 && !fasta;
  PackedInput in;
  uint64_t n_tiles;
  if (fasta) {
    in.eol = fa->pk_break; in.stream = fa->pk_stream; in.n_bytes = fa->n_chars; in.n_cover = fa->n_cover; in.n_valid = fa->n_valid; in.brk = nullptr;
    n_tiles = (fa->n_chars + Cfg::TILE - 1) / Cfg::TILE;
  } else {
    in.eol = sc->pk_eol; in.stream = sc->pk_stream; in.n_bytes = sc->n_bytes; in.n_cover = sc->n_cover; in.n_valid = sc->n_bytes; in.brk = sc->pk_brk;
    n_tiles = sc->n_tiles;
  }
  uint64_t per_tiles = (n_tiles + kPartGroups - 1) / kPartGroups;   // scatter-pass tiles per workgroup (both passes walk the same ranges)
  PartWs w;
  uint64_t n = fasta ? 0 : sc->n_tuples;
  KMI_TRY(get_part_ws(ctx, (size_t)n, NW + VW, WS_KEYS_A, WS_KEYS_B, &w));
  KMI_HIP(ctx, hipMemsetAsync(w.fine_hist, 0, sizeof(uint32_t) * kNumFine * kFineParts, ctx->stream));
  ReadDesc *reads = nullptr;
  if (quality) KMI_TRY(fastq_quality_reads(ctx, *sc, &reads));
  if (!fasta) {
    // FASTQ: the histogram pass of the fused count build -- the entry list (runs of up to eight windows of a
    // read, fastq_list_kernel, which also carries the seq / qual length rule) and fastq_hist_list_kernel over it: no per-tile
    // scans, rolled windows, the same tile ownership as tuple_scatter (2 ms where tuple_hist takes 7)
    using LPC = ListPassCfg<NW, BITS>;
    const bool split = sc->pk_brk != nullptr;
    void *pl;
    KMI_TRY(ws_get(ctx, WS_ENT_LIST, sizeof(uint16_t) * ((size_t)n_tiles * LPC::ent_stride(idx->shape.k, split) + 64), &pl)); uint16_t *ent = (uint16_t *)pl;
    KMI_TRY(ws_get(ctx, WS_ENT_CNT, sizeof(uint32_t) * (n_tiles + 8), &pl)); uint32_t *ent_cnt = (uint32_t *)pl;
    {
      // (position + quality: the list pass also leaves the read descriptors the quality kernel starts from)
      ProfScope ps(ctx, "fastq_list", n);
      const uint32_t wave_lds = LPC::wave_lds_bytes(idx->shape.k, split, quality);
      hipLaunchKernelGGL((fastq_list_kernel<NW, BITS>), dim3(kListGroups), dim3(kListThreads), wave_lds * (kListThreads / kWave), ctx->stream, in,
                         n_tiles, idx->shape.k, LPC::max_runs(idx->shape.k, split), wave_lds, sc->line_base, ctx->d_flags, ent, ent_cnt,
                         LPC::ent_stride(idx->shape.k, split), 128u, reads, sc->tile_off);
    }
    {
      ProfScope ps(ctx, "fastq_hist", n);
      hipLaunchKernelGGL((fastq_hist_list_kernel<NW, BITS>), dim3(kPartGroups), dim3(kHistThreads), 0, ctx->stream, in, n_tiles, idx->shape,
                         canonical, (const uint16_t *)ent, (const uint32_t *)ent_cnt, LPC::ent_stride(idx->shape.k, split), w.fine_hist, w.wg_hist);
    }
  } else {
    // FASTA: tuple_hist over the compacted stream -- in tiles of 512 threads whatever the scatter pass's tile is (three-word shapes:
    // twice its size), both passes over the same byte ranges per workgroup
    ProfScope ps(ctx, "tuple_hist", n);
    constexpr int NTH = 512, RATIO = NTH / Cfg::NT;
    static_assert(NTH % Cfg::NT == 0, "the histogram pass's tile is a whole number of the scatter pass's");
    per_tiles = (per_tiles + RATIO - 1) / RATIO * RATIO;
    hipLaunchKernelGGL((tuple_hist_kernel<NW, BITS, true, false, NTH>), dim3(kPartGroups), dim3(NTH), 0, ctx->stream, in, (n_tiles + RATIO - 1) / RATIO, idx->shape,
                       canonical, (const uint32_t *)nullptr, (const uint64_t *)nullptr, reads, w.fine_hist, w.wg_hist, per_tiles / RATIO);
  }
  {
    ProfScope ps(ctx, "fine_offsets", kNumFine);
    launch_fine_offsets(ctx, w.fine_hist, w.fine_off, w.part_off, w.coarse_base);
    hipLaunchKernelGGL(coarse_cursors_kernel, dim3(kNumCoarse / 4), dim3(256), 0, ctx->stream, (const uint32_t *)w.wg_hist, (uint32_t)kPartGroups,
                       (const uint64_t *)w.coarse_base, w.wg_off);
  }
  KMI_HIP(ctx, hipGetLastError());
  if (fasta) {   // the tuple count of a FASTA input is the histogram's total
    KMI_HIP(ctx, hipMemcpyAsync(&n, w.fine_off + kNumFine, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PartWs w2;
    KMI_TRY(get_part_ws(ctx, (size_t)n, NW + VW, WS_KEYS_A, WS_KEYS_B, &w2));   // (the record buffers at their real size; the tables stay where they are)
    w.buf_a = w2.buf_a; w.buf_b = w2.buf_b;
  }
  *n_out = n;
  if (n == 0) return KMI_OK;
  float *dq = nullptr;
  if (quality) {
    uint32_t f0 = 0;   // (a window crowded with tiny lines made no descriptors: the caller takes the extract pass)
    KMI_HIP(ctx, hipMemcpyAsync(&f0, ctx->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (f0 & 8u) {
      KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t), ctx->stream));
      *n_out = ~0ull;
      return KMI_OK;
    }
    void *p;
    KMI_TRY(ws_get(ctx, WS_INPUT2, (size_t)n * sizeof(float) + 64, &p)); dq = (float *)p;
    KMI_TRY(fastq_quality_launch(ctx, bytes_dev, *sc, idx->shape.k, reads, dq));
  }
  uint64_t *split_keys = nullptr, *split_vals = nullptr;
  KMI_TRY(alloc_split(n, &split_keys, &split_vals));
  {
    ProfScope ps(ctx, "tuple_scatter", n);
#define KMI_TUPLE_SCATTER(FA)                                                                                                              \
    hipLaunchKernelGGL((tuple_scatter_kernel<NW, BITS, VW, FA>), dim3(kPartGroups), dim3(Cfg::NT), 0, ctx->stream, in, n_tiles, idx->shape, canonical, \
                       fasta ? (const uint32_t *)nullptr : sc->line_base, fasta ? (const uint64_t *)nullptr : sc->hdr_base,                \
                       fasta ? (const uint64_t *)nullptr : sc->tile_off, file_offset, fasta ? fa->ids_by_rank : (const uint64_t *)nullptr, \
                       (const float *)dq, (const uint64_t *)w.wg_off, w.buf_a, ctx->d_flags, per_tiles)
    if (fasta) KMI_TUPLE_SCATTER(true); else KMI_TUPLE_SCATTER(false);
#undef KMI_TUPLE_SCATTER
  }
  {
    ProfScope ps(ctx, "scatter_fine", n);
    if constexpr (NW + VW <= 4) {
      if (ctx->lines_p2)
        hipLaunchKernelGGL((scatter_fine_records_lines_kernel<NW, BITS, VW>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream,
                           (const uint64_t *)w.buf_a, split_keys ? split_keys : w.buf_b, idx->shape, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off,
                           (const uint64_t *)w.wg_off, (uint32_t)kPartGroups, 0u, split_keys ? split_vals : (uint64_t *)nullptr);
      else
        hipLaunchKernelGGL((scatter_fine_kernel<NW, BITS, VW>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, w.buf_a,
                           split_keys ? split_keys : w.buf_b, idx->shape, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off,
                           (const uint64_t *)w.wg_off, (uint32_t)kPartGroups, (int)BUCKET_SUB, 0u, split_keys ? split_vals : (uint64_t *)nullptr);
    } else
      hipLaunchKernelGGL((scatter_fine_kernel<NW, BITS, VW>), dim3(kNumCoarse * kFineParts), dim3(kPartThreads), 0, ctx->stream, w.buf_a,
                         split_keys ? split_keys : w.buf_b, idx->shape, (const uint64_t *)w.fine_off, (const uint64_t *)w.part_off,
                         (const uint64_t *)w.wg_off, (uint32_t)kPartGroups, (int)BUCKET_SUB, 0u, split_keys ? split_vals : (uint64_t *)nullptr);
  }
  KMI_HIP(ctx, hipGetLastError());
  uint32_t fl = 0, fl0 = 0;
  KMI_HIP(ctx, hipMemcpyAsync(&fl, ctx->d_flags + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(&fl0, ctx->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (!fasta && (fl0 & 4u)) return fastq_length_verdict(ctx);   // (the list pass found a record whose sequence and quality lines differ in length)
  if (fl) {
    KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags + 3, 0, sizeof(uint32_t), ctx->stream));
    return set_err(ctx, KMI_ERR_OVERFLOW, "ShortSequenceKmerId increment overflow (k-mer more than 65535 bytes into its record)");
  }
  out->keys = split_keys ? split_keys : w.buf_b; out->fine_off = w.fine_off; out->scratch = nullptr;
  return KMI_OK;
}

// Index::build_* of a position / position + quality index from the parse: as mm_insert_vw, with partition_from_parse
template <int NW, int BITS, int VW>
static kmi_status mm_build_vw(kmi_index *idx, const uint8_t *bytes_dev, const FastqScan *sc, const FastaScan *fa, uint64_t file_offset, bool *declined) {
  kmi_ctx *ctx = idx->ctx;
  Partitioned part;
  uint64_t n = 0;
  const bool fresh = !idx->has_data || idx->n_entries == 0;
  uint64_t *nk = nullptr, *nv = nullptr, *noff = nullptr;
  size_t kb = 0, vb = 0;
  auto alloc_split = [&](uint64_t nt, uint64_t **sk, uint64_t **sv) -> kmi_status {
    if (!fresh) return KMI_OK;   // (the records go to the workspace and are concatenated with the entries afterwards)
    // an empty index: the fine partition of the records IS the index, so its last pass writes the key and value arrays directly
    KMI_TRY(alloc_mm_arrays(ctx, nt, NW, VW, &nk, &nv, &noff, &kb, &vb));
    *sk = nk; *sv = nv;
    return KMI_OK;
  };
  kmi_status st = partition_from_parse<NW, BITS, VW>(idx, bytes_dev, sc, fa, file_offset, &part, &n, alloc_split);
  if (st == KMI_OK && n == ~0ull) { *declined = true; return KMI_OK; }
  if (st == KMI_OK && n == 0) return KMI_OK;
  if (fresh) {
    if (st == KMI_OK && hipMemcpyAsync(noff, part.fine_off, kOffBytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) st = KMI_ERR_DEVICE;
    if (st == KMI_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = KMI_ERR_DEVICE;
    if (st != KMI_OK) { if (nk) { pool_free(ctx, nk, kb); pool_free(ctx, nv, vb); pool_free(ctx, noff, kOffBytes); } return st; }
    free_index_arrays(idx);
    idx->keys = nk; idx->mvals = nv; idx->bucket_off = noff; idx->n_entries = n; idx->has_data = true;
    idx->keys_bytes = kb; idx->mvals_bytes = vb;
    return KMI_OK;
  }
  KMI_TRY(st);
  const uint64_t total = n + idx->n_entries;
  KMI_TRY(alloc_mm_arrays(ctx, total, NW, VW, &nk, &nv, &noff, &kb, &vb));
  {
    ProfScope ps(ctx, "bucket_concat", n);
    hipLaunchKernelGGL((bucket_concat_kernel<NW, VW>), dim3(kNumFine), dim3(256), 0, ctx->stream, (const uint64_t *)part.keys,
                       (const uint64_t *)part.fine_off, (const uint64_t *)idx->keys, (const uint64_t *)idx->mvals,
                       (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), noff, nk, nv);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_index_arrays(idx);
  idx->keys = nk; idx->mvals = nv; idx->bucket_off = noff; idx->n_entries = total; idx->has_data = true;
  idx->keys_bytes = kb; idx->mvals_bytes = vb;
  return KMI_OK;
}
template <int NW, int BITS>
static kmi_status mm_build_impl(kmi_index *idx, const uint8_t *bytes_dev, const FastqScan *sc, const FastaScan *fa, uint64_t file_offset, bool *declined) {
  if (idx->val_words == 1) return mm_build_vw<NW, BITS, 1>(idx, bytes_dev, sc, fa, file_offset, declined);
  if (idx->val_words == 2) return mm_build_vw<NW, BITS, 2>(idx, bytes_dev, sc, fa, file_offset, declined);
  return set_err(idx->ctx, KMI_ERR_INVALID, "not a multimap index");
}
static kmi_status index_build_records_from_parse(kmi_index *idx, const uint8_t *bytes_dev, const FastqScan *sc, const FastaScan *fa, uint64_t file_offset, bool *declined) {
  *declined = false;
  KMI_DISPATCH(idx->shape, mm_build_impl, idx, bytes_dev, sc, fa, file_offset, declined);
}

template <int NW, int BITS, int VW>
static kmi_status mm_insert_vw(kmi_index *idx, const uint64_t *recs_dev, size_t n, bool transform, const float *in_q = nullptr,
                               const uint64_t *in_v = nullptr) {
  kmi_ctx *ctx = idx->ctx;
  if (n == 0) return KMI_OK;
  Partitioned part;
  if (!idx->has_data || idx->n_entries == 0) {
    // an empty index: the fine partition of the records IS the index, so its last pass writes the key and value arrays directly
    uint64_t *nk, *nv, *noff;
    size_t kb, vb;
    KMI_TRY(alloc_mm_arrays(ctx, n, NW, VW, &nk, &nv, &noff, &kb, &vb));
    kmi_status st = partition_impl<NW, BITS, VW>(ctx, &idx->cfg, idx->shape, recs_dev, n, transform, WS_KEYS_A, WS_KEYS_B, &part, 0u, nk, nv, in_q, in_v);
    if (st == KMI_OK && hipMemcpyAsync(noff, part.fine_off, kOffBytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) st = KMI_ERR_DEVICE;
    if (st == KMI_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = KMI_ERR_DEVICE;
    if (st != KMI_OK) { pool_free(ctx, nk, kb); pool_free(ctx, nv, vb); pool_free(ctx, noff, kOffBytes); return st; }
    free_index_arrays(idx);
    idx->keys = nk; idx->mvals = nv; idx->bucket_off = noff; idx->n_entries = n; idx->has_data = true;
    idx->keys_bytes = kb; idx->mvals_bytes = vb;
    return KMI_OK;
  }
  KMI_TRY((partition_impl<NW, BITS, VW>(ctx, &idx->cfg, idx->shape, recs_dev, n, transform, WS_KEYS_A, WS_KEYS_B, &part, 0u, nullptr, nullptr, in_q, in_v)));
  const uint64_t total = n + idx->n_entries;
  uint64_t *nk, *nv, *noff;
  size_t kb, vb;
  KMI_TRY(alloc_mm_arrays(ctx, total, NW, VW, &nk, &nv, &noff, &kb, &vb));
  {
    ProfScope ps(ctx, "bucket_concat", n);
    hipLaunchKernelGGL((bucket_concat_kernel<NW, VW>), dim3(kNumFine), dim3(256), 0, ctx->stream, (const uint64_t *)part.keys,
                       (const uint64_t *)part.fine_off, (const uint64_t *)idx->keys, (const uint64_t *)idx->mvals,
                       (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), noff, nk, nv);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_index_arrays(idx);
  idx->keys = nk; idx->mvals = nv; idx->bucket_off = noff; idx->n_entries = total; idx->has_data = true;
  idx->keys_bytes = kb; idx->mvals_bytes = vb;
  return KMI_OK;
}

template <int NW, int BITS>
static kmi_status mm_insert_impl(kmi_index *idx, const uint64_t *recs_dev, size_t n, bool transform, const float *in_q, const uint64_t *in_v) {
  if (idx->val_words == 1) return in_q ? set_err(idx->ctx, KMI_ERR_INVALID, "quality values for a position index") : mm_insert_vw<NW, BITS, 1>(idx, recs_dev, n, transform, nullptr, in_v);
  if (idx->val_words == 2) return mm_insert_vw<NW, BITS, 2>(idx, recs_dev, n, transform, in_q, in_v);
  return set_err(idx->ctx, KMI_ERR_INVALID, "not a multimap index");
}

static kmi_status index_insert_records(kmi_index *idx, const uint64_t *recs_dev, size_t n, bool transform, const float *in_q = nullptr,
                                       const uint64_t *in_v = nullptr) {
  // in_q: the records are (key words, id) and the quality word of record i is the float in_q[i] (position + quality index)
  // in_v: recs_dev holds the key words alone, in_v the ids (and in_q the qualities, or none: zero)
  KMI_DISPATCH(idx->shape, mm_insert_impl, idx, recs_dev, n, transform, in_q, in_v);
}

// Index::insert(std::vector<std::pair<Kmer, T>>&) of the counting maps: the value of every pair is ADDED
// (distributed_unordered_map.hpp:1603-1618). distinct_in = true: the keys are known to be distinct (a reduced map handed
// over), so into an empty index they need no table at all.
template <int NW, int BITS>
static kmi_status insert_pairs_impl(kmi_index *idx, const uint64_t *recs_dev, size_t n, bool transform, bool distinct_in, Partitioned *part_out) {
  // part_out: where the fine-partitioned records are left (workspace, valid until the next partition)
  kmi_ctx *ctx = idx->ctx;
  if (n == 0) return KMI_OK;
  KMI_TRY(ensure_layout(idx, 0u));
  Partitioned part;
  KMI_TRY((partition_impl<NW, BITS, 1>(ctx, &idx->cfg, idx->shape, recs_dev, n, transform, WS_KEYS_A, WS_KEYS_B, &part)));
  if (part_out) *part_out = part;
  if (distinct_in && !idx->has_data) {
    uint64_t *nk = nullptr, *noff = nullptr; uint32_t *nv = nullptr;
    const size_t kb = n * NW * sizeof(uint64_t), vb = n * sizeof(uint32_t);
    hipError_t e0 = pool_alloc(ctx, (void **)&noff, kOffBytes), e1 = pool_alloc(ctx, (void **)&nk, kb), e2 = pool_alloc(ctx, (void **)&nv, vb);
    if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) {
      if (nk) pool_free(ctx, nk, kb);
      if (nv) pool_free(ctx, nv, vb);
      if (noff) pool_free(ctx, noff, kOffBytes);
      return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the index arrays");
    }
    {
      ProfScope ps(ctx, "unzip_pairs", n);
      hipLaunchKernelGGL((unzip_pairs_kernel<NW>), dim3(2048), dim3(256), 0, ctx->stream, (const uint64_t *)part.keys, (uint64_t)n, nk, nv);
    }
    KMI_HIP(ctx, hipMemcpyAsync(noff, part.fine_off, kOffBytes, hipMemcpyDeviceToDevice, ctx->stream));
    KMI_HIP(ctx, hipGetLastError());
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_index_arrays(idx);
    idx->keys = nk; idx->vals = nv; idx->bucket_off = noff; idx->n_entries = n; idx->has_data = true;
    idx->keys_bytes = kb; idx->vals_bytes = vb;
    return KMI_OK;
  }
  void *p;
  const uint64_t cap = n + idx->n_entries;
  KMI_TRY(ws_get(ctx, WS_TMP_KEYS, cap * NW * sizeof(uint64_t), &p)); uint64_t *tmp_keys = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_TMP_VALS, cap * sizeof(uint32_t), &p)); uint32_t *tmp_vals = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_BUCKET_CNT, sizeof(uint32_t) * kNumFine, &p)); uint32_t *out_cnt = (uint32_t *)p;
  {
    ProfScope ps(ctx, "bucket_reduce_pairs", n);
    hipLaunchKernelGGL((bucket_reduce_pairs_kernel<NW>), dim3(kNumFine), dim3(TabCfg<NW>::NT), 0, ctx->stream, (const uint64_t *)part.keys,
                       (const uint64_t *)part.fine_off, (const uint64_t *)idx->keys, (const uint32_t *)idx->vals,
                       (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), tmp_keys, tmp_vals, out_cnt, ctx->d_flags, idx->saturating);
  }
  KMI_HIP(ctx, hipGetLastError());
  return adopt_tmp<NW>(idx, tmp_keys, tmp_vals, part.fine_off, idx->has_data ? idx->bucket_off : nullptr, out_cnt);
}

static kmi_status index_insert_pairs(kmi_index *idx, const uint64_t *recs_dev, size_t n, bool transform, bool distinct_in, Partitioned *part_out) {
  KMI_DISPATCH(idx->shape, insert_pairs_impl, idx, recs_dev, n, transform, distinct_in, part_out);
}

// queries: results compacted into out_keys_dev / out_vals_dev (max(1, val_words) u64 per result)
template <int NW, int BITS, int VW>
static kmi_status query_vw(kmi_index *idx, int mode, const uint64_t *q_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_vals_dev,
                           uint64_t out_capacity, uint64_t *n_out, uint64_t **auto_keys = nullptr, uint64_t **auto_vals = nullptr, bool hits_only = false) {
  kmi_ctx *ctx = idx->ctx;
  constexpr int OW = VW ? VW : 1;
  if (n_out) *n_out = 0;
  if (nq == 0) return KMI_OK;
  if (mode == Q_ERASE && !idx->has_data) return KMI_OK;
  if (mode == Q_ERASE || VW > 0 || idx->find_emits_index) KMI_TRY(ensure_dense(idx));
  Partitioned part;
  KMI_TRY((partition_impl<NW, BITS>(ctx, &idx->cfg, idx->shape, q_dev, nq, true, WS_QUERY_A, WS_QUERY_B, &part, idx->has_data ? idx->layout_w : 0u)));
  void *p;
  if (VW > 0 && mode == Q_FIND) {
    // a multimap find returns every entry of a queried key: sized by a pass that only counts the hits per bucket, then written
    // compactly (a slot per index entry, as erase uses, would be the size of the index -- 72 GB on one rank of config 5)
    KMI_TRY(ws_get(ctx, WS_BUCKET_CNT, sizeof(uint32_t) * kNumFine, &p)); uint32_t *out_cnt = (uint32_t *)p;
    KMI_TRY(ws_get(ctx, WS_BUCKET_OFF, sizeof(uint64_t) * (kNumFine + 1), &p)); uint64_t *res_off = (uint64_t *)p;
    {
      ProfScope ps(ctx, "bucket_query_hits", nq);
      hipLaunchKernelGGL((bucket_query_kernel<NW, VW>), dim3(kNumFine), dim3(QTabCfg<NW>::NT), 0, ctx->stream, (int)Q_HITS, (const uint64_t *)part.keys,
                         (const uint64_t *)part.fine_off, (const uint64_t *)idx->keys, (const uint32_t *)idx->vals, (const uint64_t *)idx->mvals,
                         (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                         out_cnt, ctx->d_flags, false, (const uint32_t *)nullptr, (const uint64_t *)nullptr);
      hipLaunchKernelGGL(bucket_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)out_cnt, res_off, ctx->d_totals, 5);
    }
    uint64_t total = 0;
    KMI_TRY(read_total(ctx, 5, &total));
    if (n_out) *n_out = total;
    if (hits_only) return KMI_OK;   // (the caller sizes its buffers with this and asks again)
    if (auto_keys) {   // the caller takes the results where this function puts them
      KMI_TRY(ws_get(ctx, WS_OUTPUT, (total ? total : 1) * NW * sizeof(uint64_t), &p)); out_keys_dev = (uint64_t *)p;
      KMI_TRY(ws_get(ctx, WS_OUTPUT2, (total ? total : 1) * OW * sizeof(uint64_t), &p)); out_vals_dev = (uint64_t *)p;
      *auto_keys = out_keys_dev; *auto_vals = out_vals_dev;
    } else if (total > out_capacity) return set_err(ctx, KMI_ERR_OVERFLOW, "query: result capacity too small");
    if (total) {
      ProfScope ps(ctx, "bucket_query_find", nq);
      hipLaunchKernelGGL((bucket_query_kernel<NW, VW>), dim3(kNumFine), dim3(QTabCfg<NW>::NT), 0, ctx->stream, (int)Q_FIND, (const uint64_t *)part.keys,
                         (const uint64_t *)part.fine_off, (const uint64_t *)idx->keys, (const uint32_t *)idx->vals, (const uint64_t *)idx->mvals,
                         (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), out_keys_dev, out_vals_dev, (uint32_t *)out_vals_dev,
                         out_cnt, ctx->d_flags, false, (const uint32_t *)nullptr, (const uint64_t *)res_off);
    }
    KMI_HIP(ctx, hipGetLastError());
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMI_OK;
  }
  const bool by_entries = (mode == Q_ERASE);
  const uint64_t cap = by_entries ? std::max<uint64_t>(idx->n_entries, 1) : nq;
  KMI_TRY(ws_get(ctx, WS_TMP_KEYS, cap * NW * sizeof(uint64_t), &p)); uint64_t *tmp_keys = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_TMP_VALS, cap * OW * sizeof(uint64_t), &p)); void *tmp_vals = p;
  KMI_TRY(ws_get(ctx, WS_BUCKET_CNT, sizeof(uint32_t) * kNumFine, &p)); uint32_t *out_cnt = (uint32_t *)p;
  {
    ProfScope ps(ctx, mode == Q_COUNT ? "bucket_query_count" : (mode == Q_FIND ? "bucket_query_find" : "bucket_query_erase"), nq);
    hipLaunchKernelGGL((bucket_query_kernel<NW, VW>), dim3(kNumFine), dim3(QTabCfg<NW>::NT), 0, ctx->stream, mode, (const uint64_t *)part.keys,
                       (const uint64_t *)part.fine_off, (const uint64_t *)idx->keys, (const uint32_t *)idx->vals, (const uint64_t *)idx->mvals,
                       (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), tmp_keys, (uint64_t *)tmp_vals, (uint32_t *)tmp_vals,
                       out_cnt, ctx->d_flags, idx->find_emits_index && mode == Q_FIND && VW == 0, (const uint32_t *)idx->bucket_cnt);
  }
  KMI_HIP(ctx, hipGetLastError());
  const uint64_t *src_off = (by_entries && idx->has_data) ? idx->bucket_off : part.fine_off;
  if (mode == Q_ERASE) {
    const uint64_t before = idx->n_entries;
    if (VW == 0) {
      KMI_TRY((adopt_tmp<NW>(idx, tmp_keys, (const uint32_t *)tmp_vals, idx->bucket_off, nullptr, out_cnt)));
    } else {
      KMI_TRY(ws_get(ctx, WS_BUCKET_OFF, sizeof(uint64_t) * (kNumFine + 1), &p)); uint64_t *res_off = (uint64_t *)p;
      hipLaunchKernelGGL(bucket_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)out_cnt, res_off, ctx->d_totals, 4);
      uint64_t total = 0;
      KMI_TRY(read_total(ctx, 4, &total));
      uint64_t *nk, *nv, *noff;
      size_t kb, vb;
      KMI_TRY(alloc_mm_arrays(ctx, total, NW, OW, &nk, &nv, &noff, &kb, &vb));
      KMI_HIP(ctx, hipMemcpyAsync(noff, res_off, sizeof(uint64_t) * (kNumFine + 1), hipMemcpyDeviceToDevice, ctx->stream));
      hipLaunchKernelGGL((bucket_compact_words_kernel<NW, OW>), dim3(kNumFine), dim3(256), 0, ctx->stream, (const uint64_t *)tmp_keys,
                         (const uint64_t *)tmp_vals, (const uint64_t *)idx->bucket_off, (const uint64_t *)res_off, nk, nv);
      KMI_HIP(ctx, hipGetLastError());
      KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
      free_index_arrays(idx);
      idx->keys = nk; idx->mvals = nv; idx->bucket_off = noff; idx->n_entries = total;
      idx->keys_bytes = kb; idx->mvals_bytes = vb;
    }
    if (n_out) *n_out = before - idx->n_entries;
    return KMI_OK;
  }
  KMI_TRY(ws_get(ctx, WS_BUCKET_OFF, sizeof(uint64_t) * (kNumFine + 1), &p)); uint64_t *res_off = (uint64_t *)p;
  {
    ProfScope ps(ctx, "bucket_offsets", kNumFine);
    hipLaunchKernelGGL(bucket_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)out_cnt, res_off, ctx->d_totals, 5);
  }
  uint64_t total = 0;
  KMI_TRY(read_total(ctx, 5, &total));
  if (n_out) *n_out = total;
  if (total > out_capacity) return set_err(ctx, KMI_ERR_OVERFLOW, "query: result capacity too small");
  {
    ProfScope ps(ctx, "bucket_compact", total);
    hipLaunchKernelGGL((bucket_compact_words_kernel<NW, OW>), dim3(kNumFine), dim3(256), 0, ctx->stream, (const uint64_t *)tmp_keys,
                       (const uint64_t *)tmp_vals, src_off, (const uint64_t *)res_off, out_keys_dev, out_vals_dev);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

template <int NW, int BITS>
static kmi_status query_impl(kmi_index *idx, int mode, const uint64_t *q_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_vals_dev,
                             uint64_t out_capacity, uint64_t *n_out, uint64_t **auto_keys, uint64_t **auto_vals, bool hits_only) {
  if (idx->val_words == 0) return query_vw<NW, BITS, 0>(idx, mode, q_dev, nq, out_keys_dev, out_vals_dev, out_capacity, n_out);
  if (idx->val_words == 1) return query_vw<NW, BITS, 1>(idx, mode, q_dev, nq, out_keys_dev, out_vals_dev, out_capacity, n_out, auto_keys, auto_vals, hits_only);
  return query_vw<NW, BITS, 2>(idx, mode, q_dev, nq, out_keys_dev, out_vals_dev, out_capacity, n_out, auto_keys, auto_vals, hits_only);
}

static kmi_status index_query(kmi_index *idx, int mode, const uint64_t *q_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_vals_dev,
                              uint64_t out_capacity, uint64_t *n_out, uint64_t **auto_keys = nullptr, uint64_t **auto_vals = nullptr, bool hits_only = false) {
  // auto_keys / auto_vals (find of a multimap): the results are left in workspace buffers sized to the hits, returned here
  // hits_only (find of a multimap): *n_out = the entries the find would return, nothing written
  KMI_DISPATCH(idx->shape, query_impl, idx, mode, q_dev, nq, out_keys_dev, out_vals_dev, out_capacity, n_out, auto_keys, auto_vals, hits_only);
}

// results of a multimap find are bounded by the entries, everything else by the queries
static uint64_t query_result_bound(const kmi_index *idx, int mode, size_t nq) {
  return (idx->val_words > 0 && mode == Q_FIND) ? idx->n_entries : (uint64_t)nq;
}

// imxx::distribute bucketing by destination rank
// per-rank counts = sums over the rank's sub-buckets
static kmi_status read_rank_counts(kmi_ctx *ctx, const uint64_t *cnt_dev, uint32_t nranks, uint32_t sub, uint64_t *send_counts_host) {
  uint64_t tmp[kNumCoarse];
  KMI_HIP(ctx, hipMemcpyAsync(tmp, cnt_dev, sizeof(uint64_t) * nranks * sub, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t r = 0; r < nranks; ++r) {
    uint64_t c = 0;
    for (uint32_t j = 0; j < sub; ++j) c += tmp[r * sub + j];
    send_counts_host[r] = c;
  }
  return KMI_OK;
}

template <int NW, int BITS, int VW>
static kmi_status route_vw(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint64_t *keys_dev, size_t n, uint32_t nranks,
                           uint64_t *out_keys_dev, uint64_t *send_counts_host, const float *in_q = nullptr, const uint64_t *in_v = nullptr,
                           bool by_owner = false) {
  // in_q (VW == 2): the input records are (key words, id), the quality word of record i is in_q[i]; in_v: keys_dev holds the
  // key words alone and in_v the ids (scatter_range)
  // by_owner (one-word 2-bit keys): the rank is the owner of the key's minimizer bucket (an index built through exchanged
  // super-k-mers), not KeyToRank's hash
  void *p;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); uint32_t *wg_hist = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_CURSOR, sizeof(uint64_t) * kPartGroups * kNumCoarse, &p)); uint64_t *wg_off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * kNumCoarse * 2, &p)); uint64_t *cnt = (uint64_t *)p;
  BucketFn fn; fn.mode = BUCKET_RANK; fn.shape = shape; fn.dist_hash = cfg->dist_hash; fn.farm_ndebug = cfg->farm_ndebug != 0; fn.nranks = nranks;
  fn.dist_trans = cfg->dist_trans; fn.rank_magic = rank_magic_of(nranks);
  fn.sub = rank_sub_buckets(nranks);
  if (by_owner) { fn.owner_w = sk_window_of(shape.k); fn.dist_trans = 0; }
  const uint32_t nb = nranks * fn.sub;
  {
    ProfScope ps(ctx, "hist_rank", n);
    hipLaunchKernelGGL((hist_rank_kernel<NW, BITS, VW>), dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, keys_dev, (uint64_t)n, shape,
                       cfg->strand, fn, wg_hist, (VW > 0 && in_v) ? (uint32_t)NW : (uint32_t)(NW + VW) - ((VW > 0 && in_q) ? 1u : 0u));
  }
  {
    ProfScope ps(ctx, "rank_offsets", nb);
    launch_rank_offsets(ctx->stream, (const uint32_t *)wg_hist, (uint32_t)kPartGroups, nb, cnt, cnt + kNumCoarse, wg_off);
  }
  {
    ProfScope ps(ctx, "scatter_rank", n);
    hipLaunchKernelGGL((scatter_chunks_kernel<NW, BITS, VW>), dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, keys_dev, (uint64_t)n,
                       out_keys_dev, shape, cfg->strand, true, fn, (const uint64_t *)wg_off, (VW > 0) ? in_q : (const float *)nullptr,
                       (VW > 0) ? in_v : (const uint64_t *)nullptr);
  }
  KMI_HIP(ctx, hipGetLastError());
  return read_rank_counts(ctx, cnt, nranks, fn.sub, send_counts_host);
}

// (k-mer, count) records of a counting map grouped by destination rank: KeyToRank, or the owner of the minimizer bucket
template <int NW, int BITS>
static kmi_status route_pairs_impl(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint64_t *recs_dev, size_t n, uint32_t nranks, bool by_owner,
                                   uint64_t *out_dev, uint64_t *send_counts_host) {
  return route_vw<NW, BITS, 1>(ctx, cfg, shape, recs_dev, n, nranks, out_dev, send_counts_host, nullptr, nullptr, by_owner);
}
static kmi_status route_pairs(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint64_t *recs_dev, size_t n, uint32_t nranks, bool by_owner,
                              uint64_t *out_dev, uint64_t *send_counts_host) {
  if (n == 0) { for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0; return KMI_OK; }
  KMI_DISPATCH(shape, route_pairs_impl, ctx, cfg, shape, recs_dev, n, nranks, by_owner, out_dev, send_counts_host);
}

// queries (and anything else keyed by k-mer) to the rank that owns the key's minimizer bucket
static kmi_status route_owner(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint64_t *keys_dev, size_t n, uint32_t nranks,
                              uint64_t *out_keys_dev, uint64_t *send_counts_host) {
  void *p;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); uint32_t *wg_hist = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_CURSOR, sizeof(uint64_t) * kPartGroups * kNumCoarse, &p)); uint64_t *wg_off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * kNumCoarse * 2, &p)); uint64_t *cnt = (uint64_t *)p;
  BucketFn fn; fn.mode = BUCKET_RANK; fn.shape = shape; fn.dist_hash = cfg->dist_hash; fn.farm_ndebug = cfg->farm_ndebug != 0; fn.nranks = nranks;
  fn.dist_trans = 0; fn.rank_magic = rank_magic_of(nranks);
  fn.sub = rank_sub_buckets(nranks);
  fn.owner_w = sk_window_of(shape.k);
  const uint32_t nb = nranks * fn.sub;
  {
    ProfScope ps(ctx, "hist_rank", n);
    hipLaunchKernelGGL((hist_rank_kernel<1, 2, 0>), dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, keys_dev, (uint64_t)n, shape, cfg->strand, fn, wg_hist);
  }
  {
    ProfScope ps(ctx, "rank_offsets", nb);
    launch_rank_offsets(ctx->stream, (const uint32_t *)wg_hist, (uint32_t)kPartGroups, nb, cnt, cnt + kNumCoarse, wg_off);
  }
  {
    ProfScope ps(ctx, "scatter_rank", n);
    hipLaunchKernelGGL((scatter_chunks_kernel<1, 2, 0>), dim3(kPartGroups), dim3(kPartThreads), 0, ctx->stream, keys_dev, (uint64_t)n, out_keys_dev, shape,
                       cfg->strand, true, fn, (const uint64_t *)wg_off);
  }
  KMI_HIP(ctx, hipGetLastError());
  return read_rank_counts(ctx, cnt, nranks, fn.sub, send_counts_host);
}

template <int NW, int BITS>
static kmi_status route_impl(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint64_t *keys_dev, size_t n, uint32_t nranks,
                             uint32_t value_words, uint64_t *out_keys_dev, uint64_t *send_counts_host) {
  if (value_words == 0) return route_vw<NW, BITS, 0>(ctx, cfg, shape, keys_dev, n, nranks, out_keys_dev, send_counts_host);
  if (value_words == 1) return route_vw<NW, BITS, 1>(ctx, cfg, shape, keys_dev, n, nranks, out_keys_dev, send_counts_host);
  if (value_words == 2) return route_vw<NW, BITS, 2>(ctx, cfg, shape, keys_dev, n, nranks, out_keys_dev, send_counts_host);
  return set_err(ctx, KMI_ERR_INVALID, "value_words must be 0, 1 or 2");
}

template <int NW, int BITS>
static kmi_status route_records_q(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint64_t *keys, const uint64_t *ids, size_t n, uint32_t nranks,
                                  const float *in_q, uint64_t *out, uint64_t *send_counts_host) {
  if (in_q) return route_vw<NW, BITS, 2>(ctx, cfg, shape, keys, n, nranks, out, send_counts_host, in_q, ids);
  return route_vw<NW, BITS, 1>(ctx, cfg, shape, keys, n, nranks, out, send_counts_host, nullptr, ids);
}

// read_file + the bucketing half of imxx::distribute in one go (FASTQ): keys of this rank's reads, transformed and
// grouped by destination rank, straight from the window list; the tuple array in file order never exists.
template <int NW, int BITS>
static kmi_status extract_route_impl(kmi_ctx *ctx, const kmi_config *cfg, KShape shape, const uint8_t *bytes_dev, size_t n_bytes, uint32_t nranks,
                                     uint64_t *out_keys_dev, size_t capacity, uint64_t *n_tuples, uint64_t *n_seqs, uint64_t *send_counts_host) {
  FastqScan sc;
  KMI_TRY(fastq_scan(ctx, cfg, bytes_dev, n_bytes, &sc, false));
  if (n_tuples) *n_tuples = sc.n_tuples;
  if (n_seqs) *n_seqs = sc.n_seqs;
  for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0;
  const uint64_t n = sc.n_tuples, n_tiles = sc.n_tiles;
  if (n == 0) return KMI_OK;
  if (n > capacity) return set_err(ctx, KMI_ERR_OVERFLOW, "extract_route: output capacity too small");
  PackedInput in; in.eol = sc.pk_eol; in.stream = sc.pk_stream; in.n_bytes = sc.n_bytes; in.n_cover = sc.n_cover; in.n_valid = sc.n_bytes; in.brk = sc.pk_brk;
  const bool split = sc.pk_brk != nullptr;
  const bool canonical = cfg->strand != KMI_STRAND_SINGLE;
  void *p;
  KMI_TRY(ws_get(ctx, WS_WGHIST, sizeof(uint32_t) * kPartGroups * kNumCoarse, &p)); uint32_t *wg_hist = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_CURSOR, sizeof(uint64_t) * kPartGroups * kNumCoarse, &p)); uint64_t *wg_off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * kNumCoarse * 2, &p)); uint64_t *cnt = (uint64_t *)p;
  using LPC = ListPassCfg<NW, BITS>;
  KMI_TRY(ws_get(ctx, WS_ENT_LIST, sizeof(uint16_t) * ((size_t)n_tiles * LPC::ent_stride(shape.k, split) + 64), &p)); uint16_t *ent = (uint16_t *)p;
  KMI_TRY(ws_get(ctx, WS_ENT_CNT, sizeof(uint32_t) * (n_tiles + 8), &p)); uint32_t *ent_cnt = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_ENT_BKT, sizeof(uint64_t) * ((size_t)n_tiles * ListPassCfg<NW, BITS>::ent_stride(shape.k, split) + 64), &p));
  uint64_t *ent_bkt = (uint64_t *)p;
  BucketFn fn; fn.mode = BUCKET_RANK; fn.shape = shape; fn.dist_hash = cfg->dist_hash; fn.farm_ndebug = cfg->farm_ndebug != 0; fn.nranks = nranks;
  fn.dist_trans = cfg->dist_trans; fn.rank_magic = rank_magic_of(nranks);
  fn.sub = rank_sub_buckets(nranks);
  const uint32_t nb = nranks * fn.sub;
  {
    ProfScope ps(ctx, "fastq_list", n);
    using LP = ListPassCfg<NW, BITS>;
    const uint32_t wave_lds = LP::wave_lds_bytes(shape.k, split);
    hipLaunchKernelGGL((fastq_list_kernel<NW, BITS>), dim3(kListGroups), dim3(kListThreads), wave_lds * (kListThreads / kWave), ctx->stream, in,
                       n_tiles, shape.k, LP::max_runs(shape.k, split), wave_lds, sc.line_base, ctx->d_flags, ent, ent_cnt,
                       LP::ent_stride(shape.k, split));
  }
  {
    ProfScope ps(ctx, "fastq_rank_hist", n);
    hipLaunchKernelGGL((fastq_rank_hist_list_kernel<NW, BITS, 512>), dim3(fused_groups<NW>()), dim3(512), 0, ctx->stream, in, n_tiles, shape, canonical,
                       (const uint16_t *)ent, (const uint32_t *)ent_cnt, LPC::ent_stride(shape.k, split), fn, wg_hist, ent_bkt);
  }
  {
    ProfScope ps(ctx, "rank_offsets", nb);
    launch_rank_offsets(ctx->stream, (const uint32_t *)wg_hist, (uint32_t)fused_groups<NW>(), nb, cnt, cnt + kNumCoarse, wg_off);
  }
  {
    ProfScope ps(ctx, "fastq_rank_scatter", n);
    hipLaunchKernelGGL((fastq_scatter_list_kernel<NW, BITS, true>), dim3(fused_groups<NW>()), dim3(ExCfg<NW, BITS>::NT), 0, ctx->stream, in, n_tiles,
                       shape, canonical, (const uint16_t *)ent, (const uint32_t *)ent_cnt, LPC::ent_stride(shape.k, split), (const uint64_t *)ent_bkt,
                       (const uint64_t *)wg_off, out_keys_dev);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_TRY(read_rank_counts(ctx, cnt, nranks, fn.sub, send_counts_host));
  return fastq_length_verdict(ctx);   // the length rule rode on the list pass
}

// split of a count index by destination rank (see "Combine-first distributed insert")
template <int NW, int BITS>
static kmi_status split_impl(kmi_index *idx, uint32_t nranks, uint64_t *out_keys_dev, uint32_t *out_counts_dev, size_t capacity,
                             uint32_t *bucket_cnt_dev, uint64_t *send_counts_host) {
  kmi_ctx *ctx = idx->ctx;
  const uint64_t n = idx->n_entries;
  if (n > capacity) return set_err(ctx, KMI_ERR_OVERFLOW, "output capacity is smaller than the number of index entries");
  KMI_TRY(ensure_layout(idx, preferred_layout(idx)));   // sender and receiver agree on what "bucket b" means
  void *p;
  KMI_TRY(ws_get(ctx, WS_SPLIT_RANK, n + 64, &p)); uint8_t *rank_of = (uint8_t *)p;
  KMI_TRY(ws_get(ctx, WS_SPLIT_OFF, sizeof(uint64_t) * ((size_t)nranks * (kNumFine + 1) + 2 * (kNumCoarse + 1)), &p));
  uint64_t *boff = (uint64_t *)p, *tot = boff + (size_t)nranks * (kNumFine + 1), *base = tot + kNumCoarse + 1;
  BucketFn fn; fn.mode = BUCKET_RANK; fn.shape = idx->shape; fn.dist_hash = idx->cfg.dist_hash; fn.farm_ndebug = idx->cfg.farm_ndebug != 0;
  fn.dist_trans = idx->cfg.dist_trans; fn.rank_magic = rank_magic_of(nranks);
  fn.nranks = nranks; fn.sub = 1;
  {
    ProfScope ps(ctx, "split_count", n);
    hipLaunchKernelGGL((split_count_kernel<NW>), dim3(kNumFine), dim3(256), 0, ctx->stream, (const uint64_t *)idx->keys,
                       (const uint64_t *)idx->bucket_off, fn, rank_of, bucket_cnt_dev);
  }
  {
    ProfScope ps(ctx, "split_offsets", (uint64_t)nranks * kNumFine);
    hipLaunchKernelGGL(part_offsets_kernel, dim3(nranks), dim3(1024), 0, ctx->stream, (const uint32_t *)bucket_cnt_dev, boff, tot);
    hipLaunchKernelGGL(part_bases_kernel, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *)tot, nranks, base);
  }
  {
    ProfScope ps(ctx, "split_scatter", n);
    hipLaunchKernelGGL((split_scatter_kernel<NW>), dim3(kNumFine), dim3(256), 0, ctx->stream, (const uint64_t *)idx->keys, (const uint32_t *)idx->vals,
                       (const uint64_t *)idx->bucket_off, (const uint8_t *)rank_of, nranks, (const uint64_t *)boff, (const uint64_t *)base,
                       out_keys_dev, out_counts_dev);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipMemcpyAsync(send_counts_host, tot, sizeof(uint64_t) * nranks, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

template <int NW, int BITS>
static kmi_status merge_impl(kmi_index *idx, uint32_t nparts, const uint64_t *keys_dev, const uint32_t *counts_dev, const uint32_t *bucket_cnt_dev) {
  kmi_ctx *ctx = idx->ctx;
  KMI_TRY(ensure_layout(idx, preferred_layout(idx)));
  void *p;
  KMI_TRY(ws_get(ctx, WS_SPLIT_OFF, sizeof(uint64_t) * ((size_t)nparts * (kNumFine + 1) + 2 * (kNumCoarse + 1)), &p));
  uint64_t *boff = (uint64_t *)p, *tot = boff + (size_t)nparts * (kNumFine + 1), *base = tot + kNumCoarse + 1;
  KMI_TRY(ws_get(ctx, WS_BUCKET_OFF, sizeof(uint64_t) * (kNumFine + 1), &p)); uint64_t *comb = (uint64_t *)p;
  {
    ProfScope ps(ctx, "merge_offsets", (uint64_t)nparts * kNumFine);
    hipLaunchKernelGGL(part_offsets_kernel, dim3(nparts), dim3(1024), 0, ctx->stream, bucket_cnt_dev, boff, tot);
    hipLaunchKernelGGL(part_bases_kernel, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *)tot, nparts, base);
    hipLaunchKernelGGL(parts_sum_kernel, dim3(kNumFine / 256 + 1), dim3(256), 0, ctx->stream, (const uint64_t *)boff, nparts, comb);
  }
  uint64_t n_in = 0;
  KMI_HIP(ctx, hipMemcpyAsync(&n_in, base + nparts, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_in == 0) return KMI_OK;
  const uint64_t cap = n_in + idx->n_entries;
  KMI_TRY(ws_get(ctx, WS_TMP_KEYS, cap * NW * sizeof(uint64_t), &p)); uint64_t *tmp_keys = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_TMP_VALS, cap * sizeof(uint32_t), &p)); uint32_t *tmp_vals = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_BUCKET_CNT, sizeof(uint32_t) * kNumFine, &p)); uint32_t *out_cnt = (uint32_t *)p;
  {
    ProfScope ps(ctx, "bucket_merge", n_in);
    hipLaunchKernelGGL((bucket_merge_kernel<NW>), dim3(kNumFine), dim3(TabCfg<NW>::NT), 0, ctx->stream, keys_dev, counts_dev, nparts,
                       (const uint64_t *)base, (const uint64_t *)boff, (const uint64_t *)comb, (const uint64_t *)idx->keys,
                       (const uint32_t *)idx->vals, (const uint64_t *)(idx->has_data ? idx->bucket_off : nullptr), tmp_keys, tmp_vals, out_cnt,
                       ctx->d_flags, idx->saturating);
  }
  KMI_HIP(ctx, hipGetLastError());
  return adopt_tmp<NW>(idx, tmp_keys, tmp_vals, comb, idx->has_data ? idx->bucket_off : nullptr, out_cnt);
}

}  // namespace kmi

// dispatch of the two halves on the window width
static kmi_status sk_produce(kmi_index *idx, uint32_t w, const uint8_t *bytes_dev, size_t n_bytes, uint32_t nranks, const uint64_t **recs, uint64_t *n_records,
                             uint64_t *send_counts, int *produced, uint64_t *out, size_t out_cap) {
  return w == 19u ? sk_produce_w<19>(idx, bytes_dev, n_bytes, nranks, recs, n_records, send_counts, produced, out, out_cap)
       : (w == 13u ? sk_produce_w<13>(idx, bytes_dev, n_bytes, nranks, recs, n_records, send_counts, produced, out, out_cap)
       : (w == 11u ? sk_produce_w<11>(idx, bytes_dev, n_bytes, nranks, recs, n_records, send_counts, produced, out, out_cap)
                   : sk_produce_w<7>(idx, bytes_dev, n_bytes, nranks, recs, n_records, send_counts, produced, out, out_cap)));
}
static kmi_status sk_consume(kmi_index *idx, uint32_t w, const uint64_t *recs_dev, uint64_t n_records, uint32_t nranks) {
  if (n_records == 0) { idx->owner_lp = 31u - (uint32_t)__builtin_clz(nranks); return KMI_OK; }
  return w == 19u ? sk_consume_w<19>(idx, recs_dev, n_records, nranks) : (w == 13u ? sk_consume_w<13>(idx, recs_dev, n_records, nranks)
       : (w == 11u ? sk_consume_w<11>(idx, recs_dev, n_records, nranks) : sk_consume_w<7>(idx, recs_dev, n_records, nranks)));
}
static bool sk_rank_count(uint32_t p) { return p == 1u || p == 2u || p == 4u || p == 8u; }

#include "kmi_debruijn.h"
#include "kmi_update.h"

struct kmi_comm;
static kmi_status dist_state(kmi_index *idx, kmi_comm *comm, uint64_t *holders, uint64_t *owner_p, uint64_t *owner_any);
// every rank brings the status of what it has just done on its own; all leave with an error if any did (its own, or KMI_ERR_PEER):
// one all-reduce, placed where the next step is a collective that a rank which has already returned would leave its peers waiting in
static kmi_status dist_agree(kmi_comm *comm, kmi_status mine);
static int comm_rank_of(kmi_comm *comm);

extern "C" {

kmi_status kmi_route_tuples_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *records_dev, size_t n, uint32_t nranks,
                                uint32_t value_words, uint64_t *out_records_dev, uint64_t *send_counts_host) {
  if (!ctx) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (nranks == 0 || nranks > (uint32_t)kNumCoarse || !send_counts_host) return set_err(ctx, KMI_ERR_INVALID, "nranks must be in 1..256");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) { for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0; return KMI_OK; }
  KMI_DISPATCH(shape, route_impl, ctx, cfg, shape, records_dev, n, nranks, value_words, out_records_dev, send_counts_host);
}

kmi_status kmi_route_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *keys_dev, size_t n, uint32_t nranks,
                         uint64_t *out_keys_dev, uint64_t *send_counts_host) {
  if (!ctx) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (nranks == 0 || nranks > (uint32_t)kNumCoarse || !send_counts_host) return set_err(ctx, KMI_ERR_INVALID, "nranks must be in 1..256");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) { for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0; return KMI_OK; }
  KMI_DISPATCH(shape, route_impl, ctx, cfg, shape, keys_dev, n, nranks, 0u, out_keys_dev, send_counts_host);
}

kmi_status kmi_extract_route_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint32_t nranks,
                                 uint64_t *out_keys_dev, size_t out_capacity, uint64_t *n_tuples, uint64_t *n_seqs,
                                 uint64_t *send_counts_host) {
  if (!ctx) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (nranks == 0 || nranks > (uint32_t)kNumCoarse || !send_counts_host) return set_err(ctx, KMI_ERR_INVALID, "nranks must be in 1..256");
  if (cfg->seq_format != KMI_FMT_FASTQ) return set_err(ctx, KMI_ERR_INVALID, "extract_route: FASTQ only (use kmi_extract_dev + kmi_route_dev)");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_tuples) *n_tuples = 0;
  if (n_seqs) *n_seqs = 0;
  if (n_bytes == 0) { for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0; return KMI_OK; }
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  KMI_DISPATCH(shape, extract_route_impl, ctx, cfg, shape, bytes_dev, n_bytes, nranks, out_keys_dev, out_capacity, n_tuples, n_seqs, send_counts_host);
}

// the tuples of the position indexes parsed and grouped by destination rank in one call: the tuples in file order stay in the
// workspace as the parsers' k-mer, id and quality arrays, from which the scatter by rank gathers its records
kmi_status kmi_extract_route_records_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset,
                                         uint32_t nranks, uint64_t *out_records_dev, size_t out_capacity, uint64_t *n_tuples, uint64_t *n_seqs,
                                         uint64_t *send_counts_host) {
  if (!ctx) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (nranks == 0 || nranks > (uint32_t)kNumCoarse || !send_counts_host) return set_err(ctx, KMI_ERR_INVALID, "nranks must be in 1..256");
  if (cfg->index_kind == KMI_INDEX_COUNT) return set_err(ctx, KMI_ERR_INVALID, "records are the tuples of the position indexes (index_kind POSITION / POSQUAL)");
  if (cfg->index_kind == KMI_INDEX_POSQUAL && cfg->seq_format != KMI_FMT_FASTQ) return set_err(ctx, KMI_ERR_INVALID, "quality values need FASTQ input");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_tuples) *n_tuples = 0;
  if (n_seqs) *n_seqs = 0;
  for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0;
  if (n_bytes == 0) return KMI_OK;
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  uint64_t nt = 0, ns = 0;
  KMI_TRY(extract_count(ctx, cfg, bytes_dev, n_bytes, &nt, &ns));
  if (n_tuples) *n_tuples = nt;
  if (n_seqs) *n_seqs = ns;
  if (nt == 0) return KMI_OK;
  if (nt > out_capacity) return set_err(ctx, KMI_ERR_OVERFLOW, "extract_route_records: output capacity too small");
  const uint32_t nw = shape.n_words;
  const bool q = cfg->index_kind == KMI_INDEX_POSQUAL;
  void *dk, *di, *dq = nullptr;
  KMI_TRY(ws_get(ctx, WS_OUTPUT, (size_t)nt * nw * sizeof(uint64_t), &dk));
  KMI_TRY(ws_get(ctx, WS_OUTPUT2, (size_t)nt * sizeof(uint64_t), &di));
  if (q) KMI_TRY(ws_get(ctx, WS_INPUT2, (size_t)nt * sizeof(float) + 64, &dq));
  KMI_TRY(extract_run(ctx, cfg, bytes_dev, n_bytes, file_offset, (uint64_t *)dk, (uint64_t *)di, (size_t)nt, false, true, &nt, &ns, (float *)dq));
  KMI_DISPATCH(shape, route_records_q, ctx, cfg, shape, (const uint64_t *)dk, (const uint64_t *)di, (size_t)nt, nranks, (const float *)dq, out_records_dev,
               send_counts_host);
}

kmi_status kmi_index_create(kmi_ctx *ctx, const kmi_config *cfg, kmi_index **out) {
  if (!ctx || !out) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  kmi_index *idx = new kmi_index();
  idx->ctx = ctx; idx->cfg = *cfg; idx->shape = shape;
  idx->val_words = cfg->index_kind == KMI_INDEX_COUNT ? 0u : (cfg->index_kind == KMI_INDEX_POSITION ? 1u : 2u);
  *out = idx;
  return KMI_OK;
}

kmi_status kmi_index_destroy(kmi_index *idx) {
  if (!idx) return KMI_OK;
  (void)hipSetDevice(idx->ctx->device);
  (void)hipStreamSynchronize(idx->ctx->stream);
  free_index_arrays(idx);
  delete idx;
  return KMI_OK;
}

kmi_status kmi_index_insert_dev(kmi_index *idx, const uint64_t *kmers_dev, size_t n) {
  if (!idx) return KMI_ERR_INVALID;
  if (idx->val_words) return set_err(idx->ctx, KMI_ERR_INVALID, "a position index takes (k-mer, value) tuples: kmi_index_insert_tuples_*");
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  return index_insert(idx, kmers_dev, n, true);
}

kmi_status kmi_index_insert_transformed_dev(kmi_index *idx, const uint64_t *kmers_dev, size_t n) {
  if (!idx) return KMI_ERR_INVALID;
  if (idx->val_words) return set_err(idx->ctx, KMI_ERR_INVALID, "a position index takes (k-mer, value) tuples: kmi_index_insert_tuples_*");
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  return index_insert(idx, kmers_dev, n, false);
}

kmi_status kmi_index_insert_pairs_dev(kmi_index *idx, const uint64_t *records_dev, size_t n) {
  if (!idx) return KMI_ERR_INVALID;
  if (idx->val_words) return set_err(idx->ctx, KMI_ERR_INVALID, "(k-mer, count) pairs go into a count index");
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  return index_insert_pairs(idx, records_dev, n, true, false);
}

kmi_status kmi_index_insert_pairs_host(kmi_index *idx, const uint64_t *records, size_t n) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "(k-mer, count) pairs go into a count index");
  if (n == 0) return KMI_OK;
  if (!records) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *din;
  const size_t bytes = n * (idx->shape.n_words + 1) * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, WS_INPUT, bytes, &din));
  KMI_HIP(ctx, hipMemcpyAsync(din, records, bytes, hipMemcpyHostToDevice, ctx->stream));
  return index_insert_pairs(idx, (const uint64_t *)din, n, true, false);
}

kmi_status kmi_index_insert_host(kmi_index *idx, const uint64_t *kmers, size_t n) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "a position index takes (k-mer, value) tuples: kmi_index_insert_tuples_*");
  if (n == 0) return KMI_OK;
  if (!kmers) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *din;
  const size_t bytes = n * idx->shape.n_words * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, WS_INPUT, bytes, &din));
  KMI_HIP(ctx, hipMemcpyAsync(din, kmers, bytes, hipMemcpyHostToDevice, ctx->stream));
  return index_insert(idx, (const uint64_t *)din, n, true);
}

kmi_status kmi_index_set_seq_format(kmi_index *idx, uint32_t seq_format) {
  if (!idx) return KMI_ERR_INVALID;
  if (seq_format > KMI_FMT_FASTA) return set_err(idx->ctx, KMI_ERR_INVALID, "unknown sequence format");
  idx->cfg.seq_format = seq_format;
  return KMI_OK;
}

kmi_status kmi_index_set_seq_filter(kmi_index *idx, uint32_t seq_filter) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_config c = idx->cfg;
  c.seq_filter = seq_filter;
  if (!kmi::valid_config(&c, nullptr)) return set_err(idx->ctx, KMI_ERR_INVALID, "sequence filter not available for this index (see kmi_config.seq_filter)");
  idx->cfg.seq_filter = seq_filter;
  return KMI_OK;
}

kmi_status kmi_index_build_dev(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_bytes == 0) return KMI_OK;
  // (bytes still in host memory, kmi_index_build_host: only the one-pass front end of the super-k-mer build feeds itself)
  const bool sk_fastq = idx->val_words == 0 && idx->cfg.seq_format == KMI_FMT_FASTQ && idx->shape.n_words == 1 && idx->shape.bits == 2 && ctx->fused_superkmer &&
                        sk_window_of(idx->shape.k) != 0u;
  if (!sk_fastq) KMI_TRY(feed_flush(ctx));
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  if (idx->val_words == 0 && idx->cfg.seq_format == KMI_FMT_FASTQ) return index_build_fused(idx, bytes_dev, n_bytes);
  if (idx->val_words == 0) {
    const uint32_t w = (idx->shape.n_words == 1 && idx->shape.bits == 2 && ctx->fused_superkmer) ? sk_window_of(idx->shape.k) : 0u;
    if (w) {   // FASTA count index of one-word DNA k-mers: super-k-mers cut from the compacted stream (falls through when a capacity is exceeded)
      FastaScan fa;
      KMI_TRY(fasta_scan(ctx, &idx->cfg, bytes_dev, n_bytes, file_offset, false, &fa));
      if (fa.n_chars < idx->shape.k) return KMI_OK;
      bool done = false;
      kmi_status st = w == 19u ? build_superkmer_fasta_w<19>(idx, fa, &done) : (w == 13u ? build_superkmer_fasta_w<13>(idx, fa, &done)
                    : (w == 11u ? build_superkmer_fasta_w<11>(idx, fa, &done) : build_superkmer_fasta_w<7>(idx, fa, &done)));
      if (st != KMI_OK || done) return st;
    }
    // FASTA count index: tuples from the compacted-stream extract, then the key insert path
    uint64_t nt = 0, ns = 0;
    KMI_TRY(extract_count(ctx, &idx->cfg, bytes_dev, n_bytes, &nt, &ns));
    if (nt == 0) return KMI_OK;
    void *dk;
    KMI_TRY(ws_get(ctx, WS_OUTPUT, (size_t)nt * idx->shape.n_words * sizeof(uint64_t), &dk));
    KMI_TRY(extract_run(ctx, &idx->cfg, bytes_dev, n_bytes, file_offset, (uint64_t *)dk, nullptr, (size_t)nt, true, true, &nt, &ns));
    return index_insert(idx, (const uint64_t *)dk, (size_t)nt, false);
  }
  // PositionIndex / PositionQualityIndex: KmerPosition(Quality)TupleParser tuples -> multimap insert. Partitioned straight from
  // the parse (kmi_tuples.h); KMI_TUPLES=extract keeps the round-1 order (tuple arrays in file order, then the key partition)
  if (ctx->tuples_from_parse) {
    bool declined = false;
    if (idx->cfg.seq_format == KMI_FMT_FASTQ) {
      FastqScan sc;
      KMI_TRY(fastq_scan(ctx, &idx->cfg, bytes_dev, n_bytes, &sc, false));   // (the list pass carries the seq / qual length rule)
      if (sc.n_tuples == 0) return fastq_scan(ctx, &idx->cfg, bytes_dev, n_bytes, &sc, true);   // (no k-mer at all: only the record rules are left to check)
      KMI_TRY(index_build_records_from_parse(idx, bytes_dev, &sc, nullptr, file_offset, &declined));
    } else {
      FastaScan fa;
      KMI_TRY(fasta_scan(ctx, &idx->cfg, bytes_dev, n_bytes, file_offset, true, &fa));
      if (fa.n_chars < idx->shape.k) return KMI_OK;
      KMI_TRY(index_build_records_from_parse(idx, bytes_dev, nullptr, &fa, file_offset, &declined));
    }
    if (!declined) return KMI_OK;   // (declined: lines of a few bytes crowd a window of the list pass -- the extract pass below takes the input)
  }
  const uint32_t nw = idx->shape.n_words, vw = idx->val_words;
  uint64_t nt = 0, ns = 0;
  KMI_TRY(extract_count(ctx, &idx->cfg, bytes_dev, n_bytes, &nt, &ns));
  if (nt == 0) return KMI_OK;
  // the tuples leave the extract pass as the parsers' separate arrays -- k-mers, ids and (FASTQ, position + quality) one dense
  // float array -- and the first partition pass gathers a record from them: its histogram reads the k-mers alone, and a quality
  // word written into 24-byte records afterwards would cost a read-modify-write of every line. (Without quality lines the
  // quality word of a position + quality index stays zero.)
  void *dk, *di, *dq = nullptr;
  KMI_TRY(ws_get(ctx, WS_OUTPUT, (size_t)nt * nw * sizeof(uint64_t), &dk));
  KMI_TRY(ws_get(ctx, WS_OUTPUT2, (size_t)nt * sizeof(uint64_t), &di));
  if (vw == 2 && idx->cfg.seq_format == KMI_FMT_FASTQ) KMI_TRY(ws_get(ctx, WS_INPUT2, (size_t)nt * sizeof(float) + 64, &dq));
  KMI_TRY(extract_run(ctx, &idx->cfg, bytes_dev, n_bytes, file_offset, (uint64_t *)dk, (uint64_t *)di, (size_t)nt, false, true, &nt, &ns, (float *)dq));
  return index_insert_records(idx, (const uint64_t *)dk, (size_t)nt, true, (const float *)dq, (const uint64_t *)di);
}

kmi_status kmi_index_build_host(kmi_index *idx, const uint8_t *bytes, size_t n_bytes, uint64_t file_offset) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  if (n_bytes == 0) return KMI_OK;
  if (!bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *din;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &din));
  // the copy is not queued here: the one-pass front end queues it in chunks and works on the ranges of a chunk while the next one is
  // on its way (sk_front_fast); every other path queues it whole before it reads the input (feed_flush). Worth it from 64 MB on; the
  // caller's buffer should be pinned (hipHostMalloc / hipHostRegister) -- a copy from pageable memory is staged by the runtime and
  // overlaps with nothing.
  if (ctx->host_overlap && n_bytes >= ctx->host_overlap_min) { ctx->feed_host = bytes; ctx->feed_dev = (uint8_t *)din; ctx->feed_bytes = n_bytes; }
  else KMI_HIP(ctx, hipMemcpyAsync(din, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  const kmi_status st = kmi_index_build_dev(idx, (const uint8_t *)din, n_bytes, file_offset);
  if (ctx->feed_host) { ctx->feed_host = nullptr; if (st == KMI_OK) return set_err(ctx, KMI_ERR_DEVICE, "the input was never copied"); }
  return st;
}

kmi_status kmi_index_clear(kmi_index *idx) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_index_arrays(idx);
  idx->n_entries = 0; idx->has_data = false; idx->owner_lp = 0;
  return KMI_OK;
}

kmi_status kmi_index_local_size(kmi_index *idx, uint64_t *n) {
  if (!idx || !n) return KMI_ERR_INVALID;
  *n = idx->n_entries;
  return KMI_OK;
}

kmi_status kmi_index_export_host(kmi_index *idx, uint64_t *keys, uint32_t *counts, size_t capacity, uint64_t *n) {
  if (!idx || !n) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  *n = 0;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "a position index exports tuples: kmi_index_export_tuples_host");
  if (idx->n_entries == 0) return KMI_OK;
  if (capacity < idx->n_entries) return set_err(ctx, KMI_ERR_OVERFLOW, "export: capacity too small");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  KMI_TRY(ensure_dense(idx));
  if (keys) KMI_HIP(ctx, hipMemcpyAsync(keys, idx->keys, idx->n_entries * idx->shape.n_words * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (counts) KMI_HIP(ctx, hipMemcpyAsync(counts, idx->vals, idx->n_entries * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n = idx->n_entries;
  return KMI_OK;
}

void kmi_results_free(kmi_results *r) {
  if (!r) return;
  free(r->keys); free(r->values);
  memset(r, 0, sizeof(*r));
}

static kmi_status query_host(kmi_index *idx, int mode, const uint64_t *queries, size_t nq, kmi_results *out, uint64_t *n_erased) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  if (out) memset(out, 0, sizeof(*out));
  if (n_erased) *n_erased = 0;
  if (nq == 0) return KMI_OK;
  if (!queries) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t nw = idx->shape.n_words, ow = idx->val_words ? idx->val_words : 1u;
  const uint64_t bound = query_result_bound(idx, mode, nq);
  void *dq, *dk = nullptr, *dv = nullptr;
  KMI_TRY(ws_get(ctx, WS_INPUT, nq * nw * sizeof(uint64_t), &dq));
  KMI_HIP(ctx, hipMemcpyAsync(dq, queries, nq * nw * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  const bool sized_inside = idx->val_words > 0 && mode == Q_FIND;   // a multimap find sizes its result buffers from a count of the hits
  if (mode != Q_ERASE && !sized_inside) {
    KMI_TRY(ws_get(ctx, WS_OUTPUT, (bound ? bound : 1) * nw * sizeof(uint64_t), &dk));
    KMI_TRY(ws_get(ctx, WS_OUTPUT2, (bound ? bound : 1) * ow * sizeof(uint64_t), &dv));
  }
  uint64_t n = 0;
  if (sized_inside) KMI_TRY(index_query(idx, mode, (const uint64_t *)dq, nq, nullptr, nullptr, 0, &n, (uint64_t **)&dk, (uint64_t **)&dv));
  else KMI_TRY(index_query(idx, mode, (const uint64_t *)dq, nq, (uint64_t *)dk, (uint64_t *)dv, bound, &n));
  if (mode == Q_ERASE) { if (n_erased) *n_erased = n; return KMI_OK; }
  out->n = n;
  out->keys = (uint64_t *)malloc((n ? n : 1) * nw * sizeof(uint64_t));
  out->values = (uint64_t *)malloc((n ? n : 1) * ow * sizeof(uint64_t));
  if (!out->keys || !out->values) return set_err(ctx, KMI_ERR_NOMEM, "host malloc failed");
  if (n) {
    KMI_HIP(ctx, hipMemcpyAsync(out->keys, dk, n * nw * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(out->values, dv, n * ow * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return KMI_OK;
}

kmi_status kmi_index_count_host(kmi_index *idx, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!out) return KMI_ERR_INVALID;
  return query_host(idx, Q_COUNT, queries, nq, out, nullptr);
}
kmi_status kmi_index_find_host(kmi_index *idx, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!out) return KMI_ERR_INVALID;
  return query_host(idx, Q_FIND, queries, nq, out, nullptr);
}
kmi_status kmi_index_erase_host(kmi_index *idx, const uint64_t *queries, size_t nq, uint64_t *n_erased) {
  return query_host(idx, Q_ERASE, queries, nq, nullptr, n_erased);
}
kmi_status kmi_index_count_dev(kmi_index *idx, const uint64_t *queries_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_values_dev,
                               uint64_t *n_out) {
  if (!idx) return KMI_ERR_INVALID;
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  return index_query(idx, Q_COUNT, queries_dev, nq, out_keys_dev, out_values_dev, nq, n_out);
}
kmi_status kmi_index_find_dev(kmi_index *idx, const uint64_t *queries_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_values_dev,
                              uint64_t *n_out) {
  if (!idx) return KMI_ERR_INVALID;
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  return index_query(idx, Q_FIND, queries_dev, nq, out_keys_dev, out_values_dev, query_result_bound(idx, Q_FIND, nq), n_out);
}

kmi_status kmi_index_find_hits_dev(kmi_index *idx, const uint64_t *queries_dev, size_t nq, uint64_t *n_hits) {
  if (!idx || !n_hits) return KMI_ERR_INVALID;
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  if (idx->val_words == 0) { *n_hits = nq; return KMI_OK; }   // (a counting map answers every distinct query key once at most)
  return index_query(idx, Q_FIND, queries_dev, nq, nullptr, nullptr, 0, n_hits, nullptr, nullptr, true);
}

// ---- multimap (PositionIndex / PositionQualityIndex) entry points
kmi_status kmi_index_insert_tuples_dev(kmi_index *idx, const uint64_t *records_dev, size_t n) {
  if (!idx) return KMI_ERR_INVALID;
  if (idx->val_words == 0) return set_err(idx->ctx, KMI_ERR_INVALID, "insert_tuples needs a position index");
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  return index_insert_records(idx, records_dev, n, true);
}

kmi_status kmi_index_insert_tuples_host(kmi_index *idx, const uint64_t *kmers, const uint64_t *values, size_t n) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words == 0) return set_err(ctx, KMI_ERR_INVALID, "insert_tuples needs a position index");
  if (n == 0) return KMI_OK;
  if (!kmers || !values) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t nw = idx->shape.n_words, vw = idx->val_words, rw = nw + vw;
  // interleave into records (key words, value words) -- the layout of std::pair<Kmer, value>
  uint64_t *rec = (uint64_t *)malloc(n * rw * sizeof(uint64_t));
  if (!rec) return set_err(ctx, KMI_ERR_NOMEM, "host malloc failed");
  for (size_t i = 0; i < n; ++i) {
    memcpy(rec + i * rw, kmers + i * nw, nw * sizeof(uint64_t));
    memcpy(rec + i * rw + nw, values + i * vw, vw * sizeof(uint64_t));
  }
  void *din;
  kmi_status st = ws_get(ctx, WS_INPUT, n * rw * sizeof(uint64_t), &din);
  if (st == KMI_OK && hipMemcpy(din, rec, n * rw * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess)
    st = set_err(ctx, KMI_ERR_DEVICE, "hipMemcpy failed");
  free(rec);
  if (st != KMI_OK) return st;
  return index_insert_records(idx, (const uint64_t *)din, n, true);
}

kmi_status kmi_index_export_tuples_host(kmi_index *idx, uint64_t *keys, uint64_t *values, size_t capacity, uint64_t *n) {
  if (!idx || !n) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  *n = 0;
  if (idx->val_words == 0) return set_err(ctx, KMI_ERR_INVALID, "export_tuples needs a position index");
  if (idx->n_entries == 0) return KMI_OK;
  if (capacity < idx->n_entries) return set_err(ctx, KMI_ERR_OVERFLOW, "export: capacity too small");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  KMI_TRY(ensure_dense(idx));
  if (keys) KMI_HIP(ctx, hipMemcpyAsync(keys, idx->keys, idx->n_entries * idx->shape.n_words * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (values) KMI_HIP(ctx, hipMemcpyAsync(values, idx->mvals, idx->n_entries * idx->val_words * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n = idx->n_entries;
  return KMI_OK;
}

uint32_t kmi_index_num_buckets(void) { return (uint32_t)kmi::kNumFine; }

kmi_status kmi_index_split_by_rank_dev(kmi_index *idx, uint32_t nranks, uint64_t *out_kmers_dev, uint32_t *out_counts_dev, size_t capacity,
                                       uint32_t *bucket_counts_dev, uint64_t *send_counts_host) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "only the count index splits into (k-mer, count) pairs");
  if (nranks == 0 || nranks > (uint32_t)kmi::kNumCoarse || !send_counts_host || !bucket_counts_dev)
    return set_err(ctx, KMI_ERR_INVALID, "nranks must be in 1..256 and the count buffers non-null");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (idx->n_entries == 0) {
    for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0;
    KMI_HIP(ctx, hipMemsetAsync(bucket_counts_dev, 0, sizeof(uint32_t) * (size_t)nranks * kmi::kNumFine, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMI_OK;
  }
  if (!out_kmers_dev || !out_counts_dev) return set_err(ctx, KMI_ERR_INVALID, "null output buffer");
  KMI_DISPATCH(idx->shape, kmi::split_impl, idx, nranks, out_kmers_dev, out_counts_dev, capacity, bucket_counts_dev, send_counts_host);
}

kmi_status kmi_index_merge_parts_dev(kmi_index *idx, uint32_t nparts, const uint64_t *kmers_dev, const uint32_t *counts_dev,
                                     const uint32_t *bucket_counts_dev) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "only the count index merges (k-mer, count) pairs");
  if (nparts == 0 || nparts > (uint32_t)kmi::kNumCoarse || !bucket_counts_dev) return set_err(ctx, KMI_ERR_INVALID, "nparts must be in 1..256");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  KMI_DISPATCH(idx->shape, kmi::merge_impl, idx, nparts, kmers_dev, counts_dev, bucket_counts_dev);
}

// ---- the collectives of Index<MapType, Parser> over ranks (see kmerind_hip.h) ----------------------------------------
static kmi_status dist_check(kmi_index *idx, kmi_comm *comm) {
  if (!idx || !comm) return KMI_ERR_INVALID;
  if (kmi::comm_ctx(comm) != idx->ctx) return set_err(idx->ctx, KMI_ERR_INVALID, "the communicator belongs to another context");
  KMI_HIP(idx->ctx, hipSetDevice(idx->ctx->device));
  return KMI_OK;
}

// imxx::distribute of `n` items already grouped by destination in send_dev: counts, then payload; *recv_dev (workspace
// slot `slot`) holds the concatenation by source rank, recv_counts what every source sent
static kmi_status dist_exchange(kmi_comm *comm, const void *send_dev, const uint64_t *send_counts, size_t elem_bytes, kmi::WsSlot slot,
                                void **recv_dev, std::vector<uint64_t> &recv_counts, uint64_t *total) {
  kmi_ctx *ctx = kmi::comm_ctx(comm);
  const int p = kmi::comm_size(comm);
  recv_counts.assign(p, 0);
  KMI_TRY(kmi::comm_all_to_all_counts(comm, send_counts, recv_counts.data()));
  uint64_t t = 0;
  for (int r = 0; r < p; ++r) t += recv_counts[r];
  KMI_TRY(ws_get(ctx, slot, (t + 8) * elem_bytes, recv_dev));
  KMI_TRY(kmi::comm_all_to_all_v(comm, send_dev, send_counts, *recv_dev, recv_counts.data(), elem_bytes));
  *total = t;
  return KMI_OK;
}

kmi_status kmi_index_insert_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *kmers, size_t n) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "a position index takes (k-mer, value) tuples: kmi_index_insert_tuples_dist_host");
  if (n && !kmers) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const int p = kmi::comm_size(comm);
  if (p == 1 && !ctx->force_dist) return kmi_index_insert_host(idx, kmers, n);
  const size_t kb = idx->shape.n_words * sizeof(uint64_t);
  void *d_in, *d_send, *d_recv;
  KMI_TRY(ws_get(ctx, WS_INPUT, (n + 8) * kb, &d_in));
  KMI_TRY(ws_get(ctx, WS_DIST_A, (n + 8) * kb, &d_send));
  if (n) KMI_HIP(ctx, hipMemcpyAsync(d_in, kmers, n * kb, hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint64_t> sc(p, 0), rc;
  // InputTransform + grouping by KeyToRank -- or by the owner of the minimizer's bucket when that is how the entries are distributed
  // (agreed over the ranks: a rank that happens to hold nothing must not route differently from its peers)
  uint64_t holders = 0, owner_p = 0, owner_any = 0;
  KMI_TRY(dist_state(idx, comm, &holders, &owner_p, &owner_any));
  if (owner_any && holders) {
    if (!idx->owner_lp) idx->owner_lp = 31u - (uint32_t)__builtin_clz((uint32_t)p);
    KMI_TRY(kmi_route_owner_dev(ctx, &idx->cfg, (const uint64_t *)d_in, n, (uint32_t)p, (uint64_t *)d_send, sc.data()));
  } else KMI_TRY(kmi_route_dev(ctx, &idx->cfg, (const uint64_t *)d_in, n, (uint32_t)p, (uint64_t *)d_send, sc.data()));
  uint64_t total = 0;
  KMI_TRY(dist_exchange(comm, d_send, sc.data(), kb, WS_DIST_B, &d_recv, rc, &total));
  return index_insert(idx, (const uint64_t *)d_recv, (size_t)total, false);
}

kmi_status kmi_index_insert_tuples_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *kmers, const uint64_t *values, size_t n) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words == 0) return set_err(ctx, KMI_ERR_INVALID, "insert_tuples needs a position index");
  if (n && (!kmers || !values)) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const int p = kmi::comm_size(comm);
  if (p == 1 && !ctx->force_dist) return kmi_index_insert_tuples_host(idx, kmers, values, n);
  const uint32_t nw = idx->shape.n_words, vw = idx->val_words, rw = nw + vw;
  std::vector<uint64_t> rec((n + 1) * rw);
  for (size_t i = 0; i < n; ++i) {
    memcpy(&rec[i * rw], kmers + i * nw, nw * sizeof(uint64_t));
    memcpy(&rec[i * rw + nw], values + i * vw, vw * sizeof(uint64_t));
  }
  void *d_in, *d_send, *d_recv;
  KMI_TRY(ws_get(ctx, WS_INPUT, (n + 8) * rw * sizeof(uint64_t), &d_in));
  KMI_TRY(ws_get(ctx, WS_DIST_A, (n + 8) * rw * sizeof(uint64_t), &d_send));
  if (n) KMI_HIP(ctx, hipMemcpy(d_in, rec.data(), n * rw * sizeof(uint64_t), hipMemcpyHostToDevice));
  std::vector<uint64_t> sc(p, 0), rc;
  KMI_TRY(kmi_route_tuples_dev(ctx, &idx->cfg, (const uint64_t *)d_in, n, (uint32_t)p, vw, (uint64_t *)d_send, sc.data()));
  uint64_t total = 0;
  KMI_TRY(dist_exchange(comm, d_send, sc.data(), rw * sizeof(uint64_t), WS_DIST_B, &d_recv, rc, &total));
  return index_insert_records(idx, (const uint64_t *)d_recv, (size_t)total, true);   // (the transform is idempotent)
}

// weighted insert and update() of the counting maps over ranks (distributed_unordered_map.hpp:1603-1618 behind insert's
// distribute; distributed_densehash_map.hpp:1975-2030): the (k-mer, value) pairs travel to the ranks that own their keys --
// KeyToRank, or the owner of the minimizer bucket when that is how the entries are distributed (agreed over the ranks) -- and
// are applied there. records: n x (n_words key words, one value word).
static kmi_status pairs_to_owners(kmi_index *idx, kmi_comm *comm, const uint64_t *records, size_t n, void **d_recv, uint64_t *total) {
  kmi_ctx *ctx = idx->ctx;
  const int p = kmi::comm_size(comm);
  const uint32_t rw = idx->shape.n_words + 1;
  void *d_in, *d_send;
  KMI_TRY(ws_get(ctx, WS_INPUT, (n + 8) * rw * sizeof(uint64_t), &d_in));
  KMI_TRY(ws_get(ctx, WS_DIST_A, (n + 8) * rw * sizeof(uint64_t), &d_send));
  if (n) KMI_HIP(ctx, hipMemcpyAsync(d_in, records, n * rw * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  uint64_t holders = 0, owner_p = 0, owner_any = 0;
  KMI_TRY(dist_state(idx, comm, &holders, &owner_p, &owner_any));
  const bool by_owner = owner_any != 0 && holders != 0;
  if (by_owner && !idx->owner_lp) idx->owner_lp = 31u - (uint32_t)__builtin_clz((uint32_t)p);
  std::vector<uint64_t> sc(p, 0), rc;
  KMI_TRY(route_pairs(ctx, &idx->cfg, idx->shape, (const uint64_t *)d_in, n, (uint32_t)p, by_owner, (uint64_t *)d_send, sc.data()));
  return dist_exchange(comm, d_send, sc.data(), rw * sizeof(uint64_t), WS_DIST_B, d_recv, rc, total);
}

kmi_status kmi_index_insert_pairs_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *records, size_t n) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "(k-mer, count) pairs go into a count index");
  if (n && !records) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  if (kmi::comm_size(comm) == 1 && !ctx->force_dist) return kmi_index_insert_pairs_host(idx, records, n);
  void *d_recv; uint64_t total = 0;
  KMI_TRY(pairs_to_owners(idx, comm, records, n, &d_recv, &total));
  const uint32_t lp = idx->owner_lp;
  kmi_status st = index_insert_pairs(idx, (const uint64_t *)d_recv, (size_t)total, true, false);   // (the strand transform is idempotent)
  idx->owner_lp = lp;
  return st;
}

// the routing half of the collectives above, on its own: every rank brings (k-mer, value) pairs, every rank gets back -- in host
// memory -- the pairs whose keys IT owns, keys as the map stores them (strand transform applied), grouped by source rank (the order
// inside a group is the routing scatter's, unspecified). What update() with a HOST functor needs over ranks (distributed_densehash_map.hpp:1975-2030: distribute,
// then the local update): the facade applies the functor to the owner's entries on the owner's host.
kmi_status kmi_index_route_pairs_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *records, size_t n, kmi_results *out) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (!out) return KMI_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "(k-mer, value) pairs of a counting map");
  if (n && !records) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const uint32_t nw = idx->shape.n_words, rw = nw + 1;
  void *d_recv = nullptr; uint64_t total = 0;
  if (kmi::comm_size(comm) == 1 && !ctx->force_dist) {
    // one rank: the pairs with their keys transformed (route_pairs with one destination does exactly that)
    void *d_in, *d_send;
    KMI_TRY(ws_get(ctx, WS_INPUT, (n + 8) * rw * sizeof(uint64_t), &d_in));
    KMI_TRY(ws_get(ctx, WS_DIST_A, (n + 8) * rw * sizeof(uint64_t), &d_send));
    if (n) KMI_HIP(ctx, hipMemcpyAsync(d_in, records, n * rw * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    uint64_t sc1 = 0;
    KMI_TRY(route_pairs(ctx, &idx->cfg, idx->shape, (const uint64_t *)d_in, n, 1u, false, (uint64_t *)d_send, &sc1));
    d_recv = d_send; total = n;
  } else KMI_TRY(pairs_to_owners(idx, comm, records, n, &d_recv, &total));
  if (total == 0) return KMI_OK;
  std::vector<uint64_t> h((size_t)total * rw);
  KMI_HIP(ctx, hipMemcpyAsync(h.data(), d_recv, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out->keys = (uint64_t *)malloc((size_t)total * nw * sizeof(uint64_t));
  out->values = (uint64_t *)malloc((size_t)total * sizeof(uint64_t));
  if (!out->keys || !out->values) { kmi_results_free(out); return set_err(ctx, KMI_ERR_NOMEM, "host results"); }
  for (uint64_t i = 0; i < total; ++i) {
    memcpy(out->keys + i * nw, &h[i * rw], nw * sizeof(uint64_t));
    out->values[i] = h[i * rw + nw];
  }
  out->n = total;
  return KMI_OK;
}

kmi_status kmi_index_update_pairs_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *records, size_t n, uint32_t op, uint64_t *n_updated) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (!n_updated) return KMI_ERR_INVALID;
  *n_updated = 0;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "update() is a member of the counting maps");
  if (op > KMI_UPDATE_ASSIGN) return set_err(ctx, KMI_ERR_INVALID, "unknown updater");
  if (n && !records) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  if (kmi::comm_size(comm) == 1 && !ctx->force_dist) return kmi_index_update_pairs_host(idx, records, n, op, n_updated);
  void *d_recv; uint64_t total = 0;
  KMI_TRY(pairs_to_owners(idx, comm, records, n, &d_recv, &total));   // (pairs of one key arrive in source-rank order, each source's in input order)
  if (total == 0) return KMI_OK;
  return index_update_pairs(idx, (uint64_t *)d_recv, (size_t)total, (int)op, n_updated);
}

}  // extern "C"

static int comm_rank_of(kmi_comm *comm) { return kmi::comm_rank(comm); }
static kmi_status dist_agree(kmi_comm *comm, kmi_status mine) {
  uint64_t bad = mine != KMI_OK ? 1u : 0u;
  const kmi_status st = kmi::comm_allreduce_sum(comm, &bad);
  if (mine != KMI_OK) return mine;
  if (st != KMI_OK) return st;
  if (bad) return set_err(kmi::comm_ctx(comm), KMI_ERR_PEER, "the collective was given up: another rank reported an error");
  return KMI_OK;
}

// one 64-bit all-reduce carries what every rank must agree on before it picks a route: field 0 (bits 0..15) ranks that hold
// entries, field 1 (16..31) ranks whose entries are distributed by minimizer owner over exactly p ranks, field 2 (32..47) ranks
// whose entries are distributed by owner at all
static kmi_status dist_state(kmi_index *idx, kmi_comm *comm, uint64_t *holders, uint64_t *owner_p, uint64_t *owner_any) {
  const int p = kmi::comm_size(comm);
  uint64_t v = (idx->n_entries ? 1ull : 0ull) | ((idx->owner_lp && (1u << idx->owner_lp) == (uint32_t)p) ? 1ull << 16 : 0ull) | (idx->owner_lp ? 1ull << 32 : 0ull);
  KMI_TRY(kmi::comm_allreduce_sum(comm, &v));
  *holders = v & 0xffffu; *owner_p = (v >> 16) & 0xffffu; *owner_any = (v >> 32) & 0xffffu;
  return KMI_OK;
}

// The count index over ranks through exchanged super-k-mer records, CHUNKED: the rank's share is cut into record-aligned chunks;
// the records of chunk c travel on the communicator's stream while the front end of chunk c + 1 runs on the context's; the
// counts of a chunk ride with the verdict ("this rank could not produce it": then every rank sends that chunk as k-mers to the
// same owners afterwards) and with the sender's largest message. What arrived is consumed in one go.
template <int W>
static kmi_status build_dist_superkmer(kmi_index *idx, kmi_comm *comm, const uint8_t *d_bytes, size_t n_bytes, uint64_t file_offset) {
  kmi_ctx *ctx = idx->ctx;
  const int p = kmi::comm_size(comm);
  const uint32_t lp = 31u - (uint32_t)__builtin_clz((uint32_t)p);
  const uint32_t nch = (idx->cfg.seq_format == KMI_FMT_FASTQ) ? ctx->dist_chunks : 1u;
  std::vector<uint64_t> cuts(nch + 1, 0);
  if (n_bytes) {
    if (nch > 1) KMI_TRY(kmi_fastq_partition_dev(ctx, d_bytes, n_bytes, nch, cuts.data()));
    else cuts[1] = n_bytes;
  }
  // the receive pool: every rank's bytes are known after one all-reduce; a rank receives about a p-th of all records
  uint64_t all_bytes = n_bytes;
  KMI_TRY(kmi::comm_allreduce_sum(comm, &all_bytes));
  uint64_t pool_cap = (uint64_t)((double)all_bytes / p * 0.06 * 1.3 * ctx->dist_pool_pct / 100.0) + ctx->dist_pool_slack;
  void *pv;
  KMI_TRY(ws_get(ctx, WS_DIST_B, pool_cap * 16, &pv));
  uint64_t *pool = (uint64_t *)pv;
  uint64_t pool_pos = 0;
  bool pool_in_b = true;
  uint64_t *sendbuf[2] = {nullptr, nullptr};
  size_t send_cap[2] = {0, 0};
  std::vector<uint32_t> failed;
  std::vector<uint64_t> sc(p), rc(p);
  const uint32_t lp_before = idx->owner_lp;
  idx->owner_lp = lp;   // (every rank is here: the route was agreed on; an error below puts the old value back)
  // A rank that fails on its own -- a parse or length verdict of its front end, a workspace it cannot get -- still enters the
  // chunk's count exchange and says so there (count = kHardError), so that every rank leaves this function with an error instead
  // of waiting in a send / receive for a peer that has already returned.
  constexpr uint64_t kNotProduced = ~0ull, kHardError = ~0ull - 1ull;
  kmi_status st_local = KMI_OK;
  bool peer_error = false;
  for (uint32_t c = 0; c < nch && st_local == KMI_OK && !peer_error; ++c) {
    const size_t cb = (size_t)(cuts[c + 1] - cuts[c]);
    const int sb = (int)(c & 1u);
    int produced = 0;
    auto produce_chunk = [&]() -> kmi_status {
      const size_t want = (size_t)((double)cb * 0.06) + 8192;
      if (c >= 2) KMI_TRY(kmi::comm_exchange_wait(comm));   // the buffer's previous message has left (the transfer before the last is done)
      if (send_cap[sb] < want) { KMI_TRY(ws_get(ctx, sb ? WS_DIST_C : WS_DIST_A, want * 16, &pv)); sendbuf[sb] = (uint64_t *)pv; send_cap[sb] = want; }
      const uint64_t *recs = nullptr;
      uint64_t nrec = 0;
      const uint8_t *src = d_bytes + cuts[c];   // (any address: the one-pass front end loads unaligned; the general one aligns its input itself)
      if (ctx->sk_dbg == 9 && comm_rank_of(comm) == 1 && c == 1) return set_err(ctx, KMI_ERR_PARSE, "test knob KMI_SK_DBG=9: rank 1 fails on its second chunk");
      KMI_TRY(sk_produce(idx, (uint32_t)W, src, cb, (uint32_t)p, &recs, &nrec, sc.data(), &produced, sendbuf[sb], send_cap[sb]));
      if (produced && nrec && recs != sendbuf[sb]) {   // more records than the buffer was sized for: a larger one, and a copy out of the workspace
        KMI_TRY(kmi::comm_exchange_wait(comm));
        KMI_TRY(ws_get(ctx, sb ? WS_DIST_C : WS_DIST_A, (nrec + 64) * 16, &pv)); sendbuf[sb] = (uint64_t *)pv; send_cap[sb] = nrec + 64;
        KMI_HIP(ctx, hipMemcpyAsync(sendbuf[sb], recs, nrec * 16, hipMemcpyDeviceToDevice, ctx->stream));
      }
      return KMI_OK;
    };
    st_local = produce_chunk();
    uint64_t mine = 0, largest = 0;
    for (int r = 0; r < p; ++r) mine = std::max(mine, sc[r] * 16);
    std::vector<uint64_t> tell(sc);
    if (st_local != KMI_OK) { for (int r = 0; r < p; ++r) tell[r] = kHardError; mine = 0; }
    else if (!produced) for (int r = 0; r < p; ++r) tell[r] = kNotProduced;
    {
      const kmi_status st_x = kmi::comm_all_to_all_counts2(comm, tell.data(), mine, rc.data(), &largest);
      if (st_x != KMI_OK) { idx->owner_lp = lp_before; return st_local != KMI_OK ? st_local : st_x; }   // (the exchange itself failed: nothing more to agree on)
    }
    for (int r = 0; r < p; ++r) peer_error = peer_error || rc[r] == kHardError;
    if (st_local != KMI_OK || peer_error) break;
    bool all_ok = produced != 0;
    for (int r = 0; r < p; ++r) all_ok = all_ok && rc[r] != kNotProduced;
    if (!all_ok) { failed.push_back(c); continue; }
    uint64_t n_in = 0;
    for (int r = 0; r < p; ++r) n_in += rc[r];
    // (from here to the payload exchange only allocation can fail: a rank that cannot hold what it is sent has no way left to say so)
    if (pool_pos + n_in > pool_cap) {   // the estimate was short (a very uneven input): a pool twice the need, what arrived so far moves over
      KMI_TRY(kmi::comm_exchange_wait(comm));
      const uint64_t ncap = 2 * (pool_pos + n_in) + ctx->dist_pool_slack;
      ++ctx->dist_pool_regrows;
      KMI_TRY(ws_get(ctx, pool_in_b ? WS_DIST_D : WS_DIST_B, ncap * 16, &pv));
      if (pool_pos) KMI_HIP(ctx, hipMemcpyAsync(pv, pool, pool_pos * 16, hipMemcpyDeviceToDevice, ctx->stream));
      pool = (uint64_t *)pv; pool_cap = ncap; pool_in_b = !pool_in_b;
    }
    KMI_TRY(kmi::comm_all_to_all_v_async(comm, sendbuf[sb], sc.data(), pool + 2 * pool_pos, rc.data(), 16, largest));
    pool_pos += n_in;
  }
  KMI_TRY(kmi::comm_exchange_join(comm));
  if (st_local != KMI_OK || peer_error) {
    // what arrived before the error is dropped: the index is as it was (the transfers queued so far have been joined)
    (void)hipStreamSynchronize(ctx->stream);
    idx->owner_lp = lp_before;
    if (st_local != KMI_OK) return st_local;
    return set_err(ctx, KMI_ERR_PEER, "the build over ranks was given up: another rank reported an error in its share of the input");
  }
  {
    kmi_status st_c = pool_pos ? sk_consume(idx, (uint32_t)W, pool, pool_pos, (uint32_t)p) : KMI_OK;
    // chunks some rank could not produce as records follow as k-mers, collectively: first agree that every rank is still there
    if (!failed.empty()) st_c = dist_agree(comm, st_c);
    if (st_c != KMI_OK) { if (!pool_pos || !idx->n_entries) idx->owner_lp = lp_before; return st_c; }
  }
  idx->owner_lp = lp;
  // chunks some rank could not produce as records: their k-mers, routed to the same owners (collective: every rank has the same list)
  for (uint32_t c : failed) {
    const size_t cb = (size_t)(cuts[c + 1] - cuts[c]);
    const uint32_t nw = idx->shape.n_words;
    uint64_t nt = 0, ns = 0, total = 0;
    void *d_keys = nullptr, *d_send = nullptr, *d_recv = nullptr;
    auto route_chunk = [&]() -> kmi_status {
      const uint8_t *src = d_bytes + cuts[c];
      if (cb) { KMI_TRY(align_input(ctx, &src, cb)); KMI_TRY(extract_count(ctx, &idx->cfg, src, cb, &nt, &ns)); }
      KMI_TRY(ws_get(ctx, WS_OUTPUT, (nt + 64) * nw * sizeof(uint64_t), &d_keys));
      KMI_TRY(ws_get(ctx, WS_DIST_A, (nt + 64) * nw * sizeof(uint64_t), &d_send));
      send_cap[0] = 0;
      for (int r = 0; r < p; ++r) sc[r] = 0;
      if (nt) {
        KMI_TRY(extract_run(ctx, &idx->cfg, src, cb, file_offset + cuts[c], (uint64_t *)d_keys, nullptr, (size_t)nt, true, true, &nt, &ns));
        KMI_TRY(kmi_route_owner_dev(ctx, &idx->cfg, (const uint64_t *)d_keys, (size_t)nt, (uint32_t)p, (uint64_t *)d_send, sc.data()));
      }
      return KMI_OK;
    };
    KMI_TRY(dist_agree(comm, route_chunk()));
    std::vector<uint64_t> rcv;
    KMI_TRY(dist_exchange(comm, d_send, sc.data(), nw * sizeof(uint64_t), WS_DIST_D, &d_recv, rcv, &total));
    KMI_TRY(index_insert(idx, (const uint64_t *)d_recv, (size_t)total, false));
    idx->owner_lp = lp;
  }
  return KMI_OK;
}

extern "C" {

kmi_status kmi_index_build_dist_dev(kmi_index *idx, kmi_comm *comm, const uint8_t *d_bytes, size_t n_bytes, uint64_t file_offset) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  const int p = kmi::comm_size(comm);
  if (p == 1 && !ctx->force_dist) return kmi_index_build_dev(idx, d_bytes, n_bytes, file_offset);
  if (n_bytes && !d_bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const uint32_t nw = idx->shape.n_words, vw = idx->val_words, rw = nw + vw;
  void *d_send = nullptr, *d_recv;
  std::vector<uint64_t> sc(p, 0), rc;
  uint64_t nt = 0, ns = 0, total = 0;
  if (vw == 0) {
    // the route follows from state every rank has agreed on (never from one rank's own entry count: a rank with an empty share
    // would otherwise take another branch than its peers, and the branches issue different collectives)
    uint64_t holders = 0, owner_p = 0, owner_any = 0;
    KMI_TRY(dist_state(idx, comm, &holders, &owner_p, &owner_any));
    if (owner_any && owner_any != (uint64_t)p && holders)
      return set_err(ctx, KMI_ERR_INVALID, "the ranks disagree on how this index is distributed (some by minimizer owner, some not)");
    const uint32_t skw = sk_width_of(idx);
    const bool by_owner = owner_any != 0 && holders != 0;                 // the entries are already distributed by minimizer owner
    if (skw && sk_rank_count((uint32_t)p) && (holders == 0 || owner_p == (uint64_t)p)) {
      if (holders == 0) idx->owner_lp = 0;
      switch (skw) {
        case 19: return build_dist_superkmer<19>(idx, comm, d_bytes, n_bytes, file_offset);
        case 13: return build_dist_superkmer<13>(idx, comm, d_bytes, n_bytes, file_offset);
        case 11: return build_dist_superkmer<11>(idx, comm, d_bytes, n_bytes, file_offset);
        default: return build_dist_superkmer<7>(idx, comm, d_bytes, n_bytes, file_offset);
      }
    }
    if (n_bytes) KMI_TRY(extract_count(ctx, &idx->cfg, d_bytes, n_bytes, &nt, &ns));   // (an empty share still enters the collectives)
    KMI_TRY(ws_get(ctx, WS_DIST_A, (nt + 64) * nw * sizeof(uint64_t), &d_send));
    if (by_owner) {
      // k-mers into an index that is distributed by minimizer owner go to those owners
      void *d_keys;
      KMI_TRY(ws_get(ctx, WS_OUTPUT, (nt + 64) * nw * sizeof(uint64_t), &d_keys));
      if (nt) {
        KMI_TRY(extract_run(ctx, &idx->cfg, d_bytes, n_bytes, file_offset, (uint64_t *)d_keys, nullptr, (size_t)nt, true, true, &nt, &ns));
        KMI_TRY(kmi_route_owner_dev(ctx, &idx->cfg, (const uint64_t *)d_keys, (size_t)nt, (uint32_t)p, (uint64_t *)d_send, sc.data()));
      }
    } else if (nt && idx->cfg.seq_format == KMI_FMT_FASTQ) {
      // read_file + the bucketing half of imxx::distribute, fused: the tuple array in file order never exists
      KMI_TRY(kmi_extract_route_dev(ctx, &idx->cfg, d_bytes, n_bytes, (uint32_t)p, (uint64_t *)d_send, (size_t)nt, &nt, &ns, sc.data()));
    } else if (nt) {
      void *d_keys;
      KMI_TRY(ws_get(ctx, WS_OUTPUT, (nt + 64) * nw * sizeof(uint64_t), &d_keys));
      KMI_TRY(extract_run(ctx, &idx->cfg, d_bytes, n_bytes, file_offset, (uint64_t *)d_keys, nullptr, (size_t)nt, true, true, &nt, &ns));
      KMI_TRY(kmi_route_dev(ctx, &idx->cfg, (const uint64_t *)d_keys, (size_t)nt, (uint32_t)p, (uint64_t *)d_send, sc.data()));
    }
    KMI_TRY(dist_exchange(comm, d_send, sc.data(), nw * sizeof(uint64_t), WS_DIST_B, &d_recv, rc, &total));
    return index_insert(idx, (const uint64_t *)d_recv, (size_t)total, false);
  }
  if (n_bytes) KMI_TRY(extract_count(ctx, &idx->cfg, d_bytes, n_bytes, &nt, &ns));
  if (vw == 2 && idx->cfg.seq_format != KMI_FMT_FASTQ)
    return set_err(ctx, KMI_ERR_INVALID, "a position + quality index over ranks is built from FASTQ partitions");
  KMI_TRY(ws_get(ctx, WS_DIST_A, (nt + 64) * rw * sizeof(uint64_t), &d_send));
  if (nt)
    KMI_TRY(kmi_extract_route_records_dev(ctx, &idx->cfg, d_bytes, n_bytes, file_offset, (uint32_t)p, (uint64_t *)d_send, (size_t)nt + 64,
                                          &nt, &ns, sc.data()));
  KMI_TRY(dist_exchange(comm, d_send, sc.data(), rw * sizeof(uint64_t), WS_DIST_B, &d_recv, rc, &total));
  return index_insert_records(idx, (const uint64_t *)d_recv, (size_t)total, true);
}

kmi_status kmi_index_build_range_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                           uint64_t nominal_bytes, int reaches_eof, int *need_more) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (!need_more) return KMI_ERR_INVALID;
  *need_more = 0;
  if (n_bytes && !bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  if (idx->cfg.seq_format != KMI_FMT_FASTQ) return set_err(ctx, KMI_ERR_INVALID, "a byte range of a file is cut at FASTQ record starts here (a FASTA partition comes with kmi_ctx_set_fasta_partition)");
  if (nominal_bytes > n_bytes) nominal_bytes = n_bytes;
  void *d_bytes;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &d_bytes));
  uint64_t pos[2] = {0, nominal_bytes}, cut[2] = {0, n_bytes};
  if (n_bytes) {
    KMI_HIP(ctx, hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    KMI_TRY(kmi_fastq_find_records_dev(ctx, (const uint8_t *)d_bytes, n_bytes, buffer_offset == 0, pos, 2, cut));
    if (nominal_bytes >= n_bytes) cut[1] = n_bytes;   // (the nominal range is the rest of the buffer)
  }
  if (cut[1] >= n_bytes && !reaches_eof && n_bytes) { *need_more = 1; return KMI_OK; }   // no record start behind the nominal end in what was read
  if (cut[0] >= n_bytes && !reaches_eof && n_bytes && buffer_offset != 0) { *need_more = 1; return KMI_OK; }
  if (cut[1] < cut[0]) cut[1] = cut[0];
  return kmi_index_build_dist_dev(idx, comm, (const uint8_t *)d_bytes + cut[0], (size_t)(cut[1] - cut[0]), buffer_offset + cut[0]);
}

kmi_status kmi_index_build_fasta_file_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes) {
  // every rank holds the WHOLE FASTA file; the block bookkeeping of an equal split over the ranks (where each block's valid range
  // begins, the machine state and the record count there: file.hpp:1436-1610 + fasta_loader.hpp:202-470) is computed on the device
  // from the whole buffer, every rank keeps its own block and enters the collective build with it
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (idx->cfg.seq_format != KMI_FMT_FASTA) return set_err(ctx, KMI_ERR_INVALID, "not a FASTA index");
  if (n_bytes && !bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const uint32_t p = (uint32_t)kmi::comm_size(comm), r = (uint32_t)kmi::comm_rank(comm);
  void *d_bytes;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &d_bytes));
  if (n_bytes) KMI_HIP(ctx, hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint64_t> be(2 * (size_t)p);
  std::vector<kmi_fasta_partition> parts(p);
  KMI_TRY(kmi_fasta_partition_dev(ctx, (const uint8_t *)d_bytes, n_bytes, p, idx->shape.k, be.data(), parts.data()));
  KMI_TRY(kmi_ctx_set_fasta_partition(ctx, &parts[r]));
  const kmi_status st = kmi_index_build_dist_dev(idx, comm, (const uint8_t *)d_bytes + be[2 * r], (size_t)(be[2 * r + 1] - be[2 * r]), be[2 * r]);
  (void)kmi_ctx_set_fasta_partition(ctx, nullptr);
  return st;
}

// FASTA over ranks by BYTE RANGE (file.hpp:1436-1610, fasta_loader.hpp:202-470): the rank brings only its block of an equal split of
// the file plus look-ahead, and what it cannot know from its own bytes -- the kind of line its first byte sits on, the records that
// start before it, whether the file opens with a header -- comes from the other ranks' block summaries (kmi_fasta_block_summary_dev:
// the line-kind machine over a block as a transfer function), gathered once and composed left to right. bytes = file bytes
// [buffer_offset, buffer_offset + n_bytes), the first nominal_bytes of them the rank's block; prev_byte = the file byte before the
// buffer (-1 at the file start); *need_more = 1: the k - 1 sequence characters behind the block (the last windows' overlap) do not
// end inside the look-ahead -- nothing collective has happened, the caller reads further and calls again.
// the bookkeeping of a block: *need_more (nothing collective has happened then), or the partition record and where the block's
// windows end inside the buffer. d_bytes = the buffer on the device (already in flight on the context's stream).
static kmi_status fasta_range_partition(kmi_ctx *ctx, kmi_comm *comm, uint32_t k, const uint8_t *bytes, const uint8_t *d_bytes, size_t n_bytes,
                                        uint64_t buffer_offset, uint64_t nominal_bytes, int reaches_eof, int prev_byte, int *need_more,
                                        kmi_fasta_partition *part_out, uint64_t *end_out) {
  const uint32_t p = (uint32_t)kmi::comm_size(comm), r = (uint32_t)kmi::comm_rank(comm);
  const bool first_ls = buffer_offset == 0 || prev_byte == (int)'\n';
  uint64_t mine[8] = {0, 0, 1, 0, 2, 0, 0, 0};
  KMI_TRY(kmi_fasta_block_summary_dev(ctx, d_bytes, (size_t)nominal_bytes, first_ls ? 1 : 0, mine));
  mine[6] = (buffer_offset == 0 && n_bytes) ? bytes[0] : 0;   // (rank 0: the file's first byte decides init_parser's index shift)
  mine[7] = nominal_bytes;
  // where the overlap ends: the (k - 1)-th sequence character at or behind the block's end, for every state the machine may be in there
  auto step = [](uint32_t &state, uint8_t c) {
    if (c == '>' || c == ';') state = KMI_FA_HEADER;
    else state = (state == KMI_FA_OUTSIDE) ? (uint32_t)KMI_FA_OUTSIDE : (uint32_t)KMI_FA_SEQUENCE;
  };
  uint64_t end_for[3] = {nominal_bytes, nominal_bytes, nominal_bytes};
  for (uint32_t st0 = 0; st0 < 3; ++st0) {
    if (k <= 1 || nominal_bytes >= n_bytes) { end_for[st0] = nominal_bytes < n_bytes ? nominal_bytes : n_bytes; continue; }
    uint32_t st = st0, need = k - 1u;
    uint64_t i = nominal_bytes;
    for (; i < n_bytes && need; ++i) {
      const uint8_t c = bytes[i];
      const bool ls = i == 0 ? first_ls : bytes[i - 1] == '\n';
      if (ls) step(st, c);
      if (st == KMI_FA_SEQUENCE && c != '\n' && c != '\r') --need;
    }
    if (need && !reaches_eof) { *need_more = 1; return KMI_OK; }
    end_for[st0] = need ? n_bytes : i;
  }
  if (k > 1 && nominal_bytes >= n_bytes && !reaches_eof && p > 1 && r + 1 < p) { *need_more = 1; return KMI_OK; }   // (no look-ahead at all behind a block that is not the file's last)
  // ---- collective from here on
  std::vector<uint64_t> all((size_t)p * 8);
  KMI_TRY(kmi::comm_allgather_words(comm, mine, 8, all.data()));
  uint32_t st = KMI_FA_OUTSIDE; uint64_t ev = 0;
  for (uint32_t q = 0; q < r; ++q) { const uint64_t *t = &all[(size_t)q * 8]; ev += t[2 * st + 1]; st = (uint32_t)t[2 * st]; }
  uint64_t first = 0;
  for (uint32_t q = 0; q < p; ++q) if (all[(size_t)q * 8 + 7]) { first = all[(size_t)q * 8 + 6]; break; }   // the first non-empty block opens the file
  kmi_fasta_partition part; memset(&part, 0, sizeof(part));
  part.valid_bytes = nominal_bytes;
  part.start_state = buffer_offset == 0 ? (uint32_t)KMI_FA_OUTSIDE : st;
  part.at_line_start = first_ls ? 1u : 0u;
  part.records_before = ev;
  part.index_shift = (first == '>' || first == ';') ? 0u : 1u;
  const uint32_t st_end = (uint32_t)mine[2 * part.start_state];   // the machine's state behind the block
  const uint64_t end = end_for[st_end] < nominal_bytes ? nominal_bytes : end_for[st_end];
  *part_out = part;
  *end_out = end < n_bytes ? end : n_bytes;
  return KMI_OK;
}

kmi_status kmi_index_build_fasta_range_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                                 uint64_t nominal_bytes, int reaches_eof, int prev_byte, int *need_more) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (!need_more) return KMI_ERR_INVALID;
  *need_more = 0;
  if (idx->cfg.seq_format != KMI_FMT_FASTA) return set_err(ctx, KMI_ERR_INVALID, "not a FASTA index");
  if ((n_bytes && !bytes) || nominal_bytes > n_bytes) return set_err(ctx, KMI_ERR_INVALID, "bad buffer");
  void *d_bytes;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &d_bytes));
  if (n_bytes) KMI_HIP(ctx, hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  kmi_fasta_partition part; uint64_t end = 0;
  KMI_TRY(fasta_range_partition(ctx, comm, idx->shape.k, bytes, (const uint8_t *)d_bytes, n_bytes, buffer_offset, nominal_bytes, reaches_eof, prev_byte,
                                need_more, &part, &end));
  if (*need_more) return KMI_OK;
  KMI_TRY(kmi_ctx_set_fasta_partition(ctx, &part));
  const kmi_status stb = kmi_index_build_dist_dev(idx, comm, (const uint8_t *)d_bytes, (size_t)end, buffer_offset);
  (void)kmi_ctx_set_fasta_partition(ctx, nullptr);
  return stb;
}

// read_file_* of a FASTA file on one rank of several, by byte range (kmer_file_helper.hpp:550-633 over FASTALoader's partitions,
// fasta_loader.hpp:202-470): the same bookkeeping, then the block's tuples to the host. Collective over comm (one small gather).
kmi_status kmi_extract_fasta_range_dist_host(kmi_ctx *ctx, const kmi_config *cfg, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes,
                                             uint64_t buffer_offset, uint64_t nominal_bytes, int reaches_eof, int prev_byte, int *need_more,
                                             kmi_tuples *out) {
  if (!ctx || !cfg || !comm || !out || !need_more) return KMI_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  *need_more = 0;
  if (cfg->seq_format != KMI_FMT_FASTA) return set_err(ctx, KMI_ERR_INVALID, "not FASTA");
  if ((n_bytes && !bytes) || nominal_bytes > n_bytes) return set_err(ctx, KMI_ERR_INVALID, "bad buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *d_bytes;
  KMI_TRY(ws_get(ctx, WS_INPUT2, n_bytes + 64, &d_bytes));
  if (n_bytes) KMI_HIP(ctx, hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  kmi_fasta_partition part; uint64_t end = 0;
  KMI_TRY(fasta_range_partition(ctx, comm, cfg->k, bytes, (const uint8_t *)d_bytes, n_bytes, buffer_offset, nominal_bytes, reaches_eof, prev_byte,
                                need_more, &part, &end));
  if (*need_more || end == 0) return KMI_OK;
  KMI_TRY(kmi_ctx_set_fasta_partition(ctx, &part));
  const kmi_status st = kmi_extract_host(ctx, cfg, bytes, (size_t)end, buffer_offset, out);
  (void)kmi_ctx_set_fasta_partition(ctx, nullptr);
  return st;
}

kmi_status kmi_index_build_dist_host(kmi_index *idx, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t file_offset) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  if (kmi::comm_size(comm) == 1 && !ctx->force_dist) return kmi_index_build_host(idx, bytes, n_bytes, file_offset);
  if (n_bytes && !bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  void *d_bytes;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &d_bytes));
  if (n_bytes) KMI_HIP(ctx, hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  return kmi_index_build_dist_dev(idx, comm, (const uint8_t *)d_bytes, n_bytes, file_offset);
}

static kmi_status query_dist_host(kmi_index *idx, kmi_comm *comm, int mode, const uint64_t *queries, size_t nq, kmi_results *out, uint64_t *n_erased) {
  KMI_TRY(dist_check(idx, comm));
  kmi_ctx *ctx = idx->ctx;
  const int p = kmi::comm_size(comm);
  if (p == 1 && !ctx->force_dist) return query_host(idx, mode, queries, nq, out, n_erased);
  if (out) memset(out, 0, sizeof(*out));
  if (n_erased) *n_erased = 0;
  if (nq && !queries) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const uint32_t nw = idx->shape.n_words, ow = idx->val_words ? idx->val_words : 1u;
  const size_t kb = nw * sizeof(uint64_t), vb = ow * sizeof(uint64_t);
  void *d_in, *d_send, *d_q;
  KMI_TRY(ws_get(ctx, WS_INPUT, (nq + 8) * kb, &d_in));
  KMI_TRY(ws_get(ctx, WS_DIST_A, (nq + 8) * kb, &d_send));
  if (nq) KMI_HIP(ctx, hipMemcpyAsync(d_in, queries, nq * kb, hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint64_t> sc(p, 0), rc;
  if (idx->owner_lp) KMI_TRY(kmi_route_owner_dev(ctx, &idx->cfg, (const uint64_t *)d_in, nq, (uint32_t)p, (uint64_t *)d_send, sc.data()));
  else KMI_TRY(kmi_route_dev(ctx, &idx->cfg, (const uint64_t *)d_in, nq, (uint32_t)p, (uint64_t *)d_send, sc.data()));
  uint64_t total = 0;
  KMI_TRY(dist_exchange(comm, d_send, sc.data(), kb, WS_DIST_B, &d_q, rc, &total));
  if (mode == Q_ERASE) {
    uint64_t n = 0;
    if (total) KMI_TRY(index_query(idx, Q_ERASE, (const uint64_t *)d_q, (size_t)total, nullptr, nullptr, 0, &n));
    if (n_erased) *n_erased = n;
    return KMI_OK;
  }
  // every source rank's keys are answered on their own (a key two ranks ask about is answered to both). The answers of a multimap
  // find are sized by a count of the hits per source (a bound by the entries of the index, per source, is p times the index)
  const bool mm_find = idx->val_words > 0 && mode == Q_FIND;
  std::vector<uint64_t> back(p, 0), got;
  uint64_t bound = 0, off = 0;
  for (int s = 0; s < p; ++s) {
    if (mm_find) {
      if (rc[s]) KMI_TRY(index_query(idx, mode, (const uint64_t *)d_q + off * nw, (size_t)rc[s], nullptr, nullptr, 0, &back[s], nullptr, nullptr, true));
      bound += back[s];
    } else bound += query_result_bound(idx, mode, (size_t)rc[s]);
    off += rc[s];
  }
  void *d_rk, *d_rv;
  KMI_TRY(ws_get(ctx, WS_DIST_C, (bound + 8) * kb, &d_rk));
  KMI_TRY(ws_get(ctx, WS_DIST_D, (bound + 8) * vb, &d_rv));
  uint64_t pos = 0;
  off = 0;
  for (int s = 0; s < p; ++s) {
    uint64_t n = 0;
    if (rc[s] && (!mm_find || back[s]))
      KMI_TRY(index_query(idx, mode, (const uint64_t *)d_q + off * nw, (size_t)rc[s], (uint64_t *)d_rk + pos * nw, (uint64_t *)d_rv + pos * ow,
                          mm_find ? back[s] : query_result_bound(idx, mode, (size_t)rc[s]), &n));
    back[s] = n; off += rc[s]; pos += n;
  }
  void *d_ak, *d_av;
  uint64_t n_mine = 0, n_mine2 = 0;
  KMI_TRY(dist_exchange(comm, d_rk, back.data(), kb, WS_DIST_A, &d_ak, got, &n_mine));
  KMI_TRY(dist_exchange(comm, d_rv, back.data(), vb, WS_DIST_B, &d_av, got, &n_mine2));
  out->n = n_mine;
  out->keys = (uint64_t *)malloc((n_mine ? n_mine : 1) * kb);
  out->values = (uint64_t *)malloc((n_mine ? n_mine : 1) * vb);
  if (!out->keys || !out->values) return set_err(ctx, KMI_ERR_NOMEM, "host malloc failed");
  if (n_mine) {
    KMI_HIP(ctx, hipMemcpyAsync(out->keys, d_ak, n_mine * kb, hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(out->values, d_av, n_mine * vb, hipMemcpyDeviceToHost, ctx->stream));
  }
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

kmi_status kmi_index_count_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!out) return KMI_ERR_INVALID;
  return query_dist_host(idx, comm, Q_COUNT, queries, nq, out, nullptr);
}
kmi_status kmi_index_find_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!out) return KMI_ERR_INVALID;
  return query_dist_host(idx, comm, Q_FIND, queries, nq, out, nullptr);
}
kmi_status kmi_index_erase_dist_host(kmi_index *idx, kmi_comm *comm, const uint64_t *queries, size_t nq, uint64_t *n_erased_local) {
  return query_dist_host(idx, comm, Q_ERASE, queries, nq, nullptr, n_erased_local);
}
kmi_status kmi_index_size_dist(kmi_index *idx, kmi_comm *comm, uint64_t *n) {
  if (!n) return KMI_ERR_INVALID;
  KMI_TRY(dist_check(idx, comm));
  *n = idx->n_entries;
  return kmi::comm_allreduce_sum(comm, n);
}

// ---- de Bruijn graph nodes (kmi_debruijn.h)
kmi_status kmi_dbg_create(kmi_ctx *ctx, const kmi_config *cfg, uint32_t node_kind, kmi_dbg **out) {
  if (!ctx || !out) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (node_kind > KMI_DBG_EDGE_EXISTS) return set_err(ctx, KMI_ERR_INVALID, "unknown node kind");
  if (cfg->seq_filter != KMI_SEQ_ALL)
    return set_err(ctx, KMI_ERR_INVALID, "de Bruijn nodes are built from every record of the input (no sequence filter), as the reference's engine is");
  kmi_dbg *g = new kmi_dbg();
  g->ctx = ctx; g->cfg = *cfg; g->shape = shape; g->node_kind = node_kind;
  kmi_config c = *cfg;   // the node map proper: both strands of a k-mer are one node, kept under the smaller one
  c.index_kind = KMI_INDEX_COUNT; c.strand = KMI_STRAND_CANONICAL; c.dist_trans = KMI_DIST_MODEL;
  const kmi_status st = kmi_index_create(ctx, &c, &g->nodes);
  if (st != KMI_OK) { delete g; return st; }
  *out = g;
  return KMI_OK;
}

// the SeqParser template argument of the engine's build_posix / build_mmap (FASTQParser or FASTAParser)
kmi_status kmi_dbg_set_seq_format(kmi_dbg *g, uint32_t seq_format) {
  if (!g) return KMI_ERR_INVALID;
  if (seq_format > KMI_FMT_FASTA) return set_err(g->ctx, KMI_ERR_INVALID, "unknown sequence format");
  g->cfg.seq_format = seq_format;
  return KMI_OK;
}

kmi_status kmi_dbg_destroy(kmi_dbg *g) {
  if (!g) return KMI_OK;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  if (g->edges) pool_free(g->ctx, g->edges, g->edges_bytes);
  (void)kmi_index_destroy(g->nodes);
  delete g;
  return KMI_OK;
}

kmi_status kmi_dbg_clear(kmi_dbg *g) {
  if (!g) return KMI_ERR_INVALID;
  KMI_TRY(kmi_index_clear(g->nodes));
  if (g->edges) pool_free(g->ctx, g->edges, g->edges_bytes);
  g->edges = nullptr; g->edges_bytes = 0;
  return KMI_OK;
}

kmi_status kmi_dbg_local_size(kmi_dbg *g, uint64_t *n) {
  if (!g || !n) return KMI_ERR_INVALID;
  *n = g->nodes->n_entries;
  return KMI_OK;
}

kmi_status kmi_dbg_parse_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint64_t *out_records_dev,
                             size_t out_capacity, uint64_t *n_tuples) {
  if (!ctx || !n_tuples) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t *recs = nullptr;
  KMI_TRY(dbg_parse(ctx, cfg, bytes_dev, n_bytes, false, &recs, n_tuples));
  if (!out_records_dev) return KMI_OK;   // count only
  if (*n_tuples > out_capacity) return set_err(ctx, KMI_ERR_OVERFLOW, "parse: output capacity too small");
  if (*n_tuples) KMI_HIP(ctx, hipMemcpyAsync(out_records_dev, recs, (size_t)*n_tuples * (shape.n_words + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

kmi_status kmi_dbg_build_dev(kmi_dbg *g, const uint8_t *bytes_dev, size_t n_bytes) {
  if (!g) return KMI_ERR_INVALID;
  kmi_ctx *ctx = g->ctx;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_bytes == 0) return KMI_OK;
  {   // an empty graph from clean FASTQ reads: through super-k-mer records (dbg_build_superkmer); else, and for what that declines, tuples
    bool done = false;
    KMI_TRY(dbg_build_superkmer(g, bytes_dev, n_bytes, &done));
    if (done) return KMI_OK;
  }
  uint64_t *recs = nullptr, nt = 0;
  KMI_TRY(dbg_parse(ctx, &g->cfg, bytes_dev, n_bytes, true, &recs, &nt));
  return dbg_insert(g, recs, (size_t)nt);
}

kmi_status kmi_dbg_build_host(kmi_dbg *g, const uint8_t *bytes, size_t n_bytes) {
  if (!g) return KMI_ERR_INVALID;
  kmi_ctx *ctx = g->ctx;
  if (n_bytes == 0) return KMI_OK;
  if (!bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *din;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &din));
  KMI_HIP(ctx, hipMemcpyAsync(din, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  return kmi_dbg_build_dev(g, (const uint8_t *)din, n_bytes);
}

// tuples as the parser emits them (any strand; value word = edge byte); they are rewritten in place into node form
static kmi_status dbg_insert_tuples(kmi_dbg *g, uint64_t *recs_dev, size_t n) {
  KMI_TRY(dbg_edges(g->ctx, recs_dev, n, nullptr, 0, g->shape, false, true));
  return dbg_insert(g, recs_dev, n);
}

kmi_status kmi_dbg_insert_dev(kmi_dbg *g, const uint64_t *records_dev, size_t n) {
  if (!g) return KMI_ERR_INVALID;
  kmi_ctx *ctx = g->ctx;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return KMI_OK;
  void *dr;   // (the caller's buffer stays as it is)
  const size_t bytes = n * (g->shape.n_words + 1) * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, WS_DBG_RECS, bytes + 64, &dr));
  KMI_HIP(ctx, hipMemcpyAsync(dr, records_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return dbg_insert_tuples(g, (uint64_t *)dr, n);
}

kmi_status kmi_dbg_insert_host(kmi_dbg *g, const uint64_t *records, size_t n) {
  if (!g) return KMI_ERR_INVALID;
  kmi_ctx *ctx = g->ctx;
  if (n == 0) return KMI_OK;
  if (!records) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *dr;
  const size_t bytes = n * (g->shape.n_words + 1) * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, WS_DBG_RECS, bytes + 64, &dr));
  KMI_HIP(ctx, hipMemcpyAsync(dr, records, bytes, hipMemcpyHostToDevice, ctx->stream));
  return dbg_insert_tuples(g, (uint64_t *)dr, n);
}

kmi_status kmi_dbg_find_dev(kmi_dbg *g, const uint64_t *queries_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_values_dev, uint64_t *n_out) {
  if (!g || !n_out) return KMI_ERR_INVALID;
  KMI_HIP(g->ctx, hipSetDevice(g->ctx->device));
  return dbg_find(g, queries_dev, nq, out_keys_dev, out_values_dev, n_out);
}

kmi_status kmi_dbg_find_host(kmi_dbg *g, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!g || !out) return KMI_ERR_INVALID;
  kmi_ctx *ctx = g->ctx;
  memset(out, 0, sizeof(*out));
  if (nq == 0) return KMI_OK;
  if (!queries) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t nw = g->shape.n_words;
  void *dq, *dk, *dv;
  KMI_TRY(ws_get(ctx, WS_INPUT, nq * nw * sizeof(uint64_t), &dq));
  KMI_TRY(ws_get(ctx, WS_OUTPUT, nq * nw * sizeof(uint64_t), &dk));
  KMI_TRY(ws_get(ctx, WS_OUTPUT2, nq * kDbgValueWords * sizeof(uint64_t), &dv));
  KMI_HIP(ctx, hipMemcpyAsync(dq, queries, nq * nw * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  uint64_t n = 0;
  KMI_TRY(dbg_find(g, (const uint64_t *)dq, nq, (uint64_t *)dk, (uint64_t *)dv, &n));
  out->n = n;
  out->keys = (uint64_t *)malloc((n ? n : 1) * nw * sizeof(uint64_t));
  out->values = (uint64_t *)malloc((n ? n : 1) * kDbgValueWords * sizeof(uint64_t));
  if (!out->keys || !out->values) return set_err(ctx, KMI_ERR_NOMEM, "host malloc failed");
  if (n) {
    KMI_HIP(ctx, hipMemcpyAsync(out->keys, dk, n * nw * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(out->values, dv, n * kDbgValueWords * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return KMI_OK;
}

kmi_status kmi_dbg_count_host(kmi_dbg *g, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!g) return KMI_ERR_INVALID;
  return kmi_index_count_host(g->nodes, queries, nq, out);
}

kmi_status kmi_dbg_export_host(kmi_dbg *g, uint64_t *keys, uint32_t *counts9, size_t capacity, uint64_t *n) {
  if (!g || !n) return KMI_ERR_INVALID;
  kmi_ctx *ctx = g->ctx;
  *n = 0;
  const uint64_t ne = g->nodes->n_entries;
  if (ne == 0) return KMI_OK;
  if (capacity < ne) return set_err(ctx, KMI_ERR_OVERFLOW, "export: capacity too small");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (keys) KMI_HIP(ctx, hipMemcpyAsync(keys, g->nodes->keys, ne * g->shape.n_words * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (counts9) {
    std::vector<uint32_t> e(ne * 8), s(ne);
    KMI_HIP(ctx, hipMemcpyAsync(e.data(), g->edges, ne * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(s.data(), g->nodes->vals, ne * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const bool ex = g->node_kind == KMI_DBG_EDGE_EXISTS;
    for (uint64_t i = 0; i < ne; ++i) {
      for (int t = 0; t < 8; ++t) counts9[i * 9 + t] = ex ? (e[i * 8 + t] ? 1u : 0u) : e[i * 8 + t];
      counts9[i * 9 + 8] = ex ? 0u : s[i];
    }
  }
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n = ne;
  return KMI_OK;
}

// build over ranks: parse the rank's share, group the node-form tuples by KeyToRank of the canonical k-mer, one all-to-all
// (imxx::distribute inside de_bruijn_nodes_distributed::insert, de_bruijn_nodes_distributed.hpp:243-250), insert what arrives
kmi_status kmi_dbg_build_dist_host(kmi_dbg *g, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes) {
  if (!g) return KMI_ERR_INVALID;
  KMI_TRY(dist_check(g->nodes, comm));
  kmi_ctx *ctx = g->ctx;
  const int p = kmi::comm_size(comm);
  if (p == 1 && !ctx->force_dist) return kmi_dbg_build_host(g, bytes, n_bytes);
  if (n_bytes && !bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const uint32_t rw = g->shape.n_words + 1u;
  void *d_bytes, *d_send, *d_recv;
  KMI_TRY(ws_get(ctx, WS_INPUT, n_bytes + 64, &d_bytes));
  if (n_bytes) KMI_HIP(ctx, hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
  uint64_t *recs = nullptr, nt = 0, total = 0;
  KMI_TRY(dbg_parse(ctx, &g->cfg, (const uint8_t *)d_bytes, n_bytes, true, &recs, &nt));   // (an empty share still enters the collectives)
  KMI_TRY(ws_get(ctx, WS_DIST_A, ((size_t)nt + 64) * rw * sizeof(uint64_t), &d_send));
  std::vector<uint64_t> sc(p, 0), rc;
  if (nt) KMI_TRY(kmi_route_tuples_dev(ctx, &g->nodes->cfg, recs, (size_t)nt, (uint32_t)p, 1, (uint64_t *)d_send, sc.data()));
  KMI_TRY(dist_exchange(comm, d_send, sc.data(), rw * sizeof(uint64_t), WS_DIST_B, &d_recv, rc, &total));
  return dbg_insert(g, (const uint64_t *)d_recv, (size_t)total);
}

// find() of the node map over ranks: query keys to the ranks that own their canonical k-mer (the node map's KeyToRank), every
// source's keys answered on their own, one return exchange of (k-mer, node) -- distributed_unordered_map.hpp:564-687 as the
// node map inherits it (de_bruijn_nodes_distributed.hpp:61)
kmi_status kmi_dbg_find_dist_host(kmi_dbg *g, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!g || !out) return KMI_ERR_INVALID;
  KMI_TRY(dist_check(g->nodes, comm));
  kmi_ctx *ctx = g->ctx;
  const int p = kmi::comm_size(comm);
  if (p == 1 && !ctx->force_dist) return kmi_dbg_find_host(g, queries, nq, out);
  memset(out, 0, sizeof(*out));
  if (nq && !queries) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const uint32_t nw = g->shape.n_words;
  const size_t kb = nw * sizeof(uint64_t), vb = kDbgValueWords * sizeof(uint64_t);
  void *d_in, *d_send, *d_q;
  KMI_TRY(ws_get(ctx, WS_INPUT, (nq + 8) * kb, &d_in));
  KMI_TRY(ws_get(ctx, WS_DIST_A, (nq + 8) * kb, &d_send));
  if (nq) KMI_HIP(ctx, hipMemcpyAsync(d_in, queries, nq * kb, hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint64_t> sc(p, 0), rc;
  KMI_TRY(kmi_route_dev(ctx, &g->nodes->cfg, (const uint64_t *)d_in, nq, (uint32_t)p, (uint64_t *)d_send, sc.data()));
  uint64_t total = 0;
  KMI_TRY(dist_exchange(comm, d_send, sc.data(), kb, WS_DIST_B, &d_q, rc, &total));
  void *d_rk, *d_rv;
  KMI_TRY(ws_get(ctx, WS_DIST_C, (total + 8) * kb, &d_rk));
  KMI_TRY(ws_get(ctx, WS_DIST_D, (total + 8) * vb, &d_rv));
  std::vector<uint64_t> back(p, 0), got;
  uint64_t off = 0, pos = 0;
  for (int s = 0; s < p; ++s) {
    uint64_t n = 0;
    if (rc[s]) KMI_TRY(dbg_find(g, (const uint64_t *)d_q + off * nw, (size_t)rc[s], (uint64_t *)d_rk + pos * nw, (uint64_t *)d_rv + pos * kDbgValueWords, &n));
    back[s] = n; off += rc[s]; pos += n;
  }
  void *d_ak, *d_av;
  uint64_t n_mine = 0, n_mine2 = 0;
  KMI_TRY(dist_exchange(comm, d_rk, back.data(), kb, WS_DIST_A, &d_ak, got, &n_mine));
  KMI_TRY(dist_exchange(comm, d_rv, back.data(), vb, WS_DIST_B, &d_av, got, &n_mine2));
  out->n = n_mine;
  out->keys = (uint64_t *)malloc((n_mine ? n_mine : 1) * kb);
  out->values = (uint64_t *)malloc((n_mine ? n_mine : 1) * vb);
  if (!out->keys || !out->values) return set_err(ctx, KMI_ERR_NOMEM, "host malloc failed");
  if (n_mine) {
    KMI_HIP(ctx, hipMemcpyAsync(out->keys, d_ak, n_mine * kb, hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(out->values, d_av, n_mine * vb, hipMemcpyDeviceToHost, ctx->stream));
  }
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

// build_posix / build_mmap(filename) of the engine with comm.size() > 1: the rank passes the bytes it read of the FASTQ file (its
// nominal range + look-ahead), the partition is cut at record starts on the device (kmi_index_build_range_dist_host's rule)
kmi_status kmi_dbg_build_range_dist_host(kmi_dbg *g, kmi_comm *comm, const uint8_t *bytes, size_t n_bytes, uint64_t buffer_offset,
                                         uint64_t nominal_bytes, int reaches_eof, int *need_more) {
  if (!g || !need_more) return KMI_ERR_INVALID;
  KMI_TRY(dist_check(g->nodes, comm));
  kmi_ctx *ctx = g->ctx;
  *need_more = 0;
  if (n_bytes && !bytes) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  if (nominal_bytes > n_bytes) nominal_bytes = n_bytes;
  uint64_t pos[2] = {0, nominal_bytes}, cut[2] = {0, n_bytes};
  if (n_bytes) {
    KMI_HIP(ctx, hipSetDevice(ctx->device));
    void *d_bytes;
    KMI_TRY(ws_get(ctx, WS_INPUT2, n_bytes + 64, &d_bytes));
    KMI_HIP(ctx, hipMemcpyAsync(d_bytes, bytes, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    KMI_TRY(kmi_fastq_find_records_dev(ctx, (const uint8_t *)d_bytes, n_bytes, buffer_offset == 0, pos, 2, cut));
    if (nominal_bytes >= n_bytes) cut[1] = n_bytes;
  }
  if (cut[1] >= n_bytes && !reaches_eof && n_bytes) { *need_more = 1; return KMI_OK; }
  if (cut[0] >= n_bytes && !reaches_eof && n_bytes && buffer_offset != 0) { *need_more = 1; return KMI_OK; }
  if (cut[1] < cut[0]) cut[1] = cut[0];
  return kmi_dbg_build_dist_host(g, comm, bytes + cut[0], (size_t)(cut[1] - cut[0]));
}

// erase(): the nodes of the query keys (either strand) leave the map
kmi_status kmi_dbg_erase_host(kmi_dbg *g, const uint64_t *queries, size_t nq, uint64_t *n_erased) {
  if (!g) return KMI_ERR_INVALID;
  kmi_ctx *ctx = g->ctx;
  if (n_erased) *n_erased = 0;
  if (nq == 0) return KMI_OK;
  if (!queries) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *dq;
  KMI_TRY(ws_get(ctx, WS_INPUT, (nq + 8) * g->shape.n_words * sizeof(uint64_t), &dq));
  KMI_HIP(ctx, hipMemcpyAsync(dq, queries, nq * g->shape.n_words * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  return dbg_erase(g, (const uint64_t *)dq, nq, n_erased);
}

// ... over ranks: the keys travel to the ranks that own them (*n_erased_local = nodes that left THIS rank's part)
kmi_status kmi_dbg_erase_dist_host(kmi_dbg *g, kmi_comm *comm, const uint64_t *queries, size_t nq, uint64_t *n_erased_local) {
  if (!g) return KMI_ERR_INVALID;
  KMI_TRY(dist_check(g->nodes, comm));
  kmi_ctx *ctx = g->ctx;
  const int p = kmi::comm_size(comm);
  if (p == 1 && !ctx->force_dist) return kmi_dbg_erase_host(g, queries, nq, n_erased_local);
  if (n_erased_local) *n_erased_local = 0;
  if (nq && !queries) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  const size_t kb = g->shape.n_words * sizeof(uint64_t);
  void *d_in, *d_send, *d_q;
  KMI_TRY(ws_get(ctx, WS_INPUT, (nq + 8) * kb, &d_in));
  KMI_TRY(ws_get(ctx, WS_DIST_A, (nq + 8) * kb, &d_send));
  if (nq) KMI_HIP(ctx, hipMemcpyAsync(d_in, queries, nq * kb, hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint64_t> sc(p, 0), rc;
  KMI_TRY(kmi_route_dev(ctx, &g->nodes->cfg, (const uint64_t *)d_in, nq, (uint32_t)p, (uint64_t *)d_send, sc.data()));
  uint64_t total = 0;
  KMI_TRY(dist_exchange(comm, d_send, sc.data(), kb, WS_DIST_B, &d_q, rc, &total));
  return dbg_erase(g, (const uint64_t *)d_q, (size_t)total, n_erased_local);
}

// count() over ranks: the node map's keys live in a count index, whose collective answers 1 per node held
kmi_status kmi_dbg_count_dist_host(kmi_dbg *g, kmi_comm *comm, const uint64_t *queries, size_t nq, kmi_results *out) {
  if (!g) return KMI_ERR_INVALID;
  return kmi_index_count_dist_host(g->nodes, comm, queries, nq, out);
}

kmi_status kmi_dbg_size_dist(kmi_dbg *g, kmi_comm *comm, uint64_t *n) {
  if (!g) return KMI_ERR_INVALID;
  return kmi_index_size_dist(g->nodes, comm, n);
}

// ---- builds over ranks through exchanged super-k-mer records
kmi_status kmi_index_sk_produce_dev(kmi_index *idx, const uint8_t *bytes_dev, size_t n_bytes, uint32_t nranks, uint64_t *out_records_dev,
                                    size_t out_capacity, const uint64_t **records_dev, uint64_t *n_records, uint64_t *send_counts_host, int *produced) {
  if (!idx || !records_dev || !n_records || !send_counts_host || !produced) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  *produced = 0; *records_dev = nullptr; *n_records = 0;
  const uint32_t w = sk_width_of(idx);
  if (!w || !sk_rank_count(nranks)) return KMI_OK;   // not a case of this path: the caller routes k-mers
  if (ctx->sk_dbg == 7) return KMI_OK;   // (test knob: as if the input had exceeded a capacity of the front end)
  if (n_bytes) KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  if (!out_records_dev) out_capacity = 0;
  return sk_produce(idx, w, bytes_dev, n_bytes, nranks, records_dev, n_records, send_counts_host, produced, out_records_dev, out_capacity);
}

kmi_status kmi_index_sk_consume_dev(kmi_index *idx, const uint64_t *records_dev, size_t n_records, uint32_t nranks) {
  if (!idx) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t w = sk_width_of(idx);
  if (!w || !sk_rank_count(nranks)) return set_err(ctx, KMI_ERR_INVALID, "no super-k-mer build for this index / rank count");
  if (n_records && !records_dev) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  return sk_consume(idx, w, records_dev, n_records, nranks);
}

kmi_status kmi_index_set_owner_ranks(kmi_index *idx, uint32_t nranks) {
  if (!idx) return KMI_ERR_INVALID;
  if (nranks == 0 || nranks > 8 || (nranks & (nranks - 1u))) return set_err(idx->ctx, KMI_ERR_INVALID, "owner ranks: 1, 2, 4 or 8");
  if (idx->has_data && idx->n_entries && (1u << idx->owner_lp) != nranks)
    return set_err(idx->ctx, KMI_ERR_INVALID, "the index holds entries distributed another way");
  idx->owner_lp = 31u - (uint32_t)__builtin_clz(nranks);
  return KMI_OK;
}

kmi_status kmi_index_set_saturating(kmi_index *idx, int on) {
  if (!idx) return KMI_ERR_INVALID;
  if (idx->val_words) return set_err(idx->ctx, KMI_ERR_INVALID, "saturating counts belong to the counting maps");
  idx->saturating = on != 0;
  return KMI_OK;
}

kmi_status kmi_index_sk_width(kmi_index *idx, uint32_t *w) {
  if (!idx || !w) return KMI_ERR_INVALID;
  *w = kmi::sk_width_of(idx);
  return KMI_OK;
}

kmi_status kmi_index_owner_ranks(kmi_index *idx, uint32_t *nranks) {
  if (!idx || !nranks) return KMI_ERR_INVALID;
  *nranks = 1u << idx->owner_lp;
  return KMI_OK;
}

kmi_status kmi_route_owner_dev(kmi_ctx *ctx, const kmi_config *cfg, const uint64_t *keys_dev, size_t n, uint32_t nranks, uint64_t *out_keys_dev,
                               uint64_t *send_counts_host) {
  if (!ctx) return KMI_ERR_INVALID;
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (shape.n_words != 1 || shape.bits != 2 || sk_window_of(shape.k) == 0 || nranks < 2 || nranks > 8 || (nranks & (nranks - 1u)) || !send_counts_host)
    return set_err(ctx, KMI_ERR_INVALID, "owner routing is for one-word DNA k-mers over 2, 4 or 8 ranks");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  for (uint32_t r = 0; r < nranks; ++r) send_counts_host[r] = 0;
  if (n == 0) return KMI_OK;
  return route_owner(ctx, cfg, shape, keys_dev, n, nranks, out_keys_dev, send_counts_host);
}

// ---- update() with a device-side updater (kmi_update.h)
kmi_status kmi_index_update_pairs_dev(kmi_index *idx, const uint64_t *records_dev, size_t n, uint32_t op, uint64_t *n_updated) {
  if (!idx || !n_updated) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  *n_updated = 0;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "update() is a member of the counting maps");
  if (op > KMI_UPDATE_ASSIGN) return set_err(ctx, KMI_ERR_INVALID, "unknown updater");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return KMI_OK;
  void *dr;   // (the caller's buffer stays as it is)
  const size_t bytes = n * (idx->shape.n_words + 1) * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, WS_INPUT2, bytes + 64, &dr));
  KMI_HIP(ctx, hipMemcpyAsync(dr, records_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return index_update_pairs(idx, (uint64_t *)dr, n, (int)op, n_updated);
}

kmi_status kmi_index_update_pairs_host(kmi_index *idx, const uint64_t *records, size_t n, uint32_t op, uint64_t *n_updated) {
  if (!idx || !n_updated) return KMI_ERR_INVALID;
  kmi_ctx *ctx = idx->ctx;
  *n_updated = 0;
  if (idx->val_words) return set_err(ctx, KMI_ERR_INVALID, "update() is a member of the counting maps");
  if (op > KMI_UPDATE_ASSIGN) return set_err(ctx, KMI_ERR_INVALID, "unknown updater");
  if (n == 0) return KMI_OK;
  if (!records) return set_err(ctx, KMI_ERR_INVALID, "null buffer");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  void *dr;
  const size_t bytes = n * (idx->shape.n_words + 1) * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, WS_INPUT2, bytes + 64, &dr));
  KMI_HIP(ctx, hipMemcpyAsync(dr, records, bytes, hipMemcpyHostToDevice, ctx->stream));
  return index_update_pairs(idx, (uint64_t *)dr, n, (int)op, n_updated);
}

}  // extern "C"
