// kmi_superkmer.h -- the fused count-index build of one-word 2-bit k-mers through SUPER-K-MERS (included by kmi_index.hip).
//
// What it replaces on the reference side is unchanged (Index::build_* = read_file + insert,
// kmer_index.hpp:239-372 -> kmer_file_helper.hpp:588-633 + distributed_unordered_map.hpp:1603-1618, 1826-1884); what
// changes is what travels through HBM between the parse and the per-bucket reduce. The first fused build moved every
// k-mer as an 8-byte key through the partition three times (32 of its 43 bytes per k-mer). Here the unit that is
// partitioned is the super-k-mer: the run of consecutive k-mers of a read that share their MINIMIZER (the canonical
// m-mer, m = k - W + 1, with the smallest hash among the W m-mers of the k-mer). All occurrences of a canonical k-mer --
// on either strand, in any read -- have the same minimizer, so a bucket chosen by a hash of the minimizer holds every
// copy of its k-mers and can be counted on its own; and a super-k-mer of n k-mers is k + n - 1 packed bases, one
// 16-byte record for about (W + 1) / 2 k-mers instead of 8 bytes for each.
//
//   L   fastq_list_kernel<RUNS>   runs of k-mer windows per read and tile (32-bit entries)
//   M   sk_minimizer_kernel<W>    one LANE walks one run: rolling canonical m-mer hash, sliding-window minimum over W
//                                 positions (prefix / suffix minima over blocks of W: static register indices), super-k-mer
//                                 boundaries -> items (window offset, length, 18 bucket bits) + per-workgroup coarse counts
//   S   sk_scatter_kernel         items -> 16-byte records, bucket-sorted per round in LDS (8-byte descriptors), assembled
//                                 from the stream image on the way out -> 256 coarse buckets at per-workgroup cursors
//   P   sk_fine_count / scatter_fine (record mode)   coarse bucket -> its 128 fine buckets
//   C   sk_reduce_kernel          one workgroup per fine bucket: a lane expands a record (rolling both strands), canonical
//                                 keys are compacted through a per-wavefront LDS queue and go through the flat table insert
//                                 64 at a time; distinct (k-mer, count) pairs leave through a global cursor
// The pairs are distinct; they enter the index through the weighted-pair insert (placement-hash partition), so the
// stored index, the queries and every other path are what they were.
#pragma once

namespace kmi {

// record (16 bytes): word 0 = bases 0..31 of the complement-stream slice, word 1 = bases 32..50 (38 bits) | (n - 1) << 38 |
// bucket bits << 43 (18 bits: coarse 8 | fine 7 | sub 3, most significant first)
constexpr int kRecNShift = 38, kRecHashShift = 43;
__device__ __forceinline__ uint32_t rec_hash18(uint64_t w1) { return (uint32_t)(w1 >> kRecHashShift) & 0x3ffffu; }
__device__ __forceinline__ uint32_t rec_fine_sub(uint64_t w1) { return (rec_hash18(w1) >> 3) & 127u; }   // fine bucket inside its coarse bucket

constexpr int kSkThreads = 512;      // workgroup of the minimizer and scatter passes: one lane per run
constexpr int kSkRoundTiles = 24;    // scan tiles a round may span (512 reads of 150 bases are 20 tiles)
constexpr int kSkListCap = 32;       // items per run entry (about 13: 25 come up once in 1e5 entries; see sk_segment_of)

// a round of the minimizer / scatter passes: up to kSkThreads consecutive runs of the workgroup's tiles (at most
// kSkRoundTiles scan tiles: the tile of a lane's run is found by comparing against that many prefix counts)
struct SkRound {
  uint64_t t; uint32_t ei, total; uint32_t eo[kSkRoundTiles + 1]; uint64_t nt; uint32_t nei;
};
__device__ __forceinline__ bool sk_plan(const uint32_t *__restrict__ ent_cnt, uint64_t te, uint64_t t, uint32_t ei, SkRound &r) {
  while (t < te && ei >= ent_cnt[t]) { ++t; ei = 0; }   // tiles that are used up or hold no window are skipped
  if (t >= te) return false;
  r.t = t; r.ei = ei; r.eo[0] = 0;
  r.nt = t + kSkRoundTiles; r.nei = 0;
  bool cut = false;
#pragma unroll
  for (int i = 0; i < kSkRoundTiles; ++i) {
    uint32_t avail = (t + i < te) ? ent_cnt[t + i] : 0u;
    if (i == 0) avail -= ei;
    uint32_t take = avail;
    if (r.eo[i] + take > (uint32_t)kSkThreads) take = (uint32_t)kSkThreads - r.eo[i];
    if (cut) take = 0;
    if (!cut && take < avail) { cut = true; r.nt = t + i; r.nei = (i == 0 ? ei : 0u) + take; }
    r.eo[i + 1] = r.eo[i] + take;
  }
  r.total = r.eo[kSkRoundTiles];
  return true;
}
// the run of lane `e` of the round: tile (relative to the round's first), index into the run list
__device__ __forceinline__ void sk_locate(const SkRound &r, uint32_t e, uint32_t ent_stride, uint32_t &tr, uint64_t &idx) {
  tr = 0;
#pragma unroll
  for (int i = 1; i < kSkRoundTiles; ++i) tr += (e >= r.eo[i]) ? 1u : 0u;
  uint32_t base = 0;
#pragma unroll
  for (int i = 1; i < kSkRoundTiles; ++i) base = (tr == (uint32_t)i) ? r.eo[i] : base;
  idx = (r.t + tr) * ent_stride + (tr == 0u ? r.ei : 0u) + (e - base);
}
// The packed stream is read where it lies (every lane walks its own read: 300 bits about 80 bytes apart from its
// neighbour's; the lines are served by L2): three dwords at base position ip hold 64 aligned stream bits and more.
struct SkWin { uint32_t r0, r1, r2, sh; };
__device__ __forceinline__ SkWin sk_fetch(const uint32_t *__restrict__ st, uint64_t last_dw, uint64_t ip) {
  uint64_t d = ip >> 4;
  d = d < last_dw ? d : last_dw;   // (clamped: positions past the data belong to no window)
  SkWin w;
  w.r0 = st[d]; w.r1 = st[d + 1]; w.r2 = st[d + 2]; w.sh = (uint32_t)(ip & 15u) * 2u;
  return w;
}
__device__ __forceinline__ uint32_t sk_lo(const SkWin &w) { return __builtin_amdgcn_alignbit(w.r1, w.r0, w.sh); }
__device__ __forceinline__ uint32_t sk_hi(const SkWin &w) { return __builtin_amdgcn_alignbit(w.r2, w.r1, w.sh); }

// ---------------------------------------------------------------------------
// M: minimizers and super-k-mer boundaries, one lane per run
// ---------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(kSkThreads, 2) void sk_minimizer_kernel(PackedInput in, uint64_t n_tiles, uint32_t k, const uint32_t *__restrict__ ent,
                                                                    const uint32_t *__restrict__ ent_cnt, uint32_t ent_stride, uint32_t items_per_tile,
                                                                    uint32_t *__restrict__ items, uint32_t *__restrict__ run_items,
                                                                    uint32_t *__restrict__ wg_hist, uint32_t *__restrict__ flags) {
  constexpr int NT = kSkThreads, CAP = kSkListCap;
  constexpr uint32_t INF = 0xffffffffu;
  __shared__ uint32_t s_list[(CAP + 2) * NT];   // [j][thread]; slot 0 takes the opening dummy, slot CAP + 1 what does not fit
  __shared__ uint32_t s_cnt[kNumCoarse];
  __shared__ uint32_t s_scan[NT / kWave + 2];
  const uint32_t m = k - (uint32_t)W + 1u;
  const uint32_t mmask = (m >= 16u) ? 0xffffffffu : ((1u << (2u * m)) - 1u);
  const uint32_t topsh = 2u * m - 2u;
  const uint32_t nmax = sk_nmax_of(k);
  const uint32_t *st = reinterpret_cast<const uint32_t *>(in.stream);
  const uint64_t last_dw = in.n_cover / 16 - 1;   // (the stream buffer has 64 bytes of slack behind the covered tiles)
  if (threadIdx.x < kNumCoarse) s_cnt[threadIdx.x] = 0;
  const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t tb = (uint64_t)blockIdx.x * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  const uint64_t item_base = tb * (uint64_t)items_per_tile;                 // this workgroup's item stream
  const uint64_t item_end = te * (uint64_t)items_per_tile;
  uint32_t round_base = 0;
  SkRound cur;
  bool have = tb < te && sk_plan(ent_cnt, te, tb, 0u, cur);
  while (have) {
    uint32_t tr = 0, ev = 0;
    uint64_t eidx = 0;
    const bool mine = threadIdx.x < cur.total;
    if (mine) { sk_locate(cur, threadIdx.x, ent_stride, tr, eidx); ev = ent[eidx]; }
    uint32_t cnt = 0;   // list slots written by this lane's run, the opening dummy included
    {
      const uint32_t L = mine ? (ev >> 13) + 1u : 0u;                      // windows of the run
      const uint64_t ip0 = (cur.t + tr) * 8192ull + (ev & 0x1fffu);        // stream position of its first base
      const uint32_t nblk = mine ? (L + (uint32_t)W - 2u) / (uint32_t)W + 1u : 0u;   // m-mer positions 0 .. L + W - 2
      uint32_t R, F;
      {
        const SkWin w0 = sk_fetch(st, last_dw, ip0);
        R = sk_lo(w0) & mmask;
        F = sk_fwd_of(R, m);
      }
      uint32_t sprev[W + 1];
#pragma unroll
      for (int j = 0; j <= W; ++j) sprev[j] = INF;
      // The first window opens a super-k-mer like any other boundary: len starts at nmax ("the one before is full"), and
      // the dummy this closes lands in list slot 0.
      uint32_t prev = 0, len = nmax;
      // codes of the bases b W + m - 1 .. b W + m + W - 2 (the base that completes m-mer position q = b W + j is q + m - 1);
      // the next block's are in flight while this one is walked
      SkWin wn = sk_fetch(st, last_dw, ip0 + m - 1u);
      for (uint32_t b = 0; __any(b < nblk); ++b) {
        const uint32_t clo = sk_lo(wn), chi = sk_hi(wn);
        wn = sk_fetch(st, last_dw, ip0 + (uint64_t)(b + 1u) * (uint32_t)W + m - 1u);
        uint32_t hh[W];
        uint32_t p = INF;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const uint32_t q = b * (uint32_t)W + (uint32_t)j;   // (the same in every lane)
          if (j > 0 || b > 0) {
            const uint32_t c = (j < 16) ? ((clo >> (2 * (j & 15))) & 3u) : ((chi >> (2 * (j & 15))) & 3u);
            R = (R >> 2) | (c << topsh);
            F = ((F << 2) | (c ^ 3u)) & mmask;
          }
          // (positions past the run's last m-mer hash whatever follows the read: only windows that do not exist see them)
          const uint32_t h = sk_order_hash(R < F ? R : F);
          hh[j] = h;
          p = p < h ? p : h;
          const uint32_t sp = sprev[j + 1];
          const uint32_t curv = sp < p ? sp : p;
          const bool valid = q - (uint32_t)(W - 1) < L;   // window i = q - (W - 1); wraps to a large number below zero
          const bool fresh = valid && (curv != prev || len >= nmax);
          if (fresh) {   // close (prev, len)
            const uint32_t slot = cnt < (uint32_t)CAP + 1u ? cnt : (uint32_t)CAP + 1u;
            s_list[slot * NT + threadIdx.x] = (prev << 5) | (len - 1u);
            ++cnt;
            prev = curv;
            len = 0u;
          }
          len += valid ? 1u : 0u;
        }
        sprev[W] = INF;
        sprev[W - 1] = hh[W - 1];
#pragma unroll
        for (int j = W - 2; j >= 0; --j) sprev[j] = hh[j] < sprev[j + 1] ? hh[j] : sprev[j + 1];
      }
      if (mine) {   // the last super-k-mer
        const uint32_t slot = cnt < (uint32_t)CAP + 1u ? cnt : (uint32_t)CAP + 1u;
        s_list[slot * NT + threadIdx.x] = (prev << 5) | (len - 1u);
        ++cnt;
      }
    }
    if (cnt > (uint32_t)CAP + 1u) { atomicOr(&flags[9], 1u); cnt = 1; }
    cnt = cnt ? cnt - 1u : 0u;   // real items: slots 1 .. cnt
    // items in their final form (window offset | (n - 1) << 7 | bucket bits << 12) + the coarse counts
    {
      uint32_t off = 0;
      for (uint32_t j = 1; j <= cnt; ++j) {
        const uint32_t it = s_list[j * NT + threadIdx.x];
        const uint32_t n1 = it & 31u;
        const uint32_t h18 = sk_bucket_bits(it >> 5);
        atomicAdd(&s_cnt[h18 >> 10], 1u);
        s_list[j * NT + threadIdx.x] = off | (n1 << 7) | (h18 << 12);
        off += n1 + 1u;
      }
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>(cnt, s_scan, &total);
    if (item_base + round_base + total > item_end) {   // uniform
      if (threadIdx.x == 0) atomicOr(&flags[9], 2u);
    } else {
      uint32_t *dst = items + item_base + round_base + ex;
      for (uint32_t j = 0; j < cnt; ++j) dst[j] = s_list[(j + 1u) * NT + threadIdx.x];
      if (mine) run_items[eidx] = (round_base + ex) | (cnt << 26);
    }
    round_base += total;
    SkRound nxt;
    have = sk_plan(ent_cnt, te, cur.nt, cur.nei, nxt);
    cur = nxt;
  }
  lds_barrier();
  if (threadIdx.x < kNumCoarse) wg_hist[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] = s_cnt[threadIdx.x];
}

// ---------------------------------------------------------------------------
// S: items -> records -> coarse buckets
// ---------------------------------------------------------------------------
// record of the super-k-mer whose first base is stream position ip: nb = k + n - 1 bases
__device__ __forceinline__ void sk_assemble(const uint32_t *__restrict__ st, uint64_t last_dw, uint64_t ip, uint32_t nb, uint32_t n1, uint32_t h18,
                                            uint64_t &w0, uint64_t &w1) {
  uint64_t d = ip >> 4;
  d = d < last_dw ? d : last_dw;
  const uint32_t sh = (uint32_t)(ip & 15u) * 2u;
  const uint32_t r0 = st[d], r1 = st[d + 1], r2 = st[d + 2], r3 = st[d + 3], r4 = st[d + 4];
  const uint32_t a0 = __builtin_amdgcn_alignbit(r1, r0, sh), a1 = __builtin_amdgcn_alignbit(r2, r1, sh);
  const uint32_t a2 = __builtin_amdgcn_alignbit(r3, r2, sh), a3 = __builtin_amdgcn_alignbit(r4, r3, sh);
  w0 = (uint64_t)a0 | ((uint64_t)a1 << 32);
  uint64_t hi = (uint64_t)a2 | ((uint64_t)a3 << 32);
  const uint32_t bits = 2u * nb;   // 34 .. 102
  if (bits < 64u) { w0 &= (1ull << bits) - 1ull; hi = 0; }
  else hi &= (1ull << (bits - 64u)) - 1ull;
  w1 = hi | ((uint64_t)n1 << kRecNShift) | ((uint64_t)h18 << kRecHashShift);
}

__global__ __launch_bounds__(kSkThreads) void sk_scatter_kernel(PackedInput in, uint64_t n_tiles, uint32_t k, const uint32_t *__restrict__ ent,
                                                               const uint32_t *__restrict__ ent_cnt, uint32_t ent_stride, uint32_t items_per_tile,
                                                               const uint32_t *__restrict__ items, const uint32_t *__restrict__ run_items,
                                                               const uint64_t *__restrict__ wg_off, uint64_t *__restrict__ out) {
  constexpr int NT = kSkThreads, CAP = kSkListCap;
  __shared__ uint64_t s_stage[CAP * NT];   // descriptors: stream position (41 bits) | (n - 1) << 41 | bucket bits << 46
  __shared__ uint32_t s_cnt[kNumCoarse];
  __shared__ uint32_t s_lofs[kNumCoarse];
  __shared__ uint64_t s_gbase[kNumCoarse];
  __shared__ uint32_t s_part[kNumCoarse / kWave];
  __shared__ uint32_t s_total;
  const uint32_t *st = reinterpret_cast<const uint32_t *>(in.stream);
  const uint64_t last_dw = in.n_cover / 16 - 1;   // (the stream buffer has 64 bytes of slack behind the covered tiles)
  uint64_t cursor = (threadIdx.x < kNumCoarse) ? wg_off[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] : 0ull;
  if (threadIdx.x < kNumCoarse) s_cnt[threadIdx.x] = 0;
  const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t tb = (uint64_t)blockIdx.x * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  const uint64_t item_base = tb * (uint64_t)items_per_tile;
  SkRound cur;
  bool have = tb < te && sk_plan(ent_cnt, te, tb, 0u, cur);
  while (have) {
    lds_barrier();   // the previous round's stage is done with; the counters are clear
    uint32_t tr = 0, ev = 0, cnt = 0;
    uint32_t it[CAP], rk[CAP];
    {
      uint64_t eidx = 0;
      uint32_t ri = 0;
      if (threadIdx.x < cur.total) { sk_locate(cur, threadIdx.x, ent_stride, tr, eidx); ev = ent[eidx]; ri = run_items[eidx]; }
      cnt = ri >> 26;
      const uint32_t *src = items + item_base + (ri & 0x3ffffffu);
#pragma unroll
      for (int j = 0; j < CAP; ++j) it[j] = ((uint32_t)j < cnt) ? src[j] : 0u;
    }
    // rank inside (round, coarse bucket), kept as the LDS atomic returns it
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      rk[j] = 0;
      if ((uint32_t)j < cnt) rk[j] = atomicAdd(&s_cnt[it[j] >> 22], 1u);
    }
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
      if (threadIdx.x == kNumCoarse - 1) s_total = pre + inc;
    }
    lds_barrier();
    {
      const uint64_t ip0 = (cur.t + tr) * 8192ull + (ev & 0x1fffu);
#pragma unroll
      for (int j = 0; j < CAP; ++j) {
        if ((uint32_t)j < cnt) {
          const uint32_t h18 = it[j] >> 12, n1 = (it[j] >> 7) & 31u, woff = it[j] & 127u;
          s_stage[s_lofs[h18 >> 10] + rk[j]] = (ip0 + woff) | ((uint64_t)n1 << 41) | ((uint64_t)h18 << 46);
        }
      }
    }
    lds_barrier();
    const uint32_t total = s_total;
    for (uint32_t s = threadIdx.x; s < total; s += NT) {
      const uint64_t d = s_stage[s];
      const uint64_t ip = d & ((1ull << 41) - 1ull);
      const uint32_t n1 = (uint32_t)(d >> 41) & 31u, h18 = (uint32_t)(d >> 46);
      uint64_t w0, w1;
      sk_assemble(st, last_dw, ip, k + n1, n1, h18, w0, w1);
      const uint64_t dst = s_gbase[h18 >> 10] + s;
      reinterpret_cast<ulonglong2 *>(out)[dst] = make_ulonglong2(w0, w1);
    }
    SkRound nxt;
    have = sk_plan(ent_cnt, te, cur.nt, cur.nei, nxt);
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------
// P: fine histogram of a coarse bucket's records (workgroup (c, h) counts the part the scatter groups of half h wrote):
// records and k-mers per fine bucket
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sk_fine_count_kernel(const uint64_t *__restrict__ recs, const uint64_t *__restrict__ wg_off,
                                                            const uint64_t *__restrict__ coarse_end /* [kNumCoarse]: end of every coarse bucket */,
                                                            uint32_t groups, uint32_t *__restrict__ fine_hist /* [kFineParts][kNumFine] */,
                                                            uint32_t *__restrict__ fine_kmers /* [kFineParts][kNumFine]: k-mers */) {
  __shared__ uint32_t s_h[kSubPerCoarse], s_k[kSubPerCoarse];
  const uint32_t gpp = groups / kFineParts;
  const uint32_t c = blockIdx.x / kFineParts, h = blockIdx.x % kFineParts;
  if (threadIdx.x < kSubPerCoarse) { s_h[threadIdx.x] = 0; s_k[threadIdx.x] = 0; }
  lds_barrier();
  const uint64_t b = wg_off[(uint64_t)(h * gpp) * kNumCoarse + c];
  const uint64_t e = (h + 1 < (uint32_t)kFineParts) ? wg_off[(uint64_t)((h + 1) * gpp) * kNumCoarse + c] : coarse_end[c];
  for (uint64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    const uint64_t w1 = recs[2 * i + 1];
    atomicAdd(&s_h[rec_fine_sub(w1)], 1u);
    atomicAdd(&s_k[rec_fine_sub(w1)], ((uint32_t)(w1 >> kRecNShift) & 31u) + 1u);
  }
  lds_barrier();
  if (threadIdx.x < kSubPerCoarse) {
    fine_hist[(uint64_t)h * kNumFine + c * kSubPerCoarse + threadIdx.x] = s_h[threadIdx.x];
    fine_kmers[(uint64_t)h * kNumFine + c * kSubPerCoarse + threadIdx.x] = s_k[threadIdx.x];
  }
}

// ---------------------------------------------------------------------------
// C: per fine bucket, records -> distinct (k-mer, count) pairs
// ---------------------------------------------------------------------------
// inclusive scans across the 64 lanes on the DPP path (row_shr 1, 2, 4, 8 inside the rows of 16, row_bcast 15 / 31 across
// them): six VALU instructions, no LDS crossbar
__device__ __forceinline__ uint32_t wave_inclusive_max_dpp(uint32_t v) {   // identity 0
  uint32_t t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;
  return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_sum_dpp(uint32_t v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
  return v;
}

// table + per-wavefront scratch fill the CU's LDS: (7680 + 64) x 12 B + 16 x 4352 B
struct SkTabCfg {
  static constexpr int CAP = 7680, PAD = 64, SLOTS = CAP + PAD, LIMIT = CAP * 3 / 4, NT = 1024;
  static constexpr int OWN = kWave * 32;   // k-mers of a batch of 64 records at most
};

// The k-mers of a batch of 64 records (one per lane) are numbered 0 .. T - 1 through the records' prefix sums, and lane l
// of step t takes k-mer g = 64 t + l WHATEVER record it belongs to, so every lane works in every step although the records
// hold 1 to 20 k-mers: the record of g is the last one that starts at or before g -- the records mark their first k-mer in
// a byte array (own[P_r] = r + 1), the step reads own[g] and takes a running maximum over the lanes (DPP scan) -- and its
// k-mer j = g - P_r is cut out of the record's 128 bits; the forward strand comes from one reverse complement.
template <bool CANON>
__global__ __launch_bounds__(SkTabCfg::NT) void sk_reduce_kernel(const uint64_t *__restrict__ recs, const uint64_t *__restrict__ rec_off, uint32_t k,
                                                                const uint64_t *__restrict__ kmer_off /* k-mers before every bucket */,
                                                                uint64_t *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_vals,
                                                                uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ flags, int dbg) {
  using T = SkTabCfg;
  constexpr int NWAVES = T::NT / kWave;
  __shared__ uint64_t s_tk[T::SLOTS];
  __shared__ uint32_t s_tv[T::SLOTS];
  __shared__ uint64_t s_missq[NWAVES * kMissQ];
  __shared__ ulonglong2 s_recs[NWAVES * kWave];
  __shared__ uint32_t s_pre[NWAVES * kWave];
  __shared__ uint8_t s_own[NWAVES * T::OWN];
  __shared__ uint32_t s_ctl[8];        // 0 distinct, 1 overflow, 2 special count, 3 special set, 4 emit counter, 5 stack size, 6/7 output base
  __shared__ uint32_t s_stack[64];     // pending passes: filter bits | value << 8
  const uint32_t b = blockIdx.x;
  const uint64_t rb = rec_off[b], re = rec_off[b + 1];
  if (rb == re) { if (threadIdx.x == 0) out_cnt[b] = 0; return; }
  const uint64_t tmp0 = kmer_off[b];   // the bucket's output range: as many slots as it has k-mers (same contract as bucket_reduce_kernel)
  const uint32_t lane = lane_id(), wv = wave_id();
  lds_u64_t *const tkeys = (lds_u64_t *)s_tk;
  lds_u32_t *const tvals = (lds_u32_t *)s_tv;
  lds_u32_t *const tdist = (lds_u32_t *)&s_ctl[0];
  lds_u32_t *const tovf = (lds_u32_t *)&s_ctl[1];
  uint64_t *const mq = s_missq + wv * kMissQ;
  const lds_u64_t *const mql = (const lds_u64_t *)mq;
  ulonglong2 *const wrec = s_recs + wv * kWave;
  uint32_t *const wpre = s_pre + wv * kWave;
  uint8_t *const wown = s_own + wv * T::OWN;
  const uint32_t kb = 2u * k;
  const uint64_t kmask = low_mask64(kb);
  const bool full64 = kb == 64u;   // only then can a key equal the empty marker
  const KShape shape = make_shape(k, 2);
  for (uint32_t i = threadIdx.x; i < (uint32_t)(NWAVES * T::OWN / 4); i += T::NT) reinterpret_cast<uint32_t *>(s_own)[i] = 0;
  if (threadIdx.x == 0) { s_ctl[5] = 1; s_stack[0] = 0; s_ctl[4] = 0; }
  lds_barrier();
  // this wavefront's share of the bucket's records: a contiguous range
  const uint32_t n_rec = (uint32_t)(re - rb);
  const uint32_t share = (n_rec + NWAVES - 1) / NWAVES;
  const uint32_t r_lo = wv * share < n_rec ? wv * share : n_rec;
  const uint32_t r_hi = r_lo + share < n_rec ? r_lo + share : n_rec;
  const ulonglong2 *const src = reinterpret_cast<const ulonglong2 *>(recs) + rb;
  while (true) {
    const uint32_t sp = s_ctl[5];
    if (sp == 0) break;                       // uniform
    const uint32_t pass = s_stack[sp - 1];
    const uint32_t fbits = pass & 0xffu, fval = pass >> 8;
    lds_barrier();                            // everyone has read the stack
    for (uint32_t i = threadIdx.x; i < (uint32_t)T::SLOTS; i += T::NT) { s_tk[i] = kEmptyKey; s_tv[i] = 0; }
    if (threadIdx.x == 0) { s_ctl[0] = 0; s_ctl[1] = 0; s_ctl[2] = 0; s_ctl[3] = 0; s_ctl[5] = sp - 1; }
    lds_barrier();
    // the first three filter bits are the records' sub-bucket bits (whole records are skipped), the others come from the key's hash
    const uint32_t rbits = fbits < 3u ? fbits : 3u, rmask = (1u << rbits) - 1u, rval = fval & rmask;
    const uint32_t hbits = fbits - rbits, hmask = (1u << hbits) - 1u, hval = fval >> rbits;
    uint32_t mn = 0;   // keys waiting in the miss queue (uniform)
    ulonglong2 nxt = make_ulonglong2(0, 0);
    if (r_lo + lane < r_hi) nxt = src[r_lo + lane];
    for (uint32_t r0 = r_lo; r0 < r_hi; r0 += kWave) {
      if (__atomic_load_n(&s_ctl[1], __ATOMIC_RELAXED)) break;   // this pass is lost already
      if (dbg == 3) { if (nxt.x == 12345ull) s_ctl[2] = 1; if (r0 + kWave + lane < r_hi) nxt = src[r0 + kWave + lane]; continue; }   // experiment: records read only
      const ulonglong2 rec = nxt;
      const bool have = r0 + lane < r_hi;
      if (r0 + kWave + lane < r_hi) nxt = src[r0 + kWave + lane];   // in flight while this batch is expanded
      uint32_t n = have ? ((uint32_t)(rec.y >> kRecNShift) & 31u) + 1u : 0u;
      if ((rec_hash18(rec.y) & rmask) != rval) n = 0;
      const uint32_t inc = wave_inclusive_sum_dpp(n);
      const uint32_t pre = inc - n;
      const uint32_t total = __builtin_amdgcn_readlane(inc, kWave - 1);
      wrec[lane] = rec;
      wpre[lane] = pre;
      if (n) wown[pre] = (uint8_t)(lane + 1u);
      uint32_t carry = 0;   // record (+ 1) the previous step ended in
      for (uint32_t g0 = 0; g0 < total; g0 += kWave) {
        const uint32_t g = g0 + lane;
        const bool act = g < total;
        uint32_t o = act ? (uint32_t)wown[g] : 0u;
        o = wave_inclusive_max_dpp(o);
        o = o > carry ? o : carry;
        carry = __builtin_amdgcn_readlane(o, kWave - 1);
        const uint32_t rid = o ? o - 1u : 0u;
        const uint32_t j = g - wpre[rid];
        const ulonglong2 rr = wrec[rid];
        // k-mer j of the record: 2 k bits from bit 2 j of its 128
        const uint32_t sh = 2u * j;   // 0 .. 62
        uint64_t rc = sh ? ((rr.x >> sh) | (rr.y << (64u - sh))) : rr.x;
        rc &= kmask;
        uint64_t key = rc;
        {
          const uint64_t r1[1] = {rc};
          uint64_t f1[1];
          fwd_from_rc<1, 2>(r1, f1, shape);
          key = CANON ? (f1[0] < rc ? f1[0] : rc) : f1[0];
        }
        if (dbg == 1) { if (key == 12345ull) s_ctl[2] = 1; continue; }   // experiment: expansion only
        // table fast path
        const uint64_t kk[1] = {key};
        const uint32_t h = place_hash<1>(kk);
        const uint32_t slot = slot_of(h, T::CAP);
        bool v = act;
        if (hbits) v = v && ((h >> 17) & hmask) == hval;   // (the slot uses the low 17 bits)
        if (full64 && v && key == kEmptyKey) { s_ctl[3] = 1; atomicAdd(&s_ctl[2], 1u); v = false; }
        const uint64_t cur = __atomic_load_n(&s_tk[slot], __ATOMIC_RELAXED);
        const bool hit = v && cur == key;
        if (hit) atomicAdd(&s_tv[slot], 1u);
        const bool miss = v && !hit && dbg != 2;   // (experiment 2: no slow path)
        const unsigned long long mm = __ballot(miss);
        if (mm) {
          const uint32_t pos = mn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
          if (miss) mq[pos] = key;
          mn += (uint32_t)__popcll(mm);
          if (mn >= (uint32_t)kWave) { probe_insert_lds_cap<T::CAP, T::SLOTS, T::LIMIT>(tkeys, tvals, tdist, tovf, mql, mn - kWave, kWave); mn -= kWave; }
        }
      }
      if (n) wown[pre] = 0;   // the marks go back to zero for the next batch
    }
    if (mn) { probe_insert_lds_cap<T::CAP, T::SLOTS, T::LIMIT>(tkeys, tvals, tdist, tovf, mql, 0u, mn); mn = 0; }
    lds_barrier();
    if (s_ctl[1]) {   // overflow: this pass splits in two (one more filter bit)
      if (threadIdx.x == 0) {
        if (fbits >= 18u) atomicOr(&flags[2], 1u);   // 3 record bits + 15 hash bits: 2^18 tables did not hold the bucket
        else {
          const uint32_t spn = s_ctl[5];
          s_stack[spn] = (fbits + 1u) | (fval << 8);
          s_stack[spn + 1] = (fbits + 1u) | ((fval | (1u << fbits)) << 8);
          s_ctl[5] = spn + 2u;
        }
      }
      lds_barrier();
      continue;
    }
    // emit behind what the earlier passes left (disjoint key sets)
    {
      uint32_t *s_out = &s_ctl[4];
      for (uint32_t s = threadIdx.x; s < (uint32_t)((T::SLOTS + kWave - 1) / kWave * kWave); s += T::NT) {
        const bool used = s < (uint32_t)T::SLOTS && s_tk[s] != kEmptyKey;
        const uint32_t pos = wave_alloc(s_out, used);
        if (used) { tmp_keys[tmp0 + pos] = s_tk[s]; tmp_vals[tmp0 + pos] = s_tv[s]; }
      }
      lds_barrier();
      if (threadIdx.x == 0 && s_ctl[3]) {
        const uint32_t pos = atomicAdd(s_out, 1u);
        tmp_keys[tmp0 + pos] = kEmptyKey;
        tmp_vals[tmp0 + pos] = s_ctl[2];
      }
    }
    lds_barrier();
  }
  if (threadIdx.x == 0) out_cnt[b] = s_ctl[4];
}

}  // namespace kmi
