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
constexpr uint64_t kSkPadW1 = ~0ull;   // second word of a PAD record (sk_scatter_fine_slack_lines_kernel): never a record's; readers of a fine bucket skip it
// records of the de Bruijn node build: three windows fewer per record (sk_nmax_of - kSkEdgeWindows), and the two outside bases where
// the last three bases would be: bits 32..34 the base before the first k-mer, 35..37 the base behind the last one, each 1 + its code
// (A C G T = 0..3 in the record's orientation), 0 = the read ends there
constexpr int kRecEdgeShift = 32;
constexpr uint32_t kSkEdgeWindows = 3;
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
// L (FASTA): the run list of a compacted character stream (kmi_fasta.hip: every position is a base, bit r of the break bitmap
// = character r opens a record, or lies behind an N under the N-split rule). Window r is a k-mer iff no break bit falls in
// (r, r + k) and r + k <= n_chars (and r < n_valid in a partition). One wavefront per tile of 8192 positions, lane l owns
// the 128 window positions from 128 l: its runs of valid windows, cut every `seg`, become entries (tile position | (windows
// - 1) << 13) exactly as fastq_list<RUNS> writes them for FASTQ lines.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fasta_runs_kernel(const uint32_t *__restrict__ brk, uint64_t n_chars, uint64_t n_valid, uint64_t n_cover,
                                                        uint64_t n_tiles, uint32_t k, uint32_t seg, uint32_t *__restrict__ ent,
                                                        uint32_t *__restrict__ ent_cnt, uint32_t stride, unsigned long long *__restrict__ n_windows,
                                                        uint32_t *__restrict__ flags) {
  const uint32_t lane = lane_id();
  const uint64_t n_words = n_cover / 32;
  const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x / kWave);
  for (uint64_t t = (uint64_t)blockIdx.x * (blockDim.x / kWave) + wave_id(); t < n_tiles; t += n_waves) {
    const uint64_t p0 = t * 8192ull + (uint64_t)lane * 128ull;
    // break bits of positions p0 + 1 .. p0 + 191 (k <= 32): three 64-bit words starting one position up
    uint64_t x[3];
    {
      uint32_t w[7];
      const uint64_t w0 = p0 >> 5;
#pragma unroll
      for (int i = 0; i < 7; ++i) w[i] = (w0 + i < n_words) ? brk[w0 + i] : 0u;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const uint64_t lo = (uint64_t)w[2 * j] | ((uint64_t)w[2 * j + 1] << 32), hi = (uint64_t)w[2 * j + 2];
        x[j] = (lo >> 1) | (hi << 63);   // (p0 is a multiple of 32: bit 0 of w[0] is position p0 itself)
      }
    }
    smear_right<3>(x, k - 1u);          // bit i: a break in (p0 + i, p0 + i + k)
    uint64_t v0 = ~x[0], v1 = ~x[1];    // valid window starts p0 + 0..63, p0 + 64..127
    {
      // r + k <= n_chars and r < n_valid
      const uint64_t lim_a = n_chars >= k ? n_chars - k + 1u : 0u, lim = lim_a < n_valid ? lim_a : n_valid;   // windows start below lim
      const uint64_t room = lim > p0 ? lim - p0 : 0u;
      v0 &= room >= 64 ? ~0ull : ((1ull << room) - 1ull);
      const uint64_t room1 = room > 64 ? room - 64 : 0u;
      v1 &= room1 >= 64 ? ~0ull : ((1ull << room1) - 1ull);
    }
    const uint32_t nwin = (uint32_t)__popcll(v0) + (uint32_t)__popcll(v1);
    // two walks over the lane's 128 bits: count the entries, then write them behind the lanes before
    auto walk = [&](uint32_t base, bool write) -> uint32_t {
      uint64_t a = v0, b = v1;
      uint32_t c = 0;
      while (a | b) {
        const uint32_t p = a ? (uint32_t)__builtin_ctzll(a) : 64u + (uint32_t)__builtin_ctzll(b);
        // ones from p on: shift the 128 bits down by p and count the trailing ones
        uint64_t lo, hi;
        if (p >= 64u) { lo = b >> (p - 64u); hi = 0; } else { lo = p ? ((a >> p) | (b << (64u - p))) : a; hi = b >> p; }
        uint32_t run = (~lo) ? (uint32_t)__builtin_ctzll(~lo) : 64u + ((~hi) ? (uint32_t)__builtin_ctzll(~hi) : 64u);
        run = run > seg ? seg : run;
        if (write && base + c < stride) ent[t * stride + base + c] = (lane * 128u + p) | ((run - 1u) << 13);
        ++c;
        // clear [p, p + run)
        const uint32_t e = p + run;
        const uint64_t m0 = (e >= 64u ? ~0ull : ((1ull << e) - 1ull)) & ~(p >= 64u ? ~0ull : ((1ull << p) - 1ull));
        const uint64_t m1 = (e <= 64u ? 0ull : (e >= 128u ? ~0ull : ((1ull << (e - 64u)) - 1ull))) & ~(p <= 64u ? 0ull : ((1ull << (p - 64u)) - 1ull));
        a &= ~m0; b &= ~m1;
      }
      return c;
    };
    const uint32_t cnt = walk(0u, false);
    const uint32_t inc = wave_inclusive_scan(cnt);
    const uint32_t total = __shfl(inc, kWave - 1, kWave);
    (void)walk(inc - cnt, true);
    const uint32_t wsum = wave_reduce_sum(nwin);
    if (lane == 0) {
      ent_cnt[t] = total < stride ? total : stride;
      if (total > stride) atomicOr(&flags[9], 1u);   // more runs than the tile's slots: the caller takes the k-mer path
      if (wsum) atomicAdd(n_windows, (unsigned long long)wsum);
    }
  }
}

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
      // the run's stream words, once, into registers: 16 words from the word that holds its first base (a read of 150 bases is
      // 10 of them). A block's window of codes starts at base a + c of these (a = ip0 mod 16 differs per lane, c = b W + m - 1
      // does not), i.e. in word (c >> 4) or the one after it: the four candidate words are picked by a uniform switch, the
      // lane's own carry and shift finish the job -- no memory access inside the walk.
      uint32_t r[16];
      {
        uint64_t d0 = ip0 >> 4;
        d0 = d0 < last_dw ? d0 : last_dw;   // (16 words from here stay inside the buffer and its 64 bytes of slack)
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = st[d0 + i];
      }
      const uint32_t a = (uint32_t)ip0 & 15u;
      auto window = [&](uint32_t c, uint32_t &lo, uint32_t &hi) {   // 64 stream bits from base a + c of the run's words
        uint32_t x0, x1, x2, x3;
        switch (c >> 4) {   // uniform
          case 0: x0 = r[0]; x1 = r[1]; x2 = r[2]; x3 = r[3]; break;
          case 1: x0 = r[1]; x1 = r[2]; x2 = r[3]; x3 = r[4]; break;
          case 2: x0 = r[2]; x1 = r[3]; x2 = r[4]; x3 = r[5]; break;
          case 3: x0 = r[3]; x1 = r[4]; x2 = r[5]; x3 = r[6]; break;
          case 4: x0 = r[4]; x1 = r[5]; x2 = r[6]; x3 = r[7]; break;
          case 5: x0 = r[5]; x1 = r[6]; x2 = r[7]; x3 = r[8]; break;
          case 6: x0 = r[6]; x1 = r[7]; x2 = r[8]; x3 = r[9]; break;
          case 7: x0 = r[7]; x1 = r[8]; x2 = r[9]; x3 = r[10]; break;
          case 8: x0 = r[8]; x1 = r[9]; x2 = r[10]; x3 = r[11]; break;
          case 9: x0 = r[9]; x1 = r[10]; x2 = r[11]; x3 = r[12]; break;
          case 10: x0 = r[10]; x1 = r[11]; x2 = r[12]; x3 = r[13]; break;
          case 11: x0 = r[11]; x1 = r[12]; x2 = r[13]; x3 = r[14]; break;
          default: x0 = r[12]; x1 = r[13]; x2 = r[14]; x3 = r[15]; break;   // (only blocks past the longest run get here)
        }
        const uint32_t o = a + (c & 15u);          // 0 .. 30
        const bool carry = o >= 16u;
        const uint32_t sh = (o & 15u) * 2u;
        const uint32_t y0 = carry ? x1 : x0, y1 = carry ? x2 : x1, y2 = carry ? x3 : x2;
        lo = __builtin_amdgcn_alignbit(y1, y0, sh);
        hi = __builtin_amdgcn_alignbit(y2, y1, sh);
      };
      uint32_t R, F;
      {
        uint32_t lo, hi;
        window(0u, lo, hi);
        R = lo & mmask;
        F = sk_fwd_of(R, m);
      }
      uint32_t sprev[W + 1];
#pragma unroll
      for (int j = 0; j <= W; ++j) sprev[j] = INF;
      // The first window opens a super-k-mer like any other boundary: len starts at nmax ("the one before is full"), and
      // the dummy this closes lands in list slot 0.
      uint32_t prev = 0, len = nmax;
      for (uint32_t b = 0; __any(b < nblk); ++b) {
        // codes of the bases b W + m - 1 .. b W + m + W - 2 (the base that completes m-mer position q = b W + j is q + m - 1)
        uint32_t clo, chi;
        window(b * (uint32_t)W + m - 1u, clo, chi);
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
            s_list[slot * NT + threadIdx.x] = (prev << 7) | (len - 1u);
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
        s_list[slot * NT + threadIdx.x] = (prev << 7) | (len - 1u);
        ++cnt;
      }
    }
    if (cnt > (uint32_t)CAP + 1u) { atomicOr(&flags[9], 1u); cnt = 1; }
    cnt = cnt ? cnt - 1u : 0u;   // real items: slots 1 .. cnt
    // items in their final form (window offset | (n - 1) << 7 | bucket bits << 12 | two further hash bits << 30) + the coarse counts
    {
      uint32_t off = 0;
      for (uint32_t j = 1; j <= cnt; ++j) {
        const uint32_t it = s_list[j * NT + threadIdx.x];
        const uint32_t n1 = it & 127u;
        const uint32_t h20 = sk_bucket_bits20(it >> 7), h18 = h20 >> 2;
        atomicAdd(&s_cnt[h18 >> 10], 1u);
        s_list[j * NT + threadIdx.x] = off | (n1 << 7) | (h18 << 12) | ((h20 & 3u) << 30);   // (+ two more hash bits on top)
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

// the same from a run's row of stream words in LDS: bit position `bit` of the row
// CANON (canonical strand model): of the super-k-mer and its reverse complement the smaller bit pattern is kept. The two
// strands' reads of a stretch of genome cut it into mirror-image super-k-mers (same minimizers, same boundaries); their
// k-mers are the same canonical k-mers, so which mirror image travels does not matter to the count -- but with ONE form the
// reduce finds the copies of both strands identical and expands them once (sk_reduce, T1).
template <bool CANON>
__device__ __forceinline__ void sk_assemble_row(const uint32_t *row, uint32_t bit, uint32_t nb, uint32_t n1, uint32_t h18, uint64_t &w0, uint64_t &w1,
                                                bool *flipped = nullptr) {
  const uint32_t d = bit >> 5, sh = bit & 31u;
  const uint32_t r0 = row[d], r1 = row[d + 1], r2 = row[d + 2], r3 = row[d + 3], r4 = row[d + 4];
  const uint32_t a0 = __builtin_amdgcn_alignbit(r1, r0, sh), a1 = __builtin_amdgcn_alignbit(r2, r1, sh);
  const uint32_t a2 = __builtin_amdgcn_alignbit(r3, r2, sh), a3 = __builtin_amdgcn_alignbit(r4, r3, sh);
  w0 = (uint64_t)a0 | ((uint64_t)a1 << 32);
  uint64_t hi = (uint64_t)a2 | ((uint64_t)a3 << 32);
  const uint32_t bits = 2u * nb;   // 34 .. 102
  if (bits < 64u) { w0 &= (1ull << bits) - 1ull; hi = 0; }
  else hi &= (1ull << (bits - 64u)) - 1ull;
  if (CANON) {
    // reverse complement of nb bases held as complement codes, base i at bits 2 i: reverse all 128 bits (the two bits of a
    // base change places: swap them back), complement, and bring the nb bases down from the top
    uint64_t r_hi = __builtin_bitreverse64(w0), r_lo = __builtin_bitreverse64(hi);
    r_hi = ~(((r_hi >> 1) & 0x5555555555555555ull) | ((r_hi & 0x5555555555555555ull) << 1));
    r_lo = ~(((r_lo >> 1) & 0x5555555555555555ull) | ((r_lo & 0x5555555555555555ull) << 1));
    const uint32_t sh = 128u - bits;   // 26 .. 94
    uint64_t c_lo, c_hi;
    if (sh < 64u) { c_lo = (r_lo >> sh) | (r_hi << (64u - sh)); c_hi = r_hi >> sh; }
    else { c_lo = r_hi >> (sh - 64u); c_hi = 0; }
    const bool take = c_hi < hi || (c_hi == hi && c_lo < w0);
    w0 = take ? c_lo : w0;
    hi = take ? c_hi : hi;
    if (flipped) *flipped = take;
  }
  w1 = hi | ((uint64_t)n1 << kRecNShift) | ((uint64_t)h18 << kRecHashShift);
}

constexpr int kSkRowDw = 17;   // stream words of a run kept in LDS: up to 63 bases of alignment + 128 + 31 bases = 444 bits in 16 words (odd stride: rows on different banks)

template <bool CANON>
__global__ __launch_bounds__(kSkThreads, 2) void sk_scatter_kernel(PackedInput in, uint64_t n_tiles, uint32_t k, const uint32_t *__restrict__ ent,
                                                                  const uint32_t *__restrict__ ent_cnt, uint32_t ent_stride, uint32_t items_per_tile,
                                                                  const uint32_t *__restrict__ items, const uint32_t *__restrict__ run_items,
                                                                  const uint64_t *__restrict__ wg_off, uint64_t *__restrict__ out, uint32_t lp) {
  // lp (a build over 2^lp ranks): the records are grouped by the top 8 bucket bits as always -- the top lp of them name the
  // rank that will own the record -- but carry the bucket bits shifted left by lp: what is left of them is the bucket inside
  // the owner's index (kmi_index::layout_w holds lp next to W)
  // A round = the runs of up to 512 lanes. Every lane brings the stream words of its run into LDS (three 16-byte loads of
  // consecutive memory: gathering them per RECORD on the way out cost five loads of 64 different lines each). The items are
  // bucket-sorted as 16-bit references (lane of the run << 5 | item number): count per coarse bucket, scan, place. The
  // copy-out walks the sorted references, so consecutive threads write consecutive 16-byte records of one bucket, each cut
  // from its run's row.
  constexpr int NT = kSkThreads, CAP = kSkListCap;
  __shared__ uint16_t s_stage[CAP * NT];
  __shared__ uint32_t s_row[NT * kSkRowDw + 4];   // (a record's five-word read window may reach past the last row)
  __shared__ uint32_t s_bit0[NT];       // bit of the run's first base inside its row
  __shared__ uint32_t s_ioff[NT];       // first item of every run in the workgroup's item stream
  __shared__ uint32_t s_cnt[kNumCoarse];
  __shared__ uint32_t s_cur[kNumCoarse];
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
  const uint32_t *const wg_items = items + tb * (uint64_t)items_per_tile;
  SkRound cur;
  bool have = tb < te && sk_plan(ent_cnt, te, tb, 0u, cur);
  while (have) {
    lds_barrier();   // the previous round's stage and run tables are done with; the counters are clear
    uint32_t cnt = 0;
    uint32_t it[CAP];
    {
      uint32_t tr = 0, ev = 0, ri = 0;
      uint64_t eidx = 0;
      if (threadIdx.x < cur.total) { sk_locate(cur, threadIdx.x, ent_stride, tr, eidx); ev = ent[eidx]; ri = run_items[eidx]; }
      cnt = ri >> 26;
      const uint32_t *src = wg_items + (ri & 0x3ffffffu);
#pragma unroll
      for (int j = 0; j < CAP; ++j) it[j] = ((uint32_t)j < cnt) ? src[j] : 0u;
      const uint64_t ip0 = (cur.t + tr) * 8192ull + (ev & 0x1fffu);
      uint64_t d0 = (ip0 >> 4) & ~3ull;               // 16-byte aligned: dword index, multiple of four
      d0 = d0 + 16 <= last_dw + 16 ? d0 : 0;          // (16 words from here stay inside the buffer and its 64 bytes of slack)
      const uint4 q0 = *reinterpret_cast<const uint4 *>(st + d0), q1 = *reinterpret_cast<const uint4 *>(st + d0 + 4),
                  q2 = *reinterpret_cast<const uint4 *>(st + d0 + 8), q3 = *reinterpret_cast<const uint4 *>(st + d0 + 12);
      uint32_t *row = s_row + threadIdx.x * kSkRowDw;
      row[0] = q0.x; row[1] = q0.y; row[2] = q0.z; row[3] = q0.w; row[4] = q1.x; row[5] = q1.y; row[6] = q1.z; row[7] = q1.w;
      row[8] = q2.x; row[9] = q2.y; row[10] = q2.z; row[11] = q2.w; row[12] = q3.x; row[13] = q3.y; row[14] = q3.z; row[15] = q3.w;
      row[16] = 0;
      s_bit0[threadIdx.x] = (uint32_t)(ip0 - d0 * 16) * 2u;   // 0 .. 126
      s_ioff[threadIdx.x] = ri & 0x3ffffffu;
    }
#pragma unroll
    for (int j = 0; j < CAP; ++j)
      if ((uint32_t)j < cnt) atomicAdd(&s_cnt[(it[j] >> 22) & 255u], 1u);
    lds_barrier();
    uint32_t c = 0, inc = 0;
    if (threadIdx.x < kNumCoarse) {
      c = s_cnt[threadIdx.x];
      s_cnt[threadIdx.x] = 0;
      inc = wave_inclusive_sum_dpp(c);
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
    }
    lds_barrier();
    if (threadIdx.x < kNumCoarse) {
      uint32_t pre = 0;
#pragma unroll
      for (uint32_t w = 0; w < kNumCoarse / kWave; ++w) pre += (w < wave_id()) ? s_part[w] : 0u;
      const uint32_t lo = pre + inc - c;
      s_cur[threadIdx.x] = lo;
      s_gbase[threadIdx.x] = cursor - lo;
      cursor += c;
      if (threadIdx.x == kNumCoarse - 1) s_total = pre + inc;
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < CAP; ++j)
      if ((uint32_t)j < cnt) s_stage[atomicAdd(&s_cur[(it[j] >> 22) & 255u], 1u)] = (uint16_t)((threadIdx.x << 5) | (uint32_t)j);
    lds_barrier();
    const uint32_t total = s_total;
    for (uint32_t s = threadIdx.x; s < total; s += NT) {
      const uint32_t e = s_stage[s], rl = e >> 5, j = e & 31u;
      const uint32_t item = wg_items[s_ioff[rl] + j];
      const uint32_t h18 = (item >> 12) & 0x3ffffu, n1 = (item >> 7) & 31u, extra = item >> 30;
      // the bucket bits as the owner will read them: the lp rank bits shifted out, the two spare hash bits shifted in below
      const uint32_t hs = ((h18 << lp) & 0x3ffffu) | (lp <= 2u ? extra >> (2u - lp) : extra << (lp - 2u));
      uint64_t w0, w1;
      sk_assemble_row<CANON>(s_row + rl * kSkRowDw, s_bit0[rl] + 2u * (item & 127u), k + n1, n1, hs, w0, w1);
      reinterpret_cast<ulonglong2 *>(out)[s_gbase[h18 >> 10] + s] = make_ulonglong2(w0, w1);
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

// P without a counting pass: the fine buckets of coarse bucket c get room for fine_cap[c] records each (a quarter above the
// mean share of c's records, which the front end counted), and a workgroup appends its tile's records of a fine bucket behind
// what the bucket holds so far -- one global atomic per (tile, fine bucket), 128 per 4096 records -- instead of writing to
// offsets a histogram pass (sk_fine_count: one more read of all records) would have prepared. It also adds up the k-mers per fine
// bucket (the reduce's output ranges). A bucket that outgrows its room raises flags[34]: the caller then takes the exact path.
// Bucket b's records end up at fine_region[b >> 7] + (b & 127) * fine_cap[b >> 7], fine_cnt[b] of them, in no particular order.
constexpr int kSlackTile = 4096;
__global__ __launch_bounds__(kPartThreads) void sk_scatter_fine_slack_kernel(const uint64_t *__restrict__ recs, uint64_t *__restrict__ out,
                                                                           const uint64_t *__restrict__ wg_off, const uint64_t *__restrict__ coarse_end,
                                                                           uint32_t groups, const uint64_t *__restrict__ fine_region,
                                                                           const uint32_t *__restrict__ fine_cap, uint32_t *__restrict__ fine_cnt,
                                                                           uint32_t *__restrict__ fine_kmers, uint32_t *__restrict__ flags) {
  constexpr int TILE = kSlackTile, PT = TILE / kPartThreads;
  __shared__ ulonglong2 s_stage[TILE];
  __shared__ uint8_t s_bkt[TILE];
  __shared__ uint32_t s_cnt[kSubPerCoarse], s_k[kSubPerCoarse], s_lofs[kSubPerCoarse];
  __shared__ uint64_t s_gbase[kSubPerCoarse];
  __shared__ uint32_t s_part[kSubPerCoarse / kWave];
  const uint32_t gpp = groups / kFineParts;
  const uint32_t c = blockIdx.x / kFineParts, h = blockIdx.x % kFineParts;
  const uint64_t b = wg_off[(uint64_t)(h * gpp) * kNumCoarse + c];
  const uint64_t e = (h + 1 < (uint32_t)kFineParts) ? wg_off[(uint64_t)((h + 1) * gpp) * kNumCoarse + c] : coarse_end[c];
  const uint64_t region = fine_region[c];
  const uint32_t cap = fine_cap[c];
  if (threadIdx.x < kSubPerCoarse) { s_cnt[threadIdx.x] = 0; s_k[threadIdx.x] = 0; }
  lds_barrier();
  const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(recs);
  ulonglong2 raw[PT];
  auto load_tile = [&](uint64_t t0) {
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      uint64_t i = t0 + (uint64_t)j * kPartThreads + threadIdx.x;
      i = (i < e) ? i : (e ? e - 1 : 0);
      raw[j] = src[i];
    }
  };
  if (b < e) load_tile(b);
  for (uint64_t t0 = b; t0 < e; t0 += TILE) {
    const uint32_t nt = (uint32_t)((e - t0 < (uint64_t)TILE) ? (e - t0) : (uint64_t)TILE);
    ulonglong2 rec[PT];
    uint32_t bk[PT], rk[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const uint32_t li = j * kPartThreads + threadIdx.x;
      rec[j] = raw[j];
      bk[j] = 0xffffffffu;
      if (li < nt) {
        bk[j] = rec_fine_sub(rec[j].y);
        rk[j] = atomicAdd(&s_cnt[bk[j]], 1u);
        atomicAdd(&s_k[bk[j]], ((uint32_t)(rec[j].y >> kRecNShift) & 31u) + 1u);
      }
    }
    if (t0 + TILE < e) load_tile(t0 + TILE);   // in flight until the next iteration needs it
    lds_barrier();
    uint32_t cnt = 0, inc = 0;
    if (threadIdx.x < kSubPerCoarse) {              // two whole waves
      cnt = s_cnt[threadIdx.x];
      const uint32_t km = s_k[threadIdx.x];
      s_cnt[threadIdx.x] = 0; s_k[threadIdx.x] = 0;
      inc = wave_inclusive_sum_dpp(cnt);
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
      const uint32_t f = c * kSubPerCoarse + threadIdx.x;
      uint32_t at = 0;
      if (cnt) { at = atomicAdd(&fine_cnt[f], cnt); atomicAdd(&fine_kmers[f], km); }
      // room for this tile's share? (a bucket that outgrows it keeps counting, so the caller sees by how much, but writes nothing)
      const bool fits = (uint64_t)at + cnt <= (uint64_t)cap;
      if (!fits) atomicOr(&flags[34], 1u);
      s_gbase[threadIdx.x] = fits ? region + (uint64_t)threadIdx.x * cap + at : ~0ull;
    }
    lds_barrier();
    if (threadIdx.x < kSubPerCoarse) {
      const uint32_t lo = (wave_id() ? s_part[0] : 0u) + inc - cnt;
      s_lofs[threadIdx.x] = lo;
      if (s_gbase[threadIdx.x] != ~0ull) s_gbase[threadIdx.x] -= lo;
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      if (bk[j] != 0xffffffffu) {
        const uint32_t pos = s_lofs[bk[j]] + rk[j];
        s_stage[pos] = rec[j];
        s_bkt[pos] = (uint8_t)bk[j];
      }
    }
    lds_barrier();
    for (uint32_t s = threadIdx.x; s < nt; s += kPartThreads) {
      const uint64_t gb = s_gbase[s_bkt[s]];
      if (gb != ~0ull) reinterpret_cast<ulonglong2 *>(out)[gb + s] = s_stage[s];
    }
    // (no barrier here: the next tile's counting only touches s_cnt / s_k, reset above; its staging is two barriers away)
  }
}

// The same pass writing WHOLE LINES. A tile's records of a fine bucket used to go where the bucket's counter stood: runs of about 32
// records that begin and end inside 128-byte lines, which cost 1.7 x what whole lines cost (tools/write_runs.hip, modes 11 / 12). Here a
// bucket's records leave in groups of 8 (one line): what a tile brings and what the workgroup still carries of that bucket is cut at the
// last multiple of 8, room for exactly that many is taken from the bucket's counter -- so every allocation of every workgroup is a
// multiple of 8 and, the regions starting on lines, every group is a whole line -- and the tail of < 8 records travels on in registers
// (8 slots per bucket, one per thread of the bucket's eight). At its end a workgroup fills its tails up to 8 with PAD records (second
// word all ones: never a record's, kSkPadW1), which every reader of a fine bucket skips -- at most 14 per bucket, 0.3 % of config 2's.
constexpr int kSlackLineRecs = 8;                    // 16-byte records per 128-byte line
constexpr int kSlackLT = 3072;                       // new records per tile: + 7 carried per bucket = 3968 staged records, 62 KB (two workgroups per CU)
constexpr int kSlackLCap = kSlackLT + (kSlackLineRecs - 1) * kSubPerCoarse;
__global__ __launch_bounds__(kPartThreads) void sk_scatter_fine_slack_lines_kernel(const uint64_t *__restrict__ recs, uint64_t *__restrict__ out,
                                                                                 const uint64_t *__restrict__ wg_off, const uint64_t *__restrict__ coarse_end,
                                                                                 uint32_t groups, const uint64_t *__restrict__ fine_region,
                                                                                 const uint32_t *__restrict__ fine_cap, uint32_t *__restrict__ fine_cnt,
                                                                                 uint32_t *__restrict__ fine_kmers, uint32_t *__restrict__ flags) {
  constexpr int NB = kSubPerCoarse, T = kSlackLT, PT = T / kPartThreads, SCAP = kSlackLCap, L = kSlackLineRecs, MAXG = SCAP / L + NB;
  static_assert(kPartThreads / NB == L && T % kPartThreads == 0, "one carry slot per thread: eight threads per bucket");
  __shared__ ulonglong2 s_stage[SCAP];
  __shared__ uint32_t s_cnt[NB];       // carry + new records of the tile
  __shared__ uint32_t s_k[NB];         // k-mers of the tile's new records
  __shared__ uint32_t s_lofs[NB];      // first stage slot of the bucket
  __shared__ uint32_t s_emit[NB];      // records that leave this tile
  __shared__ uint32_t s_old[NB];       // records carried into this tile
  __shared__ uint32_t s_lbase[NB];     // first destination group of the bucket
  __shared__ uint64_t s_dst[NB];       // where this tile's groups of the bucket go (~0: the bucket has outgrown its room)
  __shared__ uint32_t s_part[NB / kWave];
  __shared__ uint32_t s_ng;
  __shared__ uint8_t s_linebkt[MAXG];
  const uint32_t gpp = groups / kFineParts;
  const uint32_t c = blockIdx.x / kFineParts, h = blockIdx.x % kFineParts;
  const uint64_t b = wg_off[(uint64_t)(h * gpp) * kNumCoarse + c];
  const uint64_t e = (h + 1 < (uint32_t)kFineParts) ? wg_off[(uint64_t)((h + 1) * gpp) * kNumCoarse + c] : coarse_end[c];
  if (b >= e) return;
  const uint64_t region = fine_region[c];
  const uint32_t cap = fine_cap[c];
  if (threadIdx.x < NB) { s_cnt[threadIdx.x] = 0; s_k[threadIdx.x] = 0; }
  const uint32_t cb = threadIdx.x / L, cj = threadIdx.x % L;   // carry: slot cj of bucket cb
  ulonglong2 carry = make_ulonglong2(0, 0);
  uint32_t my_carry = 0;   // thread f < NB: records it carries of bucket f
  lds_barrier();
  const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(recs);
  ulonglong2 raw[PT];
  auto load_tile = [&](uint64_t t0) {
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      uint64_t i = t0 + (uint64_t)j * kPartThreads + threadIdx.x;
      i = (i < e) ? i : e - 1;
      raw[j] = src[i];
    }
  };
  // room for `n` records (a multiple of 8) of fine bucket f: the position inside the region, or ~0 when the bucket has outgrown it
  auto take = [&](uint32_t f, uint32_t n) -> uint64_t {
    const uint32_t at = atomicAdd(&fine_cnt[c * NB + f], n);
    if ((uint64_t)at + n > (uint64_t)cap) { atomicOr(&flags[34], 1u); return ~0ull; }
    return region + (uint64_t)f * cap + at;
  };
  load_tile(b);
  for (uint64_t t0 = b; t0 < e; t0 += T) {
    const uint32_t nt = (uint32_t)((e - t0 < (uint64_t)T) ? (e - t0) : (uint64_t)T);
    ulonglong2 rec[PT];
    uint32_t bk[PT], rk[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const uint32_t li = j * kPartThreads + threadIdx.x;
      rec[j] = raw[j];
      bk[j] = 0xffffffffu;
      if (li < nt) {
        bk[j] = rec_fine_sub(rec[j].y);
        rk[j] = atomicAdd(&s_cnt[bk[j]], 1u);   // rank behind the carried records: s_cnt starts at the carry count
        atomicAdd(&s_k[bk[j]], ((uint32_t)(rec[j].y >> kRecNShift) & 31u) + 1u);
      }
    }
    if (t0 + T < e) load_tile(t0 + T);
    lds_barrier();
    uint32_t cnt = 0, emit = 0, ng = 0, inc = 0;
    if (threadIdx.x < NB) {   // two whole waves
      cnt = s_cnt[threadIdx.x];
      const uint32_t km = s_k[threadIdx.x];
      s_k[threadIdx.x] = 0;
      emit = cnt & ~(uint32_t)(L - 1);
      ng = emit / L;
      inc = wave_inclusive_sum_dpp(cnt | (ng << 16));
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
      if (km) atomicAdd(&fine_kmers[c * NB + threadIdx.x], km);
      s_dst[threadIdx.x] = emit ? take(threadIdx.x, emit) : ~0ull;
    }
    lds_barrier();
    if (threadIdx.x < NB) {
      const uint32_t pre = wave_id() ? s_part[0] : 0u;
      const uint32_t ex = pre + inc - (cnt | (ng << 16));
      const uint32_t lo = ex & 0xffffu, lb = ex >> 16;
      s_lofs[threadIdx.x] = lo; s_emit[threadIdx.x] = emit; s_old[threadIdx.x] = my_carry; s_lbase[threadIdx.x] = lb;
      for (uint32_t i = 0; i < ng; ++i) s_linebkt[lb + i] = (uint8_t)threadIdx.x;
      if (threadIdx.x == NB - 1) s_ng = lb + ng;
      s_cnt[threadIdx.x] = cnt - emit;   // the next tile ranks behind these
      my_carry = cnt - emit;
    }
    lds_barrier();
    // stage: carried records first, then the tile's, per bucket
    if (cj < s_old[cb]) s_stage[s_lofs[cb] + cj] = carry;
#pragma unroll
    for (int j = 0; j < PT; ++j)
      if (bk[j] != 0xffffffffu) s_stage[s_lofs[bk[j]] + rk[j]] = rec[j];
    lds_barrier();
    {
      const uint32_t n_groups = s_ng, l8 = threadIdx.x & (uint32_t)(L - 1);
      for (uint32_t g = threadIdx.x / L; g < n_groups; g += kPartThreads / L) {
        const uint32_t f = s_linebkt[g];
        const uint64_t d = s_dst[f];
        const uint32_t i = (g - s_lbase[f]) * L + l8;
        if (d != ~0ull) reinterpret_cast<ulonglong2 *>(out)[d + i] = s_stage[s_lofs[f] + i];
      }
      // what stays behind the last group travels on in registers
      const uint32_t rem = s_cnt[cb], base = s_lofs[cb] + s_emit[cb];
      if (cj < rem) carry = s_stage[base + cj];
    }
    lds_barrier();   // the next tile rewrites the stage and the per-bucket tables
  }
  // the tails: filled up to a whole line with pad records
  {
    if (threadIdx.x < NB) s_dst[threadIdx.x] = my_carry ? take(threadIdx.x, (uint32_t)L) : ~0ull;
    lds_barrier();
    const uint32_t rem = s_cnt[cb];
    const uint64_t d = s_dst[cb];
    if (rem && d != ~0ull) reinterpret_cast<ulonglong2 *>(out)[d + cj] = cj < rem ? carry : make_ulonglong2(kEmptyKey, kSkPadW1);
  }
}

// the words a build zeroes before its kernels start, in one launch instead of a fill per array (a fill is a kernel of its own:
// five microseconds each on the stream, and a build has a dozen): up to three arrays of dwords and two short ranges of the flags
__global__ __launch_bounds__(1024) void sk_zero_kernel(uint32_t *__restrict__ a, uint32_t na, uint32_t *__restrict__ b, uint32_t nb,
                                                     uint32_t *__restrict__ c, uint32_t nc, uint32_t *__restrict__ flags, uint32_t f0, uint32_t f1,
                                                     uint32_t g0, uint32_t g1) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (uint32_t x = i; x < na; x += stride) a[x] = 0;
  for (uint32_t x = i; x < nb; x += stride) b[x] = 0;
  for (uint32_t x = i; x < nc; x += stride) c[x] = 0;
  if (i >= f0 && i < f1) flags[i] = 0;
  if (i >= g0 && i < g1) flags[i] = 0;
}

// Records that arrived from the other ranks of a build (a flat array, every source's part grouped by the sender's buckets):
// coarse-bucket counts of every workgroup's chunk (the chunks scatter_chunks_kernel will take) and the k-mers they hold
__global__ __launch_bounds__(kPartThreads) void sk_recv_hist_kernel(const uint64_t *__restrict__ recs, uint64_t n, uint32_t *__restrict__ wg_hist,
                                                                   unsigned long long *__restrict__ n_kmers) {
  __shared__ uint32_t s_h[kNumCoarse];
  __shared__ unsigned long long s_k;
  if (threadIdx.x < kNumCoarse) s_h[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_k = 0ull;
  lds_barrier();
  const uint64_t chunk = part_chunk(n, gridDim.x, PartCfg<2>::TILE);
  const uint64_t b = (uint64_t)blockIdx.x * chunk, e = (b + chunk < n) ? b + chunk : n;
  unsigned long long km = 0;
  for (uint64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    const uint64_t w1 = recs[2 * i + 1];
    atomicAdd(&s_h[rec_hash18(w1) >> 10], 1u);
    km += ((uint32_t)(w1 >> kRecNShift) & 31u) + 1u;
  }
  km = wave_reduce_sum(km);
  if (lane_id() == 0 && km) atomicAdd(&s_k, km);
  lds_barrier();
  if (threadIdx.x < kNumCoarse) wg_hist[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] = s_h[threadIdx.x];
  if (threadIdx.x == 0 && s_k) atomicAdd(n_kmers, s_k);
}

// ---------------------------------------------------------------------------
// C: per fine bucket, records -> distinct (k-mer, count) pairs
// ---------------------------------------------------------------------------
// ---- sk_reduce -----------------------------------------------------------------------------------------------------
// Two tables per workgroup. Reads that cover the same stretch of the genome cut it into the SAME super-k-mers (the
// boundaries depend on the bases alone), so at sequencing coverage most records of a bucket are copies of one another --
// all but the ones a read's end cut short. T1 counts identical records first (phase A: 128-bit keys); only its distinct
// records are expanded (phase B), and each of their k-mers enters the k-mer table T2 once, with the record's multiplicity
// as its weight: for 12x coverage about a quarter of the expansions and table operations. T1 is an accelerator, not a
// set: a record that finds no room in it (table full, long probe walk, an unlucky race between a claim and its second
// word) is expanded directly with weight 1, so nothing depends on T1 holding every distinct record exactly once.
//
// Expansion (both for T1's slots and for the overflow list): the k-mers of a batch of 64 records (one per lane) are taken two
// at a time -- a UNIT is two neighbouring k-mers of one record -- and the units are numbered 0 .. T - 1 through the records'
// prefix sums; lane l of step t takes unit g = 64 t + l WHATEVER record it belongs to, so every lane works in every step
// although the records hold 1 to 21 k-mers: the record of g is the last one that starts at or before g -- the records mark
// their first unit in a byte array (own[P_r] = r + 1), the step reads own[g] and takes a running maximum over the lanes (DPP
// scan) -- its words come over the lane crossbar (ds_bpermute), its first k-mer j = 2 (g - P_r) is cut out of the record's
// 128 bits with one reverse complement for the forward strand, and its second k-mer is the first one moved on by a base.
#ifndef KMI_SK_H1
#define KMI_SK_H1 1536   // (2048 left the k-mer table 1000 slots less: config 2 the same, 0.2 - 0.3 ms slower where buckets hold 3800 keys)
#endif
#ifndef KMI_SK_T1_DUP
#define KMI_SK_T1_DUP 0.5f   // distinct k-mers per occurrence up to which identical records are worth counting first
#endif
#ifndef KMI_SK_FILL
#define KMI_SK_FILL 80
#endif
#ifndef KMI_SK_NT
#define KMI_SK_NT 1024      // threads of a sk_reduce workgroup ...
#endif
#ifndef KMI_SK_LDS_KB
#define KMI_SK_LDS_KB 160   // ... and the LDS it may take. Measured on config 2 (3.55 ms): 512 threads / 80 KB, two workgroups per CU, H1 768 or
                            // 512: 3.60 - 3.63 (same wavefronts per CU, twice the passes); 1024 threads / 80 KB, H1 512 / 256: 7.0 / 8.2 ms
#endif
// slot hash of the k-mer table (private to this kernel: two multiplies instead of the placement hash's three)
#ifndef KMI_SK_HASH24
#define KMI_SK_HASH24 1
#endif
__device__ __forceinline__ uint32_t sk_slot_hash(uint64_t key) {
#if KMI_SK_HASH24
  // three 24-bit multiplies (full rate; a 32-bit multiply issues at a quarter of it) over the key's bits 0..23, 24..47, 48..63
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  uint32_t h = __umul24(lo, 0x9E3779u);
  h += __umul24(__builtin_amdgcn_alignbit(hi, lo, 24), 0x85EBCBu);
  h += __umul24(hi >> 16, 0xC2B2AFu);
#else
  uint32_t h = ((uint32_t)key ^ ((uint32_t)(key >> 32) * 0x85EBCA6Bu)) * 0x9E3779B1u;
#endif
  h ^= h >> 15;
  return h;
}
__device__ __forceinline__ uint32_t sk_slot_of(uint32_t h, uint32_t cap) { return __umul24(h >> 16, cap) >> 16; }   // (cap < 2^16: a 24-bit multiply)

// the slow path of the k-mer table, out of line: queue entries [first, first + cnt) (key, weight), one per lane
#ifndef KMI_SK_PROBE_ATTR
#define KMI_SK_PROBE_ATTR __forceinline__   // (an out-of-line function begins with s_waitcnt vmcnt(0): every call waited for the record loads in flight)
#endif
__device__ KMI_SK_PROBE_ATTR uint32_t sk_probe_insert(lds_u64_t *tkeys, lds_u32_t *tvals, lds_u32_t *distinct, lds_u32_t *overflow,
                                                         const lds_u64_t *q, const lds_u32_t *qw, uint32_t first, uint32_t cnt,
                                                         uint32_t cap, uint32_t last, uint32_t limit, uint32_t pending) {
  const uint32_t lane = lane_id();
  bool claimed = false;
  if (lane < cnt) {
    const uint64_t key = q[first + lane];
    const uint32_t wt = qw[first + lane];
    const uint32_t s0 = sk_slot_of(sk_slot_hash(key), cap);
    uint32_t s = s0;
    uint64_t c = __atomic_load_n(&tkeys[s], __ATOMIC_RELAXED);
    for (;;) {
      while (c != key && c != kEmptyKey && s - s0 < 64u) { ++s; c = __atomic_load_n(&tkeys[s], __ATOMIC_RELAXED); }   // the walk (no wrap: padded table)
      if (c == key) break;
      // an empty slot, or the walk got too long: a walk of 64 slots or one that reaches the end of the padding ends the pass
      // (the other wavefronts keep filling the table until they see the flag: without the bound a full table turns every
      // walk into a scan of all of it)
      if (c != kEmptyKey || s >= last) { *overflow = 1; s = last; break; }
      uint64_t expected = kEmptyKey;
      if (__atomic_compare_exchange_n(&tkeys[s], &expected, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { claimed = true; break; }
      c = expected;
    }
    __atomic_fetch_add(&tvals[s], wt, __ATOMIC_RELAXED);   // (slot `last` never holds a key: counts parked there are never read)
  }
  // fill level: the claims of this call (and the `pending` ones the caller made in line since its last call) go to the shared
  // counter in ONE add per wavefront; its return value is the level -- past `limit` the pass is over (a linear-probing
  // table near full turns every insert into a long walk)
  const unsigned long long cm = __ballot(claimed);
  const uint32_t add = (uint32_t)__popcll(cm) + pending;
  if (add >= 16u || (add && cnt < 64u)) {   // (small amounts ride along with the next call; the last call of a pass flushes)
    if (lane == 0 && __atomic_fetch_add(distinct, add, __ATOMIC_RELAXED) + add >= limit) *overflow = 1;
    return 0u;
  }
  return add;
}


// ---- the kernel ------------------------------------------------------------------------------------------------------
// The two tables and the expansion described above, organised so that nothing is paid per bucket that can be paid per workgroup:
//  * one workgroup per CU stays resident and pulls fine buckets from a queue (one atomic per bucket, issued a bucket ahead);
//  * the tables are cleared ONCE; afterwards the sweep that emits a pass's (k-mer, count) pairs leaves every slot it read
//    empty, and phase B empties every record slot it expands -- a pass starts on clean tables without a clear of its own;
//  * records that find no room in T1 go to a short overflow list and are expanded in phase B with everything else (the
//    direct expansions inside phase A cost a whole expansion step for the two or three records of a batch that needed one);
//  * the first records of the NEXT bucket are loaded before the emit sweep of the current one, so a bucket does not start
//    with two dependent trips to HBM (its offsets, then its records).
#ifndef KMI_SK_OVF
#define KMI_SK_OVF 448   // overflow list entries (records that T1 did not take). With H1 1536: 100 record slots + 28 list entries per wavefront in phase B = two full steps of 64 lanes; 192: 3.12 ms, 384: 2.98, 448: 2.97, 512: 3.00 (config 2)
#endif
constexpr int kSkOvf = KMI_SK_OVF;
template <int OWN_>
struct SkTab2 {
  static constexpr int NT = KMI_SK_NT, NWAVES = NT / kWave;
  static constexpr int OWN = OWN_;
  static constexpr int H1 = KMI_SK_H1, S1 = H1 + 64, L1 = H1 * 3 / 4;
  static constexpr int FIXED = NWAVES * (OWN + kMissQ * 12) + S1 * 20 + kSkOvf * 16 + 2560;
  static constexpr int S2 = ((KMI_SK_LDS_KB * 1024 - FIXED) / 12) / 64 * 64;
  static constexpr int CAP2 = S2 - 64, LIMIT2 = CAP2 * KMI_SK_FILL / 100;
};

template <bool CANON, int OWN_, bool SPECIAL>
#ifndef KMI_SK_MIN_WAVES
#define KMI_SK_MIN_WAVES 4   // wavefronts per SIMD the register budget has to allow
#endif
__global__ __launch_bounds__(KMI_SK_NT, KMI_SK_MIN_WAVES) void sk_reduce_kernel(const uint64_t *__restrict__ recs, const uint64_t *__restrict__ rec_off, uint32_t k,
                                                        const uint64_t *__restrict__ kmer_off /* k-mers before every bucket */,
                                                        uint64_t *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_vals,
                                                        uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ flags,
                                                        uint32_t *__restrict__ queue /* zero at launch: the next bucket to hand out */, uint32_t n_buckets,
                                                        uint32_t start_bits, uint32_t lp, float inv_dup,
                                                        const uint64_t *__restrict__ fine_region = nullptr, const uint32_t *__restrict__ fine_cap = nullptr,
                                                        const uint32_t *__restrict__ fine_cnt = nullptr,
                                                        const uint32_t *__restrict__ redo_list = nullptr, const uint32_t *__restrict__ redo_cnt = nullptr) {
  // redo_list / redo_cnt (behind sk_reduce2_kernel, kmi_reduce2.h): the queue hands out positions of this list of bucket numbers
  // instead of the buckets 0 .. n_buckets - 1; an empty list ends the launch at once
  if (redo_cnt) { n_buckets = __hip_atomic_load(redo_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (n_buckets == 0u) return; }
  auto bid = [&](uint32_t q) -> uint32_t { return redo_list ? redo_list[q] : q; };
  // fine_region / fine_cap / fine_cnt (the slack scatter's layout): bucket b's records are fine_cnt[b] records from
  // fine_region[b >> 7] + (b & 127) * fine_cap[b >> 7]; null: records [rec_off[b], rec_off[b + 1])
  auto records_of = [&](uint32_t bb, uint64_t &lo, uint64_t &hi) {
    if (fine_cnt) {   // (a bucket that outgrew its room voids the build; what was written before it did stays within the room)
      const uint32_t cap = fine_cap[bb >> 7], cnt = fine_cnt[bb];
      lo = fine_region[bb >> 7] + (uint64_t)(bb & 127u) * cap; hi = lo + (cnt < cap ? cnt : cap);
    }
    else { lo = rec_off[bb]; hi = rec_off[bb + 1]; }
  };
  using T = SkTab2<OWN_>;
  constexpr int NWAVES = T::NWAVES;
  constexpr uint64_t W1_INIT = ~0ull;   // never a record's second word (its top three bits are zero)
  enum { C_DIST = 0, C_OVF = 1, C_SPC = 2, C_SPS = 3, C_EMIT = 4, C_SP = 5, C_OVN = 6, C_T1N = 8, C_LVL = 9, C_NEXT = 10 /* and 11 */ };
  __shared__ uint64_t s_tk[T::S2];
  __shared__ uint32_t s_tv[T::S2];
  __shared__ ulonglong2 s_r[T::S1];
  __shared__ uint32_t s_rc[T::S1];
  __shared__ ulonglong2 s_ovf[kSkOvf];
  __shared__ uint64_t s_missq[NWAVES * kMissQ];
  __shared__ uint32_t s_missw[NWAVES * kMissQ];
  __shared__ uint8_t s_own[NWAVES * T::OWN];
  __shared__ uint32_t s_ctl[12];
  __shared__ uint32_t s_stack[320];
  const uint32_t lane = lane_id(), wv = wave_id();
  lds_u64_t *const tkeys = (lds_u64_t *)s_tk;
  lds_u32_t *const tvals = (lds_u32_t *)s_tv;
  lds_u32_t *const tdist = (lds_u32_t *)&s_ctl[C_DIST];
  lds_u32_t *const tovf = (lds_u32_t *)&s_ctl[C_OVF];
  uint64_t *const mq = s_missq + wv * kMissQ;
  uint32_t *const mw = s_missw + wv * kMissQ;
  uint8_t *const wown = s_own + wv * T::OWN;
  const uint32_t kb = 2u * k;                                   // 34 .. 64
  const uint32_t kmask_hi = kb >= 64u ? 0xffffffffu : ((1u << (kb - 32u)) - 1u);
  const uint32_t pad = 64u - kb;                                // 0 .. 30
  const bool use_t1_known = inv_dup > 0.f;
  // ---- once per workgroup: clean tables, clean marks, the first bucket
  for (uint32_t i = threadIdx.x; i < (uint32_t)(NWAVES * T::OWN / 4); i += T::NT) reinterpret_cast<uint32_t *>(s_own)[i] = 0;
  for (uint32_t i = threadIdx.x; i < (uint32_t)T::S2; i += T::NT) { s_tk[i] = kEmptyKey; s_tv[i] = 0; }
  for (uint32_t i = threadIdx.x; i < (uint32_t)T::S1; i += T::NT) { s_r[i] = make_ulonglong2(kEmptyKey, W1_INIT); s_rc[i] = 0; }
  for (uint32_t i = threadIdx.x; i < (uint32_t)(NWAVES * kMissQ); i += T::NT) s_missq[i] = 0;   // (to_table2's idle compare-and-swaps land here: never the empty marker)
  if (threadIdx.x < 12) s_ctl[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_ctl[C_NEXT] = atomicAdd(queue, 1u);
  lds_barrier();
  uint32_t b = __builtin_amdgcn_readfirstlane(s_ctl[C_NEXT]);
  uint32_t par = 1;                      // which of the two "next bucket" words this bucket publishes
  bool pf_ok = false;                    // pf / pf_rb / pf_re hold the start of bucket b
  ulonglong2 pf = make_ulonglong2(0, 0);
  uint64_t pf_rb = 0, pf_re = 0, pf_k0 = 0, pf_k1 = 0;
  // all k-mers of a batch of records (w0, w1, weight wt; n = 0: none) into the k-mer table
  uint32_t mn = 0, pending = 0, my_claims = 0, hbits = 0, hmask = 0, hval = 0;
  // one k-mer (key, weight kw; v: it is this pass's) into the k-mer table: the home slot and the one behind it in one read, a first
  // sighting with a free home slot claimed in line, everything else through the wavefront's miss queue
  auto to_table = [&](uint64_t key, uint32_t kw, bool v) {
    const uint32_t h = sk_slot_hash(key);
    const uint32_t slot = sk_slot_of(h, (uint32_t)T::CAP2);
    v = v && (h & hmask) == hval;   // (the pass's share of the hash space: mask 0 / value 0 takes everything; the slot comes from the high bits)
    if (SPECIAL && v && key == kEmptyKey) { s_ctl[C_SPS] = 1; atomicAdd(&s_ctl[C_SPC], kw); v = false; }   // (k = 32 only)
    const uint64_t c0 = __atomic_load_n(&s_tk[slot], __ATOMIC_RELAXED), c1 = __atomic_load_n(&s_tk[slot + 1u], __ATOMIC_RELAXED);
    bool hit0 = v && c0 == key;
    const bool hit1 = v && c1 == key;
    bool won = false;
    if (v && c0 == kEmptyKey) {   // first sighting with a free home slot: claimed here, in line (most first sightings are)
      const unsigned long long old = atomicCAS((unsigned long long *)&s_tk[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
      won = old == kEmptyKey;
      hit0 = won || old == key;
    }
    my_claims += won ? 1u : 0u;   // (per lane; the wavefront adds them up once per batch)
    if (hit0 || hit1) atomicAdd(&s_tv[slot + (hit0 ? 0u : 1u)], kw);
    const bool miss = v && !hit0 && !hit1;
    const unsigned long long mm = __ballot(miss);
    if (mm) {
      const uint32_t pos = mn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
      if (miss) { mq[pos] = key; mw[pos] = kw; }
      mn += (uint32_t)__popcll(mm);
      if (mn >= (uint32_t)kWave) {
        pending = sk_probe_insert(tkeys, tvals, tdist, tovf, (const lds_u64_t *)mq, (const lds_u32_t *)mw, mn - kWave, kWave, (uint32_t)T::CAP2,
                                  (uint32_t)T::S2 - 1u, (uint32_t)T::LIMIT2, pending);
        mn -= kWave;
      }
    }
  };
  // the same for the TWO k-mers of a unit, without a branch: both home slots and their neighbours are read together; a lane that has
  // nothing to claim sends its compare-and-swap to a word of its wavefront's miss queue (never the empty marker), a lane without a
  // hit adds zero. (One k-mer at a time, each behind `if`s: 270 instructions per step of which 100 moved the exec mask about; 190 so.)
  uint64_t *const cas_dummy = mq + kWave + lane;
  auto to_table2 = [&](uint64_t ka, bool va, uint64_t kc, bool vc, uint32_t kw) {
    const uint32_t hA = sk_slot_hash(ka), hC = sk_slot_hash(kc);
    const uint32_t sa = sk_slot_of(hA, (uint32_t)T::CAP2), sc = sk_slot_of(hC, (uint32_t)T::CAP2);
    va = va && (hA & hmask) == hval; vc = vc && (hC & hmask) == hval;
    if (SPECIAL) {   // (k = 32 only)
      if (va && ka == kEmptyKey) { s_ctl[C_SPS] = 1; atomicAdd(&s_ctl[C_SPC], kw); va = false; }
      if (vc && kc == kEmptyKey) { s_ctl[C_SPS] = 1; atomicAdd(&s_ctl[C_SPC], kw); vc = false; }
    }
    const uint64_t a0 = __atomic_load_n(&s_tk[sa], __ATOMIC_RELAXED), a1 = __atomic_load_n(&s_tk[sa + 1u], __ATOMIC_RELAXED);
    const uint64_t c0 = __atomic_load_n(&s_tk[sc], __ATOMIC_RELAXED), c1 = __atomic_load_n(&s_tk[sc + 1u], __ATOMIC_RELAXED);
    const bool ta = va && a0 == kEmptyKey, tc = vc && c0 == kEmptyKey;
    const unsigned long long oa = atomicCAS((unsigned long long *)(ta ? &s_tk[sa] : cas_dummy), (unsigned long long)kEmptyKey, (unsigned long long)ka);
    const unsigned long long oc = atomicCAS((unsigned long long *)(tc ? &s_tk[sc] : cas_dummy), (unsigned long long)kEmptyKey, (unsigned long long)kc);
    const bool wa_ = ta && oa == kEmptyKey, wc_ = tc && oc == kEmptyKey;
    const bool ha0 = va && (a0 == ka || wa_ || (ta && oa == ka)), hc0 = vc && (c0 == kc || wc_ || (tc && oc == kc));
    const bool ha1 = va && !ha0 && a1 == ka, hc1 = vc && !hc0 && c1 == kc;
    my_claims += (wa_ ? 1u : 0u) + (wc_ ? 1u : 0u);
    atomicAdd(&s_tv[sa + (ha1 ? 1u : 0u)], (ha0 || ha1) ? kw : 0u);
    atomicAdd(&s_tv[sc + (hc1 ? 1u : 0u)], (hc0 || hc1) ? kw : 0u);
    const bool ma = va && !ha0 && !ha1, mc = vc && !hc0 && !hc1;
    const unsigned long long mma = __ballot(ma), mmc = __ballot(mc);
    if (mma | mmc) {   // uniform
      if (ma) { const uint32_t pos = mn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mma, 0u)); mq[pos] = ka; mw[pos] = kw; }
      mn += (uint32_t)__popcll(mma);
      if (mn >= (uint32_t)kWave) {
        pending = sk_probe_insert(tkeys, tvals, tdist, tovf, (const lds_u64_t *)mq, (const lds_u32_t *)mw, mn - kWave, kWave, (uint32_t)T::CAP2,
                                  (uint32_t)T::S2 - 1u, (uint32_t)T::LIMIT2, pending);
        mn -= kWave;
      }
      if (mc) { const uint32_t pos = mn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mmc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mmc, 0u)); mq[pos] = kc; mw[pos] = kw; }
      mn += (uint32_t)__popcll(mmc);
      if (mn >= (uint32_t)kWave) {
        pending = sk_probe_insert(tkeys, tvals, tdist, tovf, (const lds_u64_t *)mq, (const lds_u32_t *)mw, mn - kWave, kWave, (uint32_t)T::CAP2,
                                  (uint32_t)T::S2 - 1u, (uint32_t)T::LIMIT2, pending);
        mn -= kWave;
      }
    }
  };
  // all k-mers of a batch of records (w0, w1, weight wt; n = 0: none) into the k-mer table. A lane takes TWO neighbouring k-mers
  // of one record (a unit): finding the record of a unit (marks, running maximum, six words over the lane crossbar) is paid once
  // for both, and the second k-mer is the first one moved on by a base -- a handful of instructions instead of a second cut
  // and reverse complement. (One k-mer per lane: 80 issue slots per k-mer, with this 60; the phase is bound by issue.)
  auto expand = [&](uint64_t w0, uint64_t w1, uint32_t wt, uint32_t n) {
    const uint32_t nu = (n + 1u) >> 1;
    const uint32_t inc = wave_inclusive_sum_dpp(nu);
    const uint32_t pre = inc - nu;
    const uint32_t total = __builtin_amdgcn_readlane(inc, kWave - 1);
    if (total == 0u) return;   // uniform
    if (nu) wown[pre] = (uint8_t)(lane + 1u);
    uint32_t carry = 0;   // record (+ 1) the previous step ended in
    const uint32_t top_sh = kb - 34u;   // where the last base of a k-mer starts in its high word (k >= 17)
    for (uint32_t g0 = 0; g0 < total; g0 += kWave) {
      const uint32_t g = g0 + lane;
      const bool act = g < total;
      uint32_t o = act ? (uint32_t)wown[g] : 0u;
      o = wave_inclusive_max_dpp(o);
      o = o > carry ? o : carry;
      carry = __builtin_amdgcn_readlane(o, kWave - 1);
      const int rl = (int)((o ? o - 1u : 0u) << 2);   // byte address of the lane that holds the record
      const uint32_t j = 2u * (g - (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)pre));
      const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)w0), a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)(w0 >> 32));
      const uint32_t a2 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)w1), a3 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)(w1 >> 32));
      const uint32_t kw = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)wt);
      const bool two = act && j + 1u <= ((a3 >> (kRecNShift - 32)) & 31u);   // k-mer j + 1 exists (the record holds n - 1)
      // k-mer j of the record: 2 k bits from bit 2 j of its 128 (j <= 31: the window starts in word 0 or 1). All of it on 32-bit
      // registers -- funnel shifts (v_alignbit) and single-word bit tricks: the 64-bit shifts this replaces issue at a quarter
      // of the rate.
      const bool w1sel = j >= 16u;
      const uint32_t bit = (2u * j) & 31u;   // (j is even: at most 28, so k-mer j + 1 starts in the same word)
      const uint32_t b0 = w1sel ? a1 : a0, b1 = w1sel ? a2 : a1, b2 = w1sel ? a3 : a2;
      const uint32_t rc_lo = __builtin_amdgcn_alignbit(b1, b0, bit);
      const uint32_t rc_hi = __builtin_amdgcn_alignbit(b2, b1, bit) & kmask_hi;   // (k >= 17: the low word is all k-mer)
      // forward strand = reverse complement of that: bit-reverse the 64 bits (the words change places), swap the two bits of
      // every base back, complement, and bring the 2 k bits down from the top (the complemented pad bits fall off below)
      const uint32_t r_hi = __builtin_bitreverse32(rc_lo), r_lo = __builtin_bitreverse32(rc_hi);
      const uint32_t s_hi = ~(((r_hi >> 1) & 0x55555555u) | ((r_hi << 1) & 0xAAAAAAAAu));
      const uint32_t s_lo = ~(((r_lo >> 1) & 0x55555555u) | ((r_lo << 1) & 0xAAAAAAAAu));
      const uint32_t fw_lo = __builtin_amdgcn_alignbit(s_hi, s_lo, pad), fw_hi = s_hi >> pad;
      // k-mer j + 1: the window one base on; its forward strand is the first one's moved up by a base, the complement of the
      // window's new last base below it
      const uint32_t rc2_lo = __builtin_amdgcn_alignbit(b1, b0, bit + 2u);
      const uint32_t rc2_hi = __builtin_amdgcn_alignbit(b2, b1, bit + 2u) & kmask_hi;
      const uint32_t fw2_lo = (fw_lo << 2) | (((rc2_hi >> top_sh) & 3u) ^ 3u);
      const uint32_t fw2_hi = __builtin_amdgcn_alignbit(fw_hi, fw_lo, 30u) & kmask_hi;
      const uint64_t rc = (uint64_t)rc_lo | ((uint64_t)rc_hi << 32), fw = (uint64_t)fw_lo | ((uint64_t)fw_hi << 32);
      const uint64_t rc2 = (uint64_t)rc2_lo | ((uint64_t)rc2_hi << 32), fw2 = (uint64_t)fw2_lo | ((uint64_t)fw2_hi << 32);
#ifndef KMI_SK_INSERT1
      to_table2(CANON ? (fw < rc ? fw : rc) : fw, act, CANON ? (fw2 < rc2 ? fw2 : rc2) : fw2, two, kw);
#else
      to_table(CANON ? (fw < rc ? fw : rc) : fw, kw, act);
      to_table(CANON ? (fw2 < rc2 ? fw2 : rc2) : fw2, kw, two);
#endif
    }
    if (nu) wown[pre] = 0;   // the marks go back to zero for the next batch
    // fill level: this batch's in-line claims go to the shared counter (one scan + one LDS add per batch of records)
    const uint32_t batch_claims = __builtin_amdgcn_readlane(wave_inclusive_sum_dpp(my_claims), kWave - 1);
    my_claims = 0;
    pending += batch_claims;
    if (pending >= 32u) {   // uniform
      if (lane == 0 && atomicAdd(&s_ctl[C_DIST], pending) + pending >= (uint32_t)T::LIMIT2) s_ctl[C_OVF] = 1;
      pending = 0;
    }
  };
  // per-wavefront phase clocks (-DKMI_SK_TIMING; printed by the host after every launch): where the 16 wavefronts of a workgroup spend
  // their time -- phase A (record table, direct expansions, waiting at the barrier), phase B (expansion, waiting), emit
#ifdef KMI_SK_TIMING
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
#define TQ_MARK(i) { const unsigned long long now_ = clock64(); acc[i] += now_ - tq; tq = now_; }
#else
#define TQ_MARK(i)
#endif
  while (b < n_buckets) {   // uniform
    uint32_t q_next = 0;
    if (threadIdx.x == 0) q_next = atomicAdd(queue, 1u);   // (stays in a register until phase A is done: nobody waits for it)
    uint64_t rb = pf_rb, re = pf_re;
    const uint32_t bkt = bid(b);
    if (!pf_ok) records_of(bkt, rb, re);
    const uint32_t n_rec = (uint32_t)(re - rb);
    const ulonglong2 *const src = reinterpret_cast<const ulonglong2 *>(recs) + rb;
    const uint32_t share = (n_rec + NWAVES - 1) / NWAVES;
    const uint32_t r_lo = wv * share < n_rec ? wv * share : n_rec;
    const uint32_t r_hi = r_lo + share < n_rec ? r_lo + share : n_rec;
    ulonglong2 first = pf;
    if (!pf_ok) { first = make_ulonglong2(0, 0); if (r_lo + lane < r_hi) first = src[r_lo + lane]; }
    const uint64_t tmp0 = pf_ok ? pf_k0 : kmer_off[bkt], tmp1 = pf_ok ? pf_k1 : kmer_off[bkt + 1];
    pf_ok = false;
    if (threadIdx.x == 0) {
      uint32_t hb = start_bits > 8u ? 8u : start_bits;
      if (use_t1_known && n_rec) {
        // a bucket expected above 88 % of what a pass takes starts one level down (a lost attempt costs a whole pass, and the
        // buckets that overflow are the large ones)
        const float pred = (float)(tmp1 - tmp0) * inv_dup;
        float room = 0.88f * (float)T::LIMIT2 * (float)(1u << hb);
        while (hb < 8u && pred > room) { ++hb; room *= 2.f; }
      }
      const uint32_t np = n_rec ? (1u << hb) : 0u;
      for (uint32_t v = 0; v < np; ++v) s_stack[v] = hb | (v << 8);
      s_ctl[C_SP] = np; s_ctl[C_EMIT] = 0; s_ctl[C_LVL] = hb;
    }
    bool first_pass = true;
    lds_barrier();
#ifdef KMI_SK_TIMING
    tq = clock64();
#endif
    while (true) {
      const uint32_t sp = s_ctl[C_SP];
      if (sp == 0) break;                       // uniform
      const uint32_t pass = s_stack[sp - 1];
      const uint32_t fbits = pass & 0xffu, fval = pass >> 8;
      lds_barrier();                            // everyone has read the stack
      if (threadIdx.x == 0) s_ctl[C_SP] = sp - 1;
      const bool use_t1 = use_t1_known ? inv_dup <= KMI_SK_T1_DUP : fbits < 2u;
      // the first three filter bits (two in a build over 8 ranks) are the records' sub-bucket bits (whole records are skipped), the
      // others come from the key's hash
      const uint32_t dead = lp > 2u ? lp - 2u : 0u, rmax = 3u - dead;
      const uint32_t rbits = fbits < rmax ? fbits : rmax, rmask = ((1u << rbits) - 1u) << dead, rval = (fval & ((1u << rbits) - 1u)) << dead;
      hbits = fbits - rbits; hmask = (1u << hbits) - 1u; hval = fval >> rbits;
      mn = 0; pending = 0; my_claims = 0;
      // ---- phase A: identical records are counted; what T1 does not take waits in the overflow list
      {
        ulonglong2 nxt = first;
        if (!first_pass) { nxt = make_ulonglong2(0, 0); if (r_lo + lane < r_hi) nxt = src[r_lo + lane]; }
        for (uint32_t r0 = r_lo; r0 < r_hi; r0 += kWave) {
          if (__atomic_load_n(&s_ctl[C_OVF], __ATOMIC_RELAXED)) break;   // this pass is lost already
          const ulonglong2 rec = nxt;
          const bool have = r0 + lane < r_hi;
          if (r0 + kWave + lane < r_hi) nxt = src[r0 + kWave + lane];   // in flight while this batch is worked on
          uint32_t n = (have && rec.y != kSkPadW1) ? ((uint32_t)(rec.y >> kRecNShift) & 31u) + 1u : 0u;
          if ((rec_hash18(rec.y) & rmask) != rval) n = 0;
          bool direct = n != 0u;   // still to be placed
          if (use_t1) {
            if (direct && rec.x != kEmptyKey) {
              // four consecutive slots in one go (independent reads): the first that holds this record takes the count, else the
              // first empty one is claimed; a second window of four for the few that find the first one taken
              uint32_t h = ((uint32_t)rec.x ^ (uint32_t)(rec.x >> 32)) * 0x9E3779B1u ^ ((uint32_t)rec.y ^ (uint32_t)(rec.y >> 32)) * 0x85EBCA6Bu;
              h ^= h >> 15;
              const uint32_t s0 = ((h >> 16) * (uint32_t)T::H1) >> 16;   // H1 home slots (+ 64 of padding)
              for (uint32_t s = s0; direct && s < s0 + 8u; s += 4u) {
                ulonglong2 e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = s_r[s + i];
                int hit = -1, free_ = -1;
#pragma unroll
                for (int i = 3; i >= 0; --i) {
                  const bool same = e[i].x == rec.x && e[i].y == rec.y, empty = e[i].x == kEmptyKey;
                  if (same) { hit = i; free_ = -1; } else if (empty) { free_ = i; hit = -1; }   // (the earliest of either kind wins)
                }
                if (hit >= 0) { atomicAdd(&s_rc[s + hit], 1u); direct = false; }
                else if (free_ >= 0) {
                  if (__atomic_load_n(&s_ctl[C_T1N], __ATOMIC_RELAXED) >= (uint32_t)T::L1) break;   // full enough
                  const unsigned long long old = atomicCAS((unsigned long long *)&s_r[s + free_].x, (unsigned long long)kEmptyKey, (unsigned long long)rec.x);
                  if (old == kEmptyKey) {   // claimed
                    __atomic_store_n(&s_r[s + free_].y, rec.y, __ATOMIC_RELAXED);
                    atomicAdd(&s_rc[s + free_], 1u);
                    atomicAdd(&s_ctl[C_T1N], 1u);
                    direct = false;
                  } else if (old == rec.x && __atomic_load_n(&s_r[s + free_].y, __ATOMIC_RELAXED) == rec.y) {
                    // lost the slot to a copy of the same record that arrived in the same step: counted with it
                    atomicAdd(&s_rc[s + free_], 1u);
                    direct = false;
                  } else break;   // lost it to another record
                }
              }
            }
            if (direct) {   // the overflow list takes it (phase B expands it with weight 1)
              const uint32_t oi = atomicAdd(&s_ctl[C_OVN], 1u);
              if (oi < (uint32_t)kSkOvf) { s_ovf[oi] = rec; direct = false; }
            }
          }
          if (__any(direct)) { TQ_MARK(0) expand(rec.x, rec.y, 1u, direct ? n : 0u); TQ_MARK(1) }
        }
      }
      TQ_MARK(0)
      if (first_pass && threadIdx.x == 0) s_ctl[C_NEXT + par] = q_next;
      lds_barrier();   // T1 and the overflow list complete; the next bucket is known
      TQ_MARK(2)
      uint32_t nbk = 0;
      if (first_pass) {   // the next bucket's range: two scalar loads that return during phase B
        nbk = __builtin_amdgcn_readfirstlane(s_ctl[C_NEXT + par]);
        if (nbk < n_buckets) { const uint32_t nbkt = bid(nbk); records_of(nbkt, pf_rb, pf_re); pf_k0 = kmer_off[nbkt]; pf_k1 = kmer_off[nbkt + 1]; }
      }
      // ---- phase B: every distinct record once with its multiplicity, then the overflow list; the slots are left empty
      if (use_t1) {
        const uint32_t n_ovf = s_ctl[C_OVN] < (uint32_t)kSkOvf ? s_ctl[C_OVN] : (uint32_t)kSkOvf;
        // every wavefront sweeps an equal share of the record slots AND an equal share of the overflow list (the list is dense, the
        // slots are half empty: whole 64-entry blocks dealt out in turn left three wavefronts with three times the k-mers of four
        // others, and a sixth of the kernel went by at the barrier behind this phase)
        constexpr uint32_t PER1 = ((uint32_t)T::S1 + NWAVES - 1) / NWAVES;
        const uint32_t per_o = (n_ovf + NWAVES - 1) / NWAVES;
        const uint32_t t_lo = wv * PER1, t_hi = t_lo + PER1 < (uint32_t)T::S1 ? t_lo + PER1 : (uint32_t)T::S1;
        const uint32_t o_lo = wv * per_o < n_ovf ? wv * per_o : n_ovf, o_hi = o_lo + per_o < n_ovf ? o_lo + per_o : n_ovf;
        const uint32_t n_mine = (t_hi - t_lo) + (o_hi - o_lo);
        for (uint32_t i0 = 0; i0 < n_mine; i0 += kWave) {
          const uint32_t i = i0 + lane;
          uint64_t w0 = kEmptyKey, w1 = 0; uint32_t wt = 0;
          if (i < t_hi - t_lo) {
            const uint32_t s = t_lo + i;
            const ulonglong2 ent = s_r[s];
            w0 = ent.x; w1 = ent.y; wt = s_rc[s];
            if (w0 != kEmptyKey) { s_r[s] = make_ulonglong2(kEmptyKey, W1_INIT); s_rc[s] = 0; }
          } else if (i < n_mine) {
            const ulonglong2 ent = s_ovf[o_lo + (i - (t_hi - t_lo))];
            w0 = ent.x; w1 = ent.y; wt = 1u;
          }
          uint32_t n = wt ? ((uint32_t)(w1 >> kRecNShift) & 31u) + 1u : 0u;
          if (__atomic_load_n(&s_ctl[C_OVF], __ATOMIC_RELAXED)) n = 0;   // lost pass: the sweep goes on only to empty the slots
          if (__any(n != 0u)) expand(w0, w1, wt, n);
        }
      }
      if (mn) {
        pending = sk_probe_insert(tkeys, tvals, tdist, tovf, (const lds_u64_t *)mq, (const lds_u32_t *)mw, 0u, mn, (uint32_t)T::CAP2, (uint32_t)T::S2 - 1u,
                                  (uint32_t)T::LIMIT2, pending);
        mn = 0;
      }
      if (pending && lane == 0 && atomicAdd(&s_ctl[C_DIST], pending) + pending >= (uint32_t)T::LIMIT2) s_ctl[C_OVF] = 1;
      if (first_pass && nbk < n_buckets) {   // the next bucket's first records, in flight during the emit sweep
        const uint32_t nn = (uint32_t)(pf_re - pf_rb), nshare = (nn + NWAVES - 1) / NWAVES;
        const uint32_t nlo = wv * nshare < nn ? wv * nshare : nn, nhi = nlo + nshare < nn ? nlo + nshare : nn;
        pf = make_ulonglong2(0, 0);
        if (nlo + lane < nhi) pf = reinterpret_cast<const ulonglong2 *>(recs)[pf_rb + nlo + lane];
        pf_ok = true;
      }
      first_pass = false;
      TQ_MARK(3)
      lds_barrier();
      TQ_MARK(4)
      const bool lost = s_ctl[C_OVF] != 0u;
      lds_barrier();   // everyone has read the verdict (the resets below overwrite it)
      if (threadIdx.x == 0) { s_ctl[C_DIST] = 0; s_ctl[C_OVF] = 0; s_ctl[C_T1N] = 0; s_ctl[C_OVN] = 0; }
      if (lost) {
        // Overflow: this pass is split -- straight to the level most buckets of this build ended at (flags[16 + L] counts the
        // buckets that finished with L filter bits) instead of one bit at a time. The table is emptied without being read.
        for (uint32_t i = threadIdx.x; i < (uint32_t)T::S2; i += T::NT) { s_tk[i] = kEmptyKey; s_tv[i] = 0; }
        if (threadIdx.x == 0) {
          s_ctl[C_SPC] = 0; s_ctl[C_SPS] = 0;
          if (fbits >= 18u) atomicOr(&flags[2], 1u);   // 3 record bits + 15 hash bits: 2^18 tables did not hold the bucket
          else {
            uint32_t best = 0, best_n = 0;
            for (uint32_t l = 0; l <= 8u; ++l) {
              const uint32_t c = __hip_atomic_load(&flags[16 + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (c > best_n) { best_n = c; best = l; }
            }
            uint32_t target = best > fbits ? best : fbits + 1u;
            if (target > fbits + 6u) target = fbits + 6u;     // (64 children at most at a time: the stack holds 320)
            uint32_t spn = s_ctl[C_SP];
            for (uint32_t v = 0; v < (1u << (target - fbits)); ++v) s_stack[spn++] = target | ((fval | (v << fbits)) << 8);
            s_ctl[C_SP] = spn;
            if (target > s_ctl[C_LVL]) s_ctl[C_LVL] = target;
          }
        }
        lds_barrier();
        continue;
      }
      // emit behind what the earlier passes left (disjoint key sets); every slot read is left empty
      {
        uint32_t *s_out = &s_ctl[C_EMIT];
        for (uint32_t s = threadIdx.x; s < (uint32_t)((T::S2 + kWave - 1) / kWave * kWave); s += T::NT) {
          uint64_t key = kEmptyKey; uint32_t val = 0;
          if (s < (uint32_t)T::S2) { key = s_tk[s]; val = s_tv[s]; }
          const bool used = key != kEmptyKey;
          const uint32_t pos = wave_alloc(s_out, used);
          if (used) { tmp_keys[tmp0 + pos] = key; tmp_vals[tmp0 + pos] = val; s_tk[s] = kEmptyKey; s_tv[s] = 0; }
        }
        lds_barrier();
        if (threadIdx.x == 0) {
          if (s_ctl[C_SPS]) {
            const uint32_t pos = atomicAdd(s_out, 1u);
            tmp_keys[tmp0 + pos] = kEmptyKey;
            tmp_vals[tmp0 + pos] = s_ctl[C_SPC];
          }
          s_ctl[C_SPC] = 0; s_ctl[C_SPS] = 0;
          s_tv[T::S2 - 1] = 0;   // (the parking slot of lost walks never holds a key; its count is dropped here)
        }
      }
      lds_barrier();
      TQ_MARK(5)
    }
    if (threadIdx.x == 0) {
      out_cnt[bkt] = s_ctl[C_EMIT];
      if (n_rec && (s_ctl[C_LVL] || start_bits)) atomicAdd(&flags[16 + (s_ctl[C_LVL] > 8u ? 8u : s_ctl[C_LVL])], 1u);
    }
    if (n_rec == 0u && threadIdx.x == 0) s_ctl[C_NEXT + par] = q_next;   // (an empty bucket never reached the pass loop)
    lds_barrier();   // everyone has left the pass loop (the next bucket's set-up rewrites the stack words)
    b = __builtin_amdgcn_readfirstlane(s_ctl[C_NEXT + par]);
    par ^= 1u;
    TQ_MARK(6)
  }
#ifdef KMI_SK_TIMING
  if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&reinterpret_cast<unsigned long long *>(flags + 48)[i], acc[i]);
#endif
}

}  // namespace kmi
