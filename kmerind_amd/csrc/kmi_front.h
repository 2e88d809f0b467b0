// kmi_front.h -- the FASTQ front end of the super-k-mer build in ONE pass over the input bytes (included by kmi_index.hip).
//
// What it replaces: fastq_scan_tiles (classify every byte, EOL bitmap, packed stream), fastq_scan_offsets (line bases),
// fastq_list (runs of k-mer windows per tile) and sk_minimizer (the walk) -- i.e. FASTQParser::get_next_record
// (fastq_loader.hpp:389-467) + KmerParser's window loop (kmer_parser.hpp:85-294) of the reference, in the formulation of
// kmi_extract.hip: a line is a maximal run of non-EOL bytes and line index % 4 is its role. Those four kernels spent 3.1 of the
// build's 9 ms issuing instructions on bytes that never become k-mers: every byte was classified into a 2-bit code although 52 %
// of a record are header, '+' and quality, and the line bookkeeping went through three passes over bitmaps.
//
// Here every WAVEFRONT owns a byte range and runs on its own (no workgroup barrier in the kernel):
//   produce   64 lanes x 64 bytes: EOL bits only (SWAR), line starts / ends ranked with one DPP scan, their positions into two
//             small rings in LDS; complete lines get their role from the line index; sequence lines become runs of windows
//   consume   as soon as 64 runs wait: one lane per run loads the read's own bytes, packs THEM (and nothing else) into 2-bit
//             complement codes -- the run's row, kept for the scatter pass -- and walks the minimizers exactly as
//             sk_minimizer did (rolling canonical m-mer, order hash, sliding minimum over W positions)
// The line index of a range's first line is INFERRED (first line that starts with '@' whose second successor starts with '+':
// the reference's own find_first_record idea, fastq_loader.hpp:269-364) and the ranges are chained afterwards (front_verify):
// the index of range r + 1 must be that of range r plus its lines. Anything the fast path is not sure about -- a marker that
// is not where the role says, sequence and quality lines of different length, an inference without exactly one candidate, a
// chain that does not close, a capacity exceeded -- raises ONE flag and the caller runs the general path (fastq_scan + list +
// minimizer), which also words the error. So this path is exact on well-formed input and silent on everything else.
#pragma once

namespace kmi {

constexpr int kFrWaves = 4, kFrThreads = kFrWaves * kWave;
constexpr uint32_t kFrRing = 256;        // line events a wavefront keeps (power of two); a 2 KB step may add half of it
constexpr uint32_t kFrRunQ = 256;        // waiting runs (power of two): 64 are taken at a time, a step adds at most 128
constexpr uint32_t kFrStep = 4096;       // bytes per produce step: 64 lanes x 64
constexpr uint32_t kFrRowDw = 12;        // a run's packed row: 192 bases (a run holds at most 128 + 31)
constexpr uint32_t kFrNone = 0xffu;      // FrRange::l0 of a range that owns no line
constexpr uint32_t kFrOverrun = 256u << 10;   // bytes a range may scan behind its own end before it hands the input over

struct FrRange { uint32_t l0, n_lines, n_runs, n_items; };
struct __attribute__((packed, aligned(1))) FrU4 { uint32_t x, y, z, w; };   // sixteen bytes at any address

// bytes per lane of a run, in dwords: seg + kmax - 1 bases
template <int W> struct FrCfg {
  static constexpr int SEG = W >= 19 ? 127 : (W >= 13 ? 80 : (W >= 11 ? 64 : 44));   // (< 128: window number 127 marks "behind the run" in the walk's entries)
  static constexpr int KMAX = W >= 19 ? 32 : (W >= 13 ? 28 : (W >= 11 ? 22 : 20));
  static constexpr int ND = (SEG + KMAX - 1 + 3) / 4;   // 40, 27, 22, 16
  static constexpr int NR = (ND + 3) / 4;               // packed words that can hold a base: 10, 7, 6, 4
};

// 0x80 in every byte of w that is '\n' or '\r' (exact): x = w ^ 0x0A.. turns them into 0x00 / 0x07; a byte is one of the two iff
// its high five bits are clear and its low three are 000 or 111, i.e. iff (x & 0xF8) | (((x & 7) + 1) & 6) is zero
__device__ __forceinline__ uint32_t eol_flags(uint32_t w) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  const uint32_t z = ((x & 0x07070707u) + 0x01010101u) & 0x06060606u;
  return zero_bytes((x & 0xF8F8F8F8u) | z);
}
// four bases -> four complement codes in the low byte; anything but A C G T (either case) counts as A, like DNA::FROM_ASCII
__device__ __forceinline__ uint32_t pack_dna4(uint32_t w, uint32_t *ok_out = nullptr) {
  const uint32_t x = w & 0xDFDFDFDFu;                          // fold case
  const uint32_t idx = (x >> 1) & 0x03030303u;                 // A 0, C 1, T 2, G 3
  const uint32_t expect = byte_perm(0u, 0x47544341u, idx);     // 'A','C','T','G'
  const uint32_t ok = zero_bytes(x ^ expect);                  // 0x80 per base byte
  if (ok_out) *ok_out = ok;
  const uint32_t lut = byte_perm(0u, 0x01000203u, idx);        // complement codes: A 3, C 2, T 0, G 1
  const uint32_t v1 = ok >> 7, vm = v1 | (v1 << 1);
  const uint32_t cc = (lut & vm) | (0x03030303u & ~vm);
  const uint32_t p1 = cc | (cc >> 6);
  return (p1 | (p1 >> 12)) & 0xFFu;
}

// 0x80 in every byte of w that is one of A C G T (either case)
__device__ __forceinline__ uint32_t dna4_ok(uint32_t w) {
  const uint32_t x = w & 0xDFDFDFDFu;
  const uint32_t idx = (x >> 1) & 0x03030303u;
  return zero_bytes(x ^ byte_perm(0u, 0x47544341u, idx));
}

template <int W, bool EDGES = false>
__global__ __launch_bounds__(kFrThreads) void sk_front_kernel(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t range_bytes, uint32_t n_ranges,
                                                             uint32_t k, bool rna, uint32_t run_cap, uint32_t item_cap, uint32_t ranges_per_group,
                                                             FrRange *__restrict__ info, uint32_t *__restrict__ run_items, uint32_t *__restrict__ rows,
                                                             uint32_t *__restrict__ items, uint32_t *__restrict__ wg_hist,
                                                             unsigned long long *__restrict__ n_windows, uint32_t *__restrict__ flags,
                                                             uint32_t r_first = 0u, uint32_t r_end = 0xffffffffu) {
  constexpr bool edges = EDGES;   // (a kernel of its own: the checks it adds cost the build that does not need them a quarter of this kernel's time)
  // EDGES (the de Bruijn node build, kmi_debruijn.h): a record also carries the base before its first and the base behind its last
  // k-mer (two 3-bit codes in the place of its last three bases: super-k-mers are cut three windows earlier), which the scatter pass
  // takes from the run's row (the bases around the run itself ride in the row's last word) -- and every base has to be one of A C G T
  // (an N is an A inside a k-mer but its own DNA16 code as a neighbour, edge_iterator.hpp:163-177: the general build decides then)
  // r_first / r_end: this launch takes the ranges [r_first, min(r_end, n_ranges)) -- a build from host memory launches the kernel
  // once per arrived chunk of the input, for the ranges whose bytes (and the bytes a range may scan behind its end) are there
  if (r_end > n_ranges) r_end = n_ranges;
  using F = FrCfg<W>;
  constexpr int CAP = kSkListCap;
  constexpr uint32_t INF = 0xffffffffu, RM = kFrRing - 1u, QM = kFrRunQ - 1u;
  __shared__ uint32_t s_S[kFrWaves][kFrRing], s_E[kFrWaves][kFrRing], s_rq[kFrWaves][kFrRunQ];
  __shared__ uint32_t s_list[kFrWaves][(CAP + 2) * kWave];   // [slot][lane]; slot 0 takes the opening dummy, slot CAP + 1 what does not fit
  __shared__ uint32_t s_hist[kFrWaves][kNumCoarse];
  const uint32_t lane = lane_id(), wv = wave_id();
  uint32_t *const S = s_S[wv], *const E = s_E[wv], *const rq = s_rq[wv], *const list = s_list[wv], *const hist = s_hist[wv];
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  const uint32_t m = k - (uint32_t)W + 1u;
  const uint32_t mmask = (m >= 16u) ? 0xffffffffu : ((1u << (2u * m)) - 1u);
  const uint32_t topsh = 2u * m - 2u;
  const uint32_t nmax = sk_nmax_of(k) - (edges ? kSkEdgeWindows : 0u);
  const uint32_t seg = (uint32_t)F::SEG;
  const uint32_t n_waves = gridDim.x * (uint32_t)kFrWaves;
#ifdef KMI_FR_TIMING
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = clock64();
#define FQ_MARK(i) { const unsigned long long now_ = clock64(); acc[i] += now_ - tq; tq = now_; }
#else
#define FQ_MARK(i)
#endif
  for (uint32_t r = r_first + blockIdx.x * (uint32_t)kFrWaves + wv; r < r_end; r += n_waves) {   // (uniform per wavefront)
    const uint64_t B = (uint64_t)r * range_bytes;
    const uint32_t len = (uint32_t)((n_bytes - B < range_bytes) ? (n_bytes - B) : range_bytes);
    for (uint32_t i = lane; i < (uint32_t)kNumCoarse; i += kWave) hist[i] = 0;
    uint32_t why = 0; bool bail = false;                               // this range gives up: the caller takes the general path
    uint32_t carry = (B == 0) ? 1u : (is_eol((uint32_t)__builtin_amdgcn_readfirstlane((int)bytes[B - 1])) ? 1u : 0u);   // EOL status of the byte before the next step
    const uint32_t eoff = carry ? 0u : 1u;           // the range starts inside a line: that line's end is not ours
    uint32_t n_starts = 0, n_ends = 0, n_owned = 0, done = 0, l0 = (r == 0) ? 0u : INF;
    if (r == 0 && __builtin_amdgcn_readfirstlane((int)bytes[0]) != (int)'@') { why |= 2u; bail = true; }      // (get_next_record refuses a partition that does not begin with '@')
    uint32_t rq_head = 0, rq_tail = 0, run_count = 0, item_count = 0;
    unsigned long long windows = 0;                  // (lane 0's copy counts)
    uint32_t p = 0;                                  // next step's position inside the range
    bool producing = true;
    // the step's 32 bytes per lane are loaded one step ahead (a step costs a few hundred instructions: without the prefetch
    // every one of them began with a trip to HBM that nothing else in this wavefront could hide)
    FrU4 n0, n1, n2, n3;
    n0.x = n0.y = n0.z = n0.w = 0; n1 = n0; n2 = n0; n3 = n0;
    if (B + 64ull * lane + 64 <= n_bytes) {
      const uint8_t *q = bytes + B + 64ull * lane;
      n0 = *reinterpret_cast<const FrU4 *>(q); n1 = *reinterpret_cast<const FrU4 *>(q + 16);
      n2 = *reinterpret_cast<const FrU4 *>(q + 32); n3 = *reinterpret_cast<const FrU4 *>(q + 48);
    }
    // the first byte of a header / '+' line is loaded when the line completes and compared one step later (nothing waits for it)
    uint32_t mk_ch = 0, mk_want = 0;
    // EOL bits of 32 bytes: the flags (0x80 per EOL byte) of two dwords gathered into one byte by two dot products with bit
    // weights (the product sits 7 bits too high)
    auto eol32 = [](const FrU4 &a, const FrU4 &b) -> uint32_t {
      uint32_t d0 = __builtin_amdgcn_udot4(eol_flags(a.x), 0x08040201u, 0u, false);
      d0 = __builtin_amdgcn_udot4(eol_flags(a.y), 0x80402010u, d0, false);
      uint32_t d1 = __builtin_amdgcn_udot4(eol_flags(a.z), 0x08040201u, 0u, false);
      d1 = __builtin_amdgcn_udot4(eol_flags(a.w), 0x80402010u, d1, false);
      uint32_t d2 = __builtin_amdgcn_udot4(eol_flags(b.x), 0x08040201u, 0u, false);
      d2 = __builtin_amdgcn_udot4(eol_flags(b.y), 0x80402010u, d2, false);
      uint32_t d3 = __builtin_amdgcn_udot4(eol_flags(b.z), 0x08040201u, 0u, false);
      d3 = __builtin_amdgcn_udot4(eol_flags(b.w), 0x80402010u, d3, false);
      return (d0 >> 7) | (d1 << 1) | (d2 << 9) | (d3 << 17);
    };
    while (!bail) {
      // ---------------------------------------------------------------- produce
      while (producing && rq_tail - rq_head < (uint32_t)kWave && !bail) {
        const uint64_t g = B + p + 64ull * lane;
        uint32_t eol_lo = 0, eol_hi = 0;
        const FrU4 v0 = n0, v1 = n1, v2 = n2, v3 = n3;
        {
          const uint64_t gn = g + kFrStep;
          if (gn + 64 <= n_bytes) {
            const uint8_t *q = bytes + gn;
            n0 = *reinterpret_cast<const FrU4 *>(q); n1 = *reinterpret_cast<const FrU4 *>(q + 16);
            n2 = *reinterpret_cast<const FrU4 *>(q + 32); n3 = *reinterpret_cast<const FrU4 *>(q + 48);
          }
        }
        FQ_MARK(0)
        if (g + 64 <= n_bytes) {
          eol_lo = eol32(v0, v1); eol_hi = eol32(v2, v3);
        } else {
#pragma unroll 1
          for (uint32_t i = 0; i < 64u; ++i) {
            const bool e = (g + i >= n_bytes) || is_eol(bytes[g + i]);   // bytes past the end count as EOL
            if (i < 32u) eol_lo |= (e ? 1u : 0u) << i; else eol_hi |= (e ? 1u : 0u) << (i - 32u);
          }
        }
        FQ_MARK(1)
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)(eol_hi >> 31), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);   // lane 0 keeps the carry
        const uint32_t before_lo = (eol_lo << 1) | prev, before_hi = (eol_hi << 1) | (eol_lo >> 31);
        uint32_t ls_lo = ~eol_lo & before_lo, ls_hi = ~eol_hi & before_hi, le_lo = eol_lo & ~before_lo, le_hi = eol_hi & ~before_hi;
        const uint32_t base = p + 64u * lane;
        const uint32_t ns = (uint32_t)__builtin_popcount(ls_lo) + (uint32_t)__builtin_popcount(ls_hi);
        const uint32_t ne = (uint32_t)__builtin_popcount(le_lo) + (uint32_t)__builtin_popcount(le_hi);
        if (__any(ns > 8u || ne > 8u)) { why |= 4u; bail = true; break; }   // lines of a few bytes: not this path's input (and the packed sums below stay inside their fields)
        // owned line starts: positions below the range's length (whole lanes: a range's length is a multiple of 64 except where the
        // buffer ends, and no line starts behind its end)
        const uint32_t no = base < len ? ns : 0u;
        const uint32_t packed = ns | (ne << 10) | (no << 20);
        const uint32_t inc = wave_inclusive_sum_dpp(packed);
        const uint32_t tot = __builtin_amdgcn_readlane(inc, kWave - 1);
        const uint32_t NS = tot & 1023u, NE = (tot >> 10) & 1023u, NO = tot >> 20;
        if (NS > kFrRing / 2u || NE > kFrRing / 2u) { why |= 4u; bail = true; break; }
        uint32_t rs = n_starts + ((inc - packed) & 1023u), re = n_ends + (((inc - packed) >> 10) & 1023u);
        while (ls_lo) { S[rs & RM] = base + (uint32_t)__builtin_ctz(ls_lo); ++rs; ls_lo &= ls_lo - 1u; }
        while (ls_hi) { S[rs & RM] = base + 32u + (uint32_t)__builtin_ctz(ls_hi); ++rs; ls_hi &= ls_hi - 1u; }
        while (le_lo) { E[re & RM] = base + (uint32_t)__builtin_ctz(le_lo); ++re; le_lo &= le_lo - 1u; }
        while (le_hi) { E[re & RM] = base + 32u + (uint32_t)__builtin_ctz(le_hi); ++re; le_hi &= le_hi - 1u; }
        n_starts += NS; n_ends += NE; n_owned += NO;
        carry = (uint32_t)__builtin_amdgcn_readlane(eol_hi, kWave - 1) >> 31;   // (the builtin returns a signed int)
        p += kFrStep;
        // a range is done two complete lines behind its last byte; a range that lies inside a line of megabytes (long-read FASTQ)
        // would scan on to that line's end, and so would every other range inside the same line: work without a bound. Past
        // kFrOverrun bytes behind its end the range gives up and the general path takes the input.
        if (p > len + kFrOverrun) { why |= 32u; bail = true; break; }
        const bool at_eof = B + p > n_bytes;                 // (a step that reached past the end has produced the last line's end)
        wave_sync();                                         // the events are in the rings
        // ---- the line index of the first line (once)
        if (l0 == INF && n_starts > 0u) {
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)S[0]) >= len) l0 = kFrNone;                     // the range owns no line: nothing to do here
          else if (at_eof && n_starts < 6u) {
            // the buffer ends here: a well-formed file ends with a quality line, so the lines of this range end on index 3 (mod 4);
            // like every inferred index this one is confirmed by the chain over the ranges
            l0 = (0u - n_starts) & 3u;
          } else if (n_starts >= 6u) {
            const uint32_t nl = 6u;
            uint32_t c = 0;
            if (lane < nl) c = bytes[B + S[lane]];
            const uint32_t c2 = __shfl_down(c, 2, kWave);
            const bool cand = lane < 4u && lane + 2u < nl && c == (uint32_t)'@' && c2 == (uint32_t)'+';
            const unsigned long long cm = __ballot(cand);
            if (__popcll(cm) != 1) { why |= 8u; bail = true; break; }   // none, or more than one: the general path decides
            l0 = (4u - (uint32_t)__builtin_ctzll(cm)) & 3u;
          }
        }
        if (l0 == kFrNone) { producing = false; break; }
        // ---- lines whose end is known: roles, markers, the length rule, runs
        if (l0 != INF) {
          const uint32_t complete = n_ends >= eoff ? (n_starts < n_ends - eoff ? n_starts : n_ends - eoff) : 0u;
          for (uint32_t j0 = done; j0 < complete && !bail; j0 += kWave) {
            const uint32_t j = j0 + lane;
            const bool have = j < complete;
            uint32_t sj = 0, L = 0, role = 4;
            if (have) { sj = S[j & RM]; L = E[(j + eoff) & RM] - sj; role = (l0 + j) & 3u; }
            const bool owned = have && sj < len;
            bool bad = false;
            if (mk_want && mk_ch != mk_want) bad = true;   // the previous batch's marker
            mk_want = 0;
            if (owned && (role == 0u || role == 2u)) { mk_ch = bytes[B + sj]; mk_want = role == 0u ? (uint32_t)'@' : (uint32_t)'+'; }
            if (have && role == 3u && j >= 2u && S[(j - 2u) & RM] < len) bad = bad || (E[(j - 2u + eoff) & RM] - S[(j - 2u) & RM] != L);
            if (__any(bad)) {
              why |= 16u; bail = true; break;
            }
            uint32_t nr = 0, nwin = 0;
            if (owned && role == 1u && L >= k) { nwin = L - k + 1u; nr = (nwin + seg - 1u) / seg; }
            const uint32_t rinc = wave_inclusive_sum_dpp(nr);
            const uint32_t rtot = __builtin_amdgcn_readlane(rinc, kWave - 1);
            if ((rq_tail + rtot - rq_head > kFrRunQ || __any(have && (sj >> 24))) && rtot) { { why |= 32u; bail = true; break; } }   // a long read's many runs, or a range past 16 MB
            uint32_t o = rq_tail + rinc - nr;
            for (uint32_t i = 0; i < nr; ++i) {
              const uint32_t left = nwin - i * seg;
              rq[(o + i) & QM] = (sj + i * seg) | (((left < seg ? left : seg) - 1u) << 24);
            }
            rq_tail += rtot;
          }
          if (bail) break;
          done = complete;
          // done when the range is scanned and the two lines behind its last one are complete (their quality lines are checked here)
          if (at_eof || (p >= len && complete >= n_owned + 2u)) producing = false;
        } else if (at_eof) { producing = false; }   // (no line start at all)
        wave_sync();                                         // the queue is written; the rings may be overwritten by the next step
      }
      if (bail) break;
      FQ_MARK(2)
      // ---------------------------------------------------------------- consume
      const uint32_t avail = rq_tail - rq_head;
      if (avail == 0u) { if (!producing) break; continue; }
      const uint32_t take = avail < (uint32_t)kWave ? avail : (uint32_t)kWave;
      const bool mine = lane < take;
      const uint32_t rv = mine ? rq[(rq_head + lane) & QM] : 0u;
      rq_head += take;
      if (run_count + take > run_cap) { why |= 64u; bail = true; break; }
      const uint32_t L = mine ? (rv >> 24) + 1u : 0u;        // windows of the run
      const uint64_t g0 = B + (rv & 0xffffffu);              // its first base
      {
        const uint32_t wsum = __builtin_amdgcn_readlane(wave_inclusive_sum_dpp(L), kWave - 1);
        windows += wsum;
      }
      // ---- the run's bytes -> its packed row (complement codes, base i at bits 2 i)
      uint32_t rw[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) rw[i] = 0;
      uint32_t other = 0;   // edges: 0x80 per byte of the run that is none of A C G T
      if (mine) {
        if (g0 + 16ull * F::NR <= n_bytes) {
#pragma unroll
          for (int q = 0; q < F::NR; ++q) {
            FrU4 v = *reinterpret_cast<const FrU4 *>(bytes + g0 + 16 * q);   // (a read starts at any byte: unaligned 16-byte loads)
            if (rna) { v.x = swap_tu_dword(v.x); v.y = swap_tu_dword(v.y); v.z = swap_tu_dword(v.z); v.w = swap_tu_dword(v.w); }
            uint32_t okd[4];
            rw[q] = pack_dna4(v.x, &okd[0]) | (pack_dna4(v.y, &okd[1]) << 8) | (pack_dna4(v.z, &okd[2]) << 16) | (pack_dna4(v.w, &okd[3]) << 24);
            if (edges) {   // uniform: every byte of the run's nb bases has to be one of A C G T
              const uint32_t nbits = 8u * (L + k - 1u);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t at = 8u * (16u * (uint32_t)q + 4u * (uint32_t)e);   // first bit of this dword in the run
                const uint32_t vm = nbits >= at + 32u ? 0x80808080u : (nbits <= at ? 0u : (0x80808080u & ((1u << (nbits - at)) - 1u)));
                other |= ~okd[e] & vm;
              }
            }
          }
        } else {   // the last reads of the buffer: byte by byte, nothing read past the end
          const uint32_t nb = L + k - 1u;
#pragma unroll 1
          for (uint32_t i = 0; i < nb; ++i) {
            uint32_t c = bytes[g0 + i];
            if (rna) c = swap_tu_dword(c) & 0xffu;
            const uint32_t code = pack_dna4(c | 0x0A0A0A00u) & 3u;
            if (edges) other |= ~dna4_ok(c | 0x41414100u) & 0x80u;
            const uint32_t wi = i >> 4, sh = (i & 15u) * 2u;
#pragma unroll
            for (int q = 0; q < F::NR; ++q) rw[q] |= (wi == (uint32_t)q) ? (code << sh) : 0u;
          }
        }
      }
      if (edges) {   // uniform
        // the bases around the run: the byte before its first base and the byte behind its last one are EOLs where the read begins and
        // ends, and bases where the read goes on in another run; as 1 + base code (0: none) into the last word of the row
        if (mine) {
          const uint32_t nb = L + k - 1u;
          uint32_t lb = bytes[g0 - 1u], rb = g0 + nb < n_bytes ? (uint32_t)bytes[g0 + nb] : (uint32_t)'\n';
          if (rna) { lb = swap_tu_dword(lb) & 0xffu; rb = swap_tu_dword(rb) & 0xffu; }
          const uint32_t lc = is_eol(lb) ? 0u : 4u - (pack_dna4(lb | 0x0A0A0A00u) & 3u), rc = is_eol(rb) ? 0u : 4u - (pack_dna4(rb | 0x0A0A0A00u) & 3u);
          if (!is_eol(lb)) other |= ~dna4_ok(lb | 0x41414100u) & 0x80u;
          if (!is_eol(rb)) other |= ~dna4_ok(rb | 0x41414100u) & 0x80u;
          rw[kFrRowDw - 1] = lc | (rc << 3);
        }
        if (__any(other != 0u)) { why |= 1024u; bail = true; break; }
      }
      FQ_MARK(3)
      // the row leaves now (the walk below consumes its registers): three 16-byte stores per lane, consecutive rows
      if (mine) {
        const uint64_t ri = (uint64_t)r * run_cap + run_count + lane;
        uint4 *rp = reinterpret_cast<uint4 *>(rows + ri * kFrRowDw);
        rp[0] = make_uint4(rw[0], rw[1], rw[2], rw[3]);
        rp[1] = make_uint4(rw[4], rw[5], rw[6], rw[7]);
        rp[2] = make_uint4(rw[8], rw[9], rw[10], rw[11]);
      }
      // ---- the walk (sk_minimizer_kernel's). The row is kept ALIGNED to the block being walked: after the first m - 1 bases are
      // shifted out once, the W codes of a block are the low 2 W bits of rw[0..1], and the row moves down by W bases per block
      // (static funnel shifts) -- no indexing of the register array by a run-time value, which the compiler answers with scratch.
      // An entry is written when a super-k-mer OPENS: (low 25 bits of the minimizer's order hash) << 7 | its first window. The walk
      // has no branch: EVERY position stores its (minimizer, window) into the slot behind the last entry, and the slot pointer
      // steps on when the minimizer differs from the one before -- a store that is not followed by a step is overwritten by the
      // next one. Lengths come from the next entry's window afterwards. Positions behind a lane's last window (short reads, the
      // last block) go on writing entries; their window numbers are >= L and the pass below skips them.
      uint32_t cnt = 0;
      {
        const uint32_t nblk = mine ? (L + (uint32_t)W - 2u) / (uint32_t)W + 1u : 0u;   // m-mer positions 0 .. L + W - 2
        uint32_t R = rw[0] & mmask, Fw = sk_fwd_of(rw[0] & mmask, m);
        {
          const uint32_t sh0 = 2u * (m - 1u);   // 4 .. 30 (m <= 16)
#pragma unroll
          for (int i = 0; i < F::NR; ++i) rw[i] = __builtin_amdgcn_alignbit(rw[i + 1], rw[i], sh0);
        }
        uint32_t sprev[W + 1];
#pragma unroll
        for (int j = 0; j <= W; ++j) sprev[j] = INF;
        // (LDS byte addresses: this lane's slot 0; the last slot a pointer may rest on is CAP: what is stored behind it lands in CAP + 1)
        const uint32_t a_first = (uint32_t)(uintptr_t)((lds_u32_t *)list + lane), a_stop = a_first + (uint32_t)CAP * 4u * kWave;
        uint32_t addr = a_first;   // the slot the next entry goes to
        uint32_t prevv = 0;
        const uint32_t nblk_max = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max_dpp(nblk), kWave - 1);
        auto block = [&](auto first_block, uint32_t wbase /* window that ends at position 0 of the block (b W - (W - 1)) */) {
          constexpr bool FIRST = decltype(first_block)::value;
          const uint32_t clo = rw[0], chi = rw[1];
          {
            constexpr int WS = (2 * W) / 32, BS = (2 * W) % 32;
#pragma unroll
            for (int i = 0; i < F::NR; ++i) rw[i] = __builtin_amdgcn_alignbit(rw[i + WS + 1], rw[i + WS], BS);
          }
          uint32_t hh[W];
          uint32_t pm = INF;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            if (j > 0 || !FIRST) {
              const uint32_t c = (j < 16) ? ((clo >> (2 * (j & 15))) & 3u) : ((chi >> (2 * (j & 15))) & 3u);
              R = (R >> 2) | (c << topsh);
              Fw = ((Fw << 2) | (c ^ 3u)) & mmask;
            }
            const uint32_t h = sk_order_hash(R < Fw ? R : Fw);
            hh[j] = h;
            pm = pm < h ? pm : h;
            if (FIRST && j < W - 1) continue;                     // no window ends before position W - 1
            const uint32_t sp = sprev[j + 1];
            const uint32_t curv = sp < pm ? sp : pm;
            const uint32_t wi = wbase + (uint32_t)j < 127u ? wbase + (uint32_t)j : 127u;   // (the same in every lane)
            *(lds_u32_t *)(uintptr_t)addr = (curv << 7) | wi;
            const uint32_t step = (FIRST || curv != prevv) ? 4u * kWave : 0u;   // (window 0 opens whatever the value)
            addr = addr + step < a_stop ? addr + step : a_stop;
            prevv = curv;
          }
          sprev[W] = INF;
          sprev[W - 1] = hh[W - 1];
#pragma unroll
          for (int j = W - 2; j >= 0; --j) sprev[j] = hh[j] < sprev[j + 1] ? hh[j] : sprev[j + 1];
        };
        if (nblk_max) block(std::true_type{}, 0u - (uint32_t)(W - 1));
        for (uint32_t b = 1; b < nblk_max; ++b) block(std::false_type{}, b * (uint32_t)W - (uint32_t)(W - 1));
        cnt = (addr - a_first) / (4u * kWave);   // entries 0 .. cnt - 1 (those behind the last window included); CAP: maybe more than fit
      }
      FQ_MARK(4)
      if (__any(cnt >= (uint32_t)CAP)) { why |= 128u; bail = true; break; }
      wave_sync();
      // items in their final form ((n - 1) | 27 hash bits << 5: the 18 bucket bits and nine more below them; the items of a run cover
      // its windows without a gap, so an item's first window is the sum of the lengths before it) + the coarse counts.
      // Backwards, written from the top of the list down: a super-k-mer longer than nmax windows (one minimizer repeated: low
      // complexity) becomes several items, and what is written must not reach what has not been read yet.
      uint32_t top = (uint32_t)CAP + 2u;   // items end up in slots top .. CAP + 1
      {
        uint32_t nxt_w = L;
        const uint32_t cmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max_dpp(cnt), kWave - 1);
        uint32_t it_nx = cmax ? list[(cmax - 1u) * kWave + lane] : 0u;   // (read one turn ahead; a turn writes above the entry it works on)
        for (uint32_t jj = cmax; jj-- > 0u;) {
          const uint32_t it = it_nx;
          if (jj) it_nx = list[(jj - 1u) * kWave + lane];
          const uint32_t w0 = it & 127u;
          if (jj < cnt && w0 < L) {
            const uint32_t h27 = sk_bucket_bits27(it >> 7);
            uint32_t left = nxt_w - w0;                 // windows of this super-k-mer (>= 1)
            nxt_w = w0;
            while (left) {                               // (one turn unless the super-k-mer is longer than nmax)
              uint32_t n = left;                         // the last piece first: what nmax-window pieces leave over
              while (n > nmax) n -= nmax;
              left -= n;
              if (top <= jj || top <= 2u) { top = 0xffffffffu; left = 0; break; }   // no room (CAP items at most, none over an entry still to be read): checked below
              --top;
              atomicAdd(&hist[h27 >> 19], 1u);
              list[top * kWave + lane] = (n - 1u) | (h27 << 5);
            }
            if (top == 0xffffffffu) break;
          }
        }
      }
      if (__any(top == 0xffffffffu)) { why |= 128u; bail = true; break; }
      cnt = cnt ? (uint32_t)CAP + 2u - top : 0u;
      FQ_MARK(5)
      const uint32_t cinc = wave_inclusive_sum_dpp(cnt);
      const uint32_t ctot = __builtin_amdgcn_readlane(cinc, kWave - 1);
      if (item_count + ctot > item_cap) { why |= 256u; bail = true; break; }
      if (mine) {
        const uint32_t ex = item_count + cinc - cnt;
        uint32_t *dst = items + (uint64_t)r * item_cap + ex;
        for (uint32_t j = 0; j < cnt; ++j) dst[j] = list[(top + j) * kWave + lane];
        const uint64_t ri = (uint64_t)r * run_cap + run_count + lane;
        run_items[ri] = ex | (cnt << 26);
      }
      run_count += take;
      item_count += ctot;
      wave_sync();   // the list is read; the next batch overwrites it
      FQ_MARK(6)
    }
    if (!bail && __any(mk_want && mk_ch != mk_want)) { why |= 16u; bail = true; }   // the last batch's markers
    if (bail) { if (lane == 0) { atomicOr(&flags[9], 4u); atomicOr(&flags[10], why); } }
    wave_sync();
    if (lane == 0) {
      FrRange fr; fr.l0 = (l0 == INF) ? kFrNone : l0; fr.n_lines = (l0 == INF || l0 == kFrNone) ? 0u : n_owned; fr.n_runs = run_count; fr.n_items = item_count;
      info[r] = fr;
      if (windows) atomicAdd(n_windows, windows);
    }
    const uint32_t grp = r / ranges_per_group;
    for (uint32_t i = lane; i < (uint32_t)kNumCoarse; i += kWave) {
      const uint32_t c = hist[i];
      if (c) atomicAdd(&wg_hist[(uint64_t)grp * kNumCoarse + i], c);
    }
    wave_sync();
    FQ_MARK(7)
  }
#ifdef KMI_FR_TIMING
  if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&reinterpret_cast<unsigned long long *>(flags + 48)[i], acc[i]);
#endif
}

// S for the fused front end: the scatter pass over the runs a group of ranges left (run_items / rows / items at fixed strides per
// range). The round structure, the LDS bucket sort and the copy-out are sk_scatter_kernel's; what differs is where a lane finds
// its run: run number G of the group lies in the range whose prefix of run counts covers it, its row is read as three aligned
// 16-byte loads and starts at the run's first base.
constexpr int kFrRowLds = 11;             // a run's ten packed words (127 + 31 bases) + one: odd stride, rows on different banks
constexpr uint32_t kFrMaxGroupRanges = 128;
constexpr int kFrScThreads = 512;         // runs per round of the scatter pass ...
constexpr int kFrScItems = 7168;          // ... as long as their items fit here (about 13 per run of 120 windows); a round takes fewer runs otherwise
template <bool CANON>
__global__ __launch_bounds__(kFrScThreads, 2) void sk_scatter_rows_kernel(const FrRange *__restrict__ info, uint32_t n_ranges, uint32_t rpg, uint32_t run_cap,
                                                                         uint32_t item_cap, uint32_t k, const uint32_t *__restrict__ run_items,
                                                                         const uint32_t *__restrict__ rows, const uint32_t *__restrict__ items,
                                                                         const uint64_t *__restrict__ wg_off, uint64_t *__restrict__ out, uint32_t lp,
                                                                         bool local_fmt, const uint32_t *__restrict__ flags, uint32_t edges = 0u) {
  // edges: the record's two outside bases (sk_front_kernel) go to bits 32..37 of word 1 as 1 + base code (A C G T = 0..3 in the
  // record's own orientation; 0: the read ends there) -- see sk_edge_codes
  // local_fmt (the records go straight to this GPU's back end, no exchange): nobody reads a record's coarse bits again -- where it
  // lies says them -- so their place (bits 53..60 of word 1) and bit 61 take NINE further hash bits for sk_reduce2's bins; otherwise
  // the record keeps its 18 bucket bits as the owner will read them and bits 61..63 take three.
  // (launched before the host has looked at the front end's verdict: a front end that gave up has left tables nobody may walk)
  if (__hip_atomic_load(&flags[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  // One lane per run; the round's items are brought into LDS ONCE (sixteen bytes at a time from each run's list) and everything
  // after that -- bucket counts, the bucket sort of (run, item) references, the copy-out that assembles the records -- reads
  // them there. (Kept in registers and re-read from global memory for the copy-out, the items made this kernel wait on the
  // texture addresser: one dword from a different cache line per lane and record.)
  constexpr int NT = kFrScThreads, CAP = kSkListCap, ICAP = kFrScItems;
  static_assert(ICAP >= CAP, "a run's items must fit a round");
  __shared__ uint16_t s_stage[ICAP];
  __shared__ uint32_t s_items[ICAP + 4];    // first window | (n - 1) << 7 | the top 20 hash bits << 12
  __shared__ uint8_t s_xbits[ICAP];         // the low seven hash bits
  __shared__ uint32_t s_row[NT * kFrRowLds + 4];
  __shared__ uint32_t s_ibase[NT];      // first item of every run in s_items
  __shared__ uint8_t s_runw[NT];        // windows of every run (edges)
  __shared__ uint8_t s_runx[NT];        // the bases around every run: 1 + code of the one before | of the one behind << 3 (edges)
  __shared__ uint32_t s_cnt[kNumCoarse];
  __shared__ uint32_t s_cur[kNumCoarse];
  __shared__ uint64_t s_gbase[kNumCoarse];
  __shared__ uint32_t s_part[kNumCoarse / kWave];
  __shared__ uint32_t s_scan[NT / kWave + 2];
  __shared__ uint32_t s_pre[kFrMaxGroupRanges + 1];   // runs before every range of the group
  const bool bl = threadIdx.x < kNumCoarse;   // the lanes that keep a coarse bucket's counters
  uint64_t cursor = bl ? wg_off[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] : 0ull;
  if (bl) s_cnt[threadIdx.x] = 0;
  const uint32_t r_first = blockIdx.x * rpg;
  const uint32_t nr = r_first >= n_ranges ? 0u : (n_ranges - r_first < rpg ? n_ranges - r_first : rpg);
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t q = 0; q < nr; ++q) { s_pre[q] = acc; acc += info[r_first + q].n_runs; }
    s_pre[nr] = acc;
  }
  lds_barrier();
  const uint32_t total_runs = s_pre[nr];
  const uint32_t *const g_items = items + (uint64_t)r_first * item_cap;
  for (uint32_t rb = 0; rb < total_runs;) {
    // ---- this round's runs: as many of the next NT as their items fit
    const uint32_t G = rb + threadIdx.x;
    uint32_t ri = 0, q = 0;
    uint64_t run = 0;
    if (G < total_runs) {
      for (uint32_t i = 1; i < nr; ++i) q += (G >= s_pre[i]) ? 1u : 0u;
      run = (uint64_t)(r_first + q) * run_cap + (G - s_pre[q]);
      ri = run_items[run];
    }
    uint32_t cnt = ri >> 26;
    uint32_t total;
    const uint32_t pre = block_exclusive_scan<uint32_t>(cnt, s_scan, &total);
    const bool taken = G < total_runs && pre + cnt <= (uint32_t)ICAP;   // (a prefix of the lanes: the sums only grow)
    const uint32_t n_taken = (uint32_t)__syncthreads_count(taken);     // >= 1: one run's items always fit
    if (!taken) cnt = 0;
    if (taken) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(rows + run * kFrRowDw);
      const uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
      uint32_t *row = s_row + threadIdx.x * kFrRowLds;
      row[0] = q0.x; row[1] = q0.y; row[2] = q0.z; row[3] = q0.w; row[4] = q1.x; row[5] = q1.y; row[6] = q1.z; row[7] = q1.w;
      row[8] = q2.x; row[9] = q2.y; row[10] = 0;   // (words 10 and 11 of the row in memory hold no base)
      if (edges) s_runx[threadIdx.x] = (uint8_t)q2.w;   // (... word 11 the two bases around the run, sk_front_kernel)
      // the items: sixteen bytes at a time (a list starts at any dword; what is read behind the run's items is not used). An item's
      // first window is the sum of the lengths before it.
      const uint32_t *src = g_items + (q * item_cap + (ri & 0x3ffffffu));
      uint32_t woff = 0;
#pragma unroll
      for (int v4 = 0; v4 < CAP / 4; ++v4) {
        if ((uint32_t)(4 * v4) < cnt) {
          const FrU4 v = *reinterpret_cast<const FrU4 *>(src + 4 * v4);
          const uint32_t it[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((uint32_t)(4 * v4 + e) < cnt) {
              const uint32_t n1 = it[e] & 31u, h27 = it[e] >> 5;
              s_items[pre + 4 * v4 + e] = woff | (n1 << 7) | ((h27 >> 7) << 12);
              s_xbits[pre + 4 * v4 + e] = (uint8_t)(h27 & 127u);
              atomicAdd(&s_cnt[h27 >> 19], 1u);
              woff += n1 + 1u;
            }
        }
      }
    }
    s_ibase[threadIdx.x] = pre;
    if (edges) {   // uniform: the run's windows = the sum of its items' lengths
      uint32_t wsum = 0;
      for (uint32_t j = 0; j < cnt; ++j) wsum += ((s_items[pre + j] >> 7) & 31u) + 1u;
      s_runw[threadIdx.x] = (uint8_t)wsum;
    }
    lds_barrier();
    // ---- bucket offsets of the round
    uint32_t c = 0, inc = 0;
    if (bl) {
      c = s_cnt[threadIdx.x];
      s_cnt[threadIdx.x] = 0;
      inc = wave_inclusive_sum_dpp(c);
      if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
    }
    lds_barrier();
    if (bl) {
      uint32_t wpre = 0;
#pragma unroll
      for (uint32_t w = 0; w < kNumCoarse / kWave; ++w) wpre += (w < wave_id()) ? s_part[w] : 0u;
      const uint32_t lo = wpre + inc - c;
      s_cur[threadIdx.x] = lo;
      s_gbase[threadIdx.x] = cursor - lo;
      cursor += c;
    }
    lds_barrier();
    // ---- references (run << 5 | item) in bucket order
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t item = s_items[pre + j];
      s_stage[atomicAdd(&s_cur[item >> 24], 1u)] = (uint16_t)((threadIdx.x << 5) | j);
    }
    lds_barrier();
    // ---- copy-out: every lane assembles the record of one sorted position
    const uint32_t n_items = pre + cnt;   // (of the last taken lane: the round's total)
    const uint32_t round_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cur[kNumCoarse - 1]);   // the last bucket's end = all items of the round
    (void)n_items;
    for (uint32_t s = threadIdx.x; s < round_items; s += NT) {
      const uint32_t e = s_stage[s], rl = e >> 5, j = e & 31u;
      const uint32_t item = s_items[s_ibase[rl] + j];
      const uint32_t h27 = ((item >> 12) << 7) | (uint32_t)s_xbits[s_ibase[rl] + j], n1 = (item >> 7) & 31u;
      // the bucket bits as the owner will read them: the lp rank bits shifted out, further hash bits shifted in below; three more
      // (nine in the local format, which gives up the coarse bits) above them for the bins of sk_reduce2
      const uint32_t hsh = (h27 << lp) & 0x7ffffffu;
      uint32_t hs = hsh >> 9, top3 = (hsh >> 6) & 7u;
      if (local_fmt) { hs = (hs & 0x3ffu) | (((hsh >> 1) & 0xffu) << 10); top3 = hsh & 1u; }
      uint64_t w0, w1;
      bool flipped = false;
      sk_assemble_row<CANON>(s_row + rl * kFrRowLds, 2u * (item & 127u), k + n1, n1, hs, w0, w1, &flipped);
      w1 |= (uint64_t)top3 << 61;
      if (edges) {   // uniform
        const uint32_t *row = s_row + rl * kFrRowLds;
        const uint32_t wfirst = item & 127u, after = wfirst + n1 + k;   // base index behind the last k-mer
        auto cc = [&](uint32_t i) -> uint32_t { return (row[i >> 4] >> (2u * (i & 15u))) & 3u; };   // complement code of the run's base i
        // (the row holds complement codes in read direction: base code = 3 - cc; a record that travels reverse-complemented sees the
        // complements of its neighbours, on the other side)
        const uint32_t runx = s_runx[rl];
        const uint32_t lc = wfirst ? 1u + (3u - cc(wfirst - 1u)) : (runx & 7u);
        const uint32_t rc = wfirst + n1 + 1u < (uint32_t)s_runw[rl] ? 1u + (3u - cc(after)) : (runx >> 3);
        const uint32_t left = flipped ? (rc ? 5u - rc : 0u) : lc, right = flipped ? (lc ? 5u - lc : 0u) : rc;
        w1 |= (uint64_t)(left | (right << 3)) << kRecEdgeShift;
      }
      reinterpret_cast<ulonglong2 *>(out)[s_gbase[h27 >> 19] + s] = make_ulonglong2(w0, w1);
    }
    rb += n_taken;
    lds_barrier();   // the stage, the items and the run tables are done with; the counters are clear
  }
}

// the chain of the ranges' line indices: range r + 1 starts where range r ended (mod 4); totals: lines, runs, items
__global__ __launch_bounds__(1024) void sk_front_verify_kernel(const FrRange *__restrict__ info, uint32_t n_ranges, uint32_t ranges_per_group,
                                                             uint64_t *__restrict__ group_runs /* [groups]: runs of every group */,
                                                             uint64_t *__restrict__ totals /* [0] lines [1] runs [2] items */, uint32_t *__restrict__ flags) {
  __shared__ uint64_t s_scan[1024 / 64 + 2];
  __shared__ uint32_t s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  lds_barrier();
  uint64_t carry = 0, runs = 0, its = 0;
  for (uint32_t r0 = 0; r0 < n_ranges; r0 += 1024) {
    const uint32_t r = r0 + threadIdx.x;
    FrRange fr; fr.l0 = kFrNone; fr.n_lines = 0; fr.n_runs = 0; fr.n_items = 0;
    if (r < n_ranges) fr = info[r];
    uint64_t tot;
    const uint64_t before = carry + block_exclusive_scan<uint64_t>((uint64_t)fr.n_lines, s_scan, &tot);
    if (fr.l0 != kFrNone && fr.n_lines && ((uint32_t)before & 3u) != fr.l0) s_bad = 1;
    carry += tot;
    uint64_t t2;
    (void)block_exclusive_scan<uint64_t>((uint64_t)fr.n_runs, s_scan, &t2); runs += t2;
    (void)block_exclusive_scan<uint64_t>((uint64_t)fr.n_items, s_scan, &t2); its += t2;
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    totals[0] = carry; totals[1] = runs; totals[2] = its;
    if (s_bad) atomicOr(&flags[9], 4u);
  }
  // runs per group (what a scatter workgroup will walk)
  const uint32_t groups = (n_ranges + ranges_per_group - 1) / ranges_per_group;
  for (uint32_t g = threadIdx.x; g < groups; g += blockDim.x) {
    uint64_t s = 0;
    for (uint32_t r = g * ranges_per_group; r < (g + 1) * ranges_per_group && r < n_ranges; ++r) s += info[r].n_runs;
    group_runs[g] = s;
  }
}

}  // namespace kmi
