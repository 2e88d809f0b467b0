// kmi_extract.hip -- FASTQ bytes -> k-mer tuples on the device.
//
// Replaces the reference's per-read iterator stack
//   SequencesIterator (src/io/sequence_iterator.hpp:96-300)
//     -> FASTQParser::get_next_record (src/io/fastq_loader.hpp:389-467)
//     -> KmerParser / KmerCountTupleParser::operator() (src/io/kmer_parser.hpp:85-294,909-1083)
//     -> KmerGenerationIterator / Kmer::nextFromChar (src/common/kmer_iterators.hpp:67-116,
//        src/common/kmer.hpp:731-741)
// driven by KmerFileHelper::read_block_old (src/io/kmer_file_helper.hpp:110-186).
//
// Formulation (MI355X-first, no per-read sequential walk):
//   * A "line" is a maximal run of non-EOL bytes; FASTQ gives line index % 4 the role
//     (0 '@' header, 1 sequence, 2 '+', 3 quality) because get_next_record always consumes
//     4 lines and skips any number of EOLs between them.
//   * pass 1 (scan_tiles): every tile counts its line starts and, for each of the 4 possible
//     role phases, how many k-windows without an EOL start in it.
//   * pass 2 (scan over tiles, one workgroup): line base and output offset per tile.
//   * pass 3 (extract): the tile's bytes become a packed stream of COMPLEMENT codes in LDS
//     (2 or 3 bits per base). A little-endian k-window over that stream is the reverse
//     complement k-mer; the forward k-mer is its bit/group reversal (v_bfrev). Valid windows
//     are ranked with a workgroup scan, staged in LDS in file order and written out with
//     fully coalesced stores.
// Every thread owns C consecutive bytes (16 for one-word k-mers): one 16-byte global load,
// all later indexing is compile-time so nothing spills.
#include <math.h>

#include "kmi_extract.h"

namespace kmi {

// ---------------------------------------------------------------------------
// pass 1
// ---------------------------------------------------------------------------
// Cfg: ExCfg<NW, BITS> or, for 2-bit one-word shapes, the same 8 KB tile with 32 bytes per lane (ScanCfg)
template <typename Cfg>
__global__ __launch_bounds__((Cfg::NT)) void fastq_scan_tiles_kernel(const uint8_t *__restrict__ bytes,
                                                                               uint64_t n_bytes, uint32_t k,
                                                                               uint8_t *__restrict__ pk_eol,
                                                                               uint8_t *__restrict__ pk_stream,
                                                                               TileInfo *__restrict__ info,
                                                                               uint32_t *__restrict__ flags,
                                                                               const uint8_t *__restrict__ brk, uint64_t n_cover, bool rna) {
  __shared__ uint32_t s_eol[Cfg::EOL_DW];
  __shared__ uint32_t s_brk[Cfg::EOL_DW];
  __shared__ uint32_t s_scan[Cfg::NT / 64 + 2];
  __shared__ uint32_t s_cnt[3];
  __shared__ uint32_t s_last[4];
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 4) s_last[threadIdx.x] = 0;
  uint32_t dw[Cfg::C / 4], eol, ls, lbl, ltot;
  tile_front_bytes<Cfg>(bytes, n_bytes, blockIdx.x, pk_eol, pk_stream, s_eol, s_scan, dw, eol, ls, lbl, ltot, rna);

  // k-windows are counted against the break bits: EOLs, plus the N positions of a sequence filter
  uint64_t e[Cfg::NE];
  load_eol_view<Cfg>(tile_break_image<Cfg>(brk, n_cover, blockIdx.x, s_eol, s_brk), threadIdx.x, e);
  smear_right<Cfg::NE>(e, k);
  const uint32_t cand = ~(uint32_t)e[0] & Cfg::CMASK;

  // windows by local phase: 16-bit fields (phase 0,1) and (phase 2,3); marker bits by local phase
  uint32_t lo = 0, hi = 0;
  {
    uint32_t cur = lbl, start = 0, rest = ls;
    while (true) {
      uint32_t q = rest ? (uint32_t)__builtin_ctz(rest) : (uint32_t)Cfg::C;
      uint32_t seg = (q >= 32u ? 0xffffffffu : ((1u << q) - 1u)) & ~((1u << start) - 1u) & Cfg::CMASK;
      uint32_t c = (uint32_t)__builtin_popcount(cand & seg);
      uint32_t ph = cur & 3u;
      if (ph < 2) lo += c << (16 * ph); else hi += c << (16 * (ph - 2));
      if (!rest) break;
      // the line that starts at q has local index `cur` (fastq_loader.hpp:421-422,437-438). Line starts are
      // sparse (one per ~70 bytes), so the few lanes that have one talk to LDS directly.
      const uint32_t ch = (dw[q >> 2] >> (8 * (q & 3))) & 0xffu;
      const uint32_t mk = ((ch != '@') ? 1u : 0u) | ((ch != '+') ? 16u : 0u);
      atomicOr(&s_cnt[2], mk << (cur & 3u));
      atomicMax(&s_last[cur & 3u], threadIdx.x * Cfg::C + q + 1u);
      cur += 1; start = q; rest &= rest - 1u;
    }
  }
  // get_next_record refuses a partition that does not begin with '@' (fastq_loader.hpp:392-393)
  if (blockIdx.x == 0 && threadIdx.x == 0 && (dw[0] & 0xffu) != '@') atomicOr(&flags[0], 1u);
  lo = wave_reduce_sum(lo); hi = wave_reduce_sum(hi);
  if (lane_id() == 0) { atomicAdd(&s_cnt[0], lo); atomicAdd(&s_cnt[1], hi); }
  lds_barrier();
  if (threadIdx.x == 0) {
    TileInfo ti;
    ti.lines = ltot;
    ti.win[0] = s_cnt[0] & 0xffffu; ti.win[1] = s_cnt[0] >> 16;
    ti.win[2] = s_cnt[1] & 0xffffu; ti.win[3] = s_cnt[1] >> 16;
    ti.marks = s_cnt[2];
    for (int r = 0; r < 4; ++r) ti.last[r] = s_last[r];
    info[blockIdx.x] = ti;
  }
}

// ---------------------------------------------------------------------------
// pass 2: offsets over tiles. A tile maps the incoming line count b (mod 4) to b + lines and
// contributes win[(2 - b) & 3] tuples, so a run of tiles is summarised by (lines, tuples for
// each of the 4 possible incoming phases) and the summaries compose associatively:
//   (a) every block of 1024 tiles reduces to one summary,
//   (b) one workgroup scans the block summaries,
//   (c) every block re-scans its tiles from the block's base.
// totals[0] = lines, totals[1] = tuples, totals[2] = sequences (lines with index % 4 == 1)
// ---------------------------------------------------------------------------
struct TileSum { uint64_t lines; uint64_t cnt[4]; uint64_t hdr[4]; };   // hdr: 1 + byte position of the last header line start, by incoming phase

__global__ __launch_bounds__(1024) void fastq_offsets_reduce_kernel(const TileInfo *__restrict__ info, uint64_t n_tiles, uint32_t tile_bytes,
                                                                   TileSum *__restrict__ sums) {
  __shared__ uint32_t s_scan[1024 / 64 + 2];
  __shared__ uint32_t s_cnt[4];
  __shared__ unsigned long long s_hdr[4];
  if (threadIdx.x < 4) s_hdr[threadIdx.x] = 0ull;
  const uint64_t t = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  TileInfo ti; ti.lines = 0; ti.win[0] = ti.win[1] = ti.win[2] = ti.win[3] = 0; ti.marks = 0; ti.last[0] = ti.last[1] = ti.last[2] = ti.last[3] = 0;
  if (t < n_tiles) ti = info[t];
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  uint32_t tot;
  const uint32_t lb = block_exclusive_scan<uint32_t>(ti.lines, s_scan, &tot);
#pragma unroll
  for (uint32_t r = 0; r < 4; ++r) {
    uint32_t c = wave_reduce_sum(ti.win[(2u - r - lb) & 3u]);
    if (lane_id() == 0) atomicAdd(&s_cnt[r], c);
    // header lines of this tile if r lines precede the block: local index == -(r + lb) mod 4
    const uint32_t l = ti.last[(0u - r - lb) & 3u];
    unsigned long long h = l ? (unsigned long long)t * tile_bytes + l : 0ull;
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) { unsigned long long o = __shfl_xor(h, d, kWave); h = o > h ? o : h; }
    if (lane_id() == 0 && h) atomicMax(&s_hdr[r], h);
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    TileSum o; o.lines = tot;
    for (int r = 0; r < 4; ++r) o.hdr[r] = s_hdr[r];
    for (int r = 0; r < 4; ++r) o.cnt[r] = s_cnt[r];
    sums[blockIdx.x] = o;
  }
}

// in place: sums[b] becomes {lines before block b, tuples before block b, -, -, -}
__global__ __launch_bounds__(1024) void fastq_offsets_scan_kernel(TileSum *__restrict__ sums, uint64_t n_blocks,
                                                                 uint64_t n_tiles, uint64_t *__restrict__ out_off,
                                                                 uint64_t *__restrict__ totals) {
  __shared__ uint64_t s_scan[1024 / 64 + 2];
  uint64_t carry_lines = 0, carry_cnt = 0, carry_hdr = 0;
  for (uint64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
    const uint64_t b = b0 + threadIdx.x;
    TileSum ts; ts.lines = 0; ts.cnt[0] = ts.cnt[1] = ts.cnt[2] = ts.cnt[3] = 0; ts.hdr[0] = ts.hdr[1] = ts.hdr[2] = ts.hdr[3] = 0;
    if (b < n_blocks) ts = sums[b];
    uint64_t tl, tc;
    const uint64_t lb = carry_lines + block_exclusive_scan<uint64_t>(ts.lines, s_scan, &tl);
    const uint64_t c = ts.cnt[(uint32_t)lb & 3u];
    const uint64_t cb = carry_cnt + block_exclusive_scan<uint64_t>(c, s_scan, &tc);
    uint64_t th;
    uint64_t hb = block_exclusive_max<uint64_t>(ts.hdr[(uint32_t)lb & 3u], s_scan, &th);
    hb = hb > carry_hdr ? hb : carry_hdr;
    if (b < n_blocks) { ts.lines = lb; ts.cnt[0] = cb; ts.hdr[0] = hb; sums[b] = ts; }
    carry_lines += tl; carry_cnt += tc; carry_hdr = th > carry_hdr ? th : carry_hdr;
  }
  if (threadIdx.x == 0) {
    out_off[n_tiles] = carry_cnt;
    totals[0] = carry_lines;
    totals[1] = carry_cnt;
    totals[2] = (carry_lines + 2) / 4;
  }
}

__global__ __launch_bounds__(1024) void fastq_offsets_apply_kernel(const TileInfo *__restrict__ info, uint64_t n_tiles, uint32_t tile_bytes,
                                                                  const TileSum *__restrict__ sums, uint64_t *__restrict__ hdr_base,
                                                                  uint32_t *__restrict__ line_base,
                                                                  uint64_t *__restrict__ out_off,
                                                                  uint32_t *__restrict__ flags) {
  __shared__ uint32_t s_scan[1024 / 64 + 2];
  const uint64_t t = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  TileInfo ti; ti.lines = 0; ti.win[0] = ti.win[1] = ti.win[2] = ti.win[3] = 0; ti.marks = 0; ti.last[0] = ti.last[1] = ti.last[2] = ti.last[3] = 0;
  if (t < n_tiles) ti = info[t];
  const TileSum base = sums[blockIdx.x];
  const uint32_t lb = (uint32_t)base.lines + block_exclusive_scan<uint32_t>(ti.lines, s_scan, (uint32_t *)nullptr);
  const uint32_t c = ti.win[(2u - lb) & 3u];
  const uint32_t co = block_exclusive_scan<uint32_t>(c, s_scan, (uint32_t *)nullptr);
  __shared__ uint64_t s_scan64[1024 / 64 + 2];
  const uint32_t hl = ti.last[(0u - lb) & 3u];
  uint64_t hb = block_exclusive_max<uint64_t>(hl ? t * tile_bytes + hl : 0ull, s_scan64, (uint64_t *)nullptr);
  hb = hb > base.hdr[0] ? hb : base.hdr[0];
  if (t < n_tiles) {
    hdr_base[t] = hb;   // 1 + byte position of the last record start before this tile (0 = none)
    line_base[t] = lb; out_off[t] = base.cnt[0] + co;
    // header lines are the ones with (lb + local index) % 4 == 0, '+' lines == 2
    uint32_t bad = 0;
    if ((ti.marks >> ((0u - lb) & 3u)) & 1u) bad |= 1u;
    if ((ti.marks >> (4u + ((2u - lb) & 3u))) & 1u) bad |= 2u;
    if (bad) atomicOr(&flags[0], bad);
  }
}

// ---------------------------------------------------------------------------
// FASTQParser::get_next_record refuses a record whose sequence and quality lines both exist but differ in
// length (fastq_loader.hpp:454-463). Lines are maximal non-EOL runs, so the check is pure EOL-bitmap work: one
// wavefront per scan tile finds the line starts in the tile's bitmap words, and every line whose index says
// "quality" (index % 4 == 3, from the scan's line bases) measures itself forward and the sequence line two lines
// back by bit scans over the global bitmap (a handful of cached word loads per record).
// ---------------------------------------------------------------------------
// Dense form: every lane takes a few words of the window (tile + context), line starts (non-EOL after EOL) and
// line ends (EOL after non-EOL) are ranked with two wave scans and their positions land in two small LDS arrays, so
// line j of the window is [S[j], E[j + eoff]) and a quality line is compared with the line two ranks before it by
// plain array reads. Only what the window cannot answer (a record longer than the context, > CAP lines in a window)
// takes the bit-scan path above.
template <int TILE>
__global__ __launch_bounds__(256) void fastq_check_lengths_kernel(const uint32_t *__restrict__ eolw, uint64_t n_words, uint64_t n_tiles,
                                                                 const uint32_t *__restrict__ line_base, uint32_t *__restrict__ flags) {
  constexpr int WORDS = TILE / 32;
  constexpr int CTX = 32;                      // words of context on either side of the tile (1 KB)
  constexpr int WIN = WORDS + 2 * CTX, WPL = WIN / kWave;
  constexpr int CAP = 512;                     // lines per window handled by the dense path
  static_assert(WIN % kWave == 0, "window words map onto the lanes of a wavefront");
  static_assert(WIN * 32 < 65536, "window positions fit 16 bits");
  __shared__ uint32_t s_img[256 / kWave][WIN];
  __shared__ uint16_t s_S[256 / kWave][CAP], s_E[256 / kWave][CAP];
  uint32_t *img = s_img[wave_id()];
  uint16_t *S = s_S[wave_id()], *E = s_E[wave_id()];
  const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x / kWave);
  const uint32_t lane = lane_id();
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  bool bad = false;
  for (uint64_t t = (uint64_t)blockIdx.x * (blockDim.x / kWave) + wave_id(); t < n_tiles; t += n_waves) {
    const int64_t gw0 = (int64_t)(t * WORDS) - CTX;   // global word index of window word 0 (negative before the buffer)
    uint32_t w[WPL];
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      const int64_t g = gw0 + (int64_t)(lane * WPL + i);
      w[i] = (g >= 0 && (uint64_t)g < n_words) ? eolw[g] : 0xffffffffu;
      img[lane * WPL + i] = w[i];
    }
    uint32_t prev = __shfl_up(w[WPL - 1] >> 31, 1, kWave);
    const uint32_t prev_win = (gw0 > 0) ? (eolw[gw0 - 1] >> 31) : 1u;   // EOL status of the byte before the window
    if (lane == 0) prev = prev_win;
    uint32_t ls[WPL], le[WPL], ns = 0, ne = 0, nsl = 0, nst = 0;
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      const uint32_t before = (w[i] << 1) | prev;      // EOL status of each position's predecessor
      ls[i] = ~w[i] & before;
      le[i] = w[i] & ~before;
      prev = w[i] >> 31;
      const uint32_t c = (uint32_t)__builtin_popcount(ls[i]);
      const uint32_t ww = lane * WPL + i;
      ns += c; ne += (uint32_t)__builtin_popcount(le[i]);
      nsl += (ww < (uint32_t)CTX) ? c : 0u;
      nst += (ww < (uint32_t)(CTX + WORDS)) ? c : 0u;
    }
    // ranks: starts in the low half, ends in the high half of one scan
    const uint32_t packed = ns | (ne << 16);
    const uint32_t inc = wave_inclusive_scan(packed);
    const uint32_t tot = __shfl(inc, kWave - 1, kWave);
    const uint32_t NS = tot & 0xffffu, NE = tot >> 16;
    const uint32_t cnt2 = wave_reduce_sum(nsl | (nst << 16));
    const uint32_t NSL = cnt2 & 0xffffu, NST = cnt2 >> 16;
    const uint32_t lb = line_base[t];
    EolBits bm;
    bm.g = eolw; bm.n_words = n_words; bm.img = img; bm.nw = WIN; bm.w0 = (uint64_t)gw0;
    wave_sync();   // image complete (the bit-scan path reads it)
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
      for (uint32_t j = NSL + lane; j < NST; j += kWave) {
        if (((lb + (j - NSL)) & 3u) != 3u) continue;
        if (j >= 2u && j + eoff < NE) {
          const uint32_t lq = (uint32_t)E[j + eoff] - (uint32_t)S[j];
          const uint32_t lsq = (uint32_t)E[j - 2u + eoff] - (uint32_t)S[j - 2u];
          bad = bad || (lq != lsq);
        } else {
          bad = bad || fastq_lengths_differ(bm, (uint64_t)((int64_t)gw0 * 32 + (int64_t)S[j]));
        }
      }
    } else {
      // crowded window: every line start of the tile proper by bit scans
      uint32_t idx = lb + ((inc - packed) & 0xffffu) - NSL;   // line index of this lane's first start (meaningful inside the tile)
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const uint32_t ww = lane * WPL + i;
        uint32_t r = ls[i];
        while (r) {
          const uint32_t bpos = (uint32_t)__builtin_ctz(r);
          if (ww >= (uint32_t)CTX && ww < (uint32_t)(CTX + WORDS) && (idx & 3u) == 3u)
            bad = bad || fastq_lengths_differ(bm, (uint64_t)((int64_t)gw0 * 32 + (int64_t)(ww * 32u + bpos)));
          ++idx; r &= r - 1u;
        }
      }
    }
    wave_sync();   // image and event arrays are rewritten for the next tile
  }
  if (bad) atomicOr(&flags[0], 4u);
}

// ---------------------------------------------------------------------------
// pass 3: tuples in file order. Work is re-distributed over the tile's compacted window list,
// so consecutive lanes produce consecutive tuples and the stores are coalesced without staging.
// ---------------------------------------------------------------------------
// a whole record (key words, one value word) of an even number of words as 16-byte stores: half the write transactions of
// word-by-word stores with a record's stride between lanes
template <int NW>
__device__ __forceinline__ void store_record_vec(uint64_t *__restrict__ recs, uint64_t i, const uint64_t (&key)[NW], uint64_t val) {
  static_assert((NW + 1) % 2 == 0, "records of an even number of words");
  uint64_t w[NW + 1];
#pragma unroll
  for (int j = 0; j < NW; ++j) w[j] = key[j];
  w[NW] = val;
  ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(recs + i * (NW + 1));
#pragma unroll
  for (int j = 0; j < (NW + 1) / 2; ++j) dst[j] = make_ulonglong2(w[2 * j], w[2 * j + 1]);
}
template <int NW> constexpr bool kRecVec = (NW + 1) % 2 == 0;

// FASTA: the same pass over the compacted character stream (in.eol = record-start bits, in.n_bytes = characters)
template <int NW, int BITS, bool WITH_IDS>
__global__ __launch_bounds__((ExCfg<NW, BITS>::NT)) void fasta_extract_kernel(
    PackedInput in, KShape shape, bool canonical, const uint64_t *__restrict__ ids_by_rank, const uint64_t *__restrict__ out_off,
    uint64_t out_capacity, uint64_t *__restrict__ out_kmers, uint64_t *__restrict__ out_ids, uint32_t *__restrict__ flags,
    uint32_t kstride /* words between the keys of consecutive tuples */, uint32_t istride /* ... between their ids */,
    const uint8_t *__restrict__ raw_edges = nullptr /* de Bruijn tuples (kmi_debruijn.h): the input bytes; the id slot then takes the edge byte */,
    uint64_t file_offset = 0) {
  using Cfg = ExCfg<NW, BITS>;
  __shared__ uint32_t s_eol[Cfg::EOL_DW];
  __shared__ uint32_t s_stream[Cfg::STREAM_DW];
  __shared__ uint32_t s_scan[Cfg::NT / 64 + 2];
  __shared__ uint16_t s_pos[Cfg::TILE];
  uint32_t eol, ls, lbl, ltot;
  tile_front_packed<Cfg>(in, blockIdx.x, s_eol, s_stream, s_scan, eol, ls, lbl, ltot);
  const uint64_t tile0 = (uint64_t)blockIdx.x * Cfg::TILE;
  const uint32_t total = tile_window_list_from<Cfg>(chunk_valid_mask_fasta<Cfg>(s_eol, shape.k, tile0, in.n_bytes, in.n_valid), s_pos, s_scan);
  const uint64_t base = out_off[blockIdx.x];
  if (base + total > out_capacity) {
    if (threadIdx.x == 0 && total) atomicOr(&flags[1], 1u);
    return;
  }
  // records of (key words, id) at a 16-byte aligned address
  const bool vec = WITH_IDS && kstride == (uint32_t)NW + 1u && istride == kstride && out_ids == out_kmers + NW && ((uintptr_t)out_kmers & 15u) == 0;
  for (uint32_t q = threadIdx.x; q < total; q += Cfg::NT) {
    uint64_t rc[NW], fw[NW], key[NW];
    const uint32_t pos = s_pos[q];
    window_at<Cfg>(s_stream, pos, shape, rc, fw);
    select_strand<NW>(rc, fw, canonical, key);
    uint64_t idw = 0;
    if (WITH_IDS) {
      idw = ids_by_rank[tile0 + pos];
      if (raw_edges) {   // uniform
        // edge_iterator.hpp:163-177 over the record's characters (EOLs are not characters: NonEOLIter): the characters left and right
        // of the k-mer in its sequence, DNA16 codes, nothing where the sequence ends. Character r of the compacted stream sits at
        // the file position its id holds (low 40 bits); a set record-start bit at r means r has no left neighbour.
        const uint64_t r = tile0 + pos, rr = r + shape.k;
        const uint32_t *brk = reinterpret_cast<const uint32_t *>(in.eol);
        uint32_t e = 0;
        if (r > 0 && !((brk[r >> 5] >> (r & 31u)) & 1u)) e |= code_dna16(raw_edges[(ids_by_rank[r - 1] & 0xFFFFFFFFFFull) - file_offset]) << 4;
        if (rr < in.n_bytes && !((brk[rr >> 5] >> (rr & 31u)) & 1u)) e |= code_dna16(raw_edges[(ids_by_rank[rr] & 0xFFFFFFFFFFull) - file_offset]);
        idw = e;
      }
    }
    if constexpr (WITH_IDS && kRecVec<NW>) {
      if (vec) { store_record_vec<NW>(out_kmers, base + q, key, idw); continue; }   // uniform
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) out_kmers[(base + q) * kstride + w] = key[w];
    if (WITH_IDS) out_ids[(base + q) * istride] = idw;
  }
}

// valid windows per tile of the compacted stream (TileInfo.win filled for every phase)
template <int NW, int BITS>
__global__ __launch_bounds__((ExCfg<NW, BITS>::NT)) void fasta_count_tiles_kernel(PackedInput in, uint32_t k, TileInfo *__restrict__ info) {
  using Cfg = ExCfg<NW, BITS>;
  __shared__ uint32_t s_eol[Cfg::EOL_DW];
  __shared__ uint32_t s_stream[Cfg::STREAM_DW];
  __shared__ uint32_t s_scan[Cfg::NT / 64 + 2];
  uint32_t eol, ls, lbl, ltot;
  tile_front_packed<Cfg>(in, blockIdx.x, s_eol, s_stream, s_scan, eol, ls, lbl, ltot);
  const uint32_t valid = chunk_valid_mask_fasta<Cfg>(s_eol, k, (uint64_t)blockIdx.x * Cfg::TILE, in.n_bytes, in.n_valid);
  uint32_t total;
  (void)block_exclusive_scan<uint32_t>((uint32_t)__builtin_popcount(valid), s_scan, &total);
  if (threadIdx.x == 0) {
    TileInfo ti; ti.lines = 0; ti.marks = 0;
    for (int r = 0; r < 4; ++r) { ti.win[r] = total; ti.last[r] = 0; }
    info[blockIdx.x] = ti;
  }
}

// ---------------------------------------------------------------------------
// k-mer quality (PositionQualityIndex): QualityScoreSlidingWindow<.., Illumina18QualityScoreCodec<float>>
// (src/index/quality_score_iterator.hpp:67-173, src/index/quality_scores.hpp:88-341,529).
// The value is a sequential float running sum (add the entering log2-probability, subtract the
// leaving one) followed by exp2, so it is replayed in exactly that order: one lane walks one read.
// Reads are found by the extract pass: the lane that owns the first window of a read records
// (position of the read's first base, output offset of its first tuple).
// ---------------------------------------------------------------------------
__constant__ float c_qual_lut[96];
__constant__ uint64_t c_exp2_tab[32];   // T[i] = bits(2^(i/32)) - (i << 47)

// std::exp2(float) as the reference's libm computes it: the published exp2f of glibc >= 2.27
// (Szabolcs Nagy / ARM optimized routines): x = k/32 + r, 2^x = 2^(k/32) * p(r) in double,
// p(r) = C0 r^3 + C1 r^2 + C2 r + 1, one rounding to float at the end. tests/cpu/exp2f_check.c
// compares the same restatement with the host libm on 2e7 inputs.
__device__ __forceinline__ float exp2f_libm(float x, const uint64_t *__restrict__ tab /* c_exp2_tab or a copy of it */) {
  if (x <= -150.0f) return 0.0f;
  const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
  const double SHIFT = 0x1.8p+52 / 32;
  const double xd = (double)x;
  double kd = xd + SHIFT;
  const uint64_t ki = (uint64_t)__double_as_longlong(kd);
  kd -= SHIFT;
  const double r = xd - kd;
  const uint64_t t = tab[ki & 31u] + (ki << (52 - 5));
  const double sc = __longlong_as_double((long long)t);
  const double z = fma(C0, r, C1);
  const double r2 = r * r;
  double y = fma(C2, r, 1.0);
  y = fma(z, r2, y);
  y = y * sc;
  return (float)y;
}


// Sixty-four reads per WAVEFRONT, one per lane, so the sequential float chain of every read (two operations per window, in
// the reference's order) runs on all lanes at once. Everything around the chain is made wave-friendly:
//  * the quality line of a read is found by the lane itself with four bit scans over the EOL bitmap of the scan pass;
//  * the quality characters go through LDS: for every read of the batch the lanes load its characters together
//    (coalesced), chunk by chunk (kQualChunk windows), into a row per read; the row stride is an odd number of dwords, so
//    the per-lane walk over "its" row is conflict free;
//  * the values leave through a 64 x 16 LDS tile: every 16 windows the lanes write each read's 16 floats as one 64-byte line.
constexpr int kQualChunk = 32;                        // windows per chunk (a multiple of 16)
constexpr int kQualThreads = 256;
inline uint32_t qual_row_bytes(uint32_t k) { return (((kQualChunk + k + 3u) / 4u + 1u) | 1u) * 4u; }   // the aligned dwords that cover chunk + k characters at any shift; odd
inline size_t qual_lds_bytes(uint32_t k) { return (size_t)(kQualThreads / kWave) * (64u * qual_row_bytes(k) + 64u * 17u * 4u + 64u * 12u) + 96u * 4u + 32u * 8u; }

__global__ __launch_bounds__(kQualThreads) void fastq_quality_kernel(const uint8_t *__restrict__ bytes, uint64_t n_bytes,
                                                                    const uint32_t *__restrict__ eolw, uint64_t n_words, uint32_t k,
                                                                    uint32_t row_bytes, const ReadDesc *__restrict__ reads,
                                                                    const uint64_t *__restrict__ n_reads, float *__restrict__ out,
                                                                    uint64_t *__restrict__ out_rec, uint32_t qstride) {
  // out: one float per tuple; or (records) out_rec[tuple * qstride] = the float's bits in the low half of a 64-bit word
  extern __shared__ __attribute__((aligned(16))) uint8_t s_q[];
  constexpr uint32_t W = kQualChunk;
  const uint32_t lane = lane_id(), wv = wave_id();
  float *s_lut = reinterpret_cast<float *>(s_q);                                            // [96]
  uint64_t *s_exp = reinterpret_cast<uint64_t *>(s_q + 96u * 4u);                           // [32]: per-lane index, so not left in constant memory
  uint8_t *rows = s_q + 96u * 4u + 32u * 8u + (size_t)wv * (64u * row_bytes + 64u * 17u * 4u + 64u * 12u);   // [64][row_bytes]
  float *tile = reinterpret_cast<float *>(rows + 64u * row_bytes);                          // [64][17]
  uint64_t *s_qp = reinterpret_cast<uint64_t *>(tile + 64u * 17u);                          // [64]: quality line of every read of the batch
  uint32_t *s_nw = reinterpret_cast<uint32_t *>(s_qp + 64);                                 // [64]: its windows
  uint32_t *rows32 = reinterpret_cast<uint32_t *>(rows);
  const uint32_t rdw = row_bytes / 4u;
  if (threadIdx.x < 96) s_lut[threadIdx.x] = c_qual_lut[threadIdx.x];
  else if (threadIdx.x < 128) s_exp[threadIdx.x - 96] = c_exp2_tab[threadIdx.x - 96];
  lds_barrier();
  const float lo = s_lut[0], hi = s_lut[95];
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  // first position >= p whose EOL bit equals `set`; past the bitmap everything is EOL
  auto next_bit = [&](uint64_t p, bool set) -> uint64_t {
    uint64_t wi = p >> 5;
    if (wi >= n_words) return set ? p : n_words * 32;
    uint32_t bits = (set ? eolw[wi] : ~eolw[wi]) & (0xffffffffu << (p & 31u));
    while (bits == 0u) { if (++wi >= n_words) return n_words * 32; bits = set ? eolw[wi] : ~eolw[wi]; }
    return wi * 32 + (uint32_t)__builtin_ctz(bits);
  };
  auto decode = [&](uint32_t c) -> float { return (c >= 33u && c < 33u + 96u) ? s_lut[c - 33u] : lo; };
  const uint64_t nr = *n_reads;
  const uint64_t n_waves = (uint64_t)gridDim.x * (kQualThreads / kWave);
  for (uint64_t b0 = ((uint64_t)blockIdx.x * (kQualThreads / kWave) + wv) * kWave; b0 < nr; b0 += n_waves * kWave) {
    const uint64_t r = b0 + lane;
    uint64_t q = 0, o = ~0ull;
    uint32_t n_win = 0, len = 0;
    if (r < nr) {
      const ReadDesc rd = reads[r];
      o = rd.out_off;
      if (o != ~0ull) {
        const uint64_t e1 = next_bit(rd.seq_pos, true);             // end of the sequence line
        len = (uint32_t)(e1 - rd.seq_pos);
        q = next_bit(next_bit(next_bit(e1, false), true), false);   // '+' line, its end, the quality line
        n_win = len >= k ? len - k + 1u : 0u;
      }
    }
    uint32_t max_win = n_win;
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)max_win, d, kWave); max_win = t > max_win ? t : max_win; }
    s_qp[lane] = q; s_nw[lane] = n_win;
    wave_sync();
    float sum = 0.0f;
    uint32_t bad = 0;
    for (uint32_t w0 = 0; w0 < max_win; w0 += W) {
      // characters cs .. of every read that has windows in this chunk (the one that leaves window w0 - 1 onward), as the ALIGNED
      // dwords that cover them: sixteen lanes per read, four reads per load instruction, no load waits on another. A dword
      // that overlaps the buffer lies in pages the buffer touches; one that does not is not read.
      const uint32_t cs = w0 ? w0 - 1u : 0u;
      const uintptr_t buf0 = (uintptr_t)bytes, buf1 = buf0 + n_bytes;
#pragma unroll 4
      for (uint32_t i4 = 0; i4 < (uint32_t)kWave; i4 += 4) {
        const uint32_t i = i4 + (lane >> 4);
        if (s_nw[i] <= w0) continue;
        const uintptr_t a0 = (buf0 + s_qp[i] + cs) & ~(uintptr_t)3;
        for (uint32_t d = lane & 15u; d < rdw; d += 16u) {
          const uintptr_t a = a0 + 4u * d;
          uint32_t v = 0;
          if (a + 4u > buf0 && a < buf1) {
            v = *reinterpret_cast<const uint32_t *>(a);
            if (a + 4u > buf1) v &= (1u << (8u * (uint32_t)(buf1 - a))) - 1u;     // past the end reads as 0, as before
          }
          rows32[i * rdw + d] = v;
        }
      }
      wave_sync();
      const uint8_t *my = rows + lane * row_bytes + (uint32_t)((buf0 + q + cs) & 3u);
      if (w0 == 0 && n_win) {                                         // init(): quality_score_iterator.hpp:99-115
        for (uint32_t i = 0; i < k; ++i) { const float x = decode(my[i]); if (x > lo && x < hi) sum += x; else ++bad; }
      }
      for (uint32_t t0 = 0; t0 < W && w0 + t0 < max_win; t0 += 16) {
#pragma unroll 4
        for (uint32_t t = 0; t < 16; ++t) {
          const uint32_t w = w0 + t0 + t;
          if (w < n_win) {
            if (w) {                                                  // next(): :127-159
              const float ov = decode(my[w - 1u - cs]), nv = decode(my[w + k - 1u - cs]);
              if (ov > lo && ov < hi) sum -= ov; else --bad;
              if (nv > lo && nv < hi) sum += nv; else ++bad;
            }
            tile[lane * 17u + t] = bad ? 0.0f : exp2f_libm(sum, s_exp);     // getValue(): :166-173
          }
        }
        wave_sync();
        // 16 floats of one read = one 64-byte line: four reads per store instruction
        for (uint32_t g = 0; g < (uint32_t)kWave; g += 4) {
          const uint32_t row = g + (lane >> 4), col = lane & 15u, w = w0 + t0 + col;
          const uint32_t nwr = (uint32_t)__shfl((int)n_win, (int)row, kWave);
          const uint64_t orow = ((uint64_t)(uint32_t)__shfl((int)(o >> 32), (int)row, kWave) << 32) | (uint32_t)__shfl((int)(uint32_t)o, (int)row, kWave);
          if (w < nwr) {
            if (out_rec) out_rec[(orow + w) * qstride] = (uint64_t)__float_as_uint(tile[row * 17u + col]);
            else out[orow + w] = tile[row * 17u + col];
          }
        }
        wave_sync();
      }
    }
  }
}

template <int NW, int BITS, bool WITH_IDS>
__global__ __launch_bounds__((ExCfg<NW, BITS>::NT)) void fastq_extract_kernel(
    PackedInput in, KShape shape, bool canonical, const uint32_t *__restrict__ line_base, const uint64_t *__restrict__ hdr_base,
    uint64_t file_offset, const uint64_t *__restrict__ out_off, uint64_t out_capacity, uint64_t *__restrict__ out_kmers,
    uint64_t *__restrict__ out_ids, ReadDesc *__restrict__ reads, uint32_t *__restrict__ flags, uint32_t kstride, uint32_t istride,
    const uint8_t *__restrict__ raw_edges /* de Bruijn tuples (kmi_debruijn.h): the input bytes; the id slot then takes 1 | edge byte << 32
                                             and the key is the smaller strand, the edge byte turned with it */) {
  using Cfg = ExCfg<NW, BITS>;
  __shared__ uint32_t s_eol[Cfg::EOL_DW];
  __shared__ uint32_t s_brk[Cfg::EOL_DW];
  __shared__ uint32_t s_stream[Cfg::STREAM_DW];
  __shared__ uint32_t s_scan[Cfg::NT / 64 + 2];
  __shared__ uint16_t s_pos[Cfg::TILE];
  __shared__ uint16_t s_hmask[WITH_IDS ? Cfg::NT : 1];   // record-start bits of every chunk
  __shared__ uint16_t s_hexcl[WITH_IDS ? Cfg::NT : 1];   // 1 + tile position of the last record start in earlier chunks
  __shared__ uint16_t s_lsmask[WITH_IDS ? Cfg::NT : 1];  // line-start bits of every chunk
  __shared__ uint16_t s_lcnt[WITH_IDS ? Cfg::NT : 1];    // line starts of the tile before every chunk
  uint32_t eol, ls, lbl, ltot;
  tile_front_packed<Cfg>(in, blockIdx.x, s_eol, s_stream, s_scan, eol, ls, lbl, ltot);
  const uint32_t lines_before = line_base[blockIdx.x] + lbl;
  if (WITH_IDS) { s_lsmask[threadIdx.x] = (uint16_t)ls; s_lcnt[threadIdx.x] = (uint16_t)lbl; }
  if (WITH_IDS) {
    // record starts = line starts whose line index % 4 == 0 (the '@' line)
    uint32_t cur = lines_before, rest = ls, hm = 0, hlast = 0;
    while (rest) {
      const uint32_t q = (uint32_t)__builtin_ctz(rest);
      if ((cur & 3u) == 0u) { hm |= 1u << q; hlast = threadIdx.x * Cfg::C + q + 1u; }
      cur += 1; rest &= rest - 1u;
    }
    s_hmask[threadIdx.x] = (uint16_t)hm;
    s_hexcl[threadIdx.x] = (uint16_t)block_exclusive_max<uint32_t>(hlast, s_scan, (uint32_t *)nullptr);
  }
  const uint32_t total = tile_window_list<Cfg>(tile_break_image<Cfg>(in.brk, in.n_cover, blockIdx.x, s_eol, s_brk), ls, lines_before,
                                               shape.k, s_pos, s_scan);
  const uint64_t base = out_off[blockIdx.x];
  if (base + total > out_capacity) {
    if (threadIdx.x == 0 && total) atomicOr(&flags[1], 1u);
    return;
  }
  const uint64_t tile0 = (uint64_t)blockIdx.x * Cfg::TILE;
  // records of (key words, id) at a 16-byte aligned address leave as 16-byte stores (store_record_vec)
  const bool vec = WITH_IDS && kRecVec<NW> && kstride == (uint32_t)NW + 1u && istride == kstride && out_ids == out_kmers + NW && ((uintptr_t)out_kmers & 15u) == 0;
  for (uint32_t q = threadIdx.x; q < total; q += Cfg::NT) {
    uint64_t rc[NW], fw[NW], key[NW];
    const uint32_t pos = s_pos[q];
    window_at<Cfg>(s_stream, pos, shape, rc, fw);
    select_strand<NW>(rc, fw, canonical || (WITH_IDS && raw_edges != nullptr), key);
    if (!vec) {
#pragma unroll
      for (int w = 0; w < NW; ++w) out_kmers[(base + q) * kstride + w] = key[w];
    }
    if (WITH_IDS && raw_edges) {   // uniform
      // edge_iterator.hpp:163-177: the bases left and right of the k-mer in its (single-line) FASTQ sequence, DNA16 codes,
      // nothing where the read ends; reverse_complement_edges (de_bruijn_node_trait.hpp:122-124) when the other strand is kept
      const uint64_t p0 = tile0 + pos;
      const uint32_t lc = p0 > 0 ? raw_edges[p0 - 1] : (uint32_t)'\n';
      const uint32_t rcch = p0 + shape.k < in.n_bytes ? raw_edges[p0 + shape.k] : (uint32_t)'\n';
      uint32_t e = ((is_eol(lc) ? 0u : code_dna16(lc)) << 4) | (is_eol(rcch) ? 0u : code_dna16(rcch));
      if (less_words<NW>(rc, fw)) e = (comp_code<4>(e & 0xFu) << 4) | comp_code<4>(e >> 4);
      if constexpr (WITH_IDS && kRecVec<NW>) {
        if (vec) { store_record_vec<NW>(out_kmers, base + q, key, 1ull | ((uint64_t)e << 32)); continue; }
      }
      out_ids[(base + q) * istride] = 1ull | ((uint64_t)e << 32);
      continue;
    }
    if (WITH_IDS) {
      // ShortSequenceKmerId (sequence.hpp:156-157): record file offset << 16 | offset of the k-mer's
      // first base from the record start (kmer_parser.hpp:378-386)
      const uint32_t j = pos / Cfg::C, p = pos % Cfg::C;
      const uint32_t m = (uint32_t)s_hmask[j] & ((2u << p) - 1u);
      uint64_t rec;   // 1 + byte position (relative to the buffer) of the record start
      if (m) rec = tile0 + j * Cfg::C + (31u - (uint32_t)__builtin_clz(m)) + 1u;
      else if (s_hexcl[j]) rec = tile0 + s_hexcl[j];
      else rec = hdr_base[blockIdx.x];
      const uint64_t rec_off = file_offset + rec - 1u, d = tile0 + pos - (rec - 1u);
      if (rec == 0 || d > 0xFFFFu) atomicOr(&flags[3], 1u);   // ShortSequenceKmerId increment overflow (sequence.hpp:177-183)
      bool stored = false;
      if constexpr (kRecVec<NW>) {
        if (vec) { store_record_vec<NW>(out_kmers, base + q, key, ((rec_off & 0xFFFFFFFFFFull) << 16) | (d & 0xFFFFull)); stored = true; }
      }
      if (!stored) out_ids[(base + q) * istride] = ((rec_off & 0xFFFFFFFFFFull) << 16) | (d & 0xFFFFull);
      // first window of its read (the window starts on the line start of the sequence line)
      if (reads && ((s_lsmask[j] >> p) & 1u)) {
        // the descriptor slot is the read's sequence index: its sequence line is line 4 * index + 1 of the buffer
        const uint32_t line = line_base[blockIdx.x] + s_lcnt[j] + (uint32_t)__builtin_popcount((uint32_t)s_lsmask[j] & ((1u << p) - 1u));
        ReadDesc rd; rd.seq_pos = tile0 + pos; rd.out_off = base + q;
        reads[(line - 1u) >> 2] = rd;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Sequence filters (the SeqIterType argument of read_file_* / build_*, filtered_sequence_iterator.hpp): the window-break
// bitmap. NSplitSequencesIterator cuts a sequence at every 'N' / 'n' (NCharFilter, :429-440), so a k-mer window is
// kept iff it covers no EOL and no such byte: break = EOL | N | n. NFilterSequencesIterator drops every record whose
// sequence holds an 'N' (NSequenceFilter, :154-165): break = EOL | every byte of a line that holds an 'N' (lines of the
// other roles carry no windows, so marking them too changes nothing). Everything downstream reads the break bitmap
// where it read the EOL bitmap for window validity; line structure stays on the EOL bitmap.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fastq_nbits_kernel(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t n_words, bool lower_too,
                                                         uint32_t *__restrict__ brk, uint32_t *__restrict__ eolw, uint32_t *__restrict__ nbw) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  const uint64_t p0 = w * 32;
  uint32_t e = 0, n = 0;
  if (p0 + 32 <= n_bytes) {
    const uint4 a = *reinterpret_cast<const uint4 *>(bytes + p0), b = *reinterpret_cast<const uint4 *>(bytes + p0 + 16);
    const uint32_t dw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const uint32_t c = (dw[i >> 2] >> (8 * (i & 3))) & 0xffu;
      e |= (is_eol(c) ? 1u : 0u) << i;
      n |= ((c == 'N' || (lower_too && c == 'n')) ? 1u : 0u) << i;
    }
  } else {
    for (int i = 0; i < 32; ++i) {
      const uint64_t p = p0 + i;
      const uint32_t c = p < n_bytes ? bytes[p] : (uint32_t)'\n';   // bytes past the end count as EOL
      e |= (is_eol(c) ? 1u : 0u) << i;
      n |= ((c == 'N' || (lower_too && c == 'n')) ? 1u : 0u) << i;
    }
  }
  brk[w] = e | n;
  if (eolw) { eolw[w] = e; nbw[w] = n; }
}

// N_FILTER: the first 'N' of a line marks the whole line in the break bitmap
__global__ __launch_bounds__(256) void fastq_poison_lines_kernel(const uint32_t *__restrict__ eolw, const uint32_t *__restrict__ nbw,
                                                                uint64_t n_words, uint32_t *__restrict__ brk) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  uint32_t rest = nbw[w];
  while (rest) {
    const uint32_t b = (uint32_t)__builtin_ctz(rest);
    rest &= rest - 1u;
    // line start: one past the last EOL before the N at word w, bit b
    uint64_t s = 0;
    {
      uint64_t wi = w;
      uint32_t bits = eolw[wi] & ((1u << b) - 1u);
      while (bits == 0u && wi > 0) bits = eolw[--wi];
      if (bits) s = wi * 32 + (31u - (uint32_t)__builtin_clz(bits)) + 1u;
    }
    // an earlier N of the same line does the marking
    bool first = true;
    for (uint64_t wi = s >> 5; wi <= w && first; ++wi) {
      uint32_t m = nbw[wi];
      if (wi == (s >> 5)) m &= ~((1u << (s & 31u)) - 1u);
      if (wi == w) m &= (1u << b) - 1u;
      if (m) first = false;
    }
    if (!first) continue;
    // line end: the first EOL at or behind p (bytes past the input are EOL)
    uint64_t e = n_words * 32;
    {
      uint64_t wi = w;
      uint32_t bits = eolw[wi] & ~((1u << b) - 1u);
      while (bits == 0u && wi + 1 < n_words) bits = eolw[++wi];
      if (bits) e = wi * 32 + (uint32_t)__builtin_ctz(bits);
    }
    for (uint64_t wi = s >> 5; wi <= ((e - 1) >> 5); ++wi) {
      uint32_t m = 0xffffffffu;
      if (wi == (s >> 5)) m &= ~((1u << (s & 31u)) - 1u);
      if (wi == ((e - 1) >> 5) && (e & 31u)) m &= (1u << (e & 31u)) - 1u;
      atomicOr(&brk[wi], m);
    }
  }
}

// what read_block counts as sequences under a filter (kmer_file_helper.hpp:128-178): N_FILTER -- the records that pass,
// i.e. the sequence lines whose first byte is not marked; N_SPLIT -- the non-empty pieces, i.e. the bytes of sequence
// lines that are no break and follow one
template <int NW, int BITS>
__global__ __launch_bounds__((ExCfg<NW, BITS>::NT)) void fastq_subseq_count_kernel(PackedInput in, const uint32_t *__restrict__ line_base,
                                                                                  bool pieces, unsigned long long *__restrict__ total) {
  using Cfg = ExCfg<NW, BITS>;
  constexpr int C = Cfg::C;
  __shared__ uint32_t s_eol[Cfg::EOL_DW];
  __shared__ uint32_t s_stream[Cfg::STREAM_DW];
  __shared__ uint32_t s_scan[Cfg::NT / 64 + 2];
  uint32_t eol, ls, lbl, ltot;
  tile_front_packed<Cfg>(in, blockIdx.x, s_eol, s_stream, s_scan, eol, ls, lbl, ltot);
  const uint32_t role = fastq_seq_role_mask(line_base[blockIdx.x] + lbl, ls, Cfg::CMASK);
  const uint64_t n_units = in.n_cover / C, g = (uint64_t)blockIdx.x * Cfg::NT + threadIdx.x;
  const uint32_t bk = g < n_units ? read_eol_unit<C>(in.brk, g) : Cfg::CMASK;
  const uint32_t prevb = g == 0 ? 1u : ((read_eol_unit<C>(in.brk, (g - 1 < n_units ? g - 1 : n_units - 1)) >> (C - 1)) & 1u);
  uint32_t c;
  if (pieces) c = (uint32_t)__builtin_popcount(~bk & ((bk << 1) | prevb) & role & Cfg::CMASK);
  else c = (uint32_t)__builtin_popcount(ls & role & ~bk);
  uint32_t tot;
  block_exclusive_scan<uint32_t>(c, s_scan, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(total, (unsigned long long)tot);
}

// ---------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------
// offsets over the tile records (reduce per 1024 tiles, scan of the block summaries, apply)
static kmi_status launch_tile_offsets(kmi_ctx *ctx, const TileInfo *info, uint64_t n_tiles, uint32_t tile_bytes, uint64_t *hdr,
                                      uint32_t *base, uint64_t *off) {
  const uint64_t n_blocks = (n_tiles + 1023) / 1024;
  void *ps_;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(TileSum) * (n_blocks + 1), &ps_));
  TileSum *sums = (TileSum *)ps_;
  ProfScope ps(ctx, "fastq_scan_offsets", n_tiles);
  if (n_blocks > 0) {
    hipLaunchKernelGGL(fastq_offsets_reduce_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, ctx->stream, info, n_tiles, tile_bytes, sums);
  }
  hipLaunchKernelGGL(fastq_offsets_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums, n_blocks, n_tiles, off, ctx->d_totals);
  if (n_blocks > 0) {
    hipLaunchKernelGGL(fastq_offsets_apply_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, ctx->stream, info, n_tiles, tile_bytes,
                       (const TileSum *)sums, hdr, base, off, ctx->d_flags);
  }
  return KMI_OK;
}

struct ScanResult {
  uint64_t n_tiles;
  uint32_t *line_base;
  uint64_t *out_off;
  uint64_t *hdr_base;
  PackedInput packed;
};

template <int NW, int BITS>
static kmi_status scan_impl(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, const KShape &shape, ScanResult *r,
                            bool reuse = false, bool check_lengths = true, uint32_t seq_filter = KMI_SEQ_ALL, bool rna = false) {
  using Cfg = ExCfg<NW, BITS>;
  const uint64_t n_tiles = (n_bytes + Cfg::TILE - 1) / Cfg::TILE;
  void *p;
  KMI_TRY(ws_get(ctx, WS_TILE_INFO, sizeof(TileInfo) * (n_tiles + 1), &p));
  TileInfo *info = (TileInfo *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_BASE, sizeof(uint32_t) * (n_tiles + 1), &p));
  uint32_t *base = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_OFF, sizeof(uint64_t) * (n_tiles + 2), &p));
  uint64_t *off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_HDR, sizeof(uint64_t) * (n_tiles + 1), &p));
  uint64_t *hdr = (uint64_t *)p;
  const uint64_t n_cover = n_tiles * Cfg::TILE;
  KMI_TRY(ws_get(ctx, WS_PK_EOL, n_cover / 8 + 64, &p));
  uint8_t *pk_eol = (uint8_t *)p;
  KMI_TRY(ws_get(ctx, WS_PK_STREAM, n_cover * BITS / 8 + 64, &p));
  uint8_t *pk_stream = (uint8_t *)p;
  r->n_tiles = n_tiles; r->line_base = base; r->out_off = off; r->hdr_base = hdr;
  r->packed.eol = pk_eol; r->packed.stream = pk_stream; r->packed.n_bytes = n_bytes; r->packed.n_cover = n_cover; r->packed.n_valid = n_bytes;
  uint32_t *brk = nullptr;
  if (seq_filter != KMI_SEQ_ALL) {
    KMI_TRY(ws_get(ctx, WS_PK_BRK, n_cover / 8 + 64, &p));
    brk = (uint32_t *)p;
    r->packed.brk = (const uint8_t *)brk;
  }
  if (reuse) return KMI_OK;   // the scan of these very bytes is still in the workspace
  if (brk && n_tiles > 0) {
    ProfScope ps(ctx, "fastq_nbits", n_bytes);
    const uint64_t n_words = n_cover / 32;
    uint32_t *eolw = nullptr, *nbw = nullptr;
    if (seq_filter == KMI_SEQ_N_FILTER) {
      KMI_TRY(ws_get(ctx, WS_PK_NB, 2 * (n_cover / 8) + 64, &p));
      eolw = (uint32_t *)p; nbw = eolw + n_words;
    }
    hipLaunchKernelGGL(fastq_nbits_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes,
                       n_words, seq_filter == KMI_SEQ_N_SPLIT, brk, eolw, nbw);
    if (seq_filter == KMI_SEQ_N_FILTER)
      hipLaunchKernelGGL(fastq_poison_lines_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream,
                         (const uint32_t *)eolw, (const uint32_t *)nbw, n_words, brk);
  }
  if (n_tiles > 0) {
    ProfScope ps(ctx, "fastq_scan_tiles", n_bytes);
    using SCfg = ScanCfg<NW, BITS>;
    static_assert(SCfg::TILE == Cfg::TILE, "the scan pass writes the packed arrays of the same tiles");
    hipLaunchKernelGGL((fastq_scan_tiles_kernel<SCfg>), dim3((unsigned)n_tiles), dim3(SCfg::NT), 0, ctx->stream,
                       bytes_dev, (uint64_t)n_bytes, shape.k, pk_eol, pk_stream, info, ctx->d_flags, (const uint8_t *)brk, n_cover, rna);
  }
  KMI_TRY(launch_tile_offsets(ctx, info, n_tiles, (uint32_t)Cfg::TILE, hdr, base, off));
  if (brk && n_tiles > 0) {
    // n_seqs under a filter: records that pass / non-empty pieces (totals[2] held lines / 4)
    KMI_HIP(ctx, hipMemsetAsync(ctx->d_totals + 2, 0, sizeof(uint64_t), ctx->stream));
    hipLaunchKernelGGL((fastq_subseq_count_kernel<NW, BITS>), dim3((unsigned)n_tiles), dim3(Cfg::NT), 0, ctx->stream, r->packed,
                       (const uint32_t *)base, seq_filter == KMI_SEQ_N_SPLIT, (unsigned long long *)(ctx->d_totals + 2));
  }
  if (n_tiles > 0 && check_lengths) {
    ProfScope ps(ctx, "fastq_check", n_bytes);
    hipLaunchKernelGGL((fastq_check_lengths_kernel<Cfg::TILE>), dim3(2048), dim3(256), 0, ctx->stream, (const uint32_t *)pk_eol,
                       (uint64_t)(n_cover / 32), n_tiles, (const uint32_t *)base, ctx->d_flags);
  }
  KMI_HIP(ctx, hipGetLastError());
  return KMI_OK;
}

// totals + FASTQ marker verdict of the last scan
static kmi_status read_totals(kmi_ctx *ctx, uint64_t *n_tuples, uint64_t *n_seqs) {
  uint32_t flags0 = 0;
  KMI_HIP(ctx, hipMemcpyAsync(ctx->h_totals, ctx->d_totals, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(&flags0, ctx->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_tuples) *n_tuples = ctx->h_totals[1];
  if (n_seqs) *n_seqs = ctx->h_totals[2];
  if (flags0 & 1u) return set_err(ctx, KMI_ERR_PARSE, "FASTQ: missing @ on first line of a record");
  if (flags0 & 2u) return set_err(ctx, KMI_ERR_PARSE, "FASTQ: missing + on third line of a record");
  if (flags0 & 4u) return set_err(ctx, KMI_ERR_PARSE, "FASTQ: truncated record? seq and qual differ in length");
  return KMI_OK;
}

template <int NW, int BITS>
static kmi_status extract_count_impl(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, KShape shape, uint32_t seq_filter, bool rna,
                                     uint64_t *n_tuples, uint64_t *n_seqs) {
  ScanResult r;
  KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t) * 16, ctx->stream));
  KMI_TRY((scan_impl<NW, BITS>(ctx, bytes_dev, n_bytes, shape, &r, false, true, seq_filter, rna)));
  return read_totals(ctx, n_tuples, n_seqs);
}

template <int NW, int BITS>
static kmi_status extract_run_impl(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                                   KShape shape, uint64_t file_offset, uint64_t *out_kmers_dev, uint64_t *out_ids_dev,
                                   float *out_quals_dev, size_t out_capacity, bool apply_strand, bool scan_done, uint64_t *n_tuples,
                                   uint64_t *n_seqs, uint32_t rec_words, bool edges) {
  using Cfg = ExCfg<NW, BITS>;
  // rec_words != 0: out_kmers_dev is a record buffer (key words, id[, quality bits]) of rec_words words per tuple
  const uint32_t kstride = rec_words ? rec_words : (uint32_t)NW, istride = rec_words ? rec_words : 1u;
  uint64_t *ids_at = rec_words ? out_kmers_dev + NW : out_ids_dev;
  const bool want_ids = ids_at != nullptr, want_quals = out_quals_dev != nullptr || rec_words == (uint32_t)NW + 2u;
  ScanResult r;
  if (!scan_done) KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t) * 16, ctx->stream));
  KMI_TRY((scan_impl<NW, BITS>(ctx, bytes_dev, n_bytes, shape, &r, scan_done, true, cfg->seq_filter, is_rna(cfg))));
  if (r.n_tiles > 0) {
    ProfScope ps(ctx, "fastq_extract", n_bytes);
    const bool canonical = apply_strand && cfg->strand != KMI_STRAND_SINGLE;
    ReadDesc *reads = nullptr;
    if (want_quals) {
      // one descriptor slot per sequence (totals[2] of the scan), empty until the read's first window fills it
      uint64_t n_seq_now = 0;
      KMI_HIP(ctx, hipMemcpyAsync(&n_seq_now, ctx->d_totals + 2, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
      KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
      void *pr;
      KMI_TRY(ws_get(ctx, WS_READS, sizeof(ReadDesc) * (n_seq_now + 16), &pr));
      reads = (ReadDesc *)pr;
      KMI_HIP(ctx, hipMemsetAsync(reads, 0xff, sizeof(ReadDesc) * (n_seq_now + 16), ctx->stream));
    }
    if (want_ids) {
      hipLaunchKernelGGL((fastq_extract_kernel<NW, BITS, true>), dim3((unsigned)r.n_tiles), dim3(Cfg::NT), 0, ctx->stream,
                         r.packed, shape, canonical, (const uint32_t *)r.line_base, (const uint64_t *)r.hdr_base, file_offset,
                         (const uint64_t *)r.out_off, (uint64_t)out_capacity, out_kmers_dev, ids_at, reads, ctx->d_flags, kstride, istride,
                         edges ? bytes_dev : (const uint8_t *)nullptr);
    } else {
      hipLaunchKernelGGL((fastq_extract_kernel<NW, BITS, false>), dim3((unsigned)r.n_tiles), dim3(Cfg::NT), 0, ctx->stream,
                         r.packed, shape, canonical, (const uint32_t *)r.line_base, (const uint64_t *)r.hdr_base, file_offset,
                         (const uint64_t *)r.out_off, (uint64_t)out_capacity, out_kmers_dev, (uint64_t *)nullptr, (ReadDesc *)nullptr,
                         ctx->d_flags, kstride, istride, (const uint8_t *)nullptr);
    }
    if (want_quals) {
      ProfScope pq(ctx, "fastq_quality", n_bytes);
      hipLaunchKernelGGL(fastq_quality_kernel, dim3(2048), dim3(kQualThreads), qual_lds_bytes(shape.k), ctx->stream, bytes_dev,
                         (uint64_t)n_bytes, (const uint32_t *)r.packed.eol, (uint64_t)(r.packed.n_cover / 32), shape.k,
                         qual_row_bytes(shape.k), (const ReadDesc *)reads, (const uint64_t *)(ctx->d_totals + 2), out_quals_dev,
                         rec_words == (uint32_t)NW + 2u ? out_kmers_dev + NW + 1 : (uint64_t *)nullptr, rec_words);
    }
  }
  KMI_HIP(ctx, hipGetLastError());
  uint32_t fl[4] = {0, 0, 0, 0};
  KMI_HIP(ctx, hipMemcpyAsync(fl, ctx->d_flags, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
  KMI_TRY(read_totals(ctx, n_tuples, n_seqs));
  if (fl[1]) return set_err(ctx, KMI_ERR_OVERFLOW, "extract: output capacity too small");
  if (fl[3]) return set_err(ctx, KMI_ERR_OVERFLOW, "ShortSequenceKmerId increment overflow (k-mer more than 65535 bytes into its record)");
  return KMI_OK;
}

template <int NW, int BITS>
static kmi_status fastq_scan_impl(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, KShape shape, FastqScan *out, bool check_lengths,
                                  bool rna, uint32_t seq_filter) {
  ScanResult r;
  KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t) * 16, ctx->stream));
  KMI_TRY((scan_impl<NW, BITS>(ctx, bytes_dev, n_bytes, shape, &r, false, check_lengths, seq_filter, rna)));
  out->pk_brk = r.packed.brk;
  out->n_tiles = r.n_tiles; out->line_base = r.line_base; out->tile_off = r.out_off; out->hdr_base = r.hdr_base;
  out->pk_eol = r.packed.eol; out->pk_stream = r.packed.stream; out->n_bytes = r.packed.n_bytes; out->n_cover = r.packed.n_cover;
  return read_totals(ctx, &out->n_tuples, &out->n_seqs);
}

kmi_status fastq_quality_reads(kmi_ctx *ctx, const FastqScan &sc, ReadDesc **reads) {
  void *pr;
  KMI_TRY(ws_get(ctx, WS_READS, sizeof(ReadDesc) * (sc.n_seqs + 16), &pr));
  KMI_HIP(ctx, hipMemsetAsync(pr, 0xff, sizeof(ReadDesc) * (sc.n_seqs + 16), ctx->stream));
  *reads = (ReadDesc *)pr;
  return KMI_OK;
}
kmi_status fastq_quality_launch(kmi_ctx *ctx, const uint8_t *bytes_dev, const FastqScan &sc, uint32_t k, const ReadDesc *reads, float *out_quals) {
  ProfScope pq(ctx, "fastq_quality", sc.n_bytes);
  hipLaunchKernelGGL(fastq_quality_kernel, dim3(2048), dim3(kQualThreads), qual_lds_bytes(k), ctx->stream, bytes_dev, (uint64_t)sc.n_bytes,
                     reinterpret_cast<const uint32_t *>(sc.pk_eol), (uint64_t)(sc.n_cover / 32), k, qual_row_bytes(k), reads,
                     (const uint64_t *)(ctx->d_totals + 2), out_quals, (uint64_t *)nullptr, 0u);
  KMI_HIP(ctx, hipGetLastError());
  return KMI_OK;
}

// verdict of the seq/qual length rule when the list pass carries it (call after a stream sync point is acceptable: it syncs)
kmi_status fastq_length_verdict(kmi_ctx *ctx) {
  uint32_t f = 0;
  KMI_HIP(ctx, hipMemcpyAsync(&f, ctx->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (f & 4u) {
    KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t), ctx->stream));
    return set_err(ctx, KMI_ERR_PARSE, "FASTQ: truncated record? seq and qual differ in length");
  }
  return KMI_OK;
}

kmi_status fastq_scan(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, FastqScan *out, bool check_lengths) {
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (cfg->seq_format != KMI_FMT_FASTQ) return set_err(ctx, KMI_ERR_INVALID, "only FASTQ is implemented on the device yet");
  KMI_DISPATCH(shape, fastq_scan_impl, ctx, bytes_dev, n_bytes, shape, out, check_lengths, is_rna(cfg), cfg->seq_filter);
}

// ---- FASTA: byte-space scan + compaction (kmi_fasta.hip), then count / extract over the compacted stream
template <int NW, int BITS>
static kmi_status fasta_extract_impl(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, KShape shape,
                                     uint64_t file_offset, uint64_t *out_kmers_dev, uint64_t *out_ids_dev, size_t out_capacity,
                                     bool apply_strand, bool count_only, uint64_t *n_tuples, uint64_t *n_seqs, uint32_t rec_words, bool edges) {
  using Cfg = ExCfg<NW, BITS>;
  const uint32_t kstride = rec_words ? rec_words : (uint32_t)NW, istride = rec_words ? rec_words : 1u;
  if (rec_words) out_ids_dev = out_kmers_dev + NW;   // records: the id follows the key words
  FastaScan fs;
  KMI_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t) * 16, ctx->stream));
  KMI_TRY(fasta_scan(ctx, cfg, bytes_dev, n_bytes, file_offset, out_ids_dev != nullptr, &fs));
  PackedInput in; in.eol = fs.pk_break; in.stream = fs.pk_stream; in.n_bytes = fs.n_chars; in.n_cover = fs.n_cover; in.n_valid = fs.n_valid;
  const uint64_t n_tiles = (fs.n_chars + Cfg::TILE - 1) / Cfg::TILE;
  void *p;
  KMI_TRY(ws_get(ctx, WS_TILE_INFO, sizeof(TileInfo) * (n_tiles + 1), &p)); TileInfo *info = (TileInfo *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_BASE, sizeof(uint32_t) * (n_tiles + 1), &p)); uint32_t *base = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_OFF, sizeof(uint64_t) * (n_tiles + 2), &p)); uint64_t *off = (uint64_t *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_HDR, sizeof(uint64_t) * (n_tiles + 1), &p)); uint64_t *hdr = (uint64_t *)p;
  if (n_tiles > 0) {
    ProfScope ps(ctx, "fasta_count_tiles", fs.n_chars);
    hipLaunchKernelGGL((fasta_count_tiles_kernel<NW, BITS>), dim3((unsigned)n_tiles), dim3(Cfg::NT), 0, ctx->stream, in, shape.k, info);
  }
  KMI_TRY(launch_tile_offsets(ctx, info, n_tiles, (uint32_t)Cfg::TILE, hdr, base, off));
  if (!count_only && n_tiles > 0) {
    ProfScope ps(ctx, "fasta_extract", fs.n_chars);
    const bool canonical = apply_strand && cfg->strand != KMI_STRAND_SINGLE;
    if (out_ids_dev)
      hipLaunchKernelGGL((fasta_extract_kernel<NW, BITS, true>), dim3((unsigned)n_tiles), dim3(Cfg::NT), 0, ctx->stream, in, shape, canonical,
                         fs.ids_by_rank, (const uint64_t *)off, (uint64_t)out_capacity, out_kmers_dev, out_ids_dev, ctx->d_flags, kstride,
                         istride, edges ? bytes_dev : (const uint8_t *)nullptr, file_offset);
    else
      hipLaunchKernelGGL((fasta_extract_kernel<NW, BITS, false>), dim3((unsigned)n_tiles), dim3(Cfg::NT), 0, ctx->stream, in, shape, canonical,
                         (const uint64_t *)nullptr, (const uint64_t *)off, (uint64_t)out_capacity, out_kmers_dev, (uint64_t *)nullptr,
                         ctx->d_flags, kstride, istride);
  }
  KMI_HIP(ctx, hipGetLastError());
  uint32_t fl[4] = {0, 0, 0, 0};
  KMI_HIP(ctx, hipMemcpyAsync(fl, ctx->d_flags, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(ctx->h_totals, ctx->d_totals, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_tuples) *n_tuples = ctx->h_totals[1];
  if (n_seqs) *n_seqs = fs.n_seqs;
  if (fl[1]) return set_err(ctx, KMI_ERR_OVERFLOW, "extract: output capacity too small");
  return KMI_OK;
}

static kmi_status fasta_extract(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, KShape shape,
                                uint64_t file_offset, uint64_t *out_kmers_dev, uint64_t *out_ids_dev, size_t out_capacity, bool apply_strand,
                                bool count_only, uint64_t *n_tuples, uint64_t *n_seqs, uint32_t rec_words = 0, bool edges = false) {
  KMI_DISPATCH(shape, fasta_extract_impl, ctx, cfg, bytes_dev, n_bytes, shape, file_offset, out_kmers_dev, out_ids_dev, out_capacity,
               apply_strand, count_only, n_tuples, n_seqs, rec_words, edges);
}

// Illumina18QualityScoreCodec<float>::DecodeLUT by its generating formula (quality_scores.hpp:111-112):
// log2(1 - 10^(-q/10)) in long double, entry 0 = lowest(), entries 94 and 95 = 0
kmi_status upload_quality_lut(kmi_ctx *ctx) {
  float lut[96];
  lut[0] = -3.402823466e+38F;
  for (int q = 1; q < 94; ++q) lut[q] = (float)(double)log2l(1.0L - exp2l((long double)q * log2l(10.0L) / (-10.0L)));
  lut[94] = 0.0f; lut[95] = 0.0f;
  KMI_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_qual_lut), lut, sizeof(lut)));
  uint64_t tab[32];
  for (int i = 0; i < 32; ++i) {
    const double v = (double)exp2l((long double)i / 32.0L);
    uint64_t u; memcpy(&u, &v, sizeof(u));
    tab[i] = u - ((uint64_t)i << 47);
  }
  KMI_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_exp2_tab), tab, sizeof(tab)));
  return KMI_OK;
}

kmi_status extract_count(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                         uint64_t *n_tuples, uint64_t *n_seqs) {
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (n_bytes == 0) { if (n_tuples) *n_tuples = 0; if (n_seqs) *n_seqs = 0; return KMI_OK; }
  if (cfg->seq_format == KMI_FMT_FASTA)
    return fasta_extract(ctx, cfg, bytes_dev, n_bytes, shape, 0, nullptr, nullptr, 0, false, true, n_tuples, n_seqs);
  KMI_DISPATCH(shape, extract_count_impl, ctx, bytes_dev, n_bytes, shape, cfg->seq_filter, is_rna(cfg), n_tuples, n_seqs);
}

kmi_status extract_run(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes,
                       uint64_t file_offset, uint64_t *out_kmers_dev, uint64_t *out_ids_dev, size_t out_capacity,
                       bool apply_strand, bool scan_done, uint64_t *n_tuples, uint64_t *n_seqs, float *out_quals_dev, uint32_t rec_words, bool edges) {
  // edges (FASTQ, records): the value word of every record is 1 | edge byte << 32 and the key the smaller strand (kmi_debruijn.h)
  // rec_words != 0: out_kmers_dev takes whole records -- key words, id, and with rec_words == n_words + 2 the quality's float
  // bits -- rec_words words per tuple, the layout the multimap insert reads (out_ids_dev is then unused; out_quals_dev beside
  // records of n_words + 1 words takes the qualities as one dense float array)
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  if (n_bytes == 0) { if (n_tuples) *n_tuples = 0; if (n_seqs) *n_seqs = 0; return KMI_OK; }
  if ((out_quals_dev && !out_ids_dev && !rec_words) || ((out_quals_dev || rec_words == shape.n_words + 2u) && cfg->seq_format != KMI_FMT_FASTQ))
    return set_err(ctx, KMI_ERR_INVALID, "k-mer qualities need FASTQ input and are produced together with the ids");
  if (edges && rec_words != shape.n_words + 1u) return set_err(ctx, KMI_ERR_INVALID, "edge tuples are records of n_words + 1 words");
  if (cfg->seq_format == KMI_FMT_FASTA)   // (edges: the value word is the edge byte, the key stays as parsed; dbg_edges makes node form of it)
    return fasta_extract(ctx, cfg, bytes_dev, n_bytes, shape, file_offset, out_kmers_dev, out_ids_dev, out_capacity, apply_strand, false,
                         n_tuples, n_seqs, rec_words, edges);
  KMI_DISPATCH(shape, extract_run_impl, ctx, cfg, bytes_dev, n_bytes, shape, file_offset, out_kmers_dev, out_ids_dev, out_quals_dev,
               out_capacity, apply_strand, scan_done, n_tuples, n_seqs, rec_words, edges);
}

// ---------------------------------------------------------------------------
// Record-aligned partition of a FASTQ buffer on the device: FASTQParser::find_first_record (fastq_loader.hpp:269-364) as
// partitioned_file<..., FASTQParser> applies it (file.hpp:1216-1430). One thread per nominal split point: skip the rest
// of the line the split falls in, look at the first characters of the next four lines; a record starts at the line
// where '@' is followed two lines later by '+' (a quality line may itself begin with '@').
// ---------------------------------------------------------------------------
// first record start at or after `pos` (n when there is none in the buffer)
__device__ __forceinline__ uint64_t fastq_first_record_from(const uint8_t *__restrict__ bytes, uint64_t n, uint64_t pos, bool starts_file = true) {
  // starts_file = false: byte 0 of the buffer lies somewhere inside the file, so position 0 is examined like any other
  if (pos == 0 && starts_file) return 0;
  if (pos >= n) return n;
  auto eol = [&](uint64_t i) { return bytes[i] == '\n' || bytes[i] == '\r'; };
  uint64_t i = pos;
  while (i < n && !eol(i)) ++i;          // the rest of this (partial) line
  uint64_t starts[4];
  uint8_t firsts[4];
  for (int l = 0; l < 4; ++l) {
    while (i < n && eol(i)) ++i;
    if (i >= n) return n;
    starts[l] = i; firsts[l] = bytes[i];
    while (i < n && !eol(i)) ++i;
  }
  uint64_t c = n;
  if (firsts[0] == '@' && firsts[2] == '+') c = starts[0];
  else if (firsts[1] == '@' && firsts[3] == '+') c = starts[1];
  else if (firsts[0] == '+' && firsts[2] == '@') c = starts[2];
  else if (firsts[1] == '+' && firsts[3] == '@') c = starts[3];
  return c;
}
__global__ void fastq_find_first_records_kernel(const uint8_t *__restrict__ bytes, uint64_t n, uint32_t n_parts, uint64_t *__restrict__ cuts) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_parts) return;
  if (r == n_parts) { cuts[r] = n; return; }
  const uint64_t pos = n / n_parts * r + (n % n_parts) * r / n_parts;   // = floor(n r / n_parts) without overflow
  cuts[r] = fastq_first_record_from(bytes, n, pos);
}
// the same for explicit positions (cuts[i] holds position i on entry, the record start on exit)
__global__ void fastq_find_records_at_kernel(const uint8_t *__restrict__ bytes, uint64_t n, uint32_t n_pos, uint64_t *__restrict__ cuts, bool starts_file) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_pos) cuts[r] = fastq_first_record_from(bytes, n, cuts[r], starts_file);
}

}  // namespace kmi

extern "C" kmi_status kmi_fastq_partition_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, uint32_t n_parts, uint64_t *cuts_host) {
  using namespace kmi;
  if (!ctx || !cuts_host || n_parts == 0) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_bytes == 0) { for (uint32_t r = 0; r <= n_parts; ++r) cuts_host[r] = 0; return KMI_OK; }
  void *p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * ((size_t)n_parts + 1), &p));
  hipLaunchKernelGGL(fastq_find_first_records_kernel, dim3((n_parts + 1 + 63) / 64), dim3(64), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, n_parts,
                     (uint64_t *)p);
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipMemcpyAsync(cuts_host, p, sizeof(uint64_t) * ((size_t)n_parts + 1), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t r = 1; r <= n_parts; ++r) cuts_host[r] = std::max(cuts_host[r], cuts_host[r - 1]);   // ranges tile the buffer
  return KMI_OK;
}

extern "C" kmi_status kmi_fastq_find_records_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, int buffer_starts_file,
                                                 const uint64_t *positions_host, uint32_t n_pos, uint64_t *starts_host) {
  using namespace kmi;
  if (!ctx || !positions_host || !starts_host) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_pos == 0) return KMI_OK;
  void *p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(uint64_t) * (size_t)n_pos, &p));
  KMI_HIP(ctx, hipMemcpyAsync(p, positions_host, sizeof(uint64_t) * n_pos, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(fastq_find_records_at_kernel, dim3((n_pos + 63) / 64), dim3(64), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, n_pos, (uint64_t *)p,
                     buffer_starts_file != 0);
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipMemcpyAsync(starts_host, p, sizeof(uint64_t) * n_pos, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

namespace kmi {
}  // namespace kmi
