// kmi_fasta.hip -- FASTA bytes -> compacted sequence-character stream on the device.
//
// Reference semantics (src/io/fasta_loader.hpp:485-604 init_parser, :618-723 get_next_record;
// k-mer windows run over the non-EOL characters of a sequence, src/io/kmer_parser.hpp:198-213):
//   * a line starts at the buffer start and after every '\n' ('\r' does not end a line);
//   * a line whose first byte is '>' or ';' is a header line; runs of header / non-header lines
//     form groups; a record = header group followed by a non-header group; its sequence is every
//     byte of that non-header group (EOLs are skipped by the k-mer parsers, windows cross lines);
//   * non-header lines before the first header belong to no record, and shift the sequence
//     indices by one (k/2 in init_parser);
//   * LongSequenceKmerId = file position of the k-mer's first base | sequence index << 40
//     (src/common/sequence.hpp:254-255, kmer_parser.hpp:378-386).
//
// GPU formulation: the kind of the line a byte sits on is a 3-state machine (O = before any
// header, H = header group, S = sequence group) driven by line starts. Every chunk / tile is
// summarised by its state map (3 x 2 bits) and, per incoming state, its number of sequence
// characters and record starts; maps compose associatively, so tiles are resolved with the same
// reduce / scan / apply scheme as FASTQ. A third pass writes the sequence characters compacted
// (BITS per character, complement codes) plus one bit per character marking the first character
// of a record, and optionally the id of every character. After that FASTA is "FASTQ without
// roles": the k-mer kernels of kmi_extract.hip run on the compacted stream (window r is valid iff
// no record starts inside (r, r+k) and r + k <= n_chars).
#include <vector>

#include "kmi_extract.h"

namespace kmi {

enum { FA_O = 0, FA_H = 1, FA_S = 2 };
constexpr uint32_t kFaIdentity = 0x24u;   // 0->0, 1->1, 2->2

__device__ __forceinline__ uint32_t fa_apply(uint32_t map, uint32_t st) { return (map >> (2u * st)) & 3u; }
// first a, then b
__device__ __forceinline__ uint32_t fa_compose(uint32_t a, uint32_t b) {
  return fa_apply(b, fa_apply(a, 0)) | (fa_apply(b, fa_apply(a, 1)) << 2) | (fa_apply(b, fa_apply(a, 2)) << 4);
}

using FaCfg = ExCfgT<1, 2, 16, 512>;   // byte-space geometry: 16 bytes per thread, 8 KB tiles

struct FaChunk { uint32_t out, seq, ev; };   // outgoing state, sequence-character mask, record-start (H->S) mask

// run the line-kind machine over one chunk for a given incoming state
__device__ __forceinline__ FaChunk fa_chunk_apply(uint32_t in, uint32_t eol, uint32_t ls, uint32_t hs) {
  constexpr uint32_t C = FaCfg::C;
  uint32_t state = in, seq = 0, ev = 0, start = 0, rest = ls;
  while (true) {
    const uint32_t q = rest ? (uint32_t)__builtin_ctz(rest) : C;
    if (state == FA_S) seq |= ((1u << q) - 1u) & ~((1u << start) - 1u);
    if (!rest) break;
    if ((hs >> q) & 1u) state = FA_H;
    else { if (state == FA_H) ev |= 1u << q; state = (state == FA_O) ? FA_O : FA_S; }
    start = q; rest &= rest - 1u;
  }
  FaChunk r; r.out = state; r.seq = seq & ~eol & FaCfg::CMASK; r.ev = ev;
  return r;
}

// exclusive scan of state maps over the workgroup (composition in thread order); scratch: NT/64 + 1
__device__ __forceinline__ uint32_t block_exclusive_scan_map(uint32_t m, uint32_t *scratch, uint32_t *total) {
  const uint32_t nw = blockDim.x >> 6;
  uint32_t inc = m;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    uint32_t o = __shfl_up(inc, d, kWave);
    if ((int)lane_id() >= d) inc = fa_compose(o, inc);
  }
  if (lane_id() == kWave - 1) scratch[wave_id()] = inc;
  lds_barrier();
  if (wave_id() == 0) {
    uint32_t w = (lane_id() < nw) ? scratch[lane_id()] : kFaIdentity;
    uint32_t winc = w;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      uint32_t o = __shfl_up(winc, d, kWave);
      if ((int)lane_id() >= d) winc = fa_compose(o, winc);
    }
    uint32_t wexc = __shfl_up(winc, 1, kWave);
    if (lane_id() == 0) wexc = kFaIdentity;
    if (lane_id() < nw) scratch[lane_id()] = wexc;
    if (lane_id() == nw - 1) scratch[nw] = winc;
  }
  lds_barrier();
  uint32_t prev = __shfl_up(inc, 1, kWave);
  if (lane_id() == 0) prev = kFaIdentity;
  const uint32_t res = fa_compose(scratch[wave_id()], prev);
  if (total) *total = scratch[nw];
  lds_barrier();
  return res;
}

struct FaTileInfo { uint32_t map; uint32_t nseq[3]; uint32_t nev[3]; };
struct FaTileSum { uint32_t map; uint64_t nseq[3]; uint64_t nev[3]; };
struct FaTileBase { uint32_t state; uint64_t rank; uint64_t ev; };   // state entering the tile, characters / record starts before it

// bytes -> (eol, newline-derived line starts, header starts) of this thread's chunk
__device__ __forceinline__ void fa_chunk_masks(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t tile0, bool first_ls,
                                               uint32_t *s_nl, uint32_t (&dw)[4], uint32_t &eol, uint32_t &ls, uint32_t &hs) {
  constexpr int C = FaCfg::C;
  const int j = threadIdx.x;
  const uint64_t g = tile0 + (uint64_t)j * C;
  const int nv = load_chunk<C>(bytes, n_bytes, g, dw);
  uint32_t nl = 0, hd = 0;
  eol = 0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const uint32_t c = (dw[i >> 2] >> (8 * (i & 3))) & 0xffu;
    const bool in = i < nv;
    nl |= ((in && c == '\n') ? 1u : 0u) << i;
    eol |= ((!in || c == '\n' || c == '\r') ? 1u : 0u) << i;
    hd |= ((in && (c == '>' || c == ';')) ? 1u : 0u) << i;
  }
  reinterpret_cast<uint16_t *>(s_nl)[j] = (uint16_t)nl;
  lds_barrier();
  bool prev_nl;
  if (j > 0) prev_nl = (reinterpret_cast<uint16_t *>(s_nl)[j - 1] >> (C - 1)) & 1u;
  else prev_nl = (tile0 == 0) ? first_ls : (bytes[tile0 - 1] == '\n');   // first_ls: byte 0 of the buffer opens a line
  const uint32_t inb = (nv >= C) ? FaCfg::CMASK : ((1u << nv) - 1u);
  ls = ((nl << 1) | (prev_nl ? 1u : 0u)) & inb;
  hs = ls & hd;
}

// ---- pass 1: per-tile summaries
__global__ __launch_bounds__((FaCfg::NT)) void fasta_scan_tiles_kernel(const uint8_t *__restrict__ bytes, uint64_t n_bytes, bool first_ls,
                                                                      FaTileInfo *__restrict__ info) {
  __shared__ uint32_t s_nl[FaCfg::NT / 2 + 2];
  __shared__ uint32_t s_scan[FaCfg::NT / 64 + 2];
  __shared__ uint32_t s_acc[6];
  if (threadIdx.x < 6) s_acc[threadIdx.x] = 0;
  uint32_t dw[4], eol, ls, hs;
  fa_chunk_masks(bytes, n_bytes, (uint64_t)blockIdx.x * FaCfg::TILE, first_ls, s_nl, dw, eol, ls, hs);
  FaChunk r[3];
#pragma unroll
  for (uint32_t in = 0; in < 3; ++in) r[in] = fa_chunk_apply(in, eol, ls, hs);
  const uint32_t f = r[0].out | (r[1].out << 2) | (r[2].out << 4);
  uint32_t total;
  const uint32_t g = block_exclusive_scan_map(f, s_scan, &total);
#pragma unroll
  for (uint32_t i0 = 0; i0 < 3; ++i0) {
    const uint32_t tin = fa_apply(g, i0);
    uint32_t ns = 0, ne = 0;
#pragma unroll
    for (uint32_t in = 0; in < 3; ++in) {
      ns = (tin == in) ? (uint32_t)__builtin_popcount(r[in].seq) : ns;
      ne = (tin == in) ? (uint32_t)__builtin_popcount(r[in].ev) : ne;
    }
    ns = wave_reduce_sum(ns); ne = wave_reduce_sum(ne);
    if (lane_id() == 0) { atomicAdd(&s_acc[i0], ns); atomicAdd(&s_acc[3 + i0], ne); }
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    FaTileInfo ti; ti.map = total;
    for (int i = 0; i < 3; ++i) { ti.nseq[i] = s_acc[i]; ti.nev[i] = s_acc[3 + i]; }
    info[blockIdx.x] = ti;
  }
}

// ---- pass 2: offsets over tiles (same three-step scheme as FASTQ, with map composition)
__global__ __launch_bounds__(1024) void fasta_offsets_reduce_kernel(const FaTileInfo *__restrict__ info, uint64_t n_tiles,
                                                                   FaTileSum *__restrict__ sums) {
  __shared__ uint32_t s_scan[1024 / 64 + 2];
  __shared__ unsigned long long s_acc[6];
  if (threadIdx.x < 6) s_acc[threadIdx.x] = 0ull;
  const uint64_t t = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  FaTileInfo ti; ti.map = kFaIdentity;
  for (int i = 0; i < 3; ++i) { ti.nseq[i] = 0; ti.nev[i] = 0; }
  if (t < n_tiles) ti = info[t];
  uint32_t total;
  const uint32_t g = block_exclusive_scan_map(ti.map, s_scan, &total);
#pragma unroll
  for (uint32_t i0 = 0; i0 < 3; ++i0) {
    const uint32_t tin = fa_apply(g, i0);
    uint32_t ns = 0, ne = 0;
#pragma unroll
    for (uint32_t in = 0; in < 3; ++in) { ns = (tin == in) ? ti.nseq[in] : ns; ne = (tin == in) ? ti.nev[in] : ne; }
    unsigned long long a = wave_reduce_sum((unsigned long long)ns), b = wave_reduce_sum((unsigned long long)ne);
    if (lane_id() == 0) { atomicAdd(&s_acc[i0], a); atomicAdd(&s_acc[3 + i0], b); }
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    FaTileSum o; o.map = total;
    for (int i = 0; i < 3; ++i) { o.nseq[i] = s_acc[i]; o.nev[i] = s_acc[3 + i]; }
    sums[blockIdx.x] = o;
  }
}

// in place: sums[b] becomes {state entering block b, characters before it (nseq[0]), record starts before it (nev[0])}
// totals[0] = sequence characters, totals[2] = records
__global__ __launch_bounds__(1024) void fasta_offsets_scan_kernel(FaTileSum *__restrict__ sums, uint64_t n_blocks, uint32_t init_state,
                                                                 uint64_t *__restrict__ totals) {
  __shared__ uint32_t s_scanm[1024 / 64 + 2];
  __shared__ uint64_t s_scan[1024 / 64 + 2];
  uint32_t carry_state = init_state;   // FA_O at the file start; a partition states what its first byte sits on
  uint64_t carry_seq = 0, carry_ev = 0;
  for (uint64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
    const uint64_t b = b0 + threadIdx.x;
    FaTileSum ts; ts.map = kFaIdentity;
    for (int i = 0; i < 3; ++i) { ts.nseq[i] = 0; ts.nev[i] = 0; }
    if (b < n_blocks) ts = sums[b];
    uint32_t tm;
    const uint32_t g = block_exclusive_scan_map(ts.map, s_scanm, &tm);
    const uint32_t st = fa_apply(g, carry_state);
    uint64_t ns = 0, ne = 0;
#pragma unroll
    for (uint32_t in = 0; in < 3; ++in) { ns = (st == in) ? ts.nseq[in] : ns; ne = (st == in) ? ts.nev[in] : ne; }
    uint64_t t1, t2;
    const uint64_t rs = carry_seq + block_exclusive_scan<uint64_t>(ns, s_scan, &t1);
    const uint64_t re = carry_ev + block_exclusive_scan<uint64_t>(ne, s_scan, &t2);
    if (b < n_blocks) { ts.map = st; ts.nseq[0] = rs; ts.nev[0] = re; sums[b] = ts; }
    carry_state = fa_apply(tm, carry_state); carry_seq += t1; carry_ev += t2;
  }
  if (threadIdx.x == 0) { totals[0] = carry_seq; totals[2] = carry_ev; }
}

__global__ __launch_bounds__(1024) void fasta_offsets_apply_kernel(const FaTileInfo *__restrict__ info, uint64_t n_tiles,
                                                                  const FaTileSum *__restrict__ sums, FaTileBase *__restrict__ base) {
  __shared__ uint32_t s_scanm[1024 / 64 + 2];
  __shared__ uint32_t s_scan[1024 / 64 + 2];
  const uint64_t t = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  FaTileInfo ti; ti.map = kFaIdentity;
  for (int i = 0; i < 3; ++i) { ti.nseq[i] = 0; ti.nev[i] = 0; }
  if (t < n_tiles) ti = info[t];
  const FaTileSum bs = sums[blockIdx.x];
  const uint32_t g = block_exclusive_scan_map(ti.map, s_scanm, (uint32_t *)nullptr);
  const uint32_t st = fa_apply(g, bs.map);
  uint32_t ns = 0, ne = 0;
#pragma unroll
  for (uint32_t in = 0; in < 3; ++in) { ns = (st == in) ? ti.nseq[in] : ns; ne = (st == in) ? ti.nev[in] : ne; }
  const uint32_t rs = block_exclusive_scan<uint32_t>(ns, s_scan, (uint32_t *)nullptr);
  const uint32_t re = block_exclusive_scan<uint32_t>(ne, s_scan, (uint32_t *)nullptr);
  if (t < n_tiles) { FaTileBase o; o.state = st; o.rank = bs.nseq[0] + rs; o.ev = bs.nev[0] + re; base[t] = o; }
}

// ---- pass 3: compaction. pk_stream / pk_break must be zero-filled; ids may be null.
template <int BITS>
__global__ __launch_bounds__((FaCfg::NT)) void fasta_compact_kernel(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t file_offset,
                                                                   uint64_t index_shift /* + records before the buffer */, bool first_ls,
                                                                   uint64_t valid_bytes, const FaTileBase *__restrict__ base,
                                                                   uint32_t *__restrict__ pk_stream, uint32_t *__restrict__ pk_break,
                                                                   uint64_t *__restrict__ ids, uint64_t *__restrict__ totals, bool split_n,
                                                                   uint64_t *__restrict__ rec_start, uint32_t *__restrict__ rec_flag, bool rna) {
  constexpr int C = FaCfg::C;
  __shared__ uint32_t s_nl[FaCfg::NT / 2 + 2];
  __shared__ uint32_t s_scanm[FaCfg::NT / 64 + 2];
  __shared__ uint32_t s_scan[FaCfg::NT / 64 + 2];
  __shared__ uint32_t s_bits[(FaCfg::TILE * BITS) / 32 + 4];   // the tile's characters, aligned to the global dword grid
  __shared__ uint32_t s_loc[FaCfg::TILE];                      // ids: every character's byte in the tile | record starts of the tile before it << 13
  static_assert(FaCfg::TILE <= 8192, "s_loc packs the byte offset into 13 bits");
  for (int i = threadIdx.x; i < (FaCfg::TILE * BITS) / 32 + 4; i += FaCfg::NT) s_bits[i] = 0;
  const uint64_t tile0 = (uint64_t)blockIdx.x * FaCfg::TILE;
  const FaTileBase tb = base[blockIdx.x];
  uint32_t dw[4], eol, ls, hs;
  fa_chunk_masks(bytes, n_bytes, tile0, first_ls, s_nl, dw, eol, ls, hs);
  if (rna) {   // uniform: RNA alphabets read U where DNA reads T (kmi_device.h swap_tu_dword)
#pragma unroll
    for (int i = 0; i < 4; ++i) dw[i] = swap_tu_dword(dw[i]);
  }
  uint32_t f = 0;
#pragma unroll
  for (uint32_t in = 0; in < 3; ++in) f |= fa_chunk_apply(in, eol, ls, hs).out << (2 * in);
  const uint32_t g = block_exclusive_scan_map(f, s_scanm, (uint32_t *)nullptr);
  const FaChunk r = fa_chunk_apply(fa_apply(g, tb.state), eol, ls, hs);
  uint32_t tile_chars;
  const uint32_t lr = block_exclusive_scan<uint32_t>((uint32_t)__builtin_popcount(r.seq), s_scan, &tile_chars);
  const uint32_t le = block_exclusive_scan<uint32_t>((uint32_t)__builtin_popcount(r.ev), s_scan, (uint32_t *)nullptr);
  const uint64_t rank0 = tb.rank + lr;                 // rank of this thread's first sequence character
  {
    // characters in the valid range = rank of the first character at or behind byte valid_bytes (partitions only)
    const uint64_t c0 = tile0 + (uint64_t)threadIdx.x * C;
    if (valid_bytes >= c0 && valid_bytes < c0 + C && valid_bytes < n_bytes)
      totals[3] = rank0 + (uint32_t)__builtin_popcount(r.seq & ((1u << (uint32_t)(valid_bytes - c0)) - 1u));
  }
  const uint32_t shift0 = (uint32_t)((tb.rank * BITS) & 31u);   // bit phase of the tile inside its first global dword
  // record starts: the first character of the record is the next sequence character = rank_before(position)
  {
    uint32_t rest = r.ev;
    while (rest) {
      const uint32_t q = (uint32_t)__builtin_ctz(rest);
      const uint64_t rk = rank0 + (uint32_t)__builtin_popcount(r.seq & ((1u << q) - 1u));
      atomicOr(&pk_break[rk >> 5], 1u << (rk & 31u));
      // N_FILTER: first character of record e (e-th record start of the buffer; slot 0 = the lines before the first one)
      if (rec_start) rec_start[tb.ev + le + (uint32_t)__builtin_popcount(r.ev & ((1u << q) - 1u)) + 1u] = rk;
      rest &= rest - 1u;
    }
  }
  // this thread's characters as one bit run, OR-ed into the tile image
  {
    uint64_t run_lo = 0; uint32_t run_hi = 0;   // up to 48 bits
    uint32_t rest = r.seq, m = 0;
    while (rest) {
      const uint32_t p = (uint32_t)__builtin_ctz(rest);
      const uint32_t c = (dw[p >> 2] >> (8 * (p & 3))) & 0xffu;
      const uint64_t cc = comp_code<BITS>(code_of<BITS>(c));
      if (rec_flag && c == 'N')   // NSequenceFilter (filtered_sequence_iterator.hpp:154-165): this record is dropped
        atomicOr(&rec_flag[tb.ev + le + (uint32_t)__builtin_popcount(r.ev & ((2u << p) - 1u))], 1u);
      if (split_n && (c == 'N' || c == 'n')) {
        // NSplitSequencesIterator (filtered_sequence_iterator.hpp:411-440): no window may hold this character. A break bit
        // at character x blocks the windows that start in [x - k + 1, x - 1], so bits at x and x + 1 block [x - k + 1, x].
        const uint64_t rk = rank0 + m;
        atomicOr(&pk_break[rk >> 5], 1u << (rk & 31u));
        atomicOr(&pk_break[(rk + 1) >> 5], 1u << ((rk + 1) & 31u));
      }
      const uint32_t bit = m * BITS;
      if (bit < 64) run_lo |= cc << bit;
      if (bit + BITS > 64) run_hi |= (uint32_t)(cc >> (64 - bit));
      if (ids)   // (written after the barrier, a tile's ids as consecutive words: one 8-byte store per character from here is one
                 // memory transaction per character, 7.8 ms per Gbp)
        s_loc[lr + m] = (threadIdx.x * C + p) | ((le + (uint32_t)__builtin_popcount(r.ev & ((2u << p) - 1u))) << 13);
      ++m; rest &= rest - 1u;
    }
    if (m) {
      const uint32_t bit0 = lr * BITS + shift0, d = bit0 >> 5, sh = bit0 & 31u;
      // 128-bit value (run_hi:run_lo) << sh spread over up to 3 dwords
      const uint64_t lo = run_lo << sh;
      const uint64_t mid = (sh ? (run_lo >> (64 - sh)) : 0ull) | ((uint64_t)run_hi << sh);
      atomicOr(&s_bits[d], (uint32_t)lo);
      if ((uint32_t)(lo >> 32)) atomicOr(&s_bits[d + 1], (uint32_t)(lo >> 32));
      if ((uint32_t)mid) atomicOr(&s_bits[d + 2], (uint32_t)mid);
      if ((uint32_t)(mid >> 32)) atomicOr(&s_bits[d + 3], (uint32_t)(mid >> 32));
    }
  }
  lds_barrier();
  // write the tile image: boundary dwords are shared with the neighbouring tiles
  const uint32_t nbits = tile_chars * BITS + shift0;
  const uint32_t ndw = (nbits + 31) >> 5;
  const uint64_t gd0 = (tb.rank * BITS) >> 5;
  for (uint32_t i = threadIdx.x; i < ndw; i += FaCfg::NT) {
    const uint32_t v = s_bits[i];
    if (i == 0 || i == ndw - 1) { if (v) atomicOr(&pk_stream[gd0 + i], v); }
    else pk_stream[gd0 + i] = v;
  }
  if (ids) {   // uniform
    for (uint32_t i = threadIdx.x; i < tile_chars; i += FaCfg::NT) {
      const uint32_t v = s_loc[i];
      // sequence index = record starts at or before this character - 1 (+1 when the buffer begins with orphan lines)
      const uint64_t seq_index = (uint64_t)tb.ev + (v >> 13) - 1u + index_shift;
      const uint64_t pos = file_offset + tile0 + (v & 8191u);
      ids[tb.rank + i] = (pos & 0xFFFFFFFFFFull) | ((seq_index & 0xFFFFull) << 40);
    }
  }
}

// N_FILTER: every character of a record that holds an 'N' gets a break bit, so no window starts in or reaches into it.
// One thread per word of the break bitmap; its records are found by binary search over the record starts.
__global__ __launch_bounds__(256) void fasta_poison_records_kernel(const uint64_t *__restrict__ rec_start, const uint32_t *__restrict__ rec_flag,
                                                                  uint64_t n_rec /* slots 0 .. n_rec, rec_start[n_rec + 1] = n_chars */,
                                                                  uint64_t n_words, uint32_t *__restrict__ pk_break) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  const uint64_t lo = w * 32, hi = lo + 32;
  uint64_t a = 0, b = n_rec;             // last slot e with rec_start[e] <= lo
  while (a < b) {
    const uint64_t m = (a + b + 1) >> 1;
    if (rec_start[m] <= lo) a = m; else b = m - 1;
  }
  uint32_t mask = 0;
  for (uint64_t e = a; e <= n_rec && rec_start[e] < hi; ++e) {
    if (!rec_flag[e]) continue;
    const uint64_t s0 = rec_start[e] > lo ? rec_start[e] : lo, s1 = rec_start[e + 1] < hi ? rec_start[e + 1] : hi;
    if (s1 > s0) mask |= (uint32_t)(((s1 - lo) >= 32 ? 0xffffffffull : ((1ull << (s1 - lo)) - 1ull)) & ~((1ull << (s0 - lo)) - 1ull));
  }
  if (mask) pk_break[w] |= mask;
}
__global__ __launch_bounds__(256) void fasta_count_dropped_kernel(const uint32_t *__restrict__ rec_flag, uint64_t n_slots,
                                                                 unsigned long long *__restrict__ total) {
  uint32_t c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * 256) c += rec_flag[i] ? 1u : 0u;
  c = wave_reduce_sum(c);
  if (lane_id() == 0 && c) atomicAdd(total, (unsigned long long)c);
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
kmi_status fasta_scan(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, uint64_t file_offset, bool want_ids,
                      FastaScan *out) {
  KShape shape;
  if (!valid_config(cfg, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  const uint64_t n_tiles = (n_bytes + FaCfg::TILE - 1) / FaCfg::TILE;
  const uint64_t n_blocks = (n_tiles + 1023) / 1024;
  void *p;
  KMI_TRY(ws_get(ctx, WS_TILE_INFO, sizeof(FaTileInfo) * (n_tiles + 1), &p)); FaTileInfo *info = (FaTileInfo *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_HDR, sizeof(FaTileBase) * (n_tiles + 1), &p)); FaTileBase *base = (FaTileBase *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(FaTileSum) * (n_blocks + 1), &p)); FaTileSum *sums = (FaTileSum *)p;
  // the compacted arrays can never hold more characters than input bytes; pad for the consumers' halo reads
  const uint64_t cover = (n_bytes / 8192 + 2) * 8192;
  const size_t stream_bytes = cover * shape.bits / 8 + 256, break_bytes = cover / 8 + 256;
  KMI_TRY(ws_get(ctx, WS_PK_STREAM, stream_bytes, &p)); uint32_t *pk_stream = (uint32_t *)p;
  KMI_TRY(ws_get(ctx, WS_PK_EOL, break_bytes, &p)); uint32_t *pk_break = (uint32_t *)p;
  uint64_t *ids = nullptr;
  if (want_ids) { KMI_TRY(ws_get(ctx, WS_FA_IDS, (n_bytes + 1) * sizeof(uint64_t), &p)); ids = (uint64_t *)p; }
  KMI_HIP(ctx, hipMemsetAsync(pk_stream, 0, stream_bytes, ctx->stream));
  KMI_HIP(ctx, hipMemsetAsync(pk_break, 0, break_bytes, ctx->stream));
  uint8_t first = 0;
  KMI_HIP(ctx, hipMemcpyAsync(&first, bytes_dev, 1, hipMemcpyDeviceToHost, ctx->stream));
  // a partition of a FASTA file (kmi_ctx_set_fasta_partition) brings what init_parser learns from the neighbours
  const bool part = ctx->fa_part_set;
  const bool first_ls = part ? ctx->fa_part.at_line_start != 0 : true;
  const uint32_t init_state = part ? ctx->fa_part.start_state : (uint32_t)FA_O;
  const uint64_t valid_bytes = (part && ctx->fa_part.valid_bytes < n_bytes) ? ctx->fa_part.valid_bytes : (uint64_t)n_bytes;
  {
    ProfScope ps(ctx, "fasta_scan_tiles", n_bytes);
    hipLaunchKernelGGL(fasta_scan_tiles_kernel, dim3((unsigned)n_tiles), dim3(FaCfg::NT), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, first_ls,
                       info);
  }
  {
    ProfScope ps(ctx, "fasta_scan_offsets", n_tiles);
    hipLaunchKernelGGL(fasta_offsets_reduce_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, ctx->stream, (const FaTileInfo *)info, n_tiles, sums);
    hipLaunchKernelGGL(fasta_offsets_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums, n_blocks, init_state, ctx->d_totals);
    hipLaunchKernelGGL(fasta_offsets_apply_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, ctx->stream, (const FaTileInfo *)info, n_tiles,
                       (const FaTileSum *)sums, base);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipMemcpyAsync(ctx->h_totals, ctx->d_totals, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // init_parser: a leading non-header group shifts the sequence indices by one; a partition adds the records before it
  const uint64_t index_shift = part ? (uint64_t)ctx->fa_part.index_shift + ctx->fa_part.records_before
                                    : ((first == '>' || first == ';') ? 0u : 1u);
  const bool drop_n = cfg->seq_filter == KMI_SEQ_N_FILTER;
  const uint64_t n_rec = ctx->h_totals[2], n_chars_total = ctx->h_totals[0];
  uint64_t *rec_start = nullptr; uint32_t *rec_flag = nullptr;
  if (drop_n) {
    KMI_TRY(ws_get(ctx, WS_PK_NB, (n_rec + 3) * (sizeof(uint64_t) + sizeof(uint32_t)) + 64, &p));
    rec_start = (uint64_t *)p; rec_flag = (uint32_t *)(rec_start + n_rec + 3);
    // every record start writes its slot (characters before it); slot 0 = the lines before the first start, slot
    // n_rec + 1 = the end
    std::vector<uint64_t> init(n_rec + 3, n_chars_total);
    init[0] = 0;
    KMI_HIP(ctx, hipMemcpyAsync(rec_start, init.data(), sizeof(uint64_t) * (n_rec + 3), hipMemcpyHostToDevice, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KMI_HIP(ctx, hipMemsetAsync(rec_flag, 0, sizeof(uint32_t) * (n_rec + 3), ctx->stream));
  }
  {
    ProfScope ps(ctx, "fasta_compact", n_bytes);
    if (shape.bits == 2)
      hipLaunchKernelGGL((fasta_compact_kernel<2>), dim3((unsigned)n_tiles), dim3(FaCfg::NT), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes,
                         file_offset, index_shift, first_ls, valid_bytes, (const FaTileBase *)base, pk_stream, pk_break, ids, ctx->d_totals,
                         cfg->seq_filter == KMI_SEQ_N_SPLIT, rec_start, rec_flag, is_rna(cfg));
    else if (shape.bits == 3)
      hipLaunchKernelGGL((fasta_compact_kernel<3>), dim3((unsigned)n_tiles), dim3(FaCfg::NT), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes,
                         file_offset, index_shift, first_ls, valid_bytes, (const FaTileBase *)base, pk_stream, pk_break, ids, ctx->d_totals,
                         cfg->seq_filter == KMI_SEQ_N_SPLIT, rec_start, rec_flag, is_rna(cfg));
    else
      hipLaunchKernelGGL((fasta_compact_kernel<4>), dim3((unsigned)n_tiles), dim3(FaCfg::NT), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes,
                         file_offset, index_shift, first_ls, valid_bytes, (const FaTileBase *)base, pk_stream, pk_break, ids, ctx->d_totals,
                         cfg->seq_filter == KMI_SEQ_N_SPLIT, rec_start, rec_flag, is_rna(cfg));
  }
  uint64_t dropped = 0;
  if (drop_n && n_chars_total > 0) {
    const uint64_t n_words = (n_chars_total + 31) / 32;
    hipLaunchKernelGGL(fasta_poison_records_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t *)rec_start, (const uint32_t *)rec_flag, n_rec, n_words, pk_break);
    KMI_HIP(ctx, hipMemsetAsync(ctx->d_totals + 1, 0, sizeof(uint64_t), ctx->stream));
    hipLaunchKernelGGL(fasta_count_dropped_kernel, dim3(64), dim3(256), 0, ctx->stream, (const uint32_t *)rec_flag, n_rec + 1,
                       (unsigned long long *)(ctx->d_totals + 1));
    KMI_HIP(ctx, hipMemcpyAsync(&dropped, ctx->d_totals + 1, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  KMI_HIP(ctx, hipGetLastError());
  out->n_chars = ctx->h_totals[0];
  out->n_seqs = ctx->h_totals[2] - dropped;
  out->n_valid = out->n_chars;
  if (valid_bytes < (uint64_t)n_bytes) {   // totals[3] was written by the compaction pass
    KMI_HIP(ctx, hipMemcpyAsync(ctx->h_totals + 3, ctx->d_totals + 3, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->n_valid = ctx->h_totals[3];
  }
  out->pk_break = (const uint8_t *)pk_break;
  out->pk_stream = (const uint8_t *)pk_stream;
  out->ids_by_rank = ids;
  out->n_cover = cover;
  return KMI_OK;
}


// ---- FASTA partition bookkeeping on the device (file.hpp:1436-1610, fasta_loader.hpp:202-470): what FASTAParser::init_parser
// learns from the neighbouring ranks, for every block of an n_parts-way split of a buffer that sits in HBM. The state of the
// line-kind machine and the records started so far at ANY byte are the tile bases of the scan above (prefix sums over the tile
// summaries) plus a walk of at most one 8 KB tile; the end of a block's overlap (k - 1 further sequence characters) is a short
// walk forward. One thread per block: the walks are a few thousand bytes.
struct FaPart { uint64_t begin, end, valid_bytes, start_state, at_line_start, records_before, index_shift; };

__global__ void fasta_partition_cuts_kernel(const uint8_t *__restrict__ bytes, uint64_t n, uint32_t n_parts, uint32_t k, const FaTileBase *__restrict__ base,
                                            FaPart *__restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_parts) return;
  auto cut = [&](uint32_t i) -> uint64_t { return i >= n_parts ? n : n / n_parts * i + (n % n_parts) * i / n_parts; };
  auto line_start = [&](uint64_t i) { return i == 0 || bytes[i - 1] == '\n'; };
  auto step = [&](uint32_t &state, uint64_t &ev, uint8_t c) {   // the line that starts with byte c
    if (c == '>' || c == ';') state = FA_H;
    else { if (state == FA_H) ++ev; state = (state == FA_O) ? FA_O : FA_S; }
  };
  // machine state just before byte pos is looked at (kind of the line byte pos - 1 sits on), records started before pos
  auto state_at = [&](uint64_t pos, uint32_t &state, uint64_t &ev) {
    const uint64_t t = pos / FaCfg::TILE, t0 = t * FaCfg::TILE;
    state = base[t].state; ev = base[t].ev;
    for (uint64_t i = t0; i < pos; ++i) if (line_start(i)) step(state, ev, bytes[i]);
  };
  const uint64_t b = cut(r), e = cut(r + 1);
  FaPart o;
  o.index_shift = (n && (bytes[0] == '>' || bytes[0] == ';')) ? 0u : 1u;
  if (b >= n) { o.begin = n; o.end = n; o.valid_bytes = 0; o.start_state = FA_O; o.at_line_start = 1; o.records_before = 0; out[r] = o; return; }
  uint32_t st; uint64_t ev;
  state_at(b, st, ev);
  const bool ls = line_start(b);
  uint32_t start_state = st;          // a line that STARTS at b is classified by the machine itself from the previous line's kind
  if (!ls) start_state = st;          // (st is the kind of the line b sits on: its start lies before b)
  o.begin = b; o.valid_bytes = e - b;
  o.start_state = (b == 0) ? (uint32_t)FA_O : start_state;
  o.at_line_start = ls ? 1u : 0u;
  o.records_before = ev;
  // the overlap: bytes up to and including the (k - 1)-th sequence character at or behind e
  uint64_t end = e;
  if (k > 1 && e < n) {
    uint32_t se; uint64_t ee;
    state_at(e, se, ee);
    uint32_t need = k - 1u;
    uint64_t i = e;
    for (; i < n && need; ++i) {
      const uint8_t c = bytes[i];
      if (line_start(i)) step(se, ee, c);
      if (se == FA_S && c != '\n' && c != '\r') --need;
    }
    end = need ? n : i;
  }
  o.end = end < e ? e : (end > n ? n : end);
  out[r] = o;
}

}  // namespace kmi

extern "C" kmi_status kmi_fasta_partition_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, uint32_t n_parts, uint32_t k,
                                              uint64_t *begin_end_host, kmi_fasta_partition *parts_host) {
  using namespace kmi;
  if (!ctx || !begin_end_host || !parts_host || n_parts == 0 || k == 0) return KMI_ERR_INVALID;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  if (n_bytes == 0) {
    for (uint32_t r = 0; r < n_parts; ++r) {
      begin_end_host[2 * r] = begin_end_host[2 * r + 1] = 0;
      kmi_fasta_partition q; memset(&q, 0, sizeof(q)); q.at_line_start = 1; q.index_shift = 1; parts_host[r] = q;
    }
    return KMI_OK;
  }
  const uint64_t n_tiles = (n_bytes + FaCfg::TILE - 1) / FaCfg::TILE;
  const uint64_t n_blocks = (n_tiles + 1023) / 1024;
  void *p;
  KMI_TRY(ws_get(ctx, WS_TILE_INFO, sizeof(FaTileInfo) * (n_tiles + 1), &p)); FaTileInfo *info = (FaTileInfo *)p;
  KMI_TRY(ws_get(ctx, WS_TILE_HDR, sizeof(FaTileBase) * (n_tiles + 1), &p)); FaTileBase *base = (FaTileBase *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(FaTileSum) * (n_blocks + 1) + sizeof(FaPart) * ((size_t)n_parts + 1), &p)); FaTileSum *sums = (FaTileSum *)p;
  FaPart *d_parts = (FaPart *)(sums + n_blocks + 1);
  {
    ProfScope ps(ctx, "fasta_scan_tiles", n_bytes);
    hipLaunchKernelGGL(fasta_scan_tiles_kernel, dim3((unsigned)n_tiles), dim3(FaCfg::NT), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, true, info);
  }
  {
    ProfScope ps(ctx, "fasta_scan_offsets", n_tiles);
    hipLaunchKernelGGL(fasta_offsets_reduce_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, ctx->stream, (const FaTileInfo *)info, n_tiles, sums);
    hipLaunchKernelGGL(fasta_offsets_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums, n_blocks, (uint32_t)FA_O, ctx->d_totals);
    hipLaunchKernelGGL(fasta_offsets_apply_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, ctx->stream, (const FaTileInfo *)info, n_tiles,
                       (const FaTileSum *)sums, base);
  }
  hipLaunchKernelGGL(fasta_partition_cuts_kernel, dim3((n_parts + 63) / 64), dim3(64), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes, n_parts, k,
                     (const FaTileBase *)base, d_parts);
  KMI_HIP(ctx, hipGetLastError());
  std::vector<FaPart> h(n_parts);
  KMI_HIP(ctx, hipMemcpyAsync(h.data(), d_parts, sizeof(FaPart) * n_parts, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t r = 0; r < n_parts; ++r) {
    begin_end_host[2 * r] = h[r].begin; begin_end_host[2 * r + 1] = h[r].end;
    kmi_fasta_partition q; memset(&q, 0, sizeof(q));
    q.valid_bytes = h[r].valid_bytes; q.start_state = (uint32_t)h[r].start_state; q.at_line_start = (uint32_t)h[r].at_line_start;
    q.records_before = h[r].records_before; q.index_shift = (uint32_t)h[r].index_shift;
    parts_host[r] = q;
  }
  return KMI_OK;
}

// The line-kind machine over bytes [0, n_bytes) of a FASTA buffer in HBM, as a transfer function: for every state the machine can
// enter the range in (KMI_FA_OUTSIDE / HEADER / SEQUENCE), the state it leaves it in and the records (header group -> sequence
// group transitions) that start inside. What a rank tells the others about its block, so that every rank can work out where ITS
// block starts (fasta_loader.hpp:232-456 does this with collectives over the ranks' first / last lines; file.hpp:1436-1610):
// summaries compose left to right. out6 = {out[O], records[O], out[H], records[H], out[S], records[S]}.
extern "C" kmi_status kmi_fasta_block_summary_dev(kmi_ctx *ctx, const uint8_t *bytes_dev, size_t n_bytes, int first_is_line_start, uint64_t *out6) {
  using namespace kmi;
  if (!ctx || !out6) return KMI_ERR_INVALID;
  for (uint32_t in = 0; in < 3; ++in) { out6[2 * in] = in; out6[2 * in + 1] = 0; }
  if (n_bytes == 0) return KMI_OK;
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n_tiles = (n_bytes + FaCfg::TILE - 1) / FaCfg::TILE;
  const uint64_t n_blocks = (n_tiles + 1023) / 1024;
  void *p;
  KMI_TRY(ws_get(ctx, WS_TILE_INFO, sizeof(FaTileInfo) * (n_tiles + 1), &p)); FaTileInfo *info = (FaTileInfo *)p;
  KMI_TRY(ws_get(ctx, WS_MISC, sizeof(FaTileSum) * (n_blocks + 1), &p)); FaTileSum *sums = (FaTileSum *)p;
  {
    ProfScope ps(ctx, "fasta_scan_tiles", n_bytes);
    hipLaunchKernelGGL(fasta_scan_tiles_kernel, dim3((unsigned)n_tiles), dim3(FaCfg::NT), 0, ctx->stream, bytes_dev, (uint64_t)n_bytes,
                       first_is_line_start != 0, info);
    hipLaunchKernelGGL(fasta_offsets_reduce_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, ctx->stream, (const FaTileInfo *)info, n_tiles, sums);
  }
  KMI_HIP(ctx, hipGetLastError());
  std::vector<FaTileSum> h(n_blocks);
  KMI_HIP(ctx, hipMemcpyAsync(h.data(), sums, sizeof(FaTileSum) * n_blocks, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t in = 0; in < 3; ++in) {
    uint32_t st = in; uint64_t ev = 0;
    for (uint64_t b = 0; b < n_blocks; ++b) { ev += h[b].nev[st]; st = (h[b].map >> (2u * st)) & 3u; }
    out6[2 * in] = st; out6[2 * in + 1] = ev;
  }
  return KMI_OK;
}

namespace kmi {
}  // namespace kmi
