// kmi_extract.h -- tile front end shared by the FASTQ kernels (scan, extract, fused build).
// See kmi_extract.hip for the formulation.
#pragma once
#include "kmi_block.h"
#include "kmi_internal.h"

namespace kmi {

template <int NW_, int BITS_, int C_, int NT_> struct ExCfgT {
  static constexpr int NW = NW_, BITS = BITS_;
  static constexpr int C = C_;                                 // bytes per thread (8 or 16)
  static constexpr int NT = NT_;                               // threads per workgroup
  static constexpr int TILE = NT * C;                          // bytes per tile
  static constexpr int KMAX = 64 * NW / BITS;                  // largest k for this word count
  static constexpr int HALO_CHUNKS = (KMAX - 1 + C - 1) / C;
  static constexpr int CHUNKS = NT + HALO_CHUNKS;
  static constexpr int NE = (C - 1 + KMAX + 63) / 64;          // eol words per thread (normalised)
  static constexpr int E_RAW = 2 * NE + 1;                     // raw eol dwords per thread
  static constexpr int NR = (BITS * (C - 1) + 64 * NW + 31) / 32 + 1;  // normalised stream dwords
  static constexpr int S_RAW = NR + 1;
  static constexpr int EOL_DW = (CHUNKS * C + 31) / 32 + E_RAW + 1;
  static constexpr int STREAM_DW = (CHUNKS * C * BITS + 31) / 32 + S_RAW + 1;
  static constexpr uint32_t CMASK = (1u << C) - 1u;
};
// default geometry: 16 bytes per thread for one-word k-mers; tile = 8 KB / 4 KB / 2 KB / 2 KB
template <int NW, int BITS> using ExCfg = ExCfgT<NW, BITS, (NW == 1) ? 16 : 8, (NW <= 2) ? 512 : 256>;
// same tile with 8 bytes per thread (twice the threads for one-word k-mers): used where LDS
// allows only one workgroup per CU
template <int NW, int BITS> using ExCfgWide = ExCfgT<NW, BITS, 8, ExCfg<NW, BITS>::TILE / 8>;

struct TileInfo {
  uint32_t lines;     // line starts in the tile
  uint32_t win[4];    // EOL-free k-windows starting in the tile, by (local line count & 3)
};

// ---- chunk load: C bytes at byte offset g (zero-filled past n_bytes); returns #valid bytes
template <int C> __device__ __forceinline__ int load_chunk(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t g,
                                                          uint32_t (&dw)[C / 4]) {
  if (g + C <= n_bytes) {
    if constexpr (C == 16) {
      uint4 v = *reinterpret_cast<const uint4 *>(bytes + g);
      dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w;
    } else {
      uint2 v = *reinterpret_cast<const uint2 *>(bytes + g);
      dw[0] = v.x; dw[1] = v.y;
    }
    return C;
  }
#pragma unroll
  for (int i = 0; i < C / 4; ++i) dw[i] = 0;
  int n = (g < n_bytes) ? (int)(n_bytes - g) : 0;
  for (int i = 0; i < n; ++i) dw[i >> 2] |= (uint32_t)bytes[g + i] << (8 * (i & 3));
  return n;
}

template <int C> __device__ __forceinline__ void store_eol_bits(uint32_t *s_eol, int chunk, uint32_t eol) {
  if constexpr (C == 16) reinterpret_cast<uint16_t *>(s_eol)[chunk] = (uint16_t)eol;
  else reinterpret_cast<uint8_t *>(s_eol)[chunk] = (uint8_t)eol;
}

template <int BITS, int C> __device__ __forceinline__ void store_stream_bits(uint32_t *s_stream, int chunk, uint64_t st) {
  constexpr int NB = BITS * C / 8;  // bytes per chunk: 4, 6, 2 or 3
  if constexpr (NB == 4) {
    s_stream[chunk] = (uint32_t)st;
  } else if constexpr (NB == 2) {
    reinterpret_cast<uint16_t *>(s_stream)[chunk] = (uint16_t)st;
  } else if constexpr (NB == 6) {
    uint16_t *p = reinterpret_cast<uint16_t *>(s_stream) + 3 * chunk;
    p[0] = (uint16_t)st; p[1] = (uint16_t)(st >> 16); p[2] = (uint16_t)(st >> 32);
  } else {
    uint8_t *p = reinterpret_cast<uint8_t *>(s_stream) + NB * chunk;
#pragma unroll
    for (int i = 0; i < NB; ++i) p[i] = (uint8_t)(st >> (8 * i));
  }
}

// per-thread view of the EOL bit array: bits [C*j, C*j + C-1+KMAX) normalised to bit 0
template <typename Cfg> __device__ __forceinline__ void load_eol_view(const uint32_t *s_eol, int j, uint64_t (&e)[Cfg::NE]) {
  const int bit0 = Cfg::C * j, d0 = bit0 >> 5, sh = bit0 & 31;
  uint32_t raw[Cfg::E_RAW];
#pragma unroll
  for (int i = 0; i < Cfg::E_RAW; ++i) raw[i] = s_eol[d0 + i];
#pragma unroll
  for (int w = 0; w < Cfg::NE; ++w) {
    uint32_t lo = sh ? ((raw[2 * w] >> sh) | (raw[2 * w + 1] << (32 - sh))) : raw[2 * w];
    uint32_t hi = sh ? ((raw[2 * w + 1] >> sh) | (raw[2 * w + 2] << (32 - sh))) : raw[2 * w + 1];
    e[w] = ((uint64_t)hi << 32) | lo;
  }
}

template <typename Cfg> __device__ __forceinline__ void load_stream_view(const uint32_t *s_stream, int j, uint32_t (&r)[Cfg::NR]) {
  const int bit0 = Cfg::C * Cfg::BITS * j, d0 = bit0 >> 5, sh = bit0 & 31;
  uint32_t raw[Cfg::S_RAW];
#pragma unroll
  for (int i = 0; i < Cfg::S_RAW; ++i) raw[i] = s_stream[d0 + i];
#pragma unroll
  for (int i = 0; i < Cfg::NR; ++i) r[i] = sh ? ((raw[i] >> sh) | (raw[i + 1] << (32 - sh))) : raw[i];
}

// common front end of passes 1 and 3: classify own chunk (+ halo chunk), publish the EOL bits
// (and optionally the stream), derive line starts and the block-exclusive line count.
template <typename Cfg, bool WITH_STREAM>
__device__ __forceinline__ void tile_front(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t tile0,
                                           uint32_t *s_eol, uint32_t *s_stream, uint32_t *s_scan,
                                           uint32_t (&dw)[Cfg::C / 4], uint32_t &eol, uint32_t &ls,
                                           uint32_t &lines_before_local, uint32_t &lines_total) {
  constexpr int C = Cfg::C;
  constexpr int BITS = Cfg::BITS;
  const int j = threadIdx.x;
  uint64_t st;
  int nv = load_chunk<C>(bytes, n_bytes, tile0 + (uint64_t)j * C, dw);
  classify_chunk<BITS, C>(dw, nv, eol, st);
  store_eol_bits<C>(s_eol, j, eol);
  if (WITH_STREAM) store_stream_bits<BITS, C>(s_stream, j, st);
  if (j < Cfg::HALO_CHUNKS) {
    uint32_t hdw[C / 4]; uint32_t he; uint64_t hs;
    int hnv = load_chunk<C>(bytes, n_bytes, tile0 + (uint64_t)(Cfg::NT + j) * C, hdw);
    classify_chunk<BITS, C>(hdw, hnv, he, hs);
    store_eol_bits<C>(s_eol, Cfg::NT + j, he);
    if (WITH_STREAM) store_stream_bits<BITS, C>(s_stream, Cfg::NT + j, hs);
  }
  __syncthreads();
  bool prev_eol;
  if (j > 0) {
    const int pb = C * j - 1;
    prev_eol = (s_eol[pb >> 5] >> (pb & 31)) & 1u;
  } else {
    // the partition starts at a record start: treat the byte before it as EOL
    prev_eol = (tile0 == 0) ? true : is_eol(bytes[tile0 - 1]);
  }
  ls = line_starts(eol, prev_eol, Cfg::CMASK);
  lines_before_local = block_exclusive_scan<uint32_t>((uint32_t)__builtin_popcount(ls), s_scan, &lines_total);
}


// Visit every valid k-mer start of this thread's chunk in position order.
// f(p, rc words, fwd words) with p the byte index inside the chunk; `valid` from the caller.
template <typename Cfg, typename F>
__device__ __forceinline__ void for_each_chunk_kmer(const uint32_t *s_stream, uint32_t valid, const KShape &shape, F f) {
  constexpr int NW = Cfg::NW, BITS = Cfg::BITS;
  if (!valid) return;
  uint32_t r[Cfg::NR];
  load_stream_view<Cfg>(s_stream, threadIdx.x, r);
#pragma unroll
  for (int p = 0; p < Cfg::C; ++p) {
    if ((valid >> p) & 1u) {
      uint64_t rc[NW], fw[NW];
      window_words<NW, Cfg::NR>(r, BITS * p, shape, rc);
      fwd_from_rc<NW, BITS>(rc, fw, shape);
      f(p, rc, fw);
    }
  }
}

// FASTQ marker checks of one chunk (fastq_loader.hpp:392-393,421-422,437-438): returns flag bits
template <typename Cfg>
__device__ __forceinline__ uint32_t fastq_marker_errors(const uint32_t (&dw)[Cfg::C / 4], uint32_t lines_before, uint32_t ls, bool first_chunk) {
  uint32_t cur = lines_before, rest = ls, bad = 0;
  while (rest) {
    uint32_t q = (uint32_t)__builtin_ctz(rest);
    uint32_t ch = (dw[q >> 2] >> (8 * (q & 3))) & 0xffu;
    uint32_t role = cur & 3u;   // index of the line that starts here
    if (role == 0 && ch != '@') bad |= 1u;
    if (role == 2 && ch != '+') bad |= 2u;
    cur += 1; rest &= rest - 1u;
  }
  if (first_chunk && (dw[0] & 0xffu) != '@') bad |= 1u;
  return bad;
}

}  // namespace kmi
