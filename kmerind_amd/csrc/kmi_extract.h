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
  static constexpr uint32_t CMASK = (C >= 32) ? 0xffffffffu : ((1u << (C & 31)) - 1u);
};
// default geometry: 16 bytes per thread for one-word k-mers; tile = 8 KB / 4 KB / 2 KB / 2 KB
template <int NW, int BITS> using ExCfg = ExCfgT<NW, BITS, (NW == 1) ? 16 : 8, (NW <= 2) ? 512 : 256>;

// scan pass geometry: 32 bytes per lane for 2-bit one-word k-mers (per-lane overheads -- scans, line bookkeeping --
// amortise over twice the bytes; 64 code bits fill one stream unit), else the default
template <int NW, int BITS> using ScanCfg = ExCfgT<NW, BITS, (NW == 1 && BITS == 2) ? 32 : ExCfg<NW, BITS>::C,
                                                    (NW == 1 && BITS == 2) ? ExCfg<NW, BITS>::TILE / 32 : ExCfg<NW, BITS>::NT>;

struct TileInfo {
  uint32_t lines;     // line starts in the tile
  uint32_t win[4];    // EOL-free k-windows starting in the tile, by (local line count & 3)
  uint32_t marks;     // bit r: a line with (local line index & 3) == r does not start with '@'; bit 4+r: ... with '+'
  uint32_t last[4];   // 1 + tile position of the last line start with (local line index & 3) == r, 0 = none
};

// ---- chunk load: C bytes at byte offset g (zero-filled past n_bytes); returns #valid bytes
template <int C> __device__ __forceinline__ int load_chunk(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t g,
                                                          uint32_t (&dw)[C / 4]) {
  if (g + C <= n_bytes) {
    if constexpr (C == 32) {
      const uint4 v = *reinterpret_cast<const uint4 *>(bytes + g), w = *reinterpret_cast<const uint4 *>(bytes + g + 16);
      dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w; dw[4] = w.x; dw[5] = w.y; dw[6] = w.z; dw[7] = w.w;
    } else if constexpr (C == 16) {
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
  if constexpr (C == 32) s_eol[chunk] = eol;
  else if constexpr (C == 16) reinterpret_cast<uint16_t *>(s_eol)[chunk] = (uint16_t)eol;
  else reinterpret_cast<uint8_t *>(s_eol)[chunk] = (uint8_t)eol;
}

template <int BITS, int C> __device__ __forceinline__ void store_stream_bits(uint32_t *s_stream, int chunk, uint64_t st) {
  constexpr int NB = BITS * C / 8;  // bytes per chunk: 8, 4, 6, 2 or 3
  if constexpr (NB == 8) {
    s_stream[2 * chunk] = (uint32_t)st; s_stream[2 * chunk + 1] = (uint32_t)(st >> 32);
  } else if constexpr (NB == 4) {
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
template <typename Cfg> __device__ __forceinline__ void load_eol_view(const uint32_t *s_eol, int j, uint64_t (&e)[Cfg::NE], int skip = 0) {
  const int bit0 = Cfg::C * j + skip, d0 = bit0 >> 5, sh = bit0 & 31;
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

// ---------------------------------------------------------------------------
// The scan pass materialises the "2-bit pack" stage once: a bitmap of EOL bytes (1 bit per input
// byte) and the packed COMPLEMENT-code stream (BITS per input byte), both plain little-endian
// bit streams in HBM (1/8 + BITS/8 bytes per input byte). Every later pass starts from these
// instead of re-classifying the raw bytes.
// ---------------------------------------------------------------------------
struct PackedInput {
  const uint8_t *eol;      // bit i  <=> input byte i is an EOL (bytes past the end count as EOL)
  const uint8_t *stream;   // bits [BITS*i, BITS*i+BITS) = complement code of input byte i
  uint64_t n_bytes;        // input length
  uint64_t n_cover;        // bytes covered by the arrays (multiple of the scan tile)
  uint64_t n_valid;        // FASTA only: windows start at positions below this (valid range of a partition); else unused
  const uint8_t *brk = nullptr;   // FASTQ with a sequence filter: bit i <=> no k-mer window may cover byte i (EOL, or an N by the
                                  // filter's rule); null = the EOL bitmap decides alone
};

template <int C> __device__ __forceinline__ uint32_t read_eol_unit(const uint8_t *__restrict__ pk, uint64_t g) {
  if constexpr (C == 32) return reinterpret_cast<const uint32_t *>(pk)[g];
  else if constexpr (C == 16) return reinterpret_cast<const uint16_t *>(pk)[g];
  else return pk[g];
}
template <int C> __device__ __forceinline__ void write_eol_unit(uint8_t *__restrict__ pk, uint64_t g, uint32_t e) {
  if constexpr (C == 32) reinterpret_cast<uint32_t *>(pk)[g] = e;
  else if constexpr (C == 16) reinterpret_cast<uint16_t *>(pk)[g] = (uint16_t)e;
  else pk[g] = (uint8_t)e;
}
template <int BITS, int C> __device__ __forceinline__ uint64_t read_stream_unit(const uint8_t *__restrict__ pk, uint64_t g) {
  constexpr int NB = BITS * C / 8;
  if constexpr (NB == 8) return reinterpret_cast<const uint64_t *>(pk)[g];
  else if constexpr (NB == 4) return reinterpret_cast<const uint32_t *>(pk)[g];
  else if constexpr (NB == 2) return reinterpret_cast<const uint16_t *>(pk)[g];
  else if constexpr (NB == 6) {
    const uint16_t *p = reinterpret_cast<const uint16_t *>(pk) + 3 * g;
    return (uint64_t)p[0] | ((uint64_t)p[1] << 16) | ((uint64_t)p[2] << 32);
  } else {
    const uint8_t *p = pk + (uint64_t)NB * g;
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
  }
}
template <int BITS, int C> __device__ __forceinline__ void write_stream_unit(uint8_t *__restrict__ pk, uint64_t g, uint64_t st) {
  constexpr int NB = BITS * C / 8;
  if constexpr (NB == 8) reinterpret_cast<uint64_t *>(pk)[g] = st;
  else if constexpr (NB == 4) reinterpret_cast<uint32_t *>(pk)[g] = (uint32_t)st;
  else if constexpr (NB == 2) reinterpret_cast<uint16_t *>(pk)[g] = (uint16_t)st;
  else if constexpr (NB == 6) {
    uint16_t *p = reinterpret_cast<uint16_t *>(pk) + 3 * g;
    p[0] = (uint16_t)st; p[1] = (uint16_t)(st >> 16); p[2] = (uint16_t)(st >> 32);
  } else {
    uint8_t *p = pk + (uint64_t)NB * g;
#pragma unroll
    for (int i = 0; i < NB; ++i) p[i] = (uint8_t)(st >> (8 * i));
  }
}

// LDS image of the window-break bits of a tile + halo (same layout as the EOL image); returns the image the window
// validity has to be read from
template <typename Cfg>
__device__ __forceinline__ const uint32_t *tile_break_image(const uint8_t *__restrict__ brk, uint64_t n_cover, uint64_t tile,
                                                            const uint32_t *s_eol, uint32_t *s_brk) {
  if (!brk) return s_eol;   // uniform
  constexpr int C = Cfg::C;
  const int j = threadIdx.x;
  const uint64_t n_units = n_cover / C;
  const uint64_t g = tile * Cfg::NT + j;
  store_eol_bits<C>(s_brk, j, g < n_units ? read_eol_unit<C>(brk, g) : Cfg::CMASK);
  if (j < Cfg::HALO_CHUNKS) {
    const uint64_t gh = tile * Cfg::NT + Cfg::NT + j;
    store_eol_bits<C>(s_brk, Cfg::NT + j, gh < n_units ? read_eol_unit<C>(brk, gh) : Cfg::CMASK);
  }
  lds_barrier();
  return s_brk;
}

// front end of the scan pass: classify the raw bytes of this thread's chunk (+ halo), publish
// EOL bits in LDS, write the packed arrays, derive line starts and the block-exclusive line count
template <typename Cfg>
__device__ __forceinline__ void tile_front_bytes(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t tile,
                                                 uint8_t *__restrict__ pk_eol, uint8_t *__restrict__ pk_stream,
                                                 uint32_t *s_eol, uint32_t *s_scan, uint32_t (&dw)[Cfg::C / 4],
                                                 uint32_t &eol, uint32_t &ls, uint32_t &lines_before_local, uint32_t &lines_total,
                                                 bool rna = false) {
  constexpr int C = Cfg::C;
  constexpr int BITS = Cfg::BITS;
  const int j = threadIdx.x;
  const uint64_t tile0 = tile * Cfg::TILE;
  uint64_t st;
  int nv = load_chunk<C>(bytes, n_bytes, tile0 + (uint64_t)j * C, dw);
  if (rna) {   // uniform
#pragma unroll
    for (int i = 0; i < C / 4; ++i) dw[i] = swap_tu_dword(dw[i]);
  }
  classify_chunk<BITS, C>(dw, nv, eol, st);
  store_eol_bits<C>(s_eol, j, eol);
  write_eol_unit<C>(pk_eol, tile * Cfg::NT + j, eol);
  write_stream_unit<BITS, C>(pk_stream, tile * Cfg::NT + j, st);
  if (j < Cfg::HALO_CHUNKS) {
    uint32_t hdw[C / 4]; uint32_t he; uint64_t hs;
    int hnv = load_chunk<C>(bytes, n_bytes, tile0 + (uint64_t)(Cfg::NT + j) * C, hdw);
    classify_chunk<BITS, C>(hdw, hnv, he, hs);   // only the EOL bits of the halo are used: no T/U swap needed
    store_eol_bits<C>(s_eol, Cfg::NT + j, he);
  }
  lds_barrier();
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

// front end of every later pass: the tile's packed units (+ halo) go straight into LDS
template <typename Cfg>
__device__ __forceinline__ void tile_front_packed(const PackedInput &in, uint64_t tile, uint32_t *s_eol, uint32_t *s_stream,
                                                  uint32_t *s_scan, uint32_t &eol, uint32_t &ls, uint32_t &lines_before_local,
                                                  uint32_t &lines_total) {
  constexpr int C = Cfg::C;
  constexpr int BITS = Cfg::BITS;
  const int j = threadIdx.x;
  const uint64_t n_units = in.n_cover / C;
  const uint64_t g = tile * Cfg::NT + j;
  eol = Cfg::CMASK;
  uint64_t st = 0;
  if (g < n_units) { eol = read_eol_unit<C>(in.eol, g); st = read_stream_unit<BITS, C>(in.stream, g); }
  store_eol_bits<C>(s_eol, j, eol);
  store_stream_bits<BITS, C>(s_stream, j, st);
  if (j < Cfg::HALO_CHUNKS) {
    const uint64_t gh = tile * Cfg::NT + Cfg::NT + j;
    uint32_t he = Cfg::CMASK; uint64_t hs = 0;
    if (gh < n_units) { he = read_eol_unit<C>(in.eol, gh); hs = read_stream_unit<BITS, C>(in.stream, gh); }
    store_eol_bits<C>(s_eol, Cfg::NT + j, he);
    store_stream_bits<BITS, C>(s_stream, Cfg::NT + j, hs);
  }
  lds_barrier();
  bool prev_eol;
  if (j > 0) {
    const int pb = C * j - 1;
    prev_eol = (s_eol[pb >> 5] >> (pb & 31)) & 1u;
  } else {
    prev_eol = (g == 0) ? true : ((read_eol_unit<C>(in.eol, g - 1) >> (C - 1)) & 1u);
  }
  ls = line_starts(eol, prev_eol, Cfg::CMASK);
  lines_before_local = block_exclusive_scan<uint32_t>((uint32_t)__builtin_popcount(ls), s_scan, &lines_total);
}

// FASTA (compacted character space, s_eol holds the record-start bits): window r is valid iff no
// record starts at r+1 .. r+k-1 and r + k <= n_chars
template <typename Cfg>
__device__ __forceinline__ uint32_t chunk_valid_mask_fasta(const uint32_t *s_eol, uint32_t k, uint64_t tile0, uint64_t n_chars, uint64_t n_valid) {
  uint64_t e[Cfg::NE];
  load_eol_view<Cfg>(s_eol, threadIdx.x, e, 1);
  uint32_t blocked = 0;
  if (k > 1) { smear_right<Cfg::NE>(e, k - 1); blocked = (uint32_t)e[0]; }
  const uint64_t g = tile0 + (uint64_t)threadIdx.x * Cfg::C;
  uint32_t room = 0;   // positions p with g + p + k <= n_chars and g + p < n_valid
  if (g + k <= n_chars) {
    uint64_t lim = n_chars - k - g + 1;
    const uint64_t lim2 = n_valid > g ? n_valid - g : 0ull;
    lim = lim < lim2 ? lim : lim2;
    room = lim >= (uint64_t)Cfg::C ? Cfg::CMASK : ((1u << (uint32_t)lim) - 1u);
  }
  return ~blocked & room & Cfg::CMASK;
}

// window list from a ready valid mask (one workgroup scan + barrier)
template <typename Cfg>
__device__ __forceinline__ uint32_t tile_window_list_from(uint32_t valid, uint16_t *s_pos, uint32_t *s_scan) {
  uint32_t total;
  uint32_t rank = block_exclusive_scan<uint32_t>((uint32_t)__builtin_popcount(valid), s_scan, &total);
  const uint32_t base = threadIdx.x * Cfg::C;
  while (valid) {
    s_pos[rank++] = (uint16_t)(base + (uint32_t)__builtin_ctz(valid));
    valid &= valid - 1u;
  }
  lds_barrier();
  return total;
}

// Compacted list of the tile's k-mer start positions (byte index inside the tile), in file
// order: s_pos[0..total). Every lane of the later window loop then has real work, instead of
// the ~38 % of lanes that sit on a sequence line.
template <typename Cfg>
__device__ __forceinline__ uint32_t tile_window_list(const uint32_t *s_eol, uint32_t ls, uint32_t lines_before, uint32_t k,
                                                     uint16_t *s_pos, uint32_t *s_scan) {
  uint64_t e[Cfg::NE];
  load_eol_view<Cfg>(s_eol, threadIdx.x, e);
  smear_right<Cfg::NE>(e, k);
  uint32_t valid = ~(uint32_t)e[0] & fastq_seq_role_mask(lines_before, ls, Cfg::CMASK);
  uint32_t total;
  uint32_t rank = block_exclusive_scan<uint32_t>((uint32_t)__builtin_popcount(valid), s_scan, &total);
  const uint32_t base = threadIdx.x * Cfg::C;
  while (valid) {
    s_pos[rank++] = (uint16_t)(base + (uint32_t)__builtin_ctz(valid));
    valid &= valid - 1u;
  }
  lds_barrier();
  return total;
}

// k-mer whose first base is tile byte `pos`: reverse complement = the little-endian window of the
// complement stream, forward = its group reversal
template <typename Cfg>
__device__ __forceinline__ void window_at(const uint32_t *s_stream, uint32_t pos, const KShape &shape,
                                          uint64_t (&rc)[Cfg::NW], uint64_t (&fw)[Cfg::NW]) {
  constexpr int NW = Cfg::NW, BITS = Cfg::BITS;
  const uint32_t bit = BITS * pos, d = bit >> 5, sh = bit & 31u;
  uint32_t raw[2 * NW + 1];
#pragma unroll
  for (int i = 0; i < 2 * NW + 1; ++i) raw[i] = s_stream[d + i];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t lo = __builtin_amdgcn_alignbit(raw[2 * w + 1], raw[2 * w], sh);
    const uint32_t hi = __builtin_amdgcn_alignbit(raw[2 * w + 2], raw[2 * w + 1], sh);
    rc[w] = ((uint64_t)hi << 32) | lo;
  }
  mask_words<NW>(rc, shape);
  fwd_from_rc<NW, BITS>(rc, fw, shape);
}

// Consecutive windows of one read (an entry of the entry list), one-word 2-bit k-mers. The lane loads 128 stream bits
// once: the first reverse complement is a funnel shift of them, the 7 codes that follow the first window are kept in
// one register, and every further window ROLLS: rc' = rc >> 2 with the new code on top, fw' = fw << 2 with its
// complement at the bottom -- no LDS reads, no group reversal and no 128-bit shifts after the first window.
// HI: k >= 17 (the top code sits in the high word; k <= 16 runs on 32-bit registers). CANON: smaller of the two strands.
// f(j, key) for the windows j < len (1 <= len <= 8) of the entry that starts at tile-image position pos.
// FULL: every lane's entry holds 8 windows (the rule for whole reads: the caller checks it for the wavefront), so the
// per-window length tests and their exec-mask bookkeeping are left out.
template <bool HI, bool CANON, bool FULL, typename F>
__device__ __forceinline__ void roll_entry_windows(const uint32_t *s_stream, uint32_t pos, uint32_t len, const KShape &shape, F f) {
  const uint32_t bit = 2u * pos, d = bit >> 5, sh = bit & 31u;
  const uint64_t lo = (uint64_t)s_stream[d] | ((uint64_t)s_stream[d + 1] << 32);
  const uint64_t hi = (uint64_t)s_stream[d + 2] | ((uint64_t)s_stream[d + 3] << 32);
  const uint32_t kb = 2u * shape.k;                       // bits of a k-mer
  const uint64_t mask = low_mask64(kb);
  uint64_t r1[1], f1[1];
  r1[0] = (sh ? ((lo >> sh) | (hi << (64u - sh))) : lo) & mask;
  fwd_from_rc<1, 2>(r1, f1, shape);
  const uint32_t s2 = sh + kb;                            // 2 .. 95: first code behind the first window
  const uint32_t nb = (uint32_t)(s2 < 64u ? ((lo >> s2) | (hi << (64u - s2))) : (hi >> (s2 - 64u)));
  if constexpr (HI) {
    uint64_t rc = r1[0], fw = f1[0];
    const uint64_t keep = ((uint64_t)(uint32_t)(mask >> 32) << 32) | 0xffffffffull;   // low word all ones: no AND there
    const uint32_t top = kb - 34u;                                                     // bit of the top code inside the high word
    uint64_t key[1];
    key[0] = CANON ? (fw < rc ? fw : rc) : fw;
    f(0u, key);
#pragma unroll
    for (uint32_t j = 1; j < 8u; ++j) {
      if (FULL || j < len) {
        const uint32_t b = (nb >> (2u * (j - 1u))) & 3u;
        rc = (rc >> 2) | ((uint64_t)(b << top) << 32);
        fw = ((fw << 2) | (uint64_t)(b ^ 3u)) & keep;
        key[0] = CANON ? (fw < rc ? fw : rc) : fw;
        f(j, key);
      }
    }
  } else {
    uint32_t rc = (uint32_t)r1[0], fw = (uint32_t)f1[0];
    const uint32_t m32 = (uint32_t)mask, top = kb - 2u;
    uint64_t key[1];
    key[0] = CANON ? (fw < rc ? fw : rc) : fw;
    f(0u, key);
#pragma unroll
    for (uint32_t j = 1; j < 8u; ++j) {
      if (FULL || j < len) {
        const uint32_t b = (nb >> (2u * (j - 1u))) & 3u;
        rc = (rc >> 2) | (b << top);
        fw = ((fw << 2) | (b ^ 3u)) & m32;
        key[0] = CANON ? (fw < rc ? fw : rc) : fw;
        f(j, key);
      }
    }
  }
}

// key stored by the map for a parsed k-mer (kmer_index.hpp:436-481): forward strand, or the
// smaller of forward / reverse complement
template <int NW>
__device__ __forceinline__ void select_strand(const uint64_t (&rc)[NW], const uint64_t (&fw)[NW], bool canonical, uint64_t (&key)[NW]) {
  const bool use_fw = !canonical || less_words<NW>(fw, rc);
#pragma unroll
  for (int w = 0; w < NW; ++w) key[w] = use_fw ? fw[w] : rc[w];
}

// ---------------------------------------------------------------------------
// EOL-bitmap scans used by the seq/qual length rule of FASTQParser::get_next_record (fastq_loader.hpp:454-463)
// ---------------------------------------------------------------------------
// Bitmap words come from a per-wavefront LDS image (the tile with 1 KB of context on either side, which holds
// whole records of ordinary reads) and from HBM only beyond it.
struct EolBits {
  const uint32_t *g; uint64_t n_words;        // global bitmap; bit set = EOL; words >= n_words count as all-EOL
  const uint32_t *img; uint64_t w0; uint32_t nw;   // LDS image of words [w0, w0 + nw)
  __device__ __forceinline__ uint32_t word(uint64_t wi) const {
    const uint64_t d = wi - w0;
    if (d < (uint64_t)nw) return img[d];
    return wi < n_words ? g[wi] : 0xffffffffu;
  }
};
__device__ __forceinline__ uint64_t eol_next_set(const EolBits &b, uint64_t p) {   // first EOL position >= p
  uint64_t wi = p >> 5;
  if (wi >= b.n_words) return p;
  uint32_t bits = b.word(wi) & (0xffffffffu << (p & 31u));
  while (bits == 0u) { if (++wi >= b.n_words) return wi << 5; bits = b.word(wi); }
  return (wi << 5) + (uint32_t)__builtin_ctz(bits);
}
template <bool SET> __device__ __forceinline__ int64_t eol_prev(const EolBits &b, int64_t p) {   // last position < p whose EOL bit == SET, -1 if none
  if (p <= 0 || b.n_words == 0) return -1;
  uint64_t q = (uint64_t)p - 1;
  if (q >= b.n_words * 32ull) { if (SET) return (int64_t)q; q = b.n_words * 32ull - 1; }
  uint64_t wi = q >> 5;
  uint32_t w = b.word(wi);
  uint32_t bits = (SET ? w : ~w) & (0xffffffffu >> (31u - (uint32_t)(q & 31u)));
  while (bits == 0u) { if (wi == 0) return -1; --wi; w = b.word(wi); bits = SET ? w : ~w; }
  return (int64_t)((wi << 5) + 31u - (uint32_t)__builtin_clz(bits));
}
// g = position of the first byte of a quality line
__device__ __forceinline__ bool fastq_lengths_differ(const EolBits &b, uint64_t g) {
  const uint64_t len_qual = eol_next_set(b, g) - g;
  const int64_t plus_last = eol_prev<false>(b, (int64_t)g);          // last byte of the '+' line
  if (plus_last < 0) return false;
  const int64_t gap = eol_prev<true>(b, plus_last);                   // an EOL between the sequence and the '+' line
  if (gap < 0) return false;
  const int64_t seq_last = eol_prev<false>(b, gap);                   // last byte of the sequence line
  if (seq_last < 0) return false;
  const int64_t seq_first = eol_prev<true>(b, seq_last) + 1;          // (-1 + 1 = 0: the sequence line opens the buffer)
  return (uint64_t)(seq_last - seq_first + 1) != len_qual;
}


}  // namespace kmi
