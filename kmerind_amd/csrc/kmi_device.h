// kmi_device.h -- per-thread building blocks shared by the HIP kernels.
//
// Everything here is a pure function of its arguments (no LDS, no atomics), marked
// __host__ __device__ so tests/cpu/test_device_fns.cpp can run the very same code on
// the CPU against the oracle. The kernels in kmi_extract.hip / kmi_index.hip are the
// glue (tiling, LDS staging, scans) around these.
//
// Reference semantics restated here:
//   alphabets         src/common/alphabets.hpp:139-161 (DNA), :225-248 (DNA6 == DNA5)
//   Kmer layout       src/common/kmer.hpp:116-177  (data[0] = least significant word,
//                     newest base in the low bits, pad bits of the top word zero)
//   reverse_complement kmer.hpp:1723-1742 (DNA: group reverse + NOT), :1807-1847
//                     (DNA6: plain bit reversal)
//   operator<         kmer.hpp:820-823 (unsigned compare from the top word)
//   lex_less          src/common/kmer_transform.hpp:108-116
//   murmur / farm     src/index/kmer_hash.hpp:242-311, ext/smhasher/MurmurHash3.cpp:255-335,
//                     ext/farmhash/src/farmhash.cc:373-466,519-529
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KMI_HD __host__ __device__ __forceinline__
#else
#define KMI_HD inline
#endif

namespace kmi {

constexpr int kMaxWords = 4;

// ---------------------------------------------------------------------------
// shape of Kmer<K, Alphabet, uint64_t>  (padding.hpp:67-90)
// ---------------------------------------------------------------------------
struct KShape {
  uint32_t k, bits, n_bits, n_words, n_bytes, pad_bits;
};

KMI_HD KShape make_shape(uint32_t k, uint32_t bits) {
  KShape s;
  s.k = k; s.bits = bits; s.n_bits = k * bits;
  s.n_words = (s.n_bits + 63) / 64;
  s.n_bytes = (s.n_bits + 7) / 8;
  s.pad_bits = s.n_words * 64 - s.n_bits;
  return s;
}

template <int NW> struct Key { uint64_t w[NW]; };

KMI_HD uint64_t low_mask64(uint32_t bits) { return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull); }

// ---------------------------------------------------------------------------
// alphabets.  The packed LDS stream stores the COMPLEMENT code of each base so that a
// little-endian bit window over it is directly the reverse-complement k-mer.
// ---------------------------------------------------------------------------
KMI_HD bool is_eol(uint32_t c) { return c == '\n' || c == '\r'; }

// DNA: A/a=0 C/c=1 G/g=2 T/t=3, everything else (N included) 0.
KMI_HD uint32_t code_dna(uint32_t c) {
  uint32_t x = c & 0xDFu;               // fold case
  uint32_t t = (x >> 1) & 3u;           // A->0 C->1 G->3 T->2
  t ^= t >> 1;                          // A->0 C->1 G->2 T->3
  bool ok = (x == 'A') | (x == 'C') | (x == 'G') | (x == 'T');
  return ok ? t : 0u;
}
// DNA5 (= DNA6): A=1 C=3 G=6 T=4 N/X=7 '-','.'=0 everything else 2.
KMI_HD uint32_t code_dna5(uint32_t c) {
  uint32_t x = c & 0xDFu;
  uint32_t r = 2u;
  r = (x == 'A') ? 1u : r;
  r = (x == 'C') ? 3u : r;
  r = (x == 'G') ? 6u : r;
  r = (x == 'T') ? 4u : r;
  r = ((x == 'N') | (x == 'X')) ? 7u : r;
  r = ((c == '-') | (c == '.')) ? 0u : r;
  return r;
}
// DNA16 (alphabets.hpp:648-733): IUPAC letters as presence bits A=1 C=2 G=4 T/U=8 (M=3 R=5 S=6 V=7 W=9 Y=A H=B K=C D=D
// B=E), '-' and '.' = 0, everything else (N, X, ...) = F. Sixteen letters per 64-bit nibble table.
KMI_HD uint32_t code_dna16(uint32_t c) {
  const uint32_t i = (c & 0xDFu) - (uint32_t)'A';               // letter index, case folded
  const uint64_t t = i < 16u ? 0xfff3fcffb4ffd2e1ull : 0xfaf978865full;
  uint32_t r = i < 26u ? (uint32_t)(t >> (4u * (i & 15u))) & 15u : 15u;
  r = ((c == '-') | (c == '.')) ? 0u : r;
  return r;
}
template <int BITS> KMI_HD uint32_t code_of(uint32_t c) { return BITS == 2 ? code_dna(c) : (BITS == 3 ? code_dna5(c) : code_dna16(c)); }
// complement code: DNA 3-c (alphabets.hpp:172-178); DNA6 = 3-bit reversal (alphabets.hpp:197-210,262-272); DNA16 = 4-bit
// reversal (alphabets.hpp:706-729)
template <int BITS> KMI_HD uint32_t comp_code(uint32_t code) {
  if (BITS == 2) return 3u - code;
  if (BITS == 3) return ((code & 1u) << 2) | (code & 2u) | ((code >> 2) & 1u);
  return ((code & 1u) << 3) | ((code & 2u) << 1) | ((code >> 1) & 2u) | ((code >> 3) & 1u);
}

// ---------------------------------------------------------------------------
// multi-word helpers (NW 64-bit words, w[0] least significant)
// ---------------------------------------------------------------------------
KMI_HD uint64_t brev64(uint64_t x) { return __builtin_bitreverse64(x); }

template <int NW> KMI_HD void shr_words(uint64_t (&x)[NW], uint32_t sh) {  // 0 <= sh < 64
  if (sh == 0) return;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint64_t hi = (w + 1 < NW) ? x[w + 1] : 0ull;
    x[w] = (x[w] >> sh) | (hi << (64 - sh));
  }
}

template <int NW> KMI_HD bool less_words(const uint64_t (&a)[NW], const uint64_t (&b)[NW]) {
  bool lt = false, decided = false;
#pragma unroll
  for (int w = NW - 1; w >= 0; --w) {
    bool ne = a[w] != b[w];
    lt = (!decided && ne) ? (a[w] < b[w]) : lt;
    decided = decided | ne;
  }
  return lt;
}

template <int NW> KMI_HD void mask_words(uint64_t (&x)[NW], const KShape &s) {
  x[NW - 1] &= low_mask64(64 - s.pad_bits);
}

// reverse complement of a k-mer held in NW words.
template <int NW, int BITS> KMI_HD void revcomp_words(const uint64_t (&in)[NW], uint64_t (&out)[NW], const KShape &s) {
  uint64_t t[NW];
  if (BITS == 2) {
    // reverse 2-bit groups and complement: bit-reverse ~x, swap the bits of every pair back
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      uint64_t r = brev64(~in[NW - 1 - w]);
      t[w] = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    }
  } else {
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = brev64(in[NW - 1 - w]);
  }
  shr_words<NW>(t, s.pad_bits);
  if (BITS == 2) mask_words<NW>(t, s);  // ~ turned the (reversed) pad bits on below the shift only for DNA
#pragma unroll
  for (int w = 0; w < NW; ++w) out[w] = t[w];
}

template <int NW, int BITS> KMI_HD void canonical_words(const uint64_t (&in)[NW], uint64_t (&out)[NW], const KShape &s) {
  uint64_t rc[NW];
  revcomp_words<NW, BITS>(in, rc, s);
  bool lt = less_words<NW>(in, rc);
#pragma unroll
  for (int w = 0; w < NW; ++w) out[w] = lt ? in[w] : rc[w];
}

// strand model applied to a k-mer as parsed: what the map stores as key
// (kmer_index.hpp:436-481; bimolecule keeps the canonical representative)
template <int NW, int BITS> KMI_HD void strand_key(const uint64_t (&fwd)[NW], uint64_t (&out)[NW], const KShape &s, uint32_t strand) {
  if (strand == 0) {
#pragma unroll
    for (int w = 0; w < NW; ++w) out[w] = fwd[w];
  } else {
    canonical_words<NW, BITS>(fwd, out, s);
  }
}

// ---------------------------------------------------------------------------
// hashes
// ---------------------------------------------------------------------------
KMI_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
KMI_HD uint64_t rotr64(uint64_t x, int r) { return r == 0 ? x : ((x >> r) | (x << (64 - r))); }
KMI_HD uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

// byte `i` of the little-endian image of the words
template <int NW> KMI_HD uint64_t load_le64(const uint64_t (&w)[NW], uint32_t byte_off) {
  // unaligned 64-bit little-endian load at byte_off from the word array (zero beyond the end)
  uint32_t wi = byte_off >> 3, sh = (byte_off & 7u) * 8u;
  uint64_t lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    lo = (wi == (uint32_t)j) ? w[j] : lo;
    hi = (wi + 1 == (uint32_t)j) ? w[j] : hi;
  }
  return sh == 0 ? lo : ((lo >> sh) | (hi << (64 - sh)));
}

// MurmurHash3_x64_128 over the first `len` bytes (len = n_bytes <= 8*NW) of the k-mer.
template <int NW> KMI_HD void murmur3_x64_128(const uint64_t (&key)[NW], uint32_t len, uint32_t seed, uint64_t &o1, uint64_t &o2) {
  const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
  uint64_t h1 = seed, h2 = seed;
  const uint32_t nblocks = len / 16;
#pragma unroll
  for (int i = 0; i < NW / 2; ++i) {
    if ((uint32_t)i < nblocks) {
      uint64_t k1 = key[2 * i], k2 = key[2 * i + 1];
      k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
      h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
      k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
      h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
  }
  const uint32_t rem = len & 15u;
  if (rem) {
    // tail words: bytes [16*nblocks, len); k-mer pad bits are zero so whole words can be
    // used after masking to `rem` bytes.
    uint64_t t1 = 0, t2 = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      t1 = ((uint32_t)j == 2 * nblocks) ? key[j] : t1;
      t2 = ((uint32_t)j == 2 * nblocks + 1) ? key[j] : t2;
    }
    if (rem < 8) { t1 &= low_mask64(rem * 8); t2 = 0; }
    else if (rem < 16) { t2 &= low_mask64((rem - 8) * 8); }
    if (rem > 8) { t2 *= c2; t2 = rotl64(t2, 33); t2 *= c1; h2 ^= t2; }
    t1 *= c1; t1 = rotl64(t1, 31); t1 *= c2; h1 ^= t1;
  }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  o1 = h1; o2 = h2;
}

// farmhashna::Hash64 for len <= 32 (k-mers are at most 4 words), then
// Hash64WithSeed = HashLen16(Hash64 - k2, seed), then util::DebugTweak unless ndebug.
constexpr uint64_t kFarmK0 = 0xc3a5c85c97cb3127ull;
constexpr uint64_t kFarmK1 = 0xb492b66fbe98f273ull;
constexpr uint64_t kFarmK2 = 0x9ae16a3b2f90404full;

KMI_HD uint64_t farm_len16_mul(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul; a ^= a >> 47;
  uint64_t b = (v ^ a) * mul; b ^= b >> 47;
  return b * mul;
}

template <int NW> KMI_HD uint64_t farm_hash64_with_seed(const uint64_t (&key)[NW], uint32_t len, uint64_t seed, bool ndebug) {
  uint64_t h;
  if (len > 16) {            // HashLen17to32
    uint64_t mul = kFarmK2 + len * 2;
    uint64_t a = load_le64<NW>(key, 0) * kFarmK1;
    uint64_t b = load_le64<NW>(key, 8);
    uint64_t c = load_le64<NW>(key, len - 8) * mul;
    uint64_t d = load_le64<NW>(key, len - 16) * kFarmK2;
    h = farm_len16_mul(rotr64(a + b, 43) + rotr64(c, 30) + d, a + rotr64(b + kFarmK2, 18) + c, mul);
  } else if (len >= 8) {     // HashLen0to16, len >= 8
    uint64_t mul = kFarmK2 + len * 2;
    uint64_t a = load_le64<NW>(key, 0) + kFarmK2;
    uint64_t b = load_le64<NW>(key, len - 8);
    uint64_t c = rotr64(b, 37) * mul + a;
    uint64_t d = (rotr64(a, 25) + b) * mul;
    h = farm_len16_mul(c, d, mul);
  } else if (len >= 4) {
    uint64_t mul = kFarmK2 + len * 2;
    uint64_t a = load_le64<NW>(key, 0) & 0xffffffffull;
    uint64_t b = load_le64<NW>(key, len - 4) & 0xffffffffull;
    h = farm_len16_mul(len + (a << 3), b, mul);
  } else if (len > 0) {
    uint32_t a = (uint32_t)(key[0] & 0xff);
    uint32_t b = (uint32_t)((key[0] >> (8 * (len >> 1))) & 0xff);
    uint32_t c = (uint32_t)((key[0] >> (8 * (len - 1))) & 0xff);
    uint32_t y = a + (b << 8);
    uint32_t z = len + (c << 2);
    uint64_t v = y * kFarmK2 ^ z * kFarmK0;
    h = (v ^ (v >> 47)) * kFarmK2;
  } else {
    h = kFarmK2;
  }
  h = farm_len16_mul(h - kFarmK2, seed, 0x9ddfea08eb382d69ull);
  if (!ndebug) h = ~__builtin_bswap64(h * kFarmK1);
  return h;
}

// ceilLog2(comm_size): the prefix_bits KeyToRank hands to DistHash's constructor
// (src/common/bit_ops.hpp:144-152, src/containers/distributed_unordered_map.hpp:153-156)
KMI_HD uint32_t ceil_log2_u32(uint32_t n) {
  uint32_t b = 0;
  while (b < 32u && (1ull << b) < (uint64_t)n) ++b;
  return b;
}

// bits [lo, lo + 64) of the k-mer value (zero above the top word)
template <int NW> KMI_HD uint64_t kmer_bits_from(const uint64_t (&key)[NW], uint32_t lo) {
  const uint32_t w = lo >> 6, sh = lo & 63u;
  uint64_t a = 0, b = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) { a = ((uint32_t)i == w) ? key[i] : a; b = ((uint32_t)i == w + 1u) ? key[i] : b; }
  return sh ? ((a >> sh) | (b << (64u - sh))) : a;
}

// bliss::kmer::hash::{murmur,farm,identity,cpp_std}<KMER,Prefix>  (src/index/kmer_hash.hpp:156-311)
// prefix_bits: the constructor argument of identity / cpp_std (0 = the class default, 24 / 32); murmur and farm
// ignore it, as the reference does.
template <int NW> KMI_HD uint64_t kmer_hash(const uint64_t (&key)[NW], const KShape &s, uint32_t which, bool prefix, bool farm_ndebug,
                                            uint32_t prefix_bits = 0) {
  if (which == 0) {
    uint64_t h1, h2;
    murmur3_x64_128<NW>(key, s.n_bytes, 42u, h1, h2);
    return prefix ? h2 : h1;
  }
  if (which == 1) return farm_hash64_with_seed<NW>(key, s.n_bytes, prefix ? 83ull : 42ull, farm_ndebug);
  if (which == 2) {
    // identity (kmer_hash.hpp:205-230): Prefix -> getPrefix(min(nBits, prefix_bits)) = the top bits of the k-mer
    // value (kmer.hpp:1203-1221); else getSuffix(min(nBits, 64)) = the low bits (kmer.hpp:1245-1249)
    if (!prefix) {
      const uint32_t sb = s.n_bits < 64u ? s.n_bits : 64u;
      return sb == 64u ? key[0] : (key[0] & ((1ull << sb) - 1ull));
    }
    uint32_t bits = prefix_bits ? prefix_bits : 24u;
    if (bits > s.n_bits) bits = s.n_bits;
    if (bits > 64u) bits = 64u;
    if (bits == 0u) return 0ull;
    const uint64_t v = kmer_bits_from<NW>(key, s.n_bits - bits);
    return bits == 64u ? v : (v & ((1ull << bits) - 1ull));
  }
  // cpp_std (kmer_hash.hpp:154-198) with 64-bit words and libstdc++'s std::hash<size_t> (the identity):
  // h = xor of (word << 1); Prefix -> h >> (min(nBits, 64) - min(prefix_bits, nBits))
  uint64_t h = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) h ^= key[w] << 1;
  if (!prefix) return h;
  const uint32_t pb = prefix_bits ? prefix_bits : 32u;
  const uint32_t hi = s.n_bits < 64u ? s.n_bits : 64u, lo = pb < s.n_bits ? pb : s.n_bits;
  const uint32_t shift = hi > lo ? hi - lo : 0u;
  return shift >= 64u ? 0ull : (h >> shift);
}

// Internal placement hash (NOT part of the reference's observable behaviour): decides
// the on-device bucket and LDS slot of a key. Cheap 32-bit multiplies only.
template <int NW> KMI_HD uint32_t place_hash(const uint64_t (&key)[NW]) {
  uint32_t h = 0x9E3779B9u;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t lo = (uint32_t)key[w], hi = (uint32_t)(key[w] >> 32);
    h = (h ^ lo) * 0x85EBCA6Bu;
    h ^= h >> 15;
    h = (h ^ hi) * 0xC2B2AE35u;
    h ^= h >> 13;
  }
  h *= 0x27D4EB2Fu;
  h ^= h >> 16;
  return h;
}

// ---------------------------------------------------------------------------
// chunk classification: C consecutive input bytes owned by one thread
// ---------------------------------------------------------------------------
// eol : bit i set <=> byte i is '\n' or '\r' (bytes at or past `n_valid` count as EOL)
// stream: complement codes, base i at bits [BITS*i, BITS*i+BITS)
// ---- four bytes at a time (SWAR + v_perm_b32 byte look-up) ----------------------------------
// perm(hi, lo, sel): result byte i = byte sel_i of the 8-byte table {lo (0..3), hi (4..7)}
KMI_HD uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  uint64_t tab = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) r |= (uint32_t)((tab >> (8 * ((sel >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
  return r;
#endif
}
// 0x80 in every byte of y that is zero (exact, no borrow artefacts)
KMI_HD uint32_t zero_bytes(uint32_t y) { return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu); }
// bit i of the result = bit 7 of byte i
KMI_HD uint32_t byte_msbs(uint32_t m) { return (((m >> 7) & 0x01010101u) * 0x01020408u) >> 24; }   // (shifts + ors measured slower: 1.39 vs 1.29 ms in the scan)

// classify the four bytes of w: eol4 = one bit per byte, packed = 4 complement codes (4*BITS bits)
template <int BITS> KMI_HD void classify_dword(uint32_t w, uint32_t &eol4, uint32_t &packed) {
  eol4 = byte_msbs(zero_bytes(w ^ 0x0A0A0A0Au) | zero_bytes(w ^ 0x0D0D0D0Du));
  const uint32_t x = w & 0xDFDFDFDFu;                 // fold case
  const uint32_t idx = (x >> 1) & 0x07070707u;        // A->0 C->1 T->2 G->3 X->4 N->7
  if (BITS == 4) {   // DNA16: byte by byte (sixteen IUPAC letters; not a hot path)
    uint32_t p = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) p |= comp_code<4>(code_dna16((w >> (8 * i)) & 0xffu)) << (4 * i);
    packed = p;
  } else if (BITS == 2) {
    // a byte is a base iff it equals the letter its index stands for
    const uint32_t expect = byte_perm(0xFFFFFFFFu, 0x47544341u, idx);            // 'A','C','T','G'
    const uint32_t ok = zero_bytes(x ^ expect);                                  // 0x80 per base byte
    const uint32_t lut = byte_perm(0x03030303u, 0x01000203u, idx);               // complement: A3 C2 T0 G1, rest 3
    const uint32_t v1 = ok >> 7;
    const uint32_t vm = v1 | (v1 << 1);                                          // 0x03 per base byte (no multiply: 32-bit ones are multi-cycle)
    const uint32_t cc = (lut & vm) | (0x03030303u & ~vm);                        // non-bases count as A -> complement 3
    const uint32_t p1 = cc | (cc >> 6);
    packed = (p1 | (p1 >> 12)) & 0xFFu;
  } else {
    const uint32_t expect = byte_perm(0x4EFFFF58u, 0x47544341u, idx);            // idx4 'X', idx7 'N'
    const uint32_t ok = zero_bytes(x ^ expect);
    const uint32_t lut = byte_perm(0x07020207u, 0x03010604u, idx);               // rev3 of A1 C3 T4 G6 -> 4 6 1 3, N/X 7
    const uint32_t vm = (ok >> 7) * 7u;
    const uint32_t gap = zero_bytes(w ^ 0x2D2D2D2Du) | zero_bytes(w ^ 0x2E2E2E2Eu);   // '-' '.' -> code 0
    const uint32_t dflt = 0x02020202u & ~((gap >> 7) * 7u);                      // everything else -> 2 (its own reversal)
    const uint32_t cc = (lut & vm) | (dflt & ~vm);
    const uint32_t p1 = (cc & 0x00070007u) | ((cc >> 5) & 0x00380038u);
    packed = (p1 | (p1 >> 10)) & 0xFFFu;
  }
}

// RNA_T / RNA6_T (alphabets.hpp:365-445, 448-530) are the DNA / DNA6 tables with 'U','u' in the place of 'T','t', and
// T an unknown character: swapping T and U in the loaded bytes (they differ in bit 0) lets the DNA classifiers serve.
KMI_HD uint32_t swap_tu_dword(uint32_t x) {
  const uint32_t y = (x & 0xDEDEDEDEu) ^ 0x54545454u;                             // zero byte <=> T, U, t or u
  const uint32_t nz = (((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;     // 0x80 in every non-zero byte
  return x ^ ((~nz & 0x80808080u) >> 7);
}

// `dw` holds the chunk's bytes little-endian, 4 per dword.
template <int BITS, int C> KMI_HD void classify_chunk(const uint32_t (&dw)[C / 4], int n_valid, uint32_t &eol, uint64_t &stream) {
  uint32_t e = 0; uint64_t s = 0;
  if (n_valid == C) {
#pragma unroll
    for (int i = 0; i < C / 4; ++i) {
      uint32_t e4, p4;
      classify_dword<BITS>(dw[i], e4, p4);
      e |= e4 << (4 * i);
      s |= (uint64_t)p4 << (4 * BITS * i);
    }
    eol = e; stream = s;
    return;
  }
#pragma unroll
  for (int i = 0; i < C; ++i) {
    uint32_t c = (i < n_valid) ? ((dw[i >> 2] >> (8 * (i & 3))) & 0xffu) : (uint32_t)'\n';
    e |= (is_eol(c) ? 1u : 0u) << i;
    s |= (uint64_t)comp_code<BITS>(code_of<BITS>(c)) << (BITS * i);
  }
  eol = e; stream = s;
}

// line starts: a non-EOL byte whose predecessor is EOL (or the partition start)
KMI_HD uint32_t line_starts(uint32_t eol, bool prev_is_eol, uint32_t cmask) {
  return (~eol) & ((eol << 1) | (prev_is_eol ? 1u : 0u)) & cmask;
}

// smear: bit p of the result set <=> any of bits [p, p+k) of x set.  x is NE 64-bit words.
template <int NE> KMI_HD void smear_right(uint64_t (&x)[NE], uint32_t k) {
  uint32_t covered = 1;
  while (covered < k) {
    uint32_t step = (k - covered < covered) ? (k - covered) : covered;
    uint32_t ws = step >> 6, bs = step & 63u;
    uint64_t y[NE];
#pragma unroll
    for (int w = 0; w < NE; ++w) {
      uint64_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        lo = ((uint32_t)j == w + ws) ? x[j] : lo;
        hi = ((uint32_t)j == w + ws + 1) ? x[j] : hi;
      }
      y[w] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
#pragma unroll
    for (int w = 0; w < NE; ++w) x[w] |= y[w];
    covered += step;
  }
}

// FASTQ role of positions in a chunk. `lines_before` = number of line starts strictly
// before this chunk (counted from the partition start), `ls` = line-start bits of the chunk.
// Returns the mask of chunk positions that lie on a sequence line (line index % 4 == 1).
KMI_HD uint32_t fastq_seq_role_mask(uint32_t lines_before, uint32_t ls, uint32_t cmask) {
  uint32_t mask = 0, start = 0;
  uint32_t cur = lines_before;           // line starts at or before the current position
  uint32_t rest = ls;
  while (rest) {
    uint32_t q = (uint32_t)__builtin_ctz(rest);
    if (((cur - 1u) & 3u) == 1u && cur != 0u) mask |= ((1u << q) - 1u) & ~((1u << start) - 1u);
    cur += 1; start = q; rest &= rest - 1u;
  }
  if (((cur - 1u) & 3u) == 1u && cur != 0u) mask |= cmask & ~((1u << start) - 1u);
  return mask & cmask;
}

// forward k-mer from its reverse complement window (the LDS stream holds complements)
template <int NW, int BITS> KMI_HD void fwd_from_rc(const uint64_t (&rc)[NW], uint64_t (&fwd)[NW], const KShape &s) {
  revcomp_words<NW, BITS>(rc, fwd, s);
}

}  // namespace kmi
