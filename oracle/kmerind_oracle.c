/*
 * kmerind_oracle.c -- TEST INFRASTRUCTURE ONLY (see kmerind_oracle.h).
 *
 * Plain-C, deliberately scalar restatement of the reference CPU algorithm.
 * It is written for obviousness, not speed: one base at a time, one k-mer at
 * a time, exactly in the order the reference's iterator stack visits them.
 */
#define _GNU_SOURCE
#include "kmerind_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* kspec                                                                    */
/* ------------------------------------------------------------------------ */

int orc_kspec_init(orc_kspec *s, uint32_t k, uint32_t alphabet) {
  if (!s || k == 0) return -1;
  s->k = k;
  s->alphabet = alphabet;
  /* AlphabetTraits::getBitsPerChar = ceilLog2(SIZE): alphabet_traits.hpp:127-130 */
  if (ORC_IS_2BIT(alphabet)) s->bits_per_char = 2;
  else if (alphabet == ORC_DNA5 || alphabet == ORC_RNA5) s->bits_per_char = 3;
  else if (alphabet == ORC_DNA16) s->bits_per_char = 4;
  else return -1;
  s->n_bits = k * s->bits_per_char;
  s->n_words = (s->n_bits + 63) / 64;   /* padding.hpp:81 */
  s->n_bytes = (s->n_bits + 7) / 8;     /* padding.hpp:79, kmer_hash.hpp:246 */
  if (s->n_words > ORC_MAX_WORDS) return -1;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* alphabets                                                                */
/* ------------------------------------------------------------------------ */

/* DNA_T::FROM_ASCII (alphabets.hpp:139-161): ACGT/acgt -> 0123, all else 0.
 * DNA6_T::FROM_ASCII (alphabets.hpp:225-248): A=1 C=3 G=6 T=4 N/X=7 '-','.'=0,
 * everything else 2 (DNA5 is an alias of DNA6, alphabets.hpp:746-747). */
uint8_t orc_from_ascii(uint32_t alphabet, uint8_t c) {
  /* RNA_T / RNA6_T::FROM_ASCII (alphabets.hpp:378-400, 459-480): U/u where DNA has T/t; T/t is an unknown character */
  if (alphabet == ORC_RNA || alphabet == ORC_RNA5) {
    if (c == 'U' || c == 'u') c = 'T';
    else if (c == 'T' || c == 't') c = '?';
    alphabet = (alphabet == ORC_RNA) ? ORC_DNA : ORC_DNA5;
  }
  if (alphabet == ORC_DNA16) {
    /* DNA16_T::FROM_ASCII (alphabets.hpp:660-681): presence bits A=1 C=2 G=4 T/U=8, IUPAC unions, '-' '.' = 0, else 0xF */
    switch (c) {
      case '-': case '.': return 0x0;
      case 'A': case 'a': return 0x1; case 'C': case 'c': return 0x2; case 'M': case 'm': return 0x3;
      case 'G': case 'g': return 0x4; case 'R': case 'r': return 0x5; case 'S': case 's': return 0x6;
      case 'V': case 'v': return 0x7; case 'T': case 't': case 'U': case 'u': return 0x8;
      case 'W': case 'w': return 0x9; case 'Y': case 'y': return 0xA; case 'H': case 'h': return 0xB;
      case 'K': case 'k': return 0xC; case 'D': case 'd': return 0xD; case 'B': case 'b': return 0xE;
      default: return 0xF;
    }
  }
  if (alphabet == ORC_DNA) {
    switch (c) {
      case 'A': case 'a': return 0;
      case 'C': case 'c': return 1;
      case 'G': case 'g': return 2;
      case 'T': case 't': return 3;
      default: return 0;
    }
  }
  switch (c) {
    case '-': case '.': return 0;
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 3;
    case 'G': case 'g': return 6;
    case 'T': case 't': return 4;
    case 'N': case 'n': case 'X': case 'x': return 7;
    default: return 2;
  }
}

/* TO_COMPLEMENT tables: alphabets.hpp:172-178 (DNA), :262-272 (DNA6) */
uint8_t orc_complement(uint32_t alphabet, uint8_t code) {
  if (ORC_IS_2BIT(alphabet)) return (uint8_t)(3 - (code & 3));   /* RNA: alphabets.hpp:411-421; RNA6: :497-512, same tables */
  if (alphabet == ORC_DNA16) {   /* TO_COMPLEMENT (alphabets.hpp:706-729): 4-bit reversal */
    static const uint8_t c16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    return c16[code & 15];
  }
  static const uint8_t c6[8] = {0, 4, 2, 6, 1, 5, 3, 7};
  return c6[code & 7];
}

/* ------------------------------------------------------------------------ */
/* Kmer value ops                                                           */
/* ------------------------------------------------------------------------ */

static inline uint64_t low_mask(uint32_t bits) {
  return bits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << bits) - 1);
}

/* do_sanitize: kmer.hpp:1454-1460 -- clear pad bits of the top word */
static inline void sanitize(const orc_kspec *s, uint64_t *kmer) {
  uint32_t inv_pad = 64 - (s->n_words * 64 - s->n_bits);
  kmer[s->n_words - 1] &= low_mask(inv_pad);
}

void orc_kmer_clear(const orc_kspec *s, uint64_t *kmer) {
  memset(kmer, 0, s->n_words * sizeof(uint64_t));
}

static void shl_bits(const orc_kspec *s, uint64_t *kmer, uint32_t sh) {
  for (int w = (int)s->n_words - 1; w > 0; --w)
    kmer[w] = (kmer[w] << sh) | (kmer[w - 1] >> (64 - sh));
  kmer[0] <<= sh;
}

static void shr_bits(const orc_kspec *s, uint64_t *kmer, uint32_t sh) {
  for (uint32_t w = 0; w + 1 < s->n_words; ++w)
    kmer[w] = (kmer[w] >> sh) | (kmer[w + 1] << (64 - sh));
  kmer[s->n_words - 1] >>= sh;
}

void orc_kmer_next_from_char(const orc_kspec *s, uint64_t *kmer, uint8_t code) {
  shl_bits(s, kmer, s->bits_per_char);
  kmer[0] |= (uint64_t)code & low_mask(s->bits_per_char);
  sanitize(s, kmer);
}

void orc_kmer_next_reverse_from_char(const orc_kspec *s, uint64_t *kmer, uint8_t code) {
  shr_bits(s, kmer, s->bits_per_char);
  uint32_t inv_pad = 64 - (s->n_words * 64 - s->n_bits);
  kmer[s->n_words - 1] |= ((uint64_t)code & low_mask(s->bits_per_char))
                          << (inv_pad - s->bits_per_char);
  sanitize(s, kmer);
}

/* character i counted from the LSB end (i = 0 is the newest base) */
static inline uint8_t get_char(const orc_kspec *s, const uint64_t *kmer, uint32_t i) {
  uint32_t pos = i * s->bits_per_char, w = pos >> 6, o = pos & 63;
  uint64_t v = kmer[w] >> o;
  if (o + s->bits_per_char > 64 && w + 1 < s->n_words) v |= kmer[w + 1] << (64 - o);
  return (uint8_t)(v & low_mask(s->bits_per_char));
}

static inline void set_char(const orc_kspec *s, uint64_t *kmer, uint32_t i, uint8_t c) {
  uint32_t pos = i * s->bits_per_char, w = pos >> 6, o = pos & 63;
  kmer[w] |= (uint64_t)c << o;
  if (o + s->bits_per_char > 64 && w + 1 < s->n_words) kmer[w + 1] |= (uint64_t)c >> (64 - o);
}

/* Same definition the reference's slow differential helper uses: reverse the
 * order of the k characters (kmer.hpp:1615-1679). */
void orc_kmer_reverse(const orc_kspec *s, const uint64_t *in, uint64_t *out) {
  uint64_t tmp[ORC_MAX_WORDS] = {0};
  for (uint32_t i = 0; i < s->k; ++i) set_char(s, tmp, s->k - 1 - i, get_char(s, in, i));
  memcpy(out, tmp, s->n_words * sizeof(uint64_t));
}

/* reverse the characters and complement each one (kmer.hpp:1118-1127; the DNA
 * fast path negates bits :1723-1742, the DNA6 fast path reverses single bits
 * :1807-1847 -- both equal the table-driven definition below). */
void orc_kmer_revcomp(const orc_kspec *s, const uint64_t *in, uint64_t *out) {
  uint64_t tmp[ORC_MAX_WORDS] = {0};
  for (uint32_t i = 0; i < s->k; ++i)
    set_char(s, tmp, s->k - 1 - i, orc_complement(s->alphabet, get_char(s, in, i)));
  memcpy(out, tmp, s->n_words * sizeof(uint64_t));
}

int orc_kmer_less(const orc_kspec *s, const uint64_t *a, const uint64_t *b) {
  for (int w = (int)s->n_words - 1; w >= 0; --w) {
    if (a[w] != b[w]) return a[w] < b[w];
  }
  return 0;
}

int orc_kmer_equal(const orc_kspec *s, const uint64_t *a, const uint64_t *b) {
  return memcmp(a, b, s->n_words * sizeof(uint64_t)) == 0;
}

void orc_kmer_canonical(const orc_kspec *s, const uint64_t *in, uint64_t *out) {
  uint64_t rc[ORC_MAX_WORDS];
  orc_kmer_revcomp(s, in, rc);
  /* lex_less: (x < rc) ? x : rc  (kmer_transform.hpp:108-116) */
  if (orc_kmer_less(s, in, rc)) memmove(out, in, s->n_words * sizeof(uint64_t));
  else memcpy(out, rc, s->n_words * sizeof(uint64_t));
}

void orc_kmer_xor_revcomp(const orc_kspec *s, const uint64_t *in, uint64_t *out) {
  uint64_t rc[ORC_MAX_WORDS];
  orc_kmer_revcomp(s, in, rc);
  for (uint32_t w = 0; w < s->n_words; ++w) out[w] = in[w] ^ rc[w];
}

void orc_kmer_from_ascii(const orc_kspec *s, const uint8_t *chars, uint64_t *out) {
  orc_kmer_clear(s, out);
  for (uint32_t i = 0; i < s->k; ++i)
    orc_kmer_next_from_char(s, out, orc_from_ascii(s->alphabet, chars[i]));
}

/* The form the reference actually executes for DNA (kmer.hpp:1723-1742 with the SWAR
 * bit-group reverse of src/utils/bitgroup_ops.hpp:489-515,770-800): byte swap, swap
 * nibbles, swap 2-bit groups, complement, shift out the pad. Used by the CPU baseline
 * driver so the timed path is not handicapped by the definitional loop above; tests pin
 * it to orc_kmer_revcomp. DNA5 falls back to the definitional form. */
static inline uint64_t swar_grouprev2(uint64_t x) {
  x = __builtin_bswap64(x);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  return x;
}

void orc_kmer_revcomp_fast(const orc_kspec *s, const uint64_t *in, uint64_t *out) {
  if (!ORC_IS_2BIT(s->alphabet)) { orc_kmer_revcomp(s, in, out); return; }
  uint64_t t[ORC_MAX_WORDS];
  const uint32_t nw = s->n_words, pad = nw * 64 - s->n_bits;
  for (uint32_t w = 0; w < nw; ++w) t[w] = ~swar_grouprev2(in[nw - 1 - w]);
  if (pad) {
    for (uint32_t w = 0; w < nw; ++w)
      t[w] = (t[w] >> pad) | ((w + 1 < nw) ? (t[w + 1] << (64 - pad)) : 0);
  }
  memcpy(out, t, nw * sizeof(uint64_t));
}

static inline void canonical_fast(const orc_kspec *s, const uint64_t *in, uint64_t *out) {
  uint64_t rc[ORC_MAX_WORDS];
  orc_kmer_revcomp_fast(s, in, rc);
  if (orc_kmer_less(s, in, rc)) memmove(out, in, s->n_words * sizeof(uint64_t));
  else memcpy(out, rc, s->n_words * sizeof(uint64_t));
}

void orc_kmers_revcomp_fast(const orc_kspec *s, const uint64_t *in, size_t n, uint64_t *out) {
  for (size_t i = 0; i < n; ++i) orc_kmer_revcomp_fast(s, in + i * s->n_words, out + i * s->n_words);
}

void orc_kmers_revcomp(const orc_kspec *s, const uint64_t *in, size_t n, uint64_t *out) {
  for (size_t i = 0; i < n; ++i) orc_kmer_revcomp(s, in + i * s->n_words, out + i * s->n_words);
}

void orc_kmers_canonical(const orc_kspec *s, const uint64_t *in, size_t n, uint64_t *out) {
  for (size_t i = 0; i < n; ++i) orc_kmer_canonical(s, in + i * s->n_words, out + i * s->n_words);
}

/* ------------------------------------------------------------------------ */
/* hashes                                                                   */
/* ------------------------------------------------------------------------ */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rotr64(uint64_t x, int r) { return r == 0 ? x : (x >> r) | (x << (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t load32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* Published MurmurHash3_x64_128 (Austin Appleby, public domain); the reference
 * vendors it unmodified at ext/smhasher/MurmurHash3.cpp:255-335. */
void orc_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2]) {
  const uint8_t *data = (const uint8_t *)key;
  const int nblocks = len / 16;
  uint64_t h1 = seed, h2 = seed;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  for (int i = 0; i < nblocks; ++i) {
    uint64_t k1 = load64(data + 16 * i), k2 = load64(data + 16 * i + 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t *tail = data + nblocks * 16;
  uint64_t k1 = 0, k2 = 0;
  int rem = len & 15;
  for (int i = rem - 1; i >= 8; --i) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
  if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; --i) k1 ^= (uint64_t)tail[i] << (8 * i);
  if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  out[0] = h1; out[1] = h2;
}

/* FarmHash (Geoff Pike, MIT) farmhashna::Hash64 short-string branches;
 * ext/farmhash/src/farmhash.cc:373-466. */
static const uint64_t FK0 = 0xc3a5c85c97cb3127ULL;
static const uint64_t FK1 = 0xb492b66fbe98f273ULL;
static const uint64_t FK2 = 0x9ae16a3b2f90404fULL;

static inline uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

static inline uint64_t hash_len16_mul(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul; a ^= a >> 47;
  uint64_t b = (v ^ a) * mul; b ^= b >> 47;
  return b * mul;
}

/* Hash128to64: ext/farmhash/src/farmhash.h:129-138 */
static inline uint64_t hash_len16(uint64_t u, uint64_t v) {
  return hash_len16_mul(u, v, 0x9ddfea08eb382d69ULL);
}

static uint64_t farm_len0to16(const uint8_t *s, size_t len) {
  if (len >= 8) {
    uint64_t mul = FK2 + len * 2;
    uint64_t a = load64(s) + FK2;
    uint64_t b = load64(s + len - 8);
    uint64_t c = rotr64(b, 37) * mul + a;
    uint64_t d = (rotr64(a, 25) + b) * mul;
    return hash_len16_mul(c, d, mul);
  }
  if (len >= 4) {
    uint64_t mul = FK2 + len * 2;
    uint64_t a = load32(s);
    return hash_len16_mul(len + (a << 3), load32(s + len - 4), mul);
  }
  if (len > 0) {
    uint8_t a = s[0], b = s[len >> 1], c = s[len - 1];
    uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
    uint32_t z = (uint32_t)len + ((uint32_t)c << 2);
    return shift_mix(y * FK2 ^ z * FK0) * FK2;
  }
  return FK2;
}

static uint64_t farm_len17to32(const uint8_t *s, size_t len) {
  uint64_t mul = FK2 + len * 2;
  uint64_t a = load64(s) * FK1;
  uint64_t b = load64(s + 8);
  uint64_t c = load64(s + len - 8) * mul;
  uint64_t d = load64(s + len - 16) * FK2;
  return hash_len16_mul(rotr64(a + b, 43) + rotr64(c, 30) + d, a + rotr64(b + FK2, 18) + c, mul);
}

static uint64_t farm_len33to64(const uint8_t *s, size_t len) {
  uint64_t mul = FK2 + len * 2;
  uint64_t a = load64(s) * FK2;
  uint64_t b = load64(s + 8);
  uint64_t c = load64(s + len - 8) * mul;
  uint64_t d = load64(s + len - 16) * FK2;
  uint64_t y = rotr64(a + b, 43) + rotr64(c, 30) + d;
  uint64_t z = hash_len16_mul(y, a + rotr64(b + FK2, 18) + c, mul);
  uint64_t e = load64(s + 16) * mul;
  uint64_t f = load64(s + 24);
  uint64_t g = (y + load64(s + len - 32)) * mul;
  uint64_t h = (z + load64(s + len - 24)) * mul;
  return hash_len16_mul(rotr64(e + f, 43) + rotr64(g, 30) + h, e + rotr64(f + a, 18) + g, mul);
}

/* util::Hash64WithSeed wraps the result in DebugTweak (farmhash.cc:334-343,1469-1471),
 * which is the identity only when NDEBUG is defined. The reference's default build
 * type is RelWithDebInfo = "-O3 -funroll-loops -g" WITHOUT -DNDEBUG
 * (CMakeLists.txt:26,200-204), so the shipped behaviour is the tweaked value
 * ~bswap64(h * k1); a Release (-DNDEBUG) build returns h itself. Default here =
 * the reference default; orc_set_farm_ndebug(1) selects the Release behaviour. */
static int g_farm_ndebug = 0;
void orc_set_farm_ndebug(int on) { g_farm_ndebug = on; }

static inline uint64_t farm_debug_tweak(uint64_t x) {
  if (g_farm_ndebug) return x;
  return ~__builtin_bswap64(x * FK1);
}

uint64_t orc_farm_hash64_with_seed(const void *key, size_t len, uint64_t seed) {
  const uint8_t *s = (const uint8_t *)key;
  uint64_t h;
  if (len <= 16) h = farm_len0to16(s, len);
  else if (len <= 32) h = farm_len17to32(s, len);
  else if (len <= 64) h = farm_len33to64(s, len);
  else return 0; /* k-mers never exceed 32 bytes here; long-string loop not restated */
  /* Hash64WithSeed(s,len,seed) = Hash64WithSeeds(s,len,k2,seed)
   *                            = HashLen16(Hash64(s,len) - k2, seed)  (farmhash.cc:519-529) */
  return farm_debug_tweak(hash_len16(h - FK2, seed));
}

/* ceilLog2 (src/common/bit_ops.hpp:139-152): KeyToRank constructs DistHash with it
 * (src/containers/distributed_unordered_map.hpp:153-156) */
static unsigned orc_log2(unsigned n, unsigned p) { return (n <= 1) ? p : orc_log2(n >> 1, p + 1); }
unsigned orc_ceil_log2(unsigned n) { return (n <= 1) ? 0 : orc_log2(n - 1, 0) + 1; }

/* Kmer::getSuffix / getPrefix for 64-bit words (src/common/kmer.hpp:1245-1249, 1203-1221). getPrefix's two
 * branches (bits inside the top word; or shift the whole k-mer right by nBits - NumBits and take the suffix)
 * both yield the top NumBits bits of the nBits-bit value, which is what is computed here bit by bit. */
static uint64_t orc_get_suffix(const orc_kspec *s, const uint64_t *kmer, unsigned nbits) {
  (void)s;
  return nbits >= 64 ? kmer[0] : (kmer[0] & ((1ull << nbits) - 1ull));
}
static uint64_t orc_get_prefix(const orc_kspec *s, const uint64_t *kmer, unsigned nbits) {
  uint64_t v = 0;
  if (nbits == 0) return 0;   /* the reference shifts by the word width here (undefined); never reached through KeyToRank (p = 1 skips) */
  for (unsigned i = 0; i < nbits; ++i) {
    unsigned b = s->n_bits - nbits + i;           /* bit of the k-mer value that lands on result bit i */
    v |= ((kmer[b >> 6] >> (b & 63u)) & 1ull) << i;
  }
  return v;
}

/* bliss::kmer::hash::{murmur,farm,identity,cpp_std}<KMER,Prefix> (src/index/kmer_hash.hpp:154-311).
 * prefix_bits = the constructor argument of identity / cpp_std, 0 = the class default (24 / 32).
 * PARITY UNPINNED for identity and cpp_std: no reference test or fixture holds values for them and kmer_hash.hpp
 * does not compile here without generated headers; they are restated from the cited lines only. */
uint64_t orc_kmer_hash_ex(const orc_kspec *s, uint32_t which, int prefix, unsigned prefix_bits, const uint64_t *kmer) {
  if (which == ORC_HASH_MURMUR) {
    uint64_t h[2];
    orc_murmur3_x64_128(kmer, (int)s->n_bytes, 42, h);   /* kmer_hash.hpp:256-275 */
    return prefix ? h[1] : h[0];
  }
  if (which == ORC_HASH_FARM) {
    /* farm: seed 42, Prefix uses (seed << 1) - 1 = 83   (kmer_hash.hpp:301-308) */
    return orc_farm_hash64_with_seed(kmer, s->n_bytes, prefix ? 83u : 42u);
  }
  if (which == ORC_HASH_IDENTITY) {                       /* kmer_hash.hpp:205-230 */
    if (prefix) {
      unsigned bits = prefix_bits ? prefix_bits : 24u;    /* default_init_value */
      if (bits > s->n_bits) bits = s->n_bits;             /* std::min(KMER::nBits, prefix_bits) */
      if (bits > 64) bits = 64;
      return orc_get_prefix(s, kmer, bits);
    }
    return orc_get_suffix(s, kmer, s->n_bits > 64 ? 64u : s->n_bits);   /* suffix_bits */
  }
  /* cpp_std, kmer_hash.hpp:154-198: size_t = KmerWordType = 8 bytes -> tuples = nWords, leftover = 0;
   * std::hash<size_t> of libstdc++ is the identity */
  uint64_t h = 0;
  for (uint32_t i = 0; i < s->n_words; ++i) h ^= (kmer[i] << 1);
  if (!prefix) return h;
  {
    unsigned pb = prefix_bits ? prefix_bits : 32u;        /* default_init_value */
    unsigned hi = s->n_bits < 64 ? s->n_bits : 64u, lo = pb < s->n_bits ? pb : s->n_bits;
    unsigned shift = hi > lo ? hi - lo : 0;
    return shift >= 64 ? 0 : (h >> shift);
  }
}

uint64_t orc_kmer_hash(const orc_kspec *s, uint32_t which, int prefix, const uint64_t *kmer) {
  return orc_kmer_hash_ex(s, which, prefix, 0, kmer);
}

void orc_kmers_hash(const orc_kspec *s, uint32_t which, int prefix, const uint64_t *kmers,
                    size_t n, uint64_t *out) {
  for (size_t i = 0; i < n; ++i) out[i] = orc_kmer_hash(s, which, prefix, kmers + i * s->n_words);
}

/* dist_trans: the DistTrans argument of SingleStrandHashMapParams (kmer_index.hpp:436-450): 0 = the model's own
 * (identity; lex_less for bimolecule), 1 = lex_less (kmer_transform.hpp:90-116), 2 = xor_rev_comp (:60-88, x ^ rc(x)) */
void orc_key_to_rank_ex(const orc_kspec *s, uint32_t dist_hash, uint32_t strand, uint32_t dist_trans,
                        const uint64_t *kmers, size_t n, uint32_t p, uint32_t *ranks) {
  uint64_t t[ORC_MAX_WORDS];
  for (size_t i = 0; i < n; ++i) {
    const uint64_t *k = kmers + i * s->n_words;
    if (strand == ORC_STRAND_BIMOLECULE || dist_trans == 1) { orc_kmer_canonical(s, k, t); k = t; }
    else if (dist_trans == 2) {
      orc_kmer_revcomp(s, k, t);
      for (uint32_t w = 0; w < s->n_words; ++w) t[w] ^= k[w];
      k = t;
    }
    ranks[i] = (uint32_t)(orc_kmer_hash_ex(s, dist_hash, 1, orc_ceil_log2(p), k) % p);
  }
}
void orc_key_to_rank(const orc_kspec *s, uint32_t dist_hash, uint32_t strand,
                     const uint64_t *kmers, size_t n, uint32_t p, uint32_t *ranks) {
  orc_key_to_rank_ex(s, dist_hash, strand, 0, kmers, n, p, ranks);
}

/* ------------------------------------------------------------------------ */
/* record parsing                                                           */
/* ------------------------------------------------------------------------ */

static inline int is_eol(uint8_t c) { return c == '\n' || c == '\r'; }

/* findNonEOL / findEOL: src/io/file_loader.hpp:146-173 */
static size_t find_non_eol(const uint8_t *b, size_t i, size_t n) { while (i < n && is_eol(b[i])) ++i; return i; }
static size_t find_eol(const uint8_t *b, size_t i, size_t n) { while (i < n && !is_eol(b[i])) ++i; return i; }

/* FASTQParser::get_next_record, src/io/fastq_loader.hpp:389-467, iterated the
 * way SequencesIterator does (src/io/sequence_iterator.hpp:96-300). */
long orc_fastq_records(const uint8_t *bytes, size_t n, uint64_t file_offset,
                       orc_record *out, size_t out_cap) {
  size_t i = 0;
  long count = 0;
  while (i < n) {
    if (bytes[i] != '@') return -1;                 /* :392-393 */
    i = find_non_eol(bytes, i, n);
    if (i < n && bytes[i] != '@') return -1;        /* :421-422 */
    size_t rec_start = i;
    i = find_eol(bytes, i, n);
    size_t sstart = find_non_eol(bytes, i, n);
    size_t send = find_eol(bytes, sstart, n);
    i = find_non_eol(bytes, send, n);
    if (i < n && bytes[i] != '+') return -1;        /* :437-438 */
    i = find_eol(bytes, i, n);
    size_t lstart = find_non_eol(bytes, i, n);
    size_t lend = find_eol(bytes, lstart, n);
    i = find_non_eol(bytes, lend, n);
    if (sstart == send || lstart == lend) {
      /* :454-460 warning only: truncated record is still returned */
    } else if (send - sstart != lend - lstart) {
      return -1;                                    /* :461-463 */
    }
    if (out && (size_t)count < out_cap) {
      orc_record *r = &out[count];
      r->record_offset = file_offset + rec_start;
      r->record_size = i - rec_start;
      r->seq_begin = file_offset + sstart;
      r->seq_end = file_offset + send;
      r->qual_begin = file_offset + lstart;
      r->qual_end = file_offset + lend;
      r->seq_index = (uint64_t)count;
    }
    ++count;
  }
  return count;
}

/* FASTAParser::init_parser (serial form, src/io/fasta_loader.hpp:485-604) restated literally:
 *  - a line start is the buffer start and every byte that follows a '\n' (a '\r' does not end a line);
 *    its flag is 1 when the byte there is '>' or ';';
 *  - a sentinel (end, 1) is appended and runs of equal flags are collapsed (std::unique);
 *  - if the first group is a non-header group it is skipped;
 *  - then (header start, first non-header line start, next header start) triples are emitted with
 *    sequence index k/2 (k = position of the middle element in the collapsed list), so a leading
 *    orphan group shifts the indices by one;
 * get_next_record (:618-723) serves [pos2, pos3) as the sequence of record (pos, index).
 * Whole-buffer (single partition) form. */
long orc_fasta_records(const uint8_t *bytes, size_t n, uint64_t file_offset,
                       orc_record *out, size_t out_cap) {
  if (n == 0) return 0;
  /* collapsed (position, flag) list */
  size_t cap = 16, m = 0;
  size_t *pos = (size_t *)malloc(cap * sizeof(size_t));
  uint8_t *flg = (uint8_t *)malloc(cap);
#define FA_PUSH(P, F) do { uint8_t f__ = (F); if (m == 0 || flg[m - 1] != f__) { \
    if (m == cap) { cap *= 2; pos = (size_t *)realloc(pos, cap * sizeof(size_t)); flg = (uint8_t *)realloc(flg, cap); } \
    pos[m] = (P); flg[m] = f__; ++m; } } while (0)
  FA_PUSH(0, (bytes[0] == ';' || bytes[0] == '>') ? 1 : 0);
  for (size_t i = 1; i < n; ++i)
    if (bytes[i - 1] == '\n') FA_PUSH(i, (bytes[i] == ';' || bytes[i] == '>') ? 1 : 0);
  FA_PUSH(n, 1);
#undef FA_PUSH
  long count = 0;
  size_t k = 0;
  if (flg[k] == 0) ++k;
  size_t p0 = (k < m) ? pos[k] : n;
  ++k;
  for (; k + 1 < m; k += 2) {
    size_t p2 = pos[k], p3 = pos[k + 1];
    if (out && (size_t)count < out_cap) {
      orc_record *r = &out[count];
      r->record_offset = file_offset + p0;
      r->record_size = p3 - p0;
      r->seq_begin = file_offset + p2;
      r->seq_end = file_offset + p3;
      r->qual_begin = r->qual_end = 0;
      r->seq_index = (uint64_t)(k / 2);
    }
    ++count;
    p0 = p3;
  }
  free(pos); free(flg);
  return count;
}

/* ------------------------------------------------------------------------ */
/* quality LUT                                                              */
/* ------------------------------------------------------------------------ */

/* Illumina18QualityScoreCodec<float> = QualityScoreCodec<float,33,126,0>
 * (quality_scores.hpp:529): DecodeLUT[q] = log2(1 - 10^(-q/10)), entry 0 =
 * lowest(), entries 94,95 = 0.0 (quality_scores.hpp:113-211). */
float orc_qual_lut(uint8_t phred_char) {
  int q = (int)phred_char - 33;
  if (q <= 0 || q > 95) return -3.402823466e+38F;
  if (q >= 94) return 0.0f;
  long double v = log2l(1.0L - exp2l((long double)q * log2l(10.0L) / (-10.0L)));
  return (float)(double)v;
}

/* ------------------------------------------------------------------------ */
/* tuple extraction                                                         */
/* ------------------------------------------------------------------------ */

/* one (sub)sequence [sb, se) of record rec through the k-mer parser; returns the running tuple count */
static size_t extract_range(const orc_kspec *s, uint32_t fmt, const uint8_t *bytes, uint64_t file_offset, const orc_record *rec,
                            size_t sb, size_t se, uint64_t *kmers, uint64_t *ids, float *quals, size_t out_cap, size_t total,
                            float *qwin) {
  const uint32_t K = s->k;
  const float q_lo = orc_qual_lut(33), q_hi = orc_qual_lut(33 + 95);
  /* KmerGenerationIterator over NotEOL-filtered, ASCII2-mapped chars
   * (kmer_parser.hpp:198-213, kmer_iterators.hpp:67-116): first K non-EOL chars
   * fill the window, every further non-EOL char slides it by one. */
  uint64_t km[ORC_MAX_WORDS];
  orc_kmer_clear(s, km);
  uint32_t filled = 0;
  /* ids of the window's chars: circular buffer of raw offsets */
  uint64_t *pos_ring = (uint64_t *)malloc(sizeof(uint64_t) * K);
  uint32_t ring = 0;
  /* quality window state (quality_score_iterator.hpp:99-173) */
  float qsum = 0.0f; uint32_t n_bad = 0, qpos = 0;
  const size_t rsb = (size_t)(rec->seq_begin - file_offset);
  size_t qb = (size_t)(rec->qual_begin - file_offset);
  for (size_t i = sb; i < se; ++i) {
    uint8_t c = bytes[i];
    if (is_eol(c)) continue;
    orc_kmer_next_from_char(s, km, orc_from_ascii(s->alphabet, c));
    pos_ring[ring] = file_offset + i; ring = (ring + 1) % K;
    if (quals && fmt == ORC_FMT_FASTQ) {
      float nv = orc_qual_lut(bytes[qb + (i - rsb)]);
      if (filled >= K) {
        float ov = qwin[qpos];
        if (ov > q_lo && ov < q_hi) qsum -= ov; else --n_bad;
      }
      qwin[qpos] = nv; qpos = (qpos + 1) % K;
      if (nv > q_lo && nv < q_hi) qsum += nv; else ++n_bad;
    }
    if (filled < K) ++filled;
    if (filled >= K) {
      if (total < out_cap) {
        if (kmers) memcpy(kmers + total * s->n_words, km, s->n_words * sizeof(uint64_t));
        if (ids) {
          uint64_t first_pos = pos_ring[ring]; /* oldest entry = first base of window */
          if (fmt == ORC_FMT_FASTQ) {
            /* ShortSequenceKmerId: sequence.hpp:156-157 + kmer_parser.hpp:378-386 */
            ids[total] = ((rec->record_offset & 0xFFFFFFFFFFULL) << 16) |
                         ((first_pos - rec->record_offset) & 0xFFFF);
          } else {
            /* LongSequenceKmerId: sequence.hpp:254-255 */
            ids[total] = (first_pos & 0xFFFFFFFFFFULL) | ((rec->seq_index & 0xFFFF) << 40);
          }
        }
        if (quals) quals[total] = (fmt == ORC_FMT_FASTQ) ? (n_bad > 0 ? 0.0f : exp2f(qsum)) : 0.0f;
      }
      ++total;
    }
  }
  free(pos_ring);
  return total;
}

/* seq_filter = the SeqIterType of read_file_* (src/io/filtered_sequence_iterator.hpp):
 * ORC_SEQ_ALL      SequencesIterator: every record;
 * ORC_SEQ_N_FILTER NFilterSequencesIterator (:154-165): records whose [seq_begin, seq_end) holds an 'N' are skipped;
 * ORC_SEQ_N_SPLIT  NSplitSequencesIterator (:166-440): each sequence is cut into the maximal runs of characters that
 *                  satisfy NCharFilter (x != 'N' && x != 'n', :411-418); a run keeps the record's id, offsets count from the
 *                  record start (split_seq, :228-249). read_block (kmer_file_helper.hpp:128-178) skips empty pieces
 *                  and counts every other one as a sequence.
 * *n_yield (may be NULL): what the reference's parse TESTS count (mpi_test_fastq_seq_parse.cpp:918-921): every
 * sequence the iterator yields, including the empty one it produces for a run of trailing N (split_seq on a
 * remainder of only-N characters). */
long orc_extract_filtered(const orc_kspec *s, uint32_t fmt, uint32_t seq_filter, const uint8_t *bytes, size_t n,
                          uint64_t file_offset, uint64_t *kmers, uint64_t *ids, float *quals,
                          size_t out_cap, size_t *n_seqs, size_t *n_yield) {
  long nrec = (fmt == ORC_FMT_FASTQ) ? orc_fastq_records(bytes, n, file_offset, NULL, 0)
                                     : orc_fasta_records(bytes, n, file_offset, NULL, 0);
  if (nrec < 0) return -1;
  orc_record *recs = (orc_record *)malloc(sizeof(orc_record) * (size_t)(nrec ? nrec : 1));
  if (fmt == ORC_FMT_FASTQ) orc_fastq_records(bytes, n, file_offset, recs, (size_t)nrec);
  else orc_fasta_records(bytes, n, file_offset, recs, (size_t)nrec);

  size_t total = 0, seqs = 0, yields = 0;
  float *qwin = (float *)malloc(sizeof(float) * s->k);
  for (long r = 0; r < nrec; ++r) {
    const orc_record *rec = &recs[r];
    size_t sb = (size_t)(rec->seq_begin - file_offset), se = (size_t)(rec->seq_end - file_offset);
    if (seq_filter == ORC_SEQ_N_SPLIT) {
      if (quals) { free(qwin); free(recs); return -1; }   /* no quality values for pieces */
      size_t nb = sb;                                      /* `next` = [nb, se) */
      while (nb < se) {                                    /* get_next skips records whose remainder is empty */
        size_t b = nb;
        while (b < se && (bytes[b] == 'N' || bytes[b] == 'n')) ++b;     /* find_if(pred) */
        size_t e = b;
        while (e < se && !(bytes[e] == 'N' || bytes[e] == 'n')) ++e;    /* find_if_not(pred) */
        ++yields;
        if (e > b) {                                       /* kmer_file_helper.hpp:139 */
          ++seqs;
          total = extract_range(s, fmt, bytes, file_offset, rec, b, e, kmers, ids, NULL, out_cap, total, qwin);
        }
        nb = e;
      }
      continue;
    }
    if (seq_filter == ORC_SEQ_N_FILTER && se > sb && memchr(bytes + sb, 'N', se - sb)) continue;
    ++yields;
    if (se == sb) continue;                               /* kmer_file_helper.hpp:139 */
    ++seqs;                                               /* :176-177 (whole buffer is valid) */
    total = extract_range(s, fmt, bytes, file_offset, rec, sb, se, kmers, ids, quals, out_cap, total, qwin);
  }
  free(qwin);
  free(recs);
  if (n_seqs) *n_seqs = seqs;
  if (n_yield) *n_yield = yields;
  return (long)total;
}

long orc_extract(const orc_kspec *s, uint32_t fmt, const uint8_t *bytes, size_t n,
                 uint64_t file_offset, uint64_t *kmers, uint64_t *ids, float *quals,
                 size_t out_cap, size_t *n_seqs) {
  return orc_extract_filtered(s, fmt, ORC_SEQ_ALL, bytes, n, file_offset, kmers, ids, quals, out_cap, n_seqs, NULL);
}

/* ------------------------------------------------------------------------ */
/* stable bucket permutation                                                */
/* ------------------------------------------------------------------------ */

void orc_stable_bucket(const uint32_t *ranks, size_t n, uint32_t p, uint64_t *bucket_sizes,
                       uint64_t *i2o) {
  /* assign_to_buckets: incremental_mxx.hpp:273-321 */
  for (uint32_t b = 0; b < p; ++b) bucket_sizes[b] = 0;
  for (size_t i = 0; i < n; ++i) ++bucket_sizes[ranks[i]];
  /* bucket_to_permutation: :324-364 -- exclusive scan then stable positions */
  uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * p);
  uint64_t acc = 0;
  for (uint32_t b = 0; b < p; ++b) { off[b] = acc; acc += bucket_sizes[b]; }
  for (size_t i = 0; i < n; ++i) i2o[i] = off[ranks[i]]++;
  free(off);
}

/* ------------------------------------------------------------------------ */
/* counting map                                                             */
/* ------------------------------------------------------------------------ */

typedef struct cm_node {
  struct cm_node *next;
  uint64_t hash;
  uint32_t count;
  uint64_t key[]; /* n_words */
} cm_node;

struct orc_count_map {
  orc_kspec spec;
  uint32_t strand, store_hash;
  cm_node **buckets;
  size_t n_buckets, size;
};

orc_count_map *orc_count_map_create(const orc_kspec *s, uint32_t strand, uint32_t store_hash) {
  orc_count_map *m = (orc_count_map *)calloc(1, sizeof(*m));
  m->spec = *s; m->strand = strand; m->store_hash = store_hash;
  m->n_buckets = 1024;
  m->buckets = (cm_node **)calloc(m->n_buckets, sizeof(cm_node *));
  return m;
}

void orc_count_map_destroy(orc_count_map *m) {
  if (!m) return;
  for (size_t b = 0; b < m->n_buckets; ++b) {
    cm_node *nd = m->buckets[b];
    while (nd) { cm_node *nx = nd->next; free(nd); nd = nx; }
  }
  free(m->buckets); free(m);
}

static void cm_rehash(orc_count_map *m) {
  size_t nb = m->n_buckets * 2;
  cm_node **nbk = (cm_node **)calloc(nb, sizeof(cm_node *));
  for (size_t b = 0; b < m->n_buckets; ++b) {
    cm_node *nd = m->buckets[b];
    while (nd) { cm_node *nx = nd->next; size_t j = nd->hash & (nb - 1); nd->next = nbk[j]; nbk[j] = nd; nd = nx; }
  }
  free(m->buckets); m->buckets = nbk; m->n_buckets = nb;
}

/* StoreTrans (bimolecule -> lex_less) then StoreHash<Prefix=false>:
 * kmer_index.hpp:436-481, fsc_container_utils.hpp:66-86 */
static uint64_t cm_hash(const orc_count_map *m, const uint64_t *key) {
  return orc_kmer_hash(&m->spec, m->store_hash, 0, key);
}

static cm_node *cm_find(const orc_count_map *m, const uint64_t *key, uint64_t h) {
  cm_node *nd = m->buckets[h & (m->n_buckets - 1)];
  while (nd) {
    if (nd->hash == h && orc_kmer_equal(&m->spec, nd->key, key)) return nd;
    nd = nd->next;
  }
  return NULL;
}

/* input transform of the map (distributed_map_base.hpp:286-289): canonical strand
 * applies lex_less to the key before anything else. For bimolecule the store
 * transform is lex_less, so equality is on the canonical form; the oracle keeps
 * the canonical form as the representative (the reference keeps whichever strand
 * was inserted first, which is unspecified -- SURVEY.md section 7 item 6). */
static void cm_transform(const orc_count_map *m, const uint64_t *in, uint64_t *out) {
  if (m->strand == ORC_STRAND_SINGLE) memcpy(out, in, m->spec.n_words * sizeof(uint64_t));
  else orc_kmer_canonical(&m->spec, in, out);
}

static void cm_add(orc_count_map *m, const uint64_t *key, uint32_t v) {
  uint64_t h = cm_hash(m, key);
  cm_node *nd = cm_find(m, key, h);
  if (nd) { nd->count += v; return; }   /* r(old,new) = std::plus<uint32_t>, wraps */
  if (m->size + 1 > m->n_buckets) cm_rehash(m);
  nd = (cm_node *)malloc(sizeof(cm_node) + m->spec.n_words * sizeof(uint64_t));
  nd->hash = h; nd->count = v;
  memcpy(nd->key, key, m->spec.n_words * sizeof(uint64_t));
  size_t b = h & (m->n_buckets - 1);
  nd->next = m->buckets[b]; m->buckets[b] = nd;
  ++m->size;
}

void orc_count_map_insert(orc_count_map *m, const uint64_t *kmers, size_t n) {
  uint64_t t[ORC_MAX_WORDS];
  for (size_t i = 0; i < n; ++i) {
    cm_transform(m, kmers + i * m->spec.n_words, t);
    cm_add(m, t, 1);   /* counting_unordered_map::insert wraps key as (k,1): :1826-1884 */
  }
}

size_t orc_count_map_size(const orc_count_map *m) { return m->size; }

size_t orc_count_map_export(const orc_count_map *m, uint64_t *keys, uint32_t *counts) {
  size_t j = 0;
  for (size_t b = 0; b < m->n_buckets; ++b)
    for (cm_node *nd = m->buckets[b]; nd; nd = nd->next) {
      if (keys) memcpy(keys + j * m->spec.n_words, nd->key, m->spec.n_words * sizeof(uint64_t));
      if (counts) counts[j] = nd->count;
      ++j;
    }
  return j;
}

/* transform + unique of the query keys (distributed_unordered_map.hpp:901-911,
 * fsc_container_utils.hpp:306-320) then per-key lookup. */
static orc_count_map *cm_unique_queries(const orc_count_map *m, const uint64_t *queries, size_t nq) {
  orc_count_map *u = orc_count_map_create(&m->spec, m->strand, m->store_hash);
  orc_count_map_insert(u, queries, nq);
  return u;
}

size_t orc_count_map_count(const orc_count_map *m, const uint64_t *queries, size_t nq,
                           uint64_t *out_keys, uint64_t *out_counts) {
  orc_count_map *u = cm_unique_queries(m, queries, nq);
  size_t j = 0;
  for (size_t b = 0; b < u->n_buckets; ++b)
    for (cm_node *q = u->buckets[b]; q; q = q->next) {
      cm_node *nd = cm_find(m, q->key, q->hash);
      if (out_keys) memcpy(out_keys + j * m->spec.n_words, q->key, m->spec.n_words * sizeof(uint64_t));
      if (out_counts) out_counts[j] = nd ? 1 : 0;  /* LocalCount = db.count(k): :231-238 (0 or 1 for a map) */
      ++j;
    }
  orc_count_map_destroy(u);
  return j;
}

size_t orc_count_map_find(const orc_count_map *m, const uint64_t *queries, size_t nq,
                          uint64_t *out_keys, uint32_t *out_counts) {
  orc_count_map *u = cm_unique_queries(m, queries, nq);
  size_t j = 0;
  for (size_t b = 0; b < u->n_buckets; ++b)
    for (cm_node *q = u->buckets[b]; q; q = q->next) {
      cm_node *nd = cm_find(m, q->key, q->hash);
      if (!nd) continue;
      if (out_keys) memcpy(out_keys + j * m->spec.n_words, nd->key, m->spec.n_words * sizeof(uint64_t));
      if (out_counts) out_counts[j] = nd->count;
      ++j;
    }
  orc_count_map_destroy(u);
  return j;
}

size_t orc_count_map_erase(orc_count_map *m, const uint64_t *queries, size_t nq) {
  uint64_t t[ORC_MAX_WORDS];
  size_t erased = 0;
  for (size_t i = 0; i < nq; ++i) {
    cm_transform(m, queries + i * m->spec.n_words, t);
    uint64_t h = cm_hash(m, t);
    cm_node **pp = &m->buckets[h & (m->n_buckets - 1)];
    while (*pp) {
      if ((*pp)->hash == h && orc_kmer_equal(&m->spec, (*pp)->key, t)) {
        cm_node *d = *pp; *pp = d->next; free(d); --m->size; ++erased; break;
      }
      pp = &(*pp)->next;
    }
  }
  return erased;
}


/* ------------------------------------------------------------------------ */
/* multimap (unordered_multimap semantics for PositionIndex)                 */
/* ------------------------------------------------------------------------ */
/* ::dsc::unordered_multimap (distributed_unordered_map.hpp:1466-1515): insert keeps every
 * (key, value); LocalCount = db.count(k) = multiplicity (:231-238); LocalFind emits the whole
 * equal_range (:1100-1131); erase removes all entries of a key (:1292-1328). Values are
 * `vw` 64-bit words (1 = Short/LongSequenceKmerId). */
typedef struct mm_node { struct mm_node *next; uint64_t hash; uint64_t data[]; /* key words, value words */ } mm_node;
struct orc_multi_map {
  orc_kspec spec; uint32_t strand, store_hash, vw;
  mm_node **buckets; size_t n_buckets, size;
};

orc_multi_map *orc_multi_map_create(const orc_kspec *s, uint32_t strand, uint32_t store_hash, uint32_t value_words) {
  orc_multi_map *m = (orc_multi_map *)calloc(1, sizeof(*m));
  m->spec = *s; m->strand = strand; m->store_hash = store_hash; m->vw = value_words;
  m->n_buckets = 1024; m->buckets = (mm_node **)calloc(m->n_buckets, sizeof(mm_node *));
  return m;
}
void orc_multi_map_destroy(orc_multi_map *m) {
  if (!m) return;
  for (size_t b = 0; b < m->n_buckets; ++b) { mm_node *nd = m->buckets[b]; while (nd) { mm_node *nx = nd->next; free(nd); nd = nx; } }
  free(m->buckets); free(m);
}
static void mm_rehash(orc_multi_map *m) {
  size_t nb = m->n_buckets * 2;
  mm_node **nbk = (mm_node **)calloc(nb, sizeof(mm_node *));
  for (size_t b = 0; b < m->n_buckets; ++b) {
    mm_node *nd = m->buckets[b];
    while (nd) { mm_node *nx = nd->next; size_t j = nd->hash & (nb - 1); nd->next = nbk[j]; nbk[j] = nd; nd = nx; }
  }
  free(m->buckets); m->buckets = nbk; m->n_buckets = nb;
}
static void mm_transform(const orc_multi_map *m, const uint64_t *in, uint64_t *out) {
  if (m->strand == ORC_STRAND_SINGLE) memcpy(out, in, m->spec.n_words * sizeof(uint64_t));
  else orc_kmer_canonical(&m->spec, in, out);
}
void orc_multi_map_insert(orc_multi_map *m, const uint64_t *kmers, const uint64_t *values, size_t n) {
  const uint32_t nw = m->spec.n_words;
  uint64_t t[ORC_MAX_WORDS];
  for (size_t i = 0; i < n; ++i) {
    mm_transform(m, kmers + i * nw, t);
    if (m->size + 1 > m->n_buckets) mm_rehash(m);
    mm_node *nd = (mm_node *)malloc(sizeof(mm_node) + (nw + m->vw) * sizeof(uint64_t));
    nd->hash = orc_kmer_hash(&m->spec, m->store_hash, 0, t);
    memcpy(nd->data, t, nw * sizeof(uint64_t));
    memcpy(nd->data + nw, values + i * m->vw, m->vw * sizeof(uint64_t));
    size_t b = nd->hash & (m->n_buckets - 1);
    nd->next = m->buckets[b]; m->buckets[b] = nd; ++m->size;
  }
}
size_t orc_multi_map_size(const orc_multi_map *m) { return m->size; }
size_t orc_multi_map_export(const orc_multi_map *m, uint64_t *keys, uint64_t *values) {
  const uint32_t nw = m->spec.n_words; size_t j = 0;
  for (size_t b = 0; b < m->n_buckets; ++b)
    for (mm_node *nd = m->buckets[b]; nd; nd = nd->next) {
      if (keys) memcpy(keys + j * nw, nd->data, nw * sizeof(uint64_t));
      if (values) memcpy(values + j * m->vw, nd->data + nw, m->vw * sizeof(uint64_t));
      ++j;
    }
  return j;
}
/* distinct transformed query keys via a counting map, then per key the equal_range */
size_t orc_multi_map_count(const orc_multi_map *m, const uint64_t *queries, size_t nq, uint64_t *out_keys, uint64_t *out_counts) {
  orc_count_map *u = orc_count_map_create(&m->spec, m->strand, m->store_hash);
  orc_count_map_insert(u, queries, nq);
  const uint32_t nw = m->spec.n_words; size_t j = 0;
  for (size_t b = 0; b < u->n_buckets; ++b)
    for (cm_node *q = u->buckets[b]; q; q = q->next) {
      uint64_t c = 0;
      for (mm_node *nd = m->buckets[q->hash & (m->n_buckets - 1)]; nd; nd = nd->next)
        if (nd->hash == q->hash && orc_kmer_equal(&m->spec, nd->data, q->key)) ++c;
      if (out_keys) memcpy(out_keys + j * nw, q->key, nw * sizeof(uint64_t));
      if (out_counts) out_counts[j] = c;
      ++j;
    }
  orc_count_map_destroy(u);
  return j;
}
size_t orc_multi_map_find(const orc_multi_map *m, const uint64_t *queries, size_t nq, uint64_t *out_keys, uint64_t *out_values, size_t cap) {
  orc_count_map *u = orc_count_map_create(&m->spec, m->strand, m->store_hash);
  orc_count_map_insert(u, queries, nq);
  const uint32_t nw = m->spec.n_words; size_t j = 0;
  for (size_t b = 0; b < u->n_buckets; ++b)
    for (cm_node *q = u->buckets[b]; q; q = q->next)
      for (mm_node *nd = m->buckets[q->hash & (m->n_buckets - 1)]; nd; nd = nd->next)
        if (nd->hash == q->hash && orc_kmer_equal(&m->spec, nd->data, q->key)) {
          if (j < cap) {
            if (out_keys) memcpy(out_keys + j * nw, nd->data, nw * sizeof(uint64_t));
            if (out_values) memcpy(out_values + j * m->vw, nd->data + nw, m->vw * sizeof(uint64_t));
          }
          ++j;
        }
  orc_count_map_destroy(u);
  return j;
}
size_t orc_multi_map_erase(orc_multi_map *m, const uint64_t *queries, size_t nq) {
  uint64_t t[ORC_MAX_WORDS]; size_t erased = 0;
  for (size_t i = 0; i < nq; ++i) {
    mm_transform(m, queries + i * m->spec.n_words, t);
    uint64_t h = orc_kmer_hash(&m->spec, m->store_hash, 0, t);
    mm_node **pp = &m->buckets[h & (m->n_buckets - 1)];
    while (*pp) {
      if ((*pp)->hash == h && orc_kmer_equal(&m->spec, (*pp)->data, t)) { mm_node *d = *pp; *pp = d->next; free(d); --m->size; ++erased; }
      else pp = &(*pp)->next;
    }
  }
  return erased;
}

/* ------------------------------------------------------------------------ */
/* CPU baseline driver                                                      */
/* ------------------------------------------------------------------------ */

typedef struct {
  /* shared */
  const uint8_t *bytes; size_t n; orc_kspec spec; uint32_t strand; uint32_t T;
  size_t *part_begin; /* T+1 record-aligned byte offsets */
  pthread_barrier_t *bar;
  /* exchange area */
  uint64_t **send_buf;      /* [T] permuted tuples per source rank */
  uint64_t **send_counts;   /* [T][T] */
  /* per thread */
  uint32_t tid;
  uint64_t n_kmers, n_distinct;
} bench_arg;

static void *bench_worker(void *vp) {
  bench_arg *a = (bench_arg *)vp;
  const orc_kspec *s = &a->spec;
  uint32_t T = a->T, me = a->tid;
  size_t b = a->part_begin[me], e = a->part_begin[me + 1];
  /* --- "read": KmerFileHelper::read_file -> vector<tuple> (kmer_file_helper.hpp:550-579) */
  size_t nseq = 0;
  /* result.reserve(estimate) then one parsing pass (kmer_file_helper.hpp:464-467): here the
   * reservation is the trivial upper bound "one tuple per input byte" */
  size_t cap = (e > b) ? (e - b) : 1;
  uint64_t *km = (uint64_t *)malloc(sizeof(uint64_t) * s->n_words * cap);
  long nk = (e > b) ? orc_extract(s, ORC_FMT_FASTQ, a->bytes + b, e - b, b, km, NULL, NULL, cap, &nseq) : 0;
  if (nk < 0) nk = 0;
  a->n_kmers = (uint64_t)nk;
  /* --- "insert": transform_input (distributed_unordered_map.hpp:1709) */
  if (a->strand == ORC_STRAND_CANONICAL)
    for (long i = 0; i < nk; ++i) canonical_fast(s, km + (size_t)i * s->n_words, km + (size_t)i * s->n_words);
  uint64_t *recv = km; size_t nrecv = (size_t)nk;
  if (T > 1) {
    /* imxx::distribute (incremental_mxx.hpp:1039-1109): bucket, permute, exchange */
    uint32_t *ranks = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nk ? nk : 1));
    uint64_t *i2o = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nk ? nk : 1));
    orc_key_to_rank(s, ORC_HASH_MURMUR, a->strand, km, (size_t)nk, T, ranks);
    orc_stable_bucket(ranks, (size_t)nk, T, a->send_counts[me], i2o);
    uint64_t *perm = (uint64_t *)malloc(sizeof(uint64_t) * s->n_words * (size_t)(nk ? nk : 1));
    for (long i = 0; i < nk; ++i)
      memcpy(perm + i2o[i] * s->n_words, km + (size_t)i * s->n_words, s->n_words * sizeof(uint64_t));
    free(km); free(ranks); free(i2o);
    a->send_buf[me] = perm;
    pthread_barrier_wait(a->bar);                 /* all2all(counts) + all2allv(payload) */
    nrecv = 0;
    for (uint32_t src = 0; src < T; ++src) nrecv += a->send_counts[src][me];
    recv = (uint64_t *)malloc(sizeof(uint64_t) * s->n_words * (nrecv ? nrecv : 1));
    size_t w = 0;
    for (uint32_t src = 0; src < T; ++src) {      /* concatenation by source rank ascending */
      uint64_t off = 0;
      for (uint32_t d = 0; d < me; ++d) off += a->send_counts[src][d];
      size_t cnt = a->send_counts[src][me];
      memcpy(recv + w * s->n_words, a->send_buf[src] + off * s->n_words, cnt * s->n_words * sizeof(uint64_t));
      w += cnt;
    }
    pthread_barrier_wait(a->bar);
    free(a->send_buf[me]);
  }
  /* local_insert (distributed_unordered_map.hpp:1603-1618); keys already input-transformed */
  orc_count_map *m = orc_count_map_create(s, ORC_STRAND_SINGLE, ORC_HASH_MURMUR);
  if (a->strand == ORC_STRAND_BIMOLECULE) m->strand = ORC_STRAND_BIMOLECULE;
  orc_count_map_insert(m, recv, nrecv);
  a->n_distinct = orc_count_map_size(m);
  orc_count_map_destroy(m);
  free(recv);
  return NULL;
}

/* record-aligned split of a FASTQ buffer into T byte ranges: the start of each
 * range is moved forward to the next true record start using the 4-line rule
 * of FASTQParser::find_first_record (fastq_loader.hpp:269-364). */
static size_t fastq_align(const uint8_t *b, size_t n, size_t pos) {
  if (pos == 0) return 0;
  size_t i = pos;
  if (!is_eol(b[i])) i = find_eol(b, i, n);
  size_t off[4]; uint8_t first[4] = {0, 0, 0, 0};
  i = find_non_eol(b, i, n);
  if (i >= n) return n;
  off[0] = i; first[0] = b[i];
  for (int j = 1; j < 4; ++j) {
    i = find_eol(b, i, n); i = find_non_eol(b, i, n);
    off[j] = i; if (i < n) first[j] = b[i];
  }
  if (i >= n) return n;
  if (first[0] == '@' && first[2] == '+') return off[0];
  if (first[1] == '@' && first[3] == '+') return off[1];
  if (first[0] == '+' && first[2] == '@') return off[2];
  if (first[1] == '+' && first[3] == '@') return off[3];
  return n;
}

/* ------------------------------------------------------------------------ */
/* de Bruijn graph nodes (test/test/debruijn/)                              */
/* ------------------------------------------------------------------------ */

/* edge_iterator<IT, DNA16> (edge_iterator.hpp:84-177) zipped with the k-mer iterator (de_bruijn_construct_engine.hpp:
 * 140-155). chars = the NotEOL-filtered characters of one sequence. _curr starts k-1 characters in, _right one further,
 * _left at "none"; every step moves all three (operator++, :119-138); operator* (:163-177) packs
 * FROM_ASCII[*_left] << 4 | FROM_ASCII[*_right], leaving out the side that has run off the sequence. */
static size_t dbg_parse_seq(const orc_kspec *s, const uint8_t *chars, size_t len, uint64_t *kmers, uint8_t *edges,
                            size_t out_cap, size_t total) {
  const uint32_t K = s->k;
  if (len < K) return total;
  uint64_t km[ORC_MAX_WORDS];
  orc_kmer_clear(s, km);
  for (uint32_t i = 0; i + 1 < K; ++i) orc_kmer_next_from_char(s, km, orc_from_ascii(s->alphabet, chars[i]));
  size_t left = len /* = _data_end: none yet */, right = K;
  for (size_t curr = K - 1; curr < len; ++curr) {
    orc_kmer_next_from_char(s, km, orc_from_ascii(s->alphabet, chars[curr]));
    uint8_t e = 0;
    if (left != len && right != len) e = (uint8_t)((orc_from_ascii(ORC_DNA16, chars[left]) << 4) | orc_from_ascii(ORC_DNA16, chars[right]));
    else if (left == len && right != len) e = orc_from_ascii(ORC_DNA16, chars[right]);
    else if (left != len && right == len) e = (uint8_t)(orc_from_ascii(ORC_DNA16, chars[left]) << 4);
    if (total < out_cap) {
      if (kmers) memcpy(kmers + total * s->n_words, km, s->n_words * sizeof(uint64_t));
      if (edges) edges[total] = e;
    }
    ++total;
    /* operator++ */
    if (left == len) left = 0; else ++left;
    if (right != len) ++right;
  }
  return total;
}

/* fmt: ORC_FMT_FASTQ / ORC_FMT_FASTA -- the parser is generic over the sequence type (de_bruijn_construct_engine.hpp:108-158): the characters of a
 * record's sequence without its EOLs (NonEOLIter), k-mers and edges over them */
long orc_dbg_parse_fmt(const orc_kspec *s, uint32_t fmt, const uint8_t *bytes, size_t n, uint64_t *kmers, uint8_t *edges, size_t out_cap) {
  long nrec = fmt == ORC_FMT_FASTA ? orc_fasta_records(bytes, n, 0, NULL, 0) : orc_fastq_records(bytes, n, 0, NULL, 0);
  if (nrec < 0) return -1;
  orc_record *recs = (orc_record *)malloc(sizeof(orc_record) * (size_t)(nrec ? nrec : 1));
  if (fmt == ORC_FMT_FASTA) orc_fasta_records(bytes, n, 0, recs, (size_t)nrec); else orc_fastq_records(bytes, n, 0, recs, (size_t)nrec);
  size_t total = 0;
  uint8_t *chars = (uint8_t *)malloc(n ? n : 1);
  for (long r = 0; r < nrec; ++r) {
    size_t len = 0;
    for (size_t i = (size_t)recs[r].seq_begin; i < (size_t)recs[r].seq_end; ++i)
      if (!is_eol(bytes[i])) chars[len++] = bytes[i];     /* NonEOLIter */
    if (len == 0) continue;                                /* :138 */
    total = dbg_parse_seq(s, chars, len, kmers, edges, out_cap, total);
  }
  free(chars); free(recs);
  return (long)total;
}

long orc_dbg_parse(const orc_kspec *s, const uint8_t *bytes, size_t n, uint64_t *kmers, uint8_t *edges, size_t out_cap) {
  return orc_dbg_parse_fmt(s, ORC_FMT_FASTQ, bytes, n, kmers, edges, out_cap);
}

uint8_t orc_dbg_edges_revcomp(uint8_t exts) {
  return (uint8_t)((orc_complement(ORC_DNA16, exts & 0xF) << 4) | orc_complement(ORC_DNA16, exts >> 4));
}

typedef struct dbg_node {
  struct dbg_node *next;
  uint64_t hash;
  uint32_t counts[9];
  uint64_t stamp;          /* last find() call that reported the node */
  uint64_t key[];          /* the strand the node was created with */
} dbg_node;

struct orc_dbg_map {
  orc_kspec spec;
  uint32_t store_hash;
  int exists_only;
  dbg_node **buckets;
  size_t n_buckets, size;
  uint64_t stamp;
};

orc_dbg_map *orc_dbg_map_create(const orc_kspec *s, uint32_t store_hash, int exists_only) {
  orc_dbg_map *m = (orc_dbg_map *)calloc(1, sizeof(orc_dbg_map));
  m->spec = *s; m->store_hash = store_hash; m->exists_only = exists_only;
  m->n_buckets = 1024;
  m->buckets = (dbg_node **)calloc(m->n_buckets, sizeof(dbg_node *));
  return m;
}

void orc_dbg_map_destroy(orc_dbg_map *m) {
  if (!m) return;
  for (size_t b = 0; b < m->n_buckets; ++b) {
    dbg_node *nd = m->buckets[b];
    while (nd) { dbg_node *nx = nd->next; free(nd); nd = nx; }
  }
  free(m->buckets); free(m);
}

/* BimoleculeHashMapParams (kmer_index.hpp:468-481): hash and equality on lex_less(key), the key itself is stored as given */
static dbg_node *dbg_find(const orc_dbg_map *m, const uint64_t *key, uint64_t *hash_out, int *same_strand) {
  uint64_t canon[ORC_MAX_WORDS], other[ORC_MAX_WORDS];
  orc_kmer_canonical(&m->spec, key, canon);
  const uint64_t h = orc_kmer_hash(&m->spec, m->store_hash, 0, canon);
  if (hash_out) *hash_out = h;
  for (dbg_node *nd = m->buckets[h & (m->n_buckets - 1)]; nd; nd = nd->next) {
    if (nd->hash != h) continue;
    orc_kmer_canonical(&m->spec, nd->key, other);
    if (orc_kmer_equal(&m->spec, other, canon)) {
      if (same_strand) *same_strand = orc_kmer_equal(&m->spec, nd->key, key);
      return nd;
    }
  }
  return NULL;
}

static void dbg_update(const orc_dbg_map *m, dbg_node *nd, uint8_t exts) {
  if (m->exists_only) {            /* edge_exists::update (:302-311): counts |= exts */
    for (int i = 0; i < 8; ++i) nd->counts[i] |= (exts >> i) & 1u;
    return;
  }
  nd->counts[8] += 1;              /* edge_counts<DNA16, int32_t>::update (:200-239): no clamping at 32 bits */
  for (int i = 0; i < 8; ++i) nd->counts[i] += (exts >> i) & 1u;
}

void orc_dbg_map_insert(orc_dbg_map *m, const uint64_t *kmers, const uint8_t *edges, size_t n) {
  const uint32_t nw = m->spec.n_words;
  for (size_t i = 0; i < n; ++i) {
    const uint64_t *key = kmers + i * nw;
    uint64_t h; int same = 1;
    dbg_node *nd = dbg_find(m, key, &h, &same);
    if (!nd) {                       /* de_bruijn_nodes_distributed.hpp:116-133: new node, SENSE */
      if (m->size + 1 > m->n_buckets) {
        size_t nb = m->n_buckets * 2;
        dbg_node **nbk = (dbg_node **)calloc(nb, sizeof(dbg_node *));
        for (size_t b = 0; b < m->n_buckets; ++b) {
          dbg_node *x = m->buckets[b];
          while (x) { dbg_node *nx = x->next; size_t j = x->hash & (nb - 1); x->next = nbk[j]; nbk[j] = x; x = nx; }
        }
        free(m->buckets); m->buckets = nbk; m->n_buckets = nb;
      }
      nd = (dbg_node *)calloc(1, sizeof(dbg_node) + nw * sizeof(uint64_t));
      nd->hash = h; memcpy(nd->key, key, nw * sizeof(uint64_t));
      size_t b = h & (m->n_buckets - 1);
      nd->next = m->buckets[b]; m->buckets[b] = nd; m->size++;
      same = 1;
    }
    /* :149-155: ANTI_SENSE -> reverse-complemented edges */
    dbg_update(m, nd, same ? edges[i] : orc_dbg_edges_revcomp(edges[i]));
  }
}

size_t orc_dbg_map_size(const orc_dbg_map *m) { return m->size; }

static void dbg_emit(const orc_dbg_map *m, const dbg_node *nd, uint64_t *key, uint32_t *c9, int canonical_orientation) {
  uint64_t rc[ORC_MAX_WORDS];
  orc_kmer_revcomp(&m->spec, nd->key, rc);
  if (canonical_orientation && orc_kmer_less(&m->spec, rc, nd->key)) {
    memcpy(key, rc, m->spec.n_words * sizeof(uint64_t));
    /* out X of the other strand = in complement(X) of this one: bit i of the edge byte moves to where
     * reverse_complement_edges would put it */
    for (int i = 0; i < 8; ++i) {
      uint8_t bit = (uint8_t)(1u << i), to = orc_dbg_edges_revcomp(bit);
      int j = 0; while (!((to >> j) & 1)) ++j;
      c9[j] = nd->counts[i];
    }
    c9[8] = nd->counts[8];
  } else {
    memcpy(key, nd->key, m->spec.n_words * sizeof(uint64_t));
    memcpy(c9, nd->counts, sizeof(uint32_t) * 9);
  }
}

size_t orc_dbg_map_export(const orc_dbg_map *m, uint64_t *keys, uint32_t *counts9, int canonical_orientation) {
  size_t j = 0;
  for (size_t b = 0; b < m->n_buckets; ++b)
    for (dbg_node *nd = m->buckets[b]; nd; nd = nd->next) {
      dbg_emit(m, nd, keys + j * m->spec.n_words, counts9 + j * 9, canonical_orientation);
      ++j;
    }
  return j;
}

/* find(): unique queries (under the map's equality) that hit, as (stored key, node) -- distributed_unordered_map.hpp:1100-1131 */
size_t orc_dbg_map_find(orc_dbg_map *m, const uint64_t *queries, size_t nq, uint64_t *out_keys, uint32_t *out_counts9,
                        int canonical_orientation) {
  size_t j = 0;
  const uint64_t stamp = ++m->stamp;
  for (size_t i = 0; i < nq; ++i) {
    dbg_node *nd = dbg_find(m, queries + i * m->spec.n_words, NULL, NULL);
    if (!nd || nd->stamp == stamp) continue;
    nd->stamp = stamp;
    dbg_emit(m, nd, out_keys + j * m->spec.n_words, out_counts9 + j * 9, canonical_orientation);
    ++j;
  }
  return j;
}

double orc_bench_count_index(const uint8_t *bytes, size_t n, uint32_t k, uint32_t strand,
                             uint32_t threads, uint64_t *n_kmers, uint64_t *n_distinct) {
  orc_kspec spec;
  if (orc_kspec_init(&spec, k, ORC_DNA) != 0 || threads == 0) return -1.0;
  uint32_t T = threads;
  size_t *pb = (size_t *)malloc(sizeof(size_t) * (T + 1));
  for (uint32_t t = 0; t < T; ++t) pb[t] = fastq_align(bytes, n, (size_t)((double)n * t / T));
  pb[T] = n;
  for (uint32_t t = 1; t <= T; ++t) if (pb[t] < pb[t - 1]) pb[t] = pb[t - 1];
  pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, T);
  uint64_t **send_buf = (uint64_t **)calloc(T, sizeof(uint64_t *));
  uint64_t **send_counts = (uint64_t **)calloc(T, sizeof(uint64_t *));
  for (uint32_t t = 0; t < T; ++t) send_counts[t] = (uint64_t *)calloc(T, sizeof(uint64_t));
  bench_arg *args = (bench_arg *)calloc(T, sizeof(bench_arg));
  pthread_t *th = (pthread_t *)calloc(T, sizeof(pthread_t));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint32_t t = 0; t < T; ++t) {
    args[t].bytes = bytes; args[t].n = n; args[t].spec = spec; args[t].strand = strand;
    args[t].T = T; args[t].part_begin = pb; args[t].bar = &bar;
    args[t].send_buf = send_buf; args[t].send_counts = send_counts; args[t].tid = t;
    pthread_create(&th[t], NULL, bench_worker, &args[t]);
  }
  uint64_t nk = 0, nd = 0;
  for (uint32_t t = 0; t < T; ++t) { pthread_join(th[t], NULL); nk += args[t].n_kmers; nd += args[t].n_distinct; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (n_kmers) *n_kmers = nk;
  if (n_distinct) *n_distinct = nd;
  for (uint32_t t = 0; t < T; ++t) free(send_counts[t]);
  free(send_counts); free(send_buf); free(args); free(th); free(pb);
  pthread_barrier_destroy(&bar);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- the whole thread-rank build in slices, with checksums of the final maps (tests: full-size parity) ------------------
 * The same path as orc_bench_count_index (parse -> canonical -> murmur KeyToRank -> stable bucket -> in-memory all-to-all ->
 * per-rank chained map), but the input goes through in `slices` record-aligned pieces and every rank keeps its map across
 * them, so the memory in flight is one slice's tuples. out[0] = k-mers, out[1] = distinct keys, out[2] = sum of counts,
 * out[3] = sum of key * count (mod 2^64), out[4] = xor of key * (2 count + 1) (mod 2^64). */
typedef struct {
  const uint8_t *bytes; size_t n;
  orc_kspec spec; uint32_t strand, T, tid;
  const size_t *part_begin;
  pthread_barrier_t *bar;
  uint64_t **send_buf; uint64_t **send_counts;
  orc_count_map **maps;
  uint64_t n_kmers;
} full_arg;

static void *full_worker(void *vp) {
  full_arg *a = (full_arg *)vp;
  const orc_kspec *s = &a->spec;
  const uint32_t T = a->T, me = a->tid;
  const size_t b = a->part_begin[me], e = a->part_begin[me + 1];
  size_t nseq = 0;
  const size_t cap = (e > b) ? (e - b) / 2 + 1024 : 1;   /* a FASTQ record gives fewer than half a tuple per byte */
  uint64_t *km = (uint64_t *)malloc(sizeof(uint64_t) * s->n_words * cap);
  long nk = (e > b) ? orc_extract(s, ORC_FMT_FASTQ, a->bytes + b, e - b, b, km, NULL, NULL, cap, &nseq) : 0;
  if (nk < 0) nk = 0;
  a->n_kmers = (uint64_t)nk;
  if (a->strand == ORC_STRAND_CANONICAL)
    for (long i = 0; i < nk; ++i) canonical_fast(s, km + (size_t)i * s->n_words, km + (size_t)i * s->n_words);
  uint32_t *ranks = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nk ? nk : 1));
  uint64_t *i2o = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nk ? nk : 1));
  orc_key_to_rank(s, ORC_HASH_MURMUR, a->strand, km, (size_t)nk, T, ranks);
  orc_stable_bucket(ranks, (size_t)nk, T, a->send_counts[me], i2o);
  uint64_t *perm = (uint64_t *)malloc(sizeof(uint64_t) * s->n_words * (size_t)(nk ? nk : 1));
  for (long i = 0; i < nk; ++i)
    memcpy(perm + i2o[i] * s->n_words, km + (size_t)i * s->n_words, s->n_words * sizeof(uint64_t));
  free(km); free(ranks); free(i2o);
  a->send_buf[me] = perm;
  pthread_barrier_wait(a->bar);
  for (uint32_t src = 0; src < T; ++src) {      /* what this rank receives, source by source, straight into its map */
    uint64_t off = 0;
    for (uint32_t d = 0; d < me; ++d) off += a->send_counts[src][d];
    orc_count_map_insert(a->maps[me], a->send_buf[src] + off * s->n_words, (size_t)a->send_counts[src][me]);
  }
  pthread_barrier_wait(a->bar);
  free(a->send_buf[me]);
  return NULL;
}

int orc_count_full(const uint8_t *bytes, size_t n, uint32_t k, uint32_t strand, uint32_t threads, uint32_t slices, uint64_t *out) {
  orc_kspec spec;
  if (orc_kspec_init(&spec, k, ORC_DNA) != 0 || threads == 0 || slices == 0 || spec.n_words != 1) return -1;
  const uint32_t T = threads;
  orc_count_map **maps = (orc_count_map **)calloc(T, sizeof(orc_count_map *));
  for (uint32_t t = 0; t < T; ++t) {
    maps[t] = orc_count_map_create(&spec, ORC_STRAND_SINGLE, ORC_HASH_MURMUR);   /* keys arrive input-transformed */
    if (strand == ORC_STRAND_BIMOLECULE) maps[t]->strand = ORC_STRAND_BIMOLECULE;
  }
  uint64_t **send_buf = (uint64_t **)calloc(T, sizeof(uint64_t *));
  uint64_t **send_counts = (uint64_t **)calloc(T, sizeof(uint64_t *));
  for (uint32_t t = 0; t < T; ++t) send_counts[t] = (uint64_t *)calloc(T, sizeof(uint64_t));
  full_arg *args = (full_arg *)calloc(T, sizeof(full_arg));
  pthread_t *th = (pthread_t *)calloc(T, sizeof(pthread_t));
  size_t *pb = (size_t *)malloc(sizeof(size_t) * (T + 1));
  uint64_t nk = 0;
  size_t s_begin = 0;
  for (uint32_t sl = 0; sl < slices; ++sl) {
    const size_t s_end = (sl + 1 == slices) ? n : fastq_align(bytes, n, (size_t)((double)n * (sl + 1) / slices));
    const size_t sn = s_end > s_begin ? s_end - s_begin : 0;
    for (uint32_t t = 0; t < T; ++t) pb[t] = s_begin + fastq_align(bytes + s_begin, sn, (size_t)((double)sn * t / T));
    pb[T] = s_begin + sn;
    for (uint32_t t = 1; t <= T; ++t) if (pb[t] < pb[t - 1]) pb[t] = pb[t - 1];
    pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, T);
    for (uint32_t t = 0; t < T; ++t) {
      memset(send_counts[t], 0, sizeof(uint64_t) * T);
      args[t].bytes = bytes; args[t].n = n; args[t].spec = spec; args[t].strand = strand; args[t].T = T; args[t].tid = t;
      args[t].part_begin = pb; args[t].bar = &bar; args[t].send_buf = send_buf; args[t].send_counts = send_counts; args[t].maps = maps;
      pthread_create(&th[t], NULL, full_worker, &args[t]);
    }
    for (uint32_t t = 0; t < T; ++t) { pthread_join(th[t], NULL); nk += args[t].n_kmers; }
    pthread_barrier_destroy(&bar);
    s_begin = s_end;
  }
  uint64_t nd = 0, sc = 0, skc = 0, x = 0;
  for (uint32_t t = 0; t < T; ++t) {
    const orc_count_map *m = maps[t];
    for (size_t b = 0; b < m->n_buckets; ++b)
      for (cm_node *node = m->buckets[b]; node; node = node->next) {
        ++nd; sc += node->count; skc += node->key[0] * (uint64_t)node->count; x ^= node->key[0] * (2ull * node->count + 1ull);
      }
    orc_count_map_destroy(maps[t]);
  }
  out[0] = nk; out[1] = nd; out[2] = sc; out[3] = skc; out[4] = x;
  for (uint32_t t = 0; t < T; ++t) free(send_counts[t]);
  free(send_counts); free(send_buf); free(args); free(th); free(pb); free(maps);
  return 0;
}

/* the record-aligned start at or after `pos` (the 4-line rule above), for the tests of the on-device partition search */
size_t orc_fastq_align(const uint8_t *bytes, size_t n, size_t pos) { return fastq_align(bytes, n, pos); }

void orc_free(void *p) { free(p); }
