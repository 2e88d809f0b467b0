// kmi_minimizer.h -- minimizer arithmetic shared by the super-k-mer build (kmi_superkmer.h) and by the bucket function
// of an index laid out by minimizer buckets (kmi_index.hip): one definition, so that the bucket a k-mer lands in during
// the build is the bucket a query for it is sent to.
#pragma once
#include "kmi_device.h"

namespace kmi {

// (W, m) by k: W = 19 for k 29..32 (m = 11..14), 13 for k 23..28 (m = 11..16), 11 for k 21, 22 (m = 11, 12), 7 for k 17..20
// (m = 11..14); an m-mer always fits 32 bits and has at least 2 M canonical values (bucket balance). Smaller k keep the k-mer
// pipeline.
__host__ __device__ inline uint32_t sk_window_of(uint32_t k) { return k >= 29u ? 19u : (k >= 23u ? 13u : (k >= 21u ? 11u : (k >= 17u ? 7u : 0u))); }
// A run entry is one lane's work and its items one lane's list: entries are cut every so many windows that they hold about
// 12 items whatever W is (a super-k-mer averages (W + 1) / 2 windows)
__host__ __device__ inline uint32_t sk_segment_of(uint32_t w) { return w >= 19u ? 128u : (w >= 13u ? 80u : (w >= 11u ? 64u : 44u)); }
// capacity of the item stream per 8 KB scan tile (26 reads of 150 bases: about 330 / 500 / 600 / 850 items are used)
__host__ __device__ inline uint32_t sk_items_per_tile(uint32_t w) { return w >= 19u ? 1024u : (w >= 13u ? 1536u : (w >= 11u ? 2048u : 2560u)); }
// longest super-k-mer kept in one record: k + n - 1 <= 51 bases (102 bits) and n - 1 in 5 bits
__host__ __device__ inline uint32_t sk_nmax_of(uint32_t k) { return (52u - k) < 32u ? (52u - k) : 32u; }

// order hash of a canonical m-mer: a bijection on 32 bits, so distinct m-mers never tie (a tie would be broken by
// position, and position order flips with the strand)
__device__ __forceinline__ uint32_t sk_order_hash(uint32_t c) {
  uint32_t h = c * 0x9E3779B1u;
  h ^= h >> 15;
  return h;
}
// bucket bits of a minimizer (from the low 25 bits of its order hash: the high bits of a MINIMUM are mostly zero, and a list entry of
// the walks keeps seven bits beside them)
__device__ __forceinline__ uint32_t sk_bucket_bits20(uint32_t hv25) {
  uint32_t h = (hv25 ^ 0x5bd1e995u) * 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h >> 12;   // 20 bits: the 18 bucket bits and two more below them (a build over ranks fills the sub-bucket bits it shifts out with them)
}
__device__ __forceinline__ uint32_t sk_bucket_bits(uint32_t hv25) { return sk_bucket_bits20(hv25) >> 2; }   // 18 bits
// the same hash, 27 bits: the 20 above and seven more below them -- the one-pass front end's items carry all of them, the records
// what fits (kmi_reduce2.h sorts a fine bucket's records by the bits beyond the bucket: k-mers that differ there never meet)
__device__ __forceinline__ uint32_t sk_bucket_bits27(uint32_t hv25) {
  uint32_t h = (hv25 ^ 0x5bd1e995u) * 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h >> 5;
}
// forward strand of an m-mer from its complement-stream window (m <= 16: 32 bits)
__device__ __forceinline__ uint32_t sk_fwd_of(uint32_t r, uint32_t m) {
  uint32_t x = __builtin_bitreverse32(~r);                    // complement codes -> forward codes, first base to the top
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);   // bit reversal swapped the two bits of every base
  return x >> (32u - 2u * m);
}

// The 18 bucket bits of a stored key (one-word 2-bit k-mer, either strand): the same minimizer the build's rolling pass finds
// for the k-mer's window -- the smallest order hash over its W = k - m + 1 canonical m-mers (forward m-mer at base p, or its
// reverse complement, whichever is smaller as an integer).
__device__ __forceinline__ uint32_t sk_key_bucket18(uint64_t key, uint32_t k, uint32_t w) {
  const uint32_t m = k - w + 1u;
  const uint32_t mmask = (m >= 16u) ? 0xffffffffu : ((1u << (2u * m)) - 1u);
  const KShape shape = make_shape(k, 2);
  const uint64_t kk[1] = {key};
  uint64_t rk[1];
  revcomp_words<1, 2>(kk, rk, shape);
  uint32_t best = 0xffffffffu;
  for (uint32_t p = 0; p < w; ++p) {
    const uint32_t f = (uint32_t)(key >> (2u * (k - m - p))) & mmask;   // forward m-mer at base p
    const uint32_t r = (uint32_t)(rk[0] >> (2u * p)) & mmask;           // its reverse complement
    const uint32_t h = sk_order_hash(f < r ? f : r);
    best = h < best ? h : best;
  }
  return sk_bucket_bits(best & 0x1ffffffu);
}

}  // namespace kmi
