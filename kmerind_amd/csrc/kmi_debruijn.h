// De Bruijn graph nodes: the reference's in-tree consumer of Index (test/test/debruijn/, driven by
// test/test/test_de_bruijn_graph_construction.cpp:96-133) on the same extract -> partition -> reduce kernels.
//
//   de_bruijn_parser (de_bruijn_construct_engine.hpp:90-158)     -> the extract pass in its edge form (fastq_extract_kernel, raw_edges) for a
//                                                                   build; dbg_edges_kernel over position tuples for the parser's own output
//   de_bruijn_nodes_distributed::local_insert (..._distributed.hpp:91-159) with
//   node::edge_counts<DNA16, int32_t> / node::edge_exists<DNA16> (de_bruijn_node_trait.hpp:139-336)
//                                                                -> weighted count insert of (k-mer, 1 | edge << 32) records
//                                                                   + dbg_accumulate_kernel per fine bucket
//
// A node is (k-mer, [out A C G T, in A C G T, occurrences]). The reference finds a node under either strand and keeps the
// strand that reached the map first (BimoleculeHashMapParams, kmer_index.hpp:468-481) -- an order that depends on how
// MPI delivered the tuples; here every node is kept in the orientation of its lexicographically smaller strand (the edge byte
// of a k-mer that is stored reverse-complemented goes through reverse_complement_edges, de_bruijn_node_trait.hpp:122-124, as
// the reference does for an ANTI_SENSE insert). The node SET and, per node up to that orientation, the counts are the same.
//
// Included by kmi_index.hip (it uses the table, partition and insert machinery of that translation unit).
#pragma once

namespace kmi {

constexpr uint32_t kDbgValueWords = 5;   // a node's value as 64-bit words: uint32_t counts[9] + one word of padding

// The value word of a parsed tuple -> the edge byte of its k-mer.
//   from_ids: the word holds the ShortSequenceKmerId of the extract pass (record offset << 16 | offset of the k-mer's first
//             base, file_offset 0): the bases around the k-mer are bytes[pos - 1] and bytes[pos + k] of the FASTQ sequence line
//             (edge_iterator.hpp:163-177: DNA16 code of the left base << 4 | code of the right base, 0 where the read ends).
//             Otherwise the word already holds the edge byte (insert of caller-made tuples).
//   node_form: the key becomes the smaller of the k-mer and its reverse complement (the edge byte follows it) and the value word
//             1 | edge << 32: weight 1 for the count insert, the edge byte above it for the accumulate pass.
//             Otherwise the tuple stays as parsed and the word is the edge byte (what de_bruijn_parser emits).
template <int NW, int BITS>
__global__ __launch_bounds__(256) void dbg_edges_kernel(uint64_t *__restrict__ recs, uint64_t n, const uint8_t *__restrict__ bytes,
                                                       uint64_t n_bytes, KShape shape, bool from_ids, bool node_form) {
  constexpr int RW = NW + 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t key[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) key[w] = recs[i * RW + w];
    const uint64_t v = recs[i * RW + NW];
    uint32_t e;
    if (from_ids) {
      const uint64_t pos = (v >> 16) + (v & 0xFFFFull);
      const uint32_t lc = pos > 0 ? bytes[pos - 1] : (uint32_t)'\n';
      const uint32_t rc = pos + shape.k < n_bytes ? bytes[pos + shape.k] : (uint32_t)'\n';
      e = ((is_eol(lc) ? 0u : code_dna16(lc)) << 4) | (is_eol(rc) ? 0u : code_dna16(rc));
    } else {
      e = (uint32_t)v & 0xFFu;
    }
    if (node_form) {
      uint64_t rck[NW];
      revcomp_words<NW, BITS>(key, rck, shape);
      if (less_words<NW>(rck, key)) {
#pragma unroll
        for (int w = 0; w < NW; ++w) recs[i * RW + w] = rck[w];
        e = (comp_code<4>(e & 0xFu) << 4) | comp_code<4>(e >> 4);   // reverse_complement_edges<DNA16>
      }
      recs[i * RW + NW] = 1ull | ((uint64_t)e << 32);
    } else {
      recs[i * RW + NW] = e;
    }
  }
}

// The accumulate pass keeps, per workgroup, a table of one chunk of a bucket's nodes (key -> row) and eight counters per row.
template <int NW> struct DbgCfg {
  static constexpr int CAP = (NW == 1) ? 6144 : 2048;    // home slots
  static constexpr int PAD = TabCfg<NW>::PAD;
  static constexpr int SLOTS = CAP + PAD;
  static constexpr int ROWS = CAP * 3 / 4;               // nodes per chunk (one-word keys: 4608, i.e. 1.5e8 nodes per index before a
                                                         // bucket of average size needs a second chunk)
  static constexpr int NT = 1024;
};

// Fine bucket b: edges[node][0..7] = the edge bits of the bucket's records added up (+ the counts of the nodes that were there
// before the insert, whose positions have changed). The nodes of the bucket are distinct keys at known positions, so the table
// is filled once per chunk of ROWS nodes and only read afterwards.
//
// Counters: 16 bits each in LDS, two to a word, exact beyond that: an add returns the word as it was, so exactly one add sees
// each carry out of a low half (n1 of them) and each wrap of the whole word (n2); with the halves lo, hi that are left,
//   sum of the low counter  = lo + 65536 n1          sum of the high counter = hi + 65536 n2 - n1     (mod 2^32, as int32 wraps)
// The adds that see one send the correction to `edges` with a global atomic and mark the row; a marked row leaves with atomic
// adds too, every other row (all of them, outside homopolymer-like repeats) as plain stores into the zero-filled array.
template <int NW>
__global__ __launch_bounds__((DbgCfg<NW>::NT)) void dbg_accumulate_kernel(const uint64_t *__restrict__ idx_keys, const uint64_t *__restrict__ idx_off,
                                                                         const uint64_t *__restrict__ recs, const uint64_t *__restrict__ rec_off,
                                                                         const uint64_t *__restrict__ old_keys, const uint32_t *__restrict__ old_edges,
                                                                         const uint64_t *__restrict__ old_off, uint32_t *__restrict__ edges) {
  using Cfg = DbgCfg<NW>;
  constexpr int RW = NW + 1;
  __shared__ uint64_t s_tk[Cfg::SLOTS * NW];
  __shared__ uint32_t s_tt[(NW == 1) ? 1 : Cfg::SLOTS];
  __shared__ uint16_t s_row[Cfg::SLOTS];
  __shared__ uint32_t s_cnt[Cfg::ROWS * 4];
  __shared__ uint32_t s_mark[(Cfg::ROWS + 31) / 32];
  __shared__ uint32_t s_ctl[8];
  LdsTable<NW> tab;
  tab.keys = s_tk; tab.vals = nullptr; tab.tags = s_tt; tab.distinct = &s_ctl[0]; tab.overflow = &s_ctl[1];
  tab.special = &s_ctl[2]; tab.special_set = &s_ctl[3]; tab.progress = &s_ctl[5];
  tab.cap = Cfg::CAP; tab.slots = Cfg::SLOTS; tab.limit = 2u * Cfg::CAP;   // (never "too loaded": the chunk size bounds the load)
  uint32_t *s_fail = &s_ctl[6], *s_special_row = &s_ctl[7];
  const uint32_t b = blockIdx.x;
  const uint64_t ib = idx_off[b], ie = idx_off[b + 1];
  if (ib == ie) return;
  const uint64_t rb = recs ? rec_off[b] : 0ull, re = recs ? rec_off[b + 1] : 0ull;
  const uint64_t ob = old_keys ? old_off[b] : 0ull, oe = old_keys ? old_off[b + 1] : 0ull;
  if (rb == re && ob == oe) return;
  uint64_t i0 = ib;
  auto row_of = [&](const uint64_t (&k)[NW]) -> uint32_t {
    const int s = table_find<NW>(tab, k, place_hash<NW>(k));
    return s >= 0 ? (uint32_t)s_row[s] : (s == -2 ? *s_special_row : ~0u);
  };
  // counter t of `row` += a (a < 65536)
  auto add = [&](uint32_t row, uint32_t t, uint32_t a) {
    uint32_t *g = edges + (i0 + row) * 8u;
    if (t & 1u) {
      const uint32_t old = atomicAdd(&s_cnt[row * 4u + (t >> 1)], a << 16);
      if ((old >> 16) + a > 0xFFFFu) { atomicOr(&s_mark[row >> 5], 1u << (row & 31u)); atomicAdd(&g[t], 65536u); }
    } else {
      const uint32_t old = atomicAdd(&s_cnt[row * 4u + (t >> 1)], a);
      if ((old & 0xFFFFu) + a > 0xFFFFu) {
        atomicOr(&s_mark[row >> 5], 1u << (row & 31u));
        atomicAdd(&g[t], 65536u);
        atomicAdd(&g[t + 1u], 0xFFFFFFFFu);                                   // the carry sits in the neighbour
        if ((uint64_t)old + a > 0xFFFFFFFFull) atomicAdd(&g[t + 1u], 65536u);   // ... and made the word wrap
      }
    }
  };
  uint32_t chunk = Cfg::ROWS;
  while (i0 < ie) {
    const uint32_t nc = (uint32_t)((ie - i0) < (uint64_t)chunk ? (ie - i0) : (uint64_t)chunk);
    for (uint32_t x = threadIdx.x; x < (uint32_t)Cfg::SLOTS; x += blockDim.x) { if (NW == 1) s_tk[x] = kEmptyKey; else s_tt[x] = kTagEmpty; }
    for (uint32_t x = threadIdx.x; x < nc * 4u; x += blockDim.x) s_cnt[x] = 0;
    for (uint32_t x = threadIdx.x; x < (nc + 31u) / 32u; x += blockDim.x) s_mark[x] = 0;
    if (threadIdx.x == 0) { s_ctl[0] = s_ctl[1] = s_ctl[2] = s_ctl[3] = s_ctl[5] = 0; *s_fail = 0; *s_special_row = ~0u; }
    lds_barrier();
    for (uint32_t j = threadIdx.x; j < nc; j += blockDim.x) {
      uint64_t k[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) k[w] = idx_keys[(i0 + j) * NW + w];
      const int s = table_upsert<NW>(tab, k, place_hash<NW>(k));
      if (s >= 0) s_row[s] = (uint16_t)j; else if (s == -2) *s_special_row = j; else *s_fail = 1;
    }
    lds_barrier();
    if (*s_fail) {   // a probe sequence ran off the end of the table (one-word keys do not wrap around): fewer nodes per chunk
      lds_barrier();
      chunk = chunk > 64u ? chunk / 2u : 32u;   // 32 nodes always fit the 64 slots of padding
      continue;
    }
    for (uint64_t i = rb + threadIdx.x; i < re; i += blockDim.x) {
      uint32_t e = (uint32_t)(recs[i * RW + NW] >> 32) & 0xFFu;
      if (e == 0u) continue;
      uint64_t k[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) k[w] = recs[i * RW + w];
      const uint32_t row = row_of(k);
      if (row == ~0u) continue;   // a node of another chunk
      while (e) { add(row, (uint32_t)__builtin_ctz(e), 1u); e &= e - 1u; }
    }
    for (uint64_t i = ob + threadIdx.x; i < oe; i += blockDim.x) {
      uint64_t k[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) k[w] = old_keys[i * NW + w];
      const uint32_t row = row_of(k);
      if (row == ~0u) continue;
#pragma unroll
      for (uint32_t t = 0; t < 8u; ++t) {
        const uint32_t c = old_edges[i * 8u + t];
        if (c & 0xFFFFu) add(row, t, c & 0xFFFFu);
        if (c >> 16) { atomicOr(&s_mark[row >> 5], 1u << (row & 31u)); atomicAdd(&edges[(i0 + row) * 8u + t], c & 0xFFFF0000u); }
      }
    }
    lds_barrier();
    for (uint32_t x = threadIdx.x; x < nc * 8u; x += blockDim.x) {
      const uint32_t row = x >> 3, t = x & 7u;
      const uint32_t c = (s_cnt[row * 4u + (t >> 1)] >> ((t & 1u) * 16u)) & 0xFFFFu;
      if ((s_mark[row >> 5] >> (row & 31u)) & 1u) { if (c) atomicAdd(&edges[i0 * 8u + x], c); }
      else edges[i0 * 8u + x] = c;
    }
    lds_barrier();
    i0 += nc;
  }
}

// The same sums from SUPER-K-MER records (the build through the count index's own front and back end, dbg_build_superkmer below).
// A record (kmi_superkmer.h) is a stretch of a read: k + n - 1 bases as complement codes, base i at bits 2 i, in whichever of its
// two orientations is the smaller number, + the base before its first and behind its last k-mer (kRecEdgeShift). Its k-mer j has the
// neighbours base j - 1 and base j + k of the SAME record -- the two outside bases for j = 0 and j = n - 1 -- so nothing but the
// record is read: edge_iterator.hpp:84-177 without a tuple per k-mer. The records of fine bucket b are still where the count
// build's scatter pass left them; the nodes of b are the (dense) entries [idx_off[b], idx_off[b + 1]) of the node index, which the
// build laid out by the same minimizer buckets. One lane per record rolls through its k-mers: window -> both strands -> the smaller
// one is the node (the edges of a node kept reverse-complemented change sides and are complemented, de_bruijn_node_trait.hpp:
// 122-124) -> row -> two counter adds. Counters and chunks of rows as in dbg_accumulate_kernel.
constexpr int kSkEdgeOwn = 64 * 10;   // unit marks of a wavefront's batch: 64 records of up to 9 units (18 k-mers) + slack
__global__ __launch_bounds__((DbgCfg<1>::NT)) void sk_edges_accumulate_kernel(const uint64_t *__restrict__ idx_keys, const uint64_t *__restrict__ idx_off,
                                                                           const uint64_t *__restrict__ recs, const uint64_t *__restrict__ rec_off,
                                                                           const uint64_t *__restrict__ fine_region, const uint32_t *__restrict__ fine_cap,
                                                                           const uint32_t *__restrict__ fine_cnt, uint32_t k, uint32_t *__restrict__ edges) {
  using Cfg = DbgCfg<1>;
  __shared__ uint64_t s_tk[Cfg::SLOTS];
  __shared__ uint32_t s_tt[1];
  __shared__ uint16_t s_row[Cfg::SLOTS];
  __shared__ uint32_t s_cnt[Cfg::ROWS * 4];
  __shared__ uint32_t s_mark[(Cfg::ROWS + 31) / 32];
  __shared__ uint8_t s_own[(Cfg::NT / kWave) * kSkEdgeOwn];
  __shared__ uint32_t s_ctl[8];
  LdsTable<1> tab;
  tab.keys = s_tk; tab.vals = nullptr; tab.tags = s_tt; tab.distinct = &s_ctl[0]; tab.overflow = &s_ctl[1];
  tab.special = &s_ctl[2]; tab.special_set = &s_ctl[3]; tab.progress = &s_ctl[5];
  tab.cap = Cfg::CAP; tab.slots = Cfg::SLOTS; tab.limit = 2u * Cfg::CAP;
  uint32_t *s_fail = &s_ctl[6], *s_special_row = &s_ctl[7];
  const uint32_t b = blockIdx.x;
  const uint64_t ib = idx_off[b], ie = idx_off[b + 1];
  if (ib == ie) return;
  for (uint32_t x = threadIdx.x; x < (uint32_t)sizeof(s_own) / 4u; x += blockDim.x) reinterpret_cast<uint32_t *>(s_own)[x] = 0;
  uint64_t rb, re;
  if (fine_cnt) {
    const uint32_t cap = fine_cap[b >> 7], cnt = fine_cnt[b];
    rb = fine_region[b >> 7] + (uint64_t)(b & 127u) * cap; re = rb + (cnt < cap ? cnt : cap);
  } else { rb = rec_off[b]; re = rec_off[b + 1]; }
  if (rb == re) return;
  uint64_t i0 = ib;
  auto add = [&](uint32_t row, uint32_t t, uint32_t a) {   // counter t of `row` += a, a < 65536 (dbg_accumulate_kernel's add)
    uint32_t *g = edges + (i0 + row) * 8u;
    if (t & 1u) {
      const uint32_t old = atomicAdd(&s_cnt[row * 4u + (t >> 1)], a << 16);
      if ((old >> 16) + a > 0xFFFFu) { atomicOr(&s_mark[row >> 5], 1u << (row & 31u)); atomicAdd(&g[t], 65536u); }
    } else {
      const uint32_t old = atomicAdd(&s_cnt[row * 4u + (t >> 1)], a);
      if ((old & 0xFFFFu) + a > 0xFFFFu) {
        atomicOr(&s_mark[row >> 5], 1u << (row & 31u));
        atomicAdd(&g[t], 65536u);
        atomicAdd(&g[t + 1u], 0xFFFFFFFFu);
        if ((uint64_t)old + a > 0xFFFFFFFFull) atomicAdd(&g[t + 1u], 65536u);
      }
    }
  };
  const uint32_t kb = 2u * k;
  const uint64_t kmask = kb >= 64u ? ~0ull : ((1ull << kb) - 1ull);
  const KShape shape = make_shape(k, 2u);
  uint32_t chunk = Cfg::ROWS;
  while (i0 < ie) {
    const uint32_t nc = (uint32_t)((ie - i0) < (uint64_t)chunk ? (ie - i0) : (uint64_t)chunk);
    for (uint32_t x = threadIdx.x; x < (uint32_t)Cfg::SLOTS; x += blockDim.x) s_tk[x] = kEmptyKey;
    for (uint32_t x = threadIdx.x; x < nc * 4u; x += blockDim.x) s_cnt[x] = 0;
    for (uint32_t x = threadIdx.x; x < (nc + 31u) / 32u; x += blockDim.x) s_mark[x] = 0;
    if (threadIdx.x == 0) { s_ctl[0] = s_ctl[1] = s_ctl[2] = s_ctl[3] = s_ctl[5] = 0; *s_fail = 0; *s_special_row = ~0u; }
    lds_barrier();
    for (uint32_t j = threadIdx.x; j < nc; j += blockDim.x) {
      const uint64_t kk[1] = {idx_keys[i0 + j]};
      const int s = table_upsert<1>(tab, kk, sk_slot_hash(kk[0]));   // (the reduce's slot hash: 24-bit multiplies -- this table is private to the kernel)
      if (s >= 0) s_row[s] = (uint16_t)j; else if (s == -2) *s_special_row = j; else *s_fail = 1;
    }
    lds_barrier();
    if (*s_fail) {
      lds_barrier();
      chunk = chunk > 64u ? chunk / 2u : 32u;
      continue;
    }
    // The records, 64 to a wavefront at a time, their k-mers dealt out as sk_reduce's expansion deals them out: a UNIT is two
    // neighbouring k-mers of one record, the units of the batch are numbered through the records' prefix sums, and lane l of step t
    // takes unit 64 t + l whatever record it belongs to -- the record is the last one that starts at or before the unit (marks in a
    // byte array, a running maximum over the lanes), its words come over the lane crossbar. So every lane works in every step although
    // the records hold 1 to 18 k-mers (one lane per record: a lane's turn was as long as its record and the others waited). The two
    // k-mers of a unit start their table walks together.
    {
      const uint32_t lane = lane_id(), wv = wave_id();
      uint8_t *const wown = s_own + wv * kSkEdgeOwn;
      const uint32_t kmask_hi = kb >= 64u ? 0xffffffffu : ((1u << (kb - 32u)) - 1u), pad = 64u - kb, top_sh = kb - 34u;   // (k >= 17)
      for (uint64_t r0 = rb + (uint64_t)wv * kWave; r0 < re; r0 += blockDim.x) {   // (uniform per wavefront)
        ulonglong2 rec = make_ulonglong2(0, 0);
        const bool have = r0 + lane < re;
        if (have) rec = reinterpret_cast<const ulonglong2 *>(recs)[r0 + lane];
        const uint32_t n = (have && rec.y != kSkPadW1) ? ((uint32_t)(rec.y >> kRecNShift) & 31u) + 1u : 0u;
        const uint32_t nu = (n + 1u) >> 1;
        const uint32_t inc = wave_inclusive_sum_dpp(nu);
        const uint32_t pre = inc - nu;
        const uint32_t total = __builtin_amdgcn_readlane(inc, kWave - 1);
        if (nu) wown[pre] = (uint8_t)(lane + 1u);
        uint32_t carry = 0;
        for (uint32_t g0 = 0; g0 < total; g0 += kWave) {
          const uint32_t g = g0 + lane;
          const bool act = g < total;
          uint32_t o = act ? (uint32_t)wown[g] : 0u;
          o = wave_inclusive_max_dpp(o);
          o = o > carry ? o : carry;
          carry = __builtin_amdgcn_readlane(o, kWave - 1);
          const int rl = (int)((o ? o - 1u : 0u) << 2);   // byte address of the lane that holds the record
          const uint32_t j = 2u * (g - (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)pre));
          const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)rec.x), a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)(rec.x >> 32));
          const uint32_t a2 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)rec.y), a3 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)(rec.y >> 32));
          const uint32_t rn = ((a3 >> (kRecNShift - 32)) & 31u) + 1u;                       // k-mers of the record
          const uint32_t lout = (a3 >> (kRecEdgeShift - 32)) & 7u, rout = (a3 >> (kRecEdgeShift - 29)) & 7u;
          const bool two = act && j + 1u < rn;
          // complement code of base i of the record (bases sit below bit 96: words a0..a2)
          auto cc = [&](uint32_t i) -> uint32_t { const uint32_t w = i >= 32u ? a2 : (i >= 16u ? a1 : a0); return (w >> (2u * (i & 15u))) & 3u; };
          // k-mer j: 2 k bits from bit 2 j (j is even, at most 16: the window starts in word 0 or 1) -- sk_reduce's cut, on 32-bit registers
          const bool w1sel = j >= 16u;
          const uint32_t bit = (2u * j) & 31u;
          const uint32_t b0 = w1sel ? a1 : a0, b1 = w1sel ? a2 : a1, b2 = w1sel ? (a3 & 0u) : a2;   // (nothing of word 3 is a base)
          const uint32_t rc_lo = __builtin_amdgcn_alignbit(b1, b0, bit);
          const uint32_t rc_hi = __builtin_amdgcn_alignbit(b2, b1, bit) & kmask_hi;
          const uint32_t r_hi = __builtin_bitreverse32(rc_lo), r_lo = __builtin_bitreverse32(rc_hi);
          const uint32_t s_hi = ~(((r_hi >> 1) & 0x55555555u) | ((r_hi << 1) & 0xAAAAAAAAu));
          const uint32_t s_lo = ~(((r_lo >> 1) & 0x55555555u) | ((r_lo << 1) & 0xAAAAAAAAu));
          const uint32_t fw_lo = __builtin_amdgcn_alignbit(s_hi, s_lo, pad), fw_hi = s_hi >> pad;
          const uint32_t rc2_lo = __builtin_amdgcn_alignbit(b1, b0, bit + 2u);
          const uint32_t rc2_hi = __builtin_amdgcn_alignbit(b2, b1, bit + 2u) & kmask_hi;
          const uint32_t c_jk = (rc2_hi >> top_sh) & 3u;                                   // complement code of base j + k (the second k-mer's last base)
          const uint32_t fw2_lo = (fw_lo << 2) | (c_jk ^ 3u);
          const uint32_t fw2_hi = __builtin_amdgcn_alignbit(fw_hi, fw_lo, 30u) & kmask_hi;
          const uint64_t rc = (uint64_t)rc_lo | ((uint64_t)rc_hi << 32), fw = (uint64_t)fw_lo | ((uint64_t)fw_hi << 32);
          const uint64_t rc2 = (uint64_t)rc2_lo | ((uint64_t)rc2_hi << 32), fw2 = (uint64_t)fw2_lo | ((uint64_t)fw2_hi << 32);
          // neighbours as 1 + base code (base code = 3 - complement code), 0: none
          const uint32_t in_a = j ? 4u - cc(j - 1u) : lout;
          const uint32_t out_a = (j + 1u < rn) ? 4u - c_jk : rout;
          const uint32_t in_c = 4u - (rc_lo & 3u);                                         // base j
          const uint32_t out_c = (j + 2u < rn) ? 4u - cc(j + k + 1u) : rout;
          const bool rev_a = rc < fw, rev_c = rc2 < fw2;                                   // the node is the reverse strand: edges change sides, complemented
          const uint64_t ka = rev_a ? rc : fw, kc = rev_c ? rc2 : fw2;
          const uint32_t ia = rev_a ? (out_a ? 5u - out_a : 0u) : in_a, oa = rev_a ? (in_a ? 5u - in_a : 0u) : out_a;
          const uint32_t ic = rev_c ? (out_c ? 5u - out_c : 0u) : in_c, oc = rev_c ? (in_c ? 5u - in_c : 0u) : out_c;
          const bool va = act && (ia | oa), vc = two && (ic | oc);
          uint32_t sa = slot_of(sk_slot_hash(ka), (int)Cfg::CAP), sc = slot_of(sk_slot_hash(kc), (int)Cfg::CAP);
          uint64_t ta = s_tk[sa], tc = s_tk[sc];
          uint32_t ra = ~0u, rcw = ~0u;
          if (va) { if (ka == kEmptyKey) ra = *s_special_row; else { while (ta != ka && ta != kEmptyKey) ta = s_tk[++sa]; if (ta == ka) ra = s_row[sa]; } }
          if (vc) { if (kc == kEmptyKey) rcw = *s_special_row; else { while (tc != kc && tc != kEmptyKey) tc = s_tk[++sc]; if (tc == kc) rcw = s_row[sc]; } }
          if (ra != ~0u) { if (oa) add(ra, oa - 1u, 1u); if (ia) add(ra, 3u + ia, 1u); }
          if (rcw != ~0u) { if (oc) add(rcw, oc - 1u, 1u); if (ic) add(rcw, 3u + ic, 1u); }
        }
        if (nu) wown[pre] = 0;   // the marks go back to zero for the next batch
      }
    }
    lds_barrier();
    for (uint32_t x = threadIdx.x; x < nc * 8u; x += blockDim.x) {
      const uint32_t row = x >> 3, t = x & 7u;
      const uint32_t c = (s_cnt[row * 4u + (t >> 1)] >> ((t & 1u) * 16u)) & 0xFFFFu;
      if ((s_mark[row >> 5] >> (row & 31u)) & 1u) { if (c) atomicAdd(&edges[i0 * 8u + x], c); }
      else edges[i0 * 8u + x] = c;
    }
    lds_barrier();
    i0 += nc;
  }
}

// find(): the node values of the hits. pos = entry positions (what the count index's find reports with find_emits_index);
// out = kDbgValueWords words per node: counts[0..7] (exists_only: 0 / 1), counts[8] = occurrences (exists_only: 0), padding
__global__ __launch_bounds__(256) void dbg_gather_kernel(const uint64_t *__restrict__ pos, uint64_t n, const uint32_t *__restrict__ edges,
                                                        const uint32_t *__restrict__ self, bool exists_only, uint64_t *__restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = pos[i];
    uint32_t c[10];
#pragma unroll
    for (int t = 0; t < 8; ++t) { const uint32_t v = edges[p * 8u + t]; c[t] = exists_only ? (v ? 1u : 0u) : v; }
    c[8] = exists_only ? 0u : self[p];
    c[9] = 0u;
#pragma unroll
    for (int w = 0; w < 5; ++w) out[i * kDbgValueWords + w] = (uint64_t)c[2 * w] | ((uint64_t)c[2 * w + 1] << 32);
  }
}

}  // namespace kmi

struct kmi_dbg {
  kmi_ctx *ctx = nullptr;
  kmi_config cfg{};          // as given (alphabet, k, hashes); the strand model of the node map is fixed (see the header comment)
  KShape shape{};
  uint32_t node_kind = 0;    // KMI_DBG_EDGE_COUNTS / KMI_DBG_EDGE_EXISTS
  kmi_index *nodes = nullptr;   // canonical k-mer -> occurrences (a count index, laid out by the placement hash)
  uint32_t *edges = nullptr;    // [n_entries][8]: out A C G T, in A C G T, in the order of nodes->keys
  size_t edges_bytes = 0;
};

namespace kmi {

template <int NW, int BITS>
static kmi_status dbg_edges_impl(kmi_ctx *ctx, uint64_t *recs, size_t n, const uint8_t *bytes_dev, size_t n_bytes, KShape shape, bool from_ids,
                                 bool node_form) {
  if (n == 0) return KMI_OK;
  ProfScope ps(ctx, "dbg_edges", n);
  hipLaunchKernelGGL((dbg_edges_kernel<NW, BITS>), dim3(4096), dim3(256), 0, ctx->stream, recs, (uint64_t)n, bytes_dev, (uint64_t)n_bytes, shape,
                     from_ids, node_form);
  KMI_HIP(ctx, hipGetLastError());
  return KMI_OK;
}
static kmi_status dbg_edges(kmi_ctx *ctx, uint64_t *recs, size_t n, const uint8_t *bytes_dev, size_t n_bytes, KShape shape, bool from_ids,
                            bool node_form) {
  KMI_DISPATCH(shape, dbg_edges_impl, ctx, recs, n, bytes_dev, n_bytes, shape, from_ids, node_form);
}

// the tuples of a FASTQ buffer as records (key words, value word) in WS_DBG_RECS; the value word is what dbg_edges_kernel leaves
static kmi_status dbg_parse(kmi_ctx *ctx, const kmi_config *cfg, const uint8_t *bytes_dev, size_t n_bytes, bool node_form, uint64_t **recs_out,
                            uint64_t *n_out) {
  *recs_out = nullptr; *n_out = 0;
  if (n_bytes == 0) return KMI_OK;
  kmi_config c = *cfg;
  c.index_kind = KMI_INDEX_POSITION; c.strand = KMI_STRAND_SINGLE; c.seq_filter = KMI_SEQ_ALL; c.dist_trans = KMI_DIST_MODEL;
  const bool fasta = c.seq_format == KMI_FMT_FASTA;
  KShape shape;
  if (!valid_config(&c, &shape)) return set_err(ctx, KMI_ERR_INVALID, "bad kmi_config");
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  uint64_t nt = 0, ns = 0;
  KMI_TRY(extract_count(ctx, &c, bytes_dev, n_bytes, &nt, &ns));
  if (nt == 0) return KMI_OK;
  void *dr;
  const uint32_t rw = shape.n_words + 1u;
  KMI_TRY(ws_get(ctx, WS_DBG_RECS, ((size_t)nt + 8) * rw * sizeof(uint64_t), &dr));
  // node form: the extract pass writes the smaller strand and 1 | edge << 32 itself; the parser's form (tuples as parsed) takes
  // the position ids and turns them into edge bytes in a second pass
  if (fasta) {
    // FASTA (the parser is generic over the sequence type, de_bruijn_construct_engine.hpp:108-158): the extract pass over the compacted
    // character stream looks the neighbours up through the characters' file positions and leaves (k-mer as parsed, edge byte)
    KMI_TRY(extract_run(ctx, &c, bytes_dev, n_bytes, 0, (uint64_t *)dr, nullptr, (size_t)nt, false, true, &nt, &ns, nullptr, rw, true));
    if (node_form) KMI_TRY(dbg_edges(ctx, (uint64_t *)dr, (size_t)nt, nullptr, 0, shape, false, true));
    *recs_out = (uint64_t *)dr; *n_out = nt;
    return KMI_OK;
  }
  KMI_TRY(extract_run(ctx, &c, bytes_dev, n_bytes, 0, (uint64_t *)dr, nullptr, (size_t)nt, false, true, &nt, &ns, nullptr, rw, node_form));
  if (!node_form) KMI_TRY(dbg_edges(ctx, (uint64_t *)dr, (size_t)nt, bytes_dev, n_bytes, shape, true, false));
  *recs_out = (uint64_t *)dr; *n_out = nt;
  return KMI_OK;
}

// A graph built through super-k-mers keeps its nodes by minimizer bucket; whatever inserts tuples into it partitions them by the
// placement hash, and the node index changes over (ensure_layout) -- the edge counts have to follow the nodes. The re-layout is a
// pure permutation of (key, count) pairs: with the counts replaced by the entries' old positions for its duration, every entry
// arrives with the place it came from, and one gather brings its count and its eight edge counters along.
__global__ __launch_bounds__(256) void dbg_iota_kernel(uint32_t *__restrict__ v, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void dbg_follow_kernel(const uint32_t *__restrict__ vals /* the entries' old positions */, uint64_t n,
                                                        const uint32_t *__restrict__ old_edges, uint32_t *__restrict__ edges) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 8u; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = i >> 3, t = i & 7u;
    edges[i] = old_edges[(uint64_t)vals[p] * 8u + t];
  }
}
// (a launch of its own behind dbg_follow_kernel, which reads the old position of every entry)
__global__ __launch_bounds__(256) void dbg_follow_counts_kernel(uint32_t *__restrict__ vals, uint64_t n, const uint32_t *__restrict__ old_counts) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) vals[i] = old_counts[vals[i]];
}
static kmi_status dbg_to_placement_layout(kmi_dbg *g) {
  kmi_ctx *ctx = g->ctx;
  kmi_index *idx = g->nodes;
  if (idx->layout_w == 0u || !idx->has_data || idx->n_entries == 0) return KMI_OK;
  KMI_TRY(ensure_dense(idx));
  const uint64_t n = idx->n_entries;
  if (n > 0xFFFFFFFFull) return set_err(ctx, KMI_ERR_OVERFLOW, "more than 2^32 nodes in one graph");
  void *p;
  KMI_TRY(ws_get(ctx, WS_DBG_OLD, (size_t)n * sizeof(uint32_t) + 64, &p));
  uint32_t *old_counts = (uint32_t *)p;
  KMI_HIP(ctx, hipMemcpyAsync(old_counts, idx->vals, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
  hipLaunchKernelGGL(dbg_iota_kernel, dim3(2048), dim3(256), 0, ctx->stream, idx->vals, n);
  KMI_TRY(ensure_layout(idx, 0u));
  uint32_t *ne = nullptr;
  const size_t eb = (size_t)n * 8 * sizeof(uint32_t);
  if (pool_alloc(ctx, (void **)&ne, eb) != hipSuccess) return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the edge counts");
  {
    ProfScope ps(ctx, "dbg_follow", n);
    hipLaunchKernelGGL(dbg_follow_kernel, dim3(4096), dim3(256), 0, ctx->stream, (const uint32_t *)idx->vals, n, (const uint32_t *)g->edges, ne);
    hipLaunchKernelGGL(dbg_follow_counts_kernel, dim3(2048), dim3(256), 0, ctx->stream, idx->vals, n, (const uint32_t *)old_counts);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pool_free(ctx, g->edges, g->edges_bytes);
  g->edges = ne; g->edges_bytes = eb;
  return KMI_OK;
}

// nodes.insert(tuples): records in node form (canonical key, 1 | edge << 32)
template <int NW, int BITS>
static kmi_status dbg_insert_impl(kmi_dbg *g, const uint64_t *recs_dev, size_t n) {
  kmi_ctx *ctx = g->ctx;
  kmi_index *idx = g->nodes;
  if (n == 0) return KMI_OK;
  KMI_TRY(dbg_to_placement_layout(g));   // (a graph built through super-k-mers: the tuples below are partitioned by the placement hash)
  // the nodes that are there keep their edge counts but not their positions: their keys and bucket offsets are set aside
  const uint64_t n_old = idx->has_data ? idx->n_entries : 0;
  uint64_t *old_keys = nullptr, *old_off = nullptr;
  if (n_old) {
    void *p;
    const size_t kb = (size_t)n_old * NW * sizeof(uint64_t);
    KMI_TRY(ws_get(ctx, WS_DBG_OLD, kb + kOffBytes + 64, &p));
    old_keys = (uint64_t *)p; old_off = (uint64_t *)((uint8_t *)p + ((kb + 15) & ~(size_t)15));
    KMI_HIP(ctx, hipMemcpyAsync(old_keys, idx->keys, kb, hipMemcpyDeviceToDevice, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(old_off, idx->bucket_off, kOffBytes, hipMemcpyDeviceToDevice, ctx->stream));
  }
  Partitioned part;
  KMI_TRY(index_insert_pairs(idx, recs_dev, n, false, false, &part));   // occurrences: the low half of the value word is the weight 1
  const size_t eb = (size_t)(idx->n_entries ? idx->n_entries : 1) * 8 * sizeof(uint32_t);
  uint32_t *ne = nullptr;
  if (pool_alloc(ctx, (void **)&ne, eb) != hipSuccess) return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the edge counts");
  KMI_HIP(ctx, hipMemsetAsync(ne, 0, eb, ctx->stream));
  {
    ProfScope ps(ctx, "dbg_accumulate", n);
    hipLaunchKernelGGL((dbg_accumulate_kernel<NW>), dim3(kNumFine), dim3(DbgCfg<NW>::NT), 0, ctx->stream, (const uint64_t *)idx->keys,
                       (const uint64_t *)idx->bucket_off, (const uint64_t *)part.keys, (const uint64_t *)part.fine_off, (const uint64_t *)old_keys,
                       (const uint32_t *)g->edges, (const uint64_t *)old_off, ne);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (g->edges) pool_free(ctx, g->edges, g->edges_bytes);
  g->edges = ne; g->edges_bytes = eb;
  return KMI_OK;
}
static kmi_status dbg_insert(kmi_dbg *g, const uint64_t *recs_dev, size_t n) { KMI_DISPATCH(g->shape, dbg_insert_impl, g, recs_dev, n); }

// The node build of an EMPTY graph from FASTQ through the count index's super-k-mer build (de_bruijn_construct_engine.hpp:90-158 +
// de_bruijn_nodes_distributed::local_insert in one): the one-pass front end with edge records (sk_front_kernel, `edges`), the back
// end as it is -- the node index with its occurrence counts --, then sk_edges_accumulate over the fine buckets' records, which are
// still in the workspace. *done = false: not this path's input (several runs per read, a base that is not A C G T, a shape without
// super-k-mers, anything the fast front end declines): the tuple path takes it.
template <int W>
static kmi_status dbg_build_superkmer_w(kmi_dbg *g, const uint8_t *bytes_dev, size_t n_bytes, bool *done) {
  kmi_ctx *ctx = g->ctx;
  kmi_index *idx = g->nodes;
  SkFront f;
  bool took = false;
  ctx->edge_records = true;
  kmi_status st = sk_front_fast<W>(ctx, &idx->cfg, idx->shape, bytes_dev, n_bytes, 0u, &f, &took, nullptr, 0, true);
  ctx->edge_records = false;
  if (st != KMI_OK || !took || !f.ok) return st;
  *done = true;
  if (f.n_kmers == 0) return KMI_OK;
  KMI_TRY((sk_back_end<W>(idx, f.recs, f.n_records, f.h_cnt, f.h_base, f.wg_off, f.n_kmers, 0u, false, true)));
  if (!ctx->sk_left.valid) return set_err(ctx, KMI_ERR_DEVICE, "the back end left no record of its fine buckets");
  KMI_TRY(ensure_dense(idx));
  const size_t eb = (size_t)(idx->n_entries ? idx->n_entries : 1) * 8 * sizeof(uint32_t);
  uint32_t *ne = nullptr;
  if (pool_alloc(ctx, (void **)&ne, eb) != hipSuccess) return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the edge counts");
  KMI_HIP(ctx, hipMemsetAsync(ne, 0, eb, ctx->stream));
  {
    ProfScope ps(ctx, "sk_edges_accumulate", f.n_kmers);
    hipLaunchKernelGGL(sk_edges_accumulate_kernel, dim3(kNumFine), dim3(DbgCfg<1>::NT), 0, ctx->stream, (const uint64_t *)idx->keys, (const uint64_t *)idx->bucket_off,
                       ctx->sk_left.recs, ctx->sk_left.rec_off, ctx->sk_left.region, ctx->sk_left.cap, ctx->sk_left.cnt, idx->shape.k, ne);
  }
  ctx->sk_left.valid = false;
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (g->edges) pool_free(ctx, g->edges, g->edges_bytes);
  g->edges = ne; g->edges_bytes = eb;
  return KMI_OK;
}
static kmi_status dbg_build_superkmer(kmi_dbg *g, const uint8_t *bytes_dev, size_t n_bytes, bool *done) {
  *done = false;
  kmi_ctx *ctx = g->ctx;
  kmi_index *idx = g->nodes;
  if (!ctx->dbg_superkmer || !ctx->fused_superkmer || !ctx->front_fused || g->cfg.seq_format != KMI_FMT_FASTQ || (idx->has_data && idx->n_entries) ||
      idx->shape.n_words != 1 || idx->shape.bits != 2 || n_bytes < 64)
    return KMI_OK;
  const uint32_t w = sk_window_of(idx->shape.k);
  if (!w) return KMI_OK;
  KMI_TRY(align_input(ctx, &bytes_dev, n_bytes));
  return w == 19u ? dbg_build_superkmer_w<19>(g, bytes_dev, n_bytes, done) : (w == 13u ? dbg_build_superkmer_w<13>(g, bytes_dev, n_bytes, done)
       : (w == 11u ? dbg_build_superkmer_w<11>(g, bytes_dev, n_bytes, done) : dbg_build_superkmer_w<7>(g, bytes_dev, n_bytes, done)));
}

// nodes.erase(keys) (the erase of the distributed map the node map derives from, distributed_unordered_map.hpp:719-779): the nodes of
// the query keys leave with their edge counts. The key array of the node index is compacted by the index's own erase; the
// counters of the nodes that stay are carried over to the new order the way an insert carries the old nodes' counters.
template <int NW, int BITS>
static kmi_status dbg_erase_impl(kmi_dbg *g, const uint64_t *q_dev, size_t nq, uint64_t *n_erased) {
  kmi_ctx *ctx = g->ctx;
  kmi_index *idx = g->nodes;
  if (n_erased) *n_erased = 0;
  const uint64_t n_old = idx->has_data ? idx->n_entries : 0;
  if (n_old == 0 || nq == 0) return KMI_OK;
  KMI_TRY(ensure_dense(idx));
  void *p;
  const size_t kb = (size_t)n_old * NW * sizeof(uint64_t);
  KMI_TRY(ws_get(ctx, WS_DBG_OLD, kb + kOffBytes + 64, &p));
  uint64_t *old_keys = (uint64_t *)p, *old_off = (uint64_t *)((uint8_t *)p + ((kb + 15) & ~(size_t)15));
  KMI_HIP(ctx, hipMemcpyAsync(old_keys, idx->keys, kb, hipMemcpyDeviceToDevice, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(old_off, idx->bucket_off, kOffBytes, hipMemcpyDeviceToDevice, ctx->stream));
  uint64_t gone = 0;
  KMI_TRY(index_query(idx, Q_ERASE, q_dev, nq, nullptr, nullptr, 0, &gone));
  if (n_erased) *n_erased = gone;
  if (gone == 0) return KMI_OK;
  const size_t eb = (size_t)(idx->n_entries ? idx->n_entries : 1) * 8 * sizeof(uint32_t);
  uint32_t *ne = nullptr;
  if (pool_alloc(ctx, (void **)&ne, eb) != hipSuccess) return set_err(ctx, KMI_ERR_NOMEM, "hipMalloc failed for the edge counts");
  KMI_HIP(ctx, hipMemsetAsync(ne, 0, eb, ctx->stream));
  if (idx->n_entries) {
    ProfScope ps(ctx, "dbg_accumulate", n_old);
    hipLaunchKernelGGL((dbg_accumulate_kernel<NW>), dim3(kNumFine), dim3(DbgCfg<NW>::NT), 0, ctx->stream, (const uint64_t *)idx->keys,
                       (const uint64_t *)idx->bucket_off, (const uint64_t *)nullptr, (const uint64_t *)nullptr, (const uint64_t *)old_keys,
                       (const uint32_t *)g->edges, (const uint64_t *)old_off, ne);
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (g->edges) pool_free(ctx, g->edges, g->edges_bytes);
  g->edges = ne; g->edges_bytes = eb;
  return KMI_OK;
}
static kmi_status dbg_erase(kmi_dbg *g, const uint64_t *q_dev, size_t nq, uint64_t *n_erased) { KMI_DISPATCH(g->shape, dbg_erase_impl, g, q_dev, nq, n_erased); }

// find(): keys of the hits + kDbgValueWords value words each
static kmi_status dbg_find(kmi_dbg *g, const uint64_t *q_dev, size_t nq, uint64_t *out_keys_dev, uint64_t *out_vals_dev, uint64_t *n_out) {
  kmi_ctx *ctx = g->ctx;
  *n_out = 0;
  if (nq == 0 || !g->nodes->has_data || g->nodes->n_entries == 0) return KMI_OK;
  void *dp;
  KMI_TRY(ws_get(ctx, WS_DBG_POS, (nq + 8) * sizeof(uint64_t), &dp));
  g->nodes->find_emits_index = true;
  const kmi_status st = index_query(g->nodes, Q_FIND, q_dev, nq, out_keys_dev, (uint64_t *)dp, nq, n_out);
  g->nodes->find_emits_index = false;
  KMI_TRY(st);
  if (*n_out) {
    ProfScope ps(ctx, "dbg_gather", *n_out);
    hipLaunchKernelGGL(dbg_gather_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const uint64_t *)dp, *n_out, (const uint32_t *)g->edges,
                       (const uint32_t *)g->nodes->vals, g->node_kind == KMI_DBG_EDGE_EXISTS, out_vals_dev);
    KMI_HIP(ctx, hipGetLastError());
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return KMI_OK;
}

}  // namespace kmi
