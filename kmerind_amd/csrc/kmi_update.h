// update() of the counting maps with a device-side updater: distributed_densehash_map.hpp:1975-2003 ->
// densehash_map.hpp:663-714 -- for every input pair whose (transformed) key is stored, op(stored value, pair value); pairs of
// absent keys are skipped; the return value is the number of calls. The reference takes any functor; the updaters that have a
// device form here are the arithmetic ones a counting map is used with (KMI_UPDATE_*). Pairs with the same key are applied
// in input order by the reference; ADD / MAX / MIN do not depend on it, ASSIGN keeps the LAST pair of the input, as there.
//
// Included by kmi_index.hip after kmi_debruijn.h (same table geometry: DbgCfg).
#pragma once

namespace kmi {

// value word of every record := value | position in the input << 32 (ASSIGN needs the order; the partition does not keep it)
template <int NW>
__global__ __launch_bounds__(256) void update_tag_order_kernel(uint64_t *__restrict__ recs, uint64_t n) {
  constexpr int RW = NW + 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    recs[i * RW + NW] = (recs[i * RW + NW] & 0xffffffffull) | (i << 32);
}

// Fine bucket b: the bucket's entries in a table (key -> row), their counts in LDS, the bucket's update records streamed
// against it, the counts written back in place. Chunks of DbgCfg<NW>::ROWS entries, as in dbg_accumulate_kernel.
template <int NW>
__global__ __launch_bounds__((DbgCfg<NW>::NT)) void bucket_update_kernel(const uint64_t *__restrict__ idx_keys, const uint64_t *__restrict__ idx_off,
                                                                        uint32_t *__restrict__ idx_vals, const uint64_t *__restrict__ recs,
                                                                        const uint64_t *__restrict__ rec_off, int op,
                                                                        unsigned long long *__restrict__ n_updated) {
  using Cfg = DbgCfg<NW>;
  constexpr int RW = NW + 1;
  __shared__ uint64_t s_tk[Cfg::SLOTS * NW];
  __shared__ uint32_t s_tt[(NW == 1) ? 1 : Cfg::SLOTS];
  __shared__ uint16_t s_row[Cfg::SLOTS];
  __shared__ uint32_t s_val[Cfg::ROWS];
  __shared__ unsigned long long s_set[Cfg::ROWS];   // ASSIGN: (input position + 1) << 32 | value of the latest pair seen
  __shared__ uint32_t s_ctl[8];
  __shared__ unsigned long long s_hits;
  LdsTable<NW> tab;
  tab.keys = s_tk; tab.vals = nullptr; tab.tags = s_tt; tab.distinct = &s_ctl[0]; tab.overflow = &s_ctl[1];
  tab.special = &s_ctl[2]; tab.special_set = &s_ctl[3]; tab.progress = &s_ctl[5];
  tab.cap = Cfg::CAP; tab.slots = Cfg::SLOTS; tab.limit = 2u * Cfg::CAP;
  uint32_t *s_fail = &s_ctl[6], *s_special_row = &s_ctl[7];
  const uint32_t b = blockIdx.x;
  const uint64_t ib = idx_off[b], ie = idx_off[b + 1];
  const uint64_t rb = rec_off[b], re = rec_off[b + 1];
  if (ib == ie || rb == re) return;
  if (threadIdx.x == 0) s_hits = 0ull;
  uint32_t chunk = Cfg::ROWS;
  uint64_t i0 = ib;
  unsigned long long hits = 0;
  while (i0 < ie) {
    const uint32_t nc = (uint32_t)((ie - i0) < (uint64_t)chunk ? (ie - i0) : (uint64_t)chunk);
    for (uint32_t x = threadIdx.x; x < (uint32_t)Cfg::SLOTS; x += blockDim.x) { if (NW == 1) s_tk[x] = kEmptyKey; else s_tt[x] = kTagEmpty; }
    for (uint32_t x = threadIdx.x; x < nc; x += blockDim.x) { s_val[x] = idx_vals[i0 + x]; s_set[x] = 0ull; }
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
    if (*s_fail) {
      lds_barrier();
      chunk = chunk > 64u ? chunk / 2u : 32u;
      continue;
    }
    for (uint64_t i = rb + threadIdx.x; i < re; i += blockDim.x) {
      uint64_t k[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) k[w] = recs[i * RW + w];
      const int s = table_find<NW>(tab, k, place_hash<NW>(k));
      const uint32_t row = s >= 0 ? (uint32_t)s_row[s] : (s == -2 ? *s_special_row : ~0u);
      if (row == ~0u) continue;   // absent, or an entry of another chunk
      const uint64_t vw = recs[i * RW + NW];
      const uint32_t v = (uint32_t)vw;
      ++hits;
      if (op == KMI_UPDATE_ADD) atomicAdd(&s_val[row], v);
      else if (op == KMI_UPDATE_MAX) atomicMax(&s_val[row], v);
      else if (op == KMI_UPDATE_MIN) atomicMin(&s_val[row], v);
      else atomicMax(&s_set[row], (unsigned long long)((((vw >> 32) + 1ull) << 32) | v));
    }
    lds_barrier();
    for (uint32_t x = threadIdx.x; x < nc; x += blockDim.x) {
      if (op == KMI_UPDATE_ASSIGN) { if (s_set[x]) idx_vals[i0 + x] = (uint32_t)s_set[x]; }
      else idx_vals[i0 + x] = s_val[x];
    }
    lds_barrier();
    i0 += nc;
  }
  hits = wave_reduce_sum(hits);
  if (lane_id() == 0 && hits) atomicAdd(&s_hits, hits);
  lds_barrier();
  if (threadIdx.x == 0 && s_hits) atomicAdd(n_updated, s_hits);
}

template <int NW, int BITS>
static kmi_status update_pairs_impl(kmi_index *idx, uint64_t *recs_dev, size_t n, int op, uint64_t *n_updated) {
  kmi_ctx *ctx = idx->ctx;
  *n_updated = 0;
  if (n == 0 || !idx->has_data || idx->n_entries == 0) return KMI_OK;   // (this->empty(): nothing to update)
  KMI_TRY(ensure_dense(idx));
  if (op == KMI_UPDATE_ASSIGN) {
    if (n >> 32) return set_err(ctx, KMI_ERR_INVALID, "update(assign): at most 2^32 - 1 pairs per call (their order rides in the record)");
    hipLaunchKernelGGL((update_tag_order_kernel<NW>), dim3(2048), dim3(256), 0, ctx->stream, recs_dev, (uint64_t)n);
  }
  Partitioned part;
  KMI_TRY((partition_impl<NW, BITS, 1>(ctx, &idx->cfg, idx->shape, recs_dev, n, true, WS_KEYS_A, WS_KEYS_B, &part, idx->layout_w)));
  KMI_HIP(ctx, hipMemsetAsync(ctx->d_totals + 7, 0, sizeof(uint64_t), ctx->stream));
  {
    ProfScope ps(ctx, "bucket_update", n);
    hipLaunchKernelGGL((bucket_update_kernel<NW>), dim3(kNumFine), dim3(DbgCfg<NW>::NT), 0, ctx->stream, (const uint64_t *)idx->keys,
                       (const uint64_t *)idx->bucket_off, idx->vals, (const uint64_t *)part.keys, (const uint64_t *)part.fine_off, op,
                       (unsigned long long *)(ctx->d_totals + 7));
  }
  KMI_HIP(ctx, hipGetLastError());
  KMI_HIP(ctx, hipMemcpyAsync(n_updated, ctx->d_totals + 7, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}
static kmi_status index_update_pairs(kmi_index *idx, uint64_t *recs_dev, size_t n, int op, uint64_t *n_updated) {
  KMI_DISPATCH(idx->shape, update_pairs_impl, idx, recs_dev, n, op, n_updated);
}

}  // namespace kmi
