// kmi_tuples.h -- position / position + quality tuples partitioned STRAIGHT FROM THE PARSE (included by kmi_index.hip).
//
// What it replaces on the reference side: KmerPositionTupleParser / KmerPositionQualityTupleParser filling a vector (kmer_parser.hpp:
// 303-569, 577-900) + unordered_multimap::insert's distribute and local insert (distributed_unordered_map.hpp:1466-1515). Until
// round 3 the device did the same in the same order: the extract pass wrote the tuple arrays in file order (16-32 bytes per tuple),
// a histogram pass read the keys back, and the coarse scatter read everything back again -- one write and two reads of arrays
// that exist only to be partitioned (21 of 31 / 44 / 56 ms for 1.2e9 / 1.2e9 / 1e9 tuples went into those passes and the fine one).
// Here the two kernels that need the tuples make them from the packed input themselves, per scan tile:
//   tuple_hist     the tile's k-mer windows -> strand transform -> placement hash -> fine-bucket histogram in LDS (what hist_fine
//                  counts), and for position + quality the read descriptors the quality kernel starts from;
//   tuple_scatter  the same windows again, now with their values -- ShortSequenceKmerId from the record starts of the tile
//                  (sequence.hpp:127-209), LongSequenceKmerId of a FASTA character from the compaction's id array, the quality
//                  float from the quality kernel's dense array -- bucket-sorted per round in LDS and written once, as records, to
//                  the coarse buckets at per-workgroup cursors.
// The fine pass (scatter_fine) and everything behind it are unchanged; the tuple arrays in file order never exist.
#pragma once

#ifndef KMI_TUP_RT
#define KMI_TUP_RT(RW) 2048
#endif

namespace kmi {

// windows a scatter round takes: 32 / 48 / 64 KB of records of 2 / 3 / 4 words
template <int RW> struct TupCfg { static constexpr int RT = KMI_TUP_RT(RW); };   // (with the tile images 60 - 80 KB of LDS: two workgroups per CU)

// NTH (threads = bytes per tile / C): the histogram pass may take larger tiles than the scatter pass -- its per-tile work (the packed
// tile into LDS, the scans, the window list) is paid per tile whatever the tile holds, and three-word shapes have tiles of 2 KB
// (256 threads: their scatter pass needs the registers): with 512 threads the histogram of config 4 takes 6.2 instead of 11.0 ms.
// `per` = tiles of THIS kernel's size per workgroup: the caller gives both passes the same byte ranges (the scatter pass writes at
// cursors that come from this pass's per-workgroup counts).
template <int NW, int BITS, bool FASTA, bool READS, int NTH = ExCfg<NW, BITS>::NT>
__global__ __launch_bounds__(NTH) void tuple_hist_kernel(PackedInput in, uint64_t n_tiles, KShape shape, bool canonical,
                                                       const uint32_t *__restrict__ line_base, const uint64_t *__restrict__ tile_off,
                                                       ReadDesc *__restrict__ reads, uint32_t *__restrict__ fine_hist /* [kFineParts][kNumFine] */,
                                                       uint32_t *__restrict__ wg_hist /* [groups][kNumCoarse] */, uint64_t per) {
  using Cfg = ExCfgT<NW, BITS, ExCfg<NW, BITS>::C, NTH>;
  __shared__ uint32_t s_hist[kNumFine];
  __shared__ uint32_t s_eol[Cfg::EOL_DW];
  __shared__ uint32_t s_brk[Cfg::EOL_DW];
  __shared__ uint32_t s_stream[Cfg::STREAM_DW];
  __shared__ uint32_t s_scan[Cfg::NT / 64 + 2];
  __shared__ uint16_t s_pos[Cfg::TILE];
  __shared__ uint16_t s_lsmask[READS ? Cfg::NT : 1];   // line-start bits of every chunk
  __shared__ uint16_t s_lcnt[READS ? Cfg::NT : 1];     // line starts of the tile before every chunk
  for (int i = threadIdx.x; i < kNumFine; i += Cfg::NT) s_hist[i] = 0;
  const uint64_t tb = (uint64_t)blockIdx.x * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  for (uint64_t tile = tb; tile < te; ++tile) {
    lds_barrier();   // the previous tile's images are done with (and the histogram is clear)
    uint32_t eol, ls, lbl, ltot;
    tile_front_packed<Cfg>(in, tile, s_eol, s_stream, s_scan, eol, ls, lbl, ltot);
    const uint64_t tile0 = tile * Cfg::TILE;
    uint32_t total;
    if (FASTA) total = tile_window_list_from<Cfg>(chunk_valid_mask_fasta<Cfg>(s_eol, shape.k, tile0, in.n_bytes, in.n_valid), s_pos, s_scan);
    else {
      if (READS) { s_lsmask[threadIdx.x] = (uint16_t)ls; s_lcnt[threadIdx.x] = (uint16_t)lbl; }
      total = tile_window_list<Cfg>(tile_break_image<Cfg>(in.brk, in.n_cover, tile, s_eol, s_brk), ls, line_base[tile] + lbl, shape.k, s_pos, s_scan);
    }
    const uint64_t base = READS ? tile_off[tile] : 0ull;
    for (uint32_t q = threadIdx.x; q < total; q += Cfg::NT) {   // (eight consecutive windows per lane, rolled, were slower here: 7.3 against 6.7 ms)
      uint64_t rc[NW], fw[NW], key[NW];
      const uint32_t pos = s_pos[q];
      window_at<Cfg>(s_stream, pos, shape, rc, fw);
      select_strand<NW>(rc, fw, canonical, key);
      atomicAdd(&s_hist[fine15_of_key<NW>(key, 0u, shape.k)], 1u);
      if (READS) {   // the first window of its read: the descriptor slot is the read's sequence index (fastq_extract_kernel)
        const uint32_t j = pos / Cfg::C, p = pos % Cfg::C;
        if ((s_lsmask[j] >> p) & 1u) {
          const uint32_t line = line_base[tile] + s_lcnt[j] + (uint32_t)__builtin_popcount((uint32_t)s_lsmask[j] & ((1u << p) - 1u));
          ReadDesc rd; rd.seq_pos = tile0 + pos; rd.out_off = base + q;
          reads[(line - 1u) >> 2] = rd;
        }
      }
    }
  }
  lds_barrier();
  uint32_t *part_hist = fine_hist + (uint64_t)(blockIdx.x / (gridDim.x / kFineParts)) * kNumFine;
  for (int i = threadIdx.x; i < kNumFine; i += Cfg::NT) {
    const uint32_t v = s_hist[i];
    if (v) atomicAdd(&part_hist[i], v);
  }
  for (uint32_t c = threadIdx.x; c < (uint32_t)kNumCoarse; c += Cfg::NT) {
    uint32_t s = 0;
    for (int i = 0; i < kSubPerCoarse; ++i) s += s_hist[c * kSubPerCoarse + ((i + c) & (kSubPerCoarse - 1))];
    wg_hist[(uint64_t)blockIdx.x * kNumCoarse + c] = s;
  }
}

template <int NW, int BITS, int VW, bool FASTA>
__global__ __launch_bounds__((ExCfg<NW, BITS>::NT)) void tuple_scatter_kernel(PackedInput in, uint64_t n_tiles, KShape shape, bool canonical,
                                                                            const uint32_t *__restrict__ line_base, const uint64_t *__restrict__ hdr_base,
                                                                            const uint64_t *__restrict__ tile_off, uint64_t file_offset,
                                                                            const uint64_t *__restrict__ ids_by_rank, const float *__restrict__ in_q,
                                                                            const uint64_t *__restrict__ wg_off, uint64_t *__restrict__ out,
                                                                            uint32_t *__restrict__ flags, uint64_t per /* tiles per workgroup */) {
  using Cfg = ExCfg<NW, BITS>;
  constexpr int RW = NW + VW, RT = TupCfg<RW>::RT, NT = Cfg::NT, PT = RT / NT;
  static_assert(RT % NT == 0 && NT >= kNumCoarse, "round geometry");
  __shared__ uint64_t s_stage[RT * RW];
  __shared__ uint8_t s_bkt[RT];
  __shared__ uint32_t s_cnt[kNumCoarse];
  __shared__ uint32_t s_lofs[kNumCoarse];
  __shared__ uint64_t s_gbase[kNumCoarse];
  __shared__ uint32_t s_part[kNumCoarse / kWave];
  __shared__ uint32_t s_eol[Cfg::EOL_DW];
  __shared__ uint32_t s_brk[Cfg::EOL_DW];
  __shared__ uint32_t s_stream[Cfg::STREAM_DW];
  __shared__ uint32_t s_scan[Cfg::NT / 64 + 2];
  __shared__ uint16_t s_pos[Cfg::TILE];
  __shared__ uint16_t s_hmask[FASTA ? 1 : Cfg::NT];   // record-start bits of every chunk
  __shared__ uint16_t s_hexcl[FASTA ? 1 : Cfg::NT];   // 1 + tile position of the last record start in earlier chunks
  uint64_t cursor = (threadIdx.x < kNumCoarse) ? wg_off[(uint64_t)blockIdx.x * kNumCoarse + threadIdx.x] : 0ull;
  if (threadIdx.x < kNumCoarse) s_cnt[threadIdx.x] = 0;
  const uint64_t tb = (uint64_t)blockIdx.x * per;
  const uint64_t te = (tb + per < n_tiles) ? tb + per : n_tiles;
  for (uint64_t tile = tb; tile < te; ++tile) {
    lds_barrier();   // the previous tile's images and its last round's stage are done with
    uint32_t eol, ls, lbl, ltot;
    tile_front_packed<Cfg>(in, tile, s_eol, s_stream, s_scan, eol, ls, lbl, ltot);
    const uint64_t tile0 = tile * Cfg::TILE;
    uint32_t total;
    if (FASTA) total = tile_window_list_from<Cfg>(chunk_valid_mask_fasta<Cfg>(s_eol, shape.k, tile0, in.n_bytes, in.n_valid), s_pos, s_scan);
    else {
      const uint32_t lines_before = line_base[tile] + lbl;
      // record starts = line starts whose line index % 4 == 0 (the '@' line)
      uint32_t cur = lines_before, rest = ls, hm = 0, hlast = 0;
      while (rest) {
        const uint32_t q = (uint32_t)__builtin_ctz(rest);
        if ((cur & 3u) == 0u) { hm |= 1u << q; hlast = threadIdx.x * Cfg::C + q + 1u; }
        cur += 1; rest &= rest - 1u;
      }
      s_hmask[threadIdx.x] = (uint16_t)hm;
      s_hexcl[threadIdx.x] = (uint16_t)block_exclusive_max<uint32_t>(hlast, s_scan, (uint32_t *)nullptr);
      total = tile_window_list<Cfg>(tile_break_image<Cfg>(in.brk, in.n_cover, tile, s_eol, s_brk), ls, lines_before, shape.k, s_pos, s_scan);
    }
    const uint64_t base = tile_off ? tile_off[tile] : 0ull;
    for (uint32_t r0 = 0; r0 < total; r0 += RT) {
      const uint32_t nt = total - r0 < (uint32_t)RT ? total - r0 : (uint32_t)RT;
      uint64_t k[PT][NW], v[PT][VW];
      uint32_t bk[PT], rk[PT];
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        // (lane l takes entries l, l + NT, ... of the round: consecutive lanes read consecutive quality values and ids; PT consecutive
        // entries per lane with rolled windows made this kernel slower -- 22 against 15 ms for position + quality)
        const uint32_t li = (uint32_t)j * NT + threadIdx.x;
        bk[j] = 0xffffffffu;
        if (li < nt) {
          const uint32_t q = r0 + li, pos = s_pos[q];
          uint64_t rc[NW], fw[NW];
          window_at<Cfg>(s_stream, pos, shape, rc, fw);
          select_strand<NW>(rc, fw, canonical, k[j]);
          if (FASTA) v[j][0] = ids_by_rank[tile0 + pos];   // LongSequenceKmerId of the window's first character
          else {
            // ShortSequenceKmerId (sequence.hpp:156-157): record file offset << 16 | offset of the k-mer's first base from the
            // record start (kmer_parser.hpp:378-386)
            const uint32_t cj = pos / Cfg::C, p = pos % Cfg::C;
            const uint32_t m = (uint32_t)s_hmask[cj] & ((2u << p) - 1u);
            uint64_t rec;   // 1 + byte position (relative to the buffer) of the record start
            if (m) rec = tile0 + cj * Cfg::C + (31u - (uint32_t)__builtin_clz(m)) + 1u;
            else if (s_hexcl[cj]) rec = tile0 + s_hexcl[cj];
            else rec = hdr_base[tile];
            const uint64_t rec_off = file_offset + rec - 1u, d = tile0 + pos - (rec - 1u);
            if (rec == 0 || d > 0xFFFFu) atomicOr(&flags[3], 1u);   // ShortSequenceKmerId increment overflow (sequence.hpp:177-183)
            v[j][0] = ((rec_off & 0xFFFFFFFFFFull) << 16) | (d & 0xFFFFull);
          }
          if (VW > 1) v[j][VW - 1] = in_q ? (uint64_t)__float_as_uint(in_q[base + q]) : 0ull;
          bk[j] = fine15_of_key<NW>(k[j], 0u, shape.k) >> (kFineBits - kCoarseBits);
          rk[j] = atomicAdd(&s_cnt[bk[j]], 1u);
        }
      }
      lds_barrier();
      uint32_t c = 0, inc = 0;
      if (threadIdx.x < kNumCoarse) {              // waves 0..3, whole waves
        c = s_cnt[threadIdx.x];
        s_cnt[threadIdx.x] = 0;
        inc = wave_inclusive_scan(c);
        if (lane_id() == kWave - 1) s_part[wave_id()] = inc;
      }
      lds_barrier();
      if (threadIdx.x < kNumCoarse) {
        uint32_t pre = 0;
#pragma unroll
        for (uint32_t w = 0; w < kNumCoarse / kWave; ++w) pre += (w < wave_id()) ? s_part[w] : 0u;
        const uint32_t lo = pre + inc - c;
        s_lofs[threadIdx.x] = lo;
        s_gbase[threadIdx.x] = cursor - lo;
        cursor += c;
      }
      lds_barrier();
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        if (bk[j] != 0xffffffffu) {
          const uint32_t pos = s_lofs[bk[j]] + rk[j];
#pragma unroll
          for (int w = 0; w < NW; ++w) s_stage[(uint64_t)pos * RW + w] = k[j][w];
#pragma unroll
          for (int w = 0; w < VW; ++w) s_stage[(uint64_t)pos * RW + NW + w] = v[j][w];
          s_bkt[pos] = (uint8_t)bk[j];
        }
      }
      lds_barrier();
      for (uint32_t s = threadIdx.x; s < nt; s += NT) {
        const uint64_t dst = s_gbase[s_bkt[s]] + s;
#pragma unroll
        for (int w = 0; w < RW; ++w) out[dst * RW + w] = s_stage[(uint64_t)s * RW + w];
      }
      // (no barrier here: the next round's first step only touches s_cnt, reset above; its stage writes are three barriers away)
    }
  }
}

}  // namespace kmi
