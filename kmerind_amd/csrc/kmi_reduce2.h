// kmi_reduce2.h -- sk_reduce, second form: WAVEFRONT-PRIVATE tables over the bins of a fine bucket (included by kmi_index.hip
// behind kmi_superkmer.h). Replaces the same reference code as sk_reduce_kernel: the local insert of the counting map,
// distributed_unordered_map.hpp:1603-1618 (find -> emplace or at() = r(at(), v)), and the bucket walk of incremental_mxx.hpp:273-364.
//
// sk_reduce_kernel (round 2/3) shares two LDS tables between the 16 wavefronts of a workgroup and meets at six barriers per
// bucket; half its wave cycles are spent waiting (profiles/r03_z_sq_counters.json). What makes a finer split free is that two
// occurrences of a canonical k-mer have the same minimizer, hence the same minimizer hash, hence the same value of ANY bits of
// that hash: records that differ in further hash bits (their BIN) never share a k-mer. So here
//   sort     the workgroup brings a fine bucket's records into LDS once, counting-sorted by bin (up to 12 more hash bits the
//            records carry beside their bucket bits): count, scan, place -- three short, evenly loaded steps;
//   batches  every wavefront then takes WINDOWS of the sorted records (all bins that start inside a window of 2^wsh records:
//            whole bins, so batches are independent) from a shared counter and works on its own, without a barrier:
//              dedupe   one record per lane; identical records of the step are found through a 256-slot byte table and the
//                       lane crossbar (a copy adds 1 to the weight of its representative, nothing else);
//              expand   the distinct records' k-mers, two per lane and step, exactly as sk_reduce_kernel's phase B does;
//              table    a PRIVATE k-mer table (home slot + neighbour in one read, claim in line, misses through a queue into an
//                       out-of-line probe walk) sized to the batch: no other wavefront ever touches it;
//              emit     the table is swept into the bucket's output range at a position taken from one LDS add per batch, and left empty.
// A batch whose k-mers do not fit the private table (or a bucket that does not fit the stage) puts its bucket on a REDO list;
// sk_reduce_kernel runs over that list afterwards (its pass splitting handles any bucket). The output contract is
// sk_reduce_kernel's: bucket b's distinct (k-mer, count) pairs at tmp_keys / tmp_vals[kmer_off[b] ...], their number in out_cnt[b].
#pragma once

namespace kmi {

#ifndef KMI_R2_WAVES
#define KMI_R2_WAVES 14     // wavefronts of a workgroup (one workgroup per CU: it takes the whole LDS; 16 leave a stage below the mean bucket of config 2)
#endif
#ifndef KMI_R2_CAP
#define KMI_R2_CAP 192      // home slots of a wavefront's k-mer table
#endif
#ifndef KMI_R2_LDS_KB
#define KMI_R2_LDS_KB 160   // LDS of a workgroup (80: two workgroups of 8 wavefronts per CU)
#endif
#ifndef KMI_R2_BINBITS
#define KMI_R2_BINBITS 12
#endif
#ifndef KMI_R2_RPT
#define KMI_R2_RPT 0        // records a thread holds between the loads and the stage (0: the stage's share)
#endif
constexpr int kR2BinBitsMax = KMI_R2_BINBITS;   // 3 sub-bucket bits + up to 9 further hash bits of the record
constexpr int kR2PerCu = KMI_R2_LDS_KB >= 160 ? 1 : 2;

template <int OWN_> struct R2Cfg {
  static constexpr int NW = KMI_R2_WAVES, NT = NW * kWave;
  static constexpr int CAP = KMI_R2_CAP, PAD = 64, SLOTS = CAP + PAD;
  static constexpr int EMIT_AT = CAP * 3 / 4;           // a table is swept before a batch that could take it past this many keys
  static constexpr int OWN = OWN_;                      // bytes: first-unit marks of a step of 64 records
  static constexpr int MQ = 2 * kWave, T1 = 256;
  static constexpr int LIST = SLOTS + kWave;            // claimed slots in claim order (+ one dump entry per lane)
  // per wavefront, overlaid by the sort's bin counters: miss queue (keys, byte weights), dedupe table, dedupe counts, claim list
  static constexpr int O_MW = MQ * 8, O_T1 = O_MW + MQ, O_TC = O_T1 + T1, O_LIST = O_TC + kWave * 4, A_WAVE = O_LIST + LIST * 2;
  static constexpr int NBINS = 1 << kR2BinBitsMax;
  static_assert(NW * A_WAVE >= (NBINS + 1) * 4, "the bin counters of the sort overlay the wavefronts' scratch");
  static_assert(OWN % 8 == 0 && A_WAVE % 8 == 0, "alignment of what follows");
  static constexpr int WAVE_BYTES = A_WAVE + OWN + SLOTS * 12;
  static constexpr int LDS = KMI_R2_LDS_KB * 1024;
  static constexpr int WIN_MIN = 16;                    // smallest window: 16 records
  static constexpr int FIXED = NW * WAVE_BYTES + (NW + 2) * 4 + 64 + 64;
  static constexpr int STAGE = ((LDS - FIXED) * 4 / 65) / 64 * 64;   // 16 bytes + a quarter of a window word per record
  static constexpr int WS = STAGE / WIN_MIN + 2 + kWave;
  static constexpr int RPT = KMI_R2_RPT ? KMI_R2_RPT : (STAGE + NT - 1) / NT;     // records a thread holds between the loads and the stage
  static constexpr int BPT = (NBINS + NT - 1) / NT;
  static constexpr int O_STAGE = 0, O_A = O_STAGE + STAGE * 16, O_OWN = O_A + NW * A_WAVE, O_TK = O_OWN + NW * OWN,
                       O_TV = O_TK + NW * SLOTS * 8, O_WS = O_TV + NW * SLOTS * 4, O_PART = O_WS + WS * 4, O_CTL = O_PART + (NW + 2) * 4,
                       TOTAL = O_CTL + 64;
  static_assert(TOTAL <= LDS, "LDS budget");
};

#ifdef KMI_R2_TIMING
__device__ unsigned long long g_r2_dbg[128];   // per wavefront number: [0..31] cycles in the batch phase until done, [32..63] batches taken, [64..95] steps, [96..127] expand iterations
#endif
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;

__device__ __forceinline__ uint32_t r2_hash(uint64_t key) { return sk_slot_hash(key); }
__device__ __forceinline__ uint32_t r2_slot(uint32_t h, uint32_t cap) { return __umul24(h >> 16, cap) >> 16; }   // (cap < 2^8)

// the slow path of a private table, out of line: queue entries [first, first + cnt) (key, weight), one per lane; the slots it claims
// go behind the n_claims entries of the claim list. Returns the slots claimed (bit 31: a walk found no room -- the batch does not fit)
__device__ __forceinline__ uint32_t r2_probe_insert(lds_u64_t *tk, lds_u32_t *tv, const lds_u64_t *q, const lds_u8_t *qw, lds_u16_t *list,
                                                         uint32_t n_claims, uint32_t first, uint32_t cnt, uint32_t cap, uint32_t last) {
  const uint32_t lane = lane_id();
  bool claimed = false, failed = false;
  uint32_t s = 0;
  if (lane < cnt) {
    const uint64_t key = q[first + lane];
    const uint32_t wt = qw[first + lane];
    const uint32_t s0 = r2_slot(r2_hash(key), cap);
    s = s0;
    uint64_t c = __atomic_load_n(&tk[s], __ATOMIC_RELAXED);
    for (;;) {
      while (c != key && c != kEmptyKey && s - s0 < 64u) { ++s; c = __atomic_load_n(&tk[s], __ATOMIC_RELAXED); }   // the walk (no wrap: padded table)
      if (c == key) break;
      if (c != kEmptyKey || s >= last) { failed = true; break; }
      uint64_t expected = kEmptyKey;
      if (__atomic_compare_exchange_n(&tk[s], &expected, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { claimed = true; break; }
      c = expected;   // lost the slot to another lane: its key is ours (the loop ends) or the walk goes on
    }
    if (!failed) __atomic_fetch_add(&tv[s], wt, __ATOMIC_RELAXED);
  }
  const unsigned long long cm = __ballot(claimed);
  if (claimed) list[n_claims + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u))] = (uint16_t)s;
  return (uint32_t)__popcll(cm) | (__ballot(failed) ? 0x80000000u : 0u);
}

template <bool CANON, int OWN_, bool SPECIAL>
__global__ __launch_bounds__(KMI_R2_WAVES * 64, KMI_R2_WAVES * kR2PerCu / 4) void sk_reduce2_kernel(const uint64_t *__restrict__ recs, const uint64_t *__restrict__ rec_off, uint32_t k,
                                                                       const uint64_t *__restrict__ kmer_off /* k-mers before every bucket */,
                                                                       uint64_t *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_vals,
                                                                       uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ flags,
                                                                       uint32_t *__restrict__ queue /* zero at launch */, uint32_t n_buckets,
                                                                       uint32_t xshift, uint32_t xbits /* the record's further hash bits: w1 >> xshift, xbits of them */,
                                                                       uint32_t win /* records of a window, >= 16 */,
                                                                       const uint64_t *__restrict__ fine_region, const uint32_t *__restrict__ fine_cap,
                                                                       const uint32_t *__restrict__ fine_cnt, uint32_t *__restrict__ redo_list,
                                                                       uint32_t *__restrict__ redo_cnt) {
  using C = R2Cfg<OWN_>;
  constexpr int NW = C::NW, NT = C::NT, RPT = C::RPT, BPT = C::BPT;
  constexpr uint32_t INF = 0xffffffffu;
  enum { C_NEXT = 0 /* and 1 */, C_GRAB = 2, C_EMIT = 3, C_REDO = 4, C_SPC = 5, C_SPS = 6 };
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[C::TOTAL];
  ulonglong2 *const s_stage = reinterpret_cast<ulonglong2 *>(s_raw + C::O_STAGE);
  uint32_t *const s_cnt = reinterpret_cast<uint32_t *>(s_raw + C::O_A);   // (the sort's bin counters / cursors: over the wavefronts' scratch)
  uint32_t *const s_ws = reinterpret_cast<uint32_t *>(s_raw + C::O_WS);
  uint32_t *const s_part = reinterpret_cast<uint32_t *>(s_raw + C::O_PART);
  uint32_t *const s_ctl = reinterpret_cast<uint32_t *>(s_raw + C::O_CTL);
  const uint32_t lane = lane_id(), wv = wave_id();
  unsigned char *const wa = s_raw + C::O_A + wv * C::A_WAVE;
  uint64_t *const mq = reinterpret_cast<uint64_t *>(wa);
  uint8_t *const mw = wa + C::O_MW;
  uint8_t *const t1 = wa + C::O_T1;
  uint32_t *const tcnt = reinterpret_cast<uint32_t *>(wa + C::O_TC);
  uint16_t *const clist = reinterpret_cast<uint16_t *>(wa + C::O_LIST);
  uint8_t *const wown = s_raw + C::O_OWN + wv * C::OWN;
  uint64_t *const tk = reinterpret_cast<uint64_t *>(s_raw + C::O_TK) + wv * C::SLOTS;
  uint32_t *const tv = reinterpret_cast<uint32_t *>(s_raw + C::O_TV) + wv * C::SLOTS;
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  const uint32_t kb = 2u * k;                                   // 34 .. 64
  const uint32_t kmask_hi = kb >= 64u ? 0xffffffffu : ((1u << (kb - 32u)) - 1u);
  const uint32_t pad = 64u - kb;                                // 0 .. 30
  const uint32_t B = 3u + xbits, nbins = 1u << B, xmask = (1u << xbits) - 1u;
  auto bin_of = [&](uint64_t w1) -> uint32_t {
    const uint32_t hi = (uint32_t)(w1 >> 32);
    return ((hi >> (kRecHashShift - 32)) & 7u) | (((hi >> (xshift - 32u)) & xmask) << 3);
  };
  // window of a sorted position: floor(off / win) up to rounding (any monotone map serves; off < 2^13, win >= 16)
  const uint32_t win_inv = (65536u + win - 1u) / win;
  auto win_of = [&](uint32_t off) -> uint32_t { return (off * win_inv) >> 16; };
#ifdef KMI_R2_TIMING
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = clock64();
#define R2_MARK(i) { const unsigned long long now_ = clock64(); acc[i] += now_ - tq; tq = now_; }
#else
#define R2_MARK(i)
#endif
  auto records_of = [&](uint32_t bb, uint64_t &lo, uint64_t &hi) {
    if (fine_cnt) {   // (a bucket that outgrew its room voids the build; what was written before it did stays within the room)
      const uint32_t cap = fine_cap[bb >> 7], cnt = fine_cnt[bb];
      lo = fine_region[bb >> 7] + (uint64_t)(bb & 127u) * cap; hi = lo + (cnt < cap ? cnt : cap);
    } else { lo = rec_off[bb]; hi = rec_off[bb + 1]; }
  };
  // ---- once per workgroup: empty tables, clear marks, the first bucket
  {
    uint64_t *const tk_all = reinterpret_cast<uint64_t *>(s_raw + C::O_TK);
    uint32_t *const tv_all = reinterpret_cast<uint32_t *>(s_raw + C::O_TV);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(NW * C::SLOTS); i += NT) { tk_all[i] = kEmptyKey; tv_all[i] = 0; }
    uint32_t *const own_all = reinterpret_cast<uint32_t *>(s_raw + C::O_OWN);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(NW * C::OWN / 4); i += NT) own_all[i] = 0;
    uint32_t *const a_all = reinterpret_cast<uint32_t *>(s_raw + C::O_A);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(NW * C::A_WAVE / 4); i += NT) a_all[i] = 0;   // (idle compare-and-swaps land in the miss queues: never the empty marker)
    if (threadIdx.x < 16) s_ctl[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_ctl[C_NEXT] = atomicAdd(queue, 1u);
  }
  lds_barrier();
  uint32_t b = __builtin_amdgcn_readfirstlane(s_ctl[C_NEXT]);
  uint32_t par = 1;
  const ulonglong2 *const recs2 = reinterpret_cast<const ulonglong2 *>(recs);
  ulonglong2 pf[RPT];
  auto load_chunk = [&](const ulonglong2 *src, uint32_t n_rec, uint32_t ch) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const uint32_t idx = ch * (uint32_t)(NT * RPT) + (uint32_t)j * NT + threadIdx.x;
      pf[j] = make_ulonglong2(0, 0);
      if (idx < n_rec) pf[j] = src[idx];
    }
  };
  uint64_t rb = 0, re = 0;
  if (b < n_buckets) { records_of(b, rb, re); load_chunk(recs2 + rb, (uint32_t)(re - rb), 0u); }
  // ---- the private table. It is never shared, so everything about it that is the same in all lanes -- keys waiting in the miss
  // queue, slots claimed since the last sweep -- lives in scalar registers; and the insert has no branch: a lane that has nothing to
  // claim sends its compare-and-swap to a word that never holds the empty marker, a lane without a hit adds zero.
  constexpr uint32_t CAPU = (uint32_t)C::CAP, LASTU = (uint32_t)C::SLOTS - 1u;
  uint32_t mn = 0, n_claims = 0;
  bool fail = false;
  lds_u64_t *const tk3 = (lds_u64_t *)tk;
  lds_u32_t *const tv3 = (lds_u32_t *)tv;
  uint64_t *const cas_dummy = mq + kWave + lane;   // (queue words hold keys or the sort's counters: never all ones)
  auto probe = [&](uint32_t first, uint32_t cnt) {
    const uint32_t r = __builtin_amdgcn_readfirstlane(r2_probe_insert(tk3, tv3, (const lds_u64_t *)mq, (const lds_u8_t *)mw, (lds_u16_t *)clist, n_claims,
                                                                      first, cnt, CAPU, LASTU));
    n_claims += r & 0x7fffffffu;
    fail = fail || (r >> 31);
  };
  auto mbcnt = [](unsigned long long m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
  // two k-mers per lane (ka if va, kc if vc; weight kw) into the table: home slot and neighbour of both in flight together
  auto insert2 = [&](uint64_t ka, bool va, uint64_t kc, bool vc, uint32_t kw) {
    if (SPECIAL) {   // (k = 32 only: a k-mer can equal the empty marker)
      if (va && ka == kEmptyKey) { s_ctl[C_SPS] = 1; atomicAdd(&s_ctl[C_SPC], kw); va = false; }
      if (vc && kc == kEmptyKey) { s_ctl[C_SPS] = 1; atomicAdd(&s_ctl[C_SPC], kw); vc = false; }
    }
    const uint32_t sa = r2_slot(r2_hash(ka), CAPU), sc = r2_slot(r2_hash(kc), CAPU);
    const uint64_t a0 = __atomic_load_n(&tk[sa], __ATOMIC_RELAXED), a1 = __atomic_load_n(&tk[sa + 1u], __ATOMIC_RELAXED);
    const uint64_t c0 = __atomic_load_n(&tk[sc], __ATOMIC_RELAXED), c1 = __atomic_load_n(&tk[sc + 1u], __ATOMIC_RELAXED);
    const bool ta = va && a0 == kEmptyKey, tc = vc && c0 == kEmptyKey;   // first sighting with a free home slot: claimed here
    const unsigned long long oa = atomicCAS((unsigned long long *)(ta ? &tk[sa] : cas_dummy), (unsigned long long)kEmptyKey, (unsigned long long)ka);
    const unsigned long long oc = atomicCAS((unsigned long long *)(tc ? &tk[sc] : cas_dummy), (unsigned long long)kEmptyKey, (unsigned long long)kc);
    const bool wa_ = ta && oa == kEmptyKey, wc_ = tc && oc == kEmptyKey;
    const bool ha0 = va && (a0 == ka || wa_ || (ta && oa == ka)), hc0 = vc && (c0 == kc || wc_ || (tc && oc == kc));
    const bool ha1 = va && !ha0 && a1 == ka, hc1 = vc && !hc0 && c1 == kc;
    atomicAdd(&tv[sa + (ha1 ? 1u : 0u)], (ha0 || ha1) ? kw : 0u);
    atomicAdd(&tv[sc + (hc1 ? 1u : 0u)], (hc0 || hc1) ? kw : 0u);
    // the slots claimed, in claim order (the sweep walks this list instead of the table)
    const unsigned long long wma = __ballot(wa_), wmc = __ballot(wc_);
    clist[wa_ ? n_claims + mbcnt(wma) : (uint32_t)C::SLOTS + lane] = (uint16_t)sa;
    n_claims += (uint32_t)__popcll(wma);
    clist[wc_ ? n_claims + mbcnt(wmc) : (uint32_t)C::SLOTS + lane] = (uint16_t)sc;
    n_claims += (uint32_t)__popcll(wmc);
    const bool ma = va && !ha0 && !ha1, mc = vc && !hc0 && !hc1;
    const unsigned long long mma = __ballot(ma), mmc = __ballot(mc);
    if (mma | mmc) {   // uniform
      if (ma) { const uint32_t pos = mn + mbcnt(mma); mq[pos] = ka; mw[pos] = (uint8_t)kw; }
      mn += (uint32_t)__popcll(mma);
      if (mn >= (uint32_t)kWave) { probe(mn - kWave, kWave); mn -= kWave; }
      if (mc) { const uint32_t pos = mn + mbcnt(mmc); mq[pos] = kc; mw[pos] = (uint8_t)kw; }
      mn += (uint32_t)__popcll(mmc);
      if (mn >= (uint32_t)kWave) { probe(mn - kWave, kWave); mn -= kWave; }
    }
  };
  // the k-mers of a step's records (w0, w1, weight wt; nu = units of two neighbouring k-mers, 0: none; inc = inclusive prefix sum of nu over
  // the lanes, total = its last) into the table: sk_reduce_kernel's dense expansion
  auto expand = [&](uint64_t w0, uint64_t w1, uint32_t wt, uint32_t nu, uint32_t inc, uint32_t total) {
    const uint32_t pre = inc - nu;
    if (nu) wown[pre] = (uint8_t)(lane + 1u);
    uint32_t carry = 0;   // record (+ 1) the previous step ended in
    const uint32_t top_sh = kb - 34u;   // where the last base of a k-mer starts in its high word (k >= 17)
#ifdef KMI_R2_TIMING
    if (lane == 0) atomicAdd(&g_r2_dbg[96 + wv], (unsigned long long)((total + 63u) / 64u));
#endif
    for (uint32_t g0 = 0; g0 < total; g0 += kWave) {
      const uint32_t g = g0 + lane;
      const bool act = g < total;
      uint32_t o = act ? (uint32_t)wown[g] : 0u;
      o = wave_inclusive_max_dpp(o);
      o = o > carry ? o : carry;
      carry = __builtin_amdgcn_readlane(o, kWave - 1);
      const int rl = (int)((o ? o - 1u : 0u) << 2);   // byte address of the lane that holds the record
      const uint32_t j = 2u * (g - (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)pre));
      const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)w0), a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)(w0 >> 32));
      const uint32_t a2 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)w1), a3 = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)(uint32_t)(w1 >> 32));
      const uint32_t kw = (uint32_t)__builtin_amdgcn_ds_bpermute(rl, (int)wt);
      const bool two = act && j + 1u <= ((a3 >> (kRecNShift - 32)) & 31u);   // k-mer j + 1 exists (the record holds n - 1)
      const bool w1sel = j >= 16u;
      const uint32_t bit = (2u * j) & 31u;   // (j is even: at most 28, so k-mer j + 1 starts in the same word)
      const uint32_t b0 = w1sel ? a1 : a0, b1 = w1sel ? a2 : a1, b2 = w1sel ? a3 : a2;
      const uint32_t rc_lo = __builtin_amdgcn_alignbit(b1, b0, bit);
      const uint32_t rc_hi = __builtin_amdgcn_alignbit(b2, b1, bit) & kmask_hi;   // (k >= 17: the low word is all k-mer)
      const uint32_t r_hi = __builtin_bitreverse32(rc_lo), r_lo = __builtin_bitreverse32(rc_hi);
      const uint32_t s_hi = ~(((r_hi >> 1) & 0x55555555u) | ((r_hi << 1) & 0xAAAAAAAAu));
      const uint32_t s_lo = ~(((r_lo >> 1) & 0x55555555u) | ((r_lo << 1) & 0xAAAAAAAAu));
      const uint32_t fw_lo = __builtin_amdgcn_alignbit(s_hi, s_lo, pad), fw_hi = s_hi >> pad;
      const uint32_t rc2_lo = __builtin_amdgcn_alignbit(b1, b0, bit + 2u);
      const uint32_t rc2_hi = __builtin_amdgcn_alignbit(b2, b1, bit + 2u) & kmask_hi;
      const uint32_t fw2_lo = (fw_lo << 2) | (((rc2_hi >> top_sh) & 3u) ^ 3u);
      const uint32_t fw2_hi = __builtin_amdgcn_alignbit(fw_hi, fw_lo, 30u) & kmask_hi;
      const uint64_t rc = (uint64_t)rc_lo | ((uint64_t)rc_hi << 32), fw = (uint64_t)fw_lo | ((uint64_t)fw_hi << 32);
      const uint64_t rc2 = (uint64_t)rc2_lo | ((uint64_t)rc2_hi << 32), fw2 = (uint64_t)fw2_lo | ((uint64_t)fw2_hi << 32);
      insert2(CANON ? (fw < rc ? fw : rc) : fw, act, CANON ? (fw2 < rc2 ? fw2 : rc2) : fw2, two, kw);
    }
    if (nu) wown[pre] = 0;   // the marks go back to zero for the next step
  };
  // the table's keys -> the bucket's output range, behind what it has emitted; the table is left empty. (Between batches only: a
  // batch is whole bins, so what has been swept cannot come again.)
  auto sweep = [&](uint64_t tmp0) {
    if (mn && !fail) probe(0u, mn);
    mn = 0;
    if (fail) {
      // a batch did not fit the private table: the bucket goes to the redo list; the table is emptied without being read
      for (uint32_t s = lane; s < (uint32_t)C::SLOTS; s += kWave) { tk[s] = kEmptyKey; tv[s] = 0; }
      if (lane == 0) { s_ctl[C_REDO] = 1; atomicAdd(&flags[45], 1u); }
      fail = false; n_claims = 0;
      wave_sync();
      return;
    }
    if (n_claims == 0u) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&s_ctl[C_EMIT], n_claims);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i0 = 0; i0 < n_claims; i0 += kWave) {
      const uint32_t i = i0 + lane;
      if (i < n_claims) {
        const uint32_t s = clist[i];
        const uint64_t key = tk[s];
        const uint32_t val = tv[s];
        tmp_keys[tmp0 + base + i] = key; tmp_vals[tmp0 + base + i] = val;
        tk[s] = kEmptyKey; tv[s] = 0;
      }
    }
    n_claims = 0;
    wave_sync();
  };
  // one batch: the sorted records [start, end) (whole bins) into the table
  auto batch = [&](uint32_t start, uint32_t end, uint64_t tmp0) {
    R2_MARK(2)
    for (uint32_t s0 = start; s0 < end; s0 += kWave) {
      const bool have = s0 + lane < end;
      ulonglong2 rec = make_ulonglong2(0, 0);
      if (have) rec = s_stage[s0 + lane];
      const uint32_t n = have ? ((uint32_t)(rec.y >> kRecNShift) & 31u) + 1u : 0u;
      // ---- identical records of the step: the lane that wrote a table slot last stands for everyone who reads it back and holds its record
      uint32_t hsh = ((uint32_t)rec.x ^ (uint32_t)(rec.x >> 32) * 0x85EBCA6Bu) ^ ((uint32_t)rec.y * 0xC2B2AE35u) ^ (uint32_t)(rec.y >> 32);
      hsh *= 0x9E3779B1u;
      const uint32_t ts = hsh >> 24;
      tcnt[lane] = 0;
      if (have) t1[ts] = (uint8_t)lane;
      wave_sync();
      const uint32_t w = have ? (uint32_t)t1[ts] : lane;
      const int wl = (int)(w << 2);
      const uint32_t x0 = (uint32_t)__builtin_amdgcn_ds_bpermute(wl, (int)(uint32_t)rec.x), x1 = (uint32_t)__builtin_amdgcn_ds_bpermute(wl, (int)(uint32_t)(rec.x >> 32));
      const uint32_t y0 = (uint32_t)__builtin_amdgcn_ds_bpermute(wl, (int)(uint32_t)rec.y), y1 = (uint32_t)__builtin_amdgcn_ds_bpermute(wl, (int)(uint32_t)(rec.y >> 32));
      const bool dup = have && w != lane && x0 == (uint32_t)rec.x && x1 == (uint32_t)(rec.x >> 32) && y0 == (uint32_t)rec.y && y1 == (uint32_t)(rec.y >> 32);
      if (dup) atomicAdd(&tcnt[w], 1u);
      wave_sync();
      const uint32_t wt = (have && !dup) ? 1u + tcnt[lane] : 0u;
      const uint32_t nu = wt ? (n + 1u) >> 1 : 0u;
      const uint32_t inc = wave_inclusive_sum_dpp(nu);
      const uint32_t total = __builtin_amdgcn_readlane(inc, kWave - 1);
      if (s0 == start) {
        // room for this batch? (a batch of one step brings at most two k-mers per unit; a longer one gets the whole table)
        const uint32_t need = 2u * total + (end - start > (uint32_t)kWave ? CAPU : 0u);
        if (n_claims && n_claims + need > (uint32_t)C::EMIT_AT) { R2_MARK(3) sweep(tmp0); R2_MARK(4) }
      } else if (n_claims > CAPU * 7u / 8u) fail = true;   // (more to come and the table is nearly full: the walks only get longer)
      if (fail) break;
      if (total) expand(rec.x, rec.y, wt, nu, inc, total);
    }
    R2_MARK(3)
    if (fail) { sweep(tmp0); R2_MARK(4) }
  };
  while (b < n_buckets) {   // uniform
    R2_MARK(7)
    uint32_t q_next = 0;
    if (threadIdx.x == 0) q_next = atomicAdd(queue, 1u);   // (stays in a register until the sort is done: nobody waits for it)
    const uint32_t n_rec = (uint32_t)(re - rb);
    const ulonglong2 *const src = recs2 + rb;
    const uint64_t tmp0 = kmer_off[b];
    const uint32_t n_chunks = (n_rec + (uint32_t)(NT * RPT) - 1u) / (uint32_t)(NT * RPT);
    // parts: a bucket larger than the stage goes through in 2 or 4 parts (the top bits of the bin), its records read again
    uint32_t pbits = 0;
    if (n_rec > (uint32_t)C::STAGE) { pbits = 1; while (pbits < 3u && (uint64_t)n_rec * 9u / 8u > ((uint64_t)C::STAGE << pbits)) ++pbits; }
    const bool give_up = pbits > 2u || pbits > B;
    if (give_up && threadIdx.x == 0) atomicAdd(&flags[44], 1u);
    if (pbits && threadIdx.x == 0) atomicAdd(&flags[47], 1u);
    if (threadIdx.x == 0) { s_ctl[C_EMIT] = 0; s_ctl[C_REDO] = give_up ? 1u : 0u; s_ctl[C_SPC] = 0; s_ctl[C_SPS] = 0; }
    bool published = false, prefetched = false;
    uint64_t nrb = 0, nre = 0;
    if (!give_up && n_rec) {
      const uint32_t n_parts = 1u << pbits;
      for (uint32_t part = 0; part < n_parts; ++part) {
        // ---- sort: count
        for (uint32_t i = threadIdx.x; i <= nbins; i += NT) s_cnt[i] = 0;
        for (uint32_t i = threadIdx.x; i < (uint32_t)C::WS; i += NT) s_ws[i] = INF;
        if (threadIdx.x == 0) s_ctl[C_GRAB] = 0;
        lds_barrier();
        for (uint32_t ch = 0; ch < n_chunks; ++ch) {
          if (n_chunks > 1u) load_chunk(src, n_rec, ch);
#pragma unroll
          for (int j = 0; j < RPT; ++j) {
            const uint32_t idx = ch * (uint32_t)(NT * RPT) + (uint32_t)j * NT + threadIdx.x;
            const uint32_t bin = bin_of(pf[j].y);
            atomicAdd(&s_cnt[(idx < n_rec && (bin >> (B - pbits)) == part) ? bin : nbins], 1u);   // (one counter past the bins takes what is not this part's)
          }
        }
        R2_MARK(0)
        lds_barrier();
        // ---- scan: bin offsets (the cursors of the placement) and the first bin start of every window
        uint32_t total;
        {
          uint32_t c[BPT], sum = 0;
#pragma unroll
          for (int i = 0; i < BPT; ++i) {
            const uint32_t bin = threadIdx.x * (uint32_t)BPT + (uint32_t)i;
            c[i] = bin < nbins ? s_cnt[bin] : 0u;
            sum += c[i];
          }
          const uint32_t inc = wave_inclusive_sum_dpp(sum);
          if (lane == kWave - 1) s_part[wv] = inc;
          lds_barrier();
          uint32_t pre = 0, all = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) { const uint32_t v = s_part[w]; pre += ((uint32_t)w < wv) ? v : 0u; all += v; }
          total = all;
          uint32_t off = pre + inc - sum;
#pragma unroll
          for (int i = 0; i < BPT; ++i) {
            const uint32_t bin = threadIdx.x * (uint32_t)BPT + (uint32_t)i;
            if (bin < nbins) {
              s_cnt[bin] = off;
              if (c[i] && off < (uint32_t)C::STAGE) atomicMin(&s_ws[win_of(off)], off);
              off += c[i];
            }
          }
        }
        if (total > (uint32_t)C::STAGE) {   // uniform: this part does not fit the stage after all
          if (threadIdx.x == 0) { s_ctl[C_REDO] = 1; atomicAdd(&flags[46], 1u); }
          lds_barrier();
          break;
        }
        lds_barrier();
        // ---- place
        for (uint32_t ch = 0; ch < n_chunks; ++ch) {
          if (n_chunks > 1u) load_chunk(src, n_rec, ch);
#pragma unroll
          for (int j = 0; j < RPT; ++j) {
            const uint32_t idx = ch * (uint32_t)(NT * RPT) + (uint32_t)j * NT + threadIdx.x;
            const uint32_t bin = bin_of(pf[j].y);
            if (idx < n_rec && (bin >> (B - pbits)) == part) s_stage[atomicAdd(&s_cnt[bin], 1u)] = pf[j];
          }
        }
        if (!published && threadIdx.x == 0) s_ctl[C_NEXT + par] = q_next;
        published = true;
        R2_MARK(1)
        lds_barrier();   // the stage is complete; the scratch under the counters is the wavefronts' again
        R2_MARK(5)
        // ---- the next bucket's records, in flight while this part's batches run (the last part: until then the registers hold this bucket's)
        if (part + 1u == n_parts) {
          const uint32_t nb = __builtin_amdgcn_readfirstlane(s_ctl[C_NEXT + par]);
          if (nb < n_buckets) { records_of(nb, nrb, nre); load_chunk(recs2 + nrb, (uint32_t)(nre - nrb), 0u); }
          prefetched = true;
        }
        // ---- batches: windows of the sorted records, handed out by a counter
        {
#ifdef KMI_R2_TIMING
          const unsigned long long t_b0 = clock64();
#endif
          const uint32_t nwin = total ? win_of(total - 1u) + 1u : 0u;
          uint32_t gi = 0;
          if (lane == 0) gi = atomicAdd(&s_ctl[C_GRAB], 1u);
          gi = __builtin_amdgcn_readfirstlane(gi);
          while (gi < nwin) {
            uint32_t gnext = 0;
            if (lane == 0) gnext = atomicAdd(&s_ctl[C_GRAB], 1u);   // (returns while this window is worked on)
            uint32_t v = (gi + lane < nwin) ? s_ws[gi + lane] : INF;   // this window's first bin start and the following windows'
            const uint32_t start = __builtin_amdgcn_readlane(v, 0);
            if (start != INF) {
              uint32_t end = total;
              unsigned long long m = __ballot(v != INF) & ~1ull;
              if (m) end = __builtin_amdgcn_readlane(v, __builtin_ctzll(m));
              else for (uint32_t j0 = gi + kWave; j0 < nwin; j0 += kWave) {   // (a bin of more than 63 windows)
                v = (j0 + lane < nwin) ? s_ws[j0 + lane] : INF;
                m = __ballot(v != INF);
                if (m) { end = __builtin_amdgcn_readlane(v, __builtin_ctzll(m)); break; }
              }
              batch(start, end, tmp0);
#ifdef KMI_R2_TIMING
              if (lane == 0) { atomicAdd(&g_r2_dbg[32 + wv], 1ull); atomicAdd(&g_r2_dbg[64 + wv], (unsigned long long)((end - start + 63u) / 64u)); }
#endif
            }
            gi = __builtin_amdgcn_readfirstlane(gnext);
          }
          sweep(tmp0);   // what this wavefront's table still holds
#ifdef KMI_R2_TIMING
          if (lane == 0) atomicAdd(&g_r2_dbg[wv], clock64() - t_b0);
#endif
        }
        R2_MARK(2)
        lds_barrier();   // every batch of this part is done: the stage and the scratch are free
        R2_MARK(6)
      }
    }
    if (!published && threadIdx.x == 0) s_ctl[C_NEXT + par] = q_next;
    lds_barrier();
    const uint32_t nb = __builtin_amdgcn_readfirstlane(s_ctl[C_NEXT + par]);
    if (!prefetched && nb < n_buckets) { records_of(nb, nrb, nre); load_chunk(recs2 + nrb, (uint32_t)(nre - nrb), 0u); }
    if (threadIdx.x == 0) {
      if (s_ctl[C_REDO]) redo_list[atomicAdd(redo_cnt, 1u)] = b;
      else {
        uint32_t e = s_ctl[C_EMIT];
        if (SPECIAL && s_ctl[C_SPS]) { tmp_keys[tmp0 + e] = kEmptyKey; tmp_vals[tmp0 + e] = s_ctl[C_SPC]; ++e; }
        out_cnt[b] = e;
      }
    }
    b = nb; rb = nrb; re = nre;
    par ^= 1u;
  }
#ifdef KMI_R2_TIMING
  if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&reinterpret_cast<unsigned long long *>(flags + 48)[i], acc[i]);
#endif
}

}  // namespace kmi
