// kmi_comm.hip -- the exchange step of the distributed maps over RCCL (xGMI inside one node).
//
// Reference being replaced: mxx::all2all of the bucket counts and mxx::all2allv of the payload in imxx::distribute
// (src/io/incremental_mxx.hpp:1087, 1098), as called by every collective of the distributed maps
// (distributed_unordered_map.hpp:1714-1721 insert, :918-955 count, :601-640 find, :742-752 erase), and the allreduce
// behind MapType::size() (distributed_map_base.hpp:227-245). One process per GPU; the communicator is built from an
// ncclUniqueId that rank 0 makes and the application hands to the other ranks (MPI_Bcast in the reference's world,
// torch.distributed in bench.py / the tests).
//
// RCCL is loaded at run time (dlopen), so libkmerind_hip.so has no link-time dependency on it and a process that never
// creates a communicator never touches it. The all-to-all is grouped ncclSend / ncclRecv over device buffers with 64-bit
// counts; a peer message travels in pieces below 1 GiB (the RCCL build of this image was seen to corrupt larger ones,
// tools/a2a_debug.py), and because that limit is a property of one library build the first exchange of every communicator
// carries per-message checksums that are verified on arrival.
#include <dlfcn.h>
#include <stdlib.h>

#include <vector>

#include "kmi_block.h"
#include "kmi_internal.h"

namespace {

typedef int ncclResult_t_;            // ncclSuccess == 0
typedef struct { char internal[128]; } ncclUniqueId_;
typedef void *ncclComm_t_;
enum { kNcclUint8 = 1, kNcclUint64 = 5, kNcclSum = 0 };   // rccl.h: ncclDataType_t / ncclRedOp_t

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t_ (*GetUniqueId)(ncclUniqueId_ *) = nullptr;
  ncclResult_t_ (*CommInitRank)(ncclComm_t_ *, int, ncclUniqueId_, int) = nullptr;
  ncclResult_t_ (*CommDestroy)(ncclComm_t_) = nullptr;
  ncclResult_t_ (*Send)(const void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  ncclResult_t_ (*Recv)(void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  ncclResult_t_ (*GroupStart)() = nullptr;
  ncclResult_t_ (*GroupEnd)() = nullptr;
  ncclResult_t_ (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t_) = nullptr;
  bool ok = false;
};

RcclApi &rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  const char *names[] = {getenv("KMI_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!n) continue;
    api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) break;
  }
  if (!api.lib) return api;
#define KMI_SYM(field, name) *(void **)(&api.field) = dlsym(api.lib, name)
  KMI_SYM(GetUniqueId, "ncclGetUniqueId"); KMI_SYM(CommInitRank, "ncclCommInitRank"); KMI_SYM(CommDestroy, "ncclCommDestroy");
  KMI_SYM(Send, "ncclSend"); KMI_SYM(Recv, "ncclRecv"); KMI_SYM(GroupStart, "ncclGroupStart"); KMI_SYM(GroupEnd, "ncclGroupEnd");
  KMI_SYM(AllReduce, "ncclAllReduce"); KMI_SYM(GetErrorString, "ncclGetErrorString");
#undef KMI_SYM
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Send && api.Recv && api.GroupStart && api.GroupEnd && api.AllReduce;
  return api;
}

// wrap-around 64-bit sums of the peer messages (message r = words [off[r], off[r + 1]) of buf)
__global__ __launch_bounds__(256) void message_sums_kernel(const uint64_t *__restrict__ buf, const uint64_t *__restrict__ off, uint64_t *__restrict__ sums) {
  __shared__ uint64_t s_part[256 / 64];
  const uint32_t r = blockIdx.x;
  uint64_t acc = 0;
  for (uint64_t i = off[r] + threadIdx.x; i < off[r + 1]; i += blockDim.x) acc += buf[i];
  acc = kmi::wave_reduce_sum(acc);
  if (kmi::lane_id() == 0) s_part[kmi::wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sums[r] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

}  // namespace

struct kmi_comm {
  kmi_ctx *ctx = nullptr;
  ncclComm_t_ nccl = nullptr;
  bool has_transport = false;     // kmi_comm_create_transport: the application's messenger over host buffers instead of RCCL
  kmi_transport transport{};
  char *h_stage[2] = {nullptr, nullptr};   // pinned staging of a transport exchange (send, receive), grown on demand
  size_t h_stage_cap[2] = {0, 0};
  int rank = 0, nranks = 1;
  bool verified = false;          // the first payload exchange carries checksums
  uint64_t verified_bytes = 0;    // the largest peer message an exchange with checksums has carried so far
  hipStream_t xstream = nullptr;  // payload exchanges that overlap the context's stream (comm_all_to_all_v_async)
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  uint64_t *d_small = nullptr;    // [8 * nranks + 8] device scratch: counts, sums, offsets
  uint64_t *h_small = nullptr;    // pinned mirror
};

#define KMI_NCCL(c, call)                                                                                   \
  do {                                                                                                      \
    ncclResult_t_ r__ = (call);                                                                             \
    if (r__ != 0) return kmi::set_err((c)->ctx, KMI_ERR_DEVICE, "%s failed: %s", #call,                    \
                                      rccl().GetErrorString ? rccl().GetErrorString(r__) : "rccl error"); \
  } while (0)

namespace kmi {

// ---- the transport backend: every collective below has a branch that goes through the application's callbacks over host memory
static kmi_status tp_fail(kmi_comm *c, const char *what) { return set_err(c->ctx, KMI_ERR_DEVICE, "the transport's %s callback reported an error", what); }
static kmi_status tp_stage(kmi_comm *c, int which, size_t bytes) {
  if (c->h_stage_cap[which] >= bytes && c->h_stage[which]) return KMI_OK;
  if (c->h_stage[which]) (void)hipHostFree(c->h_stage[which]);
  c->h_stage[which] = nullptr; c->h_stage_cap[which] = 0;
  const size_t cap = bytes + bytes / 4 + 4096;
  if (hipHostMalloc((void **)&c->h_stage[which], cap) != hipSuccess) return set_err(c->ctx, KMI_ERR_NOMEM, "pinned staging buffer of a transport exchange%s", "");
  c->h_stage_cap[which] = cap;
  return KMI_OK;
}
// host words to every peer and back (counts, riders, checksums): `per` words per peer
static kmi_status tp_words(kmi_comm *c, const uint64_t *send, uint64_t *recv, size_t per) {
  std::vector<uint64_t> b((size_t)c->nranks, per * sizeof(uint64_t));
  if (c->transport.all_to_all_v(c->transport.user, send, b.data(), recv, b.data()) != 0) return tp_fail(c, "all_to_all_v");
  return KMI_OK;
}
// device payload: everything the context's stream has queued is done first (the send buffer's producer), then D2H, the
// callback, H2D -- synchronous: the caller's buffers are free, and the received bytes in place, when this returns
static kmi_status tp_bytes(kmi_comm *c, const char *send_dev, const uint64_t *sbytes, char *recv_dev, const uint64_t *rbytes) {
  kmi_ctx *ctx = c->ctx;
  uint64_t ts = 0, tr = 0;
  for (int r = 0; r < c->nranks; ++r) { ts += sbytes[r]; tr += rbytes[r]; }
  KMI_TRY(tp_stage(c, 0, (size_t)ts));
  KMI_TRY(tp_stage(c, 1, (size_t)tr));
  if (ts) KMI_HIP(ctx, hipMemcpyAsync(c->h_stage[0], send_dev, (size_t)ts, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (c->transport.all_to_all_v(c->transport.user, c->h_stage[0], sbytes, c->h_stage[1], rbytes) != 0) return tp_fail(c, "all_to_all_v");
  if (tr) KMI_HIP(ctx, hipMemcpyAsync(recv_dev, c->h_stage[1], (size_t)tr, hipMemcpyHostToDevice, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

// mxx::all2all of one 64-bit count per peer (incremental_mxx.hpp:1087)
kmi_status comm_all_to_all_counts(kmi_comm *c, const uint64_t *send_counts, uint64_t *recv_counts) {
  kmi_ctx *ctx = c->ctx;
  const int p = c->nranks;
  if (p == 1 && !ctx->force_dist) { recv_counts[0] = send_counts[0]; return KMI_OK; }
  if (c->has_transport) return tp_words(c, send_counts, recv_counts, 1);
  uint64_t *d_s = c->d_small, *d_r = c->d_small + p;
  memcpy(c->h_small, send_counts, sizeof(uint64_t) * p);
  KMI_HIP(ctx, hipMemcpyAsync(d_s, c->h_small, sizeof(uint64_t) * p, hipMemcpyHostToDevice, ctx->stream));
  KMI_NCCL(c, rccl().GroupStart());
  for (int r = 0; r < p; ++r) {
    KMI_NCCL(c, rccl().Send(d_s + r, sizeof(uint64_t), kNcclUint8, r, c->nccl, ctx->stream));
    KMI_NCCL(c, rccl().Recv(d_r + r, sizeof(uint64_t), kNcclUint8, r, c->nccl, ctx->stream));
  }
  KMI_NCCL(c, rccl().GroupEnd());
  KMI_HIP(ctx, hipMemcpyAsync(c->h_small + p, d_r, sizeof(uint64_t) * p, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(recv_counts, c->h_small + p, sizeof(uint64_t) * p);
  return KMI_OK;
}

static kmi_status a2a_bytes(kmi_comm *c, const char *send, const uint64_t *sbytes, char *recv, const uint64_t *rbytes,
                            hipStream_t on = nullptr, uint64_t known_gmax = ~0ull, uint64_t *gmax_out = nullptr) {
  // on: the stream the transfers are queued on (default: the context's); known_gmax: the largest peer message anywhere when
  // the caller has it already (it rode on the count exchange) -- no all-reduce, no host synchronisation in here then
  kmi_ctx *ctx = c->ctx;
  const int p = c->nranks;
  hipStream_t st = on ? on : ctx->stream;
  // per peer and transfer (KMI_COMM_PIECE: a test knob that makes small messages travel in pieces)
  const uint64_t kPiece = [] { const char *e = getenv("KMI_COMM_PIECE"); const uint64_t v = e ? strtoull(e, nullptr, 10) : 0; return v >= 64 ? v : (1ull << 30) - 4096; }();
  uint64_t biggest = 0;
  for (int r = 0; r < p; ++r) { biggest = std::max(biggest, sbytes[r]); biggest = std::max(biggest, rbytes[r]); }
  // the number of pieces must agree on every rank: it follows from the largest message anywhere
  uint64_t gmax = known_gmax != ~0ull ? std::max(known_gmax, biggest) : biggest;
  if (c->has_transport) {
    // (no pieces: the message limit is RCCL's; the largest message is still agreed on, for the callers that verify by size)
    if (p > 1 && known_gmax == ~0ull && c->transport.allreduce_u64(c->transport.user, &gmax, 1, 1) != 0) return tp_fail(c, "allreduce_u64");
    if (gmax_out) *gmax_out = gmax;
    return tp_bytes(c, send, sbytes, recv, rbytes);
  }
  if (p > 1 && known_gmax == ~0ull) {
    c->h_small[4 * p] = biggest;
    KMI_HIP(ctx, hipMemcpyAsync(c->d_small + 4 * p, c->h_small + 4 * p, sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    KMI_NCCL(c, rccl().AllReduce(c->d_small + 4 * p, c->d_small + 4 * p + 1, 1, kNcclUint64, 2 /* ncclMax */, c->nccl, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(c->h_small + 4 * p + 1, c->d_small + 4 * p + 1, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    gmax = c->h_small[4 * p + 1];
  }
  if (gmax_out) *gmax_out = gmax;
  const uint64_t pieces = gmax ? (gmax + kPiece - 1) / kPiece : 1;
  std::vector<uint64_t> soff(p + 1, 0), roff(p + 1, 0);
  for (int r = 0; r < p; ++r) { soff[r + 1] = soff[r] + sbytes[r]; roff[r + 1] = roff[r] + rbytes[r]; }
  for (uint64_t q = 0; q < pieces; ++q) {
    KMI_NCCL(c, rccl().GroupStart());
    for (int r = 0; r < p; ++r) {
      const uint64_t slo = sbytes[r] * q / pieces, shi = sbytes[r] * (q + 1) / pieces;
      const uint64_t rlo = rbytes[r] * q / pieces, rhi = rbytes[r] * (q + 1) / pieces;
      if (shi > slo) KMI_NCCL(c, rccl().Send(send + soff[r] + slo, shi - slo, kNcclUint8, r, c->nccl, st));
      if (rhi > rlo) KMI_NCCL(c, rccl().Recv(recv + roff[r] + rlo, rhi - rlo, kNcclUint8, r, c->nccl, st));
    }
    KMI_NCCL(c, rccl().GroupEnd());
  }
  return KMI_OK;
}

// mxx::all2allv of the payload (incremental_mxx.hpp:1098): device buffers, element counts per peer; the receive buffer is
// the concatenation by source rank ascending. elem_bytes is a multiple of 8 (k-mer words / records).
kmi_status comm_all_to_all_v(kmi_comm *c, const void *send_dev, const uint64_t *send_counts, void *recv_dev, const uint64_t *recv_counts,
                             size_t elem_bytes) {
  kmi_ctx *ctx = c->ctx;
  const int p = c->nranks;
  std::vector<uint64_t> sb(p), rb(p);
  for (int r = 0; r < p; ++r) { sb[r] = send_counts[r] * elem_bytes; rb[r] = recv_counts[r] * elem_bytes; }
  uint64_t gmax = 0;
  KMI_TRY(a2a_bytes(c, (const char *)send_dev, sb.data(), (char *)recv_dev, rb.data(), nullptr, ~0ull, &gmax));
  if ((!c->verified || gmax > c->verified_bytes) && elem_bytes % 8 == 0) {
    // the first exchange of this communicator, and every later one whose largest peer message is larger than any that was
    // checked before (the 1 GiB behaviour is a property of message size): the sender's sum of every message travels behind it
    // and is compared on arrival (gmax is the same on every rank, so all ranks verify or none)
    c->verified = true;
    c->verified_bytes = std::max(c->verified_bytes, gmax);
    uint64_t *h = c->h_small + 5 * p + 2;           // [p + 1] send offsets (words), [p + 1] recv offsets
    h[0] = 0;
    for (int r = 0; r < p; ++r) h[r + 1] = h[r] + sb[r] / 8;
    uint64_t *h2 = h + p + 1;
    h2[0] = 0;
    for (int r = 0; r < p; ++r) h2[r + 1] = h2[r] + rb[r] / 8;
    uint64_t *d_off = c->d_small + 5 * p + 2, *d_sums = c->d_small + 2 * p;   // sums: [p] mine, [p] of what arrived
    KMI_HIP(ctx, hipMemcpyAsync(d_off, h, sizeof(uint64_t) * 2 * (p + 1), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(message_sums_kernel, dim3(p), dim3(256), 0, ctx->stream, (const uint64_t *)send_dev, (const uint64_t *)d_off, d_sums);
    hipLaunchKernelGGL(message_sums_kernel, dim3(p), dim3(256), 0, ctx->stream, (const uint64_t *)recv_dev, (const uint64_t *)(d_off + p + 1), d_sums + p);
    KMI_HIP(ctx, hipGetLastError());
    std::vector<uint64_t> eight(p, 8), theirs(p);
    // the senders' sums, one 8-byte message per peer
    uint64_t *d_theirs = c->d_small;                // (the count scratch is free here)
    KMI_TRY(a2a_bytes(c, (const char *)d_sums, eight.data(), (char *)d_theirs, eight.data()));
    std::vector<uint64_t> got(2 * p);
    KMI_HIP(ctx, hipMemcpyAsync(got.data(), d_sums + p, sizeof(uint64_t) * p, hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipMemcpyAsync(got.data() + p, d_theirs, sizeof(uint64_t) * p, hipMemcpyDeviceToHost, ctx->stream));
    KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < p; ++r)
      if (got[r] != got[p + r]) return set_err(ctx, KMI_ERR_DEVICE, "all-to-all payload corrupted: a peer message does not match its sender's checksum%s", "");
  }
  return KMI_OK;
}

// counts with a rider: every peer also receives the sender's largest message (bytes), so all ranks know the largest message of
// the coming payload exchange without another collective; a sender that has nothing valid to send says so with count = ~0
kmi_status comm_all_to_all_counts2(kmi_comm *c, const uint64_t *send_counts, uint64_t my_largest_bytes, uint64_t *recv_counts, uint64_t *largest_bytes) {
  kmi_ctx *ctx = c->ctx;
  const int p = c->nranks;
  uint64_t *d_s = c->d_small, *d_r = c->d_small + 2 * p;
  for (int r = 0; r < p; ++r) { c->h_small[2 * r] = send_counts[r]; c->h_small[2 * r + 1] = my_largest_bytes; }
  if (c->has_transport) {
    KMI_TRY(tp_words(c, c->h_small, c->h_small + 2 * p, 2));
    uint64_t big = 0;
    for (int r = 0; r < p; ++r) { recv_counts[r] = c->h_small[2 * p + 2 * r]; big = std::max(big, c->h_small[2 * p + 2 * r + 1]); }
    *largest_bytes = big;
    return KMI_OK;
  }
  KMI_HIP(ctx, hipMemcpyAsync(d_s, c->h_small, sizeof(uint64_t) * 2 * p, hipMemcpyHostToDevice, ctx->stream));
  KMI_NCCL(c, rccl().GroupStart());
  for (int r = 0; r < p; ++r) {
    KMI_NCCL(c, rccl().Send(d_s + 2 * r, 2 * sizeof(uint64_t), kNcclUint8, r, c->nccl, ctx->stream));
    KMI_NCCL(c, rccl().Recv(d_r + 2 * r, 2 * sizeof(uint64_t), kNcclUint8, r, c->nccl, ctx->stream));
  }
  KMI_NCCL(c, rccl().GroupEnd());
  KMI_HIP(ctx, hipMemcpyAsync(c->h_small + 2 * p, d_r, sizeof(uint64_t) * 2 * p, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t big = 0;
  for (int r = 0; r < p; ++r) { recv_counts[r] = c->h_small[2 * p + 2 * r]; big = std::max(big, c->h_small[2 * p + 2 * r + 1]); }
  *largest_bytes = big;
  return KMI_OK;
}

// every rank's `n` words to every rank (all[r * n ..] = rank r's): block summaries, verdicts -- a few words, host to host
kmi_status comm_allgather_words(kmi_comm *c, const uint64_t *mine, size_t n, uint64_t *all) {
  kmi_ctx *ctx = c->ctx;
  const int p = c->nranks;
  if (p == 1 && !ctx->force_dist) { memcpy(all, mine, sizeof(uint64_t) * n); return KMI_OK; }
  std::vector<uint64_t> send((size_t)p * n);
  for (int r = 0; r < p; ++r) memcpy(&send[(size_t)r * n], mine, sizeof(uint64_t) * n);
  if (c->has_transport) return tp_words(c, send.data(), all, n);
  void *d = nullptr;
  const size_t bytes = sizeof(uint64_t) * (size_t)p * n;
  KMI_TRY(ws_get(ctx, WS_MISC, 2 * bytes + 64, &d));
  uint64_t *d_s = (uint64_t *)d, *d_r = d_s + (size_t)p * n;
  KMI_HIP(ctx, hipMemcpyAsync(d_s, send.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
  KMI_NCCL(c, rccl().GroupStart());
  for (int r = 0; r < p; ++r) {
    KMI_NCCL(c, rccl().Send(d_s + (size_t)r * n, n * sizeof(uint64_t), kNcclUint8, r, c->nccl, ctx->stream));
    KMI_NCCL(c, rccl().Recv(d_r + (size_t)r * n, n * sizeof(uint64_t), kNcclUint8, r, c->nccl, ctx->stream));
  }
  KMI_NCCL(c, rccl().GroupEnd());
  KMI_HIP(ctx, hipMemcpyAsync(all, d_r, bytes, hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KMI_OK;
}

// all2allv that does not hold the context's stream: queued on the communicator's own stream behind everything the context's
// stream holds now (the send buffer's producer), so the caller's next kernels run beside the transfer. No host synchronisation.
// largest_bytes: the largest peer message anywhere (comm_all_to_all_counts2). An exchange larger than any that carried checksums
// so far is done the synchronous, verified way instead. comm_exchange_join orders the context's stream behind the transfers.
kmi_status comm_all_to_all_v_async(kmi_comm *c, const void *send_dev, const uint64_t *send_counts, void *recv_dev, const uint64_t *recv_counts,
                                   size_t elem_bytes, uint64_t largest_bytes) {
  kmi_ctx *ctx = c->ctx;
  const int p = c->nranks;
  if (!c->verified || largest_bytes > c->verified_bytes || c->has_transport)   // (a transport exchange is synchronous: nothing to overlap)
    return comm_all_to_all_v(c, send_dev, send_counts, recv_dev, recv_counts, elem_bytes);
  std::vector<uint64_t> sb(p), rb(p);
  for (int r = 0; r < p; ++r) { sb[r] = send_counts[r] * elem_bytes; rb[r] = recv_counts[r] * elem_bytes; }
  KMI_HIP(ctx, hipEventRecord(c->ev_ready, ctx->stream));
  KMI_HIP(ctx, hipStreamWaitEvent(c->xstream, c->ev_ready, 0));
  KMI_TRY(a2a_bytes(c, (const char *)send_dev, sb.data(), (char *)recv_dev, rb.data(), c->xstream, largest_bytes));
  return KMI_OK;
}
kmi_status comm_exchange_join(kmi_comm *c) {
  kmi_ctx *ctx = c->ctx;
  KMI_HIP(ctx, hipEventRecord(c->ev_done, c->xstream));
  KMI_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
  return KMI_OK;
}
// the host waits for the transfers queued so far (a send buffer is about to be rewritten)
kmi_status comm_exchange_wait(kmi_comm *c) {
  KMI_HIP(c->ctx, hipStreamSynchronize(c->xstream));
  return KMI_OK;
}

kmi_status comm_allreduce_sum(kmi_comm *c, uint64_t *value) {
  kmi_ctx *ctx = c->ctx;
  if (c->nranks == 1) return KMI_OK;
  const int p = c->nranks;
  if (c->has_transport) return c->transport.allreduce_u64(c->transport.user, value, 1, 0) != 0 ? tp_fail(c, "allreduce_u64") : KMI_OK;
  c->h_small[4 * p] = *value;
  KMI_HIP(ctx, hipMemcpyAsync(c->d_small + 4 * p, c->h_small + 4 * p, sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  KMI_NCCL(c, rccl().AllReduce(c->d_small + 4 * p, c->d_small + 4 * p + 1, 1, kNcclUint64, kNcclSum, c->nccl, ctx->stream));
  KMI_HIP(ctx, hipMemcpyAsync(c->h_small + 4 * p + 1, c->d_small + 4 * p + 1, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  KMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *value = c->h_small[4 * p + 1];
  return KMI_OK;
}

kmi_ctx *comm_ctx(kmi_comm *c) { return c->ctx; }
int comm_size(kmi_comm *c) { return c->nranks; }
int comm_rank(kmi_comm *c) { return c->rank; }

}  // namespace kmi

extern "C" {

kmi_status kmi_comm_unique_id(void *id_out) {
  if (!id_out) return KMI_ERR_INVALID;
  if (!rccl().ok) return KMI_ERR_DEVICE;
  ncclUniqueId_ id;
  if (rccl().GetUniqueId(&id) != 0) return KMI_ERR_DEVICE;
  memcpy(id_out, &id, sizeof(id));
  return KMI_OK;
}

kmi_status kmi_comm_create(kmi_ctx *ctx, const void *id, kmi_comm **out) {
  if (!ctx || !out) return KMI_ERR_INVALID;
  if (!id && ctx->nranks > 1) return kmi::set_err(ctx, KMI_ERR_INVALID, "more than one rank needs the ncclUniqueId rank 0 made (kmi_comm_unique_id)%s", "");
  if (!rccl().ok) return kmi::set_err(ctx, KMI_ERR_DEVICE, "RCCL is not available (librccl.so.1 could not be loaded; KMI_RCCL_LIB overrides the path)%s", "");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  kmi_comm *c = new kmi_comm();
  c->ctx = ctx; c->rank = ctx->rank; c->nranks = ctx->nranks;
  ncclUniqueId_ uid;
  if (id) memcpy(&uid, id, sizeof(uid));
  else if (rccl().GetUniqueId(&uid) != 0) { delete c; return kmi::set_err(ctx, KMI_ERR_DEVICE, "ncclGetUniqueId failed%s", ""); }
  ncclResult_t_ r = rccl().CommInitRank(&c->nccl, c->nranks, uid, c->rank);
  if (r != 0) {
    delete c;
    return kmi::set_err(ctx, KMI_ERR_DEVICE, "ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
  }
  const size_t small = sizeof(uint64_t) * (8 * (size_t)c->nranks + 16);
  if (hipMalloc((void **)&c->d_small, small) != hipSuccess || hipHostMalloc((void **)&c->h_small, small) != hipSuccess ||
      hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
    kmi_comm_destroy(c);
    return kmi::set_err(ctx, KMI_ERR_NOMEM, "communicator scratch%s", "");
  }
  *out = c;
  return KMI_OK;
}

kmi_status kmi_comm_create_transport(kmi_ctx *ctx, const kmi_transport *transport, kmi_comm **out) {
  if (!ctx || !out || !transport) return KMI_ERR_INVALID;
  if (!transport->all_to_all_v || !transport->allreduce_u64) return kmi::set_err(ctx, KMI_ERR_INVALID, "a transport needs both callbacks%s", "");
  KMI_HIP(ctx, hipSetDevice(ctx->device));
  kmi_comm *c = new kmi_comm();
  c->ctx = ctx; c->rank = ctx->rank; c->nranks = ctx->nranks;
  c->has_transport = true; c->transport = *transport;
  const size_t small = sizeof(uint64_t) * (8 * (size_t)c->nranks + 16);
  if (hipMalloc((void **)&c->d_small, small) != hipSuccess || hipHostMalloc((void **)&c->h_small, small) != hipSuccess ||
      hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
    kmi_comm_destroy(c);
    return kmi::set_err(ctx, KMI_ERR_NOMEM, "communicator scratch%s", "");
  }
  *out = c;
  return KMI_OK;
}

kmi_status kmi_comm_destroy(kmi_comm *c) {
  if (!c) return KMI_OK;
  (void)hipSetDevice(c->ctx->device);
  for (int i = 0; i < 2; ++i) if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->d_small) (void)hipFree(c->d_small);
  if (c->h_small) (void)hipHostFree(c->h_small);
  delete c;
  return KMI_OK;
}

kmi_status kmi_comm_all_to_all_counts(kmi_comm *c, const uint64_t *send_counts_host, uint64_t *recv_counts_host) {
  if (!c || !send_counts_host || !recv_counts_host) return KMI_ERR_INVALID;
  KMI_HIP(c->ctx, hipSetDevice(c->ctx->device));
  return kmi::comm_all_to_all_counts(c, send_counts_host, recv_counts_host);
}

kmi_status kmi_comm_all_to_all_v(kmi_comm *c, const void *send_dev, const uint64_t *send_counts_host, void *recv_dev,
                                 const uint64_t *recv_counts_host, size_t elem_bytes) {
  if (!c || !send_counts_host || !recv_counts_host || elem_bytes == 0) return KMI_ERR_INVALID;
  KMI_HIP(c->ctx, hipSetDevice(c->ctx->device));
  return kmi::comm_all_to_all_v(c, send_dev, send_counts_host, recv_dev, recv_counts_host, elem_bytes);
}

kmi_status kmi_comm_allreduce_sum_u64(kmi_comm *c, uint64_t *value_host) {
  if (!c || !value_host) return KMI_ERR_INVALID;
  KMI_HIP(c->ctx, hipSetDevice(c->ctx->device));
  return kmi::comm_allreduce_sum(c, value_host);
}

}  // extern "C"
