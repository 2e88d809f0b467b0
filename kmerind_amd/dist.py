"""The exchange step of imxx::distribute (src/io/incremental_mxx.hpp:1087,1098) over
torch.distributed: mxx::all2all(counts) -> all_to_all_single of the count vector,
mxx::all2allv(payload) -> all_to_all_single with split sizes. backend "nccl" is RCCL
over xGMI on MI355X; the same code runs on "gloo" CPU tensors (tests).

Receive buffer = concatenation by source rank ascending, as the reference's."""
import torch
import torch.distributed as dist


def exchange_counts(send_counts, group=None, device=None):
    """send_counts: sequence of world_size ints -> list of recv counts"""
    t = torch.as_tensor([int(c) for c in send_counts], dtype=torch.int64, device=device)
    r = torch.empty_like(t)
    dist.all_to_all_single(r, t, group=group)
    return [int(x) for x in r.tolist()]


# The RCCL build of this image moves a peer message correctly only up to 2^27 eight-byte elements (1 GiB); a
# one-rank all_to_all_single of 2^27 + 1 int64 comes back half wrong (tools/a2a_debug.py). Larger exchanges go in pieces.
MSG_MAX_WORDS = 1 << 27


def exchange_keys(send, send_counts, group=None):
    """send: int64 tensor [n, n_words] grouped by destination rank; returns (recv, recv_counts).
    recv is the concatenation by source rank ascending (the reference's receive buffer), also when the peer messages are
    too large for one transfer and travel in pieces (every source's rows stay contiguous and in order)."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world and send.dim() == 2
    assert int(sum(send_counts)) == send.shape[0]
    nw = send.shape[1]
    send_counts = [int(c) for c in send_counts]
    recv_counts = exchange_counts(send_counts, group, send.device)
    rows_max = max(1, MSG_MAX_WORDS // max(1, nw))
    # the number of pieces must be the same on every rank: it follows from the largest message anywhere
    biggest = torch.tensor([max(send_counts + recv_counts)], dtype=torch.int64, device=send.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    pieces = max(1, -(-int(biggest.item()) // rows_max))

    def a2a(out, inp, **kw):
        dist.all_to_all_single(out, inp.contiguous(), group=group, **kw)
    return _a2a_rows_in_order(a2a, send, send_counts, recv_counts, pieces), recv_counts


def verify_exchange(send, sc, recv, rc, group=None, stage_through_host=False):
    """End-to-end check of one all-to-all: the wrap-around 64-bit sum of every peer message, computed by the sender, travels
    next to it and is compared with the sum of what arrived. bench.py and DistributedCountIndex run it on the FIRST exchange
    of a process: the RCCL build of this image was seen to corrupt peer messages above 1 GiB (tools/a2a_debug.py), and that
    limit is a property of one library build, not a constant -- so it is asserted at run time rather than trusted.
    send / recv: integer tensors whose rows are grouped by destination / source; sc / rc rows per peer."""
    world = dist.get_world_size(group)

    def sums(t, counts):
        width = 1
        for d in t.shape[1:]:
            width *= int(d)
        flat = t.reshape(t.shape[0], width).to(torch.int64)       # (an explicit width: a rank may have nothing to send)
        out, off = [], 0
        for c in counts:
            out.append(flat[off:off + c].sum() if c else torch.zeros((), dtype=torch.int64, device=t.device))
            off += c
        return torch.stack(out)
    mine = sums(send, sc)
    got = sums(recv, rc)
    if stage_through_host:
        mine, got = mine.cpu(), got.cpu()
    theirs = torch.empty_like(mine)
    dist.all_to_all_single(theirs, mine, group=group)
    bad = (theirs != got).nonzero().flatten().tolist()
    if bad:
        raise RuntimeError("all-to-all payload corrupted: messages from ranks %s to rank %d do not match their senders' checksums "
                           "(%d ranks)" % (bad, dist.get_rank(group), world))
    return True


def _a2a_rows_in_order(a2a, send, sc, rc, pieces):
    """all-to-all of row blocks (send grouped by destination, sc / rc rows per peer) that keeps every source's rows
    contiguous and in order in the result even when the peer messages travel in `pieces` pieces (the same number on every
    rank; the RCCL build of this image moves at most 2^27 elements per message correctly)"""
    world = len(sc)
    out = torch.empty((int(sum(rc)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if pieces <= 1:
        a2a(out, send, output_split_sizes=rc, input_split_sizes=sc)
        return out
    s_off, r_off = [0], [0]
    for c in sc:
        s_off.append(s_off[-1] + c)
    for c in rc:
        r_off.append(r_off[-1] + c)
    for p in range(pieces):
        lo = [(c * p) // pieces for c in sc]
        hi = [(c * (p + 1)) // pieces for c in sc]
        rlo = [(c * p) // pieces for c in rc]
        rhi = [(c * (p + 1)) // pieces for c in rc]
        chunk = torch.cat([send[s_off[r] + lo[r]: s_off[r] + hi[r]] for r in range(world)])
        out_split = [rhi[r] - rlo[r] for r in range(world)]
        tmp = torch.empty((int(sum(out_split)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        a2a(tmp, chunk, output_split_sizes=out_split, input_split_sizes=[hi[r] - lo[r] for r in range(world)])
        pos = 0
        for src in range(world):                                # piece p of source src goes behind its earlier pieces
            out[r_off[src] + rlo[src]: r_off[src] + rhi[src]] = tmp[pos:pos + out_split[src]]
            pos += out_split[src]
    return out


def exchange_pairs(keys, counts, bucket_counts, group=None, stage_through_host=False, verify=False):
    """The exchange of the combine-first count insert (kmerind_hip.h, kmi_index_split_by_rank_dev): keys int64 [n, n_words]
    and counts int32 [n] grouped by destination rank, bucket_counts int32 [world, B] (row r describes the message to
    rank r). Returns (recv_keys, recv_counts, recv_bucket_counts [world, B] with row s = the message from rank s).
    Three collectives: the bucket-count matrix (equal splits), the keys and the counts."""
    world = dist.get_world_size(group)
    assert bucket_counts.shape[0] == world and keys.dim() == 2 and counts.shape[0] == keys.shape[0]
    dev = keys.device
    cdev = torch.device("cpu") if stage_through_host else dev

    def a2a(out, inp, **kw):
        if stage_through_host:                    # gloo rehearsal with device-resident data
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu().contiguous(), group=group, **kw)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp.contiguous(), group=group, **kw)

    rb = torch.empty_like(bucket_counts)
    a2a(rb, bucket_counts)
    sc = [int(x) for x in bucket_counts.to(torch.int64).sum(dim=1).tolist()]
    rc = [int(x) for x in rb.to(torch.int64).sum(dim=1).tolist()]
    assert sum(sc) == keys.shape[0]
    # messages above the RCCL limit travel in pieces; the piece count must agree on all ranks
    biggest = torch.tensor([max(sc + rc + [0])], dtype=torch.int64, device=cdev)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    pieces = max(1, -(-int(biggest.item()) * keys.shape[1] // MSG_MAX_WORDS))
    rk = _a2a_rows_in_order(a2a, keys, sc, rc, pieces)
    if verify:
        verify_exchange(keys, sc, rk, rc, group, stage_through_host)
    # counts of one rank's own reads are small: when every rank's fit a byte they travel as bytes (9 instead of 12 bytes
    # per pair on the link)
    big = torch.tensor([int(counts.max().item()) if counts.numel() else 0], dtype=torch.int64, device=cdev)
    dist.all_reduce(big, op=dist.ReduceOp.MAX, group=group)
    if int(big.item()) <= 255:
        rv = _a2a_rows_in_order(a2a, counts.to(torch.uint8), sc, rc, pieces).to(counts.dtype)
    else:
        rv = _a2a_rows_in_order(a2a, counts, sc, rc, pieces)
    return rk, rv, rb


class DistributedCountIndex:
    """CountIndex over all ranks of the process group (Index<CountingMap>::build_mmap with comm.size() > 1,
    kmer_index.hpp:148-190 + distributed_unordered_map.hpp:1697-1745), combine-first: every rank reduces the k-mers of its
    own reads into a scratch index, splits that by KeyToRank, exchanges (k-mer, count) pairs and merges what it receives.
    One process per GPU; the group's backend is RCCL ("nccl") on MI355X, gloo with stage_through_host=True in rehearsals."""

    def __init__(self, ctx, cfg, group=None, stage_through_host=False, device=None):
        from .core import CountIndex, num_buckets
        self.ctx, self.cfg, self.group, self.stage = ctx, cfg, group, stage_through_host
        self.device = device                       # torch device of this rank's GPU (collectives run there unless staged)
        self.world = dist.get_world_size(group)
        self.index = CountIndex(ctx, cfg)          # this rank's share of the distributed map
        self.scratch = CountIndex(ctx, cfg)        # local reduction of one input partition
        self.nb = num_buckets()
        self.n_words = self.index.n_words
        self._cap = 0
        self._verified = False                     # the first exchange of this object carries checksums (verify_exchange)
        self.last_send_counts = None               # pairs sent to every rank by the last build (peer balance)
        self._sk_cap = {}                          # records the chunks of the last super-k-mer build produced (send buffer sizes)

    def _buffers(self, n, dev):
        if n > self._cap or not hasattr(self, "_keys"):     # (a first build that yields no k-mer still needs the buffers)
            self._cap = int(n * 1.1) + 1024
            self._keys = torch.empty((self._cap, self.n_words), dtype=torch.int64, device=dev)
            self._counts = torch.empty((self._cap,), dtype=torch.int32, device=dev)
            self._bcnt = torch.empty((self.world, self.nb), dtype=torch.int32, device=dev)
        return self._keys, self._counts, self._bcnt

    def build_device(self, dptr, nbytes, device=None, mode="auto", bounds=None):
        """adds the k-mers of this rank's FASTQ/FASTA partition (device bytes) to the distributed index.
        mode "superkmer": the ranks exchange super-k-mer records (16 bytes for about nine k-mers) and a k-mer lives on the
        rank that owns its minimizer's bucket (include/kmerind_hip.h, kmi_index_sk_produce_dev); "combine": every rank
        reduces its own reads first and (k-mer, count) pairs travel to KeyToRank(k-mer); "auto": super-k-mers where they
        apply (FASTQ, one-word DNA k-mers, 2 / 4 / 8 ranks, an index that is empty or was built that way), else combine.
        bounds (super-k-mers): byte offsets [0, ..., nbytes] of record-aligned chunks, the same number on every rank -- the
        exchange of one chunk then travels while the next one is cut into records. self.last_mode says which mode ran."""
        device = device or self.device
        if mode in ("auto", "superkmer") and self._build_superkmer(dptr, nbytes, device, bounds):
            self.last_mode = "superkmer"
            return
        if mode == "superkmer":
            raise RuntimeError("the super-k-mer exchange does not apply to this index / rank count")
        if self.index.owner_ranks() > 1:
            raise RuntimeError("this index holds entries distributed by minimizer owner: a KeyToRank-routed insert would split keys over two ranks")
        self.last_mode = "combine"
        self.scratch.clear()
        self.scratch.build_device(dptr, nbytes)
        n = self.scratch.local_size()
        keys, counts, bcnt = self._buffers(n, device)
        sc = self.scratch.split_by_rank_device(self.world, keys.data_ptr(), counts.data_ptr(), self._cap, bcnt.data_ptr())
        self.last_send_counts = [int(x) for x in sc]
        rk, rv, rb = exchange_pairs(keys[:n], counts[:n], bcnt, self.group, self.stage, verify=not self._verified)
        self._verified = True
        self.index.merge_parts_device(self.world, rk.data_ptr(), rv.data_ptr(), rb.data_ptr())

    def _staged(self):
        return self.stage or dist.get_backend(self.group) == "gloo"

    def _build_superkmer(self, dptr, nbytes, device, bounds=None):
        """-> False when the path does not apply (nothing was exchanged; the caller takes another route). Per chunk: records of
        the chunk grouped by owner rank (kmi_index_sk_produce_dev) -> all-to-all, asynchronous over RCCL, so it overlaps the
        next chunk's front end; a chunk the front end cannot take (an item capacity exceeded on some rank) travels as k-mers
        routed to the same owners. What arrived is consumed in one go (kmi_index_sk_consume_dev)."""
        import ctypes as C
        import numpy as np
        from . import _lib as L
        staged = self._staged()
        cdev = torch.device("cpu") if staged else device
        # the path applies to FASTQ builds of one-word 2-bit k-mers with k >= 17 on 1 (rehearsals), 2, 4 or 8 ranks, into an index
        # that is empty or was built this way (kmi_index_sk_width: 0 for every other shape, and with KMI_FUSED_PATH=kmer)
        skw = C.c_uint32(0)
        self.ctx.check(L.lib.kmi_index_sk_width(self.index.h, C.byref(skw)))
        ok = skw.value != 0 and self.world in (1, 2, 4, 8) and (self.index.local_size() == 0 or self.index.owner_ranks() == self.world)
        bounds = [0, nbytes] if not bounds else [int(b) for b in bounds]
        assert bounds[0] == 0 and bounds[-1] == nbytes and all(a <= b for a, b in zip(bounds, bounds[1:]))
        nch = len(bounds) - 1
        # every rank or none, and the same number of chunks everywhere (the exchanges are collectives)
        flag = torch.tensor([int(ok), nch, -nch], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag[0].item()) == 0:
            return False
        if int(flag[1].item()) != nch or -int(flag[2].item()) != nch:
            raise RuntimeError("build_device(bounds=...): the ranks disagree on the number of chunks")
        self.ctx.check(L.lib.kmi_index_set_owner_ranks(self.index.h, self.world))   # (only after every rank agreed)
        # the library writes the records on the context's stream; torch's collectives are ordered behind torch's current stream
        same_stream = getattr(self.ctx, "stream", 0) == (torch.cuda.current_stream(device).cuda_stream if device is not None and device.type == "cuda" else 0)
        recv_parts, works, sends, kmer_chunks = [], [], [], []
        total_sc = [0] * self.world
        pool, pool_pos = None, 0          # one receive buffer for all chunks (sized after the first one), so nothing is concatenated
        for c in range(nch):
            recs_p, n, produced = C.c_void_p(), C.c_uint64(), C.c_int(0)
            sc = np.zeros(self.world, dtype=np.uint64)
            # the records go straight into the send buffer when it is large enough (150-bp reads give one record per 24 input
            # bytes; the size of the last chunk's is remembered), else into library workspace and are copied over
            cb = bounds[c + 1] - bounds[c]
            cap = max(int(cb * 0.05) + 4096, int(self._sk_cap.get(c, 0) * 1.05))
            send = torch.empty((cap, 2), dtype=torch.int64, device=device)
            self.ctx.check(L.lib.kmi_index_sk_produce_dev(self.index.h, C.c_void_p(dptr + bounds[c]), cb, self.world, C.c_void_p(send.data_ptr()), cap,
                                                          C.byref(recs_p), C.byref(n), sc.ctypes.data_as(C.c_void_p), C.byref(produced)))
            self._sk_cap[c] = n.value
            counts = [int(x) for x in sc]
            if not same_stream:
                self.ctx.synchronize()                                  # the records are complete before torch reads them
            # the send counts go round with the verdict (a rank that could not produce this chunk says so with -1) and with the
            # sender's largest message: every rank hears from every rank, so all see the same verdict and the same maximum
            mine = max(counts + [0])
            t = torch.tensor([[cnt if produced.value else -1, mine] for cnt in counts], dtype=torch.int64, device=cdev)
            r = torch.empty_like(t)
            dist.all_to_all_single(r, t, group=self.group)
            rl = r.tolist()
            rc = [int(x[0]) for x in rl]
            largest = max(int(x[1]) for x in rl)
            if not produced.value or min(rc) < 0:
                kmer_chunks.append(c)
                continue
            if recs_p.value == send.data_ptr() or n.value == 0:
                send = send[: n.value]
            else:
                send = torch.empty((n.value, 2), dtype=torch.int64, device=device)
                self.ctx.check(L.lib.kmi_copy_on_device(self.ctx.h, C.c_void_p(send.data_ptr()), recs_p, n.value * 16))
            total_sc = [a + b for a, b in zip(total_sc, counts)]
            n_in = sum(rc)
            first = not self._verified
            if staged or first or largest * 2 > MSG_MAX_WORDS:           # (the same branch on every rank)
                recv, rc2 = self._exchange_dev(send, counts)          # (synchronous; pieces where a message is too large)
                assert rc2 == rc
                if first:
                    verify_exchange(send, counts, recv, rc, self.group, staged)
                    self._verified = True
            else:
                if pool is None:
                    pool = torch.empty((int(n_in * (nch - c) * 1.15) + 4096, 2), dtype=torch.int64, device=device)
                if pool_pos + n_in <= pool.shape[0]:
                    recv, in_pool = pool[pool_pos:pool_pos + n_in], True
                    pool_pos += n_in
                else:
                    recv, in_pool = torch.empty((n_in, 2), dtype=torch.int64, device=device), False
                works.append(dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=counts, group=self.group, async_op=True))
                sends.append(send)                                      # stays alive until the transfer has left
                if in_pool:
                    continue                                            # (the pool is consumed as one piece)
            recv_parts.append(recv)
        for w in works:
            w.wait()
        sends.clear()
        self.last_send_counts = total_sc
        if pool is not None and pool_pos:
            recv_parts.append(pool[:pool_pos])
        if recv_parts:
            allrecv = recv_parts[0] if len(recv_parts) == 1 else torch.cat(recv_parts)
            recv_parts.clear()
            self.ctx.check(L.lib.kmi_index_sk_consume_dev(self.index.h, C.c_void_p(allrecv.data_ptr()), allrecv.shape[0], self.world))
            del allrecv
        for c in kmer_chunks:
            self._owner_routed_kmers(dptr + bounds[c], bounds[c + 1] - bounds[c], device)
        return True

    def _owner_routed_kmers(self, dptr, nbytes, device):
        """a chunk as k-mers: parse, group by the owner of the minimizer's bucket, exchange, insert"""
        import ctypes as C
        import numpy as np
        from . import _lib as L
        nt, ns = C.c_uint64(), C.c_uint64()
        if nbytes:
            self.ctx.check(L.lib.kmi_extract_count_dev(self.ctx.h, C.byref(self.cfg), C.c_void_p(dptr), nbytes, C.byref(nt), C.byref(ns)))
        keys = torch.empty((nt.value + 8, self.n_words), dtype=torch.int64, device=device)
        send = torch.empty_like(keys)
        counts = np.zeros(self.world, dtype=np.uint64)
        if nt.value:
            self.ctx.check(L.lib.kmi_extract_dev(self.ctx.h, C.byref(self.cfg), C.c_void_p(dptr), nbytes, 0, C.c_void_p(keys.data_ptr()), None,
                                                 nt.value, C.byref(nt), C.byref(ns)))
            self.ctx.check(L.lib.kmi_route_owner_dev(self.ctx.h, C.byref(self.cfg), C.c_void_p(keys.data_ptr()), nt.value, self.world,
                                                     C.c_void_p(send.data_ptr()), counts.ctypes.data_as(C.c_void_p)))
        recv, _ = self._exchange_dev(send[: nt.value], [int(x) for x in counts])
        if recv.shape[0]:
            self.index.insert_device(recv.data_ptr(), recv.shape[0], transformed=True)

    # ---- queries (distributed_unordered_map.hpp:880-983 count, :564-687 find, :719-779 erase): transform_input, route the
    # query keys to their owners (imxx::distribute), answer locally per source rank, one return all-to-all. Device buffers
    # throughout: the keys are transformed and grouped by KeyToRank on the device (kmi_route_dev), every source's segment is
    # answered by kmi_index_count_dev / kmi_index_find_dev into device buffers, and only the caller's arrays cross PCIe
    # (over gloo -- the CPU rehearsal -- the exchanged tensors are staged through the host, as everywhere).
    value_words = 0

    def _dev(self):
        return self.device if self.device is not None else torch.device("cuda", self.ctx.device)

    def _exchange_dev(self, send, counts):
        """exchange_keys for device tensors: over RCCL as they are, over gloo through the host"""
        if self.stage or dist.get_backend(self.group) == "gloo":
            recv, rc = exchange_keys(send.cpu(), counts, self.group)
            return recv.to(send.device), rc
        return exchange_keys(send, counts, self.group)

    def _route_queries(self, q):
        """-> (query keys this rank owns, grouped by source rank, as a device tensor [n, n_words]; counts per source)"""
        import ctypes as C
        import numpy as np
        from . import _lib as L
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, self.n_words)
        dev = self._dev()
        d_q = torch.from_numpy(q.view(np.int64)).to(dev)
        d_s = torch.empty_like(d_q)
        counts = np.zeros(self.world, dtype=np.uint64)
        # an index built through exchanged super-k-mers keeps a k-mer on the owner of its minimizer's bucket
        route = L.lib.kmi_route_owner_dev if self.index.owner_ranks() > 1 else L.lib.kmi_route_dev
        self.ctx.check(route(self.ctx.h, C.byref(self.cfg), C.c_void_p(d_q.data_ptr()), q.shape[0], self.world,
                             C.c_void_p(d_s.data_ptr()), counts.ctypes.data_as(C.c_void_p)))
        return self._exchange_dev(d_s, [int(c) for c in counts])

    def _answer(self, mode, q):
        """mode "count" / "find" on the local index per source rank; returns this rank's (keys, values) for its own query keys"""
        import ctypes as C
        import numpy as np
        from . import _lib as L
        mine, recv_counts = self._route_queries(q)
        dev = mine.device
        vw = max(1, self.value_words)
        fn = L.lib.kmi_index_count_dev if mode == "count" else L.lib.kmi_index_find_dev
        out_k, out_v, back = [], [], []
        off = 0
        for src in range(self.world):                                         # answers go back to the rank that asked
            n = recv_counts[src]
            seg = mine[off:off + n]
            off += n
            cap = n
            if self.value_words and mode == "find" and n:
                # a multimap returns every entry of a queried key: the buffers are sized by a count of the hits (the entry count of
                # the index per source rank would be `world` times the index)
                hits = C.c_uint64(0)
                self.ctx.check(L.lib.kmi_index_find_hits_dev(self.index.h, C.c_void_p(seg.data_ptr()), n, C.byref(hits)))
                cap = int(hits.value)
            k = torch.empty((max(cap, 1), self.n_words), dtype=torch.int64, device=dev)
            v = torch.empty((max(cap, 1), vw), dtype=torch.int64, device=dev)
            n_out = C.c_uint64(0)
            if n:
                self.ctx.check(fn(self.index.h, C.c_void_p(seg.data_ptr()), n, C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.byref(n_out)))
            out_k.append(k[:n_out.value]); out_v.append(v[:n_out.value]); back.append(int(n_out.value))
        rk, _ = self._exchange_dev(torch.cat(out_k), back)
        rv, _ = self._exchange_dev(torch.cat(out_v), back)
        keys = rk.cpu().numpy().view(np.uint64)
        vals = rv.cpu().numpy().view(np.uint64)
        if not self.value_words:
            vals = vals[:, 0]
        return keys, vals                                                       # multimaps: [n, value_words]; count() has the multiplicity in column 0

    def count(self, q):
        """one (key, count) per distinct transformed query key of THIS rank's query (0 when absent), as the reference returns"""
        return self._answer("count", q)

    def find(self, q):
        return self._answer("find", q)

    def erase(self, q):
        import numpy as np
        mine, _ = self._route_queries(q)
        host = mine.cpu().numpy().view(np.uint64)                             # (erase takes host keys; they are already transformed)
        n = self.index.erase(host) if host.shape[0] else 0
        return global_size(n, self.group, None if self.stage else self.device)

    def clear(self):
        self.index.clear()

    def local_size(self):
        return self.index.local_size()

    def size(self):
        return global_size(self.local_size(), self.group, None if self.stage else self.device)

    def close(self):
        self.index.close()
        self.scratch.close()


class DistributedPositionIndex(DistributedCountIndex):
    """PositionIndex / PositionQualityIndex over all ranks (Index<unordered_multimap>::build_* + insert with comm.size() > 1,
    kmer_index.hpp:148-225, distributed_unordered_map.hpp:1466-1515): every (k-mer, value) tuple is kept, so nothing can be
    combined before the exchange. Per batch of this rank's partition, all on the device: parse the tuples as records
    (key words, id[, quality bits]), transformed and grouped by KeyToRank (kmi_extract_route_records_dev),
    exchange the records, insert what arrives (kmi_index_insert_tuples_dev). The partition goes through in record-aligned
    batches (kmi_fastq_partition_dev), so the tuple array of a whole partition -- 72 GB per GPU for 200 M reads on 8 --
    never has to exist at once; count / find / erase are the routed queries of the base class."""

    def __init__(self, ctx, cfg, group=None, stage_through_host=False, device=None):
        from .core import PositionIndex
        self.ctx, self.cfg, self.group, self.stage, self.device = ctx, cfg, group, stage_through_host, device
        self.world = dist.get_world_size(group)
        self.index = PositionIndex(ctx, cfg)
        self.scratch = None
        self.n_words, self.value_words = self.index.n_words, self.index.value_words
        self._cap = 0

    def build(self, data, file_offset=0, batch_bytes=None):
        """adds the tuples of this rank's record-aligned partition (host bytes; file_offset = its offset in the file)"""
        import numpy as np
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data, dtype=np.uint8)
        d = torch.from_numpy(buf.copy()).to(self._dev()) if buf.size else torch.empty(0, dtype=torch.uint8, device=self._dev())
        self.build_device(d.data_ptr(), int(buf.size), file_offset, batch_bytes)

    def build_device(self, dptr, nbytes, file_offset=0, batch_bytes=None):
        """the same for a partition resident in HBM; batch_bytes = upper bound of the bytes parsed per exchange (FASTQ only)"""
        import ctypes as C
        import numpy as np
        from . import _lib as L
        from . import fileio
        dev = self._dev()
        rw = self.n_words + self.value_words
        nb = 1
        if batch_bytes and nbytes > batch_bytes and self.cfg.seq_format == L.FMT_FASTQ:
            nb = -(-nbytes // batch_bytes)
        nb = int(global_max(nb, self.group, None if self.stage else self.device))     # every rank enters every exchange
        cuts = fileio.partition_fastq_device(self.ctx, dptr, nbytes, nb) if nb > 1 else [(0, nbytes)]
        cdev = None if self.stage else self.device
        for b, e in cuts:
            nt, ns = C.c_uint64(0), C.c_uint64(0)
            if e > b:
                self.ctx.check(L.lib.kmi_extract_count_dev(self.ctx.h, C.byref(self.cfg), C.c_void_p(dptr + b), e - b, C.byref(nt), C.byref(ns)))
            n = int(nt.value)
            send = torch.empty((n + 8, rw), dtype=torch.int64, device=dev)
            counts = np.zeros(self.world, dtype=np.uint64)
            if n:
                self.ctx.check(L.lib.kmi_extract_route_records_dev(self.ctx.h, C.byref(self.cfg), C.c_void_p(dptr + b), e - b, file_offset + b,
                                                                    self.world, C.c_void_p(send.data_ptr()), n, C.byref(nt), C.byref(ns),
                                                                    counts.ctypes.data_as(C.c_void_p)))
            recv, _ = self._exchange_dev(send[:n], [int(c) for c in counts])
            if recv.shape[0]:
                self.ctx.check(L.lib.kmi_index_insert_tuples_dev(self.index.h, C.c_void_p(recv.data_ptr()), recv.shape[0]))
            del send, recv
        del cdev

    def close(self):
        self.index.close()


def global_max(value, group=None, device=None):
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def global_size(local_size, group=None, device=None):
    """MapType::size(): allreduce of local sizes (distributed_map_base.hpp:227-245)"""
    t = torch.tensor([int(local_size)], dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return int(t.item())
