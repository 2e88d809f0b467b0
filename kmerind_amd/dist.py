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
    recv is the concatenation by source rank ascending (the reference's receive buffer) when every peer message fits one
    transfer; otherwise the exchange runs in pieces and recv is ordered by piece, then by source rank (same multiset)."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world and send.dim() == 2
    assert int(sum(send_counts)) == send.shape[0]
    nw = send.shape[1]
    send_counts = [int(c) for c in send_counts]
    recv_counts = exchange_counts(send_counts, group, send.device)
    recv = torch.empty((int(sum(recv_counts)), nw), dtype=send.dtype, device=send.device)
    rows_max = max(1, MSG_MAX_WORDS // max(1, nw))
    # the number of pieces must be the same on every rank: it follows from the largest message anywhere
    biggest = torch.tensor([max(send_counts + recv_counts)], dtype=torch.int64, device=send.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    pieces = max(1, -(-int(biggest.item()) // rows_max))
    if pieces == 1:
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
        return recv, recv_counts
    s_off = [0]
    for c in send_counts:
        s_off.append(s_off[-1] + c)
    pos = 0
    for p in range(pieces):
        def part(c):                      # rows of a message that travel in piece p
            lo, hi = (c * p) // pieces, (c * (p + 1)) // pieces
            return lo, hi
        in_split = [part(c)[1] - part(c)[0] for c in send_counts]
        out_split = [part(c)[1] - part(c)[0] for c in recv_counts]
        chunk = torch.cat([send[s_off[r] + part(send_counts[r])[0]: s_off[r] + part(send_counts[r])[1]] for r in range(world)])
        n_out = int(sum(out_split))
        dist.all_to_all_single(recv[pos:pos + n_out], chunk, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        pos += n_out
    return recv, recv_counts


def exchange_pairs(keys, counts, bucket_counts, group=None, stage_through_host=False):
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
    if max(sc + rc) * keys.shape[1] > MSG_MAX_WORDS:
        raise RuntimeError("a peer message of %d pairs exceeds what this RCCL build moves correctly" % max(sc + rc))
    n_in = sum(rc)
    rk = torch.empty((n_in, keys.shape[1]), dtype=keys.dtype, device=dev)
    rv = torch.empty((n_in,), dtype=counts.dtype, device=dev)
    a2a(rk, keys, output_split_sizes=rc, input_split_sizes=sc)
    a2a(rv, counts, output_split_sizes=rc, input_split_sizes=sc)
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

    def _buffers(self, n, dev):
        if n > self._cap:
            self._cap = int(n * 1.1) + 1024
            self._keys = torch.empty((self._cap, self.n_words), dtype=torch.int64, device=dev)
            self._counts = torch.empty((self._cap,), dtype=torch.int32, device=dev)
            self._bcnt = torch.empty((self.world, self.nb), dtype=torch.int32, device=dev)
        return self._keys, self._counts, self._bcnt

    def build_device(self, dptr, nbytes, device=None):
        """adds the k-mers of this rank's FASTQ/FASTA partition (device bytes) to the distributed index"""
        device = device or self.device
        self.scratch.clear()
        self.scratch.build_device(dptr, nbytes)
        n = self.scratch.local_size()
        keys, counts, bcnt = self._buffers(n, device)
        self.scratch.split_by_rank_device(self.world, keys.data_ptr(), counts.data_ptr(), self._cap, bcnt.data_ptr())
        rk, rv, rb = exchange_pairs(keys[:n], counts[:n], bcnt, self.group, self.stage)
        self.index.merge_parts_device(self.world, rk.data_ptr(), rv.data_ptr(), rb.data_ptr())

    def clear(self):
        self.index.clear()

    def local_size(self):
        return self.index.local_size()

    def size(self):
        return global_size(self.local_size(), self.group, None if self.stage else self.device)

    def close(self):
        self.index.close()
        self.scratch.close()


def global_size(local_size, group=None, device=None):
    """MapType::size(): allreduce of local sizes (distributed_map_base.hpp:227-245)"""
    t = torch.tensor([int(local_size)], dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return int(t.item())
