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


def global_size(local_size, group=None, device=None):
    """MapType::size(): allreduce of local sizes (distributed_map_base.hpp:227-245)"""
    t = torch.tensor([int(local_size)], dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return int(t.item())
