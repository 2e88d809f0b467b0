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


def exchange_keys(send, send_counts, group=None):
    """send: int64 tensor [n, n_words] grouped by destination rank; returns (recv, recv_counts)."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world and send.dim() == 2
    assert int(sum(send_counts)) == send.shape[0]
    recv_counts = exchange_counts(send_counts, group, send.device)
    recv = torch.empty((int(sum(recv_counts)), send.shape[1]), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts,
                           input_split_sizes=[int(c) for c in send_counts], group=group)
    return recv, recv_counts


def global_size(local_size, group=None, device=None):
    """MapType::size(): allreduce of local sizes (distributed_map_base.hpp:227-245)"""
    t = torch.tensor([int(local_size)], dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return int(t.item())
