"""kmi_comm_create_transport from Python: the two callbacks of `kmi_transport` (include/kmerind_hip.h) over a torch.distributed
process group with CPU tensors (gloo). Plumbing for the tests and rehearsals -- with it the library's own multi-rank code
(kmi_index_*_dist_*, kmi_dbg_*_dist_*: the C layer above kmi_comm) runs with several ranks sharing one GPU, which RCCL refuses.
An MPI application would fill the same two function pointers with MPI_Alltoallv / MPI_Allreduce."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L

_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
_ARED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_int)


class _Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_to_all_v", _A2A), ("allreduce_u64", _ARED)]


def _bytes_at(ptr, n):
    if n == 0:
        return torch.zeros(0, dtype=torch.uint8)
    return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * n).from_address(ptr)))


class GroupComm:
    """A kmi_comm whose messenger is a torch.distributed group (backend gloo). Keep the object alive as long as the handle is used."""

    def __init__(self, ctx, group=None):
        self.ctx, self.group = ctx, group
        self.world = dist.get_world_size(group)
        self.calls = {"all_to_all_v": 0, "allreduce": 0, "bytes": 0}
        world = self.world

        def a2a(_user, send, sbytes, recv, rbytes):
            try:
                sb = [int(sbytes[r]) for r in range(world)]
                rb = [int(rbytes[r]) for r in range(world)]
                out = _bytes_at(recv, sum(rb))
                dist.all_to_all_single(out, _bytes_at(send, sum(sb)), output_split_sizes=rb, input_split_sizes=sb, group=group)
                self.calls["all_to_all_v"] += 1
                self.calls["bytes"] += sum(sb)
                return 0
            except Exception:   # (a callback must not raise through the C frames)
                import traceback
                traceback.print_exc()
                return 1

        def ared(_user, values, n, op):
            try:
                a = np.ctypeslib.as_array(values, shape=(n,))
                t = torch.from_numpy(a.view(np.int64).copy())          # (sums wrap the same in both signs; the maxima are sizes)
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
                a[:] = t.numpy().view(np.uint64)
                self.calls["allreduce"] += 1
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        self._cb = (_A2A(a2a), _ARED(ared))
        self._t = _Transport(None, self._cb[0], self._cb[1])
        h = C.c_void_p()
        ctx.check(L.lib.kmi_comm_create_transport(ctx.h, C.byref(self._t), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            L.lib.kmi_comm_destroy(self.h)
            self.h = None
