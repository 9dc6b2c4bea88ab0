"""One-process-per-GPU plumbing for the sharded BA path.

Tracks (and their observations) are partitioned across ranks
(synth.shard_tracks / any caller-side partition); cameras are replicated.  Per
LM iteration the library hands its partial reduced camera system
[S | rhs | colsq | g_c | scalars] to the all-reduce callback below, which sums
it over ranks with RCCL (torch.distributed backend "nccl" IS RCCL on ROCm) on
the library's own HIP stream.  torch is plumbing here: device pointer ->
tensor view -> ncclAllReduce.
"""
import numpy as np

from . import _capi as capi

REDUCE_SUM = 0
REDUCE_MAX = 1


class _DevArray:
    """__cuda_array_interface__ view of library-owned device memory."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 3,
            "strides": None,
        }


def make_torch_allreduce(device_index, group=None):
    """All-reduce callback for BaHandle.set_allreduce backed by
    torch.distributed (NCCL/RCCL for CUDA tensors)."""
    import torch
    import torch.distributed as dist

    streams = {}
    views = {}   # (ptr, count) -> tensor view: the library reduces the same few buffers every iteration

    def allreduce(ptr, count, op, stream):
        t = views.get((ptr, count))
        if t is None:
            t = torch.as_tensor(_DevArray(ptr, count), device=torch.device("cuda", device_index))
            views[(ptr, count)] = t
        ext = streams.get(stream)
        if ext is None:
            ext = torch.cuda.ExternalStream(stream, device=torch.device("cuda", device_index))
            streams[stream] = ext
        with torch.cuda.stream(ext):
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == REDUCE_MAX else dist.ReduceOp.SUM, group=group)
        return 0

    return allreduce


def make_host_allreduce(group=None):
    """CPU (gloo) variant used by the world_size-2 tests: reduces a host numpy
    buffer in place.  Signature (array, op)."""
    import torch
    import torch.distributed as dist

    def allreduce(arr, op=REDUCE_SUM):
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == REDUCE_MAX else dist.ReduceOp.SUM, group=group)
        return arr

    return allreduce


def gather_points(shard_points, track_ids, num_points_total, group=None):
    """Collect every rank's optimised points into the full [num_points][4]
    array (no data-path collective: only used to hand results back)."""
    import torch
    import torch.distributed as dist

    full = np.zeros((num_points_total, 4))
    full[track_ids] = shard_points
    t = torch.from_numpy(full)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return full
