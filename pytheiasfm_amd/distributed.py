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


def make_host_staged_allreduce(device_index, group=None):
    """All-reduce callback for BaHandle.set_allreduce when several ranks share ONE device (RCCL refuses two ranks on one
    GPU): device buffer -> host, gloo all-reduce, host -> device, all ordered on the library's stream.  Everything else of a
    sharded solve -- track shards, the packed reduced system, the per-rank gradient slots, the device-side step control
    across collectives -- is the product path of a multi-GPU run (tests/sharded_worker.py, bench.py --one-gpu-dry-run)."""
    import torch
    import torch.distributed as dist

    streams, views = {}, {}

    def allreduce(ptr, count, op, stream):
        t = views.get((ptr, count))
        if t is None:
            t = torch.as_tensor(_DevArray(ptr, count), device=torch.device("cuda", device_index)); views[(ptr, count)] = t
        ext = streams.get(stream)
        if ext is None:
            ext = torch.cuda.ExternalStream(stream, device=torch.device("cuda", device_index)); streams[stream] = ext
        with torch.cuda.stream(ext):
            host = t.cpu()                      # waits for the work enqueued on the library's stream so far
            dist.all_reduce(host, op=dist.ReduceOp.MAX if op == REDUCE_MAX else dist.ReduceOp.SUM, group=group)
            t.copy_(host)                       # enqueued on the same stream, ahead of what the library enqueues next
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


def exchange_owned_blocks(blocks, rank, world_size, allreduce):
    """The exchange step of the sharded inner iterations (theia_hip_ba_set_inner_global; csrc/ba_solver.hip): block i of
    `blocks` ([n][k], float64) was swept by rank i % world_size; every rank zeroes the blocks it does not own and the buffers
    are summed -- x + 0 is exact, so every rank ends with every owner's bits.  In place; returns `blocks`."""
    owned = (np.arange(blocks.shape[0]) % world_size) == rank
    blocks[~owned] = 0.0
    allreduce(blocks.reshape(-1))
    return blocks


class NativeRccl:
    """RCCL communicator owned by libtheia_hip.so (theia_hip_rccl_*): the sharded solve then issues ncclAllReduce
    itself on its own stream.  The 128-byte unique id travels from rank 0 through torch.distributed's object
    broadcast (any host channel would do); torch takes no part in the LM iterations afterwards."""

    def __init__(self, rank, world_size, group=None):
        import ctypes as C
        import torch.distributed as dist
        L = capi.lib()
        L.theia_hip_rccl_unique_id.argtypes = [C.c_void_p]
        L.theia_hip_rccl_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.theia_hip_rccl_comm_destroy.argtypes = [C.c_void_p]
        L.theia_hip_ba_set_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        buf = C.create_string_buffer(128)
        if rank == 0:
            capi.check(L.theia_hip_rccl_unique_id(buf))
        box = [bytes(buf.raw)]
        if world_size > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self._id = C.create_string_buffer(box[0], 128)
        self._comm = C.c_void_p(None)
        capi.check(L.theia_hip_rccl_comm_create(self._id, int(rank), int(world_size), C.byref(self._comm)))
        self.rank, self.world_size = int(rank), int(world_size)

    def attach(self, handle):
        """handle: ba.BaHandle.  Replaces any all-reduce callback; also declares the shard geometry."""
        capi.check(capi.lib().theia_hip_ba_set_rccl(handle._h, self._comm, self.rank, self.world_size))

    def count(self):
        """Ranks of the communicator as RCCL reports them (ncclCommCount)."""
        import ctypes as C
        L = capi.lib()
        L.theia_hip_rccl_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        n = C.c_int32(0)
        capi.check(L.theia_hip_rccl_comm_count(self._comm, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._comm:
            capi.lib().theia_hip_rccl_comm_destroy(self._comm)
            self._comm = None


def shard_problems(num_problems, rank, world_size):
    """RANSAC over several GPUs: image pairs are independent units, dealt round robin; no collective
    on the data path.  Returns the global indices of this rank's problems."""
    return np.arange(int(rank), int(num_problems), int(world_size), dtype=np.int64)


def estimate_batch_sharded(estimator, data, offsets, params, rank, world_size, estimator_params=None, gather=True, group=None):
    """theia_hip_ransac_estimate_batch on this rank's share of the problems (pairs[rank::world_size]); problem i keeps
    the seed it has in the unsharded batch (params.seed + i).  With gather, every rank receives the per-problem
    results of the whole batch (host-side all_gather_object: results only, never correspondences)."""
    from . import ransac
    offsets = np.asarray(offsets, dtype=np.int64)
    P = len(offsets) - 1
    mine = shard_problems(P, rank, world_size)
    pc = params.to_c() if isinstance(params, ransac.RansacParameters) else params
    base_seed = int(pc.seed)
    parts = [np.asarray(data[offsets[i]:offsets[i + 1]]) for i in mine]
    loc_off = np.zeros(len(mine) + 1, dtype=np.int64)
    if len(mine):
        loc_off[1:] = np.cumsum([p.shape[0] for p in parts])
    out = {"index": mine, "success": np.zeros(0, np.int32), "models": np.zeros((0, capi.THEIA_RANSAC_MODEL_STRIDE)),
           "num_inliers": np.zeros(0, np.int32), "num_iterations": np.zeros(0, np.int32), "inlier_masks": [],
           "hypotheses_evaluated": 0, "models_scored": 0, "time_fit_seconds": 0.0, "time_score_seconds": 0.0}
    if len(mine):
        # one batched launch; every problem keeps the seed it has in the unsharded batch (params.seed + global index)
        r = ransac.estimate_batch(estimator, np.concatenate(parts, axis=0), loc_off, pc, estimator_params, seeds=base_seed + mine)
        out["success"] = r["success"]; out["models"] = r["models"]; out["num_inliers"] = r["num_inliers"]
        out["num_iterations"] = r["num_iterations"]
        out["inlier_masks"] = [r["inlier_mask"][loc_off[k]:loc_off[k + 1]] for k in range(len(mine))]
        for key in ("hypotheses_evaluated", "models_scored", "time_fit_seconds", "time_score_seconds"):
            out[key] = r[key]
    if not gather or world_size == 1:
        return out
    import torch.distributed as dist
    boxes = [None] * world_size
    dist.all_gather_object(boxes, {k: out[k] for k in ("index", "success", "models", "num_inliers", "num_iterations", "inlier_masks")}, group=group)
    full = {"success": np.zeros(P, np.int32), "models": np.zeros((P, capi.THEIA_RANSAC_MODEL_STRIDE)), "num_inliers": np.zeros(P, np.int32),
            "num_iterations": np.zeros(P, np.int32), "inlier_masks": [None] * P}
    for b in boxes:
        for k, i in enumerate(b["index"]):
            full["success"][i] = b["success"][k]; full["models"][i] = b["models"][k]; full["num_inliers"][i] = b["num_inliers"][k]
            full["num_iterations"][i] = b["num_iterations"][k]; full["inlier_masks"][i] = b["inlier_masks"][k]
    return full
