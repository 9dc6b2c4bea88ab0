"""GPU: the all-reduce callback path of the sharded BA solve with a
world_size-1 RCCL group (one GPU box): device-pointer wrapping, the library's
own stream, and the SUM / MAX protocol -- results must equal the plain solve."""
import os

import numpy as np
import pytest

from pytheiasfm_amd import ba, distributed as tdist, synth

pytestmark = pytest.mark.gpu


def test_allreduce_callback_path_world_size_1():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        p = synth.synth_ba_v1(16, 800, seed=81)
        shard, ids = synth.shard_tracks(p, 0, 1)
        o = ba.default_options()
        calls = []
        cb = tdist.make_torch_allreduce(0)

        def counting(ptr, count, op, stream):
            calls.append((count, op))
            return cb(ptr, count, op, stream)

        with ba.BaHandle(shard, o) as h:
            h.set_allreduce(counting)
            s, tr = h.run()
            out = h.download(shard.copy())
        ref = p.copy()
        s0, tr0 = ba.solve(ref, o)
        assert s.num_iterations == s0.num_iterations and abs(s.final_cost - s0.final_cost) <= 1e-9 * s0.final_cost
        assert np.abs(out.cam_ext - ref.cam_ext).max() <= 1e-8 and np.abs(out.points - ref.points).max() <= 1e-8
        # per iteration: SUM of the packed reduced system (the 3 non-zero lower 64x64 tiles of the 96 x 96
        # matrix + rhs | colsq | g_c | 8 scalars), MAX of the gradient scalars, SUM of the trial-step scalars;
        # once per solve: MAX of the tile structure (2 x 2), SUM of the camera column norms, SUM of |x|^2
        n = 6 * 16
        assert (3 * 4096 + 3 * n + 8, tdist.REDUCE_SUM) in calls and (8, tdist.REDUCE_MAX) in calls
        assert (4, tdist.REDUCE_MAX) in calls and (n, tdist.REDUCE_SUM) in calls
        # with the shard geometry known the gradient max-norm rides in the packed SUM (one slot per rank):
        # no 8-double MAX any more, same solve; the gradient norm of the trace proves the slot survived
        calls.clear()
        with ba.BaHandle(shard, o) as h:
            h.set_allreduce(counting)
            h.set_shard(0, 1)
            s2, tr2 = h.run()
            out2 = h.download(shard.copy())
        assert (3 * 4096 + 3 * n + 8 + 1, tdist.REDUCE_SUM) in calls and (8, tdist.REDUCE_MAX) not in calls
        assert s2.num_iterations == s.num_iterations and s2.final_cost == s.final_cost
        assert np.array_equal(out2.cam_ext, out.cam_ext) and np.array_equal(out2.points, out.points)
        assert np.array_equal(np.asarray(tr2.gradient_max_norm), np.asarray(tr.gradient_max_norm))
    finally:
        dist.destroy_process_group()
