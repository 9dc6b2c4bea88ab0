"""Worker of tests/test_distributed_gpu.py::test_sharded_solve_two_ranks_on_one_gpu: one rank of a world_size-2 sharded BA
solve.  Both ranks share cuda:0 (RCCL refuses two ranks on one device), so the all-reduce callback goes through the
host with gloo: device buffer -> host, dist.all_reduce, host -> device, all on the library's stream.  Everything else
-- the track shards, the packed reduced system, the per-rank slots of the gradient maximum, the device-side step control
over collectives -- is the product path of a two-GPU run.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from pytheiasfm_amd import ba, distributed as tdist, synth


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allreduce = tdist.make_host_staged_allreduce(0)
    prior_only = int(os.environ.get("SHARD_PRIORS", "0"))   # position priors without inner iterations / intrinsics (held by rank 0)

    mixed = bool(int(os.environ.get("SHARD_MIXED", "0")))
    inner = int(os.environ.get("SHARD_INNER", "0"))     # 1: inner iterations, 2: inner iterations with free intrinsics + a prior
    nviews, ntracks = int(os.environ.get("SHARD_VIEWS", "24")), int(os.environ.get("SHARD_TRACKS", "1500"))
    cfg = os.environ.get("SHARD_CONFIG", "")            # "C2" / "C4": BASELINE.json's configurations at their full size
    if cfg:
        p = synth.synth_ba_v1(synth.BA_CONFIGS[cfg][0], synth.BA_CONFIGS[cfg][1], seed=synth.BA_CONFIGS[cfg][2], mixed_models=mixed or synth.BA_CONFIGS[cfg][3])
        nviews, ntracks = p.cam_ext.shape[0], p.points.shape[0]
    else:
        p = synth.synth_ba_v1(nviews, ntracks, seed=0x5AD00 + int(mixed), mixed_models=mixed)
    o = ba.default_options(); o.use_inner_iterations = 1 if inner else 0; o.max_num_iterations = int(os.environ.get("SHARD_MAX_ITERATIONS", "12"))
    if inner == 2:
        o.intrinsics_to_optimize = 0x11; o.prior_mask = 1
        mask = np.zeros(nviews, dtype=np.uint8); mask[[3, 11, 17]] = 1
        p.set_priors(mask, position=(p.cam_ext[:, :3] + 0.02, np.tile(4.0 * np.eye(3), (nviews, 1, 1))))
    if prior_only:
        o.prior_mask = 1
        mask = np.zeros(nviews, dtype=np.uint8); mask[::7] = 1
        p.set_priors(mask, position=(p.cam_ext[:, :3] + 0.02, np.tile(4.0 * np.eye(3), (nviews, 1, 1))))
    ref = p.copy()
    s0, tr0 = ba.solve(ref, o)
    shard, ids = synth.shard_tracks(p, rank, world)
    if (inner == 2 or prior_only) and rank == 0:        # the priors of a sharded solve live on exactly one rank
        shard.set_priors(p.cam_prior_mask, **p.priors)
    with ba.BaHandle(shard, o) as h:
        h.set_allreduce(allreduce)
        h.set_shard(rank, world)
        if inner:
            h.set_inner_global(p, ids)
        s, tr = h.run()
        out = h.download(shard.copy())
        h_plan = h.plan_info()
    res = {
        "rank": rank, "iterations": int(s.num_iterations), "ref_iterations": int(s0.num_iterations),
        "final_cost": float(s.final_cost), "ref_final_cost": float(s0.final_cost),
        "cam_err": float(np.abs(out.cam_ext - ref.cam_ext).max()),
        "pts_err": float(np.abs(out.points - ref.points[ids]).max()),
        "trace_cost_err": float(np.abs(np.asarray(tr.cost)[:tr.size] - np.asarray(tr0.cost)[:tr0.size]).max() / s0.initial_cost) if tr.size == tr0.size else 1.0,
        "plan": h_plan, "tracks": int(len(ids)), "total_tracks": int(ntracks),
        "pts_rel_err": float((np.abs(out.points - ref.points[ids]) / np.maximum(1.0, np.abs(ref.points[ids]))).max()),
        "trace_size": int(tr.size), "ref_trace_size": int(tr0.size), "intr_err": float((np.abs(out.intrinsics - ref.intrinsics) / np.maximum(1.0, np.abs(ref.intrinsics))).max()),
    }
    print("RESULT " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
