"""GPU box (soak, not a test): the sharded solve at BASELINE.json configs[3] (C4: 1000 views / 500 000 tracks / 3.0 M observations, 94 tiles in
the reduced system) on 2 and 4 ranks sharing the one GPU (collective staged through the host, tests/sharded_worker.py), every rank
against the unsharded solve of the same problem.  Prints one JSON line per world size."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_distributed_gpu import _run_sharded_workers, _check_sharded_against_unsharded
for world in (2, 4):
    res, outs = _run_sharded_workers(world, dict(SHARD_CONFIG="C4", SHARD_MIXED="1", SHARD_INNER="0", THEIA_HIP_CREATE_TIMING="1", SHARD_MAX_ITERATIONS="8"), 1500)
    _check_sharded_against_unsharded(res, outs, world)
    print(json.dumps({"config": "C4", "world_size": world, "iterations": [r["iterations"] for r in res], "ref_iterations": res[0]["ref_iterations"],
                      "tracks_per_rank": [r["tracks"] for r in res], "final_cost": res[0]["final_cost"], "ref_final_cost": res[0]["ref_final_cost"],
                      "max_trace_cost_err_rel": max(r["trace_cost_err"] for r in res), "max_cam_err": max(r["cam_err"] for r in res),
                      "max_pts_rel_err": max(r["pts_rel_err"] for r in res), "all_ranks_same_cost_bits": len({r["final_cost"] for r in res}) == 1,
                      "k3_lines": [l for o in outs for l in o.splitlines() if "distributed K3" in l][:world]}))
