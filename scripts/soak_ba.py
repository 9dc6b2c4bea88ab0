"""GPU box: mid-size random problems (between the unit tests' sizes and C2), the library's LM trajectory against the CPU oracle's.
A soak, not a test: prints one line per problem and a summary.  Usage: soak_ba.py [count] [first seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import _capi as capi, ba, synth
from tests import oracle_lib as ol

# SOAK_INTR: comma-separated intrinsics_to_optimize masks to draw from (default: none, none, FOCAL | RADIAL)
INTR_CHOICES = [int(x, 0) for x in os.environ.get("SOAK_INTR", "0,0,0x11").split(",")]
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; worst = 0.0
t_all = time.time()
for k in range(count):
    seed = seed0 + k
    rng = np.random.default_rng(0x50AC0000 + seed)
    nv = int(rng.integers(20, 220)); nt = int(rng.integers(1000, 16000))
    p = synth.synth_ba_v1(nv, nt, seed=0x50AC0000 + seed, num_groups=int(rng.integers(1, 9)), mixed_models=bool(rng.integers(0, 2)),
                          fix_gauge=bool(rng.integers(0, 2)))
    n = len(p.obs_pt)
    if rng.integers(0, 2):
        pc = np.zeros(p.points.shape[0], np.uint8); pc[rng.integers(0, p.points.shape[0], p.points.shape[0] // 11)] = 1; p.point_const = pc
    order = [None, "view", "random"][int(rng.integers(0, 3))]
    if order:
        cnt = np.bincount(p.obs_pt, minlength=p.points.shape[0])
        within = np.arange(n) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        perm = np.lexsort((p.obs_pt, within)) if order == "view" else rng.permutation(n)
        p = capi.FlatProblem(p.cam_ext, p.intrinsics, p.group_model, p.cam_group, p.points, p.obs_uv[perm], p.obs_cam[perm], p.obs_pt[perm],
                             p.cam_const, p.group_const, p.point_const)
    o = ba.default_options(); oo = ol.default_options()
    its = int(rng.integers(3, 6))
    cfg = dict(max_num_iterations=its, use_inner_iterations=int(rng.integers(0, 2)), intrinsics_to_optimize=int(rng.choice(INTR_CHOICES)),
               loss_function_type=int(rng.choice([0, 0, 1, 3])), robust_loss_width=2.0)
    for f, v in cfg.items():
        setattr(o, f, v); setattr(oo, f, v)
    if rng.integers(0, 4) == 0:
        os.environ["THEIA_HIP_HOST_CHUNK_MIN"] = "4096"          # the threaded creation passes at this size
    else:
        os.environ.pop("THEIA_HIP_HOST_CHUNK_MIN", None)
    pg = p.copy(); s, tr = ba.solve(pg, o)
    po = p.copy(); so, to = ol.solve(po, oo)
    same = tr.size == to.size and np.array_equal(tr.accepted[: tr.size], to.accepted[: to.size])
    rel = float(np.max(np.abs(tr.cost[: tr.size] - to.cost[: to.size]) / to.cost[: to.size])) if tr.size == to.size else np.inf
    dp = float(np.abs(pg.points - po.points).max())
    ok = same and rel < 1e-8
    bad += not ok; worst = max(worst, rel if np.isfinite(rel) else 1.0)
    print("%3d views %3d tracks %5d obs %6d order %-6s %s threaded=%d: entries %d/%d same=%s cost rel %.1e points %.1e %s" % (
        seed, nv, nt, n, order, cfg, "THEIA_HIP_HOST_CHUNK_MIN" in os.environ, tr.size, to.size, same, rel, dp, "" if ok else "<-- DIFFERS"), flush=True)
print("soak: %d problems, %d differ, worst relative cost difference %.2e, %.0f s" % (count, bad, worst, time.time() - t_all))
