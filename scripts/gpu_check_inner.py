"""GPU check: inner iterations (default ON) against the oracle on small problems."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pytheiasfm_amd import ba, synth, sfm
from tests import oracle_lib as ol

def run(name, p, **kw):
    res = []
    for mod in (ba, ol):
        q = p.copy(); o = mod.default_options(); o.max_num_iterations = 12
        for k, v in kw.items(): setattr(o, k, v)
        s, tr = mod.solve(q, o)
        res.append((s, tr, q))
    (sg, tg, qg), (so, to, qo) = res
    n = min(len(tg.cost), len(to.cost))
    rel = max(abs(tg.cost[k] - to.cost[k]) / max(1e-300, abs(to.cost[k])) for k in range(n)) if n else 0
    print(f"{name}: iters {sg.num_iterations}/{so.num_iterations} succ {sg.num_successful_steps}/{so.num_successful_steps} "
          f"final {sg.final_cost:.9e}/{so.final_cost:.9e} trace rel {rel:.2e} acc {list(tg.accepted[:n])==list(to.accepted[:n])} "
          f"cam {np.abs(qg.cam_ext-qo.cam_ext).max():.2e} pts {np.abs(qg.points-qo.points).max():.2e} intr {np.abs(qg.intrinsics-qo.intrinsics).max():.2e}")
    if n and rel > 1e-6:
        for k in range(n): print("   ", k, tg.cost[k], to.cost[k], tg.accepted[k], to.accepted[k])

p = synth.synth_ba_v1(8, 300, seed=5)
run("plain inner on ", p)
run("plain inner off", p, use_inner_iterations=0)
run("huber          ", p, loss_function_type=1, robust_loss_width=2.0)
run("intrinsics     ", p, intrinsics_to_optimize=int(sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION))
run("xyzw points    ", p, use_homogeneous_point_parametrization=0)
p2 = synth.ba_config("C1")
run("C1             ", p2)
p3 = synth.synth_ba_v1(12, 500, seed=9, mixed_models=True)
run("mixed models   ", p3)
