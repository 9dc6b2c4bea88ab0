import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pytheiasfm_amd import ba, sfm
from tests import fountain as ft, oracle_lib as ol
d = ft.load()
rng = np.random.default_rng(1105)
cam = d["cam_ext"].copy(); pts = d["points"].copy()
cam[:, :3] += rng.normal(scale=2e-3, size=(11, 3)); cam[:, 3:] += rng.normal(scale=1e-3, size=(11, 3))
pts[:, :3] += rng.normal(scale=5e-3, size=(len(pts), 3))
intr = int(sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION)
def opt(mod):
    o = mod.default_options(); o.max_num_iterations = 25; o.intrinsics_to_optimize = intr; return o
pg, po = ft.flat_problem(d, cam.copy(), pts.copy()), ft.flat_problem(d, cam.copy(), pts.copy())   # (FlatProblem wraps, it does not copy)
sg, tg = ba.solve(pg, opt(ba)); so, to = ol.solve(po, opt(ol))
print(sg.num_iterations, so.num_iterations, sg.termination_type, so.termination_type)
for k in range(min(len(tg.cost), len(to.cost))):
    print(k, "%.10e %.10e  rel %.2e  acc %d %d  step %.3e %.3e radius %.3e %.3e" % (tg.cost[k], to.cost[k], abs(tg.cost[k]-to.cost[k])/to.cost[k], tg.accepted[k], to.accepted[k], tg.step_norm[k], to.step_norm[k], tg.radius[k], to.radius[k]))
print("cam diff", np.abs(pg.cam_ext - po.cam_ext).max(), "pts diff", np.abs(pg.points - po.points).max(), "intr", pg.intrinsics[0][:7], po.intrinsics[0][:7])
