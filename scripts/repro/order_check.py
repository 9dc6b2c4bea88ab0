"""GPU box: does the observation order of the input change the solve?  (development check)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytheiasfm_amd import _capi as capi, ba, synth
const = len(sys.argv) > 1 and sys.argv[1] == "const"
p = synth.ba_config("C2")
print("longest track", np.bincount(p.obs_pt).max(), "observations", len(p.obs_pt))
if const:
    p.cam_const = np.zeros(p.cam_ext.shape[0], np.uint8); p.cam_const[[0, 5]] = 3
    p.point_const = np.zeros(p.points.shape[0], np.uint8); p.point_const[::7] = 1
cnt = np.bincount(p.obs_pt)
within = np.arange(len(p.obs_pt)) - np.repeat(np.cumsum(cnt) - cnt, cnt)
order = np.lexsort((p.obs_pt, within))
q = capi.FlatProblem(p.cam_ext.copy(), p.intrinsics.copy(), p.group_model, p.cam_group, p.points.copy(), p.obs_uv[order],
                     p.obs_cam[order], p.obs_pt[order], p.cam_const, p.group_const, p.point_const)
o = ba.default_options(); o.max_num_iterations = 4
def run(x):
    y = x.copy(); s, t = ba.solve(y, o); return t.cost[:t.size].copy(), t.step_norm[:t.size].copy(), y
a = run(p); a2 = run(p); b = run(q); b2 = run(q)
print("p vs p   ", np.array_equal(a[0], a2[0]), np.array_equal(a[1], a2[1]))
print("q vs q   ", np.array_equal(b[0], b2[0]), np.array_equal(b[1], b2[1]))
print("p vs q   ", a[0] - b[0], a[1] - b[1], np.abs(a[2].points - b[2].points).max())
