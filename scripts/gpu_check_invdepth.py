"""GPU check: inverse-depth parametrisation, device vs the dense-LM oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pytheiasfm_amd import ba
from tests import invdepth as idp, oracle_lib as ol

def run(name, p, **kw):
    res = []
    for mod in (ba, ol):
        q = p.copy(); o = mod.default_options(); o.max_num_iterations = 15; o.use_inner_iterations = 0
        for k, v in kw.items(): setattr(o, k, v)
        s, tr = (ba.solve(q, o) if mod is ba else ol.solve_inverse_depth(q, o))
        res.append((s, tr, q))
    (sg, tg, qg), (so, to, qo) = res
    n = min(len(tg.cost), len(to.cost))
    rel = max(abs(tg.cost[k] - to.cost[k]) / max(1e-300, abs(to.cost[k])) for k in range(n)) if n else 0
    print(f"{name}: iters {sg.num_iterations}/{so.num_iterations} final {sg.final_cost:.9e}/{so.final_cost:.9e} trace rel {rel:.2e} "
          f"acc {list(tg.accepted[:n]) == list(to.accepted[:n])} cam {np.abs(qg.cam_ext - qo.cam_ext).max():.2e} rho {np.abs(qg.point_inverse_depth - qo.point_inverse_depth).max():.2e}")
    if rel > 1e-6:
        for k in range(n): print("   ", k, tg.cost[k], to.cost[k], tg.accepted[k], to.accepted[k], tg.gradient_max_norm[k], to.gradient_max_norm[k])

p = idp.make()
run("plain      ", p)
run("huber      ", p, loss_function_type=1, robust_loss_width=2.0)
p2 = idp.make(12, 400, seed=9)
p2.cam_const = np.zeros(12, dtype=np.uint8); p2.cam_const[0] = 3; p2.cam_const[5] = 1
p2.point_const = (np.arange(400) % 11 == 0).astype(np.uint8)
run("const parts", p2)
run("const orient", p, constant_camera_orientation=1)
