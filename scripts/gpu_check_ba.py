"""Development check (GPU box): HIP BA path vs oracle on C1/C2."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import oracle_lib as ol
from pytheiasfm_amd import synth, ba

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
p = synth.ba_config(cfg)
print("problem", cfg, p.cam_ext.shape, p.points.shape, p.obs_uv.shape, flush=True)
o = ba.default_options()
oo = ol.default_options()
h = ba.BaHandle(p.copy(), o)
cost, r, jc, jp, valid = h.evaluate()
ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
print("cost", cost, ocost, "rel", abs(cost - ocost) / ocost)
print("res max abs diff", np.abs(r - orr).max(), "Jc", np.abs(jc - ojc).max() / np.abs(ojc).max(), "Jp", np.abs(jp - ojp).max() / np.abs(ojp).max())
if p.cam_ext.shape[0] <= 200:
    S, rhs = h.reduced_system(1e4)
    So, rhso = ol.reduced_system(p, oo, 1e4)
    print("S rel", np.abs(S - So).max() / np.abs(So).max(), "rhs rel", np.abs(rhs - rhso).max() / np.abs(rhso).max())
    x = np.linalg.solve(So, rhso)
t = time.time(); s, tr = h.run(); dt = time.time() - t
print("GPU: success", s.success, "term", s.termination_type, "iters", s.num_iterations, "cost", s.initial_cost, "->", s.final_cost, "time", dt)
print(" phases: lin %.4f solve %.4f backsub %.4f" % (s.time_linearize, s.time_solve_reduced, s.time_backsub))
print(" trace cost", tr.cost[:8]); print(" radius", tr.radius[:8]); print(" acc", tr.accepted[:8]); print(" gmax", tr.gradient_max_norm[:8]); print(" step", tr.step_norm[:8])
pg = h.download(p.copy())
po = p.copy()
t = time.time(); so, tro = ol.solve(po, oo); dto = time.time() - t
print("ORACLE: success", so.success, "term", so.termination_type, "iters", so.num_iterations, "cost", so.initial_cost, "->", so.final_cost, "time", dto)
print(" trace cost", tro.cost[:8]); print(" radius", tro.radius[:8]); print(" acc", tro.accepted[:8]); print(" gmax", tro.gradient_max_norm[:8]); print(" step", tro.step_norm[:8])
print("param diff cams", np.abs(pg.cam_ext - po.cam_ext).max(), "pts", np.abs(pg.points - po.points).max())
# timing: repeated runs on resident problem
for rep in range(3):
    h.reset(p)
    t = time.time(); s, tr = h.run(); dt = time.time() - t
    print("rep", rep, "iters", s.num_iterations, "time %.4f s -> %.1f it/s" % (dt, s.num_iterations / dt), "lin %.4f solve %.4f back %.4f" % (s.time_linearize, s.time_solve_reduced, s.time_backsub))
