"""Development check (GPU box): C4-size BA (1k views / 500k tracks, mixed models) timing + properties."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import synth, ba
t = time.time(); p = synth.ba_config("C4"); print("gen %.1fs" % (time.time() - t), p.cam_ext.shape, p.points.shape, p.obs_uv.shape, flush=True)
o = ba.default_options()
t = time.time(); h = ba.BaHandle(p.copy(), o); print("create %.2fs" % (time.time() - t), flush=True)
for rep in range(2):
    h.reset(p)
    t = time.time(); s, tr = h.run(); dt = time.time() - t
    print("rep", rep, "success", s.success, "term", s.termination_type, "iters", s.num_iterations, "cost %.4e -> %.4e" % (s.initial_cost, s.final_cost),
          "time %.3f s -> %.1f it/s" % (dt, s.num_iterations / dt), "lin %.4f solve %.4f back %.4f" % (s.time_linearize, s.time_solve_reduced, s.time_backsub), flush=True)
print(" trace cost", tr.cost[:10]); print(" acc", tr.accepted[:10])
