"""Development check (GPU box): per-kernel-group HIP-event timing of C4-size BA iterations (THEIA_HIP_PHASE_TIMING)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["THEIA_HIP_PHASE_TIMING"] = "1"
import numpy as np
from pytheiasfm_amd import synth, ba
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
p = synth.ba_config(cfg)
o = ba.default_options(); o.max_num_iterations = 8
o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
h = ba.BaHandle(p.copy(), o)
for rep in range(3):
    h.reset(p)
    t = time.time(); s, tr = h.run(); dt = time.time() - t
    n = max(1, s.num_linearize_launches)
    print(cfg, "rep", rep, "iters", s.num_iterations, "cost %.6e" % s.final_cost, "wall/it %.3f ms" % (1e3 * dt / s.num_iterations),
          "lin+schur kernels %.1f us  lin phase %.1f us  solve %.1f us  backsub %.1f us" %
          (1e6 * s.time_kernel_linearize / n, 1e6 * s.time_linearize / n, 1e6 * s.time_solve_reduced / n, 1e6 * s.time_backsub / n), flush=True)
