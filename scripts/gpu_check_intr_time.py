"""Development check (GPU box): C2 with FOCAL_LENGTH | RADIAL_DISTORTION optimised: time per LM iteration."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import synth, ba
p = synth.ba_config("C2")
o = ba.default_options(); o.intrinsics_to_optimize = 1 | 8; o.max_num_iterations = 10
o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
h = ba.BaHandle(p.copy(), o); h.snapshot()
for rep in range(3):
    h.restore()
    t = time.time(); s, tr = h.run(trace_capacity=1); dt = time.time() - t
    print("INTR rep", rep, "iters", s.num_iterations, "%.3f ms/iter" % (1e3 * dt / max(1, s.num_iterations)),
          "lin %.3f solve %.3f back %.3f" % (1e3 * s.time_linearize / 10, 1e3 * s.time_solve_reduced / 10, 1e3 * s.time_backsub / 10), flush=True)
