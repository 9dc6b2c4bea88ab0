"""Development check (GPU box): C2 with intrinsics optimisation (pipeline default FOCAL_LENGTH | RADIAL_DISTORTION)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import synth, ba
p = synth.ba_config("C2")
for mask in (0, 0x11):
    o = ba.default_options(); o.intrinsics_to_optimize = mask; o.max_num_iterations = 8
    o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
    h = ba.BaHandle(p.copy(), o)
    os.environ["THEIA_HIP_PHASE_TIMING"] = "1"
    for rep in range(2):
        h.reset(p); t = time.time(); s, tr = h.run(); dt = time.time() - t
    print("intr mask %#x: %d iters %.2f ms/iter  lin %.3f solve %.3f back %.3f ms  cost %.4e -> %.4e" % (
        mask, s.num_iterations, 1e3 * dt / s.num_iterations, 1e3 * s.time_linearize / s.num_linearize_launches,
        1e3 * s.time_solve_reduced / s.num_linearize_launches, 1e3 * s.time_backsub / s.num_linearize_launches, s.initial_cost, s.final_cost))
    h.close()
