"""GPU box: LM iteration time of C2 / C4 with the pipelines' default intrinsics subset (FOCAL_LENGTH | RADIAL_DISTORTION)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytheiasfm_amd import ba, synth
for name in ("C2", "C4"):
    p = synth.ba_config(name)
    for intr in (0, 0x01 | 0x10):
        o = ba.default_options(); o.max_num_iterations = 8
        o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 0.0
        o.use_inner_iterations = 0; o.intrinsics_to_optimize = intr
        t0 = time.perf_counter(); h = ba.BaHandle(p.copy(), o); tc = time.perf_counter() - t0
        h.reset(p); h.snapshot(); h.restore(); h.run(trace_capacity=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            h.restore(); s, _ = h.run(trace_capacity=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(name, "intrinsics 0x%02x" % intr, "create %.2f s" % tc, "%.3f ms / LM iteration" % (1e3 * dt / (3 * s.num_iterations)), "final cost %.6e" % s.final_cost, flush=True)
        h.close()
