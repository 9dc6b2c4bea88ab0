"""GPU box: LM iteration time at C4 with the reference's default use_inner_iterations = true; argument: intrinsics_to_optimize mask
(0x11 = FOCAL_LENGTH | RADIAL_DISTORTION: the pipelines' default)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytheiasfm_amd import ba, synth
p = synth.ba_config("C4")
o = ba.default_options(); o.max_num_iterations = 8
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 0.0
o.use_inner_iterations = 1
if len(sys.argv) > 1:
    o.intrinsics_to_optimize = int(sys.argv[1], 0)
h = ba.BaHandle(p.copy(), o)
h.reset(p); h.snapshot(); h.restore(); h.run(trace_capacity=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    h.restore(); s, _ = h.run(trace_capacity=1)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("C4 inner iterations: %.3f ms / LM iteration" % (1e3 * dt / (3 * s.num_iterations)), "final cost %.9e" % s.final_cost, flush=True)
h.close()
