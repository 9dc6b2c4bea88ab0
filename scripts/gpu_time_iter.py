"""GPU box: LM iteration time at C4 (headline configuration) and the per-phase HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytheiasfm_amd import ba, synth
p = synth.ba_config("C4")
o = ba.default_options(); o.max_num_iterations = 8
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 0.0; o.use_inner_iterations = 0
h = ba.BaHandle(p.copy(), o)
h.reset(p); h.snapshot(); h.restore(); h.run(trace_capacity=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 0
for _ in range(5):
    h.restore(); s, _ = h.run(trace_capacity=1); n += s.num_iterations
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("C4: %.4f ms / LM iteration, final cost %.12e" % (1e3 * dt / n, s.final_cost), flush=True)
h.close()
