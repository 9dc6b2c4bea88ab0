"""GPU box: LM iteration and handle creation time at C4 with intrinsics optimised, by number of intrinsics groups
(THEIA_HIP_INTR_PAIRS=1 selects the per-pair lists for the group blocks instead of the track sums)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytheiasfm_amd import ba, synth
for ng in [int(x) for x in os.environ.get("INTR_GROUPS", "1,2,8").split(",")]:
    p = synth.ba_config("C4", num_groups=ng)
    o = ba.default_options(); o.max_num_iterations = 8
    o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 0.0
    o.use_inner_iterations = 0; o.intrinsics_to_optimize = 0x01 | 0x10
    t0 = time.perf_counter(); h = ba.BaHandle(p.copy(), o); tc = time.perf_counter() - t0
    h.reset(p); h.snapshot(); h.restore(); h.run(trace_capacity=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        h.restore(); s, _ = h.run(trace_capacity=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("C4 groups %d" % ng, "create %.2f s" % tc, "%.3f ms / LM iteration" % (1e3 * dt / (3 * s.num_iterations)), "final cost %.9e" % s.final_cost, flush=True)
    h.close()
