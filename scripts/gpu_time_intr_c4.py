"""GPU box: LM iteration time at C4 with FOCAL_LENGTH | RADIAL_DISTORTION free, by number of intrinsics groups."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytheiasfm_amd import ba, synth
groups = [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else "8").split(",")]
for ng in groups:
    p = synth.synth_ba_v1(1000, 500000, seed=0xBA5E0004, num_groups=ng, mixed_models=ng > 1)
    if os.environ.get("MODEL"):   # every group on one camera model (1 = radial-tangential: ten intrinsics, 2 = fisheye: nine)
        k = {1: [1000.0, 1.02, 0.2, 960.0, 540.0, -0.1, 0.02, 0.001, 0.001, -0.002], 2: [600.0, 1.0, 0.1, 960.0, 540.0, 0.01, -0.002, 0.001, 0.0005]}[int(os.environ["MODEL"])]
        p.group_model[:] = int(os.environ["MODEL"]); p.intrinsics[:] = 0.0; p.intrinsics[:, :len(k)] = k
    o = ba.default_options(); o.max_num_iterations = 8
    o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 0.0
    o.use_inner_iterations = 0; o.intrinsics_to_optimize = int(os.environ.get("INTR", "0x11"), 0)
    t0 = time.perf_counter(); h = ba.BaHandle(p.copy(), o); tcold = time.perf_counter() - t0; h.close()   # (first create of the process: caches empty)
    t0 = time.perf_counter(); h = ba.BaHandle(p.copy(), o); tc = time.perf_counter() - t0
    h.reset(p); h.snapshot(); h.restore(); h.run(trace_capacity=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        h.restore(); s, _ = h.run(trace_capacity=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("C4", ng, "groups: create %.3f s (first of the process %.3f s)" % (tc, tcold), "%.3f ms / LM iteration" % (1e3 * dt / (3 * s.num_iterations)), "final cost %.6e" % s.final_cost, flush=True)
    h.close()
