"""GPU box: handle creation time at C4 (host-side plan), with THEIA_HIP_CREATE_TIMING=1 the stages are printed."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import ba, synth
p = synth.ba_config("C4")
o = ba.default_options(); o.max_num_iterations = 8
o.use_inner_iterations = int(os.environ.get("INNER", "0"))      # (the reference's default is on: two more index lists)
if len(sys.argv) > 1:
    o.intrinsics_to_optimize = int(sys.argv[1], 0)
h = ba.BaHandle(p.copy(), o); h.close()
for _ in range(3):
    pc = p.copy()
    t0 = time.perf_counter(); h = ba.BaHandle(pc, o); dt = time.perf_counter() - t0; h.close()
    print("C4 handle creation %.1f ms (use_inner_iterations = %d)" % (1e3 * dt, o.use_inner_iterations), flush=True)
