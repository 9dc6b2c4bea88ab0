"""GPU box: the drop-in call (theia_hip_ba_solve from host arrays) at C4; THEIA_HIP_CREATE_TIMING=1 prints its phases."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import ba, synth
p = synth.ba_config("C4")
o = ba.default_options(); o.max_num_iterations = 25; o.use_inner_iterations = 0
ba.solve(p.copy(), o)
for _ in range(3):
    q = p.copy()
    t0 = time.perf_counter(); s, tr = ba.solve(q, o); dt = time.perf_counter() - t0
    print("C4 one-shot solve %.1f ms, %d iterations" % (1e3 * dt, s.num_iterations), flush=True)
