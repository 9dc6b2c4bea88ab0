import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import sfm, synth, ba
p = synth.ba_config("C4")
r = sfm.Reconstruction.from_flat(p)
o = sfm.BundleAdjustmentOptions(); o.max_num_iterations = 5; o.use_inner_iterations = False
cache = sfm.set_problem_cache(1)
for i in range(4):
    r.cam_ext += 1e-4
    t0 = time.perf_counter(); s = sfm.BundleAdjustReconstruction(o, r); dt = time.perf_counter() - t0
    print(f"call {i}: {dt*1e3:.1f} ms  iterations {s.num_iterations}  hits {cache.hits} misses {cache.misses}", flush=True)
sfm.set_problem_cache(0)
for i in range(2):
    r.cam_ext += 1e-4
    t0 = time.perf_counter(); s = sfm.BundleAdjustReconstruction(o, r); dt = time.perf_counter() - t0
    print(f"uncached {i}: {dt*1e3:.1f} ms", flush=True)
t0 = time.perf_counter(); flat = sfm._flatten(r, r.ViewIds(), r.TrackIds(), options=o); t1 = time.perf_counter(); k = ba.problem_fingerprint(flat, o.to_c()); t2 = time.perf_counter()
print(f"flatten {1e3*(t1-t0):.1f} ms  fingerprint {1e3*(t2-t1):.1f} ms")
