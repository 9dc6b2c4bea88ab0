"""GPU box: inverse-depth BA at 400 views / 60 000 tracks (n = 2400), tile-sparse reduced solve against the dense schedule
(THEIA_HIP_INVDEPTH_DENSE=1).  usage: gpu_time_invdepth.py [views] [tracks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ba
from tests import invdepth as idp
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
p = idp.make(nv, nt, seed=3)
o = ba.default_options(); o.max_num_iterations = 6; o.use_inner_iterations = 0
o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 0.0
for mode in ("sparse", "dense"):
    if mode == "dense": os.environ["THEIA_HIP_INVDEPTH_DENSE"] = "1"
    else: os.environ.pop("THEIA_HIP_INVDEPTH_DENSE", None)
    h = ba.BaHandle(p.copy(), o)
    h.run()                      # warm-up
    h.reset(p); t0 = time.perf_counter(); s, _ = h.run(); dt = time.perf_counter() - t0
    print(f"{mode}: {nv} views / {nt} tracks / {p.obs_uv.shape[0]} observations: {1e3 * dt / max(1, s.num_iterations):.2f} ms per LM iteration "
          f"({s.num_iterations} iterations, final cost {s.final_cost:.6g})", flush=True)
    h.close()
