import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_create_fuzz_gpu import draw, solve, reorder
for seed in (17,):
    p, o, rng = draw(seed)
    print("seed", seed, "intr", hex(o.intrinsics_to_optimize), "inner", o.use_inner_iterations, "groups", len(p.group_model), "models", p.group_model,
          "longest", np.bincount(p.obs_pt).max(), "nobs", len(p.obs_pt))
    a = solve(p, o)[1]; b = solve(p, o)[1]
    print("p vs p", np.array_equal(a, b), a - b)
