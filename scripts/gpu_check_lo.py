"""Development check (GPU box): LO-RANSAC control flow, HIP vs oracle, per problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ransac, synth
from tests import oracle_lib as ol
data, offsets, truth = synth.synth_ransac_v1(10, 300, "absolute", seed=0x5AC50700, noise_px=1.0)
p = ransac.RansacParameters(); p.error_thresh = (4 / 1000.0) ** 2; p.use_mle = True; p.seed = 66
p.use_lo = True; p.lo_start_iterations = 5; p.min_iterations = 50; p.failure_probability = 0.001
res = ransac.estimate_batch(2, data, offsets, p)
for i in range(10):
    pc = p.to_c(); pc.seed = 66 + i
    o = ol.ransac_estimate(2, data[offsets[i]:offsets[i + 1]], pc)
    nlo = ol.rlib().oracle_last_lo_iterations()
    print(i, "iters", o["num_iterations"], res["num_iterations"][i], "nlo", nlo, res["num_lo_iterations"][i], "ninl", o["num_inliers"], res["num_inliers"][i],
          "dmodel %.2e" % np.abs(o["model"][:12] - res["models"][i][:12]).max())
if os.environ.get("LO_DEBUG_PROBLEM"):
    k = int(os.environ["LO_DEBUG_PROBLEM"])
    os.environ["ORACLE_RANSAC_DEBUG"] = "1"; os.environ["THEIA_HIP_RANSAC_DEBUG"] = "1"
    pc = p.to_c(); pc.seed = 66 + k
    ol.ransac_estimate(2, data[offsets[k]:offsets[k + 1]], pc)
    q = ransac.RansacParameters(); q.error_thresh = p.error_thresh; q.use_mle = True; q.seed = 66 + k
    q.use_lo = True; q.lo_start_iterations = 5; q.min_iterations = 50; q.failure_probability = 0.001
    ransac.estimate_batch(2, data[offsets[k]:offsets[k + 1]], np.array([0, offsets[k + 1] - offsets[k]]), q)
