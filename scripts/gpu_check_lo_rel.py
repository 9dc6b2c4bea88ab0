"""Development check (GPU box): LO-RANSAC of the relative-pose estimator on 256 pairs x 2000 correspondences --
wall time with / without use_lo, LO event counts, and the CPU oracle on a few pairs for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ransac, synth
from tests import oracle_lib as ol
P, N = 256, 2000
data, offsets, truth = synth.synth_ransac_v1(P, N, "relative", seed=0x5AC53000)
def params(lo):
    p = ransac.RansacParameters(); p.error_thresh = (2.0 / 1000.0) ** 2; p.seed = 5; p.use_mle = True
    p.min_iterations = 1024; p.max_iterations = 1024; p.use_lo = lo; p.lo_start_iterations = 10
    return p
for lo in (False, True, False, True):
    t = time.time(); res = ransac.estimate_batch(0, data, offsets, params(lo)); dt = time.time() - t
    print("use_lo", lo, "%.3f s" % dt, "LO events", int(res["num_lo_iterations"].sum()), "inliers", int(res["num_inliers"].sum()), flush=True)
t = time.time()
for i in range(4):
    pc = params(True).to_c(); pc.seed = 5 + i
    ol.ransac_estimate(0, data[offsets[i]:offsets[i + 1]], pc)
print("oracle with use_lo: %.3f s per pair" % ((time.time() - t) / 4))
