"""GPU box (development): wall time of the DLS RANSAC leg at 256 pairs x 2000 correspondences x 4096 hypotheses, three repeats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import ransac, synth
est = {"dls": ransac.EST_ABS_DLS}[sys.argv[1] if len(sys.argv) > 1 else "dls"]
data, offsets, _ = synth.synth_ransac_v1(256, 2000, "absolute", seed=0x5AC50005)
p = ransac.RansacParameters(); p.error_thresh = (4 / 1000.0) ** 2; p.min_iterations = 4096; p.max_iterations = 4096; p.seed = 1
ransac.estimate_batch(est, data, offsets, p)
ts = []
for r in range(3):
    t0 = time.perf_counter(); ransac.estimate_batch(est, data, offsets, p); ts.append(time.perf_counter() - t0)
print("dls leg: %.4f s best of 3 = %.3f M hypotheses/s" % (min(ts), 256 * 4096 / min(ts) / 1e6))
