import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pytheiasfm_amd import ransac, synth
data, offsets, truth = synth.synth_ransac_v1(64, 5, "relative", seed=3, inlier_lo=1.0, inlier_hi=1.0, noise_px=0.0)
corr = data.reshape(64, 5, 4)
t = time.time(); ns, E = ransac.FivePointRelativePose(corr[:, :, :2], corr[:, :, 2:]); print("standalone five point", ns[:8], time.time() - t, flush=True)
data, offsets, _ = synth.synth_ransac_v1(4, 200, "relative", seed=0x5AC50001)
prm = ransac.RansacParameters(); prm.error_thresh = (2.0 / 1000.0) ** 2; prm.min_iterations = 64; prm.max_iterations = 64; prm.seed = 65
t = time.time(); res = ransac.estimate_batch(0, data, offsets, prm); print("batch est 0", res["num_inliers"], time.time() - t, flush=True)
t = time.time(); res = ransac.estimate_batch(1, data, offsets, prm); print("batch est 1", res["num_inliers"], time.time() - t, flush=True)
prm.use_lo = 1; prm.lo_start_iterations = 5
t = time.time(); res = ransac.estimate_batch(0, data, offsets, prm); print("batch est 0 LO", res["num_inliers"], time.time() - t, flush=True)
