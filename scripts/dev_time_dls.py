"""GPU box (development): wall time of one RANSAC leg at 256 pairs x 2000 correspondences x 4096 hypotheses, three repeats.
usage: dev_time_dls.py [dls | upnp | p4pf | p4pfr | five_point] [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import ransac, synth
leg = sys.argv[1] if len(sys.argv) > 1 else "dls"
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
est, kind, thr = {"dls": (ransac.EST_ABS_DLS, "absolute", (4 / 1000.0) ** 2), "upnp": (ransac.EST_RIGID_TRANSFORMATION_2D3D, "absolute", (4 / 1000.0) ** 2),
                  "five_point": (ransac.EST_RELATIVE_POSE, "relative", (2 / 1000.0) ** 2),
                  "p4pf": (ransac.EST_UNCALIBRATED_ABSOLUTE_POSE, "absolute", 4.0 ** 2),
                  "p4pfr": (ransac.EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE, "absolute", 4.0 ** 2)}[leg]
data, offsets, TRUTH = synth.synth_ransac_v1(npairs, 2000, kind, seed=0x5AC50005)
if leg == "upnp": data = ransac.central_correspondence_rows(data)
if leg == "p4pf":   # pixels: focal length 1000
    data = data.copy(); data[:, :2] *= 1000.0
ep = None
if leg == "p4pfr":
    import numpy as np
    data = ransac.radial_dist_correspondence_rows(ransac.shift_world_along_optical_axis(data, offsets, TRUTH["R"], 2.0), 1000.0, -1e-7)
    ep = np.array([2000.0, 100.0, -1e-5, -1e-9, 0.0])
p = ransac.RansacParameters(); p.error_thresh = thr; p.min_iterations = 4096; p.max_iterations = 4096; p.seed = 1
ransac.estimate_batch(est, data, offsets, p, ep)
ts = []
for r in range(3):
    t0 = time.perf_counter(); ransac.estimate_batch(est, data, offsets, p, ep); ts.append(time.perf_counter() - t0)
print("%s leg: %.4f s best of 3 = %.3f M hypotheses/s" % (leg, min(ts), npairs * 4096 / min(ts) / 1e6))
