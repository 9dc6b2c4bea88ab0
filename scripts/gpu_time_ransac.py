"""GPU box: one RANSAC leg at the C5 shape (pairs x 2000 correspondences x 4096 hypotheses), kernel-time split.
usage: gpu_time_ransac.py <five_point|sqpnp|dls|kneip|p4pf|...> [pairs]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytheiasfm_amd import ransac, synth

leg = sys.argv[1] if len(sys.argv) > 1 else "five_point"
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
est, kind, thr = {"five_point": (ransac.EST_RELATIVE_POSE, "relative", (2.0 / 1000.0) ** 2),
                  "sqpnp": (ransac.EST_ABS_SQPNP, "absolute", (4.0 / 1000.0) ** 2),
                  "dls": (ransac.EST_ABS_DLS, "absolute", (4.0 / 1000.0) ** 2),
                  "fundamental": (ransac.EST_FUNDAMENTAL_MATRIX, "relative", (2.0 / 1000.0) ** 2),
                  "homography": (ransac.EST_HOMOGRAPHY, "relative", (2.0 / 1000.0) ** 2),
                  "essential": (ransac.EST_ESSENTIAL_MATRIX, "relative", (2.0 / 1000.0) ** 2),
                  "p4pf": (ransac.EST_UNCALIBRATED_ABSOLUTE_POSE, "absolute", (4.0 / 1000.0) ** 2),
                  "upnp": (ransac.EST_RIGID_TRANSFORMATION_2D3D, "absolute", (4.0 / 1000.0) ** 2),
                  "p4pfr": (ransac.EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE, "absolute", 4.0 ** 2),
                  "kneip": (ransac.EST_ABS_KNEIP, "absolute", (4.0 / 1000.0) ** 2)}[leg]
data, offsets, TRUTH = synth.synth_ransac_v1(NP, 2000, kind, seed=0x5AC50005)
if leg == "upnp":
    data = ransac.central_correspondence_rows(data)   # [u v X Y Z] seen by identity pinhole cameras (the central overload)
ep = None
if leg == "p4pfr":
    data = ransac.radial_dist_correspondence_rows(ransac.shift_world_along_optical_axis(data, offsets, TRUTH["R"], 2.0), 1000.0, -1e-7)   # pixels of a camera with focal length 1000, distortion -1e-7
    ep = np.array([2000.0, 100.0, -1e-5, -1e-9, 0.0])
p = ransac.RansacParameters(); p.error_thresh = thr; p.min_iterations = 4096; p.max_iterations = 4096; p.seed = 1
ransac.estimate_batch(est, data[:offsets[8]], offsets[:9], p, ep)
t0 = time.perf_counter()
res = ransac.estimate_batch(est, data, offsets, p, ep)
dt = time.perf_counter() - t0
h = res["hypotheses_evaluated"]
print(f"{leg}: {NP} pairs, {h} hypotheses in {dt:.3f} s = {h / dt / 1e6:.2f} M hyp/s; fit {res['time_fit_seconds']:.3f} s "
      f"score {res['time_score_seconds']:.3f} s; checksum inliers {int(res['num_inliers'].sum())}")
