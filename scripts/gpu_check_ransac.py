"""Development check (GPU box): HIP RANSAC vs oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import oracle_lib as ol
from pytheiasfm_amd import synth, ransac

# minimal solvers: bit-compare
data, offsets, truth = synth.synth_ransac_v1(64, 5, "relative", seed=3, inlier_lo=1.0, inlier_hi=1.0, noise_px=0.0)
corr = data.reshape(64, 5, 4)
ns, E = ransac.FivePointRelativePose(corr[:, :, :2], corr[:, :, 2:])
bad = 0
for i in range(64):
    Eo = ol.five_point(corr[i])
    if len(Eo) != ns[i] or not np.array_equal(Eo, E[i, :ns[i]]):
        bad += 1
        if bad < 3: print("5pt mismatch", i, len(Eo), ns[i], np.abs(Eo - E[i, :len(Eo)]).max() if len(Eo) == ns[i] else None)
print("five point: bitwise mismatches", bad, "of 64; mean solutions", ns.mean())
dataa, offa, tra = synth.synth_ransac_v1(64, 3, "absolute", seed=4, inlier_lo=1.0, inlier_hi=1.0, noise_px=0.0)
ca = dataa.reshape(64, 3, 5)
ns3, R3, t3 = ransac.PoseFromThreePoints(ca[:, :, :2], ca[:, :, 2:])
bad = 0
for i in range(64):
    Ro, to = ol.p3p(ca[i])
    same = len(Ro) == ns3[i] and np.array_equal(Ro, R3[i, :ns3[i]], equal_nan=True) and np.array_equal(to, t3[i, :ns3[i]], equal_nan=True)
    bad += 0 if same else 1
print("p3p: bitwise mismatches", bad, "of 64")

for kind, est, thr in (("relative", 0, (2 / 1000.0) ** 2), ("relative", 1, (2 / 1000.0) ** 2), ("absolute", 2, (4 / 1000.0) ** 2)):
    for use_mle in (0, 1):
        data, offsets, truth = synth.synth_ransac_v1(8, 500, kind, seed=77)
        p = ransac.RansacParameters(); p.error_thresh = thr; p.use_mle = bool(use_mle); p.seed = 65
        t = time.time(); res = ransac.estimate_batch(est, data, offsets, p); dt = time.time() - t
        nbad = 0
        ml = {0: 21, 1: 9, 2: 12}[est]
        for i in range(8):
            pc = p.to_c(); pc.seed = 65 + i
            o = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], pc)
            ok = np.array_equal(o["inlier_mask"], res["inlier_mask"][offsets[i]:offsets[i + 1]]) and o["num_iterations"] == res["num_iterations"][i] \
                and np.array_equal(o["model"][:ml], res["models"][i][:ml], equal_nan=True)
            nbad += 0 if ok else 1
            if not ok:
                print("  mismatch", i, o["num_iterations"], res["num_iterations"][i], o["num_inliers"], res["num_inliers"][i])
        print(f"est {est} mle {use_mle}: mismatching problems {nbad}/8; iters {res['num_iterations']} inliers {res['num_inliers']} true {truth['inlier'].sum(1)} time {dt:.3f}")
# (the RANSAC throughput bench lives in bench.py: ransac_block)
