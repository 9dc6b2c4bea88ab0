"""GPU check of the DLS-PnP path: standalone solver vs the oracle, RANSAC (estimator 3) vs the oracle, timing."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytheiasfm_amd import ransac, synth
from tests import oracle_lib as ol

rng = np.random.default_rng(5)
def rand_rot():
    q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
    return np.array([[1-2*(y*y+z*z),2*(x*y-w*z),2*(x*z+w*y)],[2*(x*y+w*z),1-2*(x*x+z*z),2*(y*z-w*x)],[2*(x*z-w*y),2*(y*z+w*x),1-2*(x*x+y*y)]])
fl, wl, gt = [], [], []
for k in range(64):
    n = [3, 4, 5, 10, 50, 200][k % 6]
    R = rand_rot(); t = rng.normal(size=3)
    Xc = np.c_[rng.uniform(-1, 1, (n, 2)), rng.uniform(2, 6, n)]
    fl.append(Xc[:, :2] / Xc[:, 2:3]); wl.append((Xc - t) @ R); gt.append((R, t))
ns, q, t = ransac.DlsPnp(fl, wl)
worst = 0.0; worst_gt = 0.0; bad = 0
for k in range(64):
    qo, to = ol.dls_pnp(fl[k], wl[k], call_index=k)
    if len(qo) != ns[k]:
        bad += 1; print("count mismatch", k, len(qo), ns[k]); continue
    for i in range(ns[k]):
        d = min(np.abs(q[k, i] - qo[j]).max() + np.abs(t[k, i] - to[j]).max() for j in range(len(qo)))
        worst = max(worst, d)
    R, tt = gt[k]
    best = 1e9
    for i in range(ns[k]):
        w, x, y, z = q[k, i]
        Rq = np.array([[1-2*(y*y+z*z),2*(x*y-w*z),2*(x*z+w*y)],[2*(x*y+w*z),1-2*(x*x+z*z),2*(y*z-w*x)],[2*(x*z-w*y),2*(y*z+w*x),1-2*(x*x+y*y)]])
        best = min(best, np.abs(Rq - R).max() + np.abs(t[k, i] - tt).max())
    worst_gt = max(worst_gt, best)
print("standalone: count mismatches", bad, "worst |gpu - oracle|", worst, "worst ground-truth error", worst_gt)
print("terms", ransac.dls_macaulay_terms(0, 2))

data, offsets, _ = synth.synth_ransac_v1(6, 300, "absolute", seed=0x5AC50301)
prm = ransac.RansacParameters(); prm.error_thresh = (2.0 / 1000.0) ** 2; prm.min_iterations = 64; prm.max_iterations = 64; prm.seed = 21
res = ransac.estimate_batch(ransac.EST_ABS_DLS, data, offsets, prm)
for i in range(6):
    pc = prm.to_c(); pc.seed = prm.seed + i
    o = ol.ransac_estimate(ransac.EST_ABS_DLS, data[offsets[i]:offsets[i + 1]], pc)
    same = np.array_equal(o["inlier_mask"], res["inlier_mask"][offsets[i]:offsets[i + 1]])
    print("problem", i, "inliers gpu", res["num_inliers"][i], "oracle", o["num_inliers"], "mask equal", same,
          "model diff", np.abs(res["models"][i] - o["model"]).max(), "scored", o["models_scored"])
# timing
data, offsets, _ = synth.synth_ransac_v1(200, 2000, "absolute", seed=0x5AC50302)
prm.min_iterations = prm.max_iterations = 1024
for rep in range(2):
    t0 = time.time()
    res = ransac.estimate_batch(ransac.EST_ABS_DLS, data, offsets, prm)
    dt = time.time() - t0
    print(f"200 x 1024 hypotheses: {dt:.3f} s, {200 * 1024 / dt:.0f} hyp/s, fit {res.get('time_fit_seconds')}, score {res.get('time_score_seconds')}, models scored {res.get('models_scored')}")
