"""GPU box: repeat the thread-pool estimator scenario and report which job / field deviates from the sequential result."""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ransac, synth
THR = [(2 / 1000.0) ** 2, 0, (4 / 1000.0) ** 2]
rel, orel, _ = synth.synth_ransac_v1(8, 300, "relative", seed=0x5AC50901)
ab, oab, _ = synth.synth_ransac_v1(8, 300, "absolute", seed=0x5AC50902)
five, _, _ = synth.synth_ransac_v1(200, 5, "relative", seed=0x5AC50903, inlier_lo=1.0, inlier_hi=1.0)
c5 = five.reshape(200, 5, 4)
prel = ransac.RansacParameters(); prel.error_thresh = THR[0]; prel.seed = 9; prel.min_iterations = 200; prel.max_iterations = 400
pabs = ransac.RansacParameters(); pabs.error_thresh = THR[2]; pabs.seed = 10; pabs.min_iterations = 100; pabs.max_iterations = 200
plo = ransac.RansacParameters(); plo.error_thresh = THR[2]; plo.seed = 11; plo.use_lo = True; plo.lo_start_iterations = 5
plo.min_iterations = 50; plo.max_iterations = 100
jobs = [
    lambda: ransac.estimate_batch(ransac.EST_RELATIVE_POSE, rel, orel, prel),
    lambda: ransac.estimate_batch(ransac.EST_ABS_SQPNP, ab, oab, pabs),
    lambda: ransac.estimate_batch(ransac.EST_ABS_DLS, ab, oab, pabs),
    lambda: ransac.estimate_batch(ransac.EST_ABS_KNEIP, ab, oab, plo),
    lambda: ransac.estimate_batch(ransac.EST_ESSENTIAL_MATRIX, rel, orel, prel),
    lambda: ransac.FivePointRelativePose(c5[:, :, :2], c5[:, :, 2:]),
]
names = ["relpose", "sqpnp", "dls", "kneip+lo", "essential", "fivepoint"]
KEYS = ("success", "models", "num_inliers", "inlier_mask", "num_iterations", "num_lo_iterations")
def diff(a, b):
    if isinstance(a, dict):
        return [k for k in KEYS if not np.array_equal(a[k], b[k])]
    return [i for i, (x, y) in enumerate(zip(a, b)) if not np.array_equal(x, y, equal_nan=True)]
ref = [j() for j in jobs]
ref2 = [j() for j in jobs]
print("sequential repeat diffs:", [diff(a, b) for a, b in zip(ref, ref2)])
which = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(len(jobs)))
for rep in range(12):
    out = {}; errs = []
    def worker(k):
        try:
            out[k] = jobs[k]()
        except Exception as e:
            errs.append((k, e))
    th = [threading.Thread(target=worker, args=(k,)) for k in which]
    for t in th: t.start()
    for t in th: t.join()
    bad = {names[k]: diff(ref[k], out[k]) for k in which if k in out and diff(ref[k], out[k])}
    print(rep, "errs", errs, "diffs", bad, flush=True)
