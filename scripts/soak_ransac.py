"""GPU box: random RANSAC batches, the library's inlier sets / iteration counts / models against the CPU oracle's (bit-identical is
the bar).  A soak, not a test.  Usage: soak_ransac.py [batches per estimator] [first seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import ransac, synth
from tests import oracle_lib as ol

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
EST = [(0, "relative", (2 / 1000.0) ** 2, 21), (1, "relative", (2 / 1000.0) ** 2, 9), (2, "absolute", (4 / 1000.0) ** 2, 12),
       (3, "absolute", (4 / 1000.0) ** 2, 12), (4, "absolute", (4 / 1000.0) ** 2, 12)]
bad = 0; total = 0
t0 = time.time()
for est, kind, thr, mlen in EST:
    diff_est = 0; n_est = 0
    for k in range(nb):
        seed = seed0 + k
        rng = np.random.default_rng(0x7A5AC000 + 977 * est + seed)
        npairs = int(rng.integers(2, 7)); ncorr = int(rng.integers(40, 500))
        data, offsets, truth = synth.synth_ransac_v1(npairs, ncorr, kind, seed=0x7A5AC000 + 31 * est + seed)
        p = ransac.RansacParameters(); p.error_thresh = thr * float(rng.choice([0.5, 1.0, 2.0]))
        p.use_mle = bool(rng.integers(0, 2)); p.seed = int(rng.integers(0, 2 ** 31 - 1))
        p.ransac_type = int(rng.choice([0, 0, 1, 2]))                      # RANSAC / PROSAC / LMED
        p.min_iterations = int(rng.choice([50, 100, 300])); p.max_iterations = int(rng.choice([300, 1000, 2 ** 31 - 1]))
        p.failure_probability = float(rng.choice([0.01, 0.001]))
        res = ransac.estimate_batch(est, data, offsets, p)
        for i in range(npairs):
            pc = p.to_c(); pc.seed = (p.seed + i) & 0xFFFFFFFF
            o = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], pc)
            sl = slice(offsets[i], offsets[i + 1])
            same = (np.array_equal(o["inlier_mask"], res["inlier_mask"][sl]) and o["num_iterations"] == res["num_iterations"][i]
                    and np.array_equal(o["model"][:mlen], res["models"][i][:mlen], equal_nan=True))
            n_est += 1; diff_est += not same
            if not same:
                print("DIFFERS: est %d seed %d pair %d type %d mle %d iterations %d / %d inliers %d / %d" % (
                    est, seed, i, p.ransac_type, p.use_mle, o["num_iterations"], res["num_iterations"][i], o["num_inliers"], res["num_inliers"][i]), flush=True)
    print("estimator %d: %d problems, %d differ" % (est, n_est, diff_est), flush=True)
    bad += diff_est; total += n_est
print("soak: %d problems, %d differ, %.0f s" % (total, bad, time.time() - t0))
