"""Development timing (GPU box): one RANSAC estimator at 200 pairs x 2000 correspondences x 4096 hypotheses."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytheiasfm_amd import ransac, synth
est = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kind = "relative" if est in (0, 1) else "absolute"
data, offsets, _ = synth.synth_ransac_v1(200, 2000, kind, seed=0x5AC50302)
prm = ransac.RansacParameters(); prm.error_thresh = (2.0 / 1000.0) ** 2; prm.min_iterations = prm.max_iterations = 4096; prm.seed = 3
ransac.estimate_batch(est, data[:offsets[8]], offsets[:9], prm)
for rep in range(2):
    t0 = time.time(); res = ransac.estimate_batch(est, data, offsets, prm); dt = time.time() - t0
    print(f"est {est}: {200 * 4096 / dt / 1e6:.2f} M hyp/s wall, fit {res['time_fit_seconds']:.3f} s score {res['time_score_seconds']:.3f} s, models {res['models_scored']}")
