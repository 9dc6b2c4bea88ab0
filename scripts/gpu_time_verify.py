"""GPU box: TwoViewMatchGeometricVerification over a batch of synthetic pairs (two-view BA on), wall time per pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import twoview as tv, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data, offsets, truth = synth.synth_ransac_v1(npairs, 400, "fundamental", seed=0x5AC51900, inlier_lo=0.6, inlier_hi=0.8, noise_px=0.5)
pr = tv.CameraIntrinsicsPrior(); pr.image_width = 1000; pr.image_height = 800
pr.focal_length.is_set = True; pr.focal_length.value = [1000.0]
pr.principal_point.is_set = True; pr.principal_point.value = [500.0, 400.0]
corr = [data[offsets[i]:offsets[i + 1]] for i in range(npairs)]
vo = tv.TwoViewMatchGeometricVerificationOptions()
vo.estimate_twoview_info_options.seed = 7; vo.estimate_twoview_info_options.max_sampson_error_pixels = 2.0
tv.VerifyMatchesBatch(vo, [pr] * 4, [pr] * 4, corr[:4])
for ba_on, nth in ((True, 1), (False, 1)):
    vo.bundle_adjustment = ba_on; vo.host_threads = nth
    t0 = time.perf_counter(); out = tv.VerifyMatchesBatch(vo, [pr] * npairs, [pr] * npairs, corr); dt = time.perf_counter() - t0
    print("bundle_adjustment", ba_on, "threads", nth, "%d pairs: %.1f ms total, %.2f ms per pair, %d verified" % (npairs, 1e3 * dt, 1e3 * dt / npairs, sum(1 for o in out if o[0])), flush=True)
