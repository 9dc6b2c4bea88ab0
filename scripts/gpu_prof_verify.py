import os, sys, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np
from pytheiasfm_amd import twoview as tv, synth
npairs = 400
data, offsets, truth = synth.synth_ransac_v1(npairs, 400, "fundamental", seed=0x5AC51900, inlier_lo=0.6, inlier_hi=0.8, noise_px=0.5)
pr = tv.CameraIntrinsicsPrior(); pr.image_width = 1000; pr.image_height = 800
pr.focal_length.is_set = True; pr.focal_length.value = [1000.0]
pr.principal_point.is_set = True; pr.principal_point.value = [500.0, 400.0]
corr = [data[offsets[i]:offsets[i + 1]] for i in range(npairs)]
vo = tv.TwoViewMatchGeometricVerificationOptions()
vo.estimate_twoview_info_options.seed = 7; vo.estimate_twoview_info_options.max_sampson_error_pixels = 2.0
tv.VerifyMatchesBatch(vo, [pr] * 4, [pr] * 4, corr[:4])
pf = cProfile.Profile(); pf.enable()
out = tv.VerifyMatchesBatch(vo, [pr] * npairs, [pr] * npairs, corr)
pf.disable()
pstats.Stats(pf).sort_stats("cumulative").print_stats(18)
