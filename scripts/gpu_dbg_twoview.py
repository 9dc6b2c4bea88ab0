import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from pytheiasfm_amd import twoview as tv, synth
data, offsets, truth = synth.synth_ransac_v1(3, 400, "fundamental", seed=0x5AC51800, inlier_lo=0.6, inlier_hi=0.8, noise_px=0.5)
pr = tv.CameraIntrinsicsPrior(); pr.image_width = 1000; pr.image_height = 800
pr.focal_length.is_set = True; pr.focal_length.value = [1000.0]
pr.principal_point.is_set = True; pr.principal_point.value = [500.0, 400.0]
corr = [data[offsets[i]:offsets[i + 1]] for i in range(3)]
for seed in (7, 8, 9, 10):
    vo = tv.TwoViewMatchGeometricVerificationOptions()
    vo.estimate_twoview_info_options.seed = seed; vo.estimate_twoview_info_options.max_sampson_error_pixels = 2.0
    vo.estimate_twoview_info_options.use_lo = True; vo.estimate_twoview_info_options.lo_start_iterations = 5
    out = tv.VerifyMatchesBatch(vo, [pr] * 3, [pr] * 3, corr)
    plain = tv.EstimateTwoViewInfoBatch(vo.estimate_twoview_info_options, [pr] * 3, [pr] * 3, corr)
    for i in range(3):
        ok, info, idx = out[i]
        R = synth.angle_axis_to_matrix(info.rotation_2)
        ang = np.degrees(np.arccos(np.clip((np.trace(R @ truth["R"][i].T) - 1) / 2, -1, 1)))
        Rp = synth.angle_axis_to_matrix(plain[i][1].rotation_2)
        angp = np.degrees(np.arccos(np.clip((np.trace(Rp @ truth["R"][i].T) - 1) / 2, -1, 1)))
        print(seed, i, ok, len(idx), len(plain[i][2]), int(truth["inlier"][i].sum()), "ang %.3f angp %.3f" % (ang, angp), "pos", info.position_2 @ truth["position"][i] / np.linalg.norm(truth["position"][i]))
