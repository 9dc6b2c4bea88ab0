"""Host-side mirror of the two-view geometric verification front-end
(sfm/estimate_twoview_info.cc:133-305): EstimateTwoViewInfo = feature
normalisation + EstimateRelativePose (both views calibrated) or
EstimateUncalibratedRelativePose (otherwise) + TwoViewInfo bookkeeping.  The
RANSAC itself runs on the device through theia_hip_ransac_estimate_batch; a list
of image pairs goes down as ONE batch per (branch, error threshold) group.

Only the feature normalisation of the PINHOLE prior without radial distortion is
done here (pinhole_camera_model.h:214-241 with the undistortion being the
identity); any other prior raises, so that the caller keeps its CPU path.
"""
import numpy as np

from . import _capi as capi
from . import ransac as _ransac


class Prior:  # camera_intrinsics_prior.h:55-80
    def __init__(self, n):
        self.is_set = False
        self.value = [0.0] * n


class CameraIntrinsicsPrior:  # camera_intrinsics_prior.h:83-111 (the fields this path reads)
    def __init__(self):
        self.image_width = 0
        self.image_height = 0
        self.camera_intrinsics_model_type = "PINHOLE"
        self.focal_length = Prior(1)
        self.principal_point = Prior(2)
        self.aspect_ratio = Prior(1)
        self.skew = Prior(1)
        self.radial_distortion = Prior(4)


class EstimateTwoViewInfoOptions:  # estimate_twoview_info.h:51-81
    def __init__(self):
        self.ransac_type = _ransac.RansacType.RANSAC
        self.max_sampson_error_pixels = 6.0
        self.expected_ransac_confidence = 0.9999
        self.min_ransac_iterations = 10
        self.max_ransac_iterations = 1000
        self.use_mle = True
        self.use_lo = False
        self.lo_start_iterations = 10
        self.min_focal_length = 1.0
        self.max_focal_length = 1.7976931348623157e308
        self.seed = 0   # RandomNumberGenerator seed (options.rng in the reference)


class TwoViewInfo:  # twoview_info.h:54-86
    def __init__(self):
        self.focal_length_1 = 0.0
        self.focal_length_2 = 0.0
        self.position_2 = np.zeros(3)
        self.rotation_2 = np.zeros(3)
        self.num_verified_matches = 0
        self.num_homography_inliers = 0
        self.visibility_score = 0
        self.scale_estimate = -1.0


def ComputeResolutionScaledThreshold(threshold_pixels, image_width, image_height):
    """reconstruction_estimator_utils.cc:98-110"""
    if image_width == 0 and image_height == 0:
        return threshold_pixels
    return threshold_pixels * float(max(image_width, image_height)) / 1024.0


def _pinhole_from_prior(prior):
    """PinholeCameraModel::SetFromCameraIntrinsicsPriors (pinhole_camera_model.cc:74-107) on a
    default-constructed model (focal 1, aspect 1, skew 0, principal point 0)."""
    if prior.camera_intrinsics_model_type != "PINHOLE":
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_UNSUPPORTED,
                                 "two-view feature normalisation: only the PINHOLE prior is built")
    if prior.radial_distortion.is_set and any(v != 0.0 for v in prior.radial_distortion.value[:2]):
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_UNSUPPORTED,
                                 "two-view feature normalisation with radial distortion is not built")
    f, pp = 1.0, [0.0, 0.0]
    if prior.focal_length.is_set:
        f = prior.focal_length.value[0]
    elif prior.image_width != 0 and prior.image_height != 0:
        f = 1.2 * float(max(prior.image_width, prior.image_height))
    if prior.principal_point.is_set:
        pp = [prior.principal_point.value[0], prior.principal_point.value[1]]
    elif prior.image_width != 0 and prior.image_height != 0:
        pp = [prior.image_width / 2.0, prior.image_height / 2.0]
    ar = prior.aspect_ratio.value[0] if prior.aspect_ratio.is_set else 1.0
    skew = prior.skew.value[0] if prior.skew.is_set else 0.0
    return f, ar, skew, pp


def NormalizeFeatures(prior1, prior2, correspondences):
    """estimate_twoview_info.cc:67-102: pixels -> camera coordinates; when either focal length
    prior is missing both focal lengths are reset to 1 (only the principal point is removed)."""
    c = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 4)
    cams = [_pinhole_from_prior(prior1), _pinhole_from_prior(prior2)]
    if not prior1.focal_length.is_set or not prior2.focal_length.is_set:
        cams = [(1.0, ar, sk, pp) for (_, ar, sk, pp) in cams]
    out = np.empty_like(c)
    for k, (f, ar, sk, pp) in enumerate(cams):
        y = (c[:, 2 * k + 1] - pp[1]) / (f * ar)
        x = (c[:, 2 * k] - pp[0] - y * sk) / f
        out[:, 2 * k] = x
        out[:, 2 * k + 1] = y
    return out


def _rotation_to_angle_axis(R):
    """Eigen::AngleAxisd(Matrix3d): through the quaternion (Geometry/AngleAxis.h)."""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2.0
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2.0
        q = np.zeros(4)
        q[1 + i] = 0.25 * s
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    n = np.linalg.norm(q[1:])
    if q[0] < 0:
        n = -n
    if abs(n) < 1e-300:
        return np.zeros(3)
    angle = 2.0 * np.arctan2(n, abs(q[0]))
    return angle * q[1:] / n


def _ransac_params(options, error_thresh, use_mle=None):
    """RansacParameters as the reference fills them.  use_mle: EstimateTwoViewInfoCalibrated copies options.use_mle
    (estimate_twoview_info.cc:167) and so does the homography inlier count; EstimateTwoViewInfoUncalibrated never
    assigns it (:204-232), i.e. the uncalibrated branch always scores with InlierSupport (RansacParameters default)."""
    p = _ransac.RansacParameters()
    p.failure_probability = 1.0 - options.expected_ransac_confidence
    p.min_iterations = options.min_ransac_iterations
    p.max_iterations = options.max_ransac_iterations
    p.use_lo = options.use_lo
    p.lo_start_iterations = options.lo_start_iterations
    p.error_thresh = error_thresh
    p.use_mle = options.use_mle if use_mle is None else use_mle
    p.seed = options.seed
    pc = p.to_c()
    pc.ransac_type = int(_ransac.RansacType(options.ransac_type))
    return pc


def EstimateTwoViewInfoBatch(options, priors1, priors2, correspondences_list, pair_seeds=None):
    """EstimateTwoViewInfo for a list of image pairs.  Returns a list of
    (success, TwoViewInfo, inlier_indices).  Every pair draws from its own RandomNumberGenerator(options.seed)
    (or pair_seeds[i]): a pair's result does not depend on which other pairs share the batch, and
    EstimateTwoViewInfo(pair) == EstimateTwoViewInfoBatch([.., pair, ..])[i]."""
    n = len(correspondences_list)
    if pair_seeds is None:
        pair_seeds = [options.seed] * n
    results = [None] * n
    groups = {}
    for i in range(n):
        p1, p2 = priors1[i], priors2[i]
        calibrated = p1.focal_length.is_set and p2.focal_length.is_set
        t1 = ComputeResolutionScaledThreshold(options.max_sampson_error_pixels, p1.image_width, p1.image_height)
        t2 = ComputeResolutionScaledThreshold(options.max_sampson_error_pixels, p2.image_width, p2.image_height)
        thresh = t1 * t2
        if calibrated:   # estimate_twoview_info.cc:169-171
            thresh = thresh / (p1.focal_length.value[0] * p2.focal_length.value[0])
        groups.setdefault((calibrated, thresh), []).append(i)
    for (calibrated, thresh), idx in groups.items():
        data = [NormalizeFeatures(priors1[i], priors2[i], correspondences_list[i]) for i in idx]
        offsets = np.zeros(len(idx) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([d.shape[0] for d in data])
        est = _ransac.EST_RELATIVE_POSE if calibrated else _ransac.EST_UNCALIBRATED_RELATIVE_POSE
        eparams = None if calibrated else np.array([options.min_focal_length, options.max_focal_length])
        res = _ransac.estimate_batch(est, np.concatenate(data, axis=0), offsets, _ransac_params(options, thresh, use_mle=options.use_mle if calibrated else False), eparams,
                                     seeds=[pair_seeds[i] for i in idx])
        for k, i in enumerate(idx):
            ok = bool(res["success"][k])
            info = TwoViewInfo()
            inliers = []
            if ok:
                m = res["models"][k]
                info.rotation_2 = _rotation_to_angle_axis(m[9:18].reshape(3, 3))
                info.position_2 = m[18:21].copy()
                if calibrated:
                    info.focal_length_1 = priors1[i].focal_length.value[0]
                    info.focal_length_2 = priors2[i].focal_length.value[0]
                else:
                    info.focal_length_1 = float(m[21]); info.focal_length_2 = float(m[22])
                inliers = np.nonzero(res["inlier_mask"][offsets[k]:offsets[k + 1]])[0].tolist()
                info.num_verified_matches = len(inliers)
                # estimate_twoview_info.cc:190-193: the visibility score is computed from
                # *inlier_indices BEFORE the inliers are stored in it, i.e. from an empty list
                info.visibility_score = 0
            results[i] = (ok, info, inliers)
    return results


def EstimateTwoViewInfo(options, intrinsics1, intrinsics2, correspondences):
    """estimate_twoview_info.cc:262-305 -> (success, TwoViewInfo, inlier_indices)."""
    return EstimateTwoViewInfoBatch(options, [intrinsics1], [intrinsics2], [correspondences])[0]


# ------------------------------------------------------------------------------------------------
# Two-view BA and the geometric verification of a match list (two_view_match_geometric_verification.cc)

class TwoViewBundleAdjustmentOptions:  # bundle_adjust_two_views.h:52-57
    def __init__(self):
        from . import sfm as _sfm
        self.ba_options = _sfm.BundleAdjustmentOptions()
        self.constant_camera1_intrinsics = True
        self.constant_camera2_intrinsics = True


def _two_view_ba_options(options):
    from . import sfm as _sfm
    o = _sfm.BundleAdjustmentOptions()          # Ceres' own defaults are the option defaults here
    o.max_num_iterations = options.ba_options.max_num_iterations
    o.use_homogeneous_point_parametrization = False
    o.intrinsics_to_optimize = _sfm.OptimizeIntrinsicsType.FOCAL_LENGTH
    o.use_inner_iterations = False
    o.max_trust_region_radius = 1e16            # bundle_adjust_two_views.cc:61-72 leaves Ceres' default, not BundleAdjuster's 1e12
    return o


def BundleAdjustTwoViewsBatch(options_list, correspondences_list, cameras1, cameras2, points_list):
    """N x BundleAdjustTwoViews as ONE launch (theia_hip_ba_two_views_batch, one LM solve per wavefront).  All pairs take
    options_list[0].ba_options.max_num_iterations; cameras and points are updated in place.  Returns the summaries."""
    from . import ba as _ba, sfm as _sfm
    num = len(correspondences_list)
    if num == 0:
        return []
    corr = [np.ascontiguousarray(c, dtype=np.float64).reshape(-1, 4) for c in correspondences_list]
    offsets = np.zeros(num + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(c) for c in corr])
    cam_ext = np.zeros((num, 2, 6)); intr = np.zeros((num, 2, capi.THEIA_MAX_INTRINSICS))
    model = np.zeros((num, 2), np.int32); kconst = np.zeros((num, 2), np.uint8)
    for i in range(num):
        pts = points_list[i]
        if not (isinstance(pts, np.ndarray) and pts.dtype == np.float64 and pts.shape == (len(corr[i]), 4) and pts.flags["C_CONTIGUOUS"]):
            raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, "points3d must be a C-contiguous float64 [N][4] array, one per correspondence")
        for k, cam in enumerate((cameras1[i], cameras2[i])):
            cam_ext[i, k] = cam["ext"]; v = np.asarray(cam["intr"], dtype=np.float64); intr[i, k, :len(v)] = v; model[i, k] = int(cam["model"])
        kconst[i] = (int(bool(options_list[i].constant_camera1_intrinsics)), int(bool(options_list[i].constant_camera2_intrinsics)))
    allpts = np.ascontiguousarray(np.concatenate(points_list, axis=0))
    summ = _ba.solve_two_views_batch(offsets, np.concatenate(corr, axis=0), cam_ext, intr, model, kconst, allpts,
                                     _two_view_ba_options(options_list[0]).to_c())
    for i in range(num):
        cameras2[i]["ext"][:] = cam_ext[i, 1]
        cameras1[i]["intr"][:] = intr[i, 0][:len(cameras1[i]["intr"])]; cameras2[i]["intr"][:] = intr[i, 1][:len(cameras2[i]["intr"])]
        points_list[i][:] = allpts[offsets[i]:offsets[i + 1]]
    return [_sfm.BundleAdjustmentSummary(s) for s in summ]


def BundleAdjustTwoViews(options, correspondences, camera1, camera2, points3d, batched=True):
    """bundle_adjust_two_views.cc:110-185: camera 1 fixed, camera 2's six extrinsics and (unless held constant) the two
    focal lengths free, every triangulated point a free XYZW 4-vector, no loss function; its own SetSolverOptions
    (:60-72) takes only max_num_iterations from the options.  camera = dict(ext[6], intr[<=10], model); both cameras
    and points3d [N][4] are updated in place.  batched (default): a batch of one through theia_hip_ba_two_views_batch;
    batched=False: theia_hip_ba_solve on the flat problem (the same arithmetic; the parity tests compare the two)."""
    from . import ba as _ba, sfm as _sfm
    if batched:
        return BundleAdjustTwoViewsBatch([options], [correspondences], [camera1], [camera2], [points3d])[0]
    c = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 4)
    n = c.shape[0]
    pts = points3d
    if not (isinstance(pts, np.ndarray) and pts.dtype == np.float64 and pts.shape == (n, 4) and pts.flags["C_CONTIGUOUS"]):
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, "points3d must be a C-contiguous float64 [N][4] array, one per correspondence")
    intr = np.zeros((2, capi.THEIA_MAX_INTRINSICS))
    for k, cam in enumerate((camera1, camera2)):
        v = np.asarray(cam["intr"], dtype=np.float64); intr[k, :len(v)] = v
    flat = capi.FlatProblem(np.array([camera1["ext"], camera2["ext"]], dtype=np.float64), intr,
                            [int(camera1["model"]), int(camera2["model"])], [0, 1], pts,
                            np.concatenate([c[:, 0:2], c[:, 2:4]]),
                            np.concatenate([np.zeros(n, np.int32), np.ones(n, np.int32)]),
                            np.concatenate([np.arange(n), np.arange(n)]).astype(np.int32),
                            cam_const=[3, 0],
                            group_const=[int(bool(options.constant_camera1_intrinsics)), int(bool(options.constant_camera2_intrinsics))])
    o = _two_view_ba_options(options)
    s, _ = _ba.solve(flat, o.to_c())
    camera1["ext"][:] = flat.cam_ext[0]; camera2["ext"][:] = flat.cam_ext[1]
    camera1["intr"][:] = flat.intrinsics[0][:len(camera1["intr"])]; camera2["intr"][:] = flat.intrinsics[1][:len(camera2["intr"])]
    pts[:] = flat.points
    return _sfm.BundleAdjustmentSummary(s)


class TwoViewMatchGeometricVerificationOptions:  # two_view_match_geometric_verification.h:51-99
    def __init__(self):
        self.estimate_twoview_info_options = EstimateTwoViewInfoOptions()
        self.min_num_inlier_matches = 30
        self.guided_matching = False
        self.guided_matching_max_distance_pixels = 2.0
        self.guided_matching_lowes_ratio = float(np.float32(0.8))   # a float in the reference
        self.bundle_adjustment = True
        self.triangulation_max_reprojection_error = 15.0
        self.min_triangulation_angle_degrees = 4.0
        self.final_max_reprojection_error = 5.0


def _setup_cameras(prior1, prior2, info):
    """SetupCameras (two_view_match_geometric_verification.cc:56-68): pinhole cameras from the priors with the
    estimated focal lengths; camera 1 at the origin."""
    cams = []
    for prior, f, ext in ((prior1, info.focal_length_1, np.zeros(6)),
                          (prior2, info.focal_length_2, np.concatenate([info.position_2, info.rotation_2]))):
        _, ar, skew, pp = _pinhole_from_prior(prior)
        cams.append({"ext": np.array(ext, dtype=np.float64), "intr": np.array([f, ar, skew, pp[0], pp[1], 0.0, 0.0]), "model": 0})
    return cams


def _rays(cam, uv):
    """Camera::PixelToUnitDepthRay(pixel).normalized() of a distortion-free pinhole camera (camera.cc:177-196)."""
    from . import synth as _synth
    f, ar, skew, px, py = cam["intr"][:5]
    y = (uv[:, 1] - py) / (f * ar)
    x = (uv[:, 0] - px - y * skew) / f
    d = np.stack([x, y, np.ones_like(x)], axis=1)
    R = _synth.angle_axis_to_matrix(cam["ext"][3:6])
    d = d @ R          # R^T d, row-wise
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def _per_view_reprojection(cams, corr, pts):
    """Squared reprojection error and the depth sign of every (point, view) pair, on the device: each pair becomes a
    one-observation track of theia_hip_track_statistics."""
    from . import ba as _ba
    n = corr.shape[0]
    intr = np.zeros((2, capi.THEIA_MAX_INTRINSICS)); intr[0, :7] = cams[0]["intr"]; intr[1, :7] = cams[1]["intr"]
    flat = capi.FlatProblem(np.array([cams[0]["ext"], cams[1]["ext"]]), intr, [0, 0], [0, 1], np.concatenate([pts, pts]),
                            np.concatenate([corr[:, 0:2], corr[:, 2:4]]),
                            np.concatenate([np.zeros(n, np.int32), np.ones(n, np.int32)]), np.arange(2 * n, dtype=np.int32))
    err, behind, _ = _ba.track_statistics(flat)
    return err.reshape(2, n), behind.reshape(2, n)


def _pairs_flat(cams_list, corr_list, pts_list, one_track_per_view):
    """The pairs of a batch as ONE flat problem (cameras 2 i and 2 i + 1, one intrinsics group per camera): per pair first
    the observations of camera 1, then those of camera 2.  one_track_per_view: every (point, view) is its own track (the
    per-view statistics); otherwise a point is one track seen by both cameras (triangulation)."""
    num = len(corr_list)
    cam_ext = np.zeros((2 * num, 6)); intr = np.zeros((2 * num, capi.THEIA_MAX_INTRINSICS))
    for i, cams in enumerate(cams_list):
        for k in range(2):
            cam_ext[2 * i + k] = cams[k]["ext"]; intr[2 * i + k, :7] = cams[k]["intr"]
    ns = [len(c) for c in corr_list]
    uv = np.concatenate([np.concatenate([c[:, 0:2], c[:, 2:4]]) for c in corr_list])
    ocam = np.concatenate([np.concatenate([np.full(n, 2 * i, np.int32), np.full(n, 2 * i + 1, np.int32)]) for i, n in enumerate(ns)])
    if one_track_per_view:
        pts = np.concatenate([np.concatenate([p, p]) for p in pts_list])
        opt = np.arange(2 * sum(ns), dtype=np.int32)
    else:
        pts = np.zeros((sum(ns), 4)) if pts_list is None else np.concatenate(pts_list)
        base = np.concatenate([[0], np.cumsum(ns)[:-1]])
        opt = np.concatenate([np.concatenate([b + np.arange(n), b + np.arange(n)]) for b, n in zip(base, ns)]).astype(np.int32)
    flat = capi.FlatProblem(cam_ext, intr, np.zeros(2 * num, np.int32), np.arange(2 * num, dtype=np.int32), np.ascontiguousarray(pts),
                            uv, ocam, opt)
    return flat, ns


def _per_view_reprojection_batch(cams_list, corr_list, pts_list):
    """_per_view_reprojection of every pair in ONE theia_hip_track_statistics call: a list of (err [2][n], behind [2][n])."""
    from . import ba as _ba
    if not corr_list:
        return []
    flat, ns = _pairs_flat(cams_list, corr_list, pts_list, True)
    err, behind, _ = _ba.track_statistics(flat)
    out, o = [], 0
    for n in ns:
        out.append((err[o:o + 2 * n].reshape(2, n), behind[o:o + 2 * n].reshape(2, n)))
        o += 2 * n
    return out


# ------------------------------------------------------------------------------------------------
# Guided matching (matching/guided_epipolar_matcher.cc), the optional step of VerifyMatches between the two-view geometry and
# the two-view BA.  Host side: epipolar lines, their grouping and the grid walk (a few thousand elements per pair); the
# descriptor search of all groups of a pair is ONE device launch (theia_hip_guided_knn, csrc/guided_knn.hip).
class KeypointsAndDescriptors:  # matching/keypoints_and_descriptors.h (the fields this path reads)
    def __init__(self, keypoints, descriptors):
        self.keypoints = np.ascontiguousarray(keypoints, dtype=np.float64).reshape(-1, 2)
        self.descriptors = np.ascontiguousarray(descriptors, dtype=np.float32).reshape(len(self.keypoints), -1)


def _projection_matrix(cam):
    from . import synth as _synth
    f, ar, skew, px, py = cam["intr"][:5]
    K = np.array([[f, skew, px], [0.0, f * ar, py], [0.0, 0.0, 1.0]])
    R = _synth.angle_axis_to_matrix(cam["ext"][3:6])
    return K @ np.concatenate([R, -(R @ cam["ext"][0:3])[:, None]], axis=1)


def _fundamental_from_projections(Pa, Pb):
    """FundamentalMatrixFromProjectionMatrices(Pa, Pb) (fundamental_matrix_util.cc:218-237): nine 4 x 4 determinants, each by
    2 x 2 minors of its two row pairs."""
    def minors(A, B):   # the six 2 x 2 minors of the row pair (A, B), column pairs 01 02 03 12 13 23
        return np.array([A[0] * B[1] - A[1] * B[0], A[0] * B[2] - A[2] * B[0], A[0] * B[3] - A[3] * B[0],
                         A[1] * B[2] - A[2] * B[1], A[1] * B[3] - A[3] * B[1], A[2] * B[3] - A[3] * B[2]])
    i1, i2 = (1, 2, 0), (2, 0, 1)
    F = np.zeros((3, 3))
    for r in range(3):
        lo = minors(Pa[i1[r]], Pa[i2[r]])
        for c in range(3):
            up = minors(Pb[i1[c]], Pb[i2[c]])
            F[r, c] = up[0] * lo[5] - up[1] * lo[4] + up[2] * lo[3] + up[3] * lo[2] - up[4] * lo[1] + up[5] * lo[0]
    return F


def GuidedEpipolarMatches(camera1, camera2, features1, features2, matches, guided_matching_max_distance_pixels=2.0, lowes_ratio=0.8, seed=0):
    """GuidedEpipolarMatcher(options, camera1, camera2, features1, features2).GetMatches(&matches)
    (guided_epipolar_matcher.cc:140-185): returns the input matches [(feature1, feature2)] followed by the added ones.
    camera = dict(ext[6], intr[>= 5], model PINHOLE).  Stated deviations: the candidates of a group are searched in ascending
    feature index (the reference walks an unordered_set: the order only decides ties of equal distances), the descriptor
    distance is summed in sequence in float, and the random candidates that fill a group up to 50 come from
    RandomNumberGenerator(seed) (the reference seeds it from the clock when no generator is handed in)."""
    import ctypes as C
    matches = np.asarray(matches, dtype=np.int64).reshape(-1, 2)
    kp1, kp2 = features1.keypoints, features2.keypoints
    n1, n2 = len(kp1), len(kp2)
    d = float(guided_matching_max_distance_pixels)
    free2 = np.ones(n2, bool); free2[matches[:, 1]] = False
    free1 = np.ones(n1, bool); free1[matches[:, 0]] = False
    u2 = np.flatnonzero(free2)
    if len(u2) == 0:
        return [tuple(int(v) for v in m) for m in matches]
    # the four grids (cell size 2 d; offsets 0 / d in x and y): centre of a point, features by centre
    offs = np.array([[0.0, 0.0], [d, 0.0], [0.0, d], [d, d]])
    def centres(pts):   # pts (M, 2) -> (M, 4, 2) int64
        q = np.floor((pts[:, None, :] - offs[None]) / (2.0 * d)) * 2.0 * d + d + offs[None]
        return np.trunc(q).astype(np.int64)
    cells = {}
    cu = centres(kp2[u2])
    for k, i in enumerate(u2):
        for j in range(4):
            cells.setdefault((j, int(cu[k, j, 0]), int(cu[k, j, 1])), []).append(int(i))
    tlx, tly = kp2[u2, 0].min(), kp2[u2, 1].min(); brx, bry = kp2[u2, 0].max(), kp2[u2, 1].max()
    # epipolar lines of the unmatched features of image 1, their intersections with the bounding box (left, top, right, bottom)
    F = _fundamental_from_projections(_projection_matrix(camera2), _projection_matrix(camera1))
    u1 = np.flatnonzero(free1)
    L = (F @ np.concatenate([kp1[u1], np.ones((len(u1), 1))], axis=1).T).T
    L = L / np.sqrt(L[:, 0] * L[:, 0] + L[:, 1] * L[:, 1])[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        yl = -(L[:, 2] + L[:, 0] * tlx) / L[:, 1]; xt = -(L[:, 2] + L[:, 1] * tly) / L[:, 0]
        yr = -(L[:, 2] + L[:, 0] * brx) / L[:, 1]; xb = -(L[:, 2] + L[:, 1] * bry) / L[:, 0]
    ok = np.stack([(yl >= tly) & (yl <= bry), (xt >= tlx) & (xt <= brx), (yr >= tly) & (yr <= bry), (xb >= tlx) & (xb <= brx)], axis=1)
    px = np.stack([np.full(len(u1), tlx), xt, np.full(len(u1), brx), xb], axis=1)
    py = np.stack([yl, np.full(len(u1), tly), yr, np.full(len(u1), bry)], axis=1)
    two = ok.sum(1) == 2
    sel = np.flatnonzero(two)
    enc = np.zeros(len(sel), dtype=np.uint64)
    for k, r in enumerate(sel):
        a, b = np.flatnonzero(ok[r])
        w = [int(px[r, a]) & 0xffff, int(py[r, a]) & 0xffff, int(px[r, b]) & 0xffff, int(py[r, b]) & 0xffff]
        enc[k] = (w[0] << 48) | (w[1] << 32) | (w[2] << 16) | w[3]
    order = np.lexsort((u1[sel], enc))
    # groups of similar lines (a running mean of the decoded endpoints: inherently sequential)
    groups = []
    for k in order:
        code = int(enc[k])
        e0 = np.array([float((code >> 48) & 0xffff), float((code >> 32) & 0xffff)]); e1 = np.array([float((code >> 16) & 0xffff), float(code & 0xffff)])
        if not groups or float(np.sum((groups[-1][0] - e0) ** 2)) > d * d:
            groups.append([e0.copy(), e1.copy(), []])
        g = groups[-1]
        g[2].append(int(u1[sel[k]]))
        w = 1.0 / len(g[2])
        g[0] = (1.0 - w) * g[0] + w * e0; g[1] = (1.0 - w) * g[1] + w * e1
    if not groups:
        return [tuple(int(v) for v in m) for m in matches]
    L_ = capi.lib()
    draws = np.zeros(50 * len(groups), dtype=np.int32)
    L_.theia_hip_randint_stream.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    capi.check(L_.theia_hip_randint_stream(int(seed) & 0xFFFFFFFF, len(draws), 0, n2 - 1, draws.ctypes.data_as(C.POINTER(C.c_int32))))
    nd = 0
    q_off, c_off, q_idx, c_idx = [0], [0], [], []
    for e0, e1, feats in groups:
        steps = int(np.sqrt(np.sum((e1 - e0) ** 2)) / d)
        cand = set()
        if steps > 0:
            delta = (e0 - e1) / float(steps)
            pts = np.empty((steps, 2)); pt = e1.copy()
            for k in range(steps):                                                  # the reference's running sum sample_point += line_delta
                pt = pt + delta; pts[k] = pt
            cc = centres(pts)
            dist = np.sum((cc.astype(np.float64) - pts[:, None, :]) ** 2, axis=2)
            jb = np.argmin(dist, axis=1)                                            # first minimum: the reference's strict <
            for k in range(steps):
                cand.update(cells.get((int(jb[k]), int(cc[k, jb[k], 0]), int(cc[k, jb[k], 1])), ()))
        for _ in range(len(cand), 50):
            cand.add(int(draws[nd])); nd += 1
        cl = sorted(cand)
        q_idx.extend(feats); c_idx.extend(cl)
        q_off.append(len(q_idx)); c_off.append(len(c_idx))
    q_off = np.asarray(q_off, dtype=np.int64); c_off = np.asarray(c_off, dtype=np.int64)
    q_idx = np.asarray(q_idx, dtype=np.int32); c_idx = np.asarray(c_idx, dtype=np.int32)
    nn_d = np.zeros((len(q_idx), 2), dtype=np.float32); nn_i = np.zeros((len(q_idx), 2), dtype=np.int32)
    d1 = features1.descriptors; d2 = features2.descriptors
    if d1.shape[1] != d2.shape[1]:
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, "descriptor dimensions differ")
    L_.theia_hip_guided_knn.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32,
                                        C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    capi.check(L_.theia_hip_guided_knn(len(groups), q_off.ctypes.data_as(C.POINTER(C.c_int64)), q_idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                       c_off.ctypes.data_as(C.POINTER(C.c_int64)), c_idx.ctypes.data_as(C.POINTER(C.c_int32)), n1, n2, d1.shape[1],
                                       d1.ctypes.data_as(C.POINTER(C.c_float)), d2.ctypes.data_as(C.POINTER(C.c_float)),
                                       nn_d.ctypes.data_as(C.POINTER(C.c_float)), nn_i.ctypes.data_as(C.POINTER(C.c_int32))))
    ratio_sq = float(lowes_ratio) * float(lowes_ratio)
    good = (nn_i[:, 1] >= 0) & (nn_d[:, 0].astype(np.float64) < nn_d[:, 1].astype(np.float64) * ratio_sq)
    out = [tuple(int(v) for v in m) for m in matches]
    out.extend((int(q_idx[k]), int(nn_i[k, 0])) for k in np.flatnonzero(good))
    return out


def VerifyMatchesBatch(options, priors1, priors2, correspondences_list, indexed=None):
    """TwoViewMatchGeometricVerification::VerifyMatches (two_view_match_geometric_verification.cc:114-183) for a list of
    image pairs given as pixel correspondences [(x1, y1, x2, y2)] (the reference indexes keypoint lists).  Returns a
    list of (success, TwoViewInfo, verified_indices).  The homography count (:331-368) and EstimateTwoViewInfo run as
    two RANSAC batches over all pairs, triangulation and the reprojection filters as device sweeps, the two-view BA of all
    pairs as one launch (theia_hip_ba_two_views_batch).  Guided matching (:157-170) needs the keypoints and descriptors of both
    images: it runs in the indexed form (VerifyMatchesIndexedBatch: indexed = (features1, features2, matches) per pair; the
    third element of a result is then the list of verified (feature1, feature2) pairs) and is refused here without them.
    Stated deviation: the reference draws the homography and the relative-pose samples from ONE generator in sequence; here both
    batches start from `seed`."""
    from . import ba as _ba
    if options.guided_matching and indexed is None:
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_UNSUPPORTED, "guided matching needs keypoints and descriptors: use VerifyMatchesIndexed(Batch)")
    n = len(correspondences_list)
    corr = [np.ascontiguousarray(c, dtype=np.float64).reshape(-1, 4) for c in correspondences_list]
    results = [(False, TwoViewInfo(), []) for _ in range(n)]
    live = [i for i in range(n) if len(corr[i]) >= options.min_num_inlier_matches]   # :117-119
    if not live:
        return results
    eo = options.estimate_twoview_info_options
    # CountHomographyInliers: camera1_/camera2_ are still default-constructed there (image size 0 x 0), so the
    # resolution scaling leaves the threshold at max_sampson_error_pixels^2
    hp = _ransac_params(eo, eo.max_sampson_error_pixels * eo.max_sampson_error_pixels)
    hp.use_lo = 0                                   # a fresh RansacParameters: use_lo keeps its default
    offsets = np.zeros(len(live) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(corr[i]) for i in live])
    hres = _ransac.estimate_batch(_ransac.EST_HOMOGRAPHY, np.concatenate([corr[i] for i in live]), offsets, hp, seeds=[eo.seed] * len(live))
    tv = EstimateTwoViewInfoBatch(eo, [priors1[i] for i in live], [priors2[i] for i in live], [corr[i] for i in live])
    cand = []   # pairs that go through the two-view BA: [i, info, idx, c, cams]
    for k, i in enumerate(live):
        ok, info, inliers = tv[k]
        info.num_homography_inliers = int(hres["num_inliers"][k])
        if not ok or len(inliers) < options.min_num_inlier_matches:
            results[i] = (False, info, [])
            continue
        idx = np.asarray(inliers, dtype=np.int64)
        ci = corr[i][idx]
        if indexed is not None:
            # the indexed form carries (feature1, feature2) pairs instead of positions in the input list; guided matching
            # (:157-170) appends pairs the input never held
            f1, f2, mt = indexed[i]
            idx = np.asarray(mt, dtype=np.int64).reshape(-1, 2)[idx]
            if options.guided_matching:
                cams_g = _setup_cameras(priors1[i], priors2[i], info)
                idx = np.asarray(GuidedEpipolarMatches(cams_g[0], cams_g[1], f1, f2, idx, options.guided_matching_max_distance_pixels,
                                                       options.guided_matching_lowes_ratio, seed=eo.seed), dtype=np.int64).reshape(-1, 2)
                ci = np.concatenate([f1.keypoints[idx[:, 0]], f2.keypoints[idx[:, 1]]], axis=1)
        if options.bundle_adjustment and len(idx) > options.min_num_inlier_matches:
            cand.append([i, info, idx, ci, _setup_cameras(priors1[i], priors2[i], info)])
            continue
        info.num_verified_matches = len(idx)
        results[i] = (len(idx) > options.min_num_inlier_matches, info, idx.tolist())
    todo = []   # (i, info, idx, c, pts, cams, options) of the pairs that reach BundleAdjustTwoViews
    if cand:
        # TriangulatePoints (:186-257) of all pairs as one device sweep: angle test, midpoint; then both reprojection
        # errors below the threshold (one statistics sweep)
        flat, ns = _pairs_flat([t[4] for t in cand], [t[3] for t in cand], None, False)
        rays = np.concatenate([np.concatenate([_rays(t[4][0], t[3][:, 0:2]), _rays(t[4][1], t[3][:, 2:4])]) for t in cand])
        est, _ = _ba.estimate_tracks(flat, rays, _ba.default_options(), options.min_triangulation_angle_degrees, 1e150, False)
        bounds = np.concatenate([[0], np.cumsum(ns)])
        pts_all = [np.ascontiguousarray(flat.points[bounds[j]:bounds[j + 1]]) for j in range(len(cand))]
        stats = _per_view_reprojection_batch([t[4] for t in cand], [t[3] for t in cand], pts_all)
        lim = options.triangulation_max_reprojection_error ** 2
        for j, (i, info, idx, c, cams) in enumerate(cand):
            err, behind = stats[j]
            keep = est[bounds[j]:bounds[j + 1]] & (behind.sum(0) == 0) & (err[0] < lim) & (err[1] < lim)
            idx, c, pts = idx[keep], c[keep], np.ascontiguousarray(pts_all[j][keep])
            if len(idx) < options.min_num_inlier_matches:          # :271-273
                results[i] = (False, info, [])
                continue
            bo = TwoViewBundleAdjustmentOptions()
            bo.constant_camera1_intrinsics = priors1[i].focal_length.is_set
            bo.constant_camera2_intrinsics = priors2[i].focal_length.is_set
            todo.append((i, info, idx, c, pts, cams, bo))
    # BundleAdjustTwoViews of every surviving pair as ONE launch (one LM solve per wavefront), then the final filter
    summs = BundleAdjustTwoViewsBatch([t[6] for t in todo], [t[3] for t in todo], [t[5][0] for t in todo], [t[5][1] for t in todo],
                                      [t[4] for t in todo])
    good = [t for t, sm in zip(todo, summs) if sm.success]
    for t, sm in zip(todo, summs):
        if not sm.success:
            results[t[0]] = (False, t[1], [])
    stats = _per_view_reprojection_batch([t[5] for t in good], [t[3] for t in good], [t[4] for t in good])
    lim = options.final_max_reprojection_error ** 2
    for (i, info, idx, c, pts, cams, bo), (err, behind) in zip(good, stats):
        keep = (behind.sum(0) == 0) & (err[0] < lim) & (err[1] < lim)
        idx = idx[keep]
        info.rotation_2 = cams[1]["ext"][3:6].copy()
        info.position_2 = cams[1]["ext"][0:3] / np.linalg.norm(cams[1]["ext"][0:3])
        info.focal_length_1 = float(cams[0]["intr"][0]); info.focal_length_2 = float(cams[1]["intr"][0])
        info.num_verified_matches = len(idx)
        results[i] = (len(idx) > options.min_num_inlier_matches, info, idx.tolist())
    return results


def VerifyMatches(options, intrinsics1, intrinsics2, correspondences):
    return VerifyMatchesBatch(options, [intrinsics1], [intrinsics2], [correspondences])[0]


def VerifyMatchesIndexedBatch(options, priors1, priors2, features1_list, features2_list, matches_list):
    """The reference's own form of the call: TwoViewMatchGeometricVerification(options, intrinsics1, intrinsics2, features1,
    features2, matches).VerifyMatches(...) with KeypointsAndDescriptors and IndexedFeatureMatch lists
    (two_view_match_geometric_verification.cc:86-183), guided matching included.  Returns per pair (success, TwoViewInfo,
    verified (feature1, feature2) pairs)."""
    corr, idxd = [], []
    for f1, f2, m in zip(features1_list, features2_list, matches_list):
        m = np.asarray(m, dtype=np.int64).reshape(-1, 2)
        corr.append(np.concatenate([f1.keypoints[m[:, 0]], f2.keypoints[m[:, 1]]], axis=1) if len(m) else np.zeros((0, 4)))
        idxd.append((f1, f2, m))
    out = VerifyMatchesBatch(options, priors1, priors2, corr, indexed=idxd)
    return [(ok, info, [tuple(int(v) for v in p) for p in pairs]) for ok, info, pairs in out]


def VerifyMatchesIndexed(options, intrinsics1, intrinsics2, features1, features2, matches):
    return VerifyMatchesIndexedBatch(options, [intrinsics1], [intrinsics2], [features1], [features2], [matches])[0]
