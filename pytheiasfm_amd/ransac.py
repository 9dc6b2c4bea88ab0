"""Host-side mirror of pyTheia's RANSAC estimators (pytheia.sfm.Estimate*,
src/pytheia/sfm/sfm.cc:812-850, solvers/solvers.cc:68-102) over the C-ABI
batch entry point theia_hip_ransac_estimate_batch.

Same names / argument order as the pybind wrappers
(estimators_wrapper.cc:41-57,130-142): Estimate*(ransac_params, ransac_type,
..., correspondences) -> (success, model, RansacSummary).
"""
import copy
import ctypes as C
import enum
import time

import numpy as np

from . import _capi as capi, synth


class RansacType(enum.IntEnum):  # create_and_initialize_ransac_variant.h:52
    RANSAC = 0
    PROSAC = 1
    LMED = 2
    EXHAUSTIVE = 3


class PnPType(enum.IntEnum):  # estimate_calibrated_absolute_pose.h:54
    KNEIP = 0
    SQPnP = 1
    DLS = 2


EST_RELATIVE_POSE, EST_ESSENTIAL_MATRIX, EST_ABS_KNEIP, EST_ABS_DLS, EST_ABS_SQPNP = range(5)
EST_FUNDAMENTAL_MATRIX, EST_HOMOGRAPHY, EST_DOMINANT_PLANE, EST_RELATIVE_POSE_KNOWN_ORIENTATION = range(5, 9)
EST_UNCALIBRATED_RELATIVE_POSE = 9
EST_ABSOLUTE_POSE_KNOWN_ORIENTATION = 10
EST_TRIANGULATION = 11
EST_RADIAL_HOMOGRAPHY = 12
EST_SIMILARITY_2D3D = 13
EST_UNCALIBRATED_ABSOLUTE_POSE = 14
EST_RIGID_TRANSFORMATION_2D3D = 15
EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE = 16


class RansacParameters:
    """solvers/sample_consensus_estimator.h:58-126, same fields and defaults.
    `rng` is not bound to Python in the reference either (solvers.cc:89-102);
    `seed` is this backend's handle on RandomNumberGenerator(seed)."""

    def __init__(self):
        self.error_thresh = -1.0
        self.failure_probability = 0.01
        self.min_inlier_ratio = 0.0
        self.min_iterations = 100
        self.max_iterations = 2 ** 31 - 1
        self.use_mle = False
        self.use_Tdd_test = False
        self.use_lo = False
        self.lo_start_iterations = 50
        self.seed = 0
        self.ransac_type = RansacType.RANSAC  # set by the Estimate* wrappers from their ransac_type argument

    def to_c(self):
        p = capi.RansacParams()
        p.error_thresh = float(self.error_thresh)
        p.failure_probability = float(self.failure_probability)
        p.min_inlier_ratio = float(self.min_inlier_ratio)
        p.min_iterations = int(self.min_iterations)
        p.max_iterations = int(self.max_iterations)
        p.use_mle = int(bool(self.use_mle))
        p.use_lo = int(bool(self.use_lo))
        p.lo_start_iterations = int(self.lo_start_iterations)
        p.use_Tdd_test = int(bool(self.use_Tdd_test))
        p.seed = int(self.seed) & 0xFFFFFFFF
        p.ransac_type = int(self.ransac_type)
        return p


class RansacSummary:
    """solvers/sample_consensus_estimator.h:129-144."""

    def __init__(self):
        self.inliers = []
        self.num_input_data_points = 0
        self.num_iterations = 0
        self.confidence = 0.0
        self.num_lo_iterations = 0


class RelativePose:  # estimate_relative_pose.h:49-53
    def __init__(self, m):
        self.essential_matrix = m[0:9].reshape(3, 3).copy()
        self.rotation = m[9:18].reshape(3, 3).copy()
        self.position = m[18:21].copy()


class CalibratedAbsolutePose:  # estimate_calibrated_absolute_pose.h:49-52
    def __init__(self, m):
        self.rotation = m[0:9].reshape(3, 3).copy()
        self.position = m[9:12].copy()


def _sig():
    L = capi.lib()
    if not getattr(L, "_ransac_ready", False):
        L.theia_hip_ransac_estimate_batch.argtypes = [C.POINTER(capi.RansacBatch), C.POINTER(capi.RansacParams),
                                                      C.POINTER(capi.RansacResult)]
        L.theia_hip_five_point_relative_pose.argtypes = [C.c_int32, capi.c_double_p, capi.c_double_p, capi.c_int32_p]
        L.theia_hip_pose_from_three_points.argtypes = [C.c_int32, capi.c_double_p, capi.c_double_p, capi.c_double_p,
                                                       capi.c_int32_p]
        L.theia_hip_sqpnp.argtypes = [C.c_int32, C.POINTER(C.c_int64), capi.c_double_p, capi.c_double_p, capi.c_double_p,
                                      capi.c_double_p, capi.c_int32_p]
        L.theia_hip_dls_pnp.argtypes = [C.c_int32, C.POINTER(C.c_int64), capi.c_double_p, capi.c_double_p, C.POINTER(C.c_int64),
                                        capi.c_double_p, capi.c_double_p, capi.c_int32_p]
        L.theia_hip_dls_macaulay_terms.argtypes = [C.c_int64, C.c_int64, capi.c_double_p]
        L.theia_hip_four_point_pose_and_focal_length.argtypes = [C.c_int32, capi.c_double_p, capi.c_double_p, capi.c_int32_p]
        L.theia_hip_four_point_focal_length_radial_distortion_ex.argtypes = [C.c_int32, capi.c_double_p, capi.c_double_p, capi.c_double_p,
                                                                          capi.c_double_p, capi.c_int32_p, capi.c_int32_p]
        L.theia_hip_dls_macaulay_terms.restype = None
        L.theia_ransac_params_default.argtypes = [C.POINTER(capi.RansacParams)]
        L._ransac_ready = True
    return L


def estimate_batch(estimator, data, offsets, params, estimator_params=None, seeds=None):
    """theia_hip_ransac_estimate_batch.  data [total][datum], offsets [P+1]; seeds: optional per-problem
    RandomNumberGenerator seeds (default params.seed + problem index).  Returns dict of per-problem arrays."""
    L = _sig()
    data = np.ascontiguousarray(data, dtype=np.float64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    P = len(offsets) - 1
    total = int(offsets[-1]) if P > 0 else 0
    b = capi.RansacBatch()
    b.estimator = int(estimator); b.num_problems = P
    b.offsets = capi.ptr(offsets, C.c_int64); b.data = capi.ptr(data, C.c_double)
    ep = None if estimator_params is None else np.ascontiguousarray(estimator_params, dtype=np.float64)
    # (the C side reads a fixed number of entries: two focal-length limits, or the radial-distortion metadata + first-call flag)
    need = {EST_UNCALIBRATED_RELATIVE_POSE: 2, EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE: 5}.get(int(estimator), 0)
    if ep is not None and ep.size < need:
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, f"estimator {int(estimator)} reads {need} estimator_params, got {ep.size}")
    b.estimator_params = None if ep is None else capi.ptr(ep, C.c_double)
    sd = None if seeds is None else np.ascontiguousarray(np.asarray(seeds, dtype=np.int64) & 0xFFFFFFFF, dtype=np.uint32)
    if sd is not None and sd.shape[0] != P:
        raise capi.TheiaHipError(-1, "seeds must hold one entry per problem")
    b.seeds = None if sd is None else sd.ctypes.data_as(C.POINTER(C.c_uint32))
    success = np.zeros(max(P, 1), dtype=np.int32); models = np.zeros((max(P, 1), capi.THEIA_RANSAC_MODEL_STRIDE))
    ninl = np.zeros(max(P, 1), dtype=np.int32); mask = np.zeros(max(total, 1), dtype=np.uint8)
    nit = np.zeros(max(P, 1), dtype=np.int32); conf = np.zeros(max(P, 1)); nlo = np.zeros(max(P, 1), dtype=np.int32)
    r = capi.RansacResult()
    r.num_lo_iterations = capi.ptr(nlo, C.c_int32)
    r.success = capi.ptr(success, C.c_int32); r.models = capi.ptr(models, C.c_double)
    r.num_inliers = capi.ptr(ninl, C.c_int32); r.inlier_mask = capi.ptr(mask, C.c_uint8)
    r.num_iterations = capi.ptr(nit, C.c_int32); r.confidence = capi.ptr(conf, C.c_double)
    pc = params.to_c() if isinstance(params, RansacParameters) else params
    capi.check(L.theia_hip_ransac_estimate_batch(C.byref(b), C.byref(pc), C.byref(r)))
    return {"success": success[:P], "models": models[:P], "num_inliers": ninl[:P], "inlier_mask": mask[:total],
            "num_iterations": nit[:P], "confidence": conf[:P], "num_lo_iterations": nlo[:P], "hypotheses_evaluated": r.hypotheses_evaluated,
            "models_scored": r.models_scored, "time_fit_score_seconds": r.time_fit_score_seconds,
            "time_fit_seconds": r.time_fit_seconds, "time_score_seconds": r.time_score_seconds}


_SAMPLE_SIZE = {0: 5, 1: 5, 2: 3, 3: 3, 4: 3, 5: 8, 6: 4, 7: 3, 8: 2, 9: 8, 10: 2, 11: 2, 12: 6, 13: 4, 14: 4, 15: 4, 16: 4}   # Estimator::SampleSize() by THEIA_EST_*


def _single(estimator, ransac_params, ransac_type, data, estimator_params=None):
    data = np.ascontiguousarray(data, dtype=np.float64)
    pc = ransac_params.to_c()
    pc.ransac_type = int(RansacType(ransac_type))
    if data.shape[0] < _SAMPLE_SIZE[estimator]:
        # a single Estimate() call: the reference's sampler CHECKs (random_sampler.cc:53-58); in a batch the C-ABI fails that pair only
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, "fewer data than the minimal sample size")
    res = estimate_batch(estimator, data, np.array([0, data.shape[0]], dtype=np.int64), pc, estimator_params)
    s = RansacSummary()
    s.inliers = np.nonzero(res["inlier_mask"])[0].tolist()
    s.num_input_data_points = data.shape[0]
    s.num_iterations = int(res["num_iterations"][0])
    s.confidence = float(res["confidence"][0])
    s.num_lo_iterations = int(res["num_lo_iterations"][0])
    return bool(res["success"][0]), res["models"][0], s


def EstimateRelativePose(ransac_params, ransac_type, normalized_correspondences):
    """estimate_relative_pose.cc:159-172.  correspondences: (N,4) x1 y1 x2 y2."""
    ok, m, s = _single(EST_RELATIVE_POSE, ransac_params, ransac_type, normalized_correspondences)
    return ok, RelativePose(m), s


def EstimateEssentialMatrix(ransac_params, ransac_type, normalized_correspondences):
    """estimate_essential_matrix.cc:90-106."""
    ok, m, s = _single(EST_ESSENTIAL_MATRIX, ransac_params, ransac_type, normalized_correspondences)
    return ok, m[0:9].reshape(3, 3).copy(), s


def EstimateCalibratedAbsolutePose(ransac_params, ransac_type, pnp_type, normalized_correspondences):
    """estimate_calibrated_absolute_pose.cc:176-190.  correspondences: (N,5) u v X Y Z."""
    est = {PnPType.KNEIP: EST_ABS_KNEIP, PnPType.DLS: EST_ABS_DLS, PnPType.SQPnP: EST_ABS_SQPNP}[PnPType(pnp_type)]
    ok, m, s = _single(est, ransac_params, ransac_type, normalized_correspondences)
    return ok, CalibratedAbsolutePose(m), s


class Plane:  # estimate_dominant_plane_from_points.h:48-51
    def __init__(self, m):
        self.point = m[0:3].copy()
        self.unit_normal = m[3:6].copy()


def EstimateFundamentalMatrix(ransac_params, ransac_type, correspondences):
    """estimate_fundamental_matrix.cc:105-121.  correspondences: (N,4) x1 y1 x2 y2 in pixels;
    error_thresh is a squared Sampson distance in pixels^2."""
    ok, m, s = _single(EST_FUNDAMENTAL_MATRIX, ransac_params, ransac_type, correspondences)
    return ok, m[0:9].reshape(3, 3).copy(), s


def EstimateHomography(ransac_params, ransac_type, correspondences):
    """estimate_homography.cc:120-135.  correspondences: (N,4) x1 y1 x2 y2; x2 ~ H x1."""
    ok, m, s = _single(EST_HOMOGRAPHY, ransac_params, ransac_type, correspondences)
    return ok, m[0:9].reshape(3, 3).copy(), s


def EstimateDominantPlaneFromPoints(ransac_params, ransac_type, points):
    """estimate_dominant_plane_from_points.cc:95-107.  points: (N,3)."""
    ok, m, s = _single(EST_DOMINANT_PLANE, ransac_params, ransac_type, points)
    return ok, Plane(m), s


def EstimateRelativePoseWithKnownOrientation(ransac_params, ransac_type, rotated_correspondences):
    """estimate_relative_pose_with_known_orientation.cc:66-81.  correspondences: (N,4), features
    rotated into a common frame; returns the unit position of camera 2."""
    ok, m, s = _single(EST_RELATIVE_POSE_KNOWN_ORIENTATION, ransac_params, ransac_type, rotated_correspondences)
    return ok, m[0:3].copy(), s


class UncalibratedRelativePose:  # estimate_uncalibrated_relative_pose.h:51-57
    def __init__(self, m):
        self.fundamental_matrix = m[0:9].reshape(3, 3).copy()
        self.rotation = m[9:18].reshape(3, 3).copy()
        self.position = m[18:21].copy()
        self.focal_length1 = float(m[21])
        self.focal_length2 = float(m[22])


def EstimateUncalibratedRelativePose(ransac_params, ransac_type, centered_correspondences, min_max_focal_length=(1.0, 1.7976931348623157e308)):
    """estimate_uncalibrated_relative_pose.cc:204-220.  correspondences: (N,4) pixels with the
    principal point removed."""
    ok, m, s = _single(EST_UNCALIBRATED_RELATIVE_POSE, ransac_params, ransac_type, centered_correspondences,
                       np.asarray(min_max_focal_length, dtype=np.float64))
    return ok, UncalibratedRelativePose(m), s


def RotateCorrespondences(normalized_correspondences, camera_orientation):
    """estimate_absolute_pose_with_known_orientation.cc:53-72: features into the world frame with the
    camera-to-world rotation (transpose of the angle-axis world-to-camera rotation)."""
    c = np.ascontiguousarray(normalized_correspondences, dtype=np.float64).reshape(-1, 5)
    R = synth.angle_axis_to_matrix(np.asarray(camera_orientation, dtype=np.float64))
    ray = np.column_stack([c[:, :2], np.ones(len(c))]) @ R   # rows: R^T [u v 1]
    out = c.copy()
    out[:, :2] = ray[:, :2] / ray[:, 2:]
    return out


def EstimateAbsolutePoseWithKnownOrientation(ransac_params, ransac_type, camera_orientation, normalized_correspondences):
    """estimate_absolute_pose_with_known_orientation.cc:132-153 -> (success, camera_position, summary)."""
    rot = RotateCorrespondences(normalized_correspondences, camera_orientation)
    ok, m, s = _single(EST_ABSOLUTE_POSE_KNOWN_ORIENTATION, ransac_params, ransac_type, rot)
    return ok, m[0:3].copy(), s


class Camera:  # the slice of sfm/camera/camera.h the triangulation estimator reads
    """position (3), angle-axis orientation (3), THEIA_CAM_* model id and that model's intrinsics (up to 10, the
    C-ABI's layout: focal, aspect ratio, skew, principal point x y, then the model's distortion parameters)."""

    def __init__(self, position, orientation, intrinsics, model=0):
        self.position = np.asarray(position, dtype=np.float64).reshape(3)
        self.orientation = np.asarray(orientation, dtype=np.float64).reshape(3)
        self.model = int(model)
        k = np.asarray(intrinsics, dtype=np.float64).ravel()
        if k.shape[0] > 10:
            raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, "a camera model holds at most 10 intrinsics")
        self.intrinsics = np.zeros(10)
        self.intrinsics[:k.shape[0]] = k

    def projection_matrix(self):
        """[R | -R c] (estimate_triangulation.cc:127-133: the calibration is NOT part of it)."""
        R = synth.angle_axis_to_matrix(self.orientation)
        return np.column_stack([R, -R @ self.position])

    def pixel_to_normalized(self, pixel):
        """Camera::PixelToNormalizedCoordinates(pixel).hnormalized() for the pinhole model
        (pinhole_camera_model.h:208-241: remove the calibration, then the fixed-point undistortion of :263-298)."""
        if self.model != 0:
            raise capi.TheiaHipError(capi.THEIA_HIP_ERR_UNSUPPORTED,
                                     "pixel_to_normalized mirrors the pinhole model only; pass normalized_features for the others")
        f, a, s, cx, cy, k1, k2 = self.intrinsics[:7]
        y = (float(pixel[1]) - cy) / (f * a)
        x = (float(pixel[0]) - cx - y * s) / f
        ux, uy = x, y
        for _ in range(100):
            px, py = ux, uy
            r2 = ux * ux + uy * uy
            d = 1.0 + r2 * (k1 + k2 * r2)
            ux, uy = x / d, y / d
            if abs(ux - px) < 1e-10 and abs(uy - py) < 1e-10:
                break
        return np.array([ux, uy])


def triangulation_observations(cameras, features, normalized_features=None):
    """The THEIA_EST_TRIANGULATION datum of every observation (theia_hip.h): the PointObservation of
    estimate_triangulation.cc:55-60 flattened to 33 doubles."""
    n = len(cameras)
    feats = np.asarray(features, dtype=np.float64).reshape(n, 2)
    out = np.zeros((n, 33))
    for i, cam in enumerate(cameras):
        out[i, 0:12] = cam.projection_matrix().ravel()
        out[i, 12:14] = cam.pixel_to_normalized(feats[i]) if normalized_features is None else np.asarray(normalized_features[i], dtype=np.float64)[:2]
        out[i, 14:16] = feats[i]
        out[i, 16:19] = cam.position
        out[i, 19:22] = cam.orientation
        out[i, 22] = cam.model
        out[i, 23:33] = cam.intrinsics
    return out


def EstimateTriangulation(ransac_params, cameras, features, normalized_features=None):
    """estimate_triangulation.cc:111-166 -> (success, triangulated_point (4, homogeneous), summary).  Up to 15
    observations: EXHAUSTIVE with min = max iterations = n (n - 1) / 2 (:143-153), otherwise plain RANSAC."""
    if len(cameras) != len(features):
        raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, "one feature per camera")   # CHECK_EQ :116
    n = len(cameras)
    if n < 2:
        return False, np.zeros(4), RansacSummary()   # :122-124
    data = triangulation_observations(cameras, features, normalized_features)
    params, rtype = ransac_params, RansacType.RANSAC
    if n <= 15:
        params = copy.copy(ransac_params)
        params.min_iterations = params.max_iterations = n * (n - 1) // 2
        rtype = RansacType.EXHAUSTIVE
    ok, m, s = _single(EST_TRIANGULATION, params, rtype, data)
    return ok, m[0:4].copy(), s


def EstimateTriangulationBatch(ransac_params, tracks):
    """Many EstimateTriangulation calls in a few launches (what TrackEstimator::EstimateTracks,
    sfm/track_estimator.cc, does track by track on its thread pool).  tracks: sequence of (cameras, features[,
    normalized_features]).  Tracks with the same observation count n <= 15 share one EXHAUSTIVE batch (iterations =
    n (n - 1) / 2 each, as a single call would set), all longer tracks share one RANSAC batch; problem i draws from
    RandomNumberGenerator(ransac_params.seed + i).  Returns (success [T], points [T, 4], inlier index lists)."""
    T = len(tracks)
    success = np.zeros(T, dtype=bool); points = np.zeros((T, 4)); inliers = [[] for _ in range(T)]
    data, groups = [None] * T, {}
    for i, tr in enumerate(tracks):
        n = len(tr[0])
        if n != len(tr[1]):
            raise capi.TheiaHipError(capi.THEIA_HIP_ERR_INVALID_ARGUMENT, "one feature per camera")
        if n < 2:
            continue
        data[i] = triangulation_observations(tr[0], tr[1], tr[2] if len(tr) > 2 else None)
        groups.setdefault(n if n <= 15 else 0, []).append(i)
    for n, idx in sorted(groups.items()):
        pc = ransac_params.to_c()
        pc.ransac_type = int(RansacType.RANSAC)
        if n:
            pc.min_iterations = pc.max_iterations = n * (n - 1) // 2
            pc.ransac_type = int(RansacType.EXHAUSTIVE)
        offsets = np.concatenate([[0], np.cumsum([data[i].shape[0] for i in idx])]).astype(np.int64)
        res = estimate_batch(EST_TRIANGULATION, np.concatenate([data[i] for i in idx]), offsets, pc,
                             seeds=[int(ransac_params.seed) + i for i in idx])
        for k, i in enumerate(idx):
            success[i] = bool(res["success"][k]); points[i] = res["models"][k][:4]
            inliers[i] = np.nonzero(res["inlier_mask"][offsets[k]:offsets[k + 1]])[0].tolist()
    return success, points, inliers


class RadialDistortionFeatureCorrespondence:  # estimate_radial_distortion_homography.h:56-70, same fields and defaults
    def __init__(self, feature_left=(0.0, 0.0), feature_right=(0.0, 0.0), normalized_feature_left=(0.0, 0.0),
                 normalized_feature_right=(0.0, 0.0)):
        self.feature_left = np.asarray(feature_left, dtype=np.float64)
        self.feature_right = np.asarray(feature_right, dtype=np.float64)
        self.normalized_feature_left = np.asarray(normalized_feature_left, dtype=np.float64)
        self.normalized_feature_right = np.asarray(normalized_feature_right, dtype=np.float64)
        self.focal_length_estimate_left = 1000.0
        self.focal_length_estimate_right = 1000.0
        self.min_radial_distortion = -5.0
        self.max_radial_distortion = 0.0

    def row(self):
        return np.concatenate([self.feature_left, self.feature_right, self.normalized_feature_left, self.normalized_feature_right,
                               [self.focal_length_estimate_left, self.focal_length_estimate_right,
                                self.min_radial_distortion, self.max_radial_distortion]])


class RadialHomographyResult:  # six_point_radial_distortion_homography.h: H, l1, l2
    def __init__(self, m):
        self.H = np.array(m[0:9]).reshape(3, 3)
        self.l1 = float(m[9])
        self.l2 = float(m[10])


def radial_correspondence_rows(correspondences):
    """(N, 12) rows of THEIA_EST_RADIAL_HOMOGRAPHY from a list of RadialDistortionFeatureCorrespondence (or an array
    already in that layout)."""
    if isinstance(correspondences, np.ndarray):
        return np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 12)
    return np.array([c.row() for c in correspondences], dtype=np.float64).reshape(-1, 12)


def EstimateRadialHomographyMatrix(ransac_params, ransac_type, normalized_correspondences):
    """estimate_radial_distortion_homography.cc:90-111 -> (success, RadialHomographyResult, summary)."""
    ok, m, s = _single(EST_RADIAL_HOMOGRAPHY, ransac_params, ransac_type, radial_correspondence_rows(normalized_correspondences))
    return ok, RadialHomographyResult(m), s


class SimilarityTransformation:  # sfm/similarity_transformation.h: rotation, translation, scale
    def __init__(self, m):
        self.rotation = np.array(m[0:9]).reshape(3, 3)
        self.translation = np.array(m[9:12])
        self.scale = float(m[12])


class CameraAndFeatureCorrespondence2D3D:  # estimate_similarity_transformation_2d_3d.h: camera, observation, point3d
    def __init__(self, camera, observation, point3d):
        self.camera = camera
        self.observation = np.asarray(observation, dtype=np.float64).reshape(2)
        p = np.asarray(point3d, dtype=np.float64).ravel()
        self.point3d = p if p.shape[0] == 4 else np.append(p, 1.0)


def pixel_to_unit_depth_ray(camera, pixel):
    """Camera::PixelToUnitDepthRay (camera.cc:232-243): R^T [normalized coordinates; 1] (pinhole mirror of the unprojection)."""
    n = camera.pixel_to_normalized(pixel)
    return synth.angle_axis_to_matrix(camera.orientation).T @ np.array([n[0], n[1], 1.0])


def similarity_correspondence_rows(correspondences, ray_directions=None):
    """(N, 26) rows of THEIA_EST_SIMILARITY_2D3D.  ray_directions: optional (N, 3) world-frame rays of the observations
    (any length) for camera models whose unprojection the mirror does not carry."""
    out = np.zeros((len(correspondences), 26))
    for i, c in enumerate(correspondences):
        ray = pixel_to_unit_depth_ray(c.camera, c.observation) if ray_directions is None else np.asarray(ray_directions[i], dtype=np.float64)
        out[i, 0:3] = ray / np.linalg.norm(ray)
        out[i, 3:7] = c.point3d
        out[i, 7:9] = c.observation
        out[i, 9:12] = c.camera.position
        out[i, 12:15] = c.camera.orientation
        out[i, 15] = c.camera.model
        out[i, 16:26] = c.camera.intrinsics
    return out


def EstimateSimilarityTransformation2D3D(ransac_params, ransac_type, correspondences, ray_directions=None):
    """estimate_similarity_transformation_2d_3d.cc:161-178 -> (success, SimilarityTransformation, summary)."""
    rows = correspondences if isinstance(correspondences, np.ndarray) else similarity_correspondence_rows(correspondences, ray_directions)
    ok, m, s = _single(EST_SIMILARITY_2D3D, ransac_params, ransac_type, rows)
    return ok, SimilarityTransformation(m), s


class RigidTransformation:  # sfm/rigid_transformation.h: rotation, translation
    def __init__(self, m):
        self.rotation = np.array(m[0:9]).reshape(3, 3)
        self.translation = np.array(m[9:12])


def central_correspondence_rows(normalized_correspondences):
    """(N, 26) rows for the FeatureCorrespondence2D3D overload of EstimateRigidTransformation2D3D
    (estimate_rigid_transformation_2d_3d.cc:159-182): [u v X Y Z] seen by identity pinhole cameras of focal length 1."""
    c = np.ascontiguousarray(normalized_correspondences, dtype=np.float64).reshape(-1, 5)
    out = np.zeros((c.shape[0], 26))
    ray = np.concatenate([c[:, 0:2], np.ones((c.shape[0], 1))], axis=1)
    out[:, 0:3] = ray / np.linalg.norm(ray, axis=1, keepdims=True)
    out[:, 3:6] = c[:, 2:5]; out[:, 6] = 1.0
    out[:, 7:9] = c[:, 0:2]
    out[:, 15] = 0                      # THEIA_CAM_PINHOLE
    out[:, 16] = 1.0; out[:, 17] = 1.0  # focal length, aspect ratio
    return out


def EstimateRigidTransformation2D3D(ransac_params, ransac_type, correspondences, ray_directions=None):
    """estimate_rigid_transformation_2d_3d.cc:137-182 -> (success, RigidTransformation, summary).  correspondences: a list of
    CameraAndFeatureCorrespondence2D3D, their (N, 26) rows, or (N, 5) normalised FeatureCorrespondence2D3D [u v X Y Z] (the
    central-camera overload).  UPnP on four correspondences per sample; as in the reference the estimator's cost parameters
    accumulate over the samples of the call (include/theia_hip.h)."""
    if isinstance(correspondences, np.ndarray):
        rows = central_correspondence_rows(correspondences) if correspondences.shape[1] == 5 else correspondences
    else:
        rows = similarity_correspondence_rows(correspondences, ray_directions)
    ok, m, s = _single(EST_RIGID_TRANSFORMATION_2D3D, ransac_params, ransac_type, rows)
    return ok, RigidTransformation(m), s


class UncalibratedAbsolutePose:  # estimate_uncalibrated_absolute_pose.h:48-52
    def __init__(self, rotation, position, focal_length):
        self.rotation = rotation
        self.position = position
        self.focal_length = focal_length


def DecomposeProjectionMatrix(pmatrix):
    """sfm/camera/projection_matrix_utils.cc:74-117: P = K [R | -R c] -> (success, K, angle-axis rotation, position) by the
    RQ decomposition of the left 3 x 3 block, the rotation projected onto SO(3), K made positive on its diagonal."""
    import scipy.linalg
    P = np.asarray(pmatrix, dtype=np.float64).reshape(3, 4)
    Rk, Q = scipy.linalg.rq(P[:, :3])
    U, _, Vt = np.linalg.svd(Q)                      # ProjectToRotationMatrix (util/util.h / rotation utilities)
    Rm = U @ Vt
    if np.linalg.det(Rm) < 0:
        Rm = -Rm
    k_det = np.linalg.det(Rk)
    if k_det == 0:
        return False, np.zeros((3, 3)), np.zeros(3), np.zeros(3)
    K = Rk.copy() if k_det > 0 else -Rk
    for i in range(3):
        if K[i, i] < 0:
            K[:, i] *= -1.0
            Rm[i, :] *= -1.0
    t = scipy.linalg.solve_triangular(K, P[:, 3], lower=False)
    position = -Rm.T @ t if k_det > 0 else Rm.T @ t
    return True, K, synth.matrix_to_angle_axis(Rm[None])[0], position


def EstimateUncalibratedAbsolutePose(ransac_params, ransac_type, normalized_correspondences):
    """estimate_uncalibrated_absolute_pose.cc:106-141.  correspondences: (N, 5) u v X Y Z, pixels with the principal point
    removed.  RANSAC over P4Pf samples on the device, then DecomposeProjectionMatrix of the winning projection matrix."""
    ok, m, s = _single(EST_UNCALIBRATED_ABSOLUTE_POSE, ransac_params, ransac_type, normalized_correspondences)
    good, K, aa, position = DecomposeProjectionMatrix(m[:12])
    rotation = synth.angle_axis_to_matrix(aa[None])[0]
    return ok, UncalibratedAbsolutePose(rotation, position, K[0, 0] / K[2, 2] if good else 0.0), s


def FourPointPoseAndFocalLength(feature_vectors, world_points):
    """pose_wrapper.cc / four_point_focal_length.cc:100-222 (P4Pf, four 2D-3D correspondences per problem; batched when the
    inputs carry a leading batch dimension): (number of solutions, list of 3 x 4 projection matrices)."""
    a = np.asarray(feature_vectors, dtype=np.float64); b = np.asarray(world_points, dtype=np.float64)
    single = a.ndim == 2
    if single:
        a, b = a[None], b[None]
    corr = np.ascontiguousarray(np.concatenate([a, b], axis=2))
    num = corr.shape[0]
    Pm = np.zeros((num, 10, 3, 4)); ns = np.zeros(num, dtype=np.int32)
    capi.check(_sig().theia_hip_four_point_pose_and_focal_length(num, capi.ptr(corr, C.c_double), capi.ptr(Pm, C.c_double),
                                                                  capi.ptr(ns, C.c_int32)))
    if single:
        return (int(ns[0]) if ns[0] > 0 else -1), [Pm[0, k] for k in range(ns[0])]
    return ns, Pm


class RadialDistUncalibratedAbsolutePoseMetaData:  # estimate_radial_dist_uncalibrated_absolute_pose.h:55-61
    def __init__(self, min_focal_length=200.0, max_focal_length=10000.0, min_radial_distortion=-1e-9, max_radial_distortion=-1e-5):
        self.min_focal_length = min_focal_length
        self.max_focal_length = max_focal_length
        self.min_radial_distortion = min_radial_distortion
        self.max_radial_distortion = max_radial_distortion

    def limits(self):
        return np.array([self.max_focal_length, self.min_focal_length, self.max_radial_distortion, self.min_radial_distortion])


class RadialDistUncalibratedAbsolutePose:  # estimate_radial_dist_uncalibrated_absolute_pose.h:48-53
    def __init__(self, m):
        self.rotation = np.array(m[0:9]).reshape(3, 3)
        self.translation = np.array(m[9:12])
        self.focal_length = float(m[12])
        self.radial_distortion = float(m[13])


def shift_world_along_optical_axis(correspondences, offsets, rotations, shift):
    """X' = X - R^T (0, 0, shift) per problem: the same images under the pose (R, t + (0, 0, shift)).  The radial-distortion
    estimator's Error rejects every model whose translation has a negative z (estimate_radial_dist_uncalibrated_absolute_pose.cc:
    137-139), so a synthetic scene for it needs t_z >= 0; synth_ransac_v1's unit translations have t_z in [-1, 1]."""
    c = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 5).copy()
    for i in range(len(offsets) - 1):
        c[offsets[i]:offsets[i + 1], 2:5] -= rotations[i].T @ np.array([0.0, 0.0, shift])
    return c


def radial_dist_correspondence_rows(normalized_correspondences, focal_length, radial_distortion):
    """(N, 5) rows [u v X Y Z] of normalised (focal length 1, undistorted) correspondences as a camera of the given focal length
    and division-model distortion observes them: pixels f * (u, v), distorted as DistortPoint does
    (estimate_radial_dist_uncalibrated_absolute_pose.cc:56-74).  For synthetic workloads (bench.py, the parity tests)."""
    c = np.ascontiguousarray(normalized_correspondences, dtype=np.float64).reshape(-1, 5).copy()
    px = c[:, :2] * focal_length
    r2 = np.sum(px * px, axis=1)
    den = 2.0 * radial_distortion * r2; inner = 1.0 - 4.0 * radial_distortion * r2
    keep = (np.abs(den) < 1e-15) | (inner < 0.0)
    scale = np.where(keep, 1.0, (1.0 - np.sqrt(np.maximum(inner, 0.0))) / np.where(keep, 1.0, den))
    c[:, :2] = px * scale[:, None]
    return c


def EstimateRadialDistUncalibratedAbsolutePose(ransac_params, ransac_type, normalized_correspondences, meta_data, first_call_in_process=False):
    """estimate_radial_dist_uncalibrated_absolute_pose.cc:163-189 -> (success, RadialDistUncalibratedAbsolutePose, summary).
    correspondences: (N, 5) u v X Y Z, the observed (distorted) pixels with the principal point removed.  RANSAC over P4Pfr
    samples on the device.  The solver's "random rotation" draws come out of the sampler's stream, as in the reference
    (include/theia_hip.h); first_call_in_process: the stream is re-seeded with 42 after the first sample, which is what the
    solver's static generator does the first time it runs in a process."""
    ep = np.concatenate([meta_data.limits(), [1.0 if first_call_in_process else 0.0]])
    ok, m, s = _single(EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE, ransac_params, ransac_type, normalized_correspondences, ep)
    return ok, RadialDistUncalibratedAbsolutePose(m), s


def FourPointsPoseFocalLengthRadialDistortion(feature_vectors, world_points, meta_data, rotation_draws=None):
    """pose_wrapper.cc:189-215 / four_point_focal_length_radial_distortion.cc:68-288 (P4Pfr, four 2D-3D correspondences per problem;
    batched when the inputs carry a leading batch dimension): (success, rotations, translations, radial distortions, focal lengths);
    success is the reference's `valid_solutions.size() > 0` -- the solver's count BEFORE the focal-length / distortion range tests
    (:287), so it can be True with empty lists.  Batched: (num_solutions after the tests, models, solver counts before the
    tests: success[i] = counts[i] > 0).
    rotation_draws: the three RandDouble(-0.5, 0.5) of every call ((num, 3)); None = the calls of a fresh process in order."""
    a = np.asarray(feature_vectors, dtype=np.float64); b = np.asarray(world_points, dtype=np.float64)
    single = a.ndim == 2
    if single:
        a, b = a[None], b[None]
    corr = np.ascontiguousarray(np.concatenate([a, b], axis=2))
    num = corr.shape[0]
    lim = np.ascontiguousarray(meta_data.limits(), dtype=np.float64)
    rd = None if rotation_draws is None else np.ascontiguousarray(np.asarray(rotation_draws, dtype=np.float64).reshape(num, 3))
    M = np.zeros((num, 13, 14)); ns = np.zeros(num, dtype=np.int32); nsolver = np.zeros(num, dtype=np.int32)
    capi.check(_sig().theia_hip_four_point_focal_length_radial_distortion_ex(num, capi.ptr(corr, C.c_double), capi.ptr(lim, C.c_double),
                                                                           None if rd is None else capi.ptr(rd, C.c_double),
                                                                           capi.ptr(M, C.c_double), capi.ptr(ns, C.c_int32), capi.ptr(nsolver, C.c_int32)))
    if single:
        k = int(ns[0])
        # success = the solver found solutions, whether or not any passed the range tests (:287: valid_solutions.size() > 0)
        return bool(nsolver[0] > 0), [M[0, j, :9].reshape(3, 3) for j in range(k)], [M[0, j, 9:12] for j in range(k)], [M[0, j, 13] for j in range(k)], [M[0, j, 12] for j in range(k)]
    return ns, M, nsolver


def FivePointRelativePose(image1_points, image2_points):
    """pose_wrapper.cc:166-173 (minimal, 5 correspondences per problem; batched
    when the inputs carry a leading batch dimension)."""
    a = np.asarray(image1_points, dtype=np.float64); b = np.asarray(image2_points, dtype=np.float64)
    single = a.ndim == 2
    if single:
        a, b = a[None], b[None]
    corr = np.ascontiguousarray(np.concatenate([a, b], axis=2))
    num = corr.shape[0]
    E = np.zeros((num, 10, 3, 3)); ns = np.zeros(num, dtype=np.int32)
    capi.check(_sig().theia_hip_five_point_relative_pose(num, capi.ptr(corr, C.c_double), capi.ptr(E, C.c_double),
                                                          capi.ptr(ns, C.c_int32)))
    if single:
        return bool(ns[0] > 0), [E[0, k] for k in range(ns[0])]
    return ns, E


def PoseFromThreePoints(feature_points, points_3d):
    """sfm.cc:573 / perspective_three_point.cc (Kneip P3P)."""
    a = np.asarray(feature_points, dtype=np.float64); b = np.asarray(points_3d, dtype=np.float64)
    single = a.ndim == 2
    if single:
        a, b = a[None], b[None]
    corr = np.ascontiguousarray(np.concatenate([a, b], axis=2))
    num = corr.shape[0]
    R = np.zeros((num, 4, 3, 3)); t = np.zeros((num, 4, 3)); ns = np.zeros(num, dtype=np.int32)
    capi.check(_sig().theia_hip_pose_from_three_points(num, capi.ptr(corr, C.c_double), capi.ptr(R, C.c_double),
                                                        capi.ptr(t, C.c_double), capi.ptr(ns, C.c_int32)))
    if single:
        return bool(ns[0] > 0), [R[0, k] for k in range(ns[0])], [t[0, k] for k in range(ns[0])]
    return ns, R, t


def SQPnP(feature_positions, world_points):
    """sfm.cc:592 / sqpnp.cc:58-353.  One problem: (N, 2) and (N, 3) arrays ->
    (success, [quaternion wxyz], [translation]).  A list of problems is solved as one
    batch and returns (num_solutions, quaternions[num][18][4], translations[num][18][3])."""
    single = isinstance(feature_positions, np.ndarray) and feature_positions.ndim == 2
    fl = [np.asarray(feature_positions, dtype=np.float64)] if single else [np.asarray(f, dtype=np.float64) for f in feature_positions]
    wl = [np.asarray(world_points, dtype=np.float64)] if single else [np.asarray(w, dtype=np.float64) for w in world_points]
    if len(fl) != len(wl) or any(f.shape[0] != w.shape[0] for f, w in zip(fl, wl)):
        raise capi.TheiaHipError(-1, "feature_positions / world_points size mismatch")
    num = len(fl)
    offsets = np.zeros(num + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([f.shape[0] for f in fl])
    feat = np.ascontiguousarray(np.concatenate(fl, axis=0).reshape(-1, 2)) if num else np.zeros((0, 2))
    world = np.ascontiguousarray(np.concatenate(wl, axis=0).reshape(-1, 3)) if num else np.zeros((0, 3))
    q = np.zeros((num, 18, 4)); t = np.zeros((num, 18, 3)); ns = np.zeros(num, dtype=np.int32)
    capi.check(_sig().theia_hip_sqpnp(num, offsets.ctypes.data_as(C.POINTER(C.c_int64)), capi.ptr(feat, C.c_double),
                                      capi.ptr(world, C.c_double), capi.ptr(q, C.c_double), capi.ptr(t, C.c_double),
                                      capi.ptr(ns, C.c_int32)))
    if single:
        return bool(ns[0] > 0), [q[0, k] for k in range(ns[0])], [t[0, k] for k in range(ns[0])]
    return ns, q, t


def DlsPnp(feature_positions, world_points, call_index=None):
    """sfm.cc:577 / dls_pnp.cc:67-200.  One problem: (N, 2) and (N, 3) arrays (N >= 3) ->
    (success, [quaternion wxyz], [translation]).  A list of problems is solved as one batch and returns
    (num_solutions, quaternions[num][27][4], translations[num][27][3]).  call_index: which DlsPnp call of a process each
    problem stands for (it selects the four std::rand() draws of the reference's random Macaulay terms; default: 0 for a
    single problem, 0, 1, 2, ... for a list)."""
    single = isinstance(feature_positions, np.ndarray) and feature_positions.ndim == 2
    fl = [np.asarray(feature_positions, dtype=np.float64)] if single else [np.asarray(f, dtype=np.float64) for f in feature_positions]
    wl = [np.asarray(world_points, dtype=np.float64)] if single else [np.asarray(w, dtype=np.float64) for w in world_points]
    if len(fl) != len(wl) or any(f.shape[0] != w.shape[0] for f, w in zip(fl, wl)):
        raise capi.TheiaHipError(-1, "feature_positions / world_points size mismatch")
    if any(f.shape[0] < 3 for f in fl):
        raise capi.TheiaHipError(-1, "Check failed: feature_position.size() >= 3")   # dls_pnp.cc:71
    num = len(fl)
    offsets = np.zeros(num + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([f.shape[0] for f in fl])
    feat = np.ascontiguousarray(np.concatenate(fl, axis=0).reshape(-1, 2)) if num else np.zeros((0, 2))
    world = np.ascontiguousarray(np.concatenate(wl, axis=0).reshape(-1, 3)) if num else np.zeros((0, 3))
    ci = None
    if call_index is not None:
        ci = np.ascontiguousarray(np.atleast_1d(np.asarray(call_index, dtype=np.int64)))
        if ci.shape[0] != num:
            raise capi.TheiaHipError(-1, "call_index: one entry per problem")
    q = np.zeros((num, 27, 4)); t = np.zeros((num, 27, 3)); ns = np.zeros(num, dtype=np.int32)
    capi.check(_sig().theia_hip_dls_pnp(num, offsets.ctypes.data_as(C.POINTER(C.c_int64)), capi.ptr(feat, C.c_double),
                                        capi.ptr(world, C.c_double), ci.ctypes.data_as(C.POINTER(C.c_int64)) if ci is not None else None,
                                        capi.ptr(q, C.c_double), capi.ptr(t, C.c_double), capi.ptr(ns, C.c_int32)))
    if single:
        return bool(ns[0] > 0), [q[0, k] for k in range(ns[0])], [t[0, k] for k in range(ns[0])]
    return ns, q, t


def dls_macaulay_terms(first_call, num_calls):
    """The reference's 100 * Eigen::Vector4d::Random() of DlsPnp calls [first_call, first_call + num_calls) of a process
    that never seeded rand() (dls_pnp.cc:134): (num_calls, 4).  Host only."""
    out = np.zeros((int(num_calls), 4))
    _sig().theia_hip_dls_macaulay_terms(int(first_call), int(num_calls), capi.ptr(out, C.c_double))
    return out
