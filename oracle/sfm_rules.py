"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by pytheiasfm_amd/ or bench.py's timed region): sequential
restatements of two host-side rule sets of the reference that sit on either side of the hot path and whose product-side
mirrors batch them:

  * SelectGoodTracksForBundleAdjustment   (src/theia/sfm/select_good_tracks_for_bundle_adjustment.cc:79-320)
  * TwoViewMatchGeometricVerification::VerifyMatches, its guided-matching branch included
                                          (src/theia/sfm/two_view_match_geometric_verification.cc:114-368,
                                           src/theia/matching/guided_epipolar_matcher.cc:92-450)

Both are written one element at a time in the order the reference walks (per track, per view, per match), with Python
containers standing in for the reference's (dict = unordered_map / ImageGrid, list = vector) and the numerical pieces --
camera projection, RANSAC, the two-view bundle adjustment -- taken from the C oracle (oracle/ba_oracle.cpp,
oracle/ransac_oracle.cpp) through the ctypes wrapper the caller passes in (`ol` = tests/oracle_lib).  Nothing here shares
code with pytheiasfm_amd/sfm.py or pytheiasfm_amd/twoview.py.

Parity status: unpinned against the reference binary (it cannot be built here: Ceres / Eigen / glog are absent); pinned by
the hand-made known-answer cases of tests/test_oracle_sfm_rules.py.  Where the reference iterates a hash container
(unordered_set<ViewId>, unordered_map<Vector2i, ..>) the result does not depend on the order except through the early
return of SelectTopRankedTracksInView, which this file walks in ascending track id (stated in the function).
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------------------------
# select_good_tracks_for_bundle_adjustment.cc
def _sq_reprojection_error(ol, rec, view, obs_index):
    """ComputeSqReprojectionError (:68-75): Camera::ProjectPoint of the track's point, squared pixel distance."""
    t = int(rec.obs_track[obs_index])
    g = int(rec.view_group[view])
    _, res, _, _, _ = ol.reprojection_error(int(rec.group_model[g]), rec.cam_ext[view], rec.group_intrinsics[g], rec.points[t],
                                            rec.obs_uv[obs_index])
    return float(res[0] * res[0] + res[1] * res[1])


def compute_track_statistics(ol, rec, view_ids, long_track_length_threshold):
    """ComputeTrackStatistics (:118-146) + ComputeStatisticsForTrack (:79-110): for every estimated track seen from one of
    `view_ids`, (min(#estimated views, threshold), mean squared reprojection error over its estimated views)."""
    obs_of_track = {}
    for o in range(len(rec.obs_track)):
        obs_of_track.setdefault(int(rec.obs_track[o]), []).append(o)
    obs_of_view = {}
    for o in range(len(rec.obs_view)):
        obs_of_view.setdefault(int(rec.obs_view[o]), []).append(o)
    stats = {}
    for v in view_ids:
        for o in obs_of_view.get(int(v), []):
            t = int(rec.obs_track[o])
            if t in stats or not rec.track_estimated[t]:
                continue
            ssum, nvalid = 0.0, 0
            for o2 in obs_of_track[t]:                      # the views observing the track (:92-101)
                v2 = int(rec.obs_view[o2])
                if not rec.view_estimated[v2]:
                    continue
                ssum += _sq_reprojection_error(ol, rec, v2, o2)
                nvalid += 1
            stats[t] = (min(nvalid, long_track_length_threshold), ssum / float(nvalid) if nvalid else float("nan"))
    return stats, obs_of_view


def select_good_tracks_for_bundle_adjustment(ol, rec, view_ids, long_track_length_threshold, image_grid_cell_size_pixels,
                                             min_num_optimized_tracks_per_view):
    """SelectGoodTracksForBundleAdjustment (:280-320) -> sorted list of track ids.
    rec: anything with obs_view / obs_track / obs_uv / view_estimated / track_estimated / cam_ext / view_group /
    group_model / group_intrinsics / points (the array-backed stand-in of the tests)."""
    view_ids = [int(v) for v in view_ids]
    stats, obs_of_view = compute_track_statistics(ol, rec, view_ids, long_track_length_threshold)
    chosen = set()
    inv = 1.0 / image_grid_cell_size_pixels
    # SelectBestTracksFromEachImageGridCell (:152-191)
    for v in view_ids:
        grid = {}
        for o in obs_of_view.get(v, []):
            t = int(rec.obs_track[o])
            if not rec.track_estimated[t]:
                continue
            cell = (int(rec.obs_uv[o][0] * inv), int(rec.obs_uv[o][1] * inv))   # Eigen cast<int>: truncation toward zero
            grid.setdefault(cell, []).append((t, stats[t]))
        for cell, elems in grid.items():
            best = elems[0]
            for e in elems[1:]:                              # std::min_element with element.second < element.second (:62-66)
                if e[1] < best[1]:
                    best = e
            chosen.add(best[0])
    # SelectTopRankedTracksInView (:195-250); the reference walks view.TrackIds() (a hash container): ascending ids here
    for v in view_ids:
        num_optimized, num_estimated = 0, 0
        candidates = []
        done = False
        for t in sorted(int(rec.obs_track[o]) for o in obs_of_view.get(v, [])):
            if not rec.track_estimated[t]:
                continue
            num_estimated += 1
            if t in chosen:
                num_optimized += 1
                if num_optimized >= min_num_optimized_tracks_per_view:
                    done = True
                    break
            else:
                candidates.append((t, stats[t]))
        if done:
            continue
        if num_optimized != num_estimated:
            needed = min(min_num_optimized_tracks_per_view - num_optimized, num_estimated - num_optimized)
            candidates.sort()                                # partial_sort with pair<TrackId, TrackStatistics>::operator< (:246-249)
            for i in range(needed):
                chosen.add(candidates[i][0])
    return sorted(chosen)


# ------------------------------------------------------------------------------------------------------------------
# two_view_match_geometric_verification.cc
def resolution_scaled_threshold(threshold_pixels, image_width, image_height):
    """ComputeResolutionScaledThreshold (reconstruction_estimator_utils.cc:98-110)."""
    if image_width == 0 and image_height == 0:
        return threshold_pixels
    return threshold_pixels * float(max(image_width, image_height)) / 1024.0


def _pinhole_of_prior(prior):
    """PinholeCameraModel::SetFromCameraIntrinsicsPriors on a default model (pinhole_camera_model.cc:74-107): focal length
    from the prior, else 1.2 max(w, h) when the image size is known; principal point from the prior, else the centre."""
    f, px, py = 1.0, 0.0, 0.0
    if prior.focal_length.is_set:
        f = prior.focal_length.value[0]
    elif prior.image_width != 0 and prior.image_height != 0:
        f = 1.2 * float(max(prior.image_width, prior.image_height))
    if prior.principal_point.is_set:
        px, py = prior.principal_point.value[0], prior.principal_point.value[1]
    elif prior.image_width != 0 and prior.image_height != 0:
        px, py = prior.image_width / 2.0, prior.image_height / 2.0
    ar = prior.aspect_ratio.value[0] if prior.aspect_ratio.is_set else 1.0
    skew = prior.skew.value[0] if prior.skew.is_set else 0.0
    return [f, ar, skew, px, py, 0.0, 0.0]


def _normalize(prior1, prior2, corr):
    """NormalizeFeatures (estimate_twoview_info.cc:67-102): pixel -> camera coordinates of a distortion-free pinhole camera;
    both focal lengths are 1 unless both priors carry one."""
    k1, k2 = _pinhole_of_prior(prior1), _pinhole_of_prior(prior2)
    if not (prior1.focal_length.is_set and prior2.focal_length.is_set):
        k1[0] = 1.0; k2[0] = 1.0
    out = np.zeros((len(corr), 4))
    for i, c in enumerate(corr):
        for k, kk in enumerate((k1, k2)):
            y = (c[2 * k + 1] - kk[4]) / (kk[0] * kk[1])
            x = (c[2 * k] - kk[3] - y * kk[2]) / kk[0]
            out[i, 2 * k] = x; out[i, 2 * k + 1] = y
    return out


def _rotation_matrix_to_angle_axis(R):
    """Eigen::AngleAxisd(Matrix3d) (through the quaternion), as estimate_twoview_info.cc:180-181 stores rotation_2."""
    t = R[0][0] + R[1][1] + R[2][2]
    q = [0.0, 0.0, 0.0, 0.0]   # w, x, y, z
    if t > 0:
        s = math.sqrt(t + 1.0)
        q[0] = 0.5 * s
        s = 0.5 / s
        q[1] = (R[2][1] - R[1][2]) * s; q[2] = (R[0][2] - R[2][0]) * s; q[3] = (R[1][0] - R[0][1]) * s
    else:
        i = 0
        if R[1][1] > R[0][0]:
            i = 1
        if R[2][2] > R[i][i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0)
        q[1 + i] = 0.5 * s
        s = 0.5 / s
        q[0] = (R[k][j] - R[j][k]) * s
        q[1 + j] = (R[j][i] + R[i][j]) * s
        q[1 + k] = (R[k][i] + R[i][k]) * s
    n = math.sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    if q[0] < 0:
        n = -n
    if abs(n) < 1e-300:
        return np.zeros(3)
    angle = 2.0 * math.atan2(n, abs(q[0]))
    return np.array([q[1] / n, q[2] / n, q[3] / n]) * angle


def _ransac_params(ol, eo, error_thresh, use_mle, seed):
    p = ol.default_ransac_params(error_thresh, seed=seed)
    p.failure_probability = 1.0 - eo.expected_ransac_confidence
    p.min_iterations = eo.min_ransac_iterations
    p.max_iterations = eo.max_ransac_iterations
    p.use_mle = 1 if use_mle else 0
    p.ransac_type = int(eo.ransac_type)
    return p


EST_RELATIVE_POSE, EST_HOMOGRAPHY, EST_UNCALIBRATED = 0, 6, 9   # THEIA_EST_* (include/theia_hip.h)


def estimate_two_view_info(ol, eo, prior1, prior2, corr):
    """EstimateTwoViewInfo (estimate_twoview_info.cc:262-305) -> (ok, info dict, inlier indices)."""
    calibrated = prior1.focal_length.is_set and prior2.focal_length.is_set
    t1 = resolution_scaled_threshold(eo.max_sampson_error_pixels, prior1.image_width, prior1.image_height)
    t2 = resolution_scaled_threshold(eo.max_sampson_error_pixels, prior2.image_width, prior2.image_height)
    thresh = t1 * t2
    data = _normalize(prior1, prior2, corr)
    info = {"focal_length_1": 0.0, "focal_length_2": 0.0, "rotation_2": np.zeros(3), "position_2": np.zeros(3),
            "num_verified_matches": 0, "num_homography_inliers": 0}
    if calibrated:   # EstimateTwoViewInfoCalibrated (:133-198)
        thresh = thresh / (prior1.focal_length.value[0] * prior2.focal_length.value[0])
        p = _ransac_params(ol, eo, thresh, eo.use_mle, eo.seed)
        p.use_lo = 1 if eo.use_lo else 0; p.lo_start_iterations = eo.lo_start_iterations
        r = ol.ransac_estimate(EST_RELATIVE_POSE, data, p)
    else:            # EstimateTwoViewInfoUncalibrated (:200-258): use_mle is never copied
        ol.set_estimator_params([eo.min_focal_length, eo.max_focal_length])
        p = _ransac_params(ol, eo, thresh, False, eo.seed)
        p.use_lo = 1 if eo.use_lo else 0; p.lo_start_iterations = eo.lo_start_iterations
        r = ol.ransac_estimate(EST_UNCALIBRATED, data, p)
    if not r["success"]:
        return False, info, []
    m = r["model"]
    info["rotation_2"] = _rotation_matrix_to_angle_axis(np.asarray(m[9:18]).reshape(3, 3))
    info["position_2"] = np.array(m[18:21])
    if calibrated:
        info["focal_length_1"] = prior1.focal_length.value[0]; info["focal_length_2"] = prior2.focal_length.value[0]
    else:
        info["focal_length_1"] = float(m[21]); info["focal_length_2"] = float(m[22])
    inliers = [i for i in range(len(corr)) if r["inlier_mask"][i]]
    info["num_verified_matches"] = len(inliers)
    return True, info, inliers


def _angle_axis_to_matrix(w):
    th = math.sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2])
    if th < 1e-300:
        return np.eye(3)
    k = [w[0] / th, w[1] / th, w[2] / th]
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K)


def _unit_ray(ext, intr, px):
    """Camera::PixelToUnitDepthRay(pixel).normalized() (camera.cc:177-196) of a distortion-free pinhole camera."""
    y = (px[1] - intr[4]) / (intr[0] * intr[1])
    x = (px[0] - intr[3] - y * intr[2]) / intr[0]
    d = _angle_axis_to_matrix(ext[3:6]).T @ np.array([x, y, 1.0])
    return d / math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])


def _triangulate_midpoint(origins, rays):
    """TriangulateMidpoint (triangulation.cc:130-157): sum_i (I - d d^T) X = sum_i (I - d d^T) o in homogeneous form, LLT."""
    A = np.zeros((4, 4)); b = np.zeros(4)
    for o, d in zip(origins, rays):
        dh = np.array([d[0], d[1], d[2], 0.0])
        T = np.eye(4) - np.outer(dh, dh)
        A += T
        b += T @ np.array([o[0], o[1], o[2], 1.0])
    try:
        Lc = np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        return None
    y = np.linalg.solve(Lc, b)
    return np.linalg.solve(Lc.T, y)


def _acceptable(ol, ext, intr, px, X, sq_max):
    """AcceptableReprojectionError (:72-84): in front of the camera and the squared pixel error below the bound."""
    ok, res, _, _, _ = ol.reprojection_error(0, ext, intr, X, px)
    q = _angle_axis_to_matrix(ext[3:6]) @ (np.asarray(X[:3]) - X[3] * np.asarray(ext[0:3]))
    depth = q[2] / X[3] if X[3] != 0 else q[2]
    if depth < 0:
        return False
    return float(res[0] * res[0] + res[1] * res[1]) < sq_max


def verify_matches(ol, capi, options, prior1, prior2, corr, indexed=None):
    """VerifyMatches (:114-183) on pixel correspondences [(x1, y1, x2, y2)] -> (ok, info dict, verified indices).
    indexed = (keypoints1, descriptors1, keypoints2, descriptors2, matches [(feature1, feature2)]): the reference's own form of
    the call; `corr` is then built from the matches, the guided-matching branch (:157-170) can run, and the third element of the
    result is the list of verified (feature1, feature2) pairs."""
    if indexed is not None:
        kp1, ds1, kp2, ds2, mt = indexed
        mt = [(int(a), int(b)) for a, b in mt]
        corr = np.array([[kp1[a][0], kp1[a][1], kp2[b][0], kp2[b][1]] for a, b in mt], dtype=np.float64).reshape(-1, 4)
    corr = np.ascontiguousarray(corr, dtype=np.float64).reshape(-1, 4)
    info = {"focal_length_1": 0.0, "focal_length_2": 0.0, "rotation_2": np.zeros(3), "position_2": np.zeros(3),
            "num_verified_matches": 0, "num_homography_inliers": 0}
    if len(corr) < options.min_num_inlier_matches:
        return False, info, []
    eo = options.estimate_twoview_info_options
    # CountHomographyInliers (:331-368): camera1_ / camera2_ are default cameras there (image size 0 x 0)
    hp = _ransac_params(ol, eo, eo.max_sampson_error_pixels * eo.max_sampson_error_pixels, eo.use_mle, eo.seed)
    nh = int(ol.ransac_estimate(EST_HOMOGRAPHY, corr, hp)["num_inliers"])
    ok, info, inliers = estimate_two_view_info(ol, eo, prior1, prior2, corr)
    info["num_homography_inliers"] = nh
    if not ok or len(inliers) < options.min_num_inlier_matches:
        return False, info, []
    matches = list(inliers)
    # SetupCameras (:56-68)
    k1, k2 = _pinhole_of_prior(prior1), _pinhole_of_prior(prior2)
    k1[0] = info["focal_length_1"]; k2[0] = info["focal_length_2"]
    ext1 = np.zeros(6)
    ext2 = np.concatenate([info["position_2"], info["rotation_2"]])
    tags = None
    if indexed is not None:
        tags = [mt[i] for i in matches]
        if options.guided_matching:   # :157-170
            tags = guided_epipolar_matches(ol, ext1, k1, ext2, k2, kp1, ds1, kp2, ds2, tags, options.guided_matching_max_distance_pixels,
                                           options.guided_matching_lowes_ratio, eo.seed)
        corr = np.array([[kp1[a][0], kp1[a][1], kp2[b][0], kp2[b][1]] for a, b in tags], dtype=np.float64).reshape(-1, 4)
        matches = list(range(len(tags)))
    elif options.guided_matching:
        raise NotImplementedError("guided matching needs keypoints and descriptors (indexed form)")
    if options.bundle_adjustment and len(matches) > options.min_num_inlier_matches:
        # BundleAdjustRelativePose (:259-327) -- TriangulatePoints (:186-257)
        sq_tri = options.triangulation_max_reprojection_error ** 2
        cos_min = math.cos(math.radians(options.min_triangulation_angle_degrees))
        pts, kept = [], []
        for i in matches:
            r1 = _unit_ray(ext1, k1, corr[i, 0:2]); r2 = _unit_ray(ext2, k2, corr[i, 2:4])
            if not (float(r1 @ r2) < cos_min):               # SufficientTriangulationAngle (triangulation.cc:236-250)
                continue
            X = _triangulate_midpoint([ext1[0:3], ext2[0:3]], [r1, r2])
            if X is None:
                continue
            if not _acceptable(ol, ext1, k1, corr[i, 0:2], X, sq_tri) or not _acceptable(ol, ext2, k2, corr[i, 2:4], X, sq_tri):
                continue
            pts.append(X); kept.append(i)
        matches = kept
        if len(matches) < options.min_num_inlier_matches:
            return False, info, []
        # BundleAdjustTwoViews (bundle_adjust_two_views.cc:110-185): camera 1 constant, focal lengths free unless a prior
        # holds them, XYZW points, no loss, Ceres' defaults except max_num_iterations
        n = len(matches)
        intr = np.zeros((2, capi.THEIA_MAX_INTRINSICS)); intr[0, :7] = k1; intr[1, :7] = k2
        P = np.ascontiguousarray(np.array(pts))
        c = corr[matches]
        flat = capi.FlatProblem(np.array([ext1, ext2]), intr, [0, 0], [0, 1], P, np.concatenate([c[:, 0:2], c[:, 2:4]]),
                                np.concatenate([np.zeros(n, np.int32), np.ones(n, np.int32)]),
                                np.concatenate([np.arange(n), np.arange(n)]).astype(np.int32), cam_const=[3, 0],
                                group_const=[int(bool(prior1.focal_length.is_set)), int(bool(prior2.focal_length.is_set))])
        o = ol.default_options()
        o.max_num_iterations = 100                            # BundleAdjustmentOptions default through TwoViewBundleAdjustmentOptions
        o.use_homogeneous_point_parametrization = 0
        o.intrinsics_to_optimize = 0x01                       # FOCAL_LENGTH
        o.use_inner_iterations = 0
        o.max_trust_region_radius = 1e16
        o.loss_function_type = 0
        s, _ = ol.solve(flat, o)
        if not s.success:
            return False, info, []
        ext2 = flat.cam_ext[1].copy(); k1 = list(flat.intrinsics[0][:7]); k2 = list(flat.intrinsics[1][:7])
        sq_fin = options.final_max_reprojection_error ** 2
        after = []
        for j, i in enumerate(matches):
            if _acceptable(ol, ext1, k1, corr[i, 0:2], flat.points[j], sq_fin) and _acceptable(ol, ext2, k2, corr[i, 2:4], flat.points[j], sq_fin):
                after.append(i)
        matches = after
        info["rotation_2"] = ext2[3:6].copy()
        pos = ext2[0:3]
        info["position_2"] = pos / math.sqrt(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2])
        info["focal_length_1"] = float(k1[0]); info["focal_length_2"] = float(k2[0])
    info["num_verified_matches"] = len(matches)
    return len(matches) > options.min_num_inlier_matches, info, (matches if tags is None else [tags[i] for i in matches])


# ------------------------------------------------------------------------------------------------------------------
# matching/guided_epipolar_matcher.cc (the guided_matching branch of VerifyMatches, :157-170)
# Restated one feature / one sample point at a time.  Two orders the reference leaves to its containers are fixed here and in
# the product (stated in DESIGN.md): the candidates of a group are searched in ASCENDING feature index (the reference walks a
# std::unordered_set<int>; the order only breaks ties between equal float distances), and the squared descriptor distance is
# summed over the dimensions in sequence, in float (Eigen's squaredNorm() sums in packets).  The random candidates that fill a
# group up to 50 come from RandomNumberGenerator(seed).RandInt (the reference seeds that generator from the clock when
# options.rng is null, as VerifyMatches leaves it: its own runs differ from each other there).
def _projection_matrix(ext, intr):
    """Camera::GetProjectionMatrix = K [R | -R c] (camera.cc:195-200, projection_matrix_utils.cc:50-58,119-135)."""
    K = np.array([[intr[0], intr[2], intr[3]], [0.0, intr[0] * intr[1], intr[4]], [0.0, 0.0, 1.0]])
    Rm = _angle_axis_to_matrix(ext[3:6])
    P = np.zeros((3, 4))
    P[:, :3] = Rm
    P[:, 3] = -(Rm @ np.asarray(ext[0:3], dtype=np.float64))
    return K @ P


def _det4(M):
    """4 x 4 determinant by cofactors of the first two rows against the last two (2 x 2 minors), the closed form of
    Eigen's fixed-size determinant."""
    def d2(a, b, c, d):
        return a * d - b * c
    s0 = d2(M[0][0], M[0][1], M[1][0], M[1][1]); s1 = d2(M[0][0], M[0][2], M[1][0], M[1][2]); s2 = d2(M[0][0], M[0][3], M[1][0], M[1][3])
    s3 = d2(M[0][1], M[0][2], M[1][1], M[1][2]); s4 = d2(M[0][1], M[0][3], M[1][1], M[1][3]); s5 = d2(M[0][2], M[0][3], M[1][2], M[1][3])
    c5 = d2(M[2][2], M[2][3], M[3][2], M[3][3]); c4 = d2(M[2][1], M[2][3], M[3][1], M[3][3]); c3 = d2(M[2][1], M[2][2], M[3][1], M[3][2])
    c2 = d2(M[2][0], M[2][3], M[3][0], M[3][3]); c1 = d2(M[2][0], M[2][2], M[3][0], M[3][2]); c0 = d2(M[2][0], M[2][1], M[3][0], M[3][1])
    return s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0


def fundamental_from_projections(P_a, P_b):
    """FundamentalMatrixFromProjectionMatrices(pmatrix1 = P_a, pmatrix2 = P_b) (fundamental_matrix_util.cc:218-237)."""
    i1, i2 = (1, 2, 0), (2, 0, 1)
    F = np.zeros((3, 3))
    for r in range(3):
        for c in range(3):
            F[r, c] = _det4([P_b[i1[c]], P_b[i2[c]], P_a[i1[r]], P_a[i2[r]]])
    return F


def _grid_center(x, y, cs, ox, oy):
    """ImageGrid::GetClosestGridCenter (guided_epipolar_matcher.cc:441-450): static_cast<int> truncates toward zero."""
    return (int(math.floor((x - ox) / (2.0 * cs)) * 2.0 * cs + cs + ox), int(math.floor((y - oy) / (2.0 * cs)) * 2.0 * cs + cs + oy))


def _u16(v):
    return int(v) & 0xffff      # static_cast<uint16_t>(double) for the non-negative pixel coordinates this path sees


def guided_epipolar_matches(ol, ext1, intr1, ext2, intr2, kp1, desc1, kp2, desc2, matches, max_distance_pixels, lowes_ratio, seed):
    """GuidedEpipolarMatcher::GetMatches (:140-185) -> the input matches followed by the added ones [(feature1, feature2)]."""
    matches = [(int(a), int(b)) for a, b in matches]
    n1, n2 = len(kp1), len(kp2)
    m1 = set(a for a, _ in matches); m2 = set(b for _, b in matches)
    # Initialize (:92-138): four grids of cell size 2 d, offset by d in x / y / both; bounding box of the unmatched features of image 2
    d = max_distance_pixels
    offs = ((0.0, 0.0), (d, 0.0), (0.0, d), (d, d))
    grids = [dict() for _ in range(4)]
    xs, ys = [], []
    for i in range(n2):
        if i in m2:
            continue
        for j in range(4):
            grids[j].setdefault(_grid_center(kp2[i][0], kp2[i][1], d, offs[j][0], offs[j][1]), []).append(i)
        xs.append(kp2[i][0]); ys.append(kp2[i][1])
    if not xs:
        return matches
    tl = (min(xs), min(ys)); br = (max(xs), max(ys))
    # GroupEpipolarLines (:187-262)
    F = fundamental_from_projections(_projection_matrix(ext2, intr2), _projection_matrix(ext1, intr1))
    ends = []
    for i in range(n1):
        if i in m1:
            continue
        line = F @ np.array([kp1[i][0], kp1[i][1], 1.0])
        line = line / math.sqrt(line[0] * line[0] + line[1] * line[1])
        pts = []
        yl = -(line[2] + line[0] * tl[0]) / line[1]
        if tl[1] <= yl <= br[1]:
            pts.append((tl[0], yl))
        xt = -(line[2] + line[1] * tl[1]) / line[0]
        if tl[0] <= xt <= br[0]:
            pts.append((xt, tl[1]))
        yr = -(line[2] + line[0] * br[0]) / line[1]
        if tl[1] <= yr <= br[1]:
            pts.append((br[0], yr))
        xb = -(line[2] + line[1] * br[1]) / line[0]
        if tl[0] <= xb <= br[0]:
            pts.append((xb, br[1]))
        if len(pts) != 2:
            continue
        code = (_u16(pts[0][0]) << 48) | (_u16(pts[0][1]) << 32) | (_u16(pts[1][0]) << 16) | _u16(pts[1][1])
        ends.append((code, i))
    ends.sort()
    groups = []   # [endpoint0 (2), endpoint1 (2), features]
    sq = d * d
    for k, (code, i) in enumerate(ends):
        e0 = np.array([float((code >> 48) & 0xffff), float((code >> 32) & 0xffff)]); e1 = np.array([float((code >> 16) & 0xffff), float(code & 0xffff)])
        if k == 0 or float((groups[-1][0] - e0) @ (groups[-1][0] - e0)) > sq:
            groups.append([e0.copy(), e1.copy(), []])
        g = groups[-1]
        g[2].append(i)
        w = 1.0 / len(g[2])
        g[0] = (1.0 - w) * g[0] + w * e0
        g[1] = (1.0 - w) * g[1] + w * e1
    # per group: candidates near the line (:264-306), two nearest neighbours (:356-412), Lowe's ratio (:163-177)
    ratio_sq = lowes_ratio * lowes_ratio
    need = sum(1 for _ in groups)
    draws = ol.randint_stream(seed, [0] * (50 * need), [n2 - 1] * (50 * need)) if need else []
    nd = 0
    out = list(matches)
    D1 = np.asarray(desc1, dtype=np.float32); D2 = np.asarray(desc2, dtype=np.float32)
    for e0, e1, feats in groups:
        diff = e1 - e0
        num_steps = int(math.sqrt(diff[0] * diff[0] + diff[1] * diff[1]) / d)
        cand = set()
        if num_steps > 0:
            delta = (e0 - e1) / float(num_steps)
            sp = e1.copy()
            for _ in range(num_steps):
                sp = sp + delta
                best, bd, bc = 0, float("inf"), None
                for j in range(4):
                    c = _grid_center(sp[0], sp[1], d, offs[j][0], offs[j][1])
                    dist = (c[0] - sp[0]) ** 2 + (c[1] - sp[1]) ** 2
                    if dist < bd:
                        best, bd, bc = j, dist, c
                cand.update(grids[best].get(bc, []))
        if len(cand) < 50:
            for _ in range(len(cand), 50):
                cand.add(int(draws[nd])); nd += 1
        cl = sorted(cand)
        for q in feats:
            best = []   # (distance, position)
            for pos, ci in enumerate(cl):
                acc = np.float32(0.0)
                dv = D1[q] - D2[ci]
                sqv = dv * dv
                for t in range(len(sqv)):
                    acc = np.float32(acc + sqv[t])
                best.append((float(acc), pos))
            best.sort()
            if len(best) >= 2 and best[0][0] < best[1][0] * ratio_sq:
                out.append((q, cl[best[0][1]]))
    return out
