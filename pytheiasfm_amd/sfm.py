"""Host-side mirror of pyTheia's BA interface (pytheia.sfm.BundleAdjust*).

Same names, argument order and error behaviour as the pybind surface
(src/pytheia/sfm/sfm.cc:1605-1626 -> bundle_adjustment_wrapper.cc:98-114):
the functions flatten a reconstruction into the C-ABI problem exactly the way
BundleAdjuster::AddView / AddTrack build the Ceres problem
(bundle_adjuster.cc:116-221), call libtheia_hip.so, scatter the parameters
back and run the reference's post-step (UpdateInverseDepth,
bundle_adjustment.cc:69-83).

The data model (Reconstruction/View/Track) is OUT OF SCOPE of this engine --
in the real integration it stays theia's C++ types.  The
`Reconstruction` below is a compact array-backed stand-in with the accessors
the BA entry points need, so that the parity tests read like the reference's.
"""
import enum

import numpy as np

from . import _capi as capi
from . import ba as _ba
from .synth import angle_axis_to_matrix

kInvalidViewId = 0xFFFFFFFF

# handles of the most recent full / partial BA problems (ba.ProblemCache): a BA call on an unchanged topology skips
# theia_hip_ba_create.  set_problem_cache(0) switches it off, clear_problem_cache() frees the device memory it holds.
_problem_cache = _ba.ProblemCache(capacity=1)


def set_problem_cache(capacity):
    global _problem_cache
    _problem_cache.clear()
    _problem_cache = _ba.ProblemCache(capacity=capacity)
    return _problem_cache


def clear_problem_cache():
    _problem_cache.clear()


class LossFunctionType(enum.IntEnum):  # create_loss_function.h:52-60
    TRIVIAL = 0
    HUBER = 1
    SOFTLONE = 2
    CAUCHY = 3
    ARCTAN = 4
    TUKEY = 5
    TRUNCATED = 6  # C++ only in the reference (not in the Python enum, sfm.cc:1566-1572)


class OptimizeIntrinsicsType(enum.IntFlag):  # bundle_adjustment.h:71-85
    NONE = 0x00
    FOCAL_LENGTH = 0x01
    ASPECT_RATIO = 0x02
    SKEW = 0x04
    PRINCIPAL_POINTS = 0x08
    RADIAL_DISTORTION = 0x10
    TANGENTIAL_DISTORTION = 0x20
    ALL = 0x3F


class CameraIntrinsicsModelType(enum.IntEnum):  # camera_intrinsics_model_type.h:46-56
    PINHOLE = 0
    PINHOLE_RADIAL_TANGENTIAL = 1
    FISHEYE = 2
    FOV = 3
    DIVISION_UNDISTORTION = 4
    DOUBLE_SPHERE = 5
    EXTENDED_UNIFIED = 6
    ORTHOGRAPHIC = 7


class BundleAdjustmentOptions:
    """bundle_adjustment.h:87-167, same field names and defaults.  The
    linear-algebra selectors are accepted and ignored: the HIP backend is the
    linear solver (Schur complement in the fused assembly kernels, then a tile-sparse, level-scheduled Cholesky of the
    reduced camera system with FP64-MFMA triangular solves and updates; dense schedule where the tile graph is full)."""

    def __init__(self):
        self.loss_function_type = LossFunctionType.TRIVIAL
        self.robust_loss_width = 2.0
        self.robust_loss_width_depth_prior = 0.01
        self.linear_solver_type = "SPARSE_SCHUR"
        self.preconditioner_type = "SCHUR_JACOBI"
        self.visibility_clustering_type = "CANONICAL_VIEWS"
        self.dense_linear_algebra_library_type = "EIGEN"
        self.sparse_linear_algebra_library_type = "EIGEN_SPARSE"
        self.optimize_for_forward_facing_trajectory = False
        self.use_mixed_precision_solves = False
        self.max_num_refinement_iterations = 2
        self.verbose = False
        self.constant_camera_orientation = False
        self.constant_camera_position = False
        self.use_homogeneous_point_parametrization = True
        self.use_inverse_depth_parametrization = False
        self.intrinsics_to_optimize = OptimizeIntrinsicsType.NONE
        self.num_threads = 1
        self.max_num_iterations = 100
        self.max_solver_time_in_seconds = 3600.0
        self.use_inner_iterations = True
        self.function_tolerance = 1e-6
        self.gradient_tolerance = 1e-10
        self.parameter_tolerance = 1e-8
        self.max_trust_region_radius = 1e12
        self.use_position_priors = False
        self.use_orientation_priors = False
        self.use_depth_priors = False
        self.orthographic_camera = False
        self.use_gravity_priors = False

    def to_c(self):
        o = _ba.default_options()
        o.loss_function_type = int(self.loss_function_type)
        o.robust_loss_width = float(self.robust_loss_width)
        o.robust_loss_width_depth_prior = float(self.robust_loss_width_depth_prior)
        o.intrinsics_to_optimize = int(self.intrinsics_to_optimize)
        o.prior_mask = ((capi.THEIA_PRIOR_POSITION if self.use_position_priors else 0) |
                        (capi.THEIA_PRIOR_GRAVITY if self.use_gravity_priors else 0) |
                        (capi.THEIA_PRIOR_ORIENTATION if self.use_orientation_priors else 0))
        o.max_num_iterations = int(self.max_num_iterations)
        o.use_homogeneous_point_parametrization = int(bool(self.use_homogeneous_point_parametrization))
        o.constant_camera_orientation = int(bool(self.constant_camera_orientation))
        o.constant_camera_position = int(bool(self.constant_camera_position))
        o.orthographic_camera = int(bool(self.orthographic_camera))
        o.use_inner_iterations = int(bool(self.use_inner_iterations))
        o.verbose = int(bool(self.verbose))
        o.function_tolerance = float(self.function_tolerance)
        o.gradient_tolerance = float(self.gradient_tolerance)
        o.parameter_tolerance = float(self.parameter_tolerance)
        o.max_trust_region_radius = float(self.max_trust_region_radius)
        o.max_solver_time_in_seconds = float(self.max_solver_time_in_seconds)
        return o


class BundleAdjustmentSummary:
    """bundle_adjustment.h:170-178."""

    def __init__(self, c=None):
        self.success = bool(c.success) if c is not None else False
        self.initial_cost = c.initial_cost if c is not None else 0.0
        self.final_cost = c.final_cost if c is not None else 0.0
        self.setup_time_in_seconds = c.setup_time_in_seconds if c is not None else 0.0
        self.solve_time_in_seconds = c.solve_time_in_seconds if c is not None else 0.0
        # extras of the HIP backend (not in the reference struct)
        self.num_iterations = c.num_iterations if c is not None else 0
        self.termination_type = c.termination_type if c is not None else 2


class Reconstruction:
    """Array-backed stand-in for theia::Reconstruction: ids are dense indices."""

    def __init__(self):
        self.cam_ext = np.zeros((0, 6))
        self.view_estimated = np.zeros(0, dtype=bool)
        self.view_group = np.zeros(0, dtype=np.int32)
        self.group_model = np.zeros(0, dtype=np.int32)
        self.group_intrinsics = np.zeros((0, capi.THEIA_MAX_INTRINSICS))
        self.points = np.zeros((0, 4))
        self.track_estimated = np.zeros(0, dtype=bool)
        self.track_reference_view = np.zeros(0, dtype=np.int64)
        self.inverse_depth = np.zeros(0)
        self.track_reference_bearing = np.zeros((0, 3))   # Track::ReferenceBearingVector()
        self.obs_view = np.zeros(0, dtype=np.int32)
        self.obs_track = np.zeros(0, dtype=np.int32)
        self.obs_uv = np.zeros((0, 2))
        self.obs_cov = np.zeros((0, 2))  # diagonal of Feature::covariance_
        # Feature::depth_prior_ / depth_prior_variance_ (feature.h:58-61); None = no feature carries a depth prior
        self.obs_depth_prior = None
        self.obs_depth_prior_variance = None
        # View::{Position,Gravity,Orientation}Prior (+ sqrt information), view.h; mask bits = capi.THEIA_PRIOR_*
        self.view_prior_mask = None
        self.view_priors = {}

    @classmethod
    def from_flat(cls, p):
        r = cls()
        r.cam_ext = p.cam_ext.copy()
        nv, nt = p.cam_ext.shape[0], p.points.shape[0]
        r.view_estimated = np.ones(nv, dtype=bool)
        r.view_group = p.cam_group.copy()
        r.group_model = p.group_model.copy()
        r.group_intrinsics = p.intrinsics.copy()
        r.points = p.points.copy()
        r.track_estimated = np.ones(nt, dtype=bool)
        r.obs_view = p.obs_cam.copy(); r.obs_track = p.obs_pt.copy(); r.obs_uv = p.obs_uv.copy()
        r.obs_cov = np.ones((len(r.obs_view), 2)) if p.obs_sqrt_info is None else 1.0 / p.obs_sqrt_info ** 2
        first = np.full(nt, kInvalidViewId, dtype=np.int64)
        order = np.argsort(r.obs_track, kind="stable")
        tr, idx = np.unique(r.obs_track[order], return_index=True)
        first[tr] = r.obs_view[order][idx]
        r.track_reference_view = first
        r.inverse_depth = np.zeros(nt)
        r.track_reference_bearing = np.zeros((nt, 3))
        return r

    def NumViews(self):
        return self.cam_ext.shape[0]

    def NumTracks(self):
        return self.points.shape[0]

    def ViewIds(self):
        return range(self.NumViews())

    def TrackIds(self):
        return range(self.NumTracks())


def _ids(ids):
    """id lists of the mirror -> int64 array (a range -- ViewIds() / TrackIds() -- without walking it in Python)"""
    if isinstance(ids, range):
        return np.arange(ids.start, ids.stop, ids.step, dtype=np.int64)
    if isinstance(ids, np.ndarray):
        return ids.astype(np.int64, copy=False).reshape(-1)
    return np.asarray(list(ids), dtype=np.int64)


def _flatten(recon, view_ids, track_ids, const_view_ids=(), options=None):
    """BundleAdjuster::AddView (:116-173) for `view_ids` (+ const views), then
    AddTrack (:175-221) for `track_ids`."""
    nv, nt = recon.NumViews(), recon.NumTracks()
    view_added = np.zeros(nv, dtype=bool)
    vi = np.concatenate([_ids(view_ids), _ids(const_view_ids)])
    if len(vi):
        if vi.min() < 0 or vi.max() >= nv:
            raise capi.TheiaHipError(-1, "view id out of range (reference: CHECK_NOTNULL aborts)")
        view_added[vi] = True
    view_added &= recon.view_estimated
    track_added = np.zeros(nt, dtype=bool)
    ti = _ids(track_ids)
    if len(ti):
        if ti.min() < 0 or ti.max() >= nt:
            raise capi.TheiaHipError(-1, "track id out of range (reference: CHECK_NOTNULL aborts)")
        track_added[ti] = True
    track_added &= recon.track_estimated
    ov, ot = recon.obs_view, recon.obs_track
    if recon.view_estimated.all() and recon.track_estimated.all() and (view_added.all() or track_added.all()):
        keep = np.ones(len(ov), dtype=bool)      # every observation enters: skip the four 3 M-row gathers
    else:
        est = recon.view_estimated[ov] & recon.track_estimated[ot]
        keep = est & (view_added[ov] | track_added[ot])
    cam_const = np.zeros(nv, dtype=np.uint8)
    # cameras reached only through AddTrack are frozen (:204)
    cam_const[~view_added] = capi_const_all()
    if len(const_view_ids):
        cam_const[_ids(const_view_ids)] = capi_const_all()
    point_const = (~track_added).astype(np.uint8)  # SetTrackConstant (:149) unless AddTrack'ed
    # an intrinsics group is optimised iff one of its views went through AddView
    # (:130-133); groups reached only through AddTrack stay constant (:442-455)
    group_const = np.ones(recon.group_intrinsics.shape[0], dtype=np.uint8)
    group_const[recon.view_group[view_added]] = 0
    sqrt_info = None
    everything = bool(keep.all())       # the full BA of an all-estimated reconstruction: no 3 M-row boolean gathers
    cov = recon.obs_cov if everything else recon.obs_cov[keep]
    if len(cov) and not np.all(cov == 1.0):
        sqrt_info = 1.0 / np.sqrt(cov)
    flat = capi.FlatProblem(recon.cam_ext.copy(), recon.group_intrinsics.copy(), recon.group_model,
                            recon.view_group, recon.points.copy(), recon.obs_uv if everything else recon.obs_uv[keep],
                            ov if everything else ov[keep], ot if everything else ot[keep],
                            cam_const=cam_const, group_const=group_const, point_const=point_const,
                            obs_sqrt_info=sqrt_info)
    if options is not None and options.use_depth_priors and recon.obs_depth_prior is not None:
        # AddView adds a DepthPriorError block for every feature of the view with depth_prior != 0
        # (bundle_adjuster.cc:152-156); observations reached only through AddTrack get none (:175-221)
        kept = np.flatnonzero(keep)
        rows = np.flatnonzero(view_added[ov[kept]] & (recon.obs_depth_prior[kept] != 0.0))
        if len(rows):
            var = np.ones(len(rows)) if recon.obs_depth_prior_variance is None else recon.obs_depth_prior_variance[kept][rows]
            flat.add_depth_priors(rows, recon.obs_depth_prior[kept][rows], var)
    if recon.view_prior_mask is not None:
        # priors are added for the views that went through AddView (bundle_adjuster.cc:159-172)
        mask = np.where(view_added, recon.view_prior_mask, 0).astype(np.uint8)
        flat.set_priors(mask, **recon.view_priors)
    return flat


def capi_const_all():
    return 0x3


def _update_inverse_depth(recon, track_ids):
    """UpdateInverseDepth (bundle_adjustment.cc:69-83): inverse depth of the
    track in its reference view, Camera::ProjectPoint depth (camera.cc:206-216)."""
    ti = _ids(track_ids)
    if not len(ti):
        return
    ti = ti[recon.track_estimated[ti]]
    ref = recon.track_reference_view[ti]
    ok = ref != kInvalidViewId
    ti, ref = ti[ok], ref[ok]
    if not len(ti):
        return
    X = recon.points[ti]
    ce = recon.cam_ext[ref]
    row2 = angle_axis_to_matrix(recon.cam_ext[:, 3:6])[:, 2, :]   # one rotation per VIEW, not per track
    p = X[:, :3] - X[:, 3:4] * ce[:, :3]
    depth = np.einsum("nj,nj->n", row2[ref], p) / X[:, 3]
    recon.inverse_depth[ti] = 1.0 / depth


def _flatten_inverse_depth(recon, track_ids, const_view_ids=()):
    """BundleAdjuster::AddInvTrack(track, false) for every listed track (bundle_adjuster.cc:223-289): one residual block per
    estimated view of an estimated track WITH a reference view (tracks without one are skipped with an error log), every
    camera variable unless set constant afterwards; the point block is the track's inverse depth."""
    nt = recon.NumTracks()
    added = np.zeros(nt, dtype=bool)
    ti = np.asarray(list(track_ids), dtype=np.int64)
    if len(ti):
        added[ti] = True
    added &= recon.track_estimated & (recon.track_reference_view != kInvalidViewId)
    ov, ot = recon.obs_view, recon.obs_track
    keep = recon.view_estimated[ov] & added[ot]
    cam_const = np.zeros(recon.NumViews(), dtype=np.uint8)
    if len(const_view_ids):
        cam_const[_ids(const_view_ids)] = capi_const_all()
    sqrt_info = None
    cov = recon.obs_cov[keep]
    if len(cov) and not np.all(cov == 1.0):
        sqrt_info = 1.0 / np.sqrt(cov)
    flat = capi.FlatProblem(recon.cam_ext.copy(), recon.group_intrinsics.copy(), recon.group_model, recon.view_group,
                            recon.points.copy(), recon.obs_uv[keep], ov[keep], ot[keep], cam_const=cam_const,
                            point_const=(~added).astype(np.uint8), obs_sqrt_info=sqrt_info)
    ref = np.where(recon.track_reference_view == kInvalidViewId, 0, recon.track_reference_view)
    rho = np.where(added, recon.inverse_depth, 1.0)
    flat.set_inverse_depth(ref, recon.track_reference_bearing, rho)
    return flat, added


def _update_homogeneous_point(recon, track_mask):
    """UpdateHomogeneousPoint (bundle_adjustment.cc:47-65): X = R_ref^T (bearing / inverse_depth) + c_ref for estimated tracks
    with a positive inverse depth."""
    ti = np.flatnonzero(track_mask & recon.track_estimated & (recon.inverse_depth > 0.0))
    if not len(ti):
        return
    ce = recon.cam_ext[recon.track_reference_view[ti]]
    R = angle_axis_to_matrix(ce[:, 3:6])
    b = recon.track_reference_bearing[ti] / recon.inverse_depth[ti, None]
    recon.points[ti, :3] = np.einsum("nji,nj->ni", R, b) + ce[:, :3]
    recon.points[ti, 3] = 1.0


def _run_inverse_depth(options, recon, track_ids, const_view_ids=()):
    flat, added = _flatten_inverse_depth(recon, track_ids, const_view_ids)
    c_opts = options.to_c()
    # Only AddInvTrack runs in this mode (bundle_adjustment.cc:119-123,155-162,194-198): no view is ever entered into
    # optimized_camera_intrinsics_groups_, so every intrinsics group the tracks touch is "potentially constant" and is
    # frozen by SetCameraIntrinsicsParameterization (bundle_adjuster.cc:441-459) whatever intrinsics_to_optimize says.
    c_opts.intrinsics_to_optimize = 0
    # Stated deviation: with use_inner_iterations the reference hands Ceres NO inner ordering in this mode
    # (bundle_adjuster.cc:328-333), and Ceres then builds its own independent-set ordering from its hash-ordered
    # parameter graph -- not reproducible outside Ceres.  The sweeps are switched off here; the LM trajectory of a
    # default-options call can therefore differ from the reference's after the first accepted step.
    if c_opts.use_inner_iterations:
        import warnings
        warnings.warn("inverse-depth bundle adjustment: use_inner_iterations is not reproducible outside Ceres in this mode (no inner "
                      "ordering is handed over, bundle_adjuster.cc:328-333) and is switched off; see DESIGN.md section 2", RuntimeWarning,
                      stacklevel=3)
    c_opts.use_inner_iterations = 0
    # AddViewPriors (bundle_adjuster.cc:290-313) runs in this mode over optimized_views_ = the views that observe an added
    # track or are the reference view of one: exactly the cameras the library counts as used (ba_invdepth.hip)
    if recon.view_prior_mask is not None:
        flat.set_priors(np.asarray(recon.view_prior_mask, dtype=np.uint8), **recon.view_priors)
    s, _ = _problem_cache.solve(flat, c_opts)
    recon.cam_ext[:] = flat.cam_ext
    recon.inverse_depth[added] = flat.point_inverse_depth[added]
    _update_homogeneous_point(recon, added)
    return BundleAdjustmentSummary(s)


def _run(options, recon, flat):
    c_opts = options.to_c()
    if getattr(options, "optimize_for_forward_facing_trajectory", False):
        # bundle_adjuster.cc:547-563: intrinsics and extrinsics share elimination group 1.  The reduced system is the same
        # one (only Ceres' fill-reducing order inside it changes).  But the reference hands the REVERSED ordering to the inner
        # iterations (bundle_adjuster.cc:329-333), whose first set is then {extrinsics, intrinsics}: not an independent set
        # when any intrinsics block is variable, Ceres rejects it (CoordinateDescentMinimizer::IsOrderingValid) and Solve
        # returns FAILURE without touching the parameters.
        if c_opts.use_inner_iterations and c_opts.intrinsics_to_optimize != 0:
            fail = BundleAdjustmentSummary()
            fail.success = False
            return fail
    s, _ = _problem_cache.solve(flat, c_opts)
    recon.cam_ext[:] = flat.cam_ext
    recon.points[:] = flat.points
    recon.group_intrinsics[:] = flat.intrinsics   # shared CameraIntrinsicsModel parameters (:388-389)
    return BundleAdjustmentSummary(s)


def BundleAdjustReconstruction(options, reconstruction):
    """bundle_adjustment.cc:188-217 (argument order of the pybind wrapper)."""
    if options.use_inverse_depth_parametrization:
        return _run_inverse_depth(options, reconstruction, reconstruction.TrackIds())
    flat = _flatten(reconstruction, reconstruction.ViewIds(), reconstruction.TrackIds(), options=options)
    summary = _run(options, reconstruction, flat)
    _update_inverse_depth(reconstruction, reconstruction.TrackIds())
    return summary


def BundleAdjustPartialReconstruction(options, view_ids, track_ids, reconstruction):
    """bundle_adjustment.cc:111-143."""
    if options.use_inverse_depth_parametrization:
        return _run_inverse_depth(options, reconstruction, track_ids)
    flat = _flatten(reconstruction, view_ids, track_ids, options=options)
    summary = _run(options, reconstruction, flat)
    # reference quirk (:134-135): the post-update list carries len(track_ids)
    # leading zeros, i.e. track 0 is also "updated" -- harmless, reproduced.
    post = ([0] if len(list(track_ids)) and reconstruction.NumTracks() else []) + list(track_ids)
    _update_inverse_depth(reconstruction, post)
    return summary


def BundleAdjustPartialViewsConstant(options, var_view_ids, const_view_ids, reconstruction):
    """bundle_adjustment.cc:146-185."""
    if options.use_inverse_depth_parametrization:
        return _run_inverse_depth(options, reconstruction, reconstruction.TrackIds(), const_view_ids)
    flat = _flatten(reconstruction, var_view_ids, reconstruction.TrackIds(), const_view_ids=const_view_ids, options=options)
    summary = _run(options, reconstruction, flat)
    _update_inverse_depth(reconstruction, reconstruction.TrackIds())
    return summary


def _no_inner(options):
    """BundleAdjustView(s) / BundleAdjustTrack(s) and their WithCov variants copy the options and switch the inner
    iterations off (bundle_adjustment.cc:225,245,267,296,338,395,427,461; they also ask for DENSE_QR, which has no
    meaning here)."""
    import copy
    o = copy.copy(options)
    o.use_inner_iterations = False
    return o


def BundleAdjustViews(reconstruction, options, view_ids):
    """bundle_adjustment.cc:240-258 (wrapper argument order :14-17)."""
    options = _no_inner(options)
    flat = _flatten(reconstruction, view_ids, [], options=options)
    return _run(options, reconstruction, flat)


def BundleAdjustView(reconstruction, options, view_id):
    """bundle_adjustment.cc:220-237."""
    return BundleAdjustViews(reconstruction, options, [view_id])


def BundleAdjustTracks(reconstruction, options, track_ids):
    """bundle_adjustment.cc:389-418."""
    options = _no_inner(options)
    flat = _flatten(reconstruction, [], track_ids, options=options)
    summary = _run(options, reconstruction, flat)
    _update_inverse_depth(reconstruction, track_ids)
    return summary


def BundleAdjustTrack(reconstruction, options, track_id):
    """bundle_adjustment.cc:261-285."""
    return BundleAdjustTracks(reconstruction, options, [track_id])


# --------------------------------------------------------------- batched micro-BA
# The pipelines call BundleAdjustView / BundleAdjustTrack once per view / per track from a
# thread pool (estimate_track.cc:176-184,289; camera localisation).  These two helpers run
# such a list of INDEPENDENT problems as one device batch; each entry is exactly what the
# single-item function of the reference would solve.
def BundleAdjustTwoViewsAngular(options, correspondences, two_view_info):
    """bundle_adjust_two_views.cc:189-246 through its pybind wrapper (bundle_adjustment_wrapper.cc:5-12):
    refines two_view_info.rotation_2 / position_2 in place against the angular epipolar error of the
    normalised correspondences [(x1, y1, x2, y2)]; returns the summary."""
    s = BundleAdjustTwoViewsAngularBatch(options, [correspondences], [two_view_info])
    return s[0]


def BundleAdjustTwoViewsAngularBatch(options, correspondences_list, two_view_infos):
    """N pairs as one launch (one wave per pair)."""
    n = len(correspondences_list)
    corr = [np.asarray(c, dtype=np.float64).reshape(-1, 4) for c in correspondences_list]
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(c) for c in corr])
    rp = np.array([np.concatenate([np.asarray(i.rotation_2, dtype=np.float64), np.asarray(i.position_2, dtype=np.float64)])
                   for i in two_view_infos], dtype=np.float64).reshape(n, 6)
    cgnr = str(getattr(options, "linear_solver_type", "")).upper().endswith("CGNR")   # every other type is a direct solve
    summ = _ba.solve_two_views_angular_batch(offsets, np.vstack(corr) if n else np.zeros((0, 4)), rp, options.to_c(),
                                             _ba.TWO_VIEW_CGNR if cgnr else _ba.TWO_VIEW_EXACT)
    for k, info in enumerate(two_view_infos):
        info.rotation_2 = rp[k, :3].copy()
        info.position_2 = rp[k, 3:].copy()
    return [BundleAdjustmentSummary(c) for c in summ]


def OptimizeRelativePositionWithKnownRotation(correspondences, rotation1, rotation2):
    """optimize_relative_position_with_known_rotation.cc:116-191 through its pybind wrapper
    (bundle_adjustment_wrapper.cc:141-150) -> (success, relative_position)."""
    pos = OptimizeRelativePositionWithKnownRotationBatch([correspondences], [rotation1], [rotation2])
    return True, pos[0]


def OptimizeRelativePositionWithKnownRotationBatch(correspondences_list, rotations1, rotations2):
    """One call per view pair of the view graph in the reference (RefineRelativeTranslationsWithKnownRotations,
    reconstruction_estimator_utils.cc:263-291, a thread pool); here all pairs as one launch, one wavefront per pair.
    Returns the positions [N][3]."""
    n = len(correspondences_list)
    corr = [np.asarray(c, dtype=np.float64).reshape(-1, 4) for c in correspondences_list]
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(c) for c in corr])
    rot = np.concatenate([np.asarray(rotations1, dtype=np.float64).reshape(n, 3), np.asarray(rotations2, dtype=np.float64).reshape(n, 3)], axis=1)
    pos, _ = _ba.optimize_relative_position_batch(offsets, np.vstack(corr) if n else np.zeros((0, 4)), rot)
    return pos


def OptimizeAbsolutePoseOnNormFeatures(correspondences_2d_3d, rotation_init, position_init, ba_options):
    """pose/pose_wrapper.cc:39-65 (pybind sfm.cc:1625-1626): BundleAdjustView of a one-view reconstruction whose camera is
    the default pinhole (focal length 1, principal point 0) at (rotation_init, position_init), every world point an estimated
    (hence constant) track.  correspondences_2d_3d: [N][5] = (x, y, X, Y, Z).  -> (success, rotation matrix, position)."""
    out = OptimizeAbsolutePoseOnNormFeaturesBatch([correspondences_2d_3d], [rotation_init], [position_init], ba_options)
    return out[0]


def OptimizeAbsolutePoseOnNormFeaturesBatch(correspondences_list, rotations_init, positions_init, ba_options):
    """N x OptimizeAbsolutePoseOnNormFeatures as one launch (theia_hip_ba_views_batch: one LM solve per wavefront)."""
    from . import synth as _synth
    n = len(correspondences_list)
    c = [np.asarray(x, dtype=np.float64).reshape(-1, 5) for x in correspondences_list]
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(x) for x in c])
    allc = np.vstack(c) if n else np.zeros((0, 5))
    cams = np.zeros((n, 6))
    for k in range(n):
        cams[k, :3] = np.asarray(positions_init[k], dtype=np.float64)
        cams[k, 3:] = _synth.matrix_to_angle_axis(np.asarray(rotations_init[k], dtype=np.float64).reshape(3, 3))
    intr = np.zeros((n, capi.THEIA_MAX_INTRINSICS)); intr[:, 0] = 1.0; intr[:, 1] = 1.0     # PinholeCameraModel defaults
    pts = np.concatenate([allc[:, 2:5], np.ones((len(allc), 1))], axis=1)
    summ = _ba.solve_views_batch(offsets, allc[:, 0:2], pts, cams, intr, np.zeros(n, np.int32), _no_inner(ba_options).to_c())
    return [(bool(s.success), _synth.angle_axis_to_matrix(cams[k, 3:]), cams[k, :3].copy()) for k, s in enumerate(summ)]


def BundleAdjustViewsIndependently(reconstruction, options, view_ids):
    """[BundleAdjustView(reconstruction, options, v) for v in view_ids] as one launch
    (theia_hip_ba_views_batch).  Returns the list of summaries."""
    r = reconstruction
    if options.use_depth_priors and r.obs_depth_prior is not None and np.any(r.obs_depth_prior != 0.0):
        raise capi.TheiaHipError(-3, "depth priors are not built for the batched single-view solves")
    view_ids = [int(v) for v in view_ids]
    for v in view_ids:
        if v < 0 or v >= r.NumViews():
            raise capi.TheiaHipError(-1, "view id out of range (reference: CHECK_NOTNULL aborts)")
    sel = [np.nonzero((r.obs_view == v) & r.track_estimated[r.obs_track] & bool(r.view_estimated[v]))[0] for v in view_ids]
    offsets = np.zeros(len(view_ids) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in sel])
    idx = np.concatenate(sel) if sel else np.zeros(0, dtype=np.int64)
    cams = np.ascontiguousarray(r.cam_ext[view_ids]) if view_ids else np.zeros((0, 6))
    cov = r.obs_cov[idx]
    si = None if (len(cov) == 0 or np.all(cov == 1.0)) else 1.0 / np.sqrt(cov)
    summ = _ba.solve_views_batch(offsets, r.obs_uv[idx], r.points[r.obs_track[idx]], cams,
                                 r.group_intrinsics[r.view_group[view_ids]], r.group_model[r.view_group[view_ids]],
                                 options.to_c(), obs_sqrt_info=si)
    for k, v in enumerate(view_ids):
        r.cam_ext[v] = cams[k]
    return [BundleAdjustmentSummary(s) for s in summ]


def BundleAdjustTracksIndependently(reconstruction, options, track_ids):
    """[BundleAdjustTrack(reconstruction, options, t) for t in track_ids] as one launch
    (theia_hip_ba_tracks_batch).  Returns the list of summaries."""
    r = reconstruction
    track_ids = np.asarray([int(t) for t in track_ids], dtype=np.int64)
    if len(track_ids) and (track_ids.min() < 0 or track_ids.max() >= r.NumTracks()):
        raise capi.TheiaHipError(-1, "track id out of range (reference: CHECK_NOTNULL aborts)")
    local = -np.ones(r.NumTracks(), dtype=np.int64)
    local[track_ids] = np.arange(len(track_ids))
    keep = (local[r.obs_track] >= 0) & r.view_estimated[r.obs_view] & r.track_estimated[r.obs_track]
    cov = r.obs_cov[keep]
    si = None if (len(cov) == 0 or np.all(cov == 1.0)) else 1.0 / np.sqrt(cov)
    pts = np.ascontiguousarray(r.points[track_ids])
    flat = capi.FlatProblem(r.cam_ext.copy(), r.group_intrinsics.copy(), r.group_model, r.view_group, pts, r.obs_uv[keep],
                            r.obs_view[keep], local[r.obs_track[keep]].astype(np.int32),
                            point_const=(~r.track_estimated[track_ids]).astype(np.uint8), obs_sqrt_info=si)
    summ = _ba.solve_tracks_batch(flat, options.to_c())
    r.points[track_ids] = flat.points
    _update_inverse_depth(r, track_ids.tolist())
    return [BundleAdjustmentSummary(s) for s in summ]


# ------------------------------------------------------------------- *WithCov
# bundle_adjustment.cc:288-386,420-499 through bundle_adjustment_wrapper.cc:52-96: the solve, then
# ceres::Covariance of the (block-diagonal) problem times the empirical variance factor
# 2 * final_cost / redundancy.  Returned as (summary, covariance(s), variance_factor).
def _with_cov(reconstruction, options, flat, want_points):
    c_opts = options.to_c()
    with _ba.BaHandle(flat, c_opts) as h:
        s, _ = h.run()
        h.download(flat)
        summary = BundleAdjustmentSummary(s)
        if not summary.success:
            return summary, None
        try:
            pc, cc = h.covariance(points=want_points, cameras=not want_points)
        except capi.TheiaHipError:
            summary.success = False      # GetCovarianceFor* returned false
            return summary, None
    reconstruction.cam_ext[:] = flat.cam_ext
    reconstruction.points[:] = flat.points
    reconstruction.group_intrinsics[:] = flat.intrinsics
    return summary, (pc if want_points else cc)


def BundleAdjustTracksWithCov(reconstruction, options, track_ids):
    """bundle_adjustment.cc:330-386 (forces the homogeneous manifold, :339)."""
    opts = _no_inner(options)
    opts.use_homogeneous_point_parametrization = True
    opts.use_inverse_depth_parametrization = False
    track_ids = [int(t) for t in track_ids]
    flat = _flatten(reconstruction, [], track_ids)
    summary, pc = _with_cov(reconstruction, opts, flat, True)
    _update_inverse_depth(reconstruction, track_ids)
    if pc is None:
        return summary, {}, 1.0
    nr_obs = int(sum(np.sum((reconstruction.obs_track == t) ) for t in track_ids))   # Track::NumViews()
    redundancy = nr_obs * 2 - 3.0 * len(track_ids)
    factor = (2.0 * summary.final_cost) / redundancy
    return summary, {t: pc[t] * factor for t in track_ids}, factor


def BundleAdjustTrackWithCov(reconstruction, options, track_id):
    """bundle_adjustment.cc:288-327."""
    summary, covs, factor = BundleAdjustTracksWithCov(reconstruction, options, [track_id])
    return summary, covs.get(int(track_id), np.eye(3)), factor


def BundleAdjustViewsWithCov(reconstruction, options, view_ids):
    """bundle_adjustment.cc:454-499."""
    view_ids = [int(v) for v in view_ids]
    options = _no_inner(options)
    flat = _flatten(reconstruction, view_ids, [])
    summary, cc = _with_cov(reconstruction, options, flat, False)
    if cc is None:
        return summary, {}, 1.0
    nr_obs = int(sum(np.sum(reconstruction.obs_view == v) for v in view_ids))           # View::NumFeatures()
    redundancy = nr_obs * 2 - 6.0 * len(view_ids)
    factor = (2.0 * summary.final_cost) / redundancy
    return summary, {v: cc[v] * factor for v in view_ids}, factor


def BundleAdjustViewWithCov(reconstruction, options, view_id):
    """bundle_adjustment.cc:420-452."""
    summary, covs, factor = BundleAdjustViewsWithCov(reconstruction, options, [view_id])
    return summary, covs.get(int(view_id), np.eye(6)), factor


def SetOutlierTracksToUnestimated(track_ids, max_inlier_reprojection_error, min_triangulation_angle_degrees, reconstruction):
    """set_outlier_tracks_to_unestimated.cc:64-139 (the sweep that brackets every full BA in the estimators):
    un-estimates tracks that reproject behind a camera, whose mean squared reprojection error exceeds the
    threshold, or that no two views see under a sufficient angle.  Returns the number removed."""
    r = reconstruction
    tids = np.asarray([int(t) for t in track_ids], dtype=np.int64)
    tids = tids[r.track_estimated[tids]] if len(tids) else tids
    if not len(tids):
        return 0
    local = -np.ones(r.NumTracks(), dtype=np.int64)
    local[tids] = np.arange(len(tids))
    keep = (local[r.obs_track] >= 0) & r.view_estimated[r.obs_view]
    flat = capi.FlatProblem(r.cam_ext.copy(), r.group_intrinsics.copy(), r.group_model, r.view_group, np.ascontiguousarray(r.points[tids]),
                            r.obs_uv[keep], r.obs_view[keep], local[r.obs_track[keep]].astype(np.int32))
    err, nbehind, mincos = _ba.track_statistics(flat)
    bad_reproj = (nbehind > 0) | (err > max_inlier_reprojection_error ** 2)
    bad_angle = ~bad_reproj & ~(mincos < np.cos(np.deg2rad(min_triangulation_angle_degrees)))
    r.track_estimated[tids[bad_reproj | bad_angle]] = False
    return int(bad_reproj.sum() + bad_angle.sum())


def RemoveOutlierTracks(tracks_to_check, max_reprojection_error_in_pixels, min_triangulation_angle_degrees, reconstruction):
    """incremental_reconstruction_estimator.cc:599-610 / hybrid_reconstruction_estimator.cc:850-861: the estimators'
    sweep around every (partial) BA = SetOutlierTracksToUnestimated with the estimator's own angle option; returns the
    number of points removed (the reference only logs it)."""
    return SetOutlierTracksToUnestimated(tracks_to_check, max_reprojection_error_in_pixels, min_triangulation_angle_degrees,
                                         reconstruction)


def _select_good_tracks(r, view_ids, track_len, track_err, long_track_length_threshold, image_grid_cell_size_pixels,
                        min_num_optimized_tracks_per_view):
    """The selection rules of select_good_tracks_for_bundle_adjustment.cc:164-320 on precomputed per-track statistics
    (track_len = number of estimated views seeing the track, track_err = mean squared reprojection error).
    Reference semantics kept as they are written: a grid cell keeps the MINIMUM of (truncated length, error)
    (`std::min_element` with pair `operator<`, :62-66), and the top-up of a view takes its candidates in ascending
    TRACK ID order (`partial_sort` of pair<TrackId, statistics>, :246-249).  Where the reference iterates hash
    containers (views in the second pass) this walks ascending ids."""
    view_ids = sorted(int(v) for v in view_ids if r.view_estimated[int(v)])
    est_obs = r.view_estimated[r.obs_view] & r.track_estimated[r.obs_track]
    tl = np.minimum(track_len, long_track_length_threshold)
    selected = np.zeros(r.NumTracks(), dtype=bool)
    inv = 1.0 / image_grid_cell_size_pixels
    by_view = {}
    order = np.argsort(r.obs_view, kind="stable")
    bounds = np.searchsorted(r.obs_view[order], np.arange(r.NumViews() + 1))
    for v in view_ids:
        idx = order[bounds[v]:bounds[v + 1]]
        by_view[v] = idx[est_obs[idx]]
    for v in view_ids:                                   # best track of every image grid cell
        idx = by_view[v]
        if not len(idx):
            continue
        cell = (r.obs_uv[idx] * inv).astype(np.int64)    # Eigen cast<int>: truncation toward zero
        t = r.obs_track[idx]
        key = np.lexsort((track_err[t], tl[t], cell[:, 1], cell[:, 0]))
        c = cell[key]
        first = np.ones(len(key), dtype=bool)
        first[1:] = np.any(c[1:] != c[:-1], axis=1)
        selected[t[key][first]] = True
    for v in view_ids:                                   # at least K optimised tracks per view
        t = r.obs_track[by_view[v]]
        nopt = int(selected[t].sum())
        if nopt >= min_num_optimized_tracks_per_view or nopt == len(t):
            continue
        need = min(min_num_optimized_tracks_per_view - nopt, len(t) - nopt)
        cand = np.sort(t[~selected[t]])
        selected[cand[:need]] = True
    return np.flatnonzero(selected)


def SelectGoodTracksForBundleAdjustment(reconstruction, long_track_length_threshold, image_grid_cell_size_pixels,
                                        min_num_optimized_tracks_per_view, view_ids=None):
    """select_good_tracks_for_bundle_adjustment.cc:263-320 -> (True, track ids to optimise).  The per-track mean
    squared reprojection errors (ComputeTrackStatistics, :79-140: the expensive part) come from
    theia_hip_track_statistics over the estimated views."""
    r = reconstruction
    if view_ids is None:
        view_ids = [v for v in range(r.NumViews()) if r.view_estimated[v]]
    keep = r.view_estimated[r.obs_view] & r.track_estimated[r.obs_track]
    flat = capi.FlatProblem(r.cam_ext.copy(), r.group_intrinsics.copy(), r.group_model, r.view_group, r.points.copy(),
                            r.obs_uv[keep], r.obs_view[keep], r.obs_track[keep])
    err, _, _ = _ba.track_statistics(flat)
    track_len = np.bincount(r.obs_track[keep], minlength=r.NumTracks())
    sel = _select_good_tracks(r, view_ids, track_len, np.nan_to_num(err, nan=0.0), long_track_length_threshold,
                              image_grid_cell_size_pixels, min_num_optimized_tracks_per_view)
    return True, sel.tolist()
