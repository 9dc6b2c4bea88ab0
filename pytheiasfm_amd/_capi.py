"""ctypes binding of the C-ABI in include/theia_hip.h (libtheia_hip.so).

The product path has NO CPU fallback: if the HIP extension is missing, or no
gfx950 device is visible, every compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("THEIA_HIP_LIBRARY") or os.path.join(_HERE, "libtheia_hip.so")   # (the override: development builds of the same library, scripts/dev_*.sh)

THEIA_MAX_INTRINSICS = 10
THEIA_RANSAC_MODEL_STRIDE = 24
# error codes (include/theia_hip.h:26-33)
THEIA_HIP_ERR_INVALID_ARGUMENT, THEIA_HIP_ERR_NO_DEVICE, THEIA_HIP_ERR_UNSUPPORTED = -1, -2, -3
THEIA_HIP_ERR_OUT_OF_MEMORY, THEIA_HIP_ERR_INTERNAL = -4, -5

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class TheiaHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"theia_hip error {code}: {message}")
        self.code = code


class BaProblem(C.Structure):
    """theia_ba_problem (include/theia_hip.h)."""
    _fields_ = [
        ("num_cameras", C.c_int32), ("num_groups", C.c_int32), ("num_points", C.c_int32),
        ("flags", C.c_int32), ("num_obs", C.c_int64),
        ("cam_ext", c_double_p), ("intrinsics", c_double_p), ("group_model", c_int32_p),
        ("cam_group", c_int32_p), ("cam_const", c_uint8_p), ("group_const", c_uint8_p),
        ("points", c_double_p), ("point_const", c_uint8_p),
        ("obs_uv", c_double_p), ("obs_sqrt_info", c_double_p), ("obs_cam", c_int32_p),
        ("obs_pt", c_int32_p),
        ("cam_prior_mask", c_uint8_p), ("cam_position_prior", c_double_p), ("cam_position_prior_sqrt_info", c_double_p),
        ("cam_gravity_prior", c_double_p), ("cam_gravity_prior_sqrt_info", c_double_p),
        ("cam_orientation_prior", c_double_p), ("cam_orientation_prior_sqrt_info", c_double_p),
        ("obs_kind", c_uint8_p),
        ("point_ref_cam", c_int32_p), ("point_ref_bearing", c_double_p), ("point_inverse_depth", c_double_p),
    ]


class BaOptions(C.Structure):
    """theia_ba_options."""
    _fields_ = [
        ("loss_function_type", C.c_int32), ("intrinsics_to_optimize", C.c_int32),
        ("max_num_iterations", C.c_int32), ("use_homogeneous_point_parametrization", C.c_int32),
        ("constant_camera_orientation", C.c_int32), ("constant_camera_position", C.c_int32),
        ("orthographic_camera", C.c_int32), ("use_inner_iterations", C.c_int32),
        ("verbose", C.c_int32), ("prior_mask", C.c_int32),
        ("robust_loss_width", C.c_double), ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("max_trust_region_radius", C.c_double), ("max_solver_time_in_seconds", C.c_double),
        ("robust_loss_width_depth_prior", C.c_double),
    ]


class BaSummary(C.Structure):
    """theia_ba_summary."""
    _fields_ = [
        ("success", C.c_int32), ("termination_type", C.c_int32), ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("setup_time_in_seconds", C.c_double), ("solve_time_in_seconds", C.c_double),
        ("trace_capacity", C.c_int32), ("trace_size", C.c_int32),
        ("trace_cost", c_double_p), ("trace_gradient_max_norm", c_double_p),
        ("trace_step_norm", c_double_p), ("trace_radius", c_double_p),
        ("trace_accepted", c_int32_p),
        ("time_linearize", C.c_double), ("time_solve_reduced", C.c_double),
        ("time_backsub", C.c_double),
        ("time_kernel_linearize", C.c_double), ("num_linearize_launches", C.c_int32),
        ("reserved1", C.c_int32),
    ]


THEIA_PRIOR_POSITION, THEIA_PRIOR_GRAVITY, THEIA_PRIOR_ORIENTATION = 1, 2, 4
THEIA_BA_FLAG_KEEP_UNOBSERVED_CAMERAS, THEIA_BA_FLAG_INVERSE_DEPTH = 1, 2


class BaViewBatch(C.Structure):
    """theia_ba_view_batch."""
    _fields_ = [
        ("num_problems", C.c_int32), ("offsets", C.POINTER(C.c_int64)), ("obs_uv", c_double_p),
        ("obs_sqrt_info", c_double_p), ("points", c_double_p), ("cam_ext", c_double_p),
        ("intrinsics", c_double_p), ("model", c_int32_p), ("cam_const", c_uint8_p),
    ]


class BaTwoViewBatch(C.Structure):
    """theia_ba_two_view_batch."""
    _fields_ = [
        ("num_problems", C.c_int32), ("linear_solver", C.c_int32), ("offsets", C.POINTER(C.c_int64)),
        ("correspondences", c_double_p), ("rotation_position", c_double_p),
    ]


class BaTwoViewFullBatch(C.Structure):
    """theia_ba_two_view_full_batch."""
    _fields_ = [
        ("num_problems", C.c_int32), ("offsets", C.POINTER(C.c_int64)), ("correspondences", c_double_p),
        ("cam_ext", c_double_p), ("intrinsics", c_double_p), ("model", C.POINTER(C.c_int32)),
        ("const_intrinsics", C.POINTER(C.c_uint8)), ("points", c_double_p),
    ]


class RansacParams(C.Structure):
    """theia_ransac_params."""
    _fields_ = [
        ("error_thresh", C.c_double), ("failure_probability", C.c_double),
        ("min_inlier_ratio", C.c_double),
        ("min_iterations", C.c_int32), ("max_iterations", C.c_int32), ("use_mle", C.c_int32),
        ("use_lo", C.c_int32), ("lo_start_iterations", C.c_int32), ("use_Tdd_test", C.c_int32),
        ("seed", C.c_uint32), ("ransac_type", C.c_int32),
    ]


class TrackEstimateOptions(C.Structure):
    """theia_track_estimate_options (TrackEstimator::Options, estimate_track.h:58-83)."""
    _fields_ = [("min_triangulation_angle_degrees", C.c_double), ("max_acceptable_reprojection_error_pixels", C.c_double),
                ("bundle_adjustment", C.c_int32), ("triangulation_method", C.c_int32)]


class RansacBatch(C.Structure):
    _fields_ = [("estimator", C.c_int32), ("num_problems", C.c_int32),
                ("offsets", c_int64_p), ("data", c_double_p), ("estimator_params", c_double_p),
                ("seeds", C.POINTER(C.c_uint32))]


class RansacResult(C.Structure):
    _fields_ = [("success", c_int32_p), ("models", c_double_p), ("num_inliers", c_int32_p),
                ("inlier_mask", c_uint8_p), ("num_iterations", c_int32_p),
                ("confidence", c_double_p),
                ("hypotheses_evaluated", C.c_int64), ("models_scored", C.c_int64),
                ("time_fit_score_seconds", C.c_double), ("num_lo_iterations", c_int32_p),
                ("time_fit_seconds", C.c_double), ("time_score_seconds", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

# every symbol include/theia_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = [
    "theia_hip_init", "theia_hip_shutdown", "theia_hip_device_count", "theia_hip_last_error",
    "theia_hip_version", "theia_ba_options_default", "theia_hip_ba_solve", "theia_hip_ba_views_batch", "theia_hip_ba_two_views_angular_batch", "theia_hip_ba_two_views_batch", "theia_hip_optimize_homography_batch", "theia_hip_optimize_fundamental_matrix_batch", "theia_hip_ba_tracks_batch", "theia_hip_track_statistics", "theia_hip_ba_create",
    "theia_hip_ba_reset_parameters", "theia_hip_estimate_tracks", "theia_hip_ba_set_shard", "theia_hip_ba_snapshot_parameters", "theia_hip_ba_restore_parameters", "theia_hip_ba_set_options", "theia_hip_ba_run", "theia_hip_ba_download",
    "theia_hip_ba_destroy", "theia_hip_ba_covariance", "theia_hip_ba_evaluate", "theia_hip_ba_evaluate_ex", "theia_hip_ba_reduced_system",
    "theia_hip_ba_set_allreduce", "theia_hip_ba_set_inner_global", "theia_hip_ba_plan_info", "theia_hip_rccl_unique_id", "theia_hip_rccl_comm_create",
    "theia_hip_optimize_relative_position_batch", "theia_hip_rccl_comm_destroy", "theia_hip_rccl_comm_count", "theia_hip_ba_set_rccl", "theia_hip_dense_spd_solve", "theia_ransac_params_default",
    "theia_hip_ransac_estimate_batch", "theia_hip_five_point_relative_pose",
    "theia_hip_pose_from_three_points", "theia_hip_sqpnp", "theia_hip_dls_pnp", "theia_hip_dls_macaulay_terms", "theia_hip_four_point_pose_and_focal_length", "theia_hip_four_point_focal_length_radial_distortion", "theia_hip_four_point_focal_length_radial_distortion_ex", "theia_hip_release_scratch", "theia_hip_guided_knn", "theia_hip_randint_stream", "theia_hip_selftest_wave_primitives",
]

_lib = None


def lib():
    """Load libtheia_hip.so; raises loudly if the extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
            "(pytheiasfm_amd has no CPU fallback)")
    # Processes that also use torch (bench.py, the RCCL all-reduce callback) must
    # map torch's bundled HIP runtime BEFORE this library binds libamdhip64.so.7:
    # the reverse order leaves torch with "No HIP GPUs are available" (two HIP
    # runtimes in one process; measured on the MI355X box).
    if os.environ.get("THEIA_HIP_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    L.theia_hip_last_error.restype = C.c_char_p
    if os.environ.get("THEIA_HIP_ABORT_TRACE") == "1":   # development aid: native backtrace on SIGABRT / SIGSEGV
        L.theia_hip_debug_install_abort_trace()
    L.theia_hip_version.restype = C.c_char_p
    L.theia_hip_ba_create.argtypes = [C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(C.c_void_p)]
    L.theia_hip_ba_run.argtypes = [C.c_void_p, C.POINTER(BaSummary)]
    L.theia_hip_ba_download.argtypes = [C.c_void_p, C.POINTER(BaProblem)]
    L.theia_hip_ba_reset_parameters.argtypes = [C.c_void_p, C.POINTER(BaProblem)]
    L.theia_hip_ba_set_options.argtypes = [C.c_void_p, C.POINTER(BaOptions)]
    L.theia_hip_ba_destroy.argtypes = [C.c_void_p]
    L.theia_hip_ba_solve.argtypes = [C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(BaSummary)]
    L.theia_hip_ba_evaluate.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p]
    L.theia_hip_ba_evaluate_ex.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p]
    L.theia_hip_ba_reduced_system.argtypes = [C.c_void_p, C.c_double, c_int32_p, c_double_p, c_double_p, C.c_int64]
    L.theia_hip_ba_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    L.theia_ba_options_default.argtypes = [C.POINTER(BaOptions)]
    L.theia_hip_dense_spd_solve.argtypes = [C.c_int32, c_double_p, c_double_p, c_double_p]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise TheiaHipError(rc, lib().theia_hip_last_error().decode())


def ptr(a, ctype):
    """Pointer to a C-contiguous numpy array (or NULL for None)."""
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ctype))


class FlatProblem:
    """Owns the numpy arrays behind a theia_ba_problem (keeps them alive)."""

    def __init__(self, cam_ext, intrinsics, group_model, cam_group, points, obs_uv, obs_cam, obs_pt,
                 cam_const=None, group_const=None, point_const=None, obs_sqrt_info=None, flags=0):
        self.flags = int(flags)
        f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i4 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        u1 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
        self.cam_ext = f8(cam_ext).reshape(-1, 6)
        self.points = f8(points).reshape(-1, 4)
        intr = f8(intrinsics)
        if intr.ndim == 1:
            intr = intr.reshape(1, -1)
        full = np.zeros((intr.shape[0], THEIA_MAX_INTRINSICS))
        full[:, :intr.shape[1]] = intr
        self.intrinsics = full
        self.group_model = i4(group_model).reshape(-1)
        self.cam_group = i4(cam_group).reshape(-1)
        self.obs_uv = f8(obs_uv).reshape(-1, 2)
        self.obs_cam = i4(obs_cam).reshape(-1)
        self.obs_pt = i4(obs_pt).reshape(-1)
        self.cam_const = u1(cam_const)
        self.group_const = u1(group_const)
        self.point_const = u1(point_const)
        self.obs_sqrt_info = None if obs_sqrt_info is None else f8(obs_sqrt_info).reshape(-1, 2)
        assert self.cam_group.shape[0] == self.cam_ext.shape[0]
        assert self.group_model.shape[0] == self.intrinsics.shape[0]
        assert self.obs_cam.shape[0] == self.obs_uv.shape[0] == self.obs_pt.shape[0]
        # camera priors (set_priors): mask [Nc], three (vector [Nc][3], sqrt information [Nc][3][3]) pairs
        self.cam_prior_mask = None
        self.priors = {}
        # depth-prior rows (add_depth_priors): obs_kind [N] uint8 or None
        self.obs_kind = None
        # inverse-depth parametrisation (set_inverse_depth)
        self.point_ref_cam = None; self.point_ref_bearing = None; self.point_inverse_depth = None

    def add_depth_priors(self, obs_index, depth, variance=1.0):
        """One DepthPriorError row (depth_prior_error.h) per listed observation: a new observation row of kind
        THEIA_OBS_DEPTH_PRIOR on the same camera and point, obs_uv = (depth, 0), sqrt_info = (1/sqrt(variance), 1)."""
        idx = np.asarray(obs_index, dtype=np.int64).reshape(-1)
        depth = np.broadcast_to(np.asarray(depth, dtype=np.float64), idx.shape)
        var = np.broadcast_to(np.asarray(variance, dtype=np.float64), idx.shape)
        n0 = self.obs_uv.shape[0]
        kind = np.zeros(n0, np.uint8) if self.obs_kind is None else self.obs_kind
        si = np.ones((n0, 2)) if self.obs_sqrt_info is None else self.obs_sqrt_info
        self.obs_uv = np.ascontiguousarray(np.vstack([self.obs_uv, np.column_stack([depth, np.zeros(len(idx))])]))
        self.obs_sqrt_info = np.ascontiguousarray(np.vstack([si, np.column_stack([1.0 / np.sqrt(var), np.ones(len(idx))])]))
        self.obs_cam = np.ascontiguousarray(np.concatenate([self.obs_cam, self.obs_cam[idx]]).astype(np.int32))
        self.obs_pt = np.ascontiguousarray(np.concatenate([self.obs_pt, self.obs_pt[idx]]).astype(np.int32))
        self.obs_kind = np.ascontiguousarray(np.concatenate([kind, np.ones(len(idx), np.uint8)]))
        return self

    def set_priors(self, mask, position=None, gravity=None, orientation=None):
        """Camera priors: mask[c] = THEIA_PRIOR_* bits; each kind = (vectors [Nc][3], sqrt_information [Nc][3][3])."""
        nc = self.cam_ext.shape[0]
        self.cam_prior_mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(nc)
        self.priors = {}
        for name, pr in (("position", position), ("gravity", gravity), ("orientation", orientation)):
            if pr is not None:
                v = np.ascontiguousarray(pr[0], dtype=np.float64).reshape(nc, 3)
                s = np.ascontiguousarray(pr[1], dtype=np.float64).reshape(nc, 3, 3)
                self.priors[name] = (v, s)
        return self

    def copy(self):
        q = FlatProblem(self.cam_ext.copy(), self.intrinsics.copy(), self.group_model, self.cam_group,
                        self.points.copy(), self.obs_uv, self.obs_cam, self.obs_pt,
                        self.cam_const, self.group_const, self.point_const, self.obs_sqrt_info, self.flags)
        q.cam_prior_mask = self.cam_prior_mask
        q.priors = dict(self.priors)
        q.obs_kind = self.obs_kind
        if self.point_ref_cam is not None:
            q.set_inverse_depth(self.point_ref_cam, self.point_ref_bearing, self.point_inverse_depth.copy())
        return q

    def set_inverse_depth(self, ref_cam, bearing, inverse_depth):
        """Inverse-depth parametrisation (THEIA_BA_FLAG_INVERSE_DEPTH): per point the reference camera index,
        Track::ReferenceBearingVector() and Track::InverseDepth() (in / out); `points` is ignored by the solve."""
        n = self.points.shape[0]
        self.point_ref_cam = np.ascontiguousarray(ref_cam, dtype=np.int32).reshape(n)
        self.point_ref_bearing = np.ascontiguousarray(bearing, dtype=np.float64).reshape(n, 3)
        self.point_inverse_depth = np.ascontiguousarray(inverse_depth, dtype=np.float64).reshape(n)
        self.flags |= THEIA_BA_FLAG_INVERSE_DEPTH

    def as_struct(self):
        p = BaProblem()
        p.num_cameras = self.cam_ext.shape[0]
        p.num_groups = self.intrinsics.shape[0]
        p.num_points = self.points.shape[0]
        p.num_obs = self.obs_uv.shape[0]
        p.flags = self.flags
        p.cam_ext = ptr(self.cam_ext, C.c_double)
        p.intrinsics = ptr(self.intrinsics, C.c_double)
        p.group_model = ptr(self.group_model, C.c_int32)
        p.cam_group = ptr(self.cam_group, C.c_int32)
        p.cam_const = ptr(self.cam_const, C.c_uint8)
        p.group_const = ptr(self.group_const, C.c_uint8)
        p.points = ptr(self.points, C.c_double)
        p.point_const = ptr(self.point_const, C.c_uint8)
        p.obs_uv = ptr(self.obs_uv, C.c_double)
        p.obs_sqrt_info = ptr(self.obs_sqrt_info, C.c_double)
        p.obs_cam = ptr(self.obs_cam, C.c_int32)
        p.obs_pt = ptr(self.obs_pt, C.c_int32)
        p.obs_kind = ptr(self.obs_kind, C.c_uint8)
        if self.point_ref_cam is not None:
            p.point_ref_cam = ptr(self.point_ref_cam, C.c_int32)
            p.point_ref_bearing = ptr(self.point_ref_bearing, C.c_double)
            p.point_inverse_depth = ptr(self.point_inverse_depth, C.c_double)
        if self.cam_prior_mask is not None:
            p.cam_prior_mask = ptr(self.cam_prior_mask, C.c_uint8)
            for name in ("position", "gravity", "orientation"):
                if name in self.priors:
                    setattr(p, "cam_%s_prior" % name, ptr(self.priors[name][0], C.c_double))
                    setattr(p, "cam_%s_prior_sqrt_info" % name, ptr(self.priors[name][1], C.c_double))
        return p


class Trace:
    """Caller-allocated per-iteration trace arrays of a theia_ba_summary."""

    def __init__(self, capacity=256):
        self.cost = np.zeros(capacity)
        self.gradient_max_norm = np.zeros(capacity)
        self.step_norm = np.zeros(capacity)
        self.radius = np.zeros(capacity)
        self.accepted = np.zeros(capacity, dtype=np.int32)
        self.capacity = capacity
        self.size = 0

    def attach(self, s):
        s.trace_capacity = self.capacity
        s.trace_size = 0
        s.trace_cost = ptr(self.cost, C.c_double)
        s.trace_gradient_max_norm = ptr(self.gradient_max_norm, C.c_double)
        s.trace_step_norm = ptr(self.step_norm, C.c_double)
        s.trace_radius = ptr(self.radius, C.c_double)
        s.trace_accepted = ptr(self.accepted, C.c_int32)

    def finish(self, s):
        self.size = s.trace_size
        for name in ("cost", "gradient_max_norm", "step_norm", "radius", "accepted"):
            setattr(self, name, getattr(self, name)[: self.size])
