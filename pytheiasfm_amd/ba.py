"""Thin Python wrapper over the BA C-ABI (theia_hip_ba_*): device-resident
handle, one-shot solve, and the introspection calls the parity tests use."""
import ctypes as C

import numpy as np

from . import _capi as capi


def default_options():
    o = capi.BaOptions()
    capi.lib().theia_ba_options_default(C.byref(o))
    return o


class BaHandle:
    """Problem resident in HBM (theia_hip_ba_create ... destroy)."""

    def __init__(self, problem, options):
        self.problem = problem
        self.options = options
        self._st = problem.as_struct()
        self._h = C.c_void_p()
        self._cb = None
        capi.check(capi.lib().theia_hip_ba_create(C.byref(self._st), C.byref(options), C.byref(self._h)))
        self.pd = 3 if options.use_homogeneous_point_parametrization else 4

    def close(self):
        if self._h:
            capi.lib().theia_hip_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, trace_capacity=256):
        s = capi.BaSummary()
        tr = capi.Trace(trace_capacity)
        tr.attach(s)
        capi.check(capi.lib().theia_hip_ba_run(self._h, C.byref(s)))
        tr.finish(s)
        return s, tr

    def reset(self, problem=None):
        """Re-upload parameters (of `problem`, default: the creating problem)."""
        st = (problem or self.problem).as_struct()
        capi.check(capi.lib().theia_hip_ba_reset_parameters(self._h, C.byref(st)))

    def set_shard(self, rank, world_size):
        """theia_hip_ba_set_shard: lets the MAX scalar ride in the SUM all-reduce (one collective less per iteration)."""
        L = capi.lib()
        L.theia_hip_ba_set_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        capi.check(L.theia_hip_ba_set_shard(self._h, int(rank), int(world_size)))

    def set_inner_global(self, full_problem, point_global_index):
        """theia_hip_ba_set_inner_global: inner iterations in a sharded solve -- the unsharded problem's observations (and
        priors) and the global index of each of this shard's points."""
        L = capi.lib()
        L.theia_hip_ba_set_inner_global.argtypes = [C.c_void_p, C.POINTER(capi.BaProblem), C.POINTER(C.c_int64)]
        idx = np.ascontiguousarray(point_global_index, dtype=np.int64)
        st = full_problem.as_struct()
        capi.check(L.theia_hip_ba_set_inner_global(self._h, C.byref(st), idx.ctypes.data_as(C.POINTER(C.c_int64))))

    def plan_info(self):
        """theia_hip_ba_plan_info: reduced size, K3 levels / flops per solve, fused-kernel runs, slow-path tracks."""
        L = capi.lib()
        L.theia_hip_ba_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        n = C.c_int32(0); lv = C.c_int32(0); fl = C.c_double(0.0); runs = C.c_int32(0); slow = C.c_int32(0)
        capi.check(L.theia_hip_ba_plan_info(self._h, C.byref(n), C.byref(lv), C.byref(fl), C.byref(runs), C.byref(slow)))
        return {"n": n.value, "k3_levels": lv.value, "k3_flops": fl.value, "fused_runs": runs.value, "slow_path_tracks": slow.value}

    def snapshot(self):
        """Keep a device-resident copy of the current parameters (theia_hip_ba_snapshot_parameters)."""
        L = capi.lib()
        L.theia_hip_ba_snapshot_parameters.argtypes = [C.c_void_p]
        capi.check(L.theia_hip_ba_snapshot_parameters(self._h))

    def restore(self):
        """Back to the snapshot without touching the host (theia_hip_ba_restore_parameters)."""
        L = capi.lib()
        L.theia_hip_ba_restore_parameters.argtypes = [C.c_void_p]
        capi.check(L.theia_hip_ba_restore_parameters(self._h))

    def set_options(self, options):
        capi.check(capi.lib().theia_hip_ba_set_options(self._h, C.byref(options)))
        self.options = options

    def download(self, problem=None):
        p = problem or self.problem
        st = p.as_struct()
        capi.check(capi.lib().theia_hip_ba_download(self._h, C.byref(st)))
        return p

    def covariance(self, points=False, cameras=False):
        """theia_hip_ba_covariance at the current state: (point_cov [np][d][d] or None, cam_cov [nc][6][6] or None)."""
        L = capi.lib()
        L.theia_hip_ba_covariance.argtypes = [C.c_void_p, capi.c_double_p, capi.c_double_p]
        pc = np.zeros((self.problem.points.shape[0], self.pd, self.pd)) if points else None
        cc = np.zeros((self.problem.cam_ext.shape[0], 6, 6)) if cameras else None
        capi.check(L.theia_hip_ba_covariance(self._h, capi.ptr(pc, C.c_double) if points else None,
                                             capi.ptr(cc, C.c_double) if cameras else None))
        return pc, cc

    def evaluate(self):
        n = self.problem.obs_uv.shape[0]
        cost = C.c_double(0)
        r = np.zeros((n, 2)); jc = np.zeros((n, 2, 6)); jp = np.zeros((n, 2, self.pd))
        valid = np.zeros(n, dtype=np.uint8)
        capi.check(capi.lib().theia_hip_ba_evaluate(
            self._h, C.byref(cost), capi.ptr(r, C.c_double), capi.ptr(jc, C.c_double),
            capi.ptr(jp, C.c_double), capi.ptr(valid, C.c_uint8)))
        return cost.value, r, jc, jp, valid

    def evaluate_ex(self):
        """evaluate() plus the intrinsics Jacobian J_intr[nobs][2][10]."""
        n = self.problem.obs_uv.shape[0]
        cost = C.c_double(0)
        r = np.zeros((n, 2)); jc = np.zeros((n, 2, 6)); jp = np.zeros((n, 2, self.pd)); ji = np.zeros((n, 2, 10))
        valid = np.zeros(n, dtype=np.uint8)
        capi.check(capi.lib().theia_hip_ba_evaluate_ex(
            self._h, C.byref(cost), capi.ptr(r, C.c_double), capi.ptr(jc, C.c_double),
            capi.ptr(jp, C.c_double), capi.ptr(ji, C.c_double), capi.ptr(valid, C.c_uint8)))
        return cost.value, r, jc, jp, ji, valid

    def reduced_system(self, radius):
        ncam = self.problem.cam_ext.shape[0]
        nmax = 6 * ncam + 10 * self.problem.intrinsics.shape[0]
        cap = nmax ** 2
        S = np.zeros(max(cap, 1)); rhs = np.zeros(max(nmax, 1)); n = C.c_int32(0)
        capi.check(capi.lib().theia_hip_ba_reduced_system(
            self._h, radius, C.byref(n), capi.ptr(S, C.c_double), capi.ptr(rhs, C.c_double), cap))
        n = n.value
        return S[: n * n].reshape(n, n).copy(), rhs[:n].copy()

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, op:int, stream:int) -> int (0 = ok)."""
        def tramp(ctx, buf, count, op, stream):
            try:
                return int(fn(buf, count, op, stream) or 0)
            except Exception as e:  # never propagate through the C frame
                import sys
                print(f"[pytheiasfm_amd] allreduce callback raised: {e!r}", file=sys.stderr)
                return -1
        self._cb = capi.ALLREDUCE_FN(tramp)
        capi.check(capi.lib().theia_hip_ba_set_allreduce(self._h, self._cb, None))


def dense_spd_solve(A, b):
    """K3 alone: x = A^-1 b through the reduced-camera Cholesky kernels."""
    A = np.ascontiguousarray(A, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    capi.check(capi.lib().theia_hip_dense_spd_solve(A.shape[0], capi.ptr(A, C.c_double), capi.ptr(b, C.c_double),
                                                    capi.ptr(x, C.c_double)))
    return x


def solve(problem, options, trace_capacity=256):
    """theia_hip_ba_solve: parameters of `problem` are updated in place."""
    s = capi.BaSummary()
    tr = capi.Trace(trace_capacity)
    tr.attach(s)
    st = problem.as_struct()
    capi.check(capi.lib().theia_hip_ba_solve(C.byref(st), C.byref(options), C.byref(s)))
    tr.finish(s)
    return s, tr


def problem_fingerprint(problem, options=None):
    """128-bit fingerprint of everything a handle is BUILT from -- the topology (observation -> camera / point indices,
    camera -> group, models, constant masks, observation kinds, prior masks), the observations themselves and the
    options -- and nothing a solve CHANGES (extrinsics, intrinsics, points, inverse depths).  Two problems with the same fingerprint can
    share one theia_hip_ba_create: the second only re-uploads its parameters (theia_hip_ba_reset_parameters)."""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:          # no xxhash on this machine: the standard library's 128-bit BLAKE2b (slower, same role)
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    p = problem
    h.update(np.array([p.cam_ext.shape[0], p.intrinsics.shape[0], p.points.shape[0], p.obs_uv.shape[0], p.flags], dtype=np.int64).tobytes())
    arrays = [p.obs_cam, p.obs_pt, p.obs_uv, p.obs_sqrt_info, p.group_model, p.cam_group, p.cam_const, p.group_const,
              p.point_const, p.obs_kind, p.cam_prior_mask, p.point_ref_cam, p.point_ref_bearing]
    for name in ("position", "gravity", "orientation"):
        pr = p.priors.get(name)
        arrays += [None, None] if pr is None else [pr[0], pr[1]]
    for a in arrays:
        if a is None:
            h.update(b"\x00none")
        else:
            h.update(b"\x01" + str(a.dtype).encode() + str(a.shape).encode())
            h.update(np.ascontiguousarray(a).data)
    if options is not None:
        h.update(bytes(options))
    return h.digest()


class ProblemCache:
    """The problem-IR cache of SURVEY.md 8(f) row 4: the full BA of a pipeline is called again and again on a
    reconstruction whose topology has not changed in between (outlier sweeps that remove nothing, the repeated
    BundleAdjustReconstruction of the global pipeline, re-runs with perturbed parameters), and building the handle -- sorting
    3 M observations into tiles, the fused kernel's run plan, the K3 schedule, 63-72 ms at C4 -- costs as much as sixty LM
    iterations.  The cache keeps the `capacity` most recently used handles (device-resident plan + observations) keyed by
    problem_fingerprint(); a hit re-uploads the parameters (1000 cameras + 500 000 points: < 2 ms) and runs.
    Inverse-depth problems are cached the same way (their handle keeps structure, observations and the reduced-system plan)."""

    def __init__(self, capacity=1):
        import threading
        self.capacity = int(capacity)
        self._handles = {}      # fingerprint -> idle BaHandle, insertion order = recency (a running handle is NOT in here)
        self._lock = threading.Lock()
        self.hits = 0
        self.misses = 0

    def clear(self):
        with self._lock:
            hs, self._handles = list(self._handles.values()), {}
        for h in hs:
            h.close()

    def solve(self, problem, options, trace_capacity=256):
        """As ba.solve(): parameters of `problem` are updated in place; returns (summary, trace)."""
        if self.capacity <= 0 or problem.obs_uv.shape[0] == 0:
            return solve(problem, options, trace_capacity)
        key = problem_fingerprint(problem, options)
        # a handle leaves the dictionary while it runs (ctypes releases the GIL inside run()): a second thread solving the
        # same topology builds its own handle, and no eviction can destroy a handle somebody is running
        with self._lock:
            h = self._handles.pop(key, None)
        if h is not None:
            try:
                h.reset(problem)
                with self._lock:
                    self.hits += 1
            except capi.TheiaHipError:
                h.close(); h = None
        if h is None:
            h = BaHandle(problem, options)
            h.problem = None          # the handle owns device copies only: do not pin the creating problem's host arrays
            with self._lock:
                self.misses += 1
        try:
            s, tr = h.run(trace_capacity)
            h.download(problem)
        except Exception:
            h.close()
            raise
        evict = []
        with self._lock:
            dup = self._handles.pop(key, None)     # another thread finished the same topology first: keep ours, drop theirs
            if dup is not None:
                evict.append(dup)
            self._handles[key] = h
            while len(self._handles) > self.capacity:
                evict.append(self._handles.pop(next(iter(self._handles))))
        for e in evict:
            e.close()
        return s, tr


def solve_views_batch(offsets, obs_uv, points, cam_ext, intrinsics, model, options, cam_const=None, obs_sqrt_info=None):
    """theia_hip_ba_views_batch: N independent BundleAdjustView problems
    (bundle_adjustment.cc:220-237) in one launch.  cam_ext [N][6] is updated in
    place; returns a list of BaSummary (no traces)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    num = len(offsets) - 1
    obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2)
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 4)
    intrinsics = np.ascontiguousarray(intrinsics, dtype=np.float64).reshape(num, capi.THEIA_MAX_INTRINSICS)
    model = np.ascontiguousarray(model, dtype=np.int32)
    if not (cam_ext.flags["C_CONTIGUOUS"] and cam_ext.dtype == np.float64 and cam_ext.shape == (num, 6)):
        raise capi.TheiaHipError(-1, "cam_ext must be a C-contiguous float64 [N][6] array (updated in place)")
    st = capi.BaViewBatch()
    st.num_problems = num
    st.offsets = offsets.ctypes.data_as(C.POINTER(C.c_int64))
    st.obs_uv = capi.ptr(obs_uv, C.c_double); st.points = capi.ptr(points, C.c_double)
    st.cam_ext = capi.ptr(cam_ext, C.c_double); st.intrinsics = capi.ptr(intrinsics, C.c_double)
    st.model = capi.ptr(model, C.c_int32)
    keep = [offsets, obs_uv, points, intrinsics, model]
    if cam_const is not None:
        cc = np.ascontiguousarray(cam_const, dtype=np.uint8); keep.append(cc)
        st.cam_const = capi.ptr(cc, C.c_uint8)
    if obs_sqrt_info is not None:
        si = np.ascontiguousarray(obs_sqrt_info, dtype=np.float64).reshape(-1, 2); keep.append(si)
        st.obs_sqrt_info = capi.ptr(si, C.c_double)
    summ = (capi.BaSummary * max(1, num))()
    L = capi.lib()
    L.theia_hip_ba_views_batch.argtypes = [C.POINTER(capi.BaViewBatch), C.POINTER(capi.BaOptions), C.POINTER(capi.BaSummary)]
    capi.check(L.theia_hip_ba_views_batch(C.byref(st), C.byref(options), summ))
    return [summ[i] for i in range(num)]


TWO_VIEW_EXACT, TWO_VIEW_CGNR = 0, 1


def solve_two_views_angular_batch(offsets, correspondences, rotation_position, options, linear_solver=TWO_VIEW_EXACT):
    """theia_hip_ba_two_views_angular_batch: N independent BundleAdjustTwoViewsAngular problems
    (bundle_adjust_two_views.cc:189-246) in one launch.  correspondences [total][4] = (x1, y1, x2, y2)
    in normalised image coordinates; rotation_position [N][6] (rotation_2 | position_2) is updated in
    place; returns a list of BaSummary (no traces).  linear_solver: TWO_VIEW_EXACT (any direct
    linear_solver_type) or TWO_VIEW_CGNR (ceres::CGNR + JACOBI, inexact steps)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    num = len(offsets) - 1
    corr = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 4)
    rp = rotation_position
    if not (rp.flags["C_CONTIGUOUS"] and rp.dtype == np.float64 and rp.shape == (num, 6)):
        raise capi.TheiaHipError(-1, "rotation_position must be a C-contiguous float64 [N][6] array (updated in place)")
    st = capi.BaTwoViewBatch()
    st.num_problems = num
    st.linear_solver = int(linear_solver)
    st.offsets = offsets.ctypes.data_as(C.POINTER(C.c_int64))
    st.correspondences = capi.ptr(corr, C.c_double)
    st.rotation_position = capi.ptr(rp, C.c_double)
    summ = (capi.BaSummary * max(1, num))()
    L = capi.lib()
    L.theia_hip_ba_two_views_angular_batch.argtypes = [C.POINTER(capi.BaTwoViewBatch), C.POINTER(capi.BaOptions), C.POINTER(capi.BaSummary)]
    capi.check(L.theia_hip_ba_two_views_angular_batch(C.byref(st), C.byref(options), summ))
    return [summ[i] for i in range(num)]


def solve_two_views_batch(offsets, correspondences, cam_ext, intrinsics, model, const_intrinsics, points, options):
    """theia_hip_ba_two_views_batch: N independent BundleAdjustTwoViews problems (bundle_adjust_two_views.cc:110-185) in one
    launch.  correspondences [total][4] pixels; cam_ext [N][2][6] (camera 2 updated), intrinsics [N][2][10] (focal lengths
    updated unless const_intrinsics [N][2]), model [N][2], points [total][4] XYZW (updated).  Returns a list of BaSummary."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    num = len(offsets) - 1
    corr = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 4)
    model = np.ascontiguousarray(model, dtype=np.int32).reshape(num, 2)
    kc = np.ascontiguousarray(const_intrinsics, dtype=np.uint8).reshape(num, 2)
    for name, a, shape in (("cam_ext", cam_ext, (num, 2, 6)), ("intrinsics", intrinsics, (num, 2, capi.THEIA_MAX_INTRINSICS)),
                           ("points", points, (corr.shape[0], 4))):
        if not (a.flags["C_CONTIGUOUS"] and a.dtype == np.float64 and a.shape == shape):
            raise capi.TheiaHipError(-1, f"{name} must be a C-contiguous float64 array of shape {shape} (updated in place)")
    st = capi.BaTwoViewFullBatch()
    st.num_problems = num
    st.offsets = offsets.ctypes.data_as(C.POINTER(C.c_int64))
    st.correspondences = capi.ptr(corr, C.c_double)
    st.cam_ext = capi.ptr(cam_ext, C.c_double); st.intrinsics = capi.ptr(intrinsics, C.c_double)
    st.model = capi.ptr(model, C.c_int32); st.const_intrinsics = capi.ptr(kc, C.c_uint8)
    st.points = capi.ptr(points, C.c_double)
    summ = (capi.BaSummary * max(1, num))()
    L = capi.lib()
    L.theia_hip_ba_two_views_batch.argtypes = [C.POINTER(capi.BaTwoViewFullBatch), C.POINTER(capi.BaOptions), C.POINTER(capi.BaSummary)]
    capi.check(L.theia_hip_ba_two_views_batch(C.byref(st), C.byref(options), summ))
    return [summ[i] for i in range(num)]


def optimize_homography_batch(offsets, correspondences, homographies, options):
    """theia_hip_optimize_homography_batch: N independent OptimizeHomography problems (bundle_adjust_two_views.cc:298-358).
    homographies [N][3][3] (row-major) is updated in place (and divided by H(2,2)); returns a list of BaSummary."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    num = len(offsets) - 1
    corr = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 4)
    H = homographies
    if not (H.flags["C_CONTIGUOUS"] and H.dtype == np.float64 and H.size == 9 * num):
        raise capi.TheiaHipError(-1, "homographies must be a C-contiguous float64 [N][3][3] array (updated in place)")
    summ = (capi.BaSummary * max(1, num))()
    L = capi.lib()
    L.theia_hip_optimize_homography_batch.argtypes = [C.c_int32, C.POINTER(C.c_int64), capi.c_double_p, capi.c_double_p,
                                                      C.POINTER(capi.BaOptions), C.POINTER(capi.BaSummary)]
    capi.check(L.theia_hip_optimize_homography_batch(num, offsets.ctypes.data_as(C.POINTER(C.c_int64)), capi.ptr(corr, C.c_double),
                                                     capi.ptr(H, C.c_double), C.byref(options), summ))
    return [summ[i] for i in range(num)]


def optimize_fundamental_matrix_batch(offsets, correspondences, fundamental_matrices, options):
    """theia_hip_optimize_fundamental_matrix_batch: N independent OptimizeFundamentalMatrix problems
    (bundle_adjust_two_views.cc:248-296).  fundamental_matrices [N][3][3] (row-major) is updated in place."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    num = len(offsets) - 1
    corr = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 4)
    F = fundamental_matrices
    if not (F.flags["C_CONTIGUOUS"] and F.dtype == np.float64 and F.size == 9 * num):
        raise capi.TheiaHipError(-1, "fundamental_matrices must be a C-contiguous float64 [N][3][3] array (updated in place)")
    summ = (capi.BaSummary * max(1, num))()
    L = capi.lib()
    L.theia_hip_optimize_fundamental_matrix_batch.argtypes = [C.c_int32, C.POINTER(C.c_int64), capi.c_double_p, capi.c_double_p,
                                                              C.POINTER(capi.BaOptions), C.POINTER(capi.BaSummary)]
    capi.check(L.theia_hip_optimize_fundamental_matrix_batch(num, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                                             capi.ptr(corr, C.c_double), capi.ptr(F, C.c_double), C.byref(options), summ))
    return [summ[i] for i in range(num)]


def optimize_relative_position_batch(offsets, correspondences, rotations):
    """theia_hip_optimize_relative_position_batch: N x OptimizeRelativePositionWithKnownRotation
    (optimize_relative_position_with_known_rotation.cc:116-191), one pair per wavefront.  correspondences [total][4]
    normalised, rotations [N][6] = rotation1 | rotation2 (angle-axis).  Returns (positions [N][3], iterations [N])."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    num = len(offsets) - 1
    corr = np.ascontiguousarray(correspondences, dtype=np.float64).reshape(-1, 4)
    rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(num, 6)
    pos = np.zeros((num, 3)); it = np.zeros(max(1, num), dtype=np.int32)
    L = capi.lib()
    L.theia_hip_optimize_relative_position_batch.argtypes = [C.c_int32, C.POINTER(C.c_int64), capi.c_double_p, capi.c_double_p, capi.c_double_p,
                                                             C.POINTER(C.c_int32)]
    capi.check(L.theia_hip_optimize_relative_position_batch(num, offsets.ctypes.data_as(C.POINTER(C.c_int64)), capi.ptr(corr, C.c_double),
                                                            capi.ptr(rot, C.c_double), capi.ptr(pos, C.c_double), it.ctypes.data_as(C.POINTER(C.c_int32))))
    return pos, it[:num]


def solve_tracks_batch(problem, options):
    """theia_hip_ba_tracks_batch: every point as an independent BundleAdjustTrack problem
    (bundle_adjustment.cc:262-285), cameras constant.  problem.points is updated in place;
    returns a list of BaSummary (one per point)."""
    st = problem.as_struct()
    num = problem.points.shape[0]
    summ = (capi.BaSummary * max(1, num))()
    L = capi.lib()
    L.theia_hip_ba_tracks_batch.argtypes = [C.POINTER(capi.BaProblem), C.POINTER(capi.BaOptions), C.POINTER(capi.BaSummary)]
    capi.check(L.theia_hip_ba_tracks_batch(C.byref(st), C.byref(options), summ))
    return [summ[i] for i in range(num)]


def track_statistics(problem):
    """theia_hip_track_statistics: (mean squared reprojection error, #views behind the camera,
    smallest ray cosine) per point."""
    st = problem.as_struct()
    num = problem.points.shape[0]
    err = np.zeros(max(1, num)); nb = np.zeros(max(1, num), dtype=np.int32); mc = np.zeros(max(1, num))
    L = capi.lib()
    L.theia_hip_track_statistics.argtypes = [C.POINTER(capi.BaProblem), capi.c_double_p, capi.c_int32_p, capi.c_double_p]
    capi.check(L.theia_hip_track_statistics(C.byref(st), capi.ptr(err, C.c_double), capi.ptr(nb, C.c_int32), capi.ptr(mc, C.c_double)))
    return err[:num], nb[:num], mc[:num]


def estimate_tracks(problem, obs_ray_dir, ba_options, min_triangulation_angle_degrees=3.0,
                    max_acceptable_reprojection_error_pixels=5.0, bundle_adjustment=True, triangulation_method=0):
    """theia_hip_estimate_tracks: TrackEstimator::EstimateTrack for every non-constant point of `problem`
    (triangulation_method = TriangulationMethodType: 0 MIDPOINT, 1 SVD, 2 L2_MINIMIZATION); problem.points is updated in place.  Returns (estimated [num_points] bool,
    {"bad_angles", "failed_triangulations", "bad_reprojections", "ba_failures"})."""
    st = problem.as_struct()
    num = problem.points.shape[0]
    rays = np.ascontiguousarray(obs_ray_dir, dtype=np.float64).reshape(-1, 3)
    if rays.shape[0] != problem.obs_uv.shape[0]:
        raise capi.TheiaHipError(-1, "one viewing ray per observation expected")
    eo = capi.TrackEstimateOptions()
    eo.min_triangulation_angle_degrees = float(min_triangulation_angle_degrees)
    eo.max_acceptable_reprojection_error_pixels = float(max_acceptable_reprojection_error_pixels)
    eo.bundle_adjustment = int(bool(bundle_adjustment))
    eo.triangulation_method = int(triangulation_method)
    est = np.zeros(max(1, num), dtype=np.uint8)
    cnt = (C.c_int32 * 4)()
    L = capi.lib()
    L.theia_hip_estimate_tracks.argtypes = [C.POINTER(capi.BaProblem), capi.c_double_p, C.POINTER(capi.BaOptions),
                                            C.POINTER(capi.TrackEstimateOptions), capi.c_uint8_p, C.POINTER(C.c_int32)]
    capi.check(L.theia_hip_estimate_tracks(C.byref(st), capi.ptr(rays, C.c_double), C.byref(ba_options), C.byref(eo),
                                           capi.ptr(est, C.c_uint8), cnt))
    return est[:num].astype(bool), {"bad_angles": cnt[0], "failed_triangulations": cnt[1], "bad_reprojections": cnt[2],
                                    "ba_failures": cnt[3]}
