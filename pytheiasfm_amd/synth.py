"""Deterministic synthetic workloads of BASELINE.json.

synth_ba_v1      ring-of-cameras reconstruction (configs C1/C2/C4)
synth_ransac_v1  batch of two-view / 2D-3D correspondence problems (config C5)

Randomness is a counter-based splitmix64 stream written here (identical on
every platform; no std:: / numpy distributions), so the same seed gives the
same scene in this container and on the GPU box.
"""
import numpy as np

from ._capi import FlatProblem

MASK = (1 << 64) - 1
CAM_PINHOLE = 0
CAM_DOUBLE_SPHERE = 5


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


class Stream:
    """Counter-based random stream: value k of stream (seed, tag) is a pure function."""

    def __init__(self, seed, tag):
        self.base = splitmix64(np.uint64((seed * 0x100000001B3 + tag) & MASK))

    def bits(self, idx):
        idx = np.asarray(idx, dtype=np.uint64)
        with np.errstate(over="ignore"):
            return splitmix64(self.base + idx * np.uint64(0xD1342543DE82EF95))

    def uniform(self, idx):
        return (self.bits(idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, idx):
        idx = np.asarray(idx, dtype=np.uint64)
        u1 = self.uniform(2 * idx)
        u2 = self.uniform(2 * idx + 1)
        return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)

    def integers(self, idx, n):
        return (self.bits(idx) % np.uint64(n)).astype(np.int64)


# ------------------------------------------------------------------ rotations
def angle_axis_to_matrix(w):
    """Rodrigues, batched: w (...,3) -> R (...,3,3)."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)
    small = th < 1e-12
    ths = np.where(small, 1.0, th)
    k = w / ths[..., None]
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1] = -k[..., 2]; K[..., 0, 2] = k[..., 1]
    K[..., 1, 0] = k[..., 2]; K[..., 1, 2] = -k[..., 0]
    K[..., 2, 0] = -k[..., 1]; K[..., 2, 1] = k[..., 0]
    s = np.sin(th)[..., None, None]
    c = np.cos(th)[..., None, None]
    R = np.eye(3) + s * K + (1.0 - c) * (K @ K)
    if np.any(small):
        R = np.where(small[..., None, None], np.eye(3), R)
    return R


def matrix_to_angle_axis(R):
    """Batched log map through the quaternion (robust near pi)."""
    R = np.asarray(R, dtype=np.float64)
    out = np.zeros(R.shape[:-2] + (3,))
    Rf = R.reshape(-1, 3, 3)
    of = out.reshape(-1, 3)
    for i, M in enumerate(Rf):
        tr = M[0, 0] + M[1, 1] + M[2, 2]
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2
            q = np.array([0.25 * s, (M[2, 1] - M[1, 2]) / s, (M[0, 2] - M[2, 0]) / s, (M[1, 0] - M[0, 1]) / s])
        elif M[0, 0] > M[1, 1] and M[0, 0] > M[2, 2]:
            s = np.sqrt(1.0 + M[0, 0] - M[1, 1] - M[2, 2]) * 2
            q = np.array([(M[2, 1] - M[1, 2]) / s, 0.25 * s, (M[0, 1] + M[1, 0]) / s, (M[0, 2] + M[2, 0]) / s])
        elif M[1, 1] > M[2, 2]:
            s = np.sqrt(1.0 + M[1, 1] - M[0, 0] - M[2, 2]) * 2
            q = np.array([(M[0, 2] - M[2, 0]) / s, (M[0, 1] + M[1, 0]) / s, 0.25 * s, (M[1, 2] + M[2, 1]) / s])
        else:
            s = np.sqrt(1.0 + M[2, 2] - M[0, 0] - M[1, 1]) * 2
            q = np.array([(M[1, 0] - M[0, 1]) / s, (M[0, 2] + M[2, 0]) / s, (M[1, 2] + M[2, 1]) / s, 0.25 * s])
        if q[0] < 0:
            q = -q
        sn = np.linalg.norm(q[1:])
        if sn < 1e-15:
            of[i] = 2.0 * q[1:]
        else:
            of[i] = q[1:] / sn * (2.0 * np.arctan2(sn, q[0]))
    return out


# ----------------------------------------------------------------- projection
def project(model, intr, cam_ext, X):
    """Reference projection (reprojection_error.h:54-110) for arrays of
    observations: cam_ext (M,6), intr (M,>=7), X (M,4). Returns uv (M,2), ok (M,)."""
    C = cam_ext[:, :3]
    R = angle_axis_to_matrix(cam_ext[:, 3:6])
    p = X[:, :3] - X[:, 3:4] * C
    q = np.einsum("nij,nj->ni", R, p)
    f, a, s, cx, cy = intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], intr[:, 4]
    ok = np.ones(len(X), dtype=bool)
    if model == CAM_PINHOLE:
        x = q[:, 0] / q[:, 2]; y = q[:, 1] / q[:, 2]
        r2 = x * x + y * y
        d = 1.0 + r2 * (intr[:, 5] + intr[:, 6] * r2)
        dx, dy = x * d, y * d
        ok &= q[:, 2] > 0
    elif model == CAM_DOUBLE_SPHERE:
        xi, al = intr[:, 5], intr[:, 6]
        r2 = q[:, 0] ** 2 + q[:, 1] ** 2
        d1 = np.sqrt(r2 + q[:, 2] ** 2)
        w1 = np.where(al > 0.5, (1 - al) / al, al / (1 - al))
        w2 = (w1 + xi) / np.sqrt(2 * w1 * xi + xi * xi + 1)
        ok &= q[:, 2] > -w2 * d1
        k = xi * d1 + q[:, 2]
        d2 = np.sqrt(r2 + k * k)
        n = al * d2 + (1 - al) * k
        dx, dy = q[:, 0] / n, q[:, 1] / n
        ok &= q[:, 2] > 0
    else:
        raise ValueError("model")
    u = f * dx + s * dy + cx
    v = f * a * dy + cy
    return np.stack([u, v], axis=1), ok


PINHOLE_INTR = np.array([1000.0, 1.0, 0.0, 960.0, 540.0, -0.05, 0.01])
DOUBLE_SPHERE_INTR = np.array([600.0, 1.0, 0.0, 960.0, 540.0, -0.2, 0.55])


def synth_ba_v1(num_views, num_tracks, seed=0xBA5E0000, num_groups=8, mixed_models=False,
                pixel_noise=0.5, sigma_pos=0.05, sigma_rot_deg=0.5, sigma_pt=0.02,
                fix_gauge=False, return_truth=False):
    """SURVEY.md 8(d) "synth_ba_v1": ring of cameras looking at the origin,
    tracks over contiguous windows of the ring (banded reduced system)."""
    nv, nt = int(num_views), int(num_tracks)
    num_groups = min(num_groups, nv)
    # --- cameras
    sc = Stream(seed, 1)
    ci = np.arange(nv)
    ang = 2.0 * np.pi * ci / nv
    pos = np.stack([10.0 * np.cos(ang), 10.0 * np.sin(ang), 2.0 * sc.uniform(ci) - 1.0], axis=1)
    z = -pos / np.linalg.norm(pos, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(up[None, :], z); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    Rlook = np.stack([x, y, z], axis=1)  # rows = camera axes in world
    jit = np.deg2rad(0.5) * np.stack([sc.normal(3 * ci + 100000), sc.normal(3 * ci + 100001), sc.normal(3 * ci + 100002)], axis=1)
    R = angle_axis_to_matrix(jit) @ Rlook
    aa = matrix_to_angle_axis(R)
    cam_gt = np.concatenate([pos, aa], axis=1)
    cam_group = (ci % num_groups).astype(np.int32)
    group_model = np.full(num_groups, CAM_PINHOLE, dtype=np.int32)
    intr = np.tile(PINHOLE_INTR, (num_groups, 1))
    if mixed_models:
        odd = np.arange(num_groups) % 2 == 1
        group_model[odd] = CAM_DOUBLE_SPHERE
        intr[odd] = DOUBLE_SPHERE_INTR
    # --- tracks: length 2 + (hash mod 9), contiguous window start hash2 mod nv
    st = Stream(seed, 2)
    ti = np.arange(nt)
    L = 2 + st.integers(ti, 9)
    L = np.minimum(L, nv)
    w0 = st.integers(ti + (1 << 40), nv)
    obs_pt = np.repeat(ti, L)
    within = np.arange(len(obs_pt)) - np.repeat(np.cumsum(L) - L, L)
    obs_cam = ((np.repeat(w0, L) + within) % nv).astype(np.int32)
    # --- points U[-3,3]^3, re-drawn until every observation is in the image
    sp = Stream(seed, 3)
    pts = np.zeros((nt, 4)); pts[:, 3] = 1.0
    todo = np.ones(nt, dtype=bool)
    rnd = 0
    while todo.any() and rnd < 64:
        idx = np.nonzero(todo)[0]
        for k in range(3):
            pts[idx, k] = 6.0 * sp.uniform(idx * 256 + rnd * 4 + k) - 3.0
        sel = todo[obs_pt]
        oc, op = obs_cam[sel], obs_pt[sel]
        bad_pt = np.zeros(nt, dtype=bool)
        for m in np.unique(group_model):
            mm = group_model[cam_group[oc]] == m
            if not mm.any():
                continue
            uv, ok = project(m, intr[cam_group[oc[mm]]], cam_gt[oc[mm]], pts[op[mm]])
            ok &= (uv[:, 0] >= 0) & (uv[:, 0] < 1920) & (uv[:, 1] >= 0) & (uv[:, 1] < 1080)
            bad_pt[op[mm][~ok]] = True
        todo = bad_pt
        rnd += 1
    if todo.any():
        raise RuntimeError("synth_ba_v1: could not place all points")
    # --- observations = exact projection + N(0, pixel_noise)
    nobs = len(obs_pt)
    obs_uv = np.zeros((nobs, 2))
    for m in np.unique(group_model):
        mm = group_model[cam_group[obs_cam]] == m
        uv, _ = project(m, intr[cam_group[obs_cam[mm]]], cam_gt[obs_cam[mm]], pts[obs_pt[mm]])
        obs_uv[mm] = uv
    sn = Stream(seed, 4)
    oi = np.arange(nobs)
    obs_uv[:, 0] += pixel_noise * sn.normal(2 * oi)
    obs_uv[:, 1] += pixel_noise * sn.normal(2 * oi + 1)
    # --- initial state = ground truth perturbed
    spert = Stream(seed, 5)
    cam0 = cam_gt.copy()
    for k in range(3):
        cam0[:, k] += sigma_pos * spert.normal(6 * ci + k)
        cam0[:, 3 + k] += np.deg2rad(sigma_rot_deg) * spert.normal(6 * ci + 3 + k)
    pts0 = pts.copy()
    for k in range(3):
        pts0[:, k] += sigma_pt * spert.normal((1 << 32) + 3 * ti + k)
    cam_const = None
    if fix_gauge:
        cam_const = np.zeros(nv, dtype=np.uint8)
        cam_const[0] = 3; cam_const[nv // 2] = 3
        cam0[0] = cam_gt[0]; cam0[nv // 2] = cam_gt[nv // 2]
    prob = FlatProblem(cam0, intr, group_model, cam_group, pts0, obs_uv, obs_cam, obs_pt.astype(np.int32),
                       cam_const=cam_const)
    if return_truth:
        return prob, cam_gt, pts
    return prob


BA_CONFIGS = {
    # name: (views, tracks, seed, mixed pinhole + double-sphere)
    "C1": (20, 2000, 0xBA5E0001, False),
    "C2": (200, 50000, 0xBA5E0002, False),
    "C4": (1000, 500000, 0xBA5E0004, True),
}


def ba_config(name, **kw):
    nv, nt, seed, mixed = BA_CONFIGS[name]
    return synth_ba_v1(nv, nt, seed=seed, mixed_models=mixed, **kw)


def shard_tracks(problem, rank, world_size):
    """Multi-GPU partition: tracks (with their observations) are dealt to ranks in blocks balanced by
    sum L^2 (Schur work) that are contiguous in CAMERA order -- tracks sorted by the first camera that sees them -- so
    that a rank's tracks see a window of the cameras: the columns of the reduced camera system that only one rank touches
    are then factored by that rank alone (the distributed K3 of csrc/ba_solver.hip: sync_plan).  Cameras and intrinsics
    are replicated.  Returns the rank's FlatProblem and the global indices of its tracks (ascending)."""
    npts = problem.points.shape[0]
    L = np.bincount(problem.obs_pt, minlength=npts).astype(np.float64)
    first_cam = np.full(npts, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first_cam, problem.obs_pt, problem.obs_cam.astype(np.int64))
    order = np.argsort(first_cam, kind="stable")
    w = np.cumsum((L * L)[order])
    total = w[-1] if npts else 0.0
    bounds = np.searchsorted(w, total * np.arange(1, world_size) / world_size, side="left")
    edges = np.concatenate([[0], bounds, [npts]]).astype(np.int64)
    ids = np.sort(order[edges[rank]:edges[rank + 1]])
    local = np.full(npts, -1, dtype=np.int64)
    local[ids] = np.arange(len(ids))
    sel = local[problem.obs_pt] >= 0
    pc = None if problem.point_const is None else problem.point_const[ids]
    si = None if problem.obs_sqrt_info is None else problem.obs_sqrt_info[sel]
    shard = FlatProblem(problem.cam_ext.copy(), problem.intrinsics.copy(), problem.group_model,
                        problem.cam_group, problem.points[ids].copy(), problem.obs_uv[sel],
                        problem.obs_cam[sel], local[problem.obs_pt[sel]].astype(problem.obs_pt.dtype), cam_const=problem.cam_const,
                        group_const=problem.group_const, point_const=pc, obs_sqrt_info=si, flags=problem.flags | 1)
    return shard, ids


# --------------------------------------------------------------------- RANSAC
def _random_rotations(stream, idx, max_angle_rad):
    ax = np.stack([stream.normal(4 * idx), stream.normal(4 * idx + 1), stream.normal(4 * idx + 2)], axis=1)
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = max_angle_rad * stream.uniform(4 * idx + 3)
    return angle_axis_to_matrix(ax * ang[:, None])


def synth_ransac_v1(num_problems, num_corr, kind="relative", seed=0x5AC50000, focal=1000.0,
                    noise_px=1.0, inlier_lo=0.3, inlier_hi=0.8):
    """SURVEY.md 8(d) "synth_ransac_v1" (config C5): per problem a random pose
    (rotation <= 30 deg, unit baseline), 3-D points in a frustum of depth
    [4, 10], inlier ratio U[0.3, 0.8], inliers + N(0, 1 px / f) noise, outliers
    uniform in the normalised image.
      kind = "relative": data [P*N][4] = (x1 y1 x2 y2), FeatureCorrespondence
      kind = "absolute": data [P*N][5] = (u v X Y Z),   FeatureCorrespondence2D3D
      kind = "fundamental": as "relative" but in pixels (focal, principal point 500,400)
      kind = "uncalibrated": pixels with the principal point removed, focal lengths f and 1.25 f
      kind = "homography":  as "fundamental" with the 3-D points on one plane per problem
      kind = "known_orientation": as "relative" with identity rotation (features already rotated)
      kind = "plane": data [P*N][3] = 3-D points, inliers within noise_px/focal of a plane
    Returns data, offsets, truth dict."""
    P, N = int(num_problems), int(num_corr)
    sp = Stream(seed, 11)
    pi = np.arange(P)
    R = _random_rotations(sp, pi, np.deg2rad(30.0))                       # (P,3,3)
    if kind == "known_orientation":
        R = np.broadcast_to(np.eye(3), (P, 3, 3)).copy()
    t = np.stack([sp.normal(8 * pi + (1 << 33)), sp.normal(8 * pi + 1 + (1 << 33)), sp.normal(8 * pi + 2 + (1 << 33))], axis=1)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    ratio = inlier_lo + (inlier_hi - inlier_lo) * sp.uniform(pi + (1 << 34))
    sd = Stream(seed, 12)
    gi = np.arange(P * N).reshape(P, N)
    depth = 4.0 + 6.0 * sd.uniform(6 * gi)
    x1 = 0.5 * (2.0 * sd.uniform(6 * gi + 1) - 1.0)
    y1 = 0.5 * (2.0 * sd.uniform(6 * gi + 2) - 1.0)
    plane_n = plane_d = None
    if kind in ("homography", "plane"):
        # one plane per problem, n . X = d with n within ~25 deg of the optical axis
        plane_n = np.stack([0.4 * (2.0 * sp.uniform(8 * pi + 3 + (1 << 35)) - 1.0),
                            0.4 * (2.0 * sp.uniform(8 * pi + 4 + (1 << 35)) - 1.0), np.ones(P)], axis=1)
        plane_n /= np.linalg.norm(plane_n, axis=1, keepdims=True)
        plane_d = 5.0 + 3.0 * sp.uniform(8 * pi + 5 + (1 << 35))
        on_plane = plane_d[:, None] / (plane_n[:, None, 0] * x1 + plane_n[:, None, 1] * y1 + plane_n[:, None, 2])
        if kind == "homography":
            depth = on_plane
    X = np.stack([x1 * depth, y1 * depth, depth], axis=2)                 # (P,N,3) in camera-1 / world frame
    X2 = np.einsum("pij,pnj->pni", R, X) + t[:, None, :]
    x2 = X2[:, :, :2] / X2[:, :, 2:3]
    sig = noise_px / focal
    is_in = (np.arange(N)[None, :] < np.floor(ratio * N)[:, None])
    nz = np.stack([sd.normal(4 * gi + (1 << 40)), sd.normal(4 * gi + 1 + (1 << 40)),
                   sd.normal(4 * gi + 2 + (1 << 40)), sd.normal(4 * gi + 3 + (1 << 40))], axis=2) * sig
    out2 = np.stack([2.0 * sd.uniform(6 * gi + 3) - 1.0, 2.0 * sd.uniform(6 * gi + 4) - 1.0], axis=2) * 0.6
    if kind == "relative":
        a = np.stack([x1, y1], axis=2) + nz[:, :, :2]
        b = np.where(is_in[:, :, None], x2 + nz[:, :, 2:], out2)
        data = np.concatenate([a, b], axis=2).reshape(P * N, 4)
    elif kind == "absolute":
        uv = np.where(is_in[:, :, None], x2 + nz[:, :, 2:], out2)
        data = np.concatenate([uv, X], axis=2).reshape(P * N, 5)
    elif kind in ("fundamental", "homography", "known_orientation", "uncalibrated"):
        a = np.stack([x1, y1], axis=2) + nz[:, :, :2]
        b = np.where(is_in[:, :, None], x2 + nz[:, :, 2:], out2)
        if kind == "uncalibrated":
            a = a * focal
            b = b * (1.25 * focal)
        elif kind != "known_orientation":
            pp = np.array([500.0, 400.0])
            a = a * focal + pp
            b = b * focal + pp
        data = np.concatenate([a, b], axis=2).reshape(P * N, 4)
    elif kind == "plane":
        Xp = np.stack([x1 * on_plane, y1 * on_plane, on_plane], axis=2) + nz[:, :, :1] * plane_n[:, None, :]
        data = np.where(is_in[:, :, None], Xp, X).reshape(P * N, 3)
    else:
        raise ValueError(kind)
    # shuffle within each problem so inliers are not a prefix
    perm = np.argsort(sd.uniform(gi + (1 << 44)), axis=1)
    data = data.reshape(P, N, -1)[np.arange(P)[:, None], perm].reshape(P * N, -1)
    is_in = is_in[np.arange(P)[:, None], perm]
    offsets = (np.arange(P + 1) * N).astype(np.int64)
    truth = {"R": R, "t": t, "position": -np.einsum("pji,pj->pi", R, t), "inlier": is_in, "ratio": ratio,
             "plane_normal": plane_n, "plane_d": plane_d}
    return np.ascontiguousarray(data), offsets, truth


def synth_relpos_v1(npairs, ncorr, seed=0x5AC5E100, noise=5e-4):
    """View pairs with KNOWN rotations for OptimizeRelativePositionWithKnownRotation (one call per view-graph edge in the
    reference's global pipeline): normalised correspondences [npairs * ncorr][4], offsets, rotations [npairs][6] (angle-axis of
    view 1 | view 2, world -> camera) and the true unit relative positions R1 (c2 - c1) / |.|."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    w = 0.3 * rng.randn(npairs, 2, 3)
    c1 = 0.2 * rng.randn(npairs, 3); c2 = c1 + rng.randn(npairs, 3)
    corr = np.zeros((npairs, ncorr, 4)); truth = np.zeros((npairs, 3))
    for k in range(npairs):
        R1 = angle_axis_to_matrix(w[k, 0]); R2 = angle_axis_to_matrix(w[k, 1])
        X = rng.uniform(-2, 2, size=(ncorr, 3)) + R1.T @ np.array([0.0, 0.0, 6.0]) + c1[k]
        p1 = (X - c1[k]) @ R1.T; p2 = (X - c2[k]) @ R2.T
        corr[k, :, 0:2] = p1[:, :2] / p1[:, 2:3]; corr[k, :, 2:4] = p2[:, :2] / p2[:, 2:3]
        t = R1 @ (c2[k] - c1[k]); truth[k] = t / np.linalg.norm(t)
    corr += noise * rng.randn(npairs, ncorr, 4)
    offsets = np.arange(npairs + 1, dtype=np.int64) * ncorr
    return np.ascontiguousarray(corr.reshape(-1, 4)), offsets, np.ascontiguousarray(w.reshape(npairs, 6)), truth
