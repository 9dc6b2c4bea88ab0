"""The scenes of the reference's DlsPnp tests (sfm/pose/dls_pnp_test.cc:128-316) as data, and its acceptance check
(dls_pnp_test.cc:60-126).  The reference draws points / noise from its own generator (seed 59); the numbers here come
from numpy with fixed seeds, the geometry and the tolerances are the reference's."""
import numpy as np

EIGHT = np.array([[-1, 3, 3], [1, -1, 2], [-1, 1, 2], [2, 1, 3], [-1, -3, 2], [1, -2, 1], [-1, 4, 2], [-2, 2, 3]], dtype=np.float64)


def angle_axis_quat(deg, axis):
    axis = np.asarray(axis, dtype=np.float64); axis = axis / np.linalg.norm(axis)
    h = np.deg2rad(deg) / 2.0
    return np.concatenate([[np.cos(h)], np.sin(h) * axis])


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def scenes():
    """(name, world points, expected quaternion, expected translation, noise, max reprojection error (squared), max angular
    distance, max squared translation difference)"""
    d = np.deg2rad
    out = [
        ("Basic", EIGHT[:4], angle_axis_quat(13, [0, 0, 1]), np.array([1.0, 1, 1]), 0.0, 1e-4, 1e-5, 1e-8),
        ("NoiseTest", EIGHT, angle_axis_quat(13, [0, 0, 1]), np.array([1.0, 1, 1]), 1 / 512, 5e-3, d(0.25), 1e-2),
        ("NoRotation", EIGHT, angle_axis_quat(0, [0, 0, 1]), np.array([1.0, 1, 1]), 1 / 512, 5e-3, d(0.25), 5e-4),
        ("NoTranslation", EIGHT, angle_axis_quat(13, [0, 0, 1]), np.array([0.0, 0, 0]), 1 / 512, 1e-2, d(0.2), 5e-3),
        ("OrthogonalRotation", EIGHT, angle_axis_quat(90, [0, 0, 1]), np.array([1.0, 1, 1]), 1 / 512, 5e-3, d(0.25), 5e-3),
    ]
    axes = [[0, 0, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1], [0, 1, 1], [1, 1, 1]]
    angles = [7, 12, 15, 20, 11, 0, 5, 0]
    trans = [[1, 1, 1], [3, 2, 13], [4, 5, 11], [1, 2, 15], [3, 1.5, 18], [1, 7, 11], [0, 0, 0], [0, 0, 0]]
    rng = np.random.default_rng(59)
    for i in range(8):
        for n in (100, 500, 1000):
            pts = np.c_[rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(2, 10, n)]
            out.append((f"ManyPoints[{i}][{n}]", pts, angle_axis_quat(angles[i], axes[i]), np.array(trans[i], dtype=np.float64),
                        1 / 512, 1e-2, d(0.3), 5e-3))
    return out


def project(world, q, t, noise, seed):
    cam = world @ quat_to_rot(q).T + t
    feat = cam[:, :2] / cam[:, 2:3]
    if noise:
        feat = feat + np.random.default_rng(seed).normal(scale=noise, size=feat.shape)   # AddNoiseToProjection
    return np.ascontiguousarray(feat)


def check_solutions(name, world, feat, q_exp, t_exp, quats, ts, max_reproj, max_rot, max_trans):
    """dls_pnp_test.cc:95-125: every solution reprojects within max_reproj, and one matches the expected pose."""
    assert len(quats) > 0, name
    matched = False
    for q, t in zip(quats, ts):
        cam = world @ quat_to_rot(q).T + t
        rep = cam[:, :2] / cam[:, 2:3]
        assert (((rep - feat) ** 2).sum(axis=1) <= max_reproj).all(), (name, ((rep - feat) ** 2).sum(axis=1).max())
        dot = min(1.0, abs(float(np.dot(q / np.linalg.norm(q), q_exp))))
        ang = 2.0 * np.arccos(dot)                                    # Quaternion::angularDistance
        if ang < max_rot and float(((t_exp - t) ** 2).sum()) < max_trans:
            matched = True
    assert matched, name


def dls_cost_gradient(feat, world, s):
    """Independent of both implementations: numerical gradient of the DLS cost J'(s) = (1 + s.s)^2 sum |(I - n n^T)(C X + t)|^2
    with the optimal translation substituted (dls_pnp.cc:67-118), central differences."""
    n = np.c_[feat, np.ones(len(feat))]; n /= np.linalg.norm(n, axis=1, keepdims=True)
    P = np.eye(3)[None] - n[:, :, None] * n[:, None, :]

    def cost(sv):
        sx = np.array([[0, -sv[2], sv[1]], [sv[2], 0, -sv[0]], [-sv[1], sv[0], 0]])
        Cb = (1 - sv @ sv) * np.eye(3) - 2 * sx + 2 * np.outer(sv, sv)
        CX = world @ Cb.T
        A = P.sum(axis=0); b = -np.einsum("nij,nj->i", P, CX)
        t = np.linalg.solve(A, b)
        r = np.einsum("nij,nj->ni", P, CX + t)
        return float((r ** 2).sum())
    g = np.zeros(3); h = 1e-6
    for k in range(3):
        e = np.zeros(3); e[k] = h
        g[k] = (cost(s + e) - cost(s - e)) / (2 * h)
    return g, cost(s)
