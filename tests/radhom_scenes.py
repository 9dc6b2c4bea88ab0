"""Scenes for the radial-distortion homography tests, shaped like six_point_radial_distortion_homography_test.cc:62-228
and estimate_radial_distortion_homography_test.cc: points on a plane seen by two cameras related by (R, t), each with a
focal length and a one-parameter division-model distortion; features in pixels (principal point removed), normalised
features = pixels / focal length."""
import numpy as np


def distort(p3, f, l):
    """DistortPoint (six_point_radial_distortion_homography.cc:151-175)."""
    p = f * p3[:2] / p3[2]
    r2 = p @ p
    denom = 2.0 * l * r2
    inner = 1.0 - 4.0 * l * r2
    if abs(denom) < np.finfo(float).eps or inner < 0.0:
        return p
    return p * (1.0 - np.sqrt(inner)) / denom


def undistort(p2, f, l):
    """UndistortPoint (:177-191) -> bearing (x, y, 1)."""
    und = 1.0 / (1.0 + l * (p2 @ p2))
    return np.array([p2[0] * und / f, p2[1] * und / f, 1.0])


def symmetric_error(H, l1, l2, pl, pr, f1, f2):
    """CheckRadialSymmetricError (:201-239) in plain numpy (numpy's own inverse)."""
    l1s, l2s = l1 / (f1 * f1), l2 / (f2 * f2)
    bl, br = undistort(pl, f1, l1s), undistort(pr, f2, l2s)
    y = H @ br; y = y / y[2]
    z = np.linalg.inv(H) @ bl; z = z / z[2]
    dl = pl - distort(y, f1, l1s); dr = pr - distort(z, f2, l2s)
    return 0.5 * (dl @ dl + dr @ dr)


def rotation_z(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])


def rows(points_3d, R, t, f1, f2, k1, k2, noise=0.0, rng=None, lmin=-5.0, lmax=0.0):
    """THEIA_EST_RADIAL_HOMOGRAPHY data rows (N, 12) of the points seen by camera 1 (identity) and camera 2 (R, t)."""
    out = np.zeros((len(points_3d), 12))
    for i, X in enumerate(np.asarray(points_3d, dtype=np.float64)):
        p1 = distort(X, f1, k1)
        p2 = distort(R @ X + t, f2, k2)
        if noise:
            p1 = p1 + noise * rng.normal(size=2); p2 = p2 + noise * rng.normal(size=2)
        out[i] = [p1[0], p1[1], p2[0], p2[1], p1[0] / f1, p1[1] / f1, p2[0] / f2, p2[1] / f2, f1, f2, lmin, lmax]
    return out


REFERENCE_POINTS = np.array([[-1.0, 3.0, 1.0], [1.0, -1.0, 1.0], [-1.0, 1.0, 1.0], [2.0, 1.0, 1.0], [3.0, 1.0, 1.0], [2.0, 2.0, 1.0]])
