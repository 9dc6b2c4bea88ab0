"""Scenes for the P4Pf (uncalibrated absolute pose) tests, after the reference's four_point_focal_length_test.cc."""
import numpy as np

EST = 14   # THEIA_EST_UNCALIBRATED_ABSOLUTE_POSE


def euler_rotation(x, y, z):
    """Rz * Ry * Rx with the sign convention of four_point_focal_length_test.cc:124-128."""
    c, s = np.cos, np.sin
    Rz = np.array([[c(z), s(z), 0], [-s(z), c(z), 0], [0, 0, 1.0]])
    Ry = np.array([[c(y), 0, -s(y)], [0, 1.0, 0], [s(y), 0, c(y)]])
    Rx = np.array([[1.0, 0, 0], [0, c(x), s(x)], [0, -s(x), c(x)]])
    return Rz @ Ry @ Rx


def project(P, X):
    p = P @ np.concatenate([np.asarray(X, dtype=np.float64), np.ones((len(X), 1))], axis=1).T
    return (p[:2] / p[2]).T


def projection(focal, R, t):
    return np.diag([focal, focal, 1.0]) @ np.concatenate([R, np.asarray(t, dtype=np.float64).reshape(3, 1)], axis=1)


def basic_scene():
    """BasicTest (four_point_focal_length_test.cc:119-146): f = 800, the four world points and the pose it lists."""
    R = euler_rotation(-0.10, -0.20, 0.30)
    t = np.array([-0.00950692, 0.0171496, 0.0508743])
    X = np.array([[-1.0, 0.5, 1.2], [-0.79, -0.68, 1.9], [1.42, 1.01, 2.19], [0.87, -0.49, 0.89]])
    P = projection(800.0, R, t)
    return P, X, project(P, X)


def random_scene(rng, n=4, focal=None):
    """RandomTestWithNoise (:148-181): f in [600, 650], rotations in [-0.25, 0.25] per axis, baseline 0.25, depths in [0, 4]."""
    focal = rng.uniform(600.0, 650.0) if focal is None else focal
    while True:   # points in front of the camera: with the first point behind it the depth ratios describe the mirrored
        R = euler_rotation(*rng.uniform(-0.25, 0.25, 3))   # configuration, which no rigid motion aligns (solver and reference alike)
        t = rng.uniform(-1.0, 1.0, 3) * 0.25
        X = np.stack([2.0 * rng.uniform(-1, 1, n), 2.0 * rng.uniform(-1, 1, n), 2.0 * rng.uniform(-1, 1, n) + 2.0], axis=1)
        P = projection(focal, R, t)
        if (P @ np.concatenate([X, np.ones((n, 1))], axis=1).T)[2].min() > 0.25:
            break
    return P, X, project(P, X), focal


def ransac_scene(rng, n, outlier_fraction=0.3, noise=0.5, focal=None):
    """n correspondences [pixel (principal point removed) | world point] of one camera, a fraction replaced by outliers."""
    P, X, px, focal = random_scene(rng, n, focal)
    px = px + rng.normal(0.0, noise, px.shape)
    bad = rng.random(n) < outlier_fraction
    px[bad] = rng.uniform(-400.0, 400.0, (int(bad.sum()), 2))
    return np.ascontiguousarray(np.concatenate([px, X], axis=1)), P, focal, ~bad
