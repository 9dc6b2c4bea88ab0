"""The scenes of the reference's P4Pfr tests, restated as data and generators:
  sfm/pose/four_point_focal_length_radial_distortion_test.cc:198-265  BasicTest / PlanarTestWithNoise (focal length 1000,
      distortion -1e-7, the fixed four world points / the 2 x 2 square at depth 5 with its 1e-10 .. 1e-7 offsets)
  sfm/estimators/estimate_radial_dist_uncalibrated_absolute_pose_test.cc:84-283  20 random points in [-5, 5]^2 x [3, 10], the
      division-model distortion of the projections, uniform outlier features in [-1, 1]^2, the four modes
(both tests are commented out of the reference's CMakeLists.txt (:329, :370) and the solver test's own call is commented out
(:141-144); their scenes and tolerances are what is pinned here).  Noise: the reference's AddNoiseToProjection helper
(test/test_utils: a UNIFORM offset in [-noise, noise] per coordinate), drawn from numpy here."""
import numpy as np

FOCAL = 1000.0
DISTORTION = -1e-7


def distort(p, k):   # DistortPoint (estimate_radial_dist_uncalibrated_absolute_pose.cc:56-74)
    p = np.asarray(p, dtype=np.float64)
    r2 = float(p @ p)
    den = 2.0 * k * r2; inner = 1.0 - 4.0 * k * r2
    if abs(den) < 1e-15 or inner < 0.0:
        return p.copy()
    return p * ((1.0 - np.sqrt(inner)) / den)


def rot_zyx(x, y, z):   # the tests' Rz * Ry * Rx with their sign convention (:206-211)
    Rz = np.array([[np.cos(z), np.sin(z), 0], [-np.sin(z), np.cos(z), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(y), 0, -np.sin(y)], [0, 1, 0], [np.sin(y), 0, np.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(x), np.sin(x)], [0, -np.sin(x), np.cos(x)]])
    return Rz @ Ry @ Rx


def angle_axis(deg, axis):
    a = np.deg2rad(deg); ax = np.asarray(axis, dtype=np.float64); ax = ax / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def project(R, t, X, focal=FOCAL, k=DISTORTION):
    pc = np.diag([focal, focal, 1.0]) @ (R @ X + t)
    return distort(pc[:2] / pc[2], k)


def solver_scene(name, noise=0.0, rng=None):
    """(features (4, 2), world (4, 3), R, t) of BasicTest / PlanarTest."""
    if name == "basic":
        R = rot_zyx(-0.10, -0.20, 0.30); t = np.array([-0.00950692, 0.0171496, 0.0508743])
        W = np.array([-0.42941, 0.000621211, -0.350949, -1.45205, 0.415794, -0.556605, -1.92898, -1.89976, 1.4949, 0.838307, 1.41972,
                      1.25756]).reshape(3, 4).T        # Map<Matrix<double, 3, 4>> filled by <<: row by row, one point per column
    else:
        R = rot_zyx(-0.10, -0.20, 0.10); t = np.array([1.1, 0.2, 0.0])
        size, depth = 2.0, 5.0
        W = np.array([[-size / 2, -size / 2, depth + 1e-10], [size / 2, -size / 2, depth - 1e-10], [size / 2, size / 2, depth + 1e-8],
                      [-size / 2, size / 2, depth - 1e-7]])
    f = np.array([project(R, t, X) for X in W])
    if noise:
        f = f + rng.uniform(-noise, noise, f.shape)
    return f, W, R, t


SOLVER_LIMITS = (2000.0, 0.0, -1e-5, -1e-10)      # the (commented-out) call of the solver test (:141-144)
ESTIMATOR_LIMITS = (2000.0, 100.0, -1e-5, -1e-9)  # the estimator test's metadata (:133-137)

ROTATIONS_A = [np.eye(3), angle_axis(12.0, (0.0, 1.0, 0.0)), angle_axis(-9.0, (1.0, 0.2, -0.8))]     # AllInliers* (:169-173)
POSITIONS = [np.array([1.3, 0.0, 0.0]), np.array([1.0, 1.0, 0.1])]                                    # :174-175


def estimator_scene(rng, R, t, inlier_ratio, noise, n=20):
    """ExecuteRandomTest (:84-131): (N, 5) rows u v X Y Z."""
    rows = np.zeros((n, 5))
    for i in range(n):
        X = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(3, 10)])
        f = project(R, t, X) if i < inlier_ratio * n else rng.uniform(-1, 1, 2)
        rows[i] = [f[0], f[1], X[0], X[1], X[2]]
    if noise:
        rows[:, :2] += rng.uniform(-noise, noise, (n, 2))
    return rows


# (mode, inlier ratio, noise, pose tolerance, RansacParameters fields) of the four TESTs (:158-283); all with use_mle, error
# threshold 1, failure probability 0.001
MODES = [
    ("AllInliersNoNoise", 1.0, 0.0, 1e-4, dict(min_iterations=1)),
    ("AllInliersWithNoise", 1.0, 1.0, 1e-2, dict(min_iterations=1, max_iterations=1000)),
    ("OutliersNoNoise", 0.7, 0.0, 1e-2, dict(min_iterations=10, max_iterations=1000)),
    ("OutliersWithNoise", 0.7, 1.0, 0.1, dict(min_iterations=10, max_iterations=1000)),
]


def arrays_equal_up_to_scale(a, b, tol):   # test::ArraysEqualUpToScale (test/test_utils.h:76-87): |cos| of the angle >= 1 - tol
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return bool(abs(float(a @ b) / (np.linalg.norm(a) * np.linalg.norm(b))) >= 1.0 - tol)
