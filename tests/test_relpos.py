"""CPU: the oracle's OptimizeRelativePositionWithKnownRotation (optimize_relative_position_with_known_rotation.cc:116-191)
against known answers -- the true baseline direction on noise-free pairs, robustness to gross outliers, the sign rule -- and
against the committed vectors of tests/golden/relpos_irls.npz."""
import os

import numpy as np

from pytheiasfm_amd import synth
from tests import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


def make_pair(seed, n, noise=0.0, outliers=0.0, behind=False):
    """Two calibrated views of n points: x_i = R_i (X - c_i) / z.  Returns normalised correspondences, the angle-axis rotations
    and the true relative position R1 (c2 - c1) / |.|."""
    rng = np.random.RandomState(seed)
    w1 = 0.3 * rng.randn(3); w2 = 0.3 * rng.randn(3)
    R1 = synth.angle_axis_to_matrix(w1); R2 = synth.angle_axis_to_matrix(w2)
    c1 = 0.2 * rng.randn(3); c2 = c1 + rng.randn(3)
    X = rng.uniform(-2, 2, size=(n, 3)) + R1.T @ np.array([0.0, 0.0, 6.0]) + c1
    if behind:
        X = 2 * c1 - X
    p1 = (X - c1) @ R1.T; p2 = (X - c2) @ R2.T
    corr = np.concatenate([p1[:, :2] / p1[:, 2:3], p2[:, :2] / p2[:, 2:3]], axis=1)
    corr += noise * rng.randn(n, 4)
    nout = int(outliers * n)
    if nout:
        corr[rng.choice(n, nout, replace=False), 2:] = rng.uniform(-0.5, 0.5, size=(nout, 2))
    t = R1 @ (c2 - c1)
    return np.ascontiguousarray(corr), w1, w2, t / np.linalg.norm(t)


def make_pairs():
    sizes = [0, 1, 2, 5, 8, 31, 63, 64, 65, 100, 128, 200, 257, 500, 1000, 2000]
    pairs = []
    for k, n in enumerate(sizes):
        pairs.append(make_pair(100 + k, n, noise=0.0 if k % 3 == 0 else 1e-3, outliers=0.0 if k % 2 == 0 else 0.2))
    pairs.append(make_pair(300, 300, noise=5e-4, outliers=0.4))
    pairs.append(make_pair(301, 150, noise=0.0, behind=True))
    return pairs


def test_noise_free_pairs_recover_the_baseline_direction():
    for seed, n in ((1, 8), (2, 50), (3, 400)):
        corr, w1, w2, truth = make_pair(seed, n)
        pos, it = ol.optimize_relative_position(corr, w1, w2)
        assert abs(np.linalg.norm(pos) - 1.0) < 1e-12 and np.abs(pos - truth).max() < 1e-8, (seed, pos, truth)
        assert 10 <= it <= 100


def test_outliers_and_sign():
    corr, w1, w2, truth = make_pair(9, 400, noise=5e-4, outliers=0.2)
    pos, it = ol.optimize_relative_position(corr, w1, w2)
    assert pos @ truth > 0.9999         # the L1-type IRLS shrugs 20 % gross outliers off (it is a local method: not every geometry)
    # the scene mirrored behind both cameras: same epipolar constraints, the cheirality majority flips the sign
    corr, w1, w2, truth = make_pair(8, 200, behind=True)
    pos, _ = ol.optimize_relative_position(corr, w1, w2)
    assert np.abs(pos + truth).max() < 1e-8 or np.abs(pos - truth).max() < 1e-8
    # an empty pair: A = 0 -> U = I -> t = e_z, ten "converged" iterations, no point in front -> -e_z
    pos, it = ol.optimize_relative_position(np.zeros((0, 4)), np.zeros(3), np.zeros(3))
    assert it == 10 and np.array_equal(np.abs(pos), [0.0, 0.0, 1.0]) and pos[2] == -1.0


def test_golden_vectors():
    g = np.load(os.path.join(HERE, "golden", "relpos_irls.npz"))
    pairs = make_pairs()
    assert int(g["num"]) == len(pairs)
    for k, (corr, w1, w2, truth) in enumerate(pairs):
        assert np.array_equal(corr, g[f"corr{k}"]) and np.array_equal(np.concatenate([w1, w2]), g[f"rot{k}"])
        pos, it = ol.optimize_relative_position(corr, w1, w2)
        assert it == int(g[f"it{k}"]) and np.abs(pos - g[f"pos{k}"]).max() <= 1e-12, k
        posw, itw = ol.optimize_relative_position(corr, w1, w2, order=1)
        assert itw == int(g[f"itw{k}"]) and np.array_equal(posw, g[f"posw{k}"]), k


def test_independent_numpy_route_and_conditioning():
    """The same IRLS written with numpy (LAPACK SVD, numpy's pairwise sums): a third summation order and another SVD.  On
    noise-free pairs all agree to 1e-7; on noisy pairs the iteration amplifies the rounding differences between summation
    orders (1 / w weights of near-zero residuals), which bounds what "the same result" can mean for this function."""
    worst_clean, worst_noisy = 0.0, 0.0
    for k in range(24):
        noisy = k % 2 == 1
        corr, w1, w2, truth = make_pair(500 + k, 120 + 10 * k, noise=1e-3 if noisy else 0.0, outliers=0.1 if noisy else 0.0)
        R1 = synth.angle_axis_to_matrix(w1); R2 = synth.angle_axis_to_matrix(w2)
        n = len(corr)
        a = np.c_[corr[:, :2], np.ones(n)] @ R1; b = np.c_[corr[:, 2:], np.ones(n)] @ R2
        Cm = np.cross(b, a) @ R1.T
        w = np.ones(n); cost = 0.0; inner = 0; t = np.zeros(3)
        for _ in range(100):
            if inner >= 10:
                break
            w = np.maximum(w, 1e-7)
            U, _, _ = np.linalg.svd((Cm.T / w) @ Cm)
            t = U[:, 2]
            w = np.abs(Cm @ t); nc = w.sum()
            inner = inner + 1 if max(abs(cost - nc), 1 - t @ t) <= 1e-5 else 0
            cost = nc
        p0, _ = ol.optimize_relative_position(corr, w1, w2, order=0)
        p1, _ = ol.optimize_relative_position(corr, w1, w2, order=1)
        d = max(min(np.abs(p0 - t).max(), np.abs(p0 + t).max()), np.abs(p0 - p1).max())
        if noisy:
            worst_noisy = max(worst_noisy, d)
        else:
            worst_clean = max(worst_clean, d)
    assert worst_clean < 1e-6 and worst_noisy < 5e-3, (worst_clean, worst_noisy)


def reference_test_scene(seed, pixel_noise):
    """The scene of the reference's own test (optimize_relative_position_with_known_rotation_test.cc:53-117,121-141): 100 points
    with x in [-2, 2], y = -2 (its RandDouble(-2.0, -2.0)), z in [8, 10]; two cameras at random positions in [-1, 1]^3 (the second
    normalised) with rotations 0.2 x rand, focal length 800, principal point (500, 500); pixel noise on both projections; the
    features normalised by the calibration.  Returns (correspondences, rotation1, rotation2, true unit relative position)."""
    rng = np.random.RandomState(seed)
    pts = np.stack([rng.uniform(-2, 2, 100), np.full(100, -2.0), rng.uniform(8, 10, 100)], axis=1)
    c1 = rng.uniform(-1, 1, 3); w1 = 0.2 * rng.uniform(-1, 1, 3)
    c2 = rng.uniform(-1, 1, 3); c2 /= np.linalg.norm(c2); w2 = 0.2 * rng.uniform(-1, 1, 3)
    R1 = synth.angle_axis_to_matrix(w1); R2 = synth.angle_axis_to_matrix(w2)
    corr = np.zeros((100, 4))
    for k, (R, c) in enumerate(((R1, c1), (R2, c2))):
        p = (pts - c) @ R.T
        px = 800.0 * p[:, :2] / p[:, 2:3] + 500.0 + pixel_noise * rng.randn(100, 2)
        corr[:, 2 * k:2 * k + 2] = (px - 500.0) / 800.0
    t = R1 @ (c2 - c1)
    return corr, w1, w2, t / np.linalg.norm(t)


def _angle_deg(a, b):
    return np.degrees(np.arctan2(np.linalg.norm(np.cross(a, b)), a @ b))  # acos resolves 1.2e-6 degrees only


def test_the_reference_tests_scenes_and_tolerances():
    """NoNoise: 1e-6 degrees on every one of 40 scenes.  PixelNoise (1 px): the reference asserts 2 degrees on the ONE scene its
    fixed seed draws; the baseline of the distribution can be as short as a tenth of the depth, where 1 px moves the direction by
    more than that (and the planar scene has a second local minimum the IRLS can settle in), so over 40 scenes the bound is
    asserted on the median (and 5 degrees on nine scenes of ten).  (The TranslationNoise variants perturb a start value
    the function overwrites: :127-129.)"""
    errs = {0.0: [], 1.0: []}
    for noise in errs:
        for seed in range(40):
            corr, w1, w2, truth = reference_test_scene(5200 + seed, noise)
            pos, _ = ol.optimize_relative_position(corr, w1, w2)
            errs[noise].append(_angle_deg(pos, truth))
    assert max(errs[0.0]) < 1e-6, max(errs[0.0])
    assert np.median(errs[1.0]) < 2.0, np.median(errs[1.0])
    assert np.mean(np.array(errs[1.0]) < 5.0) >= 0.9
