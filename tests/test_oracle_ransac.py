"""CPU: pins the RANSAC oracle (oracle/ransac_oracle.cpp) against the real
libstdc++ sample stream, numpy linear algebra, and the reference's own
known-answer tests restated (five_point_relative_pose_test.cc:115-200,
estimate_relative_pose_test.cc:60-196, estimate_calibrated_absolute_pose_test.cc:60-215,
perspective_three_point_test.cc)."""
import json
import os

import numpy as np
import pytest

import ctypes as C

from pytheiasfm_amd import _capi as capi, synth
from tests import oracle_lib as ol, p4pf_scenes as ps

HERE = os.path.dirname(os.path.abspath(__file__))


def test_mt19937_randint_stream_matches_libstdcxx_golden():
    g = json.load(open(os.path.join(HERE, "golden", "mt19937_randint.json")))
    ri = np.array(g["randint"], dtype=np.int64)
    for seed in (52, 65, 66):
        m = ri[:, 0] == seed
        out = ol.randint_stream(seed, ri[m, 1], ri[m, 2])
        assert np.array_equal(out, ri[m, 3])
    for c in g["sampler"]:
        out = ol.sampler_stream(c["seed"], c["N"], c["m"], 64)
        assert np.array_equal(out.ravel(), np.array(c["samples"]))


@pytest.mark.parametrize("n", [3, 4, 7, 10])
def test_eigen_solver_against_numpy(n):
    rng = np.random.default_rng(n)
    for _ in range(20):
        A = rng.standard_normal((n, n))
        ok, wr, wi, V = ol.eig(A)
        assert ok
        w = np.linalg.eigvals(A)
        assert np.abs(np.sort_complex(wr + 1j * wi) - np.sort_complex(w)).max() < 1e-10
        for i in range(n):
            if wi[i] == 0:
                v = V[:, i]
                assert np.abs(A @ v - wr[i] * v).max() < 1e-9 * max(1.0, np.abs(v).max())


def test_jacobi_svd_against_numpy():
    rng = np.random.default_rng(5)
    mats = [rng.standard_normal((3, 3)) for _ in range(30)]
    t = rng.standard_normal(3); R = synth.angle_axis_to_matrix(rng.standard_normal(3))
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    mats += [tx @ R, np.zeros((3, 3)), np.eye(3), np.diag([3.0, 0.0, -2.0])]
    for A in mats:
        U, S, V = ol.svd3(A)
        assert np.abs(U @ np.diag(S) @ V.T - A).max() < 1e-13 * max(1.0, np.abs(A).max())
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-13 and np.abs(V.T @ V - np.eye(3)).max() < 1e-13
        assert np.all(S[:-1] >= S[1:]) and np.all(S >= 0)
        assert np.abs(S - np.linalg.svd(A, compute_uv=False)).max() < 1e-13 * max(1.0, S.max())


def test_polynomial_roots_companion_matrix():
    assert np.allclose(np.sort(ol.poly_roots([1, -10, 35, -50, 24])), [1, 2, 3, 4], atol=1e-12)
    assert np.allclose(np.sort(ol.poly_roots([0, 0, 2, -6, 4])), [1, 2], atol=1e-14)  # leading zeros removed
    # complex roots: the REAL PARTS come back (reference quirk)
    assert np.allclose(np.sort(ol.poly_roots([1, 0, 0, 0, 1])), np.sort(np.roots([1, 0, 0, 0, 1]).real), atol=1e-12)


def rot(axis, deg):
    axis = np.asarray(axis, dtype=np.float64)
    return synth.angle_axis_to_matrix(axis / np.linalg.norm(axis) * np.deg2rad(deg))


def cross_mat(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def sampson(E, x, y):
    xh = np.append(x, 1.0); yh = np.append(y, 1.0)
    ex = E @ xh
    den = (yh @ E[:, 0]) ** 2 + (yh @ E[:, 1]) ** 2 + ex[0] ** 2 + ex[1] ** 2
    return (yh @ ex) ** 2 / den


FIVE_POINT_CASES = [
    # (points, R, t, noise, tolerance)  -- five_point_relative_pose_test.cc:115-183
    ([(-1, 3, 3), (1, -1, 2), (3, 1, 2.5), (-1, 1, 2), (2, 1, 3)], rot((0, 0, 1), 13.0), (1, 1, 1), 0.0, 1e-4),
    ([(-1, 3, 3), (1, -1, 2), (3, 1, 2.5), (-1, 1, 2), (2, 1, 3)], rot((0, 0, 1), 13.0), (1, 1, 1), 1.0 / 512, 1e-2),
    ([(-1, 3, 3), (1, -1, 2), (3, 1, 2.0), (-1, 1, 2), (2, 1, 3)], rot((0, 0, 1), 13.0), (0, 0, 1), 1.0 / 512, 0.15),
    ([(-1, 3, 3), (1, -1, 2), (3, 1, 2.0), (-1, 1, 2), (2, 1, 3)], np.eye(3), (1, 1, 1), 1.0 / 512, 0.01),
]


@pytest.mark.parametrize("case", range(len(FIVE_POINT_CASES)))
def test_five_point_known_answers(case):
    pts, R, t, noise, tol = FIVE_POINT_CASES[case]
    pts = np.array(pts, dtype=np.float64); t = np.array(t, dtype=np.float64)
    x1 = pts[:, :2] / pts[:, 2:]
    p2 = pts @ R.T + t
    x2 = p2[:, :2] / p2[:, 2:]
    st = synth.Stream(67, case)
    if noise:
        x1 = x1 + noise * np.stack([st.normal(np.arange(5) * 4), st.normal(np.arange(5) * 4 + 1)], 1)
        x2 = x2 + noise * np.stack([st.normal(np.arange(5) * 4 + 2), st.normal(np.arange(5) * 4 + 3)], 1)
    E = ol.five_point(np.hstack([x1, x2]))
    assert len(E) > 0
    Egt = cross_mat(t) @ R
    matched = False
    for e in E:
        for i in range(5):
            assert sampson(e, x1[i], x2[i]) < 1e-8
        cosd = abs(np.sum(e * Egt)) / (np.linalg.norm(e) * np.linalg.norm(Egt))
        matched |= cosd >= 1.0 - tol
    assert matched


def test_five_point_degenerate_input_returns_nothing():
    corr = np.zeros((5, 4))  # rank-deficient epipolar matrix: kernel dimension != 4
    assert len(ol.five_point(corr)) == 0


def test_p3p_known_answer_and_collinear():
    rng = np.random.default_rng(3)
    for _ in range(20):
        X = rng.uniform(-2, 2, (3, 3)) + np.array([0, 0, 6.0])
        R = synth.angle_axis_to_matrix(rng.standard_normal(3) * 0.3); t = rng.standard_normal(3) * 0.3
        pc = X @ R.T + t
        uv = pc[:, :2] / pc[:, 2:]
        Rs, ts = ol.p3p(np.hstack([uv, X]))
        assert len(Rs) == 4  # real parts of all four roots are used (quirk)
        err = min(np.abs(r - R).max() + np.abs(tt - t).max() for r, tt in zip(Rs, ts) if np.all(np.isfinite(r)))
        assert err < 1e-8
    X = np.array([[0, 0, 5.0], [1, 1, 6.0], [2, 2, 7.0]])
    assert len(ol.p3p(np.hstack([X[:, :2] / X[:, 2:], X]))[0]) == 0  # collinear world points


def grid_points():
    return np.array([(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (4, 5, 6)], dtype=np.float64)


REL_ROT = [rot((0, 1, 0), 12.0), rot((1.0, 0.2, -0.8), -9.0)]
REL_POS = [np.array([-0.7, 0, 0]), np.array([0, 0.1, 0.5])]


@pytest.mark.parametrize("ri", range(2))
@pytest.mark.parametrize("pj", range(2))
@pytest.mark.parametrize("mode", ["clean", "noise", "outliers"])
def test_estimate_relative_pose_reference_scenes(ri, pj, mode):
    """estimate_relative_pose_test.cc ExecuteRandomTest: 27 grid points."""
    R, position = REL_ROT[ri], REL_POS[pj] * (1.0 if mode == "clean" else 1.3 / 0.7 if pj == 0 else 1.0)
    pts = grid_points()
    t = -R @ position; t = t / np.linalg.norm(t)
    inlier_ratio = 0.7 if mode == "outliers" else 1.0
    st = synth.Stream(65, 10 * ri + pj)
    x1 = pts[:, :2] / pts[:, 2:]
    p2 = pts @ R.T + t
    x2 = p2[:, :2] / p2[:, 2:]
    out = np.arange(27) >= inlier_ratio * 27
    x1[out] = 2 * np.stack([st.uniform(4 * np.arange(27)), st.uniform(4 * np.arange(27) + 1)], 1)[out] - 1
    x2[out] = 2 * np.stack([st.uniform(4 * np.arange(27) + 2), st.uniform(4 * np.arange(27) + 3)], 1)[out] - 1
    if mode == "noise":
        x1 = x1 + 1e-3 * np.stack([st.normal(4 * np.arange(27) + 500), st.normal(4 * np.arange(27) + 501)], 1)
        x2 = x2 + 1e-3 * np.stack([st.normal(4 * np.arange(27) + 502), st.normal(4 * np.arange(27) + 503)], 1)
    prm = ol.default_ransac_params((2.0 / 1000.0) ** 2, seed=65)
    prm.use_mle = 1; prm.failure_probability = 0.0001
    r = ol.ransac_estimate(0, np.hstack([x1, x2]), prm)
    assert r["success"] and r["num_inliers"] > 5
    Rm = r["model"][9:18].reshape(3, 3); pos = r["model"][18:21]
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ Rm.T) - 1) / 2, -1, 1)))
    tdiff = np.degrees(np.arccos(np.clip(position / np.linalg.norm(position) @ pos, -1, 1)))
    tol = 1e-4 if mode == "clean" else 5.0
    assert ang < tol and tdiff < tol


ABS_ROT = [np.eye(3), rot((0, 1, 0), 12.0), rot((1.0, 0.2, -0.8), -9.0)]
ABS_POS = [np.array([-1.3, 0, 0]), np.array([0, 0, 0.5])]


@pytest.mark.parametrize("ri", range(3))
@pytest.mark.parametrize("pj", range(2))
@pytest.mark.parametrize("mode", ["clean", "noise", "outliers"])
def test_estimate_calibrated_absolute_pose_reference_scenes(ri, pj, mode):
    """estimate_calibrated_absolute_pose_test.cc ExecuteRandomTest (KNEIP), 100 points."""
    R, position = ABS_ROT[ri], ABS_POS[pj]
    st = synth.Stream(66, 10 * ri + pj)
    i = np.arange(100)
    X = np.stack([4 * st.uniform(3 * i) - 2, 4 * st.uniform(3 * i + 1) - 2, 6 + 4 * st.uniform(3 * i + 2)], 1)
    pc = (X - position) @ R.T
    uv = pc[:, :2] / pc[:, 2:]
    if mode == "outliers":
        out = i >= 70
        uv[out] = 2 * np.stack([st.uniform(2 * i + 900), st.uniform(2 * i + 901)], 1)[out] - 1
    if mode == "noise":
        uv = uv + 1e-3 * np.stack([st.normal(2 * i + 700), st.normal(2 * i + 701)], 1)
    prm = ol.default_ransac_params((4.0 / 1000.0) ** 2, seed=66)
    prm.use_mle = 1; prm.failure_probability = 0.001; prm.min_iterations = 50
    r = ol.ransac_estimate(2, np.hstack([uv, X]), prm)
    assert r["success"] and r["num_inliers"] > 3
    Rm = r["model"][0:9].reshape(3, 3); pos = r["model"][9:12]
    tol = 1e-4 if mode != "noise" else 1e-2
    cos_r = abs(np.sum(R * Rm)) / (np.linalg.norm(R) * np.linalg.norm(Rm))
    cos_p = abs(position @ pos) / (np.linalg.norm(position) * np.linalg.norm(pos))
    assert cos_r >= 1 - tol and cos_p >= 1 - tol


def test_iteration_bound_rules():
    """ComputeMaxIterations (sample_consensus_estimator.h:252-297): min/max
    clamps, fixed iteration count when min == max, and shrinking with the inlier ratio."""
    data, offsets, truth = synth.synth_ransac_v1(2, 300, "relative", seed=9, inlier_lo=0.7, inlier_hi=0.9)
    prm = ol.default_ransac_params((2 / 1000.0) ** 2, seed=1)
    r = ol.ransac_estimate(0, data[:300], prm)
    assert r["num_iterations"] == 100  # high inlier ratio -> min_iterations
    prm.min_iterations = 37; prm.max_iterations = 37
    assert ol.ransac_estimate(0, data[:300], prm)["num_iterations"] == 37
    prm.min_iterations = 5; prm.max_iterations = 10 ** 6
    low, _, _ = synth.synth_ransac_v1(1, 300, "relative", seed=10, inlier_lo=0.35, inlier_hi=0.35)
    r = ol.ransac_estimate(0, low, prm)
    assert 5 < r["num_iterations"] < 5000
    conf = 1.0 - (1.0 - (r["num_inliers"] / 300.0) ** 5.0) ** r["num_iterations"]
    assert abs(conf - r["confidence"]) < 1e-12


def test_golden_ransac_fixture():
    g = np.load(os.path.join(HERE, "golden", "ransac_small.npz"))
    assert np.array_equal(ol.sampler_stream(65, 120, 5, 64), g["sampler_65_120_5"])
    for kind, est, thr, seed in (("relative", 0, (2 / 1000.0) ** 2, 65), ("absolute", 2, (4 / 1000.0) ** 2, 66)):
        data, offsets = g[f"{kind}_data"], g[f"{kind}_offsets"]
        for use_mle in (0, 1):
            for i in range(3):
                prm = ol.default_ransac_params(thr, seed + i); prm.use_mle = use_mle
                r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm, trace_capacity=4096)
                assert np.array_equal(r["inlier_mask"], g[f"{kind}_mle{use_mle}_masks"][i])
                assert r["num_iterations"] == g[f"{kind}_mle{use_mle}_iters"][i]
                ml = 21 if est == 0 else 12
                assert np.allclose(r["model"][:ml], g[f"{kind}_mle{use_mle}_models"][i], rtol=0, atol=1e-12, equal_nan=True)
                if i == 0:
                    assert np.array_equal(r["trace"][0], g[f"{kind}_mle{use_mle}_trace_iter"])
                    assert np.array_equal(r["trace"][2], g[f"{kind}_mle{use_mle}_trace_ninl"])


def test_prosac_sampler_properties():
    """ProsacSampler: early samples come from the top of the quality-sorted data,
    indices are unique and in range; PROSAC finds the model with sorted data."""
    data, offsets, truth = synth.synth_ransac_v1(1, 300, "relative", seed=77)
    order = np.argsort(~truth["inlier"][0], kind="stable")
    data = data[order]
    prm = ol.default_ransac_params((2 / 1000.0) ** 2, seed=3); prm.ransac_type = 1
    r = ol.ransac_estimate(0, data, prm, trace_capacity=4096)
    assert r["success"] and r["num_inliers"] > 0.8 * truth["inlier"][0].sum()
    # first scored models already have many inliers (samples drawn from the best data)
    assert r["trace"][2][:5].max() > 0.5 * truth["inlier"][0].sum()


# ------------------------------------------------------------------ SQPnP
def quat_to_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_svd9_and_quaternion_round_trip_against_numpy():
    st = synth.Stream(91, 0)
    B = (2 * st.uniform(np.arange(81)) - 1).reshape(9, 9)
    for A in (B, B @ B.T, (B[:, :6] @ B[:, :6].T)):       # general, SPD, rank-6 PSD (the SQPnP Omega shape)
        U, S, V = ol.svd9(A)
        assert np.all(np.diff(S) <= 1e-12) and np.all(S >= 0)
        assert np.abs(U @ np.diag(S) @ V.T - A).max() <= 1e-12 * max(1.0, np.abs(A).max())
        assert np.abs(U.T @ U - np.eye(9)).max() <= 1e-13 and np.abs(V.T @ V - np.eye(9)).max() <= 1e-13
        assert np.abs(S - np.linalg.svd(A, compute_uv=False)).max() <= 1e-12 * S[0]
    for axis, deg in (((0, 0, 1), 13.0), ((1, 1, 1), 170.0), ((1, 0, 0), 180.0), ((0.2, -1, 0.3), -95.0)):
        R = rot(axis, deg)
        q, R2 = ol.rot_quat_roundtrip(R)
        assert abs(np.linalg.norm(q) - 1) <= 1e-14 and np.abs(R2 - R).max() <= 1e-14


SQP_POINTS = np.array([[-1.0, 3.0, 3.0], [1.0, -1.0, 2.0], [-1.0, 1.0, 2.0], [2.0, 1.0, 3.0],
                       [-1.0, -3.0, 2.0], [1.0, -2.0, 1.0], [-1.0, 4.0, 2.0], [-2.0, 2.0, 3.0]])


def _check_sqpnp(X, R, t, noise, max_reproj, max_rot_deg, max_trans_sq, seed=59):
    pc = X @ R.T + t
    uv = pc[:, :2] / pc[:, 2:]
    if noise:
        st = synth.Stream(seed, 1)
        i = np.arange(len(X))
        uv = uv + noise * np.stack([st.normal(2 * i), st.normal(2 * i + 1)], 1)
    q, ts = ol.sqpnp(uv, X)
    assert len(q) > 0
    matched = False
    for qi, ti in zip(q, ts):
        Rs = quat_to_matrix(qi)
        pr = X @ Rs.T + ti
        assert np.all(np.sum((pr[:, :2] / pr[:, 2:] - uv) ** 2, axis=1) <= max_reproj)
        ang = np.degrees(np.arccos(np.clip((np.trace(R.T @ Rs) - 1) / 2, -1, 1)))
        if ang < max_rot_deg and np.sum((t - ti) ** 2) < max_trans_sq:
            matched = True
    assert matched


def test_sqpnp_known_answers():
    """sqpnp_test.cc: Basic (:130-152), NoiseTest (:156-181), ManyPoints-style sweeps (:183-260)."""
    R = rot((0, 0, 1), 13.0); t = np.array([1.0, 1.0, 1.0])
    _check_sqpnp(SQP_POINTS[:4], R, t, 0.0, 1e-4, 1.0, 1e-2)
    _check_sqpnp(SQP_POINTS, R, t, 1.0 / 512.0, 5e-3, 0.25, 1e-2)
    axes = [(0, 0, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    angles = [7.0, 12.0, 15.0, 20.0, 11.0, 0.0]
    trans = [(1, 1, 1), (3, 2, 13), (4, 5, 11), (1, 2, 15), (3, 1.5, 18), (0, 0, 0)]
    st = synth.Stream(59, 7)
    for k, (ax, an, tr) in enumerate(zip(axes, angles, trans)):
        for npts in (100, 1000):
            i = np.arange(npts)
            X = np.stack([10 * st.uniform(3 * i + 7 * k) - 5, 10 * st.uniform(3 * i + 1 + 7 * k) - 5, 2 + 8 * st.uniform(3 * i + 2 + 7 * k)], 1)
            _check_sqpnp(X, rot(ax, an), np.array(tr, float), 1.0 / 512.0, 1.0, 0.3, 5e-2)


def test_sqpnp_minimal_three_points_models():
    """What the RANSAC estimator feeds it (SampleSize() = 3).  With three points Omega has a
    3-dimensional null space and RunSQP performs ONE iteration (sqpnp.cc:33-34), so the
    solutions only approximately reproject the sample: reference behaviour, kept.  The
    estimator's models are the quaternion round trip of those solutions."""
    st = synth.Stream(93, 0)
    for trial in range(40):
        i = np.arange(3)
        X = np.stack([4 * st.uniform(9 * trial + 3 * i) - 2, 4 * st.uniform(9 * trial + 3 * i + 1) - 2, 6 + 4 * st.uniform(9 * trial + 3 * i + 2)], 1)
        R = rot((0.3, 1.0, -0.2), 10.0 + trial); t = np.array([0.2, -0.1, 0.5])
        pc = X @ R.T + t
        uv = pc[:, :2] / pc[:, 2:]
        q, ts = ol.sqpnp(uv, X)
        m = ol.estimate_models(4, np.hstack([uv, X]))
        assert len(m) == len(q)
        for qi, ti, mi in zip(q, ts, m):
            Rs = quat_to_matrix(qi)
            pr = X @ Rs.T + ti
            assert np.abs(pr[:, :2] / pr[:, 2:] - uv).max() <= 5e-2
            assert abs(np.linalg.det(Rs) - 1) <= 1e-9 and np.abs(Rs @ Rs.T - np.eye(3)).max() <= 1e-9
            assert np.abs(mi[:9].reshape(3, 3) - Rs).max() <= 1e-14 and np.abs(mi[9:12] + Rs.T @ ti).max() <= 1e-12


@pytest.mark.parametrize("mode", ["clean", "noise", "outliers"])
def test_estimate_calibrated_absolute_pose_sqpnp(mode):
    """estimate_calibrated_absolute_pose_test.cc ExecuteRandomTest with PnPType::SQPnP."""
    R, position = ABS_ROT[1], ABS_POS[0]
    st = synth.Stream(66, 77)
    i = np.arange(100)
    X = np.stack([4 * st.uniform(3 * i) - 2, 4 * st.uniform(3 * i + 1) - 2, 6 + 4 * st.uniform(3 * i + 2)], 1)
    pc = (X - position) @ R.T
    uv = pc[:, :2] / pc[:, 2:]
    if mode == "outliers":
        out = i >= 70
        uv[out] = 2 * np.stack([st.uniform(2 * i + 900), st.uniform(2 * i + 901)], 1)[out] - 1
    if mode == "noise":
        uv = uv + 1e-3 * np.stack([st.normal(2 * i + 700), st.normal(2 * i + 701)], 1)
    prm = ol.default_ransac_params((4.0 / 1000.0) ** 2, seed=66)
    prm.use_mle = 1; prm.failure_probability = 0.001; prm.min_iterations = 50
    r = ol.ransac_estimate(4, np.hstack([uv, X]), prm)
    assert r["success"] and r["num_inliers"] > 3
    Rm = r["model"][0:9].reshape(3, 3); pos = r["model"][9:12]
    tol = 1e-2   # kPoseTolerance of the SQPnP cases (:217-248, :340-368); the minimal SQPnP models are approximate
    cos_r = abs(np.sum(R * Rm)) / (np.linalg.norm(R) * np.linalg.norm(Rm))
    cos_p = abs(position @ pos) / (np.linalg.norm(position) * np.linalg.norm(pos))
    assert cos_r >= 1 - tol and cos_p >= 1 - tol


# ----------------------------------------------------------- LO-RANSAC (absolute pose)
def test_rotation_matrix_angle_axis_round_trip():
    """ceres RotationMatrixToAngleAxis / AngleAxisToRotationMatrix as restated for the LO refinement."""
    lib = ol.rlib()
    for axis, deg in (((0, 0, 1), 13.0), ((1, 1, 1), 170.0), ((1, 0, 0), 179.999), ((0.2, -1, 0.3), -95.0), ((0, 1, 0), 1e-7)):
        R = rot(axis, deg)
        aa = np.zeros(3); R2 = np.zeros((3, 3))
        lib.oracle_rot_angle_axis_roundtrip(capi.ptr(np.ascontiguousarray(R), C.c_double), capi.ptr(aa, C.c_double), capi.ptr(R2, C.c_double))
        assert np.abs(R2 - R).max() <= 1e-12
        assert abs(np.linalg.norm(aa) - np.radians(abs(deg))) <= 1e-9


@pytest.mark.parametrize("pj", range(2))
def test_estimate_calibrated_absolute_pose_lo(pj):
    """estimate_calibrated_absolute_pose_test.cc OutliersWithNoiseKNEIP_LO (:370-400): 30 % outliers, 1 px noise,
    use_lo with lo_start_iterations = 5; the refined pose is accurate to kPoseTolerance-like bounds."""
    R = [np.eye(3), rot((0.4, -0.3, 0.85), 7.0)][pj]; position = [np.array([1.0, 0, 0]), np.array([0, 1.0, 0])][pj]
    st = synth.Stream(66, 300 + pj)
    i = np.arange(100)
    X = np.stack([4 * st.uniform(3 * i) - 2, 4 * st.uniform(3 * i + 1) - 2, 6 + 4 * st.uniform(3 * i + 2)], 1)
    pc = (X - position) @ R.T
    uv = pc[:, :2] / pc[:, 2:]
    out = i >= 70
    uv[out] = 2 * np.stack([st.uniform(2 * i + 900), st.uniform(2 * i + 901)], 1)[out] - 1
    uv = uv + 1e-3 * np.stack([st.normal(2 * i + 700), st.normal(2 * i + 701)], 1)
    data = np.hstack([uv, X])
    prm = ol.default_ransac_params((4.0 / 1000.0) ** 2, seed=66)
    prm.use_mle = 1; prm.failure_probability = 0.001; prm.min_iterations = 50
    r0 = ol.ransac_estimate(2, data, prm)
    prm.use_lo = 1; prm.lo_start_iterations = 5
    r1 = ol.ransac_estimate(2, data, prm)
    nlo = ol.rlib().oracle_last_lo_iterations()
    assert r1["success"] and nlo >= 1 and r1["num_inliers"] >= 60
    def err(r):
        Rm = r["model"][0:9].reshape(3, 3); pos = r["model"][9:12]
        return np.degrees(np.arccos(np.clip((np.trace(R.T @ Rm) - 1) / 2, -1, 1))), np.linalg.norm(pos - position)
    e0, e1 = err(r0), err(r1)
    # two Huber iterations whose width is 1.5 x the SQUARED threshold (reference quirk): a modest refinement
    assert e1[0] < 0.3 and e1[1] < 0.05
    assert e1[1] <= e0[1] + 1e-2                   # never meaningfully worse than the minimal-sample pose
    Rm = r1["model"][0:9].reshape(3, 3)
    assert np.abs(Rm @ Rm.T - np.eye(3)).max() <= 1e-12


def test_lmed_quality_measurement_rules():
    """lmed_quality_measurement.h: even n -> upper middle element, odd n -> mean of the two around it; threshold rule."""
    data, offsets, truth = synth.synth_ransac_v1(1, 301, "absolute", seed=12, inlier_lo=0.75, inlier_hi=0.75)
    for n in (301, 300):
        d = data[:n]
        prm = ol.default_ransac_params((4 / 1000.0) ** 2, seed=9); prm.ransac_type = 2; prm.min_iterations = 150; prm.max_iterations = 150
        r = ol.ransac_estimate(2, d, prm)
        assert r["success"] and r["num_iterations"] == 150
        m = r["model"]
        res = np.array([ol.rlib().oracle_model_error(2, capi.ptr(m, C.c_double), capi.ptr(np.ascontiguousarray(d[i]), C.c_double)) for i in range(n)])
        sq = np.sort(res * res)
        med = sq[n // 2] if n % 2 == 0 else 0.5 * (sq[n // 2 - 1] + sq[n // 2])
        thr = 2.5 * 1.4826 * (1 + 5.0 / (n - 3)) * np.sqrt(med)
        assert np.array_equal(r["inlier_mask"].astype(bool), res * res < thr * thr)
        assert r["inlier_mask"][truth["inlier"][0][:n]].mean() > 0.8


def test_golden_ransac_variants():
    """The oracle reproduces tests/golden/ransac_variants.npz: SQPnP solutions, PROSAC / LMED / SQPnP / LO-RANSAC runs."""
    g = np.load(os.path.join(HERE, "golden", "ransac_variants.npz"))
    for k in range(4):
        q, t = ol.sqpnp(g[f"sqpnp{k}_uv"], g[f"sqpnp{k}_X"])
        assert np.array_equal(q, g[f"sqpnp{k}_q"]) and np.array_equal(t, g[f"sqpnp{k}_t"])
    data, offsets = g["abs_data"], g["abs_offsets"]
    for name, est, setup in (("prosac", 2, dict(ransac_type=1)), ("lmed", 2, dict(ransac_type=2, min_iterations=120, max_iterations=200)),
                             ("sqpnp", 4, dict()), ("lo", 2, dict(use_lo=1, lo_start_iterations=5, min_iterations=50, use_mle=1))):
        for i in range(3):
            prm = ol.default_ransac_params((4 / 1000.0) ** 2, 66 + i)
            for kk, vv in setup.items():
                setattr(prm, kk, vv)
            r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm)
            assert r["num_iterations"] == g[f"{name}_iters"][i] and np.array_equal(r["inlier_mask"], g[f"{name}_masks"][i])
            assert np.allclose(r["model"][:12], g[f"{name}_models"][i], rtol=0, atol=1e-13, equal_nan=True)


# ---- uncalibrated / plane / known-orientation estimators (R14) ----

EIGHT_PTS = [(-1, 3, 3), (1, -1, 2), (-1, 1, 2), (2, 1, 3), (-1, -3, 2), (1, -2, 1), (-1, 4, 2), (-2, 2, 3)]


def _two_view(points, R, t):
    P = np.array(points, dtype=np.float64)
    x1 = P[:, :2] / P[:, 2:]
    P2 = P @ R.T + np.asarray(t, dtype=np.float64)
    return x1, P2[:, :2] / P2[:, 2:]


def test_eight_point_known_answers():
    """eight_point_fundamental_matrix_test.cc:148-250: exact data -> the epipolar constraint holds
    and F is the essential matrix up to scale; a repeated point has no solution."""
    R = rot((0, 0, 1), 13.0)
    for t in [(1.0, 0.5, 1.5), (1.0, 0.5, 0.0)]:
        x1, x2 = _two_view(EIGHT_PTS, R, t)
        m = ol.estimate_models(5, np.hstack([x1, x2]))
        assert len(m) == 1
        F = m[0][:9].reshape(3, 3)
        assert np.all(m[0][9:] == 0)
        assert max(sampson(F, x1[i], x2[i]) for i in range(8)) < 1e-20
        assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-14
        E = cross_mat(t) @ R
        c = np.sum(F * E) / (np.linalg.norm(F) * np.linalg.norm(E))
        assert abs(abs(c) - 1.0) < 1e-10
    pts = list(EIGHT_PTS); pts[1] = pts[0]
    x1, x2 = _two_view(pts, R, (1.0, 0.5, 0.0))
    assert len(ol.estimate_models(5, np.hstack([x1, x2]))) == 0


def test_eight_point_pixel_coordinates_are_normalised():
    """The Hartley normalisation (pose/util.cc:81-112) keeps the solve well conditioned in pixels."""
    R = rot((0.2, 1, 0.1), 9.0); t = (0.8, -0.1, 0.3)
    x1, x2 = _two_view(EIGHT_PTS, R, t)
    K = np.array([[900.0, 0, 640], [0, 900.0, 360], [0, 0, 1]])
    p1 = x1 * 900.0 + K[:2, 2]; p2 = x2 * 900.0 + K[:2, 2]
    F = ol.estimate_models(5, np.hstack([p1, p2]))[0][:9].reshape(3, 3)
    assert max(sampson(F, p1[i], p2[i]) for i in range(8)) < 1e-16
    Fe = np.linalg.inv(K).T @ cross_mat(t) @ R @ np.linalg.inv(K)
    c = np.sum(F * Fe) / (np.linalg.norm(F) * np.linalg.norm(Fe))
    assert abs(abs(c) - 1.0) < 1e-8


def test_four_point_homography_known_answers():
    """four_point_homography_test.cc:128-205: coplanar points, H = R + t n^T / d up to scale."""
    R = rot((0, 0, 1), 13.0); t = np.array([1.0, 0.5, 1.5])
    n = np.array([0.1, -0.2, 1.0]); n /= np.linalg.norm(n); d = 6.0
    xy = np.array([(-0.4, 0.3), (0.35, -0.25), (0.3, 0.4), (-0.2, -0.45)])
    depth = d / (xy @ n[:2] + n[2])
    pts = np.column_stack([xy * depth[:, None], depth])
    x1, x2 = _two_view(pts, R, t)
    m = ol.estimate_models(6, np.hstack([x1, x2]))
    assert len(m) == 1
    H = m[0][:9].reshape(3, 3)
    Ht = R + np.outer(t, n) / d
    c = np.sum(H * Ht) / (np.linalg.norm(H) * np.linalg.norm(Ht))
    assert abs(abs(c) - 1.0) < 1e-10
    for i in range(4):
        q = H @ np.append(x1[i], 1.0)
        assert np.linalg.norm(q[:2] / q[2] - x2[i]) < 1e-12


def test_plane_and_known_orientation_minimal_solvers():
    """estimate_dominant_plane_from_points.cc:62-81 and
    relative_pose_from_two_points_with_known_rotation.cc:51-88."""
    p = np.array([[0.0, 0, 1], [1, 0, 1.5], [0, 2, 0.5]])
    m = ol.estimate_models(7, p)
    assert len(m) == 1
    nrm = np.cross(p[1] - p[0], p[2] - p[0]); nrm /= np.linalg.norm(nrm)
    assert np.allclose(m[0][:3], p[0]) and np.allclose(m[0][3:6], nrm, atol=1e-15)
    # collinear (|cross|^2 < 1e-6) -> no model
    assert len(ol.estimate_models(7, np.array([[0.0, 0, 0], [1, 1, 1], [2, 2, 2.0000001]]))) == 0
    # two correspondences under pure translation: position = -t / |t|
    t = np.array([0.3, -0.2, 0.9])
    x1, x2 = _two_view([(-1, 3, 3), (1, -1, 2)], np.eye(3), t)
    m = ol.estimate_models(8, np.hstack([x1, x2]))
    assert len(m) == 1
    pos = m[0][:3]
    assert abs(np.linalg.norm(pos) - 1.0) < 1e-15
    assert abs(abs(pos @ (t / np.linalg.norm(t))) - 1.0) < 1e-12


@pytest.mark.parametrize("kind,est,thresh", [("fundamental", 5, 4.0), ("homography", 6, 16.0),
                                             ("plane", 7, 0.004), ("known_orientation", 8, (2.0 / 1000.0) ** 2)])
def test_new_estimators_recover_the_synthetic_inliers(kind, est, thresh):
    data, offsets, truth = synth.synth_ransac_v1(2, 300, kind=kind, seed=0x5AC51400, inlier_lo=0.45, inlier_hi=0.7)
    for p in range(2):
        d = data[offsets[p]:offsets[p + 1]]
        prm = ol.default_ransac_params(thresh, seed=7 + p)
        prm.failure_probability = 0.001
        r = ol.ransac_estimate(est, d, prm)
        assert r["success"]
        tin = truth["inlier"][p]
        mask = r["inlier_mask"].astype(bool)
        # nearly all true inliers are found and few outliers slip in
        assert (mask & tin).sum() >= 0.75 * tin.sum()
        assert (mask & ~tin).sum() <= 0.1 * tin.sum() + 3
        if kind == "plane":
            nrm = r["model"][3:6]
            assert abs(abs(nrm @ truth["plane_normal"][p]) - 1.0) < 1e-4
        if kind == "known_orientation":
            pos = r["model"][:3]
            assert abs(abs(pos @ truth["position"][p]) - 1.0) < 1e-3


def test_exhaustive_sampler_enumerates_pairs_in_order():
    """exhaustive_sampler.cc:61-79: (0,1), (0,2), ..., (0,n-1), (1,2), ... wrapping around."""
    data, offsets, _ = synth.synth_ransac_v1(1, 6, "known_orientation", seed=9, inlier_lo=1.0, inlier_hi=1.0)
    pairs = [(i, j) for i in range(6) for j in range(i + 1, 6)]
    for k in (1, 2, 5, 6, 15, 16):
        prm = ol.default_ransac_params(1e-30, seed=0)   # nothing is an inlier: the last... first best model is kept
        prm.ransac_type = 3; prm.min_iterations = k; prm.max_iterations = k
        r = ol.ransac_estimate(8, data, prm, trace_capacity=64)
        assert r["num_iterations"] == k
    # with a generous threshold and one iteration the model is the (0, 1) pair's
    prm = ol.default_ransac_params(1.0, seed=0); prm.ransac_type = 3; prm.min_iterations = 1; prm.max_iterations = 1
    r = ol.ransac_estimate(8, data, prm)
    assert np.array_equal(r["model"][:3], ol.estimate_models(8, data[[0, 1]])[0][:3])
    assert len(pairs) == 15


def test_uncalibrated_relative_pose_minimal_model():
    """estimate_uncalibrated_relative_pose.cc:83-138 on exact data: the focal lengths
    (fundamental_matrix_util.cc:57-130), rotation and position direction come back."""
    R = rot((0.3, 1.0, -0.2), 14.0); position = np.array([0.9, 0.15, -0.3])
    t = -R @ position
    x1, x2 = _two_view(EIGHT_PTS, R, t)
    f1, f2 = 820.0, 1130.0
    corr = np.hstack([x1 * f1, x2 * f2])
    ol.set_estimator_params([1.0, 1e9])
    m = ol.estimate_models(9, corr)
    assert len(m) == 1
    m = m[0]
    assert abs(m[21] - f1) < 1e-6 * f1 and abs(m[22] - f2) < 1e-6 * f2
    Rm = m[9:18].reshape(3, 3)
    assert np.degrees(np.arccos(np.clip((np.trace(R @ Rm.T) - 1) / 2, -1, 1))) < 1e-6
    assert abs(m[18:21] @ position / np.linalg.norm(position) - 1.0) < 1e-10
    F = m[:9].reshape(3, 3)
    assert max(sampson(F, corr[i, :2], corr[i, 2:]) for i in range(8)) < 1e-16
    # focal-length bounds reject the model (:103-109)
    ol.set_estimator_params([1.0, 1000.0])
    assert len(ol.estimate_models(9, corr)) == 0
    ol.set_estimator_params([900.0, 1e9])
    assert len(ol.estimate_models(9, corr)) == 0
    # bounds below 1 are ignored
    ol.set_estimator_params([0.0, 0.0])
    assert len(ol.estimate_models(9, corr)) == 1


def test_estimate_uncalibrated_relative_pose_on_synthetic_pairs():
    data, offsets, truth = synth.synth_ransac_v1(2, 300, kind="uncalibrated", seed=0x5AC51600, inlier_lo=0.5, inlier_hi=0.7,
                                                 noise_px=0.3)
    ol.set_estimator_params([1.0, 1e9])
    for p in range(2):
        prm = ol.default_ransac_params(4.0, seed=17 + p); prm.failure_probability = 0.001
        r = ol.ransac_estimate(9, data[offsets[p]:offsets[p + 1]], prm)
        assert r["success"]
        tin = truth["inlier"][p]; mask = r["inlier_mask"].astype(bool)
        assert (mask & tin).sum() >= 0.7 * tin.sum() and (mask & ~tin).sum() <= 0.1 * tin.sum() + 3


def test_position_from_two_rays_and_pivoted_qr():
    """position_from_two_rays.cc:55-83 (ColPivHouseholderQR of the 4 x 3 system): exact data give the camera
    position; noisy data give numpy's least-squares solution; a repeated correspondence has rank 2."""
    st = synth.Stream(0xAB5, 1)
    for k in range(20):
        i = np.arange(6) + 10 * k
        c = np.array([st.normal(i[:3]).tolist()])[0] * 0.5
        X = np.stack([st.normal(i + 100), st.normal(i + 200), 5.0 + st.uniform(i + 300)], 1)[:2]
        uv = (X - c)[:, :2] / (X - c)[:, 2:]
        m = ol.estimate_models(10, np.hstack([uv, X]))
        assert len(m) == 1 and np.allclose(m[0][:3], c, atol=1e-12)
        uvn = uv + 1e-3 * np.stack([st.normal(i[:2] + 400), st.normal(i[:2] + 500)], 1)
        A = np.array([[1, 0, -uvn[0, 0]], [0, 1, -uvn[0, 1]], [1, 0, -uvn[1, 0]], [0, 1, -uvn[1, 1]]])
        b = np.array([X[0, 0] - uvn[0, 0] * X[0, 2], X[0, 1] - uvn[0, 1] * X[0, 2],
                      X[1, 0] - uvn[1, 0] * X[1, 2], X[1, 1] - uvn[1, 1] * X[1, 2]])
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        m = ol.estimate_models(10, np.hstack([uvn, X]))
        assert np.allclose(m[0][:3], ref, rtol=1e-11, atol=1e-11)
    same = np.hstack([uv[:1], X[:1]])
    assert len(ol.estimate_models(10, np.vstack([same, same]))) == 0


def test_estimate_absolute_pose_with_known_orientation_scene():
    data, offsets, truth = synth.synth_ransac_v1(2, 300, kind="absolute", seed=0x5AC51800, inlier_lo=0.5, inlier_hi=0.7)
    from pytheiasfm_amd import ransac as rs
    for p in range(2):
        aa = synth.matrix_to_angle_axis(truth["R"][p])
        rot = rs.RotateCorrespondences(data[offsets[p]:offsets[p + 1]], aa)
        prm = ol.default_ransac_params((4.0 / 1000.0) ** 2, seed=3 + p); prm.failure_probability = 0.001
        r = ol.ransac_estimate(10, rot, prm)
        assert r["success"]
        tin = truth["inlier"][p]; mask = r["inlier_mask"].astype(bool)
        assert (mask & tin).sum() >= 0.75 * tin.sum() and (mask & ~tin).sum() <= 0.1 * tin.sum() + 3
        assert np.linalg.norm(r["model"][:3] - truth["position"][p]) < 0.05


def _golden_estimator_runs():
    """(kind, estimator, threshold, model length, ransac types) of tests/golden/ransac_estimators.npz."""
    from tests.golden.make_oracle_golden import NEW_ESTIMATORS
    for kind, est, thresh, mlen in NEW_ESTIMATORS:
        yield kind, est, thresh, mlen, ((0, 1, 2, 3) if kind == "known_orientation" else (0, 1, 2))
    yield "abs_known", 10, (4.0 / 1000.0) ** 2, 3, (0,)


def test_golden_ransac_estimators():
    """The oracle reproduces tests/golden/ransac_estimators.npz (fundamental matrix, homography, plane, known-orientation
    relative / absolute position, uncalibrated relative pose; RANSAC / PROSAC / LMED / EXHAUSTIVE)."""
    g = np.load(os.path.join(HERE, "golden", "ransac_estimators.npz"))
    ol.set_estimator_params([1.0, 1e9])
    for kind, est, thresh, mlen, rtypes in _golden_estimator_runs():
        data, offsets = g[f"{kind}_data"], g[f"{kind}_offsets"]
        for rtype in rtypes:
            for i in range(3):
                prm = ol.default_ransac_params(thresh, 40 + i); prm.ransac_type = rtype; prm.failure_probability = 0.001
                if rtype == 3:
                    prm.max_iterations = 500
                r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm)
                assert r["num_iterations"] == g[f"{kind}_t{rtype}_iters"][i], (kind, rtype, i)
                assert np.array_equal(r["inlier_mask"], g[f"{kind}_t{rtype}_masks"][i])
                assert np.allclose(r["model"][:mlen], g[f"{kind}_t{rtype}_models"][i], rtol=0, atol=1e-13)


# ---------------------------------------------------------------- relative-pose RefineModel (LO-RANSAC)
def _angular_error_numpy(w, t, c):
    """angular_epipolar_error.h:54-91 written independently with numpy (Rodrigues from scipy-free formula)."""
    th = np.linalg.norm(w)
    K = cross_mat(w / th) if th > 0 else np.zeros((3, 3))
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    f1 = np.array([c[0], c[1], 1.0]); f2 = np.array([c[2], c[3], 1.0])
    M = np.eye(3) - np.outer(t, t)
    a = f1 @ M @ f1 + (R @ f2) @ M @ (R.T @ f2)
    b = t @ np.cross(f1, R.T @ f2)
    s = a * a / 4 - b * b
    return 1000.0 if s < 0 else a / 2 - np.sqrt(s)


def test_angular_epipolar_error_against_numpy():
    obl = ol
    st = synth.Stream(77, 1)
    i = np.arange(40)
    W = 0.4 * np.stack([st.normal(3 * i), st.normal(3 * i + 1), st.normal(3 * i + 2)], 1)
    T = np.stack([st.normal(3 * i + 200), st.normal(3 * i + 201), st.normal(3 * i + 202)], 1)
    T /= np.linalg.norm(T, axis=1, keepdims=True)
    Cc = np.stack([st.uniform(4 * i + 400), st.uniform(4 * i + 401), st.uniform(4 * i + 402), st.uniform(4 * i + 403)], 1) - 0.5
    for k in range(40):
        e = obl.angular_epipolar_error(W[k], T[k], Cc[k])
        assert abs(e - _angular_error_numpy(W[k], T[k], Cc[k])) <= 1e-13
    # exact pose, exact correspondence: zero; a non-unit "position" can make the root imaginary -> 1000 (the functor never fails)
    assert obl.angular_epipolar_error(np.zeros(3), np.array([1.0, 0, 0]), np.array([0.1, 0.2, 0.3, 0.2])) == pytest.approx(0.0, abs=1e-15)
    assert obl.angular_epipolar_error(np.zeros(3), np.array([3.0, 0, 0]), np.array([0.0, 1.0, 0.0, -1.0])) == 1000.0


def test_two_views_angular_oracle_converges_on_exact_correspondences():
    """BundleAdjustTwoViewsAngular as RefineModel configures it (TRUNCATED, 15 iterations, CGNR): from a perturbed
    pose the cost of noise-free correspondences falls by orders of magnitude, the position stays on the unit sphere."""
    obl = ol
    from pytheiasfm_amd import ba
    data, off, truth = synth.synth_ransac_v1(4, 300, kind="relative", noise_px=0.0, seed=0x5AC50A00)
    for p in range(1, 4):
        c = data[off[p]:off[p + 1]][truth["inlier"][p]]
        w = synth.matrix_to_angle_axis(truth["R"][p]); pos = truth["position"][p] / np.linalg.norm(truth["position"][p])
        o = ba.default_options(); o.max_num_iterations = 15; o.loss_function_type = 6; o.robust_loss_width = 1e-3
        x0 = np.concatenate([w + 0.01, pos + 0.02]); x0[3:] /= np.linalg.norm(x0[3:])
        pose, s = obl.two_views_angular(c, x0, o)
        assert s["success"] and s["final_cost"] < 1e-6 * s["initial_cost"] and s["num_successful_steps"] >= 3
        assert abs(np.linalg.norm(pose[3:]) - 1.0) <= 1e-14
        assert np.abs(pose[:3] - w).max() < np.abs(x0[:3] - w).max() and np.abs(pose[3:] - pos).max() < np.abs(x0[3:] - pos).max()
    # every residual beyond the truncation width: zero gradient, immediate convergence, pose untouched
    o = ba.default_options(); o.max_num_iterations = 15; o.loss_function_type = 6; o.robust_loss_width = 1e-12
    c = data[off[0]:off[1]][~truth["inlier"][0]][:50]
    x0 = np.array([0.1, -0.2, 0.05, 0.0, 0.6, 0.8])
    pose, s = obl.two_views_angular(c, x0, o)
    assert s["num_iterations"] == 0 and np.array_equal(pose, x0) and s["final_cost"] == s["initial_cost"]


@pytest.mark.parametrize("ri", range(2))
@pytest.mark.parametrize("pj", range(2))
def test_estimate_relative_pose_lo(ri, pj):
    """estimate_relative_pose_test.cc OutliersWithNoise_LO (:252-281): 70 % inliers, 1 px noise, use_lo from iteration 5.
    RefineModel replaces rotation and position and leaves the model's essential matrix alone."""
    R, position = REL_ROT[ri], REL_POS[pj] * (1.3 / 0.7 if pj == 0 else 1.0)
    pts = grid_points()
    t = -R @ position; t = t / np.linalg.norm(t)
    st = synth.Stream(65, 50 + 10 * ri + pj)
    x1 = pts[:, :2] / pts[:, 2:]
    p2 = pts @ R.T + t
    x2 = p2[:, :2] / p2[:, 2:]
    out = np.arange(27) >= 0.7 * 27
    x1[out] = 2 * np.stack([st.uniform(4 * np.arange(27)), st.uniform(4 * np.arange(27) + 1)], 1)[out] - 1
    x2[out] = 2 * np.stack([st.uniform(4 * np.arange(27) + 2), st.uniform(4 * np.arange(27) + 3)], 1)[out] - 1
    x1 = x1 + 1e-3 * np.stack([st.normal(4 * np.arange(27) + 500), st.normal(4 * np.arange(27) + 501)], 1)
    x2 = x2 + 1e-3 * np.stack([st.normal(4 * np.arange(27) + 502), st.normal(4 * np.arange(27) + 503)], 1)
    prm = ol.default_ransac_params((2.0 / 1000.0) ** 2, seed=65)
    prm.use_mle = 1; prm.failure_probability = 0.001; prm.use_lo = 1; prm.lo_start_iterations = 5
    r = ol.ransac_estimate(0, np.hstack([x1, x2]), prm)
    nlo = ol.rlib().oracle_last_lo_iterations()
    assert r["success"] and r["num_inliers"] > 5 and nlo >= 1
    E = r["model"][0:9].reshape(3, 3); Rm = r["model"][9:18].reshape(3, 3); pos = r["model"][18:21]
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ Rm.T) - 1) / 2, -1, 1)))
    tdiff = np.degrees(np.arccos(np.clip(position / np.linalg.norm(position) @ pos, -1, 1)))
    assert ang < 5.0 and tdiff < 5.0
    assert np.abs(Rm @ Rm.T - np.eye(3)).max() <= 1e-12 and abs(np.linalg.norm(pos) - 1.0) <= 1e-12
    # the essential matrix is the minimal sample's: rank 2 with two equal singular values, but no longer [t]x R of the refined pose
    sv = np.linalg.svd(E, compute_uv=False)
    assert sv[2] <= 1e-9 * sv[0] and abs(sv[0] - sv[1]) <= 1e-6 * sv[0]


def test_golden_two_view_lo():
    """The oracle reproduces tests/golden/two_view_lo.npz: BundleAdjustTwoViewsAngular vectors and relative-pose LO-RANSAC runs."""
    from pytheiasfm_amd import ba
    g = np.load(os.path.join(HERE, "golden", "two_view_lo.npz"))
    o = ba.default_options(); o.max_num_iterations = 15; o.loss_function_type = 6; o.robust_loss_width = 2e-4
    for k in range(4):
        pose, s = ol.two_views_angular(g[f"tv{k}_corr"], g[f"tv{k}_x0"], o)
        assert np.allclose(pose, g[f"tv{k}_pose"], rtol=0, atol=1e-13)
        assert [s["success"], s["termination_type"], s["num_iterations"], s["num_successful_steps"]] == list(g[f"tv{k}_ints"])
        assert np.allclose([s["initial_cost"], s["final_cost"]], g[f"tv{k}_costs"], rtol=1e-12, atol=0)
    data, offsets = g["rel_data"], g["rel_offsets"]
    for i in range(3):
        prm = ol.default_ransac_params((2.0 / 1000.0) ** 2, 50 + i)
        prm.use_mle = 1; prm.use_lo = 1; prm.lo_start_iterations = 5; prm.min_iterations = 50; prm.failure_probability = 0.001
        r = ol.ransac_estimate(0, data[offsets[i]:offsets[i + 1]], prm)
        assert r["num_iterations"] == g["rel_lo_iters"][i] and ol.rlib().oracle_last_lo_iterations() == g["rel_lo_nlo"][i] >= 1
        assert np.array_equal(r["inlier_mask"], g["rel_lo_masks"][i])
        assert np.allclose(r["model"][:21], g["rel_lo_models"][i], rtol=0, atol=1e-13)


def _homography_scene(seed, n=120, noise=0.5):
    """Pixel correspondences of a plane seen by two cameras: x2 ~ H x1 (+ noise)."""
    st = synth.Stream(seed, 3)
    i = np.arange(n)
    H = np.array([[1.02, 0.03, 12.0], [-0.02, 0.98, -7.0], [1e-5, -2e-5, 1.0]])
    x1 = np.stack([1000 * st.uniform(2 * i), 800 * st.uniform(2 * i + 1)], 1)
    p = np.hstack([x1, np.ones((n, 1))]) @ H.T
    x2 = p[:, :2] / p[:, 2:] + noise * np.stack([st.normal(2 * i + 500), st.normal(2 * i + 501)], 1)
    x1 = x1 + noise * np.stack([st.normal(2 * i + 900), st.normal(2 * i + 901)], 1)
    return H, np.hstack([x1, x2])


def test_optimize_homography_oracle():
    """OptimizeHomography (bundle_adjust_two_views.cc:298-358): from a perturbed H the symmetric transfer cost falls to
    the noise level, the result is normalised by H(2,2); the residual terms agree with numpy."""
    from pytheiasfm_amd import ba
    H, corr = _homography_scene(5)
    H0 = H * 1.7 + np.array([[0.01, -0.01, 3.0], [0.01, 0.0, -2.0], [1e-6, 0, 0.0]])
    o = ba.default_options(); o.max_num_iterations = 15; o.loss_function_type = 0
    Hr, s = ol.optimize_homography(corr, H0, o)
    def cost(Hm):
        x1 = np.hstack([corr[:, :2], np.ones((len(corr), 1))]); x2 = np.hstack([corr[:, 2:], np.ones((len(corr), 1))])
        f = x1 @ Hm.T; b = x2 @ np.linalg.inv(Hm).T
        return 0.5 * (((f[:, :2] / f[:, 2:] - corr[:, 2:]) ** 2).sum() + ((b[:, :2] / b[:, 2:] - corr[:, :2]) ** 2).sum())
    assert abs(s["initial_cost"] - cost(H0)) <= 1e-9 * cost(H0) and abs(s["final_cost"] - cost(Hr)) <= 1e-9 * cost(Hr)
    assert s["success"] and s["final_cost"] < 0.1 * s["initial_cost"] and Hr[2, 2] == 1.0
    assert np.abs(Hr - H).max() < 0.5 and np.abs(Hr[:2, :2] - H[:2, :2]).max() < 5e-3
    assert s["final_cost"] / len(corr) < 2.0 * 0.5 ** 2 * 2      # ~ four residuals of 0.5 px noise


def test_estimate_homography_lo():
    """use_lo with HomographyEstimator::RefineModel (estimate_homography.cc:89-104)."""
    data, offsets, truth = synth.synth_ransac_v1(2, 300, "homography", seed=0x5AC52200, inlier_lo=0.6, inlier_hi=0.7)
    for i in range(2):
        prm = ol.default_ransac_params(16.0, seed=70 + i); prm.failure_probability = 0.001
        r0 = ol.ransac_estimate(6, data[offsets[i]:offsets[i + 1]], prm)
        prm.use_lo = 1; prm.lo_start_iterations = 5; prm.min_iterations = 30
        r1 = ol.ransac_estimate(6, data[offsets[i]:offsets[i + 1]], prm)
        nlo = ol.rlib().oracle_last_lo_iterations()
        assert r1["success"] and nlo >= 1 and r1["model"][8] == 1.0
        assert r1["inlier_mask"][truth["inlier"][i]].mean() > 0.9 and r1["num_inliers"] >= 0.95 * r0["num_inliers"]


def _fundamental_scene(seed, n=150, noise=0.5):
    data, offsets, truth = synth.synth_ransac_v1(1, n, "fundamental", seed=seed, inlier_lo=1.0, inlier_hi=1.0, noise_px=noise)
    K = np.array([[1000.0, 0, 500], [0, 1000.0, 400], [0, 0, 1]])
    E = cross_mat(truth["t"][0]) @ truth["R"][0]
    F = np.linalg.inv(K).T @ E @ np.linalg.inv(K)
    return F / np.linalg.norm(F), data


def test_fundamental_manifold_and_sampson_derivatives():
    """fundamental_matrix_parameterization.h Plus and its Jacobian at zero, sampson_error.h and its gradient:
    the closed forms against central differences; Plus(F, 0) is the rank-2, sigma_0 = 1 normalisation of F."""
    L = ol.rlib(); dp = capi.c_double_p
    L.oracle_fund_plus.argtypes = [dp, dp, dp]; L.oracle_fund_plus_jacobian.argtypes = [dp, dp]
    L.oracle_sampson_residual.argtypes = [dp, dp, dp]; L.oracle_sampson_residual.restype = C.c_double
    F, data = _fundamental_scene(0x5AC52400)
    F = np.ascontiguousarray(F + 1e-3 * np.arange(9).reshape(3, 3) / 9.0)           # full rank, arbitrary scale
    def plus(d):
        out = np.zeros(9); dd = np.ascontiguousarray(d, dtype=np.float64)
        L.oracle_fund_plus(capi.ptr(F, C.c_double), capi.ptr(dd, C.c_double), capi.ptr(out, C.c_double)); return out
    F0 = plus(np.zeros(7)).reshape(3, 3)
    sv = np.linalg.svd(F0, compute_uv=False); svF = np.linalg.svd(F, compute_uv=False)
    assert abs(sv[0] - 1) <= 1e-12 and abs(sv[1] - svF[1] / svF[0]) <= 1e-12 and sv[2] <= 1e-12
    PJ = np.zeros(63); L.oracle_fund_plus_jacobian(capi.ptr(F, C.c_double), capi.ptr(PJ, C.c_double)); PJ = PJ.reshape(9, 7)
    h = 1e-6
    for q in range(7):
        e = np.zeros(7); e[q] = h
        fd = (plus(e) - plus(-e)) / (2 * h)
        assert np.abs(fd - PJ[:, q]).max() <= 1e-8, q
    for c in data[:20]:
        cc = np.ascontiguousarray(c); J = np.zeros(9)
        r = L.oracle_sampson_residual(capi.ptr(F, C.c_double), capi.ptr(cc, C.c_double), capi.ptr(J, C.c_double))
        x1 = np.array([c[0], c[1], 1.0]); x2 = np.array([c[2], c[3], 1.0])
        Fx = F @ x1; Ftx = F.T @ x2
        assert abs(r - (x2 @ Fx) ** 2 / (Fx[0] ** 2 + Fx[1] ** 2 + Ftx[0] ** 2 + Ftx[1] ** 2)) <= 1e-12 * max(r, 1e-30)
        for k in range(9):
            Fp = F.copy().reshape(9); Fm = Fp.copy(); hk = 1e-5 * max(abs(Fp[k]), 1e-3); Fp[k] += hk; Fm[k] -= hk
            rp = L.oracle_sampson_residual(capi.ptr(Fp, C.c_double), capi.ptr(cc, C.c_double), None)
            rm_ = L.oracle_sampson_residual(capi.ptr(Fm, C.c_double), capi.ptr(cc, C.c_double), None)
            assert abs((rp - rm_) / (2 * hk) - J[k]) <= 1e-4 * max(abs(J[k]), 1.0)


def test_optimize_fundamental_oracle_and_lo():
    """OptimizeFundamentalMatrix from a perturbed F: the Sampson cost falls, the result is rank 2 with sigma_0 = 1;
    use_lo of EstimateFundamentalMatrix (RefineModel: 2 iterations) keeps / improves the inlier set."""
    from pytheiasfm_amd import ba
    F, data = _fundamental_scene(0x5AC52401)
    F0 = F + 2e-8 * np.array([[1.0, -2, 300], [2, 1, -200], [-300, 200, 5e4]])
    o = ba.default_options(); o.max_num_iterations = 20
    Fr, s = ol.optimize_fundamental(data, F0, o)
    sv = np.linalg.svd(Fr, compute_uv=False)
    assert s["success"] and s["final_cost"] < 0.2 * s["initial_cost"] and abs(sv[0] - 1) <= 1e-12 and sv[2] <= 1e-12
    data, offsets, truth = synth.synth_ransac_v1(2, 300, "fundamental", seed=0x5AC52402, inlier_lo=0.6, inlier_hi=0.7)
    for i in range(2):
        prm = ol.default_ransac_params(4.0, seed=90 + i); prm.failure_probability = 0.001
        r0 = ol.ransac_estimate(5, data[offsets[i]:offsets[i + 1]], prm)
        prm.use_lo = 1; prm.lo_start_iterations = 5; prm.min_iterations = 30
        r1 = ol.ransac_estimate(5, data[offsets[i]:offsets[i + 1]], prm)
        nlo = ol.rlib().oracle_last_lo_iterations()
        assert r1["success"] and nlo >= 1 and r1["num_inliers"] >= 0.9 * r0["num_inliers"]
        assert r1["inlier_mask"][truth["inlier"][i]].mean() > 0.75


LO_GOLDEN = (("fundamental", 5, 4.0, 9), ("homography", 6, 16.0, 9), ("uncalibrated", 9, 4.0, 23))


def test_golden_refine_model_vectors():
    """tests/golden/two_view_lo.npz, second part: OptimizeHomography / OptimizeFundamentalMatrix vectors and the LO-RANSAC runs
    of the fundamental-matrix, homography and uncalibrated relative-pose estimators."""
    from pytheiasfm_amd import ba
    g = np.load(os.path.join(HERE, "golden", "two_view_lo.npz"))
    oh = ba.default_options(); oh.max_num_iterations = 15; oh.loss_function_type = 6; oh.robust_loss_width = 50.0
    of = ba.default_options(); of.max_num_iterations = 2
    for k in range(2):
        Hr, s = ol.optimize_homography(g[f"hom{k}_corr"], g[f"hom{k}_H0"], oh)
        assert np.allclose(Hr, g[f"hom{k}_H"], rtol=1e-12, atol=0) and [s["num_iterations"], s["num_successful_steps"]] == list(g[f"hom{k}_ints"])
        Fr, s = ol.optimize_fundamental(g[f"fund{k}_corr"], g[f"fund{k}_F0"], of)
        assert np.allclose(Fr, g[f"fund{k}_F"], rtol=1e-11, atol=1e-15) and [s["num_iterations"], s["num_successful_steps"]] == list(g[f"fund{k}_ints"])
    ol.set_estimator_params([1.0, 1e9])
    for kind, est, thresh, mlen in LO_GOLDEN:
        data, offsets = g[f"lo_{kind}_data"], g[f"lo_{kind}_offsets"]
        for i in range(2):
            prm = ol.default_ransac_params(thresh, 60 + i); prm.failure_probability = 0.001
            prm.use_lo = 1; prm.lo_start_iterations = 5; prm.min_iterations = 30
            r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm)
            assert r["num_iterations"] == g[f"lo_{kind}_iters"][i] and ol.rlib().oracle_last_lo_iterations() == g[f"lo_{kind}_nlo"][i]
            assert np.array_equal(r["inlier_mask"], g[f"lo_{kind}_masks"][i])
            assert np.allclose(r["model"][:mlen], g[f"lo_{kind}_models"][i], rtol=1e-11, atol=1e-14)


def _tri_params(thresh=1.0, seed=151, min_it=5):
    pc = ol.default_ransac_params(thresh, seed=seed)
    pc.min_iterations = min_it
    return pc


def test_estimate_triangulation_scenes_of_the_reference_test():
    """estimate_triangulation_test.cc:103-166 restated against the oracle: two views give the point to 1e-6 with both
    observations inliers; ten observations and two outliers (EXHAUSTIVE over all 66 pairs) give it with >= 6 inliers."""
    from pytheiasfm_amd import ransac
    from tests import tri_scenes
    cams, feats = tri_scenes.scene(2, 0, 1)
    pc = _tri_params(min_it=1); pc.min_iterations = pc.max_iterations = 1; pc.ransac_type = 3
    o = ol.ransac_estimate(11, ransac.triangulation_observations(cams, feats), pc)
    X = o["model"][:4]
    assert o["success"] and o["num_inliers"] == 2 and np.linalg.norm(X[:3] / X[3] - tri_scenes.POINT[:3]) < 1e-6
    cams, feats = tri_scenes.scene(10, 2, 2)
    pc = _tri_params(); pc.min_iterations = pc.max_iterations = 66; pc.ransac_type = 3
    o = ol.ransac_estimate(11, ransac.triangulation_observations(cams, feats), pc)
    X = o["model"][:4]
    assert o["success"] and o["num_iterations"] == 66
    assert np.linalg.norm(X[:3] / X[3] - tri_scenes.POINT[:3]) < 1e-6
    assert np.nonzero(o["inlier_mask"])[0].tolist() == list(range(10))


def test_estimate_triangulation_error_is_the_reprojection_error_of_the_camera_model():
    """TriangulationEstimator::Error = |pixel - Camera::ProjectPoint|^2 with DBL_MAX behind the camera: the inlier set
    of a noisy 40-observation RANSAC run equals the numpy reprojection test of the returned point, for a distorted
    pinhole and a double-sphere camera."""
    from pytheiasfm_amd import ransac
    from tests import tri_scenes
    for model, intr, seed in ((synth.CAM_PINHOLE, synth.PINHOLE_INTR, 3), (synth.CAM_DOUBLE_SPHERE, synth.DOUBLE_SPHERE_INTR, 4)):
        cams, feats = tri_scenes.scene(32, 8, seed, model=model, intrinsics=intr, spread=0.05, noise=0.4)
        # normalized features of the true model: the ray direction through the pixel (the pinhole fixed point or, for the
        # double sphere, the camera-frame direction of the point itself for the inlying observations)
        norm = np.zeros((40, 2))
        for i, c in enumerate(cams):
            if model == synth.CAM_PINHOLE:
                norm[i] = c.pixel_to_normalized(feats[i])
            else:
                q = synth.angle_axis_to_matrix(c.orientation) @ (tri_scenes.POINT[:3] - c.position)
                norm[i] = q[:2] / q[2] + (feats[i] - synth.project(model, c.intrinsics[None], np.concatenate([c.position, c.orientation])[None], tri_scenes.POINT[None])[0][0]) / intr[0]
        data = ransac.triangulation_observations(cams, feats, norm)
        pc = _tri_params(thresh=4.0, seed=7, min_it=50)
        o = ol.ransac_estimate(11, data, pc)
        assert o["success"]
        X = o["model"][:4]
        ext = np.array([np.concatenate([c.position, c.orientation]) for c in cams])
        K = np.array([c.intrinsics for c in cams])
        uv, ok = synth.project(model, K, ext, np.tile(X, (40, 1)))
        depth = np.einsum("nij,nj->ni", synth.angle_axis_to_matrix(ext[:, 3:]), X[:3] - X[3] * ext[:, :3])[:, 2] / X[3]
        e = np.sum((uv - feats) ** 2, axis=1)
        expect = (e < 4.0) & (depth > 0)
        assert np.array_equal(o["inlier_mask"].astype(bool), expect)
        assert expect[:32].sum() >= 24 and np.linalg.norm(X[:3] / X[3] - tri_scenes.POINT[:3]) < 0.2


def test_estimate_triangulation_mirror_host_logic():
    """estimate_triangulation.cc:116-124: CHECK_EQ on the sizes, false below two observations (no device needed)."""
    from pytheiasfm_amd import ransac
    from tests import tri_scenes
    cams, feats = tri_scenes.scene(1, 0, 5)
    ok, X, s = ransac.EstimateTriangulation(ransac.RansacParameters(), cams, feats)
    assert not ok
    with pytest.raises(capi.TheiaHipError):
        ransac.EstimateTriangulation(ransac.RansacParameters(), cams, np.zeros((2, 2)))
    # pixel -> normalized inverts the pinhole projection (fixed point of pinhole_camera_model.h:263-298)
    cam = ransac.Camera([0.1, -0.2, 0.3], [0.02, -0.01, 0.03], synth.PINHOLE_INTR)
    n = cam.pixel_to_normalized(np.array([1200.0, 300.0]))
    f, a, sk, cx, cy, k1, k2 = synth.PINHOLE_INTR
    r2 = n @ n; d = 1 + r2 * (k1 + k2 * r2)
    assert np.allclose([f * n[0] * d + sk * n[1] * d + cx, f * a * n[1] * d + cy], [1200.0, 300.0], atol=1e-6)
    P = cam.projection_matrix()
    assert np.allclose(P[:, :3] @ cam.position + P[:, 3], 0.0, atol=1e-15)


def test_six_point_radial_distortion_homography_reference_scenes():
    """six_point_radial_distortion_homography_test.cc:171-226 restated: the six reference points, rotation about z by 10
    (13) degrees, f = 1500 / 1600, distortions -1e-7 / -2e-7; noise-free: one solution with mean symmetric error < 1e-4
    (the reference's bound), judged by a numpy statement of CheckRadialSymmetricError.  The reference's NoiseTest (:198-226)
    solves from the NOISE-FREE normalised points too -- GenerateDistortedImagePoints normalises `distorted_point`, not the
    noisy copy -- and perturbs only the pixels the error is evaluated on (point i receives 6 - i uniform +-0.5 px draws: the noise
    loop sits inside the point loop, :88-95); bound 2."""
    from tests import radhom_scenes as rh
    f1, f2, k1, k2 = 1500.0, 1600.0, -1e-7, -2e-7
    for deg, t, noise, bound in ((10.0, [1.0, 1.0, 1.0], 0.0, 1e-4), (13.0, [0.0, 1.0, 1.0], 0.5, 2.0)):
        rows = rh.rows(rh.REFERENCE_POINTS, rh.rotation_z(deg), np.array(t), f1, f2, k1, k2)
        models = ol.estimate_models(12, rows)
        assert 1 <= len(models) <= 2
        rng = np.random.default_rng(53)
        draws = []
        for trial in range(20 if noise else 1):       # the expectation of the reference's noisy evaluation (one draw of it is 1.1 .. 2.1)
            ev = rows.copy()
            for i in range(6 if noise else 0):
                ev[i, 0:4] += rng.uniform(-noise, noise, size=(6 - i, 4)).sum(axis=0)   # AddNoiseToProjection: uniform in +-noise_factor
            errs = []
            for m in models:
                H = m[:9].reshape(3, 3)
                assert np.allclose(m[11:20].reshape(3, 3) @ H, np.eye(3), atol=1e-9)       # the row carries H^-1
                e_np = np.mean([rh.symmetric_error(H, m[9], m[10], r[0:2], r[2:4], f1, f2) for r in ev])
                e_or = np.mean([ol.model_error(12, m, r) for r in ev])
                assert abs(e_np - e_or) <= 1e-9 * max(1.0, e_np)
                errs.append(e_np)
            draws.append(min(errs))
        assert np.mean(draws) < bound
        if True:
            best = models[int(np.argmin(errs))]
            # l = k f^2 in normalised units (the test's comment: "we used normalized image points for estimation")
            assert abs(best[9] - k1 * f1 * f1) < 1e-6 and abs(best[10] - k2 * f2 * f2) < 1e-6


def test_estimate_radial_homography_matrix_with_outliers():
    """EstimateRadialHomographyMatrix on 200 planar correspondences, 30 % gross outliers, 0.3 px noise: the oracle's
    RANSAC recovers the distortions and an inlier set that equals the numpy symmetric-error test of the returned model."""
    from tests import radhom_scenes as rh
    rng = np.random.default_rng(11)
    f1, f2, k1, k2 = 1200.0, 1300.0, -2e-7, -1e-7
    pts = np.column_stack([rng.uniform(-2, 2, 200), rng.uniform(-2, 2, 200), np.full(200, 4.0)])
    R = synth.angle_axis_to_matrix(np.array([[0.05, -0.1, 0.15]]))[0]
    rows = rh.rows(pts, R, np.array([0.5, -0.2, 0.1]), f1, f2, k1, k2, 0.3, rng)
    out = rng.uniform(size=200) < 0.3
    rows[out, 2:4] = rng.uniform(-600, 600, (int(out.sum()), 2)); rows[out, 6:8] = rows[out, 2:4] / f2
    pc = ol.default_ransac_params(2.0 ** 2, seed=5); pc.min_iterations = 200; pc.failure_probability = 1e-3
    o = ol.ransac_estimate(12, rows, pc)
    assert o["success"]
    m = o["model"]
    H = m[:9].reshape(3, 3)
    e = np.array([rh.symmetric_error(H, m[9], m[10], r[0:2], r[2:4], f1, f2) for r in rows])
    sure = np.abs(e - 4.0) > 1e-6
    assert np.array_equal((e < 4.0)[sure], o["inlier_mask"].astype(bool)[sure])
    assert o["inlier_mask"][~out].mean() > 0.9 and o["inlier_mask"][out].mean() < 0.05
    assert abs(m[9] - k1 * f1 * f1) < 0.05 and abs(m[10] - k2 * f2 * f2) < 0.05


def test_gdls_similarity_transform_reference_scenes():
    """gdls_similarity_transform_test.cc:158-215 restated: 4 (8) points seen from 4 camera centres under rotation 13 degrees
    about z, t = (1, 1, 1), s = 2.5; noise-free: a solution within the reference's bounds (rotation 1e-4 degrees, squared
    translation error 1e-6, relative scale 1e-6) and rays reproduced."""
    from tests import gdls_scenes as gs
    R, t, s = gs.rotation_z(13.0), np.array([1.0, 1.0, 1.0]), 2.5
    pts4 = [[-1.0, 3.0, 3.0], [1.0, -1.0, 2.0], [-1.0, 1.0, 2.0], [2.0, 1.0, 3.0]]
    pts8 = pts4 + [[-1.0, -3.0, 2.0], [1.0, -2.0, 1.0], [-1.0, 4.0, 2.0], [-2.0, 2.0, 3.0]]
    for pts, centers in ((pts4, [[-1.0, 0, 0], [0.0, 0, 0], [2.0, 0, 0], [3.0, 0, 0]]), (pts8, [[0, 1.0, 0], [0, 0, 0], [0, 2.0, 0], [0, 3.0, 0]])):
        origins, rays = gs.generalised(pts, centers, R, t, s)
        q, tt, sc = ol.gdls_similarity(origins, rays, pts)
        assert len(q) >= 1
        matched = False
        for i in range(len(q)):
            Rq = gs.quat_to_matrix(q[i])
            ang = np.arccos(np.clip((np.trace(Rq.T @ R) - 1.0) / 2.0, -1.0, 1.0))
            # every returned solution reproduces the rays (reprojection < 1 / 512 in the ray frame)
            p = (np.asarray(pts) @ Rq.T + tt[i]) / sc[i] - origins
            p /= np.linalg.norm(p, axis=1, keepdims=True)
            assert np.abs(np.cross(p, rays)).max() < 1.0 / 512.0
            if ang < np.deg2rad(1e-4) and np.sum((tt[i] - t) ** 2) < 1e-6 and abs(sc[i] - s) / s < 1e-6:
                matched = True
        assert matched


def test_estimate_similarity_transformation_2d_3d_with_outliers():
    """EstimateSimilarityTransformation2D3D through the oracle's RANSAC: a rig of five pinhole cameras in its own frame,
    120 correspondences with 25 % outliers and 0.5 px noise: the similarity that moves the rig onto the world cameras is
    recovered and the inlier set equals a numpy statement of the error (TransformCamera + projection)."""
    from pytheiasfm_amd import ransac
    from tests import gdls_scenes as gs
    corr, truth = gs.cameras(5, 120, seed=3, outlier_frac=0.25, noise=0.5)
    rows = ransac.similarity_correspondence_rows(corr)
    pc = ol.default_ransac_params(3.0 ** 2, seed=9); pc.min_iterations = 100; pc.failure_probability = 1e-3
    o = ol.ransac_estimate(13, rows, pc)
    assert o["success"]
    m = o["model"]
    Rs, ts, ss = m[:9].reshape(3, 3), m[9:12], m[12]
    assert np.abs(Rs - truth["R"]).max() < 5e-3 and np.abs(ts - truth["t"]).max() < 5e-2 and abs(ss - truth["s"]) < 2e-2
    e = np.zeros(len(corr)); depth = np.zeros(len(corr))
    for i, c in enumerate(corr):
        pos = ss * Rs @ c.camera.position + ts
        Rc = synth.angle_axis_to_matrix(c.camera.orientation[None])[0] @ Rs.T
        q = Rc @ (c.point3d[:3] - c.point3d[3] * pos)
        depth[i] = q[2] / c.point3d[3]
        f, a, sk, cx, cy = c.camera.intrinsics[:5]
        x, y = q[0] / q[2], q[1] / q[2]
        uv = np.array([f * x + sk * y + cx, f * a * y + cy])
        e[i] = np.sum((uv - c.observation) ** 2)
    expect = (e < 9.0) & (depth >= 0)
    sure = np.abs(e - 9.0) > 1e-6
    assert np.array_equal(expect[sure], o["inlier_mask"].astype(bool)[sure])
    assert o["inlier_mask"][~truth["outlier"]].mean() > 0.9 and o["inlier_mask"][truth["outlier"]].mean() < 0.1


# ---------------------------------------------------------------- P4Pf (uncalibrated absolute pose)
def _best_reprojection(models, X, px):
    errs = [np.linalg.norm(ps.project(m[:12].reshape(3, 4), X) - px, axis=1).max() for m in models]
    return (min(errs), int(np.argmin(errs))) if errs else (np.inf, -1)


def test_p4pf_reference_basic_vectors():
    """four_point_focal_length_test.cc:119-146 (BasicTest): f = 800, the listed pose and four world points; one of the
    solutions reprojects the four points within 1e-4 px (BasicTest) / 10 px with 0.5 px noise (BasicNoiseTest)."""
    P, X, px = ps.basic_scene()
    m = ol.estimate_models(ps.EST, np.concatenate([px, X], axis=1))
    assert 1 <= len(m) <= 10
    err, j = _best_reprojection(m, X, px)
    assert err < 1e-4
    Pm = m[j][:12].reshape(3, 4)
    assert abs(np.linalg.norm(Pm[0, :3]) - 800.0) < 1e-6 and np.abs(Pm / Pm[2, 3] - P / P[2, 3]).max() < 1e-6
    rng = np.random.default_rng(1)
    for _ in range(20):
        noisy = px + rng.normal(0.0, 0.5, px.shape)
        err, _ = _best_reprojection(ol.estimate_models(ps.EST, np.concatenate([noisy, X], axis=1)), X, noisy)
        assert err < 10.0


def test_p4pf_random_scenes_and_the_action_matrix():
    """RandomTest (:148-187, tolerance 0.1 px) over 200 scenes, and the elimination itself: the action matrix of z has the
    depth of the fourth point as an eigenvalue with eigenvector [1, z, y, x, w, ...] of the true solution."""
    rng = np.random.default_rng(7)
    errs, resid = [], []
    for _ in range(300):
        P, X, px, focal = ps.random_scene(rng)
        sub = np.concatenate([px, X], axis=1)
        errs.append(_best_reprojection(ol.estimate_models(ps.EST, sub), X, px)[0])
        T = np.zeros((10, 10))
        assert ol.rlib().oracle_p4pf_action(capi.ptr(np.ascontiguousarray(sub), C.c_double), capi.ptr(T, C.c_double)) == 1
        depth = (P @ np.concatenate([X, np.ones((4, 1))], axis=1).T)[2]
        fvar = np.linalg.norm(px, axis=1).mean()
        x, y, z, w = depth[1] / depth[0], depth[2] / depth[0], depth[3] / depth[0], (focal / fvar) ** 2
        v = np.array([1.0, z, y, x, w, z * z, y * z, x * z, w * z, w * y])
        resid.append(np.abs(T @ v - z * v).max() / (abs(z) * np.abs(v).max()))
    errs, resid = np.array(errs), np.array(resid)
    assert np.median(errs) < 1e-7 and np.mean(errs < 0.1) >= 0.99   # a minimal solver: a few ill-conditioned draws in a thousand
    assert np.median(resid) < 1e-10 and np.percentile(resid, 99) < 1e-5


def test_estimate_uncalibrated_absolute_pose_with_outliers():
    """EstimateUncalibratedAbsolutePose through the oracle's RANSAC loop: 150 correspondences, 30 % outliers, 0.5 px noise;
    the inlier set equals a numpy statement of the squared reprojection error of the winning projection matrix."""
    rng = np.random.default_rng(11)
    data, P, focal, good = ps.ransac_scene(rng, 150)
    pc = ol.default_ransac_params(2.0 ** 2, seed=5); pc.failure_probability = 1e-3
    o = ol.ransac_estimate(ps.EST, data, pc)
    assert o["success"]
    Pm = o["model"][:12].reshape(3, 4)
    assert abs(np.linalg.norm(Pm[0, :3]) / np.linalg.norm(Pm[2, :3]) - focal) < 0.03 * focal
    e = np.sum((ps.project(Pm, data[:, 2:5]) - data[:, :2]) ** 2, axis=1)
    sure = np.abs(e - 4.0) > 1e-6
    assert np.array_equal((e < 4.0)[sure], o["inlier_mask"].astype(bool)[sure])
    assert o["inlier_mask"][good].mean() > 0.85 and o["inlier_mask"][~good].mean() < 0.05


@pytest.mark.parametrize("num_outliers", [0, 50])
def test_get_best_pose_from_essential_matrix_reference_scenes(num_outliers):
    """essential_matrix_utils_test.cc:130-199 (GetBestPoseFromEssentialMatrix AllInliers / MostlyInliers), restated: 100 poses
    (rotation <= 15 degrees, unit translation), 100 points around (0, 0, 100) in front of both cameras and `num_outliers` around
    (0, 0, -100) behind them, E = [t]x R without noise.  The reference's criteria: the number of points in front is exactly the
    number of inliers, rotation and position within 1e-12 (Frobenius / Euclidean).  Also DecomposeEssentialMatrix's test (:55-84):
    one of the two rotations and +-translation match the generating pose."""
    rng = np.random.default_rng(51 + num_outliers)
    for _ in range(100):
        aa = rng.normal(size=3); aa *= np.deg2rad(15.0 * rng.uniform()) / np.linalg.norm(aa)
        K = np.array([[0, -aa[2], aa[1]], [aa[2], 0, -aa[0]], [-aa[1], aa[0], 0]]); th = np.linalg.norm(aa)
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)
        t = rng.uniform(-1, 1, 3); t /= np.linalg.norm(t)
        E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
        X = np.concatenate([rng.uniform(-1, 1, (100, 3)) + [0, 0, 100], rng.uniform(-1, 1, (num_outliers, 3)) + [0, 0, -100]])
        Y = X @ R.T + t
        corr = np.concatenate([X[:, :2] / X[:, 2:3], Y[:, :2] / Y[:, 2:3]], axis=1)
        n, Re, pe = ol.best_pose_from_essential(E, corr)
        assert n == 100
        assert np.linalg.norm(R - Re) < 1e-12 and np.linalg.norm(-R.T @ t - pe) < 1e-12


def test_focal_lengths_from_fundamental_matrix_reference_scenes():
    """fundamental_matrix_util_test.cc:54-80 (FundamentalMatrixUtil.FocalLengths), restated: 100 random poses (angle-axis in
    [-1, 1]^3, unit translation), F = K2^-1 [t]x R K1^-1 (ComposeFundamentalMatrix :251-269) with focal lengths 800 and 1000;
    the reference's criterion: both recovered within 1e-6."""
    rng = np.random.default_rng(51)
    for _ in range(100):
        aa = rng.uniform(-1, 1, 3); th = np.linalg.norm(aa); ax = aa / th
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        t = rng.uniform(-1, 1, 3); t /= np.linalg.norm(t)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        F = np.diag([1 / 1000.0, 1 / 1000.0, 1.0]) @ tx @ R @ np.diag([1 / 800.0, 1 / 800.0, 1.0])
        ok, f1, f2 = ol.focal_lengths_from_fundamental(F)
        assert ok and abs(f1 - 800.0) < 1e-6 and abs(f2 - 1000.0) < 1e-6, (f1, f2)
