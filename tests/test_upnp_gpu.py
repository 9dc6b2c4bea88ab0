"""GPU: THEIA_EST_RIGID_TRANSFORMATION_2D3D (UPnP) through the C-ABI against the oracle -- hypotheses, inlier sets, iteration
counts and the elected transformation BIT-IDENTICAL (the template lives in registers and every entry sees the oracle's
operations in its order: csrc/upnp_kernels.hip), the accumulating cost parameters carried across the rounds of a call."""
import numpy as np
import pytest

from pytheiasfm_amd import ransac
from tests import oracle_lib as ol
from tests import upnp_scenes as sc

pytestmark = pytest.mark.gpu
EST = ransac.EST_RIGID_TRANSFORMATION_2D3D


def _batch(nprob, seed, outlier_fraction=0.0, noise=0.3):
    rng = np.random.default_rng(seed)
    data, offsets, truth = [], [0], []
    for r in range(nprob):
        q = sc.quat_angle_axis(8.0 + 7.0 * r, rng.normal(size=3)); t = rng.uniform(-1.5, 1.5, 3)
        rows, inl = sc.rig_rows(rng, 90 + 25 * r, 1 + r % 4, q, t, outlier_fraction=outlier_fraction, pixel_noise=noise)
        data.append(rows); offsets.append(offsets[-1] + len(rows)); truth.append((q, t, inl))
    return np.concatenate(data), np.array(offsets, dtype=np.int64), truth


@pytest.mark.parametrize("rtype,use_mle", [(0, 0), (0, 1), (1, 0), (2, 0)])
def test_rigid_transformation_follows_the_oracle_bit_for_bit(rtype, use_mle):
    data, offsets, truth = _batch(6, 3 + rtype)
    p = ransac.RansacParameters(); p.error_thresh = 2.0 ** 2; p.min_iterations = 120; p.max_iterations = 400
    p.failure_probability = 1e-3; p.seed = 17; p.use_mle = bool(use_mle)
    pc0 = p.to_c(); pc0.ransac_type = rtype
    res = ransac.estimate_batch(EST, data, offsets, pc0)
    for i in range(6):
        sl = slice(offsets[i], offsets[i + 1])
        pc = p.to_c(); pc.seed = 17 + i; pc.ransac_type = rtype
        o = ol.ransac_estimate(15, data[sl], pc)
        assert bool(o["success"]) == bool(res["success"][i]) and o["success"]
        assert o["num_iterations"] == res["num_iterations"][i], i
        assert np.array_equal(o["inlier_mask"], res["inlier_mask"][sl]), i
        assert np.array_equal(o["model"][:12], res["models"][i][:12]), i
        q, t, inl = truth[i]
        R = res["models"][i][:9].reshape(3, 3)
        assert np.abs(R - sc.quat_to_rot(q)).max() < 5e-2 and np.abs(res["models"][i][9:12] - t).max() < 0.25
        if rtype != 2:
            assert res["inlier_mask"][sl].mean() > 0.8


def test_cost_parameters_carry_across_the_rounds_of_a_call():
    """1300 iterations = three rounds of the batch loop (512 + 1024-capped): hypothesis k is solved from the samples 0 .. k, so a
    lost or re-zeroed running sum at a round boundary would change every later model."""
    data, offsets, truth = _batch(3, 29, outlier_fraction=0.1)
    p = ransac.RansacParameters(); p.error_thresh = 3.0 ** 2; p.min_iterations = 1300; p.max_iterations = 1300; p.seed = 2
    res = ransac.estimate_batch(EST, data, offsets, p.to_c())
    for i in range(3):
        sl = slice(offsets[i], offsets[i + 1])
        pc = p.to_c(); pc.seed = 2 + i
        o = ol.ransac_estimate(15, data[sl], pc, trace_capacity=20000)
        assert o["num_iterations"] == res["num_iterations"][i] == 1300
        assert np.array_equal(o["inlier_mask"], res["inlier_mask"][sl]) and np.array_equal(o["model"][:12], res["models"][i][:12])
        assert o["trace"][0].max() > 1100      # models were still being produced in the last round


def test_python_mirror_central_and_rig_overloads():
    rng = np.random.default_rng(8)
    q = sc.quat_angle_axis(12.0, (1.0, 0.2, -0.8)); t = np.array([-1.3, 2.1, 0.5])      # estimate_rigid_transformation_2d_3d_test.cc:336-340
    R = sc.quat_to_rot(q)
    X = np.stack([rng.uniform(-2, 2, 100), rng.uniform(-2, 2, 100), rng.uniform(6, 10, 100)], axis=1)                 # :91-95
    Y = X @ R.T + t
    corr = np.concatenate([Y[:, :2] / Y[:, 2:3], X], axis=1)
    p = ransac.RansacParameters(); p.error_thresh = (0.5 / 800.0) ** 2; p.failure_probability = 1e-3; p.max_iterations = 300; p.use_mle = True
    ok, rt, s = ransac.EstimateRigidTransformation2D3D(p, ransac.RansacType.RANSAC, corr)
    assert ok and np.abs(rt.rotation - R).max() < 1e-6 and np.abs(rt.translation - t).max() < 1e-4 and len(s.inliers) == 100   # kPoseTolerance 1e-4 (:333)
    rows, inl = sc.rig_rows(rng, 150, 3, q, t, pixel_noise=0.0)
    p.error_thresh = 1.0
    ok, rt, s = ransac.EstimateRigidTransformation2D3D(p, ransac.RansacType.RANSAC, rows)
    assert ok and np.abs(rt.rotation - R).max() < 1e-6 and np.abs(rt.translation - t).max() < 1e-4 and len(s.inliers) == 150
