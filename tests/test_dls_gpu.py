"""GPU: DLS-PnP through the C-ABI (theia_hip_dls_pnp, estimator THEIA_EST_ABSOLUTE_POSE_DLS) against the reference's
scenes and tolerances (dls_pnp_test.cc), an implementation-independent optimality check, and the oracle.  Since round 4
the device follows the oracle's (= the reference's) elimination route -- dense partial-pivot LU of the 93 x 93 block in the
generated table's row / column order (oracle/dls_oracle.h, dls_layout.h) -- operation for operation, so solver outputs,
inlier sets, iteration counts and elected models are compared BIT FOR BIT."""
import numpy as np
import pytest

from pytheiasfm_amd import ransac, synth
from tests import dls_scenes as sc
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_reference_scenes_as_one_batch():
    scenes = sc.scenes()
    feats = [sc.project(w, q, t, noise, seed=len(w)) for (_, w, q, t, noise, *_r) in scenes]
    ns, quats, ts = ransac.DlsPnp(feats, [s[1] for s in scenes])
    for k, (name, world, q, t, noise, max_reproj, max_rot, max_trans) in enumerate(scenes):
        sc.check_solutions(name, world, feats[k], q, t, quats[k, :ns[k]], ts[k, :ns[k]], max_reproj, max_rot, max_trans)


def test_single_problem_form_and_argument_checks():
    name, world, q, t, noise, max_reproj, max_rot, max_trans = sc.scenes()[0]
    feat = sc.project(world, q, t, 0.0, seed=0)
    ok, quats, ts = ransac.DlsPnp(feat, world)
    assert ok
    sc.check_solutions(name, world, feat, q, t, quats, ts, max_reproj, max_rot, max_trans)
    with pytest.raises(Exception):
        ransac.DlsPnp(feat[:2], world[:2])          # CHECK_GE(feature_position.size(), 3)


def test_solutions_are_stationary_points_of_the_cost():
    rng = np.random.default_rng(12)
    feats, worlds = [], []
    for k in range(24):
        n = [3, 4, 6, 20][k % 4]
        qq = rng.normal(size=4); qq /= np.linalg.norm(qq)
        cam = np.c_[rng.uniform(-1, 1, (n, 2)), rng.uniform(2, 6, n)]
        t = rng.normal(size=3)
        worlds.append((cam - t) @ sc.quat_to_rot(qq))
        feats.append(cam[:, :2] / cam[:, 2:3] + rng.normal(scale=1e-3, size=(n, 2)))
    ns, quats, ts = ransac.DlsPnp(feats, worlds)
    assert (ns > 0).all()
    checked = 0
    for k in range(24):
        for i in range(ns[k]):
            qs = quats[k, i]
            s = -qs[1:] / qs[0]
            g, c = sc.dls_cost_gradient(feats[k], worlds[k], s)
            # J' is a quartic: scale the gradient bound with its size at the solution
            assert np.abs(g).max() <= 1e-4 * max(1.0, c) * (1.0 + s @ s) ** 2, (k, i, g, c)
            checked += 1
    assert checked >= 24


def test_bitwise_equal_to_the_oracle_on_problems_of_any_size():
    rng = np.random.default_rng(77)
    feats, worlds = [], []
    for k in range(48):
        n = [3, 4, 8, 30, 100, 3][k % 6]
        qq = rng.normal(size=4); qq /= np.linalg.norm(qq)
        cam = np.c_[rng.uniform(-1, 1, (n, 2)), rng.uniform(2, 6, n)]
        t = rng.normal(size=3)
        worlds.append((cam - t) @ sc.quat_to_rot(qq))
        feats.append(cam[:, :2] / cam[:, 2:3] + (rng.normal(scale=2e-3, size=(n, 2)) if k % 2 else 0.0))
    ns, quats, ts = ransac.DlsPnp(feats, worlds)
    for k in range(48):
        qo, to = ol.dls_pnp(feats[k], worlds[k], call_index=k)
        assert ns[k] == len(qo), (k, ns[k], len(qo))
        assert np.array_equal(quats[k, :ns[k]], qo) and np.array_equal(ts[k, :ns[k]], to), k
    assert (ns > 0).sum() >= 40


@pytest.mark.parametrize("ransac_type,use_mle,use_lo", [(0, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 0), (2, 0, 0)])
def test_ransac_with_the_dls_estimator_matches_the_oracle(ransac_type, use_mle, use_lo):
    data, offsets, _ = synth.synth_ransac_v1(5, 250, "absolute", seed=0x5AC50311 + ransac_type)
    prm = ransac.RansacParameters(); prm.error_thresh = (2.0 / 1000.0) ** 2; prm.min_iterations = 96; prm.max_iterations = 96
    prm.seed = 31; prm.use_mle = use_mle; prm.use_lo = use_lo; prm.lo_start_iterations = 10; prm.ransac_type = ransac_type
    res = ransac.estimate_batch(ransac.EST_ABS_DLS, data, offsets, prm)
    for i in range(5):
        pc = prm.to_c(); pc.seed = prm.seed + i
        o = ol.ransac_estimate(ransac.EST_ABS_DLS, data[offsets[i]:offsets[i + 1]], pc)
        assert o["num_iterations"] == res["num_iterations"][i]
        gm = res["inlier_mask"][offsets[i]:offsets[i + 1]]
        assert np.array_equal(o["inlier_mask"], gm), f"inlier set differs on problem {i}"
        if use_lo:   # the refinement is the batched single-view LM (two math libraries' sin / cos): 1e-9, DESIGN.md 2
            assert np.abs(res["models"][i][:12] - o["model"][:12]).max() < 1e-9
        else:
            assert np.array_equal(res["models"][i][:12], o["model"][:12]), f"model differs on problem {i}"


def test_adaptive_iteration_count_with_dls():
    data, offsets, _ = synth.synth_ransac_v1(4, 300, "absolute", seed=0x5AC50321, inlier_lo=0.6, inlier_hi=0.8)
    prm = ransac.RansacParameters(); prm.error_thresh = (2.0 / 1000.0) ** 2; prm.min_iterations = 10; prm.max_iterations = 2000; prm.seed = 5
    res = ransac.estimate_batch(ransac.EST_ABS_DLS, data, offsets, prm)
    for i in range(4):
        pc = prm.to_c(); pc.seed = prm.seed + i
        o = ol.ransac_estimate(ransac.EST_ABS_DLS, data[offsets[i]:offsets[i + 1]], pc)
        # the adaptive bound reacts to every inlier count on the way: equal only because every hypothesis is
        assert int(o["num_iterations"]) == int(res["num_iterations"][i])
        assert int(o["num_inliers"]) == int(res["num_inliers"][i])
        assert np.array_equal(o["inlier_mask"], res["inlier_mask"][offsets[i]:offsets[i + 1]])
        assert res["success"][i]
