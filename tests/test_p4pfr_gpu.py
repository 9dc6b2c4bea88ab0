"""GPU: THEIA_EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE (P4Pfr) and the directly bound solver through the C-ABI against the
oracle -- solutions, hypotheses, inlier sets, iteration counts and the elected model BIT-IDENTICAL (csrc/p4pfr_kernels.hip keeps
the oracle's operation order; the "random rotation" draws come out of the sampler's stream on the host, as in the reference)."""
import numpy as np
import pytest

from pytheiasfm_amd import ransac
from tests import oracle_lib as ol
from tests import p4pfr_scenes as sc

pytestmark = pytest.mark.gpu
EST = ransac.EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE
META = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=100.0, max_focal_length=2000.0, min_radial_distortion=-1e-9,
                                                         max_radial_distortion=-1e-5)


def _minimal_problems(num, seed):
    rng = np.random.default_rng(seed)
    F = np.zeros((num, 4, 2)); W = np.zeros((num, 4, 3))
    for i in range(num):
        R = sc.angle_axis(rng.uniform(0, 25), rng.normal(size=3)); t = rng.uniform(-1, 1, 3) * [1.0, 1.0, 0.3]
        f = rng.uniform(400, 1800); k = -10.0 ** rng.uniform(-8.5, -6.5)
        for j in range(4):
            X = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(4, 10)])
            W[i, j] = X; F[i, j] = sc.project(R, t, X, f, k)
    return F, W


def test_direct_solver_equals_the_oracle_bit_for_bit():
    num = 300
    F, W = _minimal_problems(num, 11)
    draws = np.random.default_rng(3).uniform(-0.5, 0.5, (num, 3))
    meta = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=0.0, max_focal_length=1e5, min_radial_distortion=0.0,
                                                             max_radial_distortion=-1.0)
    ns, M, nsolver = ransac.FourPointsPoseFocalLengthRadialDistortion(F, W, meta, rotation_draws=draws)
    total = 0
    for i in range(num):
        o = ol.p4pfr_solve(F[i], W[i], draws[i], meta.limits())
        assert len(o) == ns[i], i
        assert np.array_equal(o, M[i, :ns[i]]), i
        total += ns[i]
    assert total > 2 * num     # several real solutions per problem
    assert np.all(nsolver >= ns)                                     # the range tests only remove solutions
    # rotation_draws = None: the draws of the first calls of a fresh process (std::mt19937(42))
    d42 = ol.mt_randdouble_stream(42, 3 * 50, -0.5, 0.5).reshape(50, 3)
    ns2, M2, _ = ransac.FourPointsPoseFocalLengthRadialDistortion(F[:50], W[:50], meta)
    for i in range(50):
        assert np.array_equal(ol.p4pfr_solve(F[i], W[i], d42[i], meta.limits()), M2[i, :ns2[i]]), i


def test_single_problem_binding_on_the_reference_scene():
    f, W, R, t = sc.solver_scene("basic")
    meta = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=0.0, max_focal_length=2000.0, min_radial_distortion=-1e-10,
                                                             max_radial_distortion=-1e-5)
    ok, Rs, ts, ks, fs = ransac.FourPointsPoseFocalLengthRadialDistortion(f, W, meta)
    assert ok
    j = int(np.argmin([np.abs(Rj - R).max() for Rj in Rs]))
    assert np.abs(Rs[j] - R).max() < 1e-8 and np.abs(ts[j] - t).max() < 1e-7 and abs(fs[j] - sc.FOCAL) < 1e-5 and abs(ks[j] / sc.DISTORTION - 1) < 1e-3


def test_success_counts_the_solver_solutions_before_the_range_tests():
    """four_point_focal_length_radial_distortion.cc:287 returns valid_solutions.size() > 0 -- the solver's count BEFORE the
    focal-length / distortion range tests -- so limits that reject every solution still report success, with empty outputs
    (ADVICE r4); the count after the tests stays what the batched form returns."""
    f, W, R, t = sc.solver_scene("basic")
    wide = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=0.0, max_focal_length=2000.0, min_radial_distortion=-1e-10,
                                                             max_radial_distortion=-1e-5)
    none = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=1e-6, max_focal_length=2e-6, min_radial_distortion=-1e-10,
                                                             max_radial_distortion=-1e-5)
    ok, Rs, ts, ks, fs = ransac.FourPointsPoseFocalLengthRadialDistortion(f, W, wide)
    assert ok and len(Rs) > 0
    ok2, Rs2, ts2, ks2, fs2 = ransac.FourPointsPoseFocalLengthRadialDistortion(f, W, none)
    assert ok2 and Rs2 == [] and ts2 == [] and ks2 == [] and fs2 == []
    ns, M, nsol = ransac.FourPointsPoseFocalLengthRadialDistortion(f[None], W[None], none)
    assert int(ns[0]) == 0 and not M.any() and int(nsol[0]) > 0     # the batched form carries the reference's success flag too


def _batch(nprob, seed, ratio=0.75, noise=0.5):
    rng = np.random.default_rng(seed)
    data, offsets, truth = [], [0], []
    for r in range(nprob):
        R = sc.angle_axis(5.0 + 4.0 * r, rng.normal(size=3)); t = rng.uniform(-1, 1, 3) * [1.0, 1.0, 0.2]
        rows = sc.estimator_scene(rng, R, t, ratio, noise, n=60 + 15 * r)
        data.append(rows); offsets.append(offsets[-1] + len(rows)); truth.append((R, t))
    return np.concatenate(data), np.array(offsets, dtype=np.int64), truth


@pytest.mark.parametrize("rtype,use_mle,first_call", [(0, 0, 0), (0, 1, 1), (1, 0, 0), (2, 0, 1)])
def test_radial_dist_absolute_pose_follows_the_oracle_bit_for_bit(rtype, use_mle, first_call):
    data, offsets, truth = _batch(6, 3 + rtype)
    p = ransac.RansacParameters(); p.error_thresh = 2.0 ** 2; p.min_iterations = 150; p.max_iterations = 400
    p.failure_probability = 1e-3; p.seed = 17; p.use_mle = bool(use_mle)
    pc0 = p.to_c(); pc0.ransac_type = rtype
    ep = np.concatenate([META.limits(), [float(first_call)]])
    res = ransac.estimate_batch(EST, data, offsets, pc0, ep)
    ol.set_estimator_params(ep)
    try:
        for i in range(6):
            sl = slice(offsets[i], offsets[i + 1])
            pc = p.to_c(); pc.seed = 17 + i; pc.ransac_type = rtype
            o = ol.ransac_estimate(16, data[sl], pc)
            assert bool(o["success"]) == bool(res["success"][i]) and o["success"]
            assert o["num_iterations"] == res["num_iterations"][i], i
            assert np.array_equal(o["inlier_mask"], res["inlier_mask"][sl]), i
            assert np.array_equal(o["model"][:14], res["models"][i][:14]), i
            R, t = truth[i]
            m = res["models"][i]
            assert sc.arrays_equal_up_to_scale(R, m[:9].reshape(3, 3), 1e-2) and abs(m[12] - sc.FOCAL) < 0.1 * sc.FOCAL
    finally:
        ol.set_estimator_params([0.0] * 5)


def test_draws_continue_across_the_rounds_of_a_call():
    """1300 iterations = three rounds of the batch loop: the generator that deals samples and draws must run on across them."""
    data, offsets, truth = _batch(3, 29, ratio=0.6)
    p = ransac.RansacParameters(); p.error_thresh = 1.5 ** 2; p.min_iterations = 1300; p.max_iterations = 1300; p.seed = 2
    ep = np.concatenate([META.limits(), [1.0]])
    res = ransac.estimate_batch(EST, data, offsets, p.to_c(), ep)
    ol.set_estimator_params(ep)
    try:
        for i in range(3):
            sl = slice(offsets[i], offsets[i + 1])
            pc = p.to_c(); pc.seed = 2 + i
            o = ol.ransac_estimate(16, data[sl], pc, trace_capacity=40000)
            assert o["num_iterations"] == res["num_iterations"][i] == 1300
            assert np.array_equal(o["inlier_mask"], res["inlier_mask"][sl]) and np.array_equal(o["model"][:14], res["models"][i][:14])
            assert o["trace"][0].max() > 1100
    finally:
        ol.set_estimator_params([0.0] * 5)


@pytest.mark.parametrize("mode", sc.MODES, ids=[m[0] for m in sc.MODES])
def test_python_mirror_on_the_reference_estimator_scenes(mode):
    name, ratio, noise, tol, fields = mode
    rng = np.random.default_rng(640 + len(name))
    rots = sc.ROTATIONS_A if ratio == 1.0 else [np.eye(3), sc.angle_axis(15.0 * rng.uniform(0.2, 1.0), rng.normal(size=3))]
    k = 0
    for R in rots:
        for pos in sc.POSITIONS:
            rows = sc.estimator_scene(rng, R, pos, ratio, noise)
            p = ransac.RansacParameters(); p.error_thresh = 1.0; p.use_mle = True; p.failure_probability = 0.001
            p.min_iterations = fields.get("min_iterations", 100); p.seed = 64 + k; k += 1
            if "max_iterations" in fields:
                p.max_iterations = fields["max_iterations"]
            ok, pose, s = ransac.EstimateRadialDistUncalibratedAbsolutePose(p, ransac.RansacType.RANSAC, rows, META)
            assert ok
            assert sc.arrays_equal_up_to_scale(R, pose.rotation, tol) and sc.arrays_equal_up_to_scale(pos, pose.translation, 2 * tol)
            assert abs(pose.focal_length - sc.FOCAL) < 0.05 * sc.FOCAL
            if noise == 0.0:
                assert abs(pose.radial_distortion - sc.DISTORTION) < 0.1 * abs(sc.DISTORTION)


def test_metadata_checks_follow_the_reference():
    f, W, R, t = sc.solver_scene("basic")
    bad = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=500.0, max_focal_length=100.0)
    with pytest.raises(Exception):
        ransac.FourPointsPoseFocalLengthRadialDistortion(f, W, bad)
    bad2 = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_radial_distortion=1e-9)
    with pytest.raises(Exception):
        ransac.FourPointsPoseFocalLengthRadialDistortion(f, W, bad2)


def test_c5_shape_inlier_sets_against_oracle():
    """BASELINE configs[4]'s shape for this estimator -- 2000 correspondences per pair, exactly 4096 hypotheses, InlierSupport, 16
    pairs of the bench leg's first chunk (synth_ransac_v1 seen by a camera of focal length 1000 with distortion -1e-7, the world
    shifted so that t_z >= 0) -- against the oracle's sequential loop under the same per-pair seeds: inlier masks, iteration counts
    and the elected model bit for bit."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from pytheiasfm_amd import synth
    NP, CORR, HYPS = 16, 2000, 4096
    data, offsets, truth = synth.synth_ransac_v1(NP, CORR, "absolute", seed=0x5AC50005)
    data = ransac.radial_dist_correspondence_rows(ransac.shift_world_along_optical_axis(data, offsets, truth["R"], 2.0), 1000.0, -1e-7)
    ep = np.concatenate([META.limits(), [0.0]])
    p = ransac.RansacParameters(); p.error_thresh = 4.0 ** 2; p.min_iterations = HYPS; p.max_iterations = HYPS; p.seed = 1
    res = ransac.estimate_batch(EST, data, offsets, p, ep)
    assert np.all(res["num_iterations"] == HYPS) and res["hypotheses_evaluated"] == NP * HYPS
    ol.set_estimator_params(ep)
    try:
        def one(i):
            pc = p.to_c(); pc.seed = 1 + i
            return ol.ransac_estimate(16, data[offsets[i]:offsets[i + 1]], pc)
        with ThreadPoolExecutor(max_workers=min(NP, os.cpu_count() or 1)) as ex:
            ora = list(ex.map(one, range(NP)))
    finally:
        ol.set_estimator_params([0.0] * 5)
    for i in range(NP):
        sl = slice(offsets[i], offsets[i + 1])
        assert ora[i]["num_iterations"] == res["num_iterations"][i] == HYPS
        assert np.array_equal(ora[i]["inlier_mask"], res["inlier_mask"][sl]), i
        assert np.array_equal(ora[i]["model"][:14], res["models"][i][:14]), i
        # plausibility: the best of 4096 noisy minimal samples explains most of the planted inliers, with a focal length near 1000
        assert res["num_inliers"][i] > 0.5 * truth["inlier"][i].sum() and abs(res["models"][i][12] - 1000.0) < 60.0
