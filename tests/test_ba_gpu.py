"""GPU: parity of the HIP BA path (through the C-ABI) against the CPU oracle."""
import os

import numpy as np
import pytest

from pytheiasfm_amd import _capi as capi, ba, sfm, synth
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def both_options(**kw):
    o, oo = ba.default_options(), ol.default_options()
    for k, v in kw.items():
        setattr(o, k, v); setattr(oo, k, v)
    return o, oo


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("cfg", ["C1", "mixed"])
@pytest.mark.parametrize("manifold", [1, 0])
def test_residuals_and_jacobians_match_jet_oracle(cfg, manifold):
    p = synth.ba_config("C1") if cfg == "C1" else synth.synth_ba_v1(24, 1500, seed=11, mixed_models=True)
    o, oo = both_options(use_homogeneous_point_parametrization=manifold)
    with ba.BaHandle(p.copy(), o) as h:
        cost, r, jc, jp, valid = h.evaluate()
    ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
    assert ok == 1 and valid.all()
    assert abs(cost - ocost) <= 1e-13 * ocost
    assert np.abs(r - orr).max() <= 1e-10            # pixels; analytic vs Jet derivatives below
    assert rel(jc, ojc) <= 1e-12 and rel(jp, ojp) <= 1e-12


@pytest.mark.parametrize("loss", [1, 2, 3, 4, 5, 6])
def test_robust_losses_match_oracle(loss):
    p = synth.synth_ba_v1(10, 300, seed=21, pixel_noise=2.0)
    o, oo = both_options(loss_function_type=loss, robust_loss_width=1.5)
    with ba.BaHandle(p.copy(), o) as h:
        cost, r, jc, jp, _ = h.evaluate()
    ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
    assert abs(cost - ocost) <= 1e-12 * ocost
    assert rel(r, orr) <= 1e-11 and rel(jc, ojc) <= 1e-11 and rel(jp, ojp) <= 1e-11


@pytest.mark.parametrize("manifold", [1, 0])
@pytest.mark.parametrize("radius", [1e4, 37.5])
def test_reduced_camera_system_matches_oracle(manifold, radius):
    p = synth.synth_ba_v1(30, 2500, seed=31, mixed_models=True)
    o, oo = both_options(use_homogeneous_point_parametrization=manifold)
    with ba.BaHandle(p.copy(), o) as h:
        S, rhs = h.reduced_system(radius)
    So, ro = ol.reduced_system(p, oo, radius)
    assert S.shape == So.shape
    assert rel(S, So) <= 1e-10 and rel(rhs, ro) <= 1e-10   # summation-order noise of FP64 atomics


def test_lm_trajectory_matches_oracle_c1():
    p = synth.ba_config("C1")
    o, oo = both_options()
    pg, po = p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    so, tro = ol.solve(po, oo)
    assert s.success == so.success == 1 and s.termination_type == so.termination_type
    assert s.num_iterations == so.num_iterations and tr.size == tro.size
    assert np.array_equal(tr.accepted, tro.accepted)
    assert rel(tr.cost, tro.cost) <= 1e-9 and rel(tr.radius, tro.radius) <= 1e-9
    assert np.abs(tr.gradient_max_norm - tro.gradient_max_norm).max() <= 1e-6 * tro.gradient_max_norm.max()
    assert rel(tr.step_norm, tro.step_norm) <= 1e-6
    assert abs(s.initial_cost - so.initial_cost) <= 1e-12 * so.initial_cost
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-8 and np.abs(pg.points - po.points).max() <= 1e-8


@pytest.mark.parametrize("mixed", [False, True])
def test_gauge_fixed_parameters_within_1e6_relative(mixed):
    """north_star tolerance: point/pose parameters within 1e-6 relative on a
    gauge-fixed problem run to tight tolerances."""
    p = synth.synth_ba_v1(40, 4000, seed=41, mixed_models=mixed, fix_gauge=True)
    kw = dict(function_tolerance=1e-14, gradient_tolerance=1e-12, parameter_tolerance=1e-14, max_num_iterations=60)
    o, oo = both_options(**kw)
    pg, po = p.copy(), p.copy()
    s, _ = ba.solve(pg, o)
    so, _ = ol.solve(po, oo)
    assert s.success and so.success
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert rel(pg.cam_ext, po.cam_ext) <= 1e-6
    xg = pg.points[:, :3] / pg.points[:, 3:4]; xo = po.points[:, :3] / po.points[:, 3:4]
    assert rel(xg, xo) <= 1e-6
    assert np.array_equal(pg.cam_ext[[0, 20]], p.cam_ext[[0, 20]])   # constant views untouched


def test_constant_blocks_fixed_cost_and_masks():
    p = synth.synth_ba_v1(12, 400, seed=51)
    p.cam_const = np.array([3, 0, 1, 2, 0, 0, 3, 0, 0, 0, 4, 0], np.uint8)
    pc = np.zeros(400, np.uint8); pc[::4] = 1
    p.point_const = pc
    o, oo = both_options()
    pg, po = p.copy(), p.copy()
    with ba.BaHandle(pg, o) as h:
        cost, r, jc, jp, _ = h.evaluate()
        ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
        assert abs(cost - ocost) <= 1e-13 * ocost                     # includes the fixed cost
        assert np.abs(r - orr).max() <= 1e-10 and rel(jc, ojc) <= 1e-12 and rel(jp, ojp) <= 1e-12
        s, tr = h.run()
        h.download(pg)
    so, tro = ol.solve(po, oo)
    assert s.num_iterations == so.num_iterations and abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.array_equal(pg.cam_ext[[0, 6]], p.cam_ext[[0, 6]])       # whole block constant
    assert np.array_equal(pg.cam_ext[2, :3], p.cam_ext[2, :3]) and not np.array_equal(pg.cam_ext[2, 3:], p.cam_ext[2, 3:])
    assert np.array_equal(pg.cam_ext[3, 3:], p.cam_ext[3, 3:]) and not np.array_equal(pg.cam_ext[3, :3], p.cam_ext[3, :3])
    assert pg.cam_ext[10, 2] == p.cam_ext[10, 2]                        # tz constant
    assert np.array_equal(pg.points[pc == 1], p.points[pc == 1])
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-8 and np.abs(pg.points - po.points).max() <= 1e-8


def test_views_against_constant_points_straddling_tile_boundaries():
    """BundleAdjustViews-shaped problem: 40 variable cameras, every point constant.  The cameras at reduced index 10, 21,
    31, ... have their 6 rows across a 64-row tile boundary of the reduced system; their diagonal block then lives in
    three tiles, and no variable track marks any of them.  The tile-sparse plan must equal the dense schedule and the
    oracle (every camera is an independent 6 x 6 problem)."""
    p = synth.synth_ba_v1(40, 1500, seed=77)
    p.point_const = np.ones(1500, np.uint8)
    o, oo = both_options(max_num_iterations=6)
    pg, pd_, po = p.copy(), p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    os.environ["THEIA_HIP_DENSE_CHOLESKY"] = "1"
    try:
        sd, trd = ba.solve(pd_, o)
    finally:
        del os.environ["THEIA_HIP_DENSE_CHOLESKY"]
    so, tro = ol.solve(po, oo)
    assert s.num_iterations == sd.num_iterations == so.num_iterations
    assert rel(tr.cost[: tr.size], trd.cost[: trd.size]) < 1e-12 and np.array_equal(tr.accepted[: tr.size], trd.accepted[: trd.size])
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-8 and np.abs(pg.cam_ext - pd_.cam_ext).max() <= 1e-10
    assert np.array_equal(pg.points, p.points)
    with ba.BaHandle(p.copy(), o) as h:   # two reduced systems in a row: nothing may accumulate across the per-iteration clear
        S1, r1 = h.reduced_system(1e4)
        S2, r2 = h.reduced_system(1e4)
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert np.array_equal(S1, S2) and rel(S1, So) <= 1e-10 and rel(r1, ro) <= 1e-10


def test_reference_threshold_tests_on_gpu():
    """bundle_adjustment_test.cc thresholds through the Python mirror."""
    from tests.test_oracle_ba import _view_scene
    for noise in (0.0, 0.1):
        p = _view_scene(noise, 52)
        s, _ = ba.solve(p, ba.default_options())
        # the reference asserts the cost bound only (bundle_adjustment_test.cc:108-114); with noise-free pixels the
        # residuals are pure round-off (cost ~1e-25) and whether LM then stops by tolerance or by five invalid steps
        # depends on the last bit of the arithmetic
        assert 2.0 * s.final_cost / p.obs_uv.shape[0] < (1e-15 if noise == 0.0 else noise)
        assert s.success or noise == 0.0


def test_mirror_entry_points_partial_reconstruction():
    p = synth.synth_ba_v1(10, 300, seed=61)
    rec = sfm.Reconstruction.from_flat(p)
    before = rec.cam_ext.copy(), rec.points.copy()
    opts = sfm.BundleAdjustmentOptions()
    views, tracks = [0, 1, 2, 3], list(range(0, 150))
    summ = sfm.BundleAdjustPartialReconstruction(opts, views, tracks, rec)
    assert summ.success and summ.final_cost < summ.initial_cost
    assert np.array_equal(rec.cam_ext[4:], before[0][4:])               # frozen by AddTrack
    assert np.array_equal(rec.points[150:], before[1][150:])            # constant tracks
    assert not np.array_equal(rec.cam_ext[:4], before[0][:4])
    assert np.all(rec.inverse_depth[:150] > 0)                          # UpdateInverseDepth post-step
    flat = sfm._flatten(sfm.Reconstruction.from_flat(p), views, tracks)
    so, _ = ol.solve(flat, ol.default_options())
    assert abs(so.final_cost - summ.final_cost) <= 1e-9 * so.final_cost
    rec2 = sfm.Reconstruction.from_flat(p)
    s2 = sfm.BundleAdjustReconstruction(opts, rec2)
    assert s2.success and s2.final_cost < s2.initial_cost
    s3 = sfm.BundleAdjustView(sfm.Reconstruction.from_flat(p), opts, 3)
    assert s3.success


def test_mirror_partial_reconstruction_optimises_only_added_groups():
    """AddView marks the view's intrinsics group optimised (bundle_adjuster.cc:130-133);
    groups reached only through AddTrack stay constant (:442-455)."""
    p = synth.synth_ba_v1(12, 400, seed=62, num_groups=4)
    rec = sfm.Reconstruction.from_flat(p)
    k0 = rec.group_intrinsics.copy()
    opts = sfm.BundleAdjustmentOptions()
    opts.intrinsics_to_optimize = sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION
    views = [v for v in range(12) if rec.view_group[v] in (0, 2)]
    summ = sfm.BundleAdjustPartialReconstruction(opts, views, list(range(400)), rec)
    assert summ.success and summ.final_cost < summ.initial_cost
    assert np.array_equal(rec.group_intrinsics[[1, 3]], k0[[1, 3]])
    assert not np.array_equal(rec.group_intrinsics[[0, 2], 0], k0[[0, 2], 0])
    assert np.array_equal(rec.group_intrinsics[:, 1:5], k0[:, 1:5])


def test_mirror_independent_view_and_track_batches():
    """BundleAdjustView / BundleAdjustTrack lists as one batch equal the one-at-a-time mirror calls."""
    p = synth.synth_ba_v1(8, 150, seed=63)
    opts = sfm.BundleAdjustmentOptions(); opts.max_num_iterations = 10
    ra, rb = sfm.Reconstruction.from_flat(p), sfm.Reconstruction.from_flat(p)
    sa = sfm.BundleAdjustViewsIndependently(ra, opts, [1, 4, 6])
    for k, v in enumerate([1, 4, 6]):
        s1 = sfm.BundleAdjustView(sfm.Reconstruction.from_flat(p), opts, v)
        assert abs(sa[k].final_cost - s1.final_cost) <= 1e-8 * s1.final_cost and sa[k].final_cost < sa[k].initial_cost
    assert np.array_equal(ra.cam_ext[[0, 2, 3, 5, 7]], rb.cam_ext[[0, 2, 3, 5, 7]]) and not np.array_equal(ra.cam_ext[1], rb.cam_ext[1])
    tracks = [0, 3, 77, 149]
    st = sfm.BundleAdjustTracksIndependently(rb, opts, tracks)
    for k, t in enumerate(tracks):
        rc = sfm.Reconstruction.from_flat(p)
        s1 = sfm.BundleAdjustTrack(rc, opts, t)
        assert abs(st[k].final_cost - s1.final_cost) <= 1e-8 * max(s1.final_cost, 1e-300)
        assert np.abs(rb.points[t] - rc.points[t]).max() <= 1e-8
    assert np.array_equal(rb.points[1], p.points[1]) and np.all(rb.inverse_depth[tracks] > 0)


def test_covariance_blocks_match_jacobian_inverse():
    """*WithCov (bundle_adjustment.cc:288-386,420-499): ceres::Covariance of the block-diagonal problems
    = inverse of each block of J'J, J from the oracle's Jets at the solution; times 2 * cost / redundancy."""
    p = synth.synth_ba_v1(10, 200, seed=64, pixel_noise=0.7)
    opts = sfm.BundleAdjustmentOptions(); opts.max_num_iterations = 30
    rec = sfm.Reconstruction.from_flat(p)
    tracks = [3, 50, 199]
    summ, covs, factor = sfm.BundleAdjustTracksWithCov(rec, opts, tracks)
    assert summ.success and factor > 0
    nobs = sum(int(np.sum(p.obs_pt == t)) for t in tracks)
    assert abs(factor - 2.0 * summ.final_cost / (2 * nobs - 9)) <= 1e-12 * factor
    flat = sfm._flatten(rec, [], tracks)                      # the solved state
    ok, _, r, jc, jp = ol.evaluate(flat, ol.default_options())
    for t in tracks:
        J = jp[flat.obs_pt == t].reshape(-1, 3)
        ref = np.linalg.inv(J.T @ J) * factor
        assert np.abs(covs[t] - ref).max() <= 1e-8 * np.abs(ref).max()
        assert np.all(np.linalg.eigvalsh(covs[t]) > 0)
    s1, c1, f1 = sfm.BundleAdjustTrackWithCov(sfm.Reconstruction.from_flat(p), opts, 50)
    assert s1.success and c1.shape == (3, 3) and f1 > 0
    # views against constant tracks
    rec2 = sfm.Reconstruction.from_flat(p)
    views = [2, 7]
    sv, cv, fv = sfm.BundleAdjustViewsWithCov(rec2, opts, views)
    assert sv.success
    flat2 = sfm._flatten(rec2, views, [])
    ok, _, r2, jc2, jp2 = ol.evaluate(flat2, ol.default_options())
    for v in views:
        J = jc2[flat2.obs_cam == v].reshape(-1, 6)
        ref = np.linalg.inv(J.T @ J) * fv
        assert np.abs(cv[v] - ref).max() <= 1e-7 * np.abs(ref).max()
    s2, c2, f2 = sfm.BundleAdjustViewWithCov(sfm.Reconstruction.from_flat(p), opts, 7)
    assert s2.success and c2.shape == (6, 6) and np.all(np.linalg.eigvalsh(c2) > 0)
    # a problem that is not block diagonal has no covariance entry point
    with ba.BaHandle(p.copy(), ba.default_options()) as h:
        with pytest.raises(capi.TheiaHipError):
            h.covariance(points=True)


def test_outlier_track_sweep_matches_reference_rules():
    """SetOutlierTracksToUnestimated (set_outlier_tracks_to_unestimated.cc:64-139) through the mirror vs a
    plain numpy restatement: behind-camera views, mean squared reprojection error, triangulation angle."""
    p = synth.synth_ba_v1(12, 300, seed=65, mixed_models=True, pixel_noise=0.5)
    p.obs_uv[p.obs_pt == 7] += 40.0                       # gross reprojection error
    p.points[11, :3] = p.cam_ext[p.obs_cam[np.nonzero(p.obs_pt == 11)[0][0]], :3] - 5.0 * (p.points[11, :3] - p.cam_ext[p.obs_cam[np.nonzero(p.obs_pt == 11)[0][0]], :3])  # behind
    p.points[20, :3] *= 400.0                             # far away: rays nearly parallel
    rec = sfm.Reconstruction.from_flat(p)
    removed = sfm.SetOutlierTracksToUnestimated(range(300), 4.0, 1.0, rec)
    expect_bad = np.zeros(300, bool)
    for t in range(300):
        sel = np.nonzero(p.obs_pt == t)[0]
        X = p.points[t]
        errs, rays, behind = [], [], False
        for o in sel:
            c = p.obs_cam[o]
            g = p.cam_group[c]
            uv, _ = synth.project(p.group_model[g], p.intrinsics[g][None], p.cam_ext[c][None], X[None])
            Rc = synth.angle_axis_to_matrix(p.cam_ext[c, 3:6])
            depth = (Rc @ (X[:3] - X[3] * p.cam_ext[c, :3]))[2] / X[3]
            behind |= bool(depth < 0)
            errs.append(np.sum((uv[0] - p.obs_uv[o]) ** 2))
            ray = X[:3] / X[3] - p.cam_ext[c, :3]
            rays.append(ray / np.linalg.norm(ray))
        bad = behind or np.mean(errs) > 16.0
        if not bad:
            cosmin = min(float(rays[i] @ rays[j]) for i in range(len(rays)) for j in range(i + 1, len(rays)))
            bad = not (cosmin < np.cos(np.deg2rad(1.0)))
        expect_bad[t] = bad
    assert expect_bad[[7, 11, 20]].all() and removed == expect_bad.sum()
    assert np.array_equal(~rec.track_estimated, expect_bad)
    rec2 = sfm.Reconstruction.from_flat(p)   # the estimators' wrapper (incremental_reconstruction_estimator.cc:599-610)
    assert sfm.RemoveOutlierTracks(range(300), 4.0, 1.0, rec2) == removed and np.array_equal(rec2.track_estimated, rec.track_estimated)


def test_edge_cases_empty_invalid_and_errors():
    o = ba.default_options()
    empty = capi.FlatProblem(np.zeros((0, 6)), np.zeros((1, 7)), [0], np.zeros(0, np.int32), np.zeros((0, 4)),
                             np.zeros((0, 2)), np.zeros(0, np.int32), np.zeros(0, np.int32))
    s, _ = ba.solve(empty, o)
    assert s.success and s.initial_cost == 0.0 and s.num_iterations == 0
    # a point sitting on a camera centre: the functor returns false at x0 -> FAILURE (success = False)
    p = synth.synth_ba_v1(4, 20, seed=71)
    p.points[p.obs_pt[0], :3] = p.cam_ext[p.obs_cam[0], :3]; p.points[p.obs_pt[0], 3] = 1.0
    s, _ = ba.solve(p.copy(), o)
    so, _ = ol.solve(p.copy(), ol.default_options())
    assert s.success == so.success == 0 and s.termination_type == so.termination_type == 2
    # unsupported / invalid arguments fail loudly with the reference's conventions
    bad = synth.synth_ba_v1(4, 20, seed=72); bad.obs_cam[0] = 99
    with pytest.raises(capi.TheiaHipError):
        ba.solve(bad, o)
    o2 = ba.default_options(); o2.intrinsics_to_optimize = 0x40   # not an OptimizeIntrinsicsType bit
    with pytest.raises(capi.TheiaHipError):
        ba.solve(synth.synth_ba_v1(4, 20, seed=73), o2)
    unknown = synth.synth_ba_v1(4, 20, seed=74); unknown.group_model[:] = 9
    with pytest.raises(capi.TheiaHipError):
        ba.solve(unknown, o)


def test_full_size_c2_properties():
    """BASELINE configs[1] size: size-independent properties -- cost decreases
    monotonically over accepted steps, the reduced system is symmetric positive
    definite, re-running is bitwise reproducible in cost terms, and a second
    solve from the optimum stops immediately."""
    p = synth.ba_config("C2")
    o = ba.default_options()
    with ba.BaHandle(p.copy(), o) as h:
        s, tr = h.run()
        assert s.success and s.final_cost < 0.01 * s.initial_cost
        acc = tr.cost[tr.accepted == 1]
        assert np.all(np.diff(acc) < 0)
        q = h.download(p.copy())
        h.reset(p); s2, tr2 = h.run()
        assert s2.num_iterations == s.num_iterations and rel(tr2.cost, tr.cost) < 1e-12
    with ba.BaHandle(q, o) as h2:
        s3, _ = h2.run()
        assert s3.num_iterations <= 2 and abs(s3.final_cost - s.final_cost) <= 1e-6 * s.final_cost
    # mean squared residual at the optimum ~ pixel noise variance (0.5 px)
    n = p.obs_uv.shape[0]
    assert 0.15 < 2.0 * s.final_cost / (2 * n) < 0.30


def test_full_size_c4_properties():
    """BASELINE configs[3] size on one GPU (1000 views / 500k tracks / ~3 M observations, pinhole +
    double-sphere, n = 6000, 94 tiles): the same size-independent properties, plus the level-scheduled
    plan against the dense panel chain on the first iterations."""
    p = synth.ba_config("C4")
    o = ba.default_options(); o.max_num_iterations = 12
    with ba.BaHandle(p.copy(), o) as h:
        s, tr = h.run()
        assert s.success and s.final_cost < 0.01 * s.initial_cost
        acc = tr.cost[tr.accepted == 1]
        assert np.all(np.diff(acc) < 0) and len(acc) >= 5
        h.reset(p); s2, tr2 = h.run()
        assert s2.num_iterations == s.num_iterations and rel(tr2.cost, tr.cost) < 1e-12
    n = p.obs_uv.shape[0]
    assert 0.15 < 2.0 * s.final_cost / (2 * n) < 0.35
    o3 = ba.default_options(); o3.max_num_iterations = 3
    os.environ["THEIA_HIP_DENSE_CHOLESKY"] = "1"
    try:
        sd, trd = ba.solve(p.copy(), o3)
    finally:
        del os.environ["THEIA_HIP_DENSE_CHOLESKY"]
    assert rel(trd.cost[: trd.size], tr.cost[: trd.size]) < 1e-9 and np.array_equal(trd.accepted[: trd.size], tr.accepted[: trd.size])
    # the host-side plan (threaded structure passes, fused plan in fixed segments) does not depend on the number of
    # host threads: the serial build walks the same trajectory bit for bit
    os.environ["THEIA_HIP_HOST_THREADS"] = "1"
    try:
        s1, tr1 = ba.solve(p.copy(), o3)
    finally:
        del os.environ["THEIA_HIP_HOST_THREADS"]
    assert np.array_equal(tr1.cost[: tr1.size], tr.cost[: tr1.size]) and np.array_equal(tr1.step_norm[: tr1.size], tr.step_norm[: tr1.size])
    # ... nor on where the host staging lives: pinned blocks of the library cache, or plain memory when the host does
    # not let the process pin that much (THEIA_HIP_NO_PINNED forces the fallback)
    os.environ["THEIA_HIP_NO_PINNED"] = "1"
    try:
        s4, tr4 = ba.solve(p.copy(), o3)
    finally:
        del os.environ["THEIA_HIP_NO_PINNED"]
    assert np.array_equal(tr4.cost[: tr4.size], tr.cost[: tr4.size]) and np.array_equal(tr4.step_norm[: tr4.size], tr.step_norm[: tr4.size])


def test_observation_order_of_the_input_does_not_change_the_solve():
    """create() groups the input's observations by track: input that comes track by track (obs_pt non-decreasing) is walked in
    place, anything else goes through a counted scatter with atomic cursors and a per-track sort back into input order.  The
    same problem with its observations INTERLEAVED (view-major, the order BundleAdjustReconstruction adds them in; the order
    inside every track kept) must give the same plan and therefore the same trajectory bit for bit -- also with constant
    cameras / points (fixed-cost blocks) and with one host thread."""
    p = synth.ba_config("C2")     # 299 587 observations (> 262 144: the threaded passes run), tracks of <= 10: no track takes the
    assert np.all(np.diff(p.obs_pt) >= 0) and p.obs_uv.shape[0] > 262144   # slow path, whose atomics are not bit-reproducible
    p.cam_const = np.zeros(p.cam_ext.shape[0], np.uint8); p.cam_const[[0, 5]] = 3
    p.point_const = np.zeros(p.points.shape[0], np.uint8); p.point_const[::7] = 1
    within = np.arange(len(p.obs_pt)) - np.repeat(np.cumsum(np.bincount(p.obs_pt)) - np.bincount(p.obs_pt), np.bincount(p.obs_pt))
    order = np.lexsort((p.obs_pt, within))                          # all first observations, then all second ones, ...
    assert np.any(np.diff(p.obs_pt[order]) < 0)
    q = capi.FlatProblem(p.cam_ext.copy(), p.intrinsics.copy(), p.group_model, p.cam_group, p.points.copy(), p.obs_uv[order],
                         p.obs_cam[order], p.obs_pt[order], p.cam_const, p.group_const, p.point_const)
    o = ba.default_options(); o.max_num_iterations = 4
    pa, qa = p.copy(), q.copy()
    sa, ta = ba.solve(pa, o)
    sb, tb = ba.solve(qa, o)
    assert sa.success and ta.size == tb.size and ta.size >= 3
    assert np.array_equal(ta.cost[: ta.size], tb.cost[: tb.size]) and np.array_equal(ta.step_norm[: ta.size], tb.step_norm[: tb.size])
    assert np.array_equal(pa.cam_ext, qa.cam_ext) and np.array_equal(pa.points, qa.points)
    os.environ["THEIA_HIP_HOST_THREADS"] = "1"
    try:
        qc = q.copy(); sc, tc = ba.solve(qc, o)
    finally:
        del os.environ["THEIA_HIP_HOST_THREADS"]
    assert np.array_equal(tc.cost[: tc.size], ta.cost[: ta.size]) and np.array_equal(qc.points, pa.points)


@pytest.mark.parametrize("n", [1, 6, 24, 32, 33, 120, 121, 300, 1200])
def test_dense_cholesky_kernel_against_numpy(n):
    """K3 alone (FP64-MFMA blocked Cholesky + substitutions) vs numpy."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 8))
    A = M @ M.T + 1e-3 * np.eye(n)
    b = rng.standard_normal(n)
    x = ba.dense_spd_solve(np.tril(A), b)   # only the lower triangle is read
    xr = np.linalg.solve(A, b)
    assert np.abs(x - xr).max() <= 1e-9 * np.abs(xr).max() * max(1.0, np.linalg.cond(A) * 1e-6)
    with pytest.raises(capi.TheiaHipError):
        Abad = A.copy(); Abad[n // 2, n // 2] = -1.0
        ba.dense_spd_solve(np.tril(Abad), b)


CAMERA_MODEL_INTRINSICS = {
    0: [1000.0, 1.0, 0.0, 960.0, 540.0, -0.05, 0.01],
    1: [1000.0, 1.02, 0.2, 960.0, 540.0, -0.1, 0.02, 0.001, 0.001, -0.002],
    2: [600.0, 1.0, 0.1, 960.0, 540.0, 0.01, -0.002, 0.001, 0.0005],
    3: [800.0, 1.01, 960.0, 540.0, 0.9],
    4: [1000.0, 0.99, 960.0, 540.0, -1e-7],
    5: [600.0, 1.0, 0.0, 960.0, 540.0, -0.2, 0.55],
    6: [600.0, 1.0, 0.0, 960.0, 540.0, 0.6, 1.1],
    7: [80.0, 1.0, 0.1, 960.0, 540.0, 0.001, -0.0001],
}


@pytest.mark.parametrize("model", sorted(CAMERA_MODEL_INTRINSICS))
def test_every_camera_model_matches_jet_oracle(model):
    """A7: closed-form derivatives of all eight camera models vs the Jet oracle,
    plus one LM solve (trajectory parity)."""
    p = synth.synth_ba_v1(12, 500, seed=200 + model, num_groups=3)
    k = CAMERA_MODEL_INTRINSICS[model]
    p.group_model[:] = model
    p.intrinsics[:] = 0.0
    p.intrinsics[:, : len(k)] = k
    if model == 3:
        p.intrinsics[1, 4] = 5e-4      # small-omega Taylor branch of the FOV model
    # (inner iterations off: this LM leg walks a deliberately rough problem per model, where the 50-iteration inner
    # solves amplify round-off into the 5th digit; inner iterations have their own parity tests below)
    o, oo = both_options(use_inner_iterations=0)
    with ba.BaHandle(p.copy(), o) as h:
        cost, r, jc, jp, valid = h.evaluate()
    ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
    assert valid.all() == bool(ok)
    assert abs(cost - ocost) <= 1e-12 * ocost
    assert np.abs(r - orr).max() <= 1e-9 * max(1.0, np.abs(orr).max())
    assert rel(jc, ojc) <= 1e-11 and rel(jp, ojp) <= 1e-11
    o.max_num_iterations = oo.max_num_iterations = 6
    pg, po = p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    so, tro = ol.solve(po, oo)
    assert s.num_iterations == so.num_iterations and np.array_equal(tr.accepted, tro.accepted)
    fin = tro.cost < 1e300
    assert rel(tr.cost[fin], tro.cost[fin]) <= 1e-8


def _with_long_tracks(nv=80, nt=400, nlong=5, seed=301):
    """synth scene plus `nlong` tracks seen by EVERY camera (> 64 observations)."""
    p, cam_gt, pts_gt = synth.synth_ba_v1(nv, nt, seed=seed, return_truth=True)
    st = synth.Stream(seed, 99)
    i = np.arange(nlong)
    X = np.stack([st.uniform(3 * i) - 0.5, st.uniform(3 * i + 1) - 0.5, 0.4 * st.uniform(3 * i + 2) - 0.2, np.ones(nlong)], 1)
    oc = np.tile(np.arange(nv), nlong).astype(np.int32)
    op = (nt + np.repeat(i, nv)).astype(np.int32)
    uv, ok = synth.project(0, p.intrinsics[p.cam_group[oc]], cam_gt[oc], X[np.repeat(i, nv)])
    assert ok.all()
    uv = uv + 0.5 * np.stack([st.normal(2 * np.arange(len(oc)) + 1000), st.normal(2 * np.arange(len(oc)) + 1001)], 1)
    X0 = X.copy(); X0[:, :3] += 0.02 * np.stack([st.normal(3 * i + 500), st.normal(3 * i + 501), st.normal(3 * i + 502)], 1)
    return capi.FlatProblem(p.cam_ext, p.intrinsics, p.group_model, p.cam_group, np.vstack([p.points, X0]),
                            np.vstack([p.obs_uv, uv]), np.concatenate([p.obs_cam, oc]), np.concatenate([p.obs_pt, op]))


@pytest.mark.parametrize("manifold", [1, 0])
def test_long_tracks_slow_path_matches_oracle(manifold):
    """Tracks with more than 64 observations take the per-observation path."""
    p = _with_long_tracks()
    assert np.bincount(p.obs_pt).max() == 80
    o, oo = both_options(use_homogeneous_point_parametrization=manifold)
    with ba.BaHandle(p.copy(), o) as h:
        cost, r, jc, jp, valid = h.evaluate()
        S, rhs = h.reduced_system(1e4)
    ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
    assert abs(cost - ocost) <= 1e-12 * ocost and np.abs(r - orr).max() <= 1e-10
    assert rel(jc, ojc) <= 1e-12 and rel(jp, ojp) <= 1e-12
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert rel(S, So) <= 1e-10 and rel(rhs, ro) <= 1e-10
    pg, po = p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    so, tro = ol.solve(po, oo)
    assert s.success and s.num_iterations == so.num_iterations and np.array_equal(tr.accepted, tro.accepted)
    assert rel(tr.cost, tro.cost) <= 1e-9
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-7 and np.abs(pg.points - po.points).max() <= 1e-7


@pytest.mark.parametrize("intr", [0x11, 0x3f])
def test_long_tracks_with_intrinsics_match_oracle(intr):
    """The pipelines' default (FOCAL_LENGTH | RADIAL_DISTORTION, reconstruction_estimator_options.h:281-283) on a scene
    with tracks of more than 64 observations: the slow path carries the 16-wide camera-side block."""
    p = _with_long_tracks()
    assert np.bincount(p.obs_pt).max() == 80
    o, oo = both_options(intrinsics_to_optimize=intr)
    with ba.BaHandle(p.copy(), o) as h:
        S, rhs = h.reduced_system(1e4)
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert S.shape == So.shape and rel(S, So) <= 1e-10 and rel(rhs, ro) <= 1e-10
    pg, po = p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    so, tro = ol.solve(po, oo)
    assert s.success and s.num_iterations == so.num_iterations and np.array_equal(tr.accepted, tro.accepted)
    assert rel(tr.cost, tro.cost) <= 1e-9
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-7 and np.abs(pg.points - po.points).max() <= 1e-7
    assert np.abs(pg.intrinsics - po.intrinsics).max() <= 1e-6 * np.abs(po.intrinsics).max()


INTR_ALL = 0x3f
INTR_FOCAL_RADIAL = 0x11   # pipeline default (reconstruction_estimator_options.h:281-283)


@pytest.mark.parametrize("model", sorted(CAMERA_MODEL_INTRINSICS))
def test_intrinsics_jacobian_matches_jet_oracle(model):
    """A10: d residual / d intrinsics (all K parameters free) of every camera
    model, closed form vs Jets, and the reduced system with intrinsics blocks."""
    p = synth.synth_ba_v1(10, 300, seed=400 + model, num_groups=3)
    k = CAMERA_MODEL_INTRINSICS[model]
    p.group_model[:] = model
    p.intrinsics[:] = 0.0
    p.intrinsics[:, : len(k)] = k
    if model == 3:
        p.intrinsics[1, 4] = 5e-4
    o, oo = both_options(intrinsics_to_optimize=INTR_ALL)
    with ba.BaHandle(p.copy(), o) as h:
        cost, r, jc, jp, ji, valid = h.evaluate_ex()
        S, rhs = h.reduced_system(1e4)
    ok, ocost, orr, ojc, ojp, oji = ol.evaluate_ex(p, oo)
    assert abs(cost - ocost) <= 1e-12 * ocost
    assert rel(jc, ojc) <= 1e-11 and rel(jp, ojp) <= 1e-11
    assert np.abs(oji).max() > 0
    for q in range(10):   # per parameter: columns have very different magnitudes
        sc = np.abs(oji[:, :, q]).max()
        assert np.abs(ji[:, :, q] - oji[:, :, q]).max() <= 1e-10 * max(sc, 1e-300), (model, q)
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert S.shape == So.shape == (30 + 60, 30 + 60)
    assert rel(S, So) <= 1e-9 and rel(rhs, ro) <= 1e-9


@pytest.mark.parametrize("mixed", [False, True])
def test_intrinsics_optimisation_lm_parity_and_recovery(mixed):
    """LM with FOCAL_LENGTH | RADIAL_DISTORTION free (shared intrinsics groups):
    same trajectory as the oracle, and the perturbed focal lengths move back
    towards the truth on a gauge-fixed scene."""
    p = synth.synth_ba_v1(24, 1500, seed=431, num_groups=4, mixed_models=mixed, fix_gauge=True, pixel_noise=0.2)
    truth = p.intrinsics.copy()
    p.intrinsics[:, 0] *= 1.02
    o, oo = both_options(intrinsics_to_optimize=INTR_FOCAL_RADIAL, max_num_iterations=30)
    pg, po = p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    so, tro = ol.solve(po, oo)
    assert s.success and so.success and s.num_iterations == so.num_iterations
    assert np.array_equal(tr.accepted, tro.accepted)
    fin = tro.cost < 1e300
    assert rel(tr.cost[fin], tro.cost[fin]) <= 1e-8
    assert rel(pg.intrinsics, po.intrinsics) <= 1e-7 and np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-6
    if not mixed:   # with fisheye-type models focal trades against the distortion terms
        assert np.abs(pg.intrinsics[:, 0] / truth[:, 0] - 1.0).max() < 2e-3  # focal recovered (was 2 % off)
    assert s.final_cost < 0.05 * s.initial_cost
    assert np.array_equal(pg.intrinsics[:, 1:5], p.intrinsics[:, 1:5])        # aspect, skew, principal point frozen
    assert not np.array_equal(pg.intrinsics[:, 5:7], p.intrinsics[:, 5:7])    # distortion optimised


def test_intrinsics_group_constant_and_bounds():
    p = synth.synth_ba_v1(12, 400, seed=441, num_groups=3, mixed_models=True)
    p.group_const = np.array([0, 1, 0], np.uint8)
    p.intrinsics[1, 6] = 1.7     # double-sphere alpha outside [0, 1]; group constant -> untouched
    o, oo = both_options(intrinsics_to_optimize=INTR_ALL, max_num_iterations=8)
    pg, po = p.copy(), p.copy()
    s, _ = ba.solve(pg, o)
    so, _ = ol.solve(po, oo)
    assert s.num_iterations == so.num_iterations and abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert np.array_equal(pg.intrinsics[1], p.intrinsics[1])
    assert pg.intrinsics[0, 0] >= 1.0 and rel(pg.intrinsics, po.intrinsics) <= 1e-7


def test_level_scheduled_sparse_cholesky_matches_dense_schedule(monkeypatch):
    """K3: the nested-dissection / level-scheduled tile-sparse factorisation and
    the dense panel chain solve the same reduced systems (C2 size: 19 tiles on a
    ring; and a 4-dof-point problem with a long-track wrap-around)."""
    p = synth.ba_config("C2")
    o = ba.default_options(); o.max_num_iterations = 6
    runs = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("THEIA_HIP_DENSE_CHOLESKY", "1")
        q = p.copy()
        s, tr = ba.solve(q, o)
        runs.append((s, tr, q))
    (s0, t0, q0), (s1, t1, q1) = runs
    assert s0.num_iterations == s1.num_iterations and np.array_equal(t0.accepted, t1.accepted)
    assert rel(t0.cost, t1.cost) <= 1e-11
    assert np.abs(q0.cam_ext - q1.cam_ext).max() <= 1e-9 and np.abs(q0.points - q1.points).max() <= 1e-9


def _view_batch(num, seed, models=(0,), noise=0.3):
    """num independent localisation problems: 30..400 constant points each, perturbed pose."""
    st = synth.Stream(seed, 0)
    offs = [0]; uvs = []; Xs = []; cams = []; intr = []; mods = []; truth = []
    for k in range(num):
        n = 30 + int(370 * st.uniform(np.array([7 * k]))[0])
        i = np.arange(n)
        X = np.stack([6 * st.uniform(10000 * k + 3 * i) - 3, 6 * st.uniform(10000 * k + 3 * i + 1) - 3, 6 * st.uniform(10000 * k + 3 * i + 2) - 3], 1)
        model = models[k % len(models)]
        kk = np.zeros(10); kv = CAMERA_MODEL_INTRINSICS[model]; kk[: len(kv)] = kv
        pos = np.array([10.0, 0.3 * k - 1.0, 0.5]); z = -pos / np.linalg.norm(pos)
        xa = np.cross([0, 0, 1.0], z); xa /= np.linalg.norm(xa); R = np.stack([xa, np.cross(z, xa), z])
        ext = np.concatenate([pos, synth.matrix_to_angle_axis(R)])
        X4 = np.hstack([X, np.ones((n, 1))])
        truth.append(ext.copy())
        cams.append(ext + np.concatenate([0.05 * (2 * st.uniform(50 * k + np.arange(3) + 99) - 1), 0.01 * (2 * st.uniform(50 * k + np.arange(3) + 199) - 1)]))
        uvs.append((model, kk, ext, X4, noise * np.stack([st.normal(10000 * k + 2 * i + 5000), st.normal(10000 * k + 2 * i + 5001)], 1)))
        Xs.append(X4); intr.append(kk); mods.append(model); offs.append(offs[-1] + n)
    return offs, uvs, Xs, np.array(cams), np.array(intr), np.array(mods, np.int32), np.array(truth)


def test_views_batch_matches_per_problem_oracle():
    """theia_hip_ba_views_batch = N x BundleAdjustView: every problem follows the
    oracle's LM on the same one-camera problem (costs, iteration counts, pose)."""
    num = 24
    offs, obs, Xs, cams, intr, mods, truth = _view_batch(num, 0xBA7C, models=(0, 5, 2, 1))
    # the per-problem reference is BundleAdjustView, which switches the inner iterations off (bundle_adjustment.cc:225)
    o, oo = both_options(max_num_iterations=15, use_homogeneous_point_parametrization=0, use_inner_iterations=0)
    # observations: project with the ORACLE's evaluate on the true pose (residual = -uv at uv = 0)
    uv_all = []
    flats = []
    for k in range(num):
        model, kk, ext, X4, nz = obs[k]
        n = X4.shape[0]
        fp = capi.FlatProblem(ext[None].copy(), kk[None].copy(), np.array([model], np.int32), np.array([0], np.int32), X4.copy(),
                              np.zeros((n, 2)), np.zeros(n, np.int32), np.arange(n, dtype=np.int32), point_const=np.ones(n, np.uint8))
        ok, _, r, _, _ = ol.evaluate(fp, oo)
        uv = r + nz                      # r = projection - 0
        uv_all.append(uv)
        fp.obs_uv = uv.copy(); fp.cam_ext = cams[k][None].copy()
        flats.append(fp)
    cam_gpu = cams.copy()
    summ = ba.solve_views_batch(np.array(offs), np.vstack(uv_all), np.vstack(Xs), cam_gpu, intr, mods, o)
    for k in range(num):
        so, _ = ol.solve(flats[k], oo)
        s = summ[k]
        assert s.success == so.success and s.termination_type == so.termination_type, k
        assert s.num_iterations == so.num_iterations and s.num_successful_steps == so.num_successful_steps, k
        assert abs(s.initial_cost - so.initial_cost) <= 1e-10 * so.initial_cost
        assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
        assert np.abs(cam_gpu[k] - flats[k].cam_ext[0]).max() <= 1e-9
        assert np.abs(cam_gpu[k][:3] - truth[k][:3]).max() < 0.05   # localised


def test_views_batch_lo_options_and_masks():
    """The LO-RANSAC refinement settings (HUBER, 2 iterations, estimate_calibrated_absolute_pose.cc:124-129)
    and constant-position / constant-camera masks."""
    num = 8
    offs, obs, Xs, cams, intr, mods, truth = _view_batch(num, 0xBA7D, models=(0,))
    o, oo = both_options(max_num_iterations=2, use_homogeneous_point_parametrization=0, loss_function_type=1, robust_loss_width=0.5,
                         use_inner_iterations=0)
    uv_all, flats = [], []
    cc = np.array([0, 1, 2, 3, 0, 4, 0, 0], np.uint8)
    for k in range(num):
        model, kk, ext, X4, nz = obs[k]
        n = X4.shape[0]
        fp = capi.FlatProblem(ext[None].copy(), kk[None].copy(), np.array([model], np.int32), np.array([0], np.int32), X4.copy(),
                              np.zeros((n, 2)), np.zeros(n, np.int32), np.arange(n, dtype=np.int32), point_const=np.ones(n, np.uint8),
                              cam_const=np.array([cc[k]], np.uint8))
        ok, _, r, _, _ = ol.evaluate(fp, ol.default_options())
        uv = r + nz
        uv[::7] += 25.0                 # gross outliers for the Huber loss
        uv_all.append(uv)
        fp.obs_uv = uv.copy(); fp.cam_ext = cams[k][None].copy()
        flats.append(fp)
    cam_gpu = cams.copy()
    summ = ba.solve_views_batch(np.array(offs), np.vstack(uv_all), np.vstack(Xs), cam_gpu, intr, mods, o, cam_const=cc)
    for k in range(num):
        so, _ = ol.solve(flats[k], oo)
        s = summ[k]
        assert s.num_iterations == so.num_iterations and s.success == so.success, k
        assert abs(s.final_cost - so.final_cost) <= 1e-9 * max(so.final_cost, 1e-300), k
        assert np.abs(cam_gpu[k] - flats[k].cam_ext[0]).max() <= 1e-9
    assert np.array_equal(cam_gpu[3], cams[3])                       # whole camera constant
    assert np.array_equal(cam_gpu[1][:3], cams[1][:3]) and not np.array_equal(cam_gpu[1][3:], cams[1][3:])
    assert cam_gpu[5][2] == cams[5][2]


@pytest.mark.parametrize("manifold", [1, 0])
def test_tracks_batch_matches_per_track_oracle(manifold):
    """theia_hip_ba_tracks_batch = N x BundleAdjustTrack: each track follows the oracle's LM on the
    problem "this point variable, everything else constant"."""
    p = synth.synth_ba_v1(16, 120, seed=0x7AC5, mixed_models=True, sigma_pt=0.05)
    o, oo = both_options(max_num_iterations=20, use_homogeneous_point_parametrization=manifold, use_inner_iterations=0)   # BundleAdjustTrack
    pg = p.copy()
    pg.point_const = np.zeros(120, np.uint8); pg.point_const[5] = 1
    summ = ba.solve_tracks_batch(pg, o)
    assert np.array_equal(pg.points[5], p.points[5]) and summ[5].num_iterations == 0
    for q in list(range(0, 120, 7)) + [119]:
        if q == 5:
            continue
        sel = p.obs_pt == q
        fp = capi.FlatProblem(p.cam_ext.copy(), p.intrinsics.copy(), p.group_model, p.cam_group, p.points[q:q + 1].copy(), p.obs_uv[sel],
                              p.obs_cam[sel], np.zeros(sel.sum(), np.int32), cam_const=np.full(p.cam_ext.shape[0], 3, np.uint8))
        so, _ = ol.solve(fp, oo)
        s = summ[q]
        assert s.success == so.success and s.num_iterations == so.num_iterations and s.num_successful_steps == so.num_successful_steps, q
        assert abs(s.initial_cost - so.initial_cost) <= 1e-10 * so.initial_cost and abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
        assert np.abs(pg.points[q] - fp.points[0]).max() <= 1e-9
    assert np.mean([s.final_cost < s.initial_cost for s in summ if s.num_iterations]) > 0.9


def _with_priors(p, seed, kinds=7):
    """position (GPS-like), gravity and orientation priors on most cameras, ~1 % off the truth."""
    nc = p.cam_ext.shape[0]
    st = synth.Stream(seed, 3)
    i = np.arange(nc)
    mask = np.where(i % 5 == 4, 0, kinds).astype(np.uint8)          # every fifth camera has none
    mask[1] = kinds & 1
    pos = p.cam_ext[:, :3] + 0.02 * np.stack([st.normal(3 * i), st.normal(3 * i + 1), st.normal(3 * i + 2)], 1)
    R = synth.angle_axis_to_matrix(p.cam_ext[:, 3:])
    grav = R @ np.array([0, 0, -1.0]) + 0.005 * np.stack([st.normal(3 * i + 100), st.normal(3 * i + 101), st.normal(3 * i + 102)], 1)
    ori = p.cam_ext[:, 3:] + 0.003 * np.stack([st.normal(3 * i + 200), st.normal(3 * i + 201), st.normal(3 * i + 202)], 1)
    def info(scale, off):
        A = np.tile(np.eye(3) * scale, (nc, 1, 1))
        A[:, 0, 1] = 0.1 * scale * st.normal(i + off); A[:, 2, 0] = -0.2 * scale * st.normal(i + off + 50)
        return A
    p.set_priors(mask, position=(pos, info(20.0, 300)), gravity=(grav, info(50.0, 400)), orientation=(ori, info(80.0, 500)))
    return p


@pytest.mark.parametrize("kinds", [1, 2, 4, 7])
def test_camera_priors_match_oracle(kinds):
    """A12: position / gravity / orientation priors (3 residuals on the extrinsics, no loss): cost, reduced
    system (closed-form Jacobians vs Jets), LM trajectory and result vs the oracle; masks and constant cameras."""
    p = _with_priors(synth.synth_ba_v1(14, 400, seed=0x9A11), 0x9A12)
    p.cam_const = np.zeros(14, np.uint8); p.cam_const[0] = 3; p.cam_const[2] = 1; p.cam_const[3] = 2
    o, oo = both_options(prior_mask=kinds, max_num_iterations=15)
    with ba.BaHandle(p.copy(), o) as h:
        cost, r, jc, jp, valid = h.evaluate()
        S, rhs = h.reduced_system(1e4)
    ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
    assert ocost > ol.evaluate(p, ol.default_options())[1] + 1e-2            # the priors contribute
    assert abs(cost - ocost) <= 1e-12 * ocost
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert rel(S, So) <= 1e-9 and rel(rhs, ro) <= 1e-9
    pg, po = p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    so, tro = ol.solve(po, oo)
    assert s.success and s.num_iterations == so.num_iterations and np.array_equal(tr.accepted, tro.accepted)
    assert rel(tr.cost, tro.cost) <= 1e-8 and abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-7 and np.abs(pg.points - po.points).max() <= 1e-6
    assert np.array_equal(pg.cam_ext[0], p.cam_ext[0])
    # priors pull: without them the result differs
    q = p.copy(); s0, _ = ba.solve(q, ba.default_options())
    assert np.abs(q.cam_ext - pg.cam_ext).max() > 1e-6


def test_camera_priors_through_the_mirror():
    p = synth.synth_ba_v1(10, 300, seed=0x9A13)
    _with_priors(p, 0x9A14)
    rec = sfm.Reconstruction.from_flat(p)
    rec.view_prior_mask = p.cam_prior_mask.copy(); rec.view_priors = dict(p.priors)
    opts = sfm.BundleAdjustmentOptions(); opts.use_position_priors = True; opts.use_gravity_priors = True
    s = sfm.BundleAdjustReconstruction(opts, rec)
    flat = sfm._flatten(sfm.Reconstruction.from_flat(p), range(10), range(300))
    flat.set_priors(p.cam_prior_mask, **p.priors)
    oo = ol.default_options(); oo.prior_mask = 3
    so, _ = ol.solve(flat, oo)
    assert s.success and abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost


def test_golden_camera_model_vectors_on_gpu():
    """tests/golden/camera_models.npz (64 tuples per model, edge cases included) through the HIP evaluate path:
    functor return value, residual, d/d extrinsics, d/d intrinsics, d/d point (ambient 4)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera_models.npz"))
    for model in range(8):
        k = g[f"m{model}_intr"]
        n = g[f"m{model}_ext"].shape[0]
        intr = np.zeros((n, 10)); intr[:, : len(k)] = k
        si = np.stack([1.0 + 0.1 * (np.arange(n) % 3), np.full(n, 0.9)], 1)
        p = capi.FlatProblem(g[f"m{model}_ext"], intr, np.full(n, model, np.int32), np.arange(n, dtype=np.int32), g[f"m{model}_X"],
                             g[f"m{model}_uv"], np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32), obs_sqrt_info=si)
        o = ba.default_options(); o.use_homogeneous_point_parametrization = 0; o.intrinsics_to_optimize = 0x3f
        with ba.BaHandle(p, o) as h:
            cost, r, jc, jp, ji, valid = h.evaluate_ex()
        ok = g[f"m{model}_ok"].astype(bool)
        assert np.array_equal(valid.astype(bool), ok)
        assert np.abs(r[ok] - g[f"m{model}_res"][ok]).max() <= 1e-9
        for mine, ref in ((jc, g[f"m{model}_Je"]), (jp, g[f"m{model}_Jp"]), (ji[:, :, : len(k)], g[f"m{model}_Ji"])):
            scale = np.abs(ref[ok]).max(axis=(0, 1), keepdims=True)
            assert (np.abs(mine[ok] - ref[ok]) / np.maximum(scale, 1e-300)).max() <= 1e-9, model


@pytest.mark.parametrize("name,kw", [("intr", dict(intrinsics_to_optimize=0x11)), ("priors", dict(prior_mask=7)),
                                     ("huber", dict(loss_function_type=1, robust_loss_width=1.5)),
                                     ("depth", dict(loss_function_type=1, robust_loss_width_depth_prior=0.4))])
def test_golden_ba_variants_on_gpu(name, kw):
    from tests.test_oracle_ba import _variant_problem
    v = np.load(os.path.join(os.path.dirname(__file__), "golden", "ba_variants.npz"))
    base = capi.FlatProblem(v["base_cam_ext"], v["base_intrinsics"], v["base_group_model"], v["base_cam_group"], v["base_points"],
                            v["base_obs_uv"], v["base_obs_cam"], v["base_obs_pt"])
    p = _variant_problem(v, base, name)
    o = ba.default_options()
    for kk, vv in kw.items():
        setattr(o, kk, vv)
    o.max_num_iterations = 25
    s, tr = ba.solve(p, o)
    assert s.num_iterations == int(v[f"{name}_num_iterations"]) and np.array_equal(tr.accepted, v[f"{name}_trace_accepted"])
    assert rel(tr.cost, v[f"{name}_trace_cost"]) <= 1e-8
    assert np.abs(p.cam_ext - v[f"{name}_cam_ext"]).max() <= 1e-7 and rel(p.intrinsics, v[f"{name}_intrinsics"]) <= 1e-7
    assert np.abs(p.points - v[f"{name}_points"]).max() <= 1e-6


def _midpoint(origins, dirs):
    """TriangulateMidpoint (triangulation.cc:130-157) restated with numpy."""
    A = np.zeros((3, 3)); b = np.zeros(3)
    for o_, d in zip(origins, dirs):
        T = np.eye(3) - np.outer(d, d)
        A += T; b += T @ o_
    try:
        np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        return None
    return np.linalg.solve(A, b)


def test_estimate_tracks_follows_track_estimator_rules():
    """theia_hip_estimate_tracks = TrackEstimator::EstimateTrack with MIDPOINT triangulation: the angle test,
    the midpoint, the per-track BA (against the oracle's LM from the same start) and the reprojection test."""
    p = synth.synth_ba_v1(12, 90, seed=0xE577, sigma_pt=0.0, sigma_pos=0.0, sigma_rot_deg=0.0, pixel_noise=0.5)
    o, oo = both_options(max_num_iterations=15, use_inner_iterations=0)   # BundleAdjustTrack (estimate_track.cc:289)
    truth = p.points.copy()
    # viewing rays from the camera centres through the (noisy) pixels ~ truth direction + a little noise
    C_ = p.cam_ext[p.obs_cam, :3]
    X = truth[p.obs_pt, :3] / truth[p.obs_pt, 3:]
    st = synth.Stream(0xE578, 1)
    i = np.arange(len(p.obs_pt))
    rays = X - C_ + 2e-3 * np.stack([st.normal(3 * i), st.normal(3 * i + 1), st.normal(3 * i + 2)], 1)
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    pg = p.copy()
    pg.points[:] = 0.0; pg.points[:, 3] = 1.0            # nothing is known about the points beforehand
    pg.point_const = np.zeros(90, np.uint8); pg.point_const[7] = 1      # "already estimated": left alone
    # track 3: all rays (nearly) parallel -> bad angle; track 4: one observation only
    sel3 = np.flatnonzero(p.obs_pt == 3)
    rays[sel3] = rays[sel3[0]]
    keep = np.ones(len(p.obs_pt), bool); sel4 = np.flatnonzero(p.obs_pt == 4); keep[sel4[1:]] = False
    # track 5: an observation far off -> fails the reprojection test
    sel5 = np.flatnonzero(p.obs_pt == 5); pg.obs_uv[sel5[0]] += 400.0
    for name in ("obs_uv", "obs_cam", "obs_pt"):
        setattr(pg, name, np.ascontiguousarray(getattr(pg, name)[keep]))
    rays_k = np.ascontiguousarray(rays[keep])
    est, cnt = ba.estimate_tracks(pg, rays_k, o, 3.0, 5.0, True)
    assert not est[3] and not est[4] and not est[5] and not est[7]
    assert cnt["bad_angles"] == 2 and cnt["bad_reprojections"] >= 1 and cnt["failed_triangulations"] == 0
    assert np.array_equal(pg.points[7], [0, 0, 0, 1])
    assert est.sum() >= 80
    cos_min = np.cos(np.deg2rad(3.0))
    for q in range(0, 90, 6):
        sel = np.flatnonzero(pg.obs_pt == q)
        d = rays_k[sel]
        ok_angle = len(sel) >= 2 and bool(np.any(np.triu(d @ d.T < cos_min, 1)))
        if not ok_angle or q == 7:
            assert not est[q]
            continue
        X0 = _midpoint(pg.cam_ext[pg.obs_cam[sel], :3], d)
        fp = capi.FlatProblem(pg.cam_ext.copy(), pg.intrinsics.copy(), pg.group_model, pg.cam_group, np.append(X0, 1.0)[None].copy(),
                              pg.obs_uv[sel], pg.obs_cam[sel], np.zeros(len(sel), np.int32),
                              cam_const=np.full(pg.cam_ext.shape[0], 3, np.uint8))
        so, _ = ol.solve(fp, oo)
        assert np.abs(pg.points[q] - fp.points[0]).max() <= 1e-8 * max(1.0, np.abs(fp.points[0]).max()), q
        err, nb, _ = ba.track_statistics(fp)
        assert est[q] == bool(so.success and nb[0] == 0 and err[0] < 25.0), q
    # without the BA the point is the midpoint itself
    pg2 = pg.copy(); pg2.points[:] = 0.0; pg2.points[:, 3] = 1.0
    est2, _ = ba.estimate_tracks(pg2, rays_k, o, 3.0, 5.0, False)
    sel = np.flatnonzero(pg2.obs_pt == 12)
    assert np.allclose(pg2.points[12, :3], _midpoint(pg2.cam_ext[pg2.obs_cam[sel], :3], rays_k[sel]), rtol=1e-12, atol=1e-12)


def _projection_matrix(ext, k, model):
    """Camera::GetProjectionMatrix (camera.cc:195-200): K [R | -R c]."""
    R = synth.angle_axis_to_matrix(ext[None, 3:6])[0]
    noskew = model in (3, 4)   # THEIA_CAM_FOV, THEIA_CAM_DIVISION_UNDISTORTION: no skew slot
    f, a = k[0], k[1]
    sk, cx, cy = (0.0, k[2], k[3]) if noskew else (k[2], k[3], k[4])
    K = np.array([[f, sk, cx], [0.0, f * a, cy], [0.0, 0.0, 1.0]])
    return K @ np.concatenate([R, (-R @ ext[:3])[:, None]], axis=1)


@pytest.mark.parametrize("noise", [0.5, 8.0])
@pytest.mark.parametrize("method", [1, 2])
def test_estimate_tracks_svd_and_l2_triangulation(method, noise):
    """TriangulationMethodType SVD / L2_MINIMIZATION (estimate_track.cc:239-257): without the track BA the point is
    TriangulateNViewSVD / TriangulateNView (triangulation.cc:178-214) of the pixels and the cameras' projection matrices,
    restated here with numpy's SVD of the 3N x (4 + N) design matrix / eigh of the 4 x 4 one (defined up to sign); with the
    BA the tracks end where the MIDPOINT start ends (same minimum), and the counters follow the same rules.  The 8-pixel case is
    the one where the SVD's eigenvalue mu is not small against the L2 matrix: the fixed point has to be iterated to convergence."""
    p = synth.synth_ba_v1(14, 120, seed=0xE5A1, sigma_pt=0.0, sigma_pos=0.0, sigma_rot_deg=0.0, pixel_noise=noise)   # pinhole: the linear methods ignore any distortion
    o, _ = both_options(max_num_iterations=15, use_inner_iterations=0)
    C_ = p.cam_ext[p.obs_cam, :3]
    X = p.points[p.obs_pt, :3] / p.points[p.obs_pt, 3:]
    rays = X - C_; rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    pg = p.copy(); pg.points[:] = 0.0; pg.points[:, 3] = 1.0
    est, cnt = ba.estimate_tracks(pg, rays, o, 1.0, 5.0, False, triangulation_method=method)
    worst, dist = 0.0, []
    for q in range(120):
        sel = np.flatnonzero(pg.obs_pt == q)
        d = rays[sel]
        if len(sel) < 2 or not np.any(np.triu(d @ d.T < np.cos(np.deg2rad(1.0)), 1)):
            assert not est[q]
            continue
        Ps = [_projection_matrix(pg.cam_ext[c], pg.intrinsics[pg.cam_group[c]], int(pg.group_model[pg.cam_group[c]])) for c in pg.obs_cam[sel]]
        n = len(sel)
        if method == 1:
            A = np.zeros((3 * n, 4 + n))
            for i, (P, uv) in enumerate(zip(Ps, pg.obs_uv[sel])):
                A[3 * i:3 * i + 3, :4] = -P
                A[3 * i:3 * i + 3, 4 + i] = [uv[0], uv[1], 1.0]
            ref = np.linalg.svd(A)[2][-1][:4]
        else:
            D = np.zeros((4, 4))
            for P, uv in zip(Ps, pg.obs_uv[sel]):
                nn = np.array([uv[0], uv[1], 1.0]); nn /= np.linalg.norm(nn)
                Cm = P - np.outer(nn, nn) @ P
                D += Cm.T @ Cm
            ref = np.linalg.eigh(D)[1][:, 0]
        got = pg.points[q]
        sgn = np.sign(got @ ref)
        worst = max(worst, np.abs(got * sgn - ref).max() / np.abs(ref).max())
        dist.append(np.linalg.norm(got[:3] / got[3] - p.points[q, :3] / p.points[q, 3]))
    assert worst < 1e-7, worst
    assert np.median(dist) < 0.1 * noise      # lens distortion ignored by the linear methods: near the planted points
    assert cnt["failed_triangulations"] == 0
    if noise > 1.0:      # (most of these tracks then fail the 5-pixel reprojection test of EstimateTrack: the points were compared above)
        return
    assert est.sum() >= 100
    # with the track BA: the same minimum as from the midpoint start
    pa = p.copy(); pa.points[:] = 0.0; pa.points[:, 3] = 1.0
    pb = p.copy(); pb.points[:] = 0.0; pb.points[:, 3] = 1.0
    ea, _ = ba.estimate_tracks(pa, rays, o, 1.0, 5.0, True, triangulation_method=method)
    eb, _ = ba.estimate_tracks(pb, rays, o, 1.0, 5.0, True, triangulation_method=0)
    both = ea & eb
    assert both.sum() >= 100
    Xa = pa.points[both, :3] / pa.points[both, 3:]; Xb = pb.points[both, :3] / pb.points[both, 3:]
    assert np.abs(Xa - Xb).max() < 1e-4
    with pytest.raises(capi.TheiaHipError):
        ba.estimate_tracks(pa, rays, o, 1.0, 5.0, True, triangulation_method=3)


def _with_depth_priors(p, cam_gt, pts_gt, every=3, variance=1e-4, noise=0.005, seed=0xDE97):
    """Depth priors (noisy true depth) on every `every`-th observation."""
    idx = np.arange(0, p.obs_uv.shape[0], every)
    X = pts_gt[p.obs_pt[idx]]
    R = synth.angle_axis_to_matrix(cam_gt[p.obs_cam[idx], 3:])
    q = np.einsum("nij,nj->ni", R, X[:, :3] - X[:, 3:] * cam_gt[p.obs_cam[idx], :3])
    st = synth.Stream(seed, 5)
    return p.add_depth_priors(idx, q[:, 2] + noise * st.normal(np.arange(len(idx))), variance)


@pytest.mark.parametrize("loss,intr", [(0, 0), (1, 0), (3, 0), (0, 9)])
def test_depth_prior_rows_match_oracle(loss, intr):
    """A12 depth priors (DepthPriorError as observation rows of kind 1, own loss width): cost, residual and
    Jacobian rows, reduced system, LM trajectory and result vs the oracle; also with intrinsics optimised and
    with a robust loss."""
    p, cam_gt, pts_gt = synth.synth_ba_v1(14, 400, seed=0xDE91, return_truth=True)
    n0 = p.obs_uv.shape[0]
    _with_depth_priors(p, cam_gt, pts_gt)
    o, oo = both_options(max_num_iterations=12, loss_function_type=loss, intrinsics_to_optimize=intr,
                         robust_loss_width_depth_prior=0.5)
    with ba.BaHandle(p.copy(), o) as h:
        cost, r, jc, jp, valid = h.evaluate()
        S, rhs = h.reduced_system(1e4)
    ok, ocost, orr, ojc, ojp = ol.evaluate(p, oo)
    assert abs(cost - ocost) <= 1e-12 * ocost
    assert rel(r, orr) <= 1e-10 and rel(jc, ojc) <= 1e-9 and rel(jp, ojp) <= 1e-9
    assert np.all(np.asarray(r).reshape(-1, 2)[n0:, 1] == 0.0) and np.abs(np.asarray(r).reshape(-1, 2)[n0:, 0]).max() > 0.1
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert rel(S, So) <= 1e-9 and rel(rhs, ro) <= 1e-9
    pg, po = p.copy(), p.copy()
    s, tr = ba.solve(pg, o)
    so, tro = ol.solve(po, oo)
    assert s.success and s.num_iterations == so.num_iterations and np.array_equal(tr.accepted, tro.accepted)
    assert rel(tr.cost, tro.cost) <= 1e-8 and abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-7 and np.abs(pg.points - po.points).max() <= 1e-6
    # the rows matter: the same problem without them ends elsewhere
    q = capi.FlatProblem(p.cam_ext.copy(), p.intrinsics.copy(), p.group_model, p.cam_group, p.points.copy(), p.obs_uv[:n0],
                         p.obs_cam[:n0], p.obs_pt[:n0], cam_const=p.cam_const)
    ba.solve(q, o)
    assert np.abs(q.points - pg.points).max() > 1e-6


def test_depth_prior_rows_are_validated():
    p, cam_gt, pts_gt = synth.synth_ba_v1(6, 50, seed=0xDE92, return_truth=True)
    _with_depth_priors(p, cam_gt, pts_gt)
    o = ba.default_options()
    bad = p.copy(); bad.obs_kind = bad.obs_kind.copy(); bad.obs_kind[0] = 2
    with pytest.raises(capi.TheiaHipError, match="THEIA_OBS"):
        ba.solve(bad, o)
    bad = p.copy(); bad.obs_sqrt_info = None
    with pytest.raises(capi.TheiaHipError, match="obs_sqrt_info"):
        ba.solve(bad, o)
    with pytest.raises(capi.TheiaHipError, match="depth-prior rows"):
        ba.solve_tracks_batch(p.copy(), o)


def test_concurrent_creates_of_large_problems_share_the_host_thread_team():
    """create() at > 262 144 observations runs its structure passes on the library's persistent team of host threads; the team
    serves one region at a time and a second caller falls back to threads of its own.  Three host threads creating + solving the
    C2 problem (299 587 observations) at once -- with free intrinsics on one of them, an interleaved input on another -- must
    each reproduce the sequential result bit for bit."""
    import threading
    p = synth.ba_config("C2")
    cnt = np.bincount(p.obs_pt)
    within = np.arange(len(p.obs_pt)) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    order = np.lexsort((p.obs_pt, within))
    q = capi.FlatProblem(p.cam_ext.copy(), p.intrinsics.copy(), p.group_model, p.cam_group, p.points.copy(), p.obs_uv[order],
                         p.obs_cam[order], p.obs_pt[order], p.cam_const, p.group_const, p.point_const)
    o = ba.default_options(); o.max_num_iterations = 3
    oi = ba.default_options(); oi.max_num_iterations = 3; oi.intrinsics_to_optimize = 0x11
    jobs = [(p, o), (q, o), (p, oi)]

    def run(job):
        x = job[0].copy(); s, t = ba.solve(x, job[1])
        return t.cost[: t.size].copy(), x.cam_ext.copy(), x.points.copy(), x.intrinsics.copy()
    ref = [run(j) for j in jobs]
    out = [None] * 3; errs = []

    def worker(k):
        try:
            for _ in range(3):
                out[k] = run(jobs[k])
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for k in range(3):
        for a, b in zip(out[k], ref[k]):
            assert np.array_equal(a, b), k
    assert np.array_equal(ref[0][0], ref[1][0])     # the interleaved input: the same trajectory


def test_entry_points_are_reentrant_across_host_threads():
    """SURVEY 8(b) threading: the pipelines call BundleAdjustTrack / the estimators from thread-pool workers; ctypes
    releases the GIL, so concurrent calls really overlap.  Concurrent solves must equal the sequential ones."""
    import threading
    from pytheiasfm_amd import ransac
    probs = [synth.synth_ba_v1(10 + k, 300 + 40 * k, seed=0x7EAD00 + k) for k in range(4)]
    o = ba.default_options(); o.max_num_iterations = 8
    ref = []
    for p in probs:
        q = p.copy(); s, _ = ba.solve(q, o); ref.append((s.final_cost, s.num_iterations, q.cam_ext.copy(), q.points.copy()))
    data, offsets, _ = synth.synth_ransac_v1(6, 300, "relative", seed=0x7EAD10)
    rp = ransac.RansacParameters(); rp.error_thresh = (2 / 1000.0) ** 2; rp.seed = 5
    rref = ransac.estimate_batch(0, data, offsets, rp)
    out = [None] * 4; rout = [None] * 2; errs = []

    def ba_worker(k):
        try:
            for _ in range(3):
                q = probs[k].copy(); s, _ = ba.solve(q, o)
                out[k] = (s.final_cost, s.num_iterations, q.cam_ext.copy(), q.points.copy())
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    def ransac_worker(k):
        try:
            for _ in range(3):
                rout[k] = ransac.estimate_batch(0, data, offsets, rp)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    # the batched per-track entry points as well (they run on the legacy stream)
    tp = synth.synth_ba_v1(12, 500, seed=0x7EAD20)
    tb_ref = tp.copy(); ba.solve_tracks_batch(tb_ref, o)
    st_ref = ba.track_statistics(tp)
    tb_out = [None] * 2

    def track_worker(k):
        try:
            for _ in range(3):
                q = tp.copy(); ba.solve_tracks_batch(q, o)
                tb_out[k] = (q.points.copy(), ba.track_statistics(tp))
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = ([threading.Thread(target=ba_worker, args=(k,)) for k in range(4)] + [threading.Thread(target=ransac_worker, args=(k,)) for k in range(2)]
          + [threading.Thread(target=track_worker, args=(k,)) for k in range(2)])
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for k in range(4):
        assert out[k][1] == ref[k][1] and out[k][0] == ref[k][0]
        assert np.array_equal(out[k][2], ref[k][2]) and np.array_equal(out[k][3], ref[k][3])
    for k in range(2):
        assert np.array_equal(rout[k]["inlier_mask"], rref["inlier_mask"]) and np.array_equal(rout[k]["num_iterations"], rref["num_iterations"])
        assert np.array_equal(tb_out[k][0], tb_ref.points)
        assert all(np.array_equal(x, y) for x, y in zip(tb_out[k][1], st_ref))


def test_select_good_tracks_for_bundle_adjustment_mirror():
    """SelectGoodTracksForBundleAdjustment: device track statistics (mean squared reprojection error over the
    estimated views) = the oracle's residuals, and the selection covers every view with K tracks."""
    p = synth.synth_ba_v1(12, 400, seed=0x5E1EC7)
    rec = sfm.Reconstruction.from_flat(p)
    rec.view_estimated[3] = False; rec.track_estimated[::17] = False
    ok, sel = sfm.SelectGoodTracksForBundleAdjustment(rec, 10, 200, 25)
    assert ok and len(sel) > 0 and all(rec.track_estimated[t] for t in sel)
    keep = rec.view_estimated[rec.obs_view] & rec.track_estimated[rec.obs_track]
    _, _, r, _, _ = ol.evaluate(p, ol.default_options())
    sq = (np.asarray(r).reshape(-1, 2) ** 2).sum(1)
    err = np.bincount(rec.obs_track[keep], weights=sq[keep], minlength=400) / np.maximum(1, np.bincount(rec.obs_track[keep], minlength=400))
    ref = sfm._select_good_tracks(rec, [v for v in range(12) if rec.view_estimated[v]], np.bincount(rec.obs_track[keep], minlength=400),
                                  err, 10, 200, 25)
    assert sel == ref.tolist()
    # ... and against the oracle's sequential restatement of select_good_tracks_for_bundle_adjustment.cc:79-320 (oracle/sfm_rules.py:
    # per-track statistics through the oracle's camera projection, dict image grids, min_element / partial_sort semantics),
    # which shares nothing with the mirror's vectorised helper -- for three settings of (length threshold, cell size, K)
    R = ol.sfm_rules()
    views = [v for v in range(12) if rec.view_estimated[v]]
    assert sel == R.select_good_tracks_for_bundle_adjustment(ol, rec, views, 10, 200, 25)
    for (lt, cell, K) in ((3, 120, 40), (2, 400, 5), (10, 64, 0)):
        ok2, sel2 = sfm.SelectGoodTracksForBundleAdjustment(rec, lt, cell, K)
        assert ok2 and sel2 == R.select_good_tracks_for_bundle_adjustment(ol, rec, views, lt, cell, K), (lt, cell, K)
    # a subset of the views (the overload of :280-320 the incremental pipeline calls)
    ok3, sel3 = sfm.SelectGoodTracksForBundleAdjustment(rec, 10, 200, 25, view_ids=[0, 5, 7])
    assert ok3 and sel3 == R.select_good_tracks_for_bundle_adjustment(ol, rec, [0, 5, 7], 10, 200, 25)
    chosen = np.zeros(400, bool); chosen[sel] = True
    for v in range(12):
        if not rec.view_estimated[v]:
            continue
        t = rec.obs_track[(rec.obs_view == v) & keep]
        assert chosen[t].sum() >= min(25, len(t))


def test_two_views_angular_batch_follows_oracle():
    """theia_hip_ba_two_views_angular_batch = N x BundleAdjustTwoViewsAngular (bundle_adjust_two_views.cc:189-246):
    one wave per pair, closed-form Jacobians of the angular epipolar error, SphereManifold<3> position, CGNR steps.
    Against the Jet-based oracle: same iteration / step counts, poses to 1e-9, costs to 1e-9 relative."""
    data, off, truth = synth.synth_ransac_v1(12, 500, kind="relative", noise_px=0.5, seed=0x5AC50C00)
    corr, offs, x0s = [], [0], []
    for p in range(12):
        c = data[off[p]:off[p + 1]][truth["inlier"][p]]
        w = synth.matrix_to_angle_axis(truth["R"][p]); pos = truth["position"][p] / np.linalg.norm(truth["position"][p])
        x0 = np.concatenate([w + 0.004 * (p % 3 + 1), pos + 0.01 * ((p % 4) - 1.5)]); x0[3:] /= np.linalg.norm(x0[3:])
        corr.append(c); offs.append(offs[-1] + len(c)); x0s.append(x0)
    # an empty problem and one whose residuals all lie beyond the truncation width
    corr.append(np.zeros((0, 4))); offs.append(offs[-1]); x0s.append(np.array([0.1, 0.2, 0.3, 0.0, 0.0, 1.0]))
    corr.append(data[off[0]:off[1]][~truth["inlier"][0]][:40] * 3.0); offs.append(offs[-1] + 40); x0s.append(np.array([0.1, -0.2, 0.05, 0.0, 0.6, 0.8]))
    corr = np.vstack(corr); x0s = np.array(x0s)
    for loss, width in ((6, 2e-4), (0, 1.0), (1, 1e-5)):
        o = ba.default_options(); o.max_num_iterations = 15; o.loss_function_type = loss; o.robust_loss_width = width
        for solver in (ba.TWO_VIEW_CGNR, ba.TWO_VIEW_EXACT):
            pose = x0s.copy()
            summ = ba.solve_two_views_angular_batch(offs, corr, pose, o, solver)
            for p in range(len(x0s)):
                ref, s = ol.two_views_angular(corr[offs[p]:offs[p + 1]], x0s[p], o, solver)
                assert summ[p].num_iterations == s["num_iterations"] and summ[p].num_successful_steps == s["num_successful_steps"], (loss, p)
                assert summ[p].termination_type == s["termination_type"] and summ[p].success == s["success"]
                assert np.abs(pose[p] - ref).max() <= 1e-9, (loss, p, np.abs(pose[p] - ref).max())
                assert abs(summ[p].initial_cost - s["initial_cost"]) <= 1e-9 * max(s["initial_cost"], 1e-300) + 1e-300
                assert abs(summ[p].final_cost - s["final_cost"]) <= 1e-9 * max(s["final_cost"], 1e-300) + 1e-20
                if p < 12 and loss == 6:
                    assert summ[p].final_cost < summ[p].initial_cost and abs(np.linalg.norm(pose[p][3:]) - 1.0) <= 1e-14
            assert np.array_equal(pose[12], x0s[12]) and summ[12].num_iterations == 0


def test_compact_intrinsics_rows_equal_full_rows():
    """When no group frees more than four intrinsics the gather records keep four COMPACT intrinsics rows
    (ba_kernels.hip RecI<PD, 4>) instead of ten: the reduced camera system and the solve are the same as with the
    full rows (THEIA_HIP_INTR_ROWS=10 forces them) up to the summation order."""
    p = synth.synth_ba_v1(16, 900, seed=977, num_groups=3, fix_gauge=True, pixel_noise=0.3)
    o = ba.default_options(); o.intrinsics_to_optimize = INTR_FOCAL_RADIAL; o.max_num_iterations = 6
    out = []
    for rows in ("4", "10"):
        os.environ["THEIA_HIP_INTR_ROWS"] = rows
        try:
            h = ba.BaHandle(p.copy(), o)
            S, rhs = h.reduced_system(1e4)
            n = S.shape[0]
            q = p.copy()
            s, tr = ba.solve(q, o)
            out.append((n, S, rhs, q, s, tr))
        finally:
            del os.environ["THEIA_HIP_INTR_ROWS"]
    a, b = out
    assert a[0] == b[0] and rel(a[1], b[1]) <= 1e-12 and rel(a[2], b[2]) <= 1e-11
    assert a[4].num_iterations == b[4].num_iterations and np.array_equal(a[5].accepted, b[5].accepted)
    assert rel(a[3].intrinsics, b[3].intrinsics) <= 1e-9 and np.abs(a[3].cam_ext - b[3].cam_ext).max() <= 1e-8
    assert not np.array_equal(a[3].intrinsics[:, 0], p.intrinsics[:, 0])


@pytest.mark.parametrize("groups,which", [(1, "focal_radial"), (2, "focal_radial"), (4, "focal_radial"), (7, "focal_radial"), (1, "all"), (3, "all")])
def test_intrinsics_track_sums_equal_pair_lists_and_oracle(groups, which):
    """Tracks whose observations share one variable intrinsics group enter the camera x group and group x group blocks
    through the SUM of their intrinsics fields (pseudo-records written by k_lin_obs_intr) instead of through every
    ordered pair of observations (build_gather_lists_intr): the reduced system equals the one of the per-pair lists
    (THEIA_HIP_INTR_PAIRS=1) up to the summation order, and the oracle's; the LM trajectory follows.  One group: every
    track is summed; 2 - 4 interleaved groups: one sum per group of a track, next to tracks that keep explicit pairs
    (as many groups as observations); 7 groups: tracks beyond the four-group limit too, sharing cameras and blocks."""
    p = synth.synth_ba_v1(14, 1100, seed=0x7A5 + groups, num_groups=groups, fix_gauge=True, pixel_noise=0.3)
    intr = INTR_FOCAL_RADIAL if which == "focal_radial" else INTR_ALL
    o, oo = both_options(intrinsics_to_optimize=intr, max_num_iterations=6)
    out = []
    for pairs in (False, True):
        if pairs:
            os.environ["THEIA_HIP_INTR_PAIRS"] = "1"
        try:
            with ba.BaHandle(p.copy(), o) as h:
                S, rhs = h.reduced_system(1e4)
            q = p.copy()
            s, tr = ba.solve(q, o)
            out.append((S, rhs, q, s, tr))
        finally:
            os.environ.pop("THEIA_HIP_INTR_PAIRS", None)
    a, b = out
    assert a[0].shape == b[0].shape and rel(a[0], b[0]) <= 1e-12 and rel(a[1], b[1]) <= 1e-11
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert rel(a[0], So) <= 1e-9 and rel(a[1], ro) <= 1e-9
    assert a[3].num_iterations == b[3].num_iterations and np.array_equal(a[4].accepted, b[4].accepted)
    assert rel(a[2].intrinsics, b[2].intrinsics) <= 1e-9 and np.abs(a[2].cam_ext - b[2].cam_ext).max() <= 1e-8
    assert not np.array_equal(a[2].intrinsics[:, 0], p.intrinsics[:, 0])


def test_bundle_adjust_two_views_mirror_matches_oracle():
    """twoview.BundleAdjustTwoViews (bundle_adjust_two_views.cc:110-185): camera 1 fixed, camera 2 free, XYZW points
    without a manifold, focal lengths free unless held constant -- against the oracle on the same flat problem."""
    from pytheiasfm_amd import twoview as tv
    data, offsets, truth = synth.synth_ransac_v1(1, 120, "fundamental", seed=0x5AC52700, inlier_lo=1.0, inlier_hi=1.0, noise_px=0.5)
    corr = data[:120]
    R, t = truth["R"][0], truth["t"][0]
    ext2 = np.concatenate([truth["position"][0] + 0.01, synth.matrix_to_angle_axis(R) + 0.005])
    x1 = (corr[:, :2] - np.array([500.0, 400.0])) / 1000.0
    depth = 6.0 + 0.3 * np.sin(np.arange(120))
    pts = np.column_stack([x1 * depth[:, None], depth, np.ones(120)])        # rough points on the first camera's rays
    for const2 in (True, False):
        cam1 = {"ext": np.zeros(6), "intr": np.array([1000.0, 1, 0, 500, 400, 0, 0]), "model": 0}
        cam2 = {"ext": ext2.copy(), "intr": np.array([1010.0, 1, 0, 500, 400, 0, 0]), "model": 0}
        p3 = np.ascontiguousarray(pts.copy())
        bo = tv.TwoViewBundleAdjustmentOptions(); bo.constant_camera2_intrinsics = const2; bo.ba_options.max_num_iterations = 20
        summ = tv.BundleAdjustTwoViews(bo, corr, cam1, cam2, p3)
        flat = capi.FlatProblem(np.array([np.zeros(6), ext2]), np.array([[1000.0, 1, 0, 500, 400, 0, 0], [1010.0, 1, 0, 500, 400, 0, 0]]),
                                [0, 0], [0, 1], pts.copy(), np.concatenate([corr[:, :2], corr[:, 2:]]),
                                np.concatenate([np.zeros(120, np.int32), np.ones(120, np.int32)]),
                                np.concatenate([np.arange(120), np.arange(120)]).astype(np.int32), cam_const=[3, 0],
                                group_const=[1, int(const2)])
        # bundle_adjust_two_views.cc:61-72 builds its own solver options: no inner iterations
        o, oo = both_options(max_num_iterations=20, intrinsics_to_optimize=0x01, use_homogeneous_point_parametrization=0, use_inner_iterations=0,
                             max_trust_region_radius=1e16)
        so, tro = ol.solve(flat, oo)
        assert summ.success and summ.final_cost < 0.01 * summ.initial_cost
        assert rel(summ.final_cost, so.final_cost) <= 1e-8 and np.abs(cam2["ext"] - flat.cam_ext[1]).max() <= 1e-6
        assert np.array_equal(cam1["ext"], np.zeros(6)) and cam1["intr"][0] == 1000.0
        assert (cam2["intr"][0] == 1010.0) == const2 and rel(cam2["intr"], flat.intrinsics[1][:7]) <= 1e-7
        assert rel(p3, flat.points) <= 1e-6


def test_two_view_ba_batch_matches_the_general_solver():
    """theia_hip_ba_two_views_batch = N x BundleAdjustTwoViews (bundle_adjust_two_views.cc:110-185), one LM per wavefront:
    every pair of a ragged batch (8 .. 130 points, focal lengths free / constant in every combination) against
    theia_hip_ba_solve on the same flat problem (the mirror's per-pair path), which the oracle pins."""
    from pytheiasfm_amd import twoview as tv
    ns = [90, 40, 130, 64, 65, 8]
    npairs = len(ns)
    data, offsets, truth = synth.synth_ransac_v1(npairs, 130, "fundamental", seed=0x5AC52800, inlier_lo=1.0, inlier_hi=1.0, noise_px=0.5)
    off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    cam_ext = np.zeros((npairs, 2, 6)); intr = np.zeros((npairs, 2, capi.THEIA_MAX_INTRINSICS))
    pts = np.zeros((off[-1], 4)); kconst = np.zeros((npairs, 2), np.uint8); allc = np.zeros((off[-1], 4))
    ref = []
    for i, n in enumerate(ns):
        corr = data[offsets[i]:offsets[i] + n]
        ext2 = np.concatenate([truth["position"][i] + 0.01, synth.matrix_to_angle_axis(truth["R"][i]) + 0.004])
        depth = 6.0 + 0.3 * np.sin(np.arange(n) + i)
        x1 = (corr[:, :2] - np.array([500.0, 400.0])) / 1000.0
        p3 = np.column_stack([x1 * depth[:, None], depth, np.ones(n)])
        const = (i % 2 == 0, i % 3 == 0)
        cam_ext[i, 1] = ext2; intr[i, 0, :5] = [1000.0, 1, 0, 500, 400]; intr[i, 1, :5] = [1010.0, 1, 0, 500, 400]
        pts[off[i]:off[i + 1]] = p3; kconst[i] = const; allc[off[i]:off[i + 1]] = corr
        cam1 = {"ext": np.zeros(6), "intr": np.array([1000.0, 1, 0, 500, 400, 0, 0]), "model": 0}
        cam2 = {"ext": ext2.copy(), "intr": np.array([1010.0, 1, 0, 500, 400, 0, 0]), "model": 0}
        q3 = np.ascontiguousarray(p3.copy())
        bo = tv.TwoViewBundleAdjustmentOptions(); bo.constant_camera1_intrinsics = const[0]; bo.constant_camera2_intrinsics = const[1]
        bo.ba_options.max_num_iterations = 20
        summ = tv.BundleAdjustTwoViews(bo, corr, cam1, cam2, q3, batched=False)
        ref.append((summ, cam2["ext"].copy(), cam1["intr"][0], cam2["intr"][0], q3))
    o = ba.default_options(); o.max_num_iterations = 20; o.max_trust_region_radius = 1e16; o.use_inner_iterations = 0
    out = ba.solve_two_views_batch(off, allc, cam_ext, intr, np.zeros((npairs, 2), np.int32), kconst, pts, o)
    for i in range(npairs):
        s, (rs, ext2, f1, f2, q3) = out[i], ref[i]
        assert s.success == rs.success and s.num_iterations == rs.num_iterations, (i, s.num_iterations, rs.num_iterations)
        assert rel(s.initial_cost, rs.initial_cost) <= 1e-12 and rel(s.final_cost, rs.final_cost) <= 1e-8
        assert np.abs(cam_ext[i, 1] - ext2).max() <= 1e-7 and np.array_equal(cam_ext[i, 0], np.zeros(6))
        assert rel(intr[i, 0, 0], f1) <= 1e-8 and rel(intr[i, 1, 0], f2) <= 1e-8
        assert (intr[i, 0, 0] == 1000.0) == bool(kconst[i, 0]) and (intr[i, 1, 0] == 1010.0) == bool(kconst[i, 1])
        assert rel(pts[off[i]:off[i + 1]], q3) <= 1e-6
    # a point behind camera 1 at the start: the functor fails, the pair's solve reports failure like the general solver
    bad = pts[:ns[0]].copy(); bad[3, 2] = -5.0
    ce = cam_ext[:1].copy(); ki = intr[:1].copy()
    sb = ba.solve_two_views_batch(np.array([0, ns[0]]), allc[:ns[0]], ce, ki, np.zeros((1, 2), np.int32), kconst[:1], bad, o)[0]
    q = bad.copy(); q[:] = pts[:ns[0]]; q[3, 2] = -5.0
    cam1 = {"ext": np.zeros(6), "intr": ki[0, 0, :7].copy(), "model": 0}; cam2 = {"ext": ce[0, 1].copy(), "intr": ki[0, 1, :7].copy(), "model": 0}
    bo = tv.TwoViewBundleAdjustmentOptions(); bo.constant_camera1_intrinsics = bool(kconst[0, 0]); bo.constant_camera2_intrinsics = bool(kconst[0, 1])
    bo.ba_options.max_num_iterations = 20
    rsb = tv.BundleAdjustTwoViews(bo, allc[:ns[0]], cam1, cam2, np.ascontiguousarray(q), batched=False)
    assert bool(sb.success) == bool(rsb.success)


@pytest.mark.parametrize("groups,intr,manifold,mixed", [(1, 0x11, 1, False), (3, 0x11, 1, True), (7, 0x11, 0, False),
                                                          (2, 0x13, 1, False), (8, 0x11, 1, True),
                                                          # round 5: five to seven free parameters -> 13-row compound blocks
                                                          (2, 0x1b, 1, False), (3, 0x1f, 1, True), (1, 0x3f, 1, False),
                                                          (4, 0x3f, 0, True), (8, 0x3f, 1, True)])
def test_fused_intrinsics_assembly_matches_gather_kernels_and_oracle(groups, intr, manifold, mixed):
    """Intrinsics free (FOCAL_LENGTH | RADIAL_DISTORTION: three compact rows; with ASPECT_RATIO: four; up to every intrinsic of
    the pinhole / double-sphere models -- OptimizeIntrinsicsType::ALL -- seven rows in 13-wide blocks): the fused kernel of
    ba_fused_intr.hip -- compound [extrinsics | intrinsics] blocks per camera pair, the groups' rows summed afterwards
    (k_sum_items) -- against the first-generation gather kernels (THEIA_HIP_INTR_GATHER=1) and the oracle: reduced system,
    LM trajectory, parameters.  fix_gauge holds two cameras constant whose intrinsics groups stay variable (they take part
    in the runs with a zero extrinsics block); mixed = pinhole + double-sphere groups; manifold 0 = XYZW points."""
    p = synth.synth_ba_v1(24, 2500, seed=0x1F5 + groups, num_groups=groups, fix_gauge=True, pixel_noise=0.3, mixed_models=mixed)
    o, oo = both_options(intrinsics_to_optimize=intr, max_num_iterations=6, use_homogeneous_point_parametrization=manifold)
    res = []
    for gather in (False, True):
        if gather:
            os.environ["THEIA_HIP_INTR_GATHER"] = "1"
        try:
            with ba.BaHandle(p.copy(), o) as h:
                S, rhs = h.reduced_system(1e4)
                S2, rhs2 = h.reduced_system(1e4)
                if not gather:   # written, not accumulated, in a fixed order (the gather kernels use FP64 atomics on long lists)
                    assert np.array_equal(S, S2) and np.array_equal(rhs, rhs2), "fused assembly is not reproducible"
            q = p.copy()
            s, tr = ba.solve(q, o)
            res.append((S, rhs, q, s, tr))
        finally:
            os.environ.pop("THEIA_HIP_INTR_GATHER", None)
    a, b = res
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert a[0].shape == b[0].shape == So.shape
    assert rel(a[0], b[0]) <= 1e-12 and rel(a[1], b[1]) <= 1e-11
    assert rel(a[0], So) <= 1e-10 and rel(a[1], ro) <= 1e-10
    qo = p.copy()
    so, tro = ol.solve(qo, oo)
    assert a[3].num_iterations == b[3].num_iterations == so.num_iterations
    n = a[4].size
    assert np.array_equal(a[4].accepted[:n], tro.accepted[:n]) and rel(a[4].cost[:n], tro.cost[:n]) <= 1e-9
    # (five to seven free parameters per group -- skew, principal point -- leave the normal equations far worse conditioned: the
    # same reduced system and accept sequence, parameters to 1e-7 of the oracle's and of the gather kernels')
    ptol = 1e-9 if bin(intr).count("1") <= 2 or intr == 0x13 else 1e-7
    assert rel(a[2].intrinsics, qo.intrinsics) <= ptol and np.abs(a[2].cam_ext - qo.cam_ext).max() <= 10 * ptol
    assert rel(a[2].intrinsics, b[2].intrinsics) <= ptol and np.abs(a[2].cam_ext - b[2].cam_ext).max() <= 10 * ptol
    assert np.abs(a[2].points - qo.points).max() <= 10 * ptol
    assert not np.array_equal(a[2].intrinsics[:, 0], p.intrinsics[:, 0])


@pytest.mark.parametrize("model,groups,manifold", [(1, 2, 1), (2, 3, 1), (1, 1, 0)])
def test_fused_intrinsics_assembly_with_nine_and_ten_free_parameters(model, groups, manifold):
    """OptimizeIntrinsicsType::ALL on the radial-tangential (ten free parameters) and fisheye (nine) models: 16-row compound
    blocks (round 5; the 10 x 10 group blocks summed as two items of five rows) against the gather kernels and the oracle --
    reduced system, accept sequence, cost trace."""
    p = synth.synth_ba_v1(20, 1800, seed=0x2F5 + model, num_groups=groups, fix_gauge=True, pixel_noise=0.3)
    k = CAMERA_MODEL_INTRINSICS[model]
    p.group_model[:] = model
    p.intrinsics[:] = 0.0
    p.intrinsics[:, : len(k)] = k
    o, oo = both_options(intrinsics_to_optimize=0x3f, max_num_iterations=5, use_homogeneous_point_parametrization=manifold, use_inner_iterations=0)
    res = []
    for gather in (False, True):
        if gather:
            os.environ["THEIA_HIP_INTR_GATHER"] = "1"
        try:
            with ba.BaHandle(p.copy(), o) as h:
                S, rhs = h.reduced_system(1e4)
                S2, rhs2 = h.reduced_system(1e4)
                if not gather:
                    assert np.array_equal(S, S2) and np.array_equal(rhs, rhs2), "fused assembly is not reproducible"
                    assert h.plan_info()["fused_runs"] > 0
            q = p.copy()
            s, tr = ba.solve(q, o)
            res.append((S, rhs, q, s, tr))
        finally:
            os.environ.pop("THEIA_HIP_INTR_GATHER", None)
    a, b = res
    So, ro = ol.reduced_system(p, oo, 1e4)
    assert a[0].shape == b[0].shape == So.shape
    assert rel(a[0], b[0]) <= 1e-11 and rel(a[1], b[1]) <= 1e-10
    assert rel(a[0], So) <= 1e-9 and rel(a[1], ro) <= 1e-9
    so, tro = ol.solve(p.copy(), oo)
    assert a[3].num_iterations == b[3].num_iterations == so.num_iterations
    n = a[4].size
    fin = tro.cost[:n] < 1e300
    assert np.array_equal(a[4].accepted[:n], tro.accepted[:n]) and rel(a[4].cost[:n][fin], tro.cost[:n][fin]) <= 1e-7


def test_view_covariances_with_optimised_intrinsics():
    """BundleAdjustViewsWithCov with FOCAL_LENGTH | RADIAL_DISTORTION free: the views share intrinsics groups, J'J is an arrow
    (group columns against every camera of the group) and ceres::Covariance returns the extrinsics blocks of its inverse.
    Reference value: numpy inverse of J'J assembled from the oracle's Jets (J_ext, J_intr per observation)."""
    p = synth.synth_ba_v1(9, 250, seed=71, pixel_noise=0.6, num_groups=2)
    opts = sfm.BundleAdjustmentOptions(); opts.max_num_iterations = 25
    opts.intrinsics_to_optimize = 0x01 | 0x10      # FOCAL_LENGTH | RADIAL_DISTORTION
    rec = sfm.Reconstruction.from_flat(p)
    views = [1, 2, 4, 6, 7]
    sv, cv, fv = sfm.BundleAdjustViewsWithCov(rec, opts, views)
    assert sv.success and fv > 0
    flat = sfm._flatten(rec, views, [], options=opts)            # the solved state
    cols, ncol = {}, 0
    for v in views:
        cols[("c", v)] = ncol; ncol += 6
    free = [0, 5, 6]                                             # pinhole: focal length, two radial terms
    for g in sorted(set(int(flat.cam_group[v]) for v in views)):
        cols[("g", g)] = ncol; ncol += len(free)
    rows = []
    for i in np.flatnonzero(np.isin(flat.obs_cam, views)):
        c = int(flat.obs_cam[i]); g = int(flat.cam_group[c])
        ok, r, Je, Ji, Jp = ol.reprojection_error(int(flat.group_model[g]), flat.cam_ext[c], flat.intrinsics[g][:7], flat.points[flat.obs_pt[i]], flat.obs_uv[i])
        J = np.zeros((2, ncol))
        J[:, cols[("c", c)]:cols[("c", c)] + 6] = Je
        J[:, cols[("g", g)]:cols[("g", g)] + len(free)] = Ji[:, free]
        rows.append(J)
    J = np.concatenate(rows)
    cov = np.linalg.inv(J.T @ J) * fv
    for v in views:
        o = cols[("c", v)]
        ref = cov[o:o + 6, o:o + 6]
        assert np.abs(cv[v] - ref).max() <= 1e-6 * np.abs(ref).max()
        assert np.all(np.linalg.eigvalsh(cv[v]) > 0)
    # the coupling matters: the block-diagonal inverse is a different (smaller) matrix
    o = cols[("c", views[0])]
    Jb = J[:, o:o + 6]
    assert np.abs(np.linalg.inv(Jb.T @ Jb) * fv - cv[views[0]]).max() > 1e-3 * np.abs(cv[views[0]]).max()


def test_view_covariances_with_intrinsics_have_no_size_limit():
    """VERDICT r5 missing 6: camera covariances with optimised intrinsics on a reduced system of more than 2048 columns
    (360 views, three shared groups: 30 + 2160 columns) -- the arrow structure of J'J is inverted group by group, linear in the
    number of cameras (bundle_adjuster.cc:660-773 has no limit).  Reference value: numpy's dense inverse of J'J from the
    oracle's Jets."""
    p = synth.synth_ba_v1(360, 5000, seed=72, pixel_noise=0.5, num_groups=3)
    opts = sfm.BundleAdjustmentOptions(); opts.max_num_iterations = 6
    opts.intrinsics_to_optimize = 0x01 | 0x10
    rec = sfm.Reconstruction.from_flat(p)
    views = list(range(360))
    sv, cv, fv = sfm.BundleAdjustViewsWithCov(rec, opts, views)
    assert sv.success and fv > 0
    flat = sfm._flatten(rec, views, [], options=opts)
    free = [0, 5, 6]
    ncol = 6 * 360 + 3 * len(free)
    JTJ = np.zeros((ncol, ncol))
    for i in range(len(flat.obs_cam)):
        c = int(flat.obs_cam[i]); g = int(flat.cam_group[c])
        ok, r, Je, Ji, Jp = ol.reprojection_error(int(flat.group_model[g]), flat.cam_ext[c], flat.intrinsics[g][:7], flat.points[flat.obs_pt[i]], flat.obs_uv[i])
        idx = np.concatenate([np.arange(6 * c, 6 * c + 6), 2160 + 3 * g + np.arange(3)])
        Jo = np.concatenate([Je, Ji[:, free]], axis=1)
        JTJ[np.ix_(idx, idx)] += Jo.T @ Jo
    cov = np.linalg.inv(JTJ) * fv
    worst = 0.0
    for v in views:
        ref = cov[6 * v:6 * v + 6, 6 * v:6 * v + 6]
        worst = max(worst, np.abs(cv[v] - ref).max() / np.abs(ref).max())
    assert worst <= 1e-6, worst


def test_problem_cache_reuses_the_handle_on_unchanged_topology():
    """SURVEY 8(f) row 4 (problem-IR cache): BundleAdjustReconstruction twice on the same topology with different
    parameters = one theia_hip_ba_create; the cached solve equals the one-shot theia_hip_ba_solve (1e-10); an edit of
    the topology (one track set unestimated) or of the options builds a new handle."""
    import copy
    p = synth.synth_ba_v1(24, 4000, seed=11, mixed_models=True)
    opts = sfm.BundleAdjustmentOptions(); opts.max_num_iterations = 6
    rng = np.random.default_rng(3)

    def perturbed():
        r = sfm.Reconstruction.from_flat(p)
        r.cam_ext = r.cam_ext + rng.normal(0.0, 1e-3, r.cam_ext.shape)
        r.points[:, :3] += rng.normal(0.0, 1e-2, (r.points.shape[0], 3))
        return r

    cache = sfm.set_problem_cache(1)
    try:
        for round_ in range(3):
            r = perturbed()
            ref = copy.deepcopy(r)
            s = sfm.BundleAdjustReconstruction(opts, r)
            flat = sfm._flatten(ref, ref.ViewIds(), ref.TrackIds(), options=opts)
            s0, _ = ba.solve(flat, opts.to_c())
            assert s.success and s.num_iterations == s0.num_iterations and abs(s.final_cost - s0.final_cost) <= 1e-12 * s0.final_cost
            assert np.allclose(r.cam_ext, flat.cam_ext, rtol=1e-10, atol=1e-12) and np.allclose(r.points, flat.points, rtol=1e-10, atol=1e-12)
            assert np.allclose(r.group_intrinsics, flat.intrinsics, rtol=1e-10, atol=1e-12)
        assert (cache.misses, cache.hits) == (1, 2)
        r = perturbed(); r.track_estimated[17] = False
        assert sfm.BundleAdjustReconstruction(opts, r).success and cache.misses == 2
        o2 = copy.copy(opts); o2.max_num_iterations = 3
        assert sfm.BundleAdjustReconstruction(o2, perturbed()).success and cache.misses == 3 and len(cache._handles) == 1
    finally:
        sfm.set_problem_cache(1)
