"""GPU: bundle adjustment with the inverse-depth track parametrisation (csrc/ba_invdepth.hip: per-track rank-one Schur
elimination, two camera blocks per residual) against the oracle's dense LM of the same problem: LM traces to 1e-9,
parameters to 1e-8."""
import numpy as np
import pytest

from pytheiasfm_amd import _capi as capi, ba, sfm
from tests import invdepth as idp
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _both(p, iters=15, **kw):
    out = []
    for mod in (ba, ol):
        q = p.copy(); o = mod.default_options(); o.max_num_iterations = iters; o.use_inner_iterations = 0
        for k, v in kw.items():
            setattr(o, k, v)
        s, tr = ba.solve(q, o) if mod is ba else ol.solve_inverse_depth(q, o)
        out.append((q, s, tr))
    return out


def _compare(g, o):
    (qg, sg, tg), (qo, so, to) = g, o
    assert sg.success and so.success and sg.num_iterations == so.num_iterations
    assert list(tg.accepted) == list(to.accepted)
    # (the device sums with FP64 atomics: a gradient eight orders below its start carries their rounding, run to run)
    assert np.allclose(tg.cost, to.cost, rtol=1e-9)
    assert np.allclose(tg.gradient_max_norm, to.gradient_max_norm, rtol=1e-6, atol=1e-9 * max(1.0, float(to.gradient_max_norm[0])))
    assert np.abs(qg.cam_ext - qo.cam_ext).max() <= 1e-8 and np.abs(qg.point_inverse_depth - qo.point_inverse_depth).max() <= 1e-8
    assert np.allclose(qg.intrinsics, qo.intrinsics, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("kw", [dict(), dict(loss_function_type=1, robust_loss_width=2.0), dict(constant_camera_orientation=1),
                                dict(constant_camera_position=1), dict(loss_function_type=3, robust_loss_width=3.0)],
                         ids=["trivial", "huber", "const-orientation", "const-position", "cauchy"])
def test_inverse_depth_follows_the_oracle(kw):
    g, o = _both(idp.make(8, 200, seed=5), **kw)
    _compare(g, o)
    assert g[1].final_cost < 0.5 * g[1].initial_cost


def test_constant_cameras_and_tracks():
    p = idp.make(12, 400, seed=9)
    p.cam_const = np.zeros(12, dtype=np.uint8); p.cam_const[0] = 3; p.cam_const[5] = 1; p.cam_const[7] = 2
    p.point_const = (np.arange(400) % 11 == 0).astype(np.uint8)
    g, o = _both(p)
    _compare(g, o)
    qg = g[0]
    assert np.array_equal(qg.cam_ext[0], p.cam_ext[0]) and np.array_equal(qg.cam_ext[5, :3], p.cam_ext[5, :3])
    assert np.array_equal(qg.point_inverse_depth[::11], p.point_inverse_depth[::11])


def test_mirror_api_updates_the_homogeneous_points():
    p = idp.make(8, 200, seed=13)
    rec = sfm.Reconstruction.from_flat(p)
    rec.track_reference_view = p.point_ref_cam.astype(np.int64)
    rec.track_reference_bearing = p.point_ref_bearing.copy()
    rec.inverse_depth = p.point_inverse_depth.copy()
    opts = sfm.BundleAdjustmentOptions(); opts.use_inverse_depth_parametrization = True; opts.max_num_iterations = 15
    s = sfm.BundleAdjustReconstruction(opts, rec)
    q = p.copy(); o = ol.default_options(); o.max_num_iterations = 15; o.use_inner_iterations = 0
    so, _ = ol.solve_inverse_depth(q, o)
    assert s.success and abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(rec.inverse_depth - q.point_inverse_depth).max() <= 1e-8
    assert np.abs(rec.points - idp.world_points(q)).max() <= 1e-7          # UpdateHomogeneousPoint


@pytest.mark.parametrize("mask", [0x01, 0x01 | 0x10, 0x01 | 0x02 | 0x08, 0x3f], ids=["focal", "focal+radial", "focal+aspect+pp", "all"])
def test_inverse_depth_with_free_intrinsics_follows_the_oracle(mask):
    """The intrinsics of the observing camera's group are a parameter block of every inverse-depth residual
    (bundle_adjuster.cc:594-622) and are optimised on the subset of intrinsics_to_optimize when their views are optimised
    (BundleAdjuster::AddView + AddInvTrack through the registration API): shared groups (8 cameras, 3 groups), gauge fixed
    by one constant camera and one constant position."""
    p = idp.make(8, 300, seed=21)
    p.cam_group = (np.arange(8) % 3).astype(np.int32)
    p.intrinsics[:, 5] = -0.03
    p.cam_const = np.zeros(8, dtype=np.uint8); p.cam_const[0] = 3; p.cam_const[1] = 1
    g, o = _both(p, iters=12, intrinsics_to_optimize=mask, function_tolerance=1e-12)
    _compare(g, o)
    assert not np.array_equal(g[0].intrinsics[:3, 0], p.intrinsics[:3, 0])
    assert g[1].final_cost < 0.5 * g[1].initial_cost


def test_inverse_depth_with_a_constant_group_and_huber_loss():
    p = idp.make(8, 300, seed=22)
    p.cam_group = (np.arange(8) % 2).astype(np.int32)
    p.group_const = np.array([0, 1, 0, 0, 0, 0, 0, 0], dtype=np.uint8)
    p.cam_const = np.zeros(8, dtype=np.uint8); p.cam_const[0] = 3; p.cam_const[1] = 1
    g, o = _both(p, iters=12, intrinsics_to_optimize=0x11, loss_function_type=1, robust_loss_width=2.0)
    _compare(g, o)
    assert np.array_equal(g[0].intrinsics[1], p.intrinsics[1]) and not np.array_equal(g[0].intrinsics[0], p.intrinsics[0])


@pytest.mark.parametrize("kinds", [1, 2, 4, 7], ids=["position", "gravity", "orientation", "all"])
def test_inverse_depth_with_camera_priors_follows_the_oracle(kinds):
    """AddViewPriors (bundle_adjuster.cc:289-313) in inverse-depth mode: prior rows of the optimised views, none on a view
    that is not part of the problem, a fixed cost for a prior on a constant camera."""
    p = idp.make(8, 300, seed=23)
    rng = np.random.default_rng(3)
    mask = np.zeros(8, dtype=np.uint8); mask[[0, 2, 3, 6]] = 7
    p.cam_const = np.zeros(8, dtype=np.uint8); p.cam_const[0] = 3          # its priors are a fixed cost
    info = lambda sc: np.stack([sc * (np.eye(3) + 0.05 * rng.normal(size=(3, 3))) for _ in range(8)])
    from pytheiasfm_amd import synth
    grav = np.stack([synth.angle_axis_to_matrix(w[None])[0] @ np.array([0.0, 0.0, -1.0]) for w in p.cam_ext[:, 3:]]) + 0.01 * rng.normal(size=(8, 3))
    p.set_priors(mask, position=(p.cam_ext[:, :3] + 0.02 * rng.normal(size=(8, 3)), info(4.0)),
                 gravity=(grav, info(20.0)), orientation=(p.cam_ext[:, 3:] + 0.01 * rng.normal(size=(8, 3)), info(30.0)))
    g, o = _both(p, iters=12, prior_mask=kinds)
    _compare(g, o)
    g0, o0 = _both(p, iters=12, prior_mask=0)
    assert abs(g[1].final_cost - g0[1].final_cost) > 1e-6 * g0[1].final_cost    # the rows changed the problem


def test_inverse_depth_intrinsics_and_priors_together():
    p = idp.make(10, 400, seed=24)
    p.cam_group = (np.arange(10) % 2).astype(np.int32)
    mask = np.zeros(10, dtype=np.uint8); mask[[1, 4, 7]] = 1
    p.set_priors(mask, position=(p.cam_ext[:, :3] + 0.01, np.tile(5.0 * np.eye(3), (10, 1, 1))))
    p.cam_const = np.zeros(10, dtype=np.uint8); p.cam_const[0] = 3
    g, o = _both(p, iters=15, intrinsics_to_optimize=0x11, prior_mask=1, loss_function_type=3, robust_loss_width=3.0)
    _compare(g, o)


def test_unsupported_combinations_are_rejected():
    p = idp.make(4, 40, seed=2)
    h = ba.BaHandle(p.copy(), ba.default_options())
    with pytest.raises(capi.TheiaHipError):
        h.covariance(points=True)                          # the handle of this mode: create / reset / run / download only
    h.close()
    bad = p.copy(); bad.point_inverse_depth[3] = -1.0
    with pytest.raises(capi.TheiaHipError):
        ba.solve(bad, ba.default_options())


@pytest.mark.parametrize("intr", [0, 0x11])
def test_inverse_depth_tile_sparse_reduced_system(intr, monkeypatch):
    """64 cameras on a ring (n = 384 + 10 per free group: six 64-wide tiles, banded co-visibility): the reduced system is
    factored by the tile-sparse level-scheduled Cholesky of the main path (the plan built from the tracks' reference /
    observing cameras and groups); the LM trace equals the oracle's dense LM and the dense schedule of the same library."""
    p = idp.make(64, 1500, seed=57)
    p.cam_group = (np.arange(64) % 2).astype(np.int32)
    g, o = _both(p, iters=8, intrinsics_to_optimize=intr)
    _compare(g, o)
    monkeypatch.setenv("THEIA_HIP_INVDEPTH_DENSE", "1")
    q = p.copy(); od = ba.default_options(); od.max_num_iterations = 8; od.use_inner_iterations = 0; od.intrinsics_to_optimize = intr
    sd, td = ba.solve(q, od)
    assert sd.num_iterations == g[1].num_iterations and np.allclose(td.cost, g[2].cost, rtol=1e-11)
    assert np.abs(q.cam_ext - g[0].cam_ext).max() <= 1e-9


def test_inverse_depth_handle_equals_the_one_shot_solve():
    """theia_hip_ba_create / reset_parameters / run / download with THEIA_BA_FLAG_INVERSE_DEPTH (intrinsics free, a prior):
    the handle's run equals theia_hip_ba_solve of the same problem, a reset with other parameters equals the one-shot solve
    of those, a second run continues from the first one's end, and the problem cache of the mirror re-uses the handle."""
    p = idp.make(10, 400, seed=41)
    mask = np.zeros(10, dtype=np.uint8); mask[[2, 6]] = 1
    p.set_priors(mask, position=(p.cam_ext[:, :3] + 0.01, np.tile(5.0 * np.eye(3), (10, 1, 1))))
    o = ba.default_options(); o.max_num_iterations = 6; o.use_inner_iterations = 0; o.intrinsics_to_optimize = 0x11; o.prior_mask = 1

    def close(a, b):
        assert np.allclose(a.cam_ext, b.cam_ext, rtol=1e-9, atol=1e-11) and np.allclose(a.intrinsics, b.intrinsics, rtol=1e-9, atol=1e-12)
        assert np.allclose(a.point_inverse_depth, b.point_inverse_depth, rtol=1e-9, atol=1e-12)

    one = p.copy(); s1, t1 = ba.solve(one, o)
    hp = p.copy(); h = ba.BaHandle(hp, o)
    s2, t2 = h.run(); h.download(hp)
    assert s2.num_iterations == s1.num_iterations and np.allclose(t2.cost, t1.cost, rtol=1e-10)
    close(hp, one)
    # continue: the next run starts where the first ended = a one-shot solve from that state
    cont = one.copy(); s3, _ = ba.solve(cont, o)
    s4, _ = h.run(); h.download(hp)
    assert s4.num_iterations == s3.num_iterations and abs(s4.final_cost - s3.final_cost) <= 1e-9 * s3.final_cost
    close(hp, cont)
    # reset with perturbed parameters
    q = p.copy(); q.cam_ext[:, :3] += 0.003; q.point_inverse_depth *= 1.01
    ref = q.copy(); s5, _ = ba.solve(ref, o)
    h.reset(q); s6, _ = h.run(); h.download(q)
    assert s6.num_iterations == s5.num_iterations
    close(q, ref)
    # snapshot / restore (round 6: the same handle API as the main path): restore + run reproduces the run after the snapshot
    # (to rounding: this mode assembles the reduced system with FP64 atomics, whose order is not fixed)
    h.reset(p); h.snapshot()
    sa, _ = h.run(); ha = p.copy(); h.download(ha)
    h.restore(); sb, _ = h.run(); hb = p.copy(); h.download(hb)
    assert sa.num_iterations == sb.num_iterations and abs(sa.final_cost - sb.final_cost) <= 1e-12 * sa.final_cost
    assert np.abs(ha.cam_ext - hb.cam_ext).max() <= 1e-10 and np.abs(ha.point_inverse_depth - hb.point_inverse_depth).max() <= 1e-10
    info = h.plan_info()
    assert info["n"] > 0 and info["fused_runs"] == 0
    o2 = ba.default_options(); o2.max_num_iterations = 6; o2.use_inner_iterations = 0; o2.intrinsics_to_optimize = 0; o2.prior_mask = 1
    with pytest.raises(capi.TheiaHipError):
        h.set_options(o2)                                  # structural option changed
    h.close()
    cache = ba.ProblemCache(1)
    a = p.copy(); cache.solve(a, o); b = p.copy(); b.cam_ext[:, :3] += 0.003; cache.solve(b, o)
    assert (cache.misses, cache.hits) == (1, 1)
    close(a, one)
    cache.clear()


def test_mirror_api_passes_the_view_priors():
    """BundleAdjustReconstruction with use_inverse_depth_parametrization runs AddViewPriors (bundle_adjustment.cc:194-198):
    the mirror hands the priors of the reconstruction's views to the library; intrinsics stay constant in this entry
    point whatever intrinsics_to_optimize says (no view goes through AddView: bundle_adjuster.cc:441-459)."""
    p = idp.make(8, 200, seed=31)
    mask = np.zeros(8, dtype=np.uint8); mask[[1, 2, 5, 6]] = 1     # four positions: more than a similarity can absorb
    sign = np.where(np.arange(8) % 2 == 0, 1.0, -1.0)[:, None]      # not a rigid shift of the scene (that would cost nothing)
    pri = (p.cam_ext[:, :3] + 0.3 * sign, np.tile(20.0 * np.eye(3), (8, 1, 1)))
    p.set_priors(mask, position=pri)
    rec = sfm.Reconstruction.from_flat(p)
    rec.track_reference_view = p.point_ref_cam.astype(np.int64)
    rec.track_reference_bearing = p.point_ref_bearing.copy()
    rec.inverse_depth = p.point_inverse_depth.copy()
    rec.view_prior_mask = mask.copy(); rec.view_priors = dict(position=pri)
    opts = sfm.BundleAdjustmentOptions(); opts.use_inverse_depth_parametrization = True; opts.max_num_iterations = 15
    opts.use_position_priors = True
    opts.intrinsics_to_optimize = 0x11
    s = sfm.BundleAdjustReconstruction(opts, rec)
    q = p.copy(); o = ol.default_options(); o.max_num_iterations = 15; o.use_inner_iterations = 0; o.prior_mask = 1
    so, _ = ol.solve_inverse_depth(q, o)
    assert s.success and abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(rec.cam_ext - q.cam_ext).max() <= 1e-8 and np.array_equal(rec.group_intrinsics, p.intrinsics)
    q0 = p.copy(); o.prior_mask = 0
    s0, _ = ol.solve_inverse_depth(q0, o)
    assert abs(s0.final_cost - so.final_cost) > 1e-6 * so.final_cost
