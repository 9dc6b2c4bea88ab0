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
    assert np.allclose(tg.cost, to.cost, rtol=1e-9) and np.allclose(tg.gradient_max_norm, to.gradient_max_norm, rtol=1e-6, atol=1e-9)
    assert np.abs(qg.cam_ext - qo.cam_ext).max() <= 1e-8 and np.abs(qg.point_inverse_depth - qo.point_inverse_depth).max() <= 1e-8


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


def test_unsupported_combinations_are_rejected():
    p = idp.make(4, 40, seed=2)
    o = ba.default_options(); o.intrinsics_to_optimize = 1
    with pytest.raises(capi.TheiaHipError):
        ba.solve(p.copy(), o)
    with pytest.raises(capi.TheiaHipError):
        ba.BaHandle(p.copy(), ba.default_options())       # no handle API in this mode
    bad = p.copy(); bad.point_inverse_depth[3] = -1.0
    with pytest.raises(capi.TheiaHipError):
        ba.solve(bad, ba.default_options())
