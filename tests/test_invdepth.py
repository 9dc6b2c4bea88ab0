"""CPU: the oracle of the inverse-depth parametrisation (InvReprojectionError / InvReprojectionPoseError,
camera/reprojection_error.h:173-286; dense LM in oracle/ba_oracle.cpp) against an independent numpy + scipy statement of
the same least-squares problem."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from tests import invdepth as idp
from tests import oracle_lib as ol


def _residuals(p, cam, rho):
    """Plain numpy: x_other = R_o (R_r^T (b / rho) + c_r - c_o), pinhole projection; the reference view sees b / rho."""
    out = np.zeros((len(p.obs_cam), 2))
    for i, (c, t) in enumerate(zip(p.obs_cam, p.obs_pt)):
        cr = p.point_ref_cam[t]
        pr = p.point_ref_bearing[t] / rho[t]
        if c == cr:
            q = pr
        else:
            Xw = idp.aa_to_rot(cam[cr, 3:]).T @ pr + cam[cr, :3]
            q = idp.aa_to_rot(cam[c, 3:]) @ (Xw - cam[c, :3])
        f, a, s, px, py, k1, k2 = p.intrinsics[p.cam_group[c]][:7]
        x, y = q[0] / q[2], q[1] / q[2]
        r2 = x * x + y * y
        d = 1.0 + r2 * (k1 + k2 * r2)
        out[i] = (f * x * d + s * y * d + px - p.obs_uv[i, 0], f * a * y * d + py - p.obs_uv[i, 1])
    return out


def test_oracle_reaches_the_scipy_minimum_of_the_same_problem():
    p = idp.make(5, 40, seed=3)
    p.cam_const = np.zeros(5, dtype=np.uint8); p.cam_const[0] = 3; p.cam_const[1] = 1      # gauge: camera 0, position of camera 1
    o = ol.default_options(); o.max_num_iterations = 50; o.use_inner_iterations = 0
    o.function_tolerance = 1e-14; o.parameter_tolerance = 1e-14; o.gradient_tolerance = 1e-14
    q = p.copy()
    s, tr = ol.solve_inverse_depth(q, o)
    assert s.success and s.final_cost < s.initial_cost
    assert abs(s.initial_cost - 0.5 * (_residuals(p, p.cam_ext, p.point_inverse_depth) ** 2).sum()) <= 1e-9 * s.initial_cost

    free = np.ones((5, 6), dtype=bool); free[0] = False; free[1, :3] = False
    x0 = np.concatenate([p.cam_ext[free], p.point_inverse_depth])

    def fun(x):
        cam = p.cam_ext.copy(); cam[free] = x[: free.sum()]
        return _residuals(p, cam, x[free.sum():]).reshape(-1)
    ls = least_squares(fun, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert abs(ls.cost - s.final_cost) <= 1e-9 * s.final_cost
    cam = p.cam_ext.copy(); cam[free] = ls.x[: free.sum()]
    assert np.abs(cam - q.cam_ext).max() <= 1e-6 and np.abs(ls.x[free.sum():] - q.point_inverse_depth).max() <= 1e-6


def test_world_points_of_the_solution_reproject_at_noise_level():
    p = idp.make(8, 200, seed=5)
    o = ol.default_options(); o.max_num_iterations = 20; o.use_inner_iterations = 0
    q = p.copy()
    s, _ = ol.solve_inverse_depth(q, o)
    r = _residuals(q, q.cam_ext, q.point_inverse_depth)
    assert abs(0.5 * (r ** 2).sum() - s.final_cost) <= 1e-9 * s.final_cost
    assert np.sqrt((r ** 2).sum(axis=1).mean()) < 3.0          # 1 px feature noise, the reference ray carries its own
    X = idp.world_points(q)
    assert np.isfinite(X).all()


def test_oracle_with_free_intrinsics_and_a_position_prior_reaches_the_scipy_minimum():
    """FOCAL_LENGTH | RADIAL_DISTORTION of the shared group free (a parameter block of every residual of its cameras,
    bundle_adjuster.cc:594-622) and a position prior on two cameras (AddViewPriors, sqrt information 3 I): the dense-LM
    oracle against scipy on the same residual vector (reprojection rows + 3 prior rows per camera)."""
    p = idp.make(6, 60, seed=7)
    p.intrinsics[:, 5] = -0.02          # some distortion to estimate
    p.cam_group[:] = 0                   # one shared group
    p.cam_const = np.zeros(6, dtype=np.uint8); p.cam_const[0] = 3; p.cam_const[1] = 1
    mask = np.zeros(6, dtype=np.uint8); mask[2] = 1; mask[4] = 1
    prior = p.cam_ext[:, :3] + 0.01
    info = np.tile(3.0 * np.eye(3), (6, 1, 1))
    p.set_priors(mask, position=(prior, info))
    o = ol.default_options(); o.max_num_iterations = 80; o.use_inner_iterations = 0
    o.intrinsics_to_optimize = 0x01 | 0x10; o.prior_mask = 1
    o.function_tolerance = 1e-15; o.parameter_tolerance = 1e-15; o.gradient_tolerance = 1e-15
    q = p.copy()
    s, tr = ol.solve_inverse_depth(q, o)
    assert s.success and s.final_cost < s.initial_cost
    assert not np.array_equal(q.intrinsics[0, [0, 5, 6]], p.intrinsics[0, [0, 5, 6]])
    assert np.array_equal(q.intrinsics[0, [1, 2, 3, 4]], p.intrinsics[0, [1, 2, 3, 4]])    # not in the subset

    free = np.ones((6, 6), dtype=bool); free[0] = False; free[1, :3] = False
    nf = int(free.sum())

    def fun(x):
        pp = p.copy()
        cam = p.cam_ext.copy(); cam[free] = x[:nf]
        pp.intrinsics[0, [0, 5, 6]] = x[nf:nf + 3]
        r = _residuals(pp, cam, x[nf + 3:]).reshape(-1)
        pr = np.concatenate([3.0 * (prior[c] - cam[c, :3]) for c in (2, 4)])    # PositionError: sqrt_info (prior - position)
        return np.concatenate([r, pr])
    x0 = np.concatenate([p.cam_ext[free], p.intrinsics[0, [0, 5, 6]], p.point_inverse_depth])
    assert abs(0.5 * (fun(x0) ** 2).sum() - s.initial_cost) <= 1e-9 * s.initial_cost
    ls = least_squares(fun, x0, method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=400)
    assert abs(ls.cost - s.final_cost) <= 1e-7 * s.final_cost, (ls.cost, s.final_cost)
    assert np.abs(ls.x[nf:nf + 3] - q.intrinsics[0, [0, 5, 6]]).max() <= 1e-4 * np.abs(ls.x[nf:nf + 3]).max()
