"""An INDEPENDENT restatement of the BA solve for small problems, used to pin the oracle's and the library's LM trajectories
(tests/test_oracle_ba.py, tests/test_parity_gpu.py):
  * the reprojection residual (camera/reprojection_error.h:54-110 + pinhole_camera_model.h:181-260) written in torch (float64) and
    differentiated by torch.func -- reverse-mode autodiff, no code or derivation shared with oracle/ (forward-mode Jets) or csrc/
    (closed forms);
  * the FULL normal equations (intrinsics, cameras and points together, dense, numpy Cholesky) instead of the Schur complement;
  * Ceres' trust-region rules (TrustRegionMinimizer + LevenbergMarquardtStrategy, 2.2) restated from their description: Jacobi
    scaling 1 / (1 + |column|) fixed at the initial point, D^2 = clamp(diag(J^T J), 1e-6, 1e32) / radius, model cost change
    -m . (r + m / 2), step valid iff it is positive (else radius /= 2, 4, 8 ...; five in a row fail), parameter / function
    tolerance before the acceptance test, rho > 1e-3 accepts with radius /= max(1 / 3, 1 - (2 rho - 1)^3) capped at BundleAdjustmentOptions' max_trust_region_radius
    (1e12, bundle_adjustment.h), gradient tolerance after
    successful steps only;
  * optionally: the homogeneous point on ceres::SphereManifold<4> (Householder form of Plus and its Jacobian, written from the
    manifold's definition: x (+) d = |x| H(x)^T [sin|d| d / |d|; cos|d|]), the robust losses HUBER / SOFTLONE / CAUCHY / ARCTAN / TUKEY and Theia's TRUNCATED with Ceres' corrector
    for rho'' <= 0 (residual and Jacobian scaled by sqrt(rho')), and shared intrinsics blocks with a subset of free parameters
    (SubsetManifold) and the reference's lower bound on the focal length (bundle_adjuster.cc:406-409) by projection.
  * optionally the position / gravity / orientation rows of AddViewPriors (no loss), written from their definitions.
Pinhole and double-sphere cameras."""
import numpy as np
import torch
from torch.func import jacrev, vmap


def _residual(cam, pt, intr, uv, ds):
    """ds > 0.5: the double-sphere model (double_sphere_camera_model.h:161-249; Usenko et al. 2018, eq. 40-45, written from the
    paper's formula: pi(x) = (x, y) / (alpha d2 + (1 - alpha)(xi d1 + z))), else pinhole with two radial terms."""
    C, w = cam[:3], cam[3:6]
    p = pt[:3] - pt[3] * C
    th = torch.sqrt((w * w).sum())
    k = w / th
    q = p * torch.cos(th) + torch.linalg.cross(k, p) * torch.sin(th) + k * (k @ p) * (1.0 - torch.cos(th))
    # pinhole
    x, y = q[0] / q[2], q[1] / q[2]
    r2 = x * x + y * y
    d = 1.0 + r2 * (intr[5] + intr[6] * r2)
    xp, yp = x * d, y * d
    # double sphere
    xi, alpha = intr[5], intr[6]
    d1 = torch.sqrt((q * q).sum())
    kk = xi * d1 + q[2]
    d2 = torch.sqrt(q[0] * q[0] + q[1] * q[1] + kk * kk)
    nrm = alpha * d2 + (1.0 - alpha) * kk
    xd = torch.where(ds > 0.5, q[0] / nrm, xp)
    yd = torch.where(ds > 0.5, q[1] / nrm, yp)
    u = intr[0] * xd + intr[2] * yd + intr[3]
    v = intr[0] * intr[1] * yd + intr[4]
    return torch.stack([u - uv[0], v - uv[1]])


def _ds_valid(cam, pt, intr):
    """the projection's domain (DoubleSphereCameraModel::DistortPoint returns false outside it: the evaluation fails)"""
    C, w = cam[:3], cam[3:6]
    p = pt[:3] - pt[3] * C
    th = torch.sqrt((w * w).sum())
    k = w / th
    q = p * torch.cos(th) + torch.linalg.cross(k, p) * torch.sin(th) + k * (k @ p) * (1.0 - torch.cos(th))
    xi, alpha = intr[5], intr[6]
    w1 = torch.where(alpha > 0.5, (1.0 - alpha) / alpha, alpha / (1.0 - alpha))
    w2 = (w1 + xi) / torch.sqrt(2.0 * w1 * xi + xi * xi + 1.0)
    return q[2] > -w2 * torch.sqrt((q * q).sum())


_jac = vmap(jacrev(_residual, argnums=(0, 1, 2)))
_val = vmap(_ds_valid)
_res = vmap(_residual)
CAM_DOUBLE_SPHERE = 5


def project_to_bounds(model, v):
    """the reference's parameter bounds (bundle_adjuster.cc:406-427): focal >= 1; double sphere xi in [-1, 1], alpha in [0, 1]"""
    v[0] = max(v[0], 1.0)
    if model == CAM_DOUBLE_SPHERE:
        v[5] = min(max(v[5], -1.0), 1.0); v[6] = min(max(v[6], 0.0), 1.0)


def _rot(w):
    th = torch.sqrt((w * w).sum())
    k = w / th
    K = torch.stack([torch.stack([torch.zeros_like(th), -k[2], k[1]]), torch.stack([k[2], torch.zeros_like(th), -k[0]]),
                     torch.stack([-k[1], k[0], torch.zeros_like(th)])])
    return torch.eye(3, dtype=w.dtype) + torch.sin(th) * K + (1.0 - torch.cos(th)) * (K @ K)


def _prior(cam, kind, vec, S):
    """AddViewPriors' rows (position_error.h:52-60, gravity_error.h:53-65, orientation_error.h:53-63), no loss:
    position S (prior - C); gravity S (R(w) (0, 0, -1) - prior); orientation S log(R(w) R(prior)^T)."""
    if kind == 1:
        return S @ (vec - cam[:3])
    if kind == 2:
        return S @ (_rot(cam[3:6]) @ torch.tensor([0.0, 0.0, -1.0], dtype=cam.dtype) - vec)
    E = _rot(cam[3:6]) @ _rot(vec).T
    v = 0.5 * torch.stack([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]])      # sin(theta) * axis
    sn = torch.sqrt((v * v).sum())
    theta = torch.atan2(sn, 0.5 * (E[0, 0] + E[1, 1] + E[2, 2] - 1.0))
    return S @ (v * (theta / sn))


def householder(x):
    """v, beta with (I - beta v v^T) x = |x| e_n (the last axis), v_n = 1: the reflection SphereManifold is built on."""
    n = len(x)
    sigma = float(x[:n - 1] @ x[:n - 1])
    v = x.copy(); v[n - 1] = 1.0
    xp = x[n - 1]
    if sigma <= np.finfo(float).eps ** 2:
        return v, (2.0 if xp < 0 else 0.0)
    mu = np.sqrt(xp * xp + sigma)
    vp = xp - mu if xp <= 0 else -sigma / (xp + mu)
    beta = 2.0 * vp * vp / (sigma + vp * vp)
    v[:n - 1] /= vp
    return v, beta


def sphere_plus(x, d):
    nd = np.linalg.norm(d)
    if nd == 0.0:
        return x.copy()
    v, beta = householder(x)
    y = np.append(np.sin(nd) / nd * d, np.cos(nd))
    return np.linalg.norm(x) * (y - v * (beta * (v @ y)))


def sphere_plus_jacobian(x):
    v, beta = householder(x)
    H = np.eye(len(x)) - beta * np.outer(v, v)
    return np.linalg.norm(x) * H[:, :len(x) - 1]


def loss(kind, a, s):
    """rho(s), rho'(s) of ceres::{Trivial, Huber, Cauchy}Loss at squared norm s."""
    if kind == "huber":
        b = a * a
        return np.where(s > b, 2.0 * a * np.sqrt(np.maximum(s, 1e-300)) - b, s), np.where(s > b, a / np.sqrt(np.maximum(s, 1e-300)), 1.0)
    if kind == "cauchy":
        b = a * a
        return b * np.log1p(s / b), 1.0 / (1.0 + s / b)
    if kind == "softl1":                                  # 2 b (sqrt(1 + s / b) - 1)
        b = a * a
        return 2.0 * b * (np.sqrt(1.0 + s / b) - 1.0), 1.0 / np.sqrt(1.0 + s / b)
    if kind == "arctan":                                  # a atan(s / a)
        return a * np.arctan2(s, a), 1.0 / (1.0 + (s / a) ** 2)
    if kind == "tukey":                                   # a^2 / 3 (1 - (1 - s / a^2)^3) inside s <= a^2, constant outside
        b = a * a
        inside = s <= b
        v = np.where(inside, 1.0 - s / b, 0.0)
        return np.where(inside, b / 3.0 * (1.0 - v ** 3), b / 3.0), np.where(inside, v * v, 0.0)
    if kind == "truncated":                               # theia TruncatedLoss (loss_functions.cc:40-44): min(s, a^2)
        b = a * a
        return np.minimum(s, b), np.where(s < b, 1.0, 0.0)
    return s, np.ones_like(s)


class Problem:
    def __init__(self, flat, manifold, loss_kind, loss_width, free_intr, prior_mask=0):
        self.cam = np.array(flat.cam_ext, dtype=np.float64)
        self.pts = np.array(flat.points, dtype=np.float64)
        self.intr = np.array(flat.intrinsics, dtype=np.float64)
        self.grp = np.asarray(flat.cam_group)
        self.model = np.asarray(flat.group_model)
        assert np.all((self.model == 0) | (self.model == CAM_DOUBLE_SPHERE))
        self.oc = np.asarray(flat.obs_cam); self.op = np.asarray(flat.obs_pt)
        self.uv = torch.tensor(np.asarray(flat.obs_uv), dtype=torch.float64)
        self.ds = torch.tensor((self.model[self.grp[self.oc]] == CAM_DOUBLE_SPHERE).astype(np.float64))
        cc = np.asarray(flat.cam_const) if flat.cam_const is not None else np.zeros(len(self.grp), np.uint8)
        assert np.all((cc == 0) | (cc == 3)), "whole cameras constant or free"
        self.manifold, self.loss_kind, self.loss_width = manifold, loss_kind, loss_width
        self.free = list(free_intr) if free_intr else []
        self.pd = 3 if manifold else 4
        ng = self.intr.shape[0] if self.free else 0
        self.col_intr = {g: len(self.free) * g for g in range(ng)}
        off = len(self.free) * ng
        self.var_cam = np.nonzero(cc == 0)[0]
        self.col_cam = {int(c): off + 6 * i for i, c in enumerate(self.var_cam)}
        self.off_pts = off + 6 * len(self.var_cam)
        self.n = self.off_pts + self.pd * self.pts.shape[0]
        if self.free:
            for g in range(self.intr.shape[0]):
                project_to_bounds(self.model[g], self.intr[g])
        # camera priors in use: the camera's bit AND the option's bit (bundle_adjuster.cc:159-172); variable cameras only here
        self.prior_rows = []
        if prior_mask and flat.cam_prior_mask is not None:
            for c in range(len(self.grp)):
                for bit, name in ((1, "position"), (2, "gravity"), (4, "orientation")):
                    if (int(flat.cam_prior_mask[c]) & prior_mask & bit) and name in flat.priors:
                        assert c in self.col_cam, "priors on variable cameras only"
                        v, S = flat.priors[name]
                        self.prior_rows.append((c, bit, torch.tensor(np.asarray(v[c], dtype=np.float64)), torch.tensor(np.asarray(S[c], dtype=np.float64))))

    def evaluate(self, cam, pts, intr, jac):
        """cost, corrected residuals, corrected tangent-space Jacobian (or None)"""
        tc, tp = torch.tensor(cam[self.oc]), torch.tensor(pts[self.op])
        ti = torch.tensor(intr[self.grp[self.oc]])
        r = _res(tc, tp, ti, self.uv, self.ds).numpy()
        bad = (~_val(tc, tp, ti).numpy()) & (self.ds.numpy() > 0.5)
        if bad.any():
            r = r.copy(); r[bad] = np.nan                          # a failed evaluation: the caller sees a non-finite cost
        s = (r * r).sum(1)
        rho, rho1 = loss(self.loss_kind, self.loss_width, s)
        cost = 0.5 * float(rho.sum())
        sr = np.sqrt(rho1)
        rc = (r * sr[:, None]).reshape(-1)
        npr = 3 * len(self.prior_rows)
        if npr:
            rp = np.concatenate([_prior(torch.tensor(cam[c]), kind, v, S).numpy() for c, kind, v, S in self.prior_rows])
            cost += 0.5 * float(rp @ rp)
            rc = np.concatenate([rc, rp])
        if not jac:
            return cost, rc, None
        jc, jp, ji = (t.numpy() for t in _jac(tc, tp, ti, self.uv, self.ds))
        J = np.zeros((len(rc), self.n))
        PJ = {}
        for i in range(len(self.oc)):
            c, p, g = int(self.oc[i]), int(self.op[i]), int(self.grp[self.oc[i]])
            rows = slice(2 * i, 2 * i + 2)
            if self.free:
                J[rows, self.col_intr[g]:self.col_intr[g] + len(self.free)] += sr[i] * ji[i][:, self.free]
            if c in self.col_cam:
                J[rows, self.col_cam[c]:self.col_cam[c] + 6] = sr[i] * jc[i]
            if self.manifold:
                if p not in PJ:
                    PJ[p] = sphere_plus_jacobian(pts[p])
                J[rows, self.off_pts + 3 * p:self.off_pts + 3 * p + 3] = sr[i] * (jp[i] @ PJ[p])
            else:
                J[rows, self.off_pts + 4 * p:self.off_pts + 4 * p + 4] = sr[i] * jp[i]
        for k, (c, kind, v, S) in enumerate(self.prior_rows):
            Jp = jacrev(lambda x: _prior(x, kind, v, S))(torch.tensor(cam[c])).numpy()
            J[len(rc) - npr + 3 * k:len(rc) - npr + 3 * k + 3, self.col_cam[c]:self.col_cam[c] + 6] = Jp
        return cost, rc, J

    def plus(self, cam, pts, intr, delta):
        cam, pts, intr = cam.copy(), pts.copy(), intr.copy()
        for g, col in self.col_intr.items():
            intr[g, self.free] += delta[col:col + len(self.free)]
            project_to_bounds(self.model[g], intr[g])               # ParameterBlock::Plus projects onto the bounds
        for c, col in self.col_cam.items():
            cam[c] += delta[col:col + 6]
        dp = delta[self.off_pts:].reshape(-1, self.pd)
        if self.manifold:
            for p in range(len(pts)):
                pts[p] = sphere_plus(pts[p], dp[p])
        else:
            pts += dp
        return cam, pts, intr

    def norms(self, cam, pts, intr, cam2=None, pts2=None, intr2=None):
        """|x| over the variable blocks, or |x - x2| (Ceres: in the ambient parameters)"""
        acc = 0.0
        for g in self.col_intr:
            acc += float(((intr[g] - (intr2[g] if intr2 is not None else 0.0)) ** 2).sum())
        for c in self.col_cam:
            acc += float(((cam[c] - (cam2[c] if cam2 is not None else 0.0)) ** 2).sum())
        acc += float(((pts - (pts2 if pts2 is not None else 0.0)) ** 2).sum())
        return np.sqrt(acc)


def solve(flat, max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8,
          max_trust_region_radius=1e12, manifold=False, loss_kind="trivial", loss_width=1.0, free_intr=None, prior_mask=0):
    """Returns the trace -- (cost, gradient max norm, step norm, radius, accepted) per entry, as the oracle and the library record
    it -- and the final (cameras, points, intrinsics)."""
    P = Problem(flat, manifold, loss_kind, loss_width, free_intr, prior_mask)
    cam, pts, intr = P.cam, P.pts, P.intr
    x_cost, r, J = P.evaluate(cam, pts, intr, True)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    Js = J * scale
    gmax = float(np.abs(J.T @ r).max())
    x_norm = P.norms(cam, pts, intr)
    radius, decrease = 1e4, 2.0
    trace = [(x_cost, gmax, 0.0, radius, 1)]
    it, invalid, successful = 0, 0, True
    while True:
        if it >= max_num_iterations or (successful and gmax <= gradient_tolerance) or radius <= 1e-32:
            break
        it += 1
        D2 = np.clip((Js * Js).sum(0), 1e-6, 1e32) / radius
        A = Js.T @ Js + np.diag(D2)
        try:
            Lc = np.linalg.cholesky(A)
            y = np.linalg.solve(Lc.T, np.linalg.solve(Lc, Js.T @ r))
            ok = bool(np.all(np.isfinite(y)))
        except np.linalg.LinAlgError:
            ok = False
        mcc = 0.0
        if ok:
            m = Js @ (-y)
            mcc = -float(m @ (r + m / 2.0))
            ok = mcc > 0.0
        if not ok:
            invalid += 1
            if invalid >= 5:
                break
            radius /= decrease; decrease *= 2.0; successful = False
            trace.append((x_cost, gmax, 0.0, radius, 0))
            continue
        invalid = 0
        ccam, cpts, cintr = P.plus(cam, pts, intr, -y * scale)
        cand, rc, _ = P.evaluate(ccam, cpts, cintr, False)
        if not np.all(np.isfinite(rc)):
            cand = np.finfo(np.float64).max                     # Ceres: a candidate that fails to evaluate costs DBL_MAX
        step_norm = P.norms(cam, pts, intr, ccam, cpts, cintr)
        if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
            trace.append((cand, gmax, step_norm, radius, 0)); break
        change = x_cost - cand
        if abs(change) <= function_tolerance * x_cost:
            trace.append((cand, gmax, step_norm, radius, 0)); break
        rho = change / mcc
        if rho > 1e-3:
            cam, pts, intr = ccam, cpts, cintr
            x_norm = P.norms(cam, pts, intr)
            x_cost, r, J = P.evaluate(cam, pts, intr, True)
            Js = J * scale
            gmax = float(np.abs(J.T @ r).max())
            radius = min(max_trust_region_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            decrease = 2.0; successful = True
            trace.append((x_cost, gmax, step_norm, radius, 1))
        else:
            radius /= decrease; decrease *= 2.0; successful = False
            trace.append((cand, gmax, step_norm, radius, 0))
    return trace, cam, pts, intr
