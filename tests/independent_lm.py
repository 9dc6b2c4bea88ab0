"""An INDEPENDENT restatement of the BA solve for small problems, used to pin the oracle's LM trajectory (tests/test_oracle_ba.py):
  * the reprojection residual (camera/reprojection_error.h:54-110 + pinhole_camera_model.h:181-260) written in torch (float64) and
    differentiated by torch.func -- reverse-mode autodiff, no code or derivation shared with oracle/ (forward-mode Jets) or csrc/
    (closed forms);
  * the FULL normal equations (cameras and points together, dense, numpy Cholesky) instead of the Schur complement;
  * Ceres' trust-region rules (TrustRegionMinimizer + LevenbergMarquardtStrategy, 2.2) restated from their description: Jacobi
    scaling 1 / (1 + |column|) fixed at the initial point, D^2 = clamp(diag(J^T J), 1e-6, 1e32) / radius, model cost change
    -m . (r + m / 2), step valid iff it is positive (else radius /= 2, 4, 8 ...; five in a row fail), parameter / function
    tolerance before the acceptance test, rho > 1e-3 accepts with radius /= max(1 / 3, 1 - (2 rho - 1)^3), gradient tolerance after
    successful steps only.
Plain XYZW points (use_homogeneous_point_parametrization = 0: four free coordinates), pinhole cameras, trivial loss."""
import numpy as np
import torch
from torch.func import jacrev, vmap


def _residual(cam, pt, intr, uv):
    C, w = cam[:3], cam[3:6]
    p = pt[:3] - pt[3] * C
    th2 = (w * w).sum()
    th = torch.sqrt(th2)
    k = w / th
    q = p * torch.cos(th) + torch.linalg.cross(k, p) * torch.sin(th) + k * (k @ p) * (1.0 - torch.cos(th))
    x, y = q[0] / q[2], q[1] / q[2]
    r2 = x * x + y * y
    d = 1.0 + r2 * (intr[5] + intr[6] * r2)
    xd, yd = x * d, y * d
    u = intr[0] * xd + intr[2] * yd + intr[3]
    v = intr[0] * intr[1] * yd + intr[4]
    return torch.stack([u - uv[0], v - uv[1]])


_jac = vmap(jacrev(_residual, argnums=(0, 1)))
_res = vmap(_residual)


class Problem:
    def __init__(self, flat):
        self.cam = torch.tensor(np.array(flat.cam_ext), dtype=torch.float64)
        self.pts = torch.tensor(np.array(flat.points), dtype=torch.float64)
        grp = np.asarray(flat.cam_group)
        self.oc = torch.tensor(np.asarray(flat.obs_cam), dtype=torch.long)
        self.op = torch.tensor(np.asarray(flat.obs_pt), dtype=torch.long)
        self.intr = torch.tensor(np.asarray(flat.intrinsics)[grp[np.asarray(flat.obs_cam)]], dtype=torch.float64)
        self.uv = torch.tensor(np.asarray(flat.obs_uv), dtype=torch.float64)
        cc = np.asarray(flat.cam_const) if flat.cam_const is not None else np.zeros(len(grp), np.uint8)
        self.var_cam = np.nonzero(cc == 0)[0]
        assert np.all((cc == 0) | (cc == 3)), "whole cameras constant or free"
        self.col_cam = {int(c): 6 * i for i, c in enumerate(self.var_cam)}
        self.n = 6 * len(self.var_cam) + 4 * self.pts.shape[0]

    def evaluate(self, cam, pts, jac):
        r = _res(cam[self.oc], pts[self.op], self.intr, self.uv).numpy().reshape(-1)
        if not jac:
            return r, None
        jc, jp = _jac(cam[self.oc], pts[self.op], self.intr, self.uv)
        J = np.zeros((len(r), self.n))
        off = 6 * len(self.var_cam)
        for i in range(len(self.oc)):
            c, p = int(self.oc[i]), int(self.op[i])
            if c in self.col_cam:
                J[2 * i:2 * i + 2, self.col_cam[c]:self.col_cam[c] + 6] = jc[i].numpy()
            J[2 * i:2 * i + 2, off + 4 * p:off + 4 * p + 4] = jp[i].numpy()
        return r, J

    def plus(self, cam, pts, delta):
        cam = cam.clone(); pts = pts.clone()
        off = 6 * len(self.var_cam)
        for c, col in self.col_cam.items():
            cam[c] += torch.tensor(delta[col:col + 6])
        pts += torch.tensor(delta[off:].reshape(-1, 4))
        return cam, pts

    def state_norm(self, cam, pts):
        return float(np.sqrt(sum(float((cam[c] ** 2).sum()) for c in self.col_cam) + float((pts ** 2).sum())))


def solve(flat, max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8):
    """Returns a list of (cost, gradient max norm, step norm, radius, accepted) per trace entry -- the entries the oracle and the
    library record -- and the final (cameras, points)."""
    P = Problem(flat)
    cam, pts = P.cam, P.pts
    r, J = P.evaluate(cam, pts, True)
    x_cost = 0.5 * float(r @ r)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    Js = J * scale
    g = J.T @ r
    gmax = float(np.abs(g).max())
    x_norm = P.state_norm(cam, pts)
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
        delta = -y * scale
        ccam, cpts = P.plus(cam, pts, delta)
        rc, _ = P.evaluate(ccam, cpts, False)
        cand = 0.5 * float(rc @ rc) if np.all(np.isfinite(rc)) else np.inf
        step_norm = float(np.linalg.norm(delta))
        if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
            trace.append((cand, gmax, step_norm, radius, 0)); break
        change = x_cost - cand
        if abs(change) <= function_tolerance * x_cost:
            trace.append((cand, gmax, step_norm, radius, 0)); break
        rho = change / mcc
        if rho > 1e-3:
            cam, pts = ccam, cpts
            x_norm = P.state_norm(cam, pts)
            r, J = P.evaluate(cam, pts, True)
            x_cost = 0.5 * float(r @ r)
            Js = J * scale
            gmax = float(np.abs(J.T @ r).max())
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            decrease = 2.0; successful = True
            trace.append((x_cost, gmax, step_norm, radius, 1))
        else:
            radius /= decrease; decrease *= 2.0; successful = False
            trace.append((cand, gmax, step_norm, radius, 0))
    return trace, cam.numpy(), pts.numpy()
