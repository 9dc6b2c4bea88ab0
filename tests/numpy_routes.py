"""Independent numerical routes (numpy / LAPACK only) for the RANSAC legs of BASELINE configs[4], used to COUNT how often a
different implementation of the same estimator ends at the same inlier set as the device (tests/test_independent_routes*.py).
Nothing here shares code with oracle/ or csrc/: the five-point solver is the least-squares-fitted action matrix of
test_parity_gpu.py, the DLS solver builds its Macaulay system from the definitions (dls_pnp.cc:67-200 read as mathematics:
cost matrix, Cayley rotation, Jacobian cubics, resultant with one random linear form) with np.linalg.solve / np.linalg.eig
(LAPACK dgesv / dgeev) where the oracle and the device run a hand-written partial-pivot LU and hqr2.  The RANSAC loop is
the reference's (sample_consensus_estimator.h:300-415) with InlierSupport; the sample stream comes from the libstdc++-pinned
sampler of tests/oracle_lib.py."""
import itertools

import numpy as np


# ------------------------------------------------------------------------------------------------ polynomials
def _pmul(a, b):
    out = {}
    for ea, ca in a.items():
        for eb, cb in b.items():
            e = (ea[0] + eb[0], ea[1] + eb[1], ea[2] + eb[2])
            out[e] = out.get(e, 0.0) + ca * cb
    return out


def _padd(a, b, s=1.0):
    out = dict(a)
    for e, c in b.items():
        out[e] = out.get(e, 0.0) + s * c
    return out


def _pdiff(a, v):
    out = {}
    for e, c in a.items():
        if e[v]:
            f = list(e); f[v] -= 1
            out[tuple(f)] = out.get(tuple(f), 0.0) + e[v] * c
    return out


class _DlsSystem:
    """Everything of the DLS polynomial system that does not depend on the data, built once."""

    def __init__(self):
        s = [{(1, 0, 0): 1.0}, {(0, 1, 0): 1.0}, {(0, 0, 1): 1.0}]
        ss = {}
        for v in range(3):
            ss = _padd(ss, _pmul(s[v], s[v]))
        one = {(0, 0, 0): 1.0}
        skew = [[None, (2, -1.0), (1, 1.0)], [(2, 1.0), None, (0, -1.0)], [(1, -1.0), (0, 1.0), None]]   # [s]x
        rbar = []
        for a in range(3):
            for b in range(3):
                p = _padd(one, ss, -1.0) if a == b else {}
                if a != b:
                    ax, sg = skew[a][b]
                    p = _padd(p, s[ax], -2.0 * sg)
                p = _padd(p, _pmul(s[a], s[b]), 2.0)
                rbar.append(p)      # row-major entries of Cbar = (1 - s.s) I - 2 [s]x + 2 s s^T
        # f_v = d/ds_v sum_ab D_ab rbar_a rbar_b is linear in D: T[v] (monomial -> 81-vector)
        cub = [e for e in itertools.product(range(4), repeat=3) if sum(e) <= 3]
        self.cub = cub
        self.T = np.zeros((3, len(cub), 81))
        for a in range(9):
            for b in range(9):
                q = _pmul(rbar[a], rbar[b])
                for v in range(3):
                    for e, c in _pdiff(q, v).items():
                        self.T[v, cub.index(e), 9 * a + b] += c
        # monomials of degree <= 7: reduced ones (all exponents <= 2) first at 9 a + 3 b + c, then the rest lexicographically
        red = [(a, b, c) for a in range(3) for b in range(3) for c in range(3)]
        rest = [e for e in itertools.product(range(8), repeat=3) if sum(e) <= 7 and e not in red]
        self.mono = red + rest
        self.index = {e: i for i, e in enumerate(self.mono)}
        rows, cols, src = [], [], []
        for r, e in enumerate(self.mono):
            if r < 27:
                which, sh = 0, e
            else:
                which = 1 if e[0] >= 3 else (2 if e[1] >= 3 else 3)
                sh = list(e); sh[which - 1] -= 3; sh = tuple(sh)
            terms = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)] if which == 0 else cub
            for k, t in enumerate(terms):
                rows.append(r); cols.append(self.index[(sh[0] + t[0], sh[1] + t[1], sh[2] + t[2])]); src.append((which, k))
        self.rows, self.cols = np.array(rows), np.array(cols)
        self.src = np.array([w * 100 + k for w, k in src])

    def solve(self, feat, world, u):
        """feat [n][2], world [n][3], u [4] -> list of (R, t) with x ~ R X + t (dls_pnp.cc:147-198)."""
        n = len(feat)
        b = np.c_[feat, np.ones(n)]; b /= np.linalg.norm(b, axis=1, keepdims=True)
        nn = b[:, :, None] * b[:, None, :]
        H = np.linalg.inv(n * np.eye(3) - nn.sum(0))
        L = np.zeros((n, 3, 9))
        for r in range(3):
            L[:, r, 3 * r:3 * r + 3] = world
        Tf = H @ np.einsum("nij,njk->ik", nn - np.eye(3), L)
        W = L + Tf
        D = np.einsum("nia,nij,njb->ab", W, np.eye(3) - nn, W)
        coef = {0: np.asarray(u, dtype=np.float64)}
        for v in range(3):
            coef[v + 1] = self.T[v] @ D.ravel()
        vals = np.array([coef[s // 100][s % 100] for s in self.src])
        M = np.zeros((120, 120))
        M[self.rows, self.cols] = vals
        S = M[:27, :27] - M[:27, 27:] @ np.linalg.solve(M[27:, 27:], M[27:, :27])
        w, V = np.linalg.eig(S)
        out = []
        for i in range(27):
            if V[0, i] == 0:
                continue
            sv = V[[9, 3, 1], i] / V[0, i]
            if np.abs(sv.imag).max() >= 1e-6:
                continue
            q = np.array([1.0, -sv[0].real, -sv[1].real, -sv[2].real]); q /= np.linalg.norm(q)   # Quaterniond(1, s).inverse().normalized()
            w_, x, y, z = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w_ * z), 2 * (x * z + w_ * y)],
                          [2 * (x * y + w_ * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w_ * x)],
                          [2 * (x * z - w_ * y), 2 * (y * z + w_ * x), 1 - 2 * (x * x + y * y)]])
            t = Tf @ R.ravel()        # translation_factor * vec(R^T column-major) = rows of R in order
            if ((world @ R.T + t)[:, 2] < 0).any():
                continue
            out.append((R, t))
        return out


_dls = None


def dls_pnp(feat, world, u):
    global _dls
    if _dls is None:
        _dls = _DlsSystem()
    return _dls.solve(np.asarray(feat, dtype=np.float64), np.asarray(world, dtype=np.float64), u)


# ------------------------------------------------------------------------------------------------ relative pose
def five_point(x1, x2):
    """All real essential matrices through five correspondences, with numpy only.
    E = x E1 + y E2 + z E3 + E4 over the null space of the epipolar constraints (SVD); the ten cubic constraints
    det E = 0, 2 E E^T E - tr(E E^T) E = 0 are fitted as polynomials in (x, y, z) by least squares on random
    evaluation points; Gauss-Jordan on the cubic monomials leaves the multiplication-by-x matrix of the quotient ring,
    whose eigenvectors carry the solutions."""
    A = np.stack([np.outer(np.append(b, 1.0), np.append(a, 1.0)).ravel() for a, b in zip(x1, x2)])
    N = np.linalg.svd(A)[2][5:].reshape(4, 3, 3)

    def mono(v):
        x, y, z = v
        return np.array([x ** 3, x * x * y, x * y * y, y ** 3, x * x * z, x * y * z, y * y * z, x * z * z, y * z * z, z ** 3,
                         x * x, x * y, y * y, x * z, y * z, z * z, x, y, z, 1.0])

    def cons(v):
        E = v[0] * N[0] + v[1] * N[1] + v[2] * N[2] + N[3]
        EEt = E @ E.T
        return np.append((2.0 * EEt @ E - np.trace(EEt) * E).ravel(), np.linalg.det(E))

    rs = np.random.default_rng(12345)
    P = rs.normal(size=(80, 3))
    C = np.linalg.lstsq(np.stack([mono(v) for v in P]), np.stack([cons(v) for v in P]), rcond=None)[0].T   # 10 x 20
    B = np.linalg.solve(C[:, :10], C[:, 10:])        # reduced system: cubic monomial k = - B[k] . basis
    M = np.zeros((10, 10))
    for row, k in enumerate((0, 1, 2, 4, 5, 7)):     # x * {x^2, xy, y^2, xz, yz, z^2} = x^3, x^2 y, x y^2, x^2 z, x y z, x z^2
        M[row] = -B[k]
    M[6, 0] = M[7, 1] = M[8, 3] = M[9, 6] = 1.0      # x * {x, y, z, 1} = x^2, xy, xz, x
    w, V = np.linalg.eig(M)              # M b(x, y, z) = x b(x, y, z) at every solution: right eigenvectors
    sols = []
    for k in range(10):
        if abs(w[k].imag) > 1e-9 * max(1.0, abs(w[k])):
            continue
        v = (V[6:9, k] / V[9, k]).real
        E = v[0] * N[0] + v[1] * N[1] + v[2] * N[2] + N[3]
        sols.append(E / np.linalg.norm(E))
    return sols


def _in_front(x1h, x2h, R, pos):
    """IsTriangulatedPointInFrontOfCameras (triangulation.cc:216-232), vectorised over the rows of x1h / x2h."""
    d1 = x1h
    d2 = x2h @ R            # R^T x2
    d1s = (d1 * d1).sum(1); d2s = (d2 * d2).sum(1); d12 = (d1 * d2).sum(1)
    d1p = d1 @ pos; d2p = d2 @ pos
    return (d2s * d1p - d12 * d2p > 0) & (d12 * d1p - d1s * d2p > 0)


def relative_pose_models(x1, x2, five_point):
    """RelativePoseEstimator::EstimateModel (estimate_relative_pose.cc:75-108) on one minimal sample: (E, R, position)."""
    x1h = np.c_[x1, np.ones(len(x1))]; x2h = np.c_[x2, np.ones(len(x2))]
    out = []
    for E in five_point(x1, x2):
        U, _, Vt = np.linalg.svd(E)
        if np.linalg.det(U) < 0:
            U[:, 2] *= -1
        if np.linalg.det(Vt) < 0:
            Vt[2] *= -1
        Dm = np.array([[0.0, 1, 0], [-1, 0, 0], [0, 0, 1]])
        R1, R2 = U @ Dm @ Vt, U @ Dm.T @ Vt
        tr = U[:, 2] / np.linalg.norm(U[:, 2])
        cands = [(R1, -R1.T @ tr), (R1, R1.T @ tr), (R2, -R2.T @ tr), (R2, R2.T @ tr)]
        votes = [int(_in_front(x1h, x2h, R, p).sum()) for R, p in cands]
        k = int(np.argmax(votes))        # std::max_element: the first maximum
        if votes[k] >= 4:
            out.append((E, cands[k][0], cands[k][1]))
    return out


def relative_pose_errors(model, x1h, x2h):
    E, R, pos = model
    Ex1 = x1h @ E.T; Etx2 = x2h @ E
    num = (x2h * Ex1).sum(1) ** 2
    den = Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2
    err = num / den
    err[~_in_front(x1h, x2h, R, pos)] = np.inf
    return err


def absolute_pose_errors(model, feat, world):
    R, t = model
    pc = world @ R.T + t
    return ((pc[:, :2] / pc[:, 2:3] - feat) ** 2).sum(1)


def ransac_inlier_support(samples, fit, errors, thresh, ndata):
    """The reference's loop with InlierSupport and min_iterations = max_iterations = len(samples): the first model whose
    outlier count is strictly smaller wins.  Returns (inlier mask, best model)."""
    best_cost, best = np.inf, None
    mask = np.zeros(ndata, dtype=bool)
    for it, idx in enumerate(samples):
        for m in fit(it, idx):
            e = errors(m)
            inl = e < thresh
            cost = ndata - int(inl.sum())
            if cost < best_cost:
                best_cost, best, mask = cost, m, inl
    return mask, best
