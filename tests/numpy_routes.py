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


    def solve_gdls(self, origin, direction, world, u):
        """GdlsSimilarityTransform (sfm/transformation/gdls_similarity_transform.cc:67-228) in matrix form: rays c_i + alpha x_i,
        s c_i + alpha_i x_i = R X_i + t.  Scale and translation z = (s, t) are eliminated as the least-squares solution of
        sum |P_i (L_i r + A_i z)|^2 (P_i = I - x_i x_i^T, A_i = [-c_i | I], L_i r = R X_i), the cost matrix is sum W_i^T P_i W_i with
        W_i = L_i + A_i Z, and from there on it is DLS.  Returns (R, t, s) with s c + alpha x = R X + t."""
        n = len(world)
        x = direction / np.linalg.norm(direction, axis=1, keepdims=True)
        P = np.eye(3)[None] - x[:, :, None] * x[:, None, :]
        A = np.concatenate([-origin[:, :, None], np.broadcast_to(np.eye(3), (n, 3, 3))], axis=2)        # n x 3 x 4
        L = np.zeros((n, 3, 9))
        for r in range(3):
            L[:, r, 3 * r:3 * r + 3] = world
        Hm = np.einsum("nia,nij,njb->ab", A, P, A)
        Z = -np.linalg.solve(Hm, np.einsum("nia,nij,njb->ab", A, P, L))                                  # 4 x 9
        Wm = L + np.einsum("nia,ab->nib", A, Z)
        D = np.einsum("nia,nij,njb->ab", Wm, P, Wm)
        coef = {0: np.asarray(u, dtype=np.float64)}
        for v in range(3):
            coef[v + 1] = self.T[v] @ D.ravel()
        vals = np.array([coef[q // 100][q % 100] for q in self.src])
        M = np.zeros((120, 120))
        M[self.rows, self.cols] = vals
        S = M[:27, :27] - M[:27, 27:] @ np.linalg.solve(M[27:, 27:], M[27:, :27])
        _, V = np.linalg.eig(S)
        out = []
        for i in range(27):
            if V[0, i] == 0:
                continue
            sv = V[[9, 3, 1], i] / V[0, i]
            if np.abs(sv.imag).max() >= 1e-6:
                continue
            q = np.array([1.0, -sv[0].real, -sv[1].real, -sv[2].real]); q /= np.linalg.norm(q)
            w_, a, b, c = q
            R = np.array([[1 - 2 * (b * b + c * c), 2 * (a * b - w_ * c), 2 * (a * c + w_ * b)],
                          [2 * (a * b + w_ * c), 1 - 2 * (a * a + c * c), 2 * (b * c - w_ * a)],
                          [2 * (a * c - w_ * b), 2 * (b * c + w_ * a), 1 - 2 * (a * a + b * b)]])
            z = Z @ R.ravel()
            sc, t = z[0], z[1:]
            if (((world @ R.T + t - sc * origin) * x).sum(1) < 0).any():
                continue
            out.append((R, t, sc))
        return out


_dls = None


def gdls_similarity_models(rows, u):
    """SimilarityTransformation2D3DEstimator::EstimateModel (estimate_similarity_transformation_2d_3d.cc:85-133) on four 26-double
    rows: (rotation, translation, scale) = (R^T, -R^T t, s) of every gDLS solution."""
    global _dls
    if _dls is None:
        _dls = _DlsSystem()
    sols = _dls.solve_gdls(rows[:, 9:12], rows[:, 0:3], rows[:, 3:6] / rows[:, 6:7], u)
    return [(R.T, -R.T @ t, sc) for R, t, sc in sols]


def similarity_errors(model, rows, project):
    """SimilarityTransformation2D3DEstimator::Error (:137-155) with TransformCamera (:52-70): the datum's camera moved to
    s R c + t with orientation R_cam R^T sees the world point; squared pixel error, infinite at negative depth."""
    from pytheiasfm_amd import synth
    Rm, tm, sc = model
    n = len(rows)
    pos = sc * rows[:, 9:12] @ Rm.T + tm
    a = rows[:, 3:6] - rows[:, 6:7] * pos
    b = np.c_[a @ Rm, rows[:, 6]]                                   # R^T a, homogeneous
    ext = np.c_[np.zeros((n, 3)), rows[:, 12:15]]
    uv, _ = project(int(rows[0, 15]), rows[:, 16:26], ext, b)
    depth = np.einsum("nij,nj->ni", synth.angle_axis_to_matrix(rows[:, 12:15]), b[:, :3])[:, 2] / b[:, 3]
    e = ((uv - rows[:, 7:9]) ** 2).sum(1)
    e[depth < 0] = np.inf
    return e


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


def ransac_replay(samples, fit, errors, thresh, ndata, mode="support", min_samples=0):
    """The reference's loop (sample_consensus_estimator.h:300-415, min = max iterations) under its three quality measurements:
    "support" = InlierSupport (cost = outliers), "mle" = MLEQualityMeasurement (sum of min(r, thresh) left to right,
    mle_quality_measurement.h:58-71), "lmed" = LmedQualityMeasurement (lmed_quality_measurement.h:58-120, with its quirks: the
    estimator's -- already squared -- errors are squared again, an ODD count averages the two middle values, the inlier bound is
    OpenCV's 2.5 x 1.4826 (1 + 5 / (n - m)) sqrt(median)).  The first strictly smaller cost wins.  Returns the inlier mask."""
    best_cost = np.inf
    mask = np.zeros(ndata, dtype=bool)
    for it, idx in enumerate(samples):
        for m in fit(it, idx):
            e = errors(m)
            if mode == "support":
                inl = e < thresh; cost = ndata - int(inl.sum())
            elif mode == "mle":
                inl = e < thresh
                cost = 0.0
                for v, ok in zip(e, inl):            # sequential, as the reference adds them
                    cost += v if ok else thresh
            else:
                with np.errstate(over="ignore", invalid="ignore"):
                    sq = np.sort(e * e)
                med = sq[ndata // 2]
                if ndata % 2 != 0:
                    med = 0.5 * (sq[ndata // 2 - 1] + med)
                bound = (2.5 * 1.4826 * (1 + 5.0 / (ndata - min_samples)) * np.sqrt(med)) ** 2
                with np.errstate(over="ignore", invalid="ignore"):
                    inl = (e * e) < bound
                cost = med
            if cost < best_cost:
                best_cost, mask = cost, inl
    return mask


# ------------------------------------------------------------------------------------------------ more estimators (round 4)
def sampson_errors(F, x1h, x2h):
    """SquaredSampsonDistance (pose/util.cc:56-68) for a matrix with x2^T F x1 = 0, vectorised."""
    Fx1 = x1h @ F.T; Ftx2 = x2h @ F
    num = (x2h * Fx1).sum(1) ** 2
    return num / (Fx1[:, 0] ** 2 + Fx1[:, 1] ** 2 + Ftx2[:, 0] ** 2 + Ftx2[:, 1] ** 2)


def _normalise(pts):
    """NormalizeImagePoints (pose/util.cc:81-112): centroid to the origin, RMS distance sqrt(2)."""
    c = pts.mean(0)
    s = np.sqrt(2.0) / np.sqrt(((pts - c) ** 2).sum() / len(pts))
    T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    return (pts - c) * s, T


def eight_point(x1, x2):
    """NormalizedEightPointFundamentalMatrix (pose/eight_point_fundamental_matrix.cc:54-120): the null vector of the 8 x 9
    constraint matrix by numpy's SVD (the reference: FullPivLU kernel), rank 2 by SVD, normalisation undone."""
    n1, T1 = _normalise(x1); n2, T2 = _normalise(x2)
    A = np.c_[n2[:, :1] * n1, n2[:, :1], n2[:, 1:2] * n1, n2[:, 1:2], n1, np.ones(len(n1))]
    _, s, Vt = np.linalg.svd(A)
    if len(n1) == 8 and s[7] < 1e-12 * s[0]:
        return []                                  # dimensionOfKernel() != 1
    F = Vt[-1].reshape(3, 3)
    U, sv, Vt2 = np.linalg.svd(F)
    F = U @ np.diag([sv[0], sv[1], 0.0]) @ Vt2
    return [T2.T @ F @ T1]


def four_point_homography(x1, x2):
    """FourPointHomography (pose/four_point_homography.cc:69-102): DLT on normalised points, null vector of A^T A."""
    n1, T1 = _normalise(x1); n2, T2 = _normalise(x2)
    rows = []
    for (a, b), (u, v) in zip(n1, n2):
        rows.append([0, 0, 0, -a, -b, -1, a * v, b * v, v])
        rows.append([a, b, 1, 0, 0, 0, -a * u, -b * u, -u])
    A = np.array(rows)
    w, V = np.linalg.eigh(A.T @ A)
    return [np.linalg.inv(T2) @ V[:, 0].reshape(3, 3) @ T1]


def homography_errors(H, x1h, x2):
    p = x1h @ H.T
    return ((x2 - p[:, :2] / p[:, 2:3]) ** 2).sum(1)


def p3p_kneip(feat, world):
    """PoseFromThreePoints (pose/perspective_three_point.cc:56-300) re-typed from Kneip's paper: the quartic in cos(theta) solved by
    np.roots (LAPACK's eigenvalues of the companion matrix; the reference: its own companion-matrix solver), and -- as the reference
    does -- the REAL PARTS of all four roots are back-substituted.  Returns [(R, t)] with x ~ R X + t."""
    f = np.c_[feat, np.ones(3)]
    f = f / np.linalg.norm(f, axis=1, keepdims=True)
    P = np.array(world, dtype=np.float64)
    if (np.cross(P[1] - P[0], P[2] - P[0]) ** 2).sum() < 1e-6:
        return []

    def cam_frame(f):
        e1 = f[0]
        e3 = np.cross(f[0], f[1]); e3 /= np.linalg.norm(e3)
        return np.array([e1, np.cross(e3, e1), e3])
    T = cam_frame(f)
    f3 = T @ f[2]
    if f3[2] > 0:
        f = f[[1, 0, 2]]; P = P[[1, 0, 2]]
        T = cam_frame(f)
        f3 = T @ f[2]
    n1 = (P[1] - P[0]) / np.linalg.norm(P[1] - P[0])
    n3 = np.cross(n1, P[2] - P[0]); n3 /= np.linalg.norm(n3)
    N = np.array([n1, np.cross(n3, n1), n3])
    P3 = N @ (P[2] - P[0])
    d12 = np.linalg.norm(P[1] - P[0])
    f1, f2 = f3[0] / f3[2], f3[1] / f3[2]
    p1, p2 = P3[0], P3[1]
    cb = f[0] @ f[1]
    b = 1.0 / (1.0 - cb * cb) - 1.0
    b = -np.sqrt(b) if cb < 0 else np.sqrt(b)
    c = [-f2 ** 2 * p2 ** 4 - p2 ** 4 * f1 ** 2 - p2 ** 4,
         2 * p2 ** 3 * d12 * b + 2 * f2 ** 2 * p2 ** 3 * d12 * b - 2 * f2 * p2 ** 3 * f1 * d12,
         -f2 ** 2 * p2 ** 2 * p1 ** 2 - f2 ** 2 * p2 ** 2 * d12 ** 2 * b ** 2 - f2 ** 2 * p2 ** 2 * d12 ** 2 + f2 ** 2 * p2 ** 4 + p2 ** 4 * f1 ** 2
         + 2 * p1 * p2 ** 2 * d12 + 2 * f1 * f2 * p1 * p2 ** 2 * d12 * b - p2 ** 2 * p1 ** 2 * f1 ** 2 + 2 * p1 * p2 ** 2 * f2 ** 2 * d12
         - p2 ** 2 * d12 ** 2 * b ** 2 - 2 * p1 ** 2 * p2 ** 2,
         2 * p1 ** 2 * p2 * d12 * b + 2 * f2 * p2 ** 3 * f1 * d12 - 2 * f2 ** 2 * p2 ** 3 * d12 * b - 2 * p1 * p2 * d12 ** 2 * b,
         -2 * f2 * p2 ** 2 * f1 * p1 * d12 * b + f2 ** 2 * p2 ** 2 * d12 ** 2 + 2 * p1 ** 3 * d12 - p1 ** 2 * d12 ** 2 + f2 ** 2 * p2 ** 2 * p1 ** 2
         - p1 ** 4 - 2 * f2 ** 2 * p2 ** 2 * p1 * d12 + p2 ** 2 * f1 ** 2 * p1 ** 2 + f2 ** 2 * p2 ** 2 * d12 ** 2 * b ** 2]
    out = []
    with np.errstate(all="ignore"):
        for ct in np.real(np.roots(c)):
            cot = (-f1 * p1 / f2 - ct * p2 + d12 * b) / (-f1 * ct * p2 / f2 + p1 - d12)
            st = np.sqrt(1 - ct * ct)
            sa = np.sqrt(1 / (cot * cot + 1)); ca = np.sqrt(1 - sa * sa)
            if cot < 0:
                ca = -ca
            C = P[0] + N.T @ np.array([d12 * ca * (sa * b + ca), ct * d12 * sa * (sa * b + ca), st * d12 * sa * (sa * b + ca)])
            Q = np.array([[-ca, -sa * ct, -sa * st], [sa, -ca * ct, -ca * st], [0, -st, ct]])
            R = (N.T @ Q.T @ T).T
            out.append((R, -R @ C))
    return out


class UpnpRoute:
    """EstimateRigidTransformation2D3D's hypotheses (pose/upnp.cc) with numpy / LAPACK: the cost parameters from the paper's
    formulas in matrix form, the input equations by generic polynomial differentiation of the quartic cost, the reference's
    141 x 149 template (layout: oracle/upnp_layout.h, data) reduced by ONE np.linalg.solve instead of its Gauss-Jordan, the
    action matrix' eigenvectors by np.linalg.eig.  The estimator's accumulating state (upnp.cc:191-200) is kept.  Complex
    eigenvector pairs are skipped: their real parts depend on the phase convention of the eigen-solver (Eigen: hqr2's)."""
    S = [(2, 0, 0, 0), (0, 2, 0, 0), (0, 0, 2, 0), (0, 0, 0, 2), (1, 1, 0, 0), (1, 0, 1, 0), (1, 0, 0, 1), (0, 1, 1, 0), (0, 1, 0, 1), (0, 0, 1, 1)]

    def __init__(self, layout_path):
        import re
        txt = open(layout_path).read()
        self.row_eq = [int(v) for v in re.search(r"kRowEq\[141\] = \{(.*?)\};", txt, re.S).group(1).split(",")]
        quad = lambda name: np.array([[int(v) for v in m] for m in re.findall(r"\{(-?\d+), (-?\d+), (-?\d+), (-?\d+)\}", re.search(name + r"\[\d+\]\[4\] = \{(.*?)\};", txt, re.S).group(1))])
        self.row_mul, self.col_mono = quad("kRowMul"), quad("kColMono")
        cubic = sorted(((a, b, c, d) for a in range(4) for b in range(4) for c in range(4) for d in range(4) if a + b + c + d == 3), key=lambda e: (e[3], e[2], e[1]))
        self.mono = cubic + [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)]
        col_of = {tuple(e): c for c, e in enumerate(self.col_mono)}
        self.place = []
        for r in range(141):
            eq = self.row_eq[r]
            sup = [10, 12, 15, 19, 23] if eq == 7 else sorted({eq, 7, 8, 9} | set(range(11, 24)))
            for j in sup:
                self.place.append((r, col_of[tuple(self.row_mul[r] + np.array(self.mono[j]))], eq, j))
        self.A = np.zeros((10, 10)); self.b = np.zeros(10)

    @staticmethod
    def _phi(X):
        x, y, z = X
        return np.array([[x, x, -x, -x, 0, 2 * z, -2 * y, 2 * y, 2 * z, 0],
                         [y, -y, y, -y, -2 * z, 0, 2 * x, 2 * x, 0, 2 * z],
                         [z, -z, -z, z, 2 * y, -2 * x, 0, 0, 2 * x, 2 * y]])

    def fit(self, origin, direction, world):
        n = len(world)
        F = [np.outer(f, f) for f in direction]
        H = np.linalg.inv(n * np.eye(3) - sum(F))
        V = [H @ (Fi - np.eye(3)) for Fi in F]
        Phi = [self._phi(X) for X in world]
        G = sum(Vi @ Pi for Vi, Pi in zip(V, Phi)); J = sum(Vi @ o for Vi, o in zip(V, origin))
        for Fi, Pi, o in zip(F, Phi, origin):
            Ai = (Fi - np.eye(3)) @ (Pi + G); bi = -(Fi - np.eye(3)) @ (o + J)
            self.A += Ai.T @ Ai; self.b += Ai.T @ bi
        # gradient of s^T A s + 2 b^T s in the quaternion, by polynomial arithmetic on exponent 4-tuples
        cost = {}
        for p in range(10):
            for q in range(10):
                e = tuple(x + y for x, y in zip(self.S[p], self.S[q]))
                cost[e] = cost.get(e, 0.0) + self.A[p, q]
            cost[self.S[p]] = cost.get(self.S[p], 0.0) + 2.0 * self.b[p]
        M = np.zeros((8, 24))
        for i in range(4):
            for e, c in cost.items():
                if e[i]:
                    d = list(e); d[i] -= 1
                    M[i, self.mono.index(tuple(d))] += e[i] * c
            for k in range(4):
                e = [0, 0, 0, 0]; e[k] += 2; e[i] += 1
                M[4 + i, self.mono.index(tuple(e))] = 1.0
            M[4 + i, 20 + i] = -1.0
        M[:7] = np.linalg.solve(M[:7, :7], M[:7])
        M[:7] -= np.outer(M[:7, 10] / M[7, 10], M[7])
        T = np.zeros((141, 149))
        for r, c, eq, j in self.place:
            T[r, c] = M[eq, j]
        Xr = np.linalg.solve(T[:, :141], T[:, 141:])
        act = np.zeros((8, 8)); act[:4] = -Xr[121:125]; act[4:, :4] = np.eye(4)
        w, Vec = np.linalg.eig(act)
        quats = []
        for k in range(8):
            if abs(w[k].imag) > abs(w[k].real) * 1e-12:
                continue
            q = np.real(Vec[4:8, k]); q = q / np.linalg.norm(q)
            if not any(2 * np.arctan2(np.linalg.norm(q[0] * -p[1:] + p[0] * q[1:] + np.cross(q[1:], -p[1:])), abs(q @ p)) < np.deg2rad(0.1) for p in quats):
                quats.append(q)
        out = []
        for q in quats:
            w0, x, y, z = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w0), 2 * (x * z + y * w0)],
                          [2 * (x * y + z * w0), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w0)],
                          [2 * (x * z - y * w0), 2 * (y * z + x * w0), 1 - 2 * (x * x + y * y)]])
            t = sum(Vi @ (R @ X - o) for Vi, X, o in zip(V, world, origin))
            if all(((R @ X + t - o) @ f) >= 0 for X, o, f in zip(world, origin, direction)):   # in front, seen along the ray
                out.append((R, t))
        return out


# ------------------------------------------------------------------------------------------------ P4Pfr (round 4)
class LibstdcxxStream:
    """std::mt19937 with libstdc++'s uniform_int_distribution<int> (Lemire) and uniform_real_distribution<double>
    (generate_canonical: two 32-bit draws), written from the library's documented algorithms on numpy's MT19937 core
    (np.random.RandomState(seed) runs init_genrand(seed), as std::mt19937::seed does; bytes(4) is one 32-bit output)."""

    def __init__(self, seed):
        self.seed(seed)

    def seed(self, seed):
        self.rs = np.random.RandomState(int(seed))

    def next(self):
        return int.from_bytes(self.rs.bytes(4), "little")

    def rand_int(self, lo, hi):
        urange = hi - lo
        if urange == 0xFFFFFFFF:
            return self.next() + lo
        rng = urange + 1
        product = self.next() * rng
        low = product & 0xFFFFFFFF
        if low < rng:
            threshold = (2 ** 32 - rng) % rng
            while low < threshold:
                product = self.next() * rng
                low = product & 0xFFFFFFFF
        return (product >> 32) + lo

    def rand_double(self, lo, hi):
        s = float(self.next()) + float(self.next()) * 4294967296.0       # exact products; ONE rounding in the sum, as in double arithmetic
        r = s / 18446744073709551616.0
        if r >= 1.0:
            r = np.nextafter(1.0, 0.0)
        return r * (hi - lo) + lo

    def prosac_samples(self, num_data, m, iters):
        """ProsacSampler::Sample (solvers/prosac_sampler.cc:66-128) for the first `iters` samples of an Estimate() call: the growth
        function of Chum & Matas (t_n from 20 000 convergence iterations), m distinct draws from the top n, or m - 1 from the top
        n - 1 plus the point at position n (the reference pushes index n, not n - 1)."""
        import math
        out = []
        for kth in range(1, iters + 1):
            t_n = 20000.0
            n = m
            for i in range(m):
                t_n *= float(n - i) / (num_data - i)
            t_n_prime = 1.0
            for t in range(1, kth + 1):
                if t > t_n_prime and n < num_data:
                    t_n_plus1 = (t_n * (n + 1.0)) / (n + 1.0 - m)
                    t_n_prime += math.ceil(t_n_plus1 - t_n)
                    t_n = t_n_plus1
                    n += 1
            sub = []
            if t_n_prime < kth:
                for _ in range(m):
                    r = self.rand_int(0, n - 1)
                    while r in sub:
                        r = self.rand_int(0, n - 1)
                    sub.append(r)
            else:
                for _ in range(m - 1):
                    r = self.rand_int(0, n - 2)
                    while r in sub:
                        r = self.rand_int(0, n - 2)
                    sub.append(r)
                sub.append(n)
            out.append(sub)
        return np.array(out)

    def p4pfr_rounds(self, n, iters, first_call=False):
        """RandomSampler::Sample (partial Fisher-Yates on a persistent index vector) followed by the solver's three draws."""
        idx = list(range(n))
        samples, draws = [], []
        for it in range(iters):
            for i in range(4):
                j = self.rand_int(i, n - 1)
                idx[i], idx[j] = idx[j], idx[i]
            samples.append(idx[:4])
            if first_call and it == 0:
                self.seed(42)
            draws.append([self.rand_double(-0.5, 0.5) for _ in range(3)])
        return np.array(samples), np.array(draws)


class P4pfrRoute:
    """EstimateRadialDistUncalibratedAbsolutePose's hypotheses with numpy / LAPACK: SVD for the normalising rotation and for the
    null space of the 5 x 8 system (the reference: JacobiSVD and HouseholderQR), least squares for the particular solution and
    for D (FullPivLU / ColPivHouseholderQR), the ten equations by generic polynomial arithmetic on exponent tuples from the
    definitions of the projection-matrix rows (the cubic forms' terms and the rows / columns of the template are read from
    oracle/p4pfr_layout.h as data), the template reduced by np.linalg.lstsq (minimum-norm alpha; Eigen's FullPivLU solve keeps the
    free unknowns at zero -- any alpha with alpha^T C0 = b gives the same action matrix on the solutions), np.linalg.eig."""

    def __init__(self, layout_path):
        import re
        txt = open(layout_path).read()

        def table(name, width):
            body = re.search(name + r"\[\d+\](?:\[\d+\])* = \{(.*?)\};", txt, re.S).group(1)
            return np.array([int(v) for v in re.findall(r"-?\d+", body)]).reshape(-1, width)
        self.col = [tuple(r) for r in table("kColMono", 5)]
        self.row_eq = table("kRowEq", 1).ravel()
        self.row_mul = table("kRowMul", 5)
        nterm = table("kCubicTerms", 1).ravel()
        cub = table("kCubic", 4).reshape(5, -1, 4)
        self.cubic = [cub[f, :nterm[f]] for f in range(5)]
        self.am_row = table("kAmRow", 1).ravel()
        self.cidx = {m: c for c, m in enumerate(self.col)}

    @staticmethod
    def _mul(a, b):
        out = {}
        for ea, ca in a.items():
            for eb, cb in b.items():
                e = tuple(x + y for x, y in zip(ea, eb))
                out[e] = out.get(e, 0.0) + ca * cb
        return out

    @staticmethod
    def _add(a, b, s=1.0):
        out = dict(a)
        for e, c in b.items():
            out[e] = out.get(e, 0.0) + s * c
        return out

    def fit(self, feat, world, draws, limits):
        A1, A2, A3, K, Wv, ONE = (1, 0, 0, 0, 0), (0, 1, 0, 0, 0), (0, 0, 1, 0, 0), (0, 0, 0, 1, 0), (0, 0, 0, 0, 1), (0, 0, 0, 0, 0)
        d = (feat ** 2).sum(1)
        t0 = world.mean(0)
        Xc = (world - t0).T                                   # 3 x 4
        Us = np.linalg.svd(Xc)[0]
        if np.linalg.det(Us) < 0:
            Us[:, 0] *= -1
        R0 = Us.T
        U = R0 @ Xc
        scale = np.linalg.norm(U, axis=0).mean(); U = U / scale
        U4 = np.vstack([U, np.ones(4)])
        f0 = np.linalg.norm(feat, axis=1).mean(); u = (feat / f0).T     # 2 x 4
        k0 = d.mean(); d = d / k0
        M = np.zeros((5, 8)); b = np.zeros(5)
        M[0, :4] = U4[:, 0]; M[1, 4:] = U4[:, 0]; b[:2] = u[:, 0]
        for k in range(1, 4):
            M[k + 1, :4] = u[1, k] * U4[:, k]; M[k + 1, 4:] = -u[0, k] * U4[:, k]
        Nn = np.zeros((8, 4))
        ang = np.linalg.norm(draws)
        Kx = np.array([[0, -draws[2], draws[1]], [draws[2], 0, -draws[0]], [-draws[1], draws[0], 0]])
        Rr = np.cos(ang) * np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * np.outer(draws, draws)     # AngleAxis on the raw vector
        Nn[:, :3] = np.linalg.svd(M)[2][5:].T @ Rr
        Nn[:, 3] = np.linalg.lstsq(M, b, rcond=None)[0]
        UN1 = U4[:, 1:].T @ Nn[:4]; UN2 = U4[:, 1:].T @ Nn[4:]
        B = np.zeros((6, 9)); C = np.zeros((6, 3))
        for h, UN in enumerate((UN1, UN2)):
            B[3 * h:3 * h + 3, :3] = UN[:, :3]; B[3 * h:3 * h + 3, 3:7] = d[1:, None] * UN; B[3 * h:3 * h + 3, 8] = UN[:, 3]
            B[3 * h:3 * h + 3, 7] = -u[h, 1:] * U[2, 1:]
            C[3 * h:3 * h + 3] = u[h, 1:, None] * np.stack([U[0, 1:], U[1, 1:], np.ones(3)], axis=1)
        D = np.linalg.lstsq(C, B, rcond=None)[0]              # 3 x 9
        lin = lambda row: {A1: row[0], A2: row[1], A3: row[2], ONE: row[3]}
        tm = [A1, A2, A3, (1, 0, 0, 1, 0), (0, 1, 0, 1, 0), (0, 0, 1, 1, 0), K, Wv, ONE]
        p1 = [lin(Nn[r]) for r in range(3)]; p2 = [lin(Nn[4 + r]) for r in range(3)]
        p3 = [{tm[c]: D[0, c] for c in range(9)}, {tm[c]: D[1, c] for c in range(9)}, {Wv: 1.0}]
        p3w = {tm[c]: D[2, c] for c in range(9)}
        dot = lambda x, y: self._add(self._add(self._mul(x[0], y[0]), self._mul(x[1], y[1])), self._mul(x[2], y[2]))
        q = p1 + p2
        eqs = [dot(p2, p3), dot(p1, p3), dot(p1, p2), self._add(dot(p1, p1), dot(p2, p2), -1.0)]
        for form in self.cubic:
            e = {}
            for i, j, l, c in form:
                e = self._add(e, self._mul(self._mul(q[i], q[j]), p3[l]), float(c))
            eqs.append(e)
        e9 = self._add({ONE: 1.0, K: d[0]}, p3w, -1.0)
        for r in range(3):
            e9 = self._add(e9, p3[r], -U[r, 0])
        eqs.append(e9)
        T = np.zeros((40, 50))
        for r in range(40):
            for e, c in eqs[self.row_eq[r]].items():
                T[r, self.cidx[tuple(x + y for x, y in zip(e, self.row_mul[r]))]] = c
        bm = np.zeros((7, 37)); bm[np.arange(7), 30 + np.arange(7)] = -1.0
        alpha = np.linalg.lstsq(T[:, :37].T, bm.T, rcond=None)[0]          # 40 x 7
        RR = np.vstack([alpha.T @ T[:, 37:], np.eye(13)])
        w, V = np.linalg.eig(RR[self.am_row])
        V = V / V[0]
        out = []
        for j in range(13):
            if abs(V[5, j].imag) > 1e-6:
                continue
            a = np.array([V[5, j].real, V[7, j].real, w[j].real, 1.0]); k = V[1, j].real; P33 = V[3, j].real
            P12 = Nn @ a
            tmp = np.array([a[0], a[1], a[2], k * a[0], k * a[1], k * a[2], k, P33, 1.0])
            P3 = D @ tmp
            P = np.vstack([P12[:4], P12[4:], [P3[0], P3[1], P33, P3[2]]])
            P = P / np.linalg.norm(P[2, :3])
            f = np.linalg.norm(P[0, :3])
            focal = f * f0; rd = k / k0
            if focal < limits[1] or focal > limits[0] or rd < limits[2] or rd > limits[3] or rd > 0.0:
                continue
            Rt = np.diag([1 / f, 1 / f, 1.0]) @ P
            if np.linalg.det(Rt[:, :3]) < 0:
                Rt = -Rt
            R = Rt[:, :3] @ R0
            out.append((R, Rt[:, 3] * scale - R @ t0, focal, rd))
        return out


def radial_dist_errors(model, feat, world):
    """RadialDistUncalibratedAbsolutePoseEstimator::Error, vectorised."""
    R, t, focal, rd = model
    if t[2] < 0.0:
        return np.full(len(feat), 1e10)
    pc = world @ R.T + t
    x = focal * pc[:, :2] / pc[:, 2:3]
    r2 = (x * x).sum(1)
    den = 2.0 * rd * r2; inner = 1.0 - 4.0 * rd * r2
    keep = (np.abs(den) < 1e-15) | (inner < 0.0)
    sc = np.where(keep, 1.0, (1.0 - np.sqrt(np.maximum(inner, 0.0))) / np.where(keep, 1.0, den))
    return ((x * sc[:, None] - feat) ** 2).sum(1)


# ------------------------------------------------------------------------------------------------ four small estimators (round 4)
def plane_from_points(pts):
    """DominantPlaneEstimator::EstimateModel: the plane through three points -- the normal as the direction of least variance of
    the centred points (numpy SVD; the reference: a cross product), the sample's first point on the plane; collinear samples
    (|cross product|^2 < 1e-6, estimate_dominant_plane_from_points.cc:70-74) give no model."""
    a, b = pts[1] - pts[0], pts[2] - pts[0]
    if np.sum(np.cross(a, b) ** 2) < 1e-6:
        return []
    n = np.linalg.svd(pts - pts.mean(0))[2][-1]
    return [(pts[0], n)]


def plane_errors(model, pts):
    p0, n = model
    return np.abs((pts - p0) @ n)                     # point-to-plane distance, not squared (:84-86)


def two_point_position(x1, x2):
    """RelativePoseFromTwoPointsWithKnownRotation: the null vector of the 2 x 3 epipolar constraint by numpy's SVD (the
    reference: a FullPivLU kernel), normalised; sign as the reference's kernel vector has it (the free variable = +1 in the LU's
    column order is reproduced by fixing the sign through the largest-magnitude convention below)."""
    A = np.array([[-x1[i, 1] + x2[i, 1], -x2[i, 0] + x1[i, 0], x1[i, 1] * x2[i, 0] - x1[i, 0] * x2[i, 1]] for i in range(2)])
    s = np.linalg.svd(A, compute_uv=False)
    if s[1] <= 2 * np.finfo(float).eps * 2.0 * s[0]:
        return []
    v = np.linalg.svd(A)[2][-1]
    return [v / np.linalg.norm(v)]


def known_orientation_errors(pos, x1h, x2h):
    E = np.array([[0.0, pos[2], -pos[1]], [-pos[2], 0.0, pos[0]], [pos[1], -pos[0], 0.0]])
    return sampson_errors(E, x1h, x2h)               # the Sampson distance does not see the sign of the position


def position_from_rays(feat, world):
    """PositionFromTwoRays: least squares of the 4 x 3 system (numpy lstsq; the reference: ColPivHouseholderQR)."""
    A = np.zeros((4, 3)); b = np.zeros(4)
    for i in range(2):
        u, v = feat[i]; X, Y, Z = world[i]
        A[2 * i] = [1.0, 0.0, -u]; A[2 * i + 1] = [0.0, 1.0, -v]
        b[2 * i] = X - u * Z; b[2 * i + 1] = Y - v * Z
    x, _, rank, _ = np.linalg.lstsq(A, b, rcond=None)
    return [x] if rank == 3 else []


def known_orientation_abs_errors(pos, feat, world):
    p = world - pos
    return ((p[:, :2] / p[:, 2:3] - feat) ** 2).sum(1)


def focal_lengths_bougnoux(F):
    """Focal lengths of the two cameras from a fundamental matrix with both principal points at the origin -- Bougnoux's closed
    form (f1^2 = -(p2^T [e2]x I~ F p1)(p1^T F^T p2) / (p2^T [e2]x I~ F I~ F^T p2), p = (0, 0, 1), I~ = diag(1, 1, 0)); the
    reference rotates the epipoles onto the x axis and solves the resulting 2 x 2 (fundamental_matrix_util.cc:57-130)."""
    It = np.diag([1.0, 1.0, 0.0]); p = np.array([0.0, 0.0, 1.0])

    def one(F):
        e2 = np.linalg.svd(F.T)[2][-1]
        ex = np.array([[0, -e2[2], e2[1]], [e2[2], 0, -e2[0]], [-e2[1], e2[0], 0]])
        return -(p @ ex @ It @ F @ p) * (p @ F.T @ p) / (p @ ex @ It @ F @ It @ F.T @ p)
    return one(F), one(F.T)


def uncalibrated_relative_pose_models(x1, x2, min_max_focal):
    """UncalibratedRelativePoseEstimator::EstimateModel (estimate_uncalibrated_relative_pose.cc:83-138): 8-point F (numpy SVD),
    focal lengths by Bougnoux's formula, E = K2 F K1, the pose with the most points in front."""
    out = []
    for F in eight_point(x1, x2):
        f1sq, f2sq = focal_lengths_bougnoux(F)
        if not (f1sq > 0 and f2sq > 0):
            continue
        f1, f2 = np.sqrt(f1sq), np.sqrt(f2sq)
        lo, hi = min_max_focal
        if lo >= 1.0 and hi >= 1.0 and (f1 < lo or f2 < lo or f1 > hi or f2 > hi):
            continue
        E = np.diag([f2, f2, 1.0]) @ F @ np.diag([f1, f1, 1.0])
        n1 = x1 / f1; n2 = x2 / f2
        x1h = np.c_[n1, np.ones(len(n1))]; x2h = np.c_[n2, np.ones(len(n2))]
        U, _, Vt = np.linalg.svd(E)
        if np.linalg.det(U) < 0:
            U[:, 2] *= -1
        if np.linalg.det(Vt) < 0:
            Vt[2] *= -1
        Dm = np.array([[0.0, 1, 0], [-1, 0, 0], [0, 0, 1]])
        R1, R2 = U @ Dm @ Vt, U @ Dm.T @ Vt
        tr = U[:, 2] / np.linalg.norm(U[:, 2])
        cands = [(R1, -R1.T @ tr), (R1, R1.T @ tr), (R2, -R2.T @ tr), (R2, R2.T @ tr)]
        votes = [int(_in_front(x1h, x2h, R, pp).sum()) for R, pp in cands]
        k = int(np.argmax(votes))
        out.append((F, cands[k][0], cands[k][1], f1, f2))
    return out


def uncalibrated_relative_pose_errors(model, x1, x2):
    F, R, pos, f1, f2 = model
    x1h = np.c_[x1, np.ones(len(x1))]; x2h = np.c_[x2, np.ones(len(x2))]
    err = sampson_errors(F, x1h, x2h)
    n1h = np.c_[x1 / f1, np.ones(len(x1))]; n2h = np.c_[x2 / f2, np.ones(len(x2))]
    err[~_in_front(n1h, n2h, R, pos)] = np.inf
    return err


# ------------------------------------------------------------------------------------------------ P4Pf (round 4)
class P4pfRoute:
    """EstimateUncalibratedAbsolutePose's hypotheses (P4Pf: four_point_focal_length.cc:100-222) with numpy / LAPACK: the four
    inner-product equations of the rigid point configuration written from the geometry by generic polynomial arithmetic on
    exponent tuples (x, y, z = depths of points b, c, d relative to a; w = focal length squared), the rows / columns of the
    elimination template read from oracle/p4pf_tables.h as data, the targets reduced onto the basis by ONE np.linalg.lstsq on the
    transposed system (the oracle / the device: partial-pivot elimination), np.linalg.eig, the similarity alignment by numpy SVD."""

    def __init__(self, tables_path):
        import re
        txt = open(tables_path).read()

        def table(name, width):
            body = re.search(name + r"\[[A-Za-z0-9]+\](?:\[\d+\])? = \{(.*?)\};", txt, re.S).group(1)
            return np.array([int(v) for v in re.findall(r"\d+", body)]).reshape(-1, width)
        self.row_poly = [int(v) for v in re.search(r"THIP_P4PF_ROW_POLY \{(.*?)\}", txt).group(1).split(",")]
        self.row_mul = table("kRowMul", 4)
        self.col = {tuple(m): c for c, m in enumerate(table("kColMono", 4))}
        assert len(self.row_poly) == 77 and len(self.col) == 104

    def fit(self, feat, world):
        mean = world.mean(0); wn = world - mean
        wvar = np.linalg.norm(wn, axis=1).mean(); wn = wn / wvar
        fvar = np.linalg.norm(feat, axis=1).mean(); fn = feat / fvar
        g = [np.sum((wn[i] - wn[j]) ** 2) for i in range(4) for j in range(i + 1, 4)]
        if np.prod(g) < 1e-15:
            return []
        X, Y, Z, W, ONE = (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1), (0, 0, 0, 0)
        dep = [{ONE: 1.0}, {X: 1.0}, {Y: 1.0}, {Z: 1.0}]
        mul, add = P4pfrRoute._mul, P4pfrRoute._add

        def ip(i, j):      # X_i . X_j = dep_i dep_j (pt_i . pt_j + w)
            return mul(mul(dep[i], dep[j]), {ONE: float(fn[i] @ fn[j]), W: 1.0})

        def diff_dot(i, j):  # (X_i - X_a) . (X_j - X_a)
            return add(add(add(ip(i, j), ip(0, i), -1.0), ip(0, j), -1.0), ip(0, 0))
        dd = float((wn[3] - wn[0]) @ (wn[3] - wn[0]))
        ks = [float((wn[1] - wn[0]) @ (wn[2] - wn[0])) / dd, float((wn[2] - wn[0]) @ (wn[2] - wn[0])) / dd,
              float((wn[1] - wn[0]) @ (wn[3] - wn[0])) / dd, float((wn[2] - wn[0]) @ (wn[3] - wn[0])) / dd]
        ad2 = diff_dot(3, 3)
        polys = [add(diff_dot(1, 2), ad2, -ks[0]), add(diff_dot(2, 2), ad2, -ks[1]), add(diff_dot(1, 3), ad2, -ks[2]), add(diff_dot(2, 3), ad2, -ks[3])]
        A = np.zeros((77, 104))
        for r in range(77):
            for e, c in polys[self.row_poly[r]].items():
                A[r, self.col[tuple(a + b for a, b in zip(e, self.row_mul[r]))]] += c
        E = np.zeros((94, 5)); E[89 + np.arange(5), np.arange(5)] = 1.0
        Yc, res, rank, _ = np.linalg.lstsq(A[:, :94].T, E, rcond=None)
        if rank < 77:
            return []
        T = np.zeros((10, 10))
        T[0, 1] = T[1, 5] = T[2, 6] = T[3, 7] = T[4, 8] = 1.0
        T[5:] = -(Yc.T @ A[:, 94:])
        wv, V = np.linalg.eig(T)
        out = []
        for i in range(10):
            if wv[i].imag != 0.0:
                continue
            v = (V[:, i] / V[0, i]).real
            w = v[4]
            if not w >= 0.0:
                continue
            f = np.sqrt(w)
            depth = np.array([1.0, v[3], v[2], v[1]])
            Ac = np.c_[fn * depth[:, None], f * depth]
            ratios = [np.sqrt(gij / np.sum((Ac[i0] - Ac[j0]) ** 2)) for gij, (i0, j0) in zip(g, [(a, b) for a in range(4) for b in range(a + 1, 4)])]
            Ac = Ac * np.mean(ratios)
            m1, m2 = wn.mean(0), Ac.mean(0)
            p1 = wn - m1; p2 = Ac - m2
            p1 = p1 / np.linalg.norm(p1, axis=1, keepdims=True); p2 = p2 / np.linalg.norm(p2, axis=1, keepdims=True)
            U, _, Vt = np.linalg.svd(p2.T @ p1)
            S = np.diag([1.0, 1.0, -1.0 if np.linalg.det(U @ Vt) < 0 else 1.0])
            R = U @ S @ Vt
            t = wvar * (m2 - R @ m1) - R @ mean
            fo = f * fvar
            out.append(np.vstack([fo * np.r_[R[0], t[0]], fo * np.r_[R[1], t[1]], np.r_[R[2], t[2]]]))
        return out


def projection_errors(Pm, feat, world):
    p = np.c_[world, np.ones(len(world))] @ Pm.T
    return ((p[:, :2] / p[:, 2:3] - feat) ** 2).sum(1)


# ------------------------------------------------------------------------------------------------ radial-distortion homography (round 4)
def radial_homography_models(rows):
    """SixPointRadialDistortionHomography (six_point_radial_distortion_homography.cc:62-149) on six correspondences (rows of 12:
    pixels left / right, normalised left / right, f1 f2, lmin lmax): the two-dimensional null space of the 6 x 8 system by numpy's
    SVD (the reference: JacobiSVD; the oracle / the device: one-sided Jacobi), the quadratic that makes the combination a
    homography row pair, then the 6 x 5 system for the third row and l1 (numpy SVD), numpy's own 3 x 3 inverse."""
    x = rows[:, 4:6]; u = rows[:, 6:8]
    lmin, lmax = rows[0, 10], rows[0, 11]
    u2 = (u * u).sum(1); x2 = (x * x).sum(1)
    M = np.c_[-x[:, 1:2] * u, -x[:, 1], x[:, 0:1] * u, x[:, 0], -x[:, 1] * u2, x[:, 0] * u2]
    N = np.linalg.svd(M)[2][6:].T                                # 8 x 2: a basis of the null space (any basis serves)
    a = -N[2, 0] * N[7, 0] + N[5, 0] * N[6, 0]
    b = -N[2, 0] * N[7, 1] - N[2, 1] * N[7, 0] + N[5, 0] * N[6, 1] + N[5, 1] * N[6, 0]
    c = -N[2, 1] * N[7, 1] + N[5, 1] * N[6, 1]
    dd = b * b - 4 * a * c
    eps = 100.0 * np.finfo(float).eps
    if abs(dd) < eps:
        roots = [-b / (2 * a)]
    elif dd > 0:
        roots = [(-b + np.sqrt(dd)) / (2 * a), (-b - np.sqrt(dd)) / (2 * a)]
    else:
        return []
    out = []
    for r in roots:
        n = r * N[:, 0] + N[:, 1]
        l2 = n[6] / n[2]
        if not (lmin <= l2 <= lmax):
            continue
        u3 = 1.0 + l2 * u2
        rr = n[0] * u[:, 0] + n[1] * u[:, 1] + n[2] * u3
        T = np.c_[-x[:, 0] * u[:, 0], -x[:, 0] * u[:, 1], -x[:, 0] * u3, x2 * rr, rr]
        v = np.linalg.svd(T)[2][-1]
        v = v[:4] / v[4]
        l1 = v[3]
        if not (lmin <= l1 <= lmax):
            continue
        H = np.array([n[0:3], n[3:6], v[0:3]])
        out.append((H, np.linalg.inv(H), l1, l2))
    return out


def radial_homography_errors(model, rows):
    """CheckRadialSymmetricError (:201-239), vectorised."""
    H, Hi, l1, l2 = model
    f1, f2 = rows[0, 8], rows[0, 9]
    l1s, l2s = l1 / (f1 * f1), l2 / (f2 * f2)

    def undist(p, f, l):
        return np.c_[p / (f * (1.0 + l * (p * p).sum(1)))[:, None], np.ones(len(p))]

    def dist(p3, f, l):
        p = f * p3[:, :2] / p3[:, 2:3]
        r2 = (p * p).sum(1)
        den = 2.0 * l * r2; inner = 1.0 - 4.0 * l * r2
        keep = (np.abs(den) < np.finfo(float).eps) | (inner < 0.0)
        sc = np.where(keep, 1.0, (1.0 - np.sqrt(np.maximum(inner, 0.0))) / np.where(keep, 1.0, den))
        return p * sc[:, None]
    pl, pr = rows[:, 0:2], rows[:, 2:4]
    bl, br = undist(pl, f1, l1s), undist(pr, f2, l2s)
    dl = pl - dist(br @ H.T, f1, l1s); dr = pr - dist(bl @ Hi.T, f2, l2s)
    return 0.5 * ((dl * dl).sum(1) + (dr * dr).sum(1))


# ------------------------------------------------------------------------------------------------ triangulation (round 4)
def triangulate_two_views(o1, o2):
    """Triangulate (triangulation.cc:109-125) on two observation rows of THEIA_EST_TRIANGULATION in numpy matrix form: the
    essential matrix of the two poses, the optimal image-point correction (:66-103), DLT by numpy's SVD; kept when the point is in
    front of both cameras (estimate_triangulation.cc:82-87)."""
    P1, P2 = o1[:12].reshape(3, 4), o2[:12].reshape(3, 4)
    Rr = P1[:, :3] @ P2[:, :3].T
    t = P1[:, 3] - Rr @ P2[:, 3]; t = t / np.linalg.norm(t)
    E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ Rr
    p1 = np.array([o1[12], o1[13], 1.0]); p2 = np.array([o2[12], o2[13], 1.0])
    S = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    Es = E[:2, :2]
    l1 = S @ E @ p2; l2 = S @ E.T @ p1
    a = l1 @ Es @ l2; b = (l1 @ l1 + l2 @ l2) / 2.0; c = p1 @ E @ p2
    d = np.sqrt(b * b - a * c)
    lam = c / (b + d)
    l1 = l1 - Es @ (lam * l1); l2 = l2 - Es.T @ (lam * l2)
    lam = lam * (2.0 * d) / (l1 @ l1 + l2 @ l2)
    c1 = p1 - S.T @ (lam * l1); c2 = p2 - S.T @ (lam * l2)
    c1 = c1[:2] / c1[2]; c2 = c2[:2] / c2[2]
    A = np.array([c1[0] * P1[2] - P1[0], c1[1] * P1[2] - P1[1], c2[0] * P2[2] - P2[0], c2[1] * P2[2] - P2[1]])
    X = np.linalg.svd(A)[2][-1]
    # The reference's test is `point . projection_row_2 > 0` on the null vector AS RETURNED (estimate_triangulation.cc:62-66): it reads
    # the sign Eigen's JacobiSVD happens to give the vector.  LAPACK's sign is another one, so the route takes the sign-free
    # statement of the same condition -- positive depth of the finite point X / X_w in both cameras -- which is what Eigen's sign
    # amounts to on every pair of the test scenes (231 of 231 geometrically valid pairs accepted by the oracle's restatement).
    if X[3] < 0:
        X = -X
    return [X] if (P1[2] @ X > 0.0 and P2[2] @ X > 0.0) else []


def triangulation_errors(X, rows, project):
    """TriangulationEstimator::Error (estimate_triangulation.cc:92-101): squared pixel error through Camera::ProjectPoint (`project`:
    pytheiasfm_amd.synth.project, numpy), infinite when the depth is not positive."""
    n = len(rows)
    ext = rows[:, 16:22]
    uv, _ = project(int(rows[0, 22]), rows[:, 23:33], ext, np.tile(X, (n, 1)))
    from pytheiasfm_amd import synth
    depth = np.einsum("nij,nj->ni", synth.angle_axis_to_matrix(ext[:, 3:]), X[:3] - X[3] * ext[:, :3])[:, 2] / X[3]
    e = ((uv - rows[:, 14:16]) ** 2).sum(1)
    e[~(depth > 0)] = np.inf
    return e
