"""CPU: the numpy-only DLS route of tests/numpy_routes.py (LAPACK solve + eig on a Macaulay system built from the
definitions) against the oracle's DlsPnp -- the same solution sets on well-posed problems, so that the equality counts of
tests/test_independent_routes_gpu.py compare two implementations of ONE estimator."""
import numpy as np

from pytheiasfm_amd import ransac
from tests import dls_scenes as sc
from tests import numpy_routes as nr
from tests import oracle_lib as ol


def test_numpy_dls_route_finds_the_oracles_solutions():
    rng = np.random.default_rng(3)
    same = total = 0
    for k in range(60):
        n = 3 if k % 2 else 6
        qq = rng.normal(size=4); qq /= np.linalg.norm(qq)
        cam = np.c_[rng.uniform(-1, 1, (n, 2)), rng.uniform(2, 6, n)]; t = rng.normal(size=3)
        world = (cam - t) @ sc.quat_to_rot(qq); feat = cam[:, :2] / cam[:, 2:3]
        qo, to = ol.dls_pnp(feat, world, call_index=k)
        sols = nr.dls_pnp(feat, world, ransac.dls_macaulay_terms(k, 1)[0])
        # the generating pose is found by both
        truth = sc.quat_to_rot(qq)
        tol = 1e-2 if n == 3 else 1e-4      # (a minimal sample can be ill-conditioned: the elimination's own accuracy)
        assert min(np.abs(R - truth).max() for R, _ in sols) < tol
        assert min(np.abs(sc.quat_to_rot(q) - truth).max() for q in qo) < tol
        total += 1
        same += len(sols) == len(qo) and all(min(np.abs(sc.quat_to_rot(q) - R).max() + np.abs(tt - tn).max() for R, tn in sols) < 1e-6
                                             for q, tt in zip(qo, to))
    # (the remaining problems have a root whose imaginary part sits at the 1e-6 filter: two eigen-solvers split them differently)
    assert same >= 0.85 * total, (same, total)
