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


def test_numpy_routes_replay_the_oracles_ransac_on_cpu():
    """The round-4 routes (essential matrix, 8-point, 4-point homography, Kneip P3P, UPnP) against the ORACLE's RANSAC loop at a
    small shape -- the CPU-side twin of test_c5_shape_inlier_sets_against_an_independent_numpy_route: identical inlier sets."""
    import os
    from pytheiasfm_amd import synth
    NP, CORR, HY = 2, 300, 96
    layout = os.path.join(os.path.dirname(__file__), "..", "oracle", "upnp_layout.h")
    legs = {"essential": (1, "relative", 5, (2.0 / 1000.0) ** 2), "fundamental": (5, "relative", 8, (2.0 / 1000.0) ** 2),
            "p3p": (2, "absolute", 3, (4.0 / 1000.0) ** 2), "upnp": (15, "absolute", 4, (4.0 / 1000.0) ** 2)}
    for leg, (est, kind, m, thr) in legs.items():
        data, offsets, _ = synth.synth_ransac_v1(NP, CORR, kind, seed=0x5AC50005)
        if leg == "upnp":
            # a three-camera rig without outliers (with them the accumulating estimator ends at a handful of inliers, where a model
            # from a complex eigenvector pair -- which the numpy route skips -- can tie): pixel errors through the datum's camera
            from tests import upnp_scenes as us
            rng = np.random.default_rng(4)
            rows = [us.rig_rows(rng, CORR, 3, us.quat_angle_axis(10.0 + 9 * k, (0.2, 1.0, -0.4)), np.array([0.3, -0.2, 0.5 * k]), pixel_noise=0.3)[0] for k in range(NP)]
            data = np.concatenate(rows); thr = 2.0 ** 2
        for i in range(NP):
            d = data[offsets[i]:offsets[i + 1]]
            pc = ol.default_ransac_params(thr, seed=1 + i); pc.min_iterations = HY; pc.max_iterations = HY
            o = ol.ransac_estimate(est, d, pc)
            samples = ol.sampler_stream(1 + i, len(d), m, HY)
            if leg in ("essential", "fundamental"):
                x1, x2 = d[:, :2], d[:, 2:4]
                x1h = np.c_[x1, np.ones(len(d))]; x2h = np.c_[x2, np.ones(len(d))]
                solver = nr.five_point if leg == "essential" else nr.eight_point
                fit = lambda it, idx: solver(x1[idx], x2[idx])
                err = lambda F: nr.sampson_errors(F, x1h, x2h)
            elif leg == "p3p":
                feat, world = d[:, :2], d[:, 2:5]
                fit = lambda it, idx: nr.p3p_kneip(feat[idx], world[idx])
                err = lambda mm: nr.absolute_pose_errors(mm, feat, world)
            else:
                feat, world = d[:, 7:9], d[:, 3:6]
                route = nr.UpnpRoute(layout)
                fit = lambda it, idx: route.fit(d[idx, 9:12], d[idx, 0:3], world[idx])
                Rc = synth.angle_axis_to_matrix(d[:, 12:15])
                def err(mm):
                    p3 = np.einsum("nij,nj->ni", Rc, world @ mm[0].T + mm[1] - d[:, 9:12])
                    e = ((d[:, 16:17] * p3[:, :2] / p3[:, 2:3] - feat) ** 2).sum(1)      # pinhole, principal point 0, no distortion
                    e[p3[:, 2] < 0] = np.inf
                    return e
            mask, _ = nr.ransac_inlier_support(samples, fit, err, thr, len(d))
            assert np.array_equal(mask, o["inlier_mask"].astype(bool)), (leg, i, int(mask.sum()), int(o["inlier_mask"].sum()))
