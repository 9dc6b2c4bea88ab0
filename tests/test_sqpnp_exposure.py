"""CPU: how much does SQPnP's RANSAC result depend on the BASIS Eigen's JacobiSVD returns inside the null space of Omega?

VERDICT r4, weak 1: for a 3-point minimal sample Omega (9 x 9) is rank deficient and SQPnP starts its SQP iterations from its
null vectors (sfm/pose/sqpnp.cc:239-275; rank 2 n - 3 = 3, i.e. SIX of them); which orthonormal basis of that space JacobiSVD hands back is a
property of its sweep order and of the rounding of the reference build -- the oracle restates the sweeps (oracle/ransac_oracle.cpp:
svd_sq, with the Eigen steps written next to it) but cannot be pinned against Eigen here.  This test measures the exposure
instead: it rotates the oracle's null-space basis by random rotations (oracle_set_sqpnp_null_rotation) and replays the
configs[4]-shape RANSAC (2000 correspondences, 4096 hypotheses, InlierSupport) on the bench's first 16 pairs."""
import ctypes as C
import json
import os

import numpy as np

from pytheiasfm_amd import _capi as capi, ransac, synth
from tests import oracle_lib as ol


NULL_DIM = 6   # a 3-point sample: Omega has rank 2 n - 3 = 3


def _rotation(rng, d=NULL_DIM):
    """A random rotation of R^d (QR of a Gaussian matrix, determinant +1)."""
    Q, R = np.linalg.qr(rng.normal(size=(d, d)))
    Q = Q * np.sign(np.diag(R))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    return Q


def _set_rotation(R):
    L = ol.rlib()
    L.oracle_set_sqpnp_null_rotation.argtypes = [capi.c_double_p, C.c_int]
    L.oracle_set_sqpnp_null_rotation.restype = None
    if R is None:
        L.oracle_set_sqpnp_null_rotation(None, 0)
    else:
        R = np.ascontiguousarray(R, dtype=np.float64)
        L.oracle_set_sqpnp_null_rotation(capi.ptr(R, C.c_double), R.shape[0])


def _run(data, offsets, npairs, hyps):
    out = []
    for i in range(npairs):
        p = ransac.RansacParameters(); p.error_thresh = (4.0 / 1000.0) ** 2; p.min_iterations = hyps; p.max_iterations = hyps
        pc = p.to_c(); pc.seed = 1 + i
        o = ol.ransac_estimate(ransac.EST_ABS_SQPNP, data[offsets[i]:offsets[i + 1]], pc)
        out.append((o["inlier_mask"].copy(), int(o["num_inliers"]), o["model"][:12].copy()))
    return out


def test_sqpnp_inlier_sets_under_a_rotated_null_space_basis():
    NP, CORR, HYPS, NROT = 16, 2000, 4096, 6
    data, offsets, truth = synth.synth_ransac_v1(NP, CORR, "absolute", seed=0x5AC50005)
    try:
        _set_rotation(None)
        base = _run(data, offsets, NP, HYPS)
        _set_rotation(np.eye(NULL_DIM))
        same = _run(data, offsets, NP, HYPS)
        # the hook with the identity is the unhooked solver: bit for bit
        assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2], equal_nan=True) for a, b in zip(base, same))
        rng = np.random.default_rng(20260930)
        changed, worst, dn, signed, recov = [], 0, 0, [], []
        planted = np.array([truth["inlier"][i].sum() for i in range(NP)], dtype=np.float64)
        for k in range(NROT):
            _set_rotation(_rotation(rng))
            rot = _run(data, offsets, NP, HYPS)
            c = 0
            for i, (a, b) in enumerate(zip(base, rot)):
                d = int((a[0] != b[0]).sum())
                c += d > 0
                worst = max(worst, d)
                dn = max(dn, abs(a[1] - b[1]))
                signed.append(b[1] - a[1])
                recov.append(b[1] / planted[i])
            changed.append(c)
    finally:
        _set_rotation(None)
    report = {"pairs": NP, "correspondences": CORR, "hypotheses": HYPS, "rotations": NROT, "null_space_dimension": NULL_DIM,
              "pairs_with_a_changed_inlier_set": changed, "largest_symmetric_difference": worst,
              "largest_change_of_the_inlier_count": dn, "mean_change_of_the_inlier_count": float(np.mean(signed)),
              "mean_abs_change_of_the_inlier_count": float(np.mean(np.abs(signed))),
              "mean_inliers": float(np.mean([b[1] for b in base])),
              "recovered_fraction_of_planted_inliers_baseline": float(np.mean([b[1] / planted[i] for i, b in enumerate(base)])),
              "recovered_fraction_of_planted_inliers_rotated_min": float(np.min(recov)),
              "recovered_fraction_of_planted_inliers_rotated_mean": float(np.mean(recov))}
    print("\n[SQPnP exposure]", json.dumps(report))
    out = os.environ.get("THEIA_SQPNP_EXPOSURE_JSON")
    if out:
        with open(out, "w") as f:
            json.dump(report, f, indent=1)
    # What the measurement says: the basis MATTERS.  With one SQP iteration per start (sqpnp.cc:33-34) the
    # hypothesis of a 3-point sample is a function of the basis of Omega's 6-dimensional null space, so another orthonormal
    # basis elects another best-of-4096 model on every pair.  "Bit-identical" for this leg therefore means device = oracle on
    # the restated sweep order; against the reference it holds only as far as that restatement reproduces Eigen's bits.  What
    # does not depend on the basis is that the estimator works: every rotated run still finds most of the planted inliers.
    assert max(changed) > 0
    assert np.min(recov) > 0.5
