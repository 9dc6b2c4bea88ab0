"""Generates tests/golden/relpos_irls.npz: inputs and the ORACLE's outputs (oracle_optimize_relative_position_with_known_rotation,
oracle/ransac_oracle.cpp) for OptimizeRelativePositionWithKnownRotation on seeded synthetic view pairs -- regression vectors for
the oracle itself and known answers for the HIP batch (tests/test_relpos.py, tests/test_relpos_gpu.py).  The reference cannot
be built here (Ceres / Eigen / glog absent), so these are not reference outputs; the noise-free cases carry the true direction.
    python tests/golden/make_relpos_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import oracle_lib as ol            # noqa: E402
from tests.test_relpos import make_pairs      # noqa: E402

if __name__ == "__main__":
    pairs = make_pairs()
    out = {"num": len(pairs)}
    for k, (corr, r1, r2, truth) in enumerate(pairs):
        pos, it = ol.optimize_relative_position(corr, r1, r2)
        posw, itw = ol.optimize_relative_position(corr, r1, r2, order=1)   # sums in the device's wavefront order
        out[f"corr{k}"] = corr; out[f"rot{k}"] = np.concatenate([r1, r2]); out[f"truth{k}"] = truth
        out[f"pos{k}"] = pos; out[f"it{k}"] = it; out[f"posw{k}"] = posw; out[f"itw{k}"] = itw
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "relpos_irls.npz"), **out)
    print("wrote", len(pairs), "pairs")
