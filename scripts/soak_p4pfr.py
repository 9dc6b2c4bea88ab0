"""GPU box: the directly bound P4Pfr solver against the oracle on many random minimal problems (all solutions, all 14 numbers,
bit for bit).  usage: soak_p4pfr.py [problems]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import ransac
from tests import oracle_lib as ol
from tests import test_p4pfr_gpu as T
num = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
F, W = T._minimal_problems(num, 2024)
draws = np.random.default_rng(7).uniform(-0.5, 0.5, (num, 3))
meta = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=0.0, max_focal_length=1e5, min_radial_distortion=0.0, max_radial_distortion=-1.0)
ns, M, _ = ransac.FourPointsPoseFocalLengthRadialDistortion(F, W, meta, rotation_draws=draws)
bad = 0
for i in range(num):
    o = ol.p4pfr_solve(F[i], W[i], draws[i], meta.limits())
    bad += not (len(o) == ns[i] and np.array_equal(o, M[i, :ns[i]]))
print(f"P4Pfr soak: {num} minimal problems, {int(ns.sum())} solutions, {bad} problems differ from the oracle")
sys.exit(1 if bad else 0)
