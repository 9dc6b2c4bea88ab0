"""GPU box (development): wall time of the one-problem-per-wavefront batches whose inner loops are wave-wide sums:
OptimizeHomography x 4000 (300 matches each) and OptimizeRelativePositionWithKnownRotation x 4000 (400 matches each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ba, synth
from tests.test_oracle_ransac import _homography_scene
H, c = _homography_scene(20, n=300)
N = 4000
corr = np.tile(c, (N, 1)); offs = (np.arange(N + 1) * len(c)).astype(np.int64)
H0 = H * 1.3 + np.array([[0.01, -0.01, 2.0], [0.01, 0.0, -2.0], [1e-6, 0, 0.0]])
o = ba.default_options(); o.max_num_iterations = 15
for rep in range(3):
    Hd = np.tile(H0, (N, 1, 1)).copy()
    t0 = time.perf_counter(); summ = ba.optimize_homography_batch(offs, corr, Hd, o); dt = time.perf_counter() - t0
print("homography LM: %d problems x %d matches, %d iterations each: %.2f ms = %.3f M problems/s" % (N, len(c), summ[0].num_iterations, 1e3 * dt, N / dt / 1e6))
rc, offsets, rot, _ = synth.synth_relpos_v1(N, 400, seed=5, noise=5e-4)
for rep in range(3):
    t0 = time.perf_counter(); pos, it = ba.optimize_relative_position_batch(offsets, rc, rot); dt = time.perf_counter() - t0
print("relative position IRLS: %d pairs x 400 matches, %.1f iterations on average: %.2f ms = %.3f M pairs/s" % (N, float(np.mean(it)), 1e3 * dt, N / dt / 1e6))
