"""Scenes for the EstimateTriangulation tests, shaped like estimate_triangulation_test.cc:57-99: cameras with a random
rotation of up to 5 degrees, positions in a box around a diagonal, all looking at the point (0, 0, 8); outlier
observations carry a random pixel instead of the projection."""
import numpy as np

from pytheiasfm_amd import ransac, synth

POINT = np.array([0.0, 0.0, 8.0, 1.0])


def scene(num_observations, num_outliers, seed, model=synth.CAM_PINHOLE, intrinsics=None, spread=0.5, noise=0.0):
    rng = np.random.default_rng(seed)
    k = np.array([1000.0, 1.0, 0.0, 600.0, 400.0, 0.0, 0.0]) if intrinsics is None else np.asarray(intrinsics, dtype=np.float64)
    n = num_observations + num_outliers
    cams, feats = [], np.zeros((n, 2))
    for i in range(n):
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        aa = axis * np.deg2rad(5.0) * rng.uniform()
        j = i if i < num_observations else i - num_observations
        pos = rng.uniform(-1.0, 1.0, 3) + spread * j
        cam = ransac.Camera(pos, aa, k, model)
        ext = np.concatenate([pos, aa])[None]
        uv, ok = synth.project(model, cam.intrinsics[None], ext, POINT[None])
        assert ok[0]
        feats[i] = uv[0] + noise * rng.normal(size=2) if i < num_observations else rng.uniform(0.0, 1000.0, 2)
        cams.append(cam)
    return cams, feats
