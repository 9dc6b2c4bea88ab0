"""Scenes for the gDLS similarity tests.  generalised(): the reference's solver scenes
(gdls_similarity_transform_test.cc:57-215: rays from camera centres to world points under a known (R, t, s)).
cameras(): a rig of pinhole cameras observing a point cloud, the rig expressed in a frame that differs from the world by a
similarity -- the localisation problem EstimateSimilarityTransformation2D3D solves."""
import numpy as np

from pytheiasfm_amd import ransac, synth


def rotation_z(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])


def generalised(points, centers, R, t, s):
    """ray_origins, ray_directions with  s c_i + alpha_i x_i = R X_i + t  (gdls_similarity_transform_test.cc:72-86)."""
    points = np.asarray(points, dtype=np.float64)
    centers = np.asarray(centers, dtype=np.float64)
    origins = np.array([(R @ centers[i % len(centers)] + t) / s for i in range(len(points))])
    rays = points @ R.T + t - s * origins
    return origins, rays / np.linalg.norm(rays, axis=1, keepdims=True)


def quat_to_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def cameras(num_cams, num_points, seed, outlier_frac=0.0, noise=0.0, scale=1.7):
    """Returns (correspondences, truth): pinhole cameras of a rig (local frame) observing world points; the rig maps to the
    world by  X_world = s R (x_rig) + t  applied to the cameras (TransformCamera's convention)."""
    rng = np.random.default_rng(seed)
    Rw = synth.angle_axis_to_matrix(rng.uniform(-0.4, 0.4, (1, 3)))[0]
    tw = rng.uniform(-1.0, 1.0, 3)
    k = np.array([900.0, 1.0, 0.0, 640.0, 480.0, 0.0, 0.0])
    cams_world, cams_rig = [], []
    for c in range(num_cams):
        aa = rng.uniform(-0.2, 0.2, 3)
        pos = np.array([1.5 * (c - (num_cams - 1) / 2.0), rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3)])
        cams_world.append((pos, aa))
        # the rig frame: position and orientation such that TransformCamera(s, Rw, tw) gives the world camera
        Rc = synth.angle_axis_to_matrix(aa[None])[0]
        pos_rig = Rw.T @ (pos - tw) / scale
        aa_rig = synth.matrix_to_angle_axis((Rc @ Rw)[None])[0]
        cams_rig.append(ransac.Camera(pos_rig, aa_rig, k, synth.CAM_PINHOLE))
    corr, is_out = [], []
    for i in range(num_points):
        X = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(5, 9)])
        c = int(rng.integers(num_cams))
        pos, aa = cams_world[c]
        uv, ok = synth.project(synth.CAM_PINHOLE, k[None], np.concatenate([pos, aa])[None], np.append(X, 1.0)[None])
        px = uv[0] + noise * rng.normal(size=2)
        out = rng.uniform() < outlier_frac
        if out:
            px = rng.uniform([0.0, 0.0], [1280.0, 960.0])
        corr.append(ransac.CameraAndFeatureCorrespondence2D3D(cams_rig[c], px, X))
        is_out.append(out)
    return corr, dict(R=Rw, t=tw, s=scale, outlier=np.array(is_out), cams_world=cams_world, k=k)
