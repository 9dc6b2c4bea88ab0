"""Strecha fountain-P11 as the reference ships it (data/sfm/fountain11.bin: the output of a Theia pipeline run, 11
views, 16 616 tracks, 75 022 observations, one shared pinhole intrinsics group; gt_fountain11.bin: the ground-truth
cameras).  tests/golden/fountain11.npz holds the parsed arrays (tests/golden/make_fountain_fixture.py)."""
import os

import numpy as np

from pytheiasfm_amd import _capi as capi

NPZ = os.path.join(os.path.dirname(__file__), "golden", "fountain11.npz")


def load():
    d = np.load(NPZ)
    assert (d["intrinsics_model"] == "theia::PinholeCameraModel").all() and np.ptp(d["intrinsics"], axis=0).max() == 0.0
    return d


def flat_problem(d, cam_ext=None, points=None):
    """One intrinsics group (the file's cameras share one CameraIntrinsicsModel object)."""
    nv = d["cam_ext"].shape[0]
    pinhole = np.array([0], dtype=np.int32)   # CameraIntrinsicsModelType::PINHOLE
    return capi.FlatProblem(d["cam_ext"] if cam_ext is None else cam_ext, d["intrinsics"][:1], pinhole,
                            np.zeros(nv, dtype=np.int32), d["points"] if points is None else points,
                            d["obs_uv"], d["obs_cam"], d["obs_track"])


def aa_to_rot(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def reproject(d, cam_ext, points):
    """Pinhole (pinhole_camera_model.h): camera = R (X / w - c), pixel = (f x + s y + px, f a y + py), two radial terms."""
    f, a, s, px, py, k1, k2 = d["intrinsics"][0]
    R = np.stack([aa_to_rot(e[3:]) for e in cam_ext])
    X = points[d["obs_track"], :3] / points[d["obs_track"], 3:4]
    pc = np.einsum("nij,nj->ni", R[d["obs_cam"]], X - cam_ext[d["obs_cam"], :3])
    x, y = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    r2 = x * x + y * y
    dist = 1.0 + r2 * (k1 + k2 * r2)
    x, y = x * dist, y * dist
    return np.stack([f * x + s * y + px, f * a * y + py], axis=1), pc[:, 2]


def similarity_align(src, dst):
    """Umeyama: s, R, t minimising |dst - (s R src + t)| (AlignReconstructions, sfm/transformation/align_reconstructions.cc)."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    S, Dd = src - mu_s, dst - mu_d
    U, sig, Vt = np.linalg.svd(Dd.T @ S / len(src))
    E = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        E[2, 2] = -1
    R = U @ E @ Vt
    s = np.trace(np.diag(sig) @ E) / (S ** 2).sum() * len(src)
    t = mu_d - s * R @ mu_s
    return s, R, t


def pair_correspondences(d, i, j):
    """Normalised-image correspondences of the tracks views i and j share (the file's radial distortion is zero)."""
    f, a, s, px, py, _, _ = d["intrinsics"][0]
    oi = np.flatnonzero(d["obs_cam"] == i); oj = np.flatnonzero(d["obs_cam"] == j)
    common, ii, jj = np.intersect1d(d["obs_track"][oi], d["obs_track"][oj], return_indices=True)

    def norm(uv):
        y = (uv[:, 1] - py) / (f * a)
        x = (uv[:, 0] - px - s * y) / f
        return np.stack([x, y], axis=1)
    return np.ascontiguousarray(np.concatenate([norm(d["obs_uv"][oi[ii]]), norm(d["obs_uv"][oj[jj]])], axis=1))
