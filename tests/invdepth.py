"""Inverse-depth scenes for the tests: a synth_ba_v1 problem re-expressed in the reference's inverse-depth
parametrisation -- reference view = first observing view of a track, bearing = the reference camera's normalised ray
(x, y, 1) of the (noisy) feature as Reconstruction::... sets it (estimate_track.cc:281: PixelToNormalizedCoordinates),
inverse depth = 1 / depth in that view (bundle_adjustment.cc:67-83 UpdateInverseDepth)."""
import numpy as np

from pytheiasfm_amd import synth


def aa_to_rot(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make(num_views=8, num_tracks=200, seed=5, rho_noise=0.02):
    """Returns a FlatProblem with set_inverse_depth() applied (pinhole groups only)."""
    p = synth.synth_ba_v1(num_views, num_tracks, seed=seed)
    nt = p.points.shape[0]
    order = np.argsort(p.obs_pt, kind="stable")
    first = np.full(nt, -1, dtype=np.int64)
    tr, idx = np.unique(p.obs_pt[order], return_index=True)
    first[tr] = order[idx]                       # first observation of every track
    ref_cam = p.obs_cam[first].astype(np.int32)
    bearing = np.zeros((nt, 3)); rho = np.zeros(nt)
    rng = np.random.default_rng(seed)
    for t in range(nt):
        c = ref_cam[t]
        f, a, s, px, py = p.intrinsics[p.cam_group[c]][:5]
        u, v = p.obs_uv[first[t]]
        y = (v - py) / (f * a); x = (u - px - s * y) / f
        bearing[t] = (x, y, 1.0)
        X = p.points[t, :3] / p.points[t, 3]
        depth = (aa_to_rot(p.cam_ext[c, 3:]) @ (X - p.cam_ext[c, :3]))[2]
        rho[t] = (1.0 / depth) * (1.0 + rho_noise * rng.normal())
    p.set_inverse_depth(ref_cam, bearing, rho)
    return p


def world_points(p):
    """UpdateHomogeneousPoint (bundle_adjustment.cc:47-65): X = R_ref^T (b / rho) + c_ref."""
    out = np.ones((p.points.shape[0], 4))
    for t in range(p.points.shape[0]):
        c = p.point_ref_cam[t]
        out[t, :3] = aa_to_rot(p.cam_ext[c, 3:]).T @ (p.point_ref_bearing[t] / p.point_inverse_depth[t]) + p.cam_ext[c, :3]
    return out
