"""GPU: BASELINE.json configs[2] stand-in on real data -- the reference-held fountain-P11 reconstruction (real feature
tracks, real topology, shared intrinsics): full bundle adjustment with FOCAL_LENGTH | RADIAL_DISTORTION against the oracle
from the converged state and from a perturbed one, camera centres against the ground truth, and five-point RANSAC
verification of its view pairs against the oracle."""
import numpy as np
import pytest

from pytheiasfm_amd import ba, ransac, sfm
from tests import fountain as ft
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _options(mod, intr, loss=None, iters=12):
    o = mod.default_options()
    o.max_num_iterations = iters
    o.intrinsics_to_optimize = int(intr)
    if loss is not None:
        o.loss_function_type = int(loss); o.robust_loss_width = 2.0
    return o


def _compare(pg, sg, po, so):
    assert sg.success and so.success
    assert sg.num_iterations == so.num_iterations, (sg.num_iterations, so.num_iterations)
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    # north_star: point / pose parameters within 1e-6 relative
    assert np.abs(pg.cam_ext - po.cam_ext).max() <= 1e-6 * max(1.0, np.abs(po.cam_ext).max())
    assert np.abs(pg.points - po.points).max() <= 1e-6 * max(1.0, np.abs(po.points).max())
    assert np.abs(pg.intrinsics - po.intrinsics).max() <= 1e-6 * np.abs(po.intrinsics).max()


@pytest.mark.parametrize("intr,loss", [(sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION, None),
                                       (sfm.OptimizeIntrinsicsType.NONE, sfm.LossFunctionType.HUBER)])
def test_bundle_adjustment_from_the_converged_state(intr, loss):
    d = ft.load()
    pg, po = ft.flat_problem(d), ft.flat_problem(d)
    sg, _ = ba.solve(pg, _options(ba, intr, loss))
    so, _ = ol.solve(po, _options(ol, intr, loss))
    _compare(pg, sg, po, so)
    assert sg.final_cost <= sg.initial_cost


def test_bundle_adjustment_from_a_perturbed_state_recovers_the_ground_truth_cameras():
    d = ft.load()
    rng = np.random.default_rng(1105)
    cam = d["cam_ext"].copy(); pts = d["points"].copy()
    cam[:, :3] += rng.normal(scale=2e-3, size=(11, 3)); cam[:, 3:] += rng.normal(scale=1e-3, size=(11, 3))
    pts[:, :3] += rng.normal(scale=5e-3, size=(len(pts), 3))
    intr = sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION
    pg, po = ft.flat_problem(d, cam.copy(), pts.copy()), ft.flat_problem(d, cam.copy(), pts.copy())   # (FlatProblem wraps, it does not copy)
    sg, _ = ba.solve(pg, _options(ba, intr, iters=25))
    so, _ = ol.solve(po, _options(ol, intr, iters=25))
    _compare(pg, sg, po, so)
    assert sg.final_cost < 0.05 * sg.initial_cost
    uv, depth = ft.reproject(d, pg.cam_ext, pg.points)
    # (reproject() uses the file's intrinsics; the optimised focal / distortion move by < 1e-3 relative)
    assert np.sqrt(((uv - d["obs_uv"]) ** 2).sum(axis=1).mean()) < 1.0
    s, R, t = ft.similarity_align(pg.cam_ext[:, :3], d["gt_cam_ext"][:, :3])
    res = np.linalg.norm(d["gt_cam_ext"][:, :3] - (s * pg.cam_ext[:, :3] @ R.T + t), axis=1)
    assert res.max() < 1e-2, res          # incremental_reconstruction_estimator_test.cc:156


def test_mirror_api_on_the_real_reconstruction():
    d = ft.load()
    recon = sfm.Reconstruction.from_flat(ft.flat_problem(d))
    opts = sfm.BundleAdjustmentOptions()
    opts.max_num_iterations = 5
    opts.intrinsics_to_optimize = sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION
    summary = sfm.BundleAdjustReconstruction(opts, recon)
    assert summary.success and summary.final_cost <= summary.initial_cost


def test_five_point_verification_of_the_view_pairs_matches_the_oracle():
    d = ft.load()
    pairs = [(i, j) for i in range(11) for j in range(i + 1, 11)]
    corr = [ft.pair_correspondences(d, i, j) for i, j in pairs]
    keep = [k for k, c in enumerate(corr) if len(c) >= 100]
    assert len(keep) >= 20
    pairs = [pairs[k] for k in keep]; corr = [corr[k] for k in keep]
    offsets = np.zeros(len(corr) + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(c) for c in corr])
    data = np.ascontiguousarray(np.concatenate(corr, axis=0))
    prm = ransac.RansacParameters(); prm.error_thresh = (2.0 / 2759.48) ** 2; prm.min_iterations = 128; prm.max_iterations = 128; prm.seed = 9
    res = ransac.estimate_batch(ransac.EST_RELATIVE_POSE, data, offsets, prm)
    for k, (i, j) in enumerate(pairs):
        pc = prm.to_c(); pc.seed = prm.seed + k
        o = ol.ransac_estimate(ransac.EST_RELATIVE_POSE, corr[k], pc)
        gm = res["inlier_mask"][offsets[k]:offsets[k + 1]]
        assert np.array_equal(o["inlier_mask"], gm), f"inlier set differs on pair {(i, j)}"
        assert res["success"][k]
        # tracks of a finished reconstruction are inliers of its geometry
        assert gm.mean() > 0.8, ((i, j), gm.mean())
        # relative rotation of the RANSAC model against the reconstruction's cameras
        Ri, Rj = ft.aa_to_rot(d["cam_ext"][i, 3:]), ft.aa_to_rot(d["cam_ext"][j, 3:])
        Rrel = Rj @ Ri.T
        Rm = res["models"][k][9:18].reshape(3, 3)
        ang = np.degrees(np.arccos(np.clip((np.trace(Rm @ Rrel.T) - 1) / 2, -1, 1)))
        assert ang < 1.0, ((i, j), ang)


def test_reference_points_are_stationary_points_of_the_device_cost():
    """tests/test_fountain.py::test_reference_points_are_stationary_points_of_the_oracle_cost through the HIP path: started at
    the points Ceres left in fountain11.bin (cameras constant, TRIVIAL loss, SphereManifold<4>), the device LM stops by
    Ceres' own rules without moving 16 560 of the 16 616 tracks (relative move < 1e-9), exactly the oracle's set; and the
    batched per-track entry point (theia_hip_ba_tracks_batch = N x BundleAdjustTrack, the call that produced those points in
    the reference) leaves the same tracks in place."""
    from tests.test_fountain import stationary_tracks
    d = ft.load()
    o = ba.default_options(); o.max_num_iterations = 10
    still, move, rel_dec, pg = stationary_tracks(d, ba.solve, o)
    oo = ol.default_options(); oo.max_num_iterations = 10
    still_o, _, _, po = stationary_tracks(d, ol.solve, oo)
    assert still.sum() >= 16560 and np.array_equal(still, still_o)
    assert move[still].max() < 1e-9 and abs(rel_dec) < 1e-6
    assert np.abs(pg.points - po.points).max() <= 1e-9 * np.abs(po.points).max()
    # N x BundleAdjustTrack in one launch (one LM per thread)
    pt = ft.flat_problem(d, d["cam_ext"].copy(), d["points"].copy())
    ba.solve_tracks_batch(pt, o)
    mv = np.linalg.norm(pt.points - d["points"], axis=1) / np.linalg.norm(d["points"], axis=1)
    assert (mv[still] < 1e-9).all() and (mv < 1e-9).sum() >= 16560
