"""CPU: host-side logic -- synthetic generators, the AddView/AddTrack
flattening of the Python mirror, track sharding, and the world_size-2 (gloo)
reduction algebra of the multi-GPU path."""
import hashlib
import os
import sys

import numpy as np
import pytest

from pytheiasfm_amd import _capi as capi, sfm, synth
from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def test_synth_ba_is_deterministic_and_well_formed():
    a = synth.ba_config("C1"); b = synth.ba_config("C1")
    assert digest(a.cam_ext, a.points, a.obs_uv, a.obs_cam, a.obs_pt) == digest(b.cam_ext, b.points, b.obs_uv, b.obs_cam, b.obs_pt)
    assert a.cam_ext.shape == (20, 6) and a.points.shape == (2000, 4)
    L = np.bincount(a.obs_pt, minlength=2000)
    assert L.min() >= 2 and L.max() <= 10 and 5.0 < L.mean() < 7.0
    assert (a.obs_uv[:, 0] > -5).all() and (a.obs_uv[:, 0] < 1925).all()
    # every track observes distinct cameras
    key = a.obs_pt.astype(np.int64) * 1000 + a.obs_cam
    assert len(np.unique(key)) == len(key)


def test_synth_mixed_models_and_gauge():
    p = synth.synth_ba_v1(16, 200, seed=3, mixed_models=True, fix_gauge=True)
    assert set(p.group_model.tolist()) == {0, 5}
    assert p.cam_const[0] == 3 and p.cam_const[8] == 3 and p.cam_const.sum() == 6


def test_flatten_matches_add_view_add_track_semantics():
    p = synth.synth_ba_v1(6, 40, seed=5, num_groups=2)
    rec = sfm.Reconstruction.from_flat(p)
    rec.view_estimated[5] = False
    rec.track_estimated[7] = False
    # BundleAdjustPartialReconstruction(views {0,1}, tracks {0..9})
    flat = sfm._flatten(rec, [0, 1], list(range(10)))
    gc = np.ones(rec.group_intrinsics.shape[0], np.uint8); gc[rec.view_group[[0, 1]]] = 0
    assert np.array_equal(flat.group_const, gc)      # groups of AddView'ed views are optimised, others constant
    ov, ot = flat.obs_cam, flat.obs_pt
    assert not np.any(ov == 5) and not np.any(ot == 7)          # unestimated blocks add nothing
    in_view = np.isin(ov, [0, 1]); in_track = np.isin(ot, np.arange(10))
    assert np.all(in_view | in_track)                           # only AddView or AddTrack residuals
    full = (p.obs_cam != 5) & (p.obs_pt != 7) & (np.isin(p.obs_cam, [0, 1]) | np.isin(p.obs_pt, np.arange(10)))
    assert len(ov) == full.sum()
    assert list(flat.cam_const[:2]) == [0, 0] and np.all(flat.cam_const[2:] == 3)  # AddTrack freezes other cameras (:204)
    assert np.all(flat.point_const[:10][rec.track_estimated[:10]] == 0) and np.all(flat.point_const[10:] == 1)
    with pytest.raises(capi.TheiaHipError):
        sfm._flatten(rec, [99], [])


def test_options_mirror_defaults_and_unsupported():
    o = sfm.BundleAdjustmentOptions()
    assert o.loss_function_type == sfm.LossFunctionType.TRIVIAL and o.use_inner_iterations and o.max_num_iterations == 100
    assert o.intrinsics_to_optimize == sfm.OptimizeIntrinsicsType.NONE and o.use_homogeneous_point_parametrization
    o.use_position_priors = True; o.use_orientation_priors = True
    assert o.to_c().prior_mask == (capi.THEIA_PRIOR_POSITION | capi.THEIA_PRIOR_ORIENTATION)
    o.use_depth_priors = True; o.robust_loss_width_depth_prior = 0.25
    assert o.to_c().robust_loss_width_depth_prior == 0.25   # depth priors travel as observation rows (see _flatten)
    o.use_inverse_depth_parametrization = True     # travels as a problem flag (THEIA_BA_FLAG_INVERSE_DEPTH), not as an option
    assert o.to_c().max_num_iterations == 100
    # forward-facing trajectories: same reduced system; with inner iterations AND variable intrinsics Ceres rejects the
    # reversed ordering (its first set {extrinsics, intrinsics} is not independent) and the solve fails untouched
    f = sfm.BundleAdjustmentOptions(); f.optimize_for_forward_facing_trajectory = True
    assert f.to_c().use_inner_iterations == 1
    f.intrinsics_to_optimize = sfm.OptimizeIntrinsicsType.FOCAL_LENGTH
    p = synth.synth_ba_v1(4, 40, seed=3)
    rec = sfm.Reconstruction.from_flat(p)
    before = rec.cam_ext.copy()
    summ = sfm.BundleAdjustReconstruction(f, rec)       # returns before any device call
    assert not summ.success and np.array_equal(rec.cam_ext, before)


def test_flatten_emits_depth_prior_rows_for_added_views_only():
    """bundle_adjuster.cc:152-156: a DepthPriorError per feature with depth_prior != 0 of the views that went
    through AddView, when use_depth_priors is set; none for observations reached through AddTrack."""
    p = synth.synth_ba_v1(5, 30, seed=0xF1A7)
    rec = sfm.Reconstruction.from_flat(p)
    n = len(rec.obs_view)
    rec.obs_depth_prior = np.where(np.arange(n) % 2 == 0, 4.0 + 0.01 * np.arange(n), 0.0)
    rec.obs_depth_prior_variance = np.full(n, 0.04)
    o = sfm.BundleAdjustmentOptions(); o.use_depth_priors = True
    flat = sfm._flatten(rec, rec.ViewIds(), rec.TrackIds(), options=o)
    extra = flat.obs_uv.shape[0] - n
    assert extra == (rec.obs_depth_prior != 0).sum() and flat.obs_kind[n:].all() and not flat.obs_kind[:n].any()
    assert np.array_equal(flat.obs_uv[n:, 0], rec.obs_depth_prior[rec.obs_depth_prior != 0])
    assert np.allclose(flat.obs_sqrt_info[n:, 0], 5.0) and np.array_equal(flat.obs_cam[n:], rec.obs_view[rec.obs_depth_prior != 0])
    # only view 2 goes through AddView: its features carry priors, the others (reached via AddTrack) do not
    flat = sfm._flatten(rec, [2], rec.TrackIds(), options=o)
    k = flat.obs_uv.shape[0] - n
    assert k == ((rec.obs_depth_prior != 0) & (rec.obs_view == 2)).sum() and np.all(flat.obs_cam[n:] == 2)
    o.use_depth_priors = False
    assert sfm._flatten(rec, rec.ViewIds(), rec.TrackIds(), options=o).obs_kind is None


def test_shard_tracks_partitions_every_track_once():
    p = synth.ba_config("C1")
    seen = np.zeros(2000, dtype=int); work = []
    for r in range(4):
        sh, ids = synth.shard_tracks(p, r, 4)
        seen[ids] += 1
        assert sh.flags & 1 and sh.cam_ext.shape == p.cam_ext.shape
        assert np.array_equal(sh.points, p.points[ids])
        L = np.bincount(sh.obs_pt, minlength=len(ids))
        work.append(float((L * L).sum()))
        assert sh.obs_uv.shape[0] == np.isin(p.obs_pt, ids).sum()
    assert np.all(seen == 1)
    assert max(work) / min(work) < 1.2


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytheiasfm_amd import distributed as tdist
    p = synth.synth_ba_v1(8, 120, seed=0xBA5E0300, num_groups=2)
    o = ol.default_options()
    sh, ids = synth.shard_tracks(p, rank, world)
    # the protocol of the sharded GPU path: all-reduce the unscaled camera column
    # norms (Jacobi scaling), then all-reduce [S | rhs | scaled column norms]
    allreduce = tdist.make_host_allreduce()
    colsq = ol.colnorms(sh, o)
    allreduce(colsq)
    S, rhs, cs = ol.reduced_system_partial(sh, o, 1e4, colsq)
    n = S.shape[0]
    buf = np.concatenate([S.ravel(), rhs, cs])
    allreduce(buf)
    S = buf[: n * n].reshape(n, n) + np.diag(np.clip(buf[n * n + n:], 1e-6, 1e32) / 1e4)  # finalize step
    buf = np.concatenate([S.ravel(), buf[n * n: n * n + n]])
    pts = tdist.gather_points(sh.points, ids, p.points.shape[0])
    if rank == 0:
        Sf, rf = ol.reduced_system(p, o, 1e4)
        n = Sf.shape[0]
        q.put((float(np.abs(buf[: n * n].reshape(n, n) - Sf).max() / np.abs(Sf).max()),
               float(np.abs(buf[n * n:] - rf).max() / np.abs(rf).max()), bool(np.array_equal(pts, p.points))))
    dist.destroy_process_group()


def test_world_size_2_reduction_of_the_reduced_camera_system():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    s_err, r_err, pts_ok = res
    assert s_err < 1e-12 and r_err < 1e-10 and pts_ok


def _gloo_inner_exchange_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytheiasfm_amd import distributed as tdist
    p = synth.synth_ba_v1(9, 150, seed=0xBA5E0301)
    sh, ids = synth.shard_tracks(p, rank, world)
    allreduce = tdist.make_host_allreduce()
    # 1. the full candidate point set: every shard's points at their global indices, summed
    rng = np.random.default_rng(7)
    cand = p.points * (1.0 + 1e-3 * rng.standard_normal(p.points.shape))      # the same "candidate" on every rank
    full = tdist.gather_points(cand[ids], ids, p.points.shape[0])
    # 2. cameras dealt to the ranks by index: the owner's result reaches every rank bit for bit
    swept = p.cam_ext + 1e-2 * np.sin(np.arange(p.cam_ext.size)).reshape(p.cam_ext.shape)   # what the owner computes
    mine = p.cam_ext.copy()                                                                # non-owners hold something else
    own = (np.arange(len(mine)) % world) == rank
    mine[own] = swept[own]
    out = tdist.exchange_owned_blocks(mine, rank, world, allreduce)
    # 3. the sweep's scalars: the point shares summed over the ranks, the camera share added once
    part = np.array([float((cand[ids, :3] ** 2).sum()), 0.0])
    allreduce(part)
    q.put((rank, bool(np.array_equal(full, cand)), bool(np.array_equal(out, swept)), float(part[0]), float((cand[:, :3] ** 2).sum())))
    dist.destroy_process_group()


def test_world_size_2_exchange_of_the_sharded_inner_iterations():
    """The collectives of a sharded solve's inner iterations (theia_hip_ba_set_inner_global) on two CPU ranks with gloo:
    zero-filled buffers summed into the full point set / the dealt cameras (exact: x + 0), the point share of a norm
    summed over the shards."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_inner_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert {r[0] for r in res} == {0, 1}
    for _, pts_ok, cams_ok, s, ref in res:
        assert pts_ok and cams_ok and abs(s - ref) <= 1e-12 * ref


def test_ransac_pair_partition_is_a_round_robin_cover():
    """RANSAC over N GPUs: every pair belongs to exactly one rank."""
    from pytheiasfm_amd import distributed as tdist
    for P, W in ((0, 4), (1, 4), (10, 3), (10000, 8)):
        parts = [tdist.shard_problems(P, r, W) for r in range(W)]
        allp = np.sort(np.concatenate(parts)) if P else np.zeros(0, dtype=np.int64)
        assert np.array_equal(allp, np.arange(P))
        assert max(len(x) for x in parts) - min(len(x) for x in parts) <= 1


def test_two_view_front_end_host_logic():
    """estimate_twoview_info.cc:67-102,155-171 + reconstruction_estimator_utils.cc:98-110."""
    from pytheiasfm_amd import twoview as tv
    assert tv.ComputeResolutionScaledThreshold(6.0, 0, 0) == 6.0
    assert tv.ComputeResolutionScaledThreshold(6.0, 2048, 1536) == 12.0
    p1 = tv.CameraIntrinsicsPrior(); p1.image_width = 1000; p1.image_height = 800
    p1.focal_length.is_set = True; p1.focal_length.value = [900.0]
    p2 = tv.CameraIntrinsicsPrior(); p2.image_width = 640; p2.image_height = 480
    p2.focal_length.is_set = True; p2.focal_length.value = [700.0]
    p2.principal_point.is_set = True; p2.principal_point.value = [300.0, 250.0]
    c = np.array([[600.0, 500.0, 400.0, 300.0]])
    n = tv.NormalizeFeatures(p1, p2, c)
    assert np.allclose(n, [[(600 - 500) / 900.0, (500 - 400) / 900.0, (400 - 300) / 700.0, (300 - 250) / 700.0]])
    # one focal length missing: both are reset to 1, only the principal point is removed
    p2.focal_length.is_set = False
    n = tv.NormalizeFeatures(p1, p2, c)
    assert np.allclose(n, [[100.0, 100.0, 100.0, 50.0]])
    # skew and aspect ratio (pinhole_camera_model.h:228-232)
    p2.focal_length.is_set = True
    p2.aspect_ratio.is_set = True; p2.aspect_ratio.value = [1.1]; p2.skew.is_set = True; p2.skew.value = [2.0]
    n = tv.NormalizeFeatures(p1, p2, c)
    y = (300 - 250) / (700.0 * 1.1)
    assert np.allclose(n[0, 2:], [(400 - 300 - y * 2.0) / 700.0, y])
    # unsupported priors are refused, not approximated
    p2.radial_distortion.is_set = True; p2.radial_distortion.value = [0.1, 0, 0, 0]
    with pytest.raises(Exception, match="radial distortion"):
        tv.NormalizeFeatures(p1, p2, c)
    p1.camera_intrinsics_model_type = "FISHEYE"
    with pytest.raises(Exception, match="PINHOLE"):
        tv.NormalizeFeatures(p1, p1, c)
    # Eigen::AngleAxisd(R)
    from pytheiasfm_amd import synth
    for w in ([0.1, -0.2, 0.3], [2.0, 1.0, -1.5], [0.0, 0.0, 0.0], [3.0, 0.2, 0.1]):
        w = np.array(w)
        assert np.allclose(tv._rotation_to_angle_axis(synth.angle_axis_to_matrix(w)), w, atol=1e-12)


def test_select_good_tracks_rules():
    """select_good_tracks_for_bundle_adjustment.cc:164-320 on hand-made statistics: one track per grid cell (minimum
    of (truncated length, error)), then top-up in ascending track id until a view has K optimised tracks."""
    r = sfm.Reconstruction()
    r.cam_ext = np.zeros((2, 6)); r.view_estimated = np.array([True, True]); r.view_group = np.zeros(2, np.int32)
    r.group_model = np.zeros(1, np.int32); r.group_intrinsics = np.zeros((1, 10))
    nt = 8
    r.points = np.zeros((nt, 4)); r.track_estimated = np.ones(nt, dtype=bool); r.track_estimated[7] = False
    # view 0 sees tracks 0..5 and 7: cells (0,0): {0,1,2}, (1,0): {3}, (0,1): {4,5,7}; view 1 sees 0, 5, 6 in one cell
    uv0 = {0: (10, 10), 1: (20, 30), 2: (90, 99), 3: (150, 20), 4: (5, 120), 5: (60, 180), 7: (50, 150)}
    uv1 = {0: (10, 10), 5: (20, 20), 6: (30, 30)}
    ov, ot, ouv = [], [], []
    for t, p in uv0.items(): ov.append(0); ot.append(t); ouv.append(p)
    for t, p in uv1.items(): ov.append(1); ot.append(t); ouv.append(p)
    r.obs_view = np.array(ov, np.int32); r.obs_track = np.array(ot, np.int32); r.obs_uv = np.array(ouv, float)
    tlen = np.array([2, 2, 3, 5, 9, 2, 1, 1]); terr = np.array([0.5, 0.2, 0.1, 1.0, 0.3, 0.4, 0.9, 0.0])
    # threshold 4 truncates track 4's length to 4.  Cell winners: view 0: (0,0) -> track 1 (length 2, error 0.2 < 0.5),
    # (1,0) -> 3, (0,1) -> 5 (length 2 beats truncated 4; 7 is not estimated); view 1: one cell {0, 5, 6} -> 6 (length 1)
    sel = sfm._select_good_tracks(r, [0, 1], tlen, terr, 4, 100, 0)
    assert sel.tolist() == [1, 3, 5, 6]
    # K = 5: view 0 has 3 optimised of 6 estimated -> adds the lowest ids 0, 2; view 1 then has {0, 5, 6} = all
    sel = sfm._select_good_tracks(r, [0, 1], tlen, terr, 4, 100, 5)
    assert sel.tolist() == [0, 1, 2, 3, 5, 6]
    # K larger than the view: everything estimated in it
    sel = sfm._select_good_tracks(r, [0], tlen, terr, 4, 100, 50)
    assert sel.tolist() == [0, 1, 2, 3, 4, 5]


def test_pairs_of_a_verification_batch_as_one_flat_problem():
    """twoview._pairs_flat: the pairs of VerifyMatchesBatch as ONE flat problem for the triangulation and the per-view
    statistics sweeps -- cameras 2 i / 2 i + 1, per pair the observations of camera 1 then of camera 2, a point either
    one track seen twice or one track per view.  Pure host logic (no device call)."""
    from pytheiasfm_amd import twoview as tv
    rng = np.random.default_rng(5)
    ns = [7, 3, 11]
    corr = [rng.normal(size=(n, 4)) for n in ns]
    pts = [rng.normal(size=(n, 4)) for n in ns]
    cams = [[{"ext": rng.normal(size=6), "intr": np.array([900.0 + 10 * i + k, 1, 0, 500, 400, 0, 0]), "model": 0} for k in range(2)]
            for i in range(3)]
    flat, got = tv._pairs_flat(cams, corr, pts, False)
    assert got == ns and flat.cam_ext.shape == (6, 6) and flat.points.shape == (sum(ns), 4) and flat.obs_uv.shape == (2 * sum(ns), 2)
    base = 0
    o = 0
    for i, n in enumerate(ns):
        assert np.array_equal(flat.cam_ext[2 * i], cams[i][0]["ext"]) and flat.intrinsics[2 * i + 1, 0] == cams[i][1]["intr"][0]
        assert np.array_equal(flat.obs_cam[o:o + n], np.full(n, 2 * i)) and np.array_equal(flat.obs_cam[o + n:o + 2 * n], np.full(n, 2 * i + 1))
        assert np.array_equal(flat.obs_pt[o:o + n], base + np.arange(n)) and np.array_equal(flat.obs_pt[o + n:o + 2 * n], base + np.arange(n))
        assert np.array_equal(flat.obs_uv[o:o + n], corr[i][:, 0:2]) and np.array_equal(flat.obs_uv[o + n:o + 2 * n], corr[i][:, 2:4])
        assert np.array_equal(flat.points[base:base + n], pts[i])
        base += n; o += 2 * n
    flat2, _ = tv._pairs_flat(cams, corr, pts, True)
    assert flat2.points.shape == (2 * sum(ns), 4) and np.array_equal(flat2.obs_pt, np.arange(2 * sum(ns)))
    assert np.array_equal(flat2.points[:ns[0]], pts[0]) and np.array_equal(flat2.points[ns[0]:2 * ns[0]], pts[0])
    assert np.array_equal(flat2.obs_cam, flat.obs_cam) and np.array_equal(flat2.cam_group, np.arange(6))
    empty, _ = tv._pairs_flat(cams, corr, None, False)
    assert empty.points.shape == (sum(ns), 4) and not empty.points.any()


def test_decompose_projection_matrix_recovers_calibration_rotation_and_position():
    """DecomposeProjectionMatrix (projection_matrix_utils.cc:74-117) as the uncalibrated absolute-pose mirror uses it: random
    K [R | -R c] at any overall sign and scale gives back K (positive diagonal), R and c."""
    from pytheiasfm_amd import ransac, synth
    rng = np.random.default_rng(2)
    for _ in range(20):
        K = np.array([[rng.uniform(300, 900), rng.uniform(-2, 2), rng.uniform(-20, 20)], [0, rng.uniform(300, 900), rng.uniform(-20, 20)], [0, 0, 1.0]])
        R = synth.angle_axis_to_matrix(rng.uniform(-1, 1, (1, 3)))[0]
        c = rng.uniform(-2, 2, 3)
        P = K @ np.concatenate([R, (-R @ c)[:, None]], axis=1) * rng.choice([-1.0, 1.0]) * rng.uniform(0.5, 2.0)
        ok, Kd, aa, pos = ransac.DecomposeProjectionMatrix(P)
        assert ok and np.all(np.diag(Kd) > 0)
        assert np.abs(Kd / Kd[2, 2] - K).max() < 1e-8 and np.abs(pos - c).max() < 1e-9
        assert np.abs(synth.angle_axis_to_matrix(aa[None])[0] - R).max() < 1e-9


def test_problem_fingerprint_separates_topology_from_parameters():
    """ba.problem_fingerprint (the key of the problem-IR cache): unchanged by what a solve changes (extrinsics, intrinsics,
    points), changed by any edit of the topology, the observations, the constant masks, the priors or the options."""
    from pytheiasfm_amd import ba, synth
    p = synth.synth_ba_v1(12, 300, seed=5)
    o = ba.default_options() if os.path.exists(capi.LIB_PATH) else None
    k0 = ba.problem_fingerprint(p, o)
    q = p.copy()
    q.cam_ext += 0.01; q.points[:, :3] += 0.1; q.intrinsics[:, 0] *= 1.01
    assert ba.problem_fingerprint(q, o) == k0
    keys = {k0}
    q = p.copy(); q.obs_uv = q.obs_uv.copy(); q.obs_uv[7, 0] += 1e-9; keys.add(ba.problem_fingerprint(q, o))
    q = p.copy(); q.obs_cam = q.obs_cam.copy(); q.obs_cam[3] = (q.obs_cam[3] + 1) % 12; keys.add(ba.problem_fingerprint(q, o))
    q = p.copy(); q.point_const = np.zeros(300, np.uint8); q.point_const[5] = 1; keys.add(ba.problem_fingerprint(q, o))
    q = p.copy(); q.set_priors(np.ones(12, np.uint8), position=(np.zeros((12, 3)), np.tile(np.eye(3), (12, 1, 1)))); keys.add(ba.problem_fingerprint(q, o))
    q = p.copy(); q.add_depth_priors([1, 2], [3.0, 4.0]); keys.add(ba.problem_fingerprint(q, o))
    r = synth.synth_ba_v1(12, 299, seed=5); keys.add(ba.problem_fingerprint(r, o))
    assert len(keys) == 7
    if o is not None:
        o2 = ba.default_options(); o2.max_num_iterations = o.max_num_iterations + 1
        assert ba.problem_fingerprint(p, o2) != k0


def test_host_thread_team_survives_a_fork():
    """ba_solver.hip HostTeam: the worker threads of the parent do not exist in a fork child, and the region state
    (generation, worker count, active count) must not be inherited either.  The validation pass of create() runs its
    observation scan on the team (THEIA_HIP_HOST_CHUNK_MIN lowers the size at which it does) before any device call, so it
    is reachable without a GPU: the parent runs it, forks, and the child runs it again -- with an out-of-range index in the
    LAST part, which only a complete parallel region finds."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        os.environ["THEIA_HIP_HOST_CHUNK_MIN"] = "64"; os.environ["THEIA_HIP_HOST_THREADS"] = "6"
        from pytheiasfm_amd import ba, _capi as capi, synth
        p = synth.synth_ba_v1(8, 400, seed=3)
        def create(q):
            try:
                ba.BaHandle(q, ba.default_options()).close()
                return "ok"
            except capi.TheiaHipError as e:
                return str(e)
        first = create(p)                       # parent: the team's workers start here
        bad = p.copy(); bad.obs_cam = bad.obs_cam.copy(); bad.obs_cam[-1] = 10 ** 6
        pid = os.fork()
        if pid == 0:
            msgs = [create(bad) for _ in range(3)] + [create(p)]
            ok = all("out of range" in m for m in msgs[:3]) and msgs[3] == first
            os._exit(0 if ok else 3)
        _, status = os.waitpid(pid, 0)
        again = create(bad)
        sys.exit(0 if (os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0 and "out of range" in again) else 4)
        """ % ROOT)
    r = subprocess.run([sys.executable, "-c", code], timeout=120, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])


def test_bench_self_launch_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) re-runs itself under torch.distributed.run with one
    process per GPU on 127.0.0.1 and a free port, the original arguments kept (bench.py: self_launch)."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    def fake_call(cmd, env=None):
        seen["cmd"] = cmd; seen["env"] = env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:4] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    i = cmd.index("--master-addr"); assert cmd[i + 1] == "127.0.0.1"
    j = cmd.index("--master-port"); assert 1024 < int(cmd[j + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and os.path.basename(cmd[-7]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
