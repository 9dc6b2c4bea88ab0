"""GPU box: BundleAdjustViewsWithCov with free intrinsics on random small problems (views, shared groups, free-parameter masks,
view subsets drawn per seed) against numpy's dense inverse of J'J assembled from the oracle's Jets.  A soak, not a test.
usage: soak_cov_intr.py [count] [first seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import sfm, synth
from tests import oracle_lib as ol

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
FREE_OF_BIT = {0x01: [0], 0x02: [1], 0x04: [2], 0x08: [3, 4], 0x10: [5, 6]}   # pinhole: f, aspect, skew, principal point, k1 k2
bad = 0; worst = 0.0; t0 = time.time()
for k in range(count):
    seed = seed0 + k
    rng = np.random.default_rng(0xC0F00000 + seed)
    nv = int(rng.integers(4, 41)); nt = int(rng.integers(200, 700)); ng = int(rng.integers(1, min(6, nv) + 1))
    mask = 0
    while mask == 0: mask = int(rng.integers(0, 32)) & ~0x04          # (skew of a noise-free pinhole set is weakly observed: left out)
    p = synth.synth_ba_v1(nv, nt, seed=0xC0F00000 + seed, pixel_noise=0.5, num_groups=ng)
    opts = sfm.BundleAdjustmentOptions(); opts.max_num_iterations = int(rng.integers(3, 12)); opts.intrinsics_to_optimize = mask
    rec = sfm.Reconstruction.from_flat(p)
    nsel = int(rng.integers(2, nv + 1))
    views = sorted(int(v) for v in rng.choice(nv, nsel, replace=False))
    try:
        sv, cv, fv = sfm.BundleAdjustViewsWithCov(rec, opts, views)
    except Exception as e:
        print(f"seed {seed}: {nv} views {ng} groups mask {mask:#x} {nsel} selected: raised {e!r}", flush=True); bad += 1; continue
    flat = sfm._flatten(rec, views, [], options=opts)
    free = [q for b, qs in FREE_OF_BIT.items() if mask & b for q in qs]
    groups = sorted(set(int(flat.cam_group[v]) for v in views))
    gcol = {g: 6 * len(views) + len(free) * i for i, g in enumerate(groups)}
    ccol = {v: 6 * i for i, v in enumerate(views)}
    ncol = 6 * len(views) + len(free) * len(groups)
    JTJ = np.zeros((ncol, ncol)); nres = 0
    for i in np.flatnonzero(np.isin(flat.obs_cam, views)):
        c = int(flat.obs_cam[i]); g = int(flat.cam_group[c])
        ok, r, Je, Ji, Jp = ol.reprojection_error(int(flat.group_model[g]), flat.cam_ext[c], flat.intrinsics[g][:7], flat.points[flat.obs_pt[i]], flat.obs_uv[i])
        idx = np.concatenate([np.arange(ccol[c], ccol[c] + 6), gcol[g] + np.arange(len(free))])
        Jo = np.concatenate([Je, Ji[:, free]], axis=1)
        JTJ[np.ix_(idx, idx)] += Jo.T @ Jo; nres += 2
    cond = np.linalg.cond(JTJ)
    cov = np.linalg.inv(JTJ) * fv
    err = max(np.abs(cv[v] - cov[ccol[v]:ccol[v] + 6, ccol[v]:ccol[v] + 6]).max() / np.abs(cov[ccol[v]:ccol[v] + 6, ccol[v]:ccol[v] + 6]).max() for v in views)
    tol = max(1e-6, 1e-15 * cond * 10)
    flag = "" if (err <= tol and sv.success) else "  <-- DIFFERS"
    bad += bool(flag); worst = max(worst, err / tol)
    print(f"seed {seed}: {nv} views {ng} groups mask {mask:#04x} {nsel} selected cond {cond:.1e}: rel err {err:.2e}{flag}", flush=True)
print(f"soak: {count} problems, {bad} differ, worst err / tolerance {worst:.2e}, {time.time() - t0:.0f} s")
