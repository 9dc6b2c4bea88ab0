"""GPU box: random batches of the estimators scripts/soak_ransac.py does not draw -- gDLS similarity (13), P4Pf (14), UPnP (15), P4Pfr
(16) -- the library's inlier sets / iteration counts / elected models against the CPU oracle's, bit for bit.  A soak, not a test.
usage: soak_ransac_more.py [batches per estimator] [first seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ransac
from tests import oracle_lib as ol
from tests import gdls_scenes as gs, p4pf_scenes as ps, upnp_scenes as us, p4pfr_scenes as rs

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
META = ransac.RadialDistUncalibratedAbsolutePoseMetaData(min_focal_length=100.0, max_focal_length=2000.0, min_radial_distortion=-1e-9,
                                                         max_radial_distortion=-1e-5)
t_all = time.time(); bad = 0; total = 0
for est, mlen in ((13, 13), (14, 12), (15, 12), (16, 14)):
    diff = 0; n_est = 0
    for k in range(nb):
        seed = seed0 + k
        rng = np.random.default_rng(0x50AC7000 + 1009 * est + seed)
        nprob = int(rng.integers(2, 6))
        data, offsets = [], [0]
        for r in range(nprob):
            if est == 13:
                corr, _ = gs.cameras(int(rng.integers(3, 6)), int(rng.integers(60, 160)), seed=int(rng.integers(1 << 30)), outlier_frac=float(rng.choice([0.0, 0.2, 0.35])),
                                     noise=float(rng.choice([0.0, 0.5])), scale=float(rng.uniform(0.8, 2.5)))
                rows = ransac.similarity_correspondence_rows(corr)
            elif est == 14:
                rows, _, _, _ = ps.ransac_scene(rng, int(rng.integers(40, 200)), outlier_fraction=float(rng.choice([0.1, 0.25, 0.4])))
            elif est == 15:
                q = us.quat_angle_axis(float(rng.uniform(3, 40)), rng.normal(size=3)); t = rng.uniform(-1.5, 1.5, 3)
                rows, _ = us.rig_rows(rng, int(rng.integers(60, 180)), int(rng.integers(1, 5)), q, t, outlier_fraction=float(rng.choice([0.0, 0.1, 0.3])),
                                      pixel_noise=float(rng.choice([0.0, 0.3, 1.0])))
            else:
                R = rs.angle_axis(float(rng.uniform(3, 30)), rng.normal(size=3)); t = rng.uniform(-1, 1, 3) * [1.0, 1.0, 0.2]
                rows = rs.estimator_scene(rng, R, t, float(rng.choice([0.6, 0.8, 1.0])), float(rng.choice([0.0, 0.5])), n=int(rng.integers(30, 120)))
            data.append(rows); offsets.append(offsets[-1] + len(rows))
        data = np.concatenate(data); offsets = np.array(offsets, dtype=np.int64)
        p = ransac.RansacParameters(); p.error_thresh = float(rng.choice([2.0, 3.0])) ** 2
        p.min_iterations = int(rng.choice([40, 100, 200])); p.max_iterations = int(rng.choice([300, 700]))
        p.failure_probability = float(rng.choice([0.01, 0.001])); p.seed = int(rng.integers(0, 2 ** 31 - 64)); p.use_mle = bool(rng.integers(0, 2))
        pc0 = p.to_c(); pc0.ransac_type = int(rng.choice([0, 0, 1, 2]))
        ep = np.concatenate([META.limits(), [float(rng.integers(0, 2))]]) if est == 16 else None
        res = ransac.estimate_batch(est, data, offsets, pc0, ep)
        if est == 16: ol.set_estimator_params(ep)
        try:
            for i in range(nprob):
                sl = slice(offsets[i], offsets[i + 1])
                pc = p.to_c(); pc.seed = (p.seed + i) & 0xFFFFFFFF; pc.ransac_type = pc0.ransac_type
                o = ol.ransac_estimate(est, data[sl], pc)
                same = (bool(o["success"]) == bool(res["success"][i]) and o["num_iterations"] == res["num_iterations"][i]
                        and np.array_equal(o["inlier_mask"], res["inlier_mask"][sl])
                        and (not o["success"] or np.array_equal(o["model"][:mlen], res["models"][i][:mlen], equal_nan=True)))
                n_est += 1
                if not same:
                    diff += 1
                    nm = int(np.sum(o["inlier_mask"] != res["inlier_mask"][sl]))
                    md = float(np.nanmax(np.abs(o["model"][:mlen] - res["models"][i][:mlen]))) if o["success"] else -1.0
                    print(f"estimator {est} seed {seed} problem {i}: DIFFERS (iterations {o['num_iterations']} / {res['num_iterations'][i]}, success {o['success']} / {res['success'][i]}, "
                          f"{nm} of {sl.stop - sl.start} mask entries, model diff {md:.3g}, type {pc0.ransac_type}, mle {p.use_mle}, n {sl.stop - sl.start})", flush=True)
        finally:
            if est == 16: ol.set_estimator_params([0.0] * 5)
    print(f"estimator {est}: {n_est} problems, {diff} differ", flush=True)
    bad += diff; total += n_est
print(f"soak: {total} problems, {bad} differ, {time.time() - t_all:.0f} s")
