"""Writes the oracle-produced golden fixtures (inputs + expected outputs).

  python tests/golden/make_oracle_golden.py

ba_small.npz     8 views / 80 tracks synth_ba_v1 scene: reduced camera system at
                 radius 1e4, LM trace and final parameters.
ransac_small.npz 3 relative-pose + 3 absolute-pose problems: sample stream,
                 per-model (iteration, cost, #inliers) trace, final inlier
                 masks, models and iteration counts for seeds 65.. / 66..
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from pytheiasfm_amd import synth  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402


def main():
    p = synth.synth_ba_v1(8, 80, seed=0xBA5E0200, num_groups=2)
    o = ol.default_options()
    S, rhs = ol.reduced_system(p, o, 1e4)
    q = p.copy()
    s, tr = ol.solve(q, o)
    np.savez_compressed(os.path.join(HERE, "ba_small.npz"), cam_ext=p.cam_ext, intrinsics=p.intrinsics,
                        group_model=p.group_model, cam_group=p.cam_group, points=p.points, obs_uv=p.obs_uv,
                        obs_cam=p.obs_cam, obs_pt=p.obs_pt, S=S, rhs=rhs, num_iterations=s.num_iterations,
                        trace_cost=tr.cost, trace_radius=tr.radius, trace_accepted=tr.accepted,
                        cam_ext_final=q.cam_ext, points_final=q.points, final_cost=s.final_cost)
    out = {}
    for kind, est, thr, seed in (("relative", 0, (2 / 1000.0) ** 2, 65), ("absolute", 2, (4 / 1000.0) ** 2, 66)):
        data, offsets, truth = synth.synth_ransac_v1(3, 120, kind, seed=0x5AC50200 + est)
        out[f"{kind}_data"] = data; out[f"{kind}_offsets"] = offsets
        for use_mle in (0, 1):
            masks, models, iters = [], [], []
            for i in range(3):
                prm = ol.default_ransac_params(thr, seed + i); prm.use_mle = use_mle
                r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm, trace_capacity=4096)
                masks.append(r["inlier_mask"]); models.append(r["model"][: 21 if est == 0 else 12]); iters.append(r["num_iterations"])
                if i == 0:
                    out[f"{kind}_mle{use_mle}_trace_iter"], out[f"{kind}_mle{use_mle}_trace_cost"], out[f"{kind}_mle{use_mle}_trace_ninl"] = r["trace"]
            out[f"{kind}_mle{use_mle}_masks"] = np.stack(masks); out[f"{kind}_mle{use_mle}_models"] = np.stack(models)
            out[f"{kind}_mle{use_mle}_iters"] = np.array(iters)
    out["sampler_65_120_5"] = ol.sampler_stream(65, 120, 5, 64)
    np.savez_compressed(os.path.join(HERE, "ransac_small.npz"), **out)
    print("wrote ba_small.npz, ransac_small.npz")




# ---------------------------------------------------------------- second set
# camera_models.npz   per camera model 64 (extrinsics, intrinsics, point, pixel) tuples, edge cases included
#                     (small rotation angle, point behind / beside the camera, homogeneous w != 1, double-sphere
#                     and EUCM invalid regions): functor return value, residual and the three Jacobian blocks.
# ba_variants.npz     a 10-view scene solved three ways: intrinsics FOCAL|RADIAL, camera priors, HUBER loss with
#                     partially constant cameras: LM traces and final parameters.
# ransac_variants.npz SQPnP solutions of fixed inputs; PROSAC / LMED / LO-RANSAC runs (masks, models, iterations).
CAMERA_MODEL_INTRINSICS = {
    0: [900.0, 1.02, 0.001, 640.0, 360.0, -0.05, 0.01],
    1: [850.0, 0.98, 0.002, 600.0, 400.0, -0.04, 0.008, 0.0005, 0.001, -0.0007],
    2: [500.0, 1.01, 0.0, 640.0, 480.0, -0.01, 0.002, -0.0003, 0.00004],
    3: [700.0, 1.0, 640.0, 360.0, 0.6],
    4: [800.0, 1.0, 640.0, 360.0, -1e-7],
    5: [600.0, 1.0, 0.0, 640.0, 360.0, -0.2, 0.55],
    6: [600.0, 1.0, 0.0, 640.0, 360.0, 0.6, 1.1],
    7: [1.5, 1.0, 0.0, 320.0, 240.0, 0.01, 0.001],   # (magnification >= 1: the focal-length bound, bundle_adjuster.cc:406)
}


def second_set():
    out = {}
    for model, k in CAMERA_MODEL_INTRINSICS.items():
        st = synth.Stream(0x601D00 + model, 0)
        n = 64
        ext = np.zeros((n, 6)); X = np.zeros((n, 4)); uv = np.zeros((n, 2))
        for i in range(n):
            u = st.uniform(np.arange(16) + 100 * i)
            ext[i, :3] = 2 * u[0:3] - 1
            ext[i, 3:] = (0.8 if i % 8 else 1e-9) * (2 * u[3:6] - 1)          # every 8th: small-angle branch
            w = 1.0 if i % 4 else 0.25 + 2 * u[6]
            depth = (4 + 6 * u[7]) if i % 16 != 5 else -(1 + u[7])             # some behind the camera
            lateral = 1.5 if i % 16 != 9 else 40.0                              # some far off axis (invalid regions)
            Xc = np.array([lateral * (2 * u[8] - 1), lateral * (2 * u[9] - 1), depth])
            R = synth.angle_axis_to_matrix(ext[i, 3:])
            X[i, :3] = w * (R.T @ Xc + ext[i, :3]); X[i, 3] = w
            uv[i] = [1280 * u[10], 720 * u[11]]
        ok = np.zeros(n, dtype=np.uint8); res = np.zeros((n, 2)); Je = np.zeros((n, 2, 6)); Ji = np.zeros((n, 2, len(k))); Jp = np.zeros((n, 2, 4))
        for i in range(n):
            o, r, a, b, c = ol.reprojection_error(model, ext[i], k, X[i], uv[i], sqrt_info=[1.0 + 0.1 * (i % 3), 0.9])
            ok[i] = 1 if o else 0; res[i] = r; Je[i] = a; Ji[i] = b; Jp[i] = c
        for name, arr in (("ext", ext), ("X", X), ("uv", uv), ("ok", ok), ("res", res), ("Je", Je), ("Ji", Ji), ("Jp", Jp)):
            out[f"m{model}_{name}"] = arr
        out[f"m{model}_intr"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "camera_models.npz"), **out)

    # BA variants
    out = {}
    base = synth.synth_ba_v1(10, 150, seed=0xBA5E0400, num_groups=3, mixed_models=True)
    for name, kw in (("intr", dict(intrinsics_to_optimize=0x11)), ("priors", dict(prior_mask=7)), ("huber", dict(loss_function_type=1, robust_loss_width=1.5)),
                     ("depth", dict(loss_function_type=1, robust_loss_width_depth_prior=0.4))):
        p = base.copy()
        if name == "priors":
            nc = 10
            i = np.arange(nc)
            mask = np.where(i % 4 == 3, 0, 7).astype(np.uint8)
            pos = p.cam_ext[:, :3] + 0.03 * np.sin(1.0 + i)[:, None]
            grav = synth.angle_axis_to_matrix(p.cam_ext[:, 3:]) @ np.array([0, 0, -1.0]) + 0.01 * np.cos(i)[:, None]
            ori = p.cam_ext[:, 3:] + 0.004 * np.sin(2.0 * i)[:, None]
            info = lambda s: np.tile(np.eye(3) * s + 0.05 * s * np.array([[0, 1, 0], [0, 0, 0], [-1, 0, 0]]), (nc, 1, 1))
            p.set_priors(mask, position=(pos, info(20.0)), gravity=(grav, info(50.0)), orientation=(ori, info(80.0)))
            for key, (v, s_) in p.priors.items():
                out[f"priors_{key}"] = v; out[f"priors_{key}_info"] = s_
            out["priors_mask"] = mask
        if name == "depth":
            # depth priors on every 4th observation: the depth at the start + a deterministic offset, variance 0.01
            idx = np.arange(0, p.obs_uv.shape[0], 4)
            R = synth.angle_axis_to_matrix(p.cam_ext[p.obs_cam[idx], 3:])
            X = p.points[p.obs_pt[idx]]
            q = np.einsum("nij,nj->ni", R, X[:, :3] - X[:, 3:] * p.cam_ext[p.obs_cam[idx], :3])
            depth = q[:, 2] + 0.05 * np.sin(0.7 * idx)
            p.add_depth_priors(idx, depth, 0.01)
            out["depth_idx"] = idx; out["depth_value"] = depth
        if name == "huber":
            p.cam_const = np.array([3, 0, 1, 2, 0, 4, 0, 0, 0, 0], dtype=np.uint8)
            p.obs_uv = p.obs_uv.copy(); p.obs_uv[::11] += 30.0
            out["huber_cam_const"] = p.cam_const; out["huber_obs_uv"] = p.obs_uv
        o = ol.default_options()
        for kk, vv in kw.items():
            setattr(o, kk, vv)
        o.max_num_iterations = 25
        s, tr = ol.solve(p, o)
        out[f"{name}_trace_cost"] = tr.cost; out[f"{name}_trace_accepted"] = tr.accepted; out[f"{name}_trace_radius"] = tr.radius
        out[f"{name}_cam_ext"] = p.cam_ext; out[f"{name}_points"] = p.points; out[f"{name}_intrinsics"] = p.intrinsics
        out[f"{name}_final_cost"] = s.final_cost; out[f"{name}_num_iterations"] = s.num_iterations
    for key in ("cam_ext", "intrinsics", "group_model", "cam_group", "points", "obs_uv", "obs_cam", "obs_pt"):
        out[f"base_{key}"] = getattr(base, key)
    np.savez_compressed(os.path.join(HERE, "ba_variants.npz"), **out)

    # RANSAC variants
    out = {}
    st = synth.Stream(0x601D50, 0)
    for k, n in enumerate((3, 4, 12, 100)):
        i = np.arange(n)
        X = np.stack([4 * st.uniform(1000 * k + 3 * i) - 2, 4 * st.uniform(1000 * k + 3 * i + 1) - 2, 6 + 4 * st.uniform(1000 * k + 3 * i + 2)], 1)
        R = synth.angle_axis_to_matrix(np.array([0.12 * (k + 1), -0.07, 0.03 * k])); t = np.array([0.3, -0.2, 0.4])
        pc = X @ R.T + t
        uv = pc[:, :2] / pc[:, 2:] + 5e-4 * np.stack([st.normal(1000 * k + 2 * i + 500), st.normal(1000 * k + 2 * i + 501)], 1)
        q, ts = ol.sqpnp(uv, X)
        out[f"sqpnp{k}_uv"] = uv; out[f"sqpnp{k}_X"] = X; out[f"sqpnp{k}_q"] = q; out[f"sqpnp{k}_t"] = ts
    data, offsets, _ = synth.synth_ransac_v1(3, 150, "absolute", seed=0x5AC50A00, noise_px=1.0)
    out["abs_data"] = data; out["abs_offsets"] = offsets
    for name, est, setup in (("prosac", 2, dict(ransac_type=1)), ("lmed", 2, dict(ransac_type=2, min_iterations=120, max_iterations=200)),
                             ("sqpnp", 4, dict()), ("lo", 2, dict(use_lo=1, lo_start_iterations=5, min_iterations=50, use_mle=1))):
        masks, models, iters = [], [], []
        for i in range(3):
            prm = ol.default_ransac_params((4 / 1000.0) ** 2, 66 + i)
            for kk, vv in setup.items():
                setattr(prm, kk, vv)
            r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm)
            masks.append(r["inlier_mask"]); models.append(r["model"][:12]); iters.append(r["num_iterations"])
        out[f"{name}_masks"] = np.stack(masks); out[f"{name}_models"] = np.stack(models); out[f"{name}_iters"] = np.array(iters)
    np.savez_compressed(os.path.join(HERE, "ransac_variants.npz"), **out)
    print("wrote camera_models.npz, ba_variants.npz, ransac_variants.npz")


NEW_ESTIMATORS = (("fundamental", 5, 4.0, 9), ("homography", 6, 16.0, 9), ("plane", 7, 0.004, 6),
                  ("known_orientation", 8, (2.0 / 1000.0) ** 2, 3), ("uncalibrated", 9, 4.0, 23))


def third_set():
    """R14 estimators + the exhaustive sampler: data, inlier masks, iteration counts, models."""
    out = {}
    for kind, est, thresh, mlen in NEW_ESTIMATORS:
        data, offsets, _ = synth.synth_ransac_v1(3, 120, kind, seed=0x5AC52000 + est, inlier_lo=0.5, inlier_hi=0.7,
                                                 noise_px=0.3 if kind == "uncalibrated" else 1.0)
        out[f"{kind}_data"] = data; out[f"{kind}_offsets"] = offsets
        ol.set_estimator_params([1.0, 1e9])
        for rtype in ((0, 1, 2, 3) if kind == "known_orientation" else (0, 1, 2)):
            masks, models, iters = [], [], []
            for i in range(3):
                prm = ol.default_ransac_params(thresh, 40 + i); prm.ransac_type = rtype; prm.failure_probability = 0.001
                if rtype == 3:
                    prm.max_iterations = 500
                r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm)
                masks.append(r["inlier_mask"]); models.append(r["model"][:mlen]); iters.append(r["num_iterations"])
            out[f"{kind}_t{rtype}_masks"] = np.stack(masks); out[f"{kind}_t{rtype}_models"] = np.stack(models)
            out[f"{kind}_t{rtype}_iters"] = np.array(iters)
    # absolute position with known orientation: rotated correspondences
    data, offsets, truth = synth.synth_ransac_v1(3, 120, "absolute", seed=0x5AC52010, inlier_lo=0.5, inlier_hi=0.7)
    from pytheiasfm_amd import ransac as rs
    rot = np.concatenate([rs.RotateCorrespondences(data[offsets[i]:offsets[i + 1]], synth.matrix_to_angle_axis(truth["R"][i])) for i in range(3)])
    out["abs_known_data"] = rot; out["abs_known_offsets"] = offsets
    masks, models, iters = [], [], []
    for i in range(3):
        prm = ol.default_ransac_params((4.0 / 1000.0) ** 2, 40 + i); prm.failure_probability = 0.001
        r = ol.ransac_estimate(10, rot[offsets[i]:offsets[i + 1]], prm)
        masks.append(r["inlier_mask"]); models.append(r["model"][:3]); iters.append(r["num_iterations"])
    out["abs_known_t0_masks"] = np.stack(masks); out["abs_known_t0_models"] = np.stack(models); out["abs_known_t0_iters"] = np.array(iters)
    np.savez_compressed(os.path.join(HERE, "ransac_estimators.npz"), **out)
    print("wrote ransac_estimators.npz")


def two_view_cases():
    """(correspondences, start pose) of the BundleAdjustTwoViewsAngular vectors: inliers of synthetic pairs, perturbed truth."""
    data, off, truth = synth.synth_ransac_v1(4, 200, kind="relative", noise_px=0.5, seed=0x5AC52100)
    for p in range(4):
        c = data[off[p]:off[p + 1]][truth["inlier"][p]]
        w = synth.matrix_to_angle_axis(truth["R"][p]); pos = truth["position"][p] / np.linalg.norm(truth["position"][p])
        x0 = np.concatenate([w + 0.005 * (p + 1), pos + 0.01 * (p - 1.5)]); x0[3:] /= np.linalg.norm(x0[3:])
        yield c, x0


def fourth_set():
    """Relative-pose LO-RANSAC (RefineModel = BundleAdjustTwoViewsAngular) and the two-view adjustment itself."""
    from pytheiasfm_amd import ba
    out = {}
    o = ba.default_options(); o.max_num_iterations = 15; o.loss_function_type = 6; o.robust_loss_width = 2e-4
    for k, (c, x0) in enumerate(two_view_cases()):
        pose, s = ol.two_views_angular(c, x0, o)
        out[f"tv{k}_corr"] = c; out[f"tv{k}_x0"] = x0; out[f"tv{k}_pose"] = pose
        out[f"tv{k}_ints"] = np.array([s["success"], s["termination_type"], s["num_iterations"], s["num_successful_steps"]])
        out[f"tv{k}_costs"] = np.array([s["initial_cost"], s["final_cost"]])
    data, offsets, _ = synth.synth_ransac_v1(3, 150, "relative", seed=0x5AC52101, inlier_lo=0.5, inlier_hi=0.7)
    out["rel_data"] = data; out["rel_offsets"] = offsets
    masks, models, iters, nlo = [], [], [], []
    for i in range(3):
        prm = ol.default_ransac_params((2.0 / 1000.0) ** 2, 50 + i)
        prm.use_mle = 1; prm.use_lo = 1; prm.lo_start_iterations = 5; prm.min_iterations = 50; prm.failure_probability = 0.001
        r = ol.ransac_estimate(0, data[offsets[i]:offsets[i + 1]], prm)
        masks.append(r["inlier_mask"]); models.append(r["model"][:21]); iters.append(r["num_iterations"])
        nlo.append(ol.rlib().oracle_last_lo_iterations())
    out["rel_lo_masks"] = np.stack(masks); out["rel_lo_models"] = np.stack(models); out["rel_lo_iters"] = np.array(iters)
    out["rel_lo_nlo"] = np.array(nlo)
    # OptimizeHomography / OptimizeFundamentalMatrix vectors and the LO runs of the estimators that call them
    from tests.test_oracle_ransac import _homography_scene, _fundamental_scene
    oh = ba.default_options(); oh.max_num_iterations = 15; oh.loss_function_type = 6; oh.robust_loss_width = 50.0
    of = ba.default_options(); of.max_num_iterations = 2
    for k in range(2):
        H, c = _homography_scene(40 + k, n=90 + 20 * k)
        H0 = H * (1.0 + 0.5 * k) + np.array([[0.01, -0.01, 2.0 + k], [0.01, 0.0, -2.0], [1e-6, 0, 0.0]])
        Hr, s_ = ol.optimize_homography(c, H0, oh)
        out[f"hom{k}_corr"] = c; out[f"hom{k}_H0"] = H0; out[f"hom{k}_H"] = Hr
        out[f"hom{k}_ints"] = np.array([s_["num_iterations"], s_["num_successful_steps"]]); out[f"hom{k}_costs"] = np.array([s_["initial_cost"], s_["final_cost"]])
        F, c = _fundamental_scene(0x5AC52800 + k, n=100 + 20 * k)
        F0 = F * (1.0 + k) + (k + 2) * 1e-8 * np.array([[1.0, -2, 300], [2, 1, -200], [-300, 200, 5e4]])
        Fr, s_ = ol.optimize_fundamental(c, F0, of)
        out[f"fund{k}_corr"] = c; out[f"fund{k}_F0"] = F0; out[f"fund{k}_F"] = Fr
        out[f"fund{k}_ints"] = np.array([s_["num_iterations"], s_["num_successful_steps"]]); out[f"fund{k}_costs"] = np.array([s_["initial_cost"], s_["final_cost"]])
    ol.set_estimator_params([1.0, 1e9])
    for kind, est, thresh, mlen in (("fundamental", 5, 4.0, 9), ("homography", 6, 16.0, 9), ("uncalibrated", 9, 4.0, 23)):
        data, offsets, _ = synth.synth_ransac_v1(2, 150, kind, seed=0x5AC52900 + est, inlier_lo=0.5, inlier_hi=0.7,
                                                 noise_px=0.3 if kind == "uncalibrated" else 1.0)
        out[f"lo_{kind}_data"] = data; out[f"lo_{kind}_offsets"] = offsets
        masks, models, iters, nlo = [], [], [], []
        for i in range(2):
            prm = ol.default_ransac_params(thresh, 60 + i); prm.failure_probability = 0.001
            prm.use_lo = 1; prm.lo_start_iterations = 5; prm.min_iterations = 30
            r = ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], prm)
            masks.append(r["inlier_mask"]); models.append(r["model"][:mlen]); iters.append(r["num_iterations"])
            nlo.append(ol.rlib().oracle_last_lo_iterations())
        out[f"lo_{kind}_masks"] = np.stack(masks); out[f"lo_{kind}_models"] = np.stack(models)
        out[f"lo_{kind}_iters"] = np.array(iters); out[f"lo_{kind}_nlo"] = np.array(nlo)
    np.savez_compressed(os.path.join(HERE, "two_view_lo.npz"), **out)
    print("wrote two_view_lo.npz")


if __name__ == "__main__":
    import sys
    if "--only-fourth" in sys.argv:
        fourth_set()
        sys.exit(0)
    main()
    second_set()
    third_set()
    fourth_set()
