#!/usr/bin/env python
"""bench.py -- BA LM-iterations/s + residuals/s and RANSAC hypotheses/s on N x MI355X, with the kernel rooflines
and the CPU baseline (host cores of the same box) in the same JSON line.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE Levenberg-Marquardt iteration of the BA hot path (Jacobian evaluation + Schur elimination +
reduced-camera solve + back-substitution + trial-cost evaluation) on a synthetic reconstruction resident in HBM.

Headline workload (`north_star`, BASELINE.json configs[3] on one GPU): synth_ba_v1 "C4" = 1000 views / 500 000 tracks
/ ~3.0 M observations, mixed pinhole + double-sphere cameras.  N > 1 runs the SAME problem with its tracks sharded
over the ranks (strong scaling) and the reduced camera system all-reduced with RCCL every iteration.
Secondary blocks of the same line: "c2" (configs[1], 200 views / 50k tracks), "ransac" (configs[4]: 10 000 pairs x
2000 correspondences x 4096 hypotheses, five-point relative pose and SQPnP absolute pose).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ITERS_PER_SOLVE = 8      # LM iterations per solve from the perturbed start (tolerances off: exactly this many)
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = FP64 matrix peak (half of the guide's 157.3 TF FP32 vector rate)
PROFILE_TAG = "r6"       # committed rocprofv3 PMC passes of this command: profiles/<tag>_pmc_{fetch,write}_size.csv


def bench_options(ba, max_iters):
    o = ba.default_options()
    o.max_num_iterations = int(max_iters)
    # run exactly max_iters LM iterations: tolerances off (stated in DESIGN.md)
    o.function_tolerance = 0.0
    o.gradient_tolerance = 0.0
    o.parameter_tolerance = 0.0
    o.use_inner_iterations = 0
    return o


def algorithmic_bytes_linearize(p):
    """DESIGN.md "algorithmic bytes": what one linearize + Schur launch group must move.
    reads : 24 B / observation (uv f64x2 + two int32 indices)
            48 B / camera + 56 B / intrinsics group + 32 B / point (parameters)
            48 B / camera + 24 B / point (Jacobi scaling)
    writes: 8 B x nnz(lower block-triangular S) + 48 B / camera (rhs)
            72 B / point (V^-1 packed 6 + g_p 3)"""
    nobs = p.obs_uv.shape[0]
    nc, npts, ng = p.cam_ext.shape[0], p.points.shape[0], p.intrinsics.shape[0]
    order = np.argsort(p.obs_pt, kind="stable")
    pt = p.obs_pt[order]; cam = p.obs_cam[order].astype(np.int64)
    L = np.bincount(pt, minlength=npts)
    start = np.cumsum(L) - L
    Lmax = int(L.max()) if npts else 0
    pairs = []
    for a in range(Lmax):
        for b in range(a + 1):
            sel = L > a
            ia = start[sel] + a; ib = start[sel] + b
            ca, cb = cam[ia], cam[ib]
            hi, lo = np.maximum(ca, cb), np.minimum(ca, cb)
            pairs.append(np.unique(hi * nc + lo))
    up = np.unique(np.concatenate(pairs)) if pairs else np.zeros(0, dtype=np.int64)
    ndiag = int(np.sum(up // nc == up % nc))
    nnz = 21 * ndiag + 36 * (len(up) - ndiag)
    reads = 24 * nobs + 48 * nc + 56 * ng + 32 * npts + 48 * nc + 24 * npts
    writes = 8 * nnz + 48 * nc + 72 * npts
    return reads + writes


def pmc_traffic_bytes(world):
    """HBM-side bytes per launch of the roofline kernel group (k_cam_prep + k_lin_schur + k_schur_sum), from the
    committed rocprofv3 PMC passes of this same command at N = 1 (profiles/<tag>_pmc_*.csv, collected past the
    Infinity Cache: 1.1 GB working set; FETCH_SIZE and WRITE_SIZE in separate passes, KB per dispatch; FETCH doubled
    per the gfx950 note of MI355X_MICROARCH.md -- the observation / parameter streams are 16 B / lane reads)."""
    if world != 1:
        return None
    import csv
    base = os.path.join(ROOT, "profiles")
    try:
        kb = {}
        for tag in ("fetch_size", "write_size"):
            with open(os.path.join(base, "%s_pmc_%s.csv" % (PROFILE_TAG, tag))) as f:
                for line in f:
                    row = line.rstrip("\n").rsplit(",", 2)      # template arguments carry commas: split from the right
                    if len(row) == 3 and row[0] != "kernel":
                        kb[(tag, row[0])] = float(row[2])
        # (the headline kernel only: "k_lin_schur<": the same trace also holds k_lin_schur_i of the with_intrinsics block)
        group = [k for (t, k) in kb if t == "fetch_size" and (k.startswith("k_lin_schur<") or k.startswith("k_schur_sum") or k.startswith("k_cam_prep"))]
        if not any(k.startswith("k_lin_schur<") for k in group):
            return None
        fetch = sum(2.0 * kb[("fetch_size", k)] for k in group)
        write = sum(kb.get(("write_size", k), 0.0) for k in group)
        return int(1024 * (fetch + write))
    except (OSError, KeyError, ValueError):
        return None


def counters_stale():
    """True when the kernel sources differ from the ones the committed counter passes were collected on
    (profiles/<tag>_collected_at.json: sha256 prefixes written by scripts/refresh_profiles.sh), None when unknown."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "profiles", "%s_collected_at.json" % PROFILE_TAG)) as f:
            ref = json.load(f)
        for fn, h in ref.items():
            with open(os.path.join(ROOT, fn), "rb") as g:
                if hashlib.sha256(g.read()).hexdigest()[:16] != h:
                    return True
        return False
    except (OSError, ValueError):
        return None


def mfma_util(levels, solve_s):
    """MFMA busy cycles of one K3 solve (committed counter pass: per-launch means x launches per solve) over the cycles
    the chip's 1024 SIMDs offer in the measured solve time at 2.1 GHz."""
    m = mfma_busy_cycles()
    if not m or solve_s <= 0 or levels <= 0:
        return None
    busy = levels * m["k_sp_potrf_trsm"] + max(0, levels - 1) * m["k_sp_update"]
    return {"busy_cycles_per_solve": busy, "frac": busy / (1024 * 2.1e9 * solve_s),
            "source": "profiles/%s_sq_counters.txt (SQ_VALU_MFMA_BUSY_CYCLES per launch) x launches per solve / (1024 SIMDs x 2.1 GHz x solve time)" % PROFILE_TAG}


def mfma_busy_cycles():
    """SQ_VALU_MFMA_BUSY_CYCLES per launch of the K3 kernels from the committed counter pass of this command
    (profiles/<tag>_sq_counters.txt, `scripts/pmc_kernel.sh "SQ_VALU_MFMA_BUSY_CYCLES ..."`), or None."""
    import ast
    out = {}
    try:
        with open(os.path.join(ROOT, "profiles", "%s_sq_counters.txt" % PROFILE_TAG)) as f:
            for line in f:
                for k in ("k_sp_potrf_trsm", "k_sp_update"):
                    if line.startswith(k + " {") and "SQ_VALU_MFMA_BUSY_CYCLES" in line:
                        out[k] = float(ast.literal_eval(line[len(k) + 1:].strip())["SQ_VALU_MFMA_BUSY_CYCLES"])
    except (OSError, ValueError, SyntaxError, KeyError):
        return None
    return out if len(out) == 2 else None


class BaRunner:
    """K LM iterations as ceil(K / ITERS_PER_SOLVE) solves, each from the device-resident perturbed start."""

    def __init__(self, ba, prob, pristine):
        self.ba = ba
        self.opts = bench_options(ba, ITERS_PER_SOLVE)
        self.h = ba.BaHandle(prob, self.opts)
        self.pristine = pristine

    def prepare(self):
        self.h.reset(self.pristine)
        self.h.snapshot()   # the perturbed initial state stays in HBM: the timed region has no host round trip for inputs

    def run(self, k, from_host=False):
        done = 0
        acc = {"lin_kernel": 0.0, "launches": 0, "lin": 0.0, "solve": 0.0, "backsub": 0.0}
        while done < k:
            m = min(ITERS_PER_SOLVE, k - done)
            self.opts.max_num_iterations = int(m)
            self.h.set_options(self.opts)
            if from_host:
                self.h.reset(self.pristine)
            else:
                self.h.restore()
            s, _ = self.h.run(trace_capacity=1)
            if s.num_iterations != m:
                raise RuntimeError(f"solve stopped after {s.num_iterations} of {m} iterations (term {s.termination_type})")
            done += m
            acc["lin_kernel"] += s.time_kernel_linearize; acc["launches"] += s.num_linearize_launches
            acc["lin"] += s.time_linearize; acc["solve"] += s.time_solve_reduced; acc["backsub"] += s.time_backsub
        return acc

    def phase_timed(self, k):
        os.environ["THEIA_HIP_PHASE_TIMING"] = "1"
        try:
            return self.run(k)
        finally:
            del os.environ["THEIA_HIP_PHASE_TIMING"]


def cpu_ba_baseline(pristine, nobs_total, threads, n_it, what):
    """The oracle ("port", oracle/ba_oracle.cpp: Jet autodiff + Schur elimination + envelope Cholesky) on the host."""
    from tests import oracle_lib as ol
    oo = ol.default_options()
    oo.max_num_iterations = n_it
    oo.function_tolerance = 0.0; oo.gradient_tolerance = 0.0; oo.parameter_tolerance = 0.0
    oo.use_inner_iterations = 0
    pc = pristine.copy()
    prev = ol.set_threads(threads)
    try:
        t0 = time.perf_counter()
        so, _ = ol.solve(pc, oo, trace_capacity=1)
        dt = time.perf_counter() - t0
    finally:
        ol.set_threads(prev)
    it_s = so.num_iterations / max(so.solve_time_in_seconds, 1e-9)   # LM loop only (setup excluded, as on the GPU side)
    return {"value": nobs_total * it_s, "unit": "residuals/s", "lm_iterations_per_sec": it_s, "cores": threads, "kind": "port",
            "sample": f"{so.num_iterations} LM iterations of {what} (oracle/ba_oracle.cpp, FP64, Jet autodiff + Schur + envelope "
                      f"Cholesky, OpenMP over observations / camera chunks), {dt:.1f} s wall incl. {so.setup_time_in_seconds:.1f} s setup",
            "phase_s": {"linearize_schur": so.time_linearize, "reduced_solve": so.time_solve_reduced, "backsub_trial_cost": so.time_backsub}}


def ransac_block(cpu_baseline, host_cores, rank=0, world=1):
    """BASELINE.json configs[4] on one GPU: 10 000 pairs x 2000 correspondences x 4096 hypotheses, five-point relative
    pose, SQPnP and DLS absolute pose.  FLOP/s: SURVEY.md 8d counts (score: 85 FLOP per model and
    correspondence; fit: per-solve counts of the restated solvers)."""
    from pytheiasfm_amd import ransac, synth
    PAIRS, CORR, HYPS, CHUNK = 10000, 2000, 4096, 1000
    out = {"workload": f"synth_ransac_v1 C5: {PAIRS} pairs x {CORR} correspondences x {HYPS} hypotheses (min = max iterations), InlierSupport",
           "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS}
    # (name, estimator, data kind, threshold, FLOP per minimal solve, FLOP per model x correspondence, chunks of 1000 pairs run)
    # DLS: the reference's dense route -- 2/3 93^3 + 2 93^2 27 = 1.0 MFLOP for the partial-pivot LU of the Macaulay block with its 27
    # right-hand sides (dls_pnp.cc:143-146) + ~0.35 MFLOP for the 27 x 27 eigen-decomposition per solve
    # (all ten chunks: 40.96 M hypotheses, about half a minute)
    legs = (("five_point_relative_pose", ransac.EST_RELATIVE_POSE, "relative", (2.0 / 1000.0) ** 2, 2.5e4, 85.0, PAIRS // CHUNK),
            ("sqpnp_absolute_pose", ransac.EST_ABS_SQPNP, "absolute", (4.0 / 1000.0) ** 2, 3.0e4, 30.0, PAIRS // CHUNK),
            ("dls_absolute_pose", ransac.EST_ABS_DLS, "absolute", (4.0 / 1000.0) ** 2, 1.35e6, 30.0, PAIRS // CHUNK),
            # UPnP (EstimateRigidTransformation2D3D, central overload): one chunk; 2/3 141^3 + 2 141^2 8 = 2.2 MFLOP for the reference's
            # Gauss-Jordan of the 141 x 149 template per solve (build_upnp_action_matrix_using_symmetry.cc:2487)
            ("upnp_rigid_transformation", ransac.EST_RIGID_TRANSFORMATION_2D3D, "rigid", (4.0 / 1000.0) ** 2, 2.2e6, 40.0, 1),
            # P4Pfr (EstimateRadialDistUncalibratedAbsolutePose): one chunk, the correspondences seen by a camera of focal length 1000 with
            # division-model distortion -1e-7 (pixels); per solve ~47 kFLOP for the full-pivot LU of the 37 x 40 block with its seven
            # right-hand sides, ~55 kFLOP for the 13 x 13 eigen-decomposition, ~15 kFLOP for the template and the normalisation
            ("p4pfr_radial_dist_absolute_pose", ransac.EST_RADIAL_DIST_UNCALIBRATED_ABSOLUTE_POSE, "radial", 4.0 ** 2, 1.2e5, 35.0, 1))
    p4pfr_ep = np.array([2000.0, 100.0, -1e-5, -1e-9, 0.0])   # RadialDistUncalibratedAbsolutePoseMetaData of the reference's test + "not the first call"
    for name, est, kind, thresh, fit_flop, score_flop, nchunks in legs:
        p = ransac.RansacParameters(); p.error_thresh = thresh; p.min_iterations = HYPS; p.max_iterations = HYPS; p.seed = 1
        tot = {"hyp": 0, "models": 0, "wall": 0.0, "fit": 0.0, "score": 0.0, "kern": 0.0}
        first = None
        for c in range(rank, nchunks, world):   # (whole chunks of pairs per rank: same round-robin deal, coarser grain)
            data, offsets, TRUTH = synth.synth_ransac_v1(CHUNK, CORR, "absolute" if kind in ("rigid", "radial") else kind, seed=0x5AC50005 + 977 * c)
            if kind == "rigid":
                data = ransac.central_correspondence_rows(data)   # [u v X Y Z] as seen by identity pinhole cameras (26 doubles per datum)
            if kind == "radial":
                data = ransac.radial_dist_correspondence_rows(ransac.shift_world_along_optical_axis(data, offsets, TRUTH["R"], 2.0), 1000.0, -1e-7)
            ep = p4pfr_ep if kind == "radial" else None
            if first is None:
                ransac.estimate_batch(est, data[: offsets[8]], offsets[:9], p, ep)  # warm-up
                first = (data, offsets)
            p.seed = 1 + c * CHUNK
            t0 = time.perf_counter()
            res = ransac.estimate_batch(est, data, offsets, p, ep)
            tot["wall"] += time.perf_counter() - t0
            tot["hyp"] += int(res["hypotheses_evaluated"]); tot["models"] += int(res["models_scored"])
            tot["fit"] += res["time_fit_seconds"]; tot["score"] += res["time_score_seconds"]; tot["kern"] += res["time_fit_score_seconds"]
        leg = {"hypotheses_per_sec": tot["hyp"] / max(tot["wall"], 1e-12),   # (a rank that got no chunk of a short leg reports zeros)
               "hypotheses_per_sec_kernels_only": tot["hyp"] / max(tot["kern"], 1e-12),
               "hypotheses": tot["hyp"], "models_scored": tot["models"], "wall_s": tot["wall"],
               "kernel_s": {"fit": tot["fit"], "score": tot["score"]},
               "roofline_fit": {"bound": "fp64-vector", "achieved": tot["hyp"] * fit_flop / max(tot["fit"], 1e-12) / 1e12,
                                "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "flop_per_solve": fit_flop},
               "roofline_score": {"bound": "fp64-vector", "achieved": tot["models"] * CORR * score_flop / max(tot["score"], 1e-12) / 1e12,
                                  "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "flop_per_model_correspondence": score_flop},
               "note": "wall time includes the PCIe upload of the correspondences, host sample generation and host replay"}
        if nchunks != PAIRS // CHUNK:
            leg["sample"] = f"{nchunks * CHUNK} of the {PAIRS} pairs (x {CORR} correspondences x {HYPS} hypotheses)"
        for r in ("roofline_fit", "roofline_score"):
            leg[r]["frac"] = leg[r]["achieved"] / FP64_VECTOR_PEAK_TFLOPS
        if cpu_baseline:
            from concurrent.futures import ThreadPoolExecutor
            from tests import oracle_lib as ol
            data, offsets = first
            hy = 1024 if est != ransac.EST_RIGID_TRANSFORMATION_2D3D else 256   # (the UPnP oracle: a dense 141 x 149 elimination per hypothesis)

            ep = p4pfr_ep if kind == "radial" else None
            if ep is not None:
                ol.set_estimator_params(ep)    # (process-wide in the oracle: this leg is the only one that reads them)

            def one(i):
                pc = p.to_c(); pc.min_iterations = hy; pc.max_iterations = hy; pc.seed = 1 + i
                return ol.ransac_estimate(est, data[offsets[i]:offsets[i + 1]], pc)
            t0 = time.perf_counter()
            for i in range(8):
                one(i)
            dt1 = time.perf_counter() - t0
            nall = min(CHUNK, max(8, 4 * host_cores))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=host_cores) as ex:   # ctypes releases the GIL: one oracle call per thread
                ora = list(ex.map(one, range(nall)))
            dta = time.perf_counter() - t0
            # inlier-set equality count of the GPU path against the CPU path on the same pairs / seeds
            # (the CPU-baseline sample: the first `nall` pairs of the leg at `hy` iterations)
            pe = ransac.RansacParameters(); pe.error_thresh = thresh; pe.min_iterations = hy; pe.max_iterations = hy; pe.seed = 1
            rg = ransac.estimate_batch(est, data[: offsets[nall]], offsets[: nall + 1], pe, ep)
            eq = [bool(np.array_equal(ora[i]["inlier_mask"], rg["inlier_mask"][offsets[i]:offsets[i + 1]])) for i in range(nall)]
            sym = [int((ora[i]["inlier_mask"] != rg["inlier_mask"][offsets[i]:offsets[i + 1]]).sum()) for i in range(nall)]
            leg["inlier_set_equality"] = {"pairs_equal": int(sum(eq)), "pairs_compared": nall, "max_symmetric_difference": int(max(sym)),
                                          "iterations_equal": int(sum(int(ora[i]["num_iterations"]) == int(rg["num_iterations"][i]) for i in range(nall))),
                                          "sample": f"first {nall} pairs x {hy} hypotheses x {CORR} correspondences, per-pair seeds 1 + i, "
                                                    "GPU path against oracle/ransac_oracle.cpp"}
            leg["cpu_baseline"] = {"value": nall * hy / dta, "unit": "hypotheses/s", "cores": host_cores, "kind": "port",
                                   "sample": f"{nall} pairs x {hy} hypotheses x {CORR} correspondences, one pair per host thread "
                                             f"(oracle/ransac_oracle.cpp, the reference's sequential loop per pair), {dta:.1f} s"}
            leg["cpu_baseline_single_thread"] = {"value": 8 * hy / dt1, "unit": "hypotheses/s", "cores": 1, "kind": "port",
                                                 "sample": f"8 pairs x {hy} hypotheses, {dt1:.1f} s"}
        out[name] = leg
    return out


def relative_position_block(cpu_baseline):
    """N x OptimizeRelativePositionWithKnownRotation (one per view-graph edge in the reference's global pipeline) as one launch:
    4000 pairs x 400 correspondences, one IRLS per wavefront (csrc/relpos_irls.hip)."""
    from pytheiasfm_amd import ba, synth
    NP, NC = 4000, 400
    corr, offsets, rot, truth = synth.synth_relpos_v1(NP, NC)
    ba.optimize_relative_position_batch(offsets[:65], corr[:64 * NC], rot[:64])
    t0 = time.perf_counter(); pos, it = ba.optimize_relative_position_batch(offsets, corr, rot); dt = time.perf_counter() - t0
    out = {"workload": f"{NP} pairs x {NC} correspondences, noise 5e-4, IRLS to the reference's stopping rule (<= 100 iterations)",
           "pairs_per_sec": NP / dt, "wall_s": dt, "mean_iterations": float(np.mean(it)),
           "direction_recovered": float(np.mean(np.abs(np.sum(pos * truth, axis=1)) > 0.999)),
           "note": "wall time of the C entry point from host arrays (upload + kernel + download)"}
    if cpu_baseline:
        from tests import oracle_lib as ol
        ns = 200
        t0 = time.perf_counter()
        same = 0
        for k in range(ns):
            opos, oit = ol.optimize_relative_position(corr[offsets[k]:offsets[k + 1]], rot[k, :3], rot[k, 3:], order=1)
            same += int(np.array_equal(opos, pos[k]) and oit == it[k])
        dc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": ns / dc, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"the first {ns} pairs through oracle/ransac_oracle.cpp (one thread), {dc:.1f} s"}
        out["bit_identical_to_wave_order_oracle"] = {"pairs_equal": same, "pairs_compared": ns}
    return out


def self_launch(n):
    """Re-runs this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on a free local
    port; returns the launcher's exit code.  Rank 0 prints the one JSON line, the other ranks print nothing."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ransac", action="store_true")
    ap.add_argument("--no-c2", action="store_true")
    ap.add_argument("--one-gpu-dry-run", action="store_true",
                    help="with --gpus N under torch.distributed.run: all N ranks share cuda:0 and the collective goes through the host "
                         "(gloo); everything else is the multi-GPU code path -- a rehearsal of the SCALE run on a one-GPU box, not a measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = bool(args.one_gpu_dry_run) and world > 1
    if dry:
        local_rank = 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same
        # torch.distributed.run line the driver uses) and pass rank 0's single JSON line through
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    import torch
    import torch.distributed as dist
    from pytheiasfm_amd import _capi as capi, ba, synth, distributed as tdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    capi.check(capi.lib().theia_hip_init(local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if dry else "cuda"   # where the few scalar reductions of this script live

    # ---- headline workload (synthetic, deterministic): C4, strong-sharded over the ranks
    full = synth.ba_config("C4")
    nobs_total = full.obs_uv.shape[0]
    nviews, ntracks = full.cam_ext.shape[0], full.points.shape[0]
    prob = synth.shard_tracks(full, rank, world)[0] if world > 1 else full
    pristine = prob.copy()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    R = BaRunner(ba, prob, pristine)
    comm = None
    if world > 1:
        # the library issues ncclAllReduce itself (theia_hip_ba_set_rccl); THEIA_HIP_BENCH_TORCH_ALLREDUCE=1 keeps the
        # torch.distributed callback of round 1 for comparison
        if dry:
            R.h.set_allreduce(tdist.make_host_staged_allreduce(0))
            R.h.set_shard(rank, world)
        elif os.environ.get("THEIA_HIP_BENCH_TORCH_ALLREDUCE"):
            R.h.set_allreduce(tdist.make_torch_allreduce(local_rank))
            R.h.set_shard(rank, world)
        else:
            comm = tdist.NativeRccl(rank, world)
            comm.attach(R.h)
    R.prepare()
    info = R.h.plan_info()
    # initialisation, not part of the W warm-up steps: a fresh box idles at its lowest clock level -- run the workload
    # for a few hundred ms first so that the timed region sees settled clocks
    R.run(160)
    if args.warmup > 0:
        R.run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    R.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # Kernel-group durations for the rooflines: the timed region above has no events inside; the same workload is run
    # once more with HIP events around the kernel groups on the library's stream (THEIA_HIP_PHASE_TIMING=1).
    acc = R.phase_timed(max(ITERS_PER_SOLVE, min(args.steps, 40)))
    barrier()
    # for the record: the same solves with the parameters re-uploaded from host memory per solve
    tp0 = time.perf_counter()
    npcie = max(ITERS_PER_SOLVE, min(args.steps, 40))
    R.run(npcie, from_host=True)
    barrier()
    pcie_it_per_s = npcie / (time.perf_counter() - tp0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        it_per_s = args.steps / elapsed
        abytes = algorithmic_bytes_linearize(full) if world == 1 else None
        nl = max(1, acc["launches"])
        avg_lin = acc["lin_kernel"] / nl
        achieved = abytes / avg_lin / 1e9 if (abytes and avg_lin > 0) else 0.0
        Ltrk = np.bincount(prob.obs_pt)      # this rank's tracks (all of them at N = 1)
        afma = float(108.0 * np.sum(Ltrk * (Ltrk + 1) // 2) + 300.0 * prob.obs_uv.shape[0])
        k3_s = acc["solve"] / nl
        out = {
            "metric": "BA residuals/sec (LM-iterations/sec x observations); RANSAC hypotheses/sec alongside",
            "value": nobs_total * it_per_s, "unit": "residuals/s",
            "lm_iterations_per_sec": it_per_s,
            "lm_iterations_per_sec_pcie_inclusive": pcie_it_per_s,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synth_ba_v1 C4: {nviews} views / {ntracks} tracks / {nobs_total} observations, mixed pinhole + "
                                   f"double-sphere (8 intrinsics groups), TRIVIAL loss, intrinsics NONE, homogeneous-manifold points, "
                                   f"{ITERS_PER_SOLVE} LM iterations per solve, use_inner_iterations = false on the GPU and the CPU leg "
                                   f"(the trust-region step of north_star alone; the reference's default = true is measured in "
                                   f"\"with_inner_iterations\")"
                                   + ("" if world == 1 else f", tracks sharded over {world} ranks, RCCL all-reduce of the reduced camera system"),
                       "views": nviews, "tracks": ntracks, "observations": nobs_total,
                       "baseline_config": "BASELINE.json configs[3] (north_star target configuration) on %d GPU%s" % (world, "" if world == 1 else "s"),
                       "reduced_system_n": info["n"], "k3_levels": info["k3_levels"], "fused_kernel_runs": info["fused_runs"],
                       "slow_path_tracks": info["slow_path_tracks"],
                       # the width of the run as the collective library itself reports it (ncclCommCount through the
                       # library's own communicator; the one-GPU rehearsal has no RCCL communicator: gloo's group size)
                       "n_ranks_seen_by_rccl": (comm.count() if comm is not None else (None if world > 1 else 1)),
                       "n_ranks_in_process_group": (dist.get_world_size() if world > 1 else 1)},
            "roofline": {"kernel": "linearize + Schur assembly launch group (k_lin_schur + k_schur_sum; k_cam_prep only on the first launch of a solve)",
                         "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         # counters collected on other kernel sources than the tree's are not this kernel's traffic: refused (null)
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic_bytes(world) if counters_stale() is False else None,
                         "traffic_source": "profiles/%s_pmc_{fetch,write}_size.csv (separate rocprofv3 --pmc passes of this command, committed)" % PROFILE_TAG,
                         "traffic_source_stale": counters_stale(),   # True: the kernels changed since those passes
                         "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": 1e3 * avg_lin,
                         "launches": acc["launches"],
                         "timing": "HIP events around the kernel group on the library's stream, instrumented pass of the same workload right after the timed region",
                         # the governing roofline of this arithmetic: algorithmic FP64 FMAs of the launch group
                         # -- 108 per pair of observations of a track (diagonal included) + ~300 per linearised observation --
                         # over the launch time, against the FP64 vector peak (39.3 T FMA/s = 78.6 TFLOP/s)
                         "fp64_vector_frac": (afma / avg_lin / (0.5e12 * FP64_VECTOR_PEAK_TFLOPS)) if avg_lin > 0 else 0.0,
                         "algorithmic_fma_per_launch": afma,
                         "note": "FP64 latency / issue bound in practice (DESIGN.md 3.4): the pair-product FMAs + the linearisation per launch"},
            "roofline_k3": {"kernel": "reduced-camera solve (tile-sparse level-scheduled Cholesky, FP64 MFMA trsm / update)",
                            "bound": "mfma", "achieved": info["k3_flops"] / k3_s / 1e12 if k3_s > 0 else 0.0,
                            "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": (info["k3_flops"] / k3_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if k3_s > 0 else 0.0,
                            "flops_per_solve": info["k3_flops"], "avg_solve_ms": 1e3 * k3_s,
                            "mfma_utilisation": mfma_util(info["k3_levels"], k3_s)},
            "phase_ms_per_iteration": {"linearize_schur": 1e3 * acc["lin"] / nl, "reduced_solve": 1e3 * acc["solve"] / nl,
                                       "backsub_trial_cost": 1e3 * acc["backsub"] / nl},
        }
    R.h.close()
    if comm is not None:
        comm.close()

    host_cores = os.cpu_count() or 1
    if rank == 0 and world == 1:
        # the reference's DEFAULT options run Ceres' inner iterations after every accepted step (bundle_adjustment.h:144):
        # the same problem with use_inner_iterations = true, for the record (not the headline: the inner sweeps are per-block
        # LM solves outside north_star's hot path)
        oi = bench_options(ba, ITERS_PER_SOLVE); oi.use_inner_iterations = 1
        hi = ba.BaHandle(pristine.copy(), oi)
        hi.reset(pristine); hi.snapshot()
        hi.restore(); hi.run(trace_capacity=1)
        torch.cuda.synchronize(); ti0 = time.perf_counter()
        nrep = 2
        for _ in range(nrep):
            hi.restore(); si, _ = hi.run(trace_capacity=1)
        torch.cuda.synchronize(); ti = time.perf_counter() - ti0
        out["with_inner_iterations"] = {"lm_iterations_per_sec": nrep * si.num_iterations / ti,
                                        "ms_per_step": 1e3 * ti / (nrep * si.num_iterations), "iterations_per_solve": si.num_iterations,
                                        "note": "use_inner_iterations = true (reference default): coordinate descent over cameras then points after each accepted step"}
        del hi
        # the pipelines' default intrinsics subset (FOCAL_LENGTH | RADIAL_DISTORTION, reconstruction_estimator_options.h:281-283)
        # on the same problem: compound [extrinsics | intrinsics] camera blocks in the fused kernel of ba_fused_intr.hip
        ok = bench_options(ba, ITERS_PER_SOLVE); ok.intrinsics_to_optimize = 0x01 | 0x10
        tk0 = time.perf_counter(); hk = ba.BaHandle(pristine.copy(), ok); tk1 = time.perf_counter()
        hk.reset(pristine); hk.snapshot()
        hk.restore(); hk.run(trace_capacity=1)
        torch.cuda.synchronize(); tk2 = time.perf_counter()
        for _ in range(nrep):
            hk.restore(); sk, _ = hk.run(trace_capacity=1)
        torch.cuda.synchronize(); tk = time.perf_counter() - tk2
        out["with_intrinsics"] = {"lm_iterations_per_sec": nrep * sk.num_iterations / tk,
                                  "ms_per_step": 1e3 * tk / (nrep * sk.num_iterations), "iterations_per_solve": sk.num_iterations,
                                  "handle_creation_ms": 1e3 * (tk1 - tk0),
                                  "note": "intrinsics_to_optimize = FOCAL_LENGTH | RADIAL_DISTORTION over the problem's 8 shared groups "
                                          "(k_lin_schur_i + k_sum_items: fused, records in LDS, DESIGN.md 3.5); not the headline configuration"}
        del hk
        # OptimizeIntrinsicsType::ALL on the same problem: seven free parameters per group on the pinhole and double-sphere models
        # (13-wide compound blocks, round 5; until then the first-generation gather kernels)
        oa = bench_options(ba, ITERS_PER_SOLVE); oa.intrinsics_to_optimize = 0x3f
        ta0 = time.perf_counter(); ha = ba.BaHandle(pristine.copy(), oa); ta1 = time.perf_counter()
        ha.reset(pristine); ha.snapshot()
        ha.restore(); ha.run(trace_capacity=1)
        torch.cuda.synchronize(); ta2 = time.perf_counter()
        for _ in range(nrep):
            ha.restore(); sa, _ = ha.run(trace_capacity=1)
        torch.cuda.synchronize(); ta = time.perf_counter() - ta2
        out["with_all_intrinsics"] = {"lm_iterations_per_sec": nrep * sa.num_iterations / ta,
                                      "ms_per_step": 1e3 * ta / (nrep * sa.num_iterations), "iterations_per_solve": sa.num_iterations,
                                      "handle_creation_ms": 1e3 * (ta1 - ta0),
                                      "note": "intrinsics_to_optimize = ALL (0x3f): seven free parameters per group, 13-wide compound blocks "
                                              "(DESIGN.md 3.5); not the headline configuration"}
        del ha
        # the pipelines' REAL default: both of the above at once (reconstruction_estimator_options.h:281-283 frees FOCAL_LENGTH |
        # RADIAL_DISTORTION, bundle_adjustment.h:144 leaves use_inner_iterations on)
        ob = bench_options(ba, ITERS_PER_SOLVE); ob.intrinsics_to_optimize = 0x01 | 0x10; ob.use_inner_iterations = 1
        hb = ba.BaHandle(pristine.copy(), ob)
        hb.reset(pristine); hb.snapshot()
        hb.restore(); hb.run(trace_capacity=1)
        torch.cuda.synchronize(); tb0 = time.perf_counter()
        for _ in range(nrep):
            hb.restore(); sb, _ = hb.run(trace_capacity=1)
        torch.cuda.synchronize(); tb = time.perf_counter() - tb0
        out["pipeline_default"] = {"lm_iterations_per_sec": nrep * sb.num_iterations / tb,
                                   "ms_per_step": 1e3 * tb / (nrep * sb.num_iterations), "iterations_per_solve": sb.num_iterations,
                                   "note": "intrinsics_to_optimize = FOCAL_LENGTH | RADIAL_DISTORTION AND use_inner_iterations = true: what "
                                           "BundleAdjustReconstruction runs with inside the reference's pipelines; not the headline configuration"}
        del hb
    if rank == 0 and world == 1:
        # what one theia_hip_ba_solve call costs end to end (BundleAdjustReconstruction through the boundary: host-side
        # structure analysis + uploads + 25 LM iterations + download), for the record next to the steady-state rate
        o1 = ba.default_options(); o1.max_num_iterations = 25; o1.use_inner_iterations = 0
        o1.function_tolerance = o1.gradient_tolerance = o1.parameter_tolerance = 0.0
        def best_call(opts):   # best of three calls (one call is one sample of a 25 - 40 ms interval on a shared host)
            best, sm = float("inf"), None
            for _ in range(3):
                qq = pristine.copy()
                ta = time.perf_counter(); sm, _ = ba.solve(qq, opts, trace_capacity=1); best = min(best, time.perf_counter() - ta)
            return best, sm
        t1best, s1 = best_call(o1); t10, t11 = 0.0, t1best
        # ... the same call with the reference's default tolerances (it stops when Ceres would), and the handle creation alone
        # (second of two: the first one of a process also fills the library's device / pinned-block caches)
        o2 = ba.default_options(); o2.max_num_iterations = 25; o2.use_inner_iterations = 0
        t2best, s2 = best_call(o2); t20, t21 = 0.0, t2best
        tc = []
        for _ in range(2):
            pc = pristine.copy()
            t30 = time.perf_counter(); h3 = ba.BaHandle(pc, o2); tc.append(time.perf_counter() - t30); h3.close()
        out["one_shot_call"] = {"workload": "theia_hip_ba_solve on the same problem from host arrays, 25 LM iterations (best of three calls)",
                                "total_ms": 1e3 * (t11 - t10), "iterations": int(s1.num_iterations),
                                "default_tolerances_total_ms": 1e3 * (t21 - t20), "default_tolerances_iterations": int(s2.num_iterations),
                                "handle_creation_ms": 1e3 * tc[-1],
                                "note": "handle creation (host-side plan of 3.0 M observations) ~16 ms of it: see DESIGN.md 3.4"}
    if rank == 0 and world == 1 and not args.no_c2:
        # secondary block: BASELINE.json configs[1] (the round-1 headline), same measurement
        c2 = synth.ba_config("C2")
        R2 = BaRunner(ba, c2, c2.copy())
        R2.prepare()
        R2.run(400)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); R2.run(200); torch.cuda.synchronize(); e2 = time.perf_counter() - t0
        a2 = R2.phase_timed(48)
        ab2 = algorithmic_bytes_linearize(c2)
        l2 = a2["lin_kernel"] / max(1, a2["launches"])
        out["c2"] = {"workload": f"synth_ba_v1 C2: 200 views / 50000 tracks / {c2.obs_uv.shape[0]} observations, pinhole (BASELINE.json configs[1])",
                     "lm_iterations_per_sec": 200 / e2, "residuals_per_sec": c2.obs_uv.shape[0] * 200 / e2, "ms_per_step": 1e3 * e2 / 200,
                     "roofline": {"bound": "hbm", "achieved": ab2 / l2 / 1e9 if l2 > 0 else 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": (ab2 / l2 / 1e9 / HBM_PEAK_GBPS) if l2 > 0 else 0.0, "traffic": None,
                                  "algorithmic_bytes_per_launch": ab2, "avg_launch_ms": 1e3 * l2},
                     "phase_ms_per_iteration": {"linearize_schur": 1e3 * a2["lin"] / max(1, a2["launches"]),
                                                "reduced_solve": 1e3 * a2["solve"] / max(1, a2["launches"]),
                                                "backsub_trial_cost": 1e3 * a2["backsub"] / max(1, a2["launches"])}}
        R2.h.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        what = f"the same {nviews}-view / {ntracks}-track problem"
        # all-cores leg: the thread count that runs fastest on this host (the oracle's OpenMP loops stop scaling where
        # the dense reduced matrix and the Jacobian arrays saturate the memory system), each tried on 3 LM iterations
        tried = {}
        for th in sorted({host_cores, max(1, host_cores // 2), max(1, host_cores // 4), min(host_cores, 32)}, reverse=True):
            tried[th] = cpu_ba_baseline(pristine, nobs_total, th, 3, what)
        best = max(tried, key=lambda k: tried[k]["value"])
        out["cpu_baseline"] = tried[best]
        out["cpu_baseline"]["host_cores_available"] = host_cores
        out["cpu_baseline"]["threads_tried_iterations_per_sec"] = {str(k): v["lm_iterations_per_sec"] for k, v in tried.items()}
        out["cpu_baseline_single_thread"] = cpu_ba_baseline(pristine, nobs_total, 1, 1, what)
        # (no GPU / CPU ratio is quoted: the port stops scaling at a fraction of the host's cores -- `cores` is the thread count
        # that ran fastest, `threads_tried_iterations_per_sec` the others -- and the roofline fractions above, not a ratio
        # against it, say how good the kernels are)
        out["cpu_baseline"]["phase_s_note"] = ("parallel Jet evaluation and two-phase camera-chunk Schur elimination (phase_s.linearize_schur: the part that "
                                                "stops scaling -- at most ncv / (2 x track span) chunks run concurrently, and the Jet buffers saturate the "
                                                "memory system); the envelope Cholesky of the reduced system is one thread")

    if world == 1 and not args.no_ransac:
        out["ransac"] = ransac_block(not args.no_cpu_baseline, host_cores)
        out["relative_position_irls"] = relative_position_block(not args.no_cpu_baseline)
    elif not args.no_ransac:
        # configs[4] on N GPUs: the 10 000 pairs are dealt round robin over the ranks, no collective on the data path
        rb = ransac_block(False, host_cores, rank=rank, world=world)
        names = ("five_point_relative_pose", "sqpnp_absolute_pose", "dls_absolute_pose")
        t = torch.tensor([[rb[k]["wall_s"], float(rb[k]["hypotheses"])] for k in names],
                         dtype=torch.float64, device=red_dev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        if rank == 0:
            out["ransac"] = {"workload": rb["workload"] + f", pairs sharded round robin over {world} ranks"}
            for k, nm in enumerate(names):
                out["ransac"][nm] = {"hypotheses_per_sec": float(tsum[k, 1] / tmax[k, 0]), "hypotheses": float(tsum[k, 1]),
                                     "wall_s_max_over_ranks": float(tmax[k, 0])}

    if rank == 0 and dry:
        out["dry_run_one_gpu"] = {"ranks": world, "collective": "gloo through the host (RCCL refuses two ranks on one device)",
                                  "note": "a rehearsal of the multi-GPU code path of this script on one device: NOT a scaling measurement"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
