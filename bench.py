#!/usr/bin/env python
"""bench.py -- BA LM-iterations/s + residuals/s (and RANSAC hypotheses/s) on
N x MI355X, with the kernel roofline and the CPU baseline in the same JSON line.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE Levenberg-Marquardt iteration of the BA hot path (Jacobian
evaluation + Schur elimination + reduced-camera solve + back-substitution +
trial-cost evaluation) on a synthetic reconstruction resident in HBM.
N = 1 runs BASELINE.json configs[1] (200 views / 50k tracks, pinhole).  For
N > 1 the track set grows with N (50k tracks per GPU, same 200 views): tracks
are sharded over ranks and the reduced camera system is all-reduced with RCCL
every iteration -> weak scaling; `value` = residual blocks linearised per
second over all ranks.
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

VIEWS = 200
TRACKS_PER_GPU = 50000
SEED = 0xBA5E0002
ITERS_PER_SOLVE = 5  # the LM iterations the default tolerances run on this scene; more would hit exact convergence
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


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
    """DESIGN.md "algorithmic bytes": what one linearize+Schur launch must move.
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
            pairs.append(hi * nc + lo)
    up = np.unique(np.concatenate(pairs)) if pairs else np.zeros(0, dtype=np.int64)
    ndiag = int(np.sum(up // nc == up % nc))
    nnz = 21 * ndiag + 36 * (len(up) - ndiag)
    reads = 24 * nobs + 48 * nc + 56 * ng + 32 * npts + 48 * nc + 24 * npts
    writes = 8 * nnz + 48 * nc + 72 * npts
    return reads + writes


def pmc_traffic_bytes(world):
    """HBM-side bytes per launch of the roofline kernel group, from the committed
    rocprofv3 PMC passes of this same command (profiles/r1z_pmc_*.csv; FETCH_SIZE
    and WRITE_SIZE collected in separate passes, KB; the 16-B/lane record gathers
    of k_schur doubled per the gfx950 FETCH_SIZE note of MI355X_MICROARCH.md).
    None when the files are absent or the run is not the profiled N=1 workload."""
    if world != 1:
        return None
    import csv
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        kb = {}
        for tag in ("fetch_size", "write_size"):
            with open(os.path.join(base, "r1z_pmc_%s.csv" % tag)) as f:
                for row in csv.reader(f):
                    if row and row[0] != "kernel":
                        kb[(tag, row[0])] = float(row[2])
        fetch = 2.0 * kb[("fetch_size", "k_schur<3>")] + kb[("fetch_size", "k_lin_obs<3>")]
        write = sum(kb[("write_size", k)] for k in ("k_lin_obs<3>", "k_schur<3>"))
        return int(1024 * (fetch + write))
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ransac", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    import torch.distributed as dist
    from pytheiasfm_amd import _capi as capi, ba, synth, distributed as tdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    capi.check(capi.lib().theia_hip_init(local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- workload (synthetic, deterministic)
    full = synth.synth_ba_v1(VIEWS, TRACKS_PER_GPU * world, seed=SEED)
    nobs_total = full.obs_uv.shape[0]
    if world > 1:
        prob, _ = synth.shard_tracks(full, rank, world)
    else:
        prob = full
    pristine = prob.copy()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    opts = bench_options(ba, ITERS_PER_SOLVE)
    h = ba.BaHandle(prob, opts)
    if world > 1:
        h.set_allreduce(tdist.make_torch_allreduce(local_rank))
        h.set_shard(rank, world)

    def set_max_iters(m):
        opts.max_num_iterations = int(m)
        h.set_options(opts)

    h.reset(pristine)
    h.snapshot()   # the perturbed initial state stays in HBM: the timed region has no host round trip for inputs

    def run_iterations(k, from_host=False):
        """Exactly k LM iterations as ceil(k / ITERS_PER_SOLVE) solves, each from
        the perturbed initial state (device-resident snapshot; from_host: re-uploaded over PCIe)."""
        done = 0
        acc = {"lin_kernel": 0.0, "launches": 0, "lin": 0.0, "solve": 0.0, "backsub": 0.0}
        while done < k:
            m = min(ITERS_PER_SOLVE, k - done)
            set_max_iters(m)
            if from_host:
                h.reset(pristine)
            else:
                h.restore()
            s, _ = h.run(trace_capacity=1)
            if s.num_iterations != m:
                raise RuntimeError(f"solve stopped after {s.num_iterations} of {m} iterations (term {s.termination_type})")
            done += m
            acc["lin_kernel"] += s.time_kernel_linearize; acc["launches"] += s.num_linearize_launches
            acc["lin"] += s.time_linearize; acc["solve"] += s.time_solve_reduced; acc["backsub"] += s.time_backsub
        return acc

    # initialisation, not part of the W warm-up steps: a fresh box idles at its lowest clock level and W = 10
    # iterations last 5 ms -- run the workload for a few hundred ms first so the timed region sees settled clocks
    run_iterations(400)
    if args.warmup > 0:
        run_iterations(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_iterations(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # Kernel-group durations for the roofline: the timed region above has no events inside; the same
    # workload is run once more with HIP events around the kernel groups on the library's stream
    # (THEIA_HIP_PHASE_TIMING=1).
    os.environ["THEIA_HIP_PHASE_TIMING"] = "1"
    acc = run_iterations(min(args.steps, 50))
    del os.environ["THEIA_HIP_PHASE_TIMING"]
    barrier()
    # for the record: the same solves with the parameters re-uploaded from host memory per solve
    tp0 = time.perf_counter()
    npcie = min(args.steps, 50)
    run_iterations(npcie, from_host=True)
    barrier()
    pcie_it_per_s = npcie / (time.perf_counter() - tp0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        it_per_s = args.steps / elapsed
        res_per_s = nobs_total * args.steps / elapsed
        abytes = algorithmic_bytes_linearize(prob)
        avg_lin = acc["lin_kernel"] / max(1, acc["launches"])
        achieved = abytes / avg_lin / 1e9 if avg_lin > 0 else 0.0
        out = {
            "metric": "BA residuals/sec (LM-iterations/sec x observations); RANSAC hypotheses/sec alongside",
            "value": res_per_s, "unit": "residuals/s",
            "lm_iterations_per_sec": it_per_s,
            "lm_iterations_per_sec_pcie_inclusive": pcie_it_per_s,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synth_ba_v1: {VIEWS} views / {TRACKS_PER_GPU * world} tracks / {nobs_total} observations, pinhole, "
                                   f"TRIVIAL loss, intrinsics NONE, homogeneous-manifold points, {ITERS_PER_SOLVE} LM iterations per solve"
                                   + ("" if world == 1 else f", tracks sharded over {world} ranks, RCCL all-reduce of the reduced camera system"),
                       "views": VIEWS, "tracks": TRACKS_PER_GPU * world, "observations": nobs_total,
                       "baseline_config": "BASELINE.json configs[1]" if world == 1 else "configs[1] x N tracks (weak)"},
            "roofline": {"kernel": "linearize + Schur assembly launch group (k_lin_obs + k_schur)",
                         "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic_bytes(world),
                         "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": 1e3 * avg_lin,
                         "launches": acc["launches"],
                         "timing": "HIP events around the kernel group, instrumented pass of the same workload right after the timed region"},
            "phase_ms_per_iteration": {"linearize_schur": 1e3 * acc["lin"] / max(1, acc["launches"]),
                                       "reduced_solve": 1e3 * acc["solve"] / max(1, acc["launches"]),
                                       "backsub_trial_cost": 1e3 * acc["backsub"] / max(1, acc["launches"])},
        }
    h.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle ("port"), bounded sample of the same workload
        from tests import oracle_lib as ol
        oo = ol.default_options()
        n_it = 40
        oo.max_num_iterations = n_it
        oo.function_tolerance = 0.0; oo.gradient_tolerance = 0.0; oo.parameter_tolerance = 0.0
        oo.use_inner_iterations = 0
        pc = pristine.copy()
        t0 = time.perf_counter()
        so, _ = ol.solve(pc, oo, trace_capacity=1)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nobs_total * so.num_iterations / dt, "unit": "residuals/s",
                               "lm_iterations_per_sec": so.num_iterations / dt,
                               "cores": 1, "kind": "port",
                               "sample": f"{so.num_iterations} LM iterations of the same {VIEWS}-view/{TRACKS_PER_GPU}-track problem "
                                         f"(oracle/ba_oracle.cpp, scalar FP64, Jet autodiff + Schur + dense Cholesky), {dt:.1f} s",
                               "host_cores_available": os.cpu_count()}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]

    if rank == 0 and world == 1 and not args.no_ransac:
        try:
            from pytheiasfm_amd import ransac
            out["ransac"] = ransac.bench(cpu_baseline=not args.no_cpu_baseline)
        except ImportError:
            pass

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
