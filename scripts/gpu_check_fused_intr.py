"""GPU: the fused intrinsics assembly (ba_fused_intr.hip) against the gather kernels (THEIA_HIP_INTR_GATHER=1) and the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ba, synth, sfm
from tests import oracle_lib as ol

def rel(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
FR = int(sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION)
FRA = FR | int(sfm.OptimizeIntrinsicsType.ASPECT_RATIO)
for groups, intr, manifold, mixed in ((1, FR, 1, False), (3, FR, 1, True), (7, FR, 0, False), (2, FRA, 1, False), (8, FR, 1, True)):
    p = synth.synth_ba_v1(24, 2500, seed=0x1F5 + groups, num_groups=groups, fix_gauge=True, pixel_noise=0.3, mixed_models=mixed)
    o, oo = ba.default_options(), ol.default_options()
    for q in (o, oo):
        q.intrinsics_to_optimize = intr; q.max_num_iterations = 6; q.use_homogeneous_point_parametrization = manifold
    res = []
    for gather in (False, True):
        if gather: os.environ["THEIA_HIP_INTR_GATHER"] = "1"
        try:
            with ba.BaHandle(p.copy(), o) as h:
                S, rhs = h.reduced_system(1e4)
            q = p.copy(); s, tr = ba.solve(q, o)
            res.append((S, rhs, q, s, tr))
        finally:
            os.environ.pop("THEIA_HIP_INTR_GATHER", None)
    So, ro = ol.reduced_system(p, oo, 1e4)
    qo = p.copy(); so, tro = ol.solve(qo, oo)
    a, b = res
    print(f"groups {groups} intr {intr:#x} manifold {manifold} mixed {mixed}: S fused-gather {rel(a[0], b[0]):.2e} rhs {rel(a[1], b[1]):.2e} | S fused-oracle {rel(a[0], So):.2e} "
          f"rhs {rel(a[1], ro):.2e} | gather-oracle {rel(b[0], So):.2e} | it {a[3].num_iterations} {b[3].num_iterations} {so.num_iterations} cost {a[3].final_cost:.9e} {so.final_cost:.9e} "
          f"dintr {rel(a[2].intrinsics, qo.intrinsics):.2e} dcam {np.abs(a[2].cam_ext - qo.cam_ext).max():.2e}", flush=True)
    if rel(a[0], So) > 1e-8:
        d = np.abs(a[0] - So); i, j = np.unravel_index(d.argmax(), d.shape); print("   worst entry", i, j, a[0][i, j], So[i, j], "ni", 10 * groups)
# timing at C4 (8 groups, FOCAL | RADIAL)
p = synth.ba_config("C4")
o = ba.default_options(); o.intrinsics_to_optimize = FR; o.max_num_iterations = 8
for gather in (False, True):
    if gather: os.environ["THEIA_HIP_INTR_GATHER"] = "1"
    try:
        t0 = time.perf_counter()
        h = ba.BaHandle(p.copy(), o)
        t1 = time.perf_counter()
        s, tr = h.run()
        h.reset(p); 
        t2 = time.perf_counter(); s, tr = h.run(); t3 = time.perf_counter()
        print(f"C4 intr {'gather' if gather else 'fused'}: create {1e3*(t1-t0):.1f} ms, {s.num_iterations} it in {1e3*(t3-t2):.2f} ms = {1e3*(t3-t2)/max(1,s.num_iterations):.3f} ms/it, cost {s.initial_cost:.6e} -> {s.final_cost:.9e}", flush=True)
        h.close() if hasattr(h, "close") else None
    finally:
        os.environ.pop("THEIA_HIP_INTR_GATHER", None)
