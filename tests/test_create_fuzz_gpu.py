"""Handle creation under randomised structure (GPU): the host-side plan -- track CSR of the input, counting sorts, the fused
plan in segments, the staged uploads -- against itself across input orders and threading, and against the oracle.

Every case draws a small problem with the features that steer create() down its different paths: observations in track-major,
view-major or random order, unused track ids (gaps), constant cameras and points (fixed-cost blocks), unobserved cameras, free
intrinsics, inner-iteration lists.  THEIA_HIP_HOST_CHUNK_MIN=64 sends these small problems through the THREADED passes (the
persistent host-thread team, the atomic-cursor scatter, per-part histograms, 32 plan segments) that otherwise only run at
> 262 144 observations."""
import os

import numpy as np
import pytest

from pytheiasfm_amd import _capi as capi, ba, synth
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(seed)
    nv = int(rng.integers(5, 40)); nt = int(rng.integers(40, 1500))
    p = synth.synth_ba_v1(nv, nt, seed=0xF0220000 + seed, num_groups=int(rng.integers(1, 4)), mixed_models=bool(rng.integers(0, 2)),
                          fix_gauge=bool(rng.integers(0, 2)))
    np_ = p.points.shape[0]
    # unused track ids: spread the tracks over a larger id range (the skipped ids have no observations)
    if rng.integers(0, 2):
        stride = int(rng.integers(2, 4))
        pts = np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (np_ * stride, 1)); pts[::stride] = p.points
        p = capi.FlatProblem(p.cam_ext, p.intrinsics, p.group_model, p.cam_group, pts, p.obs_uv, p.obs_cam, p.obs_pt * stride,
                             p.cam_const, p.group_const, None)
        np_ = pts.shape[0]
    if rng.integers(0, 2):
        pc = np.zeros(np_, np.uint8); pc[rng.integers(0, np_, max(1, np_ // 9))] = 1; p.point_const = pc
    if rng.integers(0, 2):
        cc = np.zeros(nv, np.uint8) if p.cam_const is None else p.cam_const.copy()
        cc[rng.integers(0, nv, 2)] = 3; p.cam_const = cc
    if rng.integers(0, 3) == 0:       # a camera nobody observes: drop its observations
        keep = p.obs_cam != int(rng.integers(0, nv))
        p = capi.FlatProblem(p.cam_ext, p.intrinsics, p.group_model, p.cam_group, p.points, p.obs_uv[keep], p.obs_cam[keep],
                             p.obs_pt[keep], p.cam_const, p.group_const, p.point_const)
    o = ba.default_options(); o.max_num_iterations = 4
    o.use_inner_iterations = int(rng.integers(0, 2))
    o.intrinsics_to_optimize = int(rng.choice([0, 0, 0x11, 0x3f]))
    return p, o, rng


def reorder(p, order):
    return capi.FlatProblem(p.cam_ext.copy(), p.intrinsics.copy(), p.group_model, p.cam_group, p.points.copy(), p.obs_uv[order],
                            p.obs_cam[order], p.obs_pt[order], p.cam_const, p.group_const, p.point_const)


def solve(p, o):
    q = p.copy(); s, t = ba.solve(q, o)
    return s, t.cost[: t.size].copy(), q


@pytest.mark.parametrize("seed", range(int(os.environ.get("THEIA_FUZZ_SEED0", "0")), int(os.environ.get("THEIA_FUZZ_SEED1", "40"))))   # (a soak: other ranges)
def test_creation_paths_agree_on_random_structure(seed):
    p, o, rng = draw(seed)
    n = len(p.obs_pt)
    cnt = np.bincount(p.obs_pt, minlength=p.points.shape[0])
    within = np.arange(n) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    view_major = np.lexsort((p.obs_pt, within))                  # keeps every track's own order
    os.environ.pop("THEIA_HIP_HOST_CHUNK_MIN", None)
    s0, c0, q0 = solve(p, o)
    assert s0.success
    # against the oracle (the serial plan)
    oo = ol.default_options()
    for f in ("max_num_iterations", "use_inner_iterations", "intrinsics_to_optimize"):
        setattr(oo, f, getattr(o, f))
    po = p.copy(); so, to = ol.solve(po, oo)
    assert so.success and to.size == len(c0)
    assert np.max(np.abs(to.cost[: to.size] - c0) / to.cost[: to.size]) < 1e-8
    # input order: the same plan, the same bits (tracks longer than the fused kernel's 22 cameras take an atomics path: 1e-12 there)
    # (and more than four free intrinsics per group take the round-1 gather kernels, whose atomics are not bit-reproducible either)
    exact = cnt.max() <= 22 and o.intrinsics_to_optimize != 0x3f
    s1, c1, q1 = solve(reorder(p, view_major), o)
    assert len(c1) == len(c0) and (np.array_equal(c1, c0) if exact else np.allclose(c1, c0, rtol=1e-11, atol=0))
    # a random order changes the order of a track's observations, hence the order of its sums: the same trajectory to round-off
    s5, c5, q5 = solve(reorder(p, rng.permutation(n)), o)
    assert len(c5) == len(c0) and np.allclose(c5, c0, rtol=1e-10, atol=0) and np.abs(q5.points - q0.points).max() < 1e-7
    # the threaded passes on the same small problem, both orders: one plan (32 segments), the same bits for both orders,
    # the serial plan's trajectory to round-off
    os.environ["THEIA_HIP_HOST_CHUNK_MIN"] = "64"
    try:
        s2, c2, q2 = solve(p, o)
        s3, c3, q3 = solve(reorder(p, view_major), o)
        s6, c6, q6 = solve(reorder(p, rng.permutation(n)), o)
        os.environ["THEIA_HIP_HOST_THREADS"] = "3"
        s4, c4, q4 = solve(p, o)
    finally:
        os.environ.pop("THEIA_HIP_HOST_CHUNK_MIN", None); os.environ.pop("THEIA_HIP_HOST_THREADS", None)
    assert len(c2) == len(c3) == len(c4) == len(c0)
    if exact:
        assert np.array_equal(c2, c3) and np.array_equal(c2, c4) and np.array_equal(q2.points, q3.points)
    assert np.allclose(c2, c0, rtol=1e-10, atol=0) and np.allclose(c3, c0, rtol=1e-10, atol=0) and np.allclose(c6, c0, rtol=1e-10, atol=0)
    assert np.abs(q2.cam_ext - q0.cam_ext).max() < 1e-8 and np.abs(q2.points - q0.points).max() < 1e-7
