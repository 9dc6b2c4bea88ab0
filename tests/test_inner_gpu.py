"""GPU: inner iterations (ba_inner.hip: coordinate descent over cameras, shared intrinsics, points after every
trust-region candidate; on by default like the reference) against the oracle's restatement of Ceres'
CoordinateDescentMinimizer.  The block solves stop on relative tolerances (1e-6 function, 1e-8 parameter), so two
implementations agree to those tolerances, not to the last bit: costs 1e-6, parameters 1e-5 of their scale."""
import numpy as np
import pytest

from pytheiasfm_amd import ba, sfm, synth
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _both(p, iters=12, **kw):
    out = []
    for mod in (ba, ol):
        q = p.copy(); o = mod.default_options(); o.max_num_iterations = iters
        assert o.use_inner_iterations == 1
        for k, v in kw.items():
            setattr(o, k, v)
        s, tr = mod.solve(q, o)
        out.append((q, s, tr))
    return out


def _compare(g, o, ptol=1e-5):
    (qg, sg, tg), (qo, so, to) = g, o
    assert sg.success and so.success
    assert sg.num_iterations == so.num_iterations and sg.num_successful_steps == so.num_successful_steps
    n = min(len(tg.cost), len(to.cost))
    assert list(tg.accepted[:n]) == list(to.accepted[:n])
    assert np.allclose(tg.cost[:n], to.cost[:n], rtol=1e-6)
    assert abs(sg.final_cost - so.final_cost) <= 1e-6 * so.final_cost
    for a, b in ((qg.cam_ext, qo.cam_ext), (qg.points, qo.points), (qg.intrinsics, qo.intrinsics)):
        assert np.abs(a - b).max() <= ptol * max(1.0, np.abs(b).max())


INTR = int(sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION)


@pytest.mark.parametrize("kw", [dict(), dict(loss_function_type=1, robust_loss_width=2.0), dict(intrinsics_to_optimize=INTR),
                                dict(use_homogeneous_point_parametrization=0), dict(loss_function_type=3, robust_loss_width=3.0, intrinsics_to_optimize=INTR),
                                dict(constant_camera_orientation=1), dict(constant_camera_position=1)],
                         ids=["trivial", "huber", "intrinsics", "xyzw", "cauchy+intrinsics", "const-orientation", "const-position"])
def test_inner_iterations_follow_the_oracle(kw):
    p = synth.synth_ba_v1(8, 300, seed=5)
    g, o = _both(p, **kw)
    _compare(g, o)


def test_inner_iterations_c1_and_mixed_models():
    for p in (synth.ba_config("C1"), synth.synth_ba_v1(12, 500, seed=9, mixed_models=True)):
        g, o = _both(p)
        _compare(g, o)


def test_inner_iterations_with_constant_blocks_priors_and_depth_rows():
    p = synth.synth_ba_v1(10, 400, seed=21)
    rng = np.random.default_rng(3)
    p.cam_const = np.zeros(10, dtype=np.uint8); p.cam_const[0] = 3; p.cam_const[4] = 1
    p.point_const = (rng.uniform(size=400) < 0.1).astype(np.uint8)
    nc = 10
    mask = np.zeros(nc, dtype=np.uint8); mask[[1, 2, 5]] = [1, 2, 4]
    eye = np.tile(np.eye(3) * 3.0, (nc, 1, 1))
    p.set_priors(mask, position=(p.cam_ext[:, :3] + 0.01, eye), gravity=(np.tile([0.0, 0.0, -1.0], (nc, 1)), eye),
                 orientation=(p.cam_ext[:, 3:] + 0.005, eye))
    p.add_depth_priors(np.arange(0, 200, 7), 6.0, variance=0.25)
    g, o = _both(p, prior_mask=7)
    _compare(g, o)


def test_inner_on_beats_inner_off_per_iteration_and_handles_set_options():
    p = synth.synth_ba_v1(8, 300, seed=5)
    res = {}
    for inner in (1, 0):
        q = p.copy(); o = ba.default_options(); o.use_inner_iterations = inner; o.max_num_iterations = 1
        s, tr = ba.solve(q, o)
        res[inner] = tr.cost[1]
    assert res[1] < 0.995 * res[0]
    # a handle created without the lists refuses to switch them on later (the lists are built at create())
    o = ba.default_options(); o.use_inner_iterations = 0
    with ba.BaHandle(p.copy(), o) as h:
        o.use_inner_iterations = 1
        with pytest.raises(Exception):
            h.set_options(o)


def test_forward_facing_option_is_the_same_reduced_system():
    """optimize_for_forward_facing_trajectory (bundle_adjuster.cc:547-563) only collapses Ceres' elimination groups 1 and 2:
    same Schur complement, same steps; the inner iterations keep working while the intrinsics are constant."""
    p = synth.synth_ba_v1(8, 300, seed=5)
    out = []
    for flag in (False, True):
        rec = sfm.Reconstruction.from_flat(p)
        o = sfm.BundleAdjustmentOptions(); o.max_num_iterations = 8; o.optimize_for_forward_facing_trajectory = flag
        s = sfm.BundleAdjustReconstruction(o, rec)
        out.append((s, rec))
    assert out[0][0].success and out[1][0].success and out[0][0].final_cost == out[1][0].final_cost
    assert np.array_equal(out[0][1].cam_ext, out[1][1].cam_ext)


def test_group_sweep_gives_up_and_redoes_when_its_workgroups_cannot_meet():
    """The shared-intrinsics sweep runs on several co-resident workgroups per group that meet at arrival counters
    (k_inner_groups).  If partners never arrive -- two such launches of different processes could split the CUs -- every
    workgroup leaves after a bounded number of polls and the one-workgroup launch that always follows redoes the groups
    that are not marked done.  THEIA_HIP_INNER_GROUPS_MAX_POLLS=0 forces that path (the switch is read once per process,
    hence the subprocess): the intrinsics cases must still follow the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, THEIA_HIP_INNER_GROUPS_MAX_POLLS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(root, "tests", "test_inner_gpu.py"),
                        "-k", "intrinsics and follow_the_oracle"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_group_sweep_fallback_reproduces_the_cooperative_launch_bit_for_bit():
    """the one-workgroup launch of k_inner_groups (cooperative launch refused, stream capture, or a timed-out
    arrival) summed a group's observations in another order than the launch with several workgroups per group, so a timeout
    decided the low bits of the refined intrinsics.  Both now deal a pass to the same parts and add them in the same order:
    the solve with THEIA_HIP_INNER_GROUPS_MAX_POLLS=0 (every cooperative workgroup gives up, the follow-up launch redoes the
    groups) and with THEIA_HIP_INNER_GROUPS_SINGLE=1 (no cooperative launch at all) must equal the normal solve exactly."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np\n"
            "from pytheiasfm_amd import ba, sfm, synth\n"
            "p = synth.synth_ba_v1(16, 1200, seed=11, mixed_models=True)\n"
            "o = ba.default_options(); o.max_num_iterations = 6\n"
            "o.intrinsics_to_optimize = int(sfm.OptimizeIntrinsicsType.FOCAL_LENGTH | sfm.OptimizeIntrinsicsType.RADIAL_DISTORTION)\n"
            "s, tr = ba.solve(p, o)\n"
            "print('HEX', p.intrinsics.tobytes().hex(), p.cam_ext.tobytes().hex()[:4096], repr(float(s.final_cost)))\n")
    outs = []
    for extra in ({}, {"THEIA_HIP_INNER_GROUPS_MAX_POLLS": "0"}, {"THEIA_HIP_INNER_GROUPS_SINGLE": "1"}):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("HEX")][-1])
    assert outs[0] == outs[1], "give-up-and-redo path differs from the cooperative launch"
    assert outs[0] == outs[2], "one-workgroup launch differs from the cooperative launch"
