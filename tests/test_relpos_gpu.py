"""GPU: theia_hip_optimize_relative_position_batch (N x OptimizeRelativePositionWithKnownRotation, one pair per wavefront,
csrc/relpos_irls.hip) against the oracle and the committed golden vectors, and OptimizeAbsolutePoseOnNormFeatures (a
BundleAdjustView of a default pinhole camera, pose_wrapper.cc:39-65) against the oracle's LM on the same flat problem."""
import os

import numpy as np
import pytest

from pytheiasfm_amd import _capi as capi, ba, sfm, synth
from tests import oracle_lib as ol
from tests.test_relpos import make_pair, make_pairs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _batch(pairs):
    offsets = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
    corr = np.vstack([p[0] for p in pairs])
    rot = np.array([np.concatenate([p[1], p[2]]) for p in pairs])
    return offsets, corr, rot


def test_relative_position_batch_follows_oracle_and_golden():
    """Bit-identical to the oracle run with the device's summation order (64 interleaved partial sums + XOR butterfly); same
    iteration counts and 1e-9 against the index-order oracle on the golden pairs; the true direction on noise-free pairs."""
    pairs = make_pairs()
    offsets, corr, rot = _batch(pairs)
    pos, it = ba.optimize_relative_position_batch(offsets, corr, rot)
    g = np.load(os.path.join(HERE, "golden", "relpos_irls.npz"))
    for k, (c, w1, w2, truth) in enumerate(pairs):
        opos, oit = ol.optimize_relative_position(c, w1, w2, order=1)
        assert it[k] == oit == int(g[f"itw{k}"]), (k, it[k], oit)
        assert np.array_equal(pos[k], opos) and np.array_equal(pos[k], g[f"posw{k}"]), k
        assert np.abs(pos[k] - g[f"pos{k}"]).max() <= 1e-6, k      # index-order sums (conditioning: test below)
    for k, (c, w1, w2, truth) in enumerate(pairs[:16]):
        if k % 3 == 0 and k % 2 == 0 and len(c) >= 8:
            assert np.abs(pos[k] - truth).max() < 1e-7, k


def test_relative_position_many_pairs_and_mirror():
    """600 pairs of 40 .. 700 matches (the per-edge calls of RefineRelativeTranslationsWithKnownRotations as one launch): every
    pair bit-identical to the wave-order oracle -- positions and iteration counts -- whatever batch it travels in.  Against the
    INDEX-order oracle the IRLS shows its conditioning: the 1 / w weights of near-zero residuals amplify the rounding
    differences between two summation orders, so that on noisy pairs a few per cent of the runs stop an iteration apart and the
    unit vectors differ by up to ~1e-4 (measured on this set: 44 of 600 iteration counts, largest difference 1.2e-4, median
    4e-15; noise-free pairs 9e-8) -- that is the indeterminacy of the reference's own result under a change of its BLAS order."""
    rng = np.random.RandomState(5)
    pairs = [make_pair(1000 + k, int(rng.randint(40, 700)), noise=1e-3 * (k % 3), outliers=0.05 * (k % 4)) for k in range(600)]
    offsets, corr, rot = _batch(pairs)
    pos, it = ba.optimize_relative_position_batch(offsets, corr, rot)
    dev, mism = [], 0
    for k in range(600):
        opos, oit = ol.optimize_relative_position(*pairs[k][:3], order=1)
        assert it[k] == oit and np.array_equal(pos[k], opos), k
        if k % 5 == 0:
            ipos, iit = ol.optimize_relative_position(*pairs[k][:3], order=0)
            mism += int(iit != it[k]); dev.append(np.abs(pos[k] - ipos).max())
    dev = np.array(dev)
    assert np.median(dev) < 1e-12 and dev.max() < 2e-3 and mism <= 0.2 * len(dev), (np.median(dev), dev.max(), mism)
    mp = sfm.OptimizeRelativePositionWithKnownRotationBatch([p[0] for p in pairs[:9]], [p[1] for p in pairs[:9]], [p[2] for p in pairs[:9]])
    assert np.array_equal(mp, pos[:9])
    ok, one = sfm.OptimizeRelativePositionWithKnownRotation(pairs[4][0], pairs[4][1], pairs[4][2])
    assert ok and np.array_equal(one, pos[4])
    # argument checks of the C entry point
    with pytest.raises(capi.TheiaHipError):
        ba.optimize_relative_position_batch(np.array([1, 5]), corr[:5], rot[:1])


def test_optimize_absolute_pose_on_norm_features():
    data, offsets, truth = synth.synth_ransac_v1(6, 300, "absolute", seed=0x5AC57000, inlier_lo=1.0, inlier_hi=1.0, noise_px=0.5)
    o = sfm.BundleAdjustmentOptions(); o.max_num_iterations = 20
    c5, R0, p0 = [], [], []
    for i in range(6):
        d = data[offsets[i]:offsets[i + 1]]
        c5.append(d[:, :5])
        R0.append(synth.angle_axis_to_matrix(synth.matrix_to_angle_axis(truth["R"][i]) + 0.02))
        p0.append(truth["position"][i] + 0.05)
    out = sfm.OptimizeAbsolutePoseOnNormFeaturesBatch(c5, R0, p0, o)
    for i, (ok, R, pos) in enumerate(out):
        assert ok
        # the oracle's LM on the same one-view problem (points constant, default pinhole camera)
        n = len(c5[i])
        intr = np.zeros((1, capi.THEIA_MAX_INTRINSICS)); intr[0, :2] = 1.0
        ext = np.concatenate([p0[i], synth.matrix_to_angle_axis(R0[i])])[None]
        flat = capi.FlatProblem(ext.copy(), intr, [0], [0], np.concatenate([c5[i][:, 2:5], np.ones((n, 1))], axis=1), c5[i][:, :2],
                                np.zeros(n, np.int32), np.arange(n, dtype=np.int32), point_const=np.ones(n, np.uint8))
        oo = sfm._no_inner(o).to_c()
        so, _ = ol.solve(flat, oo)
        assert so.success
        assert np.abs(pos - flat.cam_ext[0, :3]).max() <= 1e-8 and np.abs(synth.matrix_to_angle_axis(R) - flat.cam_ext[0, 3:]).max() <= 1e-8
        assert np.linalg.norm(pos - truth["position"][i]) < 0.02
    ok1, R1, p1 = sfm.OptimizeAbsolutePoseOnNormFeatures(c5[2], R0[2], p0[2], o)
    assert ok1 and np.array_equal(p1, out[2][2]) and np.array_equal(R1, out[2][1])


def test_the_reference_tests_scenes_on_the_device():
    """The scenes of optimize_relative_position_with_known_rotation_test.cc as one batch through the C entry point: NoNoise within
    1e-6 degrees on all 40, PixelNoise within 2 degrees in the median (tests/test_relpos.py says why not on every scene), and
    the same vectors as the oracle in the device's summation order."""
    from tests.test_relpos import reference_test_scene, _angle_deg
    for noise in (0.0, 1.0):
        pairs = [reference_test_scene(5200 + seed, noise) for seed in range(40)]
        offsets, corr, rot = _batch(pairs)
        pos, it = ba.optimize_relative_position_batch(offsets, corr, rot)
        err = np.array([_angle_deg(pos[k], pairs[k][3]) for k in range(40)])
        assert (err.max() < 1e-6) if noise == 0.0 else (np.median(err) < 2.0), (noise, err.max())
        for k in range(40):
            want, wit = ol.optimize_relative_position(pairs[k][0], pairs[k][1], pairs[k][2], order=1)
            assert np.array_equal(pos[k], want) and it[k] == wit, k


def test_relative_position_ragged_and_tiny_pairs():
    """Pairs of 0, 1, 2 and 3 matches between ordinary ones, and an empty batch: the reference runs its loop on whatever it is given
    (optimize_relative_position_with_known_rotation.cc:116-); the device returns the oracle's bits for every pair."""
    pos, it = ba.optimize_relative_position_batch(np.zeros(1, np.int64), np.zeros((0, 4)), np.zeros((0, 6)))
    assert pos.shape == (0, 3) and it.shape == (0,)
    pairs = []
    for k, n in enumerate((0, 150, 1, 2, 65, 3, 0, 64)):
        c, w1, w2, _ = make_pair(7700 + k, max(n, 1))
        pairs.append((c[:n], w1, w2))
    offsets, corr, rot = _batch(pairs)
    pos, it = ba.optimize_relative_position_batch(offsets, corr, rot)
    for k, p in enumerate(pairs):
        want, wit = ol.optimize_relative_position(p[0], p[1], p[2], order=1)
        assert it[k] == wit, (k, it[k], wit)
        assert np.array_equal(pos[k], want, equal_nan=True), (k, pos[k], want)


def test_wave_primitives_selftest():
    """theia_hip_selftest_wave_primitives: the cross-lane primitives of the bit-exact kernels (butterfly sums on permlane swaps + DPP,
    the wave maximum of |a|, the row / half-row broadcasts, readlane) against shuffle loops, bit for bit, on 512 random wavefronts."""
    import ctypes as C
    L = capi.lib()
    L.theia_hip_selftest_wave_primitives.argtypes = [C.c_int32, C.POINTER(C.c_int32)]
    bad = C.c_int32(-1)
    capi.check(L.theia_hip_selftest_wave_primitives(512, C.byref(bad)))
    assert bad.value == 0
    with pytest.raises(capi.TheiaHipError):
        capi.check(L.theia_hip_selftest_wave_primitives(0, C.byref(bad)))
