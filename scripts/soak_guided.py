"""GPU box: GuidedEpipolarMatches on random two-view scenes (points, distractors, descriptor dimension, band width, Lowe ratio,
share of known matches, second-camera pose drawn per seed) against oracle/sfm_rules.py; relative-position batches of ragged
sizes against the wave-order oracle.  A soak, not a test.  usage: soak_guided.py [count] [first seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytheiasfm_amd import ba, twoview as tv
from tests import oracle_lib as ol
from tests.test_guided_matching_gpu import _scene
from tests.test_relpos import make_pair

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
R = ol.sfm_rules()
bad = 0; t0 = time.time(); total = 0
for k in range(count):
    seed = seed0 + k
    rng = np.random.default_rng(0x601D0000 + seed)
    npts = int(rng.integers(30, 900)); nextra = int(rng.integers(0, 400)); dim = int(rng.choice([8, 32, 64, 128]))
    known = float(rng.uniform(0.05, 0.9)); maxd = float(rng.choice([0.5, 1.0, 2.0, 4.0, 8.0])); ratio = float(rng.uniform(0.5, 0.95))
    cam1, cam2, f1, f2, given, truth = _scene(0x601D0000 + seed, npts, nextra, dim, known)
    cam2["ext"] = cam2["ext"] + np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.1, 0.1, 3)])
    got = tv.GuidedEpipolarMatches(cam1, cam2, f1, f2, given, maxd, ratio, seed=seed)
    ref = R.guided_epipolar_matches(ol, cam1["ext"], cam1["intr"], cam2["ext"], cam2["intr"], f1.keypoints, f1.descriptors, f2.keypoints,
                                    f2.descriptors, given, maxd, ratio, seed)
    same = got == ref
    bad += not same; total += len(got) - len(given)
    if not same or k % 10 == 0:
        print(f"guided seed {seed}: {len(f1.keypoints)} x {len(f2.keypoints)} features dim {dim} band {maxd} ratio {ratio:.2f} given {len(given)} "
              f"added {len(got) - len(given)}: {'same' if same else 'DIFFERS'}", flush=True)
print(f"guided soak: {count} scenes, {bad} differ, {total} matches added, {time.time() - t0:.0f} s", flush=True)
bad2 = 0; npairs = 0; t0 = time.time()
for k in range(max(1, count // 4)):
    rng = np.random.default_rng(0x7E1A0000 + seed0 + k)
    pairs = []
    for j in range(int(rng.integers(1, 300))):
        n = int(rng.choice([0, 1, 2, 3, 5, 63, 64, 65, 128, 129, int(rng.integers(4, 3000))]))
        c, w1, w2, _ = make_pair(int(rng.integers(1 << 30)), max(n, 1), noise=float(rng.choice([0.0, 1e-3, 1e-2])), outliers=float(rng.choice([0.0, 0.1, 0.3])))
        pairs.append((c[:n], w1, w2))
    offsets = np.concatenate([[0], np.cumsum([len(p[0]) for p in pairs])]).astype(np.int64)
    corr = np.vstack([p[0] for p in pairs]) if offsets[-1] else np.zeros((0, 4))
    rot = np.array([np.concatenate([p[1], p[2]]) for p in pairs])
    pos, it = ba.optimize_relative_position_batch(offsets, corr, rot)
    for j, p in enumerate(pairs):
        want, wit = ol.optimize_relative_position(p[0], p[1], p[2], order=1)
        if it[j] != wit or not np.array_equal(pos[j], want, equal_nan=True):
            bad2 += 1; print(f"relpos batch {k} pair {j} n {len(p[0])}: DIFFERS {pos[j]} {want} it {it[j]} {wit}", flush=True)
    npairs += len(pairs)
print(f"relpos soak: {npairs} pairs, {bad2} differ, {time.time() - t0:.0f} s")
