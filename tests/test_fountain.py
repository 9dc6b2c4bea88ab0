"""CPU: the real-data scene the reference ships (fountain-P11) -- the parsed fixture is consistent with the camera model
(the reference's own cameras and points reproject onto the reference's own features to well under a pixel), the
ground-truth cameras align, and the oracle's BA leaves the converged reconstruction where it is."""
import numpy as np

from tests import fountain as ft
from tests import oracle_lib as ol


def test_fixture_reprojects_and_aligns_with_ground_truth():
    d = ft.load()
    assert d["cam_ext"].shape == (11, 6) and d["points"].shape == (16616, 4) and d["obs_uv"].shape == (75022, 2)
    assert np.bincount(d["obs_track"]).min() >= 2 and np.bincount(d["obs_track"]).max() == 11
    uv, depth = ft.reproject(d, d["cam_ext"], d["points"])
    err = np.linalg.norm(uv - d["obs_uv"], axis=1)
    assert (depth > 0).all()
    assert np.sqrt((err ** 2).mean()) < 0.5 and np.median(err) < 0.3, (np.sqrt((err ** 2).mean()), np.median(err))
    # camera centres against the ground truth after a similarity (incremental_reconstruction_estimator_test.cc:102-134, 1e-2 m)
    s, R, t = ft.similarity_align(d["cam_ext"][:, :3], d["gt_cam_ext"][:, :3])
    res = np.linalg.norm(d["gt_cam_ext"][:, :3] - (s * d["cam_ext"][:, :3] @ R.T + t), axis=1)
    assert res.max() < 1e-2, res


def test_oracle_bundle_adjustment_on_the_converged_scene():
    d = ft.load()
    p = ft.flat_problem(d)
    o = ol.default_options(); o.max_num_iterations = 5
    s, tr = ol.solve(p, o)
    assert s.success and s.final_cost <= s.initial_cost
    # 0.5 * sum r^2 over 75 022 observations at ~0.25 px
    assert s.initial_cost / d["obs_uv"].shape[0] < 0.2
    assert np.abs(p.cam_ext - d["cam_ext"]).max() < 5e-3
