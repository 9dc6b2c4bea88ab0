"""GPU: the reference's estimator tests restated scene by scene (tests/estimator_scenes.py: essential matrix, homography,
dominant plane, both known-orientation estimators, uncalibrated absolute / relative pose, fundamental matrix -- 100 scenes)
through the library with the reference's options and pass criteria; and every scene's inlier set, iteration count and model
equal to the oracle's."""
import numpy as np
import pytest

from pytheiasfm_amd import ransac
from tests import estimator_scenes as es
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
CASES = es.cases()


def gpu_estimate(est, data, prm, ep):
    res = ransac.estimate_batch(est, data, np.array([0, len(data)], dtype=np.int64), prm.to_c(), ep)
    return bool(res["success"][0]), res["models"][0], res["inlier_mask"].astype(bool)


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_reference_estimator_scene_on_gpu(case):
    est, data, prm, ep = es.run(dict(case), gpu_estimate)
    res = ransac.estimate_batch(est, data, np.array([0, len(data)], dtype=np.int64), prm.to_c(), ep)
    ol.set_estimator_params(ep if ep is not None else np.zeros(2))
    o = ol.ransac_estimate(est, data, prm.to_c())
    assert o["num_iterations"] == res["num_iterations"][0]
    assert np.array_equal(o["inlier_mask"], res["inlier_mask"])
    assert np.allclose(o["model"], res["models"][0], rtol=1e-12, atol=1e-14)
