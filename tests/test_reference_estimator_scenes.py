"""CPU: the reference's estimator tests (estimate_{essential_matrix, homography, dominant_plane_from_points,
relative_pose_with_known_orientation, absolute_pose_with_known_orientation, uncalibrated_absolute_pose, fundamental_matrix,
uncalibrated_relative_pose}_test.cc) restated scene by scene (tests/estimator_scenes.py) and run through the ORACLE with the
reference's options and pass criteria."""
import numpy as np
import pytest

from tests import estimator_scenes as es
from tests import oracle_lib as ol

CASES = es.cases()


def oracle_estimate(est, data, prm, ep):
    ol.set_estimator_params(ep if ep is not None else np.zeros(2))
    o = ol.ransac_estimate(est, data, prm.to_c())
    return bool(o["success"]), o["model"], o["inlier_mask"].astype(bool)


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_reference_estimator_scene_through_the_oracle(case):
    es.run(dict(case), oracle_estimate)
