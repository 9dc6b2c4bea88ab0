"""CPU: inner iterations of the oracle (Ceres' coordinate descent after every trust-region candidate, on by default:
bundle_adjustment.h:144) -- properties that do not depend on any second implementation."""
import numpy as np

from pytheiasfm_amd import synth
from tests import oracle_lib as ol


def _solve(p, inner, iters, **kw):
    q = p.copy(); o = ol.default_options(); o.use_inner_iterations = inner; o.max_num_iterations = iters
    for k, v in kw.items():
        setattr(o, k, v)
    s, tr = ol.solve(q, o)
    return q, s, tr, o


def test_default_is_on_and_the_first_step_lands_lower():
    assert ol.default_options().use_inner_iterations == 1
    p = synth.synth_ba_v1(8, 300, seed=5)
    _, s1, t1, _ = _solve(p, 1, 1)
    _, s0, t0, _ = _solve(p, 0, 1)
    assert t1.cost[0] == t0.cost[0]
    # same trust-region candidate, then one sweep of block coordinate descent: never worse, here clearly better
    assert t1.cost[1] < t0.cost[1] and t1.cost[1] < 0.995 * t0.cost[1]
    qa, sa, _, _ = _solve(p, 1, 30)
    qb, sb, _, _ = _solve(p, 0, 30)
    assert sa.success and sb.success and abs(sa.final_cost - sb.final_cost) <= 1e-5 * sb.final_cost


def test_points_are_block_optimal_after_the_sweep():
    """The last independent set of the reversed ordering is the points: after an accepted step every point sits at the
    minimum of its own block (its gradient vanishes to the inner solver's tolerance), which the plain LM step does not give."""
    p = synth.synth_ba_v1(8, 300, seed=7)
    grads = {}
    for inner in (1, 0):
        q, s, tr, o = _solve(p, inner, 1, use_homogeneous_point_parametrization=0)
        assert s.num_successful_steps == 1
        ok, cost, r, jc, jp = ol.evaluate(q, o)
        g = np.zeros((q.points.shape[0], 4))
        np.add.at(g, q.obs_pt, np.einsum("nij,ni->nj", jp, r))
        grads[inner] = np.linalg.norm(g, axis=1)
    assert np.median(grads[1]) < 1e-3 * np.median(grads[0])
    assert grads[1].max() < 1e-2 * np.median(grads[0])


def test_inner_iterations_switch_themselves_off():
    """inner_iteration_tolerance = 1e-3: once a sweep improves the candidate by less than that, later iterations are the
    plain LM -- the traces of inner ON and of a solve that is ON only for the first k iterations coincide afterwards.
    Checked indirectly: a converged problem gains nothing from the sweep, ON and OFF take the same steps."""
    p = synth.synth_ba_v1(6, 200, seed=11)
    q, s, _, _ = _solve(p, 0, 50)
    _, s1, t1, _ = _solve(q, 1, 5)
    _, s0, t0, _ = _solve(q, 0, 5)
    assert s1.num_iterations == s0.num_iterations
    assert np.allclose(t1.cost[: len(t0.cost)], t0.cost, rtol=1e-9)
