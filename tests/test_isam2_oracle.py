"""Pins the ISAM2 restatement (oracle/isam2_oracle.hpp) against the reference's own expectations:
  tests/testGaussianISAM2.cpp:223-281 (isam_check): after createSlamlikeISAM2 (relinearization off) the incremental estimate equals
      fullinit.retract(fullgraph.linearize(fullinit)->optimize()) (assert_equal, 1e-9) and the Bayes tree's augmented Hessian equals
      the Hessian of the full graph linearized at the linearization point; :284-306 (TEST simple for 0..9 poses, slamlike_solution_gaussnewton)
  tests/testVisualISAM2.cpp:33-118: VisualISAM2Example's sequence (relinearizeThreshold 0.01, relinearizeSkip 1, an extra update per
      frame) ends with 16 variables and every landmark within 0.01 of the ground truth
CCOLAMD comes from oracle/_ref (the reference's vendored C source compiled in place)."""
import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import NonlinearFactorGraph, Ordering, Values
from gtsam_personal_amd.graph import VAR_DIM, symbol
from isam2_examples import create_points, slamlike_steps, visual_steps

pytestmark = pytest.mark.skipif(not oh.have_ref(), reason="oracle/_ref (CCOLAMD of the reference) not built")


def merge(steps):
    graph, init = NonlinearFactorGraph(), Values()
    for g, v in steps:
        graph.push_back(g)
        for k in v.keys():
            init.insert(k, v.type(k), v.at(k))
    return graph, init


def tree_hessian(cliques, dim_of, keys_sorted):
    """augmented Hessian of GaussianFactorGraph(isam): sum over cliques of [R S d]^T [R S d], variables ascending by key"""
    off, o = {}, 0
    for k in keys_sorted:
        off[k] = o
        o += dim_of[k]
    H = np.zeros((o + 1, o + 1))
    for keys, _, rsd, _ in cliques:
        cols = np.concatenate([np.arange(off[k], off[k] + dim_of[k]) for k in keys] + [[o]])
        H[np.ix_(cols, cols)] += rsd.T @ rsd
    return H


def isam_check(isam, fullgraph, fullinit):
    # expected = fullinit.retract(fullgraph.linearize(fullinit)->optimize())
    orc = oh.OracleProblem(fullgraph, fullinit, Ordering.Natural(fullgraph))
    orc.linearize()
    rc, delta, _, _ = orc.solve(0.0)
    assert rc == 0
    orc.retract(delta)
    expected = orc.values()
    actual = isam.calculateEstimate()
    assert actual.keys() == sorted(expected)
    for k in expected:
        assert np.allclose(actual.at(k), expected[k], rtol=0, atol=1e-9), k
    # information: Bayes tree == full graph linearized at the linearization point (relinearization is off: == fullinit)
    lin = isam.getLinearizationPoint()
    for k in lin.keys():
        assert np.array_equal(lin.at(k), fullinit.at(k))
    dim_of = {k: VAR_DIM[lin.type(k)] for k in lin.keys()}
    actualH = tree_hessian(isam.cliques(), dim_of, lin.keys())
    off, o = {}, 0
    for k in lin.keys():
        off[k] = o
        o += dim_of[k]
    expectedH = np.zeros_like(actualH)
    for i, fk in enumerate(fullgraph.factor_keys_in_graph_order()):
        Ab = orc.jacobian(i)
        cols = np.concatenate([np.arange(off[k], off[k] + dim_of[k]) for k in fk] + [[o]])
        expectedH[np.ix_(cols, cols)] += Ab.T @ Ab
    expectedH[-1, -1] = actualH[-1, -1]
    assert np.allclose(actualH, expectedH, rtol=0, atol=1e-9 * max(1.0, np.abs(expectedH).max()))
    # consistency of the tree: every variable is frontal in exactly one clique
    frontal = [k for keys, nfk, _, _ in isam.cliques() for k in keys[:nfk]]
    assert sorted(frontal) == lin.keys()


@pytest.mark.parametrize("max_poses", list(range(10)) + [10])
def test_slamlike_matches_batch_solution(max_poses):
    """TEST(ISAM2, simple) for 0..9 and slamlike_solution_gaussnewton: ISAM2Params(GaussNewton(0.001), 0.0, 0, false)"""
    steps = slamlike_steps(max_poses)
    isam = oh.OracleISAM2(relinearizeThreshold=0.0, relinearizeSkip=0, enableRelinearization=False, wildfireThreshold=0.001)
    for g, v in steps:
        isam.update(g, v)
    isam_check(isam, *merge(steps))


def test_visual_isam2_reaches_the_ground_truth():
    isam = oh.OracleISAM2(relinearizeThreshold=0.01, relinearizeSkip=1)
    for g, v in visual_steps():
        isam.update(g, v)
    result = isam.calculateEstimate()
    assert len(result.keys()) == 16
    for j, p in enumerate(create_points()):
        assert np.allclose(result.at(symbol("l", j)), p, rtol=0, atol=0.01), j
