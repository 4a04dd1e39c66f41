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
from gtsam_personal_amd import noiseModel
from isam2_examples import constrained_ordering_steps, create_points, slamlike_steps, stale_landmark_steps, visual_steps

pytestmark = pytest.mark.skipif(not oh.have_ref(), reason="oracle/_ref (CCOLAMD of the reference) not built")


def merge(steps, removed=()):
    """the full graph / initial values of a sequence; `removed`: indices (positions in the full factor list) left out"""
    graph, init = NonlinearFactorGraph(), Values()
    at = 0
    for g, v, *_ in steps:
        rec = [None] * g.size()
        for ftype, _, gi, keys, meas, _, models in g.buckets():
            for i, gidx in enumerate(gi.tolist()):
                rec[gidx] = (ftype, keys[i], meas[i], models[i])
        for ftype, keys, meas, model in rec:
            if at not in removed:
                graph._add(ftype, [keys], [meas], model)
            at += 1
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


NO_RELIN = dict(relinearizeThreshold=0.0, relinearizeSkip=0, enableRelinearization=False, wildfireThreshold=0.001)


def slamlike_isam():
    steps = slamlike_steps()
    isam = oh.OracleISAM2(**NO_RELIN)
    for g, v in steps:
        isam.update(g, v)
    return isam, steps


def test_remove_factors():
    """TEST(ISAM2, removeFactors) tests/testGaussianISAM2.cpp:380-400: remove the 2nd-to-last measurement of landmark 100 (index 12)"""
    isam, steps = slamlike_isam()
    n = isam.num_factors()
    isam.update(removeFactorIndices=[12])
    assert isam.unusedKeys() == [] and isam.num_factors() == n and not isam.factor_exists(12) and isam.factor_exists(11)
    isam_check(isam, *merge(steps, removed={12}))


def test_remove_variables():
    """TEST(ISAM2, removeVariables) :403-423: both measurements of landmark 100 go (indices 7 and 14) and the landmark with them"""
    isam, steps = slamlike_isam()
    isam.update(removeFactorIndices=[7, 14])
    assert isam.unusedKeys() == [100]
    fullgraph, fullinit = merge(steps, removed={7, 14})
    fullinit.erase(100)
    isam_check(isam, fullgraph, fullinit)
    assert 100 not in isam.getLinearizationPoint().keys() and 100 not in isam.getDelta()


def test_swap_factors():
    """TEST(ISAM2, swapFactors) :426-476: the 2nd-to-last factor is replaced by one with another range in the same update"""
    isam, steps = slamlike_isam()
    swap_idx = isam.num_factors() - 2
    swap = NonlinearFactorGraph()
    swap.add_BearingRangeFactor2D(10, 100, np.pi / 4.0 + np.pi / 16.0, 5.0, noiseModel.Diagonal.Sigmas([np.pi / 100.0, 0.1]))
    isam.update(swap, removeFactorIndices=[swap_idx])
    assert isam.unusedKeys() == [] and isam.num_factors() == swap_idx + 3 and not isam.factor_exists(swap_idx)
    isam_check(isam, *merge(steps + [(swap, Values())], removed={swap_idx}))


def test_constrained_ordering():
    """TEST(ISAM2, constrained_ordering) :479-571: constrainedKeys {3: 1, 4: 2} from the fourth odometry step on; the batch solution is
    reached, and the constrained variables end up last: x4 in the root clique (the largest group is eliminated last)"""
    steps = constrained_ordering_steps()
    isam = oh.OracleISAM2(**NO_RELIN)
    for g, v, c in steps:
        isam.update(g, v, constrainedKeys=c)
        roots = [keys[:nfk] for keys, nfk, _, par in isam.cliques() if par < 0]
        if c is not None:
            assert any(4 in r for r in roots), roots
    isam_check(isam, *merge(steps))


def test_extra_reelim_and_no_relin_keys():
    """extraReelimKeys re-eliminates the named variables' cliques (gatherInvolvedKeys, ISAM2-impl.h:210-215) without changing the solution;
    noRelinKeys keeps the linearization point of the named variables (gatherRelinearizeKeys, :388-392)"""
    steps = slamlike_steps()
    a = oh.OracleISAM2(relinearizeThreshold=0.0, relinearizeSkip=1, wildfireThreshold=0.001)
    b = oh.OracleISAM2(relinearizeThreshold=0.0, relinearizeSkip=1, wildfireThreshold=0.001)
    for g, v in steps[:-1]:
        a.update(g, v)
        b.update(g, v)
    g, v = steps[-1]
    ra = a.update(g, v)
    rb = b.update(g, v, noRelinKeys=[0, 1, 2])
    la, lb = a.getLinearizationPoint(), b.getLinearizationPoint()
    assert rb["variablesRelinearized"] < ra["variablesRelinearized"]
    for k in (0, 1, 2):
        assert not np.array_equal(la.at(k), lb.at(k))
    c = oh.OracleISAM2(**NO_RELIN)
    for g, v in steps:
        c.update(g, v)
    est = c.calculateEstimate()
    r = c.update(extraReelimKeys=[0])
    assert r["variablesReeliminated"] > 0
    est2 = c.calculateEstimate()
    for k in est.keys():
        assert np.allclose(est.at(k), est2.at(k), rtol=0, atol=1e-9)
    isam_check(c, *merge(steps))


def test_marginal_covariance():
    """TEST(ISAM2, marginalCovariance) tests/testGaussianISAM2.cpp:977-986: isam.marginalCovariance(5) equals
    Marginals(isam.getFactorsUnsafe(), isam.getLinearizationPoint()).marginalCovariance(5)"""
    isam, steps = slamlike_isam()
    fullgraph, _ = merge(steps)
    lin = isam.getLinearizationPoint()
    batch = oh.OracleProblem(fullgraph, lin, Ordering.Natural(fullgraph))
    for key in (5, 0, 11, 100):
        d = VAR_DIM[lin.type(key)]
        expected = batch.marginal_covariance(key, d)
        assert np.allclose(isam.marginalCovariance(key), expected, rtol=1e-9, atol=1e-12), key


def test_partial_relinearization_check_and_threshold_vectors():
    """TEST(ISAM2, slamlike_solution_partial_relinearization_check) tests/testGaussianISAM2.cpp:603-614 (the batch solution is reached with
    enablePartialRelinearizationCheck); with relinearization ON the partial check marks a subset of what the full check marks
    (CheckRelinearizationPartial stops below a clique without a variable above the threshold, ISAM2-impl.h:302-331), and per-character
    threshold vectors (FastMap<char, Vector>, :252-268 / :365-377) with all entries equal to the scalar mark the same variables as the
    scalar up to the > / >= difference -- here none sits exactly on the threshold"""
    steps = slamlike_steps()
    isam = oh.OracleISAM2(**NO_RELIN)
    isam.set_partial_relinearization_check(True)
    for g, v in steps:
        isam.update(g, v)
    isam_check(isam, *merge(steps))
    # a landmark whose large delta lies below cliques of unmoved poses when the first check comes (update 10): the full check
    # relinearizes it, the partial check stops at the root and leaves it alone
    full = oh.OracleISAM2(relinearizeThreshold=0.05, relinearizeSkip=10)
    part = oh.OracleISAM2(relinearizeThreshold=0.05, relinearizeSkip=10)
    part.set_partial_relinearization_check(True)
    cf = [full.update(g, v)["variablesRelinearized"] for g, v in stale_landmark_steps()]
    cp = [part.update(g, v)["variablesRelinearized"] for g, v in stale_landmark_steps()]
    assert cf[9] == 3 and cp[9] == 2 and cf[:9] == cp[:9] == [0] * 9
    assert np.abs(full.getDelta()[100]).max() < 0.1 < 0.4 < np.abs(part.getDelta()[100]).max()
    assert not np.array_equal(full.getLinearizationPoint().at(100), part.getLinearizationPoint().at(100))
    scalar = oh.OracleISAM2(relinearizeThreshold=0.02, relinearizeSkip=1)
    vec = oh.OracleISAM2(relinearizeThreshold=0.5, relinearizeSkip=1)
    vec.set_relinearize_thresholds({"x": [0.02] * 6, "l": [0.02] * 3})
    for g, v in visual_steps():
        assert scalar.update(g, v) == vec.update(g, v)
    # a vector of the wrong dimension is refused like the reference's invalid_argument
    bad = oh.OracleISAM2(relinearizeThreshold=0.02, relinearizeSkip=1)
    bad.set_relinearize_thresholds({"x": [0.02] * 6, "l": [0.02] * 2})
    vs = visual_steps()
    with pytest.raises(AssertionError):
        for g, v in vs:
            bad.update(g, v)


def test_evaluate_nonlinear_error():
    """ISAM2Params::evaluateNonlinearError: errorAfter of every update = the nonlinear error of the full graph at calculateEstimate()
    (evaluated independently through the batch oracle), errorBefore = the same graph at the estimate before the update's elimination;
    after a removal the removed factor no longer counts (ISAM2.cpp:444-446, 481-483)"""
    steps = slamlike_steps()
    isam = oh.OracleISAM2(relinearizeThreshold=0.01, relinearizeSkip=1)
    isam.set_evaluate_nonlinear_error(True)
    for n, (g, v) in enumerate(steps):
        isam.update(g, v)
        before, after = isam.errors()
        fullgraph, _ = merge(steps[:n + 1])
        est = isam.calculateEstimate()
        batch = oh.OracleProblem(fullgraph, est, Ordering.Natural(fullgraph))
        assert abs(after - batch.error()) <= 1e-12 * max(1.0, batch.error()), n
        assert abs(isam.error(0) - after) <= 1e-15 * max(1.0, after) and before >= after - 1e-9
    isam.update(removeFactorIndices=[12])
    fullgraph, _ = merge(steps, removed={12})
    batch = oh.OracleProblem(fullgraph, isam.calculateEstimate(), Ordering.Natural(fullgraph))
    assert abs(isam.errors()[1] - batch.error()) <= 1e-12 * max(1.0, batch.error())
    lin = oh.OracleProblem(fullgraph, isam.getLinearizationPoint(), Ordering.Natural(fullgraph))
    assert abs(isam.error(2) - lin.error()) <= 1e-12 * max(1.0, lin.error())


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_slamlike_solution_dogleg(mode):
    """TEST(ISAM2, slamlike_solution_dogleg) tests/testGaussianISAM2.cpp:310-319: ISAM2Params(ISAM2DoglegParams(1.0), 0.0, 0, false)
    (adaptation mode SEARCH_EACH_ITERATION, the default) reaches the batch solution (isam_check).  The other two adaptation modes only
    shrink / take one trust-region step per update: they must lower the error of the graph at the linearization point"""
    steps = slamlike_steps()
    isam = oh.OracleISAM2(**NO_RELIN)
    isam.set_dogleg(1.0, 1e-5, mode)
    for g, v in steps:
        isam.update(g, v)
    if mode == 0:
        isam_check(isam, *merge(steps))
        assert isam.doglegDelta() >= 1.0
    else:
        assert isam.error(0) < isam.error(2)


def test_dogleg_trust_region_limits_the_step():
    """a small initial trust region: the first estimates are NOT the Newton point (the dog leg cuts the step), the radius adapts, and the
    sequence still ends near the batch solution of the visual example"""
    dl = oh.OracleISAM2(relinearizeThreshold=0.01, relinearizeSkip=1)
    dl.set_dogleg(0.05, 1e-5, 2)  # ONE_STEP_PER_ITERATION: one trust-region step per updateDelta (SEARCH_EACH_ITERATION grows the radius
    #                               inside the call until the Newton point fits, which here reproduces Gauss-Newton)
    gn = oh.OracleISAM2(relinearizeThreshold=0.01, relinearizeSkip=1)
    radii, differs = [], False
    for g, v in visual_steps():
        dl.update(g, v)
        gn.update(g, v)
        radii.append(dl.doglegDelta())
        a, b = dl.calculateEstimate(), gn.calculateEstimate()
        differs = differs or any(not np.allclose(a.at(k), b.at(k), atol=1e-6) for k in b.keys())
    assert differs and radii[0] == 0.05 and radii[-1] > 5.0 and sorted(radii) == radii
    for j, p in enumerate(create_points()):
        assert np.allclose(dl.calculateEstimate().at(symbol("l", j)), p, rtol=0, atol=0.05), j
