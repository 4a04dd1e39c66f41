"""Edge cases of the LM path on the device against the oracle (and the reference's stated behaviour): a forest (disconnected
graph: several Bayes-tree roots, gtsam/inference/EliminationTree-inst.h:127-134), a single variable, a problem that starts at its
optimum (LevenbergMarquardtOptimizer.cpp:121-308 / NonlinearOptimizer.cpp:70-126: the first successful step ends the loop on the
error tolerances), an under-constrained system (IndeterminantLinearSystemException: LM raises lambda and goes on, :302-305;
Gauss-Newton lets it through), a maximum-lambda exit, and inputs the reference rejects (ordering / values that do not match the
graph: EliminationTree-inst.h:99-102, Values::at -> ValuesKeyDoesNotExist)."""
import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import (GaussNewtonOptimizer, GaussNewtonParams, LevenbergMarquardtOptimizer, LevenbergMarquardtParams, NonlinearFactorGraph,
                                Ordering, Values, _lib, noiseModel)
from test_gpu_parity import _check_lm, _check_linearize, _check_solve, _pair

pytestmark = pytest.mark.gpu
ODO = noiseModel.Diagonal.Sigmas([0.2, 0.2, 0.1])


def chain(graph, initial, first, n, prior=True, rng=None):
    rng = rng or np.random.default_rng(first)
    if prior:
        graph.add_PriorFactorPose2(first, [float(first), 0.0, 0.0], noiseModel.Diagonal.Sigmas([0.3, 0.3, 0.1]))
    for i in range(n):
        k = first + i
        initial.insert_pose2(k, first + 2.0 * i + rng.normal(0, 0.2), rng.normal(0, 0.2), rng.normal(0, 0.1))
        if i:
            graph.add_BetweenFactorPose2(k - 1, k, [2.0, 0.0, 0.0], ODO)


def test_forest_of_three_components():
    graph, initial = NonlinearFactorGraph(), Values()
    chain(graph, initial, 0, 7)
    chain(graph, initial, 100, 4)
    chain(graph, initial, 200, 1)  # a lone variable with its prior
    for ordering in (Ordering.Natural(graph), sorted(initial.keys(), reverse=True)):
        opt, orc, params = _pair(graph, initial, ordering)
        _check_linearize(opt, orc, graph)
        _check_solve(opt, orc, 1e-5)
        roots = [i for i, (_, _, _, par) in enumerate(orc.cliques()) if par < 0]
        assert len(roots) == 3
        _check_lm(opt, orc, params)
        opt.close()


def test_single_variable_with_a_prior():
    graph, initial = NonlinearFactorGraph(), Values()
    chain(graph, initial, 5, 1)
    opt, orc, params = _pair(graph, initial, [5])
    with pytest.raises(_lib.LmgpuError, match="no linearization"):
        opt.solve(0.0, False)
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 0.0)
    _check_lm(opt, orc, params)
    assert opt.error() < 1e-20
    opt.close()


def test_start_at_the_optimum():
    graph, initial = NonlinearFactorGraph(), Values()
    graph.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], ODO)
    for i in range(6):
        initial.insert_pose2(i, 2.0 * i, 0.0, 0.0)
        if i:
            graph.add_BetweenFactorPose2(i - 1, i, [2.0, 0.0, 0.0], ODO)
    opt, orc, params = _pair(graph, initial, Ordering.Natural(graph))
    assert opt.graph_error() == 0.0
    _check_lm(opt, orc, params)
    assert opt.error() == 0.0
    opt.close()


def test_gauge_freedom_lm_raises_lambda_and_gauss_newton_throws():
    graph, initial = NonlinearFactorGraph(), Values()
    chain(graph, initial, 0, 6, prior=False)
    ordering = Ordering.Natural(graph)
    # lambda = 0: the reference's Cholesky refuses the last pivot (gtsam/base/cholesky.cpp:146-158)
    opt, orc, params = _pair(graph, initial, ordering)
    opt.linearize()
    orc.linearize()
    with pytest.raises(_lib.IndeterminantLinearSystemException):
        opt.solve(0.0, False)
    assert orc.solve(0.0, False)[0] != 0
    opt.close()
    gn = GaussNewtonOptimizer(graph, initial, ordering, GaussNewtonParams(), device=0)
    with pytest.raises(_lib.IndeterminantLinearSystemException):
        gn.optimize()
    gn.close()
    # LM: the damped system is solvable; same trajectory as the oracle
    opt, orc, params = _pair(graph, initial, ordering)
    _check_lm(opt, orc, params, check_values=True)
    opt.close()


def test_lambda_upper_bound_exit():
    """a ring of poses started far from the solution with a small lambdaUpperBound: every try fails and the search for a successful step
    gives up at the bound (LevenbergMarquardtOptimizer.cpp:290-298), values untouched, like the oracle's"""
    rng = np.random.default_rng(0)
    graph, initial = NonlinearFactorGraph(), Values()
    graph.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], ODO)
    n = 8
    for i in range(n):
        initial.insert_pose2(i, rng.normal(0, 5), rng.normal(0, 5), rng.uniform(-3, 3))
        graph.add_BetweenFactorPose2(i, (i + 1) % n, [2.0, 0.0, 2 * np.pi / n], ODO)
    params = LevenbergMarquardtParams()
    params.lambdaInitial = 1e-5
    params.lambdaUpperBound = 1e-3
    opt, orc, params = _pair(graph, initial, Ordering.Natural(graph), params)
    e0 = opt.graph_error()
    _check_lm(opt, orc, params)
    assert opt.lambda_() >= params.lambdaUpperBound and opt.getInnerIterations() == 2 and opt.error() == e0
    opt.close()


def test_inputs_the_reference_rejects():
    graph, initial = NonlinearFactorGraph(), Values()
    chain(graph, initial, 0, 4)
    with pytest.raises(Exception):  # ordering without one of the variables
        LevenbergMarquardtOptimizer(graph, initial, [0, 1, 2], LevenbergMarquardtParams(), device=0)
    with pytest.raises(Exception):  # a variable twice
        LevenbergMarquardtOptimizer(graph, initial, [0, 1, 2, 2], LevenbergMarquardtParams(), device=0)
    with pytest.raises(Exception):  # a key the graph does not know
        LevenbergMarquardtOptimizer(graph, initial, [0, 1, 2, 3, 9], LevenbergMarquardtParams(), device=0)
    short = Values()
    for k in (0, 1, 2):
        short.insert(k, initial.type(k), initial.at(k))
    with pytest.raises(Exception):  # a factor on a variable without a value
        LevenbergMarquardtOptimizer(graph, short, [0, 1, 2, 3], LevenbergMarquardtParams(), device=0)
    extra = initial.copy()
    extra.insert_pose2(77, 0.0, 0.0, 0.0)
    with pytest.raises(Exception):  # a variable no factor touches: the elimination tree has no place for it
        LevenbergMarquardtOptimizer(graph, extra, [0, 1, 2, 3, 77], LevenbergMarquardtParams(), device=0).optimize()


def test_dogleg_is_bitwise_reproducible():
    """the Bayes tree's gradient (the steepest-descent half of the dog leg) is a fixed-order sum since round 3 (FP64 atomics before): two runs
    of DoglegOptimizer take bitwise the same steps -- batch path and the incremental one (ISAM2DoglegParams)"""
    from gtsam_personal_amd import DoglegOptimizer, DoglegParams
    from gtsam_personal_amd.synthetic import make_bal
    graph, initial, _, ordering = make_bal(n_cam=24, n_pt=400, obs_per_point=5, seed=17)
    runs = []
    for _ in range(2):
        p = DoglegParams()
        p.deltaInitial = 0.5  # small enough that the first steps are cut by the trust region (the gradient matters)
        opt = DoglegOptimizer(graph, initial, ordering, p, device=0)
        trace = []
        for _ in range(4):
            opt.iterate()
            trace.append((opt.error(), opt.getDelta()))
        vals = opt.values()
        runs.append((trace, np.concatenate([vals.at(k) for k in ordering])))
        opt.close()
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1])
