"""Marginals on the device path (lmgpu_marginal_covariance: per column one elimination + back-substitution of the linearized
system with a unit gradient) against the oracle's restatement of Marginals::marginalCovariance (dense inverse of the
information matrix, pinned in tests/test_oracle_golden.py to the reference's own output for the odometry example)."""
import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import (LevenbergMarquardtOptimizer, LevenbergMarquardtParams, Marginals, NonlinearFactorGraph, Ordering, Values, _lib,
                                noiseModel)
from gtsam_personal_amd.synthetic import make_bal
from test_oracle_golden import (ODOMETRY_EXACT, ODOMETRY_PRINTED, PLANAR_SLAM_EXPECTED, PLANAR_SLAM_JOINT_L2_X1_X3, _odometry_example,
                                _planar_slam_example)

pytestmark = pytest.mark.gpu


def test_odometry_example_known_answers_on_gpu():
    """doc/Code/OdometryMarginals.cpp / OdometryOutput3.txt: x1, x2, x3 covariances as the reference prints them"""
    graph, values = _odometry_example()
    for ordering in ([1, 2, 3], [3, 1, 2]):
        m = Marginals(graph, values, Ordering(ordering))
        for k in (1, 2, 3):
            cov = m.marginalCovariance(k)
            assert np.allclose(cov, np.array(ODOMETRY_PRINTED[k]), atol=6e-3), (k, cov)
            assert np.allclose(cov, np.array(ODOMETRY_EXACT[k]), rtol=1e-9, atol=1e-12), (k, cov)
            assert np.allclose(m.marginalInformation(k) @ cov, np.eye(3), atol=1e-9)
        m.close()


def test_bal_marginals_match_oracle_after_optimization():
    """cameras (9 x 9, in the HBM root) and points (3 x 3, LDS leaf fronts gathered in Schur form) of an optimized BAL problem;
    the optimizer's handle keeps working after the marginals consumed its linearization"""
    graph, initial, _, ordering = make_bal(n_cam=24, n_pt=400, obs_per_point=6, seed=9)
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
    opt.optimize()
    result = opt.values()
    orc = oh.OracleProblem(graph, result, ordering)
    m = Marginals(graph, result, ordering)
    keys = list(ordering)
    for k in [keys[0], keys[7], keys[399], keys[400], keys[411], keys[-1]]:
        d = 9 if k in keys[400:] else 3
        ref = orc.marginal_covariance(k, d)
        cov = m.marginalCovariance(k)
        assert ref is not None and cov.shape == (d, d)
        assert np.allclose(cov, cov.T, rtol=0, atol=1e-14 * np.abs(cov).max())
        assert np.linalg.norm(cov - ref) <= 1e-6 * np.linalg.norm(ref), (k, np.linalg.norm(cov - ref) / np.linalg.norm(ref))
    # the same handle afterwards: a plain solve needs (and gets) a fresh linearization
    h = m._opt
    h.linearize()
    dk, d, e0, e1 = h.solve(1e-3)
    orc.linearize()
    rc, do, o0, o1 = orc.solve(1e-3)
    assert rc == 0 and abs(e1 - o1) <= 1e-6 * max(1.0, abs(o1))
    m.close()


def test_pose3_graph_marginals_match_oracle_colamd():
    """first 300 poses of sphere2500 (Pose3 between factors + prior), COLAMD ordering when the reference build is present:
    LDS, medium and HBM fronts with separators all carry the unit right-hand side"""
    import os
    from gtsam_personal_amd.datasets import chain_initial_pose3, load3D
    graph, _ = load3D(os.path.join(os.path.dirname(__file__), "golden", "sphere2500_head.txt"))
    vals = chain_initial_pose3(graph)
    graph.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Natural(graph)
    orc = oh.OracleProblem(graph, vals, ordering)
    m = Marginals(graph, vals, ordering)
    keys = sorted(int(k) for k in ordering)
    for k in (keys[0], keys[57], keys[150], keys[-1]):
        ref, cov = orc.marginal_covariance(k, 6), m.marginalCovariance(k)
        assert ref is not None
        assert np.linalg.norm(cov - ref) <= 1e-6 * np.linalg.norm(ref), (k, np.linalg.norm(cov - ref) / np.linalg.norm(ref))
    m.close()


def test_gauge_freedom_reports_indeterminate():
    """two poses tied by one odometry factor and no prior: the information matrix is singular (here exactly: unit noise and
    axis-aligned poses make every Jacobian entry a small integer, so the second pivot block is exactly zero); the reference's
    Marginals throws IndeterminantLinearSystemException from the elimination in its constructor"""
    graph, values = NonlinearFactorGraph(), Values()
    graph.add_BetweenFactorPose2(1, 2, [2.0, 0.0, 0.0], noiseModel.Diagonal.Sigmas([1.0, 1.0, 1.0]))
    values.insert_pose2(1, 0.0, 0.0, 0.0)
    values.insert_pose2(2, 2.0, 0.0, 0.0)
    m = Marginals(graph, values, Ordering([1, 2]))
    with pytest.raises(_lib.IndeterminantLinearSystemException):
        m.marginalCovariance(2)
    m.close()


def test_planar_slam_marginals_known_answers_on_gpu():
    """tests/testMarginals.cpp:40-126: the five marginal covariances of the planar SLAM example (BearingRangeFactor<Pose2, Point2>,
    Point2 landmarks), asserted by the reference to 1e-8"""
    graph, soln, keys = _planar_slam_example()
    for ordering in ([keys[3], keys[4], keys[0], keys[1], keys[2]], list(keys)):
        m = Marginals(graph, soln, Ordering(ordering))
        for k, expected in zip(keys, PLANAR_SLAM_EXPECTED):
            cov = m.marginalCovariance(k)
            assert np.allclose(cov, np.array(expected), rtol=0, atol=1e-8), (k, cov)
        m.close()


def test_planar_slam_lm_matches_oracle():
    """the same graph from a perturbed start: linearize / solve / LM trajectory of the bearing-range factors against the oracle"""
    from test_gpu_parity import _check_linearize, _check_lm, _check_solve, _pair
    graph, soln, keys = _planar_slam_example()
    rng = np.random.default_rng(4)
    initial = Values()
    for k in keys[:3]:
        initial.insert_pose2(k, *(soln.at(k) + rng.normal(0, [0.2, 0.2, 0.1])))
    for k in keys[3:]:
        initial.insert_point2(k, soln.at(k) + rng.normal(0, 0.3, 2))
    opt, orc, params = _pair(graph, initial, Ordering([keys[3], keys[4], keys[0], keys[1], keys[2]]))
    assert abs(opt.graph_error() - orc.error()) <= 1e-10 * orc.error()
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-3)
    _check_lm(opt, orc, params)
    assert opt.error() < 1e-10


def test_planar_slam_joint_marginals_known_answers_on_gpu():
    """tests/testMarginals.cpp:128-176: jointMarginalCovariance of (x1, l2, x3) and of (l2, x1): blocks in Key order (l2, x1, x3),
    asserted by the reference to 1e-6; the single-variable joint equals the marginal"""
    graph, soln, keys = _planar_slam_example()
    x1, x2, x3, l1, l2 = keys
    m = Marginals(graph, soln, Ordering([l1, l2, x1, x2, x3]))
    joint = m.jointMarginalCovariance([x1, l2, x3])
    assert joint.keys == [l2, x1, x3] and joint.dims == [2, 3, 3]
    assert np.allclose(joint.fullMatrix(), PLANAR_SLAM_JOINT_L2_X1_X3, rtol=0, atol=1e-6)
    assert np.allclose(joint(l2, l2), PLANAR_SLAM_JOINT_L2_X1_X3[0:2, 0:2], atol=1e-6)
    assert np.allclose(joint(x1, l2), PLANAR_SLAM_JOINT_L2_X1_X3[2:5, 0:2], atol=1e-6)
    assert np.allclose(joint(x3, x1), PLANAR_SLAM_JOINT_L2_X1_X3[5:8, 2:5], atol=1e-6)
    joint2 = m.jointMarginalCovariance([l2, x1])
    assert np.allclose(joint2.fullMatrix(), PLANAR_SLAM_JOINT_L2_X1_X3[0:5, 0:5], atol=1e-6)
    joint1 = m.jointMarginalCovariance([x1])
    assert np.allclose(joint1(x1, x1), m.marginalCovariance(x1), atol=1e-12)
    # consistency with the single-variable marginals and with the information form
    assert np.allclose(joint(x3, x3), m.marginalCovariance(x3), atol=1e-12)
    assert np.allclose(m.jointMarginalInformation([l2, x1]).fullMatrix() @ joint2.fullMatrix(), np.eye(5), atol=1e-9)
    m.close()
