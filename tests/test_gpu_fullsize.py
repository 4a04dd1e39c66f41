"""Configs C1 and C3 at the reference's own size (examples/Data/city10000.g2o: 10 000 Pose2 / 20 687 EDGE_SE2;
examples/Data/sphere2500.txt: 2 500 Pose3 / 4 949 EDGE3), COLAMD and METIS orderings computed by the reference's own
C sources (oracle/_ref).  General sparse graphs: deep clique trees, LDS fronts with children, HBM fronts that have HBM
and non-leaf children and wide separators -- the parts of the product the BAL workload does not reach.
Tolerances as in test_gpu_parity.py (1e-9 errors / Jacobians, 1e-6 delta, [R S d], cost)."""
import os

import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams, noiseModel
from gtsam_personal_amd.datasets import chain_initial_pose3, load2D, load3D, readG2o

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(1e-300, np.linalg.norm(b)))


def _compare(graph, initial, ordering, lam, n_iter, sample_jacobians):
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(params)
    assert abs(opt.graph_error() - orc.error()) <= 1e-9 * max(1.0, orc.error())
    opt.linearize()
    orc.linearize()
    for g in sample_jacobians:
        J, Jo = opt.jacobian(g), orc.jacobian(g)
        assert np.allclose(J, Jo, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(Jo).max())), g
    dk, _, e0, e1 = opt.solve(lam)
    rc, do, o0, o1 = orc.solve(lam)
    assert rc == 0
    assert abs(e0 - o0) <= 1e-9 * max(1.0, abs(o0))
    assert abs(e1 - o1) <= 1e-6 * max(1.0, abs(o1))
    a = np.concatenate([dk[k] for k in sorted(dk)])
    b = np.concatenate([do[k] for k in sorted(do)])
    assert _rel(a, b) < 1e-6, _rel(a, b)
    cl = orc.cliques()
    assert opt.num_fronts() == len(cl)
    classes = set()
    for i, (keys, nfk, rsd, parent) in enumerate(cl):  # every clique: same keys in the same order, same [R S d]
        fk, R = opt.front(i)
        assert fk == keys
        assert np.allclose(R, rsd, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(rsd).max())), i
        classes.add(opt.front_info(i)["cls"])
    for _ in range(n_iter):
        opt.iterate()
        orc.lm_iterate(params)
        so = orc.lm_state()
        assert opt.getInnerIterations() == so["inner"]
        assert abs(opt.error() - so["error"]) <= 1e-6 * max(1e-12, abs(so["error"]))
        assert abs(opt.lambda_() - so["lambda_"]) <= 1e-9 * so["lambda_"]
    return classes


@pytest.mark.parametrize("which", ["colamd", "metis"])
def test_sphere2500_full(which):
    if not oh.have_ref():
        pytest.skip("oracle/_ref (CCOLAMD / METIS built from the reference's C sources) not present")
    graph, _ = load3D(os.path.join(GOLD, "sphere2500.txt"))
    initial = chain_initial_pose3(graph)
    graph.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
    ordering = oh.colamd(graph) if which == "colamd" else oh.metis(graph)
    classes = _compare(graph, initial, ordering, 1e-5, 2, range(0, graph.size(), 97))
    assert classes == {0, 1}  # both LDS and HBM fronts occur


@pytest.mark.parametrize("which", ["colamd", "metis"])
def test_city10000_full(which):
    if not oh.have_ref():
        pytest.skip("oracle/_ref not present")
    graph, initial = readG2o(os.path.join(GOLD, "city10000.g2o"))
    graph.add_PriorFactorPose2(0, initial.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))  # Pose2SLAMExample_g2o.cpp:60-66
    ordering = oh.colamd(graph) if which == "colamd" else oh.metis(graph)
    _compare(graph, initial, ordering, 1e-5, 2, range(0, graph.size(), 211))


@pytest.mark.parametrize("which", ["colamd", "metis"])
def test_victoria_park_full(which):
    """examples/Data/victoria_park.txt at full size (6 969 Pose2, 151 Point2 landmarks, 6 968 odometry and 3 640
    BearingRangeFactor<Pose2, Point2> built from its LANDMARK lines, gtsam/slam/tests/testDataset.cpp:131-145): the planar
    landmark SLAM shape -- long odometry chain, landmarks observed from hundreds of poses each."""
    if not oh.have_ref():
        pytest.skip("oracle/_ref not present")
    graph, initial = load2D(os.path.join(GOLD, "victoria_park.txt"))
    assert graph.size() == 10608 and initial.size() == 7120
    graph.add_PriorFactorPose2(0, initial.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
    ordering = oh.colamd(graph) if which == "colamd" else oh.metis(graph)
    _compare(graph, initial, ordering, 1e-5, 2, range(0, graph.size(), 173))


@pytest.mark.parametrize("name", ["sphere2500", "city10000"])
def test_general_sparse_solves_are_bitwise_reproducible(name):
    """every front is assembled in a fixed order (Schur gather for leaf children, one wave per front row for update matrices and own
    factors: no FP64 atomics anywhere on the default path), so two solves of the same linearization give bitwise the same update
    vector and the same [R S d] of the root -- on the general sparse graphs too, not only on BAL"""
    if name == "sphere2500":
        graph, _ = load3D(os.path.join(GOLD, "sphere2500.txt"))
        initial = chain_initial_pose3(graph)
        graph.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
    else:
        graph, initial = readG2o(os.path.join(GOLD, "city10000.g2o"))
        graph.add_PriorFactorPose2(0, initial.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
    ordering = oh.colamd(graph)
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
    opt.linearize()
    runs = []
    for _ in range(3):
        _, d, e0, e1 = opt.solve(1e-3)
        _, R = opt.front(opt.num_fronts() - 1)
        runs.append((d.copy(), R.copy(), e0, e1))
    for d, R, e0, e1 in runs[1:]:
        assert np.array_equal(d, runs[0][0]) and np.array_equal(R, runs[0][1]) and e0 == runs[0][2] and e1 == runs[0][3]
    opt.close()
