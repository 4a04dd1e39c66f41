"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs (sizes the oracle finishes in seconds), against the committed golden fixtures, and — at larger sizes —
through size-independent properties.  Tolerances: Jacobians / errors 1e-9 relative; delta, final cost 1e-6
relative (BASELINE.json north_star); variable ordering / front indexing bit-exact."""
import os

import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import (LevenbergMarquardtOptimizer, LevenbergMarquardtParams, NonlinearFactorGraph, Ordering, Values, _lib, noiseModel)
from gtsam_personal_amd.datasets import SfmData, bal_graph, chain_initial_pose3, load2D, load3D, readG2o
from gtsam_personal_amd.synthetic import make_bal

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.linalg.norm(a - b) / max(1e-300, np.linalg.norm(b)))


def _pair(graph, initial, ordering, params=None):
    params = params or LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(params)
    return opt, orc, params


def _check_linearize(opt, orc, graph):
    opt.linearize()
    orc.linearize()
    for g in range(graph.size()):
        J, Jo = opt.jacobian(g), orc.jacobian(g)
        assert J.shape == Jo.shape
        assert np.allclose(J, Jo, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(Jo).max())), g


def _check_solve(opt, orc, lam, diagonal=False, tol=1e-6):
    dk, _, e0, e1 = opt.solve(lam, diagonal)
    rc, do, o0, o1 = orc.solve(lam, diagonal)
    assert rc == 0
    assert abs(e0 - o0) <= 1e-9 * max(1.0, abs(o0))
    assert abs(e1 - o1) <= 1e-6 * max(1.0, abs(o1))
    a = np.concatenate([dk[k] for k in sorted(dk)])
    b = np.concatenate([do[k] for k in sorted(do)])
    assert rel(a, b) < tol, rel(a, b)
    # fronts: same keys, same [R S d]
    cl = orc.cliques()
    assert opt.num_fronts() == len(cl)
    for i, (keys, nfk, rsd, parent) in enumerate(cl):
        fk, R = opt.front(i)
        assert fk == keys
        assert R.shape == rsd.shape
        assert np.allclose(R, rsd, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(rsd).max())), i
    return dk


def _check_lm(opt, orc, params, tol=1e-6, check_values=True):
    opt.optimize()
    orc.lm_optimize(params)
    so = orc.lm_state()
    assert opt.iterations() == so["iterations"]
    assert opt.getInnerIterations() == so["inner"]
    assert abs(opt.error() - so["error"]) <= tol * max(1e-12, abs(so["error"])) + 1e-12
    assert abs(opt.lambda_() - so["lambda_"]) <= 1e-9 * so["lambda_"]
    if check_values:
        vo, vg = orc.values(), opt.values()
        for k in vo:
            assert np.allclose(vo[k], vg.at(k), rtol=1e-6, atol=1e-7), k


def test_error_and_linearize_sfm():
    graph, initial, _, ordering = make_bal(n_cam=8, n_pt=60, obs_per_point=4, seed=11)
    opt, orc, _ = _pair(graph, initial, ordering)
    assert abs(opt.graph_error() - orc.error()) <= 1e-10 * orc.error()
    _check_linearize(opt, orc, graph)
    hd, ho = opt.hessian_diagonal(), orc.hessian_diagonal()
    for k in ho:
        assert np.allclose(hd[k], ho[k], rtol=1e-10)


def test_solve_sfm_lds_fronts_and_hbm_root():
    # 20 cameras -> root front 181 x 181 lives in HBM (MFMA path); point fronts (n <= 40) in LDS
    graph, initial, _, ordering = make_bal(n_cam=20, n_pt=300, obs_per_point=6, seed=5)
    opt, orc, _ = _pair(graph, initial, ordering)
    _check_linearize(opt, orc, graph)
    info = opt.front_info(opt.num_fronts() - 1)
    assert info["cls"] == 1 and info["n"] == 181
    _check_solve(opt, orc, 1e-5)
    _check_solve(opt, orc, 1e-2, diagonal=True)


def test_retract_matches_oracle():
    graph, initial, _, ordering = make_bal(n_cam=6, n_pt=40, obs_per_point=4, seed=9)
    opt, orc, _ = _pair(graph, initial, ordering)
    opt.linearize()
    orc.linearize()
    dk = _check_solve(opt, orc, 1e-3)
    opt.retract()
    orc.retract({k: dk[k] for k in dk})
    vo, vg = orc.values(), opt.values()
    for k in vo:
        assert np.allclose(vo[k], vg.at(k), rtol=1e-9, atol=1e-10), k
    assert abs(opt.graph_error() - orc.error()) <= 1e-8 * max(1.0, orc.error())


def test_lm_optimize_synthetic_bal():
    graph, initial, _, ordering = make_bal(n_cam=10, n_pt=120, obs_per_point=5, seed=21)
    opt, orc, params = _pair(graph, initial, ordering)
    _check_lm(opt, orc, params)


def test_lm_optimize_ceres_params_diagonal_damping():
    graph, initial, _, ordering = make_bal(n_cam=6, n_pt=50, obs_per_point=4, seed=4)
    opt, orc, params = _pair(graph, initial, ordering, LevenbergMarquardtParams.CeresDefaults())
    _check_lm(opt, orc, params)


def test_golden_dubrovnik_error():
    """tests/testGeneralSFMFactorB.cpp:44-63: final error 0.0199833 +- 1e-5, on the GPU path"""
    db = SfmData.FromBalFile(os.path.join(GOLD, "dubrovnik-3-7-pre.txt"))
    graph, initial = bal_graph(db)
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Schur(graph, initial)
    opt, orc, params = _pair(graph, initial, ordering)
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, params.lambdaInitial)
    opt.optimize()
    assert abs(opt.error() - 0.0199833) < 1e-5
    assert abs(opt.graph_error() - 0.0199833) < 1e-5
    orc.lm_optimize(params)
    assert opt.getInnerIterations() == orc.lm_state()["inner"]


def test_cheirality_zeroes_factor():
    """a point behind a camera gives a zero Jacobian / zero b and zero error (GeneralSFMFactor.h:153-157)"""
    graph, initial, _, ordering = make_bal(n_cam=4, n_pt=10, obs_per_point=2, seed=8)
    fk = graph.factor_keys_in_graph_order()
    cam, pt = fk[0]
    c = initial.at(cam)
    R, t = c[:9].reshape(3, 3), c[9:12]
    initial.update(pt, t - 5.0 * R[:, 2])  # behind the camera along -z_c
    opt, orc, _ = _pair(graph, initial, ordering)
    assert abs(opt.graph_error() - orc.error()) <= 1e-10 * orc.error()
    opt.linearize()
    assert np.all(opt.jacobian(0) == 0.0)
    _check_linearize(opt, orc, graph)


def _pose2_graph(n=40, seed=0, loops=12):
    rng = np.random.default_rng(seed)
    graph, initial = NonlinearFactorGraph(), Values()
    th, x, y = 0.0, 0.0, 0.0
    poses = []
    for i in range(n):
        poses.append((x, y, th))
        initial.insert_pose2(i, x + rng.normal(0, 0.1), y + rng.normal(0, 0.1), th + rng.normal(0, 0.05))
        th += 2 * np.pi / n
        x += np.cos(th)
        y += np.sin(th)

    def between(a, b):
        xa, ya, ta = poses[a]
        xb, yb, tb = poses[b]
        c, s = np.cos(ta), np.sin(ta)
        dx, dy = xb - xa, yb - ya
        return [c * dx + s * dy, -s * dx + c * dy, np.arctan2(np.sin(tb - ta), np.cos(tb - ta))]
    m_odo = noiseModel.Diagonal.Sigmas([0.2, 0.2, 0.1])
    info = np.array([[40.0, 2.0, 1.0], [2.0, 30.0, 0.5], [1.0, 0.5, 90.0]])
    m_loop = noiseModel.Gaussian.Information(info)
    for i in range(n - 1):
        graph.add_BetweenFactorPose2(i, i + 1, between(i, i + 1), m_odo)
    for _ in range(loops):
        a, b = sorted(rng.choice(n, 2, replace=False).tolist())
        graph.add_BetweenFactorPose2(a, b, between(a, b), m_loop)
    graph.add_PriorFactorPose2(0, list(poses[0]), noiseModel.Diagonal.Sigmas([0.01, 0.01, 0.01]))
    return graph, initial


def test_pose2_slam_between_prior_gaussian_noise():
    graph, initial = _pose2_graph()
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Natural(graph)
    opt, orc, params = _pair(graph, initial, ordering)
    assert abs(opt.graph_error() - orc.error()) <= 1e-10 * orc.error()
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-4)
    _check_lm(opt, orc, params)


def test_hbm_front_with_odd_dimensions_and_separator():
    """a dense Pose2 cluster (61 poses, all pairs) hanging off a chain: the cluster's front is 184 x 184 with an ODD
    frontal dimension, so the MFMA trailing update runs with misaligned tile origins, a partial last panel and a
    non-empty separator; everything is compared with the oracle"""
    rng = np.random.default_rng(3)
    graph, initial = NonlinearFactorGraph(), Values()
    n_dense, n_chain = 62, 6
    for i in range(n_dense + n_chain):
        initial.insert_pose2(i, rng.normal(0, 2.0), rng.normal(0, 2.0), rng.normal(0, 0.5))
    model = noiseModel.Diagonal.Sigmas([0.3, 0.3, 0.1])
    truth = {i: np.array([np.cos(i * 0.1) * 5, np.sin(i * 0.1) * 5, i * 0.05]) for i in range(n_dense + n_chain)}

    def between(a, b):
        xa, ya, ta = truth[a]
        xb, yb, tb = truth[b]
        c, s = np.cos(ta), np.sin(ta)
        return [c * (xb - xa) + s * (yb - ya), -s * (xb - xa) + c * (yb - ya), tb - ta]
    for a in range(n_dense):
        for b in range(a + 1, n_dense):
            graph.add_BetweenFactorPose2(a, b, between(a, b), model)
    for i in range(n_dense - 1, n_dense + n_chain - 1):
        graph.add_BetweenFactorPose2(i, i + 1, between(i, i + 1), model)
    graph.add_PriorFactorPose2(n_dense + n_chain - 1, list(truth[n_dense + n_chain - 1]), noiseModel.Diagonal.Sigmas([0.05, 0.05, 0.02]))
    # eliminate the dense cluster first (its separator is pose n_dense-1 ... no: pose 60 is in the cluster; order cluster minus one first)
    ordering = Ordering(list(range(0, n_dense - 1)) + list(range(n_dense - 1, n_dense + n_chain)))
    opt, orc, params = _pair(graph, initial, ordering)
    big = max(range(opt.num_fronts()), key=lambda i: opt.front_info(i)["n"])
    info = opt.front_info(big)
    assert info["cls"] == 1 and info["nf"] == 183 and info["n"] == 187
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-3)
    _check_lm(opt, orc, params)


def _two_level_dense_pose2(n_a, n_b1, n_b2, seed=5):
    """dense cluster A (all pairs) whose poses also see every pose of B1; B = B1 u B2 is a dense cluster of its own.  Ordered
    A, B1, B2 the clique of A has B1 as its separator and is NOT merged into its parent (B2 is missing from the separator)."""
    rng = np.random.default_rng(seed)
    n = n_a + n_b1 + n_b2
    truth = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(-3, 3, n)], axis=1)
    graph, initial = NonlinearFactorGraph(), Values()
    for i in range(n):
        initial.insert_pose2(i, truth[i, 0] + rng.normal(0, 0.05), truth[i, 1] + rng.normal(0, 0.05), truth[i, 2] + rng.normal(0, 0.02))
    model = noiseModel.Diagonal.Sigmas([0.3, 0.3, 0.1])

    def add(a, b):
        xa, ya, ta = truth[a]
        xb, yb, tb = truth[b]
        c, s = np.cos(ta), np.sin(ta)
        graph.add_BetweenFactorPose2(a, b, [c * (xb - xa) + s * (yb - ya), -s * (xb - xa) + c * (yb - ya), np.arctan2(np.sin(tb - ta), np.cos(tb - ta))], model)
    for a in range(n_a):
        for b in range(a + 1, n_a + n_b1):
            add(a, b)
    for a in range(n_a, n):
        for b in range(a + 1, n):
            add(a, b)
    graph.add_PriorFactorPose2(n - 1, list(truth[n - 1]), noiseModel.Diagonal.Sigmas([0.05, 0.05, 0.02]))
    return graph, initial, Ordering(list(range(n)))


def test_chained_front_with_wide_separator():
    """A dense front BELOW the root: 1200 frontal columns (5 outer panels) and a 150-column separator, so its first three
    steps go as one chained launch (kernels_step.hpp) whose update tiles, column-block counters and deferred tile rows
    cover separator columns and a ragged last 128-block (1351 = 10 x 128 + 71); its parent is a second dense front.
    [R S d] of every front, delta and the LM trajectory against the oracle."""
    graph, initial, ordering = _two_level_dense_pose2(400, 50, 50)
    opt, orc, params = _pair(graph, initial, ordering)
    big = max(range(opt.num_fronts()), key=lambda i: opt.front_info(i)["nf"])
    info = opt.front_info(big)
    assert info["cls"] == 1 and info["nf"] == 1200 and info["n"] == 1351
    opt.linearize()
    orc.linearize()
    _check_solve(opt, orc, 1e-3)
    _check_lm(opt, orc, params)


def test_pose3_slam_example_file():
    """examples/Data/pose3example.txt (g2o 3D; committed fixture) as in examples/Pose3SLAMExample_g2o.cpp:42-48"""
    graph, initial = load3D(os.path.join(GOLD, "pose3example.txt"))
    graph.add_PriorFactorPose3(0, initial.at(0)[:9].reshape(3, 3), initial.at(0)[9:12],
                               noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Natural(graph)
    opt, orc, params = _pair(graph, initial, ordering)
    assert abs(opt.graph_error() - orc.error()) <= 1e-9 * max(1.0, orc.error())
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-5)
    _check_lm(opt, orc, params)


def test_sphere2500_subset_between_pose3():
    """config C3 shape: first 300 poses of sphere2500 (EDGE3, diagonal information), odometry-chained initial"""
    graph_all, _ = load3D(os.path.join(GOLD, "sphere2500_head.txt"))
    initial = chain_initial_pose3(graph_all)
    graph = graph_all
    graph.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Natural(graph)
    opt, orc, params = _pair(graph, initial, ordering)
    assert abs(opt.graph_error() - orc.error()) <= 1e-9 * max(1.0, orc.error())
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-5)
    opt.iterate()
    orc.lm_iterate(params)
    assert abs(opt.error() - orc.lm_state()["error"]) <= 1e-6 * orc.lm_state()["error"]


def test_projection_factor_visual_slam():
    """config C5 shape (examples/VisualISAM2Example.cpp:88-116, SFMdata.h:42-76): GenericProjectionFactor + priors, batch LM"""
    from gtsam_personal_amd.graph import L, X
    K = [50.0, 50.0, 0.0, 50.0, 50.0]
    pts = np.array([[10., 10, 10], [-10, 10, 10], [-10, -10, 10], [10, -10, 10], [10, 10, -10], [-10, 10, -10], [-10, -10, -10], [10, -10, -10]])
    graph, initial = NonlinearFactorGraph(), Values()
    rng = np.random.default_rng(1)
    up = np.array([0.0, 0.0, 1.0])
    poses = []
    for i in range(8):
        th = i * 2 * np.pi / 8
        eye = np.array([30 * np.cos(th), 30 * np.sin(th), 0.0])
        zc = -eye / np.linalg.norm(eye)
        xc = np.cross(-up, zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc], axis=1)
        poses.append((R, eye))
    noise = noiseModel.Isotropic.Sigma(2, 1.0)
    for i, (R, t) in enumerate(poses):
        for j, p in enumerate(pts):
            q = R.T @ (p - t)
            z = [K[0] * q[0] / q[2] + K[3], K[1] * q[1] / q[2] + K[4]]
            graph.add_GenericProjectionFactor(z, noise, X(i), L(j), K)
        initial.insert_pose3(X(i), R, t + rng.normal(0, 0.2, 3))
    for j, p in enumerate(pts):
        initial.insert_point3(L(j), p + rng.normal(0, 0.3, 3))
    graph.add_PriorFactorPose3(X(0), poses[0][0], poses[0][1], noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.1, 0.3, 0.3, 0.3]))
    graph.add_PriorFactorPoint3(L(0), pts[0], noiseModel.Isotropic.Sigma(3, 0.1))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Schur(graph, initial)
    opt, orc, params = _pair(graph, initial, ordering)
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-5)
    _check_lm(opt, orc, params)


def test_indeterminate_system_reports_like_reference():
    """no priors + lambda = 0: gauge freedom -> Cholesky failure -> IndeterminantLinearSystemException, and
    LM recovers by raising lambda exactly like the oracle (LevenbergMarquardtOptimizer.cpp:158-160, 249-264)"""
    graph, initial, _, ordering = make_bal(n_cam=4, n_pt=12, obs_per_point=3, seed=6, with_priors=False)
    opt, orc, params = _pair(graph, initial, ordering)
    opt.linearize()
    orc.linearize()
    rc, _, _, _ = orc.solve(0.0)
    if rc == 1:
        with pytest.raises(_lib.IndeterminantLinearSystemException):
            opt.solve(0.0)
    # without priors the minimiser is only defined up to the 7-dof gauge: compare costs / lambda trajectory, not coordinates
    _check_lm(opt, orc, params, check_values=False)


def test_medium_bal_properties_without_oracle():
    """size-independent properties at a size the oracle is not run on: the solve satisfies the damped normal
    equations (H + lambda I) delta = g, measured through the linear error model:
        lin_err(0) - lin_err(delta) = 0.5 * (g.delta + lambda |delta|^2)   and   costs decrease monotonically."""
    graph, initial, _, ordering = make_bal(n_cam=60, n_pt=6000, obs_per_point=8, seed=33)
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    opt.linearize()
    lam = 1e-3
    dk, d, e0, e1 = opt.solve(lam)
    # two solves with lambda and 2*lambda: step norm must shrink; linear cost at delta must be below cost at 0
    dk2, d2, _, e1b = opt.solve(2 * lam)
    assert np.linalg.norm(d2) < np.linalg.norm(d)
    assert e1 < e0 and e1 <= e1b
    errs = [opt.error()]
    for _ in range(4):
        opt.iterate()
        errs.append(opt.error())
    assert all(b <= a for a, b in zip(errs, errs[1:]))
    assert errs[-1] < 0.05 * errs[0]


def test_gauss_newton_matches_oracle():
    """GaussNewtonOptimizer (gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66; examples/Pose2SLAMExample_g2o.cpp:70-80 runs the
    Pose2 g2o graphs with it): per-iteration error and the defaultOptimize stopping point against the oracle."""
    from gtsam_personal_amd import GaussNewtonOptimizer, GaussNewtonParams
    graph, initial = readG2o(os.path.join(GOLD, "city10000_head.g2o"))
    graph.add_PriorFactorPose2(0, initial.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Natural(graph)
    params = GaussNewtonParams()
    opt = GaussNewtonOptimizer(graph, initial, ordering, params, device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(params)
    for _ in range(3):
        opt.iterate()
        assert orc.gn_iterate() == 0
        so = orc.lm_state()
        assert opt.iterations() == so["iterations"]
        assert abs(opt.error() - so["error"]) <= 1e-6 * max(1e-12, abs(so["error"])) + 1e-12
    opt.optimize()
    assert orc.gn_optimize(params) == 0
    so = orc.lm_state()
    assert opt.iterations() == so["iterations"]
    assert abs(opt.error() - so["error"]) <= 1e-6 * max(1e-12, abs(so["error"])) + 1e-12
    vo, vg = orc.values(), opt.values()
    for k in vo:
        assert np.allclose(vo[k], vg.at(k), rtol=1e-6, atol=1e-7), k


def test_gauss_newton_indeterminate_propagates():
    """no prior on a Pose2 chain: the undamped system is singular (gauge freedom); the reference's
    GaussNewtonOptimizer::iterate lets IndeterminantLinearSystemException escape, and so does the oracle"""
    from gtsam_personal_amd import GaussNewtonOptimizer, GaussNewtonParams
    graph, initial = NonlinearFactorGraph(), Values()
    m = noiseModel.Diagonal.Sigmas([0.2, 0.2, 0.1])
    for i in range(6):
        initial.insert_pose2(i, float(i), 0.0, 0.0)
    for i in range(5):
        graph.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], m)
    ordering = Ordering.Natural(graph)
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(GaussNewtonParams())
    assert orc.gn_iterate() == 1
    opt = GaussNewtonOptimizer(graph, initial, ordering, device=0)
    with pytest.raises(_lib.IndeterminantLinearSystemException):
        opt.iterate()


@pytest.mark.parametrize("est", ["Huber", "Cauchy", "Tukey", "GemanMcClure", "Welsch", "Fair", "DCS", "L2WithDeadZone"])
def test_robust_noise_pose2_with_outlier(est):
    """noiseModel::Robust around the loop-closure models of a Pose2 graph, one gross outlier among the closures:
    error (rho), reweighted Jacobians, one damped solve and three LM iterations against the oracle."""
    from gtsam_personal_amd.graph import mEstimator
    rng = np.random.default_rng(3)
    n = 30
    graph, initial = NonlinearFactorGraph(), Values()
    for i in range(n):
        initial.insert_pose2(i, float(i) + rng.normal(0, 0.05), rng.normal(0, 0.05), rng.normal(0, 0.02))
    m_odo = noiseModel.Diagonal.Sigmas([0.2, 0.2, 0.1])
    for i in range(n - 1):
        graph.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], m_odo)
    k = {"Huber": 1.345, "Cauchy": 1.0, "Tukey": 4.6851, "GemanMcClure": 1.0, "Welsch": 2.9846, "Fair": 1.3998, "DCS": 1.0,
         "L2WithDeadZone": 0.5}[est]
    base = noiseModel.Gaussian.Information(np.array([[40.0, 2.0, 1.0], [2.0, 30.0, 0.5], [1.0, 0.5, 90.0]]))
    robust = noiseModel.Robust.Create(getattr(mEstimator, est).Create(k), base)
    for a, b in [(0, 10), (5, 20), (12, 29), (3, 17)]:
        graph.add_BetweenFactorPose2(a, b, [float(b - a), 0.0, 0.0], robust)
    graph.add_BetweenFactorPose2(2, 25, [3.0, 4.0, 1.0], robust)  # outlier
    graph.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Diagonal.Sigmas([0.01, 0.01, 0.01]))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Natural(graph)
    opt, orc, params = _pair(graph, initial, ordering)
    assert abs(opt.graph_error() - orc.error()) <= 1e-10 * max(1.0, orc.error())
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-3)
    for _ in range(3):
        opt.iterate()
        orc.lm_iterate(params)
        so = orc.lm_state()
        assert opt.getInnerIterations() == so["inner"]
        assert abs(opt.error() - so["error"]) <= 1e-6 * max(1e-12, abs(so["error"]))


def test_robust_huber_on_projection_factors():
    """Huber on GeneralSFMFactor (the usual robust BA set-up), with corrupted observations"""
    from gtsam_personal_amd.graph import C, P, mEstimator
    graph0, initial, truth, ordering = make_bal(n_cam=8, n_pt=120, obs_per_point=4, seed=11)
    rng = np.random.default_rng(5)
    graph = NonlinearFactorGraph()
    robust = noiseModel.Robust.Create(mEstimator.Huber.Create(1.345), noiseModel.Isotropic.Sigma(2, 1.0))
    for ftype, kind, gi, keys, meas, noise, models in graph0.buckets():
        if keys.shape[1] != 2:
            continue
        z = meas.copy()
        bad = rng.random(len(z)) < 0.05
        z[bad] += rng.normal(0, 40.0, (int(bad.sum()), 2))  # gross outliers
        graph.add_GeneralSFMFactor(z, robust, keys[:, 0], keys[:, 1])
    graph.add_PriorFactorCamera(C(0), truth.at(C(0)), noiseModel.Isotropic.Sigma(9, 0.1))
    graph.add_PriorFactorPoint3(P(0), truth.at(P(0)), noiseModel.Isotropic.Sigma(3, 0.1))
    opt, orc, params = _pair(graph, initial, ordering)
    assert abs(opt.graph_error() - orc.error()) <= 1e-9 * max(1.0, orc.error())
    _check_linearize(opt, orc, graph)
    _check_solve(opt, orc, 1e-3)
    for _ in range(3):
        opt.iterate()
        orc.lm_iterate(params)
        so = orc.lm_state()
        assert opt.getInnerIterations() == so["inner"]
        assert abs(opt.error() - so["error"]) <= 1e-6 * max(1e-12, abs(so["error"]))


@pytest.mark.parametrize("which", ["bal", "pose2"])
def test_dogleg_matches_oracle(which):
    """DoglegOptimizer (gtsam/nonlinear/DoglegOptimizer.cpp:84-126): per-iteration error and trust radius, the dogleg step
    and the defaultOptimize stopping point against the oracle."""
    from gtsam_personal_amd import DoglegOptimizer, DoglegParams
    if which == "bal":
        graph, initial, _, ordering = make_bal(n_cam=40, n_pt=400, obs_per_point=5, seed=9)  # 361 x 361 HBM root + LDS leaves
    else:
        graph, initial = _pose2_graph()
        ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Natural(graph)
    params = DoglegParams()
    opt = DoglegOptimizer(graph, initial, ordering, params, device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.dl_init(params.deltaInitial)
    assert abs(opt.error() - orc.lm_state()["error"]) <= 1e-9 * max(1.0, opt.error())
    for it in range(4):
        opt.iterate()
        assert orc.dl_iterate() == 0
        so = orc.lm_state()
        assert opt.iterations() == so["iterations"]
        assert abs(opt.error() - so["error"]) <= 1e-6 * max(1e-12, abs(so["error"])) + 1e-12, it
        assert abs(opt.getDelta() - so["lambda_"]) <= 1e-6 * so["lambda_"], it
    opt.optimize()
    assert orc.dl_optimize(params) == 0
    so = orc.lm_state()
    assert opt.iterations() == so["iterations"]
    assert abs(opt.error() - so["error"]) <= 1e-6 * max(1e-12, abs(so["error"])) + 1e-12
    vo, vg = orc.values(), opt.values()
    for k in vo:
        a, b = np.array(vo[k], dtype=float), np.array(vg.at(k), dtype=float)
        if which == "pose2":  # theta = +pi and -pi are the same rotation
            a = np.array([a[0], a[1], np.cos(a[2]), np.sin(a[2])])
            b = np.array([b[0], b[1], np.cos(b[2]), np.sin(b[2])])
        assert np.allclose(a, b, rtol=1e-6, atol=1e-7), k


def _huber_pose2_cases():
    """tests/testNonlinearOptimizer.cpp:351-379 (Pose2OptimizationWithHuberNoOutlier) and :416-450 (Pose2OptimizationWithHuber):
    (graph, initial, expected pose 1, tolerance) exactly as the reference builds them"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.graph import mEstimator
    iso1 = noiseModel.Isotropic.Sigma(3, 1.0)
    cases = []
    g, v = NonlinearFactorGraph(), Values()
    g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], iso1)
    g.add_BetweenFactorPose2(0, 1, [1.0, 1.1, np.pi / 4], noiseModel.Robust.Create(mEstimator.Huber.Create(2.0), iso1))
    g.add_BetweenFactorPose2(0, 1, [1.0, 0.9, np.pi / 2], noiseModel.Robust.Create(mEstimator.Huber.Create(3.0), iso1))
    v.insert_pose2(0, 0.0, 0.0, 0.0)
    v.insert_pose2(1, 0.961187, 0.99965, 1.1781)
    cases.append((g, v, np.array([0.961187, 0.99965, 1.1781]), 3e-2))
    g, v = NonlinearFactorGraph(), Values()
    g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Isotropic.Sigma(3, 0.1))
    for meas in ([0.0, 9.0, np.pi / 2], [0.0, 11.0, np.pi / 2], [0.0, 10.0, np.pi / 2], [0.0, 9.0, 0.0]):
        g.add_BetweenFactorPose2(0, 1, meas, noiseModel.Robust.Create(mEstimator.Huber.Create(0.2), iso1))
    v.insert_pose2(0, 0.0, 0.0, 0.0)
    v.insert_pose2(1, 0.0, 10.0, np.pi / 4)
    cases.append((g, v, np.array([0.0, 10.0, 1.45212]), 1e-1))
    return cases


@pytest.mark.parametrize("method", ["gn", "lm", "dl"])
def test_huber_pose2_known_answers_on_gpu(method):
    """tests/testNonlinearOptimizer.cpp:351-379, :416-450: the reference's expected optima, reached by the GPU optimizers"""
    from gtsam_personal_amd import DoglegOptimizer, GaussNewtonOptimizer
    for graph, initial, expect1, tol in _huber_pose2_cases():
        ordering = Ordering.Natural(graph)
        cls = {"gn": GaussNewtonOptimizer, "lm": LevenbergMarquardtOptimizer, "dl": DoglegOptimizer}[method]
        opt = cls(graph, initial, ordering, device=0)
        vals = opt.optimize()
        assert np.abs(vals.at(0)).max() <= tol, (method, vals.at(0))
        assert np.abs(vals.at(1) - expect1).max() <= tol, (method, vals.at(1), expect1)


def _pose2_between_case():
    """gtsam/geometry/tests/testPose2.cpp:525-563 (between) and :67-76 (retract, default chart)"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose2(1, 1.0, 2.0, np.pi / 2)    # gT1: robot at (1,2) looking towards y
    v.insert_pose2(2, -1.0, 4.0, np.pi)       # gT2: robot at (-1,4) looking at negative x
    g.add_BetweenFactorPose2(1, 2, [0.0, 0.0, 0.0], noiseModel.Unit.Create(3))   # error = (x, y, theta) of gT1.between(gT2)
    H1 = np.array([[0.0, -1.0, -2.0], [1.0, 0.0, -2.0], [0.0, 0.0, -1.0]])
    return g, v, np.array([2.0, 2.0, np.pi / 2]), H1, np.eye(3)


def test_pose2_between_known_answers_on_gpu():
    """the reference's expected between pose and Jacobians (testPose2.cpp:525-563) out of the GPU linearize, and its retract (:67-76)"""
    g, v, e_exp, H1, H2 = _pose2_between_case()
    opt = LevenbergMarquardtOptimizer(g, v, Ordering.Natural(g), device=0)
    opt.linearize()
    J = opt.jacobian(0)
    assert np.allclose(J[:, 0:3], H1, atol=1e-12) and np.allclose(J[:, 3:6], H2, atol=1e-12)
    assert np.allclose(-J[:, 6], e_exp, atol=1e-12)
    opt.retract(np.array([0.01, -0.015, 0.99, 0.0, 0.0, 0.0]))  # packed in ordering order (keys 1, 2)
    assert np.allclose(opt.values().at(1), [1.015, 2.01, np.pi / 2 + 0.99], atol=1e-5)


def _pose3_expmap_cases():
    """gtsam/geometry/tests/testPose3.cpp:90-99 (expmap_a_full: R = Rodrigues(0.3, 0, 0), P = (0.2, 0.7, -2)) and :121-137
    (screw motion, expmap_c_full): (xi, expected R, expected t, tolerance)"""
    c3, s3 = np.cos(0.3), np.sin(0.3)
    Rx = np.array([[1, 0, 0], [0, c3, -s3], [0, s3, c3]])
    Rz = np.array([[c3, -s3, 0], [s3, c3, 0], [0, 0, 1]])
    return [(np.array([0.3, 0, 0, 0.2, 0.394742, -2.08998]), Rx, np.array([0.2, 0.7, -2.0]), 1e-5),
            (np.array([0.0, 0.0, 0.3, 0.3, 0.0, 1.0]), Rz, np.array([0.29552, 0.0446635, 1.0]), 1e-6)]


def test_pose3_expmap_known_answers_on_gpu():
    """Pose3::Expmap known answers (testPose3.cpp:90-99, 121-137) out of the GPU retract kernel"""
    for xi, R, t, tol in _pose3_expmap_cases():
        g, v = NonlinearFactorGraph(), Values()
        v.insert_pose3(0, np.eye(3), np.zeros(3))
        g.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Unit.Create(6))
        opt = LevenbergMarquardtOptimizer(g, v, Ordering.Natural(g), device=0)
        opt.retract(xi)
        out = opt.values().at(0)
        assert np.allclose(out[:9].reshape(3, 3), R, atol=tol) and np.allclose(out[9:12], t, atol=tol)


def _projection_factor_case():
    """gtsam/slam/tests/testProjectionFactor.cpp:96-115 (Error) and :141-163 (Jacobian): K = Cal3_S2(fov 60 deg, 640 x 480),
    pose (I, (0,0,-6)), point at the origin, measurement (323, 240)"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.graph import L, X
    fx = 640.0 / (2.0 * np.tan(60.0 * np.pi / 360.0))
    K = [fx, fx, 0.0, 320.0, 240.0]
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose3(X(1), np.eye(3), [0.0, 0.0, -6.0])
    v.insert_point3(L(1), [0.0, 0.0, 0.0])
    g.add_GenericProjectionFactor([323.0, 240.0], noiseModel.Unit.Create(2), X(1), L(1), K)
    H1 = np.array([[0., -554.256, 0., -92.376, 0., 0.], [554.256, 0., 0., 0., -92.376, 0.]])
    H2 = np.array([[92.376, 0., 0.], [0., 92.376, 0.]])
    return g, v, [X(1), L(1)], np.array([-3.0, 0.0]), H1, H2


def test_projection_factor_known_answers_on_gpu():
    g, v, order, e, H1, H2 = _projection_factor_case()
    opt = LevenbergMarquardtOptimizer(g, v, order, device=0)
    opt.linearize()
    J = opt.jacobian(0)
    assert np.allclose(J[:, 0:6], H1, atol=1e-3) and np.allclose(J[:, 6:9], H2, atol=1e-3)
    assert np.allclose(-J[:, 9], e, atol=1e-9)


def _projection_factor_bps_case():
    """gtsam/slam/tests/testProjectionFactor.cpp:118-139 (ErrorWithTransform) and :166-190 (JacobianWithTransform):
    body_P_sensor = (RzRyRx(-pi/2, 0, -pi/2), (0.25, -0.10, 1.0)), vehicle pose (I, (-6.25, 0.10, -1.0))"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.datasets import rot3_rzryrx
    from gtsam_personal_amd.graph import L, X
    fx = 640.0 / (2.0 * np.tan(60.0 * np.pi / 360.0))
    K = [fx, fx, 0.0, 320.0, 240.0]
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose3(X(1), np.eye(3), [-6.25, 0.10, -1.0])
    v.insert_point3(L(1), [0.0, 0.0, 0.0])
    g.add_GenericProjectionFactor([323.0, 240.0], noiseModel.Unit.Create(2), X(1), L(1), K,
                                  body_P_sensor=(rot3_rzryrx(-np.pi / 2, 0.0, -np.pi / 2), [0.25, -0.10, 1.0]))
    H1 = np.array([[-92.376, 0., 577.350, 0., 92.376, 0.], [-9.2376, -577.350, 0., 0., 0., 92.376]])
    H2 = np.array([[0., -92.376, 0.], [0., 0., -92.376]])
    return g, v, [X(1), L(1)], np.array([-3.0, 0.0]), H1, H2


def test_projection_factor_with_body_p_sensor_on_gpu():
    g, v, order, e, H1, H2 = _projection_factor_bps_case()
    opt = LevenbergMarquardtOptimizer(g, v, order, device=0)
    opt.linearize()
    J = opt.jacobian(0)
    assert np.allclose(-J[:, 9], e, atol=1e-9)
    assert np.allclose(J[:, 0:6], H1, atol=1e-3) and np.allclose(J[:, 6:9], H2, atol=1e-3)
    orc = oh.OracleProblem(g, v, order)
    orc.linearize()
    assert np.allclose(J, orc.jacobian(0), rtol=1e-10, atol=1e-10)


def _sfm_error_case():
    """gtsam/slam/tests/testGeneralSFMFactor_Cal3Bundler.cpp:100-113: default Cal3Bundler (f = 1, no distortion), camera at
    (I, (0,0,-6)), landmark at the origin, measurement (3, 0): unwhitened error (-3, 0)"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.graph import L, X
    g, v = NonlinearFactorGraph(), Values()
    v.insert_camera(X(1), np.eye(3), [0.0, 0.0, -6.0], 1.0, 0.0, 0.0)
    v.insert_point3(L(1), [0.0, 0.0, 0.0])
    g.add_GeneralSFMFactor(np.array([[3.0, 0.0]]), noiseModel.Unit.Create(2), np.array([X(1)], dtype=np.uint64), np.array([L(1)], dtype=np.uint64))
    return g, v, [L(1), X(1)]


def test_general_sfm_factor_error_known_answer_on_gpu():
    g, v, order = _sfm_error_case()
    opt = LevenbergMarquardtOptimizer(g, v, order, device=0)
    opt.linearize()
    assert np.allclose(-opt.jacobian(0)[:, -1], [-3.0, 0.0], atol=1e-12)
    assert abs(opt.graph_error() - 4.5) < 1e-12


def _pinhole_project_case():
    """gtsam/geometry/tests/testPinholeCamera.cpp:36-47, 125-131: K = Cal3_S2(625, 625, 0, 0, 0), pose (diag(1,-1,-1), (0,0,0.5)),
    the four points (+-0.08, +-0.08, 0) project to (-100, 100), (-100, -100), (100, -100), (100, 100).  As GenericProjectionFactor
    with a zero measurement the factor error IS the projection."""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.graph import L, X
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose3(X(0), np.diag([1.0, -1.0, -1.0]), [0.0, 0.0, 0.5])
    pts = [(-0.08, -0.08, 0.0), (-0.08, 0.08, 0.0), (0.08, 0.08, 0.0), (0.08, -0.08, 0.0)]
    for i, p in enumerate(pts):
        v.insert_point3(L(i), p)
        g.add_GenericProjectionFactor([0.0, 0.0], noiseModel.Unit.Create(2), X(0), L(i), [625.0, 625.0, 0.0, 0.0, 0.0])
    expect = np.array([[-100.0, 100.0], [-100.0, -100.0], [100.0, -100.0], [100.0, 100.0]])
    return g, v, [L(0), L(1), L(2), L(3), X(0)], expect


def test_pinhole_projection_known_answers_on_gpu():
    g, v, order, expect = _pinhole_project_case()
    opt = LevenbergMarquardtOptimizer(g, v, order, device=0)
    opt.linearize()
    for i in range(4):
        assert np.allclose(-opt.jacobian(i)[:, -1], expect[i], atol=1e-9)


def test_iterate_returns_the_linearized_graph():
    """NonlinearOptimizer::iterate() returns the linear graph (NonlinearOptimizer.h:136; tests/testNonlinearOptimizer.cpp:282 reads
    it): the whitened Jacobians at the values the iteration STARTED from, index-preserving, for LM, Gauss-Newton and Dogleg; the
    bulk tap equals the per-factor tap"""
    from gtsam_personal_amd import DoglegOptimizer, GaussNewtonOptimizer
    graph, initial, _, ordering = make_bal(n_cam=6, n_pt=50, obs_per_point=4, seed=4)
    for cls in (LevenbergMarquardtOptimizer, GaussNewtonOptimizer, DoglegOptimizer):
        opt = cls(graph, initial, ordering, device=0)
        orc = oh.OracleProblem(graph, initial, ordering)
        orc.linearize()  # at the initial values
        linear = opt.iterate()
        assert linear.size() == graph.size()
        fk = graph.factor_keys_in_graph_order()
        zero = {k: np.zeros(len(opt.delta_by_key(np.zeros(opt._ntot))[k])) for k in ordering}
        e0 = 0.0
        for g in range(graph.size()):
            f, Jo = linear.at(g), orc.jacobian(g)
            assert tuple(f.keys()) == tuple(fk[g])
            assert np.allclose(f.augmentedJacobian(), Jo, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(Jo).max())), (cls.__name__, g)
            e0 += 0.5 * float(Jo[:, -1] @ Jo[:, -1])
        assert abs(linear.error(zero) - e0) <= 1e-9 * max(1.0, e0)
        # a graph that is first read after the next linearization refuses instead of returning the wrong numbers
        stale = opt.iterate()
        opt.iterate()
        with pytest.raises(_lib.LmgpuError):
            stale.size()
        opt.close()


def test_gradient_of_the_returned_linear_graph_is_zero_at_the_optimum():
    """tests/testNonlinearOptimizer.cpp:270-283: after optimize(), optimizer.linearize()->gradientAtZero() is zero"""
    graph, initial, _, ordering = make_bal(n_cam=6, n_pt=50, obs_per_point=4, seed=4)
    params = LevenbergMarquardtParams()
    params.relativeErrorTol, params.absoluteErrorTol = 1e-14, 1e-14
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    opt.optimize()
    g = opt.linearize().gradientAtZero()
    scale = max(np.abs(opt.hessian_diagonal()[k]).max() for k in g)
    assert max(np.abs(v).max() for v in g.values()) < 1e-6 * scale
