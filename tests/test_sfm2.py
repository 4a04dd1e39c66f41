"""GeneralSFMFactor2<Cal3_S2> (three-variable SFM factor: pose, point, calibration) and Cal3_S2 as a variable.
Reference: gtsam/slam/GeneralSFMFactor.h:208-262; the workload is examples/SelfCalibrationExample.cpp:45-100.
Pinning: the reference holds no known-answer value for this factor (tests/testExpressionFactor.cpp:277-296 only checks it against an
expression of the same chain), so the oracle is pinned (a) by central differences of its own error function -- the way the
reference's factor tests use numericalDerivative -- and (b) by the reference's structural identity: at a fixed calibration the factor
is GenericProjectionFactor<Pose3, Point3, Cal3_S2> (same PinholeCamera<Cal3_S2>::project), whose oracle restatement is pinned by
tests/test_oracle_golden.py.  The device path is then checked against the oracle on SelfCalibrationExample's graph."""
import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import (DoglegOptimizer, DoglegParams, LevenbergMarquardtOptimizer, LevenbergMarquardtParams, NonlinearFactorGraph,
                                Ordering, Values, noiseModel)
from gtsam_personal_amd.datasets import pose3_compose, rot3_expmap
from gtsam_personal_amd.graph import L, VAR_DIM, X, symbol
from isam2_examples import create_points, create_poses, project_cal3_s2

K0 = symbol("K", 0)
K_TRUE = (50.0, 50.0, 0.0, 50.0, 50.0)


def self_calibration(n_poses=8, n_points=8):
    """examples/SelfCalibrationExample.cpp:45-100: graph and initial estimate"""
    points, poses = create_points()[:n_points], create_poses()[:n_poses]
    g = NonlinearFactorGraph()
    g.add_PriorFactorPose3(X(0), poses[0][0], poses[0][1], noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.1, 0.3, 0.3, 0.3]))
    meas_noise = noiseModel.Isotropic.Sigma(2, 1.0)
    for i, (R, t) in enumerate(poses):
        for j, p in enumerate(points):
            g.add_GeneralSFMFactor2(project_cal3_s2(R, t, p, K_TRUE), meas_noise, X(i), L(j), K0)
    g.add_PriorFactorPoint3(L(0), points[0], noiseModel.Isotropic.Sigma(3, 0.1))
    g.add_PriorFactorCal3_S2(K0, K_TRUE, noiseModel.Diagonal.Sigmas([500, 500, 0.1, 100, 100]))
    v = Values()
    v.insert_cal3_s2(K0, 60.0, 60.0, 0.0, 45.0, 45.0)
    dR, dt = rot3_expmap([-0.1, 0.2, 0.25]), np.array([0.05, -0.10, 0.20])
    for i, (R, t) in enumerate(poses):
        Rn, tn = pose3_compose(R, t, dR, dt)
        v.insert_pose3(X(i), Rn, tn)
    for j, p in enumerate(points):
        v.insert_point3(L(j), p + np.array([-0.25, 0.20, 0.15]))
    return g, v


def _numeric(graph, values, ordering, fidx, key, rows):
    dim = VAR_DIM[values.type(key)]
    cols = []
    for i in range(dim):
        es = []
        for sgn in (+1, -1):
            o = oh.OracleProblem(graph, values, ordering)
            d = {k: np.zeros(VAR_DIM[values.type(k)]) for k in values.keys()}
            d[key][i] = sgn * 1e-6
            o.retract(d)
            e = np.zeros(9)
            oh.lib().orc_factor_evaluate3(o.h, fidx, oh.dp(e), None, None, None)
            es.append(e[:rows].copy())
        cols.append((es[0] - es[1]) / 2e-6)
    return np.stack(cols, axis=1)


def test_sfm2_jacobians_match_numerical_derivatives():
    g, v = self_calibration(3, 3)
    ordering = oh.colamd(g)
    orc = oh.OracleProblem(g, v, ordering)
    fk = g.factor_keys_in_graph_order()
    checked = 0
    for fidx, keys in enumerate(fk):
        if len(keys) != 3:
            continue
        e, H1, H2, H3 = np.zeros(9), np.zeros(81), np.zeros(54), np.zeros(10)
        oh.lib().orc_factor_evaluate3(orc.h, fidx, oh.dp(e), oh.dp(H1), oh.dp(H2), oh.dp(H3))
        assert np.allclose(H1[:12].reshape(2, 6), _numeric(g, v, ordering, fidx, keys[0], 2), rtol=1e-5, atol=1e-4)
        assert np.allclose(H2[:6].reshape(2, 3), _numeric(g, v, ordering, fidx, keys[1], 2), rtol=1e-5, atol=1e-4)
        assert np.allclose(H3.reshape(2, 5), _numeric(g, v, ordering, fidx, keys[2], 2), rtol=1e-5, atol=1e-4)
        checked += 1
    assert checked == 9
    # the calibration prior: error = x - prior, H = I (PriorFactor.h:98-102 with Cal3_S2's vector chart)
    fidx = len(fk) - 1
    e, H1 = np.zeros(9), np.zeros(81)
    oh.lib().orc_factor_evaluate(orc.h, fidx, oh.dp(e), oh.dp(H1), None)
    assert np.allclose(e[:5], np.array([60.0, 60.0, 0.0, 45.0, 45.0]) - np.array(K_TRUE))
    assert np.allclose(H1[:25].reshape(5, 5), np.eye(5))


def test_sfm2_is_the_projection_factor_at_a_fixed_calibration():
    """same PinholeCamera<Cal3_S2>::project: error, H1 and H2 equal GenericProjectionFactor's with K = the calibration variable's value"""
    g, v = self_calibration(3, 3)
    kval = v.at(K0)
    g2 = NonlinearFactorGraph()
    fk = g.factor_keys_in_graph_order()
    rec = {}
    for ftype, kind, gi, keys, meas, noise, models in g.buckets():
        for i, gidx in enumerate(gi.tolist()):
            rec[gidx] = (ftype, keys[i], meas[i], models[i])
    sfm2 = [i for i, k in enumerate(fk) if len(k) == 3]
    for i in sfm2:
        _, keys, meas, model = rec[i]
        g2.add_GenericProjectionFactor(meas, model, int(keys[0]), int(keys[1]), kval)
    v2 = Values()
    for k in v.keys():
        if k != K0:
            v2.insert(k, v.type(k), v.at(k))
    o1 = oh.OracleProblem(g, v, oh.colamd(g))
    o2 = oh.OracleProblem(g2, v2, oh.colamd(g2))
    for j, i in enumerate(sfm2):
        e1, A1, B1, C1 = np.zeros(9), np.zeros(81), np.zeros(54), np.zeros(10)
        e2, A2, B2 = np.zeros(9), np.zeros(81), np.zeros(54)
        oh.lib().orc_factor_evaluate3(o1.h, i, oh.dp(e1), oh.dp(A1), oh.dp(B1), oh.dp(C1))
        oh.lib().orc_factor_evaluate(o2.h, j, oh.dp(e2), oh.dp(A2), oh.dp(B2))
        assert np.array_equal(e1[:2], e2[:2]) and np.array_equal(A1[:12], A2[:12]) and np.array_equal(B1[:6], B2[:6])


def test_self_calibration_oracle_recovers_the_calibration():
    g, v = self_calibration()
    ordering = oh.colamd(g)
    orc = oh.OracleProblem(g, v, ordering)
    params = LevenbergMarquardtParams()
    orc.lm_init(params)
    orc.lm_optimize(params)
    # noise-free measurements and priors at the truth: 2.0e4 -> ~6e-5 in 15 iterations (the weak calibration prior, sigma 500, leaves
    # fy / v0 to trade against the scene: they come back within 2 of the truth, fx / s / u0 within 0.05)
    assert orc.error() < 1e-3
    k = orc.values()[K0]
    assert np.allclose(k[[0, 2, 3]], np.array(K_TRUE)[[0, 2, 3]], atol=0.05) and np.allclose(k, K_TRUE, atol=2.5)


def test_sfm2_behind_the_camera_is_a_zero_factor():
    """GeneralSFMFactor.h:251-260: the CheiralityException is caught, the Jacobians are zero and the error is ZERO"""
    g, v = self_calibration(1, 1)
    R, t = create_poses()[0]
    behind = t - 5.0 * (R[:, 2])  # on the optical axis, behind the camera centre
    v.update(L(0), behind)
    orc = oh.OracleProblem(g, v, oh.colamd(g))
    e, H1, H2, H3 = np.ones(9), np.ones(81), np.ones(54), np.ones(10)
    oh.lib().orc_factor_evaluate3(orc.h, 1, oh.dp(e), oh.dp(H1), oh.dp(H2), oh.dp(H3))
    assert not e[:2].any() and not H1[:12].any() and not H2[:6].any() and not H3.any()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["colamd", "calibration_last"])
def test_gpu_self_calibration_matches_oracle(kind):
    g, v = self_calibration()
    if kind == "colamd":
        ordering = oh.colamd(g)
    else:  # points, poses, then the calibration every factor touches: the root holds K
        ordering = Ordering([L(j) for j in range(8)] + [X(i) for i in range(8)] + [K0])
    orc = oh.OracleProblem(g, v, ordering)
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(g, v, ordering, params, device=0)
    assert abs(opt.error() - orc.error()) <= 1e-9 * max(1.0, abs(orc.error()))
    # structure: the same cliques
    orc.linearize()
    assert orc.solve(1.0)[0] == 0
    assert opt.num_fronts() == len(orc.cliques())
    # linearization: every whitened [A1 A2 A3 b] as the oracle's
    opt.linearize()
    for fidx in range(g.size()):
        Jd = opt.jacobian(fidx)
        Jo = orc.jacobian(fidx)
        assert np.allclose(Jd, Jo, rtol=1e-9, atol=1e-9)
    orc.lm_init(params)
    for it in range(4):
        orc.lm_iterate(params)
        opt.iterate()
        so, sd = orc.lm_state(), opt.state
        assert abs(so["error"] - sd.error) <= 1e-6 * max(1.0, abs(so["error"])), (it, so["error"], sd.error)
        assert abs(so["lambda_"] - sd.lambda_) <= 1e-9 * so["lambda_"]
    vo, vd = orc.values(), opt.values()
    for k in vo.keys():
        assert np.allclose(vo[k], vd.at(k)[:len(vo[k])], rtol=1e-6, atol=1e-6), k
    assert np.allclose(vd.at(K0)[[0, 2, 3]], np.array(K_TRUE)[[0, 2, 3]], atol=1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("mest", ["Huber", "Cauchy"])
def test_gpu_self_calibration_with_robust_noise_matches_oracle(mest):
    """GeneralSFMFactor2 under noiseModel::Robust (gtsam/linear/NoiseModel.cpp:705-735: the whitened [A1 A2 A3 b] of the three-variable
    factor reweighted by sqrt(w(||b||)), its error the m-estimator's loss), two measurements pushed off by tens of pixels: linearization,
    error and four LM iterations against the oracle"""
    points, poses = create_points()[:8], create_poses()[:8]
    g = NonlinearFactorGraph()
    g.add_PriorFactorPose3(X(0), poses[0][0], poses[0][1], noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.1, 0.3, 0.3, 0.3]))
    est = getattr(noiseModel.mEstimator, mest)
    robust = noiseModel.Robust.Create(est.Create(1.345 if mest == "Huber" else 2.0), noiseModel.Isotropic.Sigma(2, 1.0))
    for i, (R, t) in enumerate(poses):
        for j, p in enumerate(points):
            z = project_cal3_s2(R, t, p, K_TRUE)
            if (i, j) in ((2, 3), (5, 1)):
                z = z + np.array([25.0, -40.0])
            g.add_GeneralSFMFactor2(z, robust, X(i), L(j), K0)
    g.add_PriorFactorPoint3(L(0), points[0], noiseModel.Isotropic.Sigma(3, 0.1))
    g.add_PriorFactorCal3_S2(K0, K_TRUE, noiseModel.Diagonal.Sigmas([500, 500, 0.1, 100, 100]))
    _, v = self_calibration()
    ordering = oh.colamd(g)
    orc = oh.OracleProblem(g, v, ordering)
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(g, v, ordering, params, device=0)
    assert abs(opt.error() - orc.error()) <= 1e-9 * max(1.0, abs(orc.error()))
    orc.linearize()
    opt.linearize()
    for fidx in range(g.size()):
        assert np.allclose(opt.jacobian(fidx), orc.jacobian(fidx), rtol=1e-9, atol=1e-9), fidx
    orc.lm_init(params)
    for it in range(4):
        orc.lm_iterate(params)
        opt.iterate()
        so, sd = orc.lm_state(), opt.state
        assert abs(so["error"] - sd.error) <= 1e-6 * max(1.0, abs(so["error"])), (it, so["error"], sd.error)
        assert abs(so["lambda_"] - sd.lambda_) <= 1e-9 * so["lambda_"]
    vo, vd = orc.values(), opt.values()
    for k in vo.keys():
        assert np.allclose(vo[k], vd.at(k)[:len(vo[k])], rtol=1e-6, atol=1e-6), k


def ring_problem(n_poses=30, n_points=40, seed=5):
    """the example's set-up on a larger scene: cameras on a ring of radius 30 facing the centre, random points in the 20-cube"""
    rng = np.random.default_rng(seed)
    points = [rng.uniform(-10, 10, 3) for _ in range(n_points)]
    R0, t0 = create_poses()[0]
    poses = []
    for i in range(n_poses):
        a = 2 * np.pi * i / n_poses
        Rz = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        poses.append((Rz @ R0, Rz @ t0))
    g = NonlinearFactorGraph()
    g.add_PriorFactorPose3(X(0), poses[0][0], poses[0][1], noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.1, 0.3, 0.3, 0.3]))
    meas_noise = noiseModel.Isotropic.Sigma(2, 1.0)
    for i, (R, t) in enumerate(poses):
        for j, p in enumerate(points):
            g.add_GeneralSFMFactor2(project_cal3_s2(R, t, p, K_TRUE) + rng.normal(0, 0.5, 2), meas_noise, X(i), L(j), K0)
    g.add_PriorFactorPoint3(L(0), points[0], noiseModel.Isotropic.Sigma(3, 0.1))
    g.add_PriorFactorCal3_S2(K0, K_TRUE, noiseModel.Diagonal.Sigmas([500, 500, 0.1, 100, 100]))
    v = Values()
    v.insert_cal3_s2(K0, 55.0, 55.0, 0.0, 47.0, 47.0)
    dR, dt = rot3_expmap([-0.02, 0.03, 0.04]), np.array([0.05, -0.10, 0.20])
    for i, (R, t) in enumerate(poses):
        Rn, tn = pose3_compose(R, t, dR, dt)
        v.insert_pose3(X(i), Rn, tn)
    for j, p in enumerate(points):
        v.insert_point3(L(j), p + np.array([-0.25, 0.20, 0.15]))
    return g, v, n_poses, n_points


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["colamd", "calibration_last"])
def test_gpu_self_calibration_with_a_dense_root(kind):
    """30 poses x 40 points: with the calibration last the root holds every pose and K (186 columns: an HBM front whose children are
    point leaves carrying three-variable factors -- they hand over update matrices, the Schur gather takes two-variable factors only)"""
    g, v, n_poses, n_points = ring_problem()
    ordering = oh.colamd(g) if kind == "colamd" else Ordering([L(j) for j in range(n_points)] + [X(i) for i in range(n_poses)] + [K0])
    orc = oh.OracleProblem(g, v, ordering)
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(g, v, ordering, params, device=0)
    infos = [opt.front_info(i) for i in range(opt.num_fronts())]
    if kind == "calibration_last":
        assert max(fi["n"] for fi in infos) >= 6 * n_poses + 5 + 1 and any(fi["cls"] == 1 for fi in infos)  # (+ the last point, merged into the root)
    orc.lm_init(params)
    for it in range(3):
        orc.lm_iterate(params)
        opt.iterate()
        so, sd = orc.lm_state(), opt.state
        assert abs(so["error"] - sd.error) <= 1e-6 * max(1.0, abs(so["error"])), (it, so["error"], sd.error)
    vo, vd = orc.values(), opt.values()
    for k in vo.keys():
        assert np.allclose(vo[k], vd.at(k)[:len(vo[k])], rtol=1e-6, atol=1e-6), k
    opt.linearize()
    orc.linearize()
    hd, ho = opt.hessian_diagonal(), orc.hessian_diagonal()
    for k in vo.keys():
        assert np.allclose(hd[k], ho[k], rtol=1e-9, atol=1e-9), k


@pytest.mark.gpu
def test_gpu_self_calibration_dogleg_like_the_example():
    """the example optimises with Dogleg (SelfCalibrationExample.cpp:96)"""
    g, v = self_calibration()
    ordering = oh.colamd(g)
    opt = DoglegOptimizer(g, v, ordering, DoglegParams(), device=0)
    res = opt.optimize()
    orc = oh.OracleProblem(g, v, ordering)
    orc.dl_init(1.0)
    orc.dl_optimize(DoglegParams())
    assert abs(opt.error() - orc.error()) <= 1e-6 * max(1e-3, abs(orc.error()))
    assert np.allclose(res.at(K0), orc.values()[K0], rtol=1e-6, atol=1e-6)


def self_calibration_isam2_steps():
    """VisualISAM2Example's sequence (examples/VisualISAM2Example.cpp:88-131) with the calibration as a variable: the projection factors
    become GeneralSFMFactor2<Cal3_S2>, K gets the prior of SelfCalibrationExample.cpp:80-83 and a perturbed initial value"""
    noise = noiseModel.Isotropic.Sigma(2, 1.0)
    points, poses = create_points(), create_poses()
    dR, dt = rot3_expmap([-0.1, 0.2, 0.25]), np.array([0.05, -0.10, 0.20])
    steps = []
    g, v = NonlinearFactorGraph(), Values()
    for i, (R, t) in enumerate(poses):
        for j, p in enumerate(points):
            g.add_GeneralSFMFactor2(project_cal3_s2(R, t, p, K_TRUE), noise, X(i), L(j), K0)
        Ri, ti = pose3_compose(R, t, dR, dt)
        v.insert_pose3(X(i), Ri, ti)
        if i == 0:
            g.add_PriorFactorPose3(X(0), R, t, noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.1, 0.3, 0.3, 0.3]))
            g.add_PriorFactorPoint3(L(0), points[0], noiseModel.Isotropic.Sigma(3, 0.1))
            g.add_PriorFactorCal3_S2(K0, K_TRUE, noiseModel.Diagonal.Sigmas([5, 5, 0.1, 5, 5]))
            v.insert_cal3_s2(K0, 52.0, 52.0, 0.0, 49.0, 49.0)
            for j, p in enumerate(points):
                v.insert_point3(L(j), p + np.array([-0.25, 0.20, 0.15]))
        else:
            steps.append((g, v))
            steps.append((NonlinearFactorGraph(), Values()))
            g, v = NonlinearFactorGraph(), Values()
    return steps


@pytest.mark.gpu
@pytest.mark.skipif(not oh.have_ref(), reason="oracle/_ref (CCOLAMD of the reference) not built")
def test_gpu_isam2_with_three_variable_factors():
    """the incremental path with GeneralSFMFactor2 and a Cal3_S2 variable: update by update against the oracle (bookkeeping, Bayes tree,
    [R S d], linearization point, delta, estimate), as tests/test_gpu_isam2.py does for the reference's own sequences"""
    from gtsam_personal_amd import ISAM2Params
    from test_gpu_isam2 import run_sequence
    isam, orc = run_sequence(self_calibration_isam2_steps(), ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1))
    est = isam.calculateEstimate()
    assert np.allclose(est.at(K0)[[0, 2, 3]], np.array(K_TRUE)[[0, 2, 3]], atol=1.0)
    isam.close()
