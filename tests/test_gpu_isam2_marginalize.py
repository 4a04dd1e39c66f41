"""ISAM2::marginalizeLeaves on the device against the oracle (oracle/isam2_oracle.hpp: isam2_marginalize_leaves), call by call: the same
factor slots, the same marginal factors, the same tree, and the same answers from every update that follows.  The cases are the
reference's own (tests/testGaussianISAM2.cpp:736-966; see tests/test_isam2_marginalize_oracle.py, which pins the oracle with the
reference's checks) plus the shapes the device path treats specially: cliques wider than an LDS front, dog leg, slot reuse."""
import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import ISAM2, ISAM2DoglegParams, ISAM2Params, NonlinearFactorGraph, Values, noiseModel
from gtsam_personal_amd._lib import LmgpuError
from isam2_examples import slamlike_steps
from test_gpu_isam2 import ccolamd, compare_state, dense_pose2_steps
from test_isam2_marginalize_oracle import MODEL, marked_keys_for, pose3_grid

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not oh.have_ref(), reason="oracle/_ref (CCOLAMD of the reference) not built")]


def pair(params=None, find_unused=False):
    p = params or ISAM2Params()
    p.findUnusedFactorSlots = find_unused
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
    if isinstance(p.optimizationParams, ISAM2DoglegParams):
        o = p.optimizationParams
        orc.set_dogleg(o.initialDelta, o.wildfireThreshold, o.adaptationMode)
    orc.set_find_unused_factor_slots(find_unused)
    return ISAM2(p, ccolamd=ccolamd, device=0), orc


def both_update(isam, orc, *args, **kw):
    rg, ro = isam.update(*args, **kw).as_dict(), orc.update(*args, **kw)
    assert rg == ro, (rg, ro)
    assert isam.num_factors() == orc.num_factors()
    compare_state(isam, orc)


def compare_factor_list(isam, orc, tol=1e-7):
    assert isam.num_factors() == orc.num_factors()
    for i in range(orc.num_factors()):
        assert isam.factor_exists(i) == orc.factor_exists(i), i
        mo, mg = orc.marginal_factor(i), isam.marginal_factor(i)
        assert (mo is None) == (mg is None), i
        if mo is not None:
            assert mg[0] == mo[0] and mg[1] == mo[1], (i, mg[0], mo[0])
            assert np.allclose(mg[2], mo[2], rtol=tol, atol=tol * max(1.0, np.abs(mo[2]).max())), i
    assert isam.getFixedVariables() == orc.getFixedVariables()


def both_marginalize(isam, orc, keys):
    mg, dg = isam.marginalizeLeaves(keys)
    mo, do = orc.marginalizeLeaves(keys)
    assert mg == mo and dg == do, (mg, mo, dg, do)
    compare_factor_list(isam, orc)
    assert isam.size() == len(orc.getLinearizationPoint().keys())
    compare_state(isam, orc)
    return mg, dg


def update_and_marginalize(isam, orc, g, v, marginalizable):
    """updateAndMarginalize tests/testGaussianISAM2.cpp:705-726 on both"""
    constrained = None
    if marginalizable:
        constrained = {int(k): 1 for k in orc.getDelta()}
        for k in (v.keys() if v is not None else []):
            constrained[int(k)] = 1
        for k in marginalizable:
            constrained[int(k)] = 0
    marked = marked_keys_for(orc, marginalizable) if marginalizable else []
    both_update(isam, orc, g, v, constrainedKeys=constrained, extraReelimKeys=marked)
    if marginalizable:
        both_marginalize(isam, orc, marginalizable)


def chain(n, between):
    g, v = NonlinearFactorGraph(), Values()
    g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], MODEL)
    for a, b in between:
        g.add_BetweenFactorPose2(a, b, [0.1 * (b - a), 0.02, 0.01], MODEL)
    for k in range(n):
        v.insert_pose2(k, 0.1 * k + 0.03, -0.02, 0.02)
    return g, v


@pytest.mark.parametrize("n,between,leaf", [
    (3, [(0, 1), (1, 2), (0, 2)], [0]),                                    # marginalizeLeaves1
    (4, [(0, 1), (1, 2), (0, 2), (2, 3)], [0]),                            # marginalizeLeaves2
    (6, [(0, 1), (1, 2), (0, 2), (2, 3), (3, 4), (4, 5), (3, 5)], [0]),    # marginalizeLeaves3
    (3, [(0, 2), (1, 2)], [1]),                                            # marginalizeLeaves4
    (6, [(0, 1), (1, 2), (0, 2), (2, 3), (3, 4), (4, 5), (3, 5)], [0, 1]),  # two leaves of one clique chain at once
])
def test_marginalize_leaves_small(n, between, leaf):
    isam, orc = pair()
    g, v = chain(n, between)
    both_update(isam, orc, g, v, constrainedKeys={k: k for k in range(n)})
    mi, di = both_marginalize(isam, orc, leaf)
    assert mi and di
    # the marginal factors enter what follows: an incremental update, a forced relinearization (fixed variables stay), a full re-elimination
    keep = [k for k in range(n) if k not in leaf]
    g2 = NonlinearFactorGraph()
    g2.add_PriorFactorPose2(keep[-1], [0.3, 0.0, 0.0], MODEL)
    both_update(isam, orc, g2, None)
    both_update(isam, orc, None, None, force_relinearize=True)
    both_update(isam, orc, None, None, extraReelimKeys=keep, forceFullSolve=True)
    assert abs(isam.error(0) - orc.error(0)) <= 1e-9 * max(1.0, abs(orc.error(0)))
    isam.close()


def test_marginalize_leaves_slamlike_then_more_updates():
    """marginalizeLeaves5 (:854-862), then the marginal factor through updates with relinearization"""
    isam, orc = pair(ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1))
    steps = slamlike_steps()
    for g, v in steps:
        both_update(isam, orc, g, v)
    both_marginalize(isam, orc, [0])
    odo = noiseModel.Diagonal.Sigmas([0.1, 0.1, np.pi / 100.0])
    last = max(k for k in orc.getLinearizationPoint().keys() if k < 100)
    for i in range(last, last + 3):
        g, v = NonlinearFactorGraph(), Values()
        g.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], odo)
        v.insert_pose2(i + 1, float(i + 1) + 0.2, 0.1, 0.02)
        both_update(isam, orc, g, v)
    isam.close()


@pytest.mark.parametrize("dim", [4, 10])
def test_marginalize_grid_one_by_one(dim):
    """marginalizeLeaves6 (:865-905): every variable of a Pose3 grid, one at a time, each after the update that makes it a leaf"""
    g, v = pose3_grid(dim)
    isam, orc = pair()
    update_and_marginalize(isam, orc, g, v, [])
    order = np.random.default_rng(1234).permutation(dim * dim)
    for key in order.tolist():
        update_and_marginalize(isam, orc, None, None, [key])
    assert isam.size() == 0 and len(isam.calculateBestEstimate().keys()) == 0
    isam.close()


def test_marginalize_root_and_slot_reuse():
    """MarginalizeRoot (:908-933) and marginalizationSize (:936-966)"""
    nm = noiseModel.Isotropic.Sigma(6, 1.0)
    isam, orc = pair()
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose3(0, np.eye(3), [0.0, 0.0, 0.0])
    g.add_PriorFactorPose3(0, np.eye(3), [0.0, 0.0, 0.0], nm)
    update_and_marginalize(isam, orc, g, v, [])
    update_and_marginalize(isam, orc, None, None, [0])
    assert isam.size() == 0
    isam.close()

    isam, orc = pair(find_unused=True)
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose3(0, np.eye(3), [0.0, 0.0, 0.0])
    g.add_PriorFactorPose3(0, np.eye(3), [0.0, 0.0, 0.0], nm)
    v.insert_pose3(1, np.eye(3), [0.1, 0.0, 0.0])
    g.add_BetweenFactorPose3(0, 1, np.eye(3), [0.0, 0.0, 0.0], nm)
    update_and_marginalize(isam, orc, g, v, [])
    n = isam.num_factors()
    update_and_marginalize(isam, orc, None, None, [0])
    assert isam.num_factors() == n
    g2 = NonlinearFactorGraph()
    g2.add_PriorFactorPose3(1, np.eye(3), [0.0, 0.0, 0.0], nm)
    both_update(isam, orc, g2, None)
    assert isam.num_factors() == n and all(isam.factor_exists(i) for i in range(n))
    compare_factor_list(isam, orc)
    isam.close()


def test_a_key_that_is_not_a_leaf_is_refused_and_nothing_changes():
    g, v = pose3_grid(4)
    isam, orc = pair()
    both_update(isam, orc, g, v)
    cl = orc.cliques()
    key = next(keys[0] for i, (keys, nf, _, par) in enumerate(cl) if any(p == i and keys[0] in k2[n2:] for k2, n2, _, p in cl))
    with pytest.raises(LmgpuError, match="not leaves"):
        isam.marginalizeLeaves([key])
    with pytest.raises(LmgpuError, match="not a variable"):
        isam.marginalizeLeaves([12345])
    compare_state(isam, orc)
    compare_factor_list(isam, orc)
    # ... and the handle goes on
    update_and_marginalize(isam, orc, None, None, [key])
    isam.close()


def test_marginalize_out_of_cliques_wider_than_an_lds_front():
    """the top clique of dense_pose2_steps holds every pose (181 scalar columns): one pose goes (the clique stays a dense-front block),
    then fifteen at once (what remains fits an LDS front and is stored as one), with updates in between"""
    steps = dense_pose2_steps(60)
    isam, orc = pair(ISAM2Params(relinearizeThreshold=0.05, relinearizeSkip=2))
    for g, v in steps:
        assert isam.update(g, v).as_dict() == orc.update(g, v)
    compare_state(isam, orc)
    assert max(R.shape[1] for _, _, R, _ in isam.cliques()) > 139
    update_and_marginalize(isam, orc, None, None, [0])
    assert max(R.shape[1] for _, _, R, _ in isam.cliques()) > 139
    both_update(isam, orc, None, None, force_relinearize=True)
    update_and_marginalize(isam, orc, None, None, list(range(1, 16)))
    assert max(R.shape[1] for _, _, R, _ in isam.cliques()) <= 139
    both_update(isam, orc, None, None, force_relinearize=True)
    g = NonlinearFactorGraph()
    g.add_PriorFactorPose2(59, [1.0, 1.0, 0.5], noiseModel.Diagonal.Sigmas([0.5, 0.5, 0.5]))
    both_update(isam, orc, g, None)
    isam.close()


def test_fixed_lag_run_with_dogleg():
    """an odometry chain with loop closures to recent poses, the oldest pose marginalized every step (what IncrementalFixedLagSmoother does
    with ISAM2: order the leaving keys first, update, marginalizeLeaves), dog-leg steps, slots reused"""
    rng = np.random.default_rng(5)
    p = ISAM2Params(ISAM2DoglegParams(1.0, 1e-5, 0), relinearizeThreshold=0.01, relinearizeSkip=1)
    isam, orc = pair(p, find_unused=True)
    odo = noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.05])
    lag = 6
    for i in range(25):
        g, v = NonlinearFactorGraph(), Values()
        if i == 0:
            g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Diagonal.Sigmas([0.01, 0.01, 0.01]))
        else:
            g.add_BetweenFactorPose2(i - 1, i, np.array([1.0, 0.0, 0.1]) + rng.normal(0, 0.01, 3), odo)
            if i >= 3:
                a = i - 3
                th = 0.1 * 3
                # (a rough relative pose three steps back: only its consistency between the two implementations matters)
                g.add_BetweenFactorPose2(a, i, np.array([2.9, 0.4, th]) + rng.normal(0, 0.01, 3), odo)
        v.insert_pose2(i, float(i) + rng.normal(0, 0.05), 0.05 * i * i + rng.normal(0, 0.05), 0.1 * i)
        leaving = [i - lag] if i >= lag else []
        update_and_marginalize(isam, orc, g, v, leaving)
        assert abs(isam.doglegDelta() - orc.doglegDelta()) <= 1e-9 * max(1.0, orc.doglegDelta())
    assert isam.size() == lag
    assert isam.num_factors() < 40  # slots are reused: the list does not grow with the run
    isam.close()


def test_fixed_lag_smoother_over_city10000_against_the_oracle(tmp_path):
    """300 poses of city10000 through a fixed-lag smoother's calls (lag 25: order the leaving pose first, update, marginalizeLeaves; loop
    closures inside the window stay) on the device and on the oracle, compared step by step; then the same sequence through the C ABI
    from C++ (tests/cpp/isam2_harness: C / X / M lines) must land on the same estimate."""
    import json
    import os
    import subprocess
    from gtsam_personal_amd.incremental_workloads import fixed_lag_pose2_steps, fixed_lag_update_params, write_isam2_sequence
    g2o = os.path.join(os.path.dirname(__file__), "golden", "city10000.g2o")
    p = ISAM2Params()
    isam, orc = pair(p, find_unused=True)
    est = {}
    steps = []
    for g, v, leaving in fixed_lag_pose2_steps(g2o, 300, 25, lambda k: est[k]):
        constrained, marked = fixed_lag_update_params(orc.cliques(), list(orc.getDelta().keys()), list(v.keys()), leaving)
        kw = dict(constrainedKeys=constrained, extraReelimKeys=marked)
        rg, ro = isam.update(g, v, **kw).as_dict(), orc.update(g, v, **kw)
        assert rg == ro, (rg, ro)
        if leaving:
            assert isam.marginalizeLeaves(leaving) == orc.marginalizeLeaves(leaving)
        e = isam.calculateEstimate()
        est = {int(k): np.asarray(e.at(k), dtype=float)[:3] for k in e.keys()}
        steps.append((g, v, None, dict(constrained=constrained, extra_reelim=marked, marginalize=leaving)))
        if len(steps) % 25 == 0:
            compare_state(isam, orc)
    compare_state(isam, orc)
    compare_factor_list(isam, orc)
    assert isam.size() == 25 and isam.num_factors() < 120
    final = isam.calculateEstimate()
    isam.close()
    # ---- the same calls from C++
    harness = os.path.join(os.path.dirname(__file__), "cpp", "isam2_harness")
    ref = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libccolamd_ref.so")
    if not (os.path.exists(harness) and os.path.exists(ref)):
        pytest.skip("tests/cpp/isam2_harness or oracle/_ref not built")
    seq = str(tmp_path / "fixed_lag.txt")
    write_isam2_sequence(seq, p, steps)
    out = subprocess.run([harness, seq, "0", ref], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    res = json.loads(out.stdout)
    assert res["updates"] == len(steps) and res["variables"] == 25 and res["marginalized"] == sum(len(st[3]["marginalize"]) for st in steps)
    got = {int(rec[0]): np.array(rec[1:]) for rec in res["estimate"]}
    assert sorted(got) == sorted(int(k) for k in final.keys())
    for k in final.keys():
        assert np.allclose(got[int(k)], final.at(k)[:3], rtol=1e-9, atol=1e-9), k
