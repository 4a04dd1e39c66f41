"""BASELINE config 5: ISAM2 (incremental relinearize + partial re-eliminate) on the device against the CPU oracle, update by
update, on the reference's own incremental workloads (tests/isam2_examples.py): VisualISAM2Example's 8-pose / 8-point
sequence (examples/VisualISAM2Example.cpp:88-131, relinearizeThreshold 0.01, relinearizeSkip 1, an extra update() per frame) and
createSlamlikeISAM2 (tests/testGaussianISAM2.cpp:44-168) with and without relinearization.

Per update: identical bookkeeping (variables relinearized / re-eliminated, factors recalculated, clique count, batch-or-incremental),
identical Bayes tree (cliques depth-first: keys in Scatter order, frontal counts, parents) = bit-exact ordering / indexing; [R S d]
of every clique, the linearization point, delta and calculateEstimate() within 1e-6 relative."""
import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import ISAM2, ISAM2DoglegParams, ISAM2GaussNewtonParams, ISAM2Params, NonlinearFactorGraph, Values, noiseModel
from gtsam_personal_amd.graph import symbol
from isam2_examples import constrained_ordering_steps, create_points, slamlike_steps, stale_landmark_steps, visual_steps

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not oh.have_ref(), reason="oracle/_ref (CCOLAMD of the reference) not built")]


def ccolamd(n_rows, n_cols, col_ptr, row_idx, cmember):
    return oh.ccolamd_csc(n_rows, n_cols, col_ptr, row_idx, cmember)


def compare_state(isam, orc, tol=1e-6):
    cg, co = isam.cliques(), orc.cliques()
    assert len(cg) == len(co)
    for i, ((kg, nfg, Rg, pg), (ko, nfo, Ro, po)) in enumerate(zip(cg, co)):
        assert kg == ko and nfg == nfo and pg == po, (i, kg, ko)
        assert Rg.shape == Ro.shape
        assert np.allclose(Rg, Ro, rtol=tol, atol=1e-7 * max(1.0, np.abs(Ro).max())), i
    lg, lo = isam.getLinearizationPoint(), orc.getLinearizationPoint()
    assert lg.keys() == lo.keys()
    for k in lo.keys():
        assert np.allclose(lg.at(k), lo.at(k), rtol=tol, atol=1e-9), k
    dg, do = isam.getDelta(), orc.getDelta()
    for k in do:
        assert np.allclose(dg[k], do[k], rtol=tol, atol=1e-8), (k, dg[k], do[k])
    eg, eo = isam.calculateEstimate(), orc.calculateEstimate()
    for k in eo.keys():
        assert np.allclose(eg.at(k), eo.at(k), rtol=tol, atol=1e-8), k


def run_sequence(steps, params, check_every_step=True):
    p = params
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
    for i, (g, v) in enumerate(steps):
        rg = isam.update(g, v).as_dict()
        ro = orc.update(g, v)
        assert rg == ro, (i, rg, ro)
        if check_every_step:
            compare_state(isam, orc)
    compare_state(isam, orc)
    return isam, orc


def test_visual_isam2_example_step_by_step():
    params = ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1)
    isam, orc = run_sequence(visual_steps(), params)
    result = isam.calculateEstimate()
    assert len(result.keys()) == 16
    for j, p in enumerate(create_points()):  # tests/testVisualISAM2.cpp:112-117
        assert np.allclose(result.at(symbol("l", j))[:3], p, rtol=0, atol=0.01), j
    best, best_o = isam.calculateBestEstimate(), orc.calculateBestEstimate()
    for k in best_o.keys():
        assert np.allclose(best.at(k), best_o.at(k), rtol=1e-6, atol=1e-8), k
    isam.close()


def test_updates_with_a_staging_arena_that_always_overflows(monkeypatch, dev_switches):
    """every table and index list of an update goes through the pinned staging arena (csrc/isam2.hpp); with a 64-byte arena each
    request takes the overflow path (a chunk of its own, freed at the next update) -- same results"""
    monkeypatch.setenv("LMGPU_ISAM2_STAGE_BYTES", "64")
    isam, _ = run_sequence(visual_steps(), ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1))
    isam.close()


@pytest.mark.parametrize("switch", ["LMGPU_ISAM2_NO_BYVALUE", "LMGPU_ISAM2_NO_MIRROR", "LMGPU_ISAM2_NO_PREWALK", "LMGPU_ISAM2_LATE_WALK_PREP"])
def test_walk_scheduling_forms_against_the_oracle(monkeypatch, dev_switches, switch):
    """the back-substitution walk's round-3 forms, each switched OFF in turn (the default is what every other test here runs): the
    re-eliminated top of the tree handing x over by value, the host's mirror of delta written by the walk, the walk behind the update's
    elimination when delta is going to be read anyway, its seeds / status word / all-ones fills pushed with the elimination's flush"""
    monkeypatch.setenv(switch, "1")
    isam, _ = run_sequence(visual_steps(), ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1))
    isam.close()
    isam, _ = run_sequence(slamlike_steps(), ISAM2Params())
    isam.close()


@pytest.mark.parametrize("params", [ISAM2Params(), ISAM2Params(ISAM2GaussNewtonParams(0.001), 0.01, 1, True), ISAM2Params(ISAM2GaussNewtonParams(0.001), 0.05, 3, True)],
                         ids=["defaults", "eager", "skip3"])
def test_updates_without_a_reader_in_between(params):
    """nobody asks for delta between the updates: an update then only ends with the walk when the NEXT one checks relinearization
    (relinearizeSkip), and several updates' re-eliminated sets reach one walk; counts per update and the final state against the oracle"""
    isam, _ = run_sequence(slamlike_steps(), params, check_every_step=False)
    isam.close()


@pytest.mark.parametrize("params", [
    ISAM2Params(ISAM2GaussNewtonParams(0.001), 0.0, 0, False),   # tests/testGaussianISAM2.cpp:303: relinearization off
    ISAM2Params(),                                                # defaults: threshold 0.1, every 10th update
    ISAM2Params(ISAM2GaussNewtonParams(0.001), 0.01, 1, True),   # relinearize eagerly
    ISAM2Params(ISAM2GaussNewtonParams(0.0), 0.05, 2, True),     # wildfire off: full back-substitution
], ids=["norelin", "defaults", "eager", "fullsolve"])
def test_slamlike_step_by_step(params):
    isam, _ = run_sequence(slamlike_steps(), params)
    isam.close()


def test_forced_relinearization_and_bare_updates():
    steps = slamlike_steps()
    p = ISAM2Params()
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
    for g, v in steps:
        assert isam.update(g, v).as_dict() == orc.update(g, v)
    for _ in range(3):
        assert isam.update(force_relinearize=True).as_dict() == orc.update(force_relinearize=True)
        compare_state(isam, orc)
    isam.close()


def test_incremental_pose_graph_like_time_incremental():
    """timing/timeIncremental.cpp:84-170 on the first 400 poses of city10000 (tests/golden/city10000_head.g2o, 431 edges incl. loop
    closures): one pose per update with the edges that reach back from it, the new pose initialised from the previous estimate composed
    with the odometry; default ISAM2Params (relinearizeThreshold 0.1, relinearizeSkip 10).  Counts per update, the whole state every
    40 updates and at the end, against the oracle."""
    import os
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.datasets import readG2o
    graph, _ = readG2o(os.path.join(os.path.dirname(__file__), "golden", "city10000_head.g2o"))
    edges = []  # (k1, k2, measured, model) in file order
    for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
        for i, g in enumerate(gi.tolist()):
            edges.append((g, int(keys[i][0]), int(keys[i][1]), meas[i], models[i]))
    edges.sort()
    p = ISAM2Params()
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)

    def compose(a, d):
        c, s = np.cos(a[2]), np.sin(a[2])
        return np.array([a[0] + c * d[0] - s * d[1], a[1] + s * d[0] + c * d[1], a[2] + d[2]])

    nxt, step, n_poses = 0, 1, 1 + max(max(e[1], e[2]) for e in edges)
    while nxt < len(edges):
        g, v = NonlinearFactorGraph(), Values()
        if step == 1:
            v.insert_pose2(0, 0.0, 0.0, 0.0)
            g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Unit.Create(3))
        while nxt < len(edges):
            _, k1, k2, m, model = edges[nxt]
            if k1 > step or k2 > step:
                break
            g.add_BetweenFactorPose2(k1, k2, m, model)
            if k2 == step and k1 == step - 1:
                prev = np.zeros(3) if step == 1 else orc.calculateEstimate().at(step - 1)
                v.insert(step, 0, compose(prev, m))
            nxt += 1
        rg = isam.update(g, v).as_dict()
        ro = orc.update(g, v)
        assert rg == ro, (step, rg, ro)
        if step % 40 == 0:
            compare_state(isam, orc)
        step += 1
    assert isam.size() == n_poses
    compare_state(isam, orc)
    isam.close()


def dense_pose2_steps(n_poses=60, seed=3):
    """a graph in which every new pose is tied to EVERY earlier one (one pose per update): the top clique holds all poses, i.e. it
    outgrows an LDS front (139 scalar columns) at pose 47"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    rng = np.random.default_rng(seed)
    truth = [np.array([2.0 * np.cos(0.3 * i) * (1 + 0.05 * i), 2.0 * np.sin(0.3 * i) * (1 + 0.05 * i), 0.3 * i + 1.0]) for i in range(n_poses)]

    def between(a, b):
        c, s_ = np.cos(a[2]), np.sin(a[2])
        dx, dy = b[0] - a[0], b[1] - a[1]
        th = b[2] - a[2]
        return np.array([c * dx + s_ * dy, -s_ * dx + c * dy, np.arctan2(np.sin(th), np.cos(th))])

    model = noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.05])
    steps = []
    for i in range(n_poses):
        g, v = NonlinearFactorGraph(), Values()
        if i == 0:
            g.add_PriorFactorPose2(0, truth[0], noiseModel.Diagonal.Sigmas([0.01, 0.01, 0.01]))
        for j in range(i):
            g.add_BetweenFactorPose2(j, i, between(truth[j], truth[i]) + rng.normal(0, 0.01, 3), model)
        v.insert_pose2(i, *(truth[i] + rng.normal(0, 0.05, 3)))
        steps.append((g, v))
    return steps


def test_cliques_wider_than_an_lds_front():
    """a clique of more than 139 scalar columns is eliminated in place by the dense-front kernels of the batch path (assembly by row,
    256-row panels, trailing updates on the matrix cores) and walked from memory by the wildfire; smaller cliques above and below it
    extend-add its update matrix as they do any other.  Compared with the oracle update by update like every other sequence."""
    steps = dense_pose2_steps()
    isam, orc = run_sequence(steps, ISAM2Params(relinearizeThreshold=0.05, relinearizeSkip=2), check_every_step=False)
    widest = max(R.shape[1] for _, _, R, _ in isam.cliques())
    assert widest > 139, widest
    # a few more updates on top of the wide tree: incremental paths through a wide clique (cached boundary factors, orphans)
    assert isam.update(force_relinearize=True).as_dict() == orc.update(force_relinearize=True)
    compare_state(isam, orc)
    isam.close()


def test_full_city10000_incremental_run():
    """timing/timeIncremental.cpp's workload: all 10 000 poses of city10000, one per update (tests/tools/isam2_long_run.py), the same updates
    replayed through the oracle: identical Bayes tree (7 939 cliques after the closing batch step, the widest 298 columns -- the
    dense-front fallback), identical counts of the last update, estimate within 1e-6."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("isam2_long_run", os.path.join(os.path.dirname(__file__), "tools", "isam2_long_run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(10000, check=True, verbose=False)
    assert res["stopped"] is None, res
    assert res["updates"] == 9999 and res["same_tree"] and res["last_counts_equal"], res
    assert res["max_rel_diff"] < 1e-6 and res["widest_clique"] > 139, res


# ---------------------------------------------------------------- ISAM2UpdateParams: factor removal, constraints, key lists
NO_RELIN = ISAM2Params(ISAM2GaussNewtonParams(0.001), relinearizeThreshold=0.0, relinearizeSkip=0, enableRelinearization=False)


def pair(params):
    p = params
    return (ISAM2(p, ccolamd=ccolamd, device=0),
            oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold))


def both(isam, orc, *args, **kw):
    rg, ro = isam.update(*args, **kw).as_dict(), orc.update(*args, **kw)
    assert rg == ro, (rg, ro)
    assert isam.unusedKeys() == orc.unusedKeys()
    assert isam.num_factors() == orc.num_factors() and isam.size() == len(orc.getLinearizationPoint().keys())
    compare_state(isam, orc)
    return rg


REMOVAL_PARAMS = pytest.mark.parametrize("params", [NO_RELIN, ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1)], ids=["norelin", "relin"])


def slamlike_pair(params):
    isam, orc = pair(params)
    for g, v in slamlike_steps():
        both(isam, orc, g, v)
    return isam, orc


@REMOVAL_PARAMS
def test_remove_factors(params):
    """TEST(ISAM2, removeFactors) tests/testGaussianISAM2.cpp:380-400 (index 12) against the oracle, which tests/test_isam2_oracle.py
    pins to that test's expectation; then the same index again (an empty slot: a no-op) and an index beyond the list (refused)"""
    isam, orc = slamlike_pair(params)
    both(isam, orc, removeFactorIndices=[12])
    assert not isam.factor_exists(12) and isam.factor_exists(11) and isam.unusedKeys() == []
    both(isam, orc, removeFactorIndices=[12])
    with pytest.raises(Exception):
        isam.update(removeFactorIndices=[isam.num_factors()])
    both(isam, orc)
    isam.close()


@REMOVAL_PARAMS
def test_remove_variables(params):
    """TEST(ISAM2, removeVariables) :403-423: indices 7 and 14 are the two measurements of landmark 100, which leaves the system with
    them; afterwards the landmark comes back under its old key with new measurements"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    isam, orc = slamlike_pair(params)
    both(isam, orc, removeFactorIndices=[7, 14])
    assert isam.unusedKeys() == [100] and 100 not in isam.getLinearizationPoint().keys() and 100 not in isam.getDelta()
    assert isam.size() == 13
    both(isam, orc)
    br = noiseModel.Diagonal.Sigmas([np.pi / 100.0, 0.1])
    g, v = NonlinearFactorGraph(), Values()
    g.add_BearingRangeFactor2D(10, 100, np.pi / 4.0 + np.pi / 16.0, 4.5, br)
    g.add_BearingRangeFactor2D(5, 100, np.pi / 4.0, 5.0, br)
    v.insert_point2(100, [5.0 / np.sqrt(2.0), 5.0 / np.sqrt(2.0)])
    both(isam, orc, g, v)
    assert isam.size() == 14 and isam.unusedKeys() == []
    isam.close()


@REMOVAL_PARAMS
def test_swap_factors(params):
    """TEST(ISAM2, swapFactors) :426-476: the 2nd-to-last factor is replaced by one with another range in the same update"""
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    isam, orc = slamlike_pair(params)
    swap_idx = isam.num_factors() - 2
    swap = NonlinearFactorGraph()
    swap.add_BearingRangeFactor2D(10, 100, np.pi / 4.0 + np.pi / 16.0, 5.0, noiseModel.Diagonal.Sigmas([np.pi / 100.0, 0.1]))
    both(isam, orc, swap, Values(), removeFactorIndices=[swap_idx])
    assert isam.num_factors() == swap_idx + 3 and not isam.factor_exists(swap_idx)
    isam.close()


def test_constrained_ordering():
    """TEST(ISAM2, constrained_ordering) :479-571: constrainedKeys {3: 1, 4: 2}; x4 ends up in the root clique"""
    isam, orc = pair(NO_RELIN)
    for g, v, c in constrained_ordering_steps():
        both(isam, orc, g, v, constrainedKeys=c)
        if c is not None:
            assert any(4 in keys[:nfk] for keys, nfk, _, par in isam.cliques() if par < 0)
    isam.close()


def test_no_relin_and_extra_reelim_keys_and_full_solve():
    isam, orc = pair(ISAM2Params(relinearizeThreshold=0.0, relinearizeSkip=1))
    steps = slamlike_steps()
    for g, v in steps[:-1]:
        both(isam, orc, g, v)
    g, v = steps[-1]
    both(isam, orc, g, v, noRelinKeys=[0, 1, 2])
    r = both(isam, orc, extraReelimKeys=[0, 100])
    assert r["variablesReeliminated"] > 0
    both(isam, orc, forceFullSolve=True, force_relinearize=True)
    with pytest.raises(Exception):
        isam.update(extraReelimKeys=[12345])
    both(isam, orc)
    isam.close()


def test_fixed_lag_style_run_with_removals():
    """the first 260 poses of city10000, one pose per update; from pose 150 on every update also removes the factors that reach back
    more than 150 poses from the newest one and pins the oldest remaining pose with a prior at its current estimate (so the system
    stays determined): the keys of removed factors are re-eliminated, poses that lose every factor leave the system.  Counts, unused
    keys and the whole state against the oracle"""
    import os
    from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.datasets import readG2o
    graph, _ = readG2o(os.path.join(os.path.dirname(__file__), "golden", "city10000_head.g2o"))
    edges = []
    for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
        for i, gidx in enumerate(gi.tolist()):
            edges.append((gidx, int(keys[i][0]), int(keys[i][1]), meas[i], models[i]))
    edges.sort()
    isam, orc = pair(ISAM2Params())

    def compose(a, d):
        c, s = np.cos(a[2]), np.sin(a[2])
        return np.array([a[0] + c * d[0] - s * d[1], a[1] + s * d[0] + c * d[1], a[2] + d[2]])

    live = {}  # factor index -> oldest pose it touches
    nxt, step, removed_vars = 0, 1, 0
    while nxt < len(edges) and step <= 260:
        g, v = NonlinearFactorGraph(), Values()
        first = isam.num_factors()
        if step == 1:
            v.insert_pose2(0, 0.0, 0.0, 0.0)
            g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Unit.Create(3))
            live[first] = 0
        while nxt < len(edges):
            _, k1, k2, m, model = edges[nxt]
            if k1 > step or k2 > step:
                break
            live[first + g.size()] = min(k1, k2)
            g.add_BetweenFactorPose2(k1, k2, m, model)
            if k2 == step and k1 == step - 1:
                prev = np.zeros(3) if step == 1 else orc.calculateEstimate().at(step - 1)
                v.insert(step, 0, compose(prev, m))
            nxt += 1
        oldest = step - 150
        old = sorted(i for i, k in live.items() if k < oldest and i < first)
        for i in old:
            del live[i]
        if old:
            live[first + g.size()] = oldest
            g.add_PriorFactorPose2(oldest, orc.calculateEstimate().at(oldest), noiseModel.Diagonal.Sigmas([0.05, 0.05, 0.02]))
        rg, ro = isam.update(g, v, removeFactorIndices=old).as_dict(), orc.update(g, v, removeFactorIndices=old)
        assert rg == ro, (step, rg, ro)
        assert isam.unusedKeys() == orc.unusedKeys(), step
        removed_vars += len(isam.unusedKeys())
        if step % 20 == 0:
            compare_state(isam, orc)
        step += 1
    assert removed_vars > 50 and isam.size() == len(orc.getLinearizationPoint().keys())
    compare_state(isam, orc)
    isam.close()


def test_marginal_covariance():
    """TEST(ISAM2, marginalCovariance) tests/testGaussianISAM2.cpp:977-986 on the device: every variable of the slamlike example (with
    relinearization: the tree is at the linearization point) and of VisualISAM2Example against the oracle's tree; a removed variable and
    an unknown key are refused"""
    isam, orc = slamlike_pair(ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1))
    for k in orc.getLinearizationPoint().keys():
        e = orc.marginalCovariance(k)
        assert np.allclose(isam.marginalCovariance(k), e, rtol=1e-6, atol=1e-9 * np.abs(e).max()), k
    both(isam, orc, removeFactorIndices=[7, 14])
    with pytest.raises(Exception):
        isam.marginalCovariance(100)
    with pytest.raises(Exception):
        isam.marginalCovariance(4242)
    e = orc.marginalCovariance(5)
    assert np.allclose(isam.marginalCovariance(5), e, rtol=1e-6, atol=1e-9 * np.abs(e).max())
    isam.close()
    isam, orc = run_sequence(visual_steps(), ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1), check_every_step=False)
    for k in orc.getLinearizationPoint().keys():
        e = orc.marginalCovariance(k)
        assert np.allclose(isam.marginalCovariance(k), e, rtol=1e-6, atol=1e-9 * np.abs(e).max()), k
    isam.close()


def test_marginal_covariance_through_wide_cliques():
    """the dense pose graph whose cliques exceed an LDS front: the path to the root runs through cliques walked from memory"""
    isam, orc = pair(ISAM2Params())
    for g, v in dense_pose2_steps():
        assert isam.update(g, v).as_dict() == orc.update(g, v)
    assert max(rsd.shape[1] for _, _, rsd, _ in isam.cliques()) > 139
    for k in orc.getLinearizationPoint().keys()[::7]:
        e = orc.marginalCovariance(k)
        assert np.allclose(isam.marginalCovariance(k), e, rtol=1e-6, atol=1e-9 * np.abs(e).max()), k
    isam.close()


def test_cpp_driver_runs_the_incremental_workloads(tmp_path):
    """tests/cpp/isam2_harness: the C ABI driven from C++ in the reference-side wrapper's call order (no Python between the updates) on
    VisualISAM2Example's sequence, on the slamlike sequence followed by the removals of the reference's removeVariables test, and on the
    400-pose incremental pose graph; the final estimate against the oracle, which ran the same updates"""
    import json
    import os
    import subprocess
    from isam2_examples import incremental_pose2_steps, write_isam2_sequence
    from gtsam_personal_amd import NonlinearFactorGraph, Values
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    harness = os.path.join(root, "tests", "cpp", "isam2_harness")
    ccolamd_so = os.path.join(root, "oracle", "_ref", "libccolamd_ref.so")

    def run_case(name, params, steps_iter, orc):
        steps = []
        for st in steps_iter:
            orc.update(st[0], st[1], removeFactorIndices=st[2] if len(st) > 2 else ())
            steps.append(st)
        path = str(tmp_path / f"{name}.txt")
        write_isam2_sequence(path, params, steps)
        out = subprocess.run([harness, path, "0", ccolamd_so], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert out.returncode == 0, out.stdout.decode()[-500:] + out.stderr.decode()[-500:]
        res = json.loads(out.stdout)
        assert res["updates"] == len(steps)
        est = orc.calculateEstimate()
        assert [int(r[0]) for r in res["estimate"]] == est.keys()
        for r in res["estimate"]:
            e = est.at(int(r[0]))
            assert np.allclose(np.array(r[1:])[:len(e)], e[:len(r) - 1], rtol=1e-6, atol=1e-8), (name, r[0])
        assert res["cliques"] == len(orc.cliques())
        return res

    def oracle_of(p):
        return oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)

    p = ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1)
    run_case("visual", p, visual_steps(), oracle_of(p))
    p = ISAM2Params()
    run_case("slamlike_removal", p, [(g, v) for g, v in slamlike_steps()] + [(NonlinearFactorGraph(), Values(), [7, 14])], oracle_of(p))
    orc = oracle_of(p)
    g2o = os.path.join(os.path.dirname(__file__), "golden", "city10000_head.g2o")
    res = run_case("city400", p, incremental_pose2_steps(g2o, 400, lambda k: orc.calculateEstimate().at(k)), orc)
    assert res["variables"] == 400


def test_partial_relinearization_check_and_threshold_vectors():
    """ISAM2Params::enablePartialRelinearizationCheck and relinearizeThreshold as FastMap<char, Vector> (ISAM2-impl.h:246-378): the
    visual example update by update against the oracle with each; a vector of the wrong dimension is refused"""
    for partial, thr in ((True, 0.05), (False, {"x": [0.03, 0.03, 0.03, 0.05, 0.05, 0.05], "l": [0.02, 0.02, 0.02]}),
                         (True, {"x": [0.03, 0.03, 0.03, 0.05, 0.05, 0.05], "l": [0.02, 0.02, 0.02]})):
        p = ISAM2Params(relinearizeThreshold=thr, relinearizeSkip=1, enablePartialRelinearizationCheck=partial)
        isam = ISAM2(p, ccolamd=ccolamd, device=0)
        orc = oh.OracleISAM2(0.1 if isinstance(thr, dict) else thr, 1, True, p.optimizationParams.wildfireThreshold)
        if isinstance(thr, dict):
            orc.set_relinearize_thresholds(thr)
        orc.set_partial_relinearization_check(partial)
        counts = []
        for g, v in visual_steps():
            rg, ro = isam.update(g, v).as_dict(), orc.update(g, v)
            assert rg == ro, (partial, thr, rg, ro)
            counts.append(rg["variablesRelinearized"])
            compare_state(isam, orc)
        assert 0 < min(c for c in counts if c) and max(counts) > 0
        isam.close()
    # the case that separates the two checks (tests/isam2_examples.py: stale_landmark_steps): the landmark is relinearized by the full
    # check only
    for partial in (False, True):
        p = ISAM2Params(relinearizeThreshold=0.05, relinearizeSkip=10, enablePartialRelinearizationCheck=partial)
        isam, orc = pair(p)
        orc.set_partial_relinearization_check(partial)
        counts = [both(isam, orc, g, v)["variablesRelinearized"] for g, v in stale_landmark_steps()]
        assert counts[9] == (2 if partial else 3)
        isam.close()
    p = ISAM2Params(relinearizeThreshold={"x": [0.03] * 6, "l": [0.02] * 2}, relinearizeSkip=1)
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    with pytest.raises(Exception, match="threshold"):
        for g, v in visual_steps():
            isam.update(g, v)
    isam.close()


def test_single_variable_estimate_and_device_driven_incremental_run():
    """calculateEstimate(key) (ISAM2.cpp:757-760) against the whole estimate, after updates with and without pending back-substitution;
    then timing/timeIncremental.cpp's loop with the DEVICE's own estimate of the previous pose initialising the next one, read one
    variable per step, replayed through the oracle afterwards"""
    import os
    from isam2_examples import incremental_pose2_steps
    isam, orc = slamlike_pair(ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1))
    whole = isam.calculateEstimate()
    for k in whole.keys():
        assert np.array_equal(isam.calculateEstimate(k), whole.at(k)), k
    both(isam, orc, removeFactorIndices=[12])
    assert np.allclose(isam.calculateEstimate(11), orc.calculateEstimate().at(11), rtol=1e-6, atol=1e-8)  # (brings the delta up to date itself)
    with pytest.raises(Exception):
        isam.calculateEstimate(4242)
    isam.close()
    p = ISAM2Params()
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    g2o = os.path.join(os.path.dirname(__file__), "golden", "city10000_head.g2o")
    steps = []
    for g, v in incremental_pose2_steps(g2o, 400, lambda k: isam.calculateEstimate(k)):
        isam.update(g, v)
        steps.append((g, v))
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
    for g, v in steps:
        ro = orc.update(g, v)
    compare_state(isam, orc)
    isam.close()


def test_evaluate_nonlinear_error():
    """ISAM2Params::evaluateNonlinearError on the device (errorBefore / errorAfter of every update, lmgpu_isam2_error on demand) against
    the oracle on the visual example (projection factors, priors) and on the slamlike sequence followed by a removal (the removed
    factor's error must no longer count), a removal of the empty slot and a bare update"""
    for steps, tail in ((visual_steps(), []), (slamlike_steps(), [dict(removeFactorIndices=[12]), dict(removeFactorIndices=[12]), dict()])):
        p = ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1, evaluateNonlinearError=True)
        isam, orc = pair(p)
        orc.set_evaluate_nonlinear_error(True)
        for g, v in steps:
            r = isam.update(g, v)
            assert r.as_dict() == orc.update(g, v)
            eb, ea = orc.errors()
            assert abs(r.errorBefore - eb) <= 1e-6 * max(1e-9, abs(eb)) + 1e-12, (r.errorBefore, eb)
            assert abs(r.errorAfter - ea) <= 1e-6 * max(1e-9, abs(ea)) + 1e-12, (r.errorAfter, ea)
        for kw in tail:
            r = isam.update(**kw)
            assert r.as_dict() == orc.update(**kw)
            eb, ea = orc.errors()
            assert abs(r.errorBefore - eb) <= 1e-6 * abs(eb) + 1e-12 and abs(r.errorAfter - ea) <= 1e-6 * abs(ea) + 1e-12
        for which in (0, 2):
            assert abs(isam.error(which) - orc.error(which)) <= 1e-6 * abs(orc.error(which)) + 1e-12
        compare_state(isam, orc)
        isam.close()


@pytest.mark.parametrize("mode,radius", [(0, 1.0), (2, 0.05), (1, 0.5)])
def test_dogleg(mode, radius):
    """ISAM2DoglegParams on the device (Powell's dog leg in updateDelta, ISAM2.cpp:739-779) against the oracle, whose restatement
    tests/test_isam2_oracle.py pins to TEST(ISAM2, slamlike_solution_dogleg): update by update the same tree, delta (= the dog-leg step),
    estimate and trust-region radius -- the slamlike sequence without and the visual example with relinearization; the three adaptation
    modes, a radius that cuts the first steps included"""
    for steps, kw in ((slamlike_steps(), dict(relinearizeThreshold=0.0, relinearizeSkip=0, enableRelinearization=False)),
                      (visual_steps(), dict(relinearizeThreshold=0.01, relinearizeSkip=1))):
        p = ISAM2Params(ISAM2DoglegParams(radius, 1e-5, mode), **kw)
        isam = ISAM2(p, ccolamd=ccolamd, device=0)
        orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, 1e-5)
        orc.set_dogleg(radius, 1e-5, mode)
        for g, v in steps:
            rg, ro = isam.update(g, v).as_dict(), orc.update(g, v)
            assert rg == ro, (rg, ro)
            compare_state(isam, orc)  # (calculateEstimate -> updateDelta -> one dog-leg iteration on both sides)
            assert abs(isam.doglegDelta() - orc.doglegDelta()) <= 1e-6 * orc.doglegDelta()
        if steps[0][1].exists(0):  # the slamlike sequence: landmark 100 loses its factors and leaves (its scalars must stop counting in
            both(isam, orc, removeFactorIndices=[7, 14])  # the norms of the dog leg), then a bare update
            both(isam, orc)
            assert abs(isam.doglegDelta() - orc.doglegDelta()) <= 1e-6 * orc.doglegDelta()
        isam.close()


def test_dogleg_through_wide_cliques():
    """the dog leg's tree products (gradient, R g, error) on cliques walked from memory (more than 139 columns), update by update against
    the oracle on the dense pose graph"""
    p = ISAM2Params(ISAM2DoglegParams(1.0, 1e-5, 0))
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, 1e-5)
    orc.set_dogleg(1.0, 1e-5, 0)
    for i, (g, v) in enumerate(dense_pose2_steps()):
        assert isam.update(g, v).as_dict() == orc.update(g, v)
        if i % 10 == 9:
            compare_state(isam, orc)
            assert abs(isam.doglegDelta() - orc.doglegDelta()) <= 1e-6 * orc.doglegDelta()
    assert max(rsd.shape[1] for _, _, rsd, _ in isam.cliques()) > 139
    compare_state(isam, orc)
    isam.close()


def test_dogleg_incremental_pose_graph():
    """the 400-pose incremental pose graph (loop closures, default relinearization) with ISAM2DoglegParams: counts per update, the state and
    the radius every 50 updates and at the end, against the oracle"""
    import os
    from isam2_examples import incremental_pose2_steps
    p = ISAM2Params(ISAM2DoglegParams(1.0, 1e-5, 0))
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, 1e-5)
    orc.set_dogleg(1.0, 1e-5, 0)
    g2o = os.path.join(os.path.dirname(__file__), "golden", "city10000_head.g2o")
    last = {"k": None}
    for i, (g, v) in enumerate(incremental_pose2_steps(g2o, 400, lambda k: last["k"])):
        rg, ro = isam.update(g, v).as_dict(), orc.update(g, v)
        assert rg == ro, (i, rg, ro)
        last["k"] = orc.calculateEstimate().at(i + 1) if i % 7 == 0 else (v.at(i + 1) if v.exists(i + 1) else last["k"])
        if i % 50 == 49:
            compare_state(isam, orc)
            assert abs(isam.doglegDelta() - orc.doglegDelta()) <= 1e-6 * orc.doglegDelta()
    compare_state(isam, orc)
    isam.close()


def test_batch_step_with_a_variable_that_has_no_factor_yet():
    """recalculateBatch orders the keys of variableIndex_ only (gtsam/nonlinear/ISAM2.cpp:186-190): a variable that came with a value but no
    factor stays out of the tree until a factor reaches it.  Here the first update (always a batch step) carries such a pose; the
    second one ties it in.  Update by update against the oracle."""
    odo = noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.05])
    steps = []
    g, v = NonlinearFactorGraph(), Values()
    g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], odo)
    g.add_BetweenFactorPose2(0, 1, [1.0, 0.0, 0.0], odo)
    v.insert_pose2(0, 0.01, 0.01, 0.01)
    v.insert_pose2(1, 1.1, -0.1, 0.01)
    v.insert_pose2(2, 2.1, 0.1, 0.0)  # no factor yet
    steps.append((g, v))
    g, v = NonlinearFactorGraph(), Values()
    g.add_BetweenFactorPose2(1, 2, [1.0, 0.0, 0.0], odo)
    steps.append((g, v))
    g, v = NonlinearFactorGraph(), Values()
    g.add_BetweenFactorPose2(2, 3, [1.0, 0.0, 0.0], odo)
    v.insert_pose2(3, 3.0, 0.0, 0.0)
    steps.append((g, v))
    isam, _ = run_sequence(steps, ISAM2Params(ISAM2GaussNewtonParams(0.001), 0.01, 1, True))
    isam.close()


def test_refused_update_leaves_the_handle_usable():
    """an update whose input is refused (a factor on a variable without a value) changes nothing and drops the pending input: the next,
    correct update runs as if the refused one had never been issued"""
    from gtsam_personal_amd import _lib
    odo = noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.05])
    isam = ISAM2(ISAM2Params(), ccolamd=ccolamd, device=0)
    g, v = NonlinearFactorGraph(), Values()
    g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], odo)
    v.insert_pose2(0, 0.0, 0.0, 0.0)
    isam.update(g, v)
    bad, bv = NonlinearFactorGraph(), Values()
    bad.add_BetweenFactorPose2(0, 7, [1.0, 0.0, 0.0], odo)  # pose 7 has no value
    bv.insert_pose2(1, 1.0, 0.0, 0.0)
    with pytest.raises(_lib.LmgpuError):
        isam.update(bad, bv)
    assert len(isam.calculateEstimate().keys()) == 1  # pose 1 of the refused update was not committed
    g, v = NonlinearFactorGraph(), Values()
    g.add_BetweenFactorPose2(0, 1, [1.0, 0.0, 0.0], odo)
    v.insert_pose2(1, 1.0, 0.0, 0.0)
    isam.update(g, v)
    est = isam.calculateEstimate()
    assert sorted(est.keys()) == [0, 1] and np.allclose(est.at(1), [1.0, 0.0, 0.0], atol=1e-6)
    isam.close()


def test_robust_noise_in_the_incremental_path():
    """noiseModel::Robust around the factors' Gaussian models (Huber on the odometry, Cauchy on the bearing-range measurements of the
    slamlike sequence, one landmark measurement a gross outlier): relinearization reweights the cached [A b] like the batch path, update
    by update against the oracle.  The reference holds no incremental test with robust noise (parity pinned by the oracle's batch
    equivalence, tests/test_isam2_oracle.py, and by the m-estimator known answers of testNoiseModel.cpp in test_oracle_golden.py)."""
    mEstimator = noiseModel.mEstimator
    odo = noiseModel.Robust.Create(mEstimator.Huber.Create(1.0), noiseModel.Diagonal.Sigmas([0.1, 0.1, np.pi / 100.0]))
    br = noiseModel.Robust.Create(mEstimator.Cauchy.Create(0.5), noiseModel.Diagonal.Sigmas([np.pi / 100.0, 0.1]))
    steps = []
    g, v = NonlinearFactorGraph(), Values()
    g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.05]))
    v.insert_pose2(0, 0.01, 0.01, 0.01)
    steps.append((g, v))
    for i in range(8):
        g, v = NonlinearFactorGraph(), Values()
        g.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], odo)
        v.insert_pose2(i + 1, float(i + 1) + 0.1, -0.1, 0.01)
        if i == 2:
            g.add_BearingRangeFactor2D(i, 100, np.pi / 4.0, 5.0, br)
            v.insert_point2(100, [2.0 + 5.0 / np.sqrt(2.0), 5.0 / np.sqrt(2.0)])
        if i == 4:
            g.add_BearingRangeFactor2D(i, 100, np.pi / 2.0 + 0.9, 9.0, br)  # an outlier: the Cauchy weight takes it out
        if i == 6:
            g.add_BearingRangeFactor2D(i, 100, 3.0 * np.pi / 4.0 + 0.35, 4.3, br)
        steps.append((g, v))
    isam, orc = run_sequence(steps, ISAM2Params(ISAM2GaussNewtonParams(0.001), 0.01, 1, True))
    isam.close()
