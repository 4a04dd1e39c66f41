"""ISAM2::marginalizeLeaves of the oracle, pinned the way the reference pins its own (tests/testGaussianISAM2.cpp:617-975): the marginal of
the Bayes tree and of the linearized factor graph on the kept variables must not change when leaves are marginalized out.

The reference's marginalizeLeaves1-4 use scalar (double) variables; the oracle has no 1-d variable type, so they are restated on Pose2
with the same graph shapes and the same constrained ordering (the tree structure, which is what they exercise, is the same)."""
import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import NonlinearFactorGraph, Values
from gtsam_personal_amd.graph import VAR_DIM
from gtsam_personal_amd import noiseModel
from isam2_examples import slamlike_steps


def marginal_of(H, lin, leaf_keys):
    """GaussianFactorGraph::marginal(toKeep)->augmentedHessian() of a dense augmented Hessian whose variables ascend by key"""
    leaf, keep, o = [], [], 0
    for k in lin.keys():
        idx = list(range(o, o + VAR_DIM[lin.type(k)]))
        (leaf if k in leaf_keys else keep).extend(idx)
        o += len(idx)
    keep.append(o)
    if not leaf:
        return H[np.ix_(keep, keep)]
    A, B, C = H[np.ix_(leaf, leaf)], H[np.ix_(leaf, keep)], H[np.ix_(keep, keep)]
    return C - B.T @ np.linalg.solve(A, B)


def check_marginalize_leaves(isam, leaf_keys, tol=1e-6):
    """checkMarginalizeLeaves tests/testGaussianISAM2.cpp:617-660"""
    leaf_keys = set(int(k) for k in leaf_keys)
    lin = isam.getLinearizationPoint()
    expected = marginal_of(isam.tree_augmented_hessian(), lin, leaf_keys)
    expected3 = marginal_of(isam.graph_augmented_hessian(), lin, leaf_keys)
    marg_idx, del_idx = isam.marginalizeLeaves(sorted(leaf_keys))
    actual = isam.tree_augmented_hessian()
    actual3 = isam.graph_augmented_hessian()
    assert np.all(np.isfinite(actual))
    assert set(isam.getLinearizationPoint().keys()).isdisjoint(leaf_keys)
    np.testing.assert_allclose(actual, expected, rtol=0, atol=tol)  # treeEqual
    a = actual.copy()
    a[-1, -1] = expected3[-1, -1]
    np.testing.assert_allclose(a, expected3, rtol=0, atol=tol)  # nonlinEqual
    np.testing.assert_allclose(actual3, expected3, rtol=0, atol=tol)  # afterNonlinCorrect
    for i in del_idx:  # (with findUnusedFactorSlots a marginal factor may sit in a slot this call has just freed)
        assert i in marg_idx or not isam.factor_exists(i)
    for i in marg_idx:
        assert isam.marginal_factor(i) is not None
    return marg_idx, del_idx


MODEL = noiseModel.Isotropic.Sigma(3, 1.0)


def chain_isam(n, between, prior=(0,)):
    g, v = NonlinearFactorGraph(), Values()
    for k in prior:
        g.add_PriorFactorPose2(k, [0.0, 0.0, 0.0], MODEL)
    for a, b in between:
        g.add_BetweenFactorPose2(a, b, [0.0, 0.0, 0.0], MODEL)
    for k in range(n):
        v.insert_pose2(k, 0.0, 0.0, 0.0)
    isam = oh.OracleISAM2()
    isam.update(g, v, constrainedKeys={k: k for k in range(n)})
    return isam


@pytest.mark.parametrize("n,between,leaf", [
    (3, [(0, 1), (1, 2), (0, 2)], [0]),                                    # marginalizeLeaves1 :736-759
    (4, [(0, 1), (1, 2), (0, 2), (2, 3)], [0]),                            # marginalizeLeaves2 :762-788
    (6, [(0, 1), (1, 2), (0, 2), (2, 3), (3, 4), (4, 5), (3, 5)], [0]),    # marginalizeLeaves3 :791-826
    (3, [(0, 2), (1, 2)], [1]),                                            # marginalizeLeaves4 :829-851
])
def test_marginalize_leaves_small(n, between, leaf):
    isam = chain_isam(n, between)
    marg_idx, del_idx = check_marginalize_leaves(isam, leaf)
    assert marg_idx and del_idx
    fixed = isam.getFixedVariables()
    for i in marg_idx:
        assert set(isam.marginal_factor(i)[0]) <= set(fixed)
    # the next update leaves the fixed variables' linearization point alone and still solves
    g = NonlinearFactorGraph()
    keep = [k for k in range(n) if k not in leaf]
    g.add_PriorFactorPose2(keep[-1], [0.3, 0.0, 0.0], MODEL)
    before = isam.getLinearizationPoint()
    isam.update(g, None, force_relinearize=True)
    after = isam.getLinearizationPoint()
    for k in fixed:
        assert np.array_equal(before.at(k), after.at(k))
    assert np.all(np.isfinite(np.concatenate(list(isam.getDelta().values()))))


def test_marginalize_leaves_slamlike():
    """marginalizeLeaves5 :854-862"""
    isam = oh.OracleISAM2()
    for g, v in slamlike_steps():
        isam.update(g, v)
    check_marginalize_leaves(isam, [0])


def marked_keys_for(isam, marginalizable):
    """updateAndMarginalize's additional keys (tests/testGaussianISAM2.cpp:685-718): the frontals of every clique below the key's clique
    whose separator holds the key"""
    cl = isam.cliques()
    children = {i: [] for i in range(len(cl))}
    for i, (_, _, _, par) in enumerate(cl):
        if par >= 0:
            children[par].append(i)
    marked = []
    for key in sorted(marginalizable):
        marked.append(key)
        home = next(i for i, (keys, nf, _, _) in enumerate(cl) if key in keys[:nf])
        stack = list(children[home])
        while stack:
            i = stack.pop()
            keys, nf, _, _ = cl[i]
            if key in keys[nf:]:
                marked.extend(keys[:nf])
                stack.extend(children[i])
    return marked


def update_and_marginalize(isam, g, v, marginalizable, tol=1e-6):
    """updateAndMarginalize :705-726"""
    constrained = None
    if marginalizable:
        constrained = {int(k): 1 for k in isam.getDelta()}
        for k in (v.keys() if v is not None else []):
            constrained[int(k)] = 1
        for k in marginalizable:
            constrained[int(k)] = 0
    marked = marked_keys_for(isam, marginalizable) if marginalizable else []
    isam.update(g, v, constrainedKeys=constrained, extraReelimKeys=marked)
    if marginalizable:
        check_marginalize_leaves(isam, marginalizable, tol)


def pose3_grid(dim):
    nm = noiseModel.Isotropic.Sigma(6, 1.0)
    g, v = NonlinearFactorGraph(), Values()
    I = np.eye(3)
    for i in range(dim):
        for j in range(dim):
            key = i * dim + j
            g.add_PriorFactorPose3(key, I, [float(i), float(j), 0.0], nm)
            v.insert_pose3(key, I, [float(i), float(j), 0.0])
            if i > 0:
                g.add_BetweenFactorPose3((i - 1) * dim + j, key, I, [1.0, 0.0, 0.0], nm)
            if j > 0:
                g.add_BetweenFactorPose3(i * dim + j - 1, key, I, [0.0, 1.0, 0.0], nm)
    return g, v


@pytest.mark.parametrize("dim", [4, 10])
def test_marginalize_leaves_grid_one_by_one(dim):
    """marginalizeLeaves6 :865-905: a grid of Pose3, every variable marginalized one at a time in a shuffled order (the reference shuffles with
    std::default_random_engine(1234); any fixed permutation exercises the same code)"""
    g, v = pose3_grid(dim)
    isam = oh.OracleISAM2()
    update_and_marginalize(isam, g, v, [])
    order = np.random.default_rng(1234).permutation(dim * dim)
    for key in order.tolist():
        update_and_marginalize(isam, None, None, [key], tol=1e-6)
        est = isam.calculateBestEstimate()
        assert key not in est.keys()
    assert len(isam.calculateBestEstimate().keys()) == 0


def test_marginalize_root():
    """TEST(ISAM2, MarginalizeRoot) :908-933"""
    nm = noiseModel.Isotropic.Sigma(6, 1.0)
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose3(0, np.eye(3), [0.0, 0.0, 0.0])
    g.add_PriorFactorPose3(0, np.eye(3), [0.0, 0.0, 0.0], nm)
    isam = oh.OracleISAM2()
    update_and_marginalize(isam, g, v, [])
    assert len(isam.calculateBestEstimate().keys()) == 1
    update_and_marginalize(isam, None, None, [0])
    assert len(isam.calculateBestEstimate().keys()) == 0


def test_marginalization_size():
    """TEST(ISAM2, marginalizationSize) :936-966: with findUnusedFactorSlots the marginal factor takes a freed slot"""
    nm = noiseModel.Isotropic.Sigma(6, 1.0)
    g, v = NonlinearFactorGraph(), Values()
    v.insert_pose3(0, np.eye(3), [0.0, 0.0, 0.0])
    g.add_PriorFactorPose3(0, np.eye(3), [0.0, 0.0, 0.0], nm)
    v.insert_pose3(1, np.eye(3), [0.0, 0.0, 0.0])
    g.add_BetweenFactorPose3(0, 1, np.eye(3), [0.0, 0.0, 0.0], nm)
    isam = oh.OracleISAM2()
    isam.set_find_unused_factor_slots(True)
    update_and_marginalize(isam, g, v, [])
    n = isam.num_factors()
    update_and_marginalize(isam, None, None, [0])
    assert isam.num_factors() == n
    # and an update's new factors fill the other freed slot
    g2 = NonlinearFactorGraph()
    g2.add_PriorFactorPose3(1, np.eye(3), [0.0, 0.0, 0.0], nm)
    isam.update(g2, None)
    assert isam.num_factors() == n and all(isam.factor_exists(i) for i in range(n))


def test_marginalize_non_leaf_is_refused():
    """the reference's debug-build exception (ISAM2.cpp:516-523): a clique below the marginalized key holds a variable that stays"""
    g, v = pose3_grid(4)
    isam = oh.OracleISAM2()
    isam.update(g, v)
    cl = isam.cliques()
    key = next(keys[0] for i, (keys, nf, _, par) in enumerate(cl) if any(p == i and keys[0] in k2[n2:] for k2, n2, _, p in cl))
    with pytest.raises(RuntimeError):
        isam.marginalizeLeaves([key])
