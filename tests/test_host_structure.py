"""CPU tests of the product's host logic: the C-ABI library loads and exports every declared symbol, and its
symbolic analysis (variable index -> elimination tree -> junction tree -> fronts, Scatter order) reproduces
the oracle's Bayes-tree structure clique for clique ("bit-exact variable ordering/indexing").
No compute entry point is called here (there is no GPU)."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import LevenbergMarquardtOptimizer, NonlinearFactorGraph, Ordering, Values, _lib, noiseModel
from gtsam_personal_amd.datasets import SfmData, bal_graph
from gtsam_personal_amd.synthetic import make_bal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "lmgpu.h")).read()
    hooks = re.search(r"#ifdef LMGPU_TEST_HOOKS(.*?)#endif", header, re.S)
    test_declared = set(re.findall(r"\b(lmgpu_[a-z0-9_]+)\s*\(", hooks.group(1)))
    header = header.replace(hooks.group(0), "")
    declared = set(re.findall(r"\b(lmgpu_[a-z0-9_]+)\s*\(", header))
    declared -= {"lmgpu_handle"}
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    # the test hooks: declared under LMGPU_TEST_HOOKS, exported by liblmgpu_test.so, absent from the product library
    assert test_declared == set(_lib.TEST_SYMBOLS)
    for name in test_declared:
        assert not hasattr(lib, name), name
        assert hasattr(_lib.load(test_hooks=True), name), name


def test_compute_without_device_fails_loudly():
    graph, initial, _, ordering = make_bal(n_cam=3, n_pt=5, obs_per_point=2, seed=1)
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, device=-1)
    with pytest.raises(_lib.LmgpuError, match="no HIP device"):
        opt.set_values(initial)
    with pytest.raises(_lib.LmgpuError):
        opt.graph_error()


def _structure_matches(graph, initial, ordering):
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, device=-1)
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.linearize()
    rc, _, _, _ = orc.solve(1.0)
    assert rc == 0
    cl = orc.cliques()
    assert opt.num_fronts() == len(cl)
    for i, (keys, nfk, rsd, parent) in enumerate(cl):
        fk, _ = opt.front(i, numeric=False)
        info = opt.front_info(i)
        assert fk == keys, (i, fk, keys)
        assert info["n_frontal_keys"] == nfk
        assert (info["nf"], info["n"]) == rsd.shape
        assert info["parent"] == parent
    return opt, cl


def test_fronts_match_oracle_bal_schur_and_natural():
    graph, initial, _, ordering = make_bal(n_cam=5, n_pt=30, obs_per_point=3, seed=2)
    opt, cl = _structure_matches(graph, initial, ordering)
    # BAL with points first: one clique per point + the camera root
    assert len(cl) == 31
    _structure_matches(graph, initial, Ordering.Natural(graph))


@pytest.mark.skipif(not oh.have_ref(), reason="oracle/_ref not built")
def test_fronts_match_oracle_colamd_metis_dubrovnik():
    db = SfmData.FromBalFile(os.path.join(GOLD, "dubrovnik-3-7-pre.txt"))
    graph, initial = bal_graph(db)
    _structure_matches(graph, initial, oh.colamd(graph))
    _structure_matches(graph, initial, oh.metis(graph))


def _pose2_grid(n=6, seed=0):
    rng = np.random.default_rng(seed)
    graph, initial = NonlinearFactorGraph(), Values()
    idx = lambda i, j: i * n + j  # noqa: E731
    for i in range(n):
        for j in range(n):
            initial.insert_pose2(idx(i, j), j + rng.normal(0, 0.05), i + rng.normal(0, 0.05), rng.normal(0, 0.05))
    model = noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.05])
    for i in range(n):
        for j in range(n):
            if j + 1 < n:
                graph.add_BetweenFactorPose2(idx(i, j), idx(i, j + 1), [1.0, 0.0, 0.0], model)
            if i + 1 < n:
                graph.add_BetweenFactorPose2(idx(i, j), idx(i + 1, j), [0.0, 1.0, 0.0], model)
    graph.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Diagonal.Sigmas([0.01, 0.01, 0.01]))
    return graph, initial


@pytest.mark.skipif(not oh.have_ref(), reason="oracle/_ref not built")
def test_fronts_match_oracle_pose2_grid_all_orderings():
    graph, initial = _pose2_grid()
    for ordering in (oh.colamd(graph), oh.metis(graph), Ordering.Natural(graph), Ordering.Natural(graph)[::-1]):
        _structure_matches(graph, initial, ordering)


@pytest.mark.parametrize("far_pct", [100, 70, 55, 30, 10, 1100, 1070, 1050, 1030, 1010, 101050, 10101050, 3101030, 8101100])
def test_chained_launch_ticket_order_is_a_topological_order(far_pct):
    """The chained factorisation launch of a dense front (kernels_step.hpp) hands its logical workgroups out by ticket; a
    workgroup spins for workgroups it depends on, so every dependency must hold an earlier ticket or the launch could hang.
    The library checks its own schedule on the host (no GPU): every workgroup of every step exactly once, dependencies
    earlier.  far_pct + 1000 = the schedule that applies the update in pairs of steps (one pass of depth 512 per pair);
    + 100000 (p + 1): with p percent (instead of 60) of a block's update tasks in front of its row-panel workgroups.  Front shapes: the C4 root (9001 x 9001), the 600-camera root, separator-heavy and small fronts."""
    lib = _lib.load()
    checked = 0
    for n, nf in [(9001, 9000), (5401, 5400), (2000, 1500), (1081, 1080), (30000, 29000), (10000, 4000), (777, 770), (4097, 4096)]:
        n_panels = (nf + 255) // 256
        rows = lambda i: min(nf, (i + 1) * 256) - i * 256
        steps = [i for i in range(n_panels - 1) if rows(i) == 256 and rows(i + 1) % 64 == 0 and n - (i + 1) * 256 > 0]
        for first, count in ((0, len(steps)), (1, len(steps) - 1), (0, 2)):
            if count >= 2 and first + count <= len(steps):
                assert lib.lmgpu_selftest_chain_schedule(n, nf, steps[first], count, far_pct) == 0, (n, nf, first, count)
                checked += 1
    assert checked >= 12
    assert lib.lmgpu_selftest_chain_schedule(100, 200, 0, 2, far_pct) != 0  # nonsense geometry is refused, not "valid"
