"""Multi-panel dense fronts (64 cameras -> 577 x 577 root with 3 outer panels, up to 600 cameras -> 5401 x 5401): the dense
path (dataflow panels, fused steps, the chained launch and its ticket schedule) against the oracle and, beyond the oracle's
reach, its launch forms against each other (LMGPU_CHAIN_FAR / LMGPU_NO_CHAIN / LMGPU_NO_FUSE / LMGPU_PANEL_2L)."""
import os

import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
from gtsam_personal_amd.synthetic import make_bal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_cam", [64, int(os.environ.get("LMGPU_TEST_NCAM", "200"))])
def test_outer_panels_match_oracle(n_cam):
    graph, initial, _, ordering = make_bal(n_cam=n_cam, n_pt=25 * n_cam, obs_per_point=8, seed=17)
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    info = opt.front_info(opt.num_fronts() - 1)
    assert info["cls"] == 1 and info["nf"] == 9 * n_cam
    opt.set_kernel_timing(bool(os.environ.get("LMGPU_TEST_TIMERS")))
    opt.linearize()
    orc.linearize()
    for lam in (1e-5, 1e-2):
        dk, d, e0, e1 = opt.solve(lam)
        rc, do, o0, o1 = orc.solve(lam)
        assert rc == 0
        a = np.concatenate([dk[k] for k in sorted(dk)])
        b = np.concatenate([do[k] for k in sorted(do)])
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel < 1e-6, (lam, rel)
        assert abs(e1 - o1) <= 1e-6 * max(1.0, abs(o1))


def test_launch_forms_agree_on_a_large_root(monkeypatch, dev_switches):
    """Beyond the oracle's reach (600 cameras -> 5401 x 5401 root, 22 outer panels, 15 000 leaf fronts): the default path
    (dataflow panels fused into the trailing-update launches) against the same arithmetic issued as separate launches
    (LMGPU_NO_FUSE) and with the two-launch panel form (LMGPU_PANEL_2L) -- three different synchronisation structures, one
    result."""
    graph, initial, _, ordering = make_bal(n_cam=600, n_pt=15000, obs_per_point=8, seed=23)
    params = LevenbergMarquardtParams()

    def run(env):
        for k in ("LMGPU_NO_FUSE", "LMGPU_PANEL_2L", "LMGPU_NO_CHAIN", "LMGPU_CHAIN_FAR", "LMGPU_NO_TAIL", "LMGPU_NO_GATHER_WRITE", "LMGPU_NO_INV16_REUSE",
                  "LMGPU_NO_LEAFPACK", "LMGPU_NO_MERGE", "LMGPU_SCHUR_UNMASKED"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, "100" if k == "LMGPU_CHAIN_FAR" else "1")  # CHAIN_FAR=100: plain step order in the chained launch
        opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)  # the switches are read at creation
        e0 = opt.error()
        trace = []
        for _ in range(2):
            opt.iterate()
            trace.append((opt.error(), opt.lambda_(), opt.getInnerIterations()))
        dk, d, _, _ = (opt.linearize(), opt.solve(1e-4))[1]
        opt.close()
        return e0, trace, d

    base = run([])
    assert base[1][-1][0] < 0.05 * base[0]
    # (+ the remaining A/B switches of this front: the end of the front as separate launches, the gather adding into a cleared front
    #  instead of writing it, the 16x16 inverses recomputed for the back-substitution, LDS-front descriptors unpacked)
    for env in (["LMGPU_CHAIN_FAR"], ["LMGPU_NO_MERGE"], ["LMGPU_NO_MERGE", "LMGPU_CHAIN_FAR"], ["LMGPU_NO_CHAIN"], ["LMGPU_NO_FUSE"], ["LMGPU_PANEL_2L"], ["LMGPU_NO_FUSE", "LMGPU_PANEL_2L"], ["LMGPU_NO_TAIL"],
                ["LMGPU_NO_GATHER_WRITE"], ["LMGPU_NO_INV16_REUSE"], ["LMGPU_NO_LEAFPACK"], ["LMGPU_SCHUR_UNMASKED"]):
        other = run(env)
        assert other[0] == base[0]
        for a, b in zip(other[1], base[1]):
            assert abs(a[0] - b[0]) <= 1e-9 * abs(b[0]) and a[2] == b[2] and abs(a[1] - b[1]) <= 1e-12 * b[1], (env, a, b)
        rel = np.linalg.norm(other[2] - base[2]) / np.linalg.norm(base[2])
        # (the last four change the ORDER of a few sums -- the tail kernel adds the last 40 columns' update in plain fused multiply-adds, the
        #  gather subtracts from a front that already holds the own factors instead of writing first -- and get rounding-level room)
        #  the chained launch applies the update two panels per pass (one sum of depth 512 instead of two of depth 256), which every form
        #  that updates panel by panel rounds differently.  Only another ticket order of the same passes is held to 1e-9.
        loose = env != ["LMGPU_CHAIN_FAR"]
        assert rel < (5e-8 if loose else 1e-9), (env, rel)


@pytest.mark.parametrize("n_cam", [16, 17, 21, 22, 28, 29, 35, 36, 42, 43, 50, 56, 57, 71, 72, 85, 86, 100])
def test_root_sizes_across_block_boundaries(n_cam):
    """Camera roots of 9 n_cam + 1 columns for n_cam chosen so that nf = 9 n_cam sweeps the residues mod 64 / 128 / 256 that
    select the code paths of the dense front: medium (one-panel) batched path vs. multi-panel path, dataflow vs. two-launch
    panels, 1-4 block columns, head-tile strips S < 4, one or two 128-tiles of trailing matrix, partial last blocks."""
    graph, initial, _, ordering = make_bal(n_cam=n_cam, n_pt=12 * n_cam, obs_per_point=5, seed=100 + n_cam)
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    opt.linearize()
    orc.linearize()
    for lam in (1e-6, 1e-1):
        dk, d, e0, e1 = opt.solve(lam)
        rc, do, o0, o1 = orc.solve(lam)
        assert rc == 0
        a = np.concatenate([dk[k] for k in sorted(dk)])
        b = np.concatenate([do[k] for k in sorted(do)])
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel < 1e-6, (n_cam, lam, rel)
        assert abs(e1 - o1) <= 1e-6 * max(1.0, abs(o1))
    cl = orc.cliques()
    keys, nfk, rsd, parent = cl[-1]
    fk, R = opt.front(opt.num_fronts() - 1)
    assert fk == keys
    assert np.allclose(R, rsd, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(rsd).max()))


def test_graph_replay_and_merged_backsubstitution_agree_with_eager_launches(monkeypatch, dev_switches):
    """Deep clique trees replay their solve as a hipGraph and back-substitute consecutive levels of small fronts in one dataflow
    launch; both only change HOW the same kernels / the same per-front arithmetic are issued.  victoria_park (first 1500 poses,
    natural ordering: a tree of hundreds of levels): repeated solves with different lambda in every mode give the same update."""
    from gtsam_personal_amd import Ordering, noiseModel
    from gtsam_personal_amd.datasets import load2D
    graph, initial = load2D(os.path.join(os.path.dirname(__file__), "golden", "victoria_park.txt"), max_index=1500)
    graph.add_PriorFactorPose2(0, initial.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering(sorted(initial.keys()))
    results = {}
    for mode in ((0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 0)):
        monkeypatch.setenv("LMGPU_GRAPH", str(mode[0]))
        monkeypatch.setenv("LMGPU_MERGE_BACKSUB", str(mode[1]))
        monkeypatch.setenv("LMGPU_MERGE_ELIM", str(mode[2]))  # round 3: the same on the way up (a merged launch runs every front with its four waves)
        opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
        opt.linearize()
        out = []
        for lam in (1e-5, 1e-2, 1e-5, 1.0, 1e-2):  # the replay is captured at the second solve; lambda must follow each call
            _, d, e0, e1 = opt.solve(lam)
            out.append((d.copy(), e1))
        opt.close()
        results[mode] = out
    base = results[(0, 0, 1)]
    assert np.linalg.norm(base[0][0] - base[1][0]) > 1e-6 * np.linalg.norm(base[0][0])  # the lambdas do make a difference
    assert np.array_equal(base[0][0], base[2][0]) and np.array_equal(base[1][0], base[4][0])
    for mode, out in results.items():
        # per-level launches take a small front's pivots one by one in LDS; a merged launch keeps a front of <= 16 columns in the registers of one wave
        # and sums its factors on the matrix core (other rounding; hundreds of levels deep, natural ordering: 1e-8)
        tol = 1e-12 if mode[2] == 1 else 5e-8
        for (d, e1), (d0, e10) in zip(out, base):
            assert np.linalg.norm(d - d0) <= tol * np.linalg.norm(d0), mode
            assert abs(e1 - e10) <= tol * max(1.0, abs(e10)), mode


def test_deep_tree_launch_forms_agree(monkeypatch, dev_switches):
    """Round-2 forms of the deep-tree kernels against each other on sphere2500 (20 levels, mid-size fronts of up to 546 columns): the
    block back-substitution with and without tickets (LMGPU_BSD_TICKET: the path of levels with more 64-row blocks than CUs), LDS fronts
    with four and with sixteen waves (LMGPU_NO_WIDE16), mid-size fronts batched per level or one by one (LMGPU_NO_MED: the trailing
    update as 128-tiles / quadrants of the per-front path); round 3: the LDS fronts of consecutive levels in one dataflow launch
    (LMGPU_MERGE_ELIM) and a level's LDS fronts + medium fronts in one launch (LMGPU_FUSE_LEVELS) against the per-level launches.
    Same arithmetic per entry except where a reduction is cut differently."""
    from gtsam_personal_amd import noiseModel
    from gtsam_personal_amd.datasets import chain_initial_pose3, load3D
    graph, _ = load3D(os.path.join(os.path.dirname(__file__), "golden", "sphere2500.txt"))
    initial = chain_initial_pose3(graph)
    graph.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "slam_orderings.npz"))
    keys = np.array(sorted(graph.keys()), dtype=np.uint64)
    ordering = [int(k) for k in keys[fx["sphere2500_metis"]]]
    results = {}
    for mode in ("default", "LMGPU_BSD_TICKET", "LMGPU_NO_WIDE16", "LMGPU_NO_MED", "LMGPU_MERGE_ELIM", "LMGPU_FUSE_LEVELS"):
        for sw in ("LMGPU_BSD_TICKET", "LMGPU_NO_WIDE16", "LMGPU_NO_MED", "LMGPU_MERGE_ELIM", "LMGPU_FUSE_LEVELS"):
            monkeypatch.delenv(sw, raising=False)
        if mode != "default":  # (round 3: the merged elimination launches and the fused level launches are on by default: switched OFF here)
            monkeypatch.setenv(mode, "0" if mode in ("LMGPU_MERGE_ELIM", "LMGPU_FUSE_LEVELS") else "1")
        opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
        opt.linearize()
        out = []
        for lam in (1e-5, 1e-2, 1.0):
            _, d, e0, e1 = opt.solve(lam)
            out.append((d.copy(), e1))
        opt.close()
        results[mode] = out
    base = results["default"]
    for mode, out in results.items():
        tol = 0.0 if mode == "LMGPU_BSD_TICKET" else 1e-9  # tickets only change which workgroup takes which block
        for (d, e1), (d0, e10) in zip(out, base):
            assert np.linalg.norm(d - d0) <= tol * np.linalg.norm(d0), mode
            assert abs(e1 - e10) <= max(tol, 1e-15) * max(1.0, abs(e10)), mode


def test_block_backsolve_runs_over_levels_are_bitwise_the_per_level_launches(monkeypatch, dev_switches):
    """round 3: the block-solved fronts of consecutive levels go as ONE launch (separator values awaited by value); every block computes
    what it computes in a launch per level (LMGPU_NO_BSD_RUNS), so delta is the same to the bit.  sphere2500 / COLAMD: a run of sixteen
    levels; city10000 / COLAMD: runs that pass levels which also hold LDS fronts."""
    from gtsam_personal_amd import noiseModel
    from gtsam_personal_amd.datasets import chain_initial_pose3, load3D
    import bench
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "slam_orderings.npz"))
    for wl in ("sphere2500", "city10000"):
        graph, initial = bench.slam_workload(wl)
        keys = np.array(sorted(graph.keys()), dtype=np.uint64)
        ordering = [int(k) for k in keys[fx[f"{wl}_colamd"]]]
        out = {}
        for mode in ("default", "LMGPU_NO_BSD_RUNS"):
            monkeypatch.delenv("LMGPU_NO_BSD_RUNS", raising=False)
            if mode != "default":
                monkeypatch.setenv(mode, "1")
            opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
            opt.linearize()
            out[mode] = [opt.solve(lam)[1].copy() for lam in (1e-5, 1.0)]
            opt.close()
        for d, d0 in zip(out["LMGPU_NO_BSD_RUNS"], out["default"]):
            assert np.array_equal(d, d0), wl


def test_kernel_timers_modes():
    """lmgpu_set_kernel_timing: 1 = every category, 2 = only the roofline kernels (what bench.py keeps inside its timed region); the
    timers never change results"""
    graph, initial, _, ordering = make_bal(n_cam=64, n_pt=1600, obs_per_point=8, seed=17)
    outs = {}
    for mode in (0, 1, 2):
        opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
        opt.set_kernel_timing(mode)
        opt.iterate()
        kt = opt.kernel_times()
        outs[mode] = (opt.error(), kt)
        opt.close()
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert all(v["launches"] == 0 for v in outs[0][1].values())
    assert outs[1][1]["linearize"]["launches"] > 0 and outs[1][1]["lds_front"]["launches"] > 0 and outs[1][1]["backsub_lds"]["launches"] > 0
    assert outs[2][1]["linearize"]["launches"] > 0 and outs[2][1]["linearize"]["ms"] > 0
    assert all(v["launches"] == 0 for k, v in outs[2][1].items() if k not in ("linearize", "chain", "syrk"))
    assert outs[2][1]["chain"]["launches"] + outs[2][1]["syrk"]["launches"] > 0
