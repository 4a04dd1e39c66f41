#!/usr/bin/env python3
"""bench.py — LM iterations/s of the MI355X-native Levenberg-Marquardt inner loop on a synthetic BAL-shaped graph.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).  A "step" is ONE
LevenbergMarquardtOptimizer::iterate() (linearize all factors + damped multifrontal Cholesky solve + linear-error,
retract and error evaluation) on the SAME graph FROM THE SAME perturbed initial estimate and LM state (the values are
restored from a device-side snapshot before every step, a few-microsecond device-to-device copy inside the timed
region), so every step does identical work — the first LM iteration the reference's own measurements quote
(SURVEY section 6).  W untimed steps, then exactly K timed ones between barrier + device synchronisation; MAX over
ranks; rank 0 prints ONE JSON line.  Workload = BASELINE.json configs[3] (1 000 cameras / 100 000 points / 1 000 000 projection
factors, seed 42), which fits one GPU; N > 1 shards the point subtrees over the ranks and sums the camera-separator
contributions with RCCL, i.e. STRONG scaling of the same graph.  Inputs are resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix peak (not listed in MI355X_MICROARCH.md; measured alongside)
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


# the real reference's own numbers on this workload (libgtsam built from /root/reference in the survey container,
# Release, no TBB => 1 thread, 8-vCPU Xeon 2.1 GHz; std::chrono around the calls iterate() makes; first LM iteration at lambda = 1e-5)
REFERENCE_MEASURED = {
    "source": "BASELINE.md section 2 (survey container: reference GTSAM built from /root/reference, g++ 11.4 -O3, GTSAM_WITH_TBB=OFF => 1 thread, "
              "8-vCPU Intel Xeon 2.10 GHz; the reference tree does not travel to the GPU box, so these are not re-measured here)",
    (1000, 100000, 10, 42): [
        {"ordering": "metis", "ms_per_iteration": 44523.0, "value": 1e3 / 44523.0, "linearize_ms": 964.0, "eliminate_ms": 74233.0, "cores": 1},
        {"ordering": "schur", "ms_per_iteration": 51087.0, "value": 1e3 / 51087.0, "linearize_ms": 830.0, "eliminate_ms": 51891.0, "cores": 1}],
    (100, 10000, 10, 42): [
        {"ordering": "metis", "ms_per_iteration": 947.0, "value": 1e3 / 947.0, "linearize_ms": 72.0, "eliminate_ms": 887.0, "cores": 1},
        {"ordering": "schur", "ms_per_iteration": 708.0, "value": 1e3 / 708.0, "linearize_ms": 86.1, "eliminate_ms": 844.0, "cores": 1}],
}


def cpu_baseline(graph, initial, ordering, ordering_name, size_key, full_tag=None):
    """The CPU oracle (oracle/liblm_oracle.so: a port of the reference's algorithm; its dense partial Cholesky blocked like Eigen's LLT)
    timed on the GPU box's host cores on the SAME workload as the GPU line: `value` = one LevenbergMarquardtOptimizer::iterate() at full size
    with all the host cores this process may use -- subtree-parallel elimination + parallel linearize (what the reference does with
    TBB, gtsam/base/treeTraversal/parallelTraversalTasks.h:78-93, NonlinearFactorGraph.cpp:246-261) and, beyond the reference, the rank-128
    updates of the dense root shared out over the threads (Eigen's LLT is single-threaded whatever GTSAM's TBB setting).  Nested: the
    oracle's single-thread time on the same workload (build container, cached with the parity fixture), the real reference's own numbers
    from the survey, and the 1/10-per-dimension sample this line carried in rounds 1-2."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_harness as oh  # the oracle is only ever the baseline / the checker
    from gtsam_personal_amd import LevenbergMarquardtParams
    from gtsam_personal_amd.synthetic import make_bal
    params = LevenbergMarquardtParams()
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))

    def leg(g, x0, order, threads, dense_threads, budget_s, max_it):
        oh.set_threads(threads)
        oh.set_dense_threads(dense_threads)
        orc = oh.OracleProblem(g, x0, order)
        orc.lm_init(params)
        times = []
        t_all = time.perf_counter()
        while len(times) < max_it and (time.perf_counter() - t_all) < budget_s:
            t0 = time.perf_counter()
            orc.lm_iterate(params)
            times.append(time.perf_counter() - t0)
        tm = orc.timings()
        return len(times) / sum(times), times, tm

    # same workload, all cores (one iteration: ~10-25 s of CPU work)
    vf, tf, tmf = leg(graph, initial, ordering, cores, cores, 1.0, 1)
    out = dict(value=vf, unit="LM iterations/s", cores=cores, kind="port",
               sample=f"the SAME workload as the GPU line ({graph.size()} factors, {ordering_name}), ONE LM iteration (the first, like every GPU step), "
                      f"oracle/liblm_oracle.so with {cores} threads: subtree-parallel elimination and parallel linearize like the reference's TBB "
                      f"build, plus (beyond the reference, whose Eigen LLT is single-threaded) the dense root's rank-128 updates on all threads",
               ms_per_iteration=1e3 * sum(tf) / len(tf), linearize_ms=1e3 * tmf["linearize_s"], eliminate_ms=1e3 * tmf["eliminate_s"])
    if full_tag:
        try:
            with open(os.path.join(ROOT, "tests", "golden", f"{full_tag}_timing.json")) as f:
                tj = json.load(f)
            out["single_thread_same_workload"] = {"what": tj["what"], "workload": tj["workload"],
                                                  "runs": [{"ordering": r["ordering"], "value": 1.0 / r["iterate_s"], "unit": "LM iterations/s", "cores": 1,
                                                            "ms_per_iteration": 1e3 * r["iterate_s"], "linearize_ms": 1e3 * r["linearize_s"],
                                                            "eliminate_ms": 1e3 * r["eliminate_s"]} for r in tj["runs"]]}
        except OSError:
            pass
    if size_key in REFERENCE_MEASURED and "co-visibility window" not in (full_tag or "") and not (full_tag or "").startswith("c4band"):
        out["reference_measured"] = {"what": "the real reference (libgtsam), one LevenbergMarquardtOptimizer::iterate() on this workload",
                                     "source": REFERENCE_MEASURED["source"], "unit": "LM iterations/s", "runs": REFERENCE_MEASURED[size_key]}
    # the 1/10-per-dimension sample of rounds 1-2 (3 iterations each, single thread and all cores, subtree parallelism only)
    n_cam, n_pt, obs, seed = size_key
    g10, x10, _, o10 = make_bal(max(2, n_cam // 10), max(10, n_pt // 10), obs, seed=seed)
    v1, t1, tm1 = leg(g10, x10, o10, 1, 1, 8.0, 3)
    vn, tn, tmn = leg(g10, x10, o10, cores, 1, 8.0, 3) if cores > 1 else (v1, t1, tm1)
    oh.set_threads(1)
    oh.set_dense_threads(1)
    out["tenth_scale_sample"] = {"workload": f"synthetic BAL {max(2, n_cam // 10)} cameras / {max(10, n_pt // 10)} points / {g10.size()} factors, Schur ordering",
                                 "all_cores": {"value": vn, "cores": cores, "ms_per_iteration": 1e3 * sum(tn) / len(tn),
                                               "linearize_ms": 1e3 * tmn["linearize_s"], "eliminate_ms": 1e3 * tmn["eliminate_s"]},
                                 "single_thread": {"value": v1, "cores": 1, "ms_per_iteration": 1e3 * sum(t1) / len(t1),
                                                   "linearize_ms": 1e3 * tm1["linearize_s"], "eliminate_ms": 1e3 * tm1["eliminate_s"]}}
    return out


def metis_fixture_ordering(args, schur):
    """The reference computes its ordering once at optimizer construction (LevenbergMarquardtParams.h:112-117); it is a boundary
    INPUT of the hot path.  The METIS permutation of the bench workloads is carried by a committed fixture
    (tests/tools/make_c4_fixture.py: Ordering::Metis through the reference's own METIS sources, oracle/_ref)."""
    import numpy as np
    tags = {(1000, 100000, 10, 42, None): "c4_seed42", (100, 10000, 10, 42, None): "bal100_seed42", (1000, 100000, 10, 42, 40): "c4band_seed42"}
    tag = tags.get((args.cams, args.points, args.obs, args.seed, args.window))
    if tag is None:
        raise SystemExit("--ordering metis: no METIS fixture for this size (tests/golden/*_metis.npz); use --ordering schur")
    fx = np.load(os.path.join(ROOT, "tests", "golden", f"{tag}_metis.npz"))
    srt = np.sort(np.array(list(schur), dtype=np.uint64))
    return [int(k) for k in srt[fx["ordering_perm"]]]


def slam_workload(name):
    """BASELINE configs[2] / configs[0] at the reference's own size: examples/Pose3SLAMExample_g2o.cpp on sphere2500 (2 500 Pose3,
    4 949 BetweenFactor<Pose3>, odometry-chained initial estimate because the file carries no vertices, prior on pose 0 with
    Diagonal::Variances(1e-6 x3, 1e-4 x3), :42-48) and examples/Pose2SLAMExample_g2o.cpp on city10000 (10 000 Pose2, 20 687 factors,
    prior Variances(1e-6, 1e-6, 1e-8), :55-67)"""
    import numpy as np
    from gtsam_personal_amd import noiseModel
    from gtsam_personal_amd.datasets import chain_initial_pose3, load3D, readG2o
    gold = os.path.join(ROOT, "tests", "golden")
    if name == "sphere2500":
        graph, _ = load3D(os.path.join(gold, "sphere2500.txt"))
        initial = chain_initial_pose3(graph)
        graph.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
    elif name == "city10000":
        graph, initial = readG2o(os.path.join(gold, "city10000.g2o"))
        graph.add_PriorFactorPose2(0, initial.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
    elif name == "victoria_park":  # examples/Data/victoria_park.txt: 6 969 Pose2, 151 Point2 landmarks, 6 968 odometry + 3 640 bearing-range factors
        from gtsam_personal_amd.datasets import load2D
        graph, initial = load2D(os.path.join(gold, "victoria_park.txt"))
        graph.add_PriorFactorPose2(0, initial.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
    else:
        raise SystemExit(name)
    return graph, initial


def slam_bench(args):
    """side line for the general sparse configs (deep clique trees): LM iterations/s of one iterate() from the initial estimate, with
    the bound that applies to them stated: tree depth x per-level latency (a level is one dependent launch sequence)"""
    import numpy as np
    import torch
    from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
    graph, initial = slam_workload(args.workload)
    oname = "colamd" if args.ordering == "schur" else args.ordering  # COLAMD is the examples' default ordering type
    fx = np.load(os.path.join(ROOT, "tests", "golden", "slam_orderings.npz"))
    keys = np.array(sorted(graph.keys()), dtype=np.uint64)
    ordering = [int(k) for k in keys[fx[f"{args.workload}_{oname}"]]]
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    opt.save_values()
    st = opt.copy_state()
    for _ in range(args.warmup):
        opt.restore_values(st)
        opt.iterate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inner = 0
    for _ in range(args.steps):
        opt.restore_values(st)
        opt.iterate()
        inner += opt.timings()["inner_iterations"]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    opt.set_kernel_timing(True)
    opt.restore_values(st)
    opt.iterate()
    kt = opt.kernel_times()
    nf = opt.num_fronts()
    infos = [opt.front_info(i) for i in range(nf)]
    levels = max(fi["level"] for fi in infos) + 1
    # algorithmic bytes of one solve: every Jacobian read once, every [R S d] written once and read once by the back-substitution
    from gtsam_personal_amd.graph import FACTOR_ROWS, FACTOR_VARS, VAR_DIM
    jac = sum(len(gi) * FACTOR_ROWS[ft] * (sum(VAR_DIM[t] for t in FACTOR_VARS[ft]) + 1) * 8 for ft, _, gi, _, _, _, _ in graph.buckets())
    rsd = sum(fi["nf"] * fi["n"] * 8 for fi in infos)
    solve_ms = sum(v["ms"] for k, v in kt.items() if k not in ("linearize", "retract_error"))
    out = {"metric": "LM iterations/sec", "value": args.steps / elapsed, "unit": "LM iterations/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "reference dataset file (tests/golden)",
           "config": {"workload": f"{args.workload} ({graph.size()} factors, {len(ordering)} variables), {oname.upper()} ordering of the reference "
                                  f"(fixture-carried), one LM iterate from the initial estimate", "fronts": nf, "tree_levels": levels,
                      "inner_iterations_per_step": inner / args.steps},
           "kernel_ms_one_iterate": {k: v["ms"] for k, v in kt.items() if v["ms"] > 0},
           "kernel_launches_one_iterate": {k: v["launches"] for k, v in kt.items() if v["launches"] > 0},
           "roofline": {"kernel": "the whole damped solve (all fronts + back-substitution; no single dominant kernel)", "bound": "hbm",
                        "achieved": (jac + 2 * rsd) / (solve_ms * 1e-3) / 1e9 if solve_ms > 0 else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": (jac + 2 * rsd) / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if solve_ms > 0 else None, "traffic": None,
                        "note": f"latency-bound, not bandwidth-bound: {levels} tree levels, each a dependent launch sequence up and down the "
                                f"tree; the applicable bound is levels x per-level latency (~{1e3 * solve_ms / max(1, 2 * levels * max(1, inner // args.steps or 1)):.0f} us per level and direction measured)"}}
    print(json.dumps(out), flush=True)


def isam2_sequences(tmpdir, poses):
    """the two incremental workloads as input files of the C++ driver over the C ABI (tests/cpp/isam2_harness.cpp):
    visual      examples/VisualISAM2Example.cpp:88-131 (BASELINE configs[4]: 8 Pose3 + 8 Point3, 64 projection factors + 2 priors, ISAM2Params
                relinearizeThreshold 0.01, relinearizeSkip 1, a bare update() after every frame)
    city10000   timing/timeIncremental.cpp:84-170 on city10000.g2o: one pose per update, the new pose initialised from the DEVICE's own
                calculateEstimate(previous pose) composed with the odometry (W lines), default ISAM2Params"""
    import numpy as np
    from gtsam_personal_amd import ISAM2Params
    from gtsam_personal_amd.incremental_workloads import incremental_pose2_steps, visual_steps, write_isam2_sequence
    out = {}
    path = os.path.join(tmpdir, "visual.txt")
    write_isam2_sequence(path, ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1), visual_steps())
    out["visual"] = path
    est = {0: np.zeros(3)}

    def dead_reckoning(k):  # only fills the V lines the W lines replace; keeps the generator's interface
        return est[k]

    steps = []
    for g, v in incremental_pose2_steps(os.path.join(ROOT, "tests", "golden", "city10000.g2o"), poses, dead_reckoning):
        for k in v.keys():
            est[int(k)] = np.asarray(v.at(k), dtype=float)[:3]
        steps.append((g, v))
    path = os.path.join(tmpdir, "city10000.txt")
    write_isam2_sequence(path, ISAM2Params(), steps, relative_pose2=True)
    out["city10000"] = path
    return out


def fixed_lag_sequence(tmpdir, fixture=None):
    """a fixed-lag smoother's calls on city10000 (order the pose about to leave first, update, ISAM2::marginalizeLeaves: what
    gtsam_unstable/nonlinear/IncrementalFixedLagSmoother.cpp does around its ISAM2; one pose per update, findUnusedFactorSlots) as an input
    file of the C++ driver.  The extraReelimKeys a smoother reads off its copy of the Bayes tree come from a fixture
    (tests/tools/make_fixed_lag_fixture.py), like the orderings."""
    import numpy as np
    from gtsam_personal_amd import ISAM2Params
    from gtsam_personal_amd.incremental_workloads import fixed_lag_pose2_steps, write_isam2_sequence
    fx = json.load(open(fixture or os.path.join(ROOT, "tests", "golden", "isam2_fixed_lag_city10000.json")))
    poses, lag = fx["poses"], fx["lag"]
    est = {0: np.zeros(3)}
    steps = []
    for step, (g, v, leaving) in enumerate(fixed_lag_pose2_steps(os.path.join(ROOT, "tests", "golden", "city10000.g2o"), poses, lag, lambda k: est[k]), start=1):
        for k in v.keys():  # (only fills the V lines the W lines replace)
            est[int(k)] = np.asarray(v.at(k), dtype=float)[:3]
        constrained = None
        if leaving:
            constrained = {k: 1 for k in range(max(0, step - lag), step + 1)}
            for k in leaving:
                constrained[int(k)] = 0
        steps.append((g, v, None, dict(constrained=constrained, extra_reelim=fx["extra_reelim"][step - 1], marginalize=leaving)))
    p = ISAM2Params()
    p.findUnusedFactorSlots = True
    path = os.path.join(tmpdir, "fixed_lag.txt")
    write_isam2_sequence(path, p, steps, relative_pose2=True)
    return path


def isam2_bench(args):
    """BASELINE configs[4] and the reference's incremental benchmark loop through the C ABI from C++ (no Python between the updates): ms per
    ISAM2::update inside the library calls, with percentiles; the CPU oracle on the same sequences beside it.  The constrained COLAMD
    orderings the library asks its caller for are replayed from tests/golden/isam2_orderings_*.bin (recorded once with the reference's
    CCOLAMD by tests/tools/make_isam2_orderings.py): a boundary input, like the batch benchmark's METIS permutation."""
    import subprocess
    import tempfile
    harness = os.path.join(ROOT, "tests", "cpp", "isam2_harness")
    if not os.path.exists(harness):
        raise SystemExit("tests/cpp/isam2_harness not built (python -c 'import __graft_entry__ as g; g.build()')")
    res, seqs = {}, {}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        seqs = isam2_sequences(d, args.isam2_poses)
        if os.path.exists(os.path.join(ROOT, "tests", "golden", "isam2_fixed_lag_city10000.json")):
            seqs["fixed_lag"] = fixed_lag_sequence(d)
        t_gen = time.perf_counter() - t0
        for name, path in seqs.items():
            fx = os.path.join(ROOT, "tests", "golden", f"isam2_orderings_{name}.bin")
            best = None
            for _ in range(max(1, args.steps if name == "visual" else 1)):  # the small example is repeated; the long loop runs once
                # (repeat:3 = the sequence three times in one process, the last one reported: the first pass of a process pays ~8 ms twice
                #  for the runtime's first launches of the batch and incremental kernel configurations -- an incremental smoother's host
                #  is a long-running process)
                r = subprocess.run([harness, path, "0", "replay:" + fx] + (["repeat:3"] if name == "visual" else []), stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=900)
                out = json.loads(r.stdout)
                if "error" in out:
                    raise SystemExit(f"isam2 bench ({name}): {out['error']} -- the recorded orderings no longer fit this run "
                                     f"(regenerate with tests/tools/make_isam2_orderings.py)")
                out.pop("estimate", None)
                if best is None or out["ms_per_update_after_first"] < best["ms_per_update_after_first"]:
                    best = out
            res[name] = best
        cpu = None
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_harness as oh  # the checker / baseline only
            from gtsam_personal_amd import ISAM2Params
            from gtsam_personal_amd.incremental_workloads import incremental_pose2_steps, visual_steps
            cpu = {}
            p = ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1)
            best = 1e30
            for _ in range(5):
                orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
                steps = visual_steps()
                t0 = time.perf_counter()
                for g, v in steps:
                    orc.update(g, v)
                orc.calculateEstimate()
                best = min(best, (time.perf_counter() - t0) / len(steps))
            cpu["visual"] = {"ms_per_update": 1e3 * best, "updates": len(steps)}
            p = ISAM2Params()
            orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
            t_orc, n_up, budget = 0.0, 0, time.perf_counter()
            for g, v in incremental_pose2_steps(os.path.join(ROOT, "tests", "golden", "city10000.g2o"), args.isam2_poses,
                                                lambda k: orc.calculateEstimate().at(k)):
                t0 = time.perf_counter()
                orc.update(g, v)
                t_orc += time.perf_counter() - t0
                n_up += 1
                if time.perf_counter() - budget > 25.0:  # bounded sample: the first updates of the same loop
                    break
            cpu["city10000"] = {"ms_per_update": 1e3 * t_orc / max(1, n_up), "updates": n_up,
                                "note": "ISAM2::update only (the estimate of the previous pose is read outside the timed calls), through ctypes"}
            if "fixed_lag" in res:
                from gtsam_personal_amd.incremental_workloads import fixed_lag_pose2_steps
                fx = json.load(open(os.path.join(ROOT, "tests", "golden", "isam2_fixed_lag_city10000.json")))
                orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
                orc.set_find_unused_factor_slots(True)
                t_orc, n_up, budget = 0.0, 0, time.perf_counter()
                for step, (g, v, leaving) in enumerate(fixed_lag_pose2_steps(os.path.join(ROOT, "tests", "golden", "city10000.g2o"), fx["poses"], fx["lag"],
                                                                             lambda k: orc.calculateEstimate().at(k)), start=1):
                    constrained = None
                    if leaving:
                        constrained = {k: 1 for k in range(max(0, step - fx["lag"]), step + 1)}
                        constrained[leaving[0]] = 0
                    t0 = time.perf_counter()
                    orc.update(g, v, constrainedKeys=constrained, extraReelimKeys=fx["extra_reelim"][step - 1])
                    if leaving:
                        orc.marginalizeLeaves(leaving)
                    t_orc += time.perf_counter() - t0
                    n_up += 1
                    if time.perf_counter() - budget > 15.0:
                        break
                cpu["fixed_lag"] = {"ms_per_update": 1e3 * t_orc / max(1, n_up), "updates": n_up,
                                    "note": "ISAM2::update + marginalizeLeaves (the estimate of the previous pose is read outside the timed calls), through ctypes"}
    v = res["visual"]
    out = {"metric": "ISAM2 update latency", "value": v["ms_per_update_after_first"], "unit": "ms per update", "n_gpus": 1, "steps": v["updates"] - 1, "warmup": 1,
           "ms_per_step": v["ms_per_update_after_first"], "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
           "data": "VisualISAM2Example's scene (examples/SFMdata.h) / reference dataset file city10000.g2o (tests/golden)",
           "config": {"workload": "VisualISAM2Example (BASELINE configs[4]): 8 Pose3 + 8 Point3, 64 GenericProjectionFactor + 2 priors, 14 updates, "
                                  "relinearizeThreshold 0.01, relinearizeSkip 1; value = mean over the updates after the first of a fresh handle in a warm process "
                                  "(the sequence runs three times per process, the third is reported); driven from C++ through the C ABI, constrained COLAMD orderings replayed from "
                                  "a fixture recorded with the reference's CCOLAMD", "best_of": args.steps},
           "visual_isam2_example": v,
           "city10000_incremental": dict(res["city10000"], workload=f"timing/timeIncremental.cpp on city10000.g2o, {res['city10000']['updates']} updates, one pose per update, every new pose "
                                                                    "initialised from the device's calculateEstimate(previous pose); ms_per_update includes those single-variable estimates"),
           "fixed_lag_city10000": (dict(res["fixed_lag"], workload="a fixed-lag smoother's calls on city10000.g2o (lag 50 poses, one pose per update: update with the leaving pose ordered "
                                                                  "first, then ISAM2::marginalizeLeaves; loop closures inside the window; findUnusedFactorSlots); ms_per_update = update + "
                                                                  "marginalizeLeaves + the single-variable estimate that initialises the next pose") if "fixed_lag" in res else None),
           "roofline": {"kernel": "none dominant: an update is a dozen dependent launches on a few workgroups", "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": None, "traffic": None, "note": "latency-bound at this size (16 variables / a few cliques per update); no roofline is claimed"},
           "sequence_generation_s": t_gen}
    if cpu is not None:
        out["cpu_baseline"] = {"value": cpu["visual"]["ms_per_update"], "unit": "ms per update", "cores": 1, "kind": "port",
                               "sample": "oracle/isam2_oracle.hpp (CPU restatement of ISAM2::update), the same VisualISAM2Example sequence, best of 5, through ctypes",
                               "city10000_incremental": cpu["city10000"], "fixed_lag_city10000": cpu.get("fixed_lag")}
    print(json.dumps(out), flush=True)


def scaling_model(graph, initial, ordering, worlds=(2, 4, 8)):
    """What the ownership rule (csrc/lmgpu.hip: assign_owners) gives this graph at 2 / 4 / 8 ranks, from structure-only handles (no device):
    replicated fronts, per-rank owned work, bytes all-reduced, and a time model -- a dense front = 150 us of dependent launches (measured:
    the 122 dense fronts of the --window 40 workload take 20 ms on one GPU, each its own gather / panel / update launch sequence) + its
    flop at the single-GPU rate of the chained factorisation (2 nf n^2 / 3 at 44 TFLOP/s); point leaves at the measured 16 ns each (leaf
    fronts + their share of the back-substitution); the all-reduce of a replicated front as a ring over xGMI at 100 GB/s algorithm
    bandwidth, of which the first 512 rows are exposed (the rest travels beside the factorisation).  A model, not a measurement: no
    multi-GPU node is available to the build."""
    from gtsam_personal_amd import LevenbergMarquardtOptimizer

    def t_front(f):
        return (f["nf"] * f["n"] * f["n"] / 3.0) * 2.0 / 44e12 + 150e-6 if f["cls"] == 1 else 16e-9

    out = {}
    for W in worlds:
        probe = LevenbergMarquardtOptimizer(graph, initial, ordering, device=-1, rank=0, world_size=W)
        info = [probe.front_info(i) for i in range(probe.num_fronts())]
        probe.close()
        rep = [f for f in info if f["owner"] < 0]
        t_rep = sum(t_front(f) for f in rep)
        t_own = [sum(t_front(f) for f in info if f["owner"] == r) for r in range(W)]
        t_one = sum(t_front(f) for f in info)
        ar_bytes = sum(f["n"] * f["n"] * 4 for f in rep)  # upper triangle, FP64
        t_ar_exposed = sum(min(f["n"], 512) * f["n"] * 8 for f in rep) * 2.0 * (W - 1) / W / 100e9  # the first chunks; the rest hides behind the factorisation
        out[str(W)] = {"replicated_fronts": len(rep), "replicated_dense_columns": sum(f["nf"] for f in rep),
                       "owned_dense_fronts_per_rank": [sum(1 for f in info if f["owner"] == r and f["cls"] == 1) for r in range(W)],
                       "model_ms_single": 1e3 * t_one, "model_ms_replicated": 1e3 * t_rep, "model_ms_owned_max": 1e3 * max(t_own),
                       "allreduce_mbytes": ar_bytes / 1e6, "model_ms_allreduce_exposed": 1e3 * t_ar_exposed,
                       "model_speedup": t_one / (t_rep + max(t_own) + t_ar_exposed)}
    return out


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (counters cannot be collected
    from inside the process): {kernel: bytes}.  tools/pmc_round.sh + tools/pmc_summary.py write the file, together with a sha256 of the
    kernel sources it was collected from: a summary of OTHER kernels than the ones that run now is not reported (traffic null, with the
    reason beside it)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    for rnd in ("r03", "r02"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")
        try:
            with open(path) as f:
                d = json.load(f)
        except OSError:
            continue
        from pmc_summary import kernel_source_hash
        sha = d.pop("_kernel_source_sha256", None)
        if sha != kernel_source_hash():
            return {"_stale": f"profiles/{rnd}/pmc_summary.json was collected from other kernel sources than the ones in the tree (re-run tools/pmc_round.sh)"}
        out = {k: v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"] for k, v in d.items()
               if isinstance(v, dict) and "hbm_read_bytes_corrected" in v and "hbm_write_bytes" in v}
        out["_source"] = f"profiles/{rnd}/pmc_summary.json"
        return out
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cams", type=int, default=1000)
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--obs", type=int, default=10)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--ordering", choices=["metis", "schur", "colamd"], default="metis",
                    help="elimination ordering: the reference's METIS ordering (BASELINE.json configs[3]; the permutation is a boundary input "
                         "carried by tests/golden/<tag>_metis.npz, produced once by Ordering::Metis through oracle/_ref) or Schur "
                         "(points then cameras, timing/timeSFMBAL.h:64-96)")
    ap.add_argument("--workload", choices=["bal", "sphere2500", "city10000", "victoria_park", "isam2"], default="bal",
                    help="bal = the headline synthetic BAL graph (BASELINE configs[3]); sphere2500 / city10000 = the general sparse configs at the "
                         "reference's size (single GPU side line; --ordering colamd|metis); isam2 = BASELINE configs[4] (VisualISAM2Example) and the "
                         "reference's incremental loop on city10000, ms per ISAM2::update through the C ABI")
    ap.add_argument("--isam2-poses", type=int, default=10000, help="--workload isam2: poses of the city10000 incremental loop")
    ap.add_argument("--window", type=int, default=None,
                    help="banded co-visibility: every point is seen from cameras within this window of the ring (synthetic.make_bal window=); "
                         "--window 40 at the default size has a METIS fixture.  The camera block of the Hessian is then sparse and the camera "
                         "subtrees below the top separators shard over the ranks, dense fronts included")
    ap.add_argument("--dev-library", action="store_true",
                    help="development: run on liblmgpu_test.so, the build that reads the LMGPU_* A/B switches from the environment (tools/variants/ab.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-peaks", action="store_true", help="skip the device micro-benchmarks (profiling runs: hundreds of extra launches under PMC)")
    ap.add_argument("--split-root", action="store_true",
                    help="1-GPU rehearsal of the multi-rank data path: one-rank RCCL communicator, chunked all-reduce of the root")
    args = ap.parse_args()

    if args.dev_library:
        from gtsam_personal_amd import _lib as _l
        _l.use_test_library(True)
    if args.workload != "bal":
        if args.gpus != 1:
            raise SystemExit("--workload sphere2500 / city10000 / victoria_park / isam2 is a single-GPU side line")
        return isam2_bench(args) if args.workload == "isam2" else slam_bench(args)
    if args.ordering == "colamd":
        raise SystemExit("--ordering colamd is offered for --workload sphere2500 / city10000 (the BAL fixtures carry METIS and Schur)")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # one process per GPU: N > 1 runs under torch.distributed.run (the driver's launch line); a bare `--gpus N` would silently
        # measure one GPU
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port <P> bench.py --gpus {args.gpus} ...`")
    # RCCL prints a version banner on stdout when the first communicator is created; the contract is ONE JSON line on
    # stdout, so everything before the final print goes to stderr at file-descriptor level
    saved_stdout = None
    if world > 1 or args.split_root:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
    from gtsam_personal_amd.synthetic import make_bal

    # The first process on a freshly started GPU box ran 3 % slower than any later one (112.7 vs 116.2 LM iterations/s; every kernel
    # slower, the bandwidth-bound linearize by 11 %), whatever the number of warm-up steps: the device memory the solver's pool lands in
    # is handed out for the first time.  Allocating, clearing and freeing a few GiB once before the solver allocates removes that
    # (measured: profiles/r02/first_process.txt).  Untimed set-up, reported in the JSON line; BENCH_PRETOUCH_GB=0 switches it off.
    pre_gb = float(os.environ.get("BENCH_PRETOUCH_GB", "8"))
    if pre_gb > 0:
        x = torch.empty(int(pre_gb * (1 << 30)), dtype=torch.uint8, device=f"cuda:{local_rank}")
        x.zero_()
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()
    t_setup = time.perf_counter()
    graph, initial, _, ordering = make_bal(args.cams, args.points, args.obs, seed=args.seed, window=args.window)
    ordering_name = "Schur ordering (points then cameras)"
    if args.ordering == "metis":
        ordering = metis_fixture_ordering(args, ordering)
        ordering_name = "METIS ordering (Ordering::Metis of the reference, fixture-carried permutation)"
    params = LevenbergMarquardtParams()
    comm_id = None
    if world > 1:
        obj = [LevenbergMarquardtOptimizer.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0)
        comm_id = obj[0]
    if args.split_root and world == 1:
        comm_id = LevenbergMarquardtOptimizer.comm_unique_id()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=local_rank, rank=rank, world_size=world, comm_id=comm_id,
                                      split_root=args.split_root and world == 1)
    t_setup = time.perf_counter() - t_setup
    n_factors = graph.size()
    e_initial = opt.error()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    opt.save_values()
    state0 = opt.copy_state()
    for _ in range(args.warmup):
        opt.restore_values(state0)
        opt.iterate()
    if not args.no_kernel_timing:
        # HIP events around the two roofline kernels only, live in the timed region; the other categories (whose events would cost the
        # timed region ~0.08 ms of idle device time per step) are measured on one more iteration behind it
        opt.set_kernel_timing(2)
    phases = dict(linearize_ms=0.0, eliminate_ms=0.0, backsub_ms=0.0, linear_error_ms=0.0, retract_error_ms=0.0)
    inner = 0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.restore_values(state0)
        opt.iterate()
        tm = opt.timings()
        for k in phases:
            phases[k] += tm[k]
        inner += tm["inner_iterations"]
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kt = opt.kernel_times() if not args.no_kernel_timing else None
    e_final = opt.error()  # error after ONE LM iteration from the initial estimate
    kt_all, kt_all_inner = None, 1
    if kt is not None:  # every category, outside the timed region
        opt.set_kernel_timing(True)
        opt.restore_values(state0)
        opt.iterate()
        kt_all_inner = max(1, opt.timings()["inner_iterations"])
        kt_all = opt.kernel_times()
        opt.restore_values(state0)
        opt.iterate()  # leave the values where e_final was taken

    if rank == 0:
        steps = args.steps
        out = {
            "metric": "LM iterations/sec",
            "value": steps / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"synthetic BAL {args.cams} cameras / {args.points} points / {n_factors} factors (GeneralSFMFactor<Cal3Bundler> + 2 priors), "
                                   f"seed {args.seed}, {ordering_name}, LM legacy defaults" + (f", co-visibility window {args.window}" if args.window else ""),
                       "ordering": args.ordering,
                       "cameras": args.cams, "points": args.points, "factors": n_factors, "fronts": opt.num_fronts(),
                       "parallelism": "single GPU" if world == 1 else f"elimination subtrees (dense fronts included) owned by one of {world} ranks each; fronts too heavy "
                                                                      "for one rank replicated after ncclAllReduce of their partial assemblies",
                       # what the ownership rule gives this graph with more ranks (structure-only model; the dense C4 root is all of the work above the
                       # leaves and stays replicated: <= ~1.2x; with --window the camera subtrees shard)
                       **({} if (world == 1 and args.window is None) else {"scaling_model": scaling_model(graph, initial, ordering)})},
            "ms_per_linearize": phases["linearize_ms"] / steps,
            "ms_per_eliminate": phases["eliminate_ms"] / max(1, inner),
            "ms_per_backsub": phases["backsub_ms"] / max(1, inner),
            "ms_per_linear_error": phases["linear_error_ms"] / max(1, inner),
            "ms_per_retract_error": phases["retract_error_ms"] / max(1, inner),
            "inner_iterations": inner,
            "error_initial": e_initial,
            "error_after_one_iteration": e_final,
            "setup_s": t_setup,
            "device_memory_pretouch_gib": pre_gb,
        }
        if kt is not None:
            lin = kt["linearize"]
            # the dense camera front: all its fused steps are ONE chain_kernel launch; without chaining (LMGPU_NO_CHAIN, split root)
            # the same work is one step_kernel launch per 256 columns
            chained = kt["chain"]["launches"] > 0
            syrk = kt["chain"] if chained else kt["syrk"]
            pmc = pmc_traffic()
            if syrk["launches"] > 0 and syrk["ms"] > 0:
                tf = syrk["work"] / (syrk["ms"] * 1e-3) / 1e12
                kname = "chain_kernel" if chained else "step_kernel"
                out["roofline"] = {"kernel": kname + (" (v_mfma_f64_16x16x4_f64: every trailing update of the dense camera front and the factorisation of "
                                                      "its 256-column panels, tile-level dataflow inside one launch)" if chained else
                                                      " (v_mfma_f64_16x16x4_f64: trailing update of the dense camera front with outer panel i "
                                                      "+ factorisation of panel i+1 in the same launch)"),
                                   "bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc.get(kname),
                                   "traffic_unit": "HBM bytes per launch (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE; " + pmc.get("_source", pmc.get("_stale", "no PMC summary committed")) + ")",
                                   "launches": syrk["launches"], "avg_launch_us": 1e3 * syrk["ms"] / syrk["launches"],
                                   "flop_per_launch": syrk["work"] / syrk["launches"]}
            if lin["launches"] > 0 and lin["ms"] > 0:
                gbs = lin["work"] / (lin["ms"] * 1e-3) / 1e9
                out["roofline_linearize"] = {"kernel": "sfm_linearize_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                             "frac": gbs / HBM_PEAK_GBPS, "traffic": pmc.get("sfm_linearize_kernel"), "launches": lin["launches"],
                                             "avg_launch_us": 1e3 * lin["ms"] / lin["launches"], "bytes_per_launch": lin["work"] / lin["launches"]}
            # linearize / chain / syrk: the timed region; the other categories: the one fully instrumented iteration behind it
            live = ("linearize", "chain", "syrk")
            out["kernel_ms_per_step"] = {k: (v["ms"] / steps if k in live else kt_all[k]["ms"]) for k, v in kt.items()}
            out["kernel_ms_per_step_note"] = "linearize / chain / syrk: HIP events inside the timed region; others: one more iteration with every category instrumented"
        # measured device peaks for context (not the roofline denominators)
        import ctypes as ct
        from gtsam_personal_amd import _lib
        lib = _lib.load()
        v = ct.c_double()
        if args.no_peaks:
            lib = None
        if lib is not None and lib.lmgpu_peak_mfma_f64(local_rank, 4000, ct.byref(v)) == 0:
            out["measured_peak_mfma_f64_tflops"] = v.value
            if "roofline" in out and v.value > 0:  # beside the datasheet-based frac: against what this device sustains
                out["roofline"]["frac_of_measured_peak"] = out["roofline"]["achieved"] / v.value
        # why the register-resident loop stops at ~60 % of the datasheet figure: the clock the chip holds under FP64 MFMA load
        clk, fpc = ct.c_double(), ct.c_double()
        peaks = {}
        for nacc in (4, 8, 16):
            if lib is not None and lib.lmgpu_peak_mfma_f64_clock(local_rank, 4000, nacc, ct.byref(v), ct.byref(clk), ct.byref(fpc)) == 0:
                peaks[f"{nacc}_accumulators"] = {"tflops": v.value, "sustained_sclk_mhz": clk.value, "flop_per_clk_per_simd": fpc.value}
        if peaks:
            out["fp64_mfma_microbenchmark"] = dict(peaks, note="register-resident v_mfma_f64_16x16x4_f64 loops on every CU with the in-kernel shader clock "
                                                   "(s_memtime / s_memrealtime).  4 / 8 accumulators with one loop-invariant operand pair stop near 49 TFLOP/s "
                                                   "(what round 2 took for the instruction's ceiling); the 16-accumulator loop has the register shape of the "
                                                   "update tile (4 x 4 accumulators, 4 + 4 changing operands, two workgroups per CU) and shows what the "
                                                   "instruction sustains in that shape (tools/syrk4_bench.hip: the tile itself reaches 61-69 TFLOP/s without "
                                                   "its memory traffic)")
            best = max(peaks.values(), key=lambda d: d["tflops"])
            if "roofline" in out:
                out["roofline"]["instruction_peak_tflops"] = best["tflops"]
                out["roofline"]["frac_of_instruction_peak"] = out["roofline"]["achieved"] / best["tflops"]
        if lib is not None and lib.lmgpu_peak_hbm_copy(local_rank, 1 << 30, 5, ct.byref(v)) == 0:
            out["measured_hbm_copy_gbps"] = v.value
            if "roofline_linearize" in out and v.value > 0:
                out["roofline_linearize"]["frac_of_measured_copy"] = out["roofline_linearize"]["achieved"] / v.value
        if world == 1 and not args.no_cpu_baseline:
            size_key = (args.cams, args.points, args.obs, args.seed)
            full_tag = {(1000, 100000, 10, 42): "c4_seed42", (100, 10000, 10, 42): "bal100_seed42"}.get(size_key) if args.window is None else \
                {(1000, 100000, 10, 42, 40): "c4band_seed42"}.get(size_key + (args.window,))
            out["cpu_baseline"] = cpu_baseline(graph, initial, ordering, ordering_name, size_key, full_tag)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
