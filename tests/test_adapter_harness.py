"""The reference-side adapter's C++ half (include/lmgpu_adapter_core.hpp) driven from C++ (tests/cpp/adapter_harness.cpp, built by
`make -C tests/cpp`) in exactly the call order of include/lmgpu_gtsam_adapter.h: constructor (variables in elimination order,
factors in graph order incl. the PRIORS every BASELINE config has, finalize, values, LM init), iterate() = lmgpu_iterate under
NonlinearOptimizer::defaultOptimize, and the Piecewise mode (the reference's own tryLambda around linearize() / solve()).

CPU: structure through the harness == the Python mirror == the oracle's cliques.  GPU: examples/SFMExample_bal.cpp's graph
(dubrovnik-3-7-pre, Isotropic(2, 1) projection noise, priors on C(0) and P(0), default LM, COLAMD) converges to the final error
the reference itself printed when it was run in the survey container: 0.0461375737045."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams, NonlinearFactorGraph, Ordering, Values, noiseModel
from gtsam_personal_amd.datasets import SfmData, bal_graph, load2D
from gtsam_personal_amd.graph import C, F_PRIOR_CAM, F_SFM, FACTOR_ARITY, N_UNIT, P, VAR_STORE_DEV

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")
HARNESS = os.path.join(ROOT, "tests", "cpp", "adapter_harness")

REFERENCE_SFMEXAMPLE_BAL_FINAL_ERROR = 0.0461375737045  # printed by the reference's own SFMExample_bal


def build_harness():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert os.path.exists(HARNESS)


def write_problem(path, graph, initial, ordering, params):
    """what the adapter extracts from the GTSAM objects, as text: the packings of include/lmgpu.h"""
    with open(path, "w") as f:
        f.write(f"VARIABLES {len(ordering)}\n")
        for k in ordering:
            t = initial.type(k)
            n = VAR_STORE_DEV[t]
            f.write(f"{int(k)} {t} {n} " + " ".join(repr(float(x)) for x in initial.at(k)[:n]) + "\n")
        rows = [None] * graph.size()
        for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
            for i, g in enumerate(gi.tolist()):
                m = np.array(meas[i], dtype=np.float64)
                if ftype == F_SFM:  # the constant principal point of Cal3Bundler is folded into z (lmgpu.h, CAM_BUNDLER)
                    m = m - initial.at(int(keys[i][0]))[15:17]
                if ftype == F_PRIOR_CAM:
                    m = m[:15]
                nz = [] if kind == N_UNIT else np.asarray(noise[i], dtype=np.float64).reshape(-1).tolist()
                rows[g] = (f"{ftype} {g} " + " ".join(str(int(x)) for x in keys[i][:FACTOR_ARITY[ftype]]) + f" {len(m)} " +
                           " ".join(repr(float(x)) for x in m) + f" {kind} {len(nz)} " + " ".join(repr(float(x)) for x in nz) +
                           f" {models[i].robust_kind} {float(models[i].robust_k)!r}\n")
        f.write(f"FACTORS {len(rows)}\n")
        f.writelines(rows)
        p = params
        f.write("PARAMS " + " ".join(repr(x) for x in (int(p.maxIterations), p.relativeErrorTol, p.absoluteErrorTol, p.errorTol, p.lambdaInitial,
                                                       p.lambdaFactor, p.lambdaUpperBound, p.lambdaLowerBound, p.minModelFidelity,
                                                       int(bool(p.diagonalDamping)), int(bool(p.useFixedLambdaFactor)), p.minDiagonal,
                                                       p.maxDiagonal)) + "\n")


def run_harness(problem, device, mode):
    r = subprocess.run([HARNESS, problem, str(device), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode in (0, 1), r.stderr
    return r.returncode, json.loads(r.stdout)


def sfm_example_bal():
    """examples/SFMExample_bal.cpp:45-78"""
    db = SfmData.FromBalFile(os.path.join(GOLD, "dubrovnik-3-7-pre.txt"))
    graph, initial = bal_graph(db, noise=noiseModel.Isotropic.Sigma(2, 1.0), camera_key=C, point_key=P)
    graph.add_PriorFactorCamera(C(0), initial.at(C(0)), noiseModel.Isotropic.Sigma(9, 0.1))
    graph.add_PriorFactorPoint3(P(0), initial.at(P(0)), noiseModel.Isotropic.Sigma(3, 0.1))
    ordering = oh.colamd(graph) if oh.have_ref() else Ordering.Schur(graph, initial)
    return graph, initial, ordering


def pose2_example():
    """examples/Pose2SLAMExample_g2o.cpp:55-67 on the reference's noisyToyGraph.txt: between factors with full information
    matrices + the prior on the first pose"""
    graph, initial = load2D(os.path.join(GOLD, "noisyToyGraph.txt"))
    k0 = initial.keys()[0]
    graph.add_PriorFactorPose2(k0, initial.at(k0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
    return graph, initial, Ordering.Natural(graph)


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("make", [sfm_example_bal, pose2_example])
def test_harness_structure_matches_mirror_and_oracle(tmp_path, make):
    build_harness()
    graph, initial, ordering = make()
    prob = str(tmp_path / "problem.txt")
    write_problem(prob, graph, initial, ordering, LevenbergMarquardtParams())
    rc, out = run_harness(prob, -1, "structure")
    assert rc == 0, out
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, device=-1)
    assert out["num_fronts"] == opt.num_fronts()
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.linearize()
    assert orc.solve(1.0)[0] == 0
    cl = orc.cliques()
    assert len(cl) == out["num_fronts"]
    for i, (keys, nfk, rsd, parent) in enumerate(cl):
        fr = out["fronts"][i]
        assert fr["keys"] == keys and fr["nfk"] == nfk and (fr["nf"], fr["n"]) == rsd.shape and fr["parent"] == parent


def test_harness_without_device_fails_loudly(tmp_path):
    build_harness()
    graph, initial, ordering = sfm_example_bal()
    prob = str(tmp_path / "problem.txt")
    write_problem(prob, graph, initial, ordering, LevenbergMarquardtParams())
    rc, out = run_harness(prob, -1, "optimize")
    assert rc == 1 and "no HIP device" in out["exception"]


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_sfm_example_bal_through_the_adapter_reaches_the_reference_error(tmp_path):
    build_harness()
    graph, initial, ordering = sfm_example_bal()
    params = LevenbergMarquardtParams()
    prob = str(tmp_path / "problem.txt")
    write_problem(prob, graph, initial, ordering, params)
    rc, whole = run_harness(prob, 0, "optimize")
    assert rc == 0, whole
    assert abs(whole["error"] - REFERENCE_SFMEXAMPLE_BAL_FINAL_ERROR) <= 1e-6 * REFERENCE_SFMEXAMPLE_BAL_FINAL_ERROR, whole["error"]
    # the oracle on the same graph: same trajectory
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(params)
    orc.lm_optimize(params)
    so = orc.lm_state()
    assert abs(so["error"] - REFERENCE_SFMEXAMPLE_BAL_FINAL_ERROR) <= 1e-6 * REFERENCE_SFMEXAMPLE_BAL_FINAL_ERROR
    assert whole["iterations"] == so["iterations"] and whole["inner"] == so["inner"]
    # the linear graph iterate() hands back through the adapter (one copy per bucket) is the per-factor taps, for every factor
    assert whole["linear_graph_matches_taps"] and whole["linear_graph_factors"] == graph.size()
    assert abs(whole["error"] - so["error"]) <= 1e-6 * so["error"]
    # Piecewise mode (the reference's own tryLambda around linearize() / solve()): same decisions, same result
    rc, piece = run_harness(prob, 0, "piecewise")
    assert rc == 0, piece
    assert piece["iterations"] == whole["iterations"] and piece["inner"] == whole["inner"]
    assert abs(piece["error"] - whole["error"]) <= 1e-9 * whole["error"]
    assert abs(piece["lambda"] - whole["lambda"]) <= 1e-12 * whole["lambda"]
    assert np.allclose(piece["values"], whole["values"], rtol=1e-9, atol=1e-10)
    # and the Python mirror over the same C ABI
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    opt.optimize()
    assert abs(opt.error() - whole["error"]) <= 1e-12 * whole["error"]


@pytest.mark.gpu
def test_pose2_graph_with_prior_through_the_adapter(tmp_path):
    build_harness()
    graph, initial, ordering = pose2_example()
    params = LevenbergMarquardtParams()
    prob = str(tmp_path / "problem.txt")
    write_problem(prob, graph, initial, ordering, params)
    rc, whole = run_harness(prob, 0, "optimize")
    assert rc == 0, whole
    rc, piece = run_harness(prob, 0, "piecewise")
    assert rc == 0, piece
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(params)
    orc.lm_optimize(params)
    so = orc.lm_state()
    for out in (whole, piece):
        assert out["iterations"] == so["iterations"] and out["inner"] == so["inner"]
        assert abs(out["error"] - so["error"]) <= 1e-6 * max(so["error"], 1e-12) + 1e-12
