#!/usr/bin/env python3
"""Side measurement (not the bench.py metric): wall time per ISAM2::update on the device vs. the CPU oracle, on the incremental
workloads of tests/test_gpu_isam2.py (VisualISAM2Example; the first 400 poses of city10000 played timeIncremental-style).
The second half drives the same C ABI from C++ (tests/cpp/isam2_harness: the reference-side wrapper's call order, no Python between
the updates; the sequence is recorded from the oracle's run first) -- what an update costs a C++ caller -- on the 400-pose sequence and
on ALL poses of city10000 (tests/golden/city10000.g2o, one pose per update = timing/timeIncremental.cpp).
    python tests/tools/bench_isam2.py [--full]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_harness as oh  # noqa: E402
from gtsam_personal_amd import ISAM2, ISAM2Params, NonlinearFactorGraph, Values, noiseModel  # noqa: E402
from gtsam_personal_amd.datasets import readG2o  # noqa: E402
from isam2_examples import visual_steps  # noqa: E402


def ccolamd(n_rows, n_cols, col_ptr, row_idx, cmember):
    return oh.ccolamd_csc(n_rows, n_cols, col_ptr, row_idx, cmember)


def city_steps(n_updates=399):
    graph, _ = readG2o(os.path.join(ROOT, "tests", "golden", "city10000_head.g2o"))
    edges = []
    for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
        for i, g in enumerate(gi.tolist()):
            edges.append((g, int(keys[i][0]), int(keys[i][1]), meas[i], models[i]))
    edges.sort()
    orc = oh.OracleISAM2()
    steps, nxt, step = [], 0, 1
    while nxt < len(edges) and len(steps) < n_updates:
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
                a = np.zeros(3) if step == 1 else orc.calculateEstimate().at(step - 1)
                c, s = np.cos(a[2]), np.sin(a[2])
                v.insert(step, 0, [a[0] + c * m[0] - s * m[1], a[1] + s * m[0] + c * m[1], a[2] + m[2]])
            nxt += 1
        orc.update(g, v)
        steps.append((g, v))
        step += 1
    return steps


for name, steps, params in (("VisualISAM2Example (8 poses, 8 points)", visual_steps(), ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1)),
                            ("city10000 head, 399 incremental updates", city_steps(), ISAM2Params())):
    p = params
    isam = ISAM2(p, ccolamd=ccolamd, device=0)
    isam.update(*steps[0])
    t0 = time.perf_counter()
    for g, v in steps[1:]:
        isam.update(g, v)
    isam.calculateEstimate()
    tg = (time.perf_counter() - t0) / (len(steps) - 1)
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
    orc.update(*steps[0])
    t0 = time.perf_counter()
    for g, v in steps[1:]:
        orc.update(g, v)
    orc.calculateEstimate()
    to = (time.perf_counter() - t0) / (len(steps) - 1)
    print(f"{name}: device {1e3 * tg:.3f} ms per update (incl. the Python marshalling and the ccolamd callback), CPU oracle {1e3 * to:.3f} ms per update")
    isam.close()


# ---- the C++ driver
import json  # noqa: E402
import subprocess  # noqa: E402
import tempfile  # noqa: E402
from isam2_examples import incremental_pose2_steps, write_isam2_sequence  # noqa: E402

subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
cases = [("city10000 head, 400 poses", os.path.join(ROOT, "tests", "golden", "city10000_head.g2o"), 400)]
if "--full" in sys.argv:
    cases.append(("city10000, all 10000 poses", os.path.join(ROOT, "tests", "golden", "city10000.g2o"), 10000))
for name, g2o, n in cases:
    p = ISAM2Params()
    orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
    steps, t_orc = [], 0.0
    for g, v in incremental_pose2_steps(g2o, n, lambda k: orc.calculateEstimate().at(k)):
        t0 = time.perf_counter()
        orc.update(g, v)
        t_orc += time.perf_counter() - t0
        steps.append((g, v))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "seq.txt")
        write_isam2_sequence(path, p, steps)
        out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "isam2_harness"), path, "0", os.path.join(ROOT, "oracle", "_ref", "libccolamd_ref.so")],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    res = json.loads(out.stdout)
    # the same loop with the DEVICE's estimate of the previous pose initialising the next one (one calculateEstimate(key) per step):
    # timing/timeIncremental.cpp as it is written; the trajectory then differs from the recorded one in the last digits only
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "seq_rel.txt")
        write_isam2_sequence(path, p, steps, relative_pose2=True)
        out2 = subprocess.run([os.path.join(ROOT, "tests", "cpp", "isam2_harness"), path, "0", os.path.join(ROOT, "oracle", "_ref", "libccolamd_ref.so")],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    res2 = json.loads(out2.stdout)
    print(f"{name}, C++ driver, every new pose from calculateEstimate(previous pose) on the device: {res2['ms_per_update']:.4f} ms per update "
          f"including {res2['single_estimates']} single-variable estimates at {res2['single_estimate_ms']:.4f} ms each", flush=True)
    est = orc.calculateEstimate()
    worst = 0.0
    for r in res["estimate"]:
        e = est.at(int(r[0]))
        worst = max(worst, float(np.max(np.abs(np.array(r[1:]) - e) / np.maximum(1.0, np.abs(e)))))
    print(f"{name}, C++ driver: {res['updates']} updates, {res['ms_per_update']:.4f} ms per update inside the library calls "
          f"(of which the caller's ccolamd {1e3 * res['ccolamd_callback_seconds'] / res['updates']:.4f} ms; median {res['p50_ms']:.3f}, p95 {res['p95_ms']:.3f}, "
          f"p99 {res['p99_ms']:.3f}), worst update {res['worst_update_ms']:.2f} ms, "
          f"calculateEstimate {res['calculate_estimate_ms']:.2f} ms; CPU oracle {1e3 * t_orc / len(steps):.4f} ms per update (through ctypes); "
          f"estimate vs oracle: max rel diff {worst:.2e}, cliques {res['cliques']}", flush=True)
