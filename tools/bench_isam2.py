#!/usr/bin/env python3
"""Side measurement (not the bench.py metric): wall time per ISAM2::update on the device vs. the CPU oracle, on the incremental
workloads of tests/test_gpu_isam2.py (VisualISAM2Example; the first 400 poses of city10000 played timeIncremental-style).
    python tools/bench_isam2.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
