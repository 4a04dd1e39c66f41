#!/usr/bin/env python3
"""Side experiment: timeIncremental-style ISAM2 on the first N poses of the FULL city10000 graph (tests/golden/city10000.g2o), device only,
the new pose initialised from the device's own estimate.  Reports how far the incremental path gets before a clique outgrows the LDS-front
limit (139 scalar columns), the largest clique seen and the time per update.     python tools/isam2_long_run.py [N=3000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_harness as oh  # noqa: E402  (only for the ccolamd callback: the reference's vendored CCOLAMD, compiled by oracle/Makefile)
from gtsam_personal_amd import ISAM2, ISAM2Params, NonlinearFactorGraph, Values, noiseModel  # noqa: E402
from gtsam_personal_amd.datasets import readG2o  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
graph, _ = readG2o(os.path.join(ROOT, "tests", "golden", "city10000.g2o"))
edges = []
for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
    for i, g in enumerate(gi.tolist()):
        edges.append((g, int(keys[i][0]), int(keys[i][1]), meas[i], models[i]))
edges.sort()
isam = ISAM2(ISAM2Params(), ccolamd=lambda *a: oh.ccolamd_csc(*a), device=0)


def compose(a, d):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([a[0] + c * d[0] - s * d[1], a[1] + s * d[0] + c * d[1], a[2] + d[2]])


nxt, step, t0, maxn, last = 0, 1, time.perf_counter(), 0, np.zeros(3)
try:
    while nxt < len(edges) and step <= N:
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
                if step % 50 == 1 and step > 1:
                    last = isam.calculateEstimate().at(step - 1)
                else:
                    last = last if step > 1 else np.zeros(3)
                v.insert(step, 0, compose(last, m))
                last = compose(last, m)
            nxt += 1
        r = isam.update(g, v)
        step += 1
        if step % 500 == 0:
            print(f"pose {step}: {1e3 * (time.perf_counter() - t0) / step:.3f} ms per update so far, cliques {r.as_dict().get('cliques')}", flush=True)
except Exception as e:  # noqa: BLE001
    print(f"stopped at pose {step}: {e}")
print(f"{step - 1} updates, {1e3 * (time.perf_counter() - t0) / max(1, step - 1):.3f} ms per update")
