#!/usr/bin/env python3
"""timing/timeIncremental.cpp-style ISAM2 on the first N poses of the FULL city10000 graph (tests/golden/city10000.g2o) on the device: one
pose per update with the edges that reach back from it, the new pose initialised by dead reckoning from the device's own estimate
(refreshed every 50 poses).  `--check` replays the same updates through the CPU oracle and compares the final state.
    python tests/tools/isam2_long_run.py [N=3000] [--check]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(n_poses=3000, check=False, verbose=True):
    import oracle_harness as oh  # the ccolamd callback (the reference's vendored CCOLAMD, compiled by oracle/Makefile) and, with check, the oracle
    from gtsam_personal_amd import ISAM2, ISAM2Params, NonlinearFactorGraph, Values, noiseModel
    from gtsam_personal_amd.datasets import readG2o
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

    out = dict(stopped=None)
    steps, nxt, step, t0, last, r = [], 0, 1, time.perf_counter(), np.zeros(3), None
    try:
        while nxt < len(edges) and step <= n_poses:
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
                    last = compose(last if step > 1 else np.zeros(3), m)
                    v.insert(step, 0, last)
                nxt += 1
            r = isam.update(g, v)
            if check:
                steps.append((g, v))
            step += 1
            if verbose and step % 1000 == 0:
                print(f"pose {step}: {1e3 * (time.perf_counter() - t0) / step:.3f} ms per update so far, cliques {r.as_dict().get('cliques')}", flush=True)
    except Exception as e:  # noqa: BLE001
        out["stopped"] = f"pose {step}: {e}"
    out["updates"] = step - 1
    out["ms_per_update"] = 1e3 * (time.perf_counter() - t0) / max(1, step - 1)
    if check and out["stopped"] is None:
        t1 = time.perf_counter()
        p = ISAM2Params()
        orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
        ro = None
        for g, v in steps:
            ro = orc.update(g, v)
        out["oracle_ms_per_update"] = 1e3 * (time.perf_counter() - t1) / len(steps)
        out["last_counts_equal"] = ro == r.as_dict()
        eg, eo = isam.calculateEstimate(), orc.calculateEstimate()
        out["max_rel_diff"] = max(float(np.max(np.abs(eg.at(k) - eo.at(k)) / np.maximum(1.0, np.abs(eo.at(k))))) for k in eo.keys())
        cg, co = isam.cliques(), orc.cliques()
        out["same_tree"] = len(cg) == len(co) and all(a[0] == b[0] and a[1] == b[1] and a[3] == b[3] for a, b in zip(cg, co))
        out["widest_clique"] = max(R.shape[1] for _, _, R, _ in cg)
    isam.close()
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    res = run(n, check="--check" in sys.argv)
    print(res)
