#!/usr/bin/env python3
"""Side measurement (not the bench.py metric): one LM iteration on the general sparse graphs C1 / C3 at full size,
GPU vs. the single-thread CPU oracle.   python tests/tools/bench_slam.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_harness as oh  # noqa: E402
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams, noiseModel  # noqa: E402
from gtsam_personal_amd.datasets import chain_initial_pose3, load2D, load3D, readG2o  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def run(name, graph, initial, ordering):
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    opt.save_values()
    st = opt.copy_state()
    for _ in range(2):
        opt.restore_values(st)
        opt.iterate()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        opt.restore_values(st)
        opt.iterate()
    gpu = (time.perf_counter() - t0) / n
    tm = opt.timings()
    opt.set_kernel_timing(True)
    opt.restore_values(st)
    opt.iterate()
    kt = opt.kernel_times()
    ncls = [0, 0]
    sizes = []
    for i in range(opt.num_fronts()):
        fi = opt.front_info(i)
        ncls[fi["cls"]] += 1
        if fi["cls"] == 1:
            sizes.append((fi["n"], fi["nf"]))
    print("   kernel ms:", {k: round(v["ms"], 2) for k, v in kt.items() if v["ms"] > 0}, " launches:", {k: v["launches"] for k, v in kt.items() if v["launches"] > 0})
    print("   LDS fronts", ncls[0], "HBM fronts", ncls[1], "largest (n, nf):", sorted(sizes)[-5:])
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(params)
    t0 = time.perf_counter()
    orc.lm_iterate(params)
    cpu = time.perf_counter() - t0
    levels = max(opt.front_info(i)["level"] for i in range(opt.num_fronts())) + 1 if "level" in opt.front_info(0) else -1
    print(f"{name}: fronts {opt.num_fronts()} levels {levels}  GPU {1e3 * gpu:.2f} ms/iter ({1 / gpu:.1f} it/s)   oracle {1e3 * cpu:.1f} ms/iter   x{cpu / gpu:.1f}"
          f"   [eliminate {tm['eliminate_ms']:.2f} ms, backsub {tm['backsub_ms']:.2f} ms of the last iterate]")


g, _ = load3D(os.path.join(GOLD, "sphere2500.txt"))
init = chain_initial_pose3(g)
g.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
run("sphere2500 / METIS", g, init, oh.metis(g))
run("sphere2500 / COLAMD", g, init, oh.colamd(g))
g, init = readG2o(os.path.join(GOLD, "city10000.g2o"))
g.add_PriorFactorPose2(0, init.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
run("city10000 / METIS", g, init, oh.metis(g))
run("city10000 / COLAMD", g, init, oh.colamd(g))
from gtsam_personal_amd.datasets import load2D  # noqa: E402
g, init = load2D(os.path.join(GOLD, "victoria_park.txt"))
g.add_PriorFactorPose2(0, init.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
run("victoria_park / METIS", g, init, oh.metis(g))
run("victoria_park / COLAMD", g, init, oh.colamd(g))
