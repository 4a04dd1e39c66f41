"""Repeated-solve stress of the sparse-graph paths (graph replay, merged back-substitution, batched mid-size fronts):
    python tests/tools/stress_slam.py {sphere|city|victoria} {colamd|metis} [solves]
Every solve of the same linearization must return the same update (bitwise since the row-owner assembly; the bound checked is 1e-9)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_harness as oh  # noqa: E402
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams, noiseModel  # noqa: E402
from gtsam_personal_amd.datasets import chain_initial_pose3, load2D, load3D, readG2o  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "..", "golden")
which, order = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
if which == "sphere":
    g, _ = load3D(os.path.join(GOLD, "sphere2500.txt"))
    init = chain_initial_pose3(g)
    g.add_PriorFactorPose3(0, np.eye(3), np.zeros(3), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
elif which == "city":
    g, init = readG2o(os.path.join(GOLD, "city10000.g2o"))
    g.add_PriorFactorPose2(0, init.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
else:
    g, init = load2D(os.path.join(GOLD, "victoria_park.txt"))
    g.add_PriorFactorPose2(0, init.at(0), noiseModel.Diagonal.Variances([1e-6, 1e-6, 1e-8]))
ordering = oh.colamd(g) if order == "colamd" else oh.metis(g)
opt = LevenbergMarquardtOptimizer(g, init, ordering, LevenbergMarquardtParams(), device=0)
opt.linearize()
ref, worst = {}, 0.0
t0 = time.perf_counter()
for k in range(n):
    lam = [1e-5, 1e-3, 1e-1][k % 3]
    _, d, e0, e1 = opt.solve(lam)
    if k < 3:
        ref[lam] = (d.copy(), e1)
    else:
        r, re1 = ref[lam]
        rel = float(np.linalg.norm(d - r) / np.linalg.norm(r))
        worst = max(worst, rel)
        assert rel < 1e-9 and abs(e1 - re1) <= 1e-9 * max(1.0, abs(re1)), (k, lam, rel, e1, re1)
print(f"{which}/{order}: {n} solves in {time.perf_counter() - t0:.1f} s, worst relative deviation {worst:.3e}: OK")
