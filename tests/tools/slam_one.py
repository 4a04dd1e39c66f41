"""One dataset of tests/tools/bench_slam.py for profiling: python tests/tools/slam_one.py {sphere|city|victoria} {colamd|metis} [solves]
Prints wall time per damped solve (linearize once, then `solves` x solve(1e-3))."""
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
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
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
opt.solve(1e-3)
t0 = time.perf_counter()
for _ in range(n):
    opt.solve(1e-3)
dt = (time.perf_counter() - t0) / n
print(f"{which}/{order}: {1e3 * dt:.3f} ms per solve (wall, incl. the delta download), {opt.num_fronts()} fronts")
