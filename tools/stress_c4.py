"""Stress of the dataflow launches: N damped solves of C4 from the same linearization; every solve must give the same delta
(bitwise: the assembly has no atomics, every sum has a fixed order) and status OK.
    python tools/stress_c4.py [solves] [cams] [points]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams  # noqa: E402
from gtsam_personal_amd.synthetic import make_bal  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cams = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pts = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
graph, initial, _, ordering = make_bal(n_cam=cams, n_pt=pts, obs_per_point=10, seed=42)
opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
opt.linearize()
ref = None
worst = 0.0
t0 = time.perf_counter()
for k in range(n):
    lam = [1e-5, 1e-3, 1e-1][k % 3]
    dk, d, e0, e1 = opt.solve(lam)
    if k < 3:
        ref = ref or {}
        ref[lam] = (d.copy(), e1)
    else:
        r, re1 = ref[lam]
        rel = float(np.linalg.norm(d - r) / np.linalg.norm(r))
        worst = max(worst, rel)
        assert rel < 1e-12 and abs(e1 - re1) <= 1e-12 * abs(re1), (k, lam, rel, e1, re1)
    if k % 50 == 49:
        print(f"solve {k + 1}: worst relative deviation of delta so far {worst:.3e}", flush=True)
print(f"{n} solves in {time.perf_counter() - t0:.1f} s, worst deviation {worst:.3e}: OK")
