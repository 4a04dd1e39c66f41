"""development aid: one damped solve of the C4-sized synthetic problem; prints scalars to compare stream modes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
from gtsam_personal_amd.synthetic import make_bal
ncam = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
graph, initial, _, ordering = make_bal(ncam, 100 * ncam, 10, seed=42)
opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
opt.linearize()
for rep in range(3):
    try:
        dk, d, e0, e1 = opt.solve(1e-5)
        print(f"rep {rep}: e0 {e0:.10e} e1 {e1:.10e} |d| {np.linalg.norm(d):.10e} d[:3] {d[:3]} d[-3:] {d[-3:]}")
    except Exception as e:
        print("rep", rep, "EXC", e)
