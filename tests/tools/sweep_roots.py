"""Camera roots of 9 n + 1 columns for 32 values of n (82 .. 2566 columns: every path of the dense front -- batched one-panel, dataflow
panels, chained steps with 2 .. 9 steps, front tail, two-launch panels) against the oracle at two values of lambda."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_harness as oh
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
from gtsam_personal_amd.synthetic import make_bal
worst=0
for n_cam in list(range(9, 135, 5)) + [143, 171, 199, 228, 256, 285]:
    graph, initial, _, ordering = make_bal(n_cam=n_cam, n_pt=10 * n_cam, obs_per_point=5, seed=1000 + n_cam)
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    opt.linearize(); orc.linearize()
    for lam in (1e-6, 1e-2):
        dk, d, e0, e1 = opt.solve(lam)
        rc, do, o0, o1 = orc.solve(lam)
        assert rc == 0
        a = np.concatenate([dk[k] for k in sorted(dk)]); b = np.concatenate([do[k] for k in sorted(do)])
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        worst = max(worst, rel)
        assert rel < 1e-6, (n_cam, lam, rel)
    opt.close()
print('sweep ok, worst rel', worst)
