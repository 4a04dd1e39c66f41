"""Side measurement: LevenbergMarquardt on the bench workload (C4) run to convergence, checking that the cost never increases."""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
from gtsam_personal_amd.synthetic import make_bal
graph, initial, _, ordering = make_bal(1000, 100000, 10, seed=42)
params = LevenbergMarquardtParams()
opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
e0 = opt.error()
t0 = time.perf_counter()
errs = [e0]
while True:
    prev = opt.error()
    opt.iterate()
    errs.append(opt.error())
    if opt.iterations() >= 100 or abs(prev - opt.error()) <= max(1e-5 * prev, 1e-5) or len(errs) > 60:
        break
dt = time.perf_counter() - t0
print("C4 LM to convergence: iterations", opt.iterations(), "inner", opt.getInnerIterations(), "time %.1f ms" % (1e3 * dt), "error %.6g -> %.6g" % (e0, opt.error()))
print("trajectory", ["%.6g" % e for e in errs])
assert all(b <= a * (1 + 1e-12) for a, b in zip(errs, errs[1:]))
