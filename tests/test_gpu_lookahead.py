"""Multi-panel HBM front (64 cameras -> 577 x 577 root, 3 outer panels of 256 rows): the dense path against the oracle,
in the default single-stream mode and (development aid) in the experimental two-stream look-ahead mode selected with
LMGPU_LOOKAHEAD=1 in the environment of the test process."""
import os

import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
from gtsam_personal_amd.synthetic import make_bal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_cam", [64, int(os.environ.get("LMGPU_TEST_NCAM", "200"))])
def test_outer_panels_match_oracle(n_cam):
    graph, initial, _, ordering = make_bal(n_cam=n_cam, n_pt=25 * n_cam, obs_per_point=8, seed=17)
    params = LevenbergMarquardtParams()
    opt = LevenbergMarquardtOptimizer(graph, initial, ordering, params, device=0)
    orc = oh.OracleProblem(graph, initial, ordering)
    info = opt.front_info(opt.num_fronts() - 1)
    assert info["cls"] == 1 and info["nf"] == 9 * n_cam
    opt.set_kernel_timing(bool(os.environ.get("LMGPU_TEST_TIMERS")))
    opt.linearize()
    orc.linearize()
    for lam in (1e-5, 1e-2):
        dk, d, e0, e1 = opt.solve(lam)
        rc, do, o0, o1 = orc.solve(lam)
        assert rc == 0
        a = np.concatenate([dk[k] for k in sorted(dk)])
        b = np.concatenate([do[k] for k in sorted(do)])
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel < 1e-6, (lam, rel)
        assert abs(e1 - o1) <= 1e-6 * max(1.0, abs(o1))
