#!/usr/bin/env python3
"""Writes the input of tests/cpp/isam2_harness for the incremental city10000 workload (timing/timeIncremental.cpp: one pose per update):
    python tests/tools/isam2_sequence.py <out file> [n_poses=10000] [--relative]
The initial values are recorded from the oracle's run of the same updates; --relative writes every new pose as "previous estimate (+)
odometry", resolved by the harness from the device's own estimate at update time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_harness as oh  # noqa: E402
from gtsam_personal_amd import ISAM2Params  # noqa: E402
from isam2_examples import incremental_pose2_steps, write_isam2_sequence  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 10000
p = ISAM2Params()
orc = oh.OracleISAM2(p.relinearizeThreshold, p.relinearizeSkip, p.enableRelinearization, p.optimizationParams.wildfireThreshold)
g2o = os.path.join(ROOT, "tests", "golden", "city10000.g2o" if n > 400 else "city10000_head.g2o")
steps = []
for g, v in incremental_pose2_steps(g2o, n, lambda k: orc.calculateEstimate().at(k)):
    orc.update(g, v)
    steps.append((g, v))
write_isam2_sequence(out, p, steps, relative_pose2="--relative" in sys.argv)
print(f"{len(steps)} updates -> {out}")
