#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY: the reference's COLAMD and METIS orderings (Ordering::Colamd / Ordering::Metis through oracle/_ref = the
reference's vendored C sources) of the general sparse bench workloads, as permutations of the ascending key list:
tests/golden/slam_orderings.npz.  The ordering is a boundary INPUT of the hot path (computed once at optimizer construction,
LevenbergMarquardtParams.h:112-117); bench.py --workload sphere2500|city10000 reads it from here."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_harness as oh  # noqa: E402
from bench import slam_workload  # noqa: E402

out = {}
for name in ("sphere2500", "city10000", "victoria_park"):
    graph, initial = slam_workload(name)
    keys = np.array(sorted(graph.keys()), dtype=np.uint64)
    for oname, fn in (("colamd", oh.colamd), ("metis", oh.metis)):
        order = np.array(fn(graph), dtype=np.uint64)
        out[f"{name}_{oname}"] = np.searchsorted(keys, order).astype(np.int32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "slam_orderings.npz"), **out)
print({k: v.shape for k, v in out.items()})
