#!/usr/bin/env python3
"""GPU BOX (needs the device and oracle/_ref): records the constrained-COLAMD orderings of the two incremental bench workloads, i.e. what
the reference-side caller's ccolamd returns for every ISAM2 update (lmgpu_ccolamd_fn), by running the C++ driver once with the
reference's vendored CCOLAMD (oracle/_ref/libccolamd_ref.so) and `record:`.  The files are then a fixture-carried boundary input of
`bench.py --workload isam2` (which replays them and loads nothing from oracle/), like the METIS permutation of the batch benchmark.
    python tests/tools/make_isam2_orderings.py <out dir> [poses]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import isam2_sequences  # noqa: E402

out_dir = sys.argv[1]
poses = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
os.makedirs(out_dir, exist_ok=True)
harness = os.path.join(ROOT, "tests", "cpp", "isam2_harness")
with tempfile.TemporaryDirectory() as d:
    for name, path in isam2_sequences(d, poses).items():
        rec = os.path.join(out_dir, f"isam2_orderings_{name}.bin")
        r = subprocess.run([harness, path, "0", os.path.join(ROOT, "oracle", "_ref", "libccolamd_ref.so"), "record:" + rec], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900)
        print(name, r.returncode, r.stdout[:300].decode(), os.path.getsize(rec))
