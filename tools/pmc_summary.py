#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files (one pass per counter set) into a per-kernel summary.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write [gpurun_out/pmc_sq] > profiles/rNN/pmc_summary.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B
(MI355X_MICROARCH.md, HBM section), so `hbm_read_bytes_corrected` = 2 x FETCH_SIZE x 1024 is what to compare with a
byte count for the streaming kernels; the uncorrected figure is kept beside it.  Counters are per dispatch; the
summary gives the mean per launch and the launch count of every kernel.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha256 over the kernel sources: bench.py reports the counters of a summary only while the kernels are the ones that were profiled"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gtsam_personal_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("lmgpu::", "")


def main():
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                a = agg[k][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                if row["Dispatch_Id"] not in seen:  # one duration per dispatch (rows repeat per counter)
                    seen.add(row["Dispatch_Id"])
                    dur[k][0] += 1
                    dur[k][1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    out = {}
    for k in sorted(agg):
        e = {"launches_per_pass": max(v[0] for v in agg[k].values())}
        for c, (n, s) in agg[k].items():
            e[c + "_mean"] = s / n
        if "FETCH_SIZE" in agg[k]:
            e["hbm_read_bytes_uncorrected"] = e["FETCH_SIZE_mean"] * 1024
            e["hbm_read_bytes_corrected"] = 2 * e["FETCH_SIZE_mean"] * 1024
        if "WRITE_SIZE" in agg[k]:
            e["hbm_write_bytes"] = e["WRITE_SIZE_mean"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k] and e.get("SQ_BUSY_CYCLES_mean", 0) > 0:
            e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / e["SQ_BUSY_CYCLES_mean"]
        e["avg_us_under_pmc"] = dur[k][1] / max(1, dur[k][0])
        out[k] = e
    out["_kernel_source_sha256"] = kernel_source_hash()
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
