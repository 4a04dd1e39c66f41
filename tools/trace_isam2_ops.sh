#!/bin/bash
# ordered device operations (kernels, copies) of the incremental workloads: what one ISAM2 update issues
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/isam2ops
mkdir -p $O
python - "$O" "${1:-400}" <<'PY'
import sys
sys.path.insert(0, ".")
import bench
for name, path in bench.isam2_sequences(sys.argv[1], int(sys.argv[2])).items():
    print(name, path)
PY
for w in visual city10000; do
  FX=tests/golden/isam2_orderings_$w.bin
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_$w -- tests/cpp/isam2_harness $O/$w.txt 0 replay:$FX > $O/harness_$w.json 2> $O/prof_$w.log || true
  python - "$O/prof_$w" "$O/ops_$w.txt" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-60:]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
t0 = rows[0][0] if rows else 0
with open(sys.argv[2], "w") as o:
    for a, b, n in rows:
        o.write(f"{(a - t0) / 1e3:10.1f} us  +{(b - a) / 1e3:7.1f}  {n}\n")
print(len(rows), "ops")
PY
  rm -rf $O/prof_$w
done
