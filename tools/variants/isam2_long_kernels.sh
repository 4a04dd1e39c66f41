# which device operations of the full city10000 incremental run take more than a millisecond (rocprofv3 kernel trace of the C++ driver)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/abi
mkdir -p $O
python - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
print(bench.isam2_sequences("gpurun_out/abi", 10000))
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_long -- tests/cpp/isam2_harness $O/city10000.txt 0 replay:tests/golden/isam2_orderings_city10000.bin > $O/long.json 2> $O/long.err
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/abi/prof_long/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-70:], r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", "")))
rows.sort()
t0 = rows[0][0]
print(len(rows), "kernels")
for i, (a, b, n, g, w) in enumerate(rows):
    if b - a > 1_000_000:
        print(f"#{i} at {(a - t0) / 1e6:9.1f} ms: {(b - a) / 1e6:8.2f} ms  {n} grid {g} wg {w}")
gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)), reverse=True)[:8]
for g, i in gaps:
    print(f"gap {g / 1e6:8.2f} ms after #{i} {rows[i][2]} (at {(rows[i][1] - t0) / 1e6:9.1f} ms), next {rows[i + 1][2]}")
PY
rm -rf $O/prof_long
python -c "
import json; d=json.load(open('gpurun_out/abi/long.json')); print({k: d[k] for k in ('ms_per_update_after_first','worst_update_ms','calculate_estimate_ms')})"
