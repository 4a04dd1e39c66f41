#!/bin/bash
# rocprofv3 kernel statistics of the incremental city10000 workload driven from C++ (tests/cpp/isam2_harness), device-driven loop
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/isam2prof
mkdir -p $O
python tests/tools/isam2_sequence.py $O/seq.txt 10000 --relative > $O/gen.log 2>&1
echo "sequence written" >> $O/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- tests/cpp/isam2_harness $O/seq.txt 0 oracle/_ref/libccolamd_ref.so > $O/harness.json 2> $O/prof.log
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_isam2_city10000.csv
rm -rf $O/prof $O/seq.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/isam2prof/harness.json")); d.pop("estimate")
print(d)
PY
head -12 $O/kernel_stats_isam2_city10000.csv
