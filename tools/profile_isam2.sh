#!/bin/bash
# rocprofv3 kernel statistics + the library's own phase trace of the two incremental bench workloads driven from C++ (tests/cpp/isam2_harness,
# orderings replayed from the fixtures): usage  tools/profile_isam2.sh [poses]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/isam2prof
mkdir -p $O
python - "$O" "${1:-10000}" <<'PY'
import sys
sys.path.insert(0, ".")
import bench, shutil
out, poses = sys.argv[1], int(sys.argv[2])
for name, path in bench.isam2_sequences(out, poses).items():
    print(name, path)
PY
for w in visual city10000; do
  LMGPU_ISAM2_TRACE=1 tests/cpp/isam2_harness $O/$w.txt 0 replay:tests/golden/isam2_orderings_$w.bin > $O/trace_$w.json 2> $O/trace_$w.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- tests/cpp/isam2_harness $O/$w.txt 0 replay:tests/golden/isam2_orderings_$w.bin > $O/harness_$w.json 2> $O/prof_$w.log
  find $O/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_isam2_$w.csv
  rm -rf $O/prof_$w
  echo "== $w"; cat $O/trace_$w.txt | tail -5; head -14 $O/kernel_stats_isam2_$w.csv | cut -c1-150
done
rm -f $O/*.txt.seq
