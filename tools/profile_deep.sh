#!/bin/bash
# kernel timeline of one LM iteration of the deep-tree workloads (COLAMD).  usage: tools/profile_deep.sh OUTDIR [workloads...]
set -e
O=${1:-gpurun_out/deep}
shift || true
W=${@:-sphere2500 city10000 victoria_park}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
for w in $W; do
  python bench.py --workload $w --ordering colamd --steps 20 --no-cpu-baseline > $O/bench_${w}_colamd.json 2>> $O/bench.err
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --ordering colamd --steps 5 --no-cpu-baseline > $O/prof_$w.log 2>&1
  python tools/trace_segments.py $O/prof_$w v 3 > $O/segment_${w}_colamd.txt
  rm -rf $O/prof_$w
  echo "$w done" >> $O/progress.txt
done
