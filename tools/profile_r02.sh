#!/bin/bash
# round-2 profile of the default bench command on the GPU box: JSON lines, rocprofv3 kernel stats, PMC passes (each its own run).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
python bench.py --steps 20 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --ordering schur --steps 20 --no-cpu-baseline > $O/bench_c4_schur.json 2>> $O/bench_c4.err
for w in sphere2500 city10000; do for o in colamd metis; do python bench.py --workload $w --ordering $o --steps 20 > $O/bench_${w}_${o}.json 2>> $O/bench_slam.err; done; done
echo "bench lines done" >> $O/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --no-cpu-baseline --no-peaks > $O/prof.log 2>&1
echo "kernel trace done" >> $O/progress.txt
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
timeout -k 10 300 python tests/tools/bench_isam2.py > $O/bench_isam2.txt 2>> $O/bench_slam.err
tools/backsolve_bench > $O/backsolve_bench_now.txt 2>&1
# PMC passes: tools/pmc_r02.sh (one counter group per run)
head -c 600 $O/bench_c4.json
