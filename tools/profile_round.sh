#!/bin/bash
# profile of a round on the GPU box: bench JSON lines, rocprofv3 kernel stats (each its own run).  usage: tools/profile_round.sh r03
set -e
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python bench.py --steps 20 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --ordering schur --steps 20 --no-cpu-baseline > $O/bench_c4_schur.json 2>> $O/bench_c4.err
python bench.py --window 40 --steps 10 --no-cpu-baseline > $O/bench_c4band_metis.json 2>> $O/bench_c4.err
for w in sphere2500 city10000 victoria_park; do for o in colamd metis; do python bench.py --workload $w --ordering $o --steps 20 > $O/bench_${w}_${o}.json 2>> $O/bench_slam.err; done; done
python bench.py --workload isam2 --steps 5 > $O/bench_isam2.json 2>> $O/bench_slam.err
echo "bench lines done" >> $O/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --no-cpu-baseline --no-peaks > $O/prof.log 2>&1
echo "kernel trace done" >> $O/progress.txt
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
for w in sphere2500 city10000 victoria_park; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload $w --ordering colamd --steps 10 > $O/prof_$w.log 2>&1
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${w}_colamd.csv
  rm -rf $O/prof
done
# PMC passes: tools/pmc_round.sh (one counter group per run)
head -c 600 $O/bench_c4.json
