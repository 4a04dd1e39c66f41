#!/bin/bash
# PMC passes of the default bench command, one counter group per run (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/pmc_round.sh r03
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-peaks > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?" >> $O/progress.txt
  echo "pmc $tag done"
done
python tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json
rm -rf $O/pmc_*/
tail -3 $O/progress.txt
