# A/B of library builds / switches on the ISAM2 C++ driver in one box.  usage: tools/variants/ab_isam2.sh "ENV=1" "ENV=0" ...  (test library preloaded)
python - <<'PY'
import sys
sys.path.insert(0, ".")
import bench, os
os.makedirs("gpurun_out/abi", exist_ok=True)
for name, path in bench.isam2_sequences("gpurun_out/abi", 2000).items():
    print(name, path)
print("fixed_lag", bench.fixed_lag_sequence("gpurun_out/abi"))
PY
run() { # label, env...
  label=$1; shift
  for w in visual city10000 fixed_lag; do
    FX=tests/golden/isam2_orderings_$w.bin
    rep=""; [ $w = visual ] && rep="repeat:3"
    best=999
    for i in 1 2 3; do
      env "$@" tests/cpp/isam2_harness gpurun_out/abi/$w.txt 0 replay:$FX $rep > gpurun_out/abi/out.json 2>/dev/null
      v=$(python -c "import json; d=json.load(open('gpurun_out/abi/out.json')); print(d['ms_per_update_after_first'])")
      best=$(python -c "print(min($best, $v))")
    done
    echo "$label $w best-of-3 ms_per_update_after_first $best"
  done
}
run product X=0
for v in "$@"; do
  run "testlib_$v" LD_PRELOAD=$PWD/gtsam_personal_amd/liblmgpu_test.so $v
done
