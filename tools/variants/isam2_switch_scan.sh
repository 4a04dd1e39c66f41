# the full city10000 incremental run of the C++ driver under each ISAM2 development switch (test library preloaded): mean / worst update, final estimate
python - <<'PY'
import sys
sys.path.insert(0, ".")
import bench, os
os.makedirs("gpurun_out/abi", exist_ok=True)
for name, path in bench.isam2_sequences("gpurun_out/abi", 10000).items():
    print(name, path)
PY
for sw in X LMGPU_ISAM2_NO_BYVALUE LMGPU_ISAM2_NO_MIRROR LMGPU_ISAM2_NO_PREWALK LMGPU_ISAM2_LATE_WALK_PREP "$@"; do
  env LD_PRELOAD=$PWD/gtsam_personal_amd/liblmgpu_test.so $sw=1 timeout -k 10 120 tests/cpp/isam2_harness gpurun_out/abi/city10000.txt 0 replay:tests/golden/isam2_orderings_city10000.bin > gpurun_out/abi/scan_$sw.json 2> gpurun_out/abi/scan_$sw.err
  echo "$sw rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/abi/scan_$sw.json')); print({k: d[k] for k in ('ms_per_update_after_first','p50_ms','p99_ms','worst_update_ms','calculate_estimate_ms')})"
  tail -2 gpurun_out/abi/scan_$sw.err
done
