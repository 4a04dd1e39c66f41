#!/bin/bash
# A/B of prebuilt variants of the TEST library on the box: tools/variants/liblmgpu_test_<name>.so replaces liblmgpu_test.so for one bench run
set -e
cp gtsam_personal_amd/liblmgpu_test.so /tmp/liblmgpu_test_base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/liblmgpu_test_base.so gtsam_personal_amd/liblmgpu_test.so; else cp tools/variants/liblmgpu_test_$v.so gtsam_personal_amd/liblmgpu_test.so; fi
  timeout -k 10 200 python bench.py --dev-library --no-cpu-baseline > gpurun_out/lib_$v.json 2> gpurun_out/lib_$v.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/lib_{sys.argv[1]}.json"))
k = d["kernel_ms_per_step"]
print(sys.argv[1], "value", round(d["value"], 2), "ms", round(d["ms_per_step"], 3), "chain", round(k["chain"], 3), "err", d["error_after_one_iteration"])
PY
done
cp /tmp/liblmgpu_test_base.so gtsam_personal_amd/liblmgpu_test.so
