#!/bin/bash
# A/B of prebuilt library variants on the box: the step and panel kernel times of the default bench
for v in "$@"; do
  cp tools/variants/liblmgpu_$v.so gtsam_personal_amd/liblmgpu.so || exit 1
  timeout -k 10 200 python bench.py > gpurun_out/var_$v.json || exit 1
  python - "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/var_{sys.argv[1]}.json"))
k = d["kernel_ms_per_step"]
print(sys.argv[1], "value", round(d["value"], 2), "syrk", round(k["syrk"], 3), "panel", round(k["panel"], 3), "err", d["error_after_one_iteration"])
PY
done
