#!/bin/bash
# A/B of the chained launch's schedule parameter on the box
for v in "$@"; do
  LMGPU_CHAIN_FAR=$v timeout -k 10 200 python bench.py > gpurun_out/far_$v.json || exit 1
  python - "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/far_{sys.argv[1]}.json"))
k = d["kernel_ms_per_step"]
print("far", sys.argv[1], "value", round(d["value"], 2), "syrk", round(k["syrk"], 3), "err", d["error_after_one_iteration"])
PY
done
