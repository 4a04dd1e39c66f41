set -e
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
k = d["kernel_ms_per_step"]
print(sys.argv[1], "value", round(d["value"], 2), "ms", round(d["ms_per_step"],3), "chain", round(k["chain"], 3), "err", d["error_after_one_iteration"])
PY
}
timeout -k 10 600 python -m pytest tests/test_gpu_c4.py tests/test_gpu_lookahead.py -m gpu -x -q > gpurun_out/r3_t2.log 2>&1 || { tail -30 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
run nomerge LMGPU_NO_MERGE=1
run merge50 LMGPU_CHAIN_FAR=50
run merge100 LMGPU_CHAIN_FAR=100
run merge70 LMGPU_CHAIN_FAR=70
run merge35 LMGPU_CHAIN_FAR=35
