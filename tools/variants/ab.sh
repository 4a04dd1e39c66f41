set -e
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --dev-library --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
k = d["kernel_ms_per_step"]
print(sys.argv[1], "value", round(d["value"], 2), "ms", round(d["ms_per_step"],3), "chain", round(k["chain"], 3), "err", d["error_after_one_iteration"])
PY
}
if [ -n "$TESTS" ]; then
timeout -k 10 600 python -m pytest $TESTS -m gpu -x -q > gpurun_out/r3_t2.log 2>&1 || { tail -30 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
fi
for v in "$@"; do
  run "$(echo $v | tr '= ' '__')" $v
done
