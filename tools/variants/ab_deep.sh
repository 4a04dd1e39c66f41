# A/B of development switches on the deep-tree workloads (test library).  usage: tools/variants/ab_deep.sh "ENV=1 ..." "ENV=2" ...
run() { # name, env...
  name=$1; shift
  for w in sphere2500 city10000 victoria_park; do
    env "$@" timeout -k 10 120 python bench.py --dev-library --workload $w --ordering colamd --steps 20 > gpurun_out/abd_${name}_$w.json 2> gpurun_out/abd_${name}_$w.err
    python - "$name" "$w" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abd_{sys.argv[1]}_{sys.argv[2]}.json"))
print(sys.argv[1], sys.argv[2], "ms", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms_one_iterate"].items()})
PY
  done
}
for v in "$@"; do
  run "$(echo $v | tr '= ' '__')" $v
done
