# A/B of development switches on the C4 headline bench (test library).  usage: tools/variants/ab_c4.sh "ENV=1" "X=0" ...
for v in "$@"; do
  name=$(echo $v | tr '= ' '__')
  env $v timeout -k 10 200 python bench.py --dev-library --steps 20 --no-cpu-baseline --no-peaks > gpurun_out/abc4_$name.json 2> gpurun_out/abc4_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abc4_{sys.argv[1]}.json"))
k = d.get("kernel_ms_per_step") or {}
print(sys.argv[1], "ms", round(d["ms_per_step"], 3), "error", d["error_after_one_iteration"], {a: round(b, 3) for a, b in k.items() if "assemble" in a or "schur" in a or "gather" in a})
PY
done
