#!/bin/bash
# A/B runs of bench.py's headline on ONE box (fresh process each): `label|environment|arguments` per line on stdin.
#   printf 'r12||--retained 12\nr24||--retained 24\n' | bash tools/ab_runs.sh > gpurun_out/ab.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
i=0
while IFS='|' read -r label envs args; do
  [ -z "$label" ] && continue
  i=$((i+1))
  env $envs SK_BENCH_DETAILS=/tmp/ab_details_$i.json python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-alone --no-c5 $args > /tmp/ab_$i.out 2> /tmp/ab_$i.err
  python3 - "$label" "$i" <<'PY'
import json, sys
label, i = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open("/tmp/ab_%s.out" % i).read().strip().splitlines()[-1])
    ph = d.get("phases_ms_per_step") or {}
    ch = (d.get("roofline") or {}).get("chain") or {}
    print("%-28s %.1f it/s  %.3f ms  cholesky %.3f  assemble %.3f  jac %.3f  | %s | steps %s model/measured %s" % (
        label, d["value"], d["ms_per_step"], ph.get("cholesky", 0), ph.get("schur_assemble", 0), ph.get("jacobian_eval", 0), d["config"].get("plan"), ch.get("steps"), ch.get("measured_over_model")), flush=True)
except Exception as e:  # noqa: BLE001
    print(label, "failed:", repr(e)); print(open("/tmp/ab_%s.err" % i).read()[-1500:])
PY
done
