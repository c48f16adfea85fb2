#!/bin/bash
# Run-to-run spread of the headline (round-4 verdict item 8): N fresh processes of
#   python3 bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-alone --no-c5
# on one box, each with SK_DEBUG=queues (the queue trial's table and choice on stderr).  One line per run: it/s, ms per step,
# the Cholesky phase, the queue combination chosen.
#   bash tools/run_to_run.sh 9 > gpurun_out/r05_run_to_run.txt
N=${1:-9}
shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq 1 $N); do
  SK_DEBUG=queues SK_BENCH_DETAILS=/tmp/rtr_details_$i.json python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-alone --no-c5 "$@" > /tmp/rtr_$i.out 2> /tmp/rtr_$i.err
  python3 - "$i" <<'PY'
import json, sys, re
i = sys.argv[1]
try:
    d = json.loads(open("/tmp/rtr_%s.out" % i).read().strip().splitlines()[-1])
    err = open("/tmp/rtr_%s.err" % i).read()
    m = re.search(r"combination (\d+)", err)
    ph = d.get("phases_ms_per_step") or {}
    print("run %s: %.1f it/s  %.3f ms  cholesky %.3f  assemble %.3f  jac %.3f  queue combination %s" % (
        i, d["value"], d["ms_per_step"], ph.get("cholesky", 0), ph.get("schur_assemble", 0), ph.get("jacobian_eval", 0), m.group(1) if m else "?"), flush=True)
    for line in err.splitlines():
        if "synthetic factorisations" in line or "secondary context" in line:
            print("   " + line[:1500])
except Exception as e:  # noqa: BLE001
    print("run %s failed: %r" % (i, e)); print(open("/tmp/rtr_%s.err" % i).read()[-2000:])
PY
done
