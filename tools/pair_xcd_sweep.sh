#!/bin/bash
# Developer probe (round 3): the pair kernels' logical blocks in runs of <group> per XCD (SK_PAIR_XCD_GROUP / _LONG), Schur-assembly phase per iteration.
cd "$(dirname "$0")/.."
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step %.3f  schur_assemble %.4f' % (d['ms_per_step'], d['phases_ms_per_step']['schur_assemble']))
except Exception as e:
    print('   failed:', e)"; }
for wl in ladybug-1723-156502 venice-1778-993923; do
B="bench.py --workload $wl --steps 10 --warmup 3 --cpu-iters 0 --no-c5 --no-alone"
for g in "0 0" "2 2" "8 8" "32 32" "8 0" "0 8" "4 16" "16 4"; do
set -- $g
echo "== $wl: group short $1, long $2"; SK_PAIR_XCD_GROUP=$1 SK_PAIR_XCD_GROUP_LONG=$2 python3 $B 2>/dev/null | show
done
done
