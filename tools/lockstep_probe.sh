#!/bin/bash
# Developer probe (round 3): the dissected factorisation with the tail front riding in the head's launches (SK_DISSECT_LOCKSTEP=1),
# for a range of cuts (SK_DISSECT_AT = cameras in the head).
cd "$(dirname "$0")/.."
B="bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-c5 --no-alone"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step %.3f  cholesky %.3f  value %.2f' % (d['ms_per_step'], d['phases_ms_per_step']['cholesky'], d['value']))
except Exception as e:
    print('   failed:', e)"; }
echo "== undissected"; python3 $B 2>/dev/null | show
for at in "$@"; do
echo "== head $at cameras, tail in the head's launches"; SK_DISSECT_AT=$at SK_DISSECT_LOCKSTEP=1 SK_DISSECT_TIMING=1 python3 $B --dissection on 2>gpurun_out/ls_err.txt | show; grep "dissected factorisation" gpurun_out/ls_err.txt | tail -1; grep -i "timed out\|error" gpurun_out/ls_err.txt | head -3
done
echo "== undissected"; python3 $B 2>/dev/null | show
