#!/bin/bash
# scratch: phase times of the two big problems with retained points, dissection on (auto) and off
set -o pipefail
tag=${1:-r5c}
for w in ladybug venice; do
  for d in auto off; do
    timeout -k 10 300 python tools/phase_times.py $w 10 retained=auto dissection=$d >> gpurun_out/${tag}_phases.txt 2>&1 || { tail -20 gpurun_out/${tag}_phases.txt; exit 1; }
  done
done
cat gpurun_out/${tag}_phases.txt
