#!/bin/bash
# scratch: the retained-points and dissection tests, then the phase times of the two big problems
set -o pipefail
tag=${1:-r5a}
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "retained or dissect or lockstep or lock_step or border" > gpurun_out/${tag}_tests.log 2>&1 || { tail -40 gpurun_out/${tag}_tests.log; exit 1; }
tail -3 gpurun_out/${tag}_tests.log
for w in ladybug venice; do
  for d in auto off; do
    timeout -k 10 300 python tools/phase_times.py $w 10 retained=auto dissection=$d >> gpurun_out/${tag}_phases.txt 2>&1 || { tail -20 gpurun_out/${tag}_phases.txt; exit 1; }
  done
done
cat gpurun_out/${tag}_phases.txt
