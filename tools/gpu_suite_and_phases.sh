#!/bin/bash
# Developer tool: the whole GPU suite (every failure listed) and the phase times of the two big problems, in one gpurun call:
#   gpurun --timeout 1200 -- "bash tools/gpu_suite_and_phases.sh <tag>"
set -o pipefail
tag=${1:-r5e}
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -15 gpurun_out/${tag}_tests.log
for w in ladybug venice; do
  for d in auto off; do
    timeout -k 10 300 python tools/phase_times.py $w 10 retained=auto dissection=$d >> gpurun_out/${tag}_phases.txt 2>&1 || { tail -20 gpurun_out/${tag}_phases.txt; exit 1; }
  done
done
cat gpurun_out/${tag}_phases.txt
exit $rc
