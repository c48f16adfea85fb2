#!/bin/bash
# the two chains of the dissected factorisation side by side on ONE device take 10-13 ms where they take
# 5.0 + 3.0 ms one after the other — and 6.6 ms under rocprofv3's kernel tracing.  What the profiler changes is how
# dispatches complete (a completion signal on every packet, handled by the tool); this probe runs the same bench line with
# the runtime's signal / queue knobs.
#   bash tools/dissection_probe.sh > gpurun_out/r03_dissection_probe.txt 2>&1
cd "$(dirname "$0")/.."
B="bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-c5 --no-alone"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step %.3f  cholesky %.3f' % (d['ms_per_step'], d['phases_ms_per_step']['cholesky']))
except Exception as e:
    print('   failed:', e)"; }
echo "== undissected"; python3 $B 2>/dev/null | show
echo "== dissected (side by side)"; python3 $B --dissection on 2>/dev/null | show
echo "== dissected, one after the other (SK_DISSECT_SERIAL=1)"; SK_DISSECT_SERIAL=1 python3 $B --dissection on 2>/dev/null | show
echo "== dissected, HSA_ENABLE_INTERRUPT=0 (signals polled, no interrupts)"; HSA_ENABLE_INTERRUPT=0 python3 $B --dissection on 2>/dev/null | show
echo "== dissected, GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 python3 $B --dissection on 2>/dev/null | show
echo "== dissected, HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 python3 $B --dissection on 2>/dev/null | show
echo "== dissected, ROC_ACTIVE_WAIT_TIMEOUT / ROC_SIGNAL_POOL default; DEBUG_HIP_BLOCK_SYNC no; AMD_DIRECT_DISPATCH=0"; AMD_DIRECT_DISPATCH=0 python3 $B --dissection on 2>/dev/null | show
echo "== dissected, under rocprofv3 --kernel-trace"; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/prof_dis -- python3 $OLDPWD/$B --dissection on 2>/dev/null | show
