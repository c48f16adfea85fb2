R=$PWD
for i in 1 2 3; do python tools/phase_times.py ladybug 20 2>&1 | tail -n 1; done
python tools/phase_times.py venice 8 2>&1 | tail -n 1
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/tools/phase_times.py ladybug 6 > $R/gpurun_out/r4j_lady.txt 2>&1; for w in -3; do python3 $R/tools/trace_point_phases.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) $w; echo; done > $R/gpurun_out/r4j_lady_trace.txt
rm -rf /tmp/kt2; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -- python3 $R/tools/phase_times.py venice 5 > $R/gpurun_out/r4j_venice.txt 2>&1; for w in -4 -3; do python3 $R/tools/trace_point_phases.py $(ls /tmp/kt2/*/*kernel_trace.csv | head -1) $w; echo; done > $R/gpurun_out/r4j_venice_trace.txt
cd $R; grep -h "bal_pair\|cam_diag\|precompute\|point_block" gpurun_out/r4j_lady_trace.txt gpurun_out/r4j_venice_trace.txt | head -16
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule or properties" 2>&1 | tail -n 2
