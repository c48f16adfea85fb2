R=$PWD
python -m pytest tests/test_jni_mock.py tests/test_traced_functors.py -x -q -m gpu > gpurun_out/r4f_tests.log 2>&1; tail -5 gpurun_out/r4f_tests.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "venice_1778_at_full_size or without_resident or timeout_is_reported or schedule_of" > gpurun_out/r4f_tests2.log 2>&1; tail -5 gpurun_out/r4f_tests2.log
python tools/chain_timeline.py ladybug-1723-156502 off > gpurun_out/r4f_chain_ladybug.txt 2>&1; tail -n 2 gpurun_out/r4f_chain_ladybug.txt
python tools/chain_timeline.py band:1200,100000,450000,9 off > gpurun_out/r4f_chain_band1200.txt 2>&1; tail -n 2 gpurun_out/r4f_chain_band1200.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/tools/phase_times.py ladybug 6 > $R/gpurun_out/r4f_lady.txt 2>&1; for w in -4 -3; do python3 $R/tools/trace_point_phases.py $(ls /tmp/kt/*/*kernel_trace.csv | head -1) $w; echo; done > $R/gpurun_out/r4f_lady_trace.txt
cd $R; for i in 1 2 3; do python tools/phase_times.py ladybug 20 2>&1 | tail -n 1; done
