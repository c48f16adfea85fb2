for i in 1 2 3; do python tools/phase_times.py ladybug 20 2>&1 | tail -n 1; done
for i in 1 2; do python tools/phase_times.py ladybug 20 dissection=off 2>&1 | tail -n 1; done
for i in 1 2; do SK_SCHEDULE_PLAIN=1 python tools/phase_times.py ladybug 20 dissection=off 2>&1 | tail -n 1; done
python tools/phase_times.py venice 8 2>&1 | tail -n 1
SK_SCHEDULE_PLAIN=1 python tools/phase_times.py venice 8 2>&1 | tail -n 1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "properties or schedule or bitwise or full_size_ladybug or default_plan" 2>&1 | tail -n 3
