#!/bin/bash
# Worlds of five and six ranks sharing ONE GPU with a real exchange over gloo (tests/dist_gpu_worker2.py,
# tests/dist_dense_rows_worker.py) — beyond what pytest can start on a GPU box (six processes per card; worlds of up to
# four are in tests/test_gpu_parity.py).  Eight ranks on one card are not possible there at all; the eight-segment
# arithmetic is covered at the matrix level (test_multiway_dissected_factorisation_vs_numpy).
#   bash tools/rehearse_worlds.sh > profiles/r03_worlds_5_6.txt
set -e
cd "$(dirname "$0")/.."
for w in 5 6; do
  python tools/launch_ranks.py $w tests/dist_gpu_worker2.py segmented 900,30000,70000,8
  python tools/launch_ranks.py $w tests/dist_gpu_worker2.py segmented 400,12000,27000,3 $w
  python tools/launch_ranks.py $w tests/dist_gpu_worker2.py sharded 400,12000,60000,3
  python tools/launch_ranks.py $w tests/dist_dense_rows_worker.py 5000,300
done
echo REHEARSAL_OK
