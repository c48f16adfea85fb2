#!/bin/bash
# The rocprofv3 evidence of a round, on the code as it stands: run on the GPU box from the repo root,
#   bash tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1
# writes under gpurun_out/ (copied to profiles/ afterwards): <tag>_bench.json (plain run), <tag>_bench_under_rocprof.json +
# <tag>_kernel_stats.csv + <tag>_syrk_timed_region.txt (one --kernel-trace --stats run), <tag>_pmc_traffic.json (separate FETCH_SIZE /
# WRITE_SIZE passes) and <tag>_pmc_mfma.json (MFMA counters).  Counter passes run with --no-resident-kernels
# (sk_options_set_resident_kernels(o, 0)): counter collection serialises the kernels of a process, and a resident kernel that
# waits for another would time out — the same plan of the undissected system, every block column and every step of the
# back-substitution a launch of its own.
set -e
TAG=${1:-r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SK_BENCH_DETAILS=$OUT/${TAG}_bench_details.json python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"
export SK_BENCH_DETAILS=$OUT/${TAG}_bench_under_rocprof_details.json
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-alone --no-c5 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
cp $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
python3 $ROOT/tools/trace_syrk_average.py $(ls /tmp/prof_kt/*/*kernel_trace.csv | head -1) $OUT/${TAG}_bench_under_rocprof_details.json > $OUT/${TAG}_syrk_timed_region.txt
python3 $ROOT/tools/trace_factor.py $(ls /tmp/prof_kt/*/*kernel_trace.csv | head -1) 0 700 > $OUT/${TAG}_factor_timeline.txt 2>&1 || true
echo "kernel trace done"
rm -rf /tmp/pmc_fetch /tmp/pmc_write /tmp/pmc_mfma
export SK_BENCH_DETAILS=$OUT/${TAG}_bench_under_pmc_fetch_details.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-alone --no-c5 --no-resident-kernels > $OUT/${TAG}_bench_under_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err
export SK_BENCH_DETAILS=/tmp/pmc_write_details.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-alone --no-c5 --no-resident-kernels > /dev/null 2> $OUT/${TAG}_pmc_write.err
python3 $ROOT/tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_bench_under_pmc_fetch_details.json
echo "traffic passes done"
export SK_BENCH_DETAILS=/tmp/pmc_mfma_details.json
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-alone --no-c5 --no-resident-kernels > $OUT/${TAG}_bench_under_pmc.json 2> $OUT/${TAG}_pmc_mfma.err
python3 $ROOT/tools/pmc_mfma_summary.py /tmp/pmc_mfma $OUT/${TAG}_pmc_mfma.json
echo "PROFILE_ROUND_OK"
