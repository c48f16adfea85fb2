#!/usr/bin/env python3
"""Developer tool: average duration of the trailing-SYRK launches of the bench's TIMED region in a rocprofv3
--kernel-trace CSV, to set beside `roofline.avg_launch_ms` of the bench line of the same run.

  rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps K --warmup W --cpu-iters 0 --no-alone --no-c5
  python tools/trace_syrk_average.py <..._kernel_trace.csv> <bench line .json>

With --no-alone and --no-c5 the timed steps are the last thing the process does on the GPU, so the timed region's SYRK
launches are the last `roofline.launches` SYRK launches of the trace (the queue trial at set-up, iteration 0 and the
warm-up steps come before them and are what skews the per-kernel averages of the --stats table)."""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "syrk_trailing" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_record
line = bench_record.load(sys.argv[2])  # (the run's detailed record: bench.py's side file, SK_BENCH_DETAILS)
n = int(line["roofline"]["launches"])
timed = rows[-n:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
print("SYRK launches in the trace: %d; timed region: the last %d" % (len(rows), n))
print("kernel trace: average %.2f us per launch (sum %.3f ms over %d steps)" % (sum(dur) / len(dur) / 1e3, sum(dur) / 1e6, line["steps"]))
print("bench line (HIP events, same run): avg_launch_ms %.5f = %.2f us; achieved %.2f TFLOP/s, frac %.3f" % (
    line["roofline"]["avg_launch_ms"], 1e3 * line["roofline"]["avg_launch_ms"], line["roofline"]["achieved"], line["roofline"]["frac"]))
flops = line["roofline"]["flops_per_solve"] * line["steps"]
print("from the trace: %.2f TFLOP/s = %.3f of 78.6" % (flops / (sum(dur) * 1e-9) * 1e-12, flops / (sum(dur) * 1e-9) * 1e-12 / 78.6))
