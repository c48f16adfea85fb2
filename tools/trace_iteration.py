#!/usr/bin/env python3
"""Developer tool: EVERY kernel of one LM iteration from a rocprofv3 --kernel-trace CSV, the chain's per-column launches (column
launches, thin SYRKs) collapsed into one line per run: start (us from the iteration's first kernel), duration, gap to the end of
everything before it, stream.
  python tools/trace_iteration.py <kernel_trace.csv> [iteration, default -3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('sk::', '').replace('void ', '')
rows.sort(key=lambda r: r['s'])
marks = [i for i, r in enumerate(rows) if r['n'] == 'bal_cam_step_kernel']  # one per LM iteration: the start of phase D
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
a, b = marks[which], marks[which + 1]
t0 = rows[a]['s']
prev_end = t0
collapse = ('chain_column_kernel', 'chain_column_pair_kernel', 'syrk_trailing_thin_f64_kernel', 'syrk_trailing_thin_pair_f64_kernel')
run = None


def flush():
    global run
    if run:
        print("%9.1f  %7.1f us  %-44s x%d (chain launches, first start to last end)" % ((run[0] - t0) / 1e3, (run[1] - run[0]) / 1e3, run[2], run[3]))
    run = None


for r in rows[a:b + 1]:
    if r['n'] in collapse:
        if run:
            run[1] = max(run[1], r['e']); run[3] += 1
        else:
            run = [r['s'], r['e'], 'column launches + thin SYRKs', 1]
        prev_end = max(prev_end, r['e'])
        continue
    flush()
    print("%9.1f  %7.1f us  gap %6.1f  %-34s stream %s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, (r['s'] - prev_end) / 1e3, r['n'], r['Stream_Id']))
    prev_end = max(prev_end, r['e'])
flush()
