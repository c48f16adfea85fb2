#!/usr/bin/env python3
"""Developer tool: the kernels of one LM iteration OUTSIDE the factorisation (phases A, B, D) from a rocprofv3 --kernel-trace CSV:
start (us from the end of the previous iteration's last kernel), duration, gap to the kernel before.
  python tools/trace_point_phases.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('sk::', '').replace('void ', '')
rows.sort(key=lambda r: r['s'])
skip = ('potrf128_kernel', 'trsm_gemm_f64_kernel', 'trsm_gemm_thin_f64_kernel', 'gemm_update_f64_kernel', 'gemm_update_thin_f64_kernel', 'gemm_diag_f64_kernel',
        'syrk_trailing_f64_kernel', 'syrk_trailing_thin_f64_kernel', 'potrf_server_kernel', 'chain_column_kernel', 'chain_marker_kernel',
        'chain_column_pair_kernel', 'syrk_trailing_thin_pair_f64_kernel', 'potrf_server_pair_kernel', 'border_add_kernel')
skip = skip + ('bs_resident_kernel', 'tri_pack_kernel')
marks = [i for i, r in enumerate(rows) if r['n'] == 'bal_cam_step_kernel']  # one per LM iteration: the start of phase D
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
a, b = marks[which], marks[which + 1]  # from one camera step to the next: D, A, B (and the factorisation, skipped)
prev_end = rows[a]['s']
t0 = rows[a]['s']
for r in rows[a:b + 1]:
    if r['n'] in skip:
        continue
    print("%9.1f  %7.1f us  gap %6.1f  %-34s stream %s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, (r['s'] - prev_end) / 1e3, r['n'], r['Stream_Id']))
    prev_end = max(prev_end, r['e'])
