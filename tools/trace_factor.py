#!/usr/bin/env python3
"""Developer tool: summarise the blocked-Cholesky part of a rocprofv3 --kernel-trace CSV (last LM iteration):
per-kernel totals, the SYRK's union time, and a slice of the timeline."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (60, 100)
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('sk::', '').replace('void ', '')
rows.sort(key=lambda r: r['s'])
pi = [i for i, r in enumerate(rows) if r['n'] == 'bal_pair_kernel'][-1]
it = rows[pi:]
end = [i for i, r in enumerate(it) if r['n'] in ('copy_row_kernel', 'bs_resident_kernel')][0]  # (the back-substitution: what follows the factorisation)
names = ('potrf128_kernel', 'trsm_gemm_f64_kernel', 'trsm_gemm_thin_f64_kernel', 'gemm_update_f64_kernel', 'gemm_update_thin_f64_kernel',
         'gemm_diag_f64_kernel', 'syrk_trailing_f64_kernel', 'syrk_trailing_thin_f64_kernel', 'potrf_server_kernel', 'chain_column_kernel',
         'chain_marker_kernel', 'crit_server_kernel', 'chain_column_pair_kernel', 'syrk_trailing_thin_pair_f64_kernel', 'potrf_server_pair_kernel')
ch = [r for r in it[1:end] if r['n'] in names]
t0 = ch[0]['s']; t1 = max(r['e'] for r in ch)
print("factor wall %.2f ms, kernels %d" % ((t1 - t0) / 1e6, len(ch)))
by = collections.defaultdict(lambda: [0, 0])
for r in ch:
    key = (r['n'], r['Stream_Id'])
    by[key][0] += r['e'] - r['s']; by[key][1] += 1
for k, v in sorted(by.items()):
    print("  %-28s stream %s n=%4d total %6.2f ms avg %6.1f us" % (k[0], k[1], v[1], v[0] / 1e6, v[0] / 1e3 / v[1]))


def union(iv):
    iv = sorted(iv); tot = 0; cs = ce = None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot


sy = [(r['s'], r['e']) for r in ch if r['n'].startswith('syrk_trailing')]
print("union busy %.2f ms; syrk union %.2f ms; syrk gaps: first start +%.2f ms, last end -%.2f ms" % (
    union([(r['s'], r['e']) for r in ch]) / 1e6, union(sy) / 1e6, (sy[0][0] - t0) / 1e6, (t1 - sy[-1][1]) / 1e6))
gaps = [(sy[i + 1][0] - sy[i][1]) / 1e3 for i in range(len(sy) - 1)]
print("gaps between consecutive SYRKs (us):", " ".join("%.0f" % g for g in gaps))
for r in ch[lo:hi]:
    print("%9.1f %8.1f %-26s st%s grid %s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r['n'], r['Stream_Id'], r['Grid_Size_X']))
