#!/usr/bin/env python3
"""Developer tool: the factorisation of the last LM iteration in a rocprofv3 --kernel-trace CSV, stream by stream:
when each stream starts and ends, how busy it is, its kernels' counts / average durations / average gaps.

  python tools/trace_streams.py <kernel_trace.csv> [first_line last_line]   (optional: a slice of the merged timeline)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('sk::', '').replace('void ', '')
rows.sort(key=lambda r: r['s'])
pi = [i for i, r in enumerate(rows) if r['n'] in ('bal_pair_kernel', 'bal_pair_long_kernel')][-1]
it = rows[pi + 1:]
end = [i for i, r in enumerate(it) if r['n'] == 'bal_gather_y_kernel' or r['n'] == 'bal_cam_step_kernel'][0]
ch = it[:end]
t0 = ch[0]['s']
print("window %.2f ms, %d kernels" % ((max(r['e'] for r in ch) - t0) / 1e6, len(ch)))
by = collections.defaultdict(list)
for r in ch:
    by[r['Stream_Id']].append(r)
for st, rs in sorted(by.items(), key=lambda kv: kv[1][0]['s']):
    busy = sum(r['e'] - r['s'] for r in rs)
    print("stream %3s: %4d kernels, first start %8.1f us, last end %8.1f us, busy %7.2f ms" % (st, len(rs), (rs[0]['s'] - t0) / 1e3, (max(r['e'] for r in rs) - t0) / 1e3, busy / 1e6))
    names = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        names[r['n']][0] += r['e'] - r['s']; names[r['n']][1] += 1
    for n, (tot, cnt) in sorted(names.items(), key=lambda kv: -kv[1][0]):
        print("      %-30s n=%4d avg %7.1f us" % (n, cnt, tot / 1e3 / cnt))
if len(sys.argv) > 3:
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    for r in ch[lo:hi]:
        print("%9.1f %8.1f %-28s st%s grid %s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r['n'], r['Stream_Id'], r['Grid_Size_X']))
