#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / average / total.  usage: prof_summary.py DIR [N]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0][-58:]
    agg[(n, r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print('total kernel time %.1f ms' % (tot / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print('%-60s grid=%s,%s n=%5d avg=%10.1f us total=%9.1f ms' % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
