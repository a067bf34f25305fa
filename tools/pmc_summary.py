#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel.  usage: pmc_summary.py DIR [substr]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0][-70:]
    if sub not in n:
        continue
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[n].add(r['Dispatch_Id'])
for n, dd in agg.items():
    print(n, 'dispatches=%d' % len(cnt[n]), {k: '%.4g' % (v / len(cnt[n])) for k, v in dd.items()}, '(per dispatch)')
