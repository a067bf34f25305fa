#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (CSV or rocpd sqlite): per (kernel, grid) count / average / total.

usage: prof_summary.py DIR [N] [--seq SUBSTR STEPS]
  --seq SUBSTR STEPS : also print, for kernels whose name contains SUBSTR, the average duration per position inside a
                       step (the launch sequence is split into STEPS equal parts, the last STEPS parts of the run).
"""
import collections
import csv
import glob
import sqlite3
import sys


def load(d):
    """-> list of (name, grid_x, grid_y, start_ns, end_ns) in launch order"""
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if f:
        rows = [(r['Kernel_Name'], r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''), int(r['Start_Timestamp']),
                 int(r['End_Timestamp'])) for r in csv.DictReader(open(f[0]))]
    else:
        db = glob.glob(d + '/**/*.db', recursive=True)[0]
        c = sqlite3.connect(db)
        rows = list(c.execute('select name, grid_x, grid_y, start, end from kernels'))
    return sorted(rows, key=lambda r: r[3])


def short(n):
    return n.split('(')[0][-58:]


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 40
    rows = load(d)
    agg = collections.defaultdict(list)
    for n, gx, gy, s, e in rows:
        agg[(short(n), gx, gy)].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    print('total kernel time %.1f ms' % (tot / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print('%-60s grid=%s,%s n=%5d avg=%10.1f us total=%9.1f ms' % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
    if '--seq' in sys.argv:
        i = sys.argv.index('--seq')
        sub, steps = sys.argv[i + 1], int(sys.argv[i + 2])
        sel = [(short(n), e - s) for n, gx, gy, s, e in rows if sub in n]
        per = len(sel) // (steps + 1) if len(sel) % steps else len(sel) // steps
        # use the last `steps` repetitions (earlier ones are warm-up)
        sel = sel[len(sel) - per * steps:]
        print('--- %s: %d launches per step' % (sub, per))
        for p in range(per):
            ds = [sel[s * per + p][1] for s in range(steps)]
            print('%4d %-50s %9.1f us' % (p, sel[p][0][-50:], sum(ds) / len(ds) / 1e3))


if __name__ == '__main__':
    main()
