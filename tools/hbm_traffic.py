#!/usr/bin/env python3
"""HBM traffic per launch and per kernel group from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

usage: hbm_traffic.py FETCH_DIR WRITE_DIR OUT.json
Counters are in KiB; on gfx950 FETCH_SIZE under-reports by 2x (MI355X_MICROARCH.md, HBM section), so it is doubled.
"""
import collections
import csv
import glob
import json
import sys

GROUPS = (('conv', ('conv3x3',)), ('lowpass', ('rows_fwd', 'rows_inv', 'cols', 'fft', 'mf_rows', 'mf_cols', 'mf_analysis', 'mf_synthesis')),
          ('gates', ('gates',)), ('weightgen', ('weightgen',)))


def group(name):
    for g, subs in GROUPS:
        if any(s in name for s in subs):
            return g
    return None


def collect(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    tot = collections.defaultdict(float)
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        g = group(r['Kernel_Name'])
        if g is None:
            continue
        tot[g] += float(r['Counter_Value'])
        ids[g].add(r['Dispatch_Id'])
    return {g: (tot[g] / len(ids[g]), len(ids[g])) for g in tot}


def main():
    fetch = collect(sys.argv[1], 'FETCH_SIZE')
    write = collect(sys.argv[2], 'WRITE_SIZE')
    out = {'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 '
                      '--no-cpu-baseline',
           'correction': 'gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); counters in KiB',
           'per_launch': {}}
    for g in fetch:
        f, n = fetch[g]
        w = write.get(g, (0.0, 0))[0]
        out['per_launch'][g] = {'launches': n, 'fetch_kb_raw_per_launch': f, 'write_kb_per_launch': w,
                                'hbm_gb_per_launch_corrected': (2 * f + w) * 1024 / 1e9}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out['per_launch'], indent=1))


if __name__ == '__main__':
    main()
