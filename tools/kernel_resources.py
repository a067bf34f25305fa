#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of every HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

usage: kernel_resources.py [file.hip ...]   (default: all of super_resolution_amd/csrc)
A non-zero ScratchSize on a hot-path kernel means private arrays or spills live in scratch memory: its loads and stores
queue in the same in-order memory pipe as the kernel's global traffic (findings 3 and 4), and a private
array indexed out of bounds only faults once it lives there (the round-1 33-point fault).  Run after kernel edits.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'super_resolution_amd', 'csrc')


def demangle(names):
    import shutil
    tool = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
    if not tool or not names:
        return names
    out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    for f in files:
        r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-fno-signed-zeros', '-fPIC', '-c', f,
                            '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
        recs, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|Dynamic Stack|VGPRs Spill): (\S+)', line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == 'Function Name':
                cur = {'name': v}
                recs.append(cur)
            elif cur is not None:
                cur['Spill' if k == 'VGPRs Spill' else k.split(' ')[0]] = v
        names = demangle([c['name'] for c in recs])
        print('== %s' % os.path.basename(f))
        for c, n in zip(recs, names):
            n = re.sub(r'\(.*', '', n)
            flag = '  <-- SCRATCH' if c.get('ScratchSize', '0') != '0' or c.get('Dynamic', 'False') == 'True' else ''
            print('  %-70s vgpr %3s agpr %3s sgpr %3s spilled %3s scratch %4s B/lane dynstack %-5s occ %s%s' % (
                n[-70:], c.get('VGPRs'), c.get('AGPRs'), c.get('TotalSGPRs'), c.get('Spill'), c.get('ScratchSize'), c.get('Dynamic'), c.get('Occupancy'), flag))


if __name__ == '__main__':
    main()
