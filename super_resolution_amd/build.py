"""Builds libsr_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m super_resolution_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['layout.hip', 'fbm.hip', 'fbm_fast.hip', 'fbm_fast_mixed.hip', 'fbm_mfma.hip', 'gates.hip', 'weightgen.hip', 'weightgen_bwd.hip', 'conv.hip', 'conv_plain.hip', 'net.hip', 'head.hip', 'train.hip']
OUT = os.path.join(HERE, 'libsr_amd.so')


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'sr_amd.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(HERE, '..', 'include', 'sr_amd.h')]
    newest_header = max(os.path.getmtime(h) for h in headers)
    for s in SOURCES:
        o = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(o)
        src = os.path.join(CSRC, s)
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(src), newest_header):
            continue  # object is newer than its source and every header
        cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-fno-signed-zeros', '-fPIC', '-Wall', '-Wno-unused-result', '-c',
               os.path.join(CSRC, s), '-o', o]
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT]
    subprocess.check_call(cmd)
    if verbose:
        print('built', OUT)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
