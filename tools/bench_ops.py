#!/usr/bin/env python3
"""Micro-benchmarks of single libsr_amd ops at the bench workload's shapes (for rocprofv3 / tuning).

    python tools/bench_ops.py lowpass [C H W B reps]     python tools/bench_ops.py conv [cin cout H W B reps]
"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from super_resolution_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device('cuda:0')
st = L.stream_ptr(dev)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def lowpass(Cc=32, H=512, W=512, B=16, reps=10, blocked=0, force_fft=1, ups=0):
    lib.sr_debug_force_fft_lowpass(force_fft)
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(H, W, C.byref(plan)))
    shape = (B, Cc // blocked, H, W, blocked) if blocked else (B, H, W, Cc)
    xshape = (B, Cc // blocked, H >> ups, W >> ups, blocked) if blocked else (B, H >> ups, W >> ups, Cc)
    mk = L.blocked_view if blocked else L.view
    x = torch.rand(xshape, device=dev).to(torch.bfloat16)
    bands = [torch.empty(shape, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    pooled = torch.empty(B, Cc, device=dev)
    ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device=dev)

    checksum = []
    for no_sc in ((0, 1) if ups else (0,)):  # A/B of the upsample shortcut inside one process
        lib.sr_debug_no_ups_shortcut(no_sc)
        run_once = lambda: L.check(lib.sr_fbm_lowpass(plan, 1, B, mk(x), ups, mk(bands[0]), mk(bands[1]), mk(bands[2]), pooled.data_ptr(),
                                                      Cc, ws.data_ptr(), ws.numel(), st))  # noqa: E731
        tt = timeit(run_once, reps)
        checksum.append([float(b.float().abs().mean()) for b in bands] + [float(pooled.abs().mean())])
        if ups:
            print('  ups shortcut %s: %.1f us  checksums %s' % ('off' if no_sc else 'on', tt * 1e6, checksum[-1]))
    lib.sr_debug_no_ups_shortcut(0)

    def run():
        L.check(lib.sr_fbm_lowpass(plan, 1, B, mk(x), ups, mk(bands[0]), mk(bands[1]), mk(bands[2]), pooled.data_ptr(), Cc,
                                   ws.data_ptr(), ws.numel(), st))
    t = timeit(run, reps)
    px = B * H * W * Cc
    lib.sr_debug_force_fft_lowpass(-1)
    print('lowpass C=%d %dx%d B=%d blocked=%d fft=%d ups=%d: %.1f us  (%.2f B/px/ch algorithmic 8 -> %.2f TB/s)' %
          (Cc, H, W, B, blocked, force_fft, ups, t * 1e6, 8.0, px * 8 / t / 1e12))


def conv(cin=64, cout=32, H=512, W=512, B=16, reps=10, bands=1, blocked=0, split=0, variant=0):
    segs = [64] + [32] * ((cin - 64) // 32)
    shp = (lambda c: (B, c // blocked, H, W, blocked)) if blocked else (lambda c: (B, H, W, c))
    mk = L.blocked_view if blocked else L.view
    xs = [torch.rand(shp(c), device=dev).to(torch.bfloat16) for c in segs]
    bs = [[torch.rand(shp(c), device=dev).to(torch.bfloat16) for _ in range(3)] for c in segs]
    gates = torch.rand(B, H, W, 4, device=dev) * 2
    wd = (torch.randn(2, B, 9, cout, cin, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(shp(cout), device=dev, dtype=torch.bfloat16)
    cc = L.Concat()
    cc.nseg, cc.bands = len(segs), bands
    for i in range(len(segs)):
        cc.x[i] = mk(xs[i])
        cc.l2[i], cc.l4[i], cc.l8[i] = [mk(t) for t in bs[i]]
    ep = L.Epilogue()
    ep.lrelu = 1

    def run():
        L.check(lib.sr_conv3x3(1, B, H, W, C.byref(cc), gates.data_ptr(), wd.data_ptr(), 9 * cout * cin,
                               B * 9 * cout * cin if split else 0, cout, C.byref(ep), mk(out), st))
    t = timeit(run, reps)
    px = B * H * W
    byts = px * ((4 if bands else 1) * cin * 2 + (16 if bands else 0) + cout * 2)
    print('conv %d->%d %dx%d B=%d bands=%d blocked=%d split=%d variant=%d: %.1f us  %.1f TFLOP/s  %.2f TB/s algorithmic' %
          (cin, cout, H, W, B, bands, blocked, split, variant, t * 1e6, 2.0 * 9 * cin * cout * px / t / 1e12, byts / t / 1e12))


def gates(cin=160, H=512, W=512, B=16, reps=10, blocked=0):
    segs = [64] + [32] * ((cin - 64) // 32)
    shp = (lambda c: (B, c // blocked, H, W, blocked)) if blocked else (lambda c: (B, H, W, c))
    mk = L.blocked_view if blocked else L.view
    xs = [torch.rand(shp(c), device=dev).to(torch.bfloat16) for c in segs]
    cc = L.Concat()
    cc.nseg = len(segs)
    for i in range(len(segs)):
        cc.x[i] = mk(xs[i])
    lp = L.LayerParams()
    lp.cin, lp.cout, lp.active = cin, 32, 1
    gw = [torch.randn(cin, 9, device=dev) * 0.05 for _ in range(3)]
    gb = [torch.zeros(1, device=dev) for _ in range(3)]
    for k in range(3):
        lp.gate_w[k], lp.gate_b[k] = gw[k].data_ptr(), gb[k].data_ptr()
    out = torch.empty(B, H, W, 4, device=dev)

    def run():
        L.check(lib.sr_fbm_gates(1, B, H, W, C.byref(cc), C.byref(lp), out.data_ptr(), st))
    t = timeit(run, reps)
    px = B * H * W
    print('gates cin=%d %dx%d B=%d blocked=%d: %.1f us  %.2f TB/s algorithmic' % (cin, H, W, B, blocked, t * 1e6, px * (cin * 2 + 16) / t / 1e12))


def all_ops():
    for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
        conv(cin, cout)
    conv(192, 64, bands=0)
    for cin, cout in ((64, 32), (160, 32), (192, 64)):
        conv(cin, cout, blocked=32)
    for cin in (64, 128, 192):
        gates(cin)
    gates(64, blocked=32)
    gates(192, blocked=32)
    lowpass(32)
    lowpass(64, blocked=32)
    lowpass(64)
    lowpass(64, 2048, 2048, 1, 5, 8)


if __name__ == '__main__':
    op = sys.argv[1]
    args = [int(a) for a in sys.argv[2:]]
    def conv_ab():
        for cin, cout in ((192, 64), (160, 32), (64, 32)):
            for split in (1, 0, 1):
                conv(cin, cout, 512, 512, 16, 10, 1, 32, split, 0)
    {'lowpass': lowpass, 'conv': conv, 'gates': gates, 'all': all_ops, 'conv_ab': conv_ab}[op](*args)
