#!/usr/bin/env python
"""Costing experiment for round-2 verdict item 3 (CPU only, uses the oracle as a numerical model of the bf16 path).

Proposal: L4 and L8 are band-limited to 1/4 and 1/8 of Nyquist, so the bf16 path could store L4 at (H/2, W/2) and L8 at
(H/2, W/2) or (H/4, W/4) and rebuild the full-resolution planes in the conv loader with a short polyphase
interpolation; the conv operand would shrink from 4 to 2.5 / 2.31 planes per input channel.

What this script measures is the NUMERICAL side: the 23-block owner-script network (tests' realistic_net, seed 2) is
evaluated with the oracle's bf16-storage emulation (hi + lo weights, as the kernels do), once with full-resolution
planes and once with decimated planes that are (i) pre-compensated in the frequency domain for the interpolation
kernel's pass-band droop (free in the DFT-as-GEMM synthesis: it only rescales basis functions), (ii) rounded to bf16
(what would be stored) and (iii) interpolated back by zero-stuffing + a separable FIR kernel, circularly.  Reported:
direct PSNR against the float32 oracle and north_star's PSNR-delta figure, per interpolation kernel.

    python tools/cost_halfres_planes.py [--size 64] [--blocks 23]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_params as gp  # noqa: E402
import make_golden_r3 as mg3  # noqa: E402
import oracle.fdconv_ref as fr  # noqa: E402
from oracle.rrdbnet_ref import rrdbnet_forward  # noqa: E402

KERNELS = {  # full-resolution taps of a 2x interpolator (even taps and odd taps each sum to 1)
    'linear (3 taps)': [1, 2, 1],
    'binomial-5': [1, 4, 6, 4, 1],
    'binomial-7': [1, 6, 15, 20, 15, 6, 1],
    'binomial-9': [1, 8, 28, 56, 70, 56, 28, 8, 1],
}


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 20 * math.log10(1.0 / math.sqrt(mse))


def response(taps, n):
    """Frequency response (real, the kernel is symmetric) of the full-resolution FIR on an n-point circular grid."""
    t = np.asarray(taps, dtype=np.float64)
    t = t / t.sum() * 2.0
    h = np.zeros(n)
    c = len(t) // 2
    for i, v in enumerate(t):
        h[(i - c) % n] += v
    return torch.from_numpy(np.real(np.fft.fft(h)))


def decimate_interp(L, taps, times):
    """L (..., H, W) band-limited -> decimate by 2 `times` times (pre-compensated), bf16 storage, interpolate back."""
    if times == 0:
        return L
    H, W = L.shape[-2:]
    rh, rw = response(taps, H).to(L.dtype), response(taps, W).to(L.dtype)
    # pre-compensation: divide the spectrum by the kernel's response (the plane's own band only; its images are removed
    # by the decimation + interpolation).  On the band the response is >= 0.3 for every kernel here.
    X = torch.fft.fft2(L)
    X = X / (rh.view(-1, 1) * rw.view(1, -1) / 4.0).clamp_min(1e-3)
    pre = torch.fft.ifft2(X).real
    small = decimate_interp(pre[..., ::2, ::2], taps, times - 1) if times > 1 else fr._bf16(pre[..., ::2, ::2])
    up = torch.zeros_like(L)
    up[..., ::2, ::2] = small
    U = torch.fft.fft2(up) * (rh.view(-1, 1) * rw.view(1, -1))
    return torch.fft.ifft2(U).real


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--blocks', type=int, default=23)
    a = ap.parse_args()
    torch.set_num_threads(8)
    sd = mg3.realistic_state((1, 1, 4, 64, a.blocks, 32), seed=2)
    x = torch.from_numpy(gp.image_input((1, 1, a.size, a.size), 83))
    real_bf16, real_lowpass = fr._bf16, fr.fbm_lowpass

    def hilo(v):
        if v.dim() == 5 or (v.dim() == 4 and v.shape[-1] == 3):
            hi = real_bf16(v)
            return hi + real_bf16(v - hi)
        return real_bf16(v)

    def run(taps=None, l4_times=0, l8_times=0):
        def lowpass(xx, dtype=None):
            L = real_lowpass(xx, dtype)
            if taps is None:
                return L
            return torch.stack([L[0], decimate_interp(L[1], taps, l4_times), decimate_interp(L[2], taps, l8_times)])
        fr._bf16, fr.fbm_lowpass = hilo, lowpass
        try:
            with torch.no_grad():
                return rrdbnet_forward(sd, x, emulate_bf16=True)
        finally:
            fr._bf16, fr.fbm_lowpass = real_bf16, real_lowpass

    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
    lo, hi = ref.min(), ref.max()
    n = lambda v: (v - lo) / (hi - lo)  # noqa: E731
    target = torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
    p_ref = psnr(n(ref).clamp(0, 1), target)
    base = run()
    print('%d blocks, %dx%d: fp32 oracle vs target %.3f dB; bf16 storage, full-resolution planes: direct %.1f dB, PSNR delta %.4f dB'
          % (a.blocks, a.size, a.size, p_ref, psnr(n(base), n(ref)), abs(psnr(n(base).clamp(0, 1), target) - p_ref)), flush=True)
    for name, taps in KERNELS.items():
        for l4t, l8t, label in ((1, 1, 'L4, L8 at 1/2 (2.5 planes)'), (1, 2, 'L4 at 1/2, L8 at 1/4 (2.31 planes)')):
            y = run(taps, l4t, l8t)
            print('  %-16s %-36s direct %.1f dB, PSNR delta %.4f dB' %
                  (name, label, psnr(n(y), n(ref)), abs(psnr(n(y).clamp(0, 1), target) - p_ref)), flush=True)


if __name__ == '__main__':
    main()
