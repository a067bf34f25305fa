#!/usr/bin/env python
"""Numerical costing (CPU, oracle as the model of the bf16 path): what if the conv's matrix-core operands were fp16
instead of bf16 -- weights as ONE fp16 value instead of a bf16 hi + lo pair (half the MFMA instructions, half the
weight-slab bytes and B-fragment reads), the mixed activation operand packed to fp16 by the loader?  The tensors the
network stores between layers and the low-pass planes stay bf16 as today.

    python tools/cost_fp16_operands.py [--size 64] [--blocks 23]
"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_params as gp  # noqa: E402
import make_golden_r3 as mg3  # noqa: E402
import oracle.fdconv_ref as fr  # noqa: E402
from oracle.rrdbnet_ref import rrdbnet_forward  # noqa: E402


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 20 * math.log10(1.0 / math.sqrt(mse))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--blocks', type=int, default=23)
    ap.add_argument('--seed', type=int, default=2)
    a = ap.parse_args()
    torch.set_num_threads(8)
    sd = mg3.realistic_state((1, 1, 4, 64, a.blocks, 32), seed=a.seed)
    x = torch.from_numpy(gp.image_input((1, 1, a.size, a.size), 83))
    real = fr._bf16
    bf = lambda v: v.to(torch.bfloat16).to(v.dtype)  # noqa: E731
    hf = lambda v: v.to(torch.float16).to(v.dtype)   # noqa: E731

    def is_weight(v):
        return v.dim() == 5 or (v.dim() == 4 and v.shape[-1] == 3 and v.shape[-2] == 3)

    variants = {
        'bf16 hi + lo weights, bf16 operand (today)': lambda v: (bf(v) + bf(v - bf(v))) if is_weight(v) else bf(v),
        'one bf16 weight, bf16 operand (round 1)': bf,
        'one fp16 weight, bf16 operand': lambda v: hf(v) if is_weight(v) else bf(v),
        'one fp16 weight, fp16 operand': hf,
    }
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
    lo, hi = ref.min(), ref.max()
    n = lambda v: (v - lo) / (hi - lo)  # noqa: E731
    target = torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
    p_ref = psnr(n(ref).clamp(0, 1), target)
    print('%d blocks, %dx%d, seed %d: fp32 oracle vs target %.3f dB' % (a.blocks, a.size, a.size, a.seed, p_ref), flush=True)
    for name, fn in variants.items():
        fr._bf16 = fn
        try:
            with torch.no_grad():
                y = rrdbnet_forward(sd, x, emulate_bf16=True)
        finally:
            fr._bf16 = real
        print('  %-46s direct %.1f dB, PSNR delta %.4f dB' % (name, psnr(n(y), n(ref)), abs(psnr(n(y).clamp(0, 1), target) - p_ref)), flush=True)


if __name__ == '__main__':
    main()
