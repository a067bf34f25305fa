#!/usr/bin/env python
"""PSNR delta of the two operand formats of the FDConv convs (bf16 hi + lo weight pairs; fp16 weight + remainder plane, the default)
over several (weights, input) pairs:
23-block owner-script networks (seeds), 64 x 64 and 96 x 96 inputs, fp32 oracle on the CPU as the reference (--ref gpu32: the fp32
HIP path instead, for 256 x 256 / 512 x 512 inputs).

    python tools/diag_f16_pairs.py [--seeds 1,2,3,4,5,6] [--sizes 64,96]
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
from oracle.rrdbnet_ref import rrdbnet_forward  # noqa: E402  (diagnostic tool: the oracle is the yardstick, as in tests/)


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 20 * math.log10(1.0 / math.sqrt(mse))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', default='1,2,3,4,5,6')
    ap.add_argument('--sizes', default='64,96')
    ap.add_argument('--ref', choices=('oracle', 'gpu32'), default='oracle', help='reference: the fp32 CPU oracle, or the fp32 HIP path (parity-checked against the oracle to 1e-4 in tests/) for sizes the oracle needs minutes for')
    a = ap.parse_args()
    import super_resolution_amd as sra
    from super_resolution_amd import _lib
    lib = _lib.lib()
    torch.set_num_threads(16)
    for seed in [int(v) for v in a.seeds.split(',')]:
        sd = mg3.realistic_state((1, 1, 4, 64, 23, 32), seed=seed)
        net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, 23, 32))
        net.load_state_dict(sd, strict=True)
        net = net.eval().cuda()
        for size in [int(v) for v in a.sizes.split(',')]:
            x = torch.from_numpy(gp.image_input((1, 1, size, size), 83))
            with torch.no_grad():
                ref = rrdbnet_forward(sd, x) if a.ref == 'oracle' else net(x.cuda()).cpu()
                lib.sr_debug_operand_f16(0)
                try:
                    y_def = net(x.cuda().to(torch.bfloat16)).float().cpu()
                finally:
                    lib.sr_debug_operand_f16(1)
                y_f16 = net(x.cuda().to(torch.bfloat16)).float().cpu()
            lo, hi = ref.min(), ref.max()
            n = lambda v: (v - lo) / (hi - lo)  # noqa: E731
            target = torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
            p_ref = psnr(n(ref).clamp(0, 1), target)
            print('[pairs] seed %d, %dx%d: bf16 hi + lo pairs delta %.4f dB (direct %.1f dB) | fp16 operands delta %.4f dB (direct %.1f dB)' %
                  (seed, size, size, abs(psnr(n(y_def).clamp(0, 1), target) - p_ref), psnr(n(y_def), n(ref)),
                   abs(psnr(n(y_f16).clamp(0, 1), target) - p_ref), psnr(n(y_f16), n(ref))), flush=True)


if __name__ == '__main__':
    main()
