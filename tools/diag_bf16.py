"""Diagnostic (GPU only, no oracle): bf16-vs-fp32 PSNR of the 23-block realistic network by size, seed and kernel family."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_params as gp
import super_resolution_amd as sra
from super_resolution_amd import _lib

lib = _lib.lib()


def make(seed, blocks=23):
    torch.manual_seed(seed)
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, blocks, 32))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if any(s in name for s in ('KSM_Local.conv.weight', 'freq_weight_conv_list')) and p.dim() > 1:
                p.add_(0.003 * torch.randn(p.shape, generator=g))
    return net.eval().cuda()


def psnr(a, b):
    span = float(b.max() - b.min())
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10 * math.log10(span * span / mse)


hooks = {'default': [], 'fft_lowpass': [('sr_debug_force_fft_lowpass', 1, -1)], 'generic_fft': [('sr_debug_force_generic_fft', 1, 0)],
         'generic_gates': [('sr_debug_force_generic_gates', 1, 0)], 'generic_conv': [('sr_debug_force_generic_conv', 1, 0)],
         'gates_no_direct': [('sr_debug_gates_no_direct', 1, 0)]}
sizes = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [64, 128, 256, 512]
seeds = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2, 3]
which = sys.argv[3].split(',') if len(sys.argv) > 3 else ['default']
for seed in seeds:
    net = make(seed)
    for side in sizes:
        x = torch.from_numpy(gp.image_input((1, 1, side, side), 84)).cuda()
        with torch.no_grad():
            y32 = net(x).float().cpu()
            for name in which:
                for fn, on, off in hooks[name]:
                    getattr(lib, fn)(on)
                yb = net(x.to(torch.bfloat16)).float().cpu()
                for fn, on, off in hooks[name]:
                    getattr(lib, fn)(off)
                print('seed %d side %4d %-16s bf16-vs-fp32 PSNR %.1f dB  (span %.2f)' % (seed, side, name, psnr(yb, y32), float(y32.max() - y32.min())), flush=True)
