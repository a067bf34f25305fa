"""Diagnostic (GPU only): mixed-precision vs float32 TRAIN-mode forward / gradient of RRDBNet-FDConv by depth."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import super_resolution_amd as sra

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
depths = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4, 8, 16, 23]
hr = 256
g = torch.Generator().manual_seed(4321)
srt = torch.rand(B, 1, hr, hr, generator=g) * torch.linspace(0.3, 1.0, B).view(B, 1, 1, 1)
srt = torch.nn.functional.avg_pool2d(srt, 3, 1, 1)
wf = torch.nn.functional.avg_pool2d(srt, 4)
srt, wf = srt.cuda(), wf.cuda()


def psnr(a, b):
    span = float(b.max() - b.min())
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10 * math.log10(span * span / mse)


for nb in depths:
    torch.manual_seed(1234)
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, nb, 32)).cuda()
    start = {k: v.detach().clone() for k, v in net.state_dict().items()}
    out = {}
    for mode in ('eval', 'train'):
        for name, cdt in (('f32', None), ('bf16', torch.bfloat16)):
            net.load_state_dict(start)
            net.train(mode == 'train')
            net.compute_dtype = cdt
            net.zero_grad(set_to_none=True)
            pred = net(wf)
            loss = sra.head.resize_l1_loss(pred, srt)
            loss.backward()
            gr = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None]).double()
            out[(mode, name)] = (pred.detach().float(), loss.item(), gr)
        (p32, l32, g32), (pb, lb, gb) = out[(mode, 'f32')], out[(mode, 'bf16')]
        cos = float((g32 * gb).sum() / (g32.norm() * gb.norm()))
        print('blocks %2d %-5s B=%d: loss f32 %.5f bf16 %.5f | pred PSNR %.1f dB (span %.3f) | grad cosine %.5f |g32| %.3e |gb| %.3e'
              % (nb, mode, B, l32, lb, psnr(pb, p32), float(p32.max() - p32.min()), cos, float(g32.norm()), float(gb.norm())), flush=True)
