"""Diagnostic (GPU box): where does the 512x512 23-block network lose parity?  fp32 GPU vs fp32 / fp64 oracle, bf16 GPU with
the DFT-as-GEMM low-pass vs with the FFT low-pass."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_params as gp
from oracle.rrdbnet_ref import rrdbnet_forward
import super_resolution_amd as sra
from super_resolution_amd import _lib

side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
do64 = len(sys.argv) > 2 and sys.argv[2] == '64'
SEED = int(os.environ.get('DIAG_SEED', '3'))
torch.manual_seed(SEED)
net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, 23, 32))
g = torch.Generator().manual_seed(SEED + 1)
with torch.no_grad():
    for name, p in net.named_parameters():
        if any(s in name for s in ('KSM_Local.conv.weight', 'freq_weight_conv_list')) and p.dim() > 1:
            p.add_(0.003 * torch.randn(p.shape, generator=g))
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.eval().cuda()
x = torch.from_numpy(gp.image_input((1, 1, side, side), 84))


def psnr(a, b, span):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10 * math.log10(span * span / mse)


with torch.no_grad():
    y32 = net(x.cuda()).cpu()
    yb = net(x.cuda().to(torch.bfloat16)).float().cpu()
    _lib.lib().sr_debug_force_fft_lowpass(1)
    yb_fft = net(x.cuda().to(torch.bfloat16)).float().cpu()
    _lib.lib().sr_debug_force_fft_lowpass(-1)
    _lib.lib().sr_debug_force_generic_fft(1)
    y32_gen = net(x.cuda()).cpu()
    _lib.lib().sr_debug_force_generic_fft(0)
    print('gpu32 (pow2 FFT) vs gpu32 (any-size FFT): max %.3e' % float((y32 - y32_gen).abs().max()), flush=True)
    t0 = time.time()
    ref32 = rrdbnet_forward(sd, x)
    print('oracle fp32 %.1f s' % (time.time() - t0), flush=True)
span = float(ref32.max() - ref32.min())
print('side %d span %.3f' % (side, span))
print('gpu32 vs ref32: max %.3e   psnr %.1f' % (float((y32 - ref32).abs().max()), psnr(y32, ref32, span)))
print('bf16 (mfma lowpass) vs gpu32: psnr %.1f   vs ref32 %.1f' % (psnr(yb, y32, span), psnr(yb, ref32, span)))
print('bf16 (fft  lowpass) vs gpu32: psnr %.1f   vs ref32 %.1f' % (psnr(yb_fft, y32, span), psnr(yb_fft, ref32, span)), flush=True)
if do64:
    import threading
    stop = threading.Event()
    threading.Thread(target=lambda: [print('.', end='', flush=True) for _ in iter(lambda: stop.wait(30.0), True)], daemon=True).start()
    with torch.no_grad():
        t0 = time.time()
        ref64 = rrdbnet_forward(sd, x.double(), dtype=torch.float64)
    stop.set()
    print('oracle fp64 %.1f s' % (time.time() - t0))
    print('ref32 vs ref64: max %.3e psnr %.1f' % (float((ref32.double() - ref64).abs().max()), psnr(ref32, ref64, span)))
    print('gpu32 vs ref64: max %.3e psnr %.1f' % (float((y32.double() - ref64).abs().max()), psnr(y32, ref64, span)))
    print('gpu32 (any-size FFT) vs ref64: max %.3e' % float((y32_gen.double() - ref64).abs().max()))
