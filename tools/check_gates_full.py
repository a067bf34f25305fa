#!/usr/bin/env python3
"""sr_fbm_gates_accumulate at the bench size: direct form against the T-image form on the same inputs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from super_resolution_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device('cuda:0')
st = L.stream_ptr(dev)
B, H, W = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (16, 512, 512)
C, NC = 64, 5
torch.manual_seed(0)
x = torch.randn(B, C // 32, H, W, 32, device=dev).to(torch.bfloat16)
gw = [[(torch.randn(C, 9, device=dev) * 0.05).contiguous() for _ in range(3)] for _ in range(NC)]
gb = [[torch.randn(1, device=dev) for _ in range(3)] for _ in range(NC)]
res = []
for no_direct in (0, 1):
    lib.sr_debug_gates_no_direct(no_direct)
    logits = [torch.full((B, H, W, 3), float('nan'), device=dev) for _ in range(NC)]
    gates = torch.full((B, H, W, 4), float('nan'), device=dev)
    cons = (L.GateConsumer * NC)()
    for j in range(NC):
        for k in range(3):
            cons[j].w[k] = gw[j][k].data_ptr()
            cons[j].b[k] = gb[j][k].data_ptr()
        cons[j].logits = logits[j].data_ptr()
        cons[j].gates = gates.data_ptr() if j == 0 else None
        cons[j].first = 1
    L.check(lib.sr_fbm_gates_accumulate(1, B, H, W, L.blocked_view(x), NC, cons, st))
    torch.cuda.synchronize()
    res.append([gates[..., :3].clone()] + [l.clone() for l in logits[1:]])
lib.sr_debug_gates_no_direct(0)
for i, (a, b) in enumerate(zip(*res)):
    d = (a - b).abs()
    print('output %d: max diff %.3g  nan %d/%d  max |ref| %.3g' % (i, d.nan_to_num(1e9).max().item(), int(torch.isnan(a).sum()),
                                                                  int(torch.isnan(b).sum()), b.abs().max().item()))
    if d.nan_to_num(1e9).max().item() > 1e-3:
        bad = (d.nan_to_num(1e9) > 1e-3).nonzero()
        print('   first bad', bad[:5].tolist(), ' count', bad.shape[0])
