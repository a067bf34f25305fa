#!/usr/bin/env python3
"""Host-side cost of one forward: a tiny input (GPU work negligible) through the 23-block RRDBNet-FDConv.
usage: bench_host.py [SIDE=64] [REPS=5]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
net, _ = bench.build_model(dev)
x = bench.synthetic_batch(1, side, side, dev).to(torch.bfloat16)
with torch.no_grad():
    net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        net(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print('1x1x%dx%d: %.1f ms per forward on the host (issue), %.1f ms with the final sync' % (side, side, (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3))
