#!/usr/bin/env python3
"""A/B of one libsr_amd debug hook on the bench workload, inside ONE process (box-to-box variance is +-10%).

    python tools/ab_hook.py sr_debug_gates_no_prefetch [rounds]

Runs the bench.py forward with the hook at 0, 1, 0, 1, ... and prints wall ms/step and the per-group kernel ms of each.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
from super_resolution_amd import _lib as L  # noqa: E402


def main():
    hook = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    on_value = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # value passed for the 'on' leg
    dev = torch.device('cuda:0')
    net, _ = bench.build_model(dev)
    x = bench.synthetic_batch(16, 512, 512, dev).to(torch.bfloat16)
    fn = getattr(L.lib(), hook)
    outs = {}
    with torch.no_grad():
        net(x)
        for rd in range(rounds):
            for on in (0, 1):
                fn(on * on_value)
                net(x)
                torch.cuda.synchronize()
                net.profile(2)
                t0 = time.perf_counter()
                y = net(x)
                y = net(x)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 500
                net.profile_read()
                net.profile(1)
                y = net(x)
                torch.cuda.synchronize()
                groups = {k: round(v[0], 1) for k, v in net.profile_read().items()}
                net.profile(False)
                outs.setdefault(on, y.float().clone())
                print('%s=%d: %.1f ms/step  %s' % (hook, on, ms, groups), flush=True)
        fn(0)
    d = (outs[0] - outs[1]).abs().max().item()
    print('max |out(0) - out(1)| = %.3g   (max |out| = %.3g, rms diff %.3g, rms out %.3g)' %
          (d, outs[0].abs().max().item(), (outs[0] - outs[1]).pow(2).mean().sqrt().item(), outs[0].pow(2).mean().sqrt().item()))


if __name__ == '__main__':
    main()
