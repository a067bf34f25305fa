#!/usr/bin/env python
"""What a one-shot caller pays: RealESRGANer(...).enhance(img) on a fresh process, first call against second call
(executor construction, per-shape plans and tables, workspace allocation), 4096^2 RGB image at the default tiling.

    python tools/first_call_latency.py [--side 4096]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--side', type=int, default=4096)
    a = ap.parse_args()
    from super_resolution_amd import RealESRGANer
    dev = torch.device('cuda:0')
    t0 = time.perf_counter()
    net, _ = bench.build_model(dev, 3)
    torch.cuda.synchronize()
    print('[first] model on the device: %.2f s' % (time.perf_counter() - t0), flush=True)
    img = (np.random.default_rng(1).random((a.side, a.side, 3)) * 255).astype(np.uint8)
    r = RealESRGANer(scale=4, model_path=None, model=net, tile=512, tile_pad=10, pre_pad=0, half=True, device=dev)
    for k in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _ = r.enhance(img)
        torch.cuda.synchronize()
        print('[first] enhance() call %d: %.2f s (tile_batch %d)' % (k + 1, time.perf_counter() - t0, r.tile_batch), flush=True)


if __name__ == '__main__':
    main()
