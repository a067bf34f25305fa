#!/usr/bin/env python3
"""Tiled RealESRGANer.enhance on one GPU (BASELINE.json configs[2] in small): RRDBNet-FDConv(3, 3) x4, bf16,
tile 512 / pad 10 by default.   usage: bench_tiled.py [SIDE=2048] [TILE=512] [PAD=10] [TILE_BATCH ...]
(tile 448 / pad 32 makes every window a multiple of 32 no larger than 512: those take the MFMA low-pass.)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_params as gp  # noqa: E402
from super_resolution_amd import RRDBNet, RealESRGANer, replace_conv_with_fdconv  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    pad = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    batches = [int(a) for a in sys.argv[4:]] or [1, 8]
    dev = torch.device('cuda', 0)
    net = replace_conv_with_fdconv(RRDBNet(3, 3, 4, 64, 23, 32))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gp.rrdbnet_state(3, 3, 64, 23, 32, seed=1234).items()}, strict=True)
    img = (np.random.default_rng(1234).random((side, side, 3)) * 255).astype(np.uint8)
    for tb in batches:
        r = RealESRGANer(scale=4, model_path=None, model=net, tile=tile, tile_pad=pad, pre_pad=0, half=True, device=dev, tile_batch=tb)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, _ = r.enhance(img)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print('side %d tile %d pad %d tile_batch %d: %.2f s  %.2f out-Mpix/s  (%d tiles)' % (side, tile, pad, tb, dt, out.shape[0] * out.shape[1] / 1e6 / dt,
                                                                            len(r.tile_windows())), flush=True)


if __name__ == '__main__':
    main()
