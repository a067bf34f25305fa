#!/usr/bin/env python3
"""cProfile of one tiled enhance() (host-side hot spots of the tiled path).  usage: prof_tiled.py [SIDE TILE PAD TILE_BATCH]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_params as gp  # noqa: E402
from super_resolution_amd import RRDBNet, RealESRGANer, replace_conv_with_fdconv  # noqa: E402

side, tile, pad, tb = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (2048, 448, 32, 8)
dev = torch.device('cuda', 0)
net = replace_conv_with_fdconv(RRDBNet(3, 3, 4, 64, 23, 32))
net.load_state_dict({k: torch.from_numpy(v) for k, v in gp.rrdbnet_state(3, 3, 64, 23, 32, seed=1234).items()}, strict=True)
img = (np.random.default_rng(1234).random((side, side, 3)) * 255).astype(np.uint8)
r = RealESRGANer(scale=4, model_path=None, model=net, tile=tile, tile_pad=pad, pre_pad=0, half=True, device=dev, tile_batch=tb)
r.enhance(img)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
r.enhance(img)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
