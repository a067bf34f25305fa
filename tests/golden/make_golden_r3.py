#!/usr/bin/env python
"""Round-3 fixture: the CPU oracle's output for BASELINE configs[1]'s network at its own size (23 blocks, one
1 x 1 x 512 x 512 sample), stored on a pixel lattice so that the GPU test need not spend five minutes of CPU on it.

  rrdbnet_23block_512.npz   float64 oracle and float32 oracle outputs of the owner-script network
                            (tests/test_gpu_parity_r2.py::realistic_net, seed 2) on gp.image_input((1,1,512,512), 84):
                            every 4th pixel of the 2048 x 2048 output (512 x 512 lattice) plus the central 256 x 256
                            crop at full resolution, the full-image minimum / maximum of the float32 oracle (the
                            normalisation of the PSNR criterion), the full-image float32-vs-float64 floor, and a
                            checksum of the seeded state_dict (the test refuses the fixture if the constructors'
                            initialisation ever differs from the one the fixture was made with).

The producer is oracle/ (pinned to the reference by tests/test_oracle_golden.py), not the reference itself: basicsr's
RRDBNet is not vendored, so no reference run of the whole network exists.  SR_AMD_LIVE_ORACLE=1 makes the
test run both oracle forwards itself instead of reading this file.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import golden_params as gp  # noqa: E402
from oracle.rrdbnet_ref import rrdbnet_forward  # noqa: E402

STRIDE = 4
CROP = (896, 1152)  # central 256 x 256 of the 2048 x 2048 output


def realistic_state(args, seed):
    """Same construction as tests/test_gpu_parity_r2.py::realistic_net (CPU part)."""
    import super_resolution_amd as sra
    torch.manual_seed(seed)
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(*args))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if any(s in name for s in ('KSM_Local.conv.weight', 'freq_weight_conv_list')) and p.dim() > 1:
                p.add_(0.003 * torch.randn(p.shape, generator=g))
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def state_checksum(sd):
    return float(sum(v.double().abs().sum().item() for k, v in sorted(sd.items()) if v.dtype.is_floating_point))


def main():
    torch.set_num_threads(int(os.environ.get('SR_AMD_ORACLE_THREADS', '8')))
    sd = realistic_state((1, 1, 4, 64, 23, 32), seed=2)
    x = torch.from_numpy(gp.image_input((1, 1, 512, 512), 84))
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
        print('float32 oracle done', flush=True)
        ref64 = rrdbnet_forward(sd, x.double(), dtype=torch.float64)
        print('float64 oracle done', flush=True)
    a, b = CROP
    out = {
        'stride': np.int64(STRIDE), 'crop': np.asarray(CROP, dtype=np.int64),
        'state_checksum': np.float64(state_checksum(sd)),
        'ref32_min': np.float32(ref.min().item()), 'ref32_max': np.float32(ref.max().item()),
        'floor_full': np.float64((ref.double() - ref64).abs().max().item()),
        'ref64_lattice': ref64[0, 0, ::STRIDE, ::STRIDE].numpy(),
        'ref32_lattice': ref[0, 0, ::STRIDE, ::STRIDE].numpy(),
        'ref64_crop': ref64[0, 0, a:b, a:b].numpy(),
        'ref32_crop': ref[0, 0, a:b, a:b].numpy(),
    }
    np.savez_compressed(os.path.join(HERE, 'rrdbnet_23block_512.npz'), **out)
    print('wrote rrdbnet_23block_512.npz: floor %.3e, range [%.4f, %.4f], checksum %.10e'
          % (out['floor_full'], out['ref32_min'], out['ref32_max'], out['state_checksum']))


if __name__ == '__main__':
    main()
