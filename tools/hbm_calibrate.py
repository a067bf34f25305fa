#!/usr/bin/env python
"""What a plain streaming kernel reaches on this GPU: device-to-device copy (read + write), fill (write only) and a
reduction (read only) over 2 GiB, through torch's own kernels.  Calibrates the "practical HBM roof" that the HBM-bound
kernels of DESIGN.md are compared with (the 8 TB/s of MI355X_MICROARCH.md is the pin rate).

    python tools/hbm_calibrate.py
"""
import json
import time

import torch


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n = 1 << 30  # bf16 elements = 2 GiB
    x = torch.ones(n, dtype=torch.bfloat16, device='cuda')
    y = torch.empty_like(x)
    gb = x.numel() * 2 / 1e9
    res = {
        'copy_gbs (read + write)': 2 * gb / timed(lambda: y.copy_(x)),
        'fill_gbs (write)': gb / timed(lambda: y.fill_(1.0)),
        'sum_gbs (read)': gb / timed(lambda: x.view(torch.int16).sum(dtype=torch.int64)),
        'add_gbs (2 reads + write)': 3 * gb / timed(lambda: torch.add(x, y, out=y)),
    }
    print(json.dumps({k: round(v, 1) for k, v in res.items()}))


if __name__ == '__main__':
    main()
