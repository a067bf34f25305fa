#!/usr/bin/env python
"""A/B of tile-to-rank assignments for the tiled 4096^2 configuration (BASELINE configs[2]) on ONE GPU.

Every rank of an N-rank job is played in turn by this one process: its shard of the 64 tiles runs through
RealESRGANer._run_tiles (same batching as the product path) and is timed with a device synchronise on both sides.
The job's compute time is the slowest rank's.  Two assignments:

  round_robin   tiles sorted by window area, dealt round-robin (rounds 1-2): near-perfect pixel balance, but every rank
                gets tiles of 3-4 window shapes = 3-4 small forwards, each paying the per-forward fixed cost
  blocks        tiles sorted by (area, shape), cut into N contiguous equal-count blocks: most ranks hold ONE shape

    python tools/ab_shard_policy.py [--worlds 2,4,8] [--reps 2] [--plain]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def area(w):
    return (w[0][1] - w[0][0]) * (w[0][3] - w[0][2])


def round_robin(wins, rank, world):
    order = sorted(range(len(wins)), key=lambda i: (-area(wins[i]), i))
    return [order[i] for i in range(rank, len(order), world)]


def blocks(wins, rank, world):
    order = sorted(range(len(wins)), key=lambda i: (-area(wins[i]), wins[i][0][1] - wins[i][0][0], i))
    n = len(order)
    return order[rank * n // world:(rank + 1) * n // world]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--worlds', default='2,4,8')
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--plain', action='store_true')
    ap.add_argument('--tile-batch', type=int, default=18)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    from super_resolution_amd import RealESRGANer
    dev = torch.device('cuda:0')
    net, _ = bench.build_model(dev, 3, plain=a.plain)
    img = (np.random.default_rng(1234).random((4096, 4096, 3)) * 255).astype(np.uint8)
    r = RealESRGANer(scale=4, model_path=None, model=net, tile=512, tile_pad=10, pre_pad=0, half=True, device=dev,
                     tile_batch=a.tile_batch)
    r.pre_process(img[:, :, ::-1].astype(np.float32) / 255.0)
    wins = r.tile_windows()
    res = {'tile_batch': a.tile_batch, 'plain': a.plain, 'worlds': {}}
    for world in [int(w) for w in a.worlds.split(',')]:
        row = {}
        for name, policy in (('round_robin', round_robin), ('blocks', blocks)):
            per_rank, forwards, px = [], [], []
            for rank in range(world):
                shard = policy(wins, rank, world)
                best = None
                for rep in range(a.reps + 1):  # the first pass builds the executors of this shard's batch shapes
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    r._run_tiles(wins, shard, sink=lambda i, t: None)
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t0
                    if rep:
                        best = dt if best is None else min(best, dt)
                per_rank.append(best * 1e3)
                forwards.append(len(r.batches_of(wins, shard, a.tile_batch)))
                px.append(sum(area(wins[i]) for i in shard))
                print('[ab_shard] world %d %-11s rank %d: %2d tiles, %d forwards, %.1f ms' %
                      (world, name, rank, len(shard), forwards[-1], per_rank[-1]), flush=True)
            row[name] = {'max_rank_ms': max(per_rank), 'per_rank_ms': per_rank, 'forwards': forwards,
                         'px_max_over_min': max(px) / min(px)}
        row['blocks_over_round_robin'] = row['blocks']['max_rank_ms'] / row['round_robin']['max_rank_ms']
        res['worlds'][str(world)] = row
        print('[ab_shard] world %d: slowest rank %.1f ms (round robin) -> %.1f ms (blocks), x%.3f' %
              (world, row['round_robin']['max_rank_ms'], row['blocks']['max_rank_ms'], row['blocks_over_round_robin']), flush=True)
    text = json.dumps(res, indent=1)
    if a.out:
        with open(a.out, 'w') as f:
            f.write(text + '\n')
    print(text)


if __name__ == '__main__':
    main()
