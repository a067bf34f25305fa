#!/usr/bin/env python
"""A/B: BASELINE configs[1] (batch 16 of 512x512, bf16) as ONE forward against the same batch cut into N independent
sample groups that run concurrently on N streams (every statistic of the network is per sample, so the groups are
independent chains of kernels; the persistent kernels of one chain can then fill the ramp-up / tail of the other's).

    python tools/ab_two_streams.py [--chains 2] [--reps 3]
"""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=2)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=512)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    net, _ = bench.build_model(dev, 1)
    x = bench.synthetic_batch(a.batch, a.size, a.size, dev).to(torch.bfloat16)
    with torch.no_grad():
        y_ref = net(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            net(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        one = min(ts)
        print('one forward of %d samples: %.1f ms (runs: %s)' % (a.batch, one * 1e3, ' '.join('%.1f' % (t * 1e3) for t in ts)), flush=True)
        n = a.chains
        per = a.batch // n
        nets = [net] + [bench.build_model(dev, 1)[0] for _ in range(n - 1)]  # same seed: same weights
        streams = [torch.cuda.Stream(dev) for _ in range(n)]
        parts = [x[i * per:(i + 1) * per].contiguous() for i in range(n)]
        outs = [None] * n

        def work(i):
            with torch.no_grad(), torch.cuda.stream(streams[i]):
                outs[i] = nets[i](parts[i])

        for i in range(n):  # warm-up, one chain at a time
            work(i)
            torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        split = min(ts)
        y = torch.cat(outs)
        same = bool((y == y_ref).all())
        print('%d concurrent chains of %d samples: %.1f ms (runs: %s); outputs bit-identical to the single forward: %s; ratio %.3f'
              % (n, per, split * 1e3, ' '.join('%.1f' % (t * 1e3) for t in ts), same, split / one), flush=True)
        # the same groups one after the other on one stream (what the split costs without the overlap)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            for i in range(n):
                nets[0](parts[i])
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print('%d groups of %d samples back to back on one stream: %.1f ms' % (n, per, min(ts) * 1e3), flush=True)


if __name__ == '__main__':
    main()
