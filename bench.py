#!/usr/bin/env python
"""bench.py -- output Mpix/s of RRDBNet-FDConv x4 inference (BASELINE.json configs[1]) on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one forward of RRDBNet(1,1,64,23,32) with every conv swapped for FDConv, on a synthetic batch of
16 grayscale 512x512 images that is already resident in HBM, computed in bf16 (MFMA) through libsr_amd.
The path shards by batch with no exchange step, so every rank runs the same per-GPU batch ("weak").
Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline     : the conv kernels (the dominant launch group; HBM-bound on FDConv layers), measured live with HIP
                 events recorded by the executor on the launch stream around every conv launch of the timed steps;
  cpu_baseline : the CPU oracle (torch fp32 restatement pinned by the golden fixtures) timed on this host on a
                 bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
NUM_FEAT, NUM_BLOCK, NUM_GROW = 64, 23, 32


def conv_macs_per_lr_pixel(cin0, cout_last, nf=NUM_FEAT, nb=NUM_BLOCK, gc=NUM_GROW):
    """algorithmic conv MACs of RRDBNet x4 per LR pixel (17,905,056 for 1-channel I/O)."""
    rdb = 9 * (sum((nf + k * gc) * gc for k in range(4)) + (nf + 4 * gc) * nf)
    return 9 * cin0 * nf + 3 * nb * rdb + 9 * nf * nf * (1 + 4 + 16 + 16) + 16 * 9 * nf * cout_last


def conv_bytes_per_lr_pixel(cin0, cout_last, nf=NUM_FEAT, nb=NUM_BLOCK, gc=NUM_GROW, es=2):
    """SURVEY 8(d) / DESIGN.md 4: algorithmic HBM bytes of all conv launches of one forward, per LR pixel per sample.

    An FDConv conv must read its input and the three low-pass planes of it (4 x cin elements per input pixel) plus
    the three fp32 gates (12 B per input pixel), and write cout elements per output pixel; residual operands add
    cout elements each; thin layers (conv_first / conv_last, min(c) <= 16) are plain convs.  Per-sample weights
    are a few hundred KB per launch and are not counted.  `es` = bytes per activation element (2: bf16).
    """
    def fd(cin, cout, in_px, out_px, nres=0):
        return in_px * (4 * cin * es + 12) + out_px * cout * es * (1 + nres)
    tot = 1 * (cin0 * es) + 1 * nf * es                       # conv_first (plain)
    for blk in range(3 * nb):
        for k in range(4):
            tot += fd(nf + k * gc, gc, 1, 1)
        tot += fd(nf + 4 * gc, nf, 1, 1, nres=2 if blk % 3 == 2 else 1)
    tot += fd(nf, nf, 1, 1, nres=1)                           # conv_body + trunk residual
    tot += fd(nf, nf, 1, 4)                                   # conv_up1 reads LR, writes 2x
    tot += fd(nf, nf, 4, 16)                                  # conv_up2 reads 2x, writes 4x
    tot += fd(nf, nf, 16, 16)                                 # conv_hr
    tot += 16 * nf * es + 16 * cout_last * es                 # conv_last (plain)
    return tot


def measured_conv_traffic():
    """HBM bytes per conv launch from the committed PMC passes of this same command (profiles/*_hbm_traffic.json:
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; counters cannot be read in-process)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    return d['per_launch']['conv']['hbm_gb_per_launch_corrected'] * 1e9, os.path.relpath(files[-1], ROOT)


def synthetic_batch(B, H, W, device, seed=1234):
    """DL-SMLM-like wide-field frames: sparse positive blobs (Poisson emitters blurred by a PSF) + floor noise."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    emit = torch.poisson(torch.full((B, 1, H, W), 0.05), generator=g)
    k = torch.arange(-6, 7, dtype=torch.float32)
    k = torch.exp(-k * k / (2 * 2.0 * 2.0))
    k = (k / k.sum()).view(1, 1, 1, -1)
    blur = torch.nn.functional.conv2d(torch.nn.functional.conv2d(emit, k, padding=(0, 6)), k.transpose(2, 3), padding=(6, 0))
    x = (blur * 4 + 0.02 * torch.rand(B, 1, H, W, generator=g)).clamp(0, 1)
    return x.to(device)


def build_model(device, seed=1234):
    import golden_params as gp
    from super_resolution_amd import RRDBNet, replace_conv_with_fdconv
    net = replace_conv_with_fdconv(RRDBNet(1, 1, 4, NUM_FEAT, NUM_BLOCK, NUM_GROW))
    sd = {k: torch.from_numpy(v) for k, v in gp.rrdbnet_state(1, 1, NUM_FEAT, NUM_BLOCK, NUM_GROW, seed=seed).items()}
    net.load_state_dict(sd, strict=True)
    return net.eval().to(device), sd


def cpu_baseline(sd, sample_hw):
    """Oracle forward on the host cores on a bounded sample; returns the dict for the JSON line."""
    from oracle.rrdbnet_ref import rrdbnet_forward
    threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or torch.get_num_threads()
    torch.set_num_threads(threads)
    x = synthetic_batch(1, sample_hw, sample_hw, 'cpu')
    t0 = time.perf_counter()
    with torch.no_grad():
        y = rrdbnet_forward(sd, x)
    dt = time.perf_counter() - t0
    return {'value': y.shape[-1] * y.shape[-2] / 1e6 / dt, 'unit': 'output Mpix/s', 'cores': threads, 'kind': 'port',
            'sample': 'one forward of the same 23-block RRDBNet-FDConv on 1x1x%dx%d (fp32 torch oracle, %.1f s), '
                      'no extrapolation' % (sample_hw, sample_hw, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--cpu-sample', type=int, default=208, help='side of the CPU-baseline sample image (0: skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU-baseline leg (profiling runs)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from super_resolution_amd.build import build
    if rank == 0:
        build(verbose=False)
    if world > 1:
        dist.barrier()

    net, sd = build_model(dev)
    B, H, W = args.batch, args.size, args.size
    x = synthetic_batch(B, H, W, dev, seed=1234 + rank).to(torch.bfloat16)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            y = net(x)
        sync()
        net.profile(2)  # HIP events around the conv launches only (the roofline kernel): 11 ms/step less than all groups
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = net(x)
        sync()
        elapsed = time.perf_counter() - t0
        prof = net.profile_read()
        # per-group breakdown from one extra, untimed step with events around every kernel group
        net.profile(1)
        net(x)
        sync()
        prof_all = net.profile_read()
    net.profile(False)
    finite = bool(torch.isfinite(y.float()).all())
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)

    out_mpix = world * B * (4 * H) * (4 * W) / 1e6
    ms_step = elapsed / args.steps * 1e3
    conv_ms, conv_n = prof['conv']
    conv_flops_step = 2.0 * conv_macs_per_lr_pixel(1, 1) * B * H * W
    conv_s_step = conv_ms / 1e3 / args.steps
    tflops = conv_flops_step / conv_s_step / 1e12
    conv_bytes_step = float(conv_bytes_per_lr_pixel(1, 1)) * B * H * W
    gbs = conv_bytes_step / conv_s_step / 1e9
    traffic, traffic_src = measured_conv_traffic()
    res = {
        'metric': 'output Mpix/s (4x SR) at 1/2/4/8 MI355X; conv MFMA util %',
        'value': out_mpix / (elapsed / args.steps),
        'unit': 'output Mpix/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': max(args.warmup, 1),
        'ms_per_step': ms_step,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'bf16',
        'data': 'synthetic',
        'config': {'workload': 'RRDBNet-FDConv(1,1,64,23,32) x4 inference, batch %d of %dx%d synthetic grayscale per GPU, '
                               'bf16 (BASELINE.json configs[1])' % (B, H, W),
                   'global_batch': world * B, 'parallelism': 'batch-sharded replicas x%d, no exchange' % world,
                   'output_finite': finite},
        # the FDConv conv streams 4 tensors per input channel: ~70 FLOP/B, far left of the 312 FLOP/B ridge -> HBM-bound
        'roofline': {'bound': 'hbm', 'kernel': 'conv3x3_mfma_bf16 (average over the %d conv launches of a step)' % (conv_n // args.steps),
                     'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': traffic,
                     'traffic_unit': 'bytes per launch (PMC, %s)' % traffic_src,
                     'avg_launch_ms': conv_ms / max(conv_n, 1),
                     'algorithmic_gb_per_launch': conv_bytes_step / 1e9 / max(conv_n // args.steps, 1),
                     'mfma': {'achieved': tflops, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': tflops / BF16_MFMA_PEAK_TFLOPS, 'algorithmic_tflop_per_step': conv_flops_step / 1e12}},
        'kernel_ms_per_step': {k: v[0] for k, v in prof_all.items()},
        'kernel_launches_per_step': {k: v[1] for k, v in prof_all.items()},
        'kernel_ms_source': 'one extra untimed step with events around every kernel group; the conv events of the roofline are in the timed steps',
    }
    if rank == 0:
        if world == 1 and args.cpu_sample > 0 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(sd, args.cpu_sample)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
