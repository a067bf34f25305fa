#!/usr/bin/env python
"""bench.py -- output Mpix/s of RRDBNet-FDConv x4 inference on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2                    # BASELINE.json configs[1] (the metric's config)
    python bench.py --config tiled --steps 1 --warmup 1              # BASELINE.json configs[2]: 4096^2 tiled image
    python bench.py --config train --steps 2 --warmup 1              # BASELINE.json configs[3]: training step (fp32)
    python bench.py --config srvgg                                   # BASELINE.json configs[4]: SRVGGNetCompact 1080p
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W [--config tiled]

--config batch (default): a step = one forward of RRDBNet(1,1,64,23,32) with every conv swapped for FDConv on a
synthetic batch of 16 grayscale 512x512 images already resident in HBM, bf16 (MFMA) through libsr_amd.  The path
shards by batch with no exchange step, so every rank runs the same per-GPU batch ("weak").
--config tiled: a step = RealESRGANer(tile=512, tile_pad=10, pre_pad=0, half=True, shard_tiles=True).tile_process()
+ post_process() of one 4096x4096 RGB image that pre_process has already placed in HBM, RRDBNet-FDConv(3,3,64,23,32);
the 64 tiles, sorted by window shape, go to the N ranks in contiguous blocks and rank 0 gathers the cores point to point
("strong": the image is fixed).  The host-to-host enhance() time and the non-parallel share are reported beside it.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline     : the conv kernels on the MFMA roof as SURVEY 8(d) prescribes (conv FLOPs / conv kernel time / 2.5
                 PFLOP/s), measured live with HIP events recorded by the executor on the launch stream around every
                 conv launch of the timed steps; `hbm` = the same kernels against the HBM roof on STRICT algorithmic
                 bytes (every conv input once, output once, residual operands once); `groups` = low-pass, gates and
                 head convs against the HBM roof on their strict bytes;
  cpu_baseline : the CPU oracle (torch fp32 restatement pinned by the golden fixtures) timed on this host on a
                 bounded sample of the same workload.
  secondary    : (default config only) one short timed step of each other BASELINE config; at N > 1 the ones that shard
                 -- the tiled image over the N ranks (strong scaling) and the training step with its all-reduce.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
NUM_FEAT, NUM_BLOCK, NUM_GROW = 64, 23, 32
ES = 2                          # bytes per activation element (bf16)


def conv_macs_per_lr_pixel(cin0, cout_last, nf=NUM_FEAT, nb=NUM_BLOCK, gc=NUM_GROW, head_only=False):
    """algorithmic conv MACs of RRDBNet x4 per LR pixel."""
    rdb = 9 * (sum((nf + k * gc) * gc for k in range(4)) + (nf + 4 * gc) * nf)
    head = 9 * nf * nf * (4 + 16 + 16) + 16 * 9 * nf * cout_last   # conv_up1, conv_up2, conv_hr, conv_last
    if head_only:
        return head
    return 9 * cin0 * nf + 3 * nb * rdb + 9 * nf * nf + head


def strict_bytes_per_lr_pixel(cin0, cout_last, nf=NUM_FEAT, nb=NUM_BLOCK, gc=NUM_GROW, es=ES):
    """Strict algorithmic HBM bytes per LR pixel per sample of each kernel group (read every input once,
    write every output once).
      conv    : trunk convs (conv_first, 69 RDB x 5, conv_body): cin in + cout out + cout per residual operand
      head    : conv_up1 (reads LR, writes 2x), conv_up2 (2x -> 4x), conv_hr, conv_last
      lowpass : per produced tensor that feeds an FDConv layer: read x once, write three planes (8 B / px / channel in
                bf16); 192 channels per RDB + conv_body's input; lowpass_hr: the inputs of conv_up1 / conv_up2 / conv_hr
                (read at the source resolution, written at the conv's resolution)
      gates   : per RDB read the 192 produced channels once and write 5 gate records (float4); + conv_body and the three
                FDConv head layers (read cin, write one record per pixel of their resolution)"""
    rec = 16
    conv = cin0 * es + nf * es
    for blk in range(3 * nb):
        for k in range(4):
            conv += (nf + k * gc) * es + gc * es
        conv += (nf + 4 * gc) * es + nf * es * (1 + (2 if blk % 3 == 2 else 1))
    conv += nf * es + nf * es * 2
    head = (nf * es + 4 * nf * es) + (4 * nf * es + 16 * nf * es) + (16 * nf * es + 16 * nf * es) + (16 * nf * es + 16 * cout_last * es)
    lowpass = (3 * nb * (nf + 4 * gc) + nf) * 4 * es
    lowpass_hr = (nf * es + 4 * 3 * nf * es) + (4 * nf * es + 16 * 3 * nf * es) + (16 * nf * es + 16 * 3 * nf * es)
    gates = 3 * nb * ((nf + 4 * gc) * es + 5 * rec) + (nf * es + rec) + (nf * es + 4 * rec) + (4 * nf * es + 16 * rec) + (16 * nf * es + 16 * rec)
    return {'conv': conv, 'head': head, 'lowpass': lowpass, 'lowpass_hr': lowpass_hr, 'gates': gates}


def measured_traffic():
    """HBM bytes per launch and kernel group from the committed PMC passes of this same command (profiles/
    *_hbm_traffic.json, newest round: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE;
    counters cannot be read in-process)."""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic.json')))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        d = json.load(f)
    return {k: v['hbm_gb_per_launch_corrected'] * 1e9 for k, v in d['per_launch'].items()}, os.path.relpath(files[-1], ROOT)


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


def build_model(device, ch=1, seed=1234, plain=False):
    import golden_params as gp
    from super_resolution_amd import RRDBNet, replace_conv_with_fdconv
    if plain:  # RealESRGAN_x4plus's architecture (inference_realesrgan.py:59-62): RRDBNet's own seeded initialisation, no FDConv
        torch.manual_seed(seed)
        net = RRDBNet(ch, ch, 4, NUM_FEAT, NUM_BLOCK, NUM_GROW)
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}  # CPU copy for the oracle leg, taken before the move
        return net.eval().to(device), sd
    net = replace_conv_with_fdconv(RRDBNet(ch, ch, 4, NUM_FEAT, NUM_BLOCK, NUM_GROW))
    sd = {k: torch.from_numpy(v) for k, v in gp.rrdbnet_state(ch, ch, NUM_FEAT, NUM_BLOCK, NUM_GROW, seed=seed).items()}
    net.load_state_dict(sd, strict=True)
    return net.eval().to(device), sd


def cpu_baseline(sd, sample_hw, ch=1):
    """Oracle forward on the host cores on a bounded sample (one warm-up, then the median of three runs);
    returns the dict for the JSON line."""
    from oracle.rrdbnet_ref import rrdbnet_forward
    threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or torch.get_num_threads()
    torch.set_num_threads(threads)
    x = synthetic_batch(1, sample_hw, sample_hw, 'cpu')
    if ch > 1:
        x = x.expand(1, ch, sample_hw, sample_hw).contiguous()
    times = []
    with torch.no_grad():
        for _ in range(4):
            t0 = time.perf_counter()
            y = rrdbnet_forward(sd, x)
            times.append(time.perf_counter() - t0)
    dt = float(np.median(times[1:]))
    return {'value': y.shape[-1] * y.shape[-2] / 1e6 / dt, 'unit': 'output Mpix/s', 'cores': threads, 'kind': 'port',
            'sample': 'forward of the same 23-block RRDBNet-FDConv on 1x%dx%dx%d (fp32 torch oracle): 1 warm-up (%.1f s) + median '
                      'of 3 runs (%.1f s), no extrapolation' % (ch, sample_hw, sample_hw, times[0], dt)}


def stream_calibration(dev):
    """What plain streaming kernels reach on this GPU right now (torch's copy and add over 1 GiB tensors): the practical
    HBM rate next to the 8 TB/s pin rate that `peak` states."""
    n = 1 << 29
    x = torch.ones(n, dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(x)
    out = {}
    for name, fn, mult in (('copy_gbs', lambda: y.copy_(x), 2), ('add_gbs', lambda: torch.add(x, y, out=y), 3)):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize(dev)
        out[name] = mult * n * 2 / 1e9 / ((time.perf_counter() - t0) / 10)
    del x, y
    torch.cuda.empty_cache()
    out['note'] = 'device-to-device copy (1 read + 1 write) and add (2 reads + 1 write) of 1 GiB bf16 tensors, torch kernels'
    return out


def roofline_fields(prof, prof_all, steps, lr_px_step, ch, traffic_applies=False):
    """roofline object from the executor's event timings.  prof: conv groups over the timed steps; prof_all: every
    group over one extra untimed step; lr_px_step: LR pixels x samples one step pushes through the network."""
    conv_ms = prof['conv'][0] + prof['head'][0]
    conv_n = prof['conv'][1] + prof['head'][1]
    conv_s_step = conv_ms / 1e3 / steps
    flops_step = 2.0 * conv_macs_per_lr_pixel(ch, ch) * lr_px_step
    tflops = flops_step / conv_s_step / 1e12
    strict = strict_bytes_per_lr_pixel(ch, ch)
    traffic, traffic_src = measured_traffic()
    n_step = max(conv_n // steps, 1)
    conv_strict_gb = (strict['conv'] + strict['head']) * lr_px_step / 1e9
    groups = {}
    for g, keys in (('conv_trunk', ('conv',)), ('head', ('head',)), ('lowpass', ('lowpass',)), ('lowpass_hr', ('lowpass_hr',)),
                    ('gates', ('gates',))):
        ms = sum(prof_all[k][0] for k in keys)
        gb = sum(strict[k] for k in keys) * lr_px_step / 1e9
        if ms > 0:
            groups[g] = {'ms_per_step': ms, 'launches': sum(prof_all[k][1] for k in keys), 'algorithmic_gb_per_step': gb,
                         'achieved_gbs': gb / (ms / 1e3), 'frac_of_hbm_peak': gb / (ms / 1e3) / HBM_PEAK_GBS}
    head_ms = prof_all['head'][0]
    if head_ms > 0:
        groups['head']['tflops'] = 2.0 * conv_macs_per_lr_pixel(ch, ch, head_only=True) * lr_px_step / (head_ms / 1e3) / 1e12
    hbm_measured = None
    if traffic_applies and traffic.get('conv'):
        # the PMC passes profile THIS workload (config 2): bytes per average conv launch / this run's average launch time
        gbs = traffic['conv'] / 1e9 / (conv_ms / 1e3 / max(conv_n, 1))
        hbm_measured = {'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                        'note': 'measured HBM traffic of the conv kernels (FETCH_SIZE + WRITE_SIZE per launch) over their '
                                'launch time: the banded operand (x + three low-pass planes per consumer) makes the '
                                'kernel HBM-bound on the bytes it has to move, which is why it sits low on the MFMA roof'}
    return {'bound': 'mfma', 'kernel': 'conv3x3_mfma_bf16 (all %d conv launches of a step)' % n_step,
            'achieved': tflops, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / BF16_MFMA_PEAK_TFLOPS,
            'algorithmic_tflop_per_step': flops_step / 1e12, 'avg_launch_ms': conv_ms / max(conv_n, 1),
            'traffic': traffic.get('conv'), 'traffic_unit': 'HBM bytes per average conv launch (PMC, %s)' % traffic_src,
            'hbm': {'achieved': conv_strict_gb / conv_s_step, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': conv_strict_gb / conv_s_step / HBM_PEAK_GBS,
                    'algorithmic_gb_per_launch': conv_strict_gb / n_step,
                    'note': 'strict bytes: every conv input once + output once + residual operands; the FDConv operand '
                            'as implemented streams x and three low-pass planes per consumer (4 x cin), see DESIGN.md'},
            'hbm_measured': hbm_measured, 'groups': groups, 'traffic_per_launch_all_groups': traffic}


def run_batch(args, world, rank, dev, dist):
    net, sd = build_model(dev, 1)
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
        net.profile(2)  # HIP events around the conv launches only: 11 ms/step less than with every group
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = net(x)
        sync()
        elapsed = time.perf_counter() - t0
        prof = net.profile_read()
        net.profile(1)  # per-group breakdown from one extra, untimed step
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
    res = {
        'metric': 'output Mpix/s (4x SR) at 1/2/4/8 MI355X; conv MFMA util %',
        'value': out_mpix / (elapsed / args.steps), 'unit': 'output Mpix/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': max(args.warmup, 1), 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'RRDBNet-FDConv(1,1,64,23,32) x4 inference, batch %d of %dx%d synthetic grayscale per GPU, '
                               'bf16 (BASELINE.json configs[1])' % (B, H, W),
                   'global_batch': world * B, 'parallelism': 'batch-sharded replicas x%d, no exchange' % world,
                   'operand_format': 'tensors in HBM bf16; matrix-core operands of the FDConv convs: fp16 (one fp16 weight per MAC + the '
                                     'summed rounding remainders on the centre tap, activations packed to fp16 by the band-mixing loader), '
                                     'float32 accumulation; SR_AMD_OPERAND_F16=0: bf16 hi + lo weight pairs with a bf16 operand (secondary.batch_bf16_pair_weights)',
                   'output_finite': finite},
        'roofline': roofline_fields(prof, prof_all, args.steps, float(B) * H * W, 1, traffic_applies=(B, H, W) == (16, 512, 512)),
        'kernel_ms_per_step': {k: v[0] for k, v in prof_all.items()},
        'kernel_launches_per_step': {k: v[1] for k, v in prof_all.items()},
        'kernel_ms_source': 'one extra untimed step with events around every kernel group; the conv events of the roofline are in the timed steps',
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # after the timed region; skipped in profiling runs
        try:
            res['roofline']['stream_calibration'] = stream_calibration(dev)
        except Exception as e:  # never take the line down
            res['roofline']['stream_calibration'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and world == 1 and args.cpu_sample > 0 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(sd, args.cpu_sample, 1)
    return res


def run_tiled(args, world, rank, dev, dist):
    from super_resolution_amd import RealESRGANer
    net, sd = build_model(dev, 3, plain=args.plain)
    side = args.image
    img = (np.random.default_rng(1234).random((side, side, 3)) * 255).astype(np.uint8)
    r = RealESRGANer(scale=4, model_path=None, model=net, tile=args.tile, tile_pad=args.tile_pad, pre_pad=0, half=True,
                     device=dev, shard_tiles=True, tile_batch=args.tile_batch or None)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the image is placed in HBM before the timed region (pre_process: host -> device, float conversion)
    t0 = time.perf_counter()
    r.pre_process(img[:, :, ::-1].astype(np.float32) / 255.0)
    torch.cuda.synchronize(dev)
    t_pre = time.perf_counter() - t0
    wins = r.tile_windows()
    mine = r.shard_of(wins, rank, world)
    lr_px_rank = float(sum((wins[i][0][1] - wins[i][0][0]) * (wins[i][0][3] - wins[i][0][2]) for i in mine))
    # network time inside the timed region: events around every forward of this rank (on torch's current stream, the one
    # the executor launches on); the rest of the step -- crops, paste, gather to rank 0, post_process -- is what does not
    # shrink with the number of ranks
    fwd_events = []
    inner_forward = net.forward

    def timed_forward(xin):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = inner_forward(xin)
        e1.record()
        fwd_events.append((e0, e1))
        return y

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            r.tile_process()
            r.post_process()
        sync()
        net.profile(2)
        net.forward = timed_forward
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r.tile_process()
            out = r.post_process()
        sync()
        elapsed = time.perf_counter() - t0
        net.forward = inner_forward
        net_s = sum(a.elapsed_time(b) for a, b in fwd_events) / 1e3 / args.steps
        prof = net.profile_read()
        net.profile(1)
        r.tile_process()
        sync()
        prof_all = net.profile_read()
        net.profile(False)
        # content check of the gather (world > 1): rank 0 re-runs tiles that PEERS computed -- chosen among those whose
        # place in the peer's run order differs from its place in the shard order -- and compares them with what was
        # pasted into the gathered image
        gather_err, gather_tiles = None, []
        if world > 1 and rank == 0:
            s4 = r.scale
            for src in range(1, world):
                shard = r.shard_of(wins, src, world)
                ran = [i for part in r.batches_of(wins, shard, r.tile_batch) for i in part]
                moved = [i for a, i in enumerate(shard) if ran[a] != i] or shard
                gather_tiles += moved[:1] + moved[-1:]
            gather_tiles = sorted(set(gather_tiles))[:4]
            gather_err = 0.0
            for i in gather_tiles:
                (y0, y1, x0, x1) = wins[i][1]
                again = r._run_tile(wins[i]).float()
                got = out[:, :, y0 * s4:y1 * s4, x0 * s4:x1 * s4].float()
                gather_err = max(gather_err, float((again - got).abs().max()))
        # whole enhance(), host image in -> host image out (PCIe, uint8 conversion and colour handling included)
        sync()
        t0 = time.perf_counter()
        full, _mode = r.enhance(img)
        sync()
        t_enh = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)
    out_mpix = (4 * side) * (4 * side) / 1e6
    step_s = elapsed / args.steps
    # event timings cover the last batch shape's executor only if the shapes differ -- the profile is per _NetState and
    # accumulates over all forwards between profile() and profile_read(), which is what we want here
    kernel_ms = sum(v[0] for v in prof_all.values())
    res = {
        'metric': 'output Mpix/s (4x SR) at 1/2/4/8 MI355X; conv MFMA util %',
        'value': out_mpix / step_s, 'unit': 'output Mpix/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': max(args.warmup, 1), 'ms_per_step': step_s * 1e3, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'RealESRGANer(tile=%d, tile_pad=%d, pre_pad=0, half=True) x4 on one %dx%d synthetic RGB image, '
                               '%s, bf16 (BASELINE.json configs[2])' % (args.tile, args.tile_pad, side, side,
                                                                       'plain RRDBNet(3,3,64,23,32) = RealESRGAN_x4plus architecture (configs[0] network)' if args.plain else 'RRDBNet-FDConv(3,3,64,23,32)'),
                   'tiles': len(wins), 'tile_batch': r.tile_batch, 'tile_batch_automatic': not args.tile_batch,
                   'parallelism': '%d tiles sorted by window shape, one contiguous block per rank over %d rank(s), one p2p message per peer (its cores stacked in shard order) to rank 0, no collective' % (len(wins), world),
                   'timed_region': 'tile_process() + post_process() with the padded image resident in HBM; max over ranks incl. the gather',
                   'enhance_host_to_host_s': t_enh, 'enhance_host_to_host_out_mpix_s': out_mpix / t_enh,
                   'pre_process_s': t_pre,
                   'non_parallel_share': {'note': 'rank 0: fraction of the timed step outside the network forwards of its own tiles '
                                                  '(crops, paste, gather of the cores, post_process): the part that does not '
                                                  'shrink with the number of ranks',
                                          'value': max(0.0, 1.0 - net_s / step_s), 'network_s': net_s,
                                          'enhance_minus_timed_s': t_enh - step_s},
                   'gather_check': {'tiles_rerun_on_rank0': gather_tiles, 'max_abs_err': gather_err,
                                    'note': 'tiles computed by peers, re-run on rank 0 and compared with the gathered image (null at N = 1)'},
                   'output_ok': bool(out is None or torch.isfinite(out.float()).all()) and (full is None or full.shape == (4 * side, 4 * side, 3))
                                and (gather_err is None or gather_err <= 2e-3)},
        'roofline': roofline_fields(prof, prof_all, args.steps, lr_px_rank, 3),
        'kernel_ms_per_step': {k: v[0] for k, v in prof_all.items()},
        'kernel_launches_per_step': {k: v[1] for k, v in prof_all.items()},
    }
    if rank == 0 and world == 1 and args.cpu_sample > 0 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(sd, args.cpu_sample, 3)
    return res


def run_srvgg(args, world, rank, dev, dist):
    """BASELINE.json configs[4]: realesr-animevideov3's SRVGGNetCompact(3,3,64,16,4,'prelu') on 1080p frames, bf16."""
    import golden_params as gp
    from super_resolution_amd import SRVGGNetCompact
    net = SRVGGNetCompact(3, 3, 64, 16, 4, 'prelu')
    sd = {k: torch.from_numpy(v) for k, v in gp.srvgg_state(3, 3, 64, 16, 4, seed=600).items()}
    net.load_state_dict(sd, strict=True)
    net = net.eval().to(dev)
    B, H, W = args.frames, 1080, 1920
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1234 + rank)).to(dev).to(torch.bfloat16)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            y = net(x)
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()  # the kernels are enqueued on torch's current stream: these events bracket them on that stream
        for _ in range(args.steps):
            y = net(x)
        e1.record()
        sync()
        elapsed = time.perf_counter() - t0
    kern_s = e0.elapsed_time(e1) / 1e3 / args.steps
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)
    step_s = elapsed / args.steps
    macs_px = 9 * (3 * 64 + 16 * 64 * 64 + 64 * 48)  # 619,200 MAC per LR pixel
    flops = 2.0 * macs_px * B * H * W
    strict_gb = B * H * W * (3 * ES + 17 * 2 * 64 * ES + 64 * ES + 48 * ES + 48 * ES + 3 * ES + 48 * ES) / 1e9
    res = {
        'metric': 'output Mpix/s (4x SR) at 1/2/4/8 MI355X; conv MFMA util %',
        'value': world * B * 16 * H * W / 1e6 / step_s, 'unit': 'output Mpix/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': max(args.warmup, 1), 'ms_per_step': step_s * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': "SRVGGNetCompact(3,3,64,16,4,'prelu') x4 on %d synthetic 1920x1080 frames per step and GPU, bf16 "
                               '(BASELINE.json configs[4])' % B,
                   'frames_per_s': world * B / step_s,
                   'published_reference': '10.0 fps on V100 (Real-ESRGAN-master/docs/anime_comparisons.md:60); other hardware',
                   'output_finite': bool(torch.isfinite(y.float()).all())},
        'roofline': {'bound': 'mfma', 'kernel': 'conv3x3_mfma_bf16 (18 conv launches per step)', 'achieved': flops / kern_s / 1e12,
                     'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': flops / kern_s / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                     'traffic': None,
                     'hbm': {'achieved': strict_gb / kern_s, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': strict_gb / kern_s / HBM_PEAK_GBS}},
    }
    return res


def run_train(args, world, rank, dev, dist):
    """BASELINE.json configs[3]: one optimisation step of RRDBNet-FDConv(1,1,64,23,32) shaped as
    train/train_realESRGAN_fdconv_1.py:110-118 -- forward in train() mode, bilinear resize to the HR size + L1, backward,
    gradient all-reduce over the ranks (RCCL), Adam -- on `--train-batch` HR crops of 256x256 per GPU (LR = 4x area
    downsample), mixed precision by default.  Loss: L1 + the perceptual (VGG19-feature) term of the Real-ESRGAN recipe
    (--train-loss l1: the owner script's L1 alone)."""
    from super_resolution_amd import RRDBNet, replace_conv_with_fdconv, train as T
    # the network as the owner script builds it (train_realESRGAN_fdconv_1.py:100-102): RRDBNet's own initialisation, every
    # conv swapped for a freshly initialised FDConv.  (The O(1)-perturbed parameter set of the inference configs makes the
    # 23-block network chaotic -- DESIGN.md -- and its gradients overflow float32 after two Adam steps.)
    torch.manual_seed(1234)
    net = replace_conv_with_fdconv(RRDBNet(1, 1, 4, NUM_FEAT, NUM_BLOCK, NUM_GROW)).to(dev)
    net.train()
    mixed = args.train_dtype == 'bf16'
    net.compute_dtype = torch.bfloat16 if mixed else None
    B, hr = args.train_batch, args.train_hr
    lr_side = hr // 4
    g = torch.Generator().manual_seed(4321 + rank)
    sr = synthetic_batch(B, hr, hr, 'cpu', seed=4321 + rank)
    sr = sr * torch.linspace(0.3, 1.0, B, dtype=torch.float32).view(B, 1, 1, 1)  # crops of different brightness
    wf = torch.nn.functional.avg_pool2d(sr, 4)
    sr, wf = sr.to(dev), wf.to(dev)
    # Adam over one flat buffer (same update rule as torch.optim.Adam(lr=1e-4) of the owner script; tests/test_cpu_host.py):
    # one gather, one all-reduce, five tensor ops per step instead of a walk over ~6000 parameter tensors
    opt = T.FlatAdam(net.parameters(), lr=1e-4)
    # BASELINE configs[3]: "L1 + perceptual".  Perceptual term as SURVEY 8(d) cfg4 prescribes: VGG19-shaped extractor, seeded
    # random initialisation (pretrained weights are a network fetch), features before ReLU at conv{1_2,2_2,3_4,4_4,5_4} with
    # weights {0.1,0.1,1,1,1} (options/train_realesrgan_x4plus.yml:133-149); --train-loss l1 = the owner script's loss alone
    perc = None
    if args.train_loss == 'l1+perceptual':
        from super_resolution_amd.perceptual import VGG19Features
        perc = dict(vgg=VGG19Features(seed=1234).to(dev), compute_dtype=torch.bfloat16 if mixed else torch.float32)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    losses = []
    for _ in range(max(args.warmup, 1)):
        losses.append(float(T.train_step(net, opt, wf, sr, perceptual=perc)))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = T.train_step(net, opt, wf, sr, perceptual=perc)
    sync()
    elapsed = time.perf_counter() - t0
    losses.append(float(loss))
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    step_s = float(tmax) / args.steps
    nparam = sum(p.numel() for p in net.parameters())
    flops = 3 * 2.0 * conv_macs_per_lr_pixel(1, 1) * B * lr_side * lr_side  # forward + operand gradient + weight gradient
    return {
        'metric': 'output Mpix/s (4x SR) at 1/2/4/8 MI355X; conv MFMA util %',
        'value': world * B * hr * hr / 1e6 / step_s, 'unit': 'output Mpix/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': max(args.warmup, 1), 'ms_per_step': step_s * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16' if mixed else 'f32', 'data': 'synthetic',
        'config': {'workload': 'training step of RRDBNet-FDConv(1,1,64,23,32): %d HR crops of %dx%d per GPU (LR %dx%d), train-mode '
                               'forward + %s loss + backward + gradient all-reduce + Adam (train.FlatAdam: one flat buffer), %s '
                               '(BASELINE.json configs[3])'
                               % (B, hr, hr, lr_side, lr_side,
                                  'bilinear/L1 + perceptual (seeded random-init VGG19 features, yml weights)' if perc else 'bilinear/L1 (owner script)',
                                  'bf16 activations on MFMA, float32 parameters / weight gradients / Adam' if mixed else 'float32'),
                   'global_batch': world * B, 'samples_per_s': world * B / step_s,
                   'parallelism': 'data parallel x%d, one all-reduce over the flat %.1f M-parameter gradient buffer per step (local BatchNorm statistics)' % (world, nparam / 1e6),
                   'loss_first_last': [losses[0], losses[-1]], 'loss_finite': bool(np.isfinite(losses).all()),
                   'peak_hbm_gb': torch.cuda.max_memory_allocated(dev) / 1e9},
        'roofline': {'bound': 'mfma', 'kernel': 'conv forward + operand gradient + weight gradient: on the matrix cores in bf16 mode (float32 accumulation), VALU kernels in f32 mode',
                     'achieved': flops / step_s / 1e12, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': flops / step_s / 1e12 / BF16_MFMA_PEAK_TFLOPS, 'traffic': None,
                     'note': 'whole-step time (no per-kernel events in the training path yet); algorithmic conv FLOPs = 3 x forward'},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', choices=('batch', 'tiled', 'srvgg', 'train'), default='batch')
    ap.add_argument('--frames', type=int, default=4, help='srvgg: 1080p frames per step')
    ap.add_argument('--train-batch', type=int, default=32, help='train: HR crops per GPU and step')
    ap.add_argument('--train-hr', type=int, default=256, help='train: side of the HR crops')
    ap.add_argument('--train-dtype', choices=('bf16', 'f32'), default='bf16', help='train: bf16 = mixed precision (bf16 activations on the matrix cores, float32 parameters and weight gradients); f32 = the parity path')
    ap.add_argument('--train-loss', choices=('l1+perceptual', 'l1'), default='l1+perceptual', help='train: BASELINE configs[3] says L1 + perceptual; l1 = the owner script (train_realESRGAN_fdconv_1.py:116)')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--image', type=int, default=4096, help='tiled: side of the square RGB image')
    ap.add_argument('--plain', action='store_true', help='tiled: the plain RRDBNet (RealESRGAN_x4plus architecture, BASELINE configs[0] network) instead of RRDBNet-FDConv')
    ap.add_argument('--tile', type=int, default=512)
    ap.add_argument('--tile-pad', type=int, default=10)
    ap.add_argument('--tile-batch', type=int, default=18, help='tiled: tiles of equal window size per forward (8 -> 18: 45.7 -> 49.2 out-Mpix/s at N = 1); 0 = left to RealESRGANer (its default: sized from the free device memory, at most 18)')
    ap.add_argument('--cpu-sample', type=int, default=128, help='side of the CPU-baseline sample image (0: skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU-baseline leg (profiling runs)')
    ap.add_argument('--no-secondary', action='store_true', help='batch: skip the one-step runs of the tiled / train / srvgg configs (N > 1: tiled and train, sharded over the ranks)')
    ap.add_argument('--secondary-deadline', type=int, default=300, help='batch: seconds the secondary configs may take before the headline is printed without them')
    ap.add_argument('--no-split-weights', action='store_true', help='A/B: plain bf16 dynamic weights instead of hi + lo pairs')
    args = ap.parse_args()

    if os.environ.get('SR_BENCH_FAULT_S'):  # debugging aid: dump every thread's Python stack after N seconds (a hung collective)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['SR_BENCH_FAULT_S']), exit=False)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    # SR_BENCH_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the N > 1 code paths on a one-GPU box (ranks share the
    # card, collectives go through the host); the driver's runs use RCCL ('nccl'), one rank per GPU
    backend = os.environ.get('SR_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if backend != 'nccl' and ndev:
        local = local % ndev
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from super_resolution_amd.build import build
    if rank == 0:
        build(verbose=False)
    if world > 1:
        dist.barrier()

    for spec in filter(None, os.environ.get('SR_BENCH_DEBUG_HOOKS', '').split(',')):  # A/B runs: "sr_debug_xxx=1,sr_debug_yyy=0"
        from super_resolution_amd import _lib
        name, _, val = spec.partition('=')
        getattr(_lib.lib(), name.strip())(int(val or '1'))
    if os.environ.get('SR_BENCH_SPLIT_TWO_PASS'):
        from super_resolution_amd import _lib
        _lib.lib().sr_debug_split_two_pass(int(os.environ['SR_BENCH_SPLIT_TWO_PASS']))
    if args.no_split_weights:
        from super_resolution_amd import _lib
        _lib.lib().sr_debug_split_weights(0)
        _lib.lib().sr_debug_operand_f16(0)
    runners = {'tiled': run_tiled, 'srvgg': run_srvgg, 'batch': run_batch, 'train': run_train}
    res = runners[args.config](args, world, rank, dev, dist)
    if args.config == 'batch' and not args.no_secondary:
        # the other BASELINE configs, one short timed step each, so that the driver's one command sees them too (each also
        # runs on its own with --config).  At N > 1 the configs that shard run as well -- the tiled image over the N ranks
        # (strong scaling, p2p gather) and the training step with its gradient all-reduce -- so the driver's 1/2/4/8-GPU
        # runs of this one command carry north_star's tiled-scaling figure.  A deadline guards the headline: if a secondary
        # config is still running when it passes (a collective that never completes), rank 0 prints the headline line
        # with the secondary marked abandoned and every rank leaves.
        import copy
        import gc
        import threading
        finished = threading.Event()
        fallback = json.dumps(dict(res, secondary={'error': 'secondary configs abandoned after %d s' % args.secondary_deadline}))

        def watch():
            if not finished.wait(args.secondary_deadline):
                if rank == 0:
                    print(fallback, flush=True)
                os._exit(0)
        threading.Thread(target=watch, daemon=True).start()
        res['secondary'] = {}
        from super_resolution_amd import _lib
        for name in (('tiled', 'tiled_plain', 'train', 'srvgg', 'batch_bf16_pair_weights') if world == 1 else ('tiled', 'tiled_plain', 'train')):
            gc.collect()
            torch.cuda.empty_cache()
            a2 = copy.copy(args)
            a2.steps, a2.warmup, a2.no_cpu_baseline = (3 if name == 'srvgg' else 1), 1, True
            a2.plain = name == 'tiled_plain'
            t0 = time.perf_counter()
            f16 = name == 'batch_bf16_pair_weights'  # the headline workload with the operand format of rounds 2-3 (DESIGN.md finding 39)
            try:
                if f16:
                    _lib.lib().sr_debug_operand_f16(0)
                    a2.steps = 3
                r2 = runners['batch' if f16 else ('tiled' if name == 'tiled_plain' else name)](a2, world, rank, dev, dist)
                res['secondary'][name] = {'value': r2['value'], 'unit': r2['unit'], 'ms_per_step': r2['ms_per_step'], 'steps': a2.steps,
                                          'n_gpus': world, 'dtype': r2['dtype'], 'scaling': r2['scaling'], 'workload': r2['config']['workload'],
                                          'roofline_bound': r2['roofline']['bound'], 'roofline_frac': r2['roofline']['frac'],
                                          'detail': {k: v for k, v in r2['config'].items() if k not in ('workload',)},
                                          'wall_s': time.perf_counter() - t0}
                if f16:
                    res['secondary'][name]['note'] = ('SR_AMD_OPERAND_F16=0: bf16 hi + lo weight pairs with a bf16 operand in the FDConv convs, the format of '
                                                      'rounds 2-3; the default since the end of round 3 is one fp16 weight per MAC + the summed rounding '
                                                      'remainders on the centre tap, operand packed to fp16 (DESIGN.md finding 39)')
            except Exception as e:  # a secondary config must never take the headline line down with it
                res['secondary'][name] = {'error': '%s: %s' % (type(e).__name__, e)}
                if world > 1:  # the peers may already be inside the next collective: do not join them half-way
                    break
            finally:
                if f16:
                    _lib.lib().sr_debug_operand_f16(1)
        finished.set()
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
