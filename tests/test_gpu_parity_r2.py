"""GPU parity at the depth and at the window sizes the BASELINE configs run (round-2 additions).

  * 23-block RRDBNet-FDConv (config 2's network) in fp32 against the CPU oracle, with the module constructors' own
    seeded initialisation ("realistic": contractive like a trained checkpoint) and with the O(1)-perturbed set;
  * north_star's bf16 criterion (PSNR delta <= 0.05 dB) on the realistic 23-block network;
  * config 3's windows: RealESRGANer(tile=512, tile_pad=10) produces 522- and 532-pixel windows; an FDConv network
    runs them tile by tile against the oracle, and the low-pass planes are checked at 1x, 2x and 4x of those sizes;
  * the dense-block gate accumulation outside the default nf=64 / gc=32 shape, odd sizes in bf16, and
    tile batching with per-sample dynamic weights.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

import golden_params as gp
from oracle import fdconv_ref as fr
from oracle.rrdbnet_ref import rrdbnet_forward

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4       # north_star: fp32 max-abs-err <= 1e-4
PSNR_DELTA_DB = 0.05  # north_star: bf16 PSNR delta <= 0.05 dB


def t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 20 * math.log10(1.0 / math.sqrt(mse))


@pytest.fixture(scope='module')
def sra():
    import super_resolution_amd as m
    from super_resolution_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return m


def realistic_net(sra, args, seed):
    """The network as the owner script builds it (train/train_realESRGAN_fdconv_1.py:100-101): RRDBNet's own init
    (kaiming x 0.1), every conv replaced by a freshly initialised FDConv.  Returns (GPU net, CPU state_dict)."""
    torch.manual_seed(seed)
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(*args))
    # The constructor leaves the FBM gates and KSM_Local at 1e-6 (FBM(x) = x, hr = 1): give those an effect.  The
    # KSM_Global heads stay as initialised -- measured on the CPU oracle: a 0.003 perturbation of spatial_fc / kernel_fc /
    # channel_fc already makes the 23-block network chaotic (fp32 and fp64 oracle disagree by 100 %), while this set
    # keeps the fp32-vs-fp64 floor at ~1e-6 of the output range, like a trained checkpoint.
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if any(s in name for s in ('KSM_Local.conv.weight', 'freq_weight_conv_list')) and p.dim() > 1:
                p.add_(0.003 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net.eval().cuda(), sd


def test_23_blocks_fp32_vs_oracle_realistic_weights(sra):
    net, sd = realistic_net(sra, (1, 1, 4, 64, 23, 32), seed=1)
    for shape, seed in (((1, 1, 64, 64), 80), ((2, 1, 96, 80), 81)):
        x = torch.from_numpy(gp.image_input(shape, seed))
        with torch.no_grad():
            y = net(x.cuda()).cpu()
            ref = rrdbnet_forward(sd, x)
        err = (y - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item())  # north_star's 1e-4 is for [0, 1] images; random-init outputs reach ~15
        print('23-block RRDBNet-FDConv fp32 %s: max|gpu - oracle| = %.3e (|y| max %.3f)' % (shape, err, ref.abs().max().item()))
        assert err < FP32_TOL * scale, (shape, err)


def test_23_blocks_fp32_vs_oracle_perturbed_weights(sra):
    """The O(1)-perturbed parameter set of the bench: 349 stacked layers with data-dependent weights amplify rounding, so
    the yardstick is the oracle itself -- fp32 oracle against fp64 oracle gives the noise floor of ANY fp32 evaluation
    of this network, and the GPU must stay within a small multiple of it (and within 1e-4 when the floor allows)."""
    sd = t(gp.rrdbnet_state(1, 1, 64, 23, 32, seed=1234))
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, 23, 32))
    net.load_state_dict(sd, strict=True)
    net = net.eval().cuda()
    x = torch.from_numpy(gp.image_input((1, 1, 64, 64), 82))
    with torch.no_grad():
        y = net(x.cuda()).cpu()
        ref32 = rrdbnet_forward(sd, x)
        ref64 = rrdbnet_forward(sd, x.double(), dtype=torch.float64)
    floor = (ref32.double() - ref64).abs().max().item()
    err = (y.double() - ref64).abs().max().item()
    scale = ref64.abs().max().item()
    print('perturbed 23-block net: |gpu32 - oracle64| = %.3e, |oracle32 - oracle64| = %.3e, |y| max %.3e' % (err, floor, scale))
    assert err <= max(FP32_TOL, 8.0 * floor), (err, floor)


def test_23_blocks_bf16_psnr_delta_realistic_weights(sra):
    """north_star's bf16 criterion at config depth (23 blocks, trunk on the MFMA low-pass).

    Finding of round 2 (finding 15): on this network ANY evaluation that rounds the per-sample weights to
    bf16 -- the oracle with bf16 operands included -- ends 15-20 dB from the fp32 output, because a rounded weight is an
    error coherent over the whole image and 349 data-dependent layers amplify it; bf16 activations alone cost 50 dB.
    The conv therefore multiplied by the weights as hi + lo bf16 pairs (rounds 2-3) and multiplies, since the end of round 3, by one
    fp16 weight plus the summed rounding remainders on the centre tap (DESIGN.md finding 39; `net(...)` below runs that default).
    Checked here: (1) the PSNR-delta criterion on
    the range-normalised output, (2) the direct bf16-vs-fp32 PSNR, (3) that the extra weight bits are what buys it (A/B through
    the debug hooks: single bf16 weights), (4) parity with the oracle's own bf16-storage emulation (same rounding points, hi + lo weights)."""
    from super_resolution_amd import _lib
    import oracle.fdconv_ref as fr_mod
    net, sd = realistic_net(sra, (1, 1, 4, 64, 23, 32), seed=2)
    x = torch.from_numpy(gp.image_input((1, 1, 64, 64), 83))
    real = fr_mod._bf16

    def hilo(v):  # weights (5-D per-sample, 4-D 3x3 of the thin plain layers) keep hi + lo, everything else one bf16 rounding
        if v.dim() == 5 or (v.dim() == 4 and v.shape[-1] == 3):
            hi = real(v)
            return hi + real(v - hi)
        return real(v)

    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
        fr_mod._bf16 = hilo
        try:
            emu = rrdbnet_forward(sd, x, emulate_bf16=True)
        finally:
            fr_mod._bf16 = real
        yb = net(x.cuda().to(torch.bfloat16)).float().cpu()
        _lib.lib().sr_debug_split_weights(0)
        _lib.lib().sr_debug_operand_f16(0)   # the default operand format (fp16 weight + remainder plane) takes precedence over the split
        try:
            yb_plain = net(x.cuda().to(torch.bfloat16)).float().cpu()
        finally:
            _lib.lib().sr_debug_split_weights(1)
            _lib.lib().sr_debug_operand_f16(1)
    target = torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
    # random-init outputs are not in [0, 1]: normalise by the fp32 output's own range before the PSNR
    lo, hi = ref.min(), ref.max()
    n = lambda v: (v - lo) / (hi - lo)
    p_ref, p_bf = psnr(n(ref).clamp(0, 1), target), psnr(n(yb).clamp(0, 1), target)
    d_gpu, d_plain, d_emu = psnr(n(yb), n(ref)), psnr(n(yb_plain), n(ref)), psnr(n(emu), n(ref))
    print('23-block bf16: PSNR vs target fp32 oracle %.3f dB, bf16 GPU %.3f dB; direct bf16-vs-fp32: GPU %.1f dB, GPU with plain '
          'bf16 weights %.1f dB, oracle bf16-storage emulation %.1f dB' % (p_ref, p_bf, d_gpu, d_plain, d_emu))
    assert abs(p_ref - p_bf) <= PSNR_DELTA_DB
    assert d_gpu > 40.0
    assert d_gpu > d_plain + 10.0
    assert d_gpu > d_emu - 3.0


def test_fp16_operand_mode_holds_the_psnr_budget(sra, capfd):
    """The two operand formats of the FDConv convs in bf16 networks (DESIGN.md finding 39): the default -- ONE fp16 weight per MAC
    plus the summed rounding remainders on the centre tap, operand packed to fp16 by the loader -- and the bf16 hi + lo pair with
    a bf16 operand of rounds 2-3 (SR_AMD_OPERAND_F16=0 / sr_debug_operand_f16(0); `y_def` below).
    north_star's criterion (PSNR delta <= 0.05 dB) on the budget test's ill-conditioned 64 x 64 pair and on the 512 x 512
    fixture pair, the distance from the default path, and that the mode really is a different kernel (outputs differ)."""
    from super_resolution_amd import _lib
    import make_golden_r3 as mg3
    lib = _lib.lib()

    def both(net, x):
        with torch.no_grad():
            lib.sr_debug_operand_f16(0)
            try:
                y_def = net(x.cuda().to(torch.bfloat16)).float().cpu()   # bf16 hi + lo pairs
            finally:
                lib.sr_debug_operand_f16(1)
            y_f16 = net(x.cuda().to(torch.bfloat16)).float().cpu()       # the default format
        return y_def, y_f16

    net, sd = realistic_net(sra, (1, 1, 4, 64, 23, 32), seed=2)
    x = torch.from_numpy(gp.image_input((1, 1, 64, 64), 83))
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
    y_def, y_f16 = both(net, x)
    assert torch.isfinite(y_f16).all() and not torch.equal(y_def, y_f16)
    target = torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
    lo, hi = ref.min(), ref.max()
    n = lambda v: (v - lo) / (hi - lo)  # noqa: E731
    p_ref = psnr(n(ref).clamp(0, 1), target)
    d64 = abs(psnr(n(y_f16).clamp(0, 1), target) - p_ref)
    print('fp16 operands, 23 blocks 64x64: PSNR delta %.4f dB (bf16 pairs %.4f dB); direct vs fp32 oracle %.1f dB (bf16 pairs %.1f dB)'
          % (d64, abs(psnr(n(y_def).clamp(0, 1), target) - p_ref), psnr(n(y_f16), n(ref)), psnr(n(y_def), n(ref))))
    assert d64 <= PSNR_DELTA_DB and abs(psnr(n(y_def).clamp(0, 1), target) - p_ref) <= PSNR_DELTA_DB
    assert psnr(n(y_f16), n(ref)) > 40.0

    x = torch.from_numpy(gp.image_input((1, 1, 512, 512), 84))
    y_def, y_f16 = both(net, x)
    o = _oracle_512(sd, x, capfd)
    lo, hi = o['ref32_min'], o['ref32_max']
    n = lambda v: (v - lo) / (hi - lo)  # noqa: E731
    a, b = mg3.CROP
    pick = lambda v: torch.cat([v[0, 0, ::mg3.STRIDE, ::mg3.STRIDE].reshape(-1), v[0, 0, a:b, a:b].reshape(-1)])  # noqa: E731
    ref = torch.cat([o['ref32_lattice'].reshape(-1), o['ref32_crop'].reshape(-1)]).float()
    target = pick(torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1))
    p_ref = psnr(n(ref).clamp(0, 1), target)
    d512 = abs(psnr(n(pick(y_f16)).clamp(0, 1), target) - p_ref)
    print('fp16 operands, 23 blocks 512x512: PSNR delta %.4f dB (bf16 pairs %.4f dB); direct vs fp32 oracle %.1f dB (bf16 pairs %.1f dB)'
          % (d512, abs(psnr(n(pick(y_def)).clamp(0, 1), target) - p_ref), psnr(n(pick(y_f16)), n(ref)), psnr(n(pick(y_def)), n(ref))))
    assert torch.isfinite(y_f16).all()
    assert d512 <= PSNR_DELTA_DB
    assert psnr(n(pick(y_f16)), n(ref)) > 40.0


class _Heartbeat:
    """A dot on the REAL stdout every 30 s while a long CPU oracle runs (the GPU box's watchdog ends a command that
    prints nothing for three minutes; pytest holds captured output back until the test ends)."""

    def __init__(self, capfd):
        self.capfd = capfd

    def __enter__(self):
        import threading
        self._cm = self.capfd.disabled()
        self._cm.__enter__()
        self._stop = threading.Event()

        def beat():
            while not self._stop.wait(30.0):
                print('.', end='', flush=True)
        self._t = threading.Thread(target=beat, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join()
        return self._cm.__exit__(*exc)


def _oracle_512(sd, x, capfd):
    """Oracle outputs (float32 and float64) of the 23-block network on the 512 x 512 sample, on the pixel set the test
    compares: every 4th pixel of the 2048 x 2048 output plus the central 256 x 256 crop.  Read from the committed
    fixture tests/golden/rrdbnet_23block_512.npz (made by tests/golden/make_golden_r3.py from oracle/ -- the two forwards
    are ~6 min of CPU, which does not fit the GPU suite's time budget); SR_AMD_LIVE_ORACLE=1 runs them here instead."""
    import make_golden_r3 as mg3
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rrdbnet_23block_512.npz')
    if os.environ.get('SR_AMD_LIVE_ORACLE') == '1':
        with _Heartbeat(capfd):
            ref = rrdbnet_forward(sd, x)
            ref64 = rrdbnet_forward(sd, x.double(), dtype=torch.float64)
        a, b = mg3.CROP
        return {'ref32_min': ref.min().item(), 'ref32_max': ref.max().item(),
                'floor_full': (ref.double() - ref64).abs().max().item(),
                'ref64_lattice': ref64[0, 0, ::mg3.STRIDE, ::mg3.STRIDE], 'ref32_lattice': ref[0, 0, ::mg3.STRIDE, ::mg3.STRIDE],
                'ref64_crop': ref64[0, 0, a:b, a:b], 'ref32_crop': ref[0, 0, a:b, a:b]}
    g = np.load(path)
    assert int(g['stride']) == mg3.STRIDE and tuple(g['crop']) == mg3.CROP
    # the fixture belongs to ONE seeded state_dict: refuse it if the constructors' initialisation ever differs.  The sum is
    # not bit-identical across hosts (convert2dftweight's rfft2 rounds differently on different CPUs: 1.7e-8 relative
    # between the build container and the GPU box), a different seed or init moves it by percents
    chk = mg3.state_checksum(sd)
    assert abs(chk - float(g['state_checksum'])) <= 1e-6 * abs(chk), 'fixture was made for another initialisation: rerun make_golden_r3.py'
    out = {k: float(g[k]) for k in ('ref32_min', 'ref32_max', 'floor_full')}
    out.update({k: torch.from_numpy(g[k]) for k in ('ref64_lattice', 'ref32_lattice', 'ref64_crop', 'ref32_crop')})
    return out


def test_23_blocks_512x512_fp32_and_bf16_vs_oracle(sra, capfd):
    """BASELINE configs[1] at its own size, against the oracle (round-2 verdict: the bench-size instances -- the 512-point
    DFT-as-GEMM low-pass, the folded W-synthesis, the Cout = 64 split-weight conv -- were oracle-checked per op only, the
    23-block network at 64x64 / 96x80).  One sample of 1 x 1 x 512 x 512 through the 23-block realistic-weights network:
    float32 HIP against the float32 / float64 oracle (<= 1e-4 of the output range, or the float32 oracle's own distance
    from float64) and north_star's bf16 criterion (PSNR delta <= 0.05 dB) for the bf16 HIP output of the same sample --
    the exact kernels the bench runs, end to end.  Compared on every 4th output pixel plus the central 256 x 256 crop
    (330k of the 4.2M pixels; the oracle's values there are a committed fixture, see _oracle_512).

    Conditioning (round-3 finding, tools/diag_bf16.py): a random-init 23-block network is well-conditioned for most
    (weights, input) pairs -- bf16 lands 50 dB from fp32 -- and chaotic for some (19-26 dB), whatever the kernels: the
    VALU conv, the VALU gates and the FFT low-pass give the same figure as the matrix-core kernels on the same pair
    (seed 3 at 512x512: 20.4 dB on all four; seed 2 at 256x256: 19.1 / 15.6 / 19.1 / 24.1 dB).  The criterion is
    therefore held on a pair the float32 network itself is well-conditioned on, which the test checks first: the float32
    output must move by less than 1e-3 of its range when the input is disturbed by 1e-6."""
    import make_golden_r3 as mg3
    net, sd = realistic_net(sra, (1, 1, 4, 64, 23, 32), seed=2)
    x = torch.from_numpy(gp.image_input((1, 1, 512, 512), 84))
    with torch.no_grad():
        y = net(x.cuda()).cpu()
        noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(85)) * 1e-6
        y_n = net((x + noise).cuda()).cpu()
        yb = net(x.cuda().to(torch.bfloat16)).float().cpu()
    assert y.shape == yb.shape == (1, 1, 2048, 2048)
    o = _oracle_512(sd, x, capfd)
    lo, hi = o['ref32_min'], o['ref32_max']
    span = hi - lo
    sens = (y_n - y).abs().max().item() / span
    a, b = mg3.CROP
    pick = lambda v: torch.cat([v[0, 0, ::mg3.STRIDE, ::mg3.STRIDE].reshape(-1), v[0, 0, a:b, a:b].reshape(-1)])  # noqa: E731
    ref64 = torch.cat([o['ref64_lattice'].reshape(-1), o['ref64_crop'].reshape(-1)]).double()
    ref = torch.cat([o['ref32_lattice'].reshape(-1), o['ref32_crop'].reshape(-1)]).float()
    # float32 yardstick: at this size the network turns 1e-6 of input noise into ~5e-5 of its output range, and the float32
    # rounding of 349 layers alone moves the oracle by a few 1e-4 of the range (first GPU run of this test: 4.3e-4) -- so,
    # as in test_23_blocks_fp32_vs_oracle_perturbed_weights, the float64 oracle is the truth and the float32 oracle's own
    # distance from it is the floor of ANY float32 evaluation
    err = (pick(y).double() - ref64).abs().max().item()
    floor = max((ref.double() - ref64).abs().max().item(), o['floor_full'])
    n = lambda v: (v - lo) / (hi - lo)  # noqa: E731
    target = pick(torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1))
    p_ref, p_bf = psnr(n(ref).clamp(0, 1), target), psnr(n(pick(yb)).clamp(0, 1), target)
    direct = psnr(n(pick(yb)), n(ref))
    print('23-block 1x1x512x512: conditioning %.2e of range per 1e-6 of input noise; fp32 |gpu - oracle64| = %.3e, |oracle32 - oracle64| '
          '= %.3e, range %.3f; bf16 PSNR vs target %.3f dB against %.3f dB (fp32 oracle), direct bf16-vs-fp32 %.1f dB'
          % (sens, err, floor, span, p_bf, p_ref, direct))
    assert sens < 1e-3, 'the sample is ill-conditioned in float32 already: pick another pair'
    # Live oracle (same host, same state_dict; SR_AMD_LIVE_ORACLE=1): 3 x floor, observed 4.3e-4 against a floor of 4.8e-4.
    # Fixture: the state_dict is rebuilt on the test host and differs from the fixture host's in the last bits (see
    # _oracle_512); the network turns that into ~1.4e-4 of its range (observed 2.0e-3 in all) -- 8 x floor, the bound
    # test_23_blocks_fp32_vs_oracle_perturbed_weights uses.
    mult = 3.0 if os.environ.get('SR_AMD_LIVE_ORACLE') == '1' else 8.0
    assert err <= max(FP32_TOL * max(1.0, span), mult * floor), (err, floor)
    assert abs(p_ref - p_bf) <= PSNR_DELTA_DB
    assert direct > 40.0


# ---------------------------------------------------------------------------------------------------------------
# config 3 windows
# ---------------------------------------------------------------------------------------------------------------
def _lowpass_gpu(L, x_nchw, dtype, ups=0):
    """sr_fbm_lowpass on an NCHW float tensor -> (3, B, C, H, W) float, pooled (B, C); H, W of the OUTPUT planes."""
    lib = L.lib()
    B, Cc, Hs, Ws = x_nchw.shape
    H, W = Hs << ups, Ws << ups
    xn = x_nchw.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(H, W, C.byref(plan)))
    ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda')
    bands = [torch.empty(B, H, W, Cc, dtype=dtype, device='cuda') for _ in range(3)]
    pooled = torch.empty(B, Cc, device='cuda')
    L.check(lib.sr_fbm_lowpass(plan, L.dtype_code(dtype), B, L.view(xn), ups, L.view(bands[0]), L.view(bands[1]), L.view(bands[2]),
                               pooled.data_ptr(), Cc, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    lib.sr_fbm_plan_destroy(plan)
    return torch.stack([b.float().permute(0, 3, 1, 2) for b in bands]).cpu(), pooled.cpu()


@pytest.mark.parametrize('hw', [(522, 532), (532, 522), (1044, 1064), (2088, 2128)])
def test_lowpass_config3_window_sizes_fp32(sra, hw):
    from super_resolution_amd import _lib as L
    H, W = hw
    x = torch.from_numpy(gp.signed_input((1, 2, H, W), 90))
    got, pooled = _lowpass_gpu(L, x, torch.float32)
    ref = fr.fbm_lowpass(x.double(), torch.float64)
    err = (got.double() - ref).abs().max().item()
    print('low-pass %dx%d fp32: max err %.3e' % (H, W, err))
    assert err < 2e-5
    assert (pooled.double() - x.double().mean(dim=(2, 3))).abs().max().item() < 1e-5


@pytest.mark.parametrize('hw', [(1044, 1064), (2088, 2128)])
def test_lowpass_config3_upsampled_source(sra, hw):
    """conv_up1 / conv_up2 inputs: the planes of nearest2x(x) computed from the half-resolution source (ups = 1)."""
    from super_resolution_amd import _lib as L
    H, W = hw
    x = torch.from_numpy(gp.signed_input((1, 2, H // 2, W // 2), 91))
    got, pooled = _lowpass_gpu(L, x, torch.float32, ups=1)
    up = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    ref = fr.fbm_lowpass(up.double(), torch.float64)
    err = (got.double() - ref).abs().max().item()
    print('low-pass of nearest2x -> %dx%d fp32: max err %.3e' % (H, W, err))
    assert err < 2e-5
    assert (pooled.double() - x.double().mean(dim=(2, 3))).abs().max().item() < 1e-5


@pytest.mark.parametrize('hw', [(522, 532), (532, 532), (522, 522), (1044, 1064), (96, 80), (130, 78), (544, 34), (38, 540), (512, 530)])
def test_lowpass_config3_window_sizes_bf16(sra, hw):
    """bf16 trunk tensors (32-channel blocks) at the window sizes: whatever kernel family serves them, the planes must
    match the fp64 oracle of the bf16-rounded input to bf16 rounding."""
    from super_resolution_amd import _lib as L
    H, W = hw
    x = torch.from_numpy(gp.signed_input((2, 32, H, W), 92)).to(torch.bfloat16).float()
    got, pooled = _lowpass_gpu(L, x, torch.bfloat16)
    ref = fr.fbm_lowpass(x.double(), torch.float64)
    err = (got.double() - ref).abs().max().item()
    print('low-pass %dx%d bf16: max err %.3e' % (H, W, err))
    assert err < 3 * 2.0 ** -8
    assert (pooled.double() - x.double().mean(dim=(2, 3))).abs().max().item() < 2e-4


@pytest.mark.timeout(600)
def test_realesrganer_config3_windows_fdconv_vs_oracle(sra):
    """RealESRGANer(tile=512, tile_pad=10) on a 1536 x 1536 RGB image: nine tiles with 522- and 532-pixel windows through
    a 1-block RRDBNet-FDConv(3,3) in fp32; two of them (532 x 532 and 522 x 532) are checked against the oracle run on
    the same window (tile-by-tile parity: FDConv's statistics are per window)."""
    sd = t(gp.rrdbnet_state(3, 3, 64, 1, 32, seed=31))
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(3, 3, 4, 64, 1, 32))
    net.load_state_dict(sd, strict=True)
    img = gp.image_input((1536, 1536, 3), 93)
    r = sra.RealESRGANer(scale=4, model_path=None, model=net, tile=512, tile_pad=10, pre_pad=0, half=False)
    r.pre_process(img)
    r.tile_process()
    out = r.post_process()
    assert out.shape == (1, 3, 6144, 6144)
    wins = r.tile_windows()
    sizes = sorted({(w[0][1] - w[0][0], w[0][3] - w[0][2]) for w in wins})
    assert sizes == [(522, 522), (522, 532), (532, 522), (532, 532)]
    x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))[None]
    for ti in (4, 1):  # centre tile: 532 x 532; top middle: 522 x 532
        (py0, py1, px0, px1), (y0, y1, x0, x1), (oy, ox) = wins[ti]
        with torch.no_grad():
            ref = rrdbnet_forward(sd, x[:, :, py0:py1, px0:px1])
        ref = ref[:, :, oy * 4:(oy + y1 - y0) * 4, ox * 4:(ox + x1 - x0) * 4]
        got = out[:, :, y0 * 4:y1 * 4, x0 * 4:x1 * 4].cpu()
        err = (got - ref).abs().max().item()
        print('tile %d window %dx%d: max|gpu - oracle| = %.3e' % (ti, py1 - py0, px1 - px0, err))
        assert err < FP32_TOL, (ti, err)


def test_realesrganer_config3_windows_bf16_batched(sra):
    """The same tiling in bf16 with tile_batch = 4 against one tile at a time (bitwise: every statistic is per sample),
    and against the fp32 run (PSNR)."""
    sd = t(gp.rrdbnet_state(3, 3, 64, 1, 32, seed=31))
    img = gp.image_input((1034, 1546, 3), 94)
    outs = {}
    for tag, half, tb in (('fp32', False, 1), ('bf16', True, 1), ('bf16_b4', True, 4)):
        net = sra.replace_conv_with_fdconv(sra.RRDBNet(3, 3, 4, 64, 1, 32))
        net.load_state_dict(sd, strict=True)
        r = sra.RealESRGANer(scale=4, model_path=None, model=net, tile=512, tile_pad=10, pre_pad=0, half=half, tile_batch=tb)
        r.pre_process(img)
        r.tile_process()
        outs[tag] = r.post_process().float().cpu()
    assert torch.equal(outs['bf16'], outs['bf16_b4'])
    p = psnr(outs['bf16'], outs['fp32'])
    print('config-3 windows, 1-block FDConv net: bf16 vs fp32 PSNR %.1f dB' % p)
    assert p > 40.0


# ---------------------------------------------------------------------------------------------------------------
# gate accumulation outside nf = 64 / gc = 32, odd sizes in bf16, tile batching with dynamic weights
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('nf,gc', [(64, 16), (128, 32), (80, 32), (32, 48)])
def test_bf16_network_other_widths_vs_oracle(sra, nf, gc):
    sd = t(gp.rrdbnet_state(1, 1, nf, 1, gc, seed=40 + nf + gc))
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, nf, 1, gc))
    net.load_state_dict(sd, strict=True)
    net = net.eval().cuda()
    x = torch.from_numpy(gp.image_input((2, 1, 32, 32), 95))
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
        y32 = net(x.cuda()).cpu()
        yb = net(x.cuda().to(torch.bfloat16)).float().cpu()
    err = (y32 - ref).abs().max().item()
    p = psnr(yb, ref)
    print('nf=%d gc=%d: fp32 err %.3e, bf16 vs fp32 oracle PSNR %.1f dB' % (nf, gc, err, p))
    assert err < FP32_TOL
    assert p > 35.0


def test_bf16_network_odd_size_vs_oracle(sra):
    sd = t(gp.rrdbnet_state(1, 1, 64, 2, 32, seed=12))
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, 2, 32))
    net.load_state_dict(sd, strict=True)
    net = net.eval().cuda()
    x = torch.from_numpy(gp.image_input((1, 1, 45, 70), 96))
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
        yb = net(x.cuda().to(torch.bfloat16)).float().cpu()
    p = psnr(yb, ref)
    print('2-block bf16 at 45x70: PSNR vs fp32 oracle %.1f dB' % p)
    assert p > 35.0


def test_tile_batching_with_dynamic_weights(sra):
    """RealESRGANer tiling on a network with active FDConv layers: tile_batch > 1 equals one-by-one exactly."""
    sd = t(gp.rrdbnet_state(3, 3, 64, 1, 32, seed=33))
    img = gp.image_input((100, 132, 3), 97)
    outs = []
    for tb in (1, 3):
        net = sra.replace_conv_with_fdconv(sra.RRDBNet(3, 3, 4, 64, 1, 32))
        net.load_state_dict(sd, strict=True)
        r = sra.RealESRGANer(scale=4, model_path=None, model=net, tile=48, tile_pad=8, pre_pad=0, half=False, tile_batch=tb)
        r.pre_process(img)
        r.tile_process()
        outs.append(r.post_process().cpu())
    assert torch.equal(outs[0], outs[1])
    x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))[None]
    wins = r.tile_windows()
    (py0, py1, px0, px1), (y0, y1, x0, x1), (oy, ox) = wins[4]
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x[:, :, py0:py1, px0:px1])[:, :, oy * 4:(oy + y1 - y0) * 4, ox * 4:(ox + x1 - x0) * 4]
    assert (outs[0][:, :, y0 * 4:y1 * 4, x0 * 4:x1 * 4] - ref).abs().max().item() < FP32_TOL


def test_automatic_tile_batch_equals_one_by_one_and_respects_memory(sra, monkeypatch):
    """tile_batch left to the library (the default): the batch is sized from the executor's workspace formula and the free
    device memory, the image is bit for bit the one-tile-at-a-time image, and with no memory to claim it falls back to 1."""
    sd = t(gp.rrdbnet_state(3, 3, 64, 1, 32, seed=33))
    img = gp.image_input((200, 248, 3), 98)   # 5 x 6 tiles of 48: 3 x 4 interior windows of 64 x 64

    def run(**kw):
        net = sra.replace_conv_with_fdconv(sra.RRDBNet(3, 3, 4, 64, 1, 32))
        net.load_state_dict(sd, strict=True)
        r = sra.RealESRGANer(scale=4, model_path=None, model=net, tile=48, tile_pad=8, pre_pad=0, half=True, **kw)
        r.pre_process(img)
        r.tile_process()
        return r, r.post_process().float().cpu()

    r_auto, auto = run()
    assert r_auto._tile_batch_arg is None and r_auto.tile_batch == 12, r_auto.tile_batch   # all tiles of the commonest shape
    r_one, one = run(tile_batch=1)
    assert r_one.tile_batch == 1
    assert torch.equal(auto, one)
    # the workspace of the chosen batch was what the formula said, and it was within the budget
    need = r_auto.model.workspace_bytes(12, 64, 64, torch.float32)
    assert 0 < need <= r_auto.model.workspace_held()
    monkeypatch.setattr(sra.RealESRGANer, 'AUTO_TILE_BATCH_MEMORY', 1e-9)
    r_tight, tight = run()
    assert r_tight.tile_batch == 1 and torch.equal(tight, one)
    monkeypatch.setattr(sra.RealESRGANer, 'AUTO_TILE_BATCH_MEMORY', 0.5)
    monkeypatch.setattr(sra.RealESRGANer, 'AUTO_TILE_BATCH_CAP', 4)
    r_cap, capped = run()
    assert r_cap.tile_batch == 4 and torch.equal(capped, one)


def test_train_mode_forward_records_an_autograd_graph(sra):
    """a train()-mode forward must never hand back a tensor without grad_fn (tests/test_gpu_train.py
    checks the gradients themselves)."""
    net = sra.RRDBNet(3, 3, 4, 8, 1, 4).cuda()  # modules are born in train() mode
    y = net(torch.zeros(1, 3, 8, 8, device='cuda'))
    assert y.grad_fn is not None and y.shape == (1, 3, 32, 32)
    yb = net(torch.zeros(1, 3, 8, 8, device='cuda', dtype=torch.bfloat16))  # mixed precision: bf16 activations
    assert yb.grad_fn is not None and yb.dtype == torch.bfloat16
    with pytest.raises(TypeError):
        net(torch.zeros(1, 3, 8, 8, device='cuda', dtype=torch.float16))
    net.eval()
    with torch.no_grad():
        assert net(torch.zeros(1, 3, 8, 8, device='cuda')).grad_fn is None  # executor path


def test_dni_and_checkpoint_containers(sra, tmp_path):
    """RealESRGANer.dni (utils.py:77-86): w0 * a + w1 * b on every entry, loaded through the list form of model_path."""
    a = t(gp.rrdbnet_state(3, 3, 8, 1, 4, seed=50, fdconv=False))
    b = t(gp.rrdbnet_state(3, 3, 8, 1, 4, seed=51, fdconv=False))
    pa, pb = str(tmp_path / 'a.pth'), str(tmp_path / 'b.pth')
    torch.save({'params': a}, pa)
    torch.save({'params': b}, pb)
    r = sra.RealESRGANer(scale=4, model_path=[pa, pb], dni_weight=[0.3, 0.7], model=sra.RRDBNet(3, 3, 4, 8, 1, 4), half=False)
    mixed = {k: 0.3 * a[k] + 0.7 * b[k] for k in a}
    got = r.model.state_dict()
    assert all(torch.allclose(got[k].cpu(), mixed[k], atol=1e-7) for k in mixed)
    x = torch.from_numpy(gp.image_input((1, 3, 12, 10), 98))
    with torch.no_grad():
        assert (r.model(x.cuda()).cpu() - rrdbnet_forward(mixed, x)).abs().max().item() < 1e-5


def test_small_inputs_replay_a_captured_graph_bit_for_bit(sra):
    """Opt-in graph replay (SR_AMD_GRAPH=1; off by default because it does not pay on this stack).
    Launch-bound inputs (B*H*W < 512^2): the executor captures everything between the input pack and the output unpack
    on its second call with a shape and replays the graph afterwards.  Replayed forwards must equal eager ones bit for bit,
    on changing inputs, and reloading parameters must drop the captured graphs (they carry parameter pointers)."""
    from super_resolution_amd import _lib
    sd = t(gp.rrdbnet_state(1, 1, 64, 1, 32, seed=7))
    _lib.lib().sr_debug_enable_graphs(1)
    try:
        _graph_replay_checks(sra, _lib, sd)
    finally:
        _lib.lib().sr_debug_enable_graphs(0)


def _graph_replay_checks(sra, _lib, sd):
    for dt in (torch.float32, torch.bfloat16):
        net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, 1, 32))
        net.load_state_dict(sd, strict=True)
        net = net.eval().cuda()
        xs = [torch.from_numpy(gp.image_input((2, 1, 24, 40), seed=90 + i)).cuda().to(dt) for i in range(4)]
        with torch.no_grad():
            got = [net(x).clone() for x in xs]          # eager, capture + replay, replay, replay
            replays = _lib.lib().sr_debug_graph_replays(net._sr.h)
            net.profile(True)                            # event profiling keeps the executor eager
            ref = [net(x).clone() for x in xs]
            net.profile(False)
        assert replays == 3, replays
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        # new parameter values through the same pointers are picked up by the replay; a reload rebuilds the executor
        with torch.no_grad():
            net.conv_body.bias.add_(0.5)
            y1 = net(xs[0])
            assert not torch.equal(y1, got[0])
            net.load_state_dict(sd, strict=True)
            assert torch.equal(net(xs[0]), got[0])
