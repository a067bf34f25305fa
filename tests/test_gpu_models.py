"""GPU parity of the drop-in modules (FDConv, RRDBNet, RealESRGANer) against the reference's golden
outputs and the CPU oracle.  Runs on the MI355X box (-m gpu); every forward goes through libsr_amd."""
import math

import numpy as np
import pytest
import torch

import golden_params as gp
from oracle import realesrganer_ref as rr
from oracle.rrdbnet_ref import rrdbnet_forward

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4  # north_star: fp32 max-abs-err <= 1e-4
PSNR_DELTA_DB = 0.05  # north_star: bf16 PSNR delta <= 0.05 dB


def t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def psnr(a, b):
    """train/test_optimized_model.py:107-114: 20*log10(1/sqrt(mse)) on [0,1] data."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 20 * math.log10(1.0 / math.sqrt(mse))


@pytest.fixture(scope='module')
def sra():
    import super_resolution_amd as m
    from super_resolution_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return m


def _layer_cases(golden):
    out = []
    for key in golden['fdconv_layers'].files:
        if key.startswith('c') and key.endswith('.y'):
            tag = key[:-2]
            cin, cout, h, w = [int(s.lstrip('chw')) for s in tag.split('_')]
            out.append((tag, cin, cout, h, w))
    return out


def test_fdconv_layers_vs_reference_golden(sra, golden):
    """Every golden FDConv layer case (6 channel shapes, odd sizes, the small-plane branch) in fp32."""
    g = golden['fdconv_layers']
    mods = {}
    worst = 0.0
    for tag, cin, cout, h, w in _layer_cases(golden):
        li = gp.LAYER_SHAPES.index((cin, cout))
        if (cin, cout) not in mods:
            m = sra.FDConv(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
            m.load_state_dict(t(gp.fdconv_params(cin, cout, seed=100 + li)), strict=True)
            mods[(cin, cout)] = m.eval().cuda()
        x = torch.from_numpy(gp.signed_input((2, cin, h, w), seed=200 + li)).cuda()
        with torch.no_grad():
            y = mods[(cin, cout)](x).cpu().numpy()
        err = np.abs(y - g[tag + '.y']).max()
        worst = max(worst, err)
        assert err < FP32_TOL, (tag, err)
    print('worst fp32 FDConv layer error vs reference: %.3e' % worst)


def test_fdconv_group_map_matches_reference(sra, golden):
    for cin, cout in gp.LAYER_SHAPES:
        m = sra.FDConv(cin, cout, kernel_size=3, padding=1)
        assert np.array_equal(m.group_map().numpy(), golden['group_maps']['gmap_%d_%d' % (cin, cout)])


def test_fdconv_thin_layer_is_plain_conv(sra):
    m = sra.FDConv(1, 64, kernel_size=3, padding=1).eval().cuda()
    assert not m.fdconv_active and 'weight' in m.state_dict() and 'dft_weight' not in m.state_dict()
    x = torch.from_numpy(gp.signed_input((2, 1, 13, 9), 3)).cuda()
    with torch.no_grad():
        y = m(x)
    ref = torch.nn.functional.conv2d(x.cpu(), m.weight.detach().cpu(), m.bias.detach().cpu(), padding=1)
    assert (y.cpu() - ref).abs().max().item() < 1e-5


def test_fdconv_bf16_layer(sra, golden):
    """bf16 storage + MFMA: error relative to the fp32 reference output stays at bf16 rounding level."""
    cin, cout = 64, 32
    m = sra.FDConv(cin, cout, kernel_size=3, padding=1)
    m.load_state_dict(t(gp.fdconv_params(cin, cout, seed=100)), strict=True)
    m = m.eval().cuda()
    x = torch.from_numpy(gp.signed_input((2, cin, 40, 48), seed=200)).cuda()
    with torch.no_grad():
        y = m(x.to(torch.bfloat16)).float().cpu().numpy()
    ref = golden['fdconv_layers']['c64_32_h40_w48.y']
    rel = np.abs(y - ref).max() / np.abs(ref).max()
    assert rel < 3e-2, rel


def build_net(sra, args, sd, fdconv):
    net = sra.RRDBNet(*args)
    if fdconv:
        net = sra.replace_conv_with_fdconv(net)
    net.load_state_dict(sd, strict=True)
    return net.eval().cuda()


def test_rrdbnet_fdconv_one_block_vs_reference_golden(sra, golden):
    sd = t(gp.rrdbnet_state(1, 1, 64, 1, 32, seed=7))
    net = build_net(sra, (1, 1, 4, 64, 1, 32), sd, True)
    assert len(net.state_dict()) == int(golden['rrdbnet_fdconv_1block']['n_keys'])
    x = torch.from_numpy(gp.image_input((2, 1, 32, 32), seed=70)).cuda()
    with torch.no_grad():
        y = net(x).cpu().numpy()
    ref = golden['rrdbnet_fdconv_1block']['y']
    assert y.shape == (2, 1, 128, 128)
    assert np.abs(y - ref).max() < FP32_TOL


def test_rrdbnet_fdconv_three_blocks_odd_size_vs_oracle(sra):
    sd = t(gp.rrdbnet_state(1, 1, 64, 3, 32, seed=8))
    net = build_net(sra, (1, 1, 4, 64, 3, 32), sd, True)
    x = torch.from_numpy(gp.image_input((1, 1, 29, 38), seed=71))
    with torch.no_grad():
        y = net(x.cuda()).cpu()
    ref = rrdbnet_forward(sd, x)
    assert (y - ref).abs().max().item() < FP32_TOL


def test_plain_rrdbnet_x4plus_shape_vs_oracle(sra):
    """config 1 architecture (RealESRGAN_x4plus: RRDBNet(3,3,64,23,32)), fp32, on a 48x40 RGB input."""
    sd = t(gp.rrdbnet_state(3, 3, 64, 23, 32, seed=11, fdconv=False, gain=0.5))
    net = build_net(sra, (3, 3, 4, 64, 23, 32), sd, False)
    x = torch.from_numpy(gp.image_input((1, 3, 48, 40), seed=72))
    with torch.no_grad():
        y = net(x.cuda()).cpu()
    ref = rrdbnet_forward(sd, x)
    assert y.shape == (1, 3, 192, 160)
    assert (y - ref).abs().max().item() < FP32_TOL
    # bf16 MFMA path on the same network: PSNR of the output against the fp32 oracle output
    with torch.no_grad():
        yb = net(x.cuda().to(torch.bfloat16)).float().cpu()
    p = psnr(yb.clamp(0, 1), ref.clamp(0, 1))
    print('plain RRDBNet bf16 vs fp32 oracle: PSNR %.1f dB' % p)
    assert p > 40.0


def test_rrdbnet_fdconv_bf16_psnr_delta(sra):
    """bf16 criterion of north_star: PSNR delta <= 0.05 dB.  With no ground-truth image for random weights the
    target is a synthetic 'clean' image; PSNR(out, target) is evaluated for the fp32 oracle output and for the
    bf16 GPU output and the two must agree to 0.05 dB."""
    sd = t(gp.rrdbnet_state(1, 1, 64, 2, 32, seed=12))
    net = build_net(sra, (1, 1, 4, 64, 2, 32), sd, True)
    x = torch.from_numpy(gp.image_input((2, 1, 32, 32), seed=73))
    ref = rrdbnet_forward(sd, x)
    with torch.no_grad():
        yb = net(x.cuda().to(torch.bfloat16)).float().cpu()
    target = torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
    p_ref = psnr(ref.clamp(0, 1), target)
    p_bf = psnr(yb.clamp(0, 1), target)
    p_direct = psnr(yb, ref)
    print('PSNR vs target: fp32 oracle %.3f dB, bf16 GPU %.3f dB; bf16-vs-fp32 direct %.1f dB' % (p_ref, p_bf, p_direct))
    assert abs(p_ref - p_bf) <= PSNR_DELTA_DB
    assert p_direct > 35.0


def test_scale2_network_pixel_unshuffle(sra, golden):
    sd = t(gp.rrdbnet_state(12, 3, 8, 1, 4, seed=10, fdconv=False))
    net = build_net(sra, (3, 3, 2, 8, 1, 4), sd, False)
    img = gp.image_input((21, 33, 3), seed=92)
    r = sra.RealESRGANer(scale=2, model_path=None, model=net, tile=0, pre_pad=0, half=False)
    r.pre_process(img)
    r.process()
    out = r.post_process().cpu().numpy()
    ref = golden['realesrganer']['x2_21x33']
    assert out.shape == ref.shape and np.abs(out - ref).max() < 1e-5


def test_realesrganer_vs_reference_golden(sra, golden, tmp_path):
    g = golden['realesrganer']
    sd = t(gp.rrdbnet_state(3, 3, 8, 1, 4, seed=9, fdconv=False))
    ckpt = str(tmp_path / 'net.pth')
    torch.save({'params': sd}, ckpt)
    img = gp.image_input((50, 70, 3), seed=90)
    for tag, tile, pad, pre in (('whole', 0, 10, 0), ('tile32', 32, 10, 0), ('tile32_pre7', 32, 10, 7), ('tile16_pad4', 16, 4, 3)):
        r = sra.RealESRGANer(scale=4, model_path=ckpt, model=sra.RRDBNet(3, 3, 4, 8, 1, 4), tile=tile, tile_pad=pad,
                             pre_pad=pre, half=False)
        r.pre_process(img)
        r.tile_process() if tile > 0 else r.process()
        out = r.post_process().cpu().numpy()
        assert np.abs(out - g['rgb50x70_' + tag]).max() < 1e-5, tag
    # Real-ESRGAN-master/tests/test_utils.py:33-56
    r = sra.RealESRGANer(scale=4, model_path=ckpt, model=sra.RRDBNet(3, 3, 4, 8, 1, 4), tile=10, tile_pad=10, pre_pad=2)
    img12 = gp.image_input((12, 12, 3), seed=91)
    r.pre_process(img12)
    assert r.img.shape == (1, 3, 14, 14)
    r.scale = 1
    r.pre_process(img12)
    assert r.img.shape == (1, 3, 16, 16)
    assert np.array_equal(r.img.cpu().numpy(), g['t_pre_mod_img'])
    r.scale = 4
    r.pre_process(img12)
    r.tile_process()
    assert r.output.shape == (1, 3, 64, 64)
    assert np.abs(r.output.cpu().numpy() - g['t_tile_out']).max() < 1e-5
    r.mod_scale = 4
    assert r.post_process().shape == (1, 3, 48, 48)  # 64 - mod_pad 2*4 - pre_pad 2*4


def test_enhance_modes_vs_oracle(sra):
    sd = t(gp.rrdbnet_state(3, 3, 8, 1, 4, seed=9, fdconv=False))
    net = build_net(sra, (3, 3, 4, 8, 1, 4), sd, False)
    oracle_net = lambda z: rrdbnet_forward(sd, z)
    cases = [((gp.image_input((9, 7, 3), 1) * 255).astype(np.uint8), dict(tile=0, pre_pad=2), 'RGB'),
             ((gp.image_input((9, 7), 2) * 60000).astype(np.uint16), dict(tile=4, tile_pad=2, pre_pad=0), 'L'),
             (gp.image_input((6, 5, 4), 3), dict(tile=0, pre_pad=0), 'RGBA')]
    for img, kw, mode in cases:
        r = sra.RealESRGANer(scale=4, model_path=None, model=net, half=False, **kw)
        out, m = r.enhance(img, outscale=4)
        ref, m2 = rr.enhance(oracle_net, img, 4, **{'tile_pad': 10, **kw})
        assert m == m2 == mode and out.shape == ref.shape and out.dtype == ref.dtype
        assert np.abs(out.astype(np.int64) - ref.astype(np.int64)).max() <= 1  # rounding of a 1e-5 float difference
    # outscale != scale: the 4x result resized with the Lanczos-4 restatement (utils.py:256-261; parity unpinned, no cv2)
    from super_resolution_amd.realesrganer import lanczos4_resize
    r = sra.RealESRGANer(scale=4, model_path=None, model=net, half=False, tile=0, pre_pad=2)
    out4, _ = r.enhance(cases[0][0], outscale=4)
    out2, _ = r.enhance(cases[0][0], outscale=2.5)
    assert out2.shape == (22, 17, 3) and out2.dtype == np.uint8
    assert np.array_equal(out2, lanczos4_resize(out4, 22, 17))


@pytest.mark.parametrize('half', [False, True])
def test_enhance_device_image_path_is_bit_identical_to_the_numpy_steps(sra, half):
    """enhance() on a GPU converts the caller's integer image to float planes and the result back to the integer image with
    two kernels (sr_image_to_planes / sr_image_from_planes) so that only integers cross PCIe; the numpy steps of
    utils.py:197-254 (`_enhance_host`) are the yardstick: every mode, bit for bit."""
    sd = t(gp.rrdbnet_state(3, 3, 8, 1, 4, seed=9, fdconv=False))
    net = build_net(sra, (3, 3, 4, 8, 1, 4), sd, False)
    rng = np.random.RandomState(12)
    cases = [((rng.rand(19, 23, 3) * 255).astype(np.uint8), dict(tile=0, pre_pad=3)),
             ((rng.rand(21, 18, 3) * 255).astype(np.uint8), dict(tile=8, tile_pad=3, pre_pad=0)),
             ((rng.rand(17, 13) * 65535).astype(np.uint16), dict(tile=0, pre_pad=0)),
             ((rng.rand(14, 15, 4) * 255).astype(np.uint8), dict(tile=0, pre_pad=1)),
             ((rng.rand(11, 12, 4) * 65535).astype(np.uint16), dict(tile=0, pre_pad=0, _alpha='bilinear')),
             (rng.rand(10, 9, 3).astype(np.float32) * 300, dict(tile=0, pre_pad=0)),
             (rng.rand(10, 9).astype(np.float64), dict(tile=0, pre_pad=0))]
    for img, kw in cases:
        kw = dict(kw)
        alpha = kw.pop('_alpha', 'realesrgan')
        r = sra.RealESRGANer(scale=4, model_path=None, model=net, half=half, **kw)
        out, mode = r.enhance(img, alpha_upsampler=alpha)
        sixteen = np.max(img) > 256
        ref = r._enhance_host(img, sixteen, 65535 if sixteen else 255, mode, alpha)
        assert out.dtype == ref.dtype and out.shape == ref.shape, (img.shape, img.dtype)
        assert np.array_equal(out, ref), (img.shape, img.dtype, int(np.abs(out.astype(np.int64) - ref.astype(np.int64)).max()))
    net.compute_dtype = None


def test_cpu_tensor_is_refused(sra):
    net = sra.RRDBNet(3, 3, 4, 8, 1, 4)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 8, 8))


def test_rrdbnet_fdconv_bf16_psnr_delta_mfma_lowpass(sra):
    """The same bf16 criterion on 64 x 64 frames, where every trunk tensor takes the MFMA (DFT-as-GEMM) low-pass; the run
    is repeated on the FFT low-pass kernels and the two bf16 outputs must be equally close to the fp32 oracle."""
    from super_resolution_amd import _lib
    sd = t(gp.rrdbnet_state(1, 1, 64, 2, 32, seed=21))
    net = build_net(sra, (1, 1, 4, 64, 2, 32), sd, True)
    x = torch.from_numpy(gp.image_input((2, 1, 64, 64), seed=77))
    ref = rrdbnet_forward(sd, x)
    target = torch.nn.functional.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
    p_ref = psnr(ref.clamp(0, 1), target)
    out = {}
    for name, force_fft in (('mfma', 0), ('fft', 1)):
        _lib.lib().sr_debug_force_fft_lowpass(force_fft)
        try:
            with torch.no_grad():
                out[name] = net(x.cuda().to(torch.bfloat16)).float().cpu()
        finally:
            _lib.lib().sr_debug_force_fft_lowpass(-1)
        p = psnr(out[name].clamp(0, 1), target)
        print('%s low-pass: PSNR vs target %.3f dB (fp32 oracle %.3f), direct %.1f dB' % (name, p, p_ref, psnr(out[name], ref)))
        assert abs(p_ref - p) <= PSNR_DELTA_DB
    assert psnr(out['mfma'], ref) > psnr(out['fft'], ref) - 1.0


def test_full_size_batch_independence_and_shape(sra):
    """BASELINE.json configs[1] at full size (23 blocks, batch 16 of 512 x 512, bf16): the oracle cannot run this in
    test time, so the check is a size-independent property of the path -- every statistic of the network is per sample,
    hence sample i of the batch must come out exactly as when it is run alone (batch / sample offsets, 32-bit index
    ranges, persistent tile walks at the bench's sizes) -- plus shape, range and finiteness."""
    sd = t(gp.rrdbnet_state(1, 1, 64, 23, 32, seed=1234))
    net = build_net(sra, (1, 1, 4, 64, 23, 32), sd, True)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(16, 1, 512, 512, generator=g).to(torch.bfloat16).cuda()
    with torch.no_grad():
        y = net(x)
        assert y.shape == (16, 1, 2048, 2048) and bool(torch.isfinite(y.float()).all())
        for i in (0, 11):
            yi = net(x[i:i + 1].contiguous())
            d = (yi.float() - y[i:i + 1].float()).abs().max().item()
            assert d == 0.0, (i, d)
    assert float(y.float().abs().max()) < 1e3


def test_full_size_lowpass_properties(sra):
    """FBM low-pass planes at the bench's trunk size (16 x 32 ch x 512 x 512, bf16, MFMA path) through properties that
    need no oracle: linearity, nesting L8(L2(x)) = L8(x) (the bands are nested ideal filters), mean preservation
    (every band keeps DC) and non-increasing energy x >= L2 >= L4 >= L8."""
    import ctypes as C
    from super_resolution_amd import _lib as L
    lib = L.lib()
    B, Cc, H, W = 16, 32, 512, 512
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(H, W, C.byref(plan)))
    ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda:0')
    st = torch.cuda.current_stream().cuda_stream

    def lowpass(x):
        bands = [torch.empty_like(x) for _ in range(3)]
        pooled = torch.empty(B, Cc, device='cuda:0')
        L.check(lib.sr_fbm_lowpass(plan, 1, B, L.view(x), 0, L.view(bands[0]), L.view(bands[1]), L.view(bands[2]),
                                   pooled.data_ptr(), Cc, ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
        return [b_.float() for b_ in bands], pooled

    g = torch.Generator().manual_seed(9)
    x = (torch.rand(B, H, W, Cc, generator=g) - 0.3).to(torch.bfloat16).cuda()  # NHWC, 32 channels
    y = (torch.rand(B, H, W, Cc, generator=g) - 0.5).to(torch.bfloat16).cuda()
    Lx, px = lowpass(x)
    Ly, _ = lowpass(y)
    z = (x.float() * 0.5 + y.float()).to(torch.bfloat16)
    Lz, _ = lowpass(z)
    tol = 3 * 2.0 ** -8  # a few bf16 roundings of O(1) values
    for k in range(3):
        assert (Lz[k] - (0.5 * Lx[k] + Ly[k])).abs().max().item() < tol, k
        assert (Lx[k].mean(dim=(1, 2)) - x.float().mean(dim=(1, 2))).abs().max().item() < 1e-3, k
    assert (px - x.float().mean(dim=(1, 2))).abs().max().item() < 2e-4
    L8_of_L2, _ = lowpass(Lx[0].to(torch.bfloat16))
    assert (L8_of_L2[2] - Lx[2]).abs().max().item() < tol
    e = [float(x.float().pow(2).sum())] + [float(v.pow(2).sum()) for v in Lx]
    assert e[0] >= e[1] * (1 - 1e-3) and e[1] >= e[2] * (1 - 1e-3) and e[2] >= e[3] * (1 - 1e-3), e
    lib.sr_fbm_plan_destroy(plan)
