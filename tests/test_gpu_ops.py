"""GPU parity of every libsr_amd entry point against the CPU oracle (runs on the MI355X box: -m gpu).

All calls go through the C ABI (ctypes).  fp32 tolerance is north_star's 1e-4 max-abs-err; bf16
results are compared with a bf16-appropriate relative tolerance stated at each test.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_params as gp
from oracle import fdconv_ref as fr

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


@pytest.fixture(scope='module')
def L():
    from super_resolution_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return _lib


def dev(t):
    return t.to('cuda:0')


def nhwc(t, dt=torch.float32):
    return dev(t).permute(0, 2, 3, 1).contiguous().to(dt)


def nchw(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


def blocked(t, dt=torch.float32, cb=8):
    """NCHW (C % cb == 0) -> channel-blocked (B, C/cb, H, W, cb) on the GPU."""
    B, Cc, H, W = t.shape
    return dev(t).view(B, Cc // cb, cb, H, W).permute(0, 1, 3, 4, 2).contiguous().to(dt)


def unblocked(t):
    B, nb, H, W, cb = t.shape
    return t.float().permute(0, 1, 4, 2, 3).reshape(B, nb * cb, H, W).cpu()


def st(L):
    return L.stream_ptr(torch.device('cuda:0'))


def make_lp(L, sd, cin, cout, dtype, keep):
    """sr_layer_params of an active FDConv layer from a numpy/torch state dict (+ basis via the library)."""
    lib = L.lib()
    d = {k: dev(torch.as_tensor(v)).contiguous() for k, v in sd.items()}
    keep.append(d)
    lp = L.LayerParams()
    lp.cin, lp.cout, lp.active = cin, cout, 1
    lp.loc_taps = d['KSM_Local.conv.weight'].shape[-1]
    gmap = dev(fr.freq_group_map(cout, cin))
    basis = torch.empty(4, cout, cin, 9, device='cuda:0')
    ws = torch.empty(lib.sr_fdconv_basis_workspace_bytes(cout, cin), dtype=torch.uint8, device='cuda:0')
    L.check(lib.sr_fdconv_basis(d['dft_weight'].data_ptr(), gmap.data_ptr(), fr.fdconv_alpha(cout, cin), cout, cin,
                                basis.data_ptr(), ws.data_ptr(), ws.numel(), st(L)))
    keep += [gmap, basis, ws]
    p = lambda k: d[k].data_ptr()
    lp.bias, lp.basis, lp.fc_w = p('bias'), basis.data_ptr(), p('KSM_Global.fc.weight')
    lp.bn_w, lp.bn_b = p('KSM_Global.bn.weight'), p('KSM_Global.bn.bias')
    lp.bn_rm, lp.bn_rv = p('KSM_Global.bn.running_mean'), p('KSM_Global.bn.running_var')
    lp.relu_s, lp.relu_b = p('KSM_Global.relu.scale'), p('KSM_Global.relu.bias')
    lp.ch_w, lp.ch_b = p('KSM_Global.channel_fc.weight'), p('KSM_Global.channel_fc.bias')
    lp.fl_w, lp.fl_b = p('KSM_Global.filter_fc.weight'), p('KSM_Global.filter_fc.bias')
    lp.sp_w, lp.sp_b = p('KSM_Global.spatial_fc.weight'), p('KSM_Global.spatial_fc.bias')
    lp.kn_w, lp.kn_b = p('KSM_Global.kernel_fc.weight'), p('KSM_Global.kernel_fc.bias')
    for i in range(3):
        lp.gate_w[i] = p('FBM.freq_weight_conv_list.%d.weight' % i)
        lp.gate_b[i] = p('FBM.freq_weight_conv_list.%d.bias' % i)
    lp.loc_w, lp.ln_w, lp.ln_b = p('KSM_Local.conv.weight'), p('KSM_Local.norm.weight'), p('KSM_Local.norm.bias')
    return lp, basis


# ------------------------------------------------------------------------------------------ layout
def test_pack_unpack_and_pixel_unshuffle(L):
    from oracle.rrdbnet_ref import pixel_unshuffle
    lib = L.lib()
    x = torch.from_numpy(gp.signed_input((2, 3, 8, 12), 1))
    for r in (1, 2, 4):
        xd = dev(x)
        out = torch.empty(2, 8 // r, 12 // r, 3 * r * r, device='cuda:0')
        L.check(lib.sr_pack_nchw(xd.data_ptr(), 0, out.data_ptr(), 0, 2, 3, 8, 12, r, st(L)))
        ref = pixel_unshuffle(x, r) if r > 1 else x
        assert torch.equal(nchw(out), ref)
    y = torch.empty(2, 3, 8, 12, device='cuda:0')
    xn = nhwc(x)
    L.check(lib.sr_unpack_nhwc(xn.data_ptr(), 0, y.data_ptr(), 0, 2, 3, 8, 12, st(L)))
    assert torch.equal(y.cpu(), x)
    # dtype conversion on the way in: fp32 -> bf16 is round-to-nearest-even like torch
    out = torch.empty(2, 8, 12, 3, device='cuda:0', dtype=torch.bfloat16)
    L.check(lib.sr_pack_nchw(dev(x).data_ptr(), 0, out.data_ptr(), 1, 2, 3, 8, 12, 1, st(L)))
    assert torch.equal(out.cpu(), x.permute(0, 2, 3, 1).to(torch.bfloat16))


# ------------------------------------------------------------------------------------------ FBM
def run_lowpass(L, x, dt, ups=0):
    """FFT kernels (generic and power-of-two); the DFT-as-GEMM path has its own test below."""
    lib = L.lib()
    lib.sr_debug_force_fft_lowpass(1)
    try:
        return _run_lowpass(L, lib, x, dt, ups)
    finally:
        lib.sr_debug_force_fft_lowpass(-1)


def _run_lowpass(L, lib, x, dt, ups, want=(0, 1, 2)):
    B, Cc, H, W = x.shape
    Ho, Wo = H << ups, W << ups
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(Ho, Wo, C.byref(plan)))
    xn = nhwc(x, dt)
    bands = [torch.full((B, Ho, Wo, Cc), float('nan'), device='cuda:0', dtype=dt) for _ in range(3)]
    views = [L.view(bands[k]) if k in want else L.View() for k in range(3)]  # p == NULL: band not wanted
    pooled = torch.full((B, Cc), float('nan'), device='cuda:0')
    ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda:0')
    L.check(lib.sr_fbm_lowpass(plan, L.dtype_code(dt), B, L.view(xn), ups, views[0], views[1], views[2], pooled.data_ptr(), Cc,
                               ws.data_ptr(), ws.numel(), st(L)))
    torch.cuda.synchronize()
    lib.sr_fbm_plan_destroy(plan)
    return [nchw(b) for b in bands], pooled.cpu()


@pytest.mark.parametrize('shape,dt,fft', [((2, 32, 64, 64), torch.bfloat16, 0), ((1, 64, 48, 80), torch.bfloat16, 0),
                                          ((1, 32, 522, 532), torch.bfloat16, 0), ((1, 6, 128, 64), torch.float32, 1),
                                          ((2, 32, 64, 128), torch.bfloat16, 1), ((1, 5, 33, 35), torch.float32, 1),
                                          ((1, 4, 58, 36), torch.float32, 1)])
def test_fbm_lowpass_band_selection(L, shape, dt, fft):
    """A band whose view is NULL is not produced (the adjoint of the band mix wants one band per call): the wanted
    planes are bit-identical to those of the three-band call, on the matrix-core path, the power-of-two FFT and the
    any-size FFT; the spatial means still come out."""
    lib = L.lib()
    x = torch.from_numpy(gp.signed_input(shape, 23)).to(dt).float()
    lib.sr_debug_force_fft_lowpass(1 if fft else 0)
    try:
        full, pooled = _run_lowpass(L, lib, x, dt, 0)
        for want in ((0,), (1,), (2,), (0, 2), (1, 2)):
            got, pooled_w = _run_lowpass(L, lib, x, dt, 0, want)
            for k in range(3):
                if k in want:
                    assert torch.equal(got[k], full[k]), (shape, want, k)
                else:
                    assert torch.isnan(got[k]).all()  # untouched
            assert torch.equal(pooled_w, pooled)
        with pytest.raises(Exception):
            _run_lowpass(L, lib, x, dt, 0, ())
    finally:
        lib.sr_debug_force_fft_lowpass(-1)


@pytest.mark.parametrize('shape', [(2, 6, 12, 12), (1, 5, 33, 35), (2, 32, 40, 48), (1, 64, 64, 64), (1, 8, 130, 77),
                                   (1, 4, 58, 36), (1, 2, 128, 256),
                                   # long non-power-of-two sides: 512- and 1024-thread blocks, radix 3 / 5 / 7 butterflies
                                   (1, 3, 1200, 36), (1, 4, 24, 1920), (1, 2, 2100, 20),
                                   # largest radix an odd prime > 7 on both axes: pruned prime passes (forward output
                                   # cut, inverse operand cut), even and odd lengths
                                   (1, 2, 46, 62), (2, 4, 34, 38), (1, 3, 51, 39), (1, 2, 174, 190), (1, 2, 11, 13)])
def test_fbm_lowpass_fp32(L, shape):
    x = torch.from_numpy(gp.signed_input(shape, 11))
    bands, pooled = run_lowpass(L, x, torch.float32)
    ref = fr.fbm_lowpass(x.double()).float()
    for k in range(3):
        err = (bands[k] - ref[k]).abs().max().item()
        assert err < 2e-5, (shape, k, err)
    assert (pooled - x.mean(dim=(2, 3))).abs().max().item() < 1e-6


@pytest.mark.parametrize('shape', [(2, 32, 64, 64), (1, 6, 128, 64), (1, 34, 64, 256), (1, 64, 512, 512), (1, 4, 1024, 128),
                                   (1, 2, 64, 2048), (1, 2, 2048, 64)])
def test_fbm_lowpass_pow2_fast_path(L, shape):
    """Power-of-two sizes take the register-blocked FFT; it must agree with the oracle and with the generic path."""
    x = torch.from_numpy(gp.signed_input(shape, 14))
    bands, pooled = run_lowpass(L, x, torch.float32)
    ref = fr.fbm_lowpass(x.double()).float()
    for k in range(3):
        err = (bands[k] - ref[k]).abs().max().item()
        assert err < 2e-5, (shape, k, err)
    assert (pooled - x.mean(dim=(2, 3))).abs().max().item() < 1e-6
    L.lib().sr_debug_force_generic_fft(1)
    try:
        bands_g, pooled_g = run_lowpass(L, x, torch.float32)
    finally:
        L.lib().sr_debug_force_generic_fft(0)
    for k in range(3):
        assert (bands[k] - bands_g[k]).abs().max().item() < 2e-5


def test_fbm_lowpass_pow2_upsampled_bf16(L):
    x = torch.from_numpy(gp.signed_input((2, 32, 64, 128), 15)).to(torch.bfloat16).float()
    xu = F.interpolate(x, scale_factor=2, mode='nearest')
    bands, pooled = run_lowpass(L, x, torch.bfloat16, ups=1)
    ref = fr.fbm_lowpass(xu.double()).float()
    for k in range(3):
        assert (bands[k] - ref[k]).abs().max().item() < 4e-3
    assert (pooled - xu.mean(dim=(2, 3))).abs().max().item() < 1e-5


@pytest.mark.parametrize('shape,ups,cb', [((2, 32, 64, 96), 0, 0), ((1, 64, 128, 64), 0, 32), ((1, 32, 512, 128), 0, 32),
                                          ((1, 32, 96, 160), 0, 0), ((2, 64, 32, 48), 1, 32), ((1, 32, 256, 512), 0, 32)])
def test_fbm_lowpass_mfma_bf16(L, shape, ups, cb):
    """The DFT-as-GEMM path (fp16 MFMA, fbm_mfma.hip; the default for bf16 planes of 32-channel groups) against the oracle
    (asymmetric row masks at H = 96, 128, 256, 512 included) and against the FFT path on the same input."""
    lib = L.lib()
    x = bf(torch.from_numpy(gp.signed_input(shape, 23)))
    B, Cc, H, W = x.shape
    Ho, Wo = H << ups, W << ups
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(Ho, Wo, C.byref(plan)))
    if cb:
        xd, mk = blocked(x, torch.bfloat16, cb), L.blocked_view
        mkb = lambda: torch.full((B, Cc // cb, Ho, Wo, cb), float('nan'), device='cuda:0', dtype=torch.bfloat16)
        back = unblocked
    else:
        xd, mk = nhwc(x, torch.bfloat16), L.view
        mkb = lambda: torch.full((B, Ho, Wo, Cc), float('nan'), device='cuda:0', dtype=torch.bfloat16)
        back = nchw
    ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda:0')
    res = []
    for force_fft in (0, 1):
        bands = [mkb() for _ in range(3)]
        pooled = torch.full((B, Cc), float('nan'), device='cuda:0')
        lib.sr_debug_force_fft_lowpass(force_fft)
        try:
            L.check(lib.sr_fbm_lowpass(plan, 1, B, mk(xd), ups, mk(bands[0]), mk(bands[1]), mk(bands[2]), pooled.data_ptr(), Cc,
                                       ws.data_ptr(), ws.numel(), st(L)))
            torch.cuda.synchronize()
        finally:
            lib.sr_debug_force_fft_lowpass(-1)
        res.append(([back(b_) for b_ in bands], pooled.cpu()))
    lib.sr_fbm_plan_destroy(plan)
    xu = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    ref = fr.fbm_lowpass(xu.double()).float()
    for k in range(3):
        tol = 2.0 ** -8 * max(1.0, ref[k].abs().max().item())
        assert (res[0][0][k] - ref[k]).abs().max().item() < tol, k
        assert (res[0][0][k] - res[1][0][k]).abs().max().item() < 2 * tol, k
        # the GEMM path must not be noisier than bf16 rounding of the result itself
        assert (res[0][0][k] - ref[k]).pow(2).mean().sqrt().item() < 2.0 ** -9 * ref[k].pow(2).mean().sqrt().item() + 1e-6, k
    assert (res[0][1] - xu.mean(dim=(2, 3))).abs().max().item() < 2e-4


@pytest.mark.parametrize('shape,ups,cb', [((1, 16, 64, 128), 0, 8), ((2, 8, 32, 64), 1, 8), ((1, 8, 2048, 64), 0, 8),
                                          ((1, 24, 40, 36), 0, 8), ((2, 64, 64, 128), 0, 32), ((1, 32, 32, 64), 1, 32),
                                          ((1, 64, 40, 36), 0, 32), ((1, 32, 64, 64), 0, 16)])
def test_fbm_lowpass_channel_blocked_layout(L, shape, ups, cb):
    """Channel-blocked tensors (blocks of 8: the executor's 2x / 4x layout; of 32: its trunk layout)."""
    lib = L.lib()
    x = torch.from_numpy(gp.signed_input(shape, 17))
    B, Cc, H, W = x.shape
    Ho, Wo = H << ups, W << ups
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(Ho, Wo, C.byref(plan)))
    xb = blocked(x, cb=cb)
    bands = [torch.full((B, Cc // cb, Ho, Wo, cb), float('nan'), device='cuda:0') for _ in range(3)]
    pooled = torch.full((B, Cc), float('nan'), device='cuda:0')
    ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda:0')
    L.check(lib.sr_fbm_lowpass(plan, 0, B, L.blocked_view(xb), ups, L.blocked_view(bands[0]), L.blocked_view(bands[1]),
                               L.blocked_view(bands[2]), pooled.data_ptr(), Cc, ws.data_ptr(), ws.numel(), st(L)))
    torch.cuda.synchronize()
    lib.sr_fbm_plan_destroy(plan)
    xu = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    ref = fr.fbm_lowpass(xu.double()).float()
    for k in range(3):
        assert (unblocked(bands[k]) - ref[k]).abs().max().item() < 2e-5, k
    assert (pooled.cpu() - xu.mean(dim=(2, 3))).abs().max().item() < 1e-6


@pytest.mark.parametrize('shape,ups,cb,dt', [
    ((1, 16, 1044, 1064), 0, 8, torch.float32),    # 4 9 29 x 8 7 19, channel-blocked (the executor's 2x layout)
    ((2, 8, 532, 522), 1, 8, torch.float32),       # nearest-2x source -> 1064 x 1044: half-length transforms (532, 522)
    ((1, 6, 1064, 1064), 0, 0, torch.float32),     # NHWC, 3 pairs in a group of 8
    ((1, 8, 2088, 2128), 0, 8, torch.float32),     # 8 9 29 x 16 7 19
    ((1, 8, 2128, 2088), 0, 8, torch.bfloat16),
    ((1, 16, 1064, 1044), 1, 8, torch.bfloat16),   # -> 2128 x 2088 from the 2x tensor (conv_up2)
    ((1, 8, 2128, 2128), 0, 0, torch.bfloat16),
])
def test_fbm_lowpass_mixed_radix_sizes(L, shape, ups, cb, dt):
    """The register-blocked mixed-radix FFT (fbm_fast_mixed.hip: sides 1044 / 1064 / 2088 / 2128 = the high-resolution
    tensors of a tile at the reference's default tiling) against the fp64 oracle and against the generic FFT kernels on
    the same input; one band alone as well (the adjoint of the band mix asks for single bands)."""
    lib = L.lib()
    x = torch.from_numpy(gp.signed_input(shape, 23)).to(dt).float()
    B, Cc, H, W = x.shape
    Ho, Wo = H << ups, W << ups
    xu = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    ref = fr.fbm_lowpass(xu.double(), torch.float64)

    def run(want=(0, 1, 2)):
        plan = C.c_void_p()
        L.check(lib.sr_fbm_plan_create(Ho, Wo, C.byref(plan)))
        if cb:
            xin = blocked(x, dt, cb)
            bands = [torch.full((B, Cc // cb, Ho, Wo, cb), float('nan'), device='cuda:0', dtype=dt) for _ in range(3)]
            vx, vb = L.blocked_view(xin), [L.blocked_view(b) if k in want else L.null_view() for k, b in enumerate(bands)]
        else:
            xin = dev(x).permute(0, 2, 3, 1).contiguous().to(dt)
            bands = [torch.full((B, Ho, Wo, Cc), float('nan'), device='cuda:0', dtype=dt) for _ in range(3)]
            vx, vb = L.view(xin), [L.view(b) if k in want else L.null_view() for k, b in enumerate(bands)]
        pooled = torch.full((B, Cc), float('nan'), device='cuda:0')
        ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda:0')
        L.check(lib.sr_fbm_lowpass(plan, L.dtype_code(dt), B, vx, ups, vb[0], vb[1], vb[2], pooled.data_ptr(), Cc, ws.data_ptr(),
                                   ws.numel(), st(L)))
        torch.cuda.synchronize()
        path = lib.sr_debug_last_lowpass_path()
        lib.sr_fbm_plan_destroy(plan)
        out = [unblocked(b) if cb else b.float().permute(0, 3, 1, 2).cpu() for b in bands]
        return out, pooled.cpu(), path

    lib.sr_debug_force_fft_lowpass(1)
    try:
        got, pooled, path = run()
        assert path == 2, path
        one, _, path1 = run(want=(1,))
        assert path1 == 2
        lib.sr_debug_no_mixed_fft(1)
        try:
            gen, pooled_gen, path_gen = run()
        finally:
            lib.sr_debug_no_mixed_fft(0)
        assert path_gen == 0, path_gen
    finally:
        lib.sr_debug_force_fft_lowpass(-1)
    tol = 2e-5 if dt == torch.float32 else 2.0 ** -8  # bf16: half an ulp of values below 2
    for k in range(3):
        err = (got[k].double() - ref[k]).abs().max().item()
        assert err < tol, (k, err)
        assert (got[k] - gen[k]).abs().max().item() < (2e-5 if dt == torch.float32 else 2.0 ** -7), k
    assert torch.equal(one[1], got[1]) and torch.isnan(one[0]).all() and torch.isnan(one[2]).all()
    assert (pooled.double() - xu.double().mean(dim=(2, 3))).abs().max().item() < 1e-5
    assert (pooled - pooled_gen).abs().max().item() < 1e-6


def test_fbm_lowpass_mixed_radix_full_width_properties(L):
    """The same kernels at the width the tiled configuration runs them (64 channels = 32 pairs = 8 pair groups per row
    block, batch 2, 2128 x 2088, bf16, channel-blocked): too large for the oracle in test time, so the check is through
    properties -- plane sums (the DC bin) equal the mean, every band keeps the mean, the bands are nested
    (L4(L2 x) = L4 x, L8(L4 x) = L8 x up to the bf16 rounding of the intermediate plane) -- and against the generic FFT."""
    lib = L.lib()
    B, Cc, H, W = 2, 64, 2128, 2088
    g = torch.Generator(device='cuda:0').manual_seed(5)
    x = (torch.rand(B, Cc // 8, H, W, 8, device='cuda:0', generator=g) - 0.3).to(torch.bfloat16)

    def run(src):
        plan = C.c_void_p()
        L.check(lib.sr_fbm_plan_create(H, W, C.byref(plan)))
        bands = [torch.full((B, Cc // 8, H, W, 8), float('nan'), device='cuda:0', dtype=torch.bfloat16) for _ in range(3)]
        pooled = torch.full((B, Cc), float('nan'), device='cuda:0')
        ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda:0')
        L.check(lib.sr_fbm_lowpass(plan, L.dtype_code(torch.bfloat16), B, L.blocked_view(src), 0, L.blocked_view(bands[0]),
                                   L.blocked_view(bands[1]), L.blocked_view(bands[2]), pooled.data_ptr(), Cc, ws.data_ptr(),
                                   ws.numel(), st(L)))
        torch.cuda.synchronize()
        path = lib.sr_debug_last_lowpass_path()
        lib.sr_fbm_plan_destroy(plan)
        return bands, pooled, path

    chan_mean = lambda t: t.float().mean(dim=(2, 3)).permute(0, 1, 2).reshape(B, Cc)  # noqa: E731  (B, C/8, 8) -> (B, C)
    lib.sr_debug_force_fft_lowpass(1)
    try:
        bands, pooled, path = run(x)
        assert path == 2, path
        again2, _, _ = run(bands[0])
        again4, _, _ = run(bands[1])
        lib.sr_debug_no_mixed_fft(1)
        try:
            gen, pooled_gen, path_gen = run(x)
        finally:
            lib.sr_debug_no_mixed_fft(0)
        assert path_gen == 0
    finally:
        lib.sr_debug_force_fft_lowpass(-1)
    mean = chan_mean(x)
    assert (pooled - mean).abs().max().item() < 1e-4
    assert (pooled - pooled_gen).abs().max().item() < 1e-6
    for k in range(3):
        assert torch.isfinite(bands[k].float()).all()
        assert (chan_mean(bands[k]) - mean).abs().max().item() < 2e-3, k   # bf16 rounding of 4.4 M samples averages out
        assert (bands[k].float() - gen[k].float()).abs().max().item() <= 2.0 ** -7, k
    assert (again2[1].float() - bands[1].float()).abs().max().item() <= 2.0 ** -6   # L4(L2 x) = L4 x
    assert (again2[2].float() - bands[2].float()).abs().max().item() <= 2.0 ** -6   # L8(L2 x) = L8 x
    assert (again4[2].float() - bands[2].float()).abs().max().item() <= 2.0 ** -6   # L8(L4 x) = L8 x


def test_fbm_lowpass_upsampled_and_bf16(L):
    x = torch.from_numpy(gp.signed_input((2, 8, 20, 24), 12))
    xu = F.interpolate(x, scale_factor=2, mode='nearest')
    bands, pooled = run_lowpass(L, x, torch.float32, ups=1)
    ref = fr.fbm_lowpass(xu.double()).float()
    for k in range(3):
        assert (bands[k] - ref[k]).abs().max().item() < 2e-5
    assert (pooled - xu.mean(dim=(2, 3))).abs().max().item() < 1e-6
    xb = x.to(torch.bfloat16)
    bands, pooled = run_lowpass(L, xb.float(), torch.bfloat16)
    ref = fr.fbm_lowpass(xb.double()).float()
    for k in range(3):  # output rounded to bf16: half an ulp of |x| < 1
        assert (bands[k] - ref[k]).abs().max().item() < 4e-3


@pytest.mark.parametrize('shape', [(1, 4, 33, 35), (2, 6, 29, 38), (1, 2, 87, 133), (1, 3, 20, 13), (1, 2, 64, 66)])
def test_fbm_lowpass_any_size_upsampled_source_half_length_rows(L, shape):
    """Any-size FFT with a nearest-2x upsampled source: one W/2-point transform per source row
    (FFT_W(up s)[c] = (1 + w^c) FFT_{W/2}(s)[c mod W/2]) against the oracle on the upsampled tensor and against the
    full-length transforms (debug hook); widths whose half has a prime factor beyond 7 take the pruned prime pass."""
    lib = L.lib()
    x = torch.from_numpy(gp.signed_input(shape, 17))
    xu = F.interpolate(x, scale_factor=2, mode='nearest')
    ref = fr.fbm_lowpass(xu.double()).float()
    lib.sr_debug_force_generic_fft(1)
    try:
        bands, pooled = run_lowpass(L, x, torch.float32, ups=1)
        lib.sr_debug_no_ups_shortcut(1)
        try:
            full, _ = run_lowpass(L, x, torch.float32, ups=1)
        finally:
            lib.sr_debug_no_ups_shortcut(0)
    finally:
        lib.sr_debug_force_generic_fft(0)
    for k in range(3):
        assert (bands[k] - ref[k]).abs().max().item() < 2e-5, (shape, k)
        assert (bands[k] - full[k]).abs().max().item() < 2e-5, (shape, k)
    assert (pooled - xu.mean(dim=(2, 3))).abs().max().item() < 1e-6


def test_fbm_masks_host_matches_golden(L, golden):
    lib = L.lib()
    g = golden['fbm_masks']
    for n in (12, 33, 64, 77, 130, 512, 522, 532, 2088, 2128):
        rows = np.zeros(3 * n, np.uint8)
        cols = np.zeros(3 * (n // 2 + 1), np.uint8)
        L.check(lib.sr_fbm_masks_host(n, n, rows.ctypes.data, cols.ctypes.data))
        assert np.array_equal(rows.reshape(3, n).astype(bool), g['rows_%d' % n])
        assert np.array_equal(cols.reshape(3, -1).astype(bool), g['cols_%d' % n])


def test_fbm_gates(L):
    lib = L.lib()
    keep = []
    cin, cout = 96, 32
    sd = gp.fdconv_params(cin, cout, 5)
    lp, _ = make_lp(L, sd, cin, cout, torch.float32, keep)
    x = torch.from_numpy(gp.signed_input((2, cin, 21, 37), 13))
    # the input is given as a concat of two views (64 + 32) to exercise the segment walk
    a, b = nhwc(x[:, :64]), nhwc(x[:, 64:])
    cc = L.Concat()
    cc.nseg = 2
    cc.x[0], cc.x[1] = L.view(a), L.view(b)
    gates = torch.empty(2, 21, 37, 4, device='cuda:0')
    L.check(lib.sr_fbm_gates(0, 2, 21, 37, C.byref(cc), C.byref(lp), gates.data_ptr(), st(L)))
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = fr.fbm_gates(x, [t['FBM.freq_weight_conv_list.%d.weight' % i] for i in range(3)],
                       [t['FBM.freq_weight_conv_list.%d.bias' % i] for i in range(3)])
    got = gates.cpu()[..., :3].permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 1e-5


def test_fbm_gates_bf16_mfma(L):
    """bf16: the 27-output 1x1 product on MFMA + 9-tap combine, against fp64 on the same bf16-rounded operands"""
    lib = L.lib()
    keep = []
    cin, cout = 160, 32
    sd = gp.fdconv_params(cin, cout, 6)
    lp, _ = make_lp(L, sd, cin, cout, torch.bfloat16, keep)
    for (h, w, ups) in ((21, 37, 0), (8, 32, 0), (13, 20, 1)):
        x = torch.from_numpy(gp.signed_input((2, cin, h, w), 16)).to(torch.bfloat16)
        segs = [nhwc(x[:, :64].float(), torch.bfloat16)] + [nhwc(x[:, 64 + 32 * i:96 + 32 * i].float(), torch.bfloat16) for i in range(3)]
        cc = L.Concat()
        cc.nseg, cc.ups = 4, ups
        for i, t_ in enumerate(segs):
            cc.x[i] = L.view(t_)
        H, W = h << ups, w << ups
        t = {k: torch.from_numpy(v) for k, v in sd.items()}
        gw = [t['FBM.freq_weight_conv_list.%d.weight' % i].to(torch.bfloat16).double() for i in range(3)]
        gb = [t['FBM.freq_weight_conv_list.%d.bias' % i].double() for i in range(3)]
        xr = x.double()
        if ups:
            xr = F.interpolate(xr, scale_factor=2, mode='nearest')
        ref = fr.fbm_gates(xr, gw, gb).float()
        outs = []
        for force in (0, 1):
            lib.sr_debug_force_generic_gates(force)
            try:
                gates = torch.full((2, H, W, 4), float('nan'), device='cuda:0')
                L.check(lib.sr_fbm_gates(1, 2, H, W, C.byref(cc), C.byref(lp), gates.data_ptr(), st(L)))
                torch.cuda.synchronize()
            finally:
                lib.sr_debug_force_generic_gates(0)
            outs.append(gates.cpu()[..., :3].permute(0, 3, 1, 2))
        # MFMA path: weights rounded to bf16 (as in ref); generic path keeps fp32 weights -> looser
        assert (outs[0] - ref).abs().max().item() < 2e-5, (h, w, ups)
        assert (outs[1] - ref).abs().max().item() < 2e-2


# ------------------------------------------------------------------------------------------ weights
@pytest.mark.parametrize('cin,cout', gp.LAYER_SHAPES)
def test_basis_and_weightgen(L, cin, cout):
    lib = L.lib()
    keep = []
    sd = gp.fdconv_params(cin, cout, 21)
    lp, basis = make_lp(L, sd, cin, cout, torch.float32, keep)
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref_b = fr.spectral_bases(t['dft_weight'], fr.freq_group_map(cout, cin), fr.fdconv_alpha(cout, cin), cout, cin,
                              torch.float64).float()
    got_b = basis.cpu().view(4, cout, cin, 3, 3)
    assert (got_b - ref_b).abs().max().item() < 1e-6 * max(1.0, ref_b.abs().max().item())
    B = 3
    pooled = torch.from_numpy(gp.signed_input((B, cin), 22))
    ca, fa, sa, ka = fr.ksm_global(pooled, t)
    hr = fr.ksm_local(pooled, t, cout)
    ref_w = sa.view(B, 1, 1, 3, 3) * ca.view(B, 1, cin, 1, 1) * fa.view(B, cout, 1, 1, 1) * \
        torch.einsum('bg,goiuv->boiuv', ka, ref_b) * hr
    wd = torch.empty(B, 9, cout, cin, device='cuda:0')
    pd = dev(pooled)
    mods = torch.empty(lib.sr_fdconv_weightgen_workspace_bytes(B, cin, cout) // 4, device='cuda:0')
    L.check(lib.sr_fdconv_weightgen(0, B, pd.data_ptr(), cin, C.byref(lp), 0, 0.0, wd.data_ptr(), None, mods.data_ptr(),
                                    mods.numel() * 4, st(L)))
    got_w = wd.cpu().view(B, 3, 3, cout, cin).permute(0, 3, 4, 1, 2)
    assert (got_w - ref_w).abs().max().item() < 2e-6 * max(1.0, ref_w.abs().max().item())
    # the modulation vectors left in the workspace: ca | ln | fa | sa | ka
    ms = (2 * cin + cout + 13 + 3) & ~3  # sample stride in floats (include/sr_amd.h)
    m = mods.cpu()[:B * ms].view(B, ms)
    assert (m[:, :cin] - ca).abs().max().item() < 2e-6
    assert (m[:, 2 * cin:2 * cin + cout] - fa).abs().max().item() < 2e-6
    assert (m[:, 2 * cin + cout:2 * cin + cout + 9] - sa).abs().max().item() < 2e-6
    assert (m[:, 2 * cin + cout + 9:2 * cin + cout + 13] - ka).abs().max().item() < 2e-6
    # a workspace that is too small is refused before anything is launched
    assert lib.sr_fdconv_weightgen(0, B, pd.data_ptr(), cin, C.byref(lp), 0, 0.0, wd.data_ptr(), None, mods.data_ptr(), 16, st(L)) == -2


@pytest.mark.parametrize('cin,cout,B', [(64, 32, 4), (192, 64, 2), (96, 32, 33)])
def test_weightgen_train_mode_batch_statistics(L, cin, cout, B):
    """Train-mode KSM_Global: BatchNorm over the batch of pooled vectors (FDConv.py:245-258 under model.train()), running
    statistics updated in place (momentum 0.1, unbiased variance), against the oracle's train-mode restatement."""
    lib = L.lib()
    keep = []
    sd = gp.fdconv_params(cin, cout, 23)
    lp, basis = make_lp(L, sd, cin, cout, torch.float32, keep)
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    pooled = torch.from_numpy(gp.signed_input((B, cin), 24))
    ca, fa, sa, ka = fr.ksm_global(pooled, t, training=True)
    hr = fr.ksm_local(pooled, t, cout)
    ref_b = fr.spectral_bases(t['dft_weight'], fr.freq_group_map(cout, cin), fr.fdconv_alpha(cout, cin), cout, cin).float()
    ref_w = sa.view(B, 1, 1, 3, 3) * ca.view(B, 1, cin, 1, 1) * fa.view(B, cout, 1, 1, 1) * \
        torch.einsum('bg,goiuv->boiuv', ka, ref_b) * hr
    z = pooled @ t['KSM_Global.fc.weight'].flatten(1).t()
    rm_ref = 0.9 * t['KSM_Global.bn.running_mean'] + 0.1 * z.mean(0)
    rv_ref = 0.9 * t['KSM_Global.bn.running_var'] + 0.1 * z.var(0, unbiased=True)
    bn = torch.nn.BatchNorm2d(16).train()  # torch's own update rule as a cross-check of the two lines above
    bn.running_mean.copy_(t['KSM_Global.bn.running_mean'])
    bn.running_var.copy_(t['KSM_Global.bn.running_var'])
    bn(z.view(B, 16, 1, 1))
    assert (bn.running_mean - rm_ref).abs().max().item() < 1e-6 and (bn.running_var - rv_ref).abs().max().item() < 1e-6
    wd = torch.empty(B, 9, cout, cin, device='cuda:0')
    pd = dev(pooled)
    mods = torch.empty(lib.sr_fdconv_weightgen_workspace_bytes(B, cin, cout) // 4, device='cuda:0')
    L.check(lib.sr_fdconv_weightgen(0, B, pd.data_ptr(), cin, C.byref(lp), 1, 0.1, wd.data_ptr(), None, mods.data_ptr(),
                                    mods.numel() * 4, st(L)))
    got_w = wd.cpu().view(B, 3, 3, cout, cin).permute(0, 3, 4, 1, 2)
    assert (got_w - ref_w).abs().max().item() < 3e-6 * max(1.0, ref_w.abs().max().item())
    rm, rv = keep[0]['KSM_Global.bn.running_mean'].cpu(), keep[0]['KSM_Global.bn.running_var'].cpu()
    assert (rm - rm_ref).abs().max().item() < 1e-6 * max(1.0, rm_ref.abs().max().item())
    assert (rv - rv_ref).abs().max().item() < 1e-6 * max(1.0, rv_ref.abs().max().item())
    # one value per channel: torch raises in train mode, the library refuses
    assert lib.sr_fdconv_weightgen(0, 1, pd.data_ptr(), cin, C.byref(lp), 1, 0.1, wd.data_ptr(), None, mods.data_ptr(),
                                   mods.numel() * 4, st(L)) == -1


# ------------------------------------------------------------------------------------------ conv
def run_conv(L, dt, xs, w, bias=None, lrelu=0, r1=None, s1=0.0, r2=None, s2=0.0, bands=None, gates=None, ups=0,
             per_sample=False):
    """xs: list of NCHW tensors (virtual concat); w: (cout,cin,3,3) or (B,cout,cin,3,3)."""
    lib = L.lib()
    B = xs[0].shape[0]
    H, W = xs[0].shape[2] << ups, xs[0].shape[3] << ups
    cout = w.shape[-4]
    cc = L.Concat()
    cc.nseg, cc.ups = len(xs), ups
    keep = []
    for i, x in enumerate(xs):
        t = nhwc(x, dt)
        keep.append(t)
        cc.x[i] = L.view(t)
        if bands is not None:
            cc.bands = 1
            for name, src in zip(('l2', 'l4', 'l8'), bands[i]):
                tb = nhwc(src, dt)
                keep.append(tb)
                getattr(cc, name)[i] = L.view(tb)
    wk = (w if per_sample else w[None]).permute(0, 3, 4, 1, 2).reshape(-1, 9, cout, w.shape[-3])
    wd = dev(wk).contiguous().to(dt)
    ep = L.Epilogue()
    if bias is not None:
        bd = dev(bias)
        keep.append(bd)
        ep.bias = bd.data_ptr()
    ep.lrelu = lrelu
    if r1 is not None:
        t = nhwc(r1, dt)
        keep.append(t)
        ep.r1, ep.s1 = L.view(t), s1
    if r2 is not None:
        t = nhwc(r2, dt)
        keep.append(t)
        ep.r2, ep.s2 = L.view(t), s2
    gd = None
    if gates is not None:
        g4 = torch.cat([gates, torch.zeros_like(gates[:, :1])], 1)
        gd = nhwc(g4)
    out = torch.full((B, H, W, cout), float('nan'), device='cuda:0', dtype=dt)
    L.check(lib.sr_conv3x3(L.dtype_code(dt), B, H, W, C.byref(cc), None if gd is None else gd.data_ptr(), wd.data_ptr(),
                           9 * cout * w.shape[-3] if per_sample else 0, 0, cout, C.byref(ep), L.view(out), st(L)))
    torch.cuda.synchronize()
    return nchw(out)


@pytest.mark.parametrize('cb', [0, 32])
def test_fbm_gates_accumulate_dense_block(L, cb):
    """Segment-wise gate accumulation over a dense block (x: 64 ch, x1..x4: 32 ch, five consumer layers) against the
    per-layer gates kernel on the same bf16 data and against the oracle."""
    lib = L.lib()
    B, H, W = 2, 20, 37
    segc = [64, 32, 32, 32, 32]
    xs = [bf(rnd((B, c, H, W), 300 + i)) for i, c in enumerate(segc)]
    if cb:
        xd = [blocked(x, torch.bfloat16, cb) for x in xs]
        mk = L.blocked_view
    else:
        xd = [nhwc(x, torch.bfloat16) for x in xs]
        mk = L.view
    cins = [64, 96, 128, 160, 192]
    gw = [[dev(rnd((cin, 9), 400 + 10 * j + k, 0.05)).contiguous() for k in range(3)] for j, cin in enumerate(cins)]
    gb = [[dev(rnd((1,), 500 + 10 * j + k)).contiguous() for k in range(3)] for j in range(5)]
    logits = [torch.full((B, H, W, 3), float("nan"), device="cuda:0") for _ in range(5)]
    gates = [torch.full((B, H, W, 4), float('nan'), device='cuda:0') for _ in range(5)]
    c0 = 0
    for s_ in range(5):  # segment s_ feeds consumers s_ .. 4; it is the last input of consumer s_
        cons = (L.GateConsumer * (5 - s_))()
        for n_, j in enumerate(range(s_, 5)):
            for k in range(3):
                cons[n_].w[k] = gw[j][k].data_ptr() + c0 * 9 * 4
                cons[n_].b[k] = gb[j][k].data_ptr()
            cons[n_].logits = logits[j].data_ptr()
            cons[n_].gates = gates[j].data_ptr() if j == s_ else None
            cons[n_].first = 1 if s_ == 0 else 0
        L.check(lib.sr_fbm_gates_accumulate(1, B, H, W, mk(xd[s_]), 5 - s_, cons, st(L)))
        c0 += segc[s_]
    torch.cuda.synchronize()
    for j in range(5):
        xcat = torch.cat(xs[:j + 1], dim=1).double()
        wts = [gw[j][k].cpu().to(torch.bfloat16).double().view(1, cins[j], 3, 3) for k in range(3)]
        gref = fr.fbm_gates(xcat, wts, [gb[j][k].cpu().double() for k in range(3)]).float()
        got = gates[j].cpu()[..., :3].permute(0, 3, 1, 2)
        assert (got - gref).abs().max().item() < 3e-5, j


def ref_conv(xs, w, bias=None, lrelu=0, r1=None, s1=0.0, r2=None, s2=0.0, bands=None, gates=None, ups=0,
             per_sample=False):
    x = torch.cat(xs, 1).double()
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    if bands is not None:
        l2, l4, l8 = [torch.cat([b[k] for b in bands], 1).double() for k in range(3)]
        g = gates.double()
        x = g[:, 0:1] * x + (g[:, 1:2] - g[:, 0:1]) * l2 + (g[:, 2:3] - g[:, 1:2]) * l4 + (1 - g[:, 2:3]) * l8
    if per_sample:
        y = torch.stack([F.conv2d(x[i:i + 1], w[i].double(), padding=1)[0] for i in range(x.shape[0])])
    else:
        y = F.conv2d(x, w.double(), padding=1)
    if bias is not None:
        y = y + bias.double().view(1, -1, 1, 1)
    if lrelu:
        y = F.leaky_relu(y, 0.2)
    if r1 is not None:
        y = y * s1 + r1.double()
    if r2 is not None:
        y = y * s2 + r2.double()
    return y.float()


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(gp.signed_input(shape, seed)) * scale


@pytest.mark.parametrize('cin,cout,h,w', [(3, 8, 9, 11), (8, 3, 17, 16), (64, 1, 20, 35), (12, 4, 33, 18), (20, 24, 16, 16)])
def test_conv_generic_fp32(L, cin, cout, h, w):
    x = rnd((2, cin, h, w), 31)
    wt = rnd((cout, cin, 3, 3), 32, 0.2)
    b = rnd((cout,), 33)
    r1 = rnd((2, cout, h, w), 34)
    kw = dict(bias=b, lrelu=1, r1=r1, s1=0.2)
    got = run_conv(L, torch.float32, [x], wt, **kw)
    assert (got - ref_conv([x], wt, **kw)).abs().max().item() < 1e-5


def test_conv_generic_fp32_concat_bands_persample_ups(L):
    B, h, w = 2, 10, 14
    xs = [rnd((B, 16, h, w), 41), rnd((B, 8, h, w), 42)]
    H, W = 2 * h, 2 * w
    bands = [[rnd((B, x.shape[1], H, W), 43 + 3 * i + k) for k in range(3)] for i, x in enumerate(xs)]
    gates = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(5)) * 2
    wt = rnd((B, 24, 24, 3, 3), 50, 0.2)
    kw = dict(bias=rnd((24,), 51), bands=bands, gates=gates, ups=1, per_sample=True, r1=rnd((B, 24, H, W), 52), s1=0.2,
              r2=rnd((B, 24, H, W), 53), s2=0.2)
    got = run_conv(L, torch.float32, xs, wt, **kw)
    assert (got - ref_conv(xs, wt, **kw)).abs().max().item() < 1e-5


def bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize('cin,cout,h,w', [(32, 32, 16, 32), (64, 32, 19, 45), (64, 64, 33, 70), (96, 32, 40, 48), (192, 64, 17, 31),
                                          (64, 1, 37, 50), (64, 3, 16, 33)])  # the last two: thin tail (conv_last)
def test_conv_mfma_bf16_plain(L, cin, cout, h, w):
    """MFMA kernel against an fp64 conv of the same bf16-rounded operands: only fp32 accumulation order and the
    final bf16 rounding differ -> tolerance 2^-8 relative to the output magnitude."""
    B = 2
    segs = [64] + [32] * ((cin - 64) // 32) if cin >= 64 else [cin]
    xs = [bf(rnd((B, c, h, w), 60 + i)) for i, c in enumerate(segs)]
    wt = bf(rnd((cout, cin, 3, 3), 70, 0.1))
    b = rnd((cout,), 71)
    r1 = bf(rnd((B, cout, h, w), 72))
    kw = dict(bias=b, lrelu=1, r1=r1, s1=0.2)
    got = run_conv(L, torch.bfloat16, xs, wt, **kw)
    ref = ref_conv(xs, wt, **kw)
    tol = 2.0 ** -8 * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < tol
    # and the VALU kernel on the same bf16 data agrees too
    L.lib().sr_debug_force_generic_conv(1)
    try:
        got2 = run_conv(L, torch.bfloat16, xs, wt, **kw)
    finally:
        L.lib().sr_debug_force_generic_conv(0)
    assert (got2 - ref).abs().max().item() < tol


@pytest.mark.parametrize('segs,cout,h,w,ups,cb,two_res', [
    ([64], 64, 50, 70, 0, 32, True),        # SRVGG / conv_body shape: 24-row tiles, ragged bottom and right edges
    ([64], 64, 13, 21, 1, 8, False),        # conv_up1 / conv_up2: nearest-2x folded into the copies, 8-channel blocks
    ([64, 32, 32], 32, 37, 45, 0, 32, False),  # dense-block conv3 of a plain RRDBNet: three segments, 32-row tiles
    ([64, 32, 32, 32], 32, 9, 33, 0, 0, False),  # 160 -> 32: 16-row tiles (weights leave room for 4-row strips only), NHWC
    ([32], 32, 40, 31, 0, 32, True)])
def test_conv_plain_wave_autonomous_kernel(L, segs, cout, h, w, ups, cb, two_res):
    """conv_plain.hip (shared weights, no bands, weights resident in LDS) against an fp64 conv of the same bf16 operands and
    against conv.hip's producer / consumer kernel on the same device tensors (sr_debug_no_plain_kernel): layouts, the
    upsample fold, PReLU slopes, residual operands, tile edges."""
    lib = L.lib()
    B = 2
    H, W = h << ups, w << ups
    cin = sum(segs)
    xs = [bf(rnd((B, c, h, w), 160 + i)) for i, c in enumerate(segs)]
    wt = bf(rnd((cout, cin, 3, 3), 170, 0.1))
    bias = rnd((cout,), 171)
    slope = torch.rand(cout, generator=torch.Generator().manual_seed(172)) * 0.5
    r1, r2 = bf(rnd((B, cout, H, W), 173)), bf(rnd((B, cout, H, W), 174))
    lay = (lambda t: blocked(t, torch.bfloat16, cb)) if cb else (lambda t: nhwc(t, torch.bfloat16))
    mk = L.blocked_view if cb else L.view
    xd = [lay(x) for x in xs]
    cc = L.Concat()
    cc.nseg, cc.ups = len(xs), ups
    for i, t in enumerate(xd):
        cc.x[i] = mk(t)
    wd = dev(wt.permute(2, 3, 0, 1).reshape(9, cout, cin)).contiguous().to(torch.bfloat16)
    bd, sd_ = dev(bias), dev(slope)
    r1d, r2d = lay(r1), lay(r2)
    ep = L.Epilogue()
    ep.bias, ep.slope = bd.data_ptr(), sd_.data_ptr()
    ep.r1, ep.s1 = mk(r1d), 0.2
    if two_res:
        ep.r2, ep.s2 = mk(r2d), 0.3
    outs = []
    for off in (0, 1):
        out = lay(torch.full((B, cout, H, W), float('nan')))
        lib.sr_debug_no_plain_kernel(off)
        try:
            L.check(lib.sr_conv3x3(1, B, H, W, C.byref(cc), None, wd.data_ptr(), 0, 0, cout, C.byref(ep), mk(out), st(L)))
            torch.cuda.synchronize()
        finally:
            lib.sr_debug_no_plain_kernel(0)
        outs.append(unblocked(out) if cb else nchw(out))
    x = torch.cat(xs, 1).double()
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    y = F.conv2d(x, wt.double(), padding=1) + bias.double().view(1, -1, 1, 1)
    y = torch.where(y > 0, y, y * slope.double().view(1, -1, 1, 1))
    y = y * 0.2 + r1.double()
    if two_res:
        y = y * 0.3 + r2.double()
    tol = 2.0 ** -8 * max(1.0, y.abs().max().item())
    assert torch.isfinite(outs[0]).all()
    assert (outs[0].double() - y).abs().max().item() < tol
    # same products, same fp32 accumulation per output: the two kernels differ by summation order at most
    assert (outs[0] - outs[1]).abs().max().item() <= 2.0 ** -7 * max(1.0, y.abs().max().item())


@pytest.mark.parametrize('cb', [8, 32])
def test_conv_and_gates_channel_blocked_layout(L, cb):
    """bf16 MFMA conv + gates with channel-blocked activations, bands and output, NHWC upsampled source."""
    lib = L.lib()
    B, h, w, cin, cout = 2, 12, 20, 64, 64
    H, W = 2 * h, 2 * w
    x = bf(rnd((B, cin, h, w), 110))
    bands = [bf(rnd((B, cin, H, W), 111 + k)) for k in range(3)]
    keep = []
    sd = gp.fdconv_params(cin, cout, 9)
    lp, _ = make_lp(L, sd, cin, cout, torch.bfloat16, keep)
    xn = nhwc(x, torch.bfloat16)
    bb = [blocked(b_, torch.bfloat16, cb) for b_ in bands]
    cc = L.Concat()
    cc.nseg, cc.ups, cc.bands = 1, 1, 1
    cc.x[0] = L.view(xn)
    cc.l2[0], cc.l4[0], cc.l8[0] = [L.blocked_view(b_) for b_ in bb]
    gates = torch.full((B, H, W, 4), float('nan'), device='cuda:0')
    L.check(lib.sr_fbm_gates(1, B, H, W, C.byref(cc), C.byref(lp), gates.data_ptr(), st(L)))
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    xu = F.interpolate(x, scale_factor=2, mode='nearest').double()
    gref = fr.fbm_gates(xu, [t['FBM.freq_weight_conv_list.%d.weight' % i].to(torch.bfloat16).double() for i in range(3)],
                        [t['FBM.freq_weight_conv_list.%d.bias' % i].double() for i in range(3)]).float()
    torch.cuda.synchronize()
    g3 = gates.cpu()[..., :3].permute(0, 3, 1, 2)
    assert (g3 - gref).abs().max().item() < 2e-5
    wt = bf(rnd((B, cout, cin, 3, 3), 115, 0.1))
    wd = dev(wt.permute(0, 3, 4, 1, 2).reshape(B, 9, cout, cin)).contiguous().to(torch.bfloat16)
    r1 = bf(rnd((B, cout, H, W), 116))
    r1b = blocked(r1, torch.bfloat16, cb)
    ep = L.Epilogue()
    ep.lrelu = 1
    ep.r1, ep.s1 = L.blocked_view(r1b), 0.2
    out = torch.full((B, cout // cb, H, W, cb), float('nan'), device='cuda:0', dtype=torch.bfloat16)
    L.check(lib.sr_conv3x3(1, B, H, W, C.byref(cc), gates.data_ptr(), wd.data_ptr(), 9 * cout * cin, 0, cout, C.byref(ep),
                           L.blocked_view(out), st(L)))
    torch.cuda.synchronize()
    ref = ref_conv([x], wt, lrelu=1, r1=r1, s1=0.2, bands=[bands], gates=g3, ups=1, per_sample=True)
    assert (unblocked(out) - ref).abs().max().item() < 2.0 ** -7 * max(1.0, ref.abs().max().item())


def test_conv_mfma_bf16_bands_persample_ups(L):
    B, h, w = 2, 12, 20
    xs = [bf(rnd((B, 64, h, w), 80)), bf(rnd((B, 32, h, w), 81))]
    H, W = 2 * h, 2 * w
    bands = [[bf(rnd((B, x.shape[1], H, W), 82 + 3 * i + k)) for k in range(3)] for i, x in enumerate(xs)]
    gates = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(6)) * 2
    wt = bf(rnd((B, 64, 96, 3, 3), 90, 0.1))
    kw = dict(bias=rnd((64,), 91), bands=bands, gates=gates, ups=1, per_sample=True, r1=bf(rnd((B, 64, H, W), 92)), s1=0.2,
              r2=bf(rnd((B, 64, H, W), 93)), s2=0.2)
    got = run_conv(L, torch.bfloat16, xs, wt, **kw)
    ref = ref_conv(xs, wt, **kw)
    # the modulated operand is rounded to bf16 once more before the MFMA: 2^-7 relative
    assert (got - ref).abs().max().item() < 2.0 ** -7 * max(1.0, ref.abs().max().item())


def test_fbm_lowpass_mfma_large_activations(L):
    """fp16 headroom of the DFT-as-GEMM intermediates: a 512 x 512 plane with a DC level of 3000 (coefficients up to
    512 x that in an orthonormal basis) must come out finite and as accurate, relative to its size, as an O(1) plane."""
    lib = L.lib()
    g = torch.Generator().manual_seed(41)
    x = bf(3000.0 + 1000.0 * torch.randn(1, 32, 512, 512, generator=g))
    B, Cc, H, W = x.shape
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(H, W, C.byref(plan)))
    xd = blocked(x, torch.bfloat16, 32)
    bands = [torch.full((B, Cc // 32, H, W, 32), float('nan'), device='cuda:0', dtype=torch.bfloat16) for _ in range(3)]
    pooled = torch.full((B, Cc), float('nan'), device='cuda:0')
    ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda:0')
    L.check(lib.sr_fbm_lowpass(plan, 1, B, L.blocked_view(xd), 0, L.blocked_view(bands[0]), L.blocked_view(bands[1]),
                               L.blocked_view(bands[2]), pooled.data_ptr(), Cc, ws.data_ptr(), ws.numel(), st(L)))
    torch.cuda.synchronize()
    lib.sr_fbm_plan_destroy(plan)
    ref = fr.fbm_lowpass(x.double()).float()
    for k in range(3):
        got = unblocked(bands[k])
        assert bool(torch.isfinite(got).all()), k
        scale = ref[k].abs().max().item()
        assert (got - ref[k]).abs().max().item() < 2.0 ** -7 * scale, k
    assert (pooled.cpu() - x.mean(dim=(2, 3))).abs().max().item() < 1.0
