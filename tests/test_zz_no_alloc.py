"""The one entry point that allocates nothing (sr_fbm_lowpass_static): its planes and pooled means must equal
sr_fbm_lowpass's bit for bit, on two calls in a row, the second with a smaller input; sizes beyond the fixed
scratch are refused.  Collected last (file name) so that a failure here hides no other test."""
import ctypes as C

import pytest
import torch

import golden_params as gp

pytestmark = pytest.mark.gpu


def _both(L, plan, dtype, x, B, H, W, Cc):
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    code = L.dtype_code(dtype)
    out = []
    for static in (False, True):
        bands = [torch.full((B, H, W, Cc), 7.0, dtype=dtype, device='cuda') for _ in range(3)]
        pooled = torch.full((B, Cc), 7.0, device='cuda')
        if static:
            L.check(lib.sr_fbm_lowpass_static(plan, code, B, L.view(x), 0, L.view(bands[0]), L.view(bands[1]), L.view(bands[2]),
                                              pooled.data_ptr(), Cc, st), 'sr_fbm_lowpass_static')
        else:
            ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, Cc), dtype=torch.uint8, device='cuda')
            L.check(lib.sr_fbm_lowpass(plan, code, B, L.view(x), 0, L.view(bands[0]), L.view(bands[1]), L.view(bands[2]),
                                       pooled.data_ptr(), Cc, ws.data_ptr(), ws.numel(), st), 'sr_fbm_lowpass')
        torch.cuda.synchronize()
        out.append((bands, pooled))
    return out


def test_lowpass_static_equals_workspace_variant_twice():
    from super_resolution_amd import _lib as L
    lib = L.lib()
    assert lib.sr_fbm_static_scratch_bytes() == 64 << 20
    # first call: bf16, 2 x 64 channels x 128 x 128 (MFMA path); second, smaller: fp32, 1 x 6 x 45 x 70 (any-size FFT);
    # third: bf16 again at 64 x 64 after the scratch has held other data
    for dtype, (B, Cc, H, W) in ((torch.bfloat16, (2, 64, 128, 128)), (torch.float32, (1, 6, 45, 70)), (torch.bfloat16, (1, 32, 64, 64))):
        plan = C.c_void_p()
        L.check(lib.sr_fbm_plan_create(H, W, C.byref(plan)))
        assert lib.sr_fbm_workspace_bytes(plan, B, Cc) <= lib.sr_fbm_static_scratch_bytes()
        x = torch.from_numpy(gp.signed_input((B, H, W, Cc), 5)).to(dtype).cuda()
        (b0, p0), (b1, p1) = _both(L, plan, dtype, x, B, H, W, Cc)
        for k in range(3):
            assert torch.equal(b0[k], b1[k]), (dtype, k)
            assert float(b1[k].float().abs().max()) < 6.0  # the 7.0 fill is gone
        assert torch.equal(p0, p1)
        assert (p1 - x.float().mean(dim=(1, 2))).abs().max().item() < 2e-3
        lib.sr_fbm_plan_destroy(plan)


def test_lowpass_static_rejects_sizes_beyond_the_scratch():
    from super_resolution_amd import _lib as L
    lib = L.lib()
    plan = C.c_void_p()
    L.check(lib.sr_fbm_plan_create(512, 512, C.byref(plan)))
    B, Cc = 16, 64
    assert lib.sr_fbm_workspace_bytes(plan, B, Cc) > lib.sr_fbm_static_scratch_bytes()
    x = torch.zeros(1, device='cuda')  # never touched: the call must fail before any launch
    v = L.View(x.data_ptr(), Cc, 0, Cc, 0, 0)
    rc = lib.sr_fbm_lowpass_static(plan, 1, B, v, 0, v, v, v, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == -2 and 'workspace' in lib.sr_strerror(rc).decode()
    lib.sr_fbm_plan_destroy(plan)
