"""Post-network head of the owner scripts, on libsr_amd (include/sr_amd.h "network ends").

  * bilinear_resize(x, size, sigmoid=False, expm1_scale=None): F.interpolate(x, size, mode='bilinear',
    align_corners=False) (train/train_realESRGAN_fdconv_1.py:115, train/test_realESRGAN_fdconv.py:51) with CustomRRDBNet's
    Sigmoid (train/train_realESRGAN_fdconv.py:111-115) folded in front and the expm1 de-normalisation
    (train/test_realESRGAN_fdconv.py:55-57) folded behind;
  * resize_l1_loss(pred, target, weighted=False, sigmoid=False): the training loss of the same scripts (nn.L1Loss at
    fdconv_1.py:116, weighted L1 at train_realESRGAN_fdconv.py:124-126) of the resized prediction, differentiable with
    respect to `pred` (one fused kernel computes the loss and its gradient).
"""
import torch

from . import _lib as L


def bilinear_resize(x, size, sigmoid=False, expm1_scale=None, out_dtype=None):
    L.require_gpu(x, 'bilinear_resize')
    x = x.contiguous()
    B, Cc, Hs, Ws = x.shape
    Hd, Wd = size
    out_dtype = out_dtype or x.dtype
    y = torch.empty(B, Cc, Hd, Wd, dtype=out_dtype, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().sr_bilinear_resize(x.data_ptr(), L.dtype_code(x.dtype), y.data_ptr(), L.dtype_code(out_dtype), B * Cc, Hs, Ws,
                                           Hd, Wd, int(bool(sigmoid)), int(expm1_scale is not None),
                                           float(expm1_scale if expm1_scale is not None else 1.0), L.stream_ptr(x.device)),
                'sr_bilinear_resize')
    return y


class _ResizeL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weighted, sigmoid):
        L.require_gpu(pred, 'resize_l1_loss')
        if pred.dtype != torch.float32 or target.dtype != torch.float32:
            raise TypeError('resize_l1_loss computes in float32')
        # read BEFORE .contiguous(): forward runs in no-grad mode, so the contiguous copy of a channels_last prediction
        # (any network with more than one output channel) has requires_grad = False
        need_grad = ctx.needs_input_grad[0]
        pred, target = pred.contiguous(), target.contiguous()
        B, Cc, Hs, Ws = pred.shape
        Hd, Wd = target.shape[-2:]
        lib = L.lib()
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred) if need_grad else None
        ws = torch.empty(lib.sr_resize_l1_workspace_bytes(), dtype=torch.uint8, device=pred.device)
        with torch.cuda.device(pred.device):
            L.check(lib.sr_resize_l1_loss(pred.data_ptr(), target.data_ptr(), B * Cc, Hs, Ws, Hd, Wd, int(bool(weighted)),
                                          int(bool(sigmoid)), loss.data_ptr(), L.ptr(grad), ws.data_ptr(), ws.numel(),
                                          L.stream_ptr(pred.device)), 'sr_resize_l1_loss')
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None


def resize_l1_loss(pred, target, weighted=False, sigmoid=False):
    return _ResizeL1.apply(pred, target, weighted, sigmoid)
