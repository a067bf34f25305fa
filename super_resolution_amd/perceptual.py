"""Perceptual (VGG19-feature L1) loss of the Real-ESRGAN recipe on libsr_amd -- BASELINE configs[3] "L1 + perceptual".

Reference: Real-ESRGAN-master/options/train_realesrgan_x4plus.yml:133-149 (basicsr `PerceptualLoss`: vgg19, features taken
BEFORE the ReLU at conv1_2, conv2_2, conv3_4, conv4_4, conv5_4 with weights 0.1, 0.1, 1, 1, 1, criterion l1,
use_input_norm, perceptual_weight 1, style_weight 0), used by realesrgan/models/realesrgan_model.py:221-227.  basicsr
itself is not vendored; the extractor below is its `VGGFeatureExtractor` restated (torchvision's vgg19 `features` layout,
so a torchvision checkpoint's `features.N.weight` keys load with `load_state_dict`).  No pretrained weights exist offline:
the bench and the tests use a seeded random initialisation of the same architecture.

`VGG19Features.forward` is plain torch (the CPU yardstick of the tests).  `perceptual_loss` is the product path: the
sixteen 3x3 convolutions run on sr_conv3x3 -- forward for the prediction and the target, and with transposed weights for
the gradient with respect to the prediction (the extractor is frozen: no weight gradient) --, ReLU / 2x2 max-pool, the
feature L1 terms, the input normalisation and the optional resize are the small kernels of csrc/head.hip.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L

VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
DEFAULT_LAYER_WEIGHTS = {'conv1_2': 0.1, 'conv2_2': 0.1, 'conv3_4': 1.0, 'conv4_4': 1.0, 'conv5_4': 1.0}  # yml:135-143


def _layer_names():
    names, blk, i = [], 1, 1
    for v in VGG19_CFG:
        if v == 'M':
            names.append('pool%d' % blk)
            blk, i = blk + 1, 1
        else:
            names += ['conv%d_%d' % (blk, i), 'relu%d_%d' % (blk, i)]
            i += 1
    return names


class VGG19Features(nn.Module):
    """vgg19 `features` up to conv5_4 (torchvision layout: `features.{0,2,5,...}.weight`).  forward(x) -> {name: feature}
    for the requested conv layers, taken before the ReLU; x: (B, 3, H, W) in [0, 1] (normalised inside when
    use_input_norm)."""

    def __init__(self, layer_names=tuple(DEFAULT_LAYER_WEIGHTS), use_input_norm=True, seed=None):
        super().__init__()
        self.names = _layer_names()
        self.layer_names = tuple(layer_names)
        last = max(self.names.index(n) for n in self.layer_names)
        mods, cin = [], 3
        for name, v in zip(self.names[:last + 1], [w for v in VGG19_CFG for w in ((v,) if v == 'M' else (v, 'R'))][:last + 1]):
            if v == 'M':
                mods.append(nn.MaxPool2d(2, 2))
            elif v == 'R':
                mods.append(nn.ReLU(inplace=False))
            else:
                mods.append(nn.Conv2d(cin, v, 3, 1, 1))
                cin = v
        self.features = nn.Sequential(*mods)
        self.names = self.names[:last + 1]
        self.use_input_norm = use_input_norm
        if seed is not None:  # He-normal weights, small biases: activations of O(1) through all sixteen layers
            g = torch.Generator().manual_seed(seed)
            with torch.no_grad():
                for m in self.features:
                    if isinstance(m, nn.Conv2d):
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (9 * m.in_channels)) ** 0.5)
                        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed = {}

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        if self.use_input_norm:
            mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
            std = torch.tensor(IMAGENET_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
            x = (x - mean) / std
        out = {}
        for name, m in zip(self.names, self.features):
            x = m(x)
            if name in self.layer_names:
                out[name] = x
        return out

    # ---- libsr_amd plumbing ----------------------------------------------------------------------------------------
    def packed(self, dtype):
        """[(name, cin, cout, weights [9][cout][cin], lo offset, transposed weights [9][cin][cout], lo offset, bias)] per conv."""
        convs = [(n, m) for n, m in zip(self.names, self.features) if isinstance(m, nn.Conv2d)]
        key = (dtype,) + tuple((m.weight.data_ptr(), m.weight._version) for _, m in convs)
        hit = self._packed.get('k')
        if hit is not None and hit[0] == key:
            return hit[1]
        out = []
        for name, m in convs:
            w = m.weight.data
            L.require_gpu(w, 'VGG19Features')
            cout, cin = m.out_channels, m.in_channels
            wp, lo = L.pack_plain_weight(w.float().contiguous(), dtype, cout, cin)
            # transposed, tap-mirrored weights: conv(dy, W^T) is the gradient with respect to the layer's input
            wt_src = w.float().flip(2, 3).transpose(0, 1).contiguous()
            wt, lot = L.pack_plain_weight(wt_src, dtype, cin, cout)
            out.append((name, cin, cout, wp, lo, wt, lot, m.bias.data.float().contiguous()))
        self._packed['k'] = (key, out)
        return out


def _conv(code, B, H, W, x, cin, w, lo, cout, bias, st):
    y = torch.empty(B, H, W, cout, dtype=x.dtype, device=x.device)
    cc = L.Concat()
    cc.nseg = 1
    cc.x[0] = L.view(x)
    ep = L.Epilogue()
    ep.bias = None if bias is None else bias.data_ptr()
    L.check(L.lib().sr_conv3x3(code, B, H, W, C.byref(cc), None, w.data_ptr(), 0, lo, cout, C.byref(ep), L.view(y), st), 'sr_conv3x3 (vgg)')
    return y


class _PerceptualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, vgg, weights, dtype):
        L.require_gpu(pred, 'perceptual_loss')
        if pred.dtype != torch.float32 or target.dtype != torch.float32:
            raise TypeError('perceptual_loss takes float32 images (it computes in `compute_dtype`)')
        lib, dev = L.lib(), pred.device
        need_grad = ctx.needs_input_grad[0]
        B, Cs, Hp, Wp = pred.shape
        Bt, Ct, H, W = target.shape
        if (B, Cs) != (Bt, Ct) or Cs not in (1, 3):
            raise ValueError('perceptual_loss: prediction %s and target %s must agree in batch and channels (1 or 3)'
                             % (tuple(pred.shape), tuple(target.shape)))
        code = L.dtype_code(dtype)
        layers = vgg.packed(dtype)
        names = vgg.names
        mean = (C.c_float * 3)(*IMAGENET_MEAN) if vgg.use_input_norm else None
        std = (C.c_float * 3)(*IMAGENET_STD) if vgg.use_input_norm else None
        with torch.cuda.device(dev):
            st = L.stream_ptr(dev)
            pred_c = pred.contiguous()
            resized = (Hp, Wp) != (H, W)
            if resized:  # the owner scripts resize the prediction to the target's size first (train_realESRGAN_fdconv_1.py:115)
                up = torch.empty(B, Cs, H, W, dtype=torch.float32, device=dev)
                L.check(lib.sr_bilinear_resize(pred_c.data_ptr(), 0, up.data_ptr(), 0, B * Cs, Hp, Wp, H, W, 0, 0, 1.0, st), 'sr_bilinear_resize')
            else:
                up = pred_c
            loss = torch.zeros((), dtype=torch.float32, device=dev)

            def run(img, keep):
                """conv outputs (pre-ReLU) of every layer for `img`; keep: hold all of them (the backward needs them)."""
                x = torch.empty(B, H, W, 3, dtype=dtype, device=dev)
                L.check(lib.sr_vgg_prep(code, B, Cs, H, W, img.data_ptr(), x.data_ptr(), mean, std, 0, st), 'sr_vgg_prep')
                feats, h, w, li = {}, H, W, 0
                pre = None
                for idx, name in enumerate(names):
                    if name.startswith('conv'):
                        _, cin, cout, wp, lo, _, _, bias = layers[li]
                        li += 1
                        pre = _conv(code, B, h, w, x, cin, wp, lo, cout, bias, st)
                        if keep or name in weights:
                            feats[name] = pre
                        if idx == len(names) - 1:
                            break
                        nxt = names[idx + 2] if idx + 2 < len(names) else ''
                        pool = 2 if nxt.startswith('pool') else 1
                        if pool == 2 and (h % 2 or w % 2):
                            raise ValueError('perceptual_loss: the image side must be divisible by 16')
                        x = torch.empty(B, h // pool, w // pool, cout, dtype=dtype, device=dev)
                        L.check(lib.sr_relu_pool(code, B, h // pool, w // pool, cout, pool, pre.data_ptr(), x.data_ptr(), st), 'sr_relu_pool')
                        h, w = h // pool, w // pool
                return feats

            with torch.no_grad():
                ft = run(target.contiguous(), False)
                fp = run(up, True)
                tap_grads = {}
                for name, wgt in weights.items():
                    a, b = fp[name], ft[name]
                    g = torch.empty_like(a) if need_grad else None
                    L.check(lib.sr_l1_feature_loss(code, a.numel(), a.data_ptr(), b.data_ptr(), float(wgt), loss.data_ptr(), L.ptr(g), st),
                            'sr_l1_feature_loss')
                    tap_grads[name] = g
                grad = None
                if need_grad:
                    # backward through the prediction's pass: conv^T, then ReLU / max-pool backward (+ the tap's own gradient)
                    conv_names = [n for n in names if n.startswith('conv')]
                    G = tap_grads.get(conv_names[-1])
                    if G is None:
                        G = torch.zeros_like(fp[conv_names[-1]])
                    for li in range(len(conv_names) - 1, -1, -1):
                        name = conv_names[li]
                        _, cin, cout, _, _, wt, lot, _ = layers[li]
                        h, w = fp[name].shape[1:3]
                        dxin = _conv(code, B, h, w, G, cout, wt, lot, cin, None, st)   # gradient wrt this layer's input
                        if li == 0:
                            G = dxin
                            break
                        prev = conv_names[li - 1]
                        src = fp[prev]
                        pool = src.shape[1] // h
                        G = torch.empty_like(src)
                        L.check(lib.sr_relu_pool_backward(code, B, h, w, src.shape[3], pool, src.data_ptr(), dxin.data_ptr(),
                                                          L.ptr(tap_grads.get(prev)), G.data_ptr(), st), 'sr_relu_pool_backward')
                    dup = torch.empty(B, Cs, H, W, dtype=torch.float32, device=dev)
                    L.check(lib.sr_vgg_prep(code, B, Cs, H, W, dup.data_ptr(), G.data_ptr(), mean, std, 1, st), 'sr_vgg_prep (adjoint)')
                    if resized:
                        grad = torch.empty(B, Cs, Hp, Wp, dtype=torch.float32, device=dev)
                        L.check(lib.sr_bilinear_resize_backward(dup.data_ptr(), grad.data_ptr(), B * Cs, Hp, Wp, H, W, st),
                                'sr_bilinear_resize_backward')
                    else:
                        grad = dup
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None, None


def perceptual_loss(pred, target, vgg, layer_weights=None, perceptual_weight=1.0, compute_dtype=torch.bfloat16):
    """perceptual_weight * sum_k layer_weights[k] * mean|vgg_k(pred') - vgg_k(target)|, pred' = pred resized (bilinear,
    align_corners=False) to the target's size when the sizes differ; differentiable with respect to `pred` (the loss and
    its gradient are computed in one pass).  pred / target: float32 (B, 1 or 3, H, W) images in [0, 1]; a one-channel
    image enters the extractor as three equal planes.  compute_dtype: bfloat16 (matrix cores) or float32 (parity path)."""
    lw = dict(DEFAULT_LAYER_WEIGHTS if layer_weights is None else layer_weights)
    missing = [k for k in lw if k not in vgg.layer_names]
    if missing:
        raise ValueError('perceptual_loss: the extractor does not expose %s' % missing)
    lw = {k: float(v) * float(perceptual_weight) for k, v in lw.items()}
    return _PerceptualFn.apply(pred, target, vgg, lw, compute_dtype)
