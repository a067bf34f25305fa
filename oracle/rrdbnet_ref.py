"""Oracle restatement of RRDBNet (Real-ESRGAN x4 generator), CPU / PyTorch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference imports RRDBNet from the pip package basicsr==1.4.2
(/root/reference/requirements.txt:8; call sites train/train_realESRGAN_fdconv_1.py:100,
train/test_realESRGAN_fdconv.py:19, Real-ESRGAN-master/inference_realesrgan.py:60) which is
NOT vendored and not installed here.  This file restates its published architecture
(basicsr/archs/rrdbnet_arch.py + arch_util.py: ResidualDenseBlock, RRDB, RRDBNet,
pixel_unshuffle) -- PARITY UNPINNED by any reference fixture; pinned by construction because
it is a fixed composition of torch's own conv2d / leaky_relu / cat / nearest-2x.

`rrdbnet_forward` runs the network functionally from a state_dict; every conv whose
state holds a `dft_weight` is evaluated as an FDConv layer through oracle.fdconv_ref (this is
what train/train_realESRGAN_fdconv_1.py:82-95 `replace_conv_with_fdconv` produces).
"""
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fdconv_ref import fdconv_forward, sub_state


def pixel_unshuffle(x, scale):
    """basicsr arch_util.pixel_unshuffle: (B,C,H*s,W*s) -> (B,C*s*s,H,W), channel = c*s*s + dy*s + dx."""
    b, c, hh, hw = x.shape
    h, w = hh // scale, hw // scale
    return x.view(b, c, h, scale, w, scale).permute(0, 1, 3, 5, 2, 4).reshape(b, c * scale * scale, h, w)


def num_blocks_of(sd):
    idx = [int(m.group(1)) for k in sd for m in [re.match(r'body\.(\d+)\.', k)] if m]
    return max(idx) + 1 if idx else 0


def rrdbnet_forward(sd, x, scale=4, training=False, dtype=None, gmaps=None, emulate_bf16=False):
    """RRDBNet.forward from a flat state_dict.  x: (B,C,H,W).

    emulate_bf16: every tensor the network stores between layers is rounded to bfloat16 (as any bf16 evaluation of the
    network, e.g. the reference under autocast, must), arithmetic stays in `dtype`: the yardstick the bf16 GPU path
    is compared with where the network itself amplifies rounding."""
    dtype = dtype or x.dtype
    x = x.to(dtype)
    gmaps = gmaps if gmaps is not None else {}
    q = (lambda t: t.to(torch.bfloat16).to(dtype)) if emulate_bf16 else (lambda t: t)
    x = q(x)

    def conv(name, t):
        return fdconv_forward(t, sub_state(sd, name), gmap=gmaps.get(name), training=training, dtype=dtype,
                              emulate_bf16=emulate_bf16)

    lrelu = lambda t: q(F.leaky_relu(t, 0.2))
    if scale == 2:
        x = pixel_unshuffle(x, 2)
    elif scale == 1:
        x = pixel_unshuffle(x, 4)
    feat = q(conv('conv_first', x))
    body = feat
    for i in range(num_blocks_of(sd)):
        blk_in = body
        for j in (1, 2, 3):
            p = 'body.%d.rdb%d.' % (i, j)
            t0 = body
            t1 = lrelu(conv(p + 'conv1', t0))
            t2 = lrelu(conv(p + 'conv2', torch.cat((t0, t1), 1)))
            t3 = lrelu(conv(p + 'conv3', torch.cat((t0, t1, t2), 1)))
            t4 = lrelu(conv(p + 'conv4', torch.cat((t0, t1, t2, t3), 1)))
            t5 = conv(p + 'conv5', torch.cat((t0, t1, t2, t3, t4), 1))
            body = q(t5 * 0.2 + t0) if j < 3 else q((t5 * 0.2 + t0) * 0.2 + blk_in)
    feat = q(feat + conv('conv_body', body))
    feat = lrelu(conv('conv_up1', F.interpolate(feat, scale_factor=2, mode='nearest')))
    feat = lrelu(conv('conv_up2', F.interpolate(feat, scale_factor=2, mode='nearest')))
    return q(conv('conv_last', lrelu(conv('conv_hr', feat))))


class RRDBNetRef(nn.Module):
    """Plain (non-FDConv) RRDBNet as an nn.Module with basicsr's state_dict layout.

    Used as the `model` handed to the *reference* RealESRGANer when golden vectors are made,
    and by the CPU baseline.  forward() delegates to rrdbnet_forward.
    """

    def __init__(self, num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
        super().__init__()
        self.scale = scale
        cin = num_in_ch * (4 if scale == 2 else 16 if scale == 1 else 1)
        self.conv_first = nn.Conv2d(cin, num_feat, 3, 1, 1)
        self.body = nn.Sequential()
        for i in range(num_block):
            blk = nn.Module()
            for j in (1, 2, 3):
                rdb = nn.Module()
                for k in range(1, 5):
                    setattr(rdb, 'conv%d' % k, nn.Conv2d(num_feat + (k - 1) * num_grow_ch, num_grow_ch, 3, 1, 1))
                rdb.conv5 = nn.Conv2d(num_feat + 4 * num_grow_ch, num_feat, 3, 1, 1)
                setattr(blk, 'rdb%d' % j, rdb)
            self.body.add_module(str(i), blk)
        for n in ('conv_body', 'conv_up1', 'conv_up2', 'conv_hr'):
            setattr(self, n, nn.Conv2d(num_feat, num_feat, 3, 1, 1))
        self.conv_last = nn.Conv2d(num_feat, num_out_ch, 3, 1, 1)

    def forward(self, x):
        return rrdbnet_forward(self.state_dict(), x, self.scale, self.training)
