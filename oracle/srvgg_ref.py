"""Oracle restatement of SRVGGNetCompact (realesr-animevideov3 family), CPU / PyTorch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference followed: /root/reference/Real-ESRGAN-master/realesrgan/archs/srvgg_arch.py
  * __init__ :22-58  (body = conv, act, (conv, act) x num_conv, conv -> num_out_ch * upscale^2; PixelShuffle)
  * forward  :60-69  (body, pixel shuffle, + nearest-upsampled input)
Written functionally from a flat state_dict (no module tree); pinned by tests/golden/srvgg.npz, outputs of the
reference class itself (tests/golden/make_golden_r2.py).
"""
import torch
import torch.nn.functional as F


def srvgg_forward(sd, x, upscale=4, act_type='prelu'):
    """sd: flat state_dict with keys body.{i}.weight / .bias; x: (B, C, H, W)."""
    idx = sorted({int(k.split('.')[1]) for k in sd})
    convs = [i for i in idx if sd['body.%d.weight' % i].dim() == 4]
    out = x
    for n, i in enumerate(convs):
        out = F.conv2d(out, sd['body.%d.weight' % i].to(x.dtype), sd['body.%d.bias' % i].to(x.dtype), padding=1)
        if n == len(convs) - 1:
            break
        if act_type == 'prelu':
            a = sd['body.%d.weight' % (i + 1)].to(x.dtype).view(1, -1, 1, 1)
            out = torch.where(out >= 0, out, a * out)
        elif act_type == 'relu':
            out = torch.relu(out)
        elif act_type == 'leakyrelu':
            out = F.leaky_relu(out, 0.1)
        else:
            raise ValueError(act_type)
    b, c, h, w = out.shape
    r = upscale
    # PixelShuffle: channel c*r*r + dy*r + dx -> pixel (h*r + dy, w*r + dx) of channel c
    out = out.view(b, c // (r * r), r, r, h, w).permute(0, 1, 4, 2, 5, 3).reshape(b, c // (r * r), h * r, w * r)
    base = x.repeat_interleave(r, dim=2).repeat_interleave(r, dim=3)
    return out + base
