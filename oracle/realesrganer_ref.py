"""Oracle restatement of RealESRGANer (tile / pad / enhance), CPU / PyTorch + numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference followed: /root/reference/Real-ESRGAN-master/realesrgan/utils.py
  * pre_process  :88-111   * process :113-115   * tile_process :117-180
  * post_process :182-191  * enhance :193-263
cv2 is not installed in the build image, so the colour conversions enhance() delegates to
cv2.cvtColor are restated here from OpenCV's documented float formulas (COLOR_BGR2GRAY =
0.114 B + 0.587 G + 0.299 R) -- that part is PARITY UNPINNED; the pad / tile / crop path is
pinned by tests/golden/realesrganer_*.npz produced by the reference class itself.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def tile_windows(height, width, tile, tile_pad):
    """The (input window, output paste, tile-local crop) triples of utils.py:134-180, in loop order."""
    out = []
    for y in range(math.ceil(height / tile)):
        for x in range(math.ceil(width / tile)):
            x0, y0 = x * tile, y * tile
            x1, y1 = min(x0 + tile, width), min(y0 + tile, height)
            px0, px1 = max(x0 - tile_pad, 0), min(x1 + tile_pad, width)
            py0, py1 = max(y0 - tile_pad, 0), min(y1 + tile_pad, height)
            out.append(dict(inp=(py0, py1, px0, px1), core=(y0, y1, x0, x1), off=(y0 - py0, x0 - px0)))
    return out


def pre_process(img_hwc, scale, pre_pad):
    """utils.py:88-111 -> (tensor 1CHW, mod_pad_h, mod_pad_w, mod_scale)."""
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(img_hwc, (2, 0, 1)))).float().unsqueeze(0)
    if pre_pad != 0:
        t = F.pad(t, (0, pre_pad, 0, pre_pad), 'reflect')
    mod_scale = {2: 2, 1: 4}.get(scale)
    mh = mw = 0
    if mod_scale is not None:
        h, w = t.shape[-2:]
        mh = (mod_scale - h % mod_scale) % mod_scale
        mw = (mod_scale - w % mod_scale) % mod_scale
        t = F.pad(t, (0, mw, 0, mh), 'reflect')
    return t, mh, mw, mod_scale


def tile_process(model_fn, img, scale, tile, tile_pad):
    b, c, h, w = img.shape
    out = img.new_zeros((b, c, h * scale, w * scale))
    for win in tile_windows(h, w, tile, tile_pad):
        py0, py1, px0, px1 = win['inp']
        y0, y1, x0, x1 = win['core']
        oy, ox = win['off']
        o = model_fn(img[:, :, py0:py1, px0:px1])
        out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = \
            o[:, :, oy * scale:(oy + y1 - y0) * scale, ox * scale:(ox + x1 - x0) * scale]
    return out


def post_process(out, scale, pre_pad, mh, mw, mod_scale):
    if mod_scale is not None:
        h, w = out.shape[-2:]
        out = out[:, :, 0:h - mh * scale, 0:w - mw * scale]
    if pre_pad != 0:
        h, w = out.shape[-2:]
        out = out[:, :, 0:h - pre_pad * scale, 0:w - pre_pad * scale]
    return out


def run_rgb(model_fn, img_hwc, scale, tile, tile_pad, pre_pad):
    t, mh, mw, ms = pre_process(img_hwc, scale, pre_pad)
    o = tile_process(model_fn, t, scale, tile, tile_pad) if tile > 0 else model_fn(t)
    return post_process(o, scale, pre_pad, mh, mw, ms)


def bgr2gray(img):
    return (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).astype(np.float32)


def enhance(model_fn, img, scale, tile=0, tile_pad=10, pre_pad=10, outscale=None, alpha_upsampler='realesrgan'):
    """utils.py:193-263 (outscale must be None or == scale: the LANCZOS4 resize is not restated)."""
    assert outscale is None or float(outscale) == float(scale)
    img = img.astype(np.float32)
    max_range = 65535 if np.max(img) > 256 else 255
    img = img / max_range
    alpha = None
    if img.ndim == 2:
        mode = 'L'
        img = np.stack([img] * 3, -1)
    elif img.shape[2] == 4:
        mode = 'RGBA'
        alpha = img[:, :, 3]
        img = img[:, :, 2::-1]
        if alpha_upsampler == 'realesrgan':
            alpha = np.stack([alpha] * 3, -1)
    else:
        mode = 'RGB'
        img = img[:, :, ::-1]

    def net(i):
        o = run_rgb(model_fn, i, scale, tile, tile_pad, pre_pad)
        o = o.squeeze(0).float().clamp_(0, 1).numpy()
        return np.transpose(o[[2, 1, 0]], (1, 2, 0))

    out = net(img)
    if mode == 'L':
        out = bgr2gray(out)
    if mode == 'RGBA':
        if alpha_upsampler == 'realesrgan':
            oa = bgr2gray(net(alpha))
        else:
            a = torch.from_numpy(np.ascontiguousarray(alpha))[None, None]
            oa = F.interpolate(a, scale_factor=scale, mode='bilinear', align_corners=False)[0, 0].numpy()
        out = np.concatenate([out, oa[..., None]], -1)
    if max_range == 65535:
        return (out * 65535.0).round().astype(np.uint16), mode
    return (out * 255.0).round().astype(np.uint8), mode
