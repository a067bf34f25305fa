"""Checkpoint interop: spatial conv weights -> FDConv `dft_weight` parameters.

Follows the reference converter `FDConv-main/FDConv-main/tools/convert_to_fdconv.py:125-181` -- for every conv weight
(Cout, Cin, k, k) with min(Cout, Cin) >= 32 and k in {1, 3}

    dft_weight = stack(re, im)(rfft2(weight.permute(0, 2, 1, 3).reshape(Cout k, Cin k)))[None] / (min(Cout, Cin) // 2)

replaces `<layer>.weight` -- generalised from its ResNet / ViT key patterns to every 4-D `*.weight` of a state dict
(the RRDBNet keys `conv_first`, `body.{i}.rdb{j}.conv{k}`, `conv_body`, `conv_up1/2`, `conv_hr`, `conv_last`), and to the
`{'params' | 'params_ema': state_dict}` containers of Real-ESRGAN checkpoints (`realesrgan/utils.py:63-70`).  Host-side
torch only: this is a file-format tool, not part of the GPU path.

    python -m super_resolution_amd.convert RealESRGAN_x4plus.pth RealESRGAN_x4plus_fdconv.pth
"""
import sys

import torch

CH_THRES = 32          # convert_to_fdconv.py:127
K_THRES = (1, 3)       # convert_to_fdconv.py:128


def spatial_to_dft_weight(w):
    """(Cout, Cin, k, k) float tensor -> (1, Cout k, Cin k / 2 + 1, 2), the layout of `FDConv.dft_weight`."""
    if w.dim() == 2:
        w = w[..., None, None]
    cout, cin, k, _ = w.shape
    spec = torch.fft.rfft2(w.float().permute(0, 2, 1, 3).reshape(cout * k, cin * k), dim=(0, 1))
    out = torch.stack([spec.real, spec.imag], dim=-1)[None]
    norm = min(cout, cin) // 2
    return out / norm if norm > 0 else out


def dft_weight_to_spatial(dft_weight, cout, cin, k=3):
    """Inverse of `spatial_to_dft_weight` (what an FDConv layer with unit attention convolves with)."""
    spec = torch.complex(dft_weight[0, ..., 0], dft_weight[0, ..., 1]) * (min(cout, cin) // 2)
    w2 = torch.fft.irfft2(spec, s=(cout * k, cin * k), dim=(0, 1))
    return w2.reshape(cout, k, cin, k).permute(0, 2, 1, 3).contiguous()


def convert_state_dict(state, ch_thres=CH_THRES, k_thres=K_THRES):
    """Plain state dict (or a {'params'/'params_ema': ...} container) -> the same with every eligible conv `.weight`
    replaced by `.dft_weight`.  Returns (new_state, [(old_key, new_key), ...])."""
    for key in ('params_ema', 'params'):
        if isinstance(state, dict) and key in state and isinstance(state[key], dict):
            inner, conv = convert_state_dict(state[key], ch_thres, k_thres)
            out = dict(state)
            out[key] = inner
            return out, conv
    out, converted = {}, []
    for k, v in state.items():
        if k.endswith('.weight') and torch.is_tensor(v) and v.dim() == 4 and v.shape[2] == v.shape[3] \
                and min(v.shape[0], v.shape[1]) >= ch_thres and v.shape[2] in k_thres:
            nk = k[:-len('.weight')] + '.dft_weight'
            out[nk] = spatial_to_dft_weight(v)
            converted.append((k, nk))
        else:
            out[k] = v
    return out, converted


def load_plain_into_fdconv(model, state):
    """Load a plain RRDBNet checkpoint into a model whose convs were replaced by FDConv: spectra and biases come from the
    checkpoint, the modulation parameters (KSM_Global / KSM_Local / FBM) keep the model's current values -- what loading
    the reference converter's output with strict=False does.  Every checkpoint key must find its layer."""
    conv, _ = convert_state_dict(state)
    for key in ('params_ema', 'params'):
        if key in conv and isinstance(conv[key], dict):
            conv = conv[key]
            break
    own = model.state_dict()
    missing = [k for k in conv if k not in own]
    if missing:
        raise RuntimeError('keys without a layer in the model: %s' % missing[:5])
    for k, v in conv.items():
        if own[k].shape != v.shape:
            raise RuntimeError('size mismatch for %s: checkpoint %s, model %s' % (k, tuple(v.shape), tuple(own[k].shape)))
    own.update({k: v.to(own[k].dtype) for k, v in conv.items()})
    model.load_state_dict(own, strict=True)
    return model


def main(argv):
    if len(argv) != 3:
        print(__doc__)
        return 2
    state = torch.load(argv[1], map_location='cpu', weights_only=True)
    new, conv = convert_state_dict(state)
    for a, b in conv:
        print('%s -> %s' % (a, b))
    torch.save(new, argv[2])
    print('%d weights converted, saved to %s' % (len(conv), argv[2]))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
