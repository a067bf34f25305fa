"""Deterministic (numpy MT19937) parameters and inputs shared by make_golden.py and the tests.

Nothing here depends on torch's RNG, so the same arrays are rebuilt on any box.  The
reference's own initialisation (std 1e-6 on most modulation weights) makes FDConv almost
static; every parameter is therefore drawn with an O(1) effect on the output so that the
dynamic path (KSM global/local, frequency groups, FBM gates) is exercised.
"""
import numpy as np

LAYER_SHAPES = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)]


def ksm_local_taps(cin):
    import math
    return round((math.log2(cin) / 2) + 0.5) // 2 * 2 + 1


def fdconv_params(cin, cout, seed, kn=4, att=16, wscale=1.0):
    """State of one active FDConv layer (the 27 entries of SURVEY 8b) as float32/int64 numpy arrays."""
    r = np.random.RandomState(seed)
    f = lambda *s: r.standard_normal(s).astype(np.float32)
    p = {}
    p['bias'] = 0.1 * f(cout)
    # spectrum of a kaiming-like spatial kernel, stored the way convert2dftweight stores it
    w = (f(cout, cin, 3, 3) * wscale / np.sqrt(cin * 9.0)).astype(np.float32)
    w2 = np.transpose(w, (0, 2, 1, 3)).reshape(cout * 3, cin * 3)
    spec = np.fft.rfft2(w2.astype(np.float64)) / (min(cin, cout) // 2)
    p['dft_weight'] = np.stack([spec.real, spec.imag], -1)[None].astype(np.float32)
    p['KSM_Global.fc.weight'] = (f(att, cin, 1, 1) * (2.0 / np.sqrt(cin))).astype(np.float32)
    p['KSM_Global.bn.weight'] = 1 + 0.3 * f(att)
    p['KSM_Global.bn.bias'] = 0.3 * f(att)
    p['KSM_Global.bn.running_mean'] = 0.3 * f(att)
    p['KSM_Global.bn.running_var'] = (0.5 + r.rand(att)).astype(np.float32)
    p['KSM_Global.bn.num_batches_tracked'] = np.array(7, dtype=np.int64)
    p['KSM_Global.relu.scale'] = (1 + 0.3 * f(1)).astype(np.float32)
    p['KSM_Global.relu.bias'] = 0.3 * f(1)
    for name, n in (('channel_fc', cin), ('filter_fc', cout), ('spatial_fc', 9), ('kernel_fc', kn)):
        p['KSM_Global.%s.weight' % name] = 0.5 * f(n, att, 1, 1)
        p['KSM_Global.%s.bias' % name] = 0.3 * f(n)
    for i in range(3):
        p['FBM.freq_weight_conv_list.%d.weight' % i] = (f(1, cin, 3, 3) * (2.0 / np.sqrt(cin * 9.0))).astype(np.float32)
        p['FBM.freq_weight_conv_list.%d.bias' % i] = 0.3 * f(1)
    p['KSM_Local.conv.weight'] = 0.5 * f(cout * 9, 1, ksm_local_taps(cin))
    p['KSM_Local.norm.weight'] = 1 + 0.3 * f(cin)
    p['KSM_Local.norm.bias'] = 0.3 * f(cin)
    return {k: (np.ascontiguousarray(v) if v.ndim else v) for k, v in p.items()}


def plain_conv_params(cin, cout, seed, gain=1.0):
    r = np.random.RandomState(seed)
    w = (r.standard_normal((cout, cin, 3, 3)) * gain / np.sqrt(cin * 9.0)).astype(np.float32)
    b = (0.1 * r.standard_normal(cout)).astype(np.float32)
    return {'weight': w, 'bias': b}


def signed_input(shape, seed):
    """Activations inside the network are signed; [0,1) images minus an offset, float32."""
    return (np.random.RandomState(seed).rand(*shape) - 0.35).astype(np.float32)


def image_input(shape, seed):
    return np.random.RandomState(seed).rand(*shape).astype(np.float32)


def rrdbnet_conv_shapes(num_in_ch, num_out_ch, num_feat, num_block, num_grow_ch):
    """(name, cin, cout) of every conv of RRDBNet in state_dict order."""
    out = [('conv_first', num_in_ch, num_feat)]
    for i in range(num_block):
        for j in (1, 2, 3):
            for k in range(1, 5):
                out.append(('body.%d.rdb%d.conv%d' % (i, j, k), num_feat + (k - 1) * num_grow_ch, num_grow_ch))
            out.append(('body.%d.rdb%d.conv5' % (i, j), num_feat + 4 * num_grow_ch, num_feat))
    for n in ('conv_body', 'conv_up1', 'conv_up2', 'conv_hr'):
        out.append((n, num_feat, num_feat))
    out.append(('conv_last', num_feat, num_out_ch))
    return out


def rrdbnet_state(num_in_ch, num_out_ch, num_feat, num_block, num_grow_ch, seed, fdconv=True, gain=1.0, wscale=0.25):
    """Flat numpy state_dict of RRDBNet; FDConv-active layers (min(c) > 16) get FDConv entries."""
    sd = {}
    for n, (name, cin, cout) in enumerate(rrdbnet_conv_shapes(num_in_ch, num_out_ch, num_feat, num_block, num_grow_ch)):
        if fdconv and min(cin, cout) > 16:
            lp = fdconv_params(cin, cout, seed * 1000 + n, wscale=wscale)
        else:
            lp = plain_conv_params(cin, cout, seed * 1000 + n, gain)
        for k, v in lp.items():
            sd[name + '.' + k] = v
    return sd


def srvgg_state(num_in_ch, num_out_ch, num_feat, num_conv, upscale, seed, act_type='prelu'):
    """Flat numpy state_dict of SRVGGNetCompact (Real-ESRGAN-master/realesrgan/archs/srvgg_arch.py:22-58):
    body.0 conv, body.1 act, then (conv, act) x num_conv, last conv -> num_out_ch * upscale^2."""
    r = np.random.RandomState(seed)
    sd = {}
    idx = 0
    chans = [(num_in_ch, num_feat)] + [(num_feat, num_feat)] * num_conv
    for cin, cout in chans:
        sd['body.%d.weight' % idx] = (r.standard_normal((cout, cin, 3, 3)) * np.sqrt(1.6 / (cin * 9.0))).astype(np.float32)
        sd['body.%d.bias' % idx] = (0.05 * r.standard_normal(cout)).astype(np.float32)
        idx += 1
        if act_type == 'prelu':
            sd['body.%d.weight' % idx] = (0.25 + 0.2 * r.standard_normal(cout)).astype(np.float32)
        idx += 1
    cout = num_out_ch * upscale * upscale
    sd['body.%d.weight' % idx] = (r.standard_normal((cout, num_feat, 3, 3)) * 0.5 / np.sqrt(num_feat * 9.0)).astype(np.float32)
    sd['body.%d.bias' % idx] = (0.05 * r.standard_normal(cout)).astype(np.float32)
    return sd
