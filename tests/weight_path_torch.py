"""Torch restatement of FDConv's small weight path (pooled means -> per-sample weights), differentiable by torch autograd.

TEST INFRASTRUCTURE: the product path computes these weights with sr_fdconv_weightgen and their gradients with
sr_fdconv_weightgen_backward (hand-written HIP); this file is the checker the GPU tests hold those kernels against
(it was the training path's backward glue until round 3).  Reference: FDConv.py:643-652, 661-697.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _group_onehot(m):
    """(kernel_num, bins) float one-hot of the frequency-group id of every rfft2 bin: ka[b, G(r, c)] = ka @ onehot, a
    matmul whose backward is a matmul (indexing with the group map goes back through an index_put)."""
    gmap = m.group_map()
    key = (gmap.data_ptr(), gmap.device)
    hit = m._sr_cache.get('onehot')
    if hit is None or hit[0] != key:
        oh = F.one_hot(gmap.view(-1).long(), m.kernel_num).t().to(torch.float32).contiguous()
        m._sr_cache['onehot'] = hit = (key, oh)
    return hit[1]


class _GroupLookup(torch.autograd.Function):
    """per_bin[b, j] = ka[b, G(j)] as a matmul with the one-hot map; its gradient wrt ka is a per-group sum, taken over the
    bins sorted by group (the matmul form, (B x bins) @ (bins x kernel_num), is a 55k-deep skinny GEMM: 63 us)."""

    @staticmethod
    def forward(ctx, ka, onehot, order):
        ctx.save_for_backward(order)
        ctx.kn = onehot.shape[0]
        return ka @ onehot

    @staticmethod
    def backward(ctx, g):
        (order,) = ctx.saved_tensors
        return g.index_select(1, order).view(g.shape[0], ctx.kn, -1).sum(2), None, None


def _group_order(m):
    """bin indices sorted by group id (every group holds the same number of bins: FDConv.py:604-631), cached."""
    gmap = m.group_map()
    key = (gmap.data_ptr(), gmap.device)
    hit = m._sr_cache.get('order')
    if hit is None or hit[0] != key:
        flat = gmap.view(-1).long()
        if not bool((torch.bincount(flat, minlength=m.kernel_num) == flat.numel() // m.kernel_num).all()):
            raise ValueError('FDConv: frequency groups of unequal size')
        m._sr_cache['order'] = hit = (key, torch.sort(flat, stable=True).indices)
    return hit[1]


def dynamic_weight(pooled, m, training):
    """W_b (B, 9, Cout, Cin) of FDConv `m` from the pooled means (B, Cin): FDConv.py:643-652, 661-697.

    Same result as `sr_fdconv_weightgen`, built from differentiable torch ops; used only to back-propagate the weight
    gradient into the KSM / spectrum parameters and the pooled means.  A training step spends its host time here (349
    layers, every op a launch and an autograd node), so the formulation keeps the op count low: fused batch norm, the
    four attention heads as one linear + one sigmoid, the group lookup as a matmul, and every constant factor (the four
    2 x sigmoid scalings, alpha and 2 / kernel_num) folded into the spectrum."""
    g, loc = m.KSM_Global, m.KSM_Local
    B, cin = pooled.shape
    cout, kn = m.out_channels, m.kernel_num
    z = F.linear(pooled, g.fc.weight.flatten(1))
    if training:
        z = F.batch_norm(z, None, None, g.bn.weight, g.bn.bias, True, 0.0, BN_EPS)
    else:
        z = F.batch_norm(z, g.bn.running_mean, g.bn.running_var, g.bn.weight, g.bn.bias, False, 0.0, BN_EPS)
    r = torch.relu(z)
    a = torch.addcmul(g.relu.bias, r * g.relu.scale, r)                 # StarReLU: s relu(z)^2 + b
    heads = (g.channel_fc, g.filter_fc, g.spatial_fc, g.kernel_fc)
    hw = torch.cat([h.weight.flatten(1) for h in heads], 0)
    hb = torch.cat([h.bias for h in heads], 0)
    ca, fa, sa, ka = torch.sigmoid(F.linear(a, hw, hb)).split([cin, cout, 9, kn], 1)   # without their factors 2, 2, 2, 2 / kn
    ln = F.layer_norm(pooled, (cin,), loc.norm.weight, loc.norm.bias, 1e-5)
    # KSM_Local's k-tap conv along the channel axis as a matmul over unfolded windows (a 1 -> Cout*9 channel conv1d of length
    # Cin lands on MIOpen's naive kernels in backward: 0.5 ms per call)
    k = loc.conv.kernel_size[0]
    lnu = F.pad(ln, ((k - 1) // 2, (k - 1) // 2)).unfold(1, k, 1)                        # (B, Cin, k)
    hr = torch.sigmoid(lnu @ loc.conv.weight.view(cout * 9, k).t())                      # (B, Cin, Cout*9), without its 2
    # spectrum scaled per frequency group and sample, inverse real 2-D DFT, (3Cout, 3Cin) -> [o, u, i, v]
    scale = 16.0 * float(m.alpha) * 2.0 / kn
    spec = torch.view_as_complex(m.dft_weight[0]) * scale                                # (3Cout, 3Cin/2+1)
    per_bin = _GroupLookup.apply(ka, _group_onehot(m), _group_order(m)).view(B, *spec.shape)  # ka[b, G(r, c)]
    adaptive = torch.fft.irfft2(spec * per_bin, s=(3 * cout, 3 * cin))                   # (B, 3Cout, 3Cin)
    adaptive = adaptive.view(B, cout, 3, cin, 3).permute(0, 2, 4, 1, 3).reshape(B, 9, cout, cin)
    mod = (sa[:, :, None] * fa[:, None, :])[:, :, :, None] * ca[:, None, None, :]        # (B, 9, Cout, Cin)
    return mod * hr.view(B, cin, cout, 9).permute(0, 3, 2, 1) * adaptive


_WEIGHT_PATH = ('dft_weight', 'KSM_Global.fc.weight', 'KSM_Global.bn.weight', 'KSM_Global.bn.bias', 'KSM_Global.relu.scale',
                'KSM_Global.relu.bias', 'KSM_Global.channel_fc.weight', 'KSM_Global.channel_fc.bias',
                'KSM_Global.filter_fc.weight', 'KSM_Global.filter_fc.bias', 'KSM_Global.spatial_fc.weight',
                'KSM_Global.spatial_fc.bias', 'KSM_Global.kernel_fc.weight', 'KSM_Global.kernel_fc.bias',
                'KSM_Local.conv.weight', 'KSM_Local.norm.weight', 'KSM_Local.norm.bias')
