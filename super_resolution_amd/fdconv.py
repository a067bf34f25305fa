"""FDConv (Frequency Dynamic Convolution) -- drop-in for the reference module, computed by libsr_amd.

Mirrors FDConv-main/FDConv-main/FDConv_detection/mmdet_custom/FDConv.py of the reference:
constructor signature (:520-553), sub-module tree and therefore `state_dict()` layout
(`dft_weight`, `KSM_Global.*`, `FBM.freq_weight_conv_list.*`, `KSM_Local.*`, `bias`; non-persistent
`indices` / `FBM.cached_masks`), the thin-layer fallback to a plain conv (:579-581, :641-642) and the
initialisation order (so a seeded construction draws the same parameters).  The sub-modules are
parameter holders only: `forward` packs the input to NHWC and calls the HIP kernels
(low-pass planes + pooled means, gates, per-sample weight generation, MFMA convolution).
Only the default-kwargs configuration used by the owner scripts
(train/train_realESRGAN_fdconv_1.py:85-91) is implemented; anything else raises NotImplementedError.
In train() mode, or whenever autograd is recording on a float32 input, `forward` takes the differentiable path of
super_resolution_amd/train.py (same forward kernels, BatchNorm on batch statistics in train(), HIP backward).
"""
import ctypes as C
import math

import logging

import torch
import torch.nn as nn

from . import _lib as L

_DEFAULT_FBM_CFG = {'k_list': [2, 4, 8], 'lowfreq_att': False, 'fs_feat': 'feat', 'act': 'sigmoid', 'spatial': 'conv',
                    'spatial_group': 1, 'spatial_kernel': 3, 'init': 'zero', 'global_selection': False}

_log = logging.getLogger(__name__)


def get_fft2freq(d1, d2, use_rfft=False):
    """Bins of a (d1, d2) [r]fft2 grid ordered by distance from DC (FDConv.py:481-516).

    Returns (coords (2, n) int64, freq grid).  torch's default unstable sort decides ties, exactly
    as in the reference, because `indices` is not stored in checkpoints.
    """
    fh = torch.fft.fftfreq(d1)
    fw = torch.fft.rfftfreq(d2) if use_rfft else torch.fft.fftfreq(d2)
    grid = torch.stack(torch.meshgrid(fh, fw, indexing='ij'), dim=-1)
    _, order = torch.sort(torch.norm(grid, dim=-1).view(-1))
    ncol = d2 // 2 + 1 if use_rfft else d2
    return torch.stack([order // ncol, order % ncol], dim=0), grid


class StarReLU(nn.Module):
    """s * relu(x)**2 + b (FDConv.py:19-36); holder of `scale` / `bias`."""

    def __init__(self):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(1))
        self.bias = nn.Parameter(torch.zeros(1))


class KernelSpatialModulation_Global(nn.Module):
    """Parameters of the global kernel/spatial attention (FDConv.py:38-258)."""

    def __init__(self, in_planes, out_planes, kernel_size, kernel_num, reduction=0.0625, min_channel=16, stride=1):
        super().__init__()
        att = max(int(in_planes * reduction), min_channel)
        self.fc = nn.Conv2d(in_planes, att, 1, bias=False)
        self.bn = nn.BatchNorm2d(att)
        self.relu = StarReLU()
        self.channel_fc = nn.Conv2d(att, in_planes, 1, bias=True)
        self.filter_fc = nn.Conv2d(att, out_planes, 1, stride=stride, bias=True)
        self.spatial_fc = nn.Conv2d(att, kernel_size * kernel_size, 1, bias=True)
        self.kernel_fc = nn.Conv2d(att, kernel_num, 1, bias=True)
        # FDConv.py:115-170: kaiming on every conv, then near-zero spatial/kernel/channel heads
        # (filter_fc keeps its kaiming init -- the reference's isinstance test at :141 never fires)
        for m in (self.fc, self.channel_fc, self.filter_fc, self.spatial_fc, self.kernel_fc):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.spatial_fc.weight, std=1e-6)
        nn.init.normal_(self.kernel_fc.weight, std=1e-6)
        nn.init.normal_(self.channel_fc.weight, std=1e-6)


class KernelSpatialModulation_Local(nn.Module):
    """Parameters of the local (per-weight) attention (FDConv.py:270-321)."""

    def __init__(self, channel, out_n):
        super().__init__()
        k = round((math.log2(channel) / 2) + 0.5) // 2 * 2 + 1
        self.conv = nn.Conv1d(1, out_n, kernel_size=k, padding=(k - 1) // 2, bias=False)
        nn.init.constant_(self.conv.weight, 1e-6)
        self.norm = nn.LayerNorm(channel)


class FrequencyBandModulation(nn.Module):
    """Parameters of the three band gates (FDConv.py:328-415)."""

    def __init__(self, in_channels, k_list=(2, 4, 8)):
        super().__init__()
        self.k_list = list(k_list)
        self.freq_weight_conv_list = nn.ModuleList()
        for _ in self.k_list:
            conv = nn.Conv2d(in_channels, 1, kernel_size=3, stride=1, padding=1, bias=True)
            nn.init.normal_(conv.weight, std=1e-6)
            conv.bias.data.zero_()
            self.freq_weight_conv_list.append(conv)
        _, freq = get_fft2freq(64, 64, use_rfft=True)
        radius = freq.abs().max(dim=-1)[0]
        masks = [radius < 0.5 / k + 1e-8 for k in self.k_list]
        self.register_buffer('cached_masks', torch.stack(masks, 0).unsqueeze(1), persistent=False)


class FDConv(nn.Conv2d):
    def __init__(self, *args, reduction=0.0625, kernel_num=4, use_fdconv_if_c_gt=16, use_fdconv_if_k_in=[1, 3],
                 use_fdconv_if_stride_in=[1], use_fbm_if_k_in=[3], use_fbm_for_stride=False, kernel_temp=1.0, temp=None,
                 att_multi=2.0, param_ratio=1, param_reduction=1.0, ksm_only_kernel_att=False, att_grid=1,
                 use_ksm_local=True, ksm_local_act='sigmoid', ksm_global_act='sigmoid', spatial_freq_decompose=False,
                 convert_param=True, linear_mode=False, fbm_cfg=None, **kwargs):
        super().__init__(*args, **kwargs)
        fbm_cfg = dict(_DEFAULT_FBM_CFG) if fbm_cfg is None else fbm_cfg
        given = dict(reduction=reduction, kernel_num=kernel_num, use_fdconv_if_c_gt=use_fdconv_if_c_gt,
                     use_fdconv_if_k_in=list(use_fdconv_if_k_in), use_fdconv_if_stride_in=list(use_fdconv_if_stride_in),
                     use_fbm_if_k_in=list(use_fbm_if_k_in), use_fbm_for_stride=use_fbm_for_stride, kernel_temp=kernel_temp,
                     temp=temp, att_multi=att_multi, param_ratio=param_ratio, param_reduction=param_reduction,
                     ksm_only_kernel_att=ksm_only_kernel_att, att_grid=att_grid, use_ksm_local=use_ksm_local,
                     ksm_local_act=ksm_local_act, ksm_global_act=ksm_global_act,
                     spatial_freq_decompose=spatial_freq_decompose, convert_param=convert_param, linear_mode=linear_mode,
                     fbm_cfg=fbm_cfg)
        default = dict(reduction=0.0625, kernel_num=4, use_fdconv_if_c_gt=16, use_fdconv_if_k_in=[1, 3],
                       use_fdconv_if_stride_in=[1], use_fbm_if_k_in=[3], use_fbm_for_stride=False, kernel_temp=1.0,
                       temp=None, att_multi=2.0, param_ratio=1, param_reduction=1.0, ksm_only_kernel_att=False,
                       att_grid=1, use_ksm_local=True, ksm_local_act='sigmoid', ksm_global_act='sigmoid',
                       spatial_freq_decompose=False, convert_param=True, linear_mode=False, fbm_cfg=_DEFAULT_FBM_CFG)
        odd = [k for k in given if given[k] != default[k]]
        if odd:
            raise NotImplementedError('FDConv on MI355X implements the default configuration only; got non-default %s' % odd)
        self.kernel_num = kernel_num
        self.param_ratio = param_ratio
        self.param_reduction = param_reduction
        self.att_multi = att_multi
        self.use_ksm_local = use_ksm_local
        self.use_fdconv_if_c_gt = use_fdconv_if_c_gt
        self.use_fdconv_if_k_in = list(use_fdconv_if_k_in)
        self._sr_cache = {}
        if not self.fdconv_active:
            return
        if self.kernel_size != (3, 3) or self.stride != (1, 1) or self.padding != (1, 1) or \
                self.dilation != (1, 1) or self.groups != 1:
            raise NotImplementedError('FDConv on MI355X implements 3x3, stride 1, padding 1, groups 1 (the RRDBNet case)')
        cin, cout = self.in_channels, self.out_channels
        if cin % 2 or (3 * cout * (3 * cin // 2 + 1)) % kernel_num:
            raise ValueError('FDConv needs an even in_channels and an rfft2 grid divisible by kernel_num')
        _log.debug('kernel_num: %d', self.kernel_num)  # the reference prints this line to stdout (FDConv.py:566)
        self.alpha = min(cout, cin) // 2 * self.kernel_num * self.param_ratio / param_reduction
        self.KSM_Global = KernelSpatialModulation_Global(cin, cout, 3, kernel_num, reduction=reduction, stride=self.stride)
        self.FBM = FrequencyBandModulation(cin, fbm_cfg['k_list'])
        self.KSM_Local = KernelSpatialModulation_Local(cin, cout * 9)
        if self.KSM_Global.fc.out_channels != 16:
            raise NotImplementedError('attention width %d (in_channels >= 272) is not implemented' % self.KSM_Global.fc.out_channels)
        self.convert2dftweight()

    @property
    def fdconv_active(self):
        """False for thin layers, which stay ordinary convolutions (FDConv.py:579-581, :641)."""
        return not (min(self.in_channels, self.out_channels) <= self.use_fdconv_if_c_gt
                    or self.kernel_size[0] not in self.use_fdconv_if_k_in)

    def convert2dftweight(self):
        """Spatial weight -> `dft_weight` spectrum parameter + frequency-group `indices` (FDConv.py:604-631)."""
        d1, d2 = self.out_channels * 3, self.in_channels * 3
        coords, _ = get_fft2freq(d1, d2, use_rfft=True)
        w2 = self.weight.permute(0, 2, 1, 3).reshape(d1, d2)
        spec = torch.fft.rfft2(w2, dim=(0, 1))
        spec = torch.stack([spec.real, spec.imag], dim=-1)[None] / (min(self.out_channels, self.in_channels) // 2)
        self.dft_weight = nn.Parameter(spec.detach().clone(), requires_grad=True)
        del self.weight
        self.register_buffer('indices', coords.reshape(2, self.kernel_num, -1)[None].clone(), persistent=False)

    # ------------------------------------------------------------------ libsr_amd plumbing
    def _apply(self, fn, *a, **k):
        self._sr_cache = {}
        return super()._apply(fn, *a, **k)

    def __setattr__(self, name, value):
        # a replaced parameter / submodule invalidates what is cached per module (parameter lists, packed weights)
        if isinstance(value, (nn.Parameter, nn.Module)) and '_sr_cache' in self.__dict__:
            self.__dict__['_sr_cache'] = {}
        super().__setattr__(name, value)

    def group_map(self):
        """uint8 (3*cout, 3*cin//2+1) frequency-group id of every rfft2 bin, from `indices`."""
        idx = self.indices[0]
        key = (idx.data_ptr(), idx._version, idx.device)
        hit = self._sr_cache.get('gmap')  # built once per placement: a training step asks for it twice per layer
        if hit is not None and hit[0] == key:
            return hit[1]
        gmap = torch.zeros(self.out_channels * 3, self.in_channels * 3 // 2 + 1, dtype=torch.uint8, device=idx.device)
        for g in range(idx.shape[1]):
            gmap[idx[0, g], idx[1, g]] = g
        self._sr_cache['gmap'] = (key, gmap)
        return gmap

    def sr_layer_params(self, dtype):
        """(LayerParams, keepalive) for the executor / op-level calls; derived buffers are cached."""
        dev = (self.dft_weight if self.fdconv_active else self.weight).device
        src = self.dft_weight if self.fdconv_active else self.weight
        key = (dtype, src.data_ptr(), src._version)
        hit = self._sr_cache.get('lp')
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        lib = L.lib()
        st = L.stream_ptr(dev)
        lp = L.LayerParams()
        lp.cin, lp.cout = self.in_channels, self.out_channels
        keep = []

        def f32(t):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise TypeError('libsr_amd reads float32 contiguous parameters')
            L.require_gpu(t, 'FDConv')
            keep.append(t)
            return t.data_ptr()

        lp.bias = f32(self.bias.data) if self.bias is not None else None
        if not self.fdconv_active:
            lp.active = 0
            f32(self.weight.data)
            wp, lp.wpacked_lo_off = L.pack_plain_weight(self.weight.data, dtype, self.out_channels, self.in_channels)
            keep.append(wp)
            lp.wpacked = wp.data_ptr()
        else:
            lp.active = 1
            g, loc, fbm = self.KSM_Global, self.KSM_Local, self.FBM
            lp.loc_taps = loc.conv.kernel_size[0]
            gmap = self.group_map()
            basis = torch.empty(4 * self.out_channels * self.in_channels * 9, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.sr_fdconv_basis_workspace_bytes(self.out_channels, self.in_channels), dtype=torch.uint8, device=dev)
            L.check(lib.sr_fdconv_basis(f32(self.dft_weight.data), gmap.data_ptr(), float(self.alpha), self.out_channels,
                                        self.in_channels, basis.data_ptr(), ws.data_ptr(), ws.numel(), st), 'sr_fdconv_basis')
            keep += [gmap, basis, ws]
            lp.basis = basis.data_ptr()
            lp.fc_w = f32(g.fc.weight.data)
            lp.bn_w, lp.bn_b = f32(g.bn.weight.data), f32(g.bn.bias.data)
            lp.bn_rm, lp.bn_rv = f32(g.bn.running_mean), f32(g.bn.running_var)
            lp.relu_s, lp.relu_b = f32(g.relu.scale.data), f32(g.relu.bias.data)
            lp.ch_w, lp.ch_b = f32(g.channel_fc.weight.data), f32(g.channel_fc.bias.data)
            lp.fl_w, lp.fl_b = f32(g.filter_fc.weight.data), f32(g.filter_fc.bias.data)
            lp.sp_w, lp.sp_b = f32(g.spatial_fc.weight.data), f32(g.spatial_fc.bias.data)
            lp.kn_w, lp.kn_b = f32(g.kernel_fc.weight.data), f32(g.kernel_fc.bias.data)
            for i in range(3):
                lp.gate_w[i] = f32(fbm.freq_weight_conv_list[i].weight.data)
                lp.gate_b[i] = f32(fbm.freq_weight_conv_list[i].bias.data)
            lp.loc_w = f32(loc.conv.weight.data)
            lp.ln_w, lp.ln_b = f32(loc.norm.weight.data), f32(loc.norm.bias.data)
        self._sr_cache['lp'] = (key, lp, keep)
        return lp, keep

    def _plan(self, H, W):
        plans = self._sr_cache.setdefault('plans', {})
        if (H, W) not in plans:
            h = C.c_void_p()
            L.check(L.lib().sr_fbm_plan_create(H, W, C.byref(h)), 'sr_fbm_plan_create')
            plans[(H, W)] = _PlanHandle(h)
        return plans[(H, W)].h

    def forward(self, x, want=None):
        """x: (B, Cin, H, W) float32 or bfloat16 CUDA tensor -> (B, Cout, H, W), same dtype.

        `want`: optional dict that receives intermediates as torch tensors (tests only).
        """
        L.require_gpu(x, 'FDConv.forward')
        if want is None and (self.training or (torch.is_grad_enabled() and x.dtype == torch.float32)):
            # differentiable path (float32): batch-statistics BatchNorm in train() mode, hand-written backward kernels
            from .train import conv_train
            return conv_train(self, x)
        lib = L.lib()
        dt = x.dtype
        code = L.dtype_code(dt)
        dev = x.device
        st = L.stream_ptr(dev)
        B, cin, H, W = x.shape
        cout = self.out_channels
        if cin != self.in_channels:
            raise RuntimeError('expected %d input channels, got %d' % (self.in_channels, cin))
        x = x.contiguous()
        lp, _keep = self.sr_layer_params(dt)
        xn = torch.empty(B, H, W, cin, dtype=dt, device=dev)
        L.check(lib.sr_pack_nchw(x.data_ptr(), code, xn.data_ptr(), code, B, cin, H, W, 1, st), 'sr_pack_nchw')
        yn = torch.empty(B, H, W, cout, dtype=dt, device=dev)
        cc = L.Concat()
        cc.nseg = 1
        cc.x[0] = L.view(xn)
        ep = L.Epilogue()
        ep.bias = lp.bias
        if not self.fdconv_active:
            L.check(lib.sr_conv3x3(code, B, H, W, C.byref(cc), None, lp.wpacked, 0, lp.wpacked_lo_off, cout, C.byref(ep), L.view(yn), st),
                    'sr_conv3x3')
        else:
            plan = self._plan(H, W)
            bands = [torch.empty_like(xn) for _ in range(3)]
            pooled = torch.empty(B, cin, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, cin), dtype=torch.uint8, device=dev)
            L.check(lib.sr_fbm_lowpass(plan, code, B, L.view(xn), 0, L.view(bands[0]), L.view(bands[1]), L.view(bands[2]),
                                       pooled.data_ptr(), cin, ws.data_ptr(), ws.numel(), st), 'sr_fbm_lowpass')
            gates = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
            L.check(lib.sr_fbm_gates(code, B, H, W, C.byref(cc), C.byref(lp), gates.data_ptr(), st), 'sr_fbm_gates')
            # bf16: weights as hi + lo pairs (a rounded weight is an error coherent over the whole image)
            split = dt == torch.bfloat16
            wdyn = torch.empty(2 if split else 1, B, 9, cout, cin, dtype=dt, device=dev)
            mods = torch.empty(lib.sr_fdconv_weightgen_workspace_bytes(B, cin, cout), dtype=torch.uint8, device=dev)
            L.check(lib.sr_fdconv_weightgen(code, B, pooled.data_ptr(), cin, C.byref(lp), 0, 0.0, wdyn[0].data_ptr(),
                                            wdyn[1].data_ptr() if split else None, mods.data_ptr(), mods.numel(), st),
                    'sr_fdconv_weightgen')
            cc.bands = 1
            cc.l2[0], cc.l4[0], cc.l8[0] = L.view(bands[0]), L.view(bands[1]), L.view(bands[2])
            L.check(lib.sr_conv3x3(code, B, H, W, C.byref(cc), gates.data_ptr(), wdyn.data_ptr(), 9 * cout * cin,
                                   B * 9 * cout * cin if split else 0, cout, C.byref(ep), L.view(yn), st), 'sr_conv3x3')
            if want is not None:
                want.update(pooled=pooled, gates=gates, wdyn=wdyn[0], l2=bands[0], l4=bands[1], l8=bands[2])
        y = torch.empty(B, cout, H, W, dtype=dt, device=dev)
        L.check(lib.sr_unpack_nhwc(yn.data_ptr(), code, y.data_ptr(), code, B, cout, H, W, st), 'sr_unpack_nhwc')
        return y


class _PlanHandle:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            L.lib().sr_fbm_plan_destroy(self.h)
        except Exception:
            pass


def replace_conv_with_fdconv(module):
    """Swap every nn.Conv2d child for a freshly initialised FDConv (train/train_realESRGAN_fdconv_1.py:82-95)."""
    for name, child in module.named_children():
        if isinstance(child, nn.Conv2d):
            setattr(module, name, FDConv(child.in_channels, child.out_channels, kernel_size=child.kernel_size,
                                         stride=child.stride, padding=child.padding, bias=(child.bias is not None)))
        else:
            replace_conv_with_fdconv(child)
    if hasattr(module, '_sr_invalidate'):
        module._sr_invalidate()
    return module
