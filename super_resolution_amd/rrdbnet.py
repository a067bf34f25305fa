"""RRDBNet (Real-ESRGAN x4 generator) -- drop-in for basicsr.archs.rrdbnet_arch.RRDBNet.

The reference imports the class from basicsr==1.4.2 (not vendored; call sites
train/train_realESRGAN_fdconv_1.py:100, train/test_realESRGAN_fdconv.py:19,
Real-ESRGAN-master/inference_realesrgan.py:60).  This module keeps its constructor, module tree and
`state_dict()` layout (`conv_first`, `body.{i}.rdb{1,2,3}.conv{1..5}`, `conv_body`, `conv_up1`,
`conv_up2`, `conv_hr`, `conv_last`) and its initialisation order; `forward` hands the whole network
to the C++ executor in libsr_amd (one call per forward, every kernel hand-written HIP).  Convs that
`replace_conv_with_fdconv` has turned into FDConv layers run the FDConv path.
"""
import ctypes as C

import torch
import torch.nn as nn
from torch.nn import init

from . import _lib as L
from .fdconv import FDConv


def default_init_weights(modules, scale=1.0):
    """basicsr arch_util.default_init_weights for convs: kaiming_normal_ * scale, zero bias."""
    for m in modules:
        init.kaiming_normal_(m.weight)
        m.weight.data *= scale
        if m.bias is not None:
            m.bias.data.fill_(0)


class ResidualDenseBlock(nn.Module):
    def __init__(self, num_feat=64, num_grow_ch=32):
        super().__init__()
        self.conv1 = nn.Conv2d(num_feat, num_grow_ch, 3, 1, 1)
        self.conv2 = nn.Conv2d(num_feat + num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv3 = nn.Conv2d(num_feat + 2 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv4 = nn.Conv2d(num_feat + 3 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv5 = nn.Conv2d(num_feat + 4 * num_grow_ch, num_feat, 3, 1, 1)
        default_init_weights([self.conv1, self.conv2, self.conv3, self.conv4, self.conv5], 0.1)


class RRDB(nn.Module):
    def __init__(self, num_feat, num_grow_ch=32):
        super().__init__()
        self.rdb1 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb2 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb3 = ResidualDenseBlock(num_feat, num_grow_ch)


class RRDBNet(nn.Module):
    """RRDBNet(num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32).

    forward(x): (B, num_in_ch, H, W) float32 or bfloat16 CUDA tensor -> (B, num_out_ch, H*scale... ) in
    the same dtype; the network computes in that dtype (fp32: exact-fp32 kernels; bf16: MFMA),
    or in `compute_dtype` if that attribute is set (used by RealESRGANer(half=True)).
    """

    def __init__(self, num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
        super().__init__()
        self.scale = scale
        self.num_in_ch, self.num_out_ch = num_in_ch, num_out_ch
        self.num_feat, self.num_block, self.num_grow_ch = num_feat, num_block, num_grow_ch
        if scale == 2:
            num_in_ch = num_in_ch * 4
        elif scale == 1:
            num_in_ch = num_in_ch * 16
        self.conv_first = nn.Conv2d(num_in_ch, num_feat, 3, 1, 1)
        self.body = nn.Sequential(*[RRDB(num_feat, num_grow_ch) for _ in range(num_block)])
        self.conv_body = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_hr = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_last = nn.Conv2d(num_feat, num_out_ch, 3, 1, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        self.compute_dtype = None
        self._sr = None

    # ---------------------------------------------------------------- executor state
    def conv_layers(self):
        """The conv modules in state_dict order (the order libsr_amd's executor numbers them)."""
        out = [self.conv_first]
        for blk in self.body:
            for rdb in (blk.rdb1, blk.rdb2, blk.rdb3):
                out += [rdb.conv1, rdb.conv2, rdb.conv3, rdb.conv4, rdb.conv5]
        return out + [self.conv_body, self.conv_up1, self.conv_up2, self.conv_hr, self.conv_last]

    def _sr_invalidate(self):
        self._sr = None
        self.__dict__['_sr_train_params'] = None

    def _apply(self, fn, *a, **k):
        self._sr_invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._sr = None
        return super().load_state_dict(*a, **k)

    def __getstate__(self):
        d = self.__dict__.copy()
        d['_sr'] = None
        return d

    def _signature(self, dtype, dev):
        sig = [dtype, str(dev)]
        for m in self.conv_layers():
            w = m.dft_weight if (isinstance(m, FDConv) and m.fdconv_active) else m.weight
            sig.append((id(m), w.data_ptr(), w._version))
        return tuple(sig)

    def _state(self, dtype, dev):
        sig = self._signature(dtype, dev)
        if self._sr is not None and self._sr.sig == sig:
            return self._sr
        self._sr = _NetState(self, dtype, dev, sig)
        return self._sr

    def forward(self, x):
        L.require_gpu(x, 'RRDBNet.forward')
        if self.training or (torch.is_grad_enabled() and x.dtype == torch.float32):
            # the executor below enqueues the whole network without an autograd graph and evaluates KSM_Global's
            # BatchNorm on its running statistics: train() mode, and any float32 forward that autograd is recording,
            # take the layer-by-layer differentiable path (train/train_realESRGAN_fdconv_1.py:108-118)
            from .train import rrdbnet_train_forward
            return rrdbnet_train_forward(self, x)
        io_dt = x.dtype
        dt = self.compute_dtype or io_dt
        L.dtype_code(io_dt)
        B, Cc, H, W = x.shape
        if Cc != self.num_in_ch:
            raise RuntimeError('expected %d input channels, got %d' % (self.num_in_ch, Cc))
        r = {4: 1, 2: 2, 1: 4}[self.scale]
        if H % r or W % r:
            raise RuntimeError('input size must be divisible by %d for scale %d' % (r, self.scale))
        st = self._state(dt, x.device)
        return st.forward(x.contiguous(), io_dt)

    def workspace_bytes(self, B, H, W, dtype=torch.float32):
        """Executor workspace an inference forward of (B, C, H, W) needs, in bytes (0: cannot be sized).  `dtype`: the
        input's dtype (compute_dtype overrides it, as in forward).  RealESRGANer sizes its automatic tile batch with it."""
        dt = self.compute_dtype or dtype
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('RRDBNet.workspace_bytes: the network must be on the GPU')
        st = self._state(dt, dev)
        lib = L.lib()
        with torch.cuda.device(dev):
            if (H, W) not in st.prepared:
                L.check(lib.sr_rrdbnet_prepare(st.h, H, W), 'sr_rrdbnet_prepare')
                st.prepared.add((H, W))
            return int(lib.sr_rrdbnet_workspace_bytes(st.h, L.dtype_code(dt), B, H, W))

    def workspace_held(self):
        """Bytes of workspace the executor holds already (re-used by forwards that need no more)."""
        return 0 if self._sr is None or self._sr.ws is None else int(self._sr.ws.numel())

    def profile(self, enable=True):
        """Per-kernel-group HIP-event timing of the next forward (bench.py)."""
        if self._sr is None:
            raise RuntimeError('run one forward first')
        L.check(L.lib().sr_rrdbnet_profile(self._sr.h, int(enable)), 'sr_rrdbnet_profile')

    def profile_read(self):
        ms = (C.c_float * 8)()
        n = (C.c_long * 8)()
        L.check(L.lib().sr_rrdbnet_profile_read(self._sr.h, ms, n), 'sr_rrdbnet_profile_read')
        names = ('conv', 'lowpass', 'gates', 'weightgen', 'layout', 'head', 'lowpass_hr')
        return {k: (ms[i], n[i]) for i, k in enumerate(names)}


class _NetState:
    """Executor handle + derived buffers for one (dtype, device, parameter version)."""

    def __init__(self, net, dtype, dev, sig):
        lib = L.lib()
        self.sig = sig
        self.dtype = dtype
        self.dev = dev
        self.h = C.c_void_p()
        L.check(lib.sr_rrdbnet_create(net.num_in_ch, net.num_out_ch, net.scale, net.num_feat, net.num_block,
                                      net.num_grow_ch, C.byref(self.h)), 'sr_rrdbnet_create')
        self.keep = []
        self.out_mult = 4  # the trunk always upsamples x4; scale 2 / 1 start from an unshuffled input
        self.r = {4: 1, 2: 2, 1: 4}[net.scale]
        self.num_out_ch = net.num_out_ch
        with torch.cuda.device(dev):
            for i, m in enumerate(net.conv_layers()):
                if isinstance(m, FDConv):
                    lp, keep = m.sr_layer_params(dtype)
                else:
                    lp, keep = _plain_params(m, dtype)
                self.keep.append((lp, keep))
                L.check(lib.sr_rrdbnet_set_layer(self.h, i, C.byref(lp)), 'sr_rrdbnet_set_layer(%d)' % i)
        self.prepared = set()
        self.ws = None

    def forward(self, x, io_dt):
        lib = L.lib()
        B, _, H, W = x.shape
        code, io_code = L.dtype_code(self.dtype), L.dtype_code(io_dt)
        with torch.cuda.device(self.dev):
            if (H, W) not in self.prepared:
                L.check(lib.sr_rrdbnet_prepare(self.h, H, W), 'sr_rrdbnet_prepare')
                self.prepared.add((H, W))
            need = lib.sr_rrdbnet_workspace_bytes(self.h, code, B, H, W)
            if need == 0:
                raise RuntimeError('libsr_amd: cannot size a workspace for input %s' % (tuple(x.shape),))
            if self.ws is None or self.ws.numel() < need:
                self.ws = None
                self.ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            y = torch.empty(B, self.num_out_ch, H // self.r * 4, W // self.r * 4, dtype=io_dt, device=self.dev)
            L.check(lib.sr_rrdbnet_forward(self.h, code, io_code, B, H, W, x.data_ptr(), y.data_ptr(), self.ws.data_ptr(),
                                           self.ws.numel(), L.stream_ptr(self.dev)), 'sr_rrdbnet_forward')
        return y

    def __del__(self):
        try:
            L.lib().sr_rrdbnet_destroy(self.h)
        except Exception:
            pass


def _plain_params(m, dtype):
    """LayerParams of an ordinary nn.Conv2d(.,.,3,1,1)."""
    if m.kernel_size != (3, 3) or m.stride != (1, 1) or m.padding != (1, 1) or m.groups != 1 or m.dilation != (1, 1):
        raise NotImplementedError('libsr_amd implements 3x3 / stride 1 / padding 1 convolutions')
    w = m.weight.data
    L.require_gpu(w, 'RRDBNet')
    if w.dtype != torch.float32:
        raise TypeError('parameters must stay float32 (set compute_dtype for bf16 compute)')
    lp = L.LayerParams()
    lp.cin, lp.cout, lp.active = m.in_channels, m.out_channels, 0
    wp, lp.wpacked_lo_off = L.pack_plain_weight(w, dtype, m.out_channels, m.in_channels)
    lp.wpacked = wp.data_ptr()
    keep = [wp, w]
    if m.bias is not None:
        lp.bias = m.bias.data.data_ptr()
        keep.append(m.bias.data)
    return lp, keep
