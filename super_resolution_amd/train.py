"""Differentiable train-mode path of RRDBNet-FDConv on libsr_amd (training step of the owner scripts).

Reference: train/train_realESRGAN_fdconv_1.py:98-118 -- `model.train(); pred = model(wf); pred_up = F.interpolate(pred,
(2048, 2048), 'bilinear'); loss = L1(pred_up, sr); loss.backward(); optimizer.step()` -- where the backward pass comes from
torch autograd through FDConv.forward (FDConv.py:640-722) and basicsr's RRDBNet.

Here every conv layer is a `torch.autograd.Function` whose forward AND backward are hand-written HIP kernels:

  forward   low-pass planes + pooled means, gates, per-sample weights (KSM_Global's BatchNorm on the statistics of the
            batch, running statistics updated in place -- `sr_fdconv_weightgen(train=1)`), convolution;
  backward  weight gradient (`sr_conv3x3_wgrad`, per-sample for FDConv), operand gradient (`sr_conv3x3` with the transposed
            weights), band-mix / gate backward (`sr_fbm_backward`, `sr_fbm_gates_backward`), and the three band branches
            back through `sr_fbm_lowpass` (the box low-pass is self-adjoint).

Precision: float32 inputs run every kernel in float32 (the parity path: gradients match the reference's to 1e-4);
bfloat16 inputs -- or `net.compute_dtype = torch.bfloat16` with float32 data -- run mixed precision: activations and
their gradients in bf16 on the matrix cores (MFMA convolution forward and operand gradient with split hi + lo weights,
MFMA low-pass), parameters, weight gradients, gates and every reduction in float32.

The weight path -- pooled (B, Cin) -> W_b (B, 9, Cout, Cin): KSM_Global / KSM_Local / spectral scatter + irfft2 -- has a
hand-written backward too (`sr_fdconv_weightgen_backward`: no torch.fft, no F.batch_norm; its torch restatement lives in
tests/weight_path_torch.py as a checker only).
Data-parallel training: `allreduce_gradients` (bucketed all-reduce of the parameter gradients over torch.distributed --
RCCL on the GPUs; BatchNorm statistics stay local to a rank, as under the reference's nn.DataParallel,
train/train_realESRGAN_fdconv_OST_1.py:238-246).
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .fdconv import FDConv

BN_EPS = 1e-5


# struct field of sr_weight_path_grads -> parameter name inside an FDConv module (same order as L.WEIGHT_PATH_GRADS)
_WP_FIELDS = (('dft_weight', 'dft_weight'), ('fc_w', 'KSM_Global.fc.weight'), ('bn_w', 'KSM_Global.bn.weight'), ('bn_b', 'KSM_Global.bn.bias'),
              ('relu_s', 'KSM_Global.relu.scale'), ('relu_b', 'KSM_Global.relu.bias'), ('ch_w', 'KSM_Global.channel_fc.weight'),
              ('ch_b', 'KSM_Global.channel_fc.bias'), ('fl_w', 'KSM_Global.filter_fc.weight'), ('fl_b', 'KSM_Global.filter_fc.bias'),
              ('sp_w', 'KSM_Global.spatial_fc.weight'), ('sp_b', 'KSM_Global.spatial_fc.bias'), ('kn_w', 'KSM_Global.kernel_fc.weight'),
              ('kn_b', 'KSM_Global.kernel_fc.bias'), ('loc_w', 'KSM_Local.conv.weight'), ('ln_w', 'KSM_Local.norm.weight'),
              ('ln_b', 'KSM_Local.norm.bias'))
assert tuple(f for f, _ in _WP_FIELDS) == L.WEIGHT_PATH_GRADS


def weight_path_backward(m, lp, pooled, pooled_ld, mods, dw, training, dpooled=None, dpooled_ld=None, accumulate=False):
    """dW_b (B, 9, Cout, Cin) float32 -> ({parameter name: gradient}, dpooled (B, Cin)) for the weight path of FDConv `m`
    (KSM_Global, KSM_Local, spectrum), one call into libsr_amd (`sr_fdconv_weightgen_backward`: hand-written kernels, no
    torch operator).  `mods` is the workspace the forward's sr_fdconv_weightgen left behind; `pooled` may be a wider
    buffer (pooled_ld floats per sample).  dpooled: optional destination (tensor or device address, B rows of dpooled_ld floats;
    accumulate: added to instead of overwritten), else allocated."""
    lib, dev = L.lib(), dw.device
    B = dw.shape[0]
    cin, cout = m.in_channels, m.out_channels
    hit = m._sr_cache.get('wp_layout')
    if hit is None:
        params = _named_params(m)[2]
        shapes = [tuple(params[n].shape) for _, n in _WP_FIELDS]
        sizes = [max(1, int(torch.Size(sh).numel())) for sh in shapes]
        m._sr_cache['wp_layout'] = hit = (shapes, sizes, sum(sizes))
    shapes, sizes, total = hit
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    pieces = flat.split(sizes)
    gs = L.WeightPathGrads()
    out = {}
    for (field, name), piece, shape in zip(_WP_FIELDS, pieces, shapes):
        setattr(gs, field, piece.data_ptr())
        out[name] = piece.view(shape)
    if dpooled is None:
        dpooled = torch.empty(B, cin, dtype=torch.float32, device=dev)
        dpooled_ld = cin
    ws = torch.empty(lib.sr_fdconv_weightgen_backward_workspace_bytes(B, cin, cout), dtype=torch.uint8, device=dev)
    L.check(lib.sr_fdconv_weightgen_backward(B, pooled.data_ptr(), pooled_ld, C.byref(lp), int(training), mods.data_ptr(),
                                             m.dft_weight.data_ptr(), m.group_map().data_ptr(), float(m.alpha), dw.data_ptr(),
                                             C.byref(gs), dpooled.data_ptr() if torch.is_tensor(dpooled) else dpooled, dpooled_ld,
                                             int(accumulate), ws.data_ptr(), ws.numel(), L.stream_ptr(dev)), 'sr_fdconv_weightgen_backward')
    return out, dpooled


# ------------------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------------------
def _pack(x):
    """(B, C, H, W) tensor -> contiguous (B, H, W, C) tensor the kernels read.  The layers hand each other NCHW-shaped
    tensors in torch's channels_last memory format -- which IS NHWC memory -- so between two layers of the network this is
    a view; only tensors that arrive in the default format (the network input, a loss gradient) are converted, by
    sr_pack_nchw.  (Converting at every layer boundary took 119 ms of the 2.1 s training step.)"""
    B, Cc, H, W = x.shape
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    xn = torch.empty(B, H, W, Cc, dtype=x.dtype, device=x.device)
    code = L.dtype_code(x.dtype)
    L.check(L.lib().sr_pack_nchw(x.contiguous().data_ptr(), code, xn.data_ptr(), code, B, Cc, H, W, 1, L.stream_ptr(x.device)), 'sr_pack_nchw')
    return xn


def _unpack(xn):
    """(B, H, W, C) kernel output -> the same memory as an NCHW-shaped channels_last tensor (no copy)."""
    return xn.permute(0, 3, 1, 2)


def _concat1(xn, bands=None):
    cc = L.Concat()
    cc.nseg = 1
    cc.x[0] = L.view(xn)
    if bands is not None:
        cc.bands = 1
        cc.l2[0], cc.l4[0], cc.l8[0] = L.view(bands[0]), L.view(bands[1]), L.view(bands[2])
    return cc


def _conv(xn, cc, gates, w, w_bstride, cout, bias=None, w_lo=0):
    """w_lo: elements from `w` to the low-order parts of the same weights (bf16 split weights), 0: none."""
    B, H, W, _ = xn.shape
    yn = torch.empty(B, H, W, cout, dtype=xn.dtype, device=xn.device)
    ep = L.Epilogue()
    ep.bias = bias
    L.check(L.lib().sr_conv3x3(L.dtype_code(xn.dtype), B, H, W, C.byref(cc), L.ptr(gates), w.data_ptr(), w_bstride, w_lo, cout,
                               C.byref(ep), L.view(yn), L.stream_ptr(xn.device)), 'sr_conv3x3')
    return yn


def _check_input(x, who):
    """float32: the parity path (every kernel in fp32).  bfloat16: mixed precision -- activations, their gradients and the
    conv operands in bf16 on the matrix cores (split hi + lo weights), parameters, weight gradients, gates, pooled means
    and every reduction in float32."""
    L.require_gpu(x, who)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('%s: the training path computes in float32 or bfloat16 (got %s)' % (who, x.dtype))


def _transposed(w, nb, cout, cin):
    """[parts][nb][9][cout][cin] -> [parts][nb][9][cin][cout] with mirrored taps, every part (hi, lo) separately."""
    parts = w.shape[0]
    wt = torch.empty(parts, nb, 9, cin, cout, dtype=w.dtype, device=w.device)
    for k in range(parts):
        L.check(L.lib().sr_conv_weight_transpose(w[k].data_ptr(), wt[k].data_ptr(), L.dtype_code(w.dtype), nb, cout, cin,
                                                 L.stream_ptr(w.device)), 'sr_conv_weight_transpose')
    return wt


# ------------------------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------------------------
class _PlainConvFn(torch.autograd.Function):
    """nn.Conv2d(.,.,3,1,1): forward sr_conv3x3, backward sr_conv3x3_wgrad + sr_conv3x3 on the transposed weights."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _check_input(x, 'conv (train)')
        lib, dev = L.lib(), x.device
        cout, cin = weight.shape[:2]
        with torch.cuda.device(dev):
            xn = _pack(x)
            # bf16: hi + lo pairs for every plain layer of the training path (the weights are what is being learned)
            n = 9 * cout * cin
            split = x.dtype == torch.bfloat16
            wp = torch.empty(2 if split else 1, 1, 9, cout, cin, dtype=x.dtype, device=dev)
            L.check(lib.sr_pack_conv_weight(weight.detach().contiguous().data_ptr(), wp[0].data_ptr(), wp[1].data_ptr() if split else None,
                                            L.dtype_code(x.dtype), cout, cin, L.stream_ptr(dev)), 'sr_pack_conv_weight')
            yn = _conv(xn, _concat1(xn), None, wp, 0, cout, None if bias is None else bias.detach().data_ptr(), n if split else 0)
            y = _unpack(yn)
        ctx.save_for_backward(xn, wp)
        ctx.has_bias = bias is not None
        ctx.need_dx = ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        xn, wp = ctx.saved_tensors
        lib, dev = L.lib(), dy.device
        B, H, W, cin = xn.shape
        cout = dy.shape[1]
        with torch.cuda.device(dev):
            st = L.stream_ptr(dev)
            dyn = _pack(dy.to(xn.dtype))
            dw = torch.empty(9, cout, cin, dtype=torch.float32, device=dev)
            db = torch.empty(cout, dtype=torch.float32, device=dev) if ctx.has_bias else None
            cc = _concat1(xn)
            L.check(lib.sr_conv3x3_wgrad(L.dtype_code(xn.dtype), B, H, W, C.byref(cc), None, L.view(dyn), cout, 0, dw.data_ptr(), L.ptr(db),
                                         st), 'sr_conv3x3_wgrad')
            dx = None
            if ctx.need_dx:
                wt = _transposed(wp, 1, cout, cin)
                dx = _unpack(_conv(dyn, _concat1(dyn), None, wt, 0, cin, None, 9 * cin * cout if wt.shape[0] == 2 else 0))
        return dx, dw.permute(1, 2, 0).reshape(cout, cin, 3, 3), db


class _FDConvFn(torch.autograd.Function):
    """An active FDConv layer.  Arguments after the module are its parameters in `named_parameters()` order (so that
    autograd routes their gradients); the module supplies the derived buffers (basis, group map, FFT plan)."""

    @staticmethod
    def forward(ctx, x, m, training, *params):
        _check_input(x, 'FDConv (train)')
        lib, dev = L.lib(), x.device
        B, cin, H, W = x.shape
        cout = m.out_channels
        if training and B < 2:
            raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                             % (torch.Size([B, m.KSM_Global.fc.out_channels, 1, 1]),))  # nn.BatchNorm2d's own error
        with torch.cuda.device(dev):
            st = L.stream_ptr(dev)
            dt, code = x.dtype, L.dtype_code(x.dtype)
            split = dt == torch.bfloat16
            lp, _keep = m.sr_layer_params(dt)
            plan = m._plan(H, W)
            xn = _pack(x)
            bands = [torch.empty_like(xn) for _ in range(3)]
            pooled = torch.empty(B, cin, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, cin), dtype=torch.uint8, device=dev)
            L.check(lib.sr_fbm_lowpass(plan, code, B, L.view(xn), 0, L.view(bands[0]), L.view(bands[1]), L.view(bands[2]),
                                       pooled.data_ptr(), cin, ws.data_ptr(), ws.numel(), st), 'sr_fbm_lowpass')
            gates = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
            cc = _concat1(xn)
            L.check(lib.sr_fbm_gates(code, B, H, W, C.byref(cc), C.byref(lp), gates.data_ptr(), st), 'sr_fbm_gates')
            wdyn = torch.empty(2 if split else 1, B, 9, cout, cin, dtype=dt, device=dev)
            mods = torch.empty(lib.sr_fdconv_weightgen_workspace_bytes(B, cin, cout), dtype=torch.uint8, device=dev)
            bn = m.KSM_Global.bn
            mom = 0.0
            if training and bn.track_running_stats:
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            L.check(lib.sr_fdconv_weightgen(code, B, pooled.data_ptr(), cin, C.byref(lp), int(training), float(mom),
                                            wdyn[0].data_ptr(), wdyn[1].data_ptr() if split else None, mods.data_ptr(), mods.numel(),
                                            st), 'sr_fdconv_weightgen')
            yn = _conv(xn, _concat1(xn, bands), gates, wdyn, 9 * cout * cin, cout, lp.bias, B * 9 * cout * cin if split else 0)
            y = _unpack(yn)
        ctx.m, ctx.training = m, training
        ctx.names = _named_params(m)[0]
        ctx.save_for_backward(xn, bands[0], bands[1], bands[2], gates, pooled, wdyn, mods)
        return y

    @staticmethod
    def backward(ctx, dy):
        xn, b2, b4, b8, gates, pooled, wdyn, mods = ctx.saved_tensors
        m = ctx.m
        lib, dev = L.lib(), dy.device
        B, H, W, cin = xn.shape
        cout = m.out_channels
        grads = {}
        with torch.cuda.device(dev):
            st = L.stream_ptr(dev)
            dt, code = xn.dtype, L.dtype_code(xn.dtype)
            lp, _keep = m.sr_layer_params(dt)
            plan = m._plan(H, W)
            dyn = _pack(dy.to(dt))
            # 1. weight / bias gradients of the convolution
            dw = torch.empty(B, 9, cout, cin, dtype=torch.float32, device=dev)
            db = torch.empty(cout, dtype=torch.float32, device=dev) if m.bias is not None else None
            cc = _concat1(xn, (b2, b4, b8))
            L.check(lib.sr_conv3x3_wgrad(code, B, H, W, C.byref(cc), gates.data_ptr(), L.view(dyn), cout, 1, dw.data_ptr(),
                                         L.ptr(db), st), 'sr_conv3x3_wgrad')
            if db is not None:
                grads['bias'] = db
            # 2. gradient wrt the modulated operand: convolution of dy with the transposed, mirrored weights
            wt = _transposed(wdyn, B, cout, cin)
            da = _conv(dyn, _concat1(dyn), None, wt, 9 * cin * cout, cin, None, B * 9 * cin * cout if wt.shape[0] == 2 else 0)
            # 3. band mix: four linear branches + gate logits
            dxd, dl2, dl4, dl8 = (torch.empty_like(xn) for _ in range(4))
            dz = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
            L.check(lib.sr_fbm_backward(code, B, H, W, L.view(xn), L.view(b2), L.view(b4), L.view(b8), gates.data_ptr(), L.view(da),
                                        L.view(dxd), L.view(dl2), L.view(dl4), L.view(dl8), dz.data_ptr(), 0, None, st), 'sr_fbm_backward')
            # 4. the band branches back through the (self-adjoint) low-pass: plane k of lowpass(dl_k)
            ws = torch.empty(lib.sr_fbm_workspace_bytes(plan, B, cin), dtype=torch.uint8, device=dev)
            # (a view with a NULL pointer = band not wanted: each call synthesises one band only)
            adj = []
            for k, src in enumerate((dl2, dl4, dl8)):
                out = torch.empty_like(xn)
                planes = [L.View(), L.View(), L.View()]
                planes[k] = L.view(out)
                L.check(lib.sr_fbm_lowpass(plan, code, B, L.view(src), 0, planes[0], planes[1], planes[2],
                                           None, 0, ws.data_ptr(), ws.numel(), st), 'sr_fbm_lowpass (adjoint)')
                adj.append(out)
            # 5. the small weight path: dW_b -> parameters and pooled means (hand-written backward kernels)
            wgrads, dpooled = weight_path_backward(m, lp, pooled, cin, mods, dw, ctx.training)
            grads.update(wgrads)
            # 6. gate convs + assembly of the input gradient
            dxn = torch.empty_like(xn)
            if xn.dtype == torch.bfloat16 and cin % 32 == 0:
                # gate-conv weight gradients on the matrix cores: the weight-gradient kernel with the logit gradients (three
                # channels, padded to one 32-wide tile, bf16 like every other dy of this mode) in the place of dy
                dzp = torch.zeros(B, H, W, 32, dtype=torch.bfloat16, device=dev)
                dzp[..., :4] = dz
                dw32 = torch.empty(1, 9, 32, cin, dtype=torch.float32, device=dev)
                L.check(lib.sr_conv3x3_wgrad(code, B, H, W, C.byref(_concat1(xn)), None, L.view(dzp), 32, 0, dw32.data_ptr(), None, st),
                        'sr_conv3x3_wgrad (gate convs)')
                dgw = dw32[0, :, :3, :].permute(1, 2, 0).contiguous()
                dgb = dz.sum((0, 1, 2))[:3]
                gw_ptr = gb_ptr = None
            else:
                dgw = torch.empty(3, cin, 9, dtype=torch.float32, device=dev)
                dgb = torch.empty(3, dtype=torch.float32, device=dev)
                gw_ptr, gb_ptr = dgw.data_ptr(), dgb.data_ptr()
            L.check(lib.sr_fbm_gates_backward(code, B, H, W, L.view(xn), dz.data_ptr(), C.byref(lp), L.view(dxd), L.view(adj[0]),
                                              L.view(adj[1]), L.view(adj[2]), dpooled.data_ptr(), cin, L.view(dxn), gw_ptr,
                                              gb_ptr, st), 'sr_fbm_gates_backward')
            for k in range(3):
                grads['FBM.freq_weight_conv_list.%d.weight' % k] = dgw[k].reshape(1, cin, 3, 3)
                grads['FBM.freq_weight_conv_list.%d.bias' % k] = dgb[k:k + 1]
            dx = _unpack(dxn) if ctx.needs_input_grad[0] else None
        return (dx, None, None) + tuple(grads.get(n) for n in ctx.names)


def _named_params(m):
    """(names, parameters) of an FDConv layer, cached on the module: walking named_parameters() three times per layer and
    step was 30 ms of host time per training step."""
    hit = m._sr_cache.get('named')  # (the cache is dropped by Module._apply, i.e. by .to() / .cuda() / .float())
    if hit is None:
        pairs = list(m.named_parameters())
        m._sr_cache['named'] = hit = ([n for n, _ in pairs], [p for _, p in pairs], {n: p for n, p in pairs})
    return hit


def conv_train(m, x):
    """One conv layer of the network on the differentiable path: FDConv (active or thin) or nn.Conv2d, 3x3 / 1 / 1."""
    if isinstance(m, FDConv) and m.fdconv_active:
        return _FDConvFn.apply(x, m, m.training, *_named_params(m)[1])
    if m.kernel_size != (3, 3) or m.stride != (1, 1) or m.padding != (1, 1) or m.groups != 1 or m.dilation != (1, 1):
        raise NotImplementedError('libsr_amd implements 3x3 / stride 1 / padding 1 convolutions')
    return _PlainConvFn.apply(x, m.weight, m.bias)


# ------------------------------------------------------------------------------------------------------------------
# the whole network as ONE autograd node: virtual concat, per-produced-tensor low-pass, hand-written backward schedule
# ------------------------------------------------------------------------------------------------------------------
class _Group:
    """The tensors that feed one set of conv layers together, as ONE NHWC buffer: a dense block's x, x1..x4 are channel
    slices of `x` (the torch.cat of basicsr's ResidualDenseBlock never exists: layer k reads the channel prefix, writes
    its output slice), with the three low-pass planes and the pooled means of every slice beside it -- computed once per
    PRODUCED tensor (192 channels per dense block, not 640) and shared by the layers that consume it.  Backward keeps one
    accumulator per buffer (dx, dl2, dl4, dl8): every consumer adds its branch gradients in place, and the adjoint
    low-pass runs once per slice on the sums."""

    def __init__(self, B, H, W, Cc, dt, dev, bands, x=None):
        self.B, self.H, self.W, self.C, self.dt, self.dev = B, H, W, Cc, dt, dev
        self.x = x if x is not None else torch.empty(B, H, W, Cc, dtype=dt, device=dev)
        self.has_bands = bands
        if bands:
            self.l2, self.l4, self.l8 = (torch.empty_like(self.x) for _ in range(3))
            self.pooled = torch.empty(B, Cc, dtype=torch.float32, device=dev)
        self.dx = self.dl2 = self.dl4 = self.dl8 = self.dpool = None
        self.use_dpool = False

    def open_backward(self, covers_all):
        """Called by every consumer in backward; True for the one that must WRITE the accumulators instead of adding to
        them: the first consumer, provided it covers every channel and every accumulator (in RRDBNet: conv5 of a dense
        block, or the only consumer).  Otherwise the accumulators start from zeros."""
        if self.dx is not None:
            return False
        make = torch.empty_like if covers_all else torch.zeros_like
        self.dx = make(self.x)
        if self.has_bands:
            self.dl2, self.dl4, self.dl8 = (make(self.x) for _ in range(3))
            self.dpool = torch.zeros_like(self.pooled)   # summed pooled-mean gradients of the consumers (matrix-core gate path)
        return covers_all


class _Runner:
    """Forward / backward of RRDBNet on libsr_amd for one call (holds the activations between the two passes)."""

    def __init__(self, net, x, training):
        self.net, self.training = net, training
        self.lib, self.dev = L.lib(), x.device
        self.io_dt = x.dtype
        self.dt = net.compute_dtype if net.compute_dtype is not None else x.dtype
        if self.dt not in (torch.float32, torch.bfloat16):
            raise TypeError('RRDBNet (train): the training path computes in float32 or bfloat16 (got %s)' % self.dt)
        self.code = L.dtype_code(self.dt)
        self.st = L.stream_ptr(self.dev)
        self.plans = {}
        self.ws = None
        self.bn_counters = []

    # ---- small pieces ------------------------------------------------------------------------------------------
    def plan(self, m, H, W):
        return m._plan(H, W)

    def workspace(self, nbytes):
        if self.ws is None or self.ws.numel() < nbytes:
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        return self.ws

    def lowpass(self, plan, grp, c0, Cc, src=None, only=None, outs=None):
        """Forward use: bands + pooled means of channels [c0, c0 + Cc) of grp.x.  Adjoint use (src / only / outs given):
        plane `only` of the low-pass of `src` into outs[only]."""
        lib, B = self.lib, grp.B
        ws = self.workspace(lib.sr_fbm_workspace_bytes(plan, B, Cc))
        if src is None:
            L.check(lib.sr_fbm_lowpass(plan, self.code, B, L.view(grp.x, Cc, c0), 0, L.view(grp.l2, Cc, c0), L.view(grp.l4, Cc, c0),
                                       L.view(grp.l8, Cc, c0), grp.pooled.data_ptr() + 4 * c0, grp.C, ws.data_ptr(), ws.numel(), self.st),
                    'sr_fbm_lowpass')
        else:
            planes = [L.View(), L.View(), L.View()]
            planes[only] = outs
            L.check(lib.sr_fbm_lowpass(plan, self.code, B, src, 0, planes[0], planes[1], planes[2], None, 0, ws.data_ptr(), ws.numel(),
                                       self.st), 'sr_fbm_lowpass (adjoint)')

    def combine(self, B, H, W, Cc, terms, scales, act=None, slope=1.0, out=None, bc=None, bc_ld=0, bc_scale=0.0):
        """out = mask(act) * (sum_j scales[j] * terms[j] + bc_scale * bc[b][c]); terms / act / out are sr_views, bc a device
        address of (B, bc_ld) floats or None; returns the output tensor."""
        res = None
        if out is None:
            res = torch.empty(B, H, W, Cc, dtype=self.dt, device=self.dev)
            out = L.view(res)
        tv = (L.View * len(terms))(*terms)
        sv = (C.c_float * len(terms))(*scales)
        L.check(self.lib.sr_grad_combine(self.code, B, H, W, Cc, len(terms), tv, sv, act if act is not None else L.View(), float(slope),
                                         bc, bc_ld, float(bc_scale), out, self.st), 'sr_grad_combine')
        return res

    def resample(self, src, adjoint):
        B, H, W, Cc = src.shape
        Hs, Ws = (H // 2, W // 2) if adjoint else (H, W)
        dst = torch.empty(B, Hs if adjoint else 2 * H, Ws if adjoint else 2 * W, Cc, dtype=src.dtype, device=src.device)
        L.check(self.lib.sr_upsample2x(self.code, B, Hs, Ws, Cc, L.view(src), L.view(dst), int(adjoint), self.st), 'sr_upsample2x')
        return dst

    @staticmethod
    def active(m):
        return isinstance(m, FDConv) and m.fdconv_active

    # ---- one conv layer ----------------------------------------------------------------------------------------
    def layer_forward(self, m, grp, cin, out_view, lrelu=False, r1=None, s1=0.0, r2=None, s2=0.0):
        """Layer `m` on the channel prefix [0, cin) of grp.x -> out_view (an sr_view of cout channels).  Returns the record
        backward needs."""
        if m.kernel_size != (3, 3) or m.stride != (1, 1) or m.padding != (1, 1) or m.groups != 1 or m.dilation != (1, 1):
            raise NotImplementedError('libsr_amd implements 3x3 / stride 1 / padding 1 convolutions')
        lib, dt, dev, st = self.lib, self.dt, self.dev, self.st
        B, H, W = grp.B, grp.H, grp.W
        cout = m.out_channels
        if cin != m.in_channels:
            raise RuntimeError('expected %d input channels, got %d' % (m.in_channels, cin))
        split = dt == torch.bfloat16
        ep = L.Epilogue()
        ep.lrelu = int(lrelu)
        if r1 is not None:
            ep.r1, ep.s1 = r1, s1
        if r2 is not None:
            ep.r2, ep.s2 = r2, s2
        cc = L.Concat()
        cc.nseg = 1
        cc.x[0] = L.view(grp.x, cin, 0)
        rec = dict(m=m, grp=grp, cin=cin, active=self.active(m))
        if not rec['active']:
            n = 9 * cout * cin
            wp = torch.empty(2 if split else 1, 1, 9, cout, cin, dtype=dt, device=dev)
            L.check(lib.sr_pack_conv_weight(m.weight.detach().contiguous().data_ptr(), wp[0].data_ptr(), wp[1].data_ptr() if split else None,
                                            self.code, cout, cin, st), 'sr_pack_conv_weight')
            ep.bias = None if m.bias is None else m.bias.detach().data_ptr()
            L.check(lib.sr_conv3x3(self.code, B, H, W, C.byref(cc), None, wp.data_ptr(), 0, n if split else 0, cout, C.byref(ep), out_view,
                                   st), 'sr_conv3x3')
            rec.update(w=wp)
            return rec
        if self.training and B < 2:
            raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                             % (torch.Size([B, m.KSM_Global.fc.out_channels, 1, 1]),))  # nn.BatchNorm2d's own error
        lp, _keep = m.sr_layer_params(dt)
        gates = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
        L.check(lib.sr_fbm_gates(self.code, B, H, W, C.byref(cc), C.byref(lp), gates.data_ptr(), st), 'sr_fbm_gates')
        wdyn = torch.empty(2 if split else 1, B, 9, cout, cin, dtype=dt, device=dev)
        mods = torch.empty(lib.sr_fdconv_weightgen_workspace_bytes(B, cin, cout), dtype=torch.uint8, device=dev)
        bn = m.KSM_Global.bn
        mom = 0.0
        if self.training and bn.track_running_stats:
            self.bn_counters.append(bn.num_batches_tracked)   # all counters of the network: one fused increment per forward
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
        L.check(lib.sr_fdconv_weightgen(self.code, B, grp.pooled.data_ptr(), grp.C, C.byref(lp), int(self.training), float(mom),
                                        wdyn[0].data_ptr(), wdyn[1].data_ptr() if split else None, mods.data_ptr(), mods.numel(), st),
                'sr_fdconv_weightgen')
        cc.bands = 1
        cc.l2[0], cc.l4[0], cc.l8[0] = L.view(grp.l2, cin, 0), L.view(grp.l4, cin, 0), L.view(grp.l8, cin, 0)
        ep.bias = lp.bias
        L.check(lib.sr_conv3x3(self.code, B, H, W, C.byref(cc), gates.data_ptr(), wdyn.data_ptr(), 9 * cout * cin,
                               B * 9 * cout * cin if split else 0, cout, C.byref(ep), out_view, st), 'sr_conv3x3')
        rec.update(gates=gates, w=wdyn, mods=mods, lp=lp, keep=_keep)
        return rec

    def layer_backward(self, rec, dyn, grads, need_dx=True):
        """dyn: (B, H, W, cout) gradient wrt the layer's conv output (before its epilogue activation / residuals, which the
        caller has already folded into dyn).  Adds the layer's input-side gradients to its group's accumulators and its
        parameter gradients to `grads` ({parameter: tensor})."""
        lib, st, dev = self.lib, self.st, self.dev
        m, grp, cin = rec['m'], rec['grp'], rec['cin']
        B, H, W = grp.B, grp.H, grp.W
        cout = m.out_channels
        first = grp.open_backward(cin == grp.C and (rec['active'] or not grp.has_bands))
        xv = L.view(grp.x, cin, 0)
        cc = L.Concat()
        cc.nseg = 1
        cc.x[0] = xv
        params = dict(m.named_parameters(recurse=False)) if not rec['active'] else _named_params(m)[2]
        if not rec['active']:
            wp = rec['w']
            dw = torch.empty(9, cout, cin, dtype=torch.float32, device=dev)
            db = torch.empty(cout, dtype=torch.float32, device=dev) if m.bias is not None else None
            L.check(lib.sr_conv3x3_wgrad(self.code, B, H, W, C.byref(cc), None, L.view(dyn), cout, 0, dw.data_ptr(), L.ptr(db), st),
                    'sr_conv3x3_wgrad')
            grads[params['weight']] = dw.permute(1, 2, 0).reshape(cout, cin, 3, 3)
            if db is not None:
                grads[params['bias']] = db
            if need_dx:
                wt = _transposed(wp, 1, cout, cin)
                da = _conv(dyn, _concat1(dyn), None, wt, 0, cin, None, 9 * cin * cout if wt.shape[0] == 2 else 0)
                dxv = L.view(grp.dx, cin, 0)
                self.combine(B, H, W, cin, [L.view(da)] + ([] if first else [dxv]), [1.0] + ([] if first else [1.0]), out=dxv)
            return
        lp, gates, wdyn = rec['lp'], rec['gates'], rec['w']
        cc.bands = 1
        cc.l2[0], cc.l4[0], cc.l8[0] = L.view(grp.l2, cin, 0), L.view(grp.l4, cin, 0), L.view(grp.l8, cin, 0)
        # 1. weight / bias gradients of the convolution
        dw = torch.empty(B, 9, cout, cin, dtype=torch.float32, device=dev)
        db = torch.empty(cout, dtype=torch.float32, device=dev) if m.bias is not None else None
        L.check(lib.sr_conv3x3_wgrad(self.code, B, H, W, C.byref(cc), gates.data_ptr(), L.view(dyn), cout, 1, dw.data_ptr(), L.ptr(db), st),
                'sr_conv3x3_wgrad')
        if db is not None:
            grads[params['bias']] = db
        # 2. gradient wrt the modulated operand: convolution of dy with the transposed, mirrored weights
        wt = _transposed(wdyn, B, cout, cin)
        da = _conv(dyn, _concat1(dyn), None, wt, 9 * cin * cout, cin, None, B * 9 * cin * cout if wt.shape[0] == 2 else 0)
        # 3. band mix: the four linear branches are ADDED to the group's accumulators (first consumer: written)
        dz = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
        mfma_gates = self.dt == torch.bfloat16 and cin % 32 == 0
        # gate-conv weight gradients on the matrix cores: the weight-gradient kernel with the logit gradients (three channels
        # in a 32-wide bf16 pixel, written by the same kernel) in the place of dy
        dzp = torch.empty(B, H, W, 32, dtype=torch.bfloat16, device=dev) if mfma_gates else None
        dxv = L.view(grp.dx, cin, 0)
        L.check(lib.sr_fbm_backward(self.code, B, H, W, xv, cc.l2[0], cc.l4[0], cc.l8[0], gates.data_ptr(), L.view(da), dxv,
                                    L.view(grp.dl2, cin, 0), L.view(grp.dl4, cin, 0), L.view(grp.dl8, cin, 0), dz.data_ptr(),
                                    0 if first else 1, L.ptr(dzp), st), 'sr_fbm_backward')
        # 4. the weight path: dW_b -> parameters and pooled means.  Matrix-core gate path: the pooled-mean gradients of all
        # consumers are summed in the group's (B, C) buffer and enter once per slice (slice_grad); else per layer (step 5)
        if mfma_gates:
            grp.use_dpool = True
            wgrads, dpooled = weight_path_backward(m, lp, grp.pooled, grp.C, rec['mods'], dw, self.training, dpooled=grp.dpool,
                                                   dpooled_ld=grp.C, accumulate=True)
        else:
            wgrads, dpooled = weight_path_backward(m, lp, grp.pooled, grp.C, rec['mods'], dw, self.training)
        for name, gten in wgrads.items():
            grads[params[name]] = gten
        # 5. gate convs: weight gradients, and their input gradient added to the direct branch
        if mfma_gates:
            dw32 = torch.empty(9, 32, cin, dtype=torch.float32, device=dev)
            db32 = torch.empty(32, dtype=torch.float32, device=dev)
            cx = L.Concat()
            cx.nseg = 1
            cx.x[0] = xv
            L.check(lib.sr_conv3x3_wgrad(self.code, B, H, W, C.byref(cx), None, L.view(dzp), 32, 0, dw32.data_ptr(), db32.data_ptr(), st),
                    'sr_conv3x3_wgrad (gate convs)')
            # (9, cin) slices of dw32 viewed as (1, cin, 3, 3): no copy (the optimizer's flatten reads them as they lie)
            gate_w = [dw32[:, k, :].t().unsqueeze(0).view(1, cin, 3, 3) for k in range(3)]
            gate_b = [db32[k:k + 1] for k in range(3)]
            # input gradient of the three gate convs: ONE 32 -> cin convolution of the padded logit gradients on the matrix
            # cores, the direct branch as its residual operand (the VALU kernel took 200 us per 192-channel layer)
            gwt = torch.empty(2, 9, cin, 32, dtype=torch.bfloat16, device=dev)
            L.check(lib.sr_pack_gate_weight_transposed(C.byref(lp), gwt[0].data_ptr(), gwt[1].data_ptr(), st), 'sr_pack_gate_weight_transposed')
            ep = L.Epilogue()
            ep.r1, ep.s1 = dxv, 1.0
            L.check(lib.sr_conv3x3(self.code, B, H, W, C.byref(_concat1(dzp)), None, gwt.data_ptr(), 0, 9 * cin * 32, cin, C.byref(ep), dxv, st),
                    'sr_conv3x3 (transposed gate convs)')
        else:
            dgw = torch.empty(3, cin, 9, dtype=torch.float32, device=dev)
            dgb = torch.empty(3, dtype=torch.float32, device=dev)
            gate_w = [dgw[k].reshape(1, cin, 3, 3) for k in range(3)]
            gate_b = [dgb[k:k + 1] for k in range(3)]
            nv = L.View()
            L.check(lib.sr_fbm_gates_backward(self.code, B, H, W, xv, dz.data_ptr(), C.byref(lp), dxv, nv, nv, nv, dpooled.data_ptr(), cin, dxv,
                                              dgw.data_ptr(), dgb.data_ptr(), st), 'sr_fbm_gates_backward')
        for k in range(3):
            grads[params['FBM.freq_weight_conv_list.%d.weight' % k]] = gate_w[k]
            grads[params['FBM.freq_weight_conv_list.%d.bias' % k]] = gate_b[k]

    def slice_grad(self, grp, plan, c0, Cc, act=None, slope=1.0, extras=()):
        """Gradient wrt channels [c0, c0 + Cc) of grp.x once all their consumers have run: direct branch + adjoint low-pass
        of the three band accumulators (+ residual-path terms `extras` = [(sr_view, scale)]), times the LeakyReLU mask of
        the layer that produced the slice (act = its output) -- i.e. the dy of that producing layer."""
        B, H, W = grp.B, grp.H, grp.W
        terms, scales, held = [L.view(grp.dx, Cc, c0)], [1.0], []
        if grp.has_bands:
            for k, acc in enumerate((grp.dl2, grp.dl4, grp.dl8)):
                out = torch.empty(B, H, W, Cc, dtype=self.dt, device=self.dev)
                self.lowpass(plan, grp, c0, Cc, src=L.view(acc, Cc, c0), only=k, outs=L.view(out))
                terms.append(L.view(out))
                scales.append(1.0)
                held.append(out)   # alive until the combine below has been enqueued
        for v, sc in extras:
            terms.append(v)
            scales.append(sc)
        if grp.use_dpool:  # summed pooled-mean gradients of the slice's consumers: + dpool[b][c] / (H W) at every pixel
            return self.combine(B, H, W, Cc, terms, scales, act=act, slope=slope, bc=grp.dpool.data_ptr() + 4 * c0, bc_ld=grp.C,
                                bc_scale=1.0 / (H * W))
        return self.combine(B, H, W, Cc, terms, scales, act=act, slope=slope)

    # ---- the network --------------------------------------------------------------------------------------------
    def forward(self, x):
        net, lib, dt, dev, st = self.net, self.lib, self.dt, self.dev, self.st
        r = {4: 1, 2: 2, 1: 4}[net.scale]
        B, Cin0, Hin, Win = x.shape
        if Cin0 != net.num_in_ch:
            raise RuntimeError('expected %d input channels, got %d' % (net.num_in_ch, Cin0))
        if Hin % r or Win % r:
            raise RuntimeError('input size must be divisible by %d for scale %d' % (r, net.scale))
        H, W = Hin // r, Win // r
        nf, gc, nb = net.num_feat, net.num_grow_ch, net.num_block
        act = self.active
        self.r, self.B, self.H, self.W = r, B, H, W
        # network input, NHWC (pixel_unshuffle of scale 2 / 1 folded into the pack)
        g_in = _Group(B, H, W, Cin0 * r * r, dt, dev, False)
        L.check(lib.sr_pack_nchw(x.contiguous().data_ptr(), L.dtype_code(x.dtype), g_in.x.data_ptr(), self.code, B, Cin0, Hin, Win, r, st),
                'sr_pack_nchw')
        rdbs = [rdb for blk in net.body for rdb in (blk.rdb1, blk.rdb2, blk.rdb3)]
        ctot = nf + 4 * gc
        groups = []
        for rdb in rdbs:
            convs = (rdb.conv1, rdb.conv2, rdb.conv3, rdb.conv4, rdb.conv5)
            groups.append(_Group(B, H, W, ctot, dt, dev, any(act(m) for m in convs)))
        g_body = _Group(B, H, W, nf, dt, dev, act(net.conv_body))
        first_out = groups[0] if groups else g_body
        self.rec_first = self.layer_forward(net.conv_first, g_in, Cin0 * r * r, L.view(first_out.x, nf, 0))
        self.g_in, self.groups, self.g_body = g_in, groups, g_body
        self.recs = []
        for n, rdb in enumerate(rdbs):
            grp = groups[n]
            nxt = groups[n + 1] if n + 1 < len(groups) else g_body
            convs = (rdb.conv1, rdb.conv2, rdb.conv3, rdb.conv4, rdb.conv5)
            plan = self.plan(next(m for m in convs if act(m)), H, W) if grp.has_bands else None
            if grp.has_bands:
                self.lowpass(plan, grp, 0, nf)
            recs = []
            for k, m in enumerate(convs):
                cin = nf + k * gc
                if k < 4:
                    recs.append(self.layer_forward(m, grp, cin, L.view(grp.x, gc, cin), lrelu=True))
                    if grp.has_bands:
                        self.lowpass(plan, grp, cin, gc)
                else:  # x5 * 0.2 + x ; the last dense block of an RRDB also takes the block's residual: (.) * 0.2 + block input
                    last = n % 3 == 2
                    recs.append(self.layer_forward(m, grp, cin, L.view(nxt.x, nf, 0), r1=L.view(grp.x, nf, 0), s1=0.2,
                                                   r2=L.view(groups[n - 2].x, nf, 0) if last else None, s2=0.2))
            self.recs.append(recs)
        # conv_body + trunk residual
        if g_body.has_bands:
            self.lowpass(self.plan(net.conv_body, H, W), g_body, 0, nf)
        f2 = torch.empty(B, H, W, nf, dtype=dt, device=dev)
        self.rec_body = self.layer_forward(net.conv_body, g_body, nf, L.view(f2), r1=L.view(first_out.x, nf, 0), s1=1.0)
        # upsampling head
        g_u1 = _Group(B, 2 * H, 2 * W, nf, dt, dev, act(net.conv_up1), x=self.resample(f2, False))
        if g_u1.has_bands:
            self.lowpass(self.plan(net.conv_up1, 2 * H, 2 * W), g_u1, 0, nf)
        v1 = torch.empty(B, 2 * H, 2 * W, nf, dtype=dt, device=dev)
        self.rec_up1 = self.layer_forward(net.conv_up1, g_u1, nf, L.view(v1), lrelu=True)
        g_u2 = _Group(B, 4 * H, 4 * W, nf, dt, dev, act(net.conv_up2), x=self.resample(v1, False))
        if g_u2.has_bands:
            self.lowpass(self.plan(net.conv_up2, 4 * H, 4 * W), g_u2, 0, nf)
        g_hr = _Group(B, 4 * H, 4 * W, nf, dt, dev, act(net.conv_hr))
        self.rec_up2 = self.layer_forward(net.conv_up2, g_u2, nf, L.view(g_hr.x), lrelu=True)
        if g_hr.has_bands:
            self.lowpass(self.plan(net.conv_hr, 4 * H, 4 * W), g_hr, 0, nf)
        g_last = _Group(B, 4 * H, 4 * W, nf, dt, dev, act(net.conv_last))
        self.rec_hr = self.layer_forward(net.conv_hr, g_hr, nf, L.view(g_last.x), lrelu=True)
        if g_last.has_bands:
            self.lowpass(self.plan(net.conv_last, 4 * H, 4 * W), g_last, 0, nf)
        pred = torch.empty(B, 4 * H, 4 * W, net.num_out_ch, dtype=dt, device=dev)
        self.rec_last = self.layer_forward(net.conv_last, g_last, nf, L.view(pred))
        self.v1, self.g_u1, self.g_u2, self.g_hr, self.g_last = v1, g_u1, g_u2, g_hr, g_last
        self.ws = None
        if self.bn_counters:
            torch._foreach_add_(self.bn_counters, 1)
            self.bn_counters = []
        return _unpack(pred).to(self.io_dt)

    def backward(self, dpred, need_dx):
        net = self.net
        B, H, W = self.B, self.H, self.W
        nf, gc = net.num_feat, net.num_grow_ch
        grads = {}
        pl = lambda m, h, w: self.plan(m, h, w) if self.active(m) else None
        dyn = _pack(dpred.to(self.dt))
        # conv_last <- conv_hr <- conv_up2 <- conv_up1 (each tensor has ONE consumer here)
        self.layer_backward(self.rec_last, dyn, grads)
        dy = self.slice_grad(self.g_last, pl(net.conv_last, 4 * H, 4 * W), 0, nf, act=L.view(self.g_last.x), slope=0.2)
        self.layer_backward(self.rec_hr, dy, grads)
        dy = self.slice_grad(self.g_hr, pl(net.conv_hr, 4 * H, 4 * W), 0, nf, act=L.view(self.g_hr.x), slope=0.2)
        self.layer_backward(self.rec_up2, dy, grads)
        d_u2 = self.slice_grad(self.g_u2, pl(net.conv_up2, 4 * H, 4 * W), 0, nf)
        d_v1 = self.resample(d_u2, True)
        dy = self.combine(B, 2 * H, 2 * W, nf, [L.view(d_v1)], [1.0], act=L.view(self.v1), slope=0.2)
        self.layer_backward(self.rec_up1, dy, grads)
        d_u1 = self.slice_grad(self.g_u1, pl(net.conv_up1, 2 * H, 2 * W), 0, nf)
        d_f2 = self.resample(d_u1, True)          # gradient wrt f2 = conv_body(body) + feat
        self.layer_backward(self.rec_body, d_f2, grads)
        dout = self.slice_grad(self.g_body, pl(net.conv_body, H, W), 0, nf)   # gradient wrt the trunk's output
        # the dense blocks, last to first
        for n in range(len(self.groups) - 1, -1, -1):
            grp, recs = self.groups[n], self.recs[n]
            j = n % 3
            if j == 2:
                dblk = dout                         # gradient wrt the RRDB's output: out = rdb3 * 0.2 + block input
            res_scale = 0.2 if j == 2 else 1.0      # what reaches this dense block's output from `dout`
            plan = next((self.plan(r_['m'], H, W) for r_ in recs if r_['active']), None)
            dy5 = self.combine(B, H, W, nf, [L.view(dout)], [0.2 * res_scale])   # x5 enters as x5 * 0.2
            self.layer_backward(recs[4], dy5, grads)
            for k in (3, 2, 1, 0):
                c0 = nf + k * gc
                dy = self.slice_grad(grp, plan, c0, gc, act=L.view(grp.x, gc, c0), slope=0.2)
                self.layer_backward(recs[k], dy, grads)
            extras = [(L.view(dout), res_scale)]    # the dense block's own residual (+ x)
            if j == 0:
                extras.append((L.view(dblk), 1.0))  # the RRDB's residual (+ block input)
            dout = self.slice_grad(grp, plan, 0, nf, extras=extras)
        # conv_first: the trunk input also feeds the long residual feat + conv_body(...)
        dfeat = self.combine(B, H, W, nf, [L.view(dout), L.view(d_f2)], [1.0, 1.0])
        self.layer_backward(self.rec_first, dfeat, grads, need_dx=need_dx)
        dx = None
        if need_dx:
            if self.r != 1:
                raise NotImplementedError('gradient wrt the input of a scale-%d network (pixel_unshuffle) is not implemented' % net.scale)
            dx = _unpack(self.g_in.dx).to(self.io_dt)
        return dx, grads


class _NetFn(torch.autograd.Function):
    """RRDBNet.forward as one autograd node; arguments after the module are its parameters (so that autograd routes their
    gradients)."""

    @staticmethod
    def forward(ctx, x, net, training, *params):
        with torch.cuda.device(x.device):
            run = _Runner(net, x, training)
            y = run.forward(x)
        ctx.run, ctx.params = run, params
        ctx.need_dx = ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        run = ctx.run
        if run is None:
            raise RuntimeError('RRDBNet (train): backward through the same forward twice is not supported')
        with torch.cuda.device(dy.device):
            dx, grads = run.backward(dy, ctx.need_dx)
        ctx.run = None  # free the activations
        return (dx, None, None) + tuple(grads.get(p) for p in ctx.params)


def rrdbnet_train_forward(net, x):
    """RRDBNet.forward (basicsr rrdbnet_arch.py; call site train/train_realESRGAN_fdconv_1.py:114) with an autograd graph:
    ONE node for the whole network.  No torch operator runs between the layers: the dense-block concat is a channel-sliced
    buffer, LeakyReLU / residual scaling live in the conv epilogues (forward) and in sr_grad_combine (backward), nearest 2x
    and its adjoint are sr_upsample2x, low-pass planes and gates exist once per produced tensor."""
    _check_input(x, 'RRDBNet (train)')
    cache = net.__dict__.setdefault('_sr_train_params', None)
    params = cache if cache is not None else list(net.parameters())
    net.__dict__['_sr_train_params'] = params
    return _NetFn.apply(x, net, net.training, *params)


# ------------------------------------------------------------------------------------------------------------------
# data-parallel gradient exchange
# ------------------------------------------------------------------------------------------------------------------
def allreduce_gradients(parameters, group=None, bucket_bytes=25 << 20, average=True):
    """Sum (or average) `.grad` of the parameters over the ranks of `group`: the one exchange of data-parallel training
    (reference: nn.DataParallel's gradient reduction, train/train_realESRGAN_fdconv_OST_1.py:238-246).

    Gradients are flattened into buckets of `bucket_bytes` (~25 MB: 82 MB of RRDBNet-FDConv gradients -> 4 all-reduces
    per step, each long enough to run at the xGMI link rate), every bucket's all-reduce is issued asynchronously and the
    results are scattered back once all have completed.  Parameters without a gradient count as zeros so that every rank
    issues the same collectives.  BatchNorm running statistics are NOT exchanged (local statistics, like the reference)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    params = [p for p in parameters if p.requires_grad]
    buckets, cur, size = [], [], 0
    for p in params:
        nbytes = p.numel() * p.element_size()
        if cur and (size + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        buckets.append(cur)
    work = []
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        work.append((bucket, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)))
    for bucket, flat, handle in work:
        handle.wait()
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            piece = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = piece.clone()
            else:
                p.grad.copy_(piece)
            off += n
    return len(buckets)


class FlatAdam(torch.optim.Optimizer):
    """Adam (torch.optim.Adam's update rule: `weight_decay` L2, no amsgrad) over ONE flat float32 buffer.

    RRDBNet-FDConv has ~6000 parameter tensors of a few KB each; per step torch.optim.Adam spends 0.18 s of host time
    walking them and the bucketed all-reduce another 6000 small copies scattering the result back.  Here the parameters
    are re-pointed to views of one buffer; a step gathers the gradients into a flat buffer with one `torch.cat`, runs ONE
    all-reduce over it when a process group is initialised (data-parallel training: SURVEY 8e), optionally clips the global
    norm, and updates parameters and moments with five tensor ops.  Parameters that received no gradient in a step count
    as zero gradients (torch.optim.Adam skips them; every parameter of the networks here gets one).

    It is a `torch.optim.Optimizer`: the learning-rate schedulers of the owner scripts accept it
    (train/train_realESRGAN_fdconv.py:121 ReduceLROnPlateau; the rate is read from `param_groups[0]` at every step) and
    `state_dict()` / `load_state_dict()` use torch.optim.Adam's own layout (per-parameter `step`, `exp_avg`, `exp_avg_sq`),
    so a checkpoint written as in train/train_realESRGAN_fdconv_OST_1.py:401 resumes under either optimizer.
    In a process group of more than one rank the constructor broadcasts rank 0's parameters, so that replicas start equal
    whatever their seeds (`broadcast=False` to skip).

    Usage: `opt = FlatAdam(net.parameters(), lr=1e-4)`; `opt.zero_grad()`; ...; `opt.step()` -- or pass it to `train_step`.
    Construct it AFTER the module has been moved / cast to its final device and dtype: `.to()` / `.float()` re-allocate
    the parameters and break the aliasing (step() checks and raises)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, broadcast=True, group=None):
        params = list(params)
        if params and isinstance(params[0], dict):
            raise ValueError('FlatAdam: one parameter group (pass the parameters, not a list of groups)')
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError('FlatAdam: no parameters')
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError('FlatAdam: float32 parameters on one device')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.params = self.param_groups[0]['params']
        self.sizes = [p.numel() for p in self.params]
        n = sum(self.sizes)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        self._offsets = []
        with torch.no_grad():
            for p, k in zip(self.params, self.sizes):
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)   # the module's parameters now alias the flat buffer
                self._offsets.append(off)
                off += k
        self._ptrs = [p.data_ptr() for p in self.params]
        self.grad = None  # the flat gradient of the last step
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        self._zeros = None
        import torch.distributed as dist
        if broadcast and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            torch.autograd.graph.increment_version(self.params)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.detach_().zero_()

    def gather_gradients(self):
        """p.grad of every parameter -> one flat tensor (a single C++ flatten: a Python loop of 6000 reshapes cost 0.1 s per
        step); zeros take the place of missing gradients."""
        grads = []
        for p, k in zip(self.params, self.sizes):
            g = p.grad
            if g is None:
                if self._zeros is None:
                    self._zeros = torch.zeros(max(self.sizes), dtype=torch.float32, device=self.flat.device)
                g = self._zeros[:k]
            grads.append(g)
        self.grad = torch._utils._flatten_dense_tensors(grads)
        return self.grad

    @torch.no_grad()
    def step(self, closure=None, group=None, average=True, clip_grad_norm=None):
        import torch.distributed as dist
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if [p.data_ptr() for p in self.params] != self._ptrs:
            raise RuntimeError('FlatAdam: a parameter no longer aliases the flat buffer (the module was moved or cast with '
                               '.to() / .float() / .cuda() after the optimizer was built) -- rebuild the optimizer')
        g = self.gather_gradients()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if g.is_cuda and dist.get_backend(group) == 'gloo':
                # gloo reduces host memory (its device path stalls when ranks share one GPU: seen with two ranks on one
                # MI355X); RCCL -- every real multi-GPU run -- reduces the device buffer in place
                h = g.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                g.copy_(h)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
            if average:
                g /= dist.get_world_size(group)
        if clip_grad_norm is not None:  # nn.utils.clip_grad_norm_: scale by max_norm / (norm + 1e-6), clamped to 1
            g *= (clip_grad_norm / (g.norm() + 1e-6)).clamp(max=1.0)
        hp = self.param_groups[0]
        b1, b2 = hp['betas']
        self.steps += 1
        if hp['weight_decay']:
            g.add_(self.flat, alpha=hp['weight_decay'])
        self.exp_avg.lerp_(g, 1.0 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1, bc2 = 1.0 - b1 ** self.steps, 1.0 - b2 ** self.steps
        denom = (self.exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(hp['eps'])
        self.flat.addcdiv_(self.exp_avg, denom, value=-hp['lr'] / bc1)
        # the update went through the flat buffer: a parameter re-pointed with `.data =` keeps its own version counter, and
        # everything keyed on it (FDConv's cached spectral bases, autograd's saved-tensor checks) must see the change
        torch.autograd.graph.increment_version(self.params)
        return loss

    def state_dict(self):
        """torch.optim.Adam's layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}."""
        state = {}
        if self.steps:
            for i, (p, off, k) in enumerate(zip(self.params, self._offsets, self.sizes)):
                state[i] = {'step': torch.tensor(float(self.steps)),
                            'exp_avg': self.exp_avg[off:off + k].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + k].view(p.shape).clone()}
        hp = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        hp['params'] = list(range(len(self.params)))
        return {'state': state, 'param_groups': [hp]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.params):
            raise ValueError('FlatAdam.load_state_dict: expected one group of %d parameters' % len(self.params))
        for k, v in groups[0].items():
            if k != 'params' and k in self.param_groups[0]:
                self.param_groups[0][k] = v
        state = sd['state']
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = 0
        for key, st in state.items():
            i = int(key)
            off, k = self._offsets[i], self.sizes[i]
            if tuple(st['exp_avg'].shape) != tuple(self.params[i].shape):
                raise ValueError('FlatAdam.load_state_dict: parameter %d has shape %s, state %s'
                                 % (i, tuple(self.params[i].shape), tuple(st['exp_avg'].shape)))
            self.exp_avg[off:off + k].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
            steps = max(steps, int(float(st['step'])))
        self.steps = steps


def train_step(net, optimizer, wf, sr, weighted=False, sigmoid=False, clip_grad_norm=None, group=None, perceptual=None):
    """One optimisation step shaped as train/train_realESRGAN_fdconv_1.py:113-118 (weighted / sigmoid / clip: the variants of
    train/train_realESRGAN_fdconv.py:107-142): forward, bilinear resize to the target size + L1 (one fused HIP kernel that
    also returns d loss / d pred), backward, gradient all-reduce when torch.distributed is initialised, optimizer step.
    perceptual: None, or a dict(vgg=VGG19Features, layer_weights=..., weight=1.0, compute_dtype=...) -- adds the VGG-feature
    L1 term of the Real-ESRGAN recipe (options/train_realesrgan_x4plus.yml:129-149: L1 + perceptual, BASELINE configs[3]).
    Returns the loss (a 0-d device tensor)."""
    from .head import resize_l1_loss
    net.train()
    optimizer.zero_grad()
    pred = net(wf)
    loss = resize_l1_loss(pred, sr, weighted=weighted, sigmoid=sigmoid)
    if perceptual is not None:
        from .perceptual import perceptual_loss
        loss = loss + perceptual_loss(pred, sr, perceptual['vgg'], perceptual.get('layer_weights'), perceptual.get('weight', 1.0),
                                      perceptual.get('compute_dtype', torch.bfloat16))
    loss.backward()
    if isinstance(optimizer, FlatAdam):  # gathers, all-reduces and clips the flat gradient itself
        optimizer.step(group=group, clip_grad_norm=clip_grad_norm)
        return loss.detach()
    allreduce_gradients(net.parameters(), group=group)
    if clip_grad_norm is not None:
        nn.utils.clip_grad_norm_(net.parameters(), clip_grad_norm)
    optimizer.step()
    return loss.detach()
