"""SRVGGNetCompact (realesr-animevideov3 / realesr-general family) -- drop-in for
Real-ESRGAN-master/realesrgan/archs/srvgg_arch.py:7-69 of the reference.

Same constructor and module tree (`body` = conv, act, (conv, act) x num_conv, conv; `upsampler`), hence the same
`state_dict()` layout (`body.{i}.weight` / `.bias`, PReLU slopes as `body.{i}.weight` of shape (num_feat,)).
`forward` runs on libsr_amd: every 3x3 conv on the conv kernels with the activation as a per-channel-slope epilogue
(PReLU: the learned slopes, ReLU: zeros, LeakyReLU: 0.1), the last conv's output channels padded to a whole MFMA tile, and
one fused kernel for PixelShuffle + the nearest-upsampled input (srvgg_arch.py:65-68).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L


class SRVGGNetCompact(nn.Module):
    """SRVGGNetCompact(num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4, act_type='prelu').

    forward(x): (B, num_in_ch, H, W) float32 or bfloat16 CUDA tensor -> (B, num_out_ch, H*upscale, W*upscale), same dtype;
    computes in that dtype or in `compute_dtype` if set (RealESRGANer(half=True) sets bfloat16)."""

    def __init__(self, num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4, act_type='prelu'):
        super().__init__()
        if act_type not in ('relu', 'prelu', 'leakyrelu'):
            raise ValueError('act_type must be relu, prelu or leakyrelu')
        if num_in_ch != num_out_ch:
            raise ValueError('the nearest-upsampled input is added to the output: num_in_ch must equal num_out_ch')
        self.num_in_ch, self.num_out_ch, self.num_feat = num_in_ch, num_out_ch, num_feat
        self.num_conv, self.upscale, self.act_type = num_conv, upscale, act_type

        def act():
            if act_type == 'relu':
                return nn.ReLU(inplace=True)
            if act_type == 'prelu':
                return nn.PReLU(num_parameters=num_feat)
            return nn.LeakyReLU(negative_slope=0.1, inplace=True)

        self.body = nn.ModuleList([nn.Conv2d(num_in_ch, num_feat, 3, 1, 1), act()])
        for _ in range(num_conv):
            self.body.append(nn.Conv2d(num_feat, num_feat, 3, 1, 1))
            self.body.append(act())
        self.body.append(nn.Conv2d(num_feat, num_out_ch * upscale * upscale, 3, 1, 1))
        self.upsampler = nn.PixelShuffle(upscale)
        self.compute_dtype = None
        self._sr = None

    def _apply(self, fn, *a, **k):
        self._sr = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._sr = None
        return super().load_state_dict(*a, **k)

    def _layers(self, dtype, dev):
        """[(LayerParams-like dict)] per conv: packed weights, bias, slopes; cached per dtype / parameter version."""
        convs = [m for m in self.body if isinstance(m, nn.Conv2d)]
        acts = [m for m in self.body if not isinstance(m, nn.Conv2d)]
        sig = (dtype, str(dev)) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._sr is not None and self._sr[0] == sig:
            return self._sr[1]
        out = []
        for i, m in enumerate(convs):
            w, b = m.weight.data, m.bias.data
            L.require_gpu(w, 'SRVGGNetCompact')
            cout, cin = m.out_channels, m.in_channels
            last = i == len(convs) - 1
            # bf16: the MFMA kernel takes 32-channel output tiles or thin (<= 16) outputs; pad the LAST conv's 48 to 64 (its
            # destination is sized for the padded count and sr_pixel_shuffle_add reads the first num_out_ch * r * r).
            # Hidden layers are never padded -- their slopes, ping-pong buffers and the next layer's cin are num_feat wide;
            # a num_feat that is not a multiple of 32 takes the generic conv kernel.
            cpad = cout
            if last and dtype == torch.bfloat16 and cout > 16 and cout % 32 and cin % 32 == 0:
                cpad = (cout + 31) // 32 * 32
            if cpad != cout:
                w = torch.cat([w, w.new_zeros(cpad - cout, cin, 3, 3)], 0)
                b = torch.cat([b, b.new_zeros(cpad - cout)], 0)
            # bf16 head: the image is packed as ONE 32-channel K-chunk (zeros beyond num_in_ch, sr_pack_nchw_padded) and the
            # weights are padded to match, so the 3 -> 64 layer runs on the matrix cores like the body (the VALU kernel that
            # takes thin inputs is bound by its instruction count: 1.17 ms against 0.45 ms for four 1080p frames)
            if i == 0 and dtype == torch.bfloat16 and cin < 32 and cout % 32 == 0:
                w = torch.cat([w, w.new_zeros(w.shape[0], 32 - cin, 3, 3)], 1)
                cin = 32
            wp, lo = L.pack_plain_weight(w.float().contiguous(), dtype, cpad, cin)
            slope = None
            if not last:
                a = acts[i]
                if isinstance(a, nn.PReLU):
                    slope = a.weight.data.float().contiguous()
                elif isinstance(a, nn.LeakyReLU):
                    slope = torch.full((cout,), a.negative_slope, dtype=torch.float32, device=dev)
                else:
                    slope = torch.zeros(cout, dtype=torch.float32, device=dev)
            out.append(dict(cin=cin, cout=cpad, w=wp, lo=lo, bias=b.float().contiguous(), slope=slope))
        self._sr = (sig, out)
        return out

    def forward(self, x):
        L.require_gpu(x, 'SRVGGNetCompact.forward')
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError('SRVGGNetCompact on MI355X is inference only: call .eval() / torch.no_grad()')
        lib = L.lib()
        io_dt = x.dtype
        dt = self.compute_dtype or io_dt
        code, io_code = L.dtype_code(dt), L.dtype_code(io_dt)
        B, Cc, H, W = x.shape
        if Cc != self.num_in_ch:
            raise RuntimeError('expected %d input channels, got %d' % (self.num_in_ch, Cc))
        dev = x.device
        x = x.contiguous()
        r = self.upscale
        with torch.cuda.device(dev):
            st = L.stream_ptr(dev)
            layers = self._layers(dt, dev)
            cin0 = layers[0]['cin']
            cur = torch.empty(B, H, W, cin0, dtype=dt, device=dev)
            if cin0 != Cc:  # padded bf16 head (see _layers)
                L.check(lib.sr_pack_nchw_padded(x.data_ptr(), io_code, cur.data_ptr(), B, Cc, H, W, cin0, st), 'sr_pack_nchw_padded')
            else:
                L.check(lib.sr_pack_nchw(x.data_ptr(), io_code, cur.data_ptr(), code, B, Cc, H, W, 1, st), 'sr_pack_nchw')
            bufs = [torch.empty(B, H, W, self.num_feat, dtype=dt, device=dev) for _ in range(2)]
            for i, ly in enumerate(layers):
                last = i == len(layers) - 1
                dst = torch.empty(B, H, W, ly['cout'], dtype=dt, device=dev) if last else bufs[i & 1]
                cc = L.Concat()
                cc.nseg = 1
                cc.x[0] = L.view(cur)
                ep = L.Epilogue()
                ep.bias = ly['bias'].data_ptr()
                ep.slope = None if ly['slope'] is None else ly['slope'].data_ptr()
                L.check(lib.sr_conv3x3(code, B, H, W, C.byref(cc), None, ly['w'].data_ptr(), 0, ly['lo'], ly['cout'], C.byref(ep),
                                       L.view(dst), st), 'sr_conv3x3')
                cur = dst
            y = torch.empty(B, self.num_out_ch, H * r, W * r, dtype=io_dt, device=dev)
            L.check(lib.sr_pixel_shuffle_add(L.view(cur), code, x.data_ptr(), y.data_ptr(), io_code, B, self.num_out_ch, H, W, r, st),
                    'sr_pixel_shuffle_add')
        return y
