"""ctypes binding of libsr_amd.so (include/sr_amd.h).

The library is built in-tree by `super_resolution_amd/build.py` (hipcc, gfx950).  There is no
fallback: if it is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SR_AMD_LIB: an alternative build of the same library (A/B experiments; never a different implementation)
LIB_PATH = os.environ.get('SR_AMD_LIB') or os.path.join(_HERE, 'libsr_amd.so')

SR_F32, SR_BF16 = 0, 1
SR_MAX_SEG = 8


class View(C.Structure):
    _fields_ = [('p', C.c_void_p), ('ld', C.c_int32), ('c0', C.c_int32), ('C', C.c_int32), ('cbs', C.c_int32),
                ('bs', C.c_int64)]


class Concat(C.Structure):
    _fields_ = [('nseg', C.c_int32), ('bands', C.c_int32), ('ups', C.c_int32), ('_pad', C.c_int32),
                ('x', View * SR_MAX_SEG), ('l2', View * SR_MAX_SEG), ('l4', View * SR_MAX_SEG), ('l8', View * SR_MAX_SEG)]


class Epilogue(C.Structure):
    _fields_ = [('bias', C.c_void_p), ('lrelu', C.c_int32), ('_pad', C.c_int32),
                ('r1', View), ('s1', C.c_float), ('_pad1', C.c_float),
                ('r2', View), ('s2', C.c_float), ('_pad2', C.c_float), ('slope', C.c_void_p)]


class LayerParams(C.Structure):
    _fields_ = [('cin', C.c_int32), ('cout', C.c_int32), ('active', C.c_int32), ('loc_taps', C.c_int32),
                ('bias', C.c_void_p), ('wpacked', C.c_void_p), ('basis', C.c_void_p), ('fc_w', C.c_void_p),
                ('bn_w', C.c_void_p), ('bn_b', C.c_void_p), ('bn_rm', C.c_void_p), ('bn_rv', C.c_void_p),
                ('relu_s', C.c_void_p), ('relu_b', C.c_void_p),
                ('ch_w', C.c_void_p), ('ch_b', C.c_void_p), ('fl_w', C.c_void_p), ('fl_b', C.c_void_p),
                ('sp_w', C.c_void_p), ('sp_b', C.c_void_p), ('kn_w', C.c_void_p), ('kn_b', C.c_void_p),
                ('gate_w', C.c_void_p * 3), ('gate_b', C.c_void_p * 3),
                ('loc_w', C.c_void_p), ('ln_w', C.c_void_p), ('ln_b', C.c_void_p), ('wpacked_lo_off', C.c_int64)]


WEIGHT_PATH_GRADS = ('dft_weight', 'fc_w', 'bn_w', 'bn_b', 'relu_s', 'relu_b', 'ch_w', 'ch_b', 'fl_w', 'fl_b', 'sp_w', 'sp_b', 'kn_w',
                     'kn_b', 'loc_w', 'ln_w', 'ln_b')


class WeightPathGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in WEIGHT_PATH_GRADS]


class GateConsumer(C.Structure):
    _fields_ = [('w', C.c_void_p * 3), ('b', C.c_void_p * 3), ('logits', C.c_void_p), ('gates', C.c_void_p), ('first', C.c_int32)]


_SIGS = {
    'sr_strerror': (C.c_char_p, [C.c_int]),
    'sr_last_hip_error': (C.c_int, []),
    'sr_version': (C.c_int, []),
    'sr_pack_nchw': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_pack_nchw_padded': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_unpack_nhwc': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_pack_conv_weight': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_fbm_plan_create': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'sr_fbm_plan_destroy': (None, [C.c_void_p]),
    'sr_fbm_masks_host': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'sr_fbm_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'sr_fbm_lowpass': (C.c_int, [C.c_void_p, C.c_int, C.c_int, View, C.c_int, View, View, View, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_fbm_static_scratch_bytes': (C.c_size_t, []),
    'sr_fbm_lowpass_static': (C.c_int, [C.c_void_p, C.c_int, C.c_int, View, C.c_int, View, View, View, C.c_void_p, C.c_int,
                                        C.c_void_p]),
    'sr_fbm_gates': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Concat), C.POINTER(LayerParams), C.c_void_p,
                               C.c_void_p]),
    'sr_fbm_gates_accumulate': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, View, C.c_int, C.POINTER(GateConsumer), C.c_void_p]),
    'sr_fdconv_basis_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'sr_fdconv_basis': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_void_p]),
    'sr_fdconv_weightgen_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'sr_fdconv_weightgen': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(LayerParams), C.c_int, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_fdconv_weightgen_backward_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'sr_fdconv_weightgen_backward': (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.POINTER(LayerParams), C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_float, C.c_void_p, C.POINTER(WeightPathGrads), C.c_void_p, C.c_int,
                                               C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_conv_weight_transpose': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_conv3x3_wgrad': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Concat), C.c_void_p, View, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_fbm_backward': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, View, View, View, View, C.c_void_p, View, View, View, View,
                                  View, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'sr_grad_combine': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(View), C.POINTER(C.c_float), View,
                                  C.c_float, C.c_void_p, C.c_int, C.c_float, View, C.c_void_p]),
    'sr_pack_gate_weight_transposed': (C.c_int, [C.POINTER(LayerParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_upsample2x': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, View, View, C.c_int, C.c_void_p]),
    'sr_fbm_gates_backward': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, View, C.c_void_p, C.POINTER(LayerParams), View, View,
                                        View, View, C.c_void_p, C.c_int, View, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_conv3x3': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Concat), C.c_void_p, C.c_void_p, C.c_long, C.c_long,
                             C.c_int, C.POINTER(Epilogue), View, C.c_void_p]),
    'sr_pixel_shuffle_add': (C.c_int, [View, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    'sr_bilinear_resize': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_float, C.c_void_p]),
    'sr_resize_l1_workspace_bytes': (C.c_size_t, []),
    'sr_resize_l1_loss': (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_relu_pool': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_relu_pool_backward': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'sr_l1_feature_loss': (C.c_int, [C.c_int, C.c_long, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_vgg_prep': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                              C.c_int, C.c_void_p]),
    'sr_bilinear_resize_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_image_to_planes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                     C.c_void_p]),
    'sr_image_from_planes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_void_p]),
    'sr_rrdbnet_create': (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_void_p)]),
    'sr_rrdbnet_destroy': (None, [C.c_void_p]),
    'sr_rrdbnet_num_layers': (C.c_int, [C.c_void_p]),
    'sr_rrdbnet_set_layer': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LayerParams)]),
    'sr_rrdbnet_prepare': (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    'sr_rrdbnet_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    'sr_rrdbnet_forward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_rrdbnet_profile': (C.c_int, [C.c_void_p, C.c_int]),
    'sr_rrdbnet_profile_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_long)]),
    'sr_debug_force_generic_conv': (None, [C.c_int]),
    'sr_debug_force_generic_fft': (None, [C.c_int]),
    'sr_debug_no_mixed_fft': (None, [C.c_int]),
    'sr_debug_operand_f16': (None, [C.c_int]),
    'sr_debug_last_lowpass_path': (C.c_int, []),
    'sr_debug_force_fft_lowpass': (None, [C.c_int]),
    'sr_debug_no_ups_shortcut': (None, [C.c_int]),
    'sr_debug_gates_no_direct': (None, [C.c_int]),
    'sr_debug_mfma_no_fold': (None, [C.c_int]),
    'sr_debug_force_generic_gates': (None, [C.c_int]),
    'sr_debug_split_weights': (None, [C.c_int]),
    'sr_debug_split_two_pass': (None, [C.c_int]),
    'sr_debug_no_resident_weights': (None, [C.c_int]),
    'sr_debug_no_plain_kernel': (None, [C.c_int]),
    'sr_debug_graph_replays': (C.c_long, [C.c_void_p]),
    'sr_debug_enable_graphs': (None, [C.c_int]),
    'sr_debug_wgrad_valu': (None, [C.c_int]),
}

# every symbol include/sr_amd.h declares (the debug hook is not part of the header)
HEADER_SYMBOLS = [n for n in _SIGS if not n.startswith('sr_debug_')]

_lib = None


def lib():
    """The loaded library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('%s is missing: build it with `python -m super_resolution_amd.build` '
                               '(hipcc --offload-arch=gfx950); there is no fallback path' % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(h, name)
            f.restype = res
            f.argtypes = args
        _lib = h
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().sr_strerror(rc).decode()
        raise RuntimeError('libsr_amd %s failed: %s (code %d, hip error %d)' % (what, msg, rc, lib().sr_last_hip_error()))


def dtype_code(dt):
    if dt == torch.float32:
        return SR_F32
    if dt == torch.bfloat16:
        return SR_BF16
    raise TypeError('libsr_amd computes in float32 or bfloat16, got %s' % dt)


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def view(t, C_=None, c0=0):
    """sr_view over the last (channel) axis of a contiguous NHWC tensor."""
    ld = t.shape[-1]
    return View(t.data_ptr(), ld, c0, ld - c0 if C_ is None else C_, 0, 0)


def blocked_view(t):
    """sr_view over a channel-blocked tensor of shape (B, C/cb, H, W, cb), cb a power of two >= 8."""
    B, nb, H, W, e = t.shape
    assert e >= 8 and e & (e - 1) == 0 and t.is_contiguous()
    return View(t.data_ptr(), e, 0, nb * e, e * H * W, nb * e * H * W)


def null_view():
    return View(None, 0, 0, 0, 0, 0)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def require_gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError('%s runs on an MI355X through libsr_amd only; got a %s tensor (no CPU path exists)'
                           % (who, t.device))


def pack_plain_weight(w, dtype, cout, cin):
    """(packed tensor, lo offset in elements) of an nn.Conv2d weight for the conv kernels.  bf16 and a thin layer
    (min(cin, cout) <= 16: conv_first / conv_last, bandwidth-bound whatever the arithmetic): the weights are kept as
    hi + lo pairs -- the rounding error of a weight is coherent over the whole image and, at the first layer of a deep
    network, cost 12 dB of bf16 output PSNR (finding 15).  Wide plain layers (a plain RRDBNet is bound by
    the matrix cores) keep single bf16 weights."""
    split = dtype == torch.bfloat16 and min(cin, cout) <= 16
    n = 9 * cout * cin
    wp = torch.empty((2 if split else 1) * n, dtype=dtype, device=w.device)
    check(lib().sr_pack_conv_weight(w.contiguous().data_ptr(), wp.data_ptr(), wp.data_ptr() + n * wp.element_size() if split else None,
                                    dtype_code(dtype), cout, cin, stream_ptr(w.device)), 'sr_pack_conv_weight')
    return wp, (n if split else 0)
