"""MI355X-native RRDBNet / FDConv / RealESRGANer (drop-in for the cdrookie/Super_resolution hot path).

    from super_resolution_amd import RRDBNet, FDConv, replace_conv_with_fdconv, RealESRGANer

Everything numerical runs in hand-written HIP kernels (libsr_amd.so, gfx950) behind a C ABI
(include/sr_amd.h); PyTorch supplies device memory, streams and torch.distributed only.
"""
from .fdconv import FDConv, replace_conv_with_fdconv  # noqa: F401
from .rrdbnet import RRDBNet  # noqa: F401
from .realesrganer import RealESRGANer  # noqa: F401
from .srvgg import SRVGGNetCompact  # noqa: F401
from . import convert, head, perceptual, train  # noqa: F401

__all__ = ['FDConv', 'RRDBNet', 'RealESRGANer', 'SRVGGNetCompact', 'replace_conv_with_fdconv']
