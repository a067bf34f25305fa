// FBM low-pass planes, register-blocked FFT: the mixed-radix sides of the high-resolution tensors of a tiled image at the
// reference's default tiling (tile 512, pad 10: windows of 522 / 532 pixels, x2 and x4 after conv_up1 / conv_up2):
// 1044 = 4 9 29, 1064 = 8 7 19, 2088 = 8 9 29, 2128 = 16 7 19.  Kernels: fbm_fast.h.
#define SR_FAST_SIZES(X, CALL) X(1044, CALL) X(1064, CALL) X(2088, CALL) X(2128, CALL)
#define SR_FAST_ENTRY fbm_lowpass_mixed
#include "fbm_fast.h"
