// FBM low-pass planes, register-blocked FFT: the power-of-two sizes (64 .. 2048).  Kernels: fbm_fast.h.
#define SR_FAST_SIZES(X, CALL) X(64, CALL) X(128, CALL) X(256, CALL) X(512, CALL) X(1024, CALL) X(2048, CALL)
#define SR_FAST_ENTRY fbm_lowpass_pow2
#define SR_FAST_DEFINE_GLOBALS 1
#include "fbm_fast.h"
