// Shared device/host helpers for libsr_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sr_amd.h"

namespace sr {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return SR_ERR_HIP;
}
#define SR_HIP(call)                                   \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) return sr::hip_fail(_e); \
    } while (0)
#define SR_LAUNCH_CHECK() SR_HIP(hipGetLastError())
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device, raised when a launch asks for more: the call
// costs microseconds of host time and a forward issues thousands of launches (small inputs are launch-bound)
#define SR_LDS_ATTR(bytes, ...)                                           \
    do {                                                                  \
        static sr::LdsAttr attr_;                                         \
        SR_HIP(sr::lds_attr(attr_, (const void*)(__VA_ARGS__), (size_t)(bytes))); \
    } while (0)
#define SR_REQUIRE(cond)            \
    do {                            \
        if (!(cond)) return SR_ERR_ARG; \
    } while (0)

struct LdsAttr {
    int set[16] = {};
};
inline hipError_t lds_attr(LdsAttr& a, const void* fn, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int& have = a.set[dev & 15];
    if (have && have >= (int)bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = (int)bytes > 0 ? (int)bytes : 1;
    return e;
}

// compute units of the current device, cached per device index (a process may drive several GPUs)
inline int device_cus(int* out) {
    static int cus[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e);
    int& c = cus[dev & 15];
    if (!c) {
        int v = 0;
        e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return hip_fail(e);
        c = v;
    }
    *out = c;
    return SR_OK;
}
inline int current_device_slot() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev & 15;
}

typedef uint16_t bf16_t;  // raw bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN (the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---- packed bf16 pairs (two channels per 32-bit word) ----
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// g1*x + (g2-g1)*L2 + (g3-g2)*L4 + (1-g3)*L8 on 8 bf16 channels: channel pairs as float2, one v_pk_mul_f32 + three
// v_pk_fma_f32 per pair (written as four products and three adds the compiler used ten instructions per pair)
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v unpack2(uint32_t u) {
    float2v r;
    r.x = lo_f(u);
    r.y = hi_f(u);
    return r;
}
__device__ __forceinline__ uint32_t modulate2(uint32_t x, uint32_t a, uint32_t b, uint32_t c, float2v c0, float2v c1, float2v c2,
                                              float2v c3) {
    float2v r = c0 * unpack2(x);
    r = __builtin_elementwise_fma(c1, unpack2(a), r);
    r = __builtin_elementwise_fma(c2, unpack2(b), r);
    r = __builtin_elementwise_fma(c3, unpack2(c), r);
    return pack2(r.x, r.y);
}
// fp16 operands (DESIGN.md finding 39): the same mix packed to fp16 (v_cvt_pk_f16_f32, round to nearest even, saturating)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
    // saturating: an activation beyond fp16's range becomes +-65504, not inf (bf16 operands have float32's range)
    a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
    const half2v h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
template <bool F16>
__device__ __forceinline__ uint32_t modulate2t(uint32_t x, uint32_t a, uint32_t b, uint32_t c, float2v c0, float2v c1, float2v c2,
                                               float2v c3) {
    float2v r = c0 * unpack2(x);
    r = __builtin_elementwise_fma(c1, unpack2(a), r);
    r = __builtin_elementwise_fma(c2, unpack2(b), r);
    r = __builtin_elementwise_fma(c3, unpack2(c), r);
    return F16 ? pack2h(r.x, r.y) : pack2(r.x, r.y);
}
template <bool F16>
__device__ __forceinline__ uint4 modulate8t(uint4 x, uint4 a, uint4 b, uint4 c, float c0, float c1, float c2, float c3) {
    const float2v k0 = {c0, c0}, k1 = {c1, c1}, k2 = {c2, c2}, k3 = {c3, c3};
    uint4 r;
    r.x = modulate2t<F16>(x.x, a.x, b.x, c.x, k0, k1, k2, k3);
    r.y = modulate2t<F16>(x.y, a.y, b.y, c.y, k0, k1, k2, k3);
    r.z = modulate2t<F16>(x.z, a.z, b.z, c.z, k0, k1, k2, k3);
    r.w = modulate2t<F16>(x.w, a.w, b.w, c.w, k0, k1, k2, k3);
    return r;
}
__device__ __forceinline__ uint4 modulate8(uint4 x, uint4 a, uint4 b, uint4 c, float c0, float c1, float c2, float c3) {
    const float2v k0 = {c0, c0}, k1 = {c1, c1}, k2 = {c2, c2}, k3 = {c3, c3};
    uint4 r;
    r.x = modulate2(x.x, a.x, b.x, c.x, k0, k1, k2, k3);
    r.y = modulate2(x.y, a.y, b.y, c.y, k0, k1, k2, k3);
    r.z = modulate2(x.z, a.z, b.z, c.z, k0, k1, k2, k3);
    r.w = modulate2(x.w, a.w, b.w, c.w, k0, k1, k2, k3);
    return r;
}

inline size_t dtype_size(int dtype) { return dtype == SR_F32 ? 4 : 2; }
inline bool dtype_ok(int dtype) { return dtype == SR_F32 || dtype == SR_BF16; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// device-side mirrors of the ABI structs (plain data, passed by value as kernel arguments)
// Channel-blocked tensors keep 2^csh channels per block ([B][C >> csh][pixels][1 << csh], ld == 1 << csh);
// NHWC is the degenerate case csh = 3, cbs = 8 (block term + in-block term add up to the channel index).
struct View {
    char* p;
    int ld, c0, C, cbs, csh;
    long bs;
};
// element offset of channel c of pixel `pix` (h*W + w of the view's own resolution) of sample b
__host__ __device__ __forceinline__ long voff(const View& v, int b, long pix, int c) {
    const int cc = v.c0 + c;
    return (long)b * v.bs + (long)(cc >> v.csh) * v.cbs + pix * v.ld + (cc & ((1 << v.csh) - 1));
}
// same without the sample term, in 32 bits (a sample is < 2^31 elements; checked on the host)
__device__ __forceinline__ unsigned voff32(const View& v, unsigned pix, int c) {
    const unsigned cc = (unsigned)(v.c0 + c);
    return (cc >> v.csh) * (unsigned)v.cbs + pix * (unsigned)v.ld + (cc & ((1u << v.csh) - 1u));
}
// npix: pixels per sample of the viewed tensor (fills the NHWC defaults)
// a blocked sr_view (cbs != 0) keeps ld channels per block; ld must then be a power of two >= 8 (view_block_ok)
inline bool view_block_ok(const sr_view& v) { return !v.cbs || (v.ld >= 8 && (v.ld & (v.ld - 1)) == 0); }
inline View to_view(const sr_view& v, long npix) {
    int csh = 3;
    if (v.cbs)
        while ((1 << csh) < v.ld) ++csh;
    return View{(char*)v.p, v.ld, v.c0, v.C, v.cbs ? v.cbs : 8, csh, v.bs ? (long)v.bs : npix * v.ld};
}
inline View nhwc_view(void* p, int C, long npix) { return View{(char*)p, C, 0, C, 8, 3, npix * C}; }
// cb = channels per block (8 or 32)
inline View blocked_view(void* p, int C, long npix, int cb = 8) {
    int csh = 3;
    while ((1 << csh) < cb) ++csh;
    return View{(char*)p, cb, 0, C, (int)(cb * npix), csh, npix * C};
}
// 16-byte vector access to 8-channel groups / 8-byte access to 4-channel groups is legal
inline bool view_vec_ok(const View& v, int mult) {
    return v.p && ((uintptr_t)v.p % 16 == 0) && v.ld % mult == 0 && v.c0 % mult == 0 && v.cbs % 8 == 0 && v.bs % 8 == 0;
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Map launch index L of T so that every XCD
// walks one contiguous range of the logical index space: neighbouring tiles (shared halos) then meet in one L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned L, unsigned T) {
    const unsigned q = T >> 3, r = T & 7u, x = L & 7u, i = L >> 3;
    return x * q + (x < r ? x : r) + i;
}

struct Concat {
    int nseg, bands, ups, ctot;
    View x[SR_MAX_SEG], l2[SR_MAX_SEG], l4[SR_MAX_SEG], l8[SR_MAX_SEG];
    int cstart[SR_MAX_SEG + 1];  // first concat channel of each segment
};
int to_concat(const sr_concat* in, long npix_x, long npix_bands, Concat* out);  // validates; SR_OK / SR_ERR_ARG

struct Epilogue {
    const float* bias;
    const float* slope;  // per-channel negative slopes (PReLU) or nullptr
    int lrelu;
    View r1, r2;
    float s1, s2;
};

// conv_plain.hip: shared-weight, band-free bf16 layers whose weights fit LDS (SR_NOT_ELIGIBLE: conv.hip's kernels take it)
int conv3x3_plain_launch(int B, int H, int W, const Concat& in, const void* wts, long wbs, long wlo, int cout, const Epilogue& ep,
                         View out, hipStream_t st);
inline Epilogue to_epilogue(const sr_epilogue* e, long npix) {
    Epilogue o{};
    if (e) {
        o.bias = e->bias;
        o.lrelu = e->lrelu;
        o.slope = e->slope;
        o.r1 = to_view(e->r1, npix);
        o.r2 = to_view(e->r2, npix);
        o.s1 = e->s1;
        o.s2 = e->s2;
    }
    return o;
}

// internal launchers shared with the network executor (same semantics as the extern "C" entry points)
// w_lo_off != 0 (bf16 MFMA kernel only): weights + w_lo_off holds the low-order bf16 parts of the same weights
// w_lo_off == SR_WLO_F16: fp16 weights + fp16 operand (see below); only where conv3x3_f16_ok() holds
int conv3x3_launch(int dtype, int B, int H, int W, const Concat& in, const float* gates, const void* weights,
                   long w_bstride, long w_lo_off, int cout, const Epilogue& ep, View out, hipStream_t st);
bool conv3x3_f16_ok(int dtype, const Concat& in, const void* weights, long w_bstride, int cout, const Epilogue& ep, View out);

// bmask: bit k set = band k (l2 / l4 / l8) is wanted; views of unwanted bands are never written (callers pass a copy of a
// wanted one so that the layout checks see a valid view)
int fbm_lowpass_launch(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8,
                       float* pooled, int pooled_ld, void* ws, size_t ws_bytes, hipStream_t st, int bmask = 7);
// power-of-two fast path; returns SR_NOT_ELIGIBLE when the shape / alignment is not covered
#define SR_NOT_ELIGIBLE (-100)
int fbm_lowpass_pow2(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8, float* pooled,
                     int pooled_ld, void* ws, hipStream_t st, int bmask = 7);
// the same kernels for the mixed-radix sides 1044 / 1064 / 2088 / 2128 (high-resolution tensors of tiles at tile 512, pad 10)
int fbm_lowpass_mixed(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8, float* pooled,
                      int pooled_ld, void* ws, hipStream_t st, int bmask = 7);
// bf16 only: the three passes as real-basis DFT GEMMs on the matrix cores (SR_NOT_ELIGIBLE otherwise)
int fbm_lowpass_mfma(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8, float* pooled,
                     int pooled_ld, void* ws, hipStream_t st, int bmask = 7);
int fbm_gates_launch(int dtype, int B, int H, int W, const Concat& in, const sr_layer_params& lp, float* gates,
                     hipStream_t st);
// consumers of sr_fbm_gates_accumulate (device-side mirror of sr_gate_consumer)
struct GateCons {
    const float* w[3];
    const float* b[3];
    float* logits;
    float* gates;
    int first;
};
struct GateConsList {
    int n;
    GateCons c[SR_MAX_GATE_CONSUMERS];
};

int fbm_gates_acc_launch(int B, int H, int W, View seg, const GateConsList& cl, hipStream_t st);  // bf16; SR_NOT_ELIGIBLE
// mods: weightgen_ws_bytes(B, cin, cout) of scratch (the modulation vectors ca | ln | fa | sa | ka per sample stay there);
// train: BatchNorm on batch statistics, lp.bn_rm / lp.bn_rv updated in place with `momentum`
size_t weightgen_ws_bytes(int B, int cin, int cout);
// wdyn_lo == SR_WDYN_F16 (bf16 only): wdyn receives ONE fp16 value per weight instead of a bf16 hi (+ lo) pair, as [B][10][cout][cin]:
// plane 9 holds the 9-tap sums of the rounding remainders (needs 10/9 of the single-weight buffer: the split layout's 2x covers it)
#define SR_WDYN_F16 ((void*)(intptr_t)-1)
#define SR_WLO_F16 (-1L)   // conv3x3_launch: w_lo_offset value that says "weights are fp16, pack the banded operand to fp16"
int weightgen_launch(int dtype, int B, const float* pooled, int pooled_ld, const sr_layer_params& lp, int train, float momentum,
                     void* wdyn, void* wdyn_lo, float* mods, hipStream_t st);

}  // namespace sr
