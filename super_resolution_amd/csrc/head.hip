// Bandwidth-bound kernels around the networks' ends:
//   sr_pixel_shuffle_add : SRVGGNetCompact's tail (srvgg_arch.py:65-68): nn.PixelShuffle(r) of the last conv's output plus
//                          the nearest-upsampled input, NHWC -> NCHW.
//   sr_bilinear_resize   : F.interpolate(pred, size, mode='bilinear', align_corners=False) of the owner scripts
//                          (train/train_realESRGAN_fdconv_1.py:115, train/test_realESRGAN_fdconv.py:51) with the Sigmoid head
//                          of CustomRRDBNet (train/train_realESRGAN_fdconv.py:111-115) folded in front and the expm1
//                          de-normalisation of test_realESRGAN_fdconv.py:55-57 folded behind.
//   sr_resize_l1_loss    : the training loss of the same scripts -- bilinear resize of the prediction to the target's size,
//                          then nn.L1Loss (fdconv_1.py:116) or the weighted L1 of train_realESRGAN_fdconv.py:124-126 -- and
//                          its gradient with respect to the prediction.
#include "common.h"

namespace sr {

// torch's area_pixel_compute_source_index for align_corners = False: src = scale * (dst + 0.5) - 0.5, clamped at 0
__device__ __forceinline__ void bilinear_taps(int d, float scale, int n_in, int& i0, int& i1, float& l1) {
    float s = scale * ((float)d + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 > n_in - 1 ? n_in - 1 : i0;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename TS, typename TD>
__global__ void pixel_shuffle_add_kernel(View src, const TD* __restrict__ base, TD* __restrict__ dst, int B, int C, int H, int W,
                                         int r) {
    const int Ho = H * r, Wo = W * r;
    const long n = (long)B * C * Ho * Wo;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int wo = (int)(i % Wo);
        long t = i / Wo;
        const int ho = (int)(t % Ho);
        t /= Ho;
        const int c = (int)(t % C), b = (int)(t / C);
        const int h = ho / r, dy = ho - h * r, w = wo / r, dx = wo - w * r;
        const float v = Elem<TS>::ld((const TS*)src.p + voff(src, b, (long)h * W + w, c * r * r + dy * r + dx));
        const float x = Elem<TD>::ld(base + (((long)b * C + c) * H + h) * W + w);
        Elem<TD>::st(dst + i, v + x);
    }
}

// r = 4, bf16 source whose 8-channel groups are 16-byte loads: one work item per SOURCE pixel.  It reads the pixel's C * 16
// channels once (two 16-byte loads per output channel) and writes, per output channel and output row, the four pixels
// dx = 0..3 as one piece -- neighbouring items write neighbouring pieces, so a wave's store covers 64 * 4 consecutive output
// pixels.  (The element-per-thread kernel above reads 2 bytes per thread from a 128-byte pixel record and touches every
// record 12 times: 1.93 ms for four 1080p frames against 0.45 ms here.)
template <typename TD>
__global__ void pixel_shuffle4_add_kernel(View src, const TD* __restrict__ base, TD* __restrict__ dst, int B, int C, int H, int W) {
    const long n = (long)B * H * W;
    const int Wo = W * 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long t = i / W;
        const int h = (int)(t % H), b = (int)(t / H);
        const long pix = (long)h * W + w;
        for (int c = 0; c < C; ++c) {
            const float x = Elem<TD>::ld(base + (((long)b * C + c) * H + h) * W + w);
            const uint4 lo = *(const uint4*)((const bf16_t*)src.p + voff(src, b, pix, c * 16));
            const uint4 hi = *(const uint4*)((const bf16_t*)src.p + voff(src, b, pix, c * 16 + 8));
            const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};  // channel c*16 + dy*4 + dx: word dy*2 + dx/2
            TD* o = dst + (((long)b * C + c) * (H * 4) + h * 4) * (long)Wo + w * 4;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const float v0 = lo_f(q[2 * dy]) + x, v1 = hi_f(q[2 * dy]) + x, v2 = lo_f(q[2 * dy + 1]) + x, v3 = hi_f(q[2 * dy + 1]) + x;
                if (sizeof(TD) == 2) {
                    *(uint2*)(o + (long)dy * Wo) = make_uint2(pack2(v0, v1), pack2(v2, v3));
                } else {
                    *(float4*)(o + (long)dy * Wo) = make_float4(v0, v1, v2, v3);
                }
            }
        }
    }
}

template <typename TS, typename TD>
__global__ void bilinear_resize_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long planes, int Hs, int Ws, int Hd,
                                       int Wd, int pre_sigmoid, int post_expm1, float post_scale) {
    const long n = planes * Hd * Wd;
    const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wd);
        long t = i / Wd;
        const int y = (int)(t % Hd);
        const long p = t / Hd;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_taps(y, sh, Hs, y0, y1, ly);
        bilinear_taps(x, sw, Ws, x0, x1, lx);
        const TS* s = src + p * Hs * Ws;
        float a = Elem<TS>::ld(s + (long)y0 * Ws + x0), b = Elem<TS>::ld(s + (long)y0 * Ws + x1);
        float c = Elem<TS>::ld(s + (long)y1 * Ws + x0), d = Elem<TS>::ld(s + (long)y1 * Ws + x1);
        if (pre_sigmoid) { a = sigmoid_f(a); b = sigmoid_f(b); c = sigmoid_f(c); d = sigmoid_f(d); }
        float v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
        if (post_expm1) v = fmaxf(expm1f(v * post_scale), 0.f);
        Elem<TD>::st(dst + i, v);
    }
}

// one block reduces its grid-stride share; partial sums go to `partial[blockIdx]` (a second tiny kernel adds them in a
// fixed order: the loss is deterministic).  grad (may be null): d loss / d pred, accumulated with float atomics.
__global__ __launch_bounds__(256) void resize_l1_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        long planes, int Hs, int Ws, int Hd, int Wd, int weighted,
                                                        int pre_sigmoid, float inv_n, float* __restrict__ partial,
                                                        float* __restrict__ grad) {
    __shared__ float red[256];
    const long n = planes * Hd * Wd;
    const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
    float acc = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wd);
        long t = i / Wd;
        const int y = (int)(t % Hd);
        const long p = t / Hd;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_taps(y, sh, Hs, y0, y1, ly);
        bilinear_taps(x, sw, Ws, x0, x1, lx);
        const float* s = pred + p * Hs * Ws;
        float a = s[(long)y0 * Ws + x0], b = s[(long)y0 * Ws + x1], c = s[(long)y1 * Ws + x0], d = s[(long)y1 * Ws + x1];
        if (pre_sigmoid) { a = sigmoid_f(a); b = sigmoid_f(b); c = sigmoid_f(c); d = sigmoid_f(d); }
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        const float v = w00 * a + w01 * b + w10 * c + w11 * d;
        const float tg = target[i];
        const float wt = weighted ? fminf(fmaxf(tg * 10.f, 0.1f), 1.0f) : 1.0f;  // train_realESRGAN_fdconv.py:125
        const float diff = v - tg;
        acc += wt * fabsf(diff);
        if (grad) {
            const float g = wt * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * inv_n;  // torch: sign(0) = 0
            float* gp = grad + p * Hs * Ws;
            const float ga = pre_sigmoid ? a * (1.f - a) : 1.f, gb = pre_sigmoid ? b * (1.f - b) : 1.f;
            const float gc = pre_sigmoid ? c * (1.f - c) : 1.f, gd = pre_sigmoid ? d * (1.f - d) : 1.f;
            atomicAdd(gp + (long)y0 * Ws + x0, g * w00 * ga);
            atomicAdd(gp + (long)y0 * Ws + x1, g * w01 * gb);
            atomicAdd(gp + (long)y1 * Ws + x0, g * w10 * gc);
            atomicAdd(gp + (long)y1 * Ws + x1, g * w11 * gd);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)partial[i];
        *out = (float)(s * (double)scale);
    }
}

// ---- RealESRGANer.enhance's image ends ----------------------------------------------------------------------------
// in:  HWC integer / float image -> three float planes  plane[j] = (float)img[h][w][chmap[j]] / full   (utils.py:197-216:
//      astype(float32), / 255 or / 65535, BGR -> RGB / gray -> RGB / alpha -> 3 equal planes)
template <typename TS>
__global__ void image_to_planes_kernel(const TS* __restrict__ src, int H, int W, int channels, int c0, int c1, int c2, float full,
                                       float* __restrict__ dst) {
    const long n = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const TS* px = src + i * channels;
        dst[i] = __fdiv_rn((float)px[c0], full);
        dst[n + i] = __fdiv_rn((float)px[c1], full);
        dst[2 * n + i] = __fdiv_rn((float)px[c2], full);
    }
}

// out: three float / bf16 planes (R, G, B) -> HWC integer image (utils.py:225-228, 251-254): clamp to [0, 1], RGB -> BGR or
//      BGR2GRAY (0.114 B + 0.587 G + 0.299 R, float32, left to right, no contraction), x full, round half to even, saturate.
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

template <typename TS, typename TD>
__global__ void image_from_planes_kernel(const TS* __restrict__ src, int H, int W, long plane_stride, long row_stride, int gray,
                                         float full, TD* __restrict__ dst, int dst_channels, int dst_c0) {
    const long n = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int h = (int)(i / W), w = (int)(i - (long)h * W);
        const TS* p = src + (long)h * row_stride + w;
        const float r = clamp01(Elem<TS>::ld(p)), g = clamp01(Elem<TS>::ld(p + plane_stride)), b = clamp01(Elem<TS>::ld(p + 2 * plane_stride));
        TD* o = dst + i * dst_channels + dst_c0;
        if (gray) {
            const float y = __fadd_rn(__fadd_rn(__fmul_rn(0.114f, b), __fmul_rn(0.587f, g)), __fmul_rn(0.299f, r));
            o[0] = (TD)rintf(__fmul_rn(y, full));
        } else {
            o[0] = (TD)rintf(__fmul_rn(b, full));
            o[1] = (TD)rintf(__fmul_rn(g, full));
            o[2] = (TD)rintf(__fmul_rn(r, full));
        }
    }
}

// ---- perceptual (VGG-feature L1) loss: the pieces between the 3x3 convolutions ---------------------------------------
// relu + 2x2 max-pool (pool = 1: relu only) on NHWC tensors, and its backward: the gradient goes to the first maximum of
// the window in scan order (torch's max_pool2d) where the pre-activation is positive
template <typename T>
__global__ __launch_bounds__(256) void relu_pool_kernel(int B, int Ho, int Wo, int C, int pool, const T* __restrict__ src, T* __restrict__ dst) {
    const int Hs = Ho * pool, Ws = Wo * pool;
    const long n = (long)B * Ho * Wo * C;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long r = e / C;
        const int w = (int)(r % Wo);
        r /= Wo;
        const int h = (int)(r % Ho), b = (int)(r / Ho);
        float m = 0.f;
        for (int dy = 0; dy < pool; ++dy)
            for (int dx = 0; dx < pool; ++dx)
                m = fmaxf(m, Elem<T>::ld(src + (((long)b * Hs + h * pool + dy) * Ws + w * pool + dx) * C + c));
        Elem<T>::st(dst + e, m);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void relu_pool_bwd_kernel(int B, int Ho, int Wo, int C, int pool, const T* __restrict__ src,
                                                            const T* __restrict__ dy, const T* __restrict__ add, T* __restrict__ dx) {
    const int Hs = Ho * pool, Ws = Wo * pool;
    const long n = (long)B * Ho * Wo * C;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long r = e / C;
        const int w = (int)(r % Wo);
        r /= Wo;
        const int h = (int)(r % Ho), b = (int)(r / Ho);
        const float g = Elem<T>::ld(dy + e);
        float m = -INFINITY;
        int at = 0;
        for (int k = 0; k < pool * pool; ++k) {
            const float v = Elem<T>::ld(src + (((long)b * Hs + h * pool + k / pool) * Ws + w * pool + k % pool) * C + c);
            if (v > m) { m = v; at = k; }
        }
        for (int k = 0; k < pool * pool; ++k) {
            const long o = (((long)b * Hs + h * pool + k / pool) * Ws + w * pool + k % pool) * C + c;
            float v = (k == at && m > 0.f) ? g : 0.f;
            if (add) v += Elem<T>::ld(add + o);   // a tapped layer's own loss gradient
            Elem<T>::st(dx + o, v);
        }
    }
}
// loss += weight * mean |a - b|;  grad (may be null) = weight * sign(a - b) / n
template <typename T>
__global__ __launch_bounds__(256) void l1_feature_kernel(long n, const T* __restrict__ a, const T* __restrict__ b, float weight,
                                                         float* __restrict__ loss, T* __restrict__ grad) {
    __shared__ float red[256];
    float acc = 0.f;
    const float gs = weight / (float)n;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float d = Elem<T>::ld(a + e) - Elem<T>::ld(b + e);
        acc += fabsf(d);
        if (grad) Elem<T>::st(grad + e, d > 0.f ? gs : (d < 0.f ? -gs : 0.f));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss, red[0] * gs);
}
// (x - mean[c]) / std[c] of a (B, Cs, H, W) float image, Cs = 1 (grey, repeated) or 3, into NHWC 3 channels; and its adjoint
template <typename T>
__global__ __launch_bounds__(256) void vgg_prep_kernel(int B, int Cs, long npix, const float* __restrict__ src, T* __restrict__ dst,
                                                       float m0, float m1, float m2, float s0, float s1, float s2, int norm) {
    const long n = (long)B * npix;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long b = e / npix, p = e - b * npix;
        const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
        for (int c = 0; c < 3; ++c) {
            float v = src[(b * Cs + (Cs == 1 ? 0 : c)) * npix + p];
            if (norm) v = (v - mean[c]) / sd[c];
            Elem<T>::st(dst + e * 3 + c, v);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void vgg_prep_bwd_kernel(int B, int Cs, long npix, const T* __restrict__ dy, float* __restrict__ dsrc,
                                                           float s0, float s1, float s2, int norm) {
    const long n = (long)B * npix;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long b = e / npix, p = e - b * npix;
        const float sd[3] = {s0, s1, s2};
        float g[3];
        for (int c = 0; c < 3; ++c) g[c] = Elem<T>::ld(dy + e * 3 + c) / (norm ? sd[c] : 1.0f);
        if (Cs == 1) dsrc[b * npix + p] = g[0] + g[1] + g[2];
        else
            for (int c = 0; c < 3; ++c) dsrc[(b * 3 + c) * npix + p] = g[c];
    }
}
// adjoint of bilinear_resize_kernel (float32): dsrc (zeroed by the caller) += weights * ddst
__global__ __launch_bounds__(256) void bilinear_resize_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc, long planes, int Hs,
                                                                  int Ws, int Hd, int Wd) {
    const long n = planes * Hd * Wd;
    const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % Wd);
        long t = i / Wd;
        const int y = (int)(t % Hd);
        const long p = t / Hd;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_taps(y, sh, Hs, y0, y1, ly);
        bilinear_taps(x, sw, Ws, x0, x1, lx);
        const float g = ddst[i];
        float* gp = dsrc + p * Hs * Ws;
        atomicAdd(gp + (long)y0 * Ws + x0, g * (1.f - ly) * (1.f - lx));
        atomicAdd(gp + (long)y0 * Ws + x1, g * (1.f - ly) * lx);
        atomicAdd(gp + (long)y1 * Ws + x0, g * ly * (1.f - lx));
        atomicAdd(gp + (long)y1 * Ws + x1, g * ly * lx);
    }
}

static int head_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}
}  // namespace sr

using namespace sr;

extern "C" int sr_pixel_shuffle_add(sr_view src, int src_dtype, const void* base, void* dst, int io_dtype, int B, int C,
                                    int H, int W, int r, void* stream) {
    SR_REQUIRE(src.p && base && dst && dtype_ok(src_dtype) && dtype_ok(io_dtype) && B > 0 && C > 0 && H > 0 && W > 0 && r >= 1);
    SR_REQUIRE(src.C >= C * r * r && src.c0 >= 0 && view_block_ok(src));
    const View v = to_view(src, (long)H * W);
    hipStream_t st = (hipStream_t)stream;
    if (r == 4 && src_dtype == SR_BF16 && view_vec_ok(v, 8) && v.C >= C * 16 && (uintptr_t)dst % 16 == 0) {
        const int g4 = head_grid((long)B * H * W);
        if (io_dtype == SR_F32) pixel_shuffle4_add_kernel<float><<<g4, 256, 0, st>>>(v, (const float*)base, (float*)dst, B, C, H, W);
        else pixel_shuffle4_add_kernel<bf16_t><<<g4, 256, 0, st>>>(v, (const bf16_t*)base, (bf16_t*)dst, B, C, H, W);
        SR_LAUNCH_CHECK();
        return SR_OK;
    }
    const int g = head_grid((long)B * C * H * W * r * r);
#define SR_PS(TS, TD) pixel_shuffle_add_kernel<TS, TD><<<g, 256, 0, st>>>(v, (const TD*)base, (TD*)dst, B, C, H, W, r)
    if (src_dtype == SR_F32 && io_dtype == SR_F32) SR_PS(float, float);
    else if (src_dtype == SR_F32) SR_PS(float, bf16_t);
    else if (io_dtype == SR_F32) SR_PS(bf16_t, float);
    else SR_PS(bf16_t, bf16_t);
#undef SR_PS
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_bilinear_resize(const void* src, int src_dtype, void* dst, int dst_dtype, long planes, int Hs, int Ws, int Hd,
                                  int Wd, int pre_sigmoid, int post_expm1, float post_scale, void* stream) {
    SR_REQUIRE(src && dst && dtype_ok(src_dtype) && dtype_ok(dst_dtype) && planes > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0);
    hipStream_t st = (hipStream_t)stream;
    const int g = head_grid(planes * Hd * Wd);
#define SR_RS(TS, TD) \
    bilinear_resize_kernel<TS, TD><<<g, 256, 0, st>>>((const TS*)src, (TD*)dst, planes, Hs, Ws, Hd, Wd, pre_sigmoid, post_expm1, post_scale)
    if (src_dtype == SR_F32 && dst_dtype == SR_F32) SR_RS(float, float);
    else if (src_dtype == SR_F32) SR_RS(float, bf16_t);
    else if (dst_dtype == SR_F32) SR_RS(bf16_t, float);
    else SR_RS(bf16_t, bf16_t);
#undef SR_RS
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" size_t sr_resize_l1_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int sr_resize_l1_loss(const float* pred, const float* target, long planes, int Hs, int Ws, int Hd, int Wd, int weighted,
                                 int pre_sigmoid, float* loss, float* grad_pred, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    SR_REQUIRE(pred && target && loss && workspace && planes > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0);
    if (workspace_bytes < sr_resize_l1_workspace_bytes()) return SR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long n = planes * Hd * Wd;
    int g = head_grid(n);
    if (g > 1024) g = 1024;
    if (grad_pred) SR_HIP(hipMemsetAsync(grad_pred, 0, (size_t)planes * Hs * Ws * sizeof(float), st));
    resize_l1_kernel<<<g, 256, 0, st>>>(pred, target, planes, Hs, Ws, Hd, Wd, weighted, pre_sigmoid, 1.0f / (float)n,
                                        (float*)workspace, grad_pred);
    SR_LAUNCH_CHECK();
    sum_partials_kernel<<<1, 64, 0, st>>>((const float*)workspace, g, 1.0f / (float)n, loss);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_image_to_planes(const void* src, int src_kind, int H, int W, int channels, int c0, int c1, int c2, float full,
                                  float* dst, void* stream) {
    SR_REQUIRE(src && dst && H > 0 && W > 0 && channels >= 1 && channels <= 4 && full > 0.f);
    SR_REQUIRE(c0 >= 0 && c0 < channels && c1 >= 0 && c1 < channels && c2 >= 0 && c2 < channels);
    hipStream_t st = (hipStream_t)stream;
    const int g = head_grid((long)H * W);
    if (src_kind == 0) image_to_planes_kernel<uint8_t><<<g, 256, 0, st>>>((const uint8_t*)src, H, W, channels, c0, c1, c2, full, dst);
    else if (src_kind == 1) image_to_planes_kernel<uint16_t><<<g, 256, 0, st>>>((const uint16_t*)src, H, W, channels, c0, c1, c2, full, dst);
    else if (src_kind == 2) image_to_planes_kernel<float><<<g, 256, 0, st>>>((const float*)src, H, W, channels, c0, c1, c2, full, dst);
    else return SR_ERR_ARG;
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_image_from_planes(const void* src, int src_dtype, int H, int W, long plane_stride, long row_stride, int gray,
                                    int sixteen, void* dst, int dst_channels, int dst_c0, void* stream) {
    SR_REQUIRE(src && dst && dtype_ok(src_dtype) && H > 0 && W > 0 && row_stride >= W && plane_stride >= (long)(H - 1) * row_stride + W);
    SR_REQUIRE(dst_channels >= 1 && dst_channels <= 4 && dst_c0 >= 0 && dst_c0 + (gray ? 1 : 3) <= dst_channels);
    hipStream_t st = (hipStream_t)stream;
    const int g = head_grid((long)H * W);
    const float full = sixteen ? 65535.f : 255.f;
#define SR_IM(TS, TD) \
    image_from_planes_kernel<TS, TD><<<g, 256, 0, st>>>((const TS*)src, H, W, plane_stride, row_stride, gray, full, (TD*)dst, dst_channels, dst_c0)
    if (src_dtype == SR_F32 && sixteen) SR_IM(float, uint16_t);
    else if (src_dtype == SR_F32) SR_IM(float, uint8_t);
    else if (sixteen) SR_IM(bf16_t, uint16_t);
    else SR_IM(bf16_t, uint8_t);
#undef SR_IM
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_relu_pool(int dtype, int B, int Ho, int Wo, int C, int pool, const void* src, void* dst, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && Ho > 0 && Wo > 0 && C > 0 && (pool == 1 || pool == 2) && src && dst);
    hipStream_t st = (hipStream_t)stream;
    const int g = head_grid((long)B * Ho * Wo * C);
    if (dtype == SR_F32) relu_pool_kernel<float><<<g, 256, 0, st>>>(B, Ho, Wo, C, pool, (const float*)src, (float*)dst);
    else relu_pool_kernel<bf16_t><<<g, 256, 0, st>>>(B, Ho, Wo, C, pool, (const bf16_t*)src, (bf16_t*)dst);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_relu_pool_backward(int dtype, int B, int Ho, int Wo, int C, int pool, const void* src, const void* dy, const void* add,
                                     void* dx, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && Ho > 0 && Wo > 0 && C > 0 && (pool == 1 || pool == 2) && src && dy && dx);
    hipStream_t st = (hipStream_t)stream;
    const int g = head_grid((long)B * Ho * Wo * C);
    if (dtype == SR_F32)
        relu_pool_bwd_kernel<float><<<g, 256, 0, st>>>(B, Ho, Wo, C, pool, (const float*)src, (const float*)dy, (const float*)add, (float*)dx);
    else
        relu_pool_bwd_kernel<bf16_t><<<g, 256, 0, st>>>(B, Ho, Wo, C, pool, (const bf16_t*)src, (const bf16_t*)dy, (const bf16_t*)add, (bf16_t*)dx);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_l1_feature_loss(int dtype, long n, const void* a, const void* b, float weight, float* loss, void* grad_a, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && n > 0 && a && b && loss);
    hipStream_t st = (hipStream_t)stream;
    int g = head_grid(n);
    if (g > 1024) g = 1024;
    if (dtype == SR_F32) l1_feature_kernel<float><<<g, 256, 0, st>>>(n, (const float*)a, (const float*)b, weight, loss, (float*)grad_a);
    else l1_feature_kernel<bf16_t><<<g, 256, 0, st>>>(n, (const bf16_t*)a, (const bf16_t*)b, weight, loss, (bf16_t*)grad_a);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_vgg_prep(int dtype, int B, int Cs, int H, int W, const float* src, void* dst, const float* mean3, const float* std3,
                           int adjoint, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && (Cs == 1 || Cs == 3) && H > 0 && W > 0 && src && dst);
    SR_REQUIRE((mean3 == nullptr) == (std3 == nullptr));
    hipStream_t st = (hipStream_t)stream;
    const long npix = (long)H * W;
    const int g = head_grid((long)B * npix);
    const int norm = mean3 != nullptr;
    const float m0 = norm ? mean3[0] : 0.f, m1 = norm ? mean3[1] : 0.f, m2 = norm ? mean3[2] : 0.f;
    const float s0 = norm ? std3[0] : 1.f, s1 = norm ? std3[1] : 1.f, s2 = norm ? std3[2] : 1.f;
    if (!adjoint) {  // src: float image (B,Cs,H,W) -> dst: (B,H,W,3) in dtype
        if (dtype == SR_F32) vgg_prep_kernel<float><<<g, 256, 0, st>>>(B, Cs, npix, src, (float*)dst, m0, m1, m2, s0, s1, s2, norm);
        else vgg_prep_kernel<bf16_t><<<g, 256, 0, st>>>(B, Cs, npix, src, (bf16_t*)dst, m0, m1, m2, s0, s1, s2, norm);
    } else {         // dst: gradient (B,H,W,3) in dtype -> src (written): float gradient (B,Cs,H,W)
        if (dtype == SR_F32) vgg_prep_bwd_kernel<float><<<g, 256, 0, st>>>(B, Cs, npix, (const float*)dst, const_cast<float*>(src), s0, s1, s2, norm);
        else vgg_prep_bwd_kernel<bf16_t><<<g, 256, 0, st>>>(B, Cs, npix, (const bf16_t*)dst, const_cast<float*>(src), s0, s1, s2, norm);
    }
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_bilinear_resize_backward(const float* ddst, float* dsrc, long planes, int Hs, int Ws, int Hd, int Wd, void* stream) {
    SR_REQUIRE(ddst && dsrc && planes > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0);
    hipStream_t st = (hipStream_t)stream;
    SR_HIP(hipMemsetAsync(dsrc, 0, (size_t)planes * Hs * Ws * sizeof(float), st));
    bilinear_resize_bwd_kernel<<<head_grid(planes * Hd * Wd), 256, 0, st>>>(ddst, dsrc, planes, Hs, Ws, Hd, Wd);
    SR_LAUNCH_CHECK();
    return SR_OK;
}
