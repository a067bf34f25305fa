// Backward kernels of the FDConv / conv layers (training step of train/train_realESRGAN_fdconv_1.py:113-118: the reference
// gets these from torch autograd through FDConv.py:640-722).  Every tensor is NHWC in `dtype`; sums run in float32.
//
//   y[b,p,o] = bias[o] + sum_{t,i} W_b[t][o][i] * a[b, p + off(t), i],   a = FBM(x) = g1 x + (g2-g1) L2 + (g3-g2) L4 + (1-g3) L8
//
//   sr_conv_weight_transpose : W^T with mirrored taps, the operand with which sr_conv3x3 computes da = conv(dy, W^T)
//   sr_conv3x3_wgrad         : dW_b[t][o][i] = sum_p dy[b,p,o] a[b,p+off(t),i]  (a formed on the fly, as the forward loader
//                              does), dbias[o] = sum dy
//   sr_fbm_backward          : da -> the four linear branches (dx, dL2, dL4, dL8) and the gate-logit gradients
//   sr_fbm_gates_backward    : gate-conv weight / bias gradients and the assembly of the layer's input gradient
//                              (direct branch + the three low-passed band branches + transposed gate conv + pooled mean)
// The box low-pass is self-adjoint (a real symmetric mask between a unitary transform pair), so the band branches go
// back through sr_fbm_lowpass itself.
#include <cstdlib>

#include "common.h"

namespace sr {

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void weight_transpose_kernel(const T* __restrict__ w, T* __restrict__ wt, int B, int cout, int cin) {
    const long n = (long)B * 9 * cout * cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // destination index: [b][t'][i][o]
        const int o = (int)(i % cout);
        long r = i / cout;
        const int ci = (int)(r % cin);
        r /= cin;
        const int tp = (int)(r % 9);
        const long b = r / 9;
        wt[i] = w[((b * 9 + (8 - tp)) * cout + o) * cin + ci];
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  Workgroup: a 32 x 32 (o, i) block of all 9 taps for a run of 4-row x 32-column pixel strips of one
// sample; thread (to, ti) keeps a 2 x 2 (o, i) patch x 9 taps in registers.  The dy strip and the haloed operand strip
// are staged in LDS (reads are broadcasts: 16 lanes share a dy address, 4 share an operand address).
constexpr int WG_RS = 4, WG_WC = 32;

template <typename T>
__device__ __forceinline__ float operand_at(const Concat& in, const float* __restrict__ gates, int b, int H, int W, int h, int w,
                                            int ci) {
    if (h < 0 || h >= H || w < 0 || w >= W || ci >= in.ctot) return 0.f;
    int s = 0;
    while (s + 1 < in.nseg && ci >= in.cstart[s + 1]) ++s;
    const int c = ci - in.cstart[s];
    const int Ws = W >> in.ups;
    float v = Elem<T>::ld((const T*)in.x[s].p + voff(in.x[s], b, (long)(h >> in.ups) * Ws + (w >> in.ups), c));
    if (in.bands) {
        const long p = (long)h * W + w;
        const float4 g = ((const float4*)gates)[(long)b * H * W + p];
        const float v2 = Elem<T>::ld((const T*)in.l2[s].p + voff(in.l2[s], b, p, c));
        const float v4 = Elem<T>::ld((const T*)in.l4[s].p + voff(in.l4[s], b, p, c));
        const float v8 = Elem<T>::ld((const T*)in.l8[s].p + voff(in.l8[s], b, p, c));
        v = g.x * v + (g.y - g.x) * v2 + (g.z - g.y) * v4 + (1.0f - g.z) * v8;
    }
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(Concat in, int H, int W, const float* __restrict__ gates, View dy, int cout,
                                                    float* __restrict__ dw, long dw_bstride, int strips_per_block) {
    __shared__ __attribute__((aligned(16))) float a_sh[WG_RS + 2][WG_WC + 2][32];
    __shared__ __attribute__((aligned(16))) float d_sh[WG_RS][WG_WC][32];
    const int cin = in.ctot;
    const int itiles = (cin + 31) / 32;
    const int i0 = (blockIdx.x % itiles) * 32, o0 = (blockIdx.x / itiles) * 32;
    const int b = blockIdx.z;
    const int sx = (W + WG_WC - 1) / WG_WC, nstrips = sx * ((H + WG_RS - 1) / WG_RS);
    const int tid = threadIdx.x, to = tid >> 4, ti = tid & 15;
    float acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t][0][0] = acc[t][0][1] = acc[t][1][0] = acc[t][1][1] = 0.f;
    const int s_begin = blockIdx.y * strips_per_block;
    const int s_end = min(nstrips, s_begin + strips_per_block);
    for (int st = s_begin; st < s_end; ++st) {
        const int h0 = (st / sx) * WG_RS, w0 = (st % sx) * WG_WC;
        __syncthreads();
        for (int it = tid; it < (WG_RS + 2) * (WG_WC + 2) * 32; it += 256) {
            const int c = it & 31, pi = it >> 5;
            const int py = pi / (WG_WC + 2), px = pi - py * (WG_WC + 2);
            a_sh[py][px][c] = operand_at<T>(in, gates, b, H, W, h0 + py - 1, w0 + px - 1, i0 + c);
        }
        for (int it = tid; it < WG_RS * WG_WC * 32; it += 256) {
            const int c = it & 31, pi = it >> 5;
            const int py = pi / WG_WC, px = pi - py * WG_WC;
            const int h = h0 + py, w = w0 + px;
            float v = 0.f;
            if (h < H && w < W && o0 + c < cout) v = Elem<T>::ld((const T*)dy.p + voff(dy, b, (long)h * W + w, o0 + c));
            d_sh[py][px][c] = v;
        }
        __syncthreads();
        for (int py = 0; py < WG_RS; ++py)
            for (int px = 0; px < WG_WC; ++px) {
                const float2 d = *(const float2*)&d_sh[py][px][2 * to];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float2 a = *(const float2*)&a_sh[py + t / 3][px + t % 3][2 * ti];
                    acc[t][0][0] += d.x * a.x;
                    acc[t][0][1] += d.x * a.y;
                    acc[t][1][0] += d.y * a.x;
                    acc[t][1][1] += d.y * a.y;
                }
            }
    }
    float* out = dw + (long)b * dw_bstride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int oo = 0; oo < 2; ++oo)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int o = o0 + 2 * to + oo, ci = i0 + 2 * ti + ii;
                if (o < cout && ci < cin) atomicAdd(out + ((long)t * cout + o) * cin + ci, acc[t][oo][ii]);
            }
}

// ---------------------------------------------------------------------------------------------
// weight gradient on the matrix cores (bf16 tensors, Cout = 32 or 64, 32-channel input chunks inside one segment).
// dW[t][o][i] = sum_p dy[p][o] a[p + off(t)][i] is a GEMM whose contraction runs over PIXELS, while both tensors keep
// channels innermost: the dy strip and the haloed operand strip are staged pixel-major in LDS exactly as they lie in
// memory (16-byte stores, the operand band-mixed on the way as in the forward loader) and ds_read_b64_tr_b16 hands every
// lane the 8 pixel values of its channel -- the MFMA fragment layout of BOTH operands, no software transpose.  A tap is
// a pixel offset of the operand fragment's base address.
//   workgroup : sample b, 32-channel chunk i0, a run of 4 x 32-pixel strips (8 k-steps of 16 pixels each)
//   wave      : Cout = 32: strip row = wave, all 9 taps (9 accumulator tiles);
//               Cout = 64: output-channel half = wave & 1, strip rows (wave >> 1) and (wave >> 1) + 2
//   epilogue  : the waves' partial tiles are summed through LDS and added to dW with one float atomic per element.
constexpr int WM_RS = 4, WM_WC = 32, WM_AW = WM_WC + 2;
constexpr int WM_A_BYTES = (WM_RS + 2) * WM_AW * 64;  // haloed operand strip, 64 B (32 channels) per pixel
constexpr int WM_D_BYTES = WM_RS * WM_WC * 64;        // dy strip per 32-channel half
constexpr int WM_RED_BYTES = 9 * 16 * 64 * 4;         // one wave's accumulators
typedef __attribute__((ext_vector_type(8))) short wm_short8;
typedef __attribute__((ext_vector_type(16))) float wm_float16;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short wm_s4;

// fragment (32 channels x 16 pixels from pixel `pix0` of a 64-byte-per-pixel image): lane l gets channel l % 32, pixels
// 8 (l / 32) .. + 7.  Lane 4q + p of a 16-lane group supplies pixel row q, channels 4p .. 4p + 3 of the group's 16.
__device__ __forceinline__ wm_short8 wm_frag(const char* img, int pix0, int lane) {
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int off = (pix0 + 8 * (g4 >> 1) + q) * 64 + 32 * (g4 & 1) + 8 * p;
    const wm_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wm_s4*)(img + off));
    const wm_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wm_s4*)(img + off + 4 * 64));
    wm_short8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

extern __shared__ __attribute__((aligned(16))) char wm_smem[];

struct WmSeg {  // the segment that holds this launch's channel chunks (plain NHWC views, 16-byte aligned)
    const bf16_t *x, *l2, *l4, *l8;
    long x_bs, b_bs;   // sample strides (elements) of x and of the band planes
    int x_ld, b_ld;    // pixel strides
    int c0, nchunk;    // first concat channel of the segment, 32-channel chunks in it
};

template <int MT>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WmSeg sg, int bands, int ups, int H, int W, int cin,
                                                         const float* __restrict__ gates, const bf16_t* __restrict__ dy,
                                                         long dy_bs, int dy_ld, float* __restrict__ dw, long dw_bstride,
                                                         int strips_per_block) {
    constexpr int COUT = 32 * MT;
    char* a_sh = wm_smem;
    char* d_sh = wm_smem + WM_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk_i = blockIdx.x;          // 32-channel chunk inside the segment
    const int b = blockIdx.z;
    const int sx = (W + WM_WC - 1) / WM_WC, nstrips = sx * ((H + WM_RS - 1) / WM_RS);
    const int s_begin = blockIdx.y * strips_per_block;
    const int s_end = min(nstrips, s_begin + strips_per_block);
    const int Ws = W >> ups;
    const bf16_t* xb = sg.x + (long)b * sg.x_bs + chunk_i * 32;
    const bf16_t* l2b = bands ? sg.l2 + (long)b * sg.b_bs + chunk_i * 32 : nullptr;
    const bf16_t* l4b = bands ? sg.l4 + (long)b * sg.b_bs + chunk_i * 32 : nullptr;
    const bf16_t* l8b = bands ? sg.l8 + (long)b * sg.b_bs + chunk_i * 32 : nullptr;
    const float4* gb = bands ? (const float4*)gates + (long)b * H * W : nullptr;
    const bf16_t* dyb = dy + (long)b * dy_bs;
    wm_float16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int mb = MT == 2 ? (wave & 1) : 0;
    const char* d_img = d_sh + mb * WM_D_BYTES;
    for (int st = s_begin; st < s_end; ++st) {
        const int h0 = (st / sx) * WM_RS, w0 = (st % sx) * WM_WC;
        __syncthreads();  // the previous strip's fragments have been read
        // operand strip with halo: 16-byte items (pixel, 8-channel group)
        for (int it = tid; it < (WM_RS + 2) * WM_AW * 4; it += 256) {
            const int ch = it & 3, pi = it >> 2;
            const int py = pi / WM_AW, px = pi - py * WM_AW;
            const int h = h0 + py - 1, w = w0 + px - 1;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (h >= 0 && h < H && w >= 0 && w < W) {
                const long ps = (long)(h >> ups) * Ws + (w >> ups);
                v = *(const uint4*)(xb + ps * sg.x_ld + 8 * ch);
                if (bands) {
                    const long p = (long)h * W + w;
                    const float4 g = gb[p];
                    const uint4 v2 = *(const uint4*)(l2b + p * sg.b_ld + 8 * ch);
                    const uint4 v4 = *(const uint4*)(l4b + p * sg.b_ld + 8 * ch);
                    const uint4 v8 = *(const uint4*)(l8b + p * sg.b_ld + 8 * ch);
                    v = modulate8(v, v2, v4, v8, g.x, g.y - g.x, g.z - g.y, 1.0f - g.z);
                }
            }
            *(uint4*)(a_sh + pi * 64 + ch * 16) = v;
        }
        // dy strip: [half][pixel][32 channels]
        for (int it = tid; it < WM_RS * WM_WC * 4 * MT; it += 256) {
            const int c8 = it % (4 * MT), pi = it / (4 * MT);
            const int py = pi / WM_WC, px = pi - py * WM_WC;
            const int h = h0 + py, w = w0 + px;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (h < H && w < W) v = *(const uint4*)(dyb + ((long)h * W + w) * dy_ld + 8 * c8);
            *(uint4*)(d_sh + (c8 >> 2) * WM_D_BYTES + pi * 64 + (c8 & 3) * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < MT; ++rr) {
            const int py = MT == 2 ? (wave >> 1) + 2 * rr : wave;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const wm_short8 fa = wm_frag(d_img, py * WM_WC + 16 * ks, lane);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const wm_short8 fb = wm_frag(a_sh, (py + t / 3) * WM_AW + 16 * ks + t % 3, lane);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // sum the partial tiles of the waves that share an output-channel half, then one atomic per element
    __syncthreads();
    float* red = (float*)wm_smem + mb * (WM_RED_BYTES / 4);
    constexpr int ROUNDS = MT == 2 ? 2 : 4;
    const int my_round = MT == 2 ? (wave >> 1) : wave;
#pragma unroll 1
    for (int r0 = 0; r0 < ROUNDS; ++r0) {
        if (my_round == r0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* q = red + (t * 16 + r) * 64 + lane;
                    *q = r0 == 0 ? acc[t][r] : *q + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = dw + (long)b * dw_bstride;
    const int i0 = sg.c0 + chunk_i * 32;
    for (int e = tid; e < MT * 9 * 16 * 64; e += 256) {
        const int l = e & 63, r = (e >> 6) & 15, t = (e >> 10) % 9, m = e / (9 * 1024);
        const int o = m * 32 + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3), ci = i0 + (l & 31);
        atomicAdd(out + ((long)t * COUT + o) * cin + ci, ((const float*)wm_smem)[e]);
    }
}

int g_wgrad_valu = 0;  // debug hook: keep bf16 weight gradients on the VALU kernel

// true when (segment s of) the operand can be staged with 16-byte loads in 32-channel chunks
static bool wm_view_ok(const View& v) {
    return v.p && v.csh == 3 && v.cbs == 8 && v.ld % 8 == 0 && v.c0 % 8 == 0 && v.bs % 8 == 0 && (uintptr_t)v.p % 16 == 0;
}

static int wgrad_mfma_launch(const Concat& c, int B, int H, int W, const float* gates, const View& dyv, int cout, int per_sample,
                             float* dw, hipStream_t st) {
    if (g_wgrad_valu || (cout != 32 && cout != 64) || !wm_view_ok(dyv) || dyv.c0 + cout > dyv.ld) return SR_NOT_ELIGIBLE;
    for (int s = 0; s < c.nseg; ++s) {
        const int cs = c.cstart[s + 1] - c.cstart[s];
        if (cs % 32 || c.cstart[s] % 32 || !wm_view_ok(c.x[s])) return SR_NOT_ELIGIBLE;
        if (c.bands && (!wm_view_ok(c.l2[s]) || !wm_view_ok(c.l4[s]) || !wm_view_ok(c.l8[s]) || c.l4[s].ld != c.l2[s].ld ||
                        c.l8[s].ld != c.l2[s].ld || c.l4[s].bs != c.l2[s].bs || c.l8[s].bs != c.l2[s].bs))
            return SR_NOT_ELIGIBLE;
    }
    const int cin = c.ctot;
    const long wn = 9L * cout * cin;
    const int nstrips = cdiv(W, WM_WC) * cdiv(H, WM_RS);
    const int mt = cout / 32;
    const size_t lds = (size_t)(WM_A_BYTES + mt * WM_D_BYTES) > (size_t)mt * WM_RED_BYTES ? (size_t)(WM_A_BYTES + mt * WM_D_BYTES)
                                                                                         : (size_t)mt * WM_RED_BYTES;
    // about 1024 workgroups per launch over all segments: fills the chip, few atomic rounds per element
    static const int target_wgs = getenv("SR_AMD_WGRAD_WGS") && atoi(getenv("SR_AMD_WGRAD_WGS")) > 0 ? atoi(getenv("SR_AMD_WGRAD_WGS")) : 1024;
    int spb = (int)(((long)nstrips * (cin / 32) * B + target_wgs - 1) / target_wgs);
    if (spb < 1) spb = 1;
    for (int s = 0; s < c.nseg; ++s) {
        WmSeg sg;
        sg.x = (const bf16_t*)c.x[s].p + c.x[s].c0;
        sg.x_bs = c.x[s].bs;
        sg.x_ld = c.x[s].ld;
        sg.l2 = sg.l4 = sg.l8 = nullptr;
        sg.b_bs = 0;
        sg.b_ld = 0;
        if (c.bands) {
            sg.l2 = (const bf16_t*)c.l2[s].p + c.l2[s].c0;
            sg.l4 = (const bf16_t*)c.l4[s].p + c.l4[s].c0;
            sg.l8 = (const bf16_t*)c.l8[s].p + c.l8[s].c0;
            sg.b_bs = c.l2[s].bs;
            sg.b_ld = c.l2[s].ld;
        }
        sg.c0 = c.cstart[s];
        sg.nchunk = (c.cstart[s + 1] - c.cstart[s]) / 32;
        const dim3 grid(sg.nchunk, cdiv(nstrips, spb), B);
        const bf16_t* dyp = (const bf16_t*)dyv.p + dyv.c0;
        if (mt == 1) {
            SR_LDS_ATTR(lds, wgrad_mfma_kernel<1>);
            wgrad_mfma_kernel<1><<<grid, 256, lds, st>>>(sg, c.bands, c.ups, H, W, cin, gates, dyp, dyv.bs, dyv.ld, dw,
                                                        per_sample ? wn : 0, spb);
        } else {
            SR_LDS_ATTR(lds, wgrad_mfma_kernel<2>);
            wgrad_mfma_kernel<2><<<grid, 256, lds, st>>>(sg, c.bands, c.ups, H, W, cin, gates, dyp, dyv.bs, dyv.ld, dw,
                                                        per_sample ? wn : 0, spb);
        }
        SR_LAUNCH_CHECK();
    }
    return SR_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(View dy, long npix, long total_pix, int cout, float* __restrict__ dbias,
                                                        int pix_per_block) {
    __shared__ float sh[256];
    const long p0 = (long)blockIdx.x * pix_per_block;
    const long p1 = min(total_pix, p0 + pix_per_block);
    for (int ob = 0; ob < cout; ob += 256) {
        const int nc = min(256, cout - ob);
        const int sub = threadIdx.x / nc, nsub = 256 / nc, c = threadIdx.x - sub * nc;
        float s = 0.f;
        if (sub < nsub)
            for (long p = p0 + sub; p < p1; p += nsub) s += Elem<T>::ld((const T*)dy.p + voff(dy, (int)(p / npix), p % npix, ob + c));
        sh[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < nc) {
            float tot = 0.f;
            for (int k = 0; k < nsub; ++k) tot += sh[k * nc + threadIdx.x];
            atomicAdd(dbias + ob + threadIdx.x, tot);
        }
        __syncthreads();
    }
}

// bf16, 8 | cout <= 256: cout / 8 lanes per pixel (16 bytes each), 256 / (cout / 8) pixels per iteration of the workgroup,
// several iterations in flight; per-thread sums meet in LDS and leave as `cout` atomics per workgroup
__global__ __launch_bounds__(256) void bias_grad_vec_kernel(View dy, long npix, long total_pix, int cout, float* __restrict__ dbias,
                                                            int pix_per_block) {
    __shared__ float sh[256][9];
    const int tpp = cout >> 3;            // threads per pixel
    const int slots = 256 / tpp;          // pixels per iteration
    const int sub = threadIdx.x % tpp, slot = threadIdx.x / tpp;
    const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(total_pix, p0 + pix_per_block);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (slot < slots) {
#pragma unroll 4
        for (long p = p0 + slot; p < p1; p += slots) {
            const int b = (int)(p / npix);
            const uint4 q = *(const uint4*)((const bf16_t*)dy.p + voff(dy, b, p - (long)b * npix, sub * 8));
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] += lo_f(w[i]);
                acc[2 * i + 1] += hi_f(w[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sh[threadIdx.x][i] = acc[i];
    __syncthreads();
    if ((int)threadIdx.x < cout) {
        const int s = threadIdx.x >> 3, i = threadIdx.x & 7;
        float tot = 0.f;
        for (int k = 0; k < slots; ++k) tot += sh[k * tpp + s][i];
        atomicAdd(dbias + threadIdx.x, tot);
    }
}

// ---------------------------------------------------------------------------------------------
// FBM backward: one wavefront per pixel, lanes over channels
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void fbm_backward_kernel(long total_pix, long npix, int C, View x, View l2, View l4, View l8,
                                                           const float* __restrict__ gates, View da, View dx, View dl2, View dl4,
                                                           View dl8, float* __restrict__ dlogits, int acc, bf16_t* __restrict__ dzpad) {
    const int lane = threadIdx.x & 63;
    const long wave0 = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    for (long pg = wave0; pg < total_pix; pg += nwaves) {
        const int b = (int)(pg / npix);
        const long p = pg - (long)b * npix;
        const float4 g = ((const float4*)gates)[pg];
        const float c1 = g.y - g.x, c2 = g.z - g.y, c3 = 1.0f - g.z;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = Elem<T>::ld((const T*)da.p + voff(da, b, p, c));
            const float xv = Elem<T>::ld((const T*)x.p + voff(x, b, p, c));
            const float v2 = Elem<T>::ld((const T*)l2.p + voff(l2, b, p, c));
            const float v4 = Elem<T>::ld((const T*)l4.p + voff(l4, b, p, c));
            const float v8 = Elem<T>::ld((const T*)l8.p + voff(l8, b, p, c));
            s1 += d * (xv - v2);
            s2 += d * (v2 - v4);
            s3 += d * (v4 - v8);
            T* o0 = (T*)dx.p + voff(dx, b, p, c);
            T* o2 = (T*)dl2.p + voff(dl2, b, p, c);
            T* o4 = (T*)dl4.p + voff(dl4, b, p, c);
            T* o8 = (T*)dl8.p + voff(dl8, b, p, c);
            Elem<T>::st(o0, g.x * d + (acc ? Elem<T>::ld(o0) : 0.f));
            Elem<T>::st(o2, c1 * d + (acc ? Elem<T>::ld(o2) : 0.f));
            Elem<T>::st(o4, c2 * d + (acc ? Elem<T>::ld(o4) : 0.f));
            Elem<T>::st(o8, c3 * d + (acc ? Elem<T>::ld(o8) : 0.f));
        }
        s1 = wsum(s1);
        s2 = wsum(s2);
        s3 = wsum(s3);
        // g = 2 sigmoid(z)  ->  dg/dz = g (1 - g/2)
        const float z1 = s1 * g.x * (1.0f - 0.5f * g.x), z2 = s2 * g.y * (1.0f - 0.5f * g.y), z3 = s3 * g.z * (1.0f - 0.5f * g.z);
        if (lane == 0) ((float4*)dlogits)[pg] = make_float4(z1, z2, z3, 0.f);
        if (dzpad && lane < 4)  // the same three values as a 32-channel bf16 pixel (zeros beyond): dy operand of the gate-conv weight gradient
            ((uint4*)dzpad)[pg * 4 + lane] = lane == 0 ? make_uint4(pack2(z1, z2), pack2(z3, 0.f), 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    }
}

// bf16 tensors with 8 | C <= 256: PC lanes per pixel (a power of two >= C / 8), 16 bytes (8 channels) per lane and tensor
// -- an eighth of the memory instructions of the channel-per-lane kernel above, the per-pixel sums by shuffles inside
// the PC-lane group.
template <int PC>
__global__ __launch_bounds__(256) void fbm_backward_vec_kernel(long total_pix, long npix, int C, View x, View l2, View l4, View l8,
                                                               const float* __restrict__ gates, View da, View dx, View dl2,
                                                               View dl4, View dl8, float* __restrict__ dlogits, int acc,
                                                               bf16_t* __restrict__ dzpad) {
    constexpr int PPW = 64 / PC;  // pixels per wavefront
    const int lane = threadIdx.x & 63, sub = lane & (PC - 1);
    const long wave0 = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    const bool live = sub * 8 < C;
    for (long pb = wave0 * PPW; pb < total_pix; pb += nwaves * PPW) {
        const long pg = pb + lane / PC;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pg < total_pix) g = ((const float4*)gates)[pg];
        if (live && pg < total_pix) {
            const int b = (int)(pg / npix);
            const long p = pg - (long)b * npix;
            const float c1 = g.y - g.x, c2 = g.z - g.y, c3 = 1.0f - g.z;
            const uint4 dv = *(const uint4*)((const bf16_t*)da.p + voff(da, b, p, sub * 8));
            const uint4 xv = *(const uint4*)((const bf16_t*)x.p + voff(x, b, p, sub * 8));
            const uint4 a2 = *(const uint4*)((const bf16_t*)l2.p + voff(l2, b, p, sub * 8));
            const uint4 a4 = *(const uint4*)((const bf16_t*)l4.p + voff(l4, b, p, sub * 8));
            const uint4 a8 = *(const uint4*)((const bf16_t*)l8.p + voff(l8, b, p, sub * 8));
            const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w}, w2[4] = {a2.x, a2.y, a2.z, a2.w},
                           w4[4] = {a4.x, a4.y, a4.z, a4.w}, w8[4] = {a8.x, a8.y, a8.z, a8.w};
            uint32_t o0[4], o2[4], o4[4], o8[4];
            // accumulate mode (a tensor with several consumers: the consumers' branch gradients are summed in place)
            uint32_t p0[4] = {0u, 0u, 0u, 0u}, p2[4] = {0u, 0u, 0u, 0u}, p4[4] = {0u, 0u, 0u, 0u}, p8[4] = {0u, 0u, 0u, 0u};
            if (acc) {
                const uint4 q0 = *(const uint4*)((const bf16_t*)dx.p + voff(dx, b, p, sub * 8));
                const uint4 q2 = *(const uint4*)((const bf16_t*)dl2.p + voff(dl2, b, p, sub * 8));
                const uint4 q4 = *(const uint4*)((const bf16_t*)dl4.p + voff(dl4, b, p, sub * 8));
                const uint4 q8 = *(const uint4*)((const bf16_t*)dl8.p + voff(dl8, b, p, sub * 8));
                p0[0] = q0.x; p0[1] = q0.y; p0[2] = q0.z; p0[3] = q0.w;
                p2[0] = q2.x; p2[1] = q2.y; p2[2] = q2.z; p2[3] = q2.w;
                p4[0] = q4.x; p4[1] = q4.y; p4[2] = q4.z; p4[3] = q4.w;
                p8[0] = q8.x; p8[1] = q8.y; p8[2] = q8.z; p8[3] = q8.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d0 = lo_f(dw[i]), d1 = hi_f(dw[i]);
                s1 += d0 * (lo_f(xw[i]) - lo_f(w2[i])) + d1 * (hi_f(xw[i]) - hi_f(w2[i]));
                s2 += d0 * (lo_f(w2[i]) - lo_f(w4[i])) + d1 * (hi_f(w2[i]) - hi_f(w4[i]));
                s3 += d0 * (lo_f(w4[i]) - lo_f(w8[i])) + d1 * (hi_f(w4[i]) - hi_f(w8[i]));
                o0[i] = pack2(g.x * d0 + lo_f(p0[i]), g.x * d1 + hi_f(p0[i]));
                o2[i] = pack2(c1 * d0 + lo_f(p2[i]), c1 * d1 + hi_f(p2[i]));
                o4[i] = pack2(c2 * d0 + lo_f(p4[i]), c2 * d1 + hi_f(p4[i]));
                o8[i] = pack2(c3 * d0 + lo_f(p8[i]), c3 * d1 + hi_f(p8[i]));
            }
            *(uint4*)((bf16_t*)dx.p + voff(dx, b, p, sub * 8)) = make_uint4(o0[0], o0[1], o0[2], o0[3]);
            *(uint4*)((bf16_t*)dl2.p + voff(dl2, b, p, sub * 8)) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
            *(uint4*)((bf16_t*)dl4.p + voff(dl4, b, p, sub * 8)) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            *(uint4*)((bf16_t*)dl8.p + voff(dl8, b, p, sub * 8)) = make_uint4(o8[0], o8[1], o8[2], o8[3]);
        }
#pragma unroll
        for (int m = PC / 2; m >= 1; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
            s3 += __shfl_xor(s3, m, 64);
        }
        const float z1 = s1 * g.x * (1.0f - 0.5f * g.x), z2 = s2 * g.y * (1.0f - 0.5f * g.y), z3 = s3 * g.z * (1.0f - 0.5f * g.z);
        if (sub == 0 && pg < total_pix) ((float4*)dlogits)[pg] = make_float4(z1, z2, z3, 0.f);
        if (dzpad && sub < 4 && pg < total_pix)  // PC >= 8 lanes per pixel: lanes 0..3 write the 64-byte padded bf16 pixel
            ((uint4*)dzpad)[pg * 4 + sub] = sub == 0 ? make_uint4(pack2(z1, z2), pack2(z3, 0.f), 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// input-gradient assembly: dx = dx_direct + a2 + a4 + a8 + sum_{k,t} gw[k][c][t] dz_k[p - off(t)] + dpooled[b][c] / (H W)
// lane <-> channel (the 27 gate weights of the channel stay in registers), a wavefront walks a run of pixels; the nine
// neighbouring logit gradients of a pixel are the same address for all lanes (one broadcast load each).
template <typename T>
__global__ __launch_bounds__(256) void input_grad_kernel(int B, int H, int W, int C, const float* __restrict__ dz, sr_layer_params lp,
                                                         View dxd, View a2, View a4, View a8, const float* __restrict__ dpooled,
                                                         int dpooled_ld, View out, int pix_per_wave) {
    const long npix = (long)H * W, total = (long)B * npix;
    const float inv = 1.0f / (float)npix;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.y * 64 + lane;
    const bool live = c < C;
    const int cc = live ? c : C - 1;
    float gw[3][9];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) gw[k][t] = lp.gate_w[k][cc * 9 + t];
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long q0 = wave * pix_per_wave, q1 = min(total, q0 + pix_per_wave);
    // branch-free body (clamped neighbour address, validity as a factor), four pixels deep: the loads of several pixels
    // are in flight together -- one pixel at a time the wave is a chain of dependent memory round trips
#pragma unroll 4
    for (long pg = q0; pg < q1; ++pg) {
        const int b = (int)(pg / npix);
        const long p = pg - (long)b * npix;
        const int h = (int)(p / W), w = (int)(p - (long)h * W);
        float v = Elem<T>::ld((const T*)dxd.p + voff(dxd, b, p, cc));
        if (a2.p)  // (all three or none: checked on the host)
            v += Elem<T>::ld((const T*)a2.p + voff(a2, b, p, cc)) + Elem<T>::ld((const T*)a4.p + voff(a4, b, p, cc)) +
                 Elem<T>::ld((const T*)a8.p + voff(a8, b, p, cc));
        if (dpooled) v += dpooled[(long)b * dpooled_ld + cc] * inv;
        float gsum = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h - (t / 3 - 1), ww = w - (t % 3 - 1);
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;  // wave-uniform
            const float4 d = ((const float4*)dz)[(long)b * npix + (long)min(max(hh, 0), H - 1) * W + min(max(ww, 0), W - 1)];
            const float f = ok ? 1.f : 0.f;
            gsum += f * (gw[0][t] * d.x + gw[1][t] * d.y + gw[2][t] * d.z);
        }
        if (live) Elem<T>::st((T*)out.p + voff(out, b, p, c), v + gsum);
    }
}

// bf16 tensors with 8 | C <= 256: lane <-> (pixel of the run, 8-channel group), 16 bytes per lane and tensor; the 27 gate
// weights of every channel are staged once per workgroup in LDS as [tap][gate][C] floats and read as float4 pairs.
template <int PC>  // lanes per pixel (a power of two >= C / 8)
__global__ __launch_bounds__(256) void input_grad_vec_kernel(int B, int H, int W, int C, const float* __restrict__ dz,
                                                             sr_layer_params lp, View dxd, View a2, View a4, View a8,
                                                             const float* __restrict__ dpooled, int dpooled_ld, View out,
                                                             int pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float gw_lds[];  // [9][3][C]
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) {
        const int c = i % C, k = (i / C) % 3, t = i / (3 * C);
        gw_lds[i] = lp.gate_w[k][c * 9 + t];
    }
    __syncthreads();
    const long npix = (long)H * W, total = (long)B * npix;
    const float inv = 1.0f / (float)npix;
    constexpr int PPI = 256 / PC;  // pixels per iteration of the workgroup
    const int sub = threadIdx.x & (PC - 1), slot = threadIdx.x / PC;
    const bool live = sub * 8 < C;
    const int c0 = live ? sub * 8 : 0;
    const long q0 = (long)blockIdx.x * pix_per_block, q1 = min(total, q0 + pix_per_block);
#pragma unroll 2
    for (long pb = q0; pb < q1; pb += PPI) {
        const long pg = pb + slot;
        if (pg >= q1 || !live) continue;
        const int b = (int)(pg / npix);
        const long p = pg - (long)b * npix;
        const int h = (int)(p / W), w = (int)(p - (long)h * W);
        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
        const uint4 v0 = *(const uint4*)((const bf16_t*)dxd.p + voff(dxd, b, p, c0));
        const uint4 v2 = a2.p ? *(const uint4*)((const bf16_t*)a2.p + voff(a2, b, p, c0)) : zero4;
        const uint4 v4 = a2.p ? *(const uint4*)((const bf16_t*)a4.p + voff(a4, b, p, c0)) : zero4;
        const uint4 v8 = a2.p ? *(const uint4*)((const bf16_t*)a8.p + voff(a8, b, p, c0)) : zero4;
        const uint32_t w0[4] = {v0.x, v0.y, v0.z, v0.w}, w2[4] = {v2.x, v2.y, v2.z, v2.w}, w4[4] = {v4.x, v4.y, v4.z, v4.w},
                       w8[4] = {v8.x, v8.y, v8.z, v8.w};
        float acc[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] = lo_f(w0[i]) + lo_f(w2[i]) + lo_f(w4[i]) + lo_f(w8[i]);
            acc[2 * i + 1] = hi_f(w0[i]) + hi_f(w2[i]) + hi_f(w4[i]) + hi_f(w8[i]);
        }
        if (dpooled) {
            const float4 d0 = *(const float4*)(dpooled + (long)b * dpooled_ld + c0), d1 = *(const float4*)(dpooled + (long)b * dpooled_ld + c0 + 4);
            acc[0] += d0.x * inv; acc[1] += d0.y * inv; acc[2] += d0.z * inv; acc[3] += d0.w * inv;
            acc[4] += d1.x * inv; acc[5] += d1.y * inv; acc[6] += d1.z * inv; acc[7] += d1.w * inv;
        }
        float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h - (t / 3 - 1), ww = w - (t % 3 - 1);
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            float4 d = ((const float4*)dz)[(long)b * npix + (long)min(max(hh, 0), H - 1) * W + min(max(ww, 0), W - 1)];
            if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);
            const float dk[3] = {d.x, d.y, d.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 ga = *(const float4*)(gw_lds + (t * 3 + k) * C + c0), gb = *(const float4*)(gw_lds + (t * 3 + k) * C + c0 + 4);
                gs[0] += ga.x * dk[k]; gs[1] += ga.y * dk[k]; gs[2] += ga.z * dk[k]; gs[3] += ga.w * dk[k];
                gs[4] += gb.x * dk[k]; gs[5] += gb.y * dk[k]; gs[6] += gb.z * dk[k]; gs[7] += gb.w * dk[k];
            }
        }
        uint4 o;
        o.x = pack2(acc[0] + gs[0], acc[1] + gs[1]);
        o.y = pack2(acc[2] + gs[2], acc[3] + gs[3]);
        o.z = pack2(acc[4] + gs[4], acc[5] + gs[5]);
        o.w = pack2(acc[6] + gs[6], acc[7] + gs[7]);
        *(uint4*)((bf16_t*)out.p + voff(out, b, p, c0)) = o;
    }
}

// gate-conv weight / bias gradients: workgroup = 64 channels x a run of pixels; lane <-> channel, the 4 waves take
// interleaved pixels; dgw[k][c][t] += x[q][c] * dz_k[q - off(t)].  The body is branch-free (clamped neighbour address,
// validity as a factor) and unrolled four pixels deep so that the loads of several pixels are in flight: the kernel is a
// stream of dependent-looking loads otherwise.  Few, long runs per workgroup: every workgroup ends in 27 x 64 atomics on
// the same 27 x C addresses (1024 workgroups per channel tile spent their time queueing there: 662 us).
template <typename T>
__global__ __launch_bounds__(256) void gate_wgrad_kernel(int B, int H, int W, int C, View x, const float* __restrict__ dz,
                                                         float* __restrict__ dgw, float* __restrict__ dgb, int pix_per_block) {
    __shared__ float red[4][27][64];
    const long npix = (long)H * W, total = (long)B * npix;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int cc = c < C ? c : C - 1;
    const long q0 = (long)blockIdx.x * pix_per_block, q1 = min(total, q0 + pix_per_block);
    float acc[3][9];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[k][t] = 0.f;
    float bsum[3] = {0.f, 0.f, 0.f};
    const bool do_bias = blockIdx.y == 0 && lane == 0;
#pragma unroll 4
    for (long qg = q0 + wave; qg < q1; qg += 4) {
        const int b = (int)(qg / npix);
        const long q = qg - (long)b * npix;
        const int h = (int)(q / W), w = (int)(q - (long)h * W);
        const float xv = Elem<T>::ld((const T*)x.p + voff(x, b, q, cc));
        if (do_bias) {
            const float4 d = ((const float4*)dz)[qg];
            bsum[0] += d.x;
            bsum[1] += d.y;
            bsum[2] += d.z;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h - (t / 3 - 1), ww = w - (t % 3 - 1);
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const int hc = min(max(hh, 0), H - 1), wc = min(max(ww, 0), W - 1);
            const float4 d = ((const float4*)dz)[(long)b * npix + (long)hc * W + wc];
            const float xm = ok ? xv : 0.f;
            acc[0][t] += xm * d.x;
            acc[1][t] += xm * d.y;
            acc[2][t] += xm * d.z;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) red[wave][k * 9 + t][lane] = acc[k][t];
    __syncthreads();
    if (wave == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int j = k * 9 + t;
                atomicAdd(dgw + ((long)k * C + c) * 9 + t, red[0][j][lane] + red[1][j][lane] + red[2][j][lane] + red[3][j][lane]);
            }
    }
    if (do_bias)
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(dgb + k, bsum[k]);
}

// ---------------------------------------------------------------------------------------------
// elementwise pieces of the network-level backward (no torch operator between the layers):
//   grad_combine : out = mask(act) * sum_j scale_j term_j,  mask = act > 0 ? 1 : slope  (LeakyReLU / PReLU-free backward folded
//                  into the sum of a tensor's gradient contributions: direct branch, three adjoint low-pass planes, residuals)
//   upsample2x   : F.interpolate(scale_factor=2, mode='nearest') of RRDBNet.forward, materialised for the training path
//   downsum2x    : its adjoint (2 x 2 sums)
constexpr int GC_MAX = 8;
struct CombineArgs {
    int n;
    View t[GC_MAX];
    float s[GC_MAX];
};
template <typename T>
__global__ __launch_bounds__(256) void grad_combine_kernel(long total_pix, long npix, int C, CombineArgs a, View act, float slope, View out,
                                                           const float* __restrict__ bc, int bc_ld, float bc_scale) {
    const long n = total_pix * C;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long pg = e / C;
        const int b = (int)(pg / npix);
        const long p = pg - (long)b * npix;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < GC_MAX; ++j)
            if (j < a.n) v += a.s[j] * Elem<T>::ld((const T*)a.t[j].p + voff(a.t[j], b, p, c));
        if (bc) v += bc_scale * bc[(long)b * bc_ld + c];
        if (act.p && !(Elem<T>::ld((const T*)act.p + voff(act, b, p, c)) > 0.f)) v *= slope;
        Elem<T>::st((T*)out.p + voff(out, b, p, c), v);
    }
}
// bf16, 8 channels (16 bytes) per thread
__global__ __launch_bounds__(256) void grad_combine_vec_kernel(long total_pix, long npix, int C8, CombineArgs a, View act, float slope, View out,
                                                               const float* __restrict__ bc, int bc_ld, float bc_scale) {
    const long n = total_pix * C8;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C8) * 8;
        const long pg = e / C8;
        const int b = (int)(pg / npix);
        const long p = pg - (long)b * npix;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < GC_MAX; ++j)
            if (j < a.n) {
                const uint4 q = *(const uint4*)((const bf16_t*)a.t[j].p + voff(a.t[j], b, p, c));
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[2 * i] += a.s[j] * lo_f(w[i]);
                    v[2 * i + 1] += a.s[j] * hi_f(w[i]);
                }
            }
        if (bc) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bc_scale * bc[(long)b * bc_ld + c + i];
        }
        if (act.p) {
            const uint4 q = *(const uint4*)((const bf16_t*)act.p + voff(act, b, p, c));
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!(lo_f(w[i]) > 0.f)) v[2 * i] *= slope;
                if (!(hi_f(w[i]) > 0.f)) v[2 * i + 1] *= slope;
            }
        }
        *(uint4*)((bf16_t*)out.p + voff(out, b, p, c)) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}

// transposed gate convs as ONE 32 -> C convolution for sr_conv3x3: W'[t][c][k] = gate_w_k[c][8 - t] for k < 3, zero for the
// other 29 input channels (the logit gradients travel as 32-channel bf16 pixels), split into hi + lo bf16 parts
__global__ __launch_bounds__(256) void pack_gate_wt_kernel(const float* __restrict__ g0, const float* __restrict__ g1,
                                                           const float* __restrict__ g2, int C, bf16_t* __restrict__ hi,
                                                           bf16_t* __restrict__ lo) {
    const int n = 9 * C * 32;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int k = e & 31, c = (e >> 5) % C, t = e / (32 * C);
        float v = 0.f;
        if (k < 3) v = (k == 0 ? g0 : (k == 1 ? g1 : g2))[c * 9 + (8 - t)];
        const bf16_t h = f32_to_bf16(v);
        hi[e] = h;
        lo[e] = f32_to_bf16(v - bf16_to_f32(h));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(int B, int Hs, int Ws, int C, View src, View dst) {
    const int H = 2 * Hs, W = 2 * Ws;
    const long n = (long)B * H * W * C;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long r = e / C;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H), b = (int)(r / H);
        Elem<T>::st((T*)dst.p + voff(dst, b, (long)h * W + w, c), Elem<T>::ld((const T*)src.p + voff(src, b, (long)(h >> 1) * Ws + (w >> 1), c)));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void downsum2x_kernel(int B, int Hs, int Ws, int C, View src, View dst) {
    const int W = 2 * Ws;
    const long n = (long)B * Hs * Ws * C;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long r = e / C;
        const int w = (int)(r % Ws);
        r /= Ws;
        const int h = (int)(r % Hs), b = (int)(r / Hs);
        const long p = (long)(2 * h) * W + 2 * w;
        const T* sp = (const T*)src.p;
        const float v = Elem<T>::ld(sp + voff(src, b, p, c)) + Elem<T>::ld(sp + voff(src, b, p + 1, c)) +
                        Elem<T>::ld(sp + voff(src, b, p + W, c)) + Elem<T>::ld(sp + voff(src, b, p + W + 1, c));
        Elem<T>::st((T*)dst.p + voff(dst, b, (long)h * Ws + w, c), v);
    }
}

static bool plain_view_ok(const sr_view& v, int C) {
    return v.p && v.C == C && v.c0 >= 0 && v.ld >= 1 && (v.cbs || v.c0 + v.C <= v.ld) && view_block_ok(v);
}

}  // namespace sr

using namespace sr;

extern "C" int sr_conv_weight_transpose(const void* w, void* wt, int dtype, int B, int cout, int cin, void* stream) {
    SR_REQUIRE(w && wt && w != wt && dtype_ok(dtype) && B > 0 && cout > 0 && cin > 0);
    const long n = (long)B * 9 * cout * cin;
    const int grid = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SR_F32)
        weight_transpose_kernel<float><<<grid, 256, 0, st>>>((const float*)w, (float*)wt, B, cout, cin);
    else
        weight_transpose_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)w, (bf16_t*)wt, B, cout, cin);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" void sr_debug_wgrad_valu(int on) { sr::g_wgrad_valu = on; }

extern "C" int sr_conv3x3_wgrad(int dtype, int B, int H, int W, const sr_concat* in, const float* gates, sr_view dy, int cout,
                                int per_sample, float* dw, float* dbias, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && B <= 65535 && H > 0 && W > 0 && in && cout > 0 && dw && plain_view_ok(dy, cout));
    SR_REQUIRE(!in->ups || (H % 2 == 0 && W % 2 == 0));
    const long npix = (long)H * W;
    Concat c;
    int rc = to_concat(in, in->ups ? npix / 4 : npix, npix, &c);
    if (rc != SR_OK) return rc;
    SR_REQUIRE(!c.bands || gates);
    hipStream_t st = (hipStream_t)stream;
    const int cin = c.ctot;
    const long wn = 9L * cout * cin;
    SR_HIP(hipMemsetAsync(dw, 0, (size_t)(per_sample ? B : 1) * wn * sizeof(float), st));
    const int tiles = cdiv(cin, 32) * cdiv(cout, 32);
    const int nstrips = cdiv(W, WG_WC) * cdiv(H, WG_RS);
    // enough workgroups to fill the chip, but as few atomic rounds as that allows
    int spb = (int)(((long)nstrips * tiles * B + 2047) / 2048);
    if (spb < 1) spb = 1;
    dim3 grid(tiles, cdiv(nstrips, spb), B);
    const View dyv = to_view(dy, npix);
    rc = dtype == SR_BF16 ? wgrad_mfma_launch(c, B, H, W, gates, dyv, cout, per_sample, dw, st) : SR_NOT_ELIGIBLE;
    if (rc != SR_OK && rc != SR_NOT_ELIGIBLE) return rc;
    if (rc == SR_NOT_ELIGIBLE) {
        if (dtype == SR_F32)
            wgrad_kernel<float><<<grid, 256, 0, st>>>(c, H, W, gates, dyv, cout, dw, per_sample ? wn : 0, spb);
        else
            wgrad_kernel<bf16_t><<<grid, 256, 0, st>>>(c, H, W, gates, dyv, cout, dw, per_sample ? wn : 0, spb);
        SR_LAUNCH_CHECK();
    }
    if (dbias) {
        SR_HIP(hipMemsetAsync(dbias, 0, (size_t)cout * sizeof(float), st));
        const long total = (long)B * npix;
        int ppb = 256;
        if (dtype == SR_BF16 && cout % 8 == 0 && cout <= 256 && 256 % (cout / 8) == 0 && wm_view_ok(dyv)) {
            // every workgroup ends in `cout` atomics on the same addresses: ~256 workgroups of long, vectorised runs
            while (ppb < 8192 && total / ppb > 256) ppb *= 2;
            bias_grad_vec_kernel<<<cdiv(total, ppb), 256, 0, st>>>(dyv, npix, total, cout, dbias, ppb);
        } else if (dtype == SR_F32)
            bias_grad_kernel<float><<<cdiv(total, ppb), 256, 0, st>>>(dyv, npix, total, cout, dbias, ppb);
        else
            bias_grad_kernel<bf16_t><<<cdiv(total, ppb), 256, 0, st>>>(dyv, npix, total, cout, dbias, ppb);
        SR_LAUNCH_CHECK();
    }
    return SR_OK;
}

extern "C" int sr_fbm_backward(int dtype, int B, int H, int W, sr_view x, sr_view l2, sr_view l4, sr_view l8, const float* gates,
                               sr_view da, sr_view dx, sr_view dl2, sr_view dl4, sr_view dl8, float* dlogits, int accumulate,
                               void* dlogits_bf16x32, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && gates && dlogits && x.C > 0);
    SR_REQUIRE(!dlogits_bf16x32 || (uintptr_t)dlogits_bf16x32 % 16 == 0);
    const int C = x.C;
    const sr_view* vs[9] = {&x, &l2, &l4, &l8, &da, &dx, &dl2, &dl4, &dl8};
    for (const sr_view* v : vs) SR_REQUIRE(plain_view_ok(*v, C));
    const long npix = (long)H * W, total = (long)B * npix;
    const int grid = (int)((total + 3) / 4 < 16384 ? (total + 3) / 4 : 16384);
    hipStream_t st = (hipStream_t)stream;
#define SR_FB(T)                                                                                                                 \
    fbm_backward_kernel<T><<<grid, 256, 0, st>>>(total, npix, C, to_view(x, npix), to_view(l2, npix), to_view(l4, npix),      \
                                                 to_view(l8, npix), gates, to_view(da, npix), to_view(dx, npix),                \
                                                 to_view(dl2, npix), to_view(dl4, npix), to_view(dl8, npix), dlogits, accumulate, (bf16_t*)dlogits_bf16x32)
    bool vec = dtype == SR_BF16 && C % 8 == 0 && C <= 256 && !g_wgrad_valu;  // (the hook keeps the scalar kernel: A/B test)
    for (const sr_view* v : vs) vec = vec && wm_view_ok(to_view(*v, npix));
    if (vec) {
        const int pc = C <= 64 ? 8 : (C <= 128 ? 16 : 32);
        const long waves = (total + 64 / pc - 1) / (64 / pc);
        const int gridv = (int)((waves + 3) / 4 < 16384 ? (waves + 3) / 4 : 16384);
#define SR_FBV(PC_)                                                                                                             \
    fbm_backward_vec_kernel<PC_><<<gridv, 256, 0, st>>>(total, npix, C, to_view(x, npix), to_view(l2, npix), to_view(l4, npix), \
                                                        to_view(l8, npix), gates, to_view(da, npix), to_view(dx, npix),          \
                                                        to_view(dl2, npix), to_view(dl4, npix), to_view(dl8, npix), dlogits, accumulate, (bf16_t*)dlogits_bf16x32)
        if (pc == 8) SR_FBV(8);
        else if (pc == 16) SR_FBV(16);
        else SR_FBV(32);
#undef SR_FBV
    } else if (dtype == SR_F32) SR_FB(float);
    else SR_FB(bf16_t);
#undef SR_FB
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_fbm_gates_backward(int dtype, int B, int H, int W, sr_view x, const float* dlogits, const sr_layer_params* lp,
                                     sr_view dx_direct, sr_view a2, sr_view a4, sr_view a8, const float* dpooled, int dpooled_ld,
                                     sr_view dx_out, float* dgate_w, float* dgate_b, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && dlogits && lp && (!dgate_w) == (!dgate_b) && x.C > 0 && x.C == lp->cin);
    SR_REQUIRE(lp->gate_w[0] && lp->gate_w[1] && lp->gate_w[2]);
    SR_REQUIRE(!dpooled || dpooled_ld >= x.C);
    const int C = x.C;
    const bool adj = a2.p != nullptr;  // the three adjoint low-pass terms: all or none
    SR_REQUIRE((a4.p != nullptr) == adj && (a8.p != nullptr) == adj);
    const sr_view* vs[3] = {&x, &dx_direct, &dx_out};
    for (const sr_view* v : vs) SR_REQUIRE(plain_view_ok(*v, C));
    if (adj) {
        const sr_view* va[3] = {&a2, &a4, &a8};
        for (const sr_view* v : va) SR_REQUIRE(plain_view_ok(*v, C));
    } else {
        a2 = a4 = a8 = sr_view{};
    }
    const long npix = (long)H * W, total = (long)B * npix;
    hipStream_t st = (hipStream_t)stream;
    if (dgate_w) {
        SR_HIP(hipMemsetAsync(dgate_w, 0, (size_t)27 * C * sizeof(float), st));
        SR_HIP(hipMemsetAsync(dgate_b, 0, 3 * sizeof(float), st));
    }
    const int ppw = 16;  // pixels per wavefront
    dim3 g1(cdiv(total, 4 * ppw), cdiv(C, 64));
    const int ppb = total >= 512L * 512 ? 2048 : 512;  // pixels per workgroup: ~2 workgroups per CU and channel tile at the training sizes
    dim3 g2(cdiv(total, ppb), cdiv(C, 64));
#define SR_GB(T)                                                                                                                 \
    do {                                                                                                                         \
        input_grad_kernel<T><<<g1, 256, 0, st>>>(B, H, W, C, dlogits, *lp, to_view(dx_direct, npix), to_view(a2, npix),         \
                                                 to_view(a4, npix), to_view(a8, npix), dpooled, dpooled_ld, to_view(dx_out, npix), ppw); \
        SR_LAUNCH_CHECK();                                                                                                       \
        if (dgate_w) {                                                                                                           \
            gate_wgrad_kernel<T><<<g2, 256, 0, st>>>(B, H, W, C, to_view(x, npix), dlogits, dgate_w, dgate_b, ppb);              \
            SR_LAUNCH_CHECK();                                                                                                   \
        }                                                                                                                        \
    } while (0)
    bool vec = dtype == SR_BF16 && C % 8 == 0 && C <= 256 && !g_wgrad_valu && (!dpooled || (dpooled_ld % 4 == 0 && (uintptr_t)dpooled % 16 == 0));
    const sr_view* vv[5] = {&dx_direct, &dx_out, &a2, &a4, &a8};
    for (int i = 0; i < (adj ? 5 : 2); ++i) vec = vec && wm_view_ok(to_view(*vv[i], npix));
    if (vec) {
        const int pc = C <= 64 ? 8 : (C <= 128 ? 16 : 32);
        const int ppbv = 128;  // pixels per workgroup (the 27 x C gate weights are staged once for them); 1024 left half the CUs idle at the training sizes: 485 us against 208 us for the channel-per-lane kernel
        const size_t ldsv = (size_t)27 * C * sizeof(float);
#define SR_IGV(PC_)                                                                                                              \
    input_grad_vec_kernel<PC_><<<cdiv(total, ppbv), 256, ldsv, st>>>(B, H, W, C, dlogits, *lp, to_view(dx_direct, npix),         \
                                                                     to_view(a2, npix), to_view(a4, npix), to_view(a8, npix),    \
                                                                     dpooled, dpooled_ld, to_view(dx_out, npix), ppbv)
        if (pc == 8) SR_IGV(8);
        else if (pc == 16) SR_IGV(16);
        else SR_IGV(32);
#undef SR_IGV
        SR_LAUNCH_CHECK();
        if (dgate_w) {
            gate_wgrad_kernel<bf16_t><<<g2, 256, 0, st>>>(B, H, W, C, to_view(x, npix), dlogits, dgate_w, dgate_b, ppb);
            SR_LAUNCH_CHECK();
        }
    } else if (dtype == SR_F32) SR_GB(float);
    else SR_GB(bf16_t);
#undef SR_GB
    return SR_OK;
}

extern "C" int sr_grad_combine(int dtype, int B, int H, int W, int C, int nterms, const sr_view* terms, const float* scales, sr_view act,
                               float slope, const float* bc, int bc_ld, float bc_scale, sr_view out, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && nterms >= 1 && nterms <= GC_MAX && terms && scales);
    SR_REQUIRE(!bc || bc_ld >= C);
    SR_REQUIRE(plain_view_ok(out, C) && (!act.p || plain_view_ok(act, C)));
    const long npix = (long)H * W, total = (long)B * npix;
    CombineArgs a{};
    a.n = nterms;
    bool vec = dtype == SR_BF16 && C % 8 == 0 && wm_view_ok(to_view(out, npix)) && (!act.p || wm_view_ok(to_view(act, npix)));
    for (int j = 0; j < nterms; ++j) {
        SR_REQUIRE(plain_view_ok(terms[j], C));
        a.t[j] = to_view(terms[j], npix);
        a.s[j] = scales[j];
        vec = vec && wm_view_ok(a.t[j]);
    }
    View av{};
    if (act.p) av = to_view(act, npix);
    hipStream_t st = (hipStream_t)stream;
    const long work = vec ? total * (C / 8) : total * C;
    const int grid = (int)((work + 255) / 256 < 65535 ? (work + 255) / 256 : 65535);
    if (vec) grad_combine_vec_kernel<<<grid, 256, 0, st>>>(total, npix, C / 8, a, av, slope, to_view(out, npix), bc, bc_ld, bc_scale);
    else if (dtype == SR_F32) grad_combine_kernel<float><<<grid, 256, 0, st>>>(total, npix, C, a, av, slope, to_view(out, npix), bc, bc_ld, bc_scale);
    else grad_combine_kernel<bf16_t><<<grid, 256, 0, st>>>(total, npix, C, a, av, slope, to_view(out, npix), bc, bc_ld, bc_scale);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_upsample2x(int dtype, int B, int Hs, int Ws, int C, sr_view src, sr_view dst, int adjoint, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && Hs > 0 && Ws > 0 && C > 0 && plain_view_ok(src, C) && plain_view_ok(dst, C));
    hipStream_t st = (hipStream_t)stream;
    const long lo = (long)Hs * Ws, hi = 4 * lo;
    const long work = (long)B * (adjoint ? lo : hi) * C;
    const int grid = (int)((work + 255) / 256 < 65535 ? (work + 255) / 256 : 65535);
    if (!adjoint) {
        if (dtype == SR_F32) upsample2x_kernel<float><<<grid, 256, 0, st>>>(B, Hs, Ws, C, to_view(src, lo), to_view(dst, hi));
        else upsample2x_kernel<bf16_t><<<grid, 256, 0, st>>>(B, Hs, Ws, C, to_view(src, lo), to_view(dst, hi));
    } else {
        if (dtype == SR_F32) downsum2x_kernel<float><<<grid, 256, 0, st>>>(B, Hs, Ws, C, to_view(src, hi), to_view(dst, lo));
        else downsum2x_kernel<bf16_t><<<grid, 256, 0, st>>>(B, Hs, Ws, C, to_view(src, hi), to_view(dst, lo));
    }
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_pack_gate_weight_transposed(const sr_layer_params* lp, void* hi, void* lo, void* stream) {
    SR_REQUIRE(lp && hi && lo && lp->cin > 0 && lp->gate_w[0] && lp->gate_w[1] && lp->gate_w[2]);
    const int n = 9 * lp->cin * 32;
    pack_gate_wt_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(lp->gate_w[0], lp->gate_w[1], lp->gate_w[2], lp->cin, (bf16_t*)hi,
                                                                       (bf16_t*)lo);
    SR_LAUNCH_CHECK();
    return SR_OK;
}
