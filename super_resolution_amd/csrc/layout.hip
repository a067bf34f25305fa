// Layout packers (HBM-bound): NCHW <-> NHWC, pixel_unshuffle fold, conv-weight repack.
#include "common.h"

namespace sr {
thread_local int g_last_hip_error = 0;

int to_concat(const sr_concat* in, long npix_x, long npix_bands, Concat* out) {
    if (!in || in->nseg < 1 || in->nseg > SR_MAX_SEG) return SR_ERR_ARG;
    Concat c{};
    c.nseg = in->nseg;
    c.bands = in->bands ? 1 : 0;
    c.ups = in->ups ? 1 : 0;
    int tot = 0;
    for (int s = 0; s < in->nseg; ++s) {
        const sr_view& v = in->x[s];
        if (!v.p || v.C < 1 || v.c0 < 0 || v.ld < 1 || (!v.cbs && v.c0 + v.C > v.ld) || !view_block_ok(v)) return SR_ERR_ARG;
        c.x[s] = to_view(v, npix_x);
        c.cstart[s] = tot;
        tot += v.C;
        if (c.bands) {
            const sr_view* bs[3] = {&in->l2[s], &in->l4[s], &in->l8[s]};
            for (int k = 0; k < 3; ++k)
                if (!bs[k]->p || bs[k]->C != v.C || bs[k]->c0 < 0 || (!bs[k]->cbs && bs[k]->c0 + v.C > bs[k]->ld) ||
                    !view_block_ok(*bs[k]))
                    return SR_ERR_ARG;
            c.l2[s] = to_view(in->l2[s], npix_bands);
            c.l4[s] = to_view(in->l4[s], npix_bands);
            c.l8[s] = to_view(in->l8[s], npix_bands);
        }
    }
    for (int s = in->nseg; s <= SR_MAX_SEG; ++s) c.cstart[s] = tot;
    c.ctot = tot;
    *out = c;
    return SR_OK;
}

template <typename TS, typename TD>
__global__ void pack_nchw_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int B, int C, int H, int W, int r) {
    // one thread per destination element; dst (B, H/r, W/r, C*r*r)
    const int Ho = H / r, Wo = W / r, Co = C * r * r;
    const long n = (long)B * Ho * Wo * Co;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int co = (int)(i % Co);
        long t = i / Co;
        int wo = (int)(t % Wo);
        t /= Wo;
        int ho = (int)(t % Ho);
        int b = (int)(t / Ho);
        int c = co / (r * r), dy = (co / r) % r, dx = co % r;
        float v = Elem<TS>::ld(src + (((long)b * C + c) * H + ho * r + dy) * W + wo * r + dx);
        Elem<TD>::st(dst + i, v);
    }
}

// NCHW -> NHWC with the channel count padded to `cpad` (zeros beyond C): one thread per destination PIXEL writes the
// whole 2 * cpad-byte record (bf16) in 16-byte pieces.  A 3-channel image becomes a 32-channel tensor that the MFMA
// kernels take as one K-chunk (SRVGGNetCompact's 3 -> 64 head: 1.17 ms on the VALU kernel, 0.45 ms this way incl. the pack).
template <typename TS>
__global__ void pack_nchw_padded_kernel(const TS* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int H, int W, int cpad) {
    const long n = (long)B * H * W;
    const long hw = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / hw, p = i - b * hw;
        uint4* o = (uint4*)(dst + i * cpad);
        for (int g = 0; g < cpad / 8; ++g) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = g * 8 + j;
                v[j] = c < C ? Elem<TS>::ld(src + (b * C + c) * hw + p) : 0.f;
            }
            o[g] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
        }
    }
}

template <typename TS, typename TD>
__global__ void unpack_nhwc_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int B, int C, int H, int W) {
    // one thread per destination (NCHW) element: coalesced writes; reads stride C (C is 1..3 at the boundary)
    const long n = (long)B * C * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int w = (int)(i % W);
        long t = i / W;
        int h = (int)(t % H);
        t /= H;
        int c = (int)(t % C);
        int b = (int)(t / C);
        float v = Elem<TS>::ld(src + (((long)b * H + h) * W + w) * C + c);
        Elem<TD>::st(dst + i, v);
    }
}

template <typename TD>
__global__ void pack_weight_kernel(const float* __restrict__ w, TD* __restrict__ dst, TD* __restrict__ dst_lo, int cout,
                                   int cin) {
    const int n = 9 * cout * cin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int c = i % cin;
        int o = (i / cin) % cout;
        int tap = i / (cin * cout);
        const float v = w[((long)o * cin + c) * 9 + tap];
        Elem<TD>::st(dst + i, v);
        if (dst_lo) Elem<TD>::st(dst_lo + i, v - Elem<TD>::ld(dst + i));  // what rounding to TD lost (w = hi + lo)
    }
}

static int grid_for(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
}  // namespace sr

using namespace sr;

extern "C" const char* sr_strerror(int code) {
    switch (code) {
        case SR_OK: return "ok";
        case SR_ERR_ARG: return "bad argument or unsupported shape";
        case SR_ERR_WORKSPACE: return "workspace too small";
        case SR_ERR_HIP: return "HIP runtime error";
        case SR_ERR_STATE: return "object not prepared for this shape";
        default: return "unknown error";
    }
}
extern "C" int sr_last_hip_error(void) { return g_last_hip_error; }
extern "C" int sr_version(void) { return 100; }

extern "C" int sr_pack_nchw_padded(const void* src, int sdt, void* dst, int B, int C, int H, int W, int cpad, void* stream) {
    SR_REQUIRE(src && dst && dtype_ok(sdt) && B > 0 && C > 0 && H > 0 && W > 0 && cpad >= C && cpad % 8 == 0);
    SR_REQUIRE((uintptr_t)dst % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for((long)B * H * W);
    if (sdt == SR_F32) pack_nchw_padded_kernel<float><<<g, 256, 0, st>>>((const float*)src, (bf16_t*)dst, B, C, H, W, cpad);
    else pack_nchw_padded_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, B, C, H, W, cpad);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_pack_nchw(const void* src, int sdt, void* dst, int ddt, int B, int C, int H, int W, int r,
                            void* stream) {
    SR_REQUIRE(src && dst && dtype_ok(sdt) && dtype_ok(ddt) && B > 0 && C > 0 && H > 0 && W > 0 && r >= 1);
    SR_REQUIRE(H % r == 0 && W % r == 0);
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for((long)B * C * H * W);
    if (sdt == SR_F32 && ddt == SR_F32)
        pack_nchw_kernel<float, float><<<g, 256, 0, st>>>((const float*)src, (float*)dst, B, C, H, W, r);
    else if (sdt == SR_F32)
        pack_nchw_kernel<float, bf16_t><<<g, 256, 0, st>>>((const float*)src, (bf16_t*)dst, B, C, H, W, r);
    else if (ddt == SR_F32)
        pack_nchw_kernel<bf16_t, float><<<g, 256, 0, st>>>((const bf16_t*)src, (float*)dst, B, C, H, W, r);
    else
        pack_nchw_kernel<bf16_t, bf16_t><<<g, 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, B, C, H, W, r);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_unpack_nhwc(const void* src, int sdt, void* dst, int ddt, int B, int C, int H, int W, void* stream) {
    SR_REQUIRE(src && dst && dtype_ok(sdt) && dtype_ok(ddt) && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for((long)B * C * H * W);
    if (sdt == SR_F32 && ddt == SR_F32)
        unpack_nhwc_kernel<float, float><<<g, 256, 0, st>>>((const float*)src, (float*)dst, B, C, H, W);
    else if (sdt == SR_F32)
        unpack_nhwc_kernel<float, bf16_t><<<g, 256, 0, st>>>((const float*)src, (bf16_t*)dst, B, C, H, W);
    else if (ddt == SR_F32)
        unpack_nhwc_kernel<bf16_t, float><<<g, 256, 0, st>>>((const bf16_t*)src, (float*)dst, B, C, H, W);
    else
        unpack_nhwc_kernel<bf16_t, bf16_t><<<g, 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, B, C, H, W);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_pack_conv_weight(const float* w, void* dst, void* dst_lo, int dtype, int cout, int cin, void* stream) {
    SR_REQUIRE(w && dst && dtype_ok(dtype) && cout > 0 && cin > 0);
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for(9L * cout * cin);
    if (dtype == SR_F32)
        pack_weight_kernel<float><<<g, 256, 0, st>>>(w, (float*)dst, (float*)dst_lo, cout, cin);
    else
        pack_weight_kernel<bf16_t><<<g, 256, 0, st>>>(w, (bf16_t*)dst, (bf16_t*)dst_lo, cout, cin);
    SR_LAUNCH_CHECK();
    return SR_OK;
}
