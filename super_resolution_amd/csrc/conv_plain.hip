// 3x3 convolution with SHARED weights and no band modulation (plain nn.Conv2d layers: SRVGGNetCompact's body -- BASELINE
// configs[4], srvgg_arch.py:32-49 --, basicsr's RRDBNet before replace_conv_with_fdconv -- configs[0] --, the VGG19 feature
// extractor's 64-channel layers), bf16, same operand layouts, concat list, nearest-2x fold and epilogue as sr_conv3x3's
// MFMA kernel in conv.hip, which keeps every other case.
//
// Why a second kernel (DESIGN.md finding 32): conv.hip's producer / consumer workgroup is built around the banded FDConv
// operand, whose pole is the loader waves' HBM stream.  A plain layer has a quarter of those bytes and nothing to mix: there
// the four MFMA waves are the pole -- multiply, then an epilogue whose sixteen stores per lane leave a CU at ~4 B per cycle,
// back to back, while the loaders wait -- and the layer ran at 0.19 of the MFMA roof.
//
// Here every wave is autonomous.  A workgroup of 8 waves walks 8*NR x 32 pixel tiles (XCD-contiguous, as conv.hip does);
// wave v owns NR rows of the tile and, per 32-channel chunk,
//   1. copies its own haloed (NR + 2) x 34 pixel strip global -> LDS with global_load_lds_dwordx4 (no registers, no VALU:
//      the 16-byte-slot XOR swizzle of the LDS image is applied to the SOURCE address; out-of-image pixels read a 64-byte
//      block of zeros; nearest-2x is an address map),
//   2. waits for its own copies (s_waitcnt vmcnt(0)) -- no barrier: nobody else reads the strip,
//   3. multiplies: 9 taps x 2 k-steps of v_mfma_f32_32x32x16_bf16, weight fragments from the layer's WHOLE weight tensor,
//      which stays resident in LDS for the kernel's lifetime (fetched once per workgroup), pixel fragments from the strip.
// After the last chunk the wave runs its epilogue and stores its rows.  The only barrier of the kernel follows the weight
// fetch.  With two waves per SIMD one wave's copies and stores sit under the other's multiplies, and the epilogue is off
// the critical path because no other wave waits for it.  Price: the strips' row halos are fetched per wave ((NR + 2) / NR of
// the rows, from L2), and LDS holds weights + 8 strips, so NR is 3 for a 64 -> 64 layer (24 x 32 tiles).
#include "common.h"

namespace sr {
namespace {

typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(16))) float float16v;

constexpr int PPW = 34;  // haloed tile width

__device__ __forceinline__ int pswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void pglds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ uint4 g_zero_px[4];  // 64 bytes of zeros: the copy source of pixels outside the image (zero padding)

template <int MT, int NR>
__global__ __launch_bounds__(512) void conv3x3_plain_mfma(Concat in, int B, int H, int W, const bf16_t* __restrict__ wts, int cin, int cout,
                                                          Epilogue ep, View out, unsigned total) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TH = 8 * NR;
    constexpr int SPIX = (NR + 2) * PPW;  // pixels of a strip
    constexpr int STRIP = SPIX * 64;      // bytes: 64-byte rows, one per pixel
    constexpr int WSLAB = 9 * MT * 32 * 64;
    const int nchunks = cin / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hk = lane >> 5;
    char* const lds_w = lds;
    float* const bias_lds = (float*)(lds + nchunks * WSLAB);  // [MT * 32]
    float* const slope_lds = bias_lds + MT * 32;              // [MT * 32] negative slopes: PReLU's, 0.2 (LeakyReLU), 1 (none)
    char* const strip = lds + nchunks * WSLAB + 2 * MT * 32 * (int)sizeof(float) + wave * STRIP;

    // ---- the layer's weights: rows (chunk, tap, o) x 32 channels of [9][cout][cin], 1 KiB (16 rows) per copy instruction;
    // lane l fills 16-byte slot l & 3 of row l >> 2, which holds channel group (l & 3) ^ ((row >> 2) & 3)
    {
        const int NI = nchunks * (WSLAB / 1024);
        for (int j = wave; j < NI; j += 8) {
            const int row_all = j * 16 + (lane >> 2);
            const int c = row_all / (9 * MT * 32), row = row_all - c * (9 * MT * 32);
            const int cgrp = (lane & 3) ^ ((row >> 2) & 3);
            const int tap = row / (MT * 32), o = row - tap * (MT * 32);
            const int oc = min(o, cout - 1);
            pglds16(wts + ((long)tap * cout + oc) * cin + c * 32 + cgrp * 8, lds_w + j * 1024);
        }
        for (int i = threadIdx.x; i < MT * 32; i += 512) {
            bias_lds[i] = (ep.bias && i < cout) ? ep.bias[i] : 0.f;
            slope_lds[i] = i < cout ? (ep.slope ? ep.slope[i] : (ep.lrelu ? 0.2f : 1.0f)) : 1.0f;
        }
    }
    __syncthreads();  // drains this wave's copies (vmcnt) and LDS writes: the only barrier of the kernel

    // LDS byte offsets of the fragments, once per kernel.  A fragment row index is K + r with K a compile-time constant:
    // pswz(16 a + k + r, chunk) = 1024 a + pswz(k + r, chunk) (the swizzle has period 16 rows), so one register per (k, k-step
    // half) -- k = K % 16 = 2 (n + dy) + dx takes 2 NR + 5 values for the pixel rows (K = (n + dy) * 34 + dx), 0 for the weight rows (multiples of
    // 32) -- plus the instruction's immediate offset addresses every ds_read of the multiply: no address arithmetic in it
    // (left to the compiler, the ~70 distinct swizzled addresses are hoisted out of the tile loop and spill).
    constexpr int NCLS = 2 * NR + 5;  // K % 16 = 2 (n + dy) + dx <= 2 NR + 4
    static_assert(NCLS <= 16, "strip rows alias in the swizzle period");
    int bb[NCLS][2], wa[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        wa[kk] = pswz(r, kk * 2 + hk);
#pragma unroll
        for (int k = 0; k < NCLS; ++k) bb[k][kk] = (int)(strip - lds) + pswz(k + r, kk * 2 + hk);
    }

    const int tilesx = (W + 31) / 32;
    const unsigned tiles = (unsigned)tilesx * ((H + TH - 1) / TH);
    const int Ws = W >> in.ups;
    // XCD-contiguous walk (conv.hip's TileWalk): workgroup j lives on XCD j % 8 and strides through that XCD's range
    unsigned wnext, wend, wstride;
    {
        const unsigned j = blockIdx.x, x = j & 7u, q = total >> 3, rr = total & 7u;
        const unsigned start = x * q + (x < rr ? x : rr);
        wend = start + q + (x < rr ? 1u : 0u);
        wstride = (gridDim.x - x + 7u) >> 3;
        wnext = start + (j >> 3);
    }
    for (unsigned lin = wnext; lin < wend; lin += wstride) {
        const int tile = lin % tiles, b = lin / tiles;
        const int h0 = (tile / tilesx) * TH, w0 = (tile % tilesx) * 32;
        const int hw0 = h0 + wave * NR;  // this wave's first output row
        if (hw0 >= H) continue;          // nothing of this wave's rows lies in the image (no barrier follows)

        // per tile: clamped source column (in source pixels) and inside-image bit of the lane's entry in each part of a row
        unsigned colb[3];
        unsigned x_in = 0;
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            const int wq = w0 - 1 + ((part * 64 + lane) >> 2);
            const int wc = min(max(wq, 0), W - 1);
            colb[part] = in.ups ? (unsigned)(wc >> 1) : (unsigned)wc;
            if (wq == wc) x_in |= 1u << part;
        }

        float16v acc[MT][NR];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

        int c = 0;
        for (int s = 0; s < in.nseg; ++s) {
            const View vx = in.x[s];
            const char* xb = (const char*)((const bf16_t*)vx.p + (long)b * vx.bs);
            for (int cc = 0; cc < vx.C; cc += 32, ++c) {
                // ---- 1. the strip of this chunk, row by row: 3 copy instructions per 34-pixel row (136 entries of 16 bytes; the
                // third moves 8).  Entry q = part * 64 + lane of a row is 16-byte slot q & 3 of pixel q >> 2 and holds channel
                // group slot ^ ((strip pixel >> 2) & 3).  The row term of every address is wave-uniform; the column term
                // (colb, x_in: per tile) and the channel-group step are all that is per lane.
                const unsigned cbase = (unsigned)(vx.c0 + cc);
                const unsigned soff0 = (cbase >> vx.csh) * (unsigned)vx.cbs + (cbase & ((1u << vx.csh) - 1u));
                const unsigned cgstep = vx.csh == 3 ? (unsigned)vx.cbs : 8u;  // element distance of consecutive 8-channel groups
#pragma unroll
                for (int row = 0; row < NR + 2; ++row) {
                    const int h = hw0 - 1 + row;  // uniform
                    const int hc = min(max(h, 0), H - 1);
                    const unsigned rowb = in.ups ? (unsigned)(hc >> 1) * (unsigned)Ws : (unsigned)hc * (unsigned)W;
                    const bool h_in = h == hc;
#pragma unroll
                    for (int part = 0; part < 3; ++part) {
                        const int q = part * 64 + lane;
                        const int col = q >> 2;
                        const unsigned cg = (unsigned)((q & 3) ^ (((2 * row + col) >> 2) & 3));  // ((row * 34 + col) >> 2) & 3
                        const unsigned el = (rowb + colb[part]) * (unsigned)vx.ld + soff0 + cg * cgstep;
                        const char* src = (h_in && ((x_in >> part) & 1u)) ? xb + (size_t)(2u * el) : (const char*)g_zero_px;
                        if (part < 2 || lane < 8) pglds16(src, strip + row * (PPW * 64) + part * 1024);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // 2. own copies landed (and own earlier stores retired)
                // ---- 3. multiply, fully unrolled, in six groups (k-step half kk, column tap dx).  A group reads the NR + 2 strip
                // rows at that column shift ONCE and uses row n + dy for all three row taps: NR + 2 + 3 MT fragment reads per
                // 3 NR MT MFMAs (11 per 18 at 64 -> 64) instead of NR + MT per NR MT when every tap reads its own rows (15 per
                // 18) -- LDS bandwidth is what the MFMAs wait for here.  One fragment set: the other wave of the SIMD fills the
                // read latency at a group's start.
                const int wc0 = wa[0] + c * WSLAB, wc1 = wa[1] + c * WSLAB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        short8 fa[3][MT], fb[NR + 2];
#pragma unroll
                        for (int m = 0; m < MT; ++m) fa[0][m] = *(const short8*)&lds[(kk ? wc1 : wc0) + (dx * (MT * 32) + m * 32) * 64];
#pragma unroll
                        for (int R = 0; R < NR + 2; ++R) {
                            const int K = R * PPW + dx;
                            fb[R] = *(const short8*)&lds[bb[K & 15][kk] + (K >> 4) * 1024];
                        }
#pragma unroll
                        for (int dy = 1; dy < 3; ++dy)
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                fa[dy][m] = *(const short8*)&lds[(kk ? wc1 : wc0) + ((dy * 3 + dx) * (MT * 32) + m * 32) * 64];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int n = 0; n < NR; ++n)
#pragma unroll
                                for (int m = 0; m < MT; ++m)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[dy][m], fb[n + dy], acc[m][n], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);  // groups stay in sequence: unrolled, the scheduler pulls every read up and spills
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment read before the strip is overwritten
            }
        }

        // ---- epilogue (conv.hip's arithmetic), one output row at a time so that the live set stays small (no other wave waits
        // for this one): lane (r, hk) holds, for pixel (h, w0 + r), channels 8q + 4hk + {0..3} of each 32-channel tile;
        // v_permlane32_swap between lanes r and r + 32 regroups them into 16 consecutive channels per lane
        int w = w0 + r;
        asm volatile("" : "+v"(w));  // opaque: keeps the epilogue's 64-bit addresses from being computed (and spilled) ahead of the multiplies
        auto fold_row = [&](const View& rv, float sc, int n, long pix) {  // residual operand of one row
            uint4 rr[MT][2];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    rr[m][half] = *(const uint4*)((const bf16_t*)rv.p + voff(rv, b, pix, m * 32 + hk * 16 + 8 * half));
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const uint32_t rw[4] = {rr[m][half].x, rr[m][half].y, rr[m][half].z, rr[m][half].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[m][n][8 * half + 2 * j] = acc[m][n][8 * half + 2 * j] * sc + lo_f(rw[j]);
                        acc[m][n][8 * half + 2 * j + 1] = acc[m][n][8 * half + 2 * j + 1] * sc + hi_f(rw[j]);
                    }
                }
        };
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int h = hw0 + n;
            const long pixc = (long)min(h, H - 1) * W + min(w, W - 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float y[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][n][j]), __float_as_uint(acc[m][n][8 + j]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][n][4 + j]), __float_as_uint(acc[m][n][12 + j]), false, false);
                    y[j] = __uint_as_float(s0[0]);
                    y[4 + j] = __uint_as_float(s0[1]);
                    y[8 + j] = __uint_as_float(s1[0]);
                    y[12 + j] = __uint_as_float(s1[1]);
                }
                const float4* bl = (const float4*)(bias_lds + m * 32 + hk * 16);
                const float4* sl = (const float4*)(slope_lds + m * 32 + hk * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bb4 = bl[q], ss = sl[q];
                    const float bq[4] = {bb4.x, bb4.y, bb4.z, bb4.w}, sq[4] = {ss.x, ss.y, ss.z, ss.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = y[4 * q + j] + bq[j];
                        acc[m][n][4 * q + j] = v > 0.f ? v : sq[j] * v;
                    }
                }
            }
            if (ep.r1.p) fold_row(ep.r1, ep.s1, n, pixc);
            if (ep.r2.p) fold_row(ep.r2, ep.s2, n, pixc);
            if (h < H && w < W) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int o = 8 * half;
                        uint4 st;
                        st.x = pack2(acc[m][n][o], acc[m][n][o + 1]);
                        st.y = pack2(acc[m][n][o + 2], acc[m][n][o + 3]);
                        st.z = pack2(acc[m][n][o + 4], acc[m][n][o + 5]);
                        st.w = pack2(acc[m][n][o + 6], acc[m][n][o + 7]);
                        *(uint4*)((bf16_t*)out.p + voff(out, b, pixc, m * 32 + hk * 16 + 8 * half)) = st;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);  // rows stay in sequence
        }
    }
}

template <int MT, int NR>
int launch_plain(int B, int H, int W, const Concat& in, const void* wts, int cout, const Epilogue& ep, View out, hipStream_t st) {
    const size_t ldsb = (size_t)(in.ctot / 32) * 9 * MT * 32 * 64 + 2 * MT * 32 * sizeof(float) + 8ul * (NR + 2) * PPW * 64;
    if (ldsb > 160 * 1024) return SR_NOT_ELIGIBLE;
    SR_LDS_ATTR(ldsb, conv3x3_plain_mfma<MT, NR>);
    const long nblk = (long)cdiv(H, 8 * NR) * cdiv(W, 32) * B;
    if (nblk >= (1L << 31)) return SR_ERR_ARG;
    int ncu = 0;
    if (int rc = device_cus(&ncu)) return rc;
    const unsigned grid = (unsigned)(nblk < ncu ? nblk : ncu);
    conv3x3_plain_mfma<MT, NR><<<dim3(grid), 512, ldsb, st>>>(in, B, H, W, (const bf16_t*)wts, in.ctot, cout, ep, out, (unsigned)nblk);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

}  // namespace

int g_no_plain_kernel = 0;  // test hook (A/B): plain layers through conv.hip's producer / consumer kernel

// Called by conv3x3_launch for bf16 layers that passed mfma_eligible().  SR_NOT_ELIGIBLE: the caller's kernels take it.
int conv3x3_plain_launch(int B, int H, int W, const Concat& in, const void* wts, long wbs, long wlo, int cout, const Epilogue& ep,
                         View out, hipStream_t st) {
    if (g_no_plain_kernel || in.bands || wbs != 0 || wlo != 0 || (cout != 32 && cout != 64)) return SR_NOT_ELIGIBLE;
    if (W >= 65536 || H >= 65536) return SR_NOT_ELIGIBLE;
    for (int s = 0; s < in.nseg; ++s) {  // the copies step through a chunk's 8-channel groups at a fixed element distance
        const View& v = in.x[s];
        if (!(v.csh == 3 || (v.csh >= 5 && v.c0 % 32 == 0))) return SR_NOT_ELIGIBLE;
    }
    int rc = SR_NOT_ELIGIBLE;
    if (cout == 64) {
        rc = launch_plain<2, 3>(B, H, W, in, wts, cout, ep, out, st);
        if (rc == SR_NOT_ELIGIBLE) rc = launch_plain<2, 2>(B, H, W, in, wts, cout, ep, out, st);
    } else {
        rc = launch_plain<1, 4>(B, H, W, in, wts, cout, ep, out, st);
        if (rc == SR_NOT_ELIGIBLE) rc = launch_plain<1, 3>(B, H, W, in, wts, cout, ep, out, st);
        if (rc == SR_NOT_ELIGIBLE) rc = launch_plain<1, 2>(B, H, W, in, wts, cout, ep, out, st);
    }
    return rc;
}
}  // namespace sr

extern "C" void sr_debug_no_plain_kernel(int on) { sr::g_no_plain_kernel = on; }
