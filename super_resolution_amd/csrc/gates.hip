// FBM spatial gates: g_k = 2*sigmoid(conv3x3(x; w_k) + b_k), k = 1..3  (FDConv.py:460-461, 417-419).
// Cin -> 3 output planes: a bandwidth-shaped kernel.  One workgroup = one 16x16 pixel tile; the
// halo tile of a few channels is staged in LDS plane-major so that the 9-tap reads of
// neighbouring lanes hit neighbouring banks.
#include <cstdlib>
#include <algorithm>

#include "common.h"

namespace sr {

constexpr int GT = 16;        // tile side
constexpr int GCC = 16;       // channels per LDS stage
constexpr int GROW = GT + 3;  // padded row length (GT+2 used)

template <typename T>
__global__ __launch_bounds__(256) void fbm_gates_kernel(Concat in, int H, int W, const float* __restrict__ w0,
                                                        const float* __restrict__ w1, const float* __restrict__ w2,
                                                        const float* __restrict__ b0, const float* __restrict__ b1,
                                                        const float* __restrict__ b2, float* __restrict__ gates) {
    __shared__ float tile[GCC][GT + 2][GROW];
    __shared__ float wsh[3][GCC][9];
    const int tilesx = (W + GT - 1) / GT;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tilesx) * GT, tx0 = (blockIdx.x % tilesx) * GT;
    const int ty = threadIdx.x / GT, tx = threadIdx.x % GT;
    const int Ws = W >> in.ups;
    const float* wk[3] = {w0, w1, w2};
    float acc[3] = {0.f, 0.f, 0.f};
    for (int s = 0; s < in.nseg; ++s) {
        const View v = in.x[s];
        for (int cc = 0; cc < v.C; cc += GCC) {
            const int nc = min(GCC, v.C - cc);
            __syncthreads();
            for (int it = threadIdx.x; it < (GT + 2) * (GT + 2) * nc; it += blockDim.x) {
                const int c = it % nc, pi = it / nc;
                const int py = pi / (GT + 2), px = pi % (GT + 2);
                const int h = ty0 + py - 1, w = tx0 + px - 1;
                float val = 0.f;
                if (h >= 0 && h < H && w >= 0 && w < W)
                    val = Elem<T>::ld((const T*)v.p + voff(v, b, (long)(h >> in.ups) * Ws + (w >> in.ups), cc + c));
                tile[c][py][px] = val;
            }
            for (int it = threadIdx.x; it < 3 * nc * 9; it += blockDim.x) {
                const int tap = it % 9, c = (it / 9) % nc, k = it / (9 * nc);
                wsh[k][c][tap] = wk[k][(long)(in.cstart[s] + cc + c) * 9 + tap];
            }
            __syncthreads();
            for (int c = 0; c < nc; ++c) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float x = tile[c][ty + tap / 3][tx + tap % 3];
                    acc[0] += x * wsh[0][c][tap];
                    acc[1] += x * wsh[1][c][tap];
                    acc[2] += x * wsh[2][c][tap];
                }
            }
        }
    }
    const int h = ty0 + ty, w = tx0 + tx;
    if (h < H && w < W) {
        float4 g;
        g.x = 2.0f / (1.0f + expf(-(acc[0] + b0[0])));
        g.y = 2.0f / (1.0f + expf(-(acc[1] + b1[0])));
        g.z = 2.0f / (1.0f + expf(-(acc[2] + b2[0])));
        g.w = 0.f;
        ((float4*)gates)[((long)b * H + h) * W + w] = g;
    }
}


// ---------------------------------------------------------------------------------------------
// bf16 MFMA version.  The 3x3 conv with 3 outputs is rewritten as a 1x1 product with 27 outputs,
//     T[p][tap*3+k] = sum_c x[p][c] * w_k[c][tap],
// followed by a 9-term shifted sum  logit_k[p] = sum_tap T[p + d(tap)][tap*3+k].  The product is
// an M=32 (27 used) x N=pixels x K=Cin GEMM on v_mfma_f32_32x32x16_bf16: 9x fewer matrix ops than
// the conv form, the activation operand goes global -> registers (each lane's fragment is 16
// contiguous bytes of one pixel), and the kernel is bound by reading x once (+ tile halo).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short g_short8;
typedef __attribute__((ext_vector_type(16))) float g_float16;

constexpr int GMH = 8, GMW = 32;                 // interior tile
constexpr int GPH = GMH + 2, GPW = GMW + 2;      // haloed tile: 340 pixel slots
constexpr int GSLOTS = GPH * GPW;
constexpr int GNT = (GSLOTS + 31) / 32;          // 11 N-tiles of 32 slots
constexpr int GTROW = 27;                        // floats per slot in the T image (odd: conflict-free)

__global__ __launch_bounds__(256) void fbm_gates_mfma(Concat in, int B, int H, int W, const float* __restrict__ w0,
                                                      const float* __restrict__ w1, const float* __restrict__ w2,
                                                      const float* __restrict__ b0, const float* __restrict__ b1,
                                                      const float* __restrict__ b2, float* __restrict__ gates) {
    extern __shared__ __attribute__((aligned(16))) char glds[];
    float* T = (float*)glds;                                  // [GNT*32][27]
    char* A = glds + GNT * 32 * GTROW * sizeof(float);        // [32][cin*2 + 16] bf16 rows, padded
    const int cin = in.ctot;
    const int astride = cin * 2 + 16;
    const int tilesx = (W + GMW - 1) / GMW;
    const unsigned tiles = (unsigned)tilesx * ((H + GMH - 1) / GMH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hk = lane >> 5;
    const int Ws = W >> in.ups;
    // stage the 27 x cin weight matrix as bf16 (rows 27..31 are zero) -- once per workgroup: the kernel is persistent
    // (staging it per 8 x 32 tile cost more than the tile itself)
    const float* wk[3] = {w0, w1, w2};
    for (int it = tid; it < 32 * cin; it += 256) {
        const int c = it % cin, row = it / cin;
        float v = 0.f;
        if (row < 27) v = wk[row % 3][(long)c * 9 + row / 3];
        *(bf16_t*)(A + row * astride + c * 2) = f32_to_bf16(v);
    }
    // XCD-contiguous walk over (tile, sample): workgroup j lives on XCD j % 8 and strides through that XCD's range
    const unsigned total = tiles * (unsigned)B;
    const unsigned xj = blockIdx.x & 7u, q8 = total >> 3, r8 = total & 7u;
    const unsigned wstart = xj * q8 + (xj < r8 ? xj : r8), wend = wstart + q8 + (xj < r8 ? 1u : 0u);
    const unsigned wstride = (gridDim.x - xj + 7u) >> 3;
    for (unsigned lin = wstart + (blockIdx.x >> 3); lin < wend; lin += wstride) {
    const int tile = lin % tiles, b = lin / tiles;
    const int h0 = (tile / tilesx) * GMH, w0p = (tile % tilesx) * GMW;
    __syncthreads();  // weights staged / T of the previous tile consumed
    // Two N-tiles (64 pixel slots) per wave and iteration, and all fragments of up to 64 channels of both fetched before
    // the first product: 4-8 independent 16-byte loads per lane in flight instead of one per dependent MFMA.
    for (int nt0 = wave * 2; nt0 < GNT; nt0 += 8) {
        bool ok[2];
        long ps[2];
        int slot[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            slot[u] = (nt0 + u) * 32 + r;
            const int py = slot[u] / GPW, px = slot[u] - py * GPW;
            const int h = h0 + py - 1, w = w0p + px - 1;
            ok[u] = nt0 + u < GNT && slot[u] < GSLOTS && h >= 0 && h < H && w >= 0 && w < W;
            // clamped address, zeroed afterwards: no branch around the loads
            const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
            ps[u] = (long)(hc >> in.ups) * Ws + (wc >> in.ups);
        }
        g_float16 acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
        int kc = 0;
        for (int s = 0; s < in.nseg; ++s) {
            const View v = in.x[s];
            for (int cc = 0; cc < v.C; cc += 64) {
                g_short8 bf[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int c = min(cc + 16 * t, v.C - 16);
                        bf[u][t] = *(const g_short8*)((const bf16_t*)v.p + voff(v, b, ps[u], c + hk * 8));
                    }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (cc + 16 * t >= v.C) break;
                    const g_short8 afrag = *(const g_short8*)(A + r * astride + (kc + 16 * t + hk * 8) * 2);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const g_short8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, ok[u] ? bf[u][t] : z, acc[u], 0, 0, 0);
                    }
                }
                kc += min(64, v.C - cc);
            }
        }
        // acc[i] = T[row = (i&3) + 8*(i>>2) + 4*hk][slot]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (nt0 + u >= GNT) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hk;
                if (row < 27) T[slot[u] * GTROW + row] = acc[u][i];
            }
        }
    }
    __syncthreads();
    const int ty = tid / GMW, tx = tid % GMW;
    const int h = h0 + ty, w = w0p + tx;
    if (h < H && w < W) {
        float lg[3] = {b0[0], b1[0], b2[0]};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* t = T + ((ty + tap / 3) * GPW + tx + tap % 3) * GTROW + tap * 3;
            lg[0] += t[0];
            lg[1] += t[1];
            lg[2] += t[2];
        }
        float4 g;
        g.x = 2.0f / (1.0f + expf(-lg[0]));
        g.y = 2.0f / (1.0f + expf(-lg[1]));
        g.z = 2.0f / (1.0f + expf(-lg[2]));
        g.w = 0.f;
        ((float4*)gates)[((long)b * H + h) * W + w] = g;
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Segment-wise accumulation (sr_fbm_gates_accumulate): one segment (<= 64 channels), up to 8 consumer layers.
// The wave's activation fragments stay in registers while it loops over the consumers (A operand = that consumer's
// 27 x C weight slice); per consumer the T image goes through the same LDS buffer and the 9-tap sum is added to the
// consumer's logits (or finished as 2*sigmoid into the gates buffer).
// ---------------------------------------------------------------------------------------------
// running logits of one consumer: three floats per pixel, packed (12-byte loads and stores)
struct Logit3 {
    float x, y, z;
};

// KMAX = k-steps (16 channels each) this instance can hold: 2 for segments of up to 32 channels (half the fragment loads and
// registers of the general one), 4 for up to 64
template <int NC, int KMAX>
__global__ __launch_bounds__(256) void fbm_gates_acc_mfma(View seg, int B, int H, int W, GateConsList cl) {
    extern __shared__ __attribute__((aligned(16))) char glds[];
    float* T = (float*)glds;                                  // [GNT*32][27]
    char* A = glds + GNT * 32 * GTROW * sizeof(float);        // [consumer][32][C*2 + 16] bf16 rows, padded
    const int C = seg.C;
    const int astride = C * 2 + 16;
    const int tilesx = (W + GMW - 1) / GMW;
    const unsigned tiles = (unsigned)tilesx * ((H + GMH - 1) / GMH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hk = lane >> 5;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        const float* w0 = cl.c[ci].w[0];
        const float* w1 = cl.c[ci].w[1];
        const float* w2 = cl.c[ci].w[2];
        for (int it = tid; it < 32 * C; it += 256) {
            const int c = it % C, row = it / C;
            float v = 0.f;
            if (row < 27) {
                const float* wk = row % 3 == 0 ? w0 : (row % 3 == 1 ? w1 : w2);
                v = wk[(long)c * 9 + row / 3];
            }
            *(bf16_t*)(A + (ci * 32 + row) * astride + c * 2) = f32_to_bf16(v);
        }
    }
    const unsigned total = tiles * (unsigned)B;
    const unsigned xj = blockIdx.x & 7u, q8 = total >> 3, r8 = total & 7u;
    const unsigned wstart = xj * q8 + (xj < r8 ? xj : r8), wend = wstart + q8 + (xj < r8 ? 1u : 0u);
    const unsigned wstride = (gridDim.x - xj + 7u) >> 3;
    const int nks = C / 16;  // 1..4
    for (unsigned lin = wstart + (blockIdx.x >> 3); lin < wend; lin += wstride) {
        const int tile = lin % tiles, b = lin / tiles;
        const int h0 = (tile / tilesx) * GMH, w0p = (tile % tilesx) * GMW;
        // this wave's N-tiles: wave, wave + 4, wave + 8 (< GNT = 11); all their fragments are fetched once
        g_short8 frag[3][KMAX];
        bool ok[3];
        int slot[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int nt = wave + 4 * u;
            slot[u] = nt * 32 + r;
            const int py = slot[u] / GPW, px = slot[u] - py * GPW;
            const int h = h0 + py - 1, w = w0p + px - 1;
            ok[u] = nt < GNT && slot[u] < GSLOTS && h >= 0 && h < H && w >= 0 && w < W;
            const long ps = (long)min(max(h, 0), H - 1) * W + min(max(w, 0), W - 1);
#pragma unroll
            for (int t = 0; t < KMAX; ++t)
                frag[u][t] = *(const g_short8*)((const bf16_t*)seg.p + voff(seg, b, ps, min(16 * t, C - 16) + hk * 8));
        }
        const int ty = tid / GMW, tx = tid % GMW;
        const int h = h0 + ty, w = w0p + tx;
        // running logits of every consumer, fetched together with the fragments (unconditional, clamped: a load behind
        // a branch, or issued after the previous consumer's store, waits for the memory pipe to drain)
        const long pix = ((long)b * H + min(h, H - 1)) * W + min(w, W - 1);
        Logit3 lgp[NC];
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            // a consumer that starts from its bias reads (and ignores) 12 bytes of the segment itself: cached, always valid
            const char* src = cl.c[ci].first ? (const char*)((const bf16_t*)seg.p + voff(seg, b, pix - (long)b * H * W, 0))
                                             : (const char*)(cl.c[ci].logits + 3 * pix);
            lgp[ci] = *(const Logit3*)src;
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            __syncthreads();  // weights staged / T of the previous consumer (or tile) consumed
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (wave + 4 * u >= GNT) continue;
                g_float16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int t = 0; t < KMAX; ++t) {
                    if (t >= nks) break;
                    const g_short8 afrag = *(const g_short8*)(A + (ci * 32 + r) * astride + (16 * t + hk * 8) * 2);
                    const g_short8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, ok[u] ? frag[u][t] : z, acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * hk;
                    if (row < 27) T[slot[u] * GTROW + row] = acc[i];
                }
            }
            __syncthreads();
            if (h < H && w < W) {
                const GateCons& gc = cl.c[ci];
                float lg[3];
                lg[0] = gc.first ? gc.b[0][0] : lgp[ci].x;
                lg[1] = gc.first ? gc.b[1][0] : lgp[ci].y;
                lg[2] = gc.first ? gc.b[2][0] : lgp[ci].z;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float* t = T + ((ty + tap / 3) * GPW + tx + tap % 3) * GTROW + tap * 3;
                    lg[0] += t[0];
                    lg[1] += t[1];
                    lg[2] += t[2];
                }
                float4 g;
                if (gc.gates) {
                    g.x = 2.0f / (1.0f + expf(-lg[0]));
                    g.y = 2.0f / (1.0f + expf(-lg[1]));
                    g.z = 2.0f / (1.0f + expf(-lg[2]));
                    g.w = 0.f;
                    ((float4*)gc.gates)[pix] = g;
                } else {
                    ((Logit3*)gc.logits)[pix] = Logit3{lg[0], lg[1], lg[2]};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Direct form of the same accumulation (32- and 64-channel segments): D[4 * consumer + gate][pixel] as ONE GEMM over
// K = 9 taps x C channels, all consumers at once.  The haloed activation tile (10 x 34 slots) is staged in LDS once per
// tile and the B fragments are read at the 9 shifted positions; the weight rows (4 per consumer, the 4th zero) sit in
// LDS.  No T image, no per-consumer barrier: two barriers per tile.  Each wave owns two pixel rows of the 8 x 32 tile.
// Lane (n = lane & 31, hk = lane >> 5) ends up with the three logits of pixel n for the consumers 2 j + hk.
// ---------------------------------------------------------------------------------------------
template <int NC, int CK>
__global__ __launch_bounds__(256) void fbm_gates_acc_direct(View seg, int B, int H, int W, GateConsList cl) {
    constexpr int C = 16 * CK;
    constexpr int XS = C * 2 + 16;       // bytes per slot of the activation tile (padded: conflict-free 16-byte reads)
    constexpr int KA = 9 * C;            // K of a weight row
    constexpr int AS = KA * 2 + 16;      // bytes per weight row
    constexpr int NROW = 4 * NC;
    constexpr int CP = C / 8;            // 16-byte parts per slot
    constexpr int NCHUNK = GSLOTS * CP;
    constexpr int NPRE = (NCHUNK + 255) / 256;
    constexpr int NJ = (NC + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char glds[];
    char* X = glds;                // [GSLOTS][XS]
    char* A = glds + GSLOTS * XS;  // [NROW][AS]
    const int tilesx = (W + GMW - 1) / GMW;
    const unsigned tiles = (unsigned)tilesx * ((H + GMH - 1) / GMH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hk = lane >> 5;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        const float* w0 = cl.c[ci].w[0];
        const float* w1 = cl.c[ci].w[1];
        const float* w2 = cl.c[ci].w[2];
        for (int it = tid; it < 4 * KA; it += 256) {
            const int g = it / KA, k = it - g * KA;
            const int tap = k / C, c = k - tap * C;
            const float* wk = g == 0 ? w0 : (g == 1 ? w1 : w2);
            const float v = g < 3 ? wk[(long)c * 9 + tap] : 0.f;
            *(bf16_t*)(A + (ci * 4 + g) * AS + k * 2) = f32_to_bf16(v);
        }
    }
    const unsigned total = tiles * (unsigned)B;
    const unsigned xj = blockIdx.x & 7u, q8 = total >> 3, r8 = total & 7u;
    const unsigned wstart = xj * q8 + (xj < r8 ? xj : r8), wend = wstart + q8 + (xj < r8 ? 1u : 0u);
    const unsigned wstride = (gridDim.x - xj + 7u) >> 3;
    unsigned lin = wstart + (blockIdx.x >> 3);
    if (lin >= wend) return;  // block-uniform
    // chunk q of the haloed tile -> (slot, part); loads are clamped into the image, the stash writes zeros outside it
    auto fetch = [&](unsigned l, uint4 (&pre)[NPRE]) {
        const int tile = l % tiles, b = l / tiles;
        const int h0 = (tile / tilesx) * GMH, w0p = (tile % tilesx) * GMW;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int q = min(tid + 256 * i, NCHUNK - 1);
            const int slot = q / CP, part = q - slot * CP;
            const int py = slot / GPW, px = slot - py * GPW;
            const int h = min(max(h0 + py - 1, 0), H - 1), w = min(max(w0p + px - 1, 0), W - 1);
            pre[i] = *(const uint4*)((const bf16_t*)seg.p + voff(seg, b, (long)h * W + w, part * 8));
        }
    };
    auto stash = [&](unsigned l, const uint4 (&pre)[NPRE]) {
        const int tile = l % tiles;
        const int h0 = (tile / tilesx) * GMH, w0p = (tile % tilesx) * GMW;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int q = tid + 256 * i;
            if (q >= NCHUNK) break;
            const int slot = q / CP, part = q - slot * CP;
            const int py = slot / GPW, px = slot - py * GPW;
            const int h = h0 + py - 1, w = w0p + px - 1;
            const bool in = h >= 0 && h < H && w >= 0 && w < W;
            *(uint4*)(X + slot * XS + part * 16) = in ? pre[i] : make_uint4(0, 0, 0, 0);
        }
    };
    uint4 pre[NPRE];
    fetch(lin, pre);
    const int ra = min(n, NROW - 1);
    for (; lin < wend; lin += wstride) {
        const int tile = lin % tiles, b = lin / tiles;
        const int h0 = (tile / tilesx) * GMH, w0p = (tile % tilesx) * GMW;
        __syncthreads();  // the previous tile's fragment reads are done (first tile: nothing pending)
        stash(lin, pre);
        // this tile's running logits first, then the next tile's activations (in-order counters: the logits are needed
        // first); all loads unconditional and clamped, issued before any store of this tile
        Logit3 lgp[2][NJ];
        long pix[2];
        bool inside[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int h = h0 + wave * 2 + p, w = w0p + n;
            inside[p] = h < H && w < W;
            const long pin = (long)min(h, H - 1) * W + min(w, W - 1);
            pix[p] = (long)b * H * W + pin;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const GateCons& g0 = cl.c[2 * j];
                const GateCons& g1 = cl.c[2 * j + 1 < NC ? 2 * j + 1 : 2 * j];
                const float* lp = hk ? g1.logits : g0.logits;
                const int first = hk ? g1.first : g0.first;
                // a consumer that starts from its bias reads (and ignores) 12 bytes of the segment itself: always valid
                const char* src = first ? (const char*)((const bf16_t*)seg.p + voff(seg, b, pin, 0)) : (const char*)(lp + 3 * pix[p]);
                lgp[p][j] = *(const Logit3*)src;
            }
        }
        fetch(min(lin + wstride, wend - 1), pre);  // past the end: any valid tile, dropped
        __syncthreads();
        g_float16 acc[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int ks = 0; ks < CK; ++ks) {
                const g_short8 afrag = *(const g_short8*)(A + ra * AS + (tap * C + ks * 16 + hk * 8) * 2);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int slot = (wave * 2 + p + tap / 3) * GPW + n + tap % 3;
                    const g_short8 bfrag = *(const g_short8*)(X + slot * XS + (ks * 16 + hk * 8) * 2);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[p], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const GateCons& g0 = cl.c[2 * j];
                const GateCons& g1 = cl.c[2 * j + 1 < NC ? 2 * j + 1 : 2 * j];
                const bool mine = inside[p] && (2 * j + 1 < NC || !hk);
                const int first = hk ? g1.first : g0.first;
                float lg[3];
                lg[0] = first ? (hk ? g1.b[0][0] : g0.b[0][0]) : lgp[p][j].x;
                lg[1] = first ? (hk ? g1.b[1][0] : g0.b[1][0]) : lgp[p][j].y;
                lg[2] = first ? (hk ? g1.b[2][0] : g0.b[2][0]) : lgp[p][j].z;
                lg[0] += acc[p][4 * j + 0];
                lg[1] += acc[p][4 * j + 1];
                lg[2] += acc[p][4 * j + 2];
                float* gp = hk ? g1.gates : g0.gates;
                float* lo = hk ? g1.logits : g0.logits;
                if (mine) {
                    if (gp) {
                        float4 g;
                        g.x = 2.0f / (1.0f + expf(-lg[0]));
                        g.y = 2.0f / (1.0f + expf(-lg[1]));
                        g.z = 2.0f / (1.0f + expf(-lg[2]));
                        g.w = 0.f;
                        ((float4*)gp)[pix[p]] = g;
                    } else {
                        ((Logit3*)lo)[pix[p]] = Logit3{lg[0], lg[1], lg[2]};
                    }
                }
            }
        }
    }
}

int g_gates_no_direct = 0;  // debug hook (A/B inside one process)

int fbm_gates_acc_launch(int B, int H, int W, View seg, const GateConsList& cl, hipStream_t st) {
    if (cl.n < 1 || cl.n > SR_MAX_GATE_CONSUMERS) return SR_ERR_ARG;
    if (seg.C % 16 || seg.C > 64 || !view_vec_ok(seg, 8)) return SR_NOT_ELIGIBLE;
    const long nblk = (long)cdiv(H, GMH) * cdiv(W, GMW) * B;
    if (nblk >= (1L << 31)) return SR_ERR_ARG;
    int ncu = 0;
    if (int rc = device_cus(&ncu)) return rc;
    // direct form: 64-channel segments (278 vs 405 us with five consumers at batch 16 of 512 x 512); at 32 channels the
    // T-image kernel below is as fast or faster (211 vs 246 us with four consumers), SR_AMD_GATES_DIRECT=1 forces it
    static const bool direct32 = getenv("SR_AMD_GATES_DIRECT") && atoi(getenv("SR_AMD_GATES_DIRECT")) == 1;
    // (up to five consumers, a dense block's worth: the variants for more would spill at 64 channels)
    if ((seg.C == 64 || (seg.C == 32 && direct32)) && cl.n <= 5 && !g_gates_no_direct) {
        const int ck = seg.C / 16;
        const size_t ldsd = (size_t)GSLOTS * (seg.C * 2 + 16) + (size_t)4 * cl.n * (9 * seg.C * 2 + 16);
        static int resident[16][SR_MAX_GATE_CONSUMERS + 1][2];  // persistent grid = resident workgroups, per device
        int& per_cu = resident[current_device_slot()][cl.n][ck == 4];
#define SR_GATES_DIRECT_V(NC, CK)                                                                                        \
    do {                                                                                                                 \
        const void* fn = (const void*)fbm_gates_acc_direct<NC, CK>;                                                      \
        if (!per_cu) {                                                                                                   \
            SR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));                      \
            SR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, ldsd));                                \
            per_cu = std::max(1, std::min(4, per_cu));                                                                   \
        }                                                                                                                \
        const dim3 grid((unsigned)std::min<long>(nblk, (long)per_cu * ncu));                                             \
        fbm_gates_acc_direct<NC, CK><<<grid, 256, ldsd, st>>>(seg, B, H, W, cl);                                         \
    } while (0)
#define SR_GATES_DIRECT(NC)                                                                                              \
    case NC:                                                                                                             \
        if (ck == 2) SR_GATES_DIRECT_V(NC, 2);                                                                           \
        else SR_GATES_DIRECT_V(NC, 4);                                                                                   \
        break;
        switch (cl.n) {
            SR_GATES_DIRECT(1) SR_GATES_DIRECT(2) SR_GATES_DIRECT(3) SR_GATES_DIRECT(4) SR_GATES_DIRECT(5)
        }
#undef SR_GATES_DIRECT
#undef SR_GATES_DIRECT_V
        SR_LAUNCH_CHECK();
        return SR_OK;
    }
    const size_t lds = (size_t)GNT * 32 * GTROW * sizeof(float) + (size_t)cl.n * 32 * (seg.C * 2 + 16);
    const long per_cu = std::max<long>(1, std::min<long>(4, (160 * 1024) / (long)lds));
    const dim3 grid((unsigned)std::min<long>(nblk, per_cu * ncu));
#define SR_GATES_ACC(NC)                                                                                                 \
    case NC:                                                                                                             \
        if (seg.C <= 32) {                                                                                               \
            SR_LDS_ATTR(lds, fbm_gates_acc_mfma<NC, 2>);                                                                 \
            fbm_gates_acc_mfma<NC, 2><<<grid, 256, lds, st>>>(seg, B, H, W, cl);                                         \
        } else {                                                                                                         \
            SR_LDS_ATTR(lds, fbm_gates_acc_mfma<NC, 4>);                                                                 \
            fbm_gates_acc_mfma<NC, 4><<<grid, 256, lds, st>>>(seg, B, H, W, cl);                                         \
        }                                                                                                                \
        break;
    switch (cl.n) {
        SR_GATES_ACC(1) SR_GATES_ACC(2) SR_GATES_ACC(3) SR_GATES_ACC(4) SR_GATES_ACC(5) SR_GATES_ACC(6) SR_GATES_ACC(7) SR_GATES_ACC(8)
    }
#undef SR_GATES_ACC
    SR_LAUNCH_CHECK();
    return SR_OK;
}

static bool gates_mfma_eligible(const Concat& in) {
    if (in.ctot > 1024) return false;
    for (int s = 0; s < in.nseg; ++s) {
        const View& v = in.x[s];
        if (v.C % 16 || !view_vec_ok(v, 8)) return false;
    }
    return true;
}

int g_force_generic_gates = 0;  // test hook

int fbm_gates_launch(int dtype, int B, int H, int W, const Concat& in, const sr_layer_params& lp, float* gates,
                     hipStream_t st) {
    if (dtype == SR_BF16 && !g_force_generic_gates && gates_mfma_eligible(in)) {
        const size_t lds = (size_t)GNT * 32 * GTROW * sizeof(float) + 32ul * (in.ctot * 2 + 16);
        SR_LDS_ATTR(lds, fbm_gates_mfma);
        const long nblk = (long)cdiv(H, GMH) * cdiv(W, GMW) * B;
        if (nblk >= (1L << 31)) return SR_ERR_ARG;
        int ncu = 0;
        if (int rc = device_cus(&ncu)) return rc;
        const long per_cu = std::max<long>(1, std::min<long>(4, (160 * 1024) / (long)lds));
        const unsigned grid = (unsigned)std::min<long>(nblk, per_cu * ncu);
        fbm_gates_mfma<<<dim3(grid), 256, lds, st>>>(
            in, B, H, W, lp.gate_w[0], lp.gate_w[1], lp.gate_w[2], lp.gate_b[0], lp.gate_b[1], lp.gate_b[2], gates);
        SR_LAUNCH_CHECK();
        return SR_OK;
    }
    dim3 grid(cdiv(H, GT) * cdiv(W, GT), B);
    if (dtype == SR_F32)
        fbm_gates_kernel<float><<<grid, 256, 0, st>>>(in, H, W, lp.gate_w[0], lp.gate_w[1], lp.gate_w[2], lp.gate_b[0],
                                                      lp.gate_b[1], lp.gate_b[2], gates);
    else
        fbm_gates_kernel<bf16_t><<<grid, 256, 0, st>>>(in, H, W, lp.gate_w[0], lp.gate_w[1], lp.gate_w[2], lp.gate_b[0],
                                                       lp.gate_b[1], lp.gate_b[2], gates);
    SR_LAUNCH_CHECK();
    return SR_OK;
}
}  // namespace sr

using namespace sr;

extern "C" int sr_fbm_gates(int dtype, int B, int H, int W, const sr_concat* in, const sr_layer_params* lp, float* gates,
                            void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && in && lp && gates);
    SR_REQUIRE(!in->ups || (H % 2 == 0 && W % 2 == 0));
    Concat c;
    const long npix = (long)H * W;
    int rc = to_concat(in, in->ups ? npix / 4 : npix, npix, &c);
    if (rc != SR_OK) return rc;
    SR_REQUIRE(c.ctot == lp->cin);
    for (int k = 0; k < 3; ++k) SR_REQUIRE(lp->gate_w[k] && lp->gate_b[k]);
    return fbm_gates_launch(dtype, B, H, W, c, *lp, gates, (hipStream_t)stream);
}

extern "C" int sr_fbm_gates_accumulate(int dtype, int B, int H, int W, sr_view seg, int nconsumers,
                                       const sr_gate_consumer* consumers, void* stream) {
    SR_REQUIRE(dtype == SR_BF16 && B > 0 && H > 0 && W > 0 && seg.p && consumers);
    SR_REQUIRE(nconsumers >= 1 && nconsumers <= SR_MAX_GATE_CONSUMERS && view_block_ok(seg) && seg.C > 0 && seg.c0 >= 0);
    GateConsList cl{};
    cl.n = nconsumers;
    for (int i = 0; i < nconsumers; ++i) {
        const sr_gate_consumer& c = consumers[i];
        for (int k = 0; k < 3; ++k) {
            SR_REQUIRE(c.w[k] && (c.b[k] || !c.first));
            cl.c[i].w[k] = c.w[k];
            cl.c[i].b[k] = c.b[k];
        }
        SR_REQUIRE(c.logits || (c.first && c.gates));
        cl.c[i].logits = c.logits;
        cl.c[i].gates = c.gates;
        cl.c[i].first = c.first ? 1 : 0;
    }
    const int rc = fbm_gates_acc_launch(B, H, W, to_view(seg, (long)H * W), cl, (hipStream_t)stream);
    return rc == SR_NOT_ELIGIBLE ? SR_ERR_ARG : rc;
}

extern "C" void sr_debug_force_generic_gates(int on) { sr::g_force_generic_gates = on; }
extern "C" void sr_debug_gates_no_direct(int on) { sr::g_gates_no_direct = on; }
