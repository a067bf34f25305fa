// FBM low-pass planes on the matrix cores (bf16 activations, H and W multiples of 32 up to 512).
//
// The three box masks keep few frequencies, so the masked FFT round trip is four small dense products per plane,
//     T   = x Qw            analysis along W   (K = W,      M = jw)      bf16 image rows -> x-basis coefficients
//     S   = Ph^T T          analysis along H   (K = H,      M = nh)      -> coefficients of both axes
//     U_k = P_k S'_k        synthesis along H  (K = nhk[k], M = H)       mask weights of band k, back to rows
//     L_k = U_k Qw_k^T      synthesis along W  (K = nw[k],  M = W)       -> the three bf16 planes
// in REAL orthonormal Fourier bases (q_0 = 1/sqrt(W); q_2c = sqrt(2/W) cos, q_2c+1 = sqrt(2/W) sin; index 1 is an
// all-zero dummy so that (cos, sin) pairs stay 2-aligned).  The VALU FFT of fbm_fast.hip is bound by vector
// instruction issue and by its fp32 complex intermediates; as v_mfma_f32_32x32x16_f16 GEMMs the arithmetic is a few
// per cent of the time and the kernels stream x, the fp16 intermediates and the three output planes.
//
// Kernel shape (both kernels): one persistent workgroup of 8 waves per CU.  Every wave keeps ITS slice of the basis
// (32 or 64 basis functions x all k) in 128 registers for the whole kernel, so the only global loads are the data
// rows of the NEXT group, issued before the matrix phase of the current one and parked in registers until the
// current image has been consumed (two LDS images, one barrier per group).  A wave's loads are therefore always
// ahead of its own stores in the in-order memory pipe -- the reason the first version of this path (basis streamed
// from L2 behind the output stores) could not overlap anything.
//
// Row masks of the reference are not symmetric in +-fy (nearest resize of the 64-row prototype).  With
// s_f = (m[+f] + m[-f]) / 2 and d_f = (m[+f] - m[-f]) / 2 the filtered coefficient columns A (cos c) and B (sin c)
// of a column frequency c > 0 are
//     A' = sum_n s_n SA_n p_n + sum_f d_f (SB_cos,f p_sin,f - SB_sin,f p_cos,f)
//     B' = sum_n s_n SB_n p_n + sum_f d_f (SA_sin,f p_cos,f - SA_cos,f p_sin,f)
// (SA, SB = y-coefficients of A, B).  The synthesis basis is shared by the three bands (it lives in registers), so
// the weights go on the data: y-coefficients below the first row with s < 1 enter with weight one straight from the
// staged image ("main" k-steps); the at most 32 rows after it (weights 1, 1/2, 0 and the partner terms) are rebuilt
// per band into a small "window" image, s S_own +- d S_partner, and multiplied with the same basis registers.
//
// Every product has the data operand [k][32 channels] (channels innermost in memory, as the tensors are).  Data tiles
// are staged row-major in LDS (plain 16-byte stores of what was loaded) and read through ds_read_b64_tr_b16, which
// hands each lane the 4 k-values of its channel: no software transpose.  D = data^T x basis^T puts 4 consecutive
// channels of one output row in a lane; v_permlane32_swap pairs lanes l / l+32 so that each lane stores 32
// contiguous bytes.
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "fbm_plan.h"

namespace sr {
namespace mf {

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s4;

constexpr int THREADS = 512;
// (one staged data image = K rows x G items x 32 channels x fp16: 64 KiB at 32 x 16 rows of 128 B or 16 x 16 rows of 256 B)
// Basis tables are stored fragment-major: the 64 x 8 halfs one wave needs for (row tile nt, k-step ks) are 1 KiB
// contiguous (lane = 32 * (k / 8 % 2) + row % 32).
__host__ __device__ inline size_t frag_index(int row, int k, int nks) {
    return ((size_t)((row >> 5) * nks + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
}

// Byte offset of 16-byte chunk ch of image row `row` (a row = G items x 64 bytes): XOR swizzles that keep both the
// 16-byte row stores and the transposed reads spread over the banks (programming guide T10, image (b)).
template <int G>
__device__ __forceinline__ int img_off(int row, int ch);
template <>
__device__ __forceinline__ int img_off<4>(int row, int ch) {
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
template <>
__device__ __forceinline__ int img_off<2>(int row, int ch) {
    return 128 * row + 16 * (ch ^ ((row & 3) << 1));
}

// Address of this lane's share of the transposed read that delivers, for item mt, the k-values
// 8 * (lane >> 5) + 4 * half + {0..3} of k-step 0: lane 4q + p of a 16-lane group supplies row q, columns 4p..4p+3.
template <int G>
__device__ __forceinline__ int tr_addr(int lane, int mt, int half) {
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 8 * (g4 >> 1) + 4 * half + q;
    return img_off<G>(row, mt * 4 + 2 * (g4 & 1) + (p >> 1)) + 8 * (p & 1);
}

__device__ __forceinline__ half8 tr_read(const char* img, int addr_lo, int addr_hi) {
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(img + addr_lo));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(img + addr_hi));
    union {
        struct {
            s4 a, b;
        } p;
        half8 v;
    } u;
    u.p.a = lo;
    u.p.b = hi;
    return u.v;
}

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    const __half2 h = __floats2half2_rn(a, b);
    return *(const uint32_t*)&h;
}
// fp16 intermediates saturate instead of overflowing to infinity (NaN passes through)
__device__ __forceinline__ uint32_t pack_h2_sat(float a, float b) {
    return pack_h2(fminf(fmaxf(a, -65504.f), 65504.f), fminf(fmaxf(b, -65504.f), 65504.f));
}
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) { return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }

// One 32 x 32 accumulator tile (rows = channels, column = this lane's output row) -> the 32 contiguous bytes of
// channels 16 * (lane >> 5) .. + 15 of that output row, as two uint4.
template <bool BF16>
__device__ __forceinline__ void tile_to_rows(const f32x16& acc, float scale, uint4& lo, uint4& hi) {
    uint32_t P[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = acc[4 * q] * scale, b = acc[4 * q + 1] * scale, c = acc[4 * q + 2] * scale, d = acc[4 * q + 3] * scale;
        P[q][0] = BF16 ? pack_bf2(a, b) : pack_h2_sat(a, b);
        P[q][1] = BF16 ? pack_bf2(c, d) : pack_h2_sat(c, d);
    }
    // lanes l < 32 hold channels 8q + {0..3}, lanes l + 32 channels 8q + 4 + {0..3}: after the swaps the low lanes own
    // channels 0..15 and the high lanes channels 16..31 of their row
    uint32_t r[4][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const auto s = __builtin_amdgcn_permlane32_swap(P[h][w], P[h + 2][w], false, false);
            r[h][w][0] = s[0];
            r[h][w][1] = s[1];
        }
    lo = make_uint4(r[0][0][0], r[0][1][0], r[0][0][1], r[0][1][1]);
    hi = make_uint4(r[1][0][0], r[1][1][0], r[1][0][1], r[1][1][1]);
}

__device__ __forceinline__ uint4 bf16x8_to_h16x8(uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = __uint_as_float(w[i] << 16), b = __uint_as_float(w[i] & 0xffff0000u);
        a = fminf(fmaxf(a, -65504.f), 65504.f);  // fp16 range (NaN passes through)
        b = fminf(fmaxf(b, -65504.f), 65504.f);
        o[i] = pack_h2(a, b);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

template <typename T>
__device__ __forceinline__ T sel3(int i, T a, T b, T c) {
    return i == 0 ? a : (i == 1 ? b : c);
}

__device__ __forceinline__ half8 zero_h8() {
    half8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (h16)0.f;
    return z;
}

// Thread index rebuilt from the lane counter and the (scalar) wave index.  Volatile: the per-thread addresses derived
// from it are then recomputed where they are used instead of being hoisted out of the group loop and spilled (a scratch
// reload queues behind the output stores like any other memory operation).
__device__ __forceinline__ int fresh_tid(int wave) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return wave * 64 + l;
}

__device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// analysis: out[item][m] = sum_k data[item][k] basis[m][k];  K <= 512, M <= 256: one 32-function basis tile per wave,
// two items per staged image (K rows x 128 bytes)
// ---------------------------------------------------------------------------------------------------------------
struct AnaArgs {
    View x;           // FROM_VIEW: bf16 source tensor; item = image row, k = pixel of the row
    int ups, Ws;      // nearest-upsampled source: pixel (row >> ups, k >> ups), source row pitch Ws
    const h16* src;   // otherwise fp16 rows of 32 channels at src + bcg * src_b + k * src_k + item * src_i
    long src_b;
    int src_k, src_i;
    h16* dst;         // dst + bcg * dst_b + m * dst_m + item * dst_i
    long dst_b;
    int dst_m, dst_i;
    const h16* basis; // fragment-major [pad32(M)][K]
    int K, M, NB, CG, nitems;
    int tile0;        // first basis tile of this launch (waves take tiles tile0 .. tile0 + 7; M > 256 takes two launches)
    float scale;
    int xtile1;       // KS = 34 only: 1 + index of one more basis tile with at most 16 valid functions, served from LDS by
                      // waves 0 and 1 in the same launch (0 = none); see mf_analysis
    float* pooled;    // plane mean = DC coefficient * pool_scale (analysis along H only)
    int pooled_ld;
    float pool_scale;
};

// KS = k-steps of 16 (32: K <= 512, the power-of-two trunk; 34: K <= 544, the 522 / 532-pixel windows of a tiled image).
template <bool FROM_VIEW, int KS>
__global__ __launch_bounds__(THREADS) void mf_analysis(AnaArgs a) {
    constexpr int G = 2, ROWB = 128;
    constexpr int IMG = KS * 16 * ROWB;   // bytes of one staged image
    constexpr int NR = (KS * 16 + 63) / 64;  // loader rows per thread
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // M is a multiple of 32 (host).  Waves beyond the last basis tile repeat it (same values to the same addresses):
    // with no predicate around the stores their number per group is a constant, which lets the compiler wait for the
    // prefetched loads only (vmcnt = stores issued since) instead of draining the stores at the end of every group.
    const int NT = a.M >> 5, nks = (a.K + 15) >> 4;
    const int tw = min(a.tile0 + wave, NT - 1);
    half8 bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        bq[ks] = ks < nks ? *(const half8*)(a.basis + ((size_t)(tw * nks + ks) * 64 + lane) * 8) : zero_h8();
    int ta[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        ta[g][0] = tr_addr<G>(lane, g, 0);
        ta[g][1] = tr_addr<G>(lane, g, 1);
    }
    // Ninth basis tile of the 522 / 532-pixel sides (262 / 266 functions: 8 tiles in the waves' registers + 6 / 10 functions):
    // a second launch for it streamed the whole source again.  Its at most 16 distinct functions (the rest of the tile
    // repeats the last one) take 512 bytes per k-step in the 24 KiB of LDS the two staged images leave free; after their own
    // tile, waves 0 and 1 multiply one item each by it (lanes of functions 16..31 read function 15's fragment: zeros, as
    // the table rows beyond its last function are).
    const bool xt = KS == 34 && a.xtile1 > 0;
    char* const xbasis = lds + 2 * IMG;
    if (xt) {
        for (int ks = wave; ks < KS; ks += THREADS / 64)
            if ((lane & 31) < 16)
                *(half8*)(xbasis + ((ks * 2 + (lane >> 5)) * 16 + (lane & 31)) * 16) =
                    ks < nks ? *(const half8*)(a.basis + ((size_t)((a.xtile1 - 1) * nks + ks) * 64 + lane) * 8) : zero_h8();
    }
    const int gpb = (a.nitems + G - 1) / G, ngroups = a.NB * gpb;
    const int ch = tid & 7, r0 = tid >> 3, oct = ch & 3;  // loader: rows r0 + 64 i, 16-byte chunk ch (item ch >> 2)
    auto fetch = [&](int grp, uint4 (&pre)[NR]) {
        const int bcg = grp / gpb, i0 = (grp - bcg * gpb) * G;
        const int item = min(i0 + (ch >> 2), a.nitems - 1);
        if (FROM_VIEW) {
            const int b = bcg / a.CG, cg = bcg - b * a.CG;
            const bf16_t* xb = (const bf16_t*)a.x.p + (long)b * a.x.bs;
            const unsigned rowpix = (unsigned)(item >> a.ups) * a.Ws;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int k = min(r0 + 64 * i, a.K - 1);
                pre[i] = *(const uint4*)(xb + voff32(a.x, rowpix + (k >> a.ups), cg * 32 + oct * 8));
            }
        } else {
            const h16* sb = a.src + (long)bcg * a.src_b + (long)item * a.src_i + oct * 8;
#pragma unroll
            for (int i = 0; i < NR; ++i) pre[i] = *(const uint4*)(sb + (long)min(r0 + 64 * i, a.K - 1) * a.src_k);
        }
    };
    auto stash = [&](const uint4 (&pre)[NR], char* img) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = r0 + 64 * i;
            if (row < a.K) *(uint4*)(img + img_off<G>(row, ch)) = FROM_VIEW ? bf16x8_to_h16x8(pre[i]) : pre[i];
        }
    };

    int grp = blockIdx.x;
    if (grp >= ngroups) return;
    uint4 pre[NR];
    fetch(grp, pre);
    // rows K .. 16 KS - 1 of both images stay zero for ever: the k loop below always runs its KS steps (a branch per
    // k-step costs a copy of the accumulators at every join), the basis registers of the missing steps are zero as well
    for (int i = tid; i < 2 * IMG / 16; i += THREADS) ((uint4*)lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    stash(pre, lds);
    __syncthreads();
    for (unsigned it = 0;; ++it) {
        const int ngrp = grp + gridDim.x;
        const bool more = ngrp < ngroups;
        fetch(more ? ngrp : grp, pre);  // unconditional: a branch around a load drains the memory pipe at the join
        asm volatile("" ::: "memory");  // the loads stay ahead of this group's stores
        const char* img = lds + (it & 1) * IMG;
        {
            const int bcg = grp / gpb, i0 = (grp - bcg * gpb) * G;
            f32x16 acc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) zero_acc(acc[g]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const char* base = img + ks * 16 * ROWB;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr_read(base, ta[g][0], ta[g][1]), bq[ks], acc[g], 0, 0, 0);
            }
            const int m = tw * 32 + (lane & 31), hk = lane >> 5;
            uint4 lo[G], hi[G];
#pragma unroll
            for (int g = 0; g < G; ++g) tile_to_rows<false>(acc[g], a.scale, lo[g], hi[g]);
            // The next image is staged BEFORE this group's stores are issued: the wait for the prefetched loads then has
            // nothing younger than them in the memory pipe (the compiler cannot assume that loads return ahead of younger
            // stores, it would drain every store here), and the stores drain under the next group's matrix phase.
            if (more) stash(pre, lds + ((it + 1) & 1) * IMG);
            asm volatile("" ::: "memory");
            if (a.pooled && i0 == 0 && wave == 0 && m == 0) {  // mean = <x, q_0 p_0> / sqrt(HW)
                const int b = bcg / a.CG, cg = bcg - b * a.CG;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    a.pooled[(long)b * a.pooled_ld + cg * 32 + (i & 3) + 8 * (i >> 2) + 4 * hk] = acc[0][i] * a.pool_scale;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // an item beyond the last one repeats it (the loader clamped the same way): same values, same address
                uint4* d = (uint4*)(a.dst + (long)bcg * a.dst_b + (long)m * a.dst_m + (long)min(i0 + g, a.nitems - 1) * a.dst_i + hk * 16);
                d[0] = lo[g];
                d[1] = hi[g];
            }
            if (KS == 34 && xt && wave < G) {  // wave-uniform
                const int g = wave;
                const int t0a = tr_addr<G>(lane, g, 0), t1a = tr_addr<G>(lane, g, 1);
                const char* xl = xbasis + ((lane >> 5) * 16 + min(lane & 31, 15)) * 16;
                f32x16 ax;
                zero_acc(ax);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr_read(img + ks * 16 * ROWB, t0a, t1a), *(const half8*)(xl + ks * 512), ax, 0, 0, 0);
                uint4 xlo, xhi;
                tile_to_rows<false>(ax, a.scale, xlo, xhi);
                const int mx = (a.xtile1 - 1) * 32 + (lane & 31);
                uint4* d = (uint4*)(a.dst + (long)bcg * a.dst_b + (long)mx * a.dst_m + (long)min(i0 + g, a.nitems - 1) * a.dst_i + hk * 16);
                d[0] = xlo;
                d[1] = xhi;
            }
        }
        if (!more) break;
        __syncthreads();
        grp = ngrp;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// synthesis: out[item][m] = sum_k data[item][k] basis[m][k];  K <= 256, M <= 512: two 32-function basis tiles per
// wave, four items per staged image (K rows x 256 bytes), three bands per item group
// ---------------------------------------------------------------------------------------------------------------
// U_k (the coefficient rows between the two synthesis passes) is stored by groups of four coefficient columns,
// U_k[bcg][j / 4][h][j % 4][32]: the H-synthesis writes, for its four columns, 256 contiguous bytes per row h and whole
// kilobytes per wave (column-major rows of 64 bytes were half-line writes 16 KiB apart); the W-synthesis reads 1 KiB runs.
// `base` already points at (bcg, row h, channel octet); k = coefficient column.
__device__ __forceinline__ const h16* u_addr(const h16* base, int k, int H) { return base + ((long)(k >> 2) * H) * 128 + (k & 3) * 32; }

constexpr int WIN_ROWS = 32;                   // window rows per band (two k-steps)
constexpr int WIN_B = 3 * WIN_ROWS * 256;      // 24 KiB
struct SynArgs {
    const h16* src[3];  // fp16 rows of 32 channels at src + bcg * src_b + k * src_k + item * src_i
    long src_b[3];
    int src_k[3], src_i[3];
    int K[3];           // data rows of band k (SHARED: K[0] rows serve all bands)
    int ksm[3];         // k-steps of band k taken straight from the staged image
    int nwin[3];        // SHARED: window k-steps after them (0..2)
    const float2* win;  // SHARED: [3][32] weights (s, d) of the window rows
    h16* dst[3];        // SHARED: dst + bcg * dst_b + m * dst_m + item * dst_i
    long dst_b[3];
    int dst_m[3], dst_i[3];
    View out[3];        // !SHARED: bf16 planes, pixel = item * Wout + m
    int Wout;
    int nitems[3];      // valid items of band k (nested, nitems[0] the largest)
    int M, NB, CG;      // M: output rows rounded up to whole basis tiles
    int Mv;             // valid output rows (rows Mv .. M - 1 of the basis table repeat row Mv - 1 and store to its address)
    int tile0;          // first basis tile of this launch (waves take tiles tile0 .. tile0 + 15; M > 512 takes two launches)
    const h16* basis;   // fragment-major [pad32(M)][16 * nks_tab]
    int nks_tab;
    int xtile;          // !SHARED: one more basis tile (index) served from LDS by waves 0..3, or -1 (see mf_synthesis)
    int bmask;          // bit k set = band k is produced
    int bfirst;         // first wanted band
    int bnext[3];       // wanted band after band k, or -1
};

// SHARED = true: synthesis along H (one staged coefficient image per item group, mask weights through the window rows,
// fp16 out);  false: synthesis along W (one image per band, bf16 views out).
// KS = k-steps of 16 (16: K <= 256; 17: K <= 272, the 522 / 532-pixel windows of a tiled image -- with the window image
// that is exactly the 160 KiB of LDS).
// SOLO (synthesis along H of the 522 / 532-row windows): the launch serves ONE basis tile, a.tile0 -- the 17th, which the
// 16 tiles in the registers of the first launch leave over.  Wave g < 4 multiplies item g of every group by it and waves
// 4..7 only stage; without it all eight waves computed that tile twice over (clamped tile indices), i.e. the second
// launch repeated the MFMA work of the first for 20 of 532 rows.
template <bool SHARED, int KS, bool SOLO = false>
__global__ __launch_bounds__(THREADS) void mf_synthesis(SynArgs a) {
    constexpr int G = 4, ROWB = 256;
    constexpr int IMG = KS * 16 * ROWB;        // bytes of one staged image
    constexpr int NR = (KS * 16 + 31) / 32;    // loader rows per thread
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* winimg = lds + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // M is a multiple of 32 (host); tiles beyond the last one repeat it, items beyond the last one repeat it (the loader
    // clamps the same way): no predicate around the stores, see mf_analysis
    const int NT = a.M >> 5;
    constexpr int NJ = SOLO ? 1 : 2;  // basis tiles per wave
    const int tj[2] = {SOLO ? min(a.tile0, NT - 1) : min(a.tile0 + 2 * wave, NT - 1), min(a.tile0 + 2 * wave + 1, NT - 1)};
    half8 bq[KS][NJ];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            bq[ks][j] = ks < a.nks_tab ? *(const half8*)(a.basis + ((size_t)(tj[j] * a.nks_tab + ks) * 64 + lane) * 8) : zero_h8();
    // !SHARED with 17 basis tiles (the 522 / 532-pixel rows of a tiled image: 544 columns, 16 tiles in the waves' registers):
    // the 17th tile's fragments live in the window-image region of LDS, which the W-synthesis does not use, and waves 0..3
    // multiply one item each by it after their own tiles -- a second launch for that tile streamed all of U again.
    const bool xt = !SHARED && a.xtile >= 0;
    if (xt) {
        for (int ks = wave; ks < KS; ks += THREADS / 64)
            *(half8*)(winimg + (ks * 64 + lane) * 16) =
                ks < a.nks_tab ? *(const half8*)(a.basis + ((size_t)(a.xtile * a.nks_tab + ks) * 64 + lane) * 8) : zero_h8();
    }
    // window weights (s, d) of this thread's window row (tid >> 4), one pair per band: kept in registers
    float2 wts[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) wts[k] = SHARED ? a.win[k * WIN_ROWS + (tid >> 4)] : make_float2(0.f, 0.f);
    // SHARED: the narrower bands exist for the first item groups only (3, 2 or 1 bands of work per group); with the sample
    // index fastest a workgroup's groups (grp = blockIdx + n * gridDim) are spread over the whole item range
    // (groups cover the items of the widest WANTED band: the first band that runs in a group stages the next image)
    const int gpb = (sel3(a.bfirst, a.nitems[0], a.nitems[1], a.nitems[2]) + G - 1) / G, ngroups = a.NB * gpb;
    // loader: rows r0 + 32 i, 16-byte chunk ch (item ch >> 2); tl = fresh_tid()
    auto fetch = [&](int tl, int grp, int band, uint4 (&pre)[NR]) {
        const int ch = tl & 15, r0 = tl >> 4, oct = ch & 3;
        const int bcg = SHARED ? grp % a.NB : grp / gpb, i0 = (SHARED ? grp / a.NB : grp - bcg * gpb) * G;
        const int K = sel3(band, a.K[0], a.K[1], a.K[2]);
        const int item = min(i0 + (ch >> 2), sel3(band, a.nitems[0], a.nitems[1], a.nitems[2]) - 1);
        const h16* sb = sel3(band, a.src[0], a.src[1], a.src[2]) + (long)bcg * sel3(band, a.src_b[0], a.src_b[1], a.src_b[2]) +
                        (long)item * sel3(band, a.src_i[0], a.src_i[1], a.src_i[2]) + oct * 8;
        const int sk = sel3(band, a.src_k[0], a.src_k[1], a.src_k[2]);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int k = min(r0 + 32 * i, K - 1);
            pre[i] = *(const uint4*)(SHARED ? sb + (long)k * sk : u_addr(sb, k, a.nitems[0]));
        }
    };
    // all 16 KS rows are written (zeros from row K on): the k loop always runs its steps -- a branch per k-step costs a
    // copy of the accumulators at every join -- and the basis registers are shared by bands of different K
    auto stash = [&](int tl, const uint4 (&pre)[NR], char* img, int band) {
        const int ch = tl & 15, r0 = tl >> 4;
        const int K = sel3(band, a.K[0], a.K[1], a.K[2]);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = r0 + 32 * i;
            if (row < KS * 16) *(uint4*)(img + img_off<G>(row, ch)) = row < K ? pre[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };

    const int band_first = SHARED ? 0 : a.bfirst;
    int grp = blockIdx.x, band = band_first;
    if (grp >= ngroups) return;
    uint4 pre[NR];
    fetch(tid, grp, band, pre);
    stash(tid, pre, lds, band);
    __syncthreads();
    for (unsigned it = 0;; ++it) {
        int ngrp = grp, nband = SHARED ? -1 : sel3(band, a.bnext[0], a.bnext[1], a.bnext[2]);
        if (nband < 0) {
            nband = band_first;
            ngrp = grp + gridDim.x;
        }
        const bool more = ngrp < ngroups;
        fetch(fresh_tid(wave), more ? ngrp : grp, more ? nband : band, pre);  // unconditional (see mf_analysis)
        asm volatile("" ::: "memory");
        const char* img = lds + (it & 1) * IMG;
        const int bcg = SHARED ? grp % a.NB : grp / gpb, i0 = (SHARED ? grp / a.NB : grp - bcg * gpb) * G;
        const int b = bcg / a.CG, cg = bcg - b * a.CG;
        const int hk = lane >> 5;

        if (SHARED) {
            // window rows of the three bands: s * own + (+-d) * partner (the other function of the frequency pair, taken
            // from the partner column j ^ 1 = item ^ 1); thread -> (row tid >> 4, chunk tid & 15) of every band
            const int tl = fresh_tid(wave);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = tl >> 4, ch = tl & 15;
                const int n = 16 * a.ksm[k] + r;
                const float2 w = sel3(k, wts[0], wts[1], wts[2]);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (r < 16 * a.nwin[k] && n < a.K[0] && (w.x != 0.f || w.y != 0.f)) {
                    const uint4 own = *(const uint4*)(img + img_off<G>(n, ch));
                    const uint4 par = *(const uint4*)(img + img_off<G>(n ^ 1, ch ^ 4));
                    const int mt = ch >> 2;
                    const bool dc = i0 + mt < 2;  // DC column (and its dummy): the reference averages the mask there
                    const float cp = dc ? 0.f : ((((n ^ mt) & 1) == 1) ? w.y : -w.y);
                    const uint32_t o[4] = {own.x, own.y, own.z, own.w}, p[4] = {par.x, par.y, par.z, par.w};
                    uint32_t q[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float2 fo = __half22float2(*(const __half2*)&o[i]), fp = __half22float2(*(const __half2*)&p[i]);
                        q[i] = pack_h2(w.x * fo.x + cp * fp.x, w.x * fo.y + cp * fp.y);
                    }
                    v = make_uint4(q[0], q[1], q[2], q[3]);
                }
                if (r < 16 * a.nwin[k]) *(uint4*)(winimg + img_off<G>(k * WIN_ROWS + r, ch)) = v;
            }
            __syncthreads();
        }

        {
            bool staged = false;  // band 0 of a group always runs (i0 < nitems[0])
#pragma unroll 1
            for (int k = SHARED ? 0 : band; k < (SHARED ? 3 : band + 1); ++k) {
                const int nit = sel3(k, a.nitems[0], a.nitems[1], a.nitems[2]);
                if (i0 >= nit || !((a.bmask >> k) & 1)) continue;
                const int ksm = sel3(k, a.ksm[0], a.ksm[1], a.ksm[2]);
                const int kend = ksm + (SHARED ? sel3(k, a.nwin[0], a.nwin[1], a.nwin[2]) : 0);
                const char* wbase = winimg + (k * WIN_ROWS - 16 * ksm) * ROWB;  // window row of k-step ks: k * 32 + 16 (ks - ksm)
                auto store = [&](int g, int j, const uint4& lo, const uint4& hi) {
                    // SHARED: the staged items are those of band 0, a narrower band drops the ones it does not have
                    if (SHARED && i0 + g >= nit) return;
                    const int ig = min(i0 + g, nit - 1), m = min(tj[j] * 32 + (lane & 31), a.Mv - 1);
                    uint4* d;
                    if (SHARED) {
                        d = (uint4*)(sel3(k, a.dst[0], a.dst[1], a.dst[2]) + (long)bcg * sel3(k, a.dst_b[0], a.dst_b[1], a.dst_b[2]) +
                                     ((long)(ig >> 2) * a.Mv + m) * 128 + (ig & 3) * 32 + hk * 16);
                    } else {
                        const View o = sel3(k, a.out[0], a.out[1], a.out[2]);
                        d = (uint4*)((bf16_t*)o.p + (long)b * o.bs + voff32(o, (unsigned)ig * a.Wout + m, cg * 32 + hk * 16));
                    }
                    d[0] = lo;
                    d[1] = hi;
                };
                uint4 held[2][NJ][2];  // packed rows of items 0 and 1, kept until the next image has been staged
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (!SOLO || wave == g) {  // wave-uniform
                        const int ta0 = tr_addr<G>(lane, g, 0), ta1 = tr_addr<G>(lane, g, 1);
                        // k-steps 0 .. kend - 1: the loop leaves early instead of branching around every step (an exit edge
                        // keeps the accumulators where they are, a skipped step costs a copy of them at the join)
                        f32x16 acc[NJ];
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            if (ks && ks >= kend) break;
                            const char* base = img + ks * 16 * ROWB;
                            if (SHARED) base = ks < ksm ? base : wbase + ks * 16 * ROWB;
                            const half8 d = tr_read(base, ta0, ta1);
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
                                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d, bq[ks][j], ks == 0 ? z : acc[j], 0, 0, 0);
                        }
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            uint4 lo, hi;
                            tile_to_rows<!SHARED>(acc[j], 1.0f, lo, hi);
                            if (g < 2) {
                                held[g][j][0] = lo;
                                held[g][j][1] = hi;
                            } else {
                                store(g, j, lo, hi);
                            }
                        }
                    }
                    if (g == 1) {
                        // stage the next image before the first store of this group goes out (see mf_analysis)
                        if (!staged && more) stash(fresh_tid(wave), pre, lds + ((it + 1) & 1) * IMG, nband);
                        staged = true;
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg)
                            if (!SOLO || wave == gg) {
#pragma unroll
                                for (int j = 0; j < NJ; ++j) store(gg, j, held[gg][j][0], held[gg][j][1]);
                            }
                    }
                }
                if (xt && wave < G) {  // wave-uniform: item `wave` of this group times the tile kept in LDS
                    const int g = wave;
                    const int ta0 = tr_addr<G>(lane, g, 0), ta1 = tr_addr<G>(lane, g, 1);
                    f32x16 ax;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks && ks >= kend) break;
                        const half8 d = tr_read(img + ks * 16 * ROWB, ta0, ta1);
                        const half8 bx = *(const half8*)(winimg + (ks * 64 + lane) * 16);
                        if (ks == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(d, bx, z, 0, 0, 0);
                        } else {
                            ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(d, bx, ax, 0, 0, 0);
                        }
                    }
                    uint4 lo, hi;
                    tile_to_rows<!SHARED>(ax, 1.0f, lo, hi);
                    const int ig = min(i0 + g, nit - 1), m = min(a.xtile * 32 + (lane & 31), a.Mv - 1);
                    const View o = sel3(k, a.out[0], a.out[1], a.out[2]);
                    uint4* dq = (uint4*)((bf16_t*)o.p + (long)b * o.bs + voff32(o, (unsigned)ig * a.Wout + m, cg * 32 + hk * 16));
                    dq[0] = lo;
                    dq[1] = hi;
                }
            }
        }
        if (!more) break;
        __syncthreads();
        grp = ngrp;
        band = nband;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// synthesis along W, folded (W a multiple of 64): the cos functions are even and the sin functions odd about w = 0, so
//     L[w] = E[w] + O[w],   L[W - w] = E[w] - O[w]      (E = sum over the cos coefficients, O = over the sin ones)
// for w = 0 .. W/2 - 1: two products of half the depth for half the columns -- half the MFMA work of mf_synthesis<false>
// and half the basis registers.  The staged image holds the cos coefficients (even j) in rows 0..127 and the sin
// coefficients in rows 128..255.  The column w = W/2 (where every sin vanishes) is one more basis tile, run by wave 0.
// Basis table: tiles 0 .. W/64 - 1 = columns w (k-steps 0..7 cos, 8..15 sin), tile W/64 = column W/2 in its row 0.
// ---------------------------------------------------------------------------------------------------------------
// KH = k-steps per half (8: up to 128 cos + 128 sin coefficients; 4 / 2 / 1 for short rows, whose staged image, loads and
// basis registers shrink with it -- at W = 64 the 8-step instance moved 64 KiB per item group for 2 KiB of coefficients).
template <int KH>
__global__ __launch_bounds__(THREADS) void mf_synthesis_wf(SynArgs a) {
    constexpr int G = 4, ROWB = 256, SIN0 = 16 * KH, IMG_W = 32 * KH * ROWB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTh = a.M >> 6;  // column tiles of the first half
    const int tw = min(wave, NTh - 1);
    half8 bE[KH], bO[KH], bN[KH];
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
        bE[ks] = *(const half8*)(a.basis + ((size_t)(tw * 16 + ks) * 64 + lane) * 8);
        bO[ks] = *(const half8*)(a.basis + ((size_t)(tw * 16 + 8 + ks) * 64 + lane) * 8);  // table: 8 cos + 8 sin steps per tile
        bN[ks] = *(const half8*)(a.basis + ((size_t)(NTh * 16 + ks) * 64 + lane) * 8);
    }
    const int gpb = (a.nitems[0] + G - 1) / G, ngroups = a.NB * gpb;
    auto fetch = [&](int tl, int grp, int band, uint4 (&pre)[KH]) {
        const int ch = tl & 15, r0 = tl >> 4, oct = ch & 3;
        const int bcg = grp / gpb, i0 = (grp - bcg * gpb) * G;
        const int K = sel3(band, a.K[0], a.K[1], a.K[2]);
        const int item = min(i0 + (ch >> 2), sel3(band, a.nitems[0], a.nitems[1], a.nitems[2]) - 1);
        const h16* sb = sel3(band, a.src[0], a.src[1], a.src[2]) + (long)bcg * sel3(band, a.src_b[0], a.src_b[1], a.src_b[2]) +
                        (long)item * sel3(band, a.src_i[0], a.src_i[1], a.src_i[2]) + oct * 8;
        const int sk = sel3(band, a.src_k[0], a.src_k[1], a.src_k[2]);
        (void)sk;
#pragma unroll
        for (int i = 0; i < KH; ++i) pre[i] = *(const uint4*)u_addr(sb, min(r0 + 32 * i, K - 1), a.nitems[0]);
    };
    // coefficient j = r0 + 32 i -> image row (j >> 1) + SIN0 (j & 1); all 32 KH rows are written (zeros from j = K on)
    auto stash = [&](int tl, const uint4 (&pre)[KH], char* img, int band) {
        const int ch = tl & 15, r0 = tl >> 4;
        const int K = sel3(band, a.K[0], a.K[1], a.K[2]);
#pragma unroll
        for (int i = 0; i < KH; ++i) {
            const int row = (r0 >> 1) + 16 * i + (r0 & 1) * SIN0;
            *(uint4*)(img + img_off<G>(row, ch)) = r0 + 32 * i < K ? pre[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };

    int grp = blockIdx.x, band = a.bfirst;
    if (grp >= ngroups) return;
    uint4 pre[KH];
    fetch(tid, grp, band, pre);
    stash(tid, pre, lds, band);
    __syncthreads();
    for (unsigned it = 0;; ++it) {
        int ngrp = grp, nband = sel3(band, a.bnext[0], a.bnext[1], a.bnext[2]);
        if (nband < 0) {
            nband = a.bfirst;
            ngrp = grp + gridDim.x;
        }
        const bool more = ngrp < ngroups;
        fetch(fresh_tid(wave), more ? ngrp : grp, more ? nband : band, pre);  // unconditional (see mf_analysis)
        asm volatile("" ::: "memory");
        const char* img = lds + (it & 1) * IMG_W;
        const int bcg = grp / gpb, i0 = (grp - bcg * gpb) * G;
        const int b = bcg / a.CG, cg = bcg - b * a.CG;
        const int hk = lane >> 5;
        const int nit = sel3(band, a.nitems[0], a.nitems[1], a.nitems[2]);
        const int kend = sel3(band, a.ksm[0], a.ksm[1], a.ksm[2]);  // k-steps per half
        const View o = sel3(band, a.out[0], a.out[1], a.out[2]);
        bf16_t* ob = (bf16_t*)o.p + (long)b * o.bs;
        const int m = tw * 32 + (lane & 31);
        auto store = [&](int g, int col, const uint4& lo, const uint4& hi) {
            uint4* d = (uint4*)(ob + voff32(o, (unsigned)min(i0 + g, nit - 1) * a.Wout + col, cg * 32 + hk * 16));
            d[0] = lo;
            d[1] = hi;
        };
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto nyquist = [&](int g) {  // wave 0: the column W / 2 of item g
            const int ta0 = tr_addr<G>(lane, g, 0), ta1 = tr_addr<G>(lane, g, 1);
            f32x16 accN;
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) {
                if (ks && ks >= kend) break;
                accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr_read(img + ks * 16 * ROWB, ta0, ta1), bN[ks], ks ? accN : z, 0, 0, 0);
            }
            uint4 lo, hi;
            tile_to_rows<true>(accN, 1.0f, lo, hi);
            if ((lane & 31) == 0) store(g, a.Wout >> 1, lo, hi);
        };
        uint4 held[2][2][2];  // packed sum / difference rows of items 0 and 1, kept until the next image has been staged
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int ta0 = tr_addr<G>(lane, g, 0), ta1 = tr_addr<G>(lane, g, 1);
            f32x16 accE, accO;
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) {
                if (ks && ks >= kend) break;
                const half8 dE = tr_read(img + ks * 16 * ROWB, ta0, ta1);
                const half8 dO = tr_read(img + (SIN0 + ks * 16) * ROWB, ta0, ta1);
                accE = __builtin_amdgcn_mfma_f32_32x32x16_f16(dE, bE[ks], ks ? accE : z, 0, 0, 0);
                accO = __builtin_amdgcn_mfma_f32_32x32x16_f16(dO, bO[ks], ks ? accO : z, 0, 0, 0);
            }
            f32x16 sum, dif;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sum[i] = accE[i] + accO[i];
                dif[i] = accE[i] - accO[i];
            }
            uint4 slo, shi, dlo, dhi;
            tile_to_rows<true>(sum, 1.0f, slo, shi);
            tile_to_rows<true>(dif, 1.0f, dlo, dhi);
            if (g < 2) {
                held[g][0][0] = slo;
                held[g][0][1] = shi;
                held[g][1][0] = dlo;
                held[g][1][1] = dhi;
            } else {
                store(g, m, slo, shi);
                if (m > 0) store(g, a.Wout - m, dlo, dhi);
                if (wave == 0) nyquist(g);
            }
            if (g == 1) {
                // stage the next image before the first store of this group goes out (see mf_analysis)
                if (more) stash(fresh_tid(wave), pre, lds + ((it + 1) & 1) * IMG_W, nband);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    store(gg, m, held[gg][0][0], held[gg][0][1]);
                    if (m > 0) store(gg, a.Wout - m, held[gg][1][0], held[gg][1][1]);
                }
                if (wave == 0) {
                    nyquist(0);
                    nyquist(1);
                }
            }
        }
        if (!more) break;
        __syncthreads();
        grp = ngrp;
        band = nband;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host: tables, eligibility, launch
// ---------------------------------------------------------------------------------------------------------------
static int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// orthonormal real Fourier basis value: index 0 DC, 1 dummy, 2c cos, 2c+1 sin
static double basis_val(int idx, int pos, int N) {
    if (idx == 0) return 1.0 / std::sqrt((double)N);
    if (idx == 1) return 0.0;
    const int f = idx >> 1;
    const double th = 2.0 * M_PI * (double)(((long)f * pos) % N) / N;
    return std::sqrt(2.0 / N) * ((idx & 1) ? std::sin(th) : std::cos(th));
}

}  // namespace mf

void fbm_mfma_plan_free(sr_fbm_plan* pl) {
    if (pl && pl->mf_block) (void)hipFree(pl->mf_block);
    if (pl) pl->mf_block = nullptr;
}

int fbm_mfma_plan_build(sr_fbm_plan* pl, const unsigned char* rows) {
    using namespace mf;
    pl->mf_ok = 0;
    const int H = pl->H, W = pl->W;
    // any even size from 32 to 544 (tables are zero-padded to whole k-steps / basis tiles); multiples of 32 up to 512 take
    // the narrower kernel instances, everything else (the 522 / 532-pixel windows of a tiled image) the wider ones
    if (!pl->contiguous || H % 2 || W % 2 || H > 544 || W > 544 || H < 32 || W < 32) return SR_OK;
    // x axis: mask k keeps column frequencies 0 .. half[k] - 1, nested
    for (int k = 0; k < 3; ++k) {
        if (pl->half[k] < 1 || 2 * (pl->half[k] - 1) >= W) return SR_OK;
        if (k && pl->half[k] > pl->half[k - 1]) return SR_OK;
    }
    const int jw = 2 * pl->half[0];
    if (jw > 272) return SR_OK;  // synthesis depth: 17 k-steps
    // y axis: symmetric / antisymmetric parts of each row mask
    std::vector<double> s(3 * (H / 2 + 1), 0.0), d(3 * (H / 2 + 1), 0.0);
    int fmax[3];
    for (int k = 0; k < 3; ++k) {
        const unsigned char* rm = rows + k * H;
        if (rm[H / 2]) return SR_OK;  // Nyquist row kept: not representable
        fmax[k] = -1;
        for (int f = 0; f < H / 2; ++f) {
            const double mp = rm[f] ? 1.0 : 0.0, mn = rm[(H - f) % H] ? 1.0 : 0.0;
            s[k * (H / 2 + 1) + f] = f ? 0.5 * (mp + mn) : mp;
            d[k * (H / 2 + 1) + f] = f ? 0.5 * (mp - mn) : 0.0;
            if (mp + mn > 0) fmax[k] = f;
        }
        if (fmax[k] < 0) return SR_OK;
        if (k && fmax[k] > fmax[k - 1]) return SR_OK;
    }
    const int nh = 2 * (fmax[0] + 1);
    if (nh > 272) return SR_OK;
    // the kernels see both coefficient counts rounded up to whole basis tiles (the extra functions are zero)
    pl->mf_jw_fill = jw;
    pl->mf_nh_fill = nh;
    pl->mf_jw = pad_to(jw, 32);
    pl->mf_nh = pad_to(nh, 32);
    // y weights per band: ones up to the first frequency with s < 1 (main k-steps), then a window of 32 rows
    std::vector<float> win(3 * 32 * 2, 0.f);
    for (int k = 0; k < 3; ++k) {
        pl->mf_nw[k] = 2 * pl->half[k];
        pl->mf_nhk[k] = 2 * (fmax[k] + 1);
        const double* sk = &s[k * (H / 2 + 1)];
        const double* dk = &d[k * (H / 2 + 1)];
        if (sk[0] != 1.0) return SR_OK;
        int f1 = 0;
        while (f1 < H / 2 && sk[f1] == 1.0) ++f1;
        const int ksm = (2 * f1) >> 4;
        int last = -1;
        for (int f = 8 * ksm; f < H / 2; ++f) {
            if (sk[f] == 0.0 && dk[f] == 0.0) continue;
            const int r = 2 * f - 16 * ksm;
            if (r + 1 >= 32 || 2 * f + 1 >= nh) return SR_OK;  // does not fit the window
            win[(k * 32 + r) * 2] = win[(k * 32 + r + 1) * 2] = (float)sk[f];
            win[(k * 32 + r) * 2 + 1] = win[(k * 32 + r + 1) * 2 + 1] = (float)dk[f];
            last = r + 1;
        }
        pl->mf_ksm[k] = ksm;
        pl->mf_nwin[k] = last < 0 ? 0 : (last >> 4) + 1;
        if (ksm + pl->mf_nwin[k] > (pad_to(nh, 16) >> 4)) return SR_OK;
    }
    pl->mf_ld_qw = pad_to(jw, 16);
    pl->mf_ld_ph = pad_to(nh, 16);
    const int W16 = pad_to(W, 16), H16 = pad_to(H, 16);  // k extents of the analysis tables (zero beyond W / H)
    const bool fold = W % 64 == 0 && jw <= 256;  // folded synthesis along W (mf_synthesis_wf)
    const size_t n_qwf = fold ? (size_t)(W / 64 + 1) * 16 * 64 * 8 : 0;
    const size_t n_qwt = (size_t)pad_to(jw, 32) * W16, n_qw = (size_t)pad_to(W, 32) * pl->mf_ld_qw, n_pht = (size_t)pad_to(nh, 32) * H16,
                 n_ph = (size_t)pad_to(H, 32) * pl->mf_ld_ph;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_qwt = carve(n_qwt * sizeof(h16)), o_qw = carve(n_qw * sizeof(h16)), o_pht = carve(n_pht * sizeof(h16)),
                 o_ph = carve(n_ph * sizeof(h16)), o_win = carve(win.size() * sizeof(float)), o_qwf = carve(n_qwf * sizeof(h16));
    std::vector<char> host(off, 0);
    h16* qwt = (h16*)&host[o_qwt];
    h16* qw = (h16*)&host[o_qw];
    h16* pht = (h16*)&host[o_pht];
    h16* ph = (h16*)&host[o_ph];
    // synthesis tables: output rows beyond the last valid one (whole-tile padding) repeat it; the kernels store them to
    // the last row's address (same values, same address: no predicate around the stores)
    for (int j = 0; j < jw; ++j) {
        for (int w = 0; w < W; ++w) qwt[frag_index(j, w, W16 / 16)] = (h16)basis_val(j, w, W);
        for (int w = 0; w < pad_to(W, 32); ++w)
            qw[frag_index(w, j, pl->mf_ld_qw / 16)] = (h16)(4.0 * basis_val(j, std::min(w, W - 1), W));  // U is kept as U / 4
    }
    for (int n = 0; n < nh; ++n) {
        for (int h = 0; h < H; ++h) pht[frag_index(n, h, H16 / 16)] = (h16)basis_val(n, h, H);
        for (int h = 0; h < pad_to(H, 32); ++h)
            ph[frag_index(h, n, pl->mf_ld_ph / 16)] = (h16)(16.0 * basis_val(n, std::min(h, H - 1), H));  // S is kept as S / 64, U as U / 4
    }
    memcpy(&host[o_win], win.data(), win.size() * sizeof(float));
    if (fold) {
        h16* qwf = (h16*)&host[o_qwf];
        for (int w = 0; w < W / 2; ++w)
            for (int c = 0; c < jw / 2; ++c) {  // k = c: cos coefficient 2c; k = 128 + c: sin coefficient 2c + 1
                qwf[frag_index(w, c, 16)] = (h16)(4.0 * basis_val(2 * c, w, W));
                qwf[frag_index(w, 128 + c, 16)] = (h16)(4.0 * basis_val(2 * c + 1, w, W));
            }
        for (int c = 0; c < jw / 2; ++c) qwf[frag_index(W / 2, c, 16)] = (h16)(4.0 * basis_val(2 * c, W / 2, W));
    }
    char* dev = nullptr;
    SR_HIP(hipMalloc((void**)&dev, off));
    const hipError_t e = hipMemcpy(dev, host.data(), off, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(dev);
        return hip_fail(e);
    }
    pl->mf_block = dev;
    pl->mf_qwt = dev + o_qwt;
    pl->mf_qw = dev + o_qw;
    pl->mf_pht = dev + o_pht;
    pl->mf_ph = dev + o_ph;
    pl->mf_win = dev + o_win;
    pl->mf_qwf = fold ? dev + o_qwf : nullptr;
    pl->mf_ok = 1;
    return SR_OK;
}

// 32 consecutive channels starting at a multiple of 32 are contiguous and 16-byte aligned in this view
static bool view32_ok(const View& v, int C) {
    if (!v.p || (uintptr_t)v.p % 16 || C % 32 || v.c0 % 32 || v.bs % 8 || v.bs >= (1L << 31)) return false;
    if (v.csh >= 5) return v.cbs % 8 == 0;            // blocks of >= 32 channels
    return v.csh == 3 && v.cbs == 8 && v.ld % 8 == 0;  // NHWC
}

int g_mfma_cus = 0;
int g_mfma_no_fold = 0;  // debug hook: W-synthesis without the even / odd fold

int fbm_lowpass_mfma(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8, float* pooled,
                     int pooled_ld, void* ws, hipStream_t st, int bmask) {
    using namespace mf;
    bmask &= 7;
    if (!bmask) return SR_ERR_ARG;
    int bfirst = 0, bnext[3] = {-1, -1, -1};
    while (!((bmask >> bfirst) & 1)) ++bfirst;
    for (int k = 0; k < 3; ++k)
        for (int j = 2; j > k; --j)
            if ((bmask >> j) & 1) bnext[k] = j;
    if (dtype != SR_BF16 || !pl->mf_ok) return SR_NOT_ELIGIBLE;
    const int H = pl->H, W = pl->W, C = x.C;
    if (!view32_ok(x, C) || !view32_ok(l2, C) || !view32_ok(l4, C) || !view32_ok(l8, C)) return SR_NOT_ELIGIBLE;
    if (int rc = device_cus(&g_mfma_cus)) return rc;
    const int CG = C / 32, NB = B * CG;
    const int jw = pl->mf_jw, nh = pl->mf_nh;
    h16* T = (h16*)ws;                      // [NB][H][jw][32]
    h16* S = T + (size_t)NB * H * jw * 32;  // [NB][nh][jw][32]
    h16* U[3];                              // [NB][nw[k] / 4][H][4][32]
    size_t usz[3];  // halfs per bcg: column groups of four, padded
    for (int k = 0; k < 3; ++k) usz[k] = (size_t)((pl->mf_nw[k] + 3) / 4) * H * 128;
    U[0] = S + (size_t)NB * nh * jw * 32;
    U[1] = U[0] + (size_t)NB * usz[0];
    U[2] = U[1] + (size_t)NB * usz[1];
    if ((size_t)((char*)(U[2] + (size_t)NB * usz[2]) - (char*)ws) > sr_fbm_workspace_bytes(pl, B, C)) return SR_NOT_ELIGIBLE;
    auto grid_for = [&](long items, int G) { return (int)std::min<long>((long)NB * ((items + G - 1) / G), g_mfma_cus); };
    const int nh_real = pl->mf_nhk[0];
    // wide instances: K up to 544 in the analysis (34 k-steps), up to 272 in the synthesis (17 k-steps)
    const bool wide_s = pl->mf_ld_qw > 256 || pl->mf_ld_ph > 256;
    const int lds_syn = 2 * (wide_s ? 17 : 16) * 16 * 256 + WIN_B;

    AnaArgs a1{};  // T = x Qw
    a1.x = x;
    a1.ups = ups;
    a1.Ws = W >> ups;
    a1.dst = T;
    a1.dst_b = (long)H * jw * 32;
    a1.dst_m = 32;
    a1.dst_i = jw * 32;
    a1.basis = (const h16*)pl->mf_qwt;
    a1.K = W;
    a1.M = jw;
    a1.NB = NB;
    a1.CG = CG;
    a1.nitems = H;
    // fp16 range of the intermediates: an orthonormal coefficient can reach sqrt(W) (sqrt(H W)) times the largest
    // activation, so T and U are kept as T / 4, U / 4 and S as S / 64 (no overflow below |x| ~ 8000; the epilogues saturate)
    a1.scale = 0.25f;
    // k-steps of the instance: the k loop always runs all of them (see mf_analysis), so short transforms -- the 64 x 64
    // crops of a training step, small inference inputs -- take instances of 4 or 8 steps instead of multiplying zeros
#define SR_ANA(FV, KS_, ARGS, ITEMS)                                                                          \
    do {                                                                                                      \
        SR_LDS_ATTR(2 * KS_ * 16 * 128 + (KS_ == 34 ? 34 * 512 : 0), mf_analysis<FV, KS_>);                   \
        mf_analysis<FV, KS_><<<grid_for(ITEMS, 2), THREADS, 2 * KS_ * 16 * 128 + (KS_ == 34 ? 34 * 512 : 0), st>>>(ARGS); \
    } while (0)
#define SR_ANA_BY_K(FV, K_, ARGS, ITEMS)                  \
    do {                                                  \
        if ((K_) <= 64) SR_ANA(FV, 4, ARGS, ITEMS);       \
        else if ((K_) <= 128) SR_ANA(FV, 8, ARGS, ITEMS); \
        else if ((K_) <= 512) SR_ANA(FV, 32, ARGS, ITEMS);\
        else SR_ANA(FV, 34, ARGS, ITEMS);                 \
    } while (0)
    static const bool no_xtile_a = getenv("SR_AMD_NO_XTILE") != nullptr;  // A/B: the ninth tile in a second launch
    // nine tiles whose last one has at most 16 distinct functions, on the 34-step instance: one launch
    // (at most 15: the analysis tables hold zeros beyond their last function, and lanes 16..31 read function 15's fragment)
    auto ninth_in_lds = [&](int K, int tiles, int valid) { return !no_xtile_a && K > 512 && tiles == 9 && valid - 256 <= 15; };
    a1.xtile1 = ninth_in_lds(W, jw / 32, pl->mf_jw_fill) ? 9 : 0;
    for (int t0 = 0; t0 < (a1.xtile1 ? 8 : jw / 32); t0 += 8) {  // more than 8 basis tiles: the data is streamed once per 8 tiles
        a1.tile0 = t0;
        SR_ANA_BY_K(true, W, a1, H);
        SR_LAUNCH_CHECK();
    }

    AnaArgs a2{};  // S = Ph^T T / 16
    a2.src = T;
    a2.src_b = (long)H * jw * 32;
    a2.src_k = jw * 32;
    a2.src_i = 32;
    a2.dst = S;
    a2.dst_b = (long)nh * jw * 32;
    a2.dst_m = jw * 32;
    a2.dst_i = 32;
    a2.basis = (const h16*)pl->mf_pht;
    a2.K = H;
    a2.M = nh;
    a2.NB = NB;
    a2.CG = CG;
    a2.nitems = pl->mf_nw[bfirst] + (pl->mf_nw[bfirst] & 1);  // coefficient columns of the widest wanted band
    a2.scale = 1.0f / 16.0f;
    a2.pool_scale = 4.0f / std::sqrt((float)H * (float)W);
    a2.xtile1 = ninth_in_lds(H, nh / 32, pl->mf_nh_fill) ? 9 : 0;
    for (int t0 = 0; t0 < (a2.xtile1 ? 8 : nh / 32); t0 += 8) {
        a2.tile0 = t0;
        a2.pooled = t0 == 0 ? pooled : nullptr;
        a2.pooled_ld = pooled_ld;
        SR_ANA_BY_K(false, H, a2, a2.nitems);
        SR_LAUNCH_CHECK();
    }

    SynArgs s3{};  // U_k = P (weights_k S)
    s3.xtile = -1;
    for (int k = 0; k < 3; ++k) {
        s3.src[k] = S;
        s3.src_b[k] = (long)nh * jw * 32;
        s3.src_k[k] = jw * 32;
        s3.src_i[k] = 32;
        s3.K[k] = nh_real;
        s3.ksm[k] = pl->mf_ksm[k];
        s3.nwin[k] = pl->mf_nwin[k];
        s3.dst[k] = U[k];
        s3.dst_b[k] = (long)usz[k];  // grouped layout, see u_addr
        s3.dst_m[k] = 128;
        s3.dst_i[k] = 32;
        s3.nitems[k] = pl->mf_nw[k];
    }
    s3.bmask = bmask;
    s3.bfirst = bfirst;
    for (int k = 0; k < 3; ++k) s3.bnext[k] = bnext[k];
    s3.win = (const float2*)pl->mf_win;
    s3.M = pad_to(H, 32);
    s3.Mv = H;
    s3.NB = NB;
    s3.CG = CG;
    s3.basis = (const h16*)pl->mf_ph;
    s3.nks_tab = pl->mf_ld_ph / 16;
    for (int t0 = 0; t0 < s3.M / 32; t0 += 16) {  // more than 16 basis tiles: S is streamed once per 16 tiles
        s3.tile0 = t0;
        // k-steps the instance must hold: the coefficient rows and every band's main + window steps
        int need = (nh_real + 15) / 16;
        for (int k = 0; k < 3; ++k) need = std::max(need, pl->mf_ksm[k] + pl->mf_nwin[k]);
#define SR_SYNH(KS_, ...)                                                                                                      \
    do {                                                                                                                       \
        SR_LDS_ATTR(2 * KS_ * 16 * 256 + WIN_B, mf_synthesis<true, KS_, ##__VA_ARGS__>);                                       \
        mf_synthesis<true, KS_, ##__VA_ARGS__><<<grid_for(pl->mf_nw[bfirst], 4), THREADS, 2 * KS_ * 16 * 256 + WIN_B, st>>>(s3); \
    } while (0)
        static const bool no_solo = getenv("SR_AMD_NO_SOLO") != nullptr;  // A/B: the 17th tile by all eight waves
        if (wide_s && t0 > 0 && t0 + 1 == s3.M / 32 && !no_solo) SR_SYNH(17, true);  // one tile left over: one wave per item
        else if (wide_s) SR_SYNH(17);
        else if (need <= 4) SR_SYNH(4);
        else if (need <= 8) SR_SYNH(8);
        else SR_SYNH(16);
#undef SR_SYNH
        SR_LAUNCH_CHECK();
    }

    SynArgs s4{};  // L_k = U_k Qw_k^T
    s4.xtile = -1;
    const View outs[3] = {l2, l4, l8};
    for (int k = 0; k < 3; ++k) {
        s4.src[k] = U[k];
        s4.src_b[k] = (long)usz[k];  // grouped layout: row h at + h * 128, column k through u_addr
        s4.src_k[k] = 32;
        s4.src_i[k] = 128;
        s4.K[k] = pl->mf_nw[k];
        s4.ksm[k] = (pl->mf_nw[k] + 15) / 16;
        s4.out[k] = outs[k];
        s4.nitems[k] = H;
    }
    s4.bmask = bmask;
    s4.bfirst = bfirst;
    for (int k = 0; k < 3; ++k) s4.bnext[k] = bnext[k];
    s4.Wout = W;
    s4.M = pad_to(W, 32);
    s4.Mv = W;
    s4.NB = NB;
    s4.CG = CG;
    if (pl->mf_qwf && !g_mfma_no_fold) {
        for (int k = 0; k < 3; ++k) s4.ksm[k] = (pl->mf_nw[k] / 2 + 15) / 16;  // k-steps per half (cos / sin)
        s4.basis = (const h16*)pl->mf_qwf;
        s4.nks_tab = 16;
        const int kh = s4.ksm[0];  // k-steps per half of the widest band
#define SR_WF(KH_)                                                                              \
    do {                                                                                        \
        SR_LDS_ATTR(2 * 32 * KH_ * 256, mf_synthesis_wf<KH_>);                                  \
        mf_synthesis_wf<KH_><<<grid_for(H, 4), THREADS, 2 * 32 * KH_ * 256, st>>>(s4);          \
    } while (0)
        if (kh <= 1) SR_WF(1);
        else if (kh <= 2) SR_WF(2);
        else if (kh <= 4) SR_WF(4);
        else SR_WF(8);
#undef SR_WF
        SR_LAUNCH_CHECK();
    } else {
        s4.basis = (const h16*)pl->mf_qw;
        s4.nks_tab = pl->mf_ld_qw / 16;
        const int ntile = s4.M / 32;
        static const bool no_xtile = getenv("SR_AMD_NO_XTILE") != nullptr;  // A/B: the 17th tile in a second launch
        s4.xtile = (ntile == 17 && !no_xtile) ? 16 : -1;  // 17 tiles: the last one from LDS in the same launch
        for (int t0 = 0; t0 < (s4.xtile >= 0 ? 16 : ntile); t0 += 16) {
            s4.tile0 = t0;
            if (wide_s) {
                SR_LDS_ATTR(lds_syn, mf_synthesis<false, 17>);
                mf_synthesis<false, 17><<<grid_for(H, 4), THREADS, lds_syn, st>>>(s4);
            } else {
                SR_LDS_ATTR(lds_syn, mf_synthesis<false, 16>);
                mf_synthesis<false, 16><<<grid_for(H, 4), THREADS, lds_syn, st>>>(s4);
            }
            SR_LAUNCH_CHECK();
        }
    }
    return SR_OK;
}
}  // namespace sr
