// FBM low-pass planes on the matrix cores (bf16 activations, H and W multiples of 32 up to 512).
//
// The three box masks keep few frequencies, so the masked FFT round trip is three small dense products per plane,
//     T = x Qw         rows forward  : x-basis coefficients of every image row            (K = W,        M = jw)
//     U_k = B2_k (Ph^T T)   columns  : y-coefficients, mask weights, back to rows         (K = H / nh_k, M = nh / H)
//     L_k = U_k Qw_k^T   rows inverse : synthesis from the first nw[k] coefficients        (K = nw[k],    M = W)
// in REAL orthonormal Fourier bases (q_0 = 1/sqrt(W); q_2c = sqrt(2/W) cos, q_2c+1 = sqrt(2/W) sin; index 1 is an
// all-zero dummy so that (cos, sin) pairs stay 2-aligned).  The VALU FFT of fbm_fast.hip is bound by vector
// instruction issue (~60 % VALU busy); the same work as v_mfma_f32_32x32x16_f16 GEMMs leaves the kernels bound by
// the HBM stream of x, the fp16 intermediates and the three output planes.
//
// Row masks of the reference are not symmetric in +-fy (nearest resize of the 64-row prototype).  With
// s_f = (m[+f] + m[-f]) / 2 and d_f = (m[+f] - m[-f]) / 2 the filtered coefficient columns A (cos c) and B (sin c)
// of a column frequency c > 0 are
//     A' = sum_n s_n SA_n p_n + sum_f d_f (SB_cos,f p_sin,f - SB_sin,f p_cos,f)
//     B' = sum_n s_n SB_n p_n + sum_f d_f (SA_sin,f p_cos,f - SA_cos,f p_sin,f)
// (SA, SB = y-coefficients of A, B).  d_f is non-zero for a handful of boundary rows only: they become one extra
// 16-deep k-step whose data rows hold the partner column's coefficients (sign folded in) -- everything stays one GEMM.
//
// Every product has the data operand [k][32 channels] (channels innermost in memory, as the tensors are) and the
// basis operand [m][k] (k innermost, read straight from L2).  Data tiles are staged row-major in LDS (plain 16-byte
// stores of what was loaded) and read through ds_read_b64_tr_b16, which hands each lane the 4 k-values of its
// channel: no software transpose.  D = data^T x basis^T puts 4 consecutive channels of one output row in a lane;
// v_permlane32_swap pairs lanes l / l+32 so that each lane stores 32 contiguous bytes.
#include <hip/hip_fp16.h>

#include <cmath>
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
constexpr int G = 4;              // batch items (image rows / coefficient columns) per workgroup pass: 128 GEMM rows
constexpr int ROWB = G * 32 * 2;  // LDS image row: G items x 32 channels x fp16 = 256 B
// Basis tables are stored fragment-major: the 64 x 8 halfs one wave loads for (row tile nt, k-step ks) are 1 KiB
// contiguous (lane = 32 * (k / 8 % 2) + row % 32), so a fragment load is eight fully used cache lines instead of 32
// quarter-used ones -- the tables are re-read from L2 by every workgroup for every batch group.
constexpr int FRAG = 64 * 8;
__host__ __device__ inline size_t frag_index(int row, int k, int nks) {
    return ((size_t)((row >> 5) * nks + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
}

// Byte offset of 16-byte chunk ch (0..15) of image row `row`: XOR swizzle that keeps both the 16-byte row stores and
// the transposed reads conflict-free (programming guide T10, image (b)).
__device__ __forceinline__ int img_off(int row, int ch) { return ROWB * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// Address of this lane's share of the transposed read that delivers, for M-tile mt (= batch item), the k-values
// 8 * (lane >> 5) + 4 * half + {0..3} of k-step 0: lane 4q + p of a 16-lane group supplies row q, columns 4p..4p+3.
__device__ __forceinline__ int tr_addr(int lane, int mt, int half) {
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 8 * (g4 >> 1) + 4 * half + q;
    return img_off(row, mt * 4 + 2 * (g4 & 1) + (p >> 1)) + 8 * (p & 1);
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

// acc[mt][j] += data(mt)^T x basis(tile j)^T over `nks` k-steps of 16 starting at image row `row0`.
// bptr[j]: this lane's 8 basis values of the first k-step (tables are fragment-major, see frag_index).  The basis streams from L2 and is
// the only operand with a long latency: PF k-steps of fragments stay in flight per wave, in a rotating register
// queue.  Every load is unconditional (indices clamped to the last k-step): a branch around a load makes the
// compiler drain the whole memory pipeline at the join.  Tiles beyond the matrix (clamped pointers) are computed
// and dropped by the caller: a uniform branch around each MFMA would split the loop body into tiny blocks.
template <int NTW, int PF>
__device__ __forceinline__ void mma_steps(f32x16 (&acc)[G][NTW], const char* img, int row0, int nks, const h16* const (&bptr)[NTW],
                                          const int (&ta)[G][2]) {
    half8 ring[PF][NTW];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int j = 0; j < NTW; ++j) ring[u][j] = *(const half8*)(bptr[j] + min(u, nks - 1) * FRAG);
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
        half8 bc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            bc[j] = ring[0][j];
#pragma unroll
            for (int u = 0; u + 1 < PF; ++u) ring[u][j] = ring[u + 1][j];
            ring[PF - 1][j] = *(const half8*)(bptr[j] + min(ks + PF, nks - 1) * FRAG);
        }
        const char* base = img + (row0 + ks * 16) * ROWB;
#pragma unroll
        for (int mt = 0; mt < G; ++mt) {
            const half8 a = tr_read(base, ta[mt][0], ta[mt][1]);
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bc[j], acc[mt][j], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    const __half2 h = __floats2half2_rn(a, b);
    return *(const uint32_t*)&h;
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
        P[q][0] = BF16 ? pack_bf2(a, b) : pack_h2(a, b);
        P[q][1] = BF16 ? pack_bf2(c, d) : pack_h2(c, d);
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

// ---------------------------------------------------------------------------------------------------------------
// rows forward (MODE 0) and rows inverse (MODE 2): one basis, K in chunks of 256 through two LDS images
// ---------------------------------------------------------------------------------------------------------------
constexpr int KC = 256;
constexpr int IMG_B = KC * ROWB;  // 64 KiB

struct RowsArgs {
    View x;  // MODE 0 input (bf16)
    int ups;
    View out[3];     // MODE 2 outputs (bf16)
    h16* T;          // MODE 0 output / [B*CG][H][jw][32]
    const h16* U[3]; // MODE 2 input  / [B*CG][H][nw[k]][32]
    const h16* basis;
    int nks;  // k-steps per row tile of the (fragment-major) basis table
    int B, CG, H, W, jw, nw[3];
};

template <typename T>
__device__ __forceinline__ T sel3(int i, T a, T b, T c) {
    return i == 0 ? a : (i == 1 ? b : c);
}

// Workgroups of 4 waves with one 64 KiB image each: two of them share a CU, so that one streams its output rows (a
// wave's later loads cannot complete before its earlier stores have drained -- one in-order counter) while the other
// multiplies.  Per batch group: stage the K x 128 data chunk, barrier, then every wave runs its basis-row tiles
// {wave, wave + 4, ...} two at a time (128 accumulator registers) and stores them.
constexpr int RTHREADS = 256;

template <int MODE>
__global__ __launch_bounds__(RTHREADS, 2) void mf_rows(RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = MODE == 0 ? a.jw : a.W;
    const int NT = (M + 31) / 32;
    int ta[G][2];
#pragma unroll
    for (int mt = 0; mt < G; ++mt) {
        ta[mt][0] = tr_addr(lane, mt, 0);
        ta[mt][1] = tr_addr(lane, mt, 1);
    }
    const int Ws = a.W >> a.ups;
    const int ngroups = a.B * a.CG * (a.H / G);
    const int oct = tid & 3, kk0 = tid >> 2;  // item i -> batch member i >> 2, row kk0 + 64 (i & 3), channel octet oct
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int bcg = grp / (a.H / G), h0 = (grp % (a.H / G)) * G;
        const int b = bcg / a.CG, cg = bcg % a.CG;
#pragma unroll 1
        for (int band = 0; band < (MODE == 0 ? 1 : 3); ++band) {
            const int K = MODE == 0 ? a.W : a.nw[band];
            const int nch = (K + KC - 1) / KC;
            const int npass = nch == 1 ? (NT + 7) / 8 : 1;  // several chunks: the accumulators persist, NT <= 8 (host check)
            const h16* usrc = MODE == 2 ? a.U[band] : nullptr;
            f32x16 acc[G][2];
#pragma unroll 1
            for (int chunk = 0; chunk < nch; ++chunk) {
                const int kc = min(K - chunk * KC, KC), kc16 = (kc + 15) & ~15;
                uint4 pre[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int g = i >> 2, kk = kk0 + 64 * (i & 3), k = min(chunk * KC + kk, K - 1);  // clamped, zeroed below
                    if (MODE == 0) {
                        const unsigned pix = (unsigned)((h0 + g) >> a.ups) * Ws + (k >> a.ups);
                        pre[i] = *(const uint4*)((const bf16_t*)a.x.p + (long)b * a.x.bs + voff32(a.x, pix, cg * 32 + oct * 8));
                    } else {
                        pre[i] = *(const uint4*)(usrc + (((long)bcg * a.H + h0 + g) * K + k) * 32 + oct * 8);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int g = i >> 2, kk = kk0 + 64 * (i & 3);
                    uint4 v = MODE == 0 ? bf16x8_to_h16x8(pre[i]) : pre[i];
                    if (kk >= kc) v = make_uint4(0u, 0u, 0u, 0u);
                    if (kk < kc16) *(uint4*)(lds + img_off(kk, g * 4 + oct)) = v;
                }
                __syncthreads();
#pragma unroll 1
                for (int pass = 0; pass < npass; ++pass) {
                    if (chunk == 0) {
#pragma unroll
                        for (int mt = 0; mt < G; ++mt)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;
                    }
                    const int nt[2] = {wave + 8 * pass, wave + 8 * pass + 4};
                    const h16* bptr[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bptr[j] = a.basis + ((size_t)(min(nt[j], NT - 1) * a.nks + chunk * (KC / 16)) * 64 + lane) * 8;
                    mma_steps<2, 4>(acc, lds, 0, kc16 >> 4, bptr, ta);
                    if (chunk + 1 < nch) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int m = nt[j] * 32 + (lane & 31);
#pragma unroll
                        for (int mt = 0; mt < G; ++mt) {
                            uint4 lo, hi;
                            tile_to_rows<MODE == 2>(acc[mt][j], 1.0f, lo, hi);
                            if (m >= M) continue;
                            uint4* dst;
                            if (MODE == 0) {
                                dst = (uint4*)(a.T + (((long)bcg * a.H + h0 + mt) * a.jw + m) * 32 + (lane >> 5) * 16);
                            } else {
                                const View o = a.out[band];
                                dst = (uint4*)((bf16_t*)o.p + (long)b * o.bs + voff32(o, (unsigned)(h0 + mt) * a.W + m, cg * 32 + (lane >> 5) * 16));
                            }
                            dst[0] = lo;
                            dst[1] = hi;
                        }
                    }
                }
                __syncthreads();  // image consumed
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// columns: y-analysis (K = H in chunks of 128), mask weights + asymmetric partner rows, y-synthesis per band
// ---------------------------------------------------------------------------------------------------------------
constexpr int KC1 = 128;
constexpr int IMG1_B = KC1 * ROWB;         // 32 KiB, two of them
constexpr int IMG2_ROWS = 256 + 3 * 16;    // coefficients + 16 partner rows per band
constexpr int IMG2_B = IMG2_ROWS * ROWB;   // 76 KiB
constexpr float S_SCALE = 1.0f / 16.0f;    // coefficients are kept as S / 16 in fp16 (the synthesis basis carries the 16)

struct ColsArgs {
    const h16* T;  // [B*CG][H][jw][32]
    h16* U[3];     // [B*CG][H][nw[k]][32]
    const h16* pht;
    const h16* b2[3];
    int ld_b2[3];
    int B, CG, H, jw, nw[3], nh, nhk[3], asym0[3], nasym[3];
    float* pooled;
    int pooled_ld;
    float pool_scale;
};

__global__ __launch_bounds__(THREADS) void mf_cols(ColsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* img2 = lds + 2 * IMG1_B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ta[G][2];
#pragma unroll
    for (int mt = 0; mt < G; ++mt) {
        ta[mt][0] = tr_addr(lane, mt, 0);
        ta[mt][1] = tr_addr(lane, mt, 1);
    }
    const int jgroups = (a.jw + G - 1) / G;
    const int ngroups = a.B * a.CG * jgroups;
    const int nch1 = (a.H + KC1 - 1) / KC1;
    const int NT1 = (a.nh + 31) / 32, NT2 = a.H / 32;
    // partner rows that no band writes must read as zero for ever
    for (int i = tid; i < 48 * 16; i += THREADS) *(uint4*)(img2 + img_off(256 + (i >> 4), i & 15)) = make_uint4(0u, 0u, 0u, 0u);

    const int ch16 = tid & 15, kk0 = tid >> 4;  // item i -> row kk0 + 32 i, 16-byte chunk ch16 (batch member ch16 >> 2)
    auto fetch = [&](int grp, int chunk, uint4 (&pre)[4]) {
        const int bcg = grp / jgroups, j0 = (grp % jgroups) * G;
        const int j = j0 + (ch16 >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = chunk * KC1 + kk0 + 32 * i;
            pre[i] = make_uint4(0u, 0u, 0u, 0u);
            if (k < a.H && j < a.jw) pre[i] = *(const uint4*)(a.T + (((long)bcg * a.H + k) * a.jw + j) * 32 + (ch16 & 3) * 8);
        }
    };
    auto stash = [&](const uint4 (&pre)[4], char* img) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(uint4*)(img + img_off(kk0 + 32 * i, ch16)) = pre[i];
    };

    int grp = blockIdx.x;
    if (grp >= ngroups) return;
    uint4 pre[4];
    fetch(grp, 0, pre);
    __syncthreads();  // zero fill of the partner rows
    stash(pre, lds);
    __syncthreads();
    unsigned step = 0;
    for (; grp < ngroups; grp += gridDim.x) {
        const int bcg = grp / jgroups, j0 = (grp % jgroups) * G;
        // ---- analysis along H -------------------------------------------------------------------------------
        f32x16 acc1[G][1];
#pragma unroll
        for (int mt = 0; mt < G; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc1[mt][0][i] = 0.f;
        const bool on1[1] = {wave < NT1};
        bool more = true;
        for (int c = 0; c < nch1; ++c) {
            int ng = grp, nc = c + 1;
            if (nc == nch1) {
                nc = 0;
                ng = grp + gridDim.x;
            }
            more = ng < ngroups;
            if (more) fetch(ng, nc, pre);
            const char* img = lds + (step & 1) * IMG1_B;
            const int kc = min(a.H - c * KC1, KC1);
            const h16* bptr[1] = {a.pht + ((size_t)(min(wave, NT1 - 1) * (a.H / 16) + c * (KC1 / 16)) * 64 + lane) * 8};
            mma_steps<1, 4>(acc1, img, 0, (kc + 15) >> 4, bptr, ta);
            if (c + 1 < nch1) {
                stash(pre, lds + ((step + 1) & 1) * IMG1_B);
                __syncthreads();
                ++step;
            }
        }
        // (the prefetched first chunk of the next group stays in `pre` until the synthesis below is done)
        // ---- coefficients -> image 2 ---------------------------------------------------------------------------
        if (on1[0]) {
            const int n = wave * 32 + (lane & 31), hk = lane >> 5;
            if (n < 256) {
#pragma unroll
                for (int mt = 0; mt < G; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint2 v;
                        v.x = pack_h2(acc1[mt][0][4 * q] * S_SCALE, acc1[mt][0][4 * q + 1] * S_SCALE);
                        v.y = pack_h2(acc1[mt][0][4 * q + 2] * S_SCALE, acc1[mt][0][4 * q + 3] * S_SCALE);
                        *(uint2*)(img2 + img_off(n, mt * 4 + q) + 8 * hk) = v;
                    }
            }
            // partner rows of the asymmetric mask rows: lane n = 2f (+1) feeds the sin (cos) slot of its frequency
            const int f = n >> 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = f - a.asym0[k];
                if (idx < 0 || idx >= a.nasym[k]) continue;
                const int row = 256 + 16 * k + 2 * idx + (1 - (n & 1));
#pragma unroll
                for (int mt = 0; mt < G; ++mt) {
                    const bool dc = (j0 + mt) < 2;  // DC column (and its dummy): the reference averages the mask there
                    const float sg = dc ? 0.f : ((((n ^ mt) & 1) == 0) ? S_SCALE : -S_SCALE);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint2 v;
                        v.x = pack_h2(acc1[mt ^ 1][0][4 * q] * sg, acc1[mt ^ 1][0][4 * q + 1] * sg);
                        v.y = pack_h2(acc1[mt ^ 1][0][4 * q + 2] * sg, acc1[mt ^ 1][0][4 * q + 3] * sg);
                        *(uint2*)(img2 + img_off(row, mt * 4 + q) + 8 * hk) = v;
                    }
                }
            }
            if (a.pooled && j0 == 0 && n == 0) {  // mean = <x, q_0 p_0> / sqrt(HW)
                const int b = bcg / a.CG, cg = bcg % a.CG;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    a.pooled[(long)b * a.pooled_ld + cg * 32 + (i & 3) + 8 * (i >> 2) + 4 * hk] = acc1[0][0][i] * a.pool_scale;
            }
        }
        __syncthreads();  // image 2 complete; image-1 stage `step` consumed
        // ---- synthesis along H, one band at a time --------------------------------------------------------------
        const int n2[2] = {wave, wave + 8};
        const bool on2[2] = {n2[0] < NT2, n2[1] < NT2};
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            if (j0 >= a.nw[k]) continue;
            f32x16 acc2[G][2];
#pragma unroll
            for (int mt = 0; mt < G; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc2[mt][j][i] = 0.f;
            const int nks = (a.nhk[k] + 15) >> 4;
            const h16* bptr[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bptr[j] = a.b2[k] + ((size_t)(min(n2[j], NT2 - 1) * (a.ld_b2[k] / 16)) * 64 + lane) * 8;
            mma_steps<2, 3>(acc2, img2, 0, nks, bptr, ta);
            if (a.nasym[k] > 0) {
                const h16* bx[2] = {bptr[0] + nks * FRAG, bptr[1] + nks * FRAG};
                mma_steps<2, 1>(acc2, img2, 256 + 16 * k, 1, bx, ta);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!on2[j]) continue;
                const int m = n2[j] * 32 + (lane & 31);
#pragma unroll
                for (int mt = 0; mt < G; ++mt) {
                    uint4 lo, hi;
                    tile_to_rows<false>(acc2[mt][j], 1.0f, lo, hi);
                    if (j0 + mt >= a.nw[k]) continue;
                    uint4* dst = (uint4*)(a.U[k] + (((long)bcg * a.H + m) * a.nw[k] + j0 + mt) * 32 + (lane >> 5) * 16);
                    dst[0] = lo;
                    dst[1] = hi;
                }
            }
        }
        if (more) stash(pre, lds + ((step + 1) & 1) * IMG1_B);
        __syncthreads();  // next group's first chunk staged; image 2 free again
        ++step;
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
    if (!pl->contiguous || H % 32 || W % 32 || H > 512 || W > 512 || H < 32 || W < 32) return SR_OK;
    // x axis: mask k keeps column frequencies 0 .. half[k] - 1, nested
    for (int k = 0; k < 3; ++k) {
        if (pl->half[k] < 1 || 2 * (pl->half[k] - 1) >= W) return SR_OK;
        if (k && pl->half[k] > pl->half[k - 1]) return SR_OK;
    }
    const int jw = 2 * pl->half[0];
    if (jw > 256) return SR_OK;
    // y axis: symmetric / antisymmetric parts of each row mask
    std::vector<double> s(3 * (H / 2 + 1), 0.0), d(3 * (H / 2 + 1), 0.0);
    int fmax[3], asym0[3], nasym[3];
    for (int k = 0; k < 3; ++k) {
        const unsigned char* rm = rows + k * H;
        if (rm[H / 2]) return SR_OK;  // Nyquist row kept: not representable
        fmax[k] = -1;
        asym0[k] = 0;
        nasym[k] = 0;
        int last_asym = -1;
        for (int f = 0; f < H / 2; ++f) {
            const double mp = rm[f] ? 1.0 : 0.0, mn = rm[(H - f) % H] ? 1.0 : 0.0;
            s[k * (H / 2 + 1) + f] = f ? 0.5 * (mp + mn) : mp;
            d[k * (H / 2 + 1) + f] = f ? 0.5 * (mp - mn) : 0.0;
            if (mp + mn > 0) fmax[k] = f;
            if (f && mp != mn) {
                if (!nasym[k]) asym0[k] = f;
                if (last_asym >= 0 && f != last_asym + 1) return SR_OK;  // must be one run
                last_asym = f;
                ++nasym[k];
            }
        }
        if (fmax[k] < 0 || nasym[k] > 8) return SR_OK;
        if (k && fmax[k] > fmax[k - 1]) return SR_OK;
    }
    const int nh = 2 * (fmax[0] + 1);
    if (nh > 256) return SR_OK;
    pl->mf_jw = jw;
    pl->mf_nh = nh;
    for (int k = 0; k < 3; ++k) {
        pl->mf_nw[k] = 2 * pl->half[k];
        pl->mf_nhk[k] = 2 * (fmax[k] + 1);
        pl->mf_asym0[k] = asym0[k];
        pl->mf_nasym[k] = nasym[k];
        pl->mf_ld_b2[k] = pad_to(pl->mf_nhk[k], 16) + 16;
    }
    pl->mf_ld_qw = pad_to(jw, 16);
    const size_t n_qwt = (size_t)pad_to(jw, 32) * W, n_qw = (size_t)W * pl->mf_ld_qw, n_pht = (size_t)pad_to(nh, 32) * H;
    size_t n_b2[3], off = 0;
    auto carve = [&](size_t n) { size_t o = off; off = align_up(off + n * sizeof(h16), 256); return o; };
    const size_t o_qwt = carve(n_qwt), o_qw = carve(n_qw), o_pht = carve(n_pht);
    size_t o_b2[3];
    for (int k = 0; k < 3; ++k) {
        n_b2[k] = (size_t)H * pl->mf_ld_b2[k];
        o_b2[k] = carve(n_b2[k]);
    }
    std::vector<char> host(off, 0);
    h16* qwt = (h16*)&host[o_qwt];
    h16* qw = (h16*)&host[o_qw];
    h16* pht = (h16*)&host[o_pht];
    for (int j = 0; j < jw; ++j)
        for (int w = 0; w < W; ++w) {
            const double v = basis_val(j, w, W);
            qwt[frag_index(j, w, W / 16)] = (h16)v;
            qw[frag_index(w, j, pl->mf_ld_qw / 16)] = (h16)v;
        }
    for (int n = 0; n < nh; ++n)
        for (int h = 0; h < H; ++h) pht[frag_index(n, h, H / 16)] = (h16)basis_val(n, h, H);
    for (int k = 0; k < 3; ++k) {
        h16* b2 = (h16*)&host[o_b2[k]];
        const int ld = pl->mf_ld_b2[k], kx = pad_to(pl->mf_nhk[k], 16);
        for (int h = 0; h < H; ++h) {
            for (int n = 0; n < pl->mf_nhk[k]; ++n) b2[frag_index(h, n, ld / 16)] = (h16)(16.0 * s[k * (H / 2 + 1) + (n >> 1)] * basis_val(n, h, H));
            for (int i = 0; i < nasym[k]; ++i) {
                const int f = asym0[k] + i;
                const double df = d[k * (H / 2 + 1) + f];
                // slot 2i pairs with the cos function (data: -/+ partner sin coefficient), slot 2i+1 with the sin function
                b2[frag_index(h, kx + 2 * i, ld / 16)] = (h16)(16.0 * df * basis_val(2 * f, h, H));
                b2[frag_index(h, kx + 2 * i + 1, ld / 16)] = (h16)(16.0 * df * basis_val(2 * f + 1, h, H));
            }
        }
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
    for (int k = 0; k < 3; ++k) pl->mf_b2[k] = dev + o_b2[k];
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

int fbm_lowpass_mfma(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8, float* pooled,
                     int pooled_ld, void* ws, hipStream_t st) {
    using namespace mf;
    if (dtype != SR_BF16 || !pl->mf_ok) return SR_NOT_ELIGIBLE;
    const int H = pl->H, W = pl->W, C = x.C;
    if (!view32_ok(x, C) || !view32_ok(l2, C) || !view32_ok(l4, C) || !view32_ok(l8, C)) return SR_NOT_ELIGIBLE;
    if (!g_mfma_cus) {
        int dev = 0;
        SR_HIP(hipGetDevice(&dev));
        SR_HIP(hipDeviceGetAttribute(&g_mfma_cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int CG = C / 32;
    const long rows = (long)B * CG * H;
    h16* T = (h16*)ws;
    h16* U[3];
    U[0] = T + rows * pl->mf_jw * 32;
    U[1] = U[0] + rows * pl->mf_nw[0] * 32;
    U[2] = U[1] + rows * pl->mf_nw[1] * 32;

    RowsArgs ra{};
    ra.x = x;
    ra.ups = ups;
    ra.T = T;
    ra.basis = (const h16*)pl->mf_qwt;
    ra.nks = W / 16;
    ra.B = B;
    ra.CG = CG;
    ra.H = H;
    ra.W = W;
    ra.jw = pl->mf_jw;
    for (int k = 0; k < 3; ++k) ra.nw[k] = pl->mf_nw[k];
    const int grid_rows = (int)std::min<long>(rows / G, 2L * g_mfma_cus);
    SR_HIP(hipFuncSetAttribute((const void*)mf_rows<0>, hipFuncAttributeMaxDynamicSharedMemorySize, IMG_B));
    mf_rows<0><<<grid_rows, RTHREADS, IMG_B, st>>>(ra);
    SR_LAUNCH_CHECK();

    ColsArgs ca{};
    ca.T = T;
    ca.pht = (const h16*)pl->mf_pht;
    ca.B = B;
    ca.CG = CG;
    ca.H = H;
    ca.jw = pl->mf_jw;
    ca.nh = pl->mf_nh;
    for (int k = 0; k < 3; ++k) {
        ca.U[k] = U[k];
        ca.b2[k] = (const h16*)pl->mf_b2[k];
        ca.ld_b2[k] = pl->mf_ld_b2[k];
        ca.nw[k] = pl->mf_nw[k];
        ca.nhk[k] = pl->mf_nhk[k];
        ca.asym0[k] = pl->mf_asym0[k];
        ca.nasym[k] = pl->mf_nasym[k];
    }
    ca.pooled = pooled;
    ca.pooled_ld = pooled_ld;
    ca.pool_scale = 1.0f / std::sqrt((float)H * (float)W);
    const long cgroups = (long)B * CG * ((pl->mf_jw + G - 1) / G);
    const int lds_cols = 2 * IMG1_B + IMG2_B;
    SR_HIP(hipFuncSetAttribute((const void*)mf_cols, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cols));
    mf_cols<<<(int)std::min<long>(cgroups, g_mfma_cus), THREADS, lds_cols, st>>>(ca);
    SR_LAUNCH_CHECK();

    ra.basis = (const h16*)pl->mf_qw;
    ra.nks = pl->mf_ld_qw / 16;
    ra.out[0] = l2;
    ra.out[1] = l4;
    ra.out[2] = l8;
    for (int k = 0; k < 3; ++k) ra.U[k] = U[k];
    SR_HIP(hipFuncSetAttribute((const void*)mf_rows<2>, hipFuncAttributeMaxDynamicSharedMemorySize, IMG_B));
    mf_rows<2><<<grid_rows, RTHREADS, IMG_B, st>>>(ra);
    SR_LAUNCH_CHECK();
    return SR_OK;
}
}  // namespace sr
