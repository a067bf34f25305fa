// FBM low-pass planes: L_k = irfft2(rfft2(x) * mask_k), k = 2,4,8  (FDConv.py:427-457).
//
// MI355X formulation.  Activations are NHWC, so the channel axis is the SIMD axis: two real
// channels are packed into one complex signal z = x_a + i*x_b and a plain complex 2-D FFT is
// run with consecutive lanes on consecutive channel pairs (conflict-free LDS, wave-uniform
// twiddles).  Because the box masks are real and applied identically to both channels,
//     L_a + i*L_b = ifft2( fft2(z) * Mfull_k )
// where Mfull_k is the full-plane mask equivalent to the reference's half-spectrum mask plus
// C2R semantics (oracle/fdconv_ref.py: fbm_full_masks).  Three kernels, all in fp32:
//   rows_fwd : FFT along W of every image row, keeps only the columns mask k=2 keeps
//   cols     : FFT along H of every kept column, then per mask: row mask, inverse FFT along H
//   rows_inv : per mask: scatter kept columns, inverse FFT along W, real/imag -> channel pair
// The DC bin of the column pass is the plane sum, so adaptive_avg_pool2d (FDConv.py:643) is a
// by-product.  The FFT is a Stockham autosort with a runtime radix list (any length whose
// factors are small), one output per work item per pass.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <cstdlib>
#include <cstring>

#include <mutex>

#include "common.h"

#include "fbm_plan.h"

namespace sr {

// ---- host: masks --------------------------------------------------------------------------
static void base_masks(bool rows[3][64], bool cols[3][33]) {
    // FDConv.py:393-415 evaluated in float32: k=2 strict (|f| < 0.25), k=4,8 inclusive.
    const int K[3] = {2, 4, 8};
    for (int k = 0; k < 3; ++k) {
        const float thr = (float)(0.5 / K[k] + 1e-8);
        for (int r = 0; r < 64; ++r) {
            const int s = r < 32 ? r : r - 64;  // fftfreq(64): 0..31, -32..-1
            rows[k][r] = fabsf((float)s / 64.0f) < thr;
        }
        for (int c = 0; c < 33; ++c) cols[k][c] = ((float)c / 64.0f) < thr;
    }
}

static int nearest_src(int dst, int in, int out) {
    // ATen legacy 'nearest': scale = (float)in / out; src = min((int)floorf(dst * scale), in - 1)
    if (out == in) return dst;
    if (out == 2 * in) return dst >> 1;
    const float scale = (float)in / (float)out;
    int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

static void resized_masks(int H, int W, uint8_t* rows /*3*H*/, uint8_t* cols /*3*(W/2+1)*/) {
    bool r64[3][64], c33[3][33];
    base_masks(r64, c33);
    const int wh = W / 2 + 1;
    for (int k = 0; k < 3; ++k) {
        for (int r = 0; r < H; ++r) rows[k * H + r] = r64[k][nearest_src(r, 64, H)];
        for (int c = 0; c < wh; ++c) cols[k * wh + c] = c33[k][nearest_src(c, 33, wh)];
    }
}

static int factorize(int n, int* rad) {
    int nf = 0;
    while (n % 4 == 0) { rad[nf++] = 4; n /= 4; }
    for (int p = 2; p * p <= n; ++p)
        while (n % p == 0) { rad[nf++] = p; n /= p; }
    if (n > 1) rad[nf++] = n;
    return nf;
}

// complex signals (channel pairs) per workgroup: the largest power of two (<= 16) whose `nbuf` LDS buffers of n points
// plus the twiddle table fit 160 KiB.  More signals per workgroup mean longer contiguous runs in the intermediate tensors
// (a run is P x 8 bytes) and more butterflies per pass and barrier; measured at 2088 x 2128 x 64 channels, batch 8:
// P = 1 / 1 (the old 32 KiB-per-buffer cap) 144 ms, P = 4 rows / 2 columns 96 ms.  `cap` (points per buffer) overrides.
static int pairs_per_group(int n, int nbuf, int cap) {
    int p = 16;
    while (p > 1 && ((size_t)(nbuf * p + 1) * n * sizeof(float2) > 160ul * 1024 || (cap > 0 && (long)n * p > cap))) p >>= 1;
    return p;
}
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && atoi(e) > 0 ? atoi(e) : dflt;  // 0: no override
}

// ---- device: Stockham pass -----------------------------------------------------------------
struct Radix {
    int nf;
    int r[24];
};
struct ColCuts {  // fbm_cols: output cut of the forward transform, operand cuts of the inverse ones [mask k][variant]
    int u;
    int t[9];
};
struct RowCuts {  // fbm_rows_inv: operand cut per band
    int t[3];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// One autosort pass of radix R over `src` (N points x P signals, signal index innermost).
// Each work item produces one output point of PP consecutive signals (index arithmetic and the twiddle of every
// butterfly term are shared by them).  `premul`, when set, scales source point i by premul[i] (the row mask,
// applied while reading the spectrum).
template <bool INV, int PP>
__device__ __forceinline__ void fft_pass_pp(const float2* __restrict__ src, float2* __restrict__ dst, int N, int R, int Ns,
                                            int P, const float2* __restrict__ tw, const float* __restrict__ premul) {
    const int NR = N / R;
    const int tstep = N / (Ns * R);
    const int PG = P / PP;
    for (int it = threadIdx.x; it < N * PG; it += blockDim.x) {
        const int pg = it % PG;
        const int o = it / PG;
        const int k = o % Ns;
        const int u = (o / Ns) % R;
        const int j = (o / (Ns * R)) * Ns + k;
        const int q = k * tstep + u * NR;  // < N
        float2 acc[PP];
#pragma unroll
        for (int i = 0; i < PP; ++i) acc[i] = make_float2(0.f, 0.f);
        int tq = 0;
        for (int t = 0; t < R; ++t) {
            float2 w = tw[tq];
            if (INV) w.y = -w.y;
            const int si = j + t * NR;
            if (premul) {
                const float m = premul[si];
                w.x *= m;
                w.y *= m;
            }
            const float2* sp = src + si * P + pg * PP;
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const float2 v = sp[i];
                acc[i].x += v.x * w.x - v.y * w.y;
                acc[i].y += v.x * w.y + v.y * w.x;
            }
            tq += q;
            if (tq >= N) tq -= N;
        }
        float2* dp = dst + o * P + pg * PP;
#pragma unroll
        for (int i = 0; i < PP; ++i) dp[i] = acc[i];
    }
}

// Odd prime radices beyond 7 (config 3's 2088 = 8 * 9 * 29 and 2128 = 16 * 7 * 19 point sides): each work item produces
// the output pair (u, R - u) of one butterfly of one signal.  With the twiddled operands z_t = x_t w^(t k tstep),
//   X[u], X[R-u] = z_0 + sum_{t=1}^{(R-1)/2} (z_t + z_{R-t}) cos(2 pi u t / R)  -/+  i (z_t - z_{R-t}) sin(2 pi u t / R)
// (signs for the forward transform): (R - 1) / 2 real-coefficient terms for two outputs instead of R complex terms for
// one -- a quarter of the multiplications and half of the LDS reads of fft_pass_pp, with (R + 1) / 2 work items per
// butterfly (1080 for a 2088-point signal, where whole butterflies would leave 72).
// Pruning (the masks keep at most half of a spectrum): as the LAST pass of a forward transform (Ns = N / R) output u covers
// result indices [u N/R, (u + 1) N/R), and output pairs u >= ucut lie wholly on frequencies no mask keeps: they are neither
// computed nor stored.  As the FIRST pass of an inverse transform (Ns = 1) operand t covers source indices
// [t N/R, (t + 1) N/R), and the pairs t >= tcut are zero in that band: they are not read.  The host computes both cuts
// (sr_fbm_plan_create) such that tcut <= ucut on one axis, so no inverse pass reads what the forward pass left unwritten.
// 0 = no pruning.
template <bool INV>
__device__ __forceinline__ void fft_pass_prime(const float2* __restrict__ src, float2* __restrict__ dst, int N, int R, int Ns,
                                               int P, const float2* __restrict__ tw, const float* __restrict__ premul, int tcut,
                                               int ucut) {
    const int NR = N / R;
    const int tstep = NR / Ns;
    const int half = (R + 1) / 2;  // outputs u = 0 .. half-1 (u = 0 pairs with itself)
    const int nt = tcut > 0 && tcut < half ? tcut : half;
    const int nu = ucut > 0 && ucut < half ? ucut : half;
    const bool notw = Ns == 1;  // first pass: every pass twiddle is 1
    for (int it = threadIdx.x; it < NR * nu * P; it += blockDim.x) {
        const int p = it % P;
        const int o = it / P;
        const int u = o % nu;
        const int j = o / nu;
        const int jhi = j / Ns, k = j - jhi * Ns;
        const int ks = k * tstep;  // pass twiddle of operand t: tw[t * ks], t * ks < N
        // complex numbers as float2v: sums / differences / scaled accumulations are one packed instruction each
        const float2v* srcv = (const float2v*)src;
        float2v a = srcv[j * P + p];
        if (premul) a *= premul[j];
        float2v bsum = {0.f, 0.f};
        int m = 0;      // (u * t) mod R
        int q1 = 0, q2 = R * ks;  // t * ks and (R - t) * ks
        for (int t = 1; t < nt; ++t) {
            m += u;
            if (m >= R) m -= R;
            const int s1 = j + t * NR, s2 = j + (R - t) * NR;
            float2v z1 = srcv[s1 * P + p], z2 = srcv[s2 * P + p];
            if (premul) {
                z1 *= premul[s1];
                z2 *= premul[s2];
            }
            if (!notw) {  // block-uniform
                q1 += ks;
                q2 -= ks;
                float2 w1 = tw[q1], w2 = tw[q2 >= N ? q2 - N : q2];
                if (INV) {
                    w1.y = -w1.y;
                    w2.y = -w2.y;
                }
                // (x + i y)(c + i s) = x (c, s) + y (-s, c)
                const float2v c1 = {w1.x, w1.y}, r1 = {-w1.y, w1.x}, c2 = {w2.x, w2.y}, r2 = {-w2.y, w2.x};
                z1 = __builtin_elementwise_fma(z1.yy, r1, z1.xx * c1);
                z2 = __builtin_elementwise_fma(z2.yy, r2, z2.xx * c2);
            }
            const float2 cs = tw[m * NR];  // (cos, -sin) of 2 pi u t / R
            const float2v cc = {cs.x, cs.x}, ss = {cs.y, cs.y};
            a = __builtin_elementwise_fma(z1 + z2, cc, a);
            bsum = __builtin_elementwise_fma(z1 - z2, ss, bsum);
        }
        // e^{-i th} -> z1 (cos - i sin) + z2 (cos + i sin) = (z1 + z2) cos - i (z1 - z2) sin; cs.y = -sin, so
        // -i (z1 - z2) sin = i (z1 - z2) cs.y = i * bsum (forward);  i * (x + i y) = (-y, x).
        float2 up, um;
        if (!INV) {
            up = make_float2(a.x - bsum.y, a.y + bsum.x);  // X[u]
            um = make_float2(a.x + bsum.y, a.y - bsum.x);  // X[R - u]
        } else {
            up = make_float2(a.x + bsum.y, a.y - bsum.x);
            um = make_float2(a.x - bsum.y, a.y + bsum.x);
        }
        float2* dp = dst + (jhi * R * Ns + k) * P + p;
        dp[u * Ns * P] = up;
        if (u) dp[(R - u) * Ns * P] = um;
    }
}

// The same pass with whole radix-R butterflies per work item (R a compile-time constant): butterfly j of PP consecutive
// signals reads its R operands once (source points j + t N/R), applies the pass twiddles w^(t k tstep), and writes the R
// outputs of the small DFT to ((j / Ns) R + u) Ns + k, k = j % Ns.  One integer division per butterfly instead of three per
// output point, and one LDS read per point and pass instead of R.
template <bool INV, int PP, int R>
__device__ __forceinline__ void fft_pass_bfly(const float2* __restrict__ src, float2* __restrict__ dst, int N, int Ns, int P,
                                              const float2* __restrict__ tw, const float* __restrict__ premul) {
    const int NR = N / R;
    const int tstep = NR / Ns;
    const int PG = P / PP;  // a power of two
    const int pgs = 31 - __clz(PG);
    for (int it = threadIdx.x; it < NR * PG; it += blockDim.x) {
        const int pg = it & (PG - 1);
        const int j = it >> pgs;
        const int jhi = j / Ns, k = j - jhi * Ns;
        float2 v[R][PP];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int si = j + t * NR;
            float2 w = tw[t * k * tstep];  // < N
            if (INV) w.y = -w.y;
            if (premul) {
                const float m = premul[si];
                w.x *= m;
                w.y *= m;
            }
            const float2* sp = src + si * P + pg * PP;
#pragma unroll
            for (int i = 0; i < PP; ++i) v[t][i] = t == 0 && !premul ? sp[i] : cmul(sp[i], w);
        }
        float2* dp = dst + (jhi * R * Ns + k) * P + pg * PP;
        if constexpr (R == 2) {
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                dp[i] = make_float2(v[0][i].x + v[1][i].x, v[0][i].y + v[1][i].y);
                dp[Ns * P + i] = make_float2(v[0][i].x - v[1][i].x, v[0][i].y - v[1][i].y);
            }
        } else if constexpr (R == 4) {
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const float2 a = make_float2(v[0][i].x + v[2][i].x, v[0][i].y + v[2][i].y);
                const float2 b = make_float2(v[0][i].x - v[2][i].x, v[0][i].y - v[2][i].y);
                const float2 c = make_float2(v[1][i].x + v[3][i].x, v[1][i].y + v[3][i].y);
                float2 d = make_float2(v[1][i].x - v[3][i].x, v[1][i].y - v[3][i].y);
                d = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);  // times w_4 (forward: -i)
                dp[i] = make_float2(a.x + c.x, a.y + c.y);
                dp[Ns * P + i] = make_float2(b.x + d.x, b.y + d.y);
                dp[2 * Ns * P + i] = make_float2(a.x - c.x, a.y - c.y);
                dp[3 * Ns * P + i] = make_float2(b.x - d.x, b.y - d.y);
            }
        } else {
            // one output at a time (R x PP operands stay in registers, PP accumulators); w_R^(t u) comes from the LDS table
#pragma unroll 1
            for (int u = 0; u < R; ++u) {
                float2 acc[PP];
#pragma unroll
                for (int i = 0; i < PP; ++i) acc[i] = v[0][i];
                int m = 0;
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    m += u;
                    if (m >= R) m -= R;
                    float2 o = tw[m * NR];
                    if (INV) o.y = -o.y;
#pragma unroll
                    for (int i = 0; i < PP; ++i) {
                        acc[i].x += v[t][i].x * o.x - v[t][i].y * o.y;
                        acc[i].y += v[t][i].x * o.y + v[t][i].y * o.x;
                    }
                }
#pragma unroll
                for (int i = 0; i < PP; ++i) dp[u * Ns * P + i] = acc[i];
            }
        }
    }
}

template <bool INV, int PP>
__device__ __forceinline__ void fft_pass_r(const float2* __restrict__ src, float2* __restrict__ dst, int N, int R, int Ns,
                                           int P, const float2* __restrict__ tw, const float* __restrict__ premul, int tcut,
                                           int ucut) {
    switch (R) {  // block-uniform
        case 2: fft_pass_bfly<INV, PP, 2>(src, dst, N, Ns, P, tw, premul); break;
        case 3: fft_pass_bfly<INV, PP, 3>(src, dst, N, Ns, P, tw, premul); break;
        case 4: fft_pass_bfly<INV, PP, 4>(src, dst, N, Ns, P, tw, premul); break;
        case 5: fft_pass_bfly<INV, PP, 5>(src, dst, N, Ns, P, tw, premul); break;
        case 7: fft_pass_bfly<INV, (PP < 2 ? PP : 2), 7>(src, dst, N, Ns, P, tw, premul); break;  // 7 x PP operands in registers
        default:
            if (R & 1) fft_pass_prime<INV>(src, dst, N, R, Ns, P, tw, premul, tcut, ucut);  // odd primes: conjugate output pairs
            else fft_pass_pp<INV, PP>(src, dst, N, R, Ns, P, tw, premul);
    }
}

template <bool INV>
__device__ __forceinline__ void fft_pass(const float2* __restrict__ src, float2* __restrict__ dst, int N, int R, int Ns,
                                         int P, const float2* __restrict__ tw, const float* __restrict__ premul, int tcut,
                                         int ucut) {
    // signals per work item: as many as leave at least one butterfly for every thread of the block
    const int bf = N / R, nt = blockDim.x;
    if ((P & 3) == 0 && bf * (P >> 2) >= nt) fft_pass_r<INV, 4>(src, dst, N, R, Ns, P, tw, premul, tcut, ucut);
    else if ((P & 1) == 0 && bf * (P >> 1) >= nt) fft_pass_r<INV, 2>(src, dst, N, R, Ns, P, tw, premul, tcut, ucut);
    else fft_pass_r<INV, 1>(src, dst, N, R, Ns, P, tw, premul, tcut, ucut);
}

// Runs all passes; returns the buffer that holds the result.  bufA holds the input.  When
// `keep_input` is set the input buffer is never written (needs a third buffer bufC).
// (forced inline: as an out-of-line function -- which hipcc chose once the pass switch grew by four more radices -- the
// kernels took a stack, passed the Radix kernel argument by reference and faulted in the 33-point test)
// `cut`: pruning of the prime pass (fft_pass_prime) -- the output cut of the last pass of a forward transform, the operand
// cut of the first pass of an inverse transform; 0 = none.
template <bool INV>
__device__ __forceinline__ float2* fft_run(float2* bufA, float2* bufB, float2* bufC, bool keep_input, int N, const Radix& rx, int P,
                           const float2* tw, const float* premul, int cut) {
    float2* src = bufA;
    float2* dst = bufB;
    int Ns = 1;
    for (int f = 0; f < rx.nf; ++f) {
        fft_pass<INV>(src, dst, N, rx.r[f], Ns, P, tw, f == 0 ? premul : nullptr, (INV && f == 0) ? cut : 0,
                      (!INV && f == rx.nf - 1) ? cut : 0);
        __syncthreads();
        Ns *= rx.r[f];
        float2* nxt = (src == bufA && keep_input) ? bufC : src;
        src = dst;
        dst = nxt;
    }
    return src;
}

extern __shared__ __attribute__((aligned(16))) char smem[];

// both channels of a complex signal in one access (channel-contiguous layouts with an even channel count)
template <typename T>
struct PairIO;
template <>
struct PairIO<float> {
    static __device__ __forceinline__ float2 ld(const float* p) { return *(const float2*)p; }
    static __device__ __forceinline__ void st(float* p, float2 v) { *(float2*)p = v; }
};
template <>
struct PairIO<bf16_t> {
    static __device__ __forceinline__ float2 ld(const bf16_t* p) {
        const uint32_t u = *(const uint32_t*)p;
        return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
    }
    static __device__ __forceinline__ void st(bf16_t* p, float2 v) {
        *(uint32_t*)p = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    }
};

// ---- kernel 1: forward FFT along W ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void fbm_rows_fwd(View x, int ups, int pair, int H, int W, int C, int PT, int P, Radix rx,
                                                    const float2* __restrict__ tw, const int* __restrict__ cols, int n2,
                                                    float2* __restrict__ t1, int ucut) {
    const int lp = 31 - __clz(P);  // P is a power of two (pairs_per_group)
    float2* bufA = (float2*)smem;
    float2* bufB = bufA + W * P;
    float2* twl = bufB + W * P;  // twiddles staged in LDS: every butterfly term reads one
    for (int i = threadIdx.x; i < W; i += blockDim.x) twl[i] = tw[i];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int p0 = blockIdx.y * P;
    const int Ws = W >> ups;
    #pragma unroll 8
    for (int it = threadIdx.x; it < W * P; it += blockDim.x) {
        const int p = it & (P - 1), w = it >> lp;
        const int c = 2 * (p0 + p);
        const long pix = (long)(h >> ups) * Ws + (w >> ups);
        float2 v = make_float2(0.f, 0.f);
        if (pair) {
            if (c < C) v = PairIO<T>::ld((const T*)x.p + voff(x, b, pix, c));
        } else {
            if (c < C) v.x = Elem<T>::ld((const T*)x.p + voff(x, b, pix, c));
            if (c + 1 < C) v.y = Elem<T>::ld((const T*)x.p + voff(x, b, pix, c + 1));
        }
        bufA[it] = v;
    }
    __syncthreads();
    const float2* res = fft_run<false>(bufA, bufB, nullptr, false, W, rx, P, twl, nullptr, ucut);
    // t1[b][idx][h][PT]
    #pragma unroll 8
    for (int it = threadIdx.x; it < n2 * P; it += blockDim.x) {
        const int p = it & (P - 1), idx = it >> lp;
        if (p0 + p < PT) t1[(((long)b * n2 + idx) * H + h) * PT + p0 + p] = res[cols[idx] * P + p];
    }
}

// ---- kernel 1b: forward FFT along W of a nearest-2x upsampled tensor ---------------------------
// The source row s (N = W / 2 pixels) appears as rows 2 hs, 2 hs + 1 of the upsampled image, every pixel twice:
//     FFT_W(up s)[c] = (1 + w^c) FFT_N(s)[c mod N],   w = exp(-2 pi i / W),
// so one N-point transform per SOURCE row serves two rows of T1: a quarter of the butterflies of fbm_rows_fwd on the
// upsampled rows (conv_up1 / conv_up2 of a tiled image: 522 -> 1044, 1064 -> 2128).  tw = twiddles of length W.
template <typename T>
__global__ __launch_bounds__(1024) void fbm_rows_fwd_up(View x, int pair, int H, int W, int C, int PT, int P, Radix rx,
                                                       const float2* __restrict__ tw, const int* __restrict__ cols, int n2,
                                                       float2* __restrict__ t1, int ucut) {
    const int lp = 31 - __clz(P);
    const int N = W >> 1, Hs = H >> 1;
    float2* bufA = (float2*)smem;
    float2* bufB = bufA + N * P;
    float2* twl = bufB + N * P;  // twiddles of the N-point transform: every second one of the W-point table
    for (int i = threadIdx.x; i < N; i += blockDim.x) twl[i] = tw[2 * i];
    const int b = blockIdx.x / Hs, hs = blockIdx.x % Hs;
    const int p0 = blockIdx.y * P;
    #pragma unroll 8
    for (int it = threadIdx.x; it < N * P; it += blockDim.x) {
        const int p = it & (P - 1), w = it >> lp;
        const int c = 2 * (p0 + p);
        const long pix = (long)hs * N + w;
        float2 v = make_float2(0.f, 0.f);
        if (pair) {
            if (c < C) v = PairIO<T>::ld((const T*)x.p + voff(x, b, pix, c));
        } else {
            if (c < C) v.x = Elem<T>::ld((const T*)x.p + voff(x, b, pix, c));
            if (c + 1 < C) v.y = Elem<T>::ld((const T*)x.p + voff(x, b, pix, c + 1));
        }
        bufA[it] = v;
    }
    __syncthreads();
    const float2* res = fft_run<false>(bufA, bufB, nullptr, false, N, rx, P, twl, nullptr, ucut);
    #pragma unroll 4
    for (int it = threadIdx.x; it < n2 * P; it += blockDim.x) {
        const int p = it & (P - 1), idx = it >> lp;
        if (p0 + p >= PT) continue;
        const int c = cols[idx];
        const float2 v = res[(c >= N ? c - N : c) * P + p];
        const float2 w = tw[c];
        const float2 o = make_float2(v.x + v.x * w.x - v.y * w.y, v.y + v.x * w.y + v.y * w.x);
        float2* d = t1 + (((long)b * n2 + idx) * H + 2 * hs) * PT + p0 + p;
        d[0] = o;
        d[PT] = o;
    }
}

// ---- kernel 2: FFT along H, row masks, inverse FFT along H ----------------------------------
__global__ __launch_bounds__(1024) void fbm_cols(int H, int W, int PT, int P, Radix rx, Radix rxi, ColCuts cuts,
                                                const float2* __restrict__ tw,
                                                const int* __restrict__ colinfo, int n2, int n4, int n8,
                                                const float* __restrict__ rowmask, const float2* __restrict__ t1,
                                                float2* __restrict__ y2, float2* __restrict__ y4,
                                                float2* __restrict__ y8, float* __restrict__ pooled, int pooled_ld,
                                                int C, int bmask) {
    const int lp = 31 - __clz(P);  // P is a power of two
    float2* buf0 = (float2*)smem;
    float2* buf1 = buf0 + H * P;
    float2* buf2 = buf1 + H * P;
    float2* twl = buf2 + H * P;
    for (int i = threadIdx.x; i < H; i += blockDim.x) twl[i] = tw[i];
    const int b = blockIdx.x / n2, idx = blockIdx.x % n2;
    const int p0 = blockIdx.y * P;
    const float2* col = t1 + ((long)b * n2 + idx) * H * PT;
    #pragma unroll 8
    for (int it = threadIdx.x; it < H * P; it += blockDim.x) {
        const int p = it & (P - 1), h = it >> lp;
        buf0[it] = (p0 + p < PT) ? col[(long)h * PT + p0 + p] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    float2* S = fft_run<false>(buf0, buf1, nullptr, false, H, rx, P, twl, nullptr, cuts.u);
    float2* other = (S == buf0) ? buf1 : buf0;
    const int variant = colinfo[idx * 4 + 0];
    if (pooled && idx == 0 && threadIdx.x < P) {  // column 0 comes first: S[0] = plane sums of the pair
        const int c = 2 * (p0 + threadIdx.x);
        const float inv = 1.0f / ((float)H * (float)W);
        if (c < C) pooled[(long)b * pooled_ld + c] = S[threadIdx.x].x * inv;
        if (c + 1 < C) pooled[(long)b * pooled_ld + c + 1] = S[threadIdx.x].y * inv;
    }
    const int kidx[3] = {idx, colinfo[idx * 4 + 1], colinfo[idx * 4 + 2]};
    const int nk[3] = {n2, n4, n8};
    float2* ys[3] = {y2, y4, y8};
    for (int k = 0; k < 3; ++k) {
        if (kidx[k] < 0 || !((bmask >> k) & 1)) continue;  // block-uniform
        const float* rm = rowmask + (k * 3 + variant) * H;
        const int tc = k == 0 ? (variant == 0 ? cuts.t[0] : (variant == 1 ? cuts.t[1] : cuts.t[2]))
                              : (k == 1 ? (variant == 0 ? cuts.t[3] : (variant == 1 ? cuts.t[4] : cuts.t[5]))
                                        : (variant == 0 ? cuts.t[6] : (variant == 1 ? cuts.t[7] : cuts.t[8])));
        const float2* res = fft_run<true>(S, buf2, other, true, H, rxi, P, twl, rm, tc);
        float2* y = ys[k];
        #pragma unroll 8
        for (int it = threadIdx.x; it < H * P; it += blockDim.x) {
            const int p = it & (P - 1), h = it >> lp;
            if (p0 + p < PT) y[(((long)b * H + h) * nk[k] + kidx[k]) * PT + p0 + p] = res[it];
        }
        __syncthreads();
    }
}

// ---- kernel 3: inverse FFT along W, write the three low-pass tensors ------------------------
template <typename T>
__global__ __launch_bounds__(1024) void fbm_rows_inv(View l2, View l4, View l8, int pair, int H, int W, int C, int PT, int P,
                                                    Radix rx, RowCuts cuts, const float2* __restrict__ tw,
                                                    const int* __restrict__ cols2, const int* __restrict__ cols4,
                                                    const int* __restrict__ cols8, int n2, int n4, int n8,
                                                    const float2* __restrict__ y2, const float2* __restrict__ y4,
                                                    const float2* __restrict__ y8, int bmask) {
    const int lp = 31 - __clz(P);  // P is a power of two
    float2* bufA = (float2*)smem;
    float2* bufB = bufA + W * P;
    float2* twl = bufB + W * P;
    for (int i = threadIdx.x; i < W; i += blockDim.x) twl[i] = tw[i];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int p0 = blockIdx.y * P;
    const float norm = 1.0f / ((float)H * (float)W);
    const int nk[3] = {n2, n4, n8};
    const int* ck[3] = {cols2, cols4, cols8};
    const float2* ys[3] = {y2, y4, y8};
    for (int k = 0; k < 3; ++k) {
        if (!((bmask >> k) & 1)) continue;  // block-uniform
        // selected field by field: an array of View structs indexed by the runtime k lived in scratch memory (128 B per lane)
        const View out = k == 0 ? l2 : (k == 1 ? l4 : l8);
        for (int it = threadIdx.x; it < W * P; it += blockDim.x) bufA[it] = make_float2(0.f, 0.f);
        __syncthreads();
        const float2* yr = ys[k] + ((long)b * H + h) * nk[k] * PT;
        #pragma unroll 8
        for (int it = threadIdx.x; it < nk[k] * P; it += blockDim.x) {
            const int p = it & (P - 1), i = it >> lp;
            if (p0 + p < PT) bufA[ck[k][i] * P + p] = yr[(long)i * PT + p0 + p];
        }
        __syncthreads();
        const float2* res = fft_run<true>(bufA, bufB, nullptr, false, W, rx, P, twl, nullptr, k == 0 ? cuts.t[0] : (k == 1 ? cuts.t[1] : cuts.t[2]));
        #pragma unroll 8
        for (int it = threadIdx.x; it < W * P; it += blockDim.x) {
            const int p = it & (P - 1), w = it >> lp;
            const int c = 2 * (p0 + p);
            const float2 v = res[it];
            const long pix = (long)h * W + w;
            if (pair) {
                if (c < C) PairIO<T>::st((T*)out.p + voff(out, b, pix, c), make_float2(v.x * norm, v.y * norm));
            } else {
                if (c < C) Elem<T>::st((T*)out.p + voff(out, b, pix, c), v.x * norm);
                if (c + 1 < C) Elem<T>::st((T*)out.p + voff(out, b, pix, c + 1), v.y * norm);
            }
        }
        __syncthreads();
    }
}

int g_force_generic_fft = 0;  // test hook
int g_no_mixed_fft = 0;       // test / A-B hook
int g_last_lowpass_path = -1; // test hook: kernel family of the last launch (0 generic FFT, 1 power-of-two FFT, 2 mixed-radix FFT, 3 DFT-as-GEMM)
namespace fast {
extern int g_no_ups_shortcut;  // debug hook (fbm_fast.hip): full-length transforms of upsampled sources
}
// bf16 planes of shapes that fbm_mfma.hip covers take its DFT-as-GEMM kernels; SR_AMD_LOWPASS=fft (or the debug hook
// below) keeps them on the FFT kernels.
int lowpass_default() {
    const char* e = getenv("SR_AMD_LOWPASS");
    return (e && !strcmp(e, "fft")) ? 1 : 0;
}
int g_force_fft_lowpass = lowpass_default();

static Radix make_radix(const int* r, int nf) {
    Radix x{};
    x.nf = nf;
    for (int i = 0; i < nf; ++i) x.r[i] = r[i];
    return x;
}
static Radix reversed(const Radix& a) {
    Radix x{};
    x.nf = a.nf;
    for (int i = 0; i < a.nf; ++i) x.r[i] = a.r[a.nf - 1 - i];
    return x;
}

int fbm_lowpass_launch(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8,
                       float* pooled, int pooled_ld, void* ws, size_t ws_bytes, hipStream_t st, int bmask) {
    const int H = pl->H, W = pl->W, C = x.C;
    const int PT = (C + 1) / 2;
    if (ws_bytes < sr_fbm_workspace_bytes(pl, B, C)) return SR_ERR_WORKSPACE;
    if (!g_force_generic_fft && !g_force_fft_lowpass) {
        const int rc = fbm_lowpass_mfma(pl, dtype, B, x, ups, l2, l4, l8, pooled, pooled_ld, ws, st, bmask);
        if (rc != SR_NOT_ELIGIBLE) { g_last_lowpass_path = 3; return rc; }
    }
    if (!g_force_generic_fft) {
        int rc = fbm_lowpass_pow2(pl, dtype, B, x, ups, l2, l4, l8, pooled, pooled_ld, ws, st, bmask);
        if (rc != SR_NOT_ELIGIBLE) { g_last_lowpass_path = 1; return rc; }
        if (!g_no_mixed_fft) rc = fbm_lowpass_mixed(pl, dtype, B, x, ups, l2, l4, l8, pooled, pooled_ld, ws, st, bmask);
        if (rc != SR_NOT_ELIGIBLE) { g_last_lowpass_path = 2; return rc; }
    }
    g_last_lowpass_path = 0;
    const int n2 = pl->n[0], n4 = pl->n[1], n8 = pl->n[2];
    float2* t1 = (float2*)ws;
    float2* y2 = t1 + (size_t)B * n2 * H * PT;
    float2* y4 = y2 + (size_t)B * H * n2 * PT;
    float2* y8 = y4 + (size_t)B * H * n4 * PT;
    const Radix rw = make_radix(pl->radW, pl->nfW), rh = make_radix(pl->radH, pl->nfH);
    const Radix rwi = reversed(rw), rhi = reversed(rh);  // inverse transforms run the largest radix first (pruned operands)
    ColCuts cc;
    cc.u = pl->ucutH;
    for (int i = 0; i < 9; ++i) cc.t[i] = pl->tcutH[i];
    RowCuts rc3;
    for (int i = 0; i < 3; ++i) rc3.t[i] = pl->tcutW[i];
    const int Pw = pl->Pw, Ph = pl->Ph;
    const size_t lds_w = (2ul * W * Pw + W) * sizeof(float2), lds_h = (3ul * H * Ph + H) * sizeof(float2);  // buffers + twiddles
    dim3 g1(B * H, cdiv(PT, Pw)), g2(B * n2, cdiv(PT, Ph));
    // threads per block: about 16 waves per CU whatever the LDS footprint allows to be resident (1, 2 or >= 4 blocks)
    static const int forced = getenv("SR_AMD_GENERIC_THREADS") ? atoi(getenv("SR_AMD_GENERIC_THREADS")) : 0;
    auto threads_for = [](size_t lds) {
        if (forced == 256 || forced == 512 || forced == 1024) return forced;
        const size_t resident = (160ul * 1024) / lds;
        return resident >= 4 ? 256 : (resident >= 2 ? 512 : 1024);
    };
    const int tw_ = threads_for(lds_w), th_ = threads_for(lds_h);
    const int pair_in = C % 2 == 0 && view_vec_ok(x, 2);
    const int pair_out = C % 2 == 0 && view_vec_ok(l2, 2) && view_vec_ok(l4, 2) && view_vec_ok(l8, 2);
    if (dtype == SR_F32) {
        SR_LDS_ATTR(lds_w, fbm_rows_fwd<float>);
        SR_LDS_ATTR(lds_w, fbm_rows_inv<float>);
    } else {
        SR_LDS_ATTR(lds_w, fbm_rows_fwd<bf16_t>);
        SR_LDS_ATTR(lds_w, fbm_rows_inv<bf16_t>);
    }
    SR_LDS_ATTR(lds_h, fbm_cols);
    if (ups && pl->nfWu > 0 && !fast::g_no_ups_shortcut) {  // one half-length transform per source row
        const int Pu = pl->Pwu;
        const size_t lds_u = (2ul * (W / 2) * Pu + W / 2) * sizeof(float2);
        const Radix ru = make_radix(pl->radWu, pl->nfWu);
        const dim3 gu(B * (H / 2), cdiv(PT, Pu));
        const int tu = threads_for(lds_u);
        if (dtype == SR_F32) {
            SR_LDS_ATTR(lds_u, fbm_rows_fwd_up<float>);
            fbm_rows_fwd_up<float><<<gu, tu, lds_u, st>>>(x, pair_in, H, W, C, PT, Pu, ru, pl->tw_w, pl->cols, n2, t1, pl->ucutWu);
        } else {
            SR_LDS_ATTR(lds_u, fbm_rows_fwd_up<bf16_t>);
            fbm_rows_fwd_up<bf16_t><<<gu, tu, lds_u, st>>>(x, pair_in, H, W, C, PT, Pu, ru, pl->tw_w, pl->cols, n2, t1, pl->ucutWu);
        }
    } else if (dtype == SR_F32)
        fbm_rows_fwd<float><<<g1, tw_, lds_w, st>>>(x, ups, pair_in, H, W, C, PT, Pw, rw, pl->tw_w, pl->cols, n2, t1, pl->ucutW);
    else
        fbm_rows_fwd<bf16_t><<<g1, tw_, lds_w, st>>>(x, ups, pair_in, H, W, C, PT, Pw, rw, pl->tw_w, pl->cols, n2, t1, pl->ucutW);
    SR_LAUNCH_CHECK();
    fbm_cols<<<g2, th_, lds_h, st>>>(H, W, PT, Ph, rh, rhi, cc, pl->tw_h, pl->colinfo, n2, n4, n8, pl->rowmask, t1, y2, y4, y8,
                                     pooled, pooled_ld, C, bmask);
    SR_LAUNCH_CHECK();
    if (dtype == SR_F32)
        fbm_rows_inv<float><<<g1, tw_, lds_w, st>>>(l2, l4, l8, pair_out, H, W, C, PT, Pw, rwi, rc3, pl->tw_w, pl->cols, pl->cols4,
                                                    pl->cols8, n2, n4, n8, y2, y4, y8, bmask);
    else
        fbm_rows_inv<bf16_t><<<g1, tw_, lds_w, st>>>(l2, l4, l8, pair_out, H, W, C, PT, Pw, rwi, rc3, pl->tw_w, pl->cols, pl->cols4,
                                                     pl->cols8, n2, n4, n8, y2, y4, y8, bmask);
    SR_LAUNCH_CHECK();
    return SR_OK;
}
}  // namespace sr

using namespace sr;

extern "C" void sr_debug_force_generic_fft(int on) { sr::g_force_generic_fft = on; }
extern "C" int sr_debug_last_lowpass_path(void) { return sr::g_last_lowpass_path; }
extern "C" void sr_debug_no_mixed_fft(int on) { sr::g_no_mixed_fft = on; }  // 1: the 1044 / 1064 / 2088 / 2128 sides take the generic FFT
namespace sr { extern int g_mfma_no_fold; }
extern "C" void sr_debug_mfma_no_fold(int on) { sr::g_mfma_no_fold = on; }
namespace sr { namespace fast { extern int g_no_ups_shortcut; } }
extern "C" void sr_debug_no_ups_shortcut(int on) { sr::fast::g_no_ups_shortcut = on; }
extern "C" void sr_debug_force_fft_lowpass(int on) { sr::g_force_fft_lowpass = on < 0 ? sr::lowpass_default() : on; }

extern "C" int sr_fbm_masks_host(int H, int W, uint8_t* rows_host, uint8_t* cols_host) {
    SR_REQUIRE(H > 0 && W > 0 && rows_host && cols_host);
    resized_masks(H, W, rows_host, cols_host);
    return SR_OK;
}

extern "C" int sr_fbm_plan_create(int H, int W, sr_fbm_plan** out) {
    SR_REQUIRE(out && H >= 1 && W >= 1 && H <= 4096 && W <= 4096);
    sr_fbm_plan* pl = (sr_fbm_plan*)calloc(1, sizeof(sr_fbm_plan));
    if (!pl) return SR_ERR_ARG;
    pl->H = H;
    pl->W = W;
    pl->nfW = factorize(W, pl->radW);
    pl->nfH = factorize(H, pl->radH);
    pl->Pw = pairs_per_group(W, 2, env_int("SR_AMD_GENERIC_CAP_W", 0));
    pl->Ph = pairs_per_group(H, 3, env_int("SR_AMD_GENERIC_CAP_H", 0));
    const int wh = W / 2 + 1;
    std::vector<uint8_t> rows(3 * H), cols(3 * wh);
    resized_masks(H, W, rows.data(), cols.data());
    // kept full-spectrum columns per mask, ascending; variant: 0 = rm[r], 1 = rm[-r], 2 = average
    std::vector<int> lists[3];
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < W; ++c) {
            const int ch = c <= W / 2 ? c : W - c;
            if (cols[k * wh + ch]) lists[k].push_back(c);
        }
    for (int k = 0; k < 3; ++k) pl->n[k] = (int)lists[k].size();
    pl->contiguous = 1;
    for (int k = 0; k < 3; ++k) {
        int a = 0;
        while (a <= W / 2 && cols[k * wh + a]) ++a;  // prefix of kept half-spectrum columns
        pl->half[k] = a;
        std::vector<int> want;
        for (int c = 0; c < W; ++c)
            if (c < a || c > W - a) want.push_back(c);
        if (want != lists[k] || a < 1 || 2 * a - 1 > W) pl->contiguous = 0;
    }
    if (pl->n[0] < 1) { free(pl); return SR_ERR_ARG; }
    std::vector<int> info(4 * lists[0].size());
    for (size_t i = 0; i < lists[0].size(); ++i) {
        const int c = lists[0][i];
        int variant;
        if (c == 0 || (W % 2 == 0 && c == W / 2)) variant = 2;
        else variant = c < (W + 1) / 2 ? 0 : 1;
        info[4 * i + 0] = variant;
        for (int k = 1; k < 3; ++k) {
            int pos = -1;
            for (size_t j = 0; j < lists[k].size(); ++j)
                if (lists[k][j] == c) pos = (int)j;
            info[4 * i + k] = pos;
        }
        info[4 * i + 3] = 0;
    }
    for (int k = 1; k < 3; ++k)  // masks must be nested, or a k=4/8 column would never be computed
        for (int c : lists[k]) {
            bool found = false;
            for (int c2 : lists[0]) found |= (c2 == c);
            if (!found) { free(pl); return SR_ERR_ARG; }
        }
    std::vector<float> rmask(9 * H);
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < H; ++r) {
            const float a = rows[k * H + r], bneg = rows[k * H + (H - r) % H];
            rmask[(k * 3 + 0) * H + r] = a;
            rmask[(k * 3 + 1) * H + r] = bneg;
            rmask[(k * 3 + 2) * H + r] = 0.5f * (a + bneg);
        }
    {   // generic FFT: cuts of the prime pass (fft_pass_prime).  used[i] != 0: index i of the spectrum carries weight.
        auto pair_cut = [](const std::vector<char>& used, int N, int R) {
            // 1 + the last conjugate pair (t, R - t), t < (R + 1) / 2, of N/R-index ranges that holds a used index
            const int nr = N / R, half = (R + 1) / 2;
            auto range_used = [&](int t) {
                for (int i = t * nr; i < (t + 1) * nr; ++i)
                    if (used[i]) return true;
                return false;
            };
            int cut = 1;
            for (int t = 1; t < half; ++t)
                if (range_used(t) || range_used(R - t)) cut = t + 1;
            return cut;
        };
        auto prime_radix = [](const int* rad, int nf) {
            const int R = nf > 0 ? rad[nf - 1] : 0;
            return (R > 7 && (R & 1)) ? R : 0;
        };
        const int Rw = prime_radix(pl->radW, pl->nfW), Rh = prime_radix(pl->radH, pl->nfH);
        pl->ucutW = pl->ucutH = 0;
        for (int i = 0; i < 3; ++i) pl->tcutW[i] = 0;
        for (int i = 0; i < 9; ++i) pl->tcutH[i] = 0;
        pl->nfWu = 0;
        pl->ucutWu = 0;
        if (W % 2 == 0 && H % 2 == 0 && W >= 4) {  // half-length row transform for nearest-2x upsampled sources
            const int N = W / 2;
            pl->nfWu = factorize(N, pl->radWu);
            pl->Pwu = pairs_per_group(N, 2, env_int("SR_AMD_GENERIC_CAP_W", 0));
            const int Ru = prime_radix(pl->radWu, pl->nfWu);
            if (Ru) {
                std::vector<char> used(N, 0);
                for (int c : lists[0]) used[c >= N ? c - N : c] = 1;
                pl->ucutWu = pair_cut(used, N, Ru);
            }
        }
        if (Rw) {
            for (int k = 0; k < 3; ++k) {
                std::vector<char> used(W, 0);
                for (int c : lists[k]) used[c] = 1;
                pl->tcutW[k] = pair_cut(used, W, Rw);
                if (k == 0) pl->ucutW = pl->tcutW[0];  // rows_fwd stores the columns of the widest mask
            }
        }
        if (Rh) {
            std::vector<char> any(H, 0);
            for (int kv = 0; kv < 9; ++kv) {
                std::vector<char> used(H, 0);
                for (int r = 0; r < H; ++r) {
                    used[r] = rmask[(size_t)kv * H + r] != 0.f;
                    any[r] |= used[r];
                }
                pl->tcutH[kv] = pair_cut(used, H, Rh);
            }
            pl->ucutH = pair_cut(any, H, Rh);  // >= every tcutH: the inverse passes read only what the forward pass wrote
        }
    }
    for (int k = 0; k < 3; ++k) {  // extent of band k along H over its three variants (fbm_fast.h prunes passes by it)
        int a = 1;
        for (int v = 0; v < 3; ++v)
            for (int r = 0; r < H; ++r)
                if (rmask[(size_t)(k * 3 + v) * H + r] != 0.f) a = std::max(a, std::min(r + 1, H - r + 1));  // r < a or r > H - a
        pl->row_half[k] = a;
    }
    std::vector<float2> tww(W), twh(H);
    for (int j = 0; j < W; ++j) tww[j] = make_float2((float)cos(-2.0 * M_PI * j / W), (float)sin(-2.0 * M_PI * j / W));
    for (int j = 0; j < H; ++j) twh[j] = make_float2((float)cos(-2.0 * M_PI * j / H), (float)sin(-2.0 * M_PI * j / H));
    // one device allocation, 256-byte aligned carve
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_tww = carve(W * sizeof(float2)), o_twh = carve(H * sizeof(float2));
    const size_t o_cols = carve(lists[0].size() * 4), o_info = carve(info.size() * 4);
    const size_t o_c4 = carve((lists[1].size() + 1) * 4), o_c8 = carve((lists[2].size() + 1) * 4);
    const size_t o_rm = carve(rmask.size() * 4);
    char* dev = nullptr;
    hipError_t e = hipMalloc((void**)&dev, off);
    if (e != hipSuccess) { free(pl); return hip_fail(e); }
    std::vector<char> host(off, 0);
    memcpy(&host[o_tww], tww.data(), W * sizeof(float2));
    memcpy(&host[o_twh], twh.data(), H * sizeof(float2));
    memcpy(&host[o_cols], lists[0].data(), lists[0].size() * 4);
    memcpy(&host[o_info], info.data(), info.size() * 4);
    if (!lists[1].empty()) memcpy(&host[o_c4], lists[1].data(), lists[1].size() * 4);
    if (!lists[2].empty()) memcpy(&host[o_c8], lists[2].data(), lists[2].size() * 4);
    memcpy(&host[o_rm], rmask.data(), rmask.size() * 4);
    e = hipMemcpy(dev, host.data(), off, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(dev); free(pl); return hip_fail(e); }
    pl->dev_block = dev;
    pl->tw_w = (float2*)(dev + o_tww);
    pl->tw_h = (float2*)(dev + o_twh);
    pl->cols = (int*)(dev + o_cols);
    pl->colinfo = (int*)(dev + o_info);
    pl->cols4 = (int*)(dev + o_c4);
    pl->cols8 = (int*)(dev + o_c8);
    pl->rowmask = (float*)(dev + o_rm);
    const int mrc = fbm_mfma_plan_build(pl, rows.data());
    if (mrc != SR_OK) { (void)hipFree(dev); free(pl); return mrc; }
    *out = pl;
    return SR_OK;
}

extern "C" void sr_fbm_plan_destroy(sr_fbm_plan* pl) {
    if (!pl) return;
    if (pl->dev_block) (void)hipFree(pl->dev_block);
    fbm_mfma_plan_free(pl);
    free(pl);
}

extern "C" size_t sr_fbm_workspace_bytes(const sr_fbm_plan* pl, int B, int C) {
    if (!pl || B < 1 || C < 1) return 0;
    const size_t PT = align_up((size_t)(C + 1) / 2, 16);  // the pow2 path stores whole pair groups
    const size_t n = (size_t)pl->n[0] * 2 + pl->n[1] + pl->n[2];
    return align_up((size_t)B * pl->H * n * PT * sizeof(float2), 256);
}

extern "C" int sr_fbm_lowpass(const sr_fbm_plan* pl, int dtype, int B, sr_view x, int ups, sr_view l2, sr_view l4,
                              sr_view l8, float* pooled, int pooled_ld, void* ws, size_t ws_bytes, void* stream) {
    SR_REQUIRE(pl && dtype_ok(dtype) && B > 0 && x.p && ws);
    // a band whose view has p == NULL is not wanted (the adjoint of the band mix needs one band per call)
    sr_view* outs[3] = {&l2, &l4, &l8};
    int bmask = 0, first = -1;
    for (int k = 0; k < 3; ++k)
        if (outs[k]->p) {
            bmask |= 1 << k;
            if (first < 0) first = k;
        }
    SR_REQUIRE(bmask != 0);
    for (int k = 0; k < 3; ++k)
        if (!outs[k]->p) *outs[k] = *outs[first];  // never written; keeps the layout checks of the kernels' launchers uniform
    SR_REQUIRE(x.C > 0 && l2.C == x.C && l4.C == x.C && l8.C == x.C);
    SR_REQUIRE(x.c0 >= 0 && l2.c0 >= 0 && l4.c0 >= 0 && l8.c0 >= 0);
    SR_REQUIRE(view_block_ok(x) && view_block_ok(l2) && view_block_ok(l4) && view_block_ok(l8));
    SR_REQUIRE(ups == 0 || (pl->H % 2 == 0 && pl->W % 2 == 0));
    SR_REQUIRE(!pooled || pooled_ld >= x.C);
    const long npix = (long)pl->H * pl->W;
    return fbm_lowpass_launch(pl, dtype, B, to_view(x, ups ? npix / 4 : npix), ups ? 1 : 0, to_view(l2, npix),
                              to_view(l4, npix), to_view(l8, npix), pooled, pooled_ld, ws, ws_bytes, (hipStream_t)stream, bmask);
}

// ---- the one entry point that allocates nothing --------------------------------------------------------------------
// For callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it):
// sr_fbm_lowpass with its workspace (the FFT / DFT intermediates T, S, U_k or T1, Y_k between the passes) in a
// file-scope device array instead of a caller buffer.  Each GPU holds its own copy of the array at its own address
// until the process ends; the kernels take that address as the `ws` argument exactly as in sr_fbm_lowpass.  Every
// intermediate is written by one pass before the next pass reads it, so nothing has to start at zero, but all calls
// on one GPU share the array: calls that overlap on one GPU (two streams, two threads) race.
static __device__ __align__(256) unsigned char sr_static_scratch[SR_STATIC_SCRATCH_BYTES];
static void* g_static_scratch_ptr[16];  // per device: address of that device's copy of the array
static std::mutex g_static_scratch_mutex;

static int static_scratch_for_current_device(void** out) {
    int dev = 0;
    SR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return SR_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (!g_static_scratch_ptr[dev]) {
        void* p = nullptr;  // in host code the array's name is a stand-in, not a device address: ask the runtime
        SR_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(sr_static_scratch)));
        if (!p) return SR_ERR_HIP;
        g_static_scratch_ptr[dev] = p;
    }
    *out = g_static_scratch_ptr[dev];
    return SR_OK;
}

extern "C" size_t sr_fbm_static_scratch_bytes(void) { return SR_STATIC_SCRATCH_BYTES; }

extern "C" int sr_fbm_lowpass_static(const sr_fbm_plan* pl, int dtype, int B, sr_view x, int ups, sr_view l2, sr_view l4,
                                     sr_view l8, float* pooled, int pooled_ld, void* stream) {
    SR_REQUIRE(pl && B > 0 && x.C > 0);
    if (sr_fbm_workspace_bytes(pl, B, x.C) > SR_STATIC_SCRATCH_BYTES) return SR_ERR_WORKSPACE;  // beyond the fixed maximum
    void* ws = nullptr;
    const int rc = static_scratch_for_current_device(&ws);
    if (rc != SR_OK) return rc;
    return sr_fbm_lowpass(pl, dtype, B, x, ups, l2, l4, l8, pooled, pooled_ld, ws, SR_STATIC_SCRATCH_BYTES, stream);
}
