// FBM low-pass planes, register-blocked fast path: powers of two (64 <= H, W <= 2048, fbm_fast.hip) and the mixed-radix
// sides of config 3's tiles (1044 = 4 9 29, 1064 = 8 7 19, 2088 = 8 9 29, 2128 = 16 7 19, fbm_fast_mixed.hip).  This
// header holds the kernels; the including source file names its sizes (SR_FAST_SIZES) and its entry point (SR_FAST_ENTRY).
//
// Same three-kernel decomposition as fbm.hip (rows forward -> columns forward/mask/inverse ->
// rows inverse, two channels per complex signal, channel pairs on consecutive lanes), but the FFT
// is a register-blocked Stockham: every thread owns whole butterflies (radix 8 / 4 / 2, and 3 / 7 / 9 / 19 / 29 with
// compile-time coefficients for the mixed-radix sizes), a pass is
//   [barrier] read R points per butterfly -> twiddle -> DFT_R in registers -> [barrier] write,
// in place in one LDS buffer of N x P complex (64 KiB), with the first pass fed straight from
// global memory and the last pass stored straight to global memory (no staging copies):
//   rows_fwd : reads the activation row (bf16/fp32 channel pairs), writes only the kept columns;
//   cols     : forward along H, then keeps each thread's first-pass operands of the spectrum in
//              registers so the three masked inverse transforms reuse one LDS buffer;
//   rows_inv : reads the sparse kept columns (zero elsewhere), writes the low-pass rows in the
//              activation dtype.
// Kept columns are the two contiguous runs [0, half) and (N - half, N) of the box masks.
#include <type_traits>

#include "common.h"
#include "fbm_plan.h"

namespace sr {
namespace fast {

constexpr int THREADS = 512;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[2]) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <bool INV>
struct Dft<4, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[4]) {
        const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        const float2 t2 = cadd(v[1], v[3]), t3 = rot90<INV>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[1] = cadd(t1, t3);
        v[2] = csub(t0, t2);
        v[3] = csub(t1, t3);
    }
};

template <bool INV>
struct Dft<8, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[8]) {
        float2 e[4] = {v[0], v[2], v[4], v[6]};
        float2 o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4, INV>::run(e);
        Dft<4, INV>::run(o);
        const float h = 0.70710678118654752440f;
        // w8^u * o[u], w8 = exp(-+ 2 pi i / 8)
        const float2 o1 = INV ? make_float2(h * (o[1].x - o[1].y), h * (o[1].x + o[1].y))
                              : make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
        const float2 o2 = rot90<INV>(o[2]);
        const float2 o3 = INV ? make_float2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y))
                              : make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
        v[0] = cadd(e[0], o[0]);
        v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);
        v[5] = csub(e[1], o1);
        v[2] = cadd(e[2], o2);
        v[6] = csub(e[2], o2);
        v[3] = cadd(e[3], o3);
        v[7] = csub(e[3], o3);
    }
};

// ---- odd radices: compile-time coefficients --------------------------------------------------------------------
// DFT_R of an odd R by conjugate pairs: with s_t = v_t + v_{R-t}, d_t = v_t - v_{R-t}, t = 1 .. (R-1)/2,
//     X[u], X[R-u] = (v_0 + sum_t s_t cos(2 pi u t / R))  -/+  i (sum_t d_t sin(2 pi u t / R))      (forward signs)
// -- (R-1)/2 real-coefficient terms for two outputs.  Every coefficient is a compile-time constant, so an operand that
// is a literal zero (a masked-out run of the spectrum, see ZM below) removes its terms, and an output that is never
// stored (OZ) removes its sums.
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double taylor_sin(double x) {  // |x| <= pi
    double term = x, sum = x;
    for (int n = 1; n < 24; ++n) {
        term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double taylor_cos(double x) {
    double term = 1.0, sum = 1.0;
    for (int n = 1; n < 24; ++n) {
        term *= -x * x / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}
constexpr double unit_angle(int m, int r) { return 2.0 * kPi * (m % r) / r - ((m % r) * 2 > r ? 2.0 * kPi : 0.0); }  // in (-pi, pi]
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
template <int R, bool INV>
struct DftOdd {
    static __device__ __forceinline__ void run(float2 (&v)[R]) {
        constexpr int HH = (R - 1) / 2;
        float2 s[HH + 1], d[HH + 1];
        const float2 x0 = v[0];
        float2 sum = x0;
        static_for<1, HH + 1>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            s[t] = cadd(v[t], v[R - t]);
            d[t] = csub(v[t], v[R - t]);
            sum = cadd(sum, s[t]);
        });
        v[0] = sum;
        static_for<1, HH + 1>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            float ax = x0.x, ay = x0.y, bx = 0.f, by = 0.f;
            static_for<1, HH + 1>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr float c = (float)taylor_cos(unit_angle(u * t, R)), sn = (float)taylor_sin(unit_angle(u * t, R));
                ax += s[t].x * c;
                ay += s[t].y * c;
                bx += d[t].x * sn;
                by += d[t].y * sn;
            });
            // forward: a - i b; inverse: a + i b
            const float2 lo = make_float2(ax + by, ay - bx), hi = make_float2(ax - by, ay + bx);
            v[u] = INV ? hi : lo;
            v[R - u] = INV ? lo : hi;
        });
    }
};
template <bool INV> struct Dft<3, INV> : DftOdd<3, INV> {};
template <bool INV> struct Dft<7, INV> : DftOdd<7, INV> {};
template <bool INV> struct Dft<9, INV> : DftOdd<9, INV> {};
template <bool INV> struct Dft<19, INV> : DftOdd<19, INV> {};
template <bool INV> struct Dft<29, INV> : DftOdd<29, INV> {};

// Radix of the next pass over a remaining length m.  Forward transforms take the power-of-two factors first and the
// largest odd factor LAST (the outputs of a last pass are contiguous runs of the spectrum: the runs no mask keeps are
// neither computed nor stored); inverse transforms take the largest odd factor FIRST (the operands of a first pass are
// contiguous runs of the spectrum: the masked-out runs are not read).
// Powers of two: 8, 4, 2 either way.  (Radix 16 where it saves a pass -- 2048 = 8 16 16, 2128 = 19 7 16 -- was measured:
// the row kernels 6-8 % faster, the column kernels, which also hold the spectrum's first-pass operands, 6-17 % slower at
// the 128 registers two workgroups per CU leave them; nothing in sum.)
constexpr int pick_pow2(int m) { return (m % 8 == 0 && m != 16) ? 8 : (m % 4 == 0 ? 4 : 2); }
template <bool INV>
constexpr int pick_radix(int m) {
    if (INV) return m % 29 == 0 ? 29 : (m % 19 == 0 ? 19 : (m % 7 == 0 ? 7 : (m % 9 == 0 ? 9 : (m % 3 == 0 ? 3 : pick_pow2(m)))));
    return m % 2 == 0 ? pick_pow2(m & -m) : (m % 9 == 0 ? 9 : (m % 3 == 0 ? 3 : (m % 7 == 0 ? 7 : (m % 19 == 0 ? 19 : 29))));
}
constexpr int first_radix(int n) { return pick_radix<true>(n); }  // of an inverse transform

// Work item i of a pass with NBJ butterflies per signal -> (signal p, butterfly j).
//   BLK = false (NHWC, channels contiguous per pixel): p fastest, so a wave touches P*8 contiguous bytes.
//   BLK = true  (channel-blocked [C/8][H][W][8]): the 4 pairs of a channel block fastest, then the
//               butterfly, so a wave's global accesses run along the image row of one channel block.
template <int P, int NBJ, bool BLK>
__device__ __forceinline__ void item_to_pj(int i, int& p, int& j) {
    if (BLK) {
        p = (i & 3) + 4 * (i / (4 * NBJ));
        j = (i >> 2) % NBJ;
    } else {
        p = i % P;
        j = i / P;
    }
}
// LDS image of the N x P complex working set, matching the lane order above (conflict-free ds access)
template <int N, int P, bool BLK>
__device__ __forceinline__ int lds_idx(int idx, int p) {
    return BLK ? (((p >> 2) * N + idx) * 4 + (p & 3)) : (idx * P + p);
}

// One Stockham pass.  load(idx, p, q, t) / store(idx, p, v) address point idx of signal p.
// ZM: bit t set = operand t of every butterfly of this (first) pass is zero (a masked-out range of the spectrum); the
// load is dropped and, with -fno-signed-zeros, so is every butterfly operation it feeds.
// OZ: bit u set = output u of every butterfly of this (last) pass is never used; it is neither stored nor computed.
template <int N, int P, int R, int NS, bool INV, bool BLK, bool LOAD_LDS, bool STORE_LDS, int ZM, int OZ, typename LoadF, typename StoreF>
__device__ __forceinline__ void pass(const float2* __restrict__ twl, LoadF load, StoreF store) {
    constexpr int NR = N / R;
    constexpr int NB = NR * P;
    constexpr int IT = (NB + THREADS - 1) / THREADS;
    constexpr int TSTEP = N / (NS * R);
    float2 v[IT][R];
    if (LOAD_LDS) __syncthreads();
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int i = threadIdx.x + q * THREADS;
        if (NB % THREADS != 0 && i >= NB) continue;
        int p, j;
        item_to_pj<P, NR, BLK>(i, p, j);
        const int k = j % NS;
#pragma unroll
        for (int t = 0; t < R; ++t) v[q][t] = ((ZM >> t) & 1) ? make_float2(0.f, 0.f) : load(j + t * NR, p, q, t);
        if (NS > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) {
                float2 w = twl[k * t * TSTEP];
                if (INV) w.y = -w.y;
                v[q][t] = cmulf(v[q][t], w);
            }
        }
        Dft<R, INV>::run(v[q]);
    }
    if (STORE_LDS) __syncthreads();
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int i = threadIdx.x + q * THREADS;
        if (NB % THREADS != 0 && i >= NB) continue;
        int p, j;
        item_to_pj<P, NR, BLK>(i, p, j);
        const int k = j % NS;
        const int base = (j / NS) * NS * R + k;
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (!((OZ >> u) & 1)) store(base + u * NS, p, v[q][u]);
    }
}

// All passes of a length-N transform: first pass loads through `lf`, last pass stores through `sl`,
// everything in between lives in `buf` (N x P complex, in place).
// LAST_LDS: the caller's `sl` writes into `buf` (needs the read/write barrier).
template <int N, int P, bool INV, bool BLK, int NS, bool LAST_LDS, int ZM = 0, int OZ = 0, typename LoadFirst, typename StoreLast>
__device__ __forceinline__ void fft(float2* __restrict__ buf, const float2* __restrict__ twl, LoadFirst lf, StoreLast sl) {
    constexpr int M = N / NS;
    constexpr int R = pick_radix<INV>(M);
    constexpr bool first = (NS == 1), last = (M == R);
    auto lds_load = [&](int idx, int p, int, int) { return buf[lds_idx<N, P, BLK>(idx, p)]; };
    auto lds_store = [&](int idx, int p, float2 v) { buf[lds_idx<N, P, BLK>(idx, p)] = v; };
    if constexpr (first && last)
        pass<N, P, R, NS, INV, BLK, false, true, ZM, OZ>(twl, lf, sl);
    else if constexpr (first)
        pass<N, P, R, NS, INV, BLK, false, true, ZM, 0>(twl, lf, lds_store);
    else if constexpr (last)
        pass<N, P, R, NS, INV, BLK, true, LAST_LDS, 0, OZ>(twl, lds_load, sl);
    else
        pass<N, P, R, NS, INV, BLK, true, true, 0, 0>(twl, lds_load, lds_store);
    if constexpr (!last) fft<N, P, INV, BLK, NS * R, LAST_LDS, 0, OZ>(buf, twl, lf, sl);
}

extern __shared__ __attribute__((aligned(16))) char smem_fast[];

// Twiddles exp(-2 pi i j / N) are staged in LDS behind the working set -- unless that pushes a workgroup past half of a
// CU's 160 KiB (the 2088 / 2128-point sides at 4 pairs: 68 + 17 KiB): those read the 17 KiB table through L1 / L2 instead
// and two workgroups share a CU.
constexpr bool twiddles_in_lds(int n, int p) { return (size_t)(n * p + n) * sizeof(float2) <= 80 * 1024; }
constexpr size_t lds_bytes(int n, int p) { return (size_t)n * p * sizeof(float2) + (twiddles_in_lds(n, p) ? (size_t)n * sizeof(float2) : 0); }
template <int N, int P>
__device__ __forceinline__ const float2* stage_twiddles(const float2* __restrict__ tw, float2* twl) {
    if constexpr (!twiddles_in_lds(N, P)) return tw;
    for (int i = threadIdx.x; i < N; i += THREADS) twl[i] = tw[i];
    __syncthreads();
    return twl;
}

template <typename T>
struct Pair;
template <>
struct Pair<float> {
    static __device__ __forceinline__ float2 ld(const float* p) { return *(const float2*)p; }
    static __device__ __forceinline__ void st(float* p, float2 v) { *(float2*)p = v; }
};
template <>
struct Pair<bf16_t> {
    static __device__ __forceinline__ float2 ld(const bf16_t* p) {
        const uint32_t u = *(const uint32_t*)p;
        return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
    }
    static __device__ __forceinline__ void st(bf16_t* p, float2 v) {
        *(uint32_t*)p = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    }
};

// position of full-spectrum column c in the kept list [0,a) u (N-a, N), or -1
template <int N>
__device__ __forceinline__ int kept_pos(int c, int a) {
    return c < a ? c : (c > N - a ? c - (N - (2 * a - 1)) : -1);
}

// Intermediates are stored per group of Q pairs: T1[b][pg][pos][h][Q], Y_k[b][pg][h][pos][Q].
struct Inter {
    int Q, NPG, QL;  // pairs per group (power of two), groups per sample, log2(Q)
};
// offsets inside one sample (32-bit: a sample's intermediates are < 2^31 elements, checked on the host)
__device__ __forceinline__ unsigned t1_off(const Inter& it, int gp, int pos, int h, int n2, int H) {
    return ((((unsigned)(gp >> it.QL) * n2 + pos) * H + h) << it.QL) + (gp & (it.Q - 1));
}
__device__ __forceinline__ unsigned y_off(const Inter& it, int gp, int h, int pos, int nk, int H) {
    return ((((unsigned)(gp >> it.QL) * H + h) * nk + pos) << it.QL) + (gp & (it.Q - 1));
}

// Workgroup -> (x, y) of the launch grid.  With groups of fewer than 16 pairs an intermediate row is narrower than a
// 128-byte line, and the workgroups that complete a line are neighbours along x: the hardware deals consecutive
// workgroups to different XCDs (different L2s, each left with a partial line), so those launches walk the grid
// XCD-contiguously instead.
__device__ __forceinline__ void block_xy(const Inter& it, unsigned& bx, unsigned& by) {
    bx = blockIdx.x;
    by = blockIdx.y;
    if (it.Q < 16) {
        const unsigned t = xcd_remap(by * gridDim.x + bx, gridDim.x * gridDim.y);
        bx = t % gridDim.x;
        by = t / gridDim.x;
    }
}

// Bit t of the result: first-pass operand range [t * N/R, (t + 1) * N/R) holds no kept index of [0, a) u (N - a, N).
template <int N>
__host__ __device__ inline int zero_mask(int a) {
    constexpr int R0 = first_radix(N), NR0 = N / R0;
    int m = 0;
#pragma unroll
    for (int t = 0; t < R0; ++t)
        if (t * NR0 >= a && (t + 1) * NR0 - 1 <= N - a) m |= 1 << t;
    return m;
}
// The kernels below take the zero masks of the three bands as template arguments.  PZ = true: the masks of the STANDARD
// band extents -- the reference's 64-point base masks keep 16 / 9 / 5 sixty-fourths of an axis on either side of DC
// (fbm.hip resized_masks), so band k of a length-N axis lies inside [0, a) u (N - a, N) with a = ceil(N {16, 9, 5} / 64);
// for a radix-8 first pass that is {0x3c, 0x3c, 0x7e}: the middle 4 / 4 / 6 eighths of the spectrum are masked out.  The
// host checks the plan's real extents against these bounds and picks the instance (PZ = false prunes nothing).  One
// specialised body per band keeps the code inside the instruction cache (dispatching among several bodies per band at
// run time did not).
constexpr int band_amax(int n, int k) { return (n * (k == 0 ? 16 : (k == 1 ? 9 : 5)) + 63) / 64; }
constexpr int range_mask(int n, int r, int a) {  // bit t: [t n/r, (t + 1) n/r) holds no index of [0, a) u (n - a, n)
    int m = 0;
    for (int t = 0; t < r; ++t)
        if (t * (n / r) >= a && (t + 1) * (n / r) - 1 <= n - a) m |= 1 << t;
    return m;
}
template <int N, bool PZ>
constexpr int band_zm(int k) { return PZ ? range_mask(N, first_radix(N), band_amax(N, k)) : 0; }
// Radix of the last pass of a forward transform, and the mask of its outputs that lie wholly outside the widest band
// when PZ holds (forward transforms neither compute nor store them).
constexpr int last_radix(int n) { return pick_radix<false>(n) == n ? n : last_radix(n / pick_radix<false>(n)); }
template <int N, bool PZ>
constexpr int fwd_oz() { return PZ ? range_mask(N, last_radix(N), band_amax(N, 0)) : 0; }
// waves per SIMD the register budget is set for: radix-29 butterflies and the unpruned fallback instances need more than 128
constexpr int min_waves(int n, bool pz) { return (n % 29 == 0 || (!pz && (n & (n - 1)) != 0)) ? 2 : 4; }


// ---- rows forward -------------------------------------------------------------------------
template <typename T, int N, int P, bool BLK, bool PZ>
__global__ __launch_bounds__(THREADS) void rows_fwd(View x, int ups, int H, int PT, Inter im,
                                                    const float2* __restrict__ tw, int a2, float2* __restrict__ t1) {
    float2* buf = (float2*)smem_fast;
    const float2* twl = stage_twiddles<N, P>(tw, buf + N * P);
    unsigned bx, by;
    block_xy(im, bx, by);
    const int b = bx / H, h = bx % H;
    const int p0 = by * P;
    const int n2 = 2 * a2 - 1;
    const unsigned rowpix = (unsigned)(h >> ups) * (N >> ups);
    const T* xb = (const T*)x.p + (long)b * x.bs;
    float2* t1b = t1 + ((long)b * im.NPG * n2 * H << im.QL);
    auto lf = [&](int idx, int p, int, int) {
        return (p0 + p < PT) ? Pair<T>::ld(xb + voff32(x, rowpix + (idx >> ups), 2 * (p0 + p))) : make_float2(0.f, 0.f);
    };
    auto sl = [&](int idx, int p, float2 v) {
        const int pos = kept_pos<N>(idx, a2);
        if (pos >= 0 && p0 + p < PT) t1b[t1_off(im, p0 + p, pos, h, n2, H)] = v;
    };
    fft<N, P, false, BLK, 1, false, 0, fwd_oz<N, PZ>()>(buf, twl, lf, sl);
}

// ---- rows forward of a nearest-2x upsampled tensor -----------------------------------------------
// The source row s (N = W / 2 pixels) appears as rows 2h, 2h + 1 of the upsampled image, every pixel twice:
//     FFT_W(up(s))[k] = (1 + w^k) FFT_N(s)[k mod N],   w = exp(-2 pi i / W),
// so one N-point transform per SOURCE row serves two rows of the upsampled image (a quarter of the butterflies of
// rows_fwd on the upsampled rows); T1 keeps the H / 2 distinct rows and cols<.., UPS = true> applies the same identity
// along H.  tw = twiddles of length W (every second one is the N-point table).
template <typename T, int N, int P, bool BLK>
__global__ __launch_bounds__(THREADS) void rows_fwd_up(View x, int H, int PT, Inter im, const float2* __restrict__ tw, int a2,
                                                       float2* __restrict__ t1) {
    constexpr int W = 2 * N;
    float2* buf = (float2*)smem_fast;
    float2* twl = buf + N * P;
    for (int i = threadIdx.x; i < N; i += THREADS) twl[i] = tw[2 * i];
    __syncthreads();
    unsigned bx, by;
    block_xy(im, bx, by);
    const int Hs = H >> 1;
    const int b = bx / Hs, hs = bx % Hs;
    const int p0 = by * P;
    const int n2 = 2 * a2 - 1;
    const unsigned rowpix = (unsigned)hs * N;
    const T* xb = (const T*)x.p + (long)b * x.bs;
    float2* t1b = t1 + ((long)b * im.NPG * n2 * Hs << im.QL);  // T1 holds the Hs distinct rows only (see cols<.., UPS>)
    auto lf = [&](int idx, int p, int, int) {
        return (p0 + p < PT) ? Pair<T>::ld(xb + voff32(x, rowpix + idx, 2 * (p0 + p))) : make_float2(0.f, 0.f);
    };
    auto sl = [&](int idx, int p, float2 v) {
        if (p0 + p >= PT) return;
        const float2 w = tw[idx];  // w^idx; w^(idx + N) = -w^idx
        if (idx < a2) {
            const float2 o = make_float2(v.x + v.x * w.x - v.y * w.y, v.y + v.x * w.y + v.y * w.x);
            t1b[t1_off(im, p0 + p, idx, hs, n2, Hs)] = o;
        }
        if (idx > N - a2) {
            const float2 o = make_float2(v.x - (v.x * w.x - v.y * w.y), v.y - (v.x * w.y + v.y * w.x));
            const int pos = kept_pos<W>(idx + N, a2);
            t1b[t1_off(im, p0 + p, pos, hs, n2, Hs)] = o;
        }
    };
    fft<N, P, false, BLK, 1, false, 0, 0>(buf, twl, lf, sl);
}

// ---- columns: forward, mask, inverse x3 ------------------------------------------------------
template <int N, int P, bool BLK, bool PZ, bool UPS = false>
__global__ __launch_bounds__(THREADS, min_waves(N, PZ)) void cols(int W, int PT, Inter im, const float2* __restrict__ tw, int a2, int a4,
                                                int a8, const float* __restrict__ rowmask,
                                                const float2* __restrict__ t1, float2* __restrict__ y2,
                                                float2* __restrict__ y4, float2* __restrict__ y8,
                                                float* __restrict__ pooled, int pooled_ld, int C, int bmask) {
    constexpr int H = N;
    float2* buf = (float2*)smem_fast;
    const float2* twl = stage_twiddles<N, P>(tw, buf + N * P);
    const int n2 = 2 * a2 - 1;
    unsigned bx, by;
    block_xy(im, bx, by);
    const int b = bx / n2, idx = bx % n2;
    const int p0 = by * P;
    const int col = idx < a2 ? idx : idx + (W - n2);  // full-spectrum column of this block
    // UPS: T1 holds the H / 2 rows of a tensor that is nearest-upsampled along H as well; its column spectrum is
    // (1 + w^k) F[k mod H/2] with F the half-length transform (w = exp(-2 pi i / H)).  The half-length twiddles live in the
    // upper half of `buf` while the forward transform uses the lower one.
    constexpr int NF = UPS ? N / 2 : N;
    const float2* t1b = t1 + ((long)b * im.NPG * n2 * NF << im.QL);
    auto lf = [&](int h, int p, int, int) {
        return (p0 + p < PT) ? t1b[t1_off(im, p0 + p, idx, h, n2, NF)] : make_float2(0.f, 0.f);
    };
    auto s_lds = [&](int r, int p, float2 v) { buf[lds_idx<NF, P, BLK>(r, p)] = v; };
    // forward: the last pass stores back into LDS
    if constexpr (UPS) {
        float2* twh = buf + NF * P;
        for (int i = threadIdx.x; i < NF; i += THREADS) twh[i] = twl[2 * i];
        __syncthreads();
        fft<NF, P, false, BLK, 1, true, 0, 0>(buf, twh, lf, s_lds);
    } else {
        fft<N, P, false, BLK, 1, true, 0, fwd_oz<N, PZ>()>(buf, twl, lf, s_lds);
    }
    __syncthreads();
    if (pooled && idx == 0 && threadIdx.x < P && p0 + threadIdx.x < PT) {
        const int c = 2 * (p0 + threadIdx.x);
        const float inv = (UPS ? 2.0f : 1.0f) / ((float)H * (float)W);
        const float2 dc = buf[lds_idx<NF, P, BLK>(0, threadIdx.x)];
        if (c < C) pooled[(long)b * pooled_ld + c] = dc.x * inv;
        if (c + 1 < C) pooled[(long)b * pooled_ld + c + 1] = dc.y * inv;
    }
    // keep this thread's first-pass operands of the spectrum in registers
    constexpr int R0 = first_radix(N);
    constexpr int NR0 = N / R0;
    constexpr int NB = NR0 * P;
    constexpr int IT = (NB + THREADS - 1) / THREADS;
    float2 sreg[IT][R0];
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int i = threadIdx.x + q * THREADS;
        int p, j;
        item_to_pj<P, NR0, BLK>(i, p, j);
#pragma unroll
        for (int t = 0; t < R0; ++t) {
            const int k = j + t * NR0;
            float2 v = (NB % THREADS == 0 || i < NB) ? buf[lds_idx<NF, P, BLK>(UPS ? (k % NF) : k, p)] : make_float2(0.f, 0.f);
            if constexpr (UPS) {
                const float2 w = twl[k];
                v = make_float2(v.x + v.x * w.x - v.y * w.y, v.y + v.x * w.y + v.y * w.x);
            }
            sreg[q][t] = v;
        }
    }
    const int variant = (col == 0 || 2 * col == W) ? 2 : (2 * col < W ? 0 : 1);
    const int ak[3] = {a2, a4, a8};
    float2* ys[3] = {y2, y4, y8};
    auto band = [&](auto kc) {  // k as a compile-time constant: it selects the pruned first pass
        constexpr int k = decltype(kc)::value;
        const int a = ak[k];
        const int nk = 2 * a - 1;
        const int pos = col < a ? col : (col > W - a ? col - (W - nk) : -1);
        if (pos < 0 || !((bmask >> k) & 1)) return;  // block-uniform
        const float* rm = rowmask + (k * 3 + variant) * H;
        // the first inverse pass takes the saved operands, scaled by the row mask
        auto lm = [&](int r, int, int q, int t) {
            const float m = rm[r];
            return make_float2(sreg[q][t].x * m, sreg[q][t].y * m);
        };
        float2* y = ys[k] + ((long)b * im.NPG * H * nk << im.QL);
        auto sl = [&](int h, int p, float2 v) {
            if (p0 + p < PT) y[y_off(im, p0 + p, h, pos, nk, H)] = v;
        };
        fft<N, P, true, BLK, 1, false, band_zm<N, PZ>(k)>(buf, twl, lm, sl);
    };
    band(std::integral_constant<int, 0>{});
    band(std::integral_constant<int, 1>{});
    band(std::integral_constant<int, 2>{});
}

// ---- rows inverse -------------------------------------------------------------------------
template <typename T, int N, int P, bool BLK, bool PZ>
__global__ __launch_bounds__(THREADS, min_waves(N, PZ)) void rows_inv(View l2, View l4, View l8, int H, int PT, Inter im,
                                                    const float2* __restrict__ tw, int a2, int a4, int a8,
                                                    const float2* __restrict__ y2, const float2* __restrict__ y4,
                                                    const float2* __restrict__ y8, int bmask) {
    float2* buf = (float2*)smem_fast;
    const float2* twl = stage_twiddles<N, P>(tw, buf + N * P);
    unsigned bx, by;
    block_xy(im, bx, by);
    const int b = bx / H, h = bx % H;
    const int p0 = by * P;
    const float norm = 1.0f / ((float)H * (float)N);
    const int ak[3] = {a2, a4, a8};
    const float2* ys[3] = {y2, y4, y8};
    const View outs[3] = {l2, l4, l8};
    const unsigned rowpix = (unsigned)h * N;
    // With the pruned masks a thread needs only 4 + 4 + 2 first-pass operands per butterfly for the three bands: all
    // of them are fetched up front, so the three transforms run back to back without another trip to memory (and no
    // load is queued behind the stores of the previous band).
    constexpr int R0 = first_radix(N), NR0 = N / R0, NB0 = NR0 * P, IT0 = (NB0 + THREADS - 1) / THREADS;
    float2 pre[3][IT0][R0];
    auto fetch = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (!((bmask >> k) & 1)) return;  // block-uniform
        const int a = ak[k];
        const int nk = 2 * a - 1;
        const float2* y = ys[k] + ((long)b * im.NPG * H * nk << im.QL);
        constexpr int ZMK = band_zm<N, PZ>(k);  // constant-evaluated here, not per thread
#pragma unroll
        for (int q = 0; q < IT0; ++q) {
            const int i = threadIdx.x + q * THREADS;
            int p, j;
            item_to_pj<P, NR0, BLK>(i, p, j);
#pragma unroll
            for (int t = 0; t < R0; ++t) {
                if ((ZMK >> t) & 1) continue;
                const int pos = kept_pos<N>(j + t * NR0, a);
                const bool ok = pos >= 0 && p0 + p < PT && (NB0 % THREADS == 0 || i < NB0);
                pre[k][q][t] = ok ? y[y_off(im, p0 + p, h, pos, nk, H)] : make_float2(0.f, 0.f);
            }
        }
    };
    if constexpr (PZ) {
        fetch(std::integral_constant<int, 0>{});
        fetch(std::integral_constant<int, 1>{});
        fetch(std::integral_constant<int, 2>{});
    }
    auto band = [&](auto kc) {  // k as a compile-time constant: it selects the pruned first pass
        constexpr int k = decltype(kc)::value;
        if (!((bmask >> k) & 1)) return;  // block-uniform
        const int a = ak[k];
        const int nk = 2 * a - 1;
        const float2* y = ys[k] + ((long)b * im.NPG * H * nk << im.QL);
        auto lf = [&](int c, int p, int q, int t) {
            if constexpr (PZ) return pre[k][q][t];
            const int pos = kept_pos<N>(c, a);
            return (pos >= 0 && p0 + p < PT) ? y[y_off(im, p0 + p, h, pos, nk, H)] : make_float2(0.f, 0.f);
        };
        const View o = outs[k];
        T* ob = (T*)o.p + (long)b * o.bs;
        auto sl = [&](int w, int p, float2 v) {
            if (p0 + p < PT) Pair<T>::st(ob + voff32(o, rowpix + w, 2 * (p0 + p)), make_float2(v.x * norm, v.y * norm));
        };
        fft<N, P, true, BLK, 1, false, band_zm<N, PZ>(k)>(buf, twl, lf, sl);
    };
    band(std::integral_constant<int, 0>{});
    band(std::integral_constant<int, 1>{});
    band(std::integral_constant<int, 2>{});
}

// ---- host dispatch ---------------------------------------------------------------------------
constexpr int pairs_for(int n) { return n <= 544 ? 16 : (n <= 1088 ? 8 : 4); }  // N x P complex <= 68 KiB
inline int pairs_for_rt(int n) { return pairs_for(n); }

#ifdef SR_FAST_DEFINE_GLOBALS
int g_no_ups_shortcut = 0;  // debug hook
#else
extern int g_no_ups_shortcut;
#endif

template <typename T, int N, bool BLK>
static int launch_rows_fwd(View x, int ups, bool shortcut, int B, int H, int PT, Inter im, const sr_fbm_plan* pl, float2* t1,
                           hipStream_t st) {
    if constexpr (N >= 128) {
        if (shortcut) {  // one half-length transform per source row
            constexpr int NS = N / 2, PS = pairs_for(NS);
            const size_t ldss = (size_t)NS * PS * sizeof(float2) + (size_t)NS * sizeof(float2);
            SR_LDS_ATTR(ldss, rows_fwd_up<T, NS, PS, BLK>);
            rows_fwd_up<T, NS, PS, BLK><<<dim3(B * (H / 2), cdiv(PT, PS)), THREADS, ldss, st>>>(x, H, PT, im, pl->tw_w, pl->half[0], t1);
            SR_LAUNCH_CHECK();
            return SR_OK;
        }
    }
    constexpr int P = pairs_for(N);
    const size_t lds = lds_bytes(N, P);
    // kept columns of the widest mask inside the outer quarters: the last pass skips the middle half of its outputs
    const bool pz = pl->half[0] <= band_amax(N, 0) && fwd_oz<N, true>() != 0;
    if (pz) {
        SR_LDS_ATTR(lds, rows_fwd<T, N, P, BLK, true>);
        rows_fwd<T, N, P, BLK, true><<<dim3(B * H, cdiv(PT, P)), THREADS, lds, st>>>(x, ups, H, PT, im, pl->tw_w, pl->half[0], t1);
    } else {
        SR_LDS_ATTR(lds, rows_fwd<T, N, P, BLK, false>);
        rows_fwd<T, N, P, BLK, false><<<dim3(B * H, cdiv(PT, P)), THREADS, lds, st>>>(x, ups, H, PT, im, pl->tw_w, pl->half[0], t1);
    }
    SR_LAUNCH_CHECK();
    return SR_OK;
}

template <typename T, int N, bool BLK>
static int launch_rows_inv(View l2, View l4, View l8, int B, int H, int PT, Inter im, const sr_fbm_plan* pl,
                           const float2* y2, const float2* y4, const float2* y8, int bmask, hipStream_t st) {
    constexpr int P = pairs_for(N);
    const size_t lds = lds_bytes(N, P);
    // standard band extents along W: the operand ranges of the first pass that band_zm names are empty
    const bool pz = pl->half[0] <= band_amax(N, 0) && pl->half[1] <= band_amax(N, 1) && pl->half[2] <= band_amax(N, 2) &&
                    (band_zm<N, true>(0) | band_zm<N, true>(1) | band_zm<N, true>(2)) != 0;
    if (pz) {
        SR_LDS_ATTR(lds, rows_inv<T, N, P, BLK, true>);
        rows_inv<T, N, P, BLK, true><<<dim3(B * H, cdiv(PT, P)), THREADS, lds, st>>>(l2, l4, l8, H, PT, im, pl->tw_w, pl->half[0],
                                                                            pl->half[1], pl->half[2], y2, y4, y8, bmask);
    } else {
        SR_LDS_ATTR(lds, rows_inv<T, N, P, BLK, false>);
        rows_inv<T, N, P, BLK, false><<<dim3(B * H, cdiv(PT, P)), THREADS, lds, st>>>(l2, l4, l8, H, PT, im, pl->tw_w, pl->half[0],
                                                                            pl->half[1], pl->half[2], y2, y4, y8, bmask);
    }
    SR_LAUNCH_CHECK();
    return SR_OK;
}

template <int N, bool BLK>
static int launch_cols(bool shortcut, int B, int W, int PT, int C, Inter im, const sr_fbm_plan* pl, const float2* t1, float2* y2,
                       float2* y4, float2* y8, float* pooled, int pooled_ld, int bmask, hipStream_t st) {
    constexpr int P = pairs_for(N);
    const size_t lds = lds_bytes(N, P);
    // every variant of every row mask must lie inside the standard band extents
    const bool pz = pl->row_half[0] <= band_amax(N, 0) && pl->row_half[1] <= band_amax(N, 1) && pl->row_half[2] <= band_amax(N, 2) &&
                    (band_zm<N, true>(0) | band_zm<N, true>(1) | band_zm<N, true>(2) | fwd_oz<N, true>()) != 0;
    const dim3 grid(B * pl->n[0], cdiv(PT, P));
#define SR_COLS(PZ_, UPS_)                                                                                                   \
    do {                                                                                                                     \
        SR_LDS_ATTR(lds, cols<N, P, BLK, PZ_, UPS_>); \
        cols<N, P, BLK, PZ_, UPS_><<<grid, THREADS, lds, st>>>(W, PT, im, pl->tw_h, pl->half[0], pl->half[1], pl->half[2],    \
                                                             pl->rowmask, t1, y2, y4, y8, pooled, pooled_ld, C, bmask);      \
    } while (0)
    if constexpr (N >= 128) {
        if (shortcut) {
            if (pz) SR_COLS(true, true);
            else SR_COLS(false, true);
            SR_LAUNCH_CHECK();
            return SR_OK;
        }
    }
    if (pz) SR_COLS(true, false);
    else SR_COLS(false, false);
#undef SR_COLS
    SR_LAUNCH_CHECK();
    return SR_OK;
}

#define SR_FAST_CASE(n_, CALL) case n_: { constexpr int NN = n_; CALL; } break;
#define SR_FAST_OK(n_, CALL) case n_:
#define SR_DISPATCH_N(n, CALL)            \
    switch (n) {                          \
        SR_FAST_SIZES(SR_FAST_CASE, CALL) \
        default: return SR_NOT_ELIGIBLE;  \
    }
static bool size_ok(int n) {
    switch (n) {
        SR_FAST_SIZES(SR_FAST_OK, 0)
            return true;
        default: return false;
    }
}

template <typename T>
static int run_all(const sr_fbm_plan* pl, int B, View x, int ups, View l2, View l4, View l8, float* pooled,
                   int pooled_ld, void* ws, bool blk_x, bool blk_o, int bmask, hipStream_t st) {
    const int H = pl->H, W = pl->W, C = x.C;
    const int PT = C / 2;
    Inter im;
    im.Q = pairs_for_rt(W) < pairs_for_rt(H) ? pairs_for_rt(W) : pairs_for_rt(H);
    im.NPG = cdiv(PT, im.Q);
    im.QL = im.Q == 16 ? 4 : (im.Q == 8 ? 3 : 2);
    const size_t PTP = (size_t)im.NPG * im.Q;
    const int n2 = pl->n[0], n4 = pl->n[1];
    float2* t1 = (float2*)ws;
    float2* y2 = t1 + (size_t)B * n2 * H * PTP;
    float2* y4 = y2 + (size_t)B * H * n2 * PTP;
    float2* y8 = y4 + (size_t)B * H * n4 * PTP;
    const bool blk_c = im.Q == 4;  // 4-pair groups: butterflies of one group on consecutive lanes
    // nearest-2x upsampled source: half-length transforms along both axes, T1 keeps the H / 2 distinct rows
    const bool shortcut = ups == 1 && H >= 128 && W >= 128 && pl->half[0] <= W / 2 && !g_no_ups_shortcut;
    int rc = SR_OK;
    if (blk_x) { SR_DISPATCH_N(W, rc = (launch_rows_fwd<T, NN, true>(x, ups, shortcut, B, H, PT, im, pl, t1, st))); }
    else { SR_DISPATCH_N(W, rc = (launch_rows_fwd<T, NN, false>(x, ups, shortcut, B, H, PT, im, pl, t1, st))); }
    if (rc != SR_OK) return rc;
    if (blk_c) { SR_DISPATCH_N(H, rc = (launch_cols<NN, true>(shortcut, B, W, PT, C, im, pl, t1, y2, y4, y8, pooled, pooled_ld, bmask, st))); }
    else { SR_DISPATCH_N(H, rc = (launch_cols<NN, false>(shortcut, B, W, PT, C, im, pl, t1, y2, y4, y8, pooled, pooled_ld, bmask, st))); }
    if (rc != SR_OK) return rc;
    if (blk_o) { SR_DISPATCH_N(W, rc = (launch_rows_inv<T, NN, true>(l2, l4, l8, B, H, PT, im, pl, y2, y4, y8, bmask, st))); }
    else { SR_DISPATCH_N(W, rc = (launch_rows_inv<T, NN, false>(l2, l4, l8, B, H, PT, im, pl, y2, y4, y8, bmask, st))); }
    return rc;
}

}  // namespace fast

int SR_FAST_ENTRY(const sr_fbm_plan* pl, int dtype, int B, View x, int ups, View l2, View l4, View l8, float* pooled,
                     int pooled_ld, void* ws, hipStream_t st, int bmask) {
    using namespace fast;
    const int H = pl->H, W = pl->W, C = x.C;
    if (!pl->contiguous || !size_ok(H) || !size_ok(W) || (C & 1)) return SR_NOT_ELIGIBLE;
    if ((long)x.bs >= (1L << 31) || (long)l2.bs >= (1L << 31) || (long)l4.bs >= (1L << 31) || (long)l8.bs >= (1L << 31) ||
        (long)pl->n[0] * H * ((C / 2 + 15) / 16 * 16) >= (1L << 31))
        return SR_NOT_ELIGIBLE;
    // pair loads / stores need 2-element alignment of every view
    const View* vs[4] = {&x, &l2, &l4, &l8};
    const size_t es = dtype_size(dtype);
    for (const View* v : vs)
        if ((v->ld & 1) || (v->c0 & 1) || (v->cbs & 1) || (v->bs & 1) || ((uintptr_t)v->p % (2 * es))) return SR_NOT_ELIGIBLE;
    // the row-major lane order pays for channel-blocked tensors (a row of 8 channels is contiguous)
    auto is_blocked = [&](const View& v) { return v.ld == 8 && v.c0 % 8 == 0 && C % 8 == 0; };
    const bool blk_x = is_blocked(x), blk_o = is_blocked(l2) && is_blocked(l4) && is_blocked(l8);
    if (dtype == SR_F32) return run_all<float>(pl, B, x, ups, l2, l4, l8, pooled, pooled_ld, ws, blk_x, blk_o, bmask, st);
    return run_all<bf16_t>(pl, B, x, ups, l2, l4, l8, pooled, pooled_ld, ws, blk_x, blk_o, bmask, st);
}
}  // namespace sr
