// FDConv weight path.
//   sr_fdconv_basis     : Basis_g = irfft2(alpha * dft_weight * [group == g])  (FDConv.py:661-684)
//                         run once per weight update; exact two-stage inverse real DFT in fp64.
//   sr_fdconv_weightgen : per-sample dynamic weights from the pooled means (FDConv.py:643-652, 690-697)
//                         KSM_Global (fc -> BN(eval) -> StarReLU -> 4 sigmoid heads), KSM_Local
//                         (LayerNorm + k-tap conv along the channel axis), and
//                         W_b = sa * ca * fa * hr * sum_g ka_g * Basis_g, written in the conv kernels'
//                         [tap][cout][cin] operand layout.
#include <math.h>

#include "common.h"

namespace sr {

constexpr int KN = 4;    // kernel_num (frequency groups), owner scripts use the default
constexpr int ATT = 16;  // attention channels: max(int(cin*0.0625), 16) = 16 for cin < 272

// ---- basis: stage A  Z[g][y][c] = sum_r [G[r][c]==g] D[r][c] e^{+2 pi i r y / N1} ------------
__global__ void basis_stage_a(const float* __restrict__ dft, const uint8_t* __restrict__ gmap, int N1, int n2h,
                              double2* __restrict__ Z) {
    const long total = (long)KN * N1 * n2h;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % n2h);
        const int y = (int)((i / n2h) % N1);
        const int g = (int)(i / ((long)n2h * N1));
        double re = 0.0, im = 0.0;
        for (int r = 0; r < N1; ++r) {
            if (gmap[(long)r * n2h + c] != g) continue;
            const double dr = dft[((long)r * n2h + c) * 2], di = dft[((long)r * n2h + c) * 2 + 1];
            const int ph = (int)(((long)r * y) % N1);
            double s, co;
            sincospi(2.0 * ph / N1, &s, &co);
            re += dr * co - di * s;
            im += dr * s + di * co;
        }
        Z[i] = make_double2(re, im);
    }
}

// ---- basis: stage B  M[g][y][x] = 1/(N1 N2) sum_c wgt_c Re(Z[g][y][c] e^{+2 pi i c x / N2}) ---
// C2R semantics: only the real part of the self-conjugate columns contributes, weight 1; others 2.
__global__ void basis_stage_b(const double2* __restrict__ Z, float alpha, int cout, int cin, float* __restrict__ basis) {
    const int N1 = 3 * cout, N2 = 3 * cin, n2h = N2 / 2 + 1;
    const long total = (long)KN * N1 * N2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % N2);
        const int y = (int)((i / N2) % N1);
        const int g = (int)(i / ((long)N2 * N1));
        const double2* z = Z + ((long)g * N1 + y) * n2h;
        double acc = 0.0;
        for (int c = 0; c < n2h; ++c) {
            const int ph = (int)(((long)c * x) % N2);
            double s, co;
            sincospi(2.0 * ph / N2, &s, &co);
            const double wgt = (c == 0 || (N2 % 2 == 0 && c == N2 / 2)) ? 1.0 : 2.0;
            acc += wgt * (z[c].x * co - z[c].y * s);
        }
        acc *= (double)alpha / ((double)N1 * (double)N2);
        // M[y = o*3+u][x = ci*3+v] -> basis[g][o][ci][u*3+v]   (reshape+permute at FDConv.py:683-684)
        const int o = y / 3, u = y % 3, ci = x / 3, v = x % 3;
        basis[(((long)g * cout + o) * cin + ci) * 9 + u * 3 + v] = (float)acc;
    }
}

// ---- per-sample weights -------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256) void weightgen_kernel(const float* __restrict__ pooled, int pooled_ld,
                                                        sr_layer_params lp, T* __restrict__ wdyn, T* __restrict__ wdyn_lo,
                                                        int o_per_block) {
    extern __shared__ float sh[];
    const int cin = lp.cin, cout = lp.cout, taps = lp.loc_taps;
    float* g = sh;              // [cin] pooled means
    float* ln = g + cin;        // [cin] LayerNorm(g)
    float* ca = ln + cin;       // [cin]
    float* fa = ca + cin;       // [cout]
    float* a = fa + cout;       // [ATT]
    float* sa = a + ATT;        // [9]
    float* ka = sa + 9;         // [KN]
    float* red = ka + KN;       // [2]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < cin; i += blockDim.x) g[i] = pooled[(long)b * pooled_ld + i];
    __syncthreads();
    // KSM_Global: fc (no bias) -> BatchNorm (running stats) -> StarReLU
    if (tid < ATT) {
        float z = 0.f;
        for (int i = 0; i < cin; ++i) z += lp.fc_w[tid * cin + i] * g[i];
        z = (z - lp.bn_rm[tid]) / sqrtf(lp.bn_rv[tid] + 1e-5f) * lp.bn_w[tid] + lp.bn_b[tid];
        const float r = fmaxf(z, 0.f);
        a[tid] = lp.relu_s[0] * r * r + lp.relu_b[0];
    }
    // KSM_Local: LayerNorm statistics over the channel axis (biased variance, eps 1e-5)
    if (tid == 64) {
        float m = 0.f;
        for (int i = 0; i < cin; ++i) m += g[i];
        m /= cin;
        float v = 0.f;
        for (int i = 0; i < cin; ++i) v += (g[i] - m) * (g[i] - m);
        v /= cin;
        red[0] = m;
        red[1] = 1.0f / sqrtf(v + 1e-5f);
    }
    __syncthreads();
    for (int i = tid; i < cin; i += blockDim.x) {
        ln[i] = (g[i] - red[0]) * red[1] * lp.ln_w[i] + lp.ln_b[i];
        float z = lp.ch_b[i];
        for (int k = 0; k < ATT; ++k) z += lp.ch_w[i * ATT + k] * a[k];
        ca[i] = 2.0f * sigmoidf(z);
    }
    for (int i = tid; i < cout; i += blockDim.x) {
        float z = lp.fl_b[i];
        for (int k = 0; k < ATT; ++k) z += lp.fl_w[i * ATT + k] * a[k];
        fa[i] = 2.0f * sigmoidf(z);
    }
    if (tid < 9) {
        float z = lp.sp_b[tid];
        for (int k = 0; k < ATT; ++k) z += lp.sp_w[tid * ATT + k] * a[k];
        sa[tid] = 2.0f * sigmoidf(z);
    }
    if (tid >= 32 && tid < 32 + KN) {
        const int q = tid - 32;
        float z = lp.kn_b[q];
        for (int k = 0; k < ATT; ++k) z += lp.kn_w[q * ATT + k] * a[k];
        ka[q] = sigmoidf(z) * 2.0f / KN;
    }
    __syncthreads();
    // aggregate: one item per (o, tap, ci); ci innermost so stores are contiguous
    const int o0 = blockIdx.y * o_per_block;
    const int o1 = min(cout, o0 + o_per_block);
    const int pad = (taps - 1) / 2;
    const long bstride = 9L * cout * cin;
    for (int it = tid; it < (o1 - o0) * 9 * cin; it += blockDim.x) {
        const int ci = it % cin;
        const int tap = (it / cin) % 9;
        const int o = o0 + it / (9 * cin);
        const float* lw = lp.loc_w + (long)(o * 9 + tap) * taps;
        float logit = 0.f;
        for (int t = 0; t < taps; ++t) {
            const int j = ci + t - pad;
            if (j >= 0 && j < cin) logit += lw[t] * ln[j];
        }
        const float hr = 2.0f * sigmoidf(logit);
        float ad = 0.f;
#pragma unroll
        for (int q = 0; q < KN; ++q) ad += ka[q] * lp.basis[(((long)q * cout + o) * cin + ci) * 9 + tap];
        const float w = sa[tap] * ca[ci] * fa[o] * ad * hr;
        const long at = b * bstride + ((long)tap * cout + o) * cin + ci;
        Elem<T>::st(wdyn + at, w);
        if (wdyn_lo) Elem<T>::st(wdyn_lo + at, w - Elem<T>::ld(wdyn + at));  // w = hi + lo, both in T (bf16: 16 significant bits)
    }
}

int weightgen_launch(int dtype, int B, const float* pooled, int pooled_ld, const sr_layer_params& lp, void* wdyn,
                     void* wdyn_lo, hipStream_t st) {
    const int opb = 8;
    dim3 grid(B, cdiv(lp.cout, opb));
    const size_t sh = (3 * lp.cin + lp.cout + ATT + 9 + KN + 2) * sizeof(float);
    if (dtype == SR_F32)
        weightgen_kernel<float><<<grid, 256, sh, st>>>(pooled, pooled_ld, lp, (float*)wdyn, (float*)wdyn_lo, opb);
    else
        weightgen_kernel<bf16_t><<<grid, 256, sh, st>>>(pooled, pooled_ld, lp, (bf16_t*)wdyn, (bf16_t*)wdyn_lo, opb);
    SR_LAUNCH_CHECK();
    return SR_OK;
}
}  // namespace sr

using namespace sr;

static bool layer_ok_active(const sr_layer_params* lp) {
    return lp && lp->cin > 0 && lp->cout > 0 && lp->cin <= 4096 && lp->cout <= 4096 && lp->loc_taps >= 1 &&
           (lp->loc_taps & 1) && lp->basis && lp->fc_w && lp->bn_w && lp->bn_b && lp->bn_rm && lp->bn_rv &&
           lp->relu_s && lp->relu_b && lp->ch_w && lp->ch_b && lp->fl_w && lp->fl_b && lp->sp_w && lp->sp_b &&
           lp->kn_w && lp->kn_b && lp->loc_w && lp->ln_w && lp->ln_b;
}

extern "C" size_t sr_fdconv_basis_workspace_bytes(int cout, int cin) {
    if (cout < 1 || cin < 1) return 0;
    return align_up((size_t)KN * 3 * cout * (3 * cin / 2 + 1) * sizeof(double2), 256);
}

extern "C" int sr_fdconv_basis(const float* dft_weight, const uint8_t* gmap, float alpha, int cout, int cin, float* basis,
                               void* ws, size_t ws_bytes, void* stream) {
    SR_REQUIRE(dft_weight && gmap && basis && ws && cout > 0 && cin > 0);
    SR_REQUIRE((3 * cin) % 2 == 0);  // irfft2 at FDConv.py:683 only returns 3*cin columns when that is even
    if (ws_bytes < sr_fdconv_basis_workspace_bytes(cout, cin)) return SR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int N1 = 3 * cout, n2h = 3 * cin / 2 + 1;
    const long na = (long)KN * N1 * n2h, nb = (long)KN * N1 * 3 * cin;
    basis_stage_a<<<(int)((na + 255) / 256), 256, 0, st>>>(dft_weight, gmap, N1, n2h, (double2*)ws);
    SR_LAUNCH_CHECK();
    basis_stage_b<<<(int)((nb + 255) / 256), 256, 0, st>>>((const double2*)ws, alpha, cout, cin, basis);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

extern "C" int sr_fdconv_weightgen(int dtype, int B, const float* pooled, int pooled_ld, const sr_layer_params* lp,
                                   void* wdyn, void* wdyn_lo, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && pooled && wdyn && layer_ok_active(lp) && pooled_ld >= lp->cin);
    return weightgen_launch(dtype, B, pooled, pooled_ld, *lp, wdyn, wdyn_lo, (hipStream_t)stream);
}
