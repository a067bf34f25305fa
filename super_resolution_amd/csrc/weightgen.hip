// FDConv weight path.
//   sr_fdconv_basis     : Basis_g = irfft2(alpha * dft_weight * [group == g])  (FDConv.py:661-684)
//                         run once per weight update; exact two-stage inverse real DFT in fp64.
//   sr_fdconv_weightgen : per-sample dynamic weights from the pooled means (FDConv.py:643-652, 690-697)
//                         KSM_Global (fc -> BN (running or batch statistics) -> StarReLU -> 4 sigmoid heads), KSM_Local
//                         (LayerNorm + k-tap conv along the channel axis), and
//                         W_b = sa * ca * fa * hr * sum_g ka_g * Basis_g, written in the conv kernels'
//                         [tap][cout][cin] operand layout.
#include <math.h>

#include "common.h"

namespace sr {

constexpr int KN = 4;    // kernel_num (frequency groups), owner scripts use the default
constexpr int ATT = 16;  // attention channels: max(int(cin*0.0625), 16) = 16 for cin < 272

// ---- basis: stage A  Z[g][y][c] = sum_r [G[r][c]==g] D[r][c] e^{+2 pi i r y / N1} ------------
// The phase factors e^{2 pi i k / N} come from a per-workgroup LDS table (same sincospi values as evaluating them per
// term, which cost ~100 fp64 instructions each: 390 -> tens of microseconds for the 192 -> 64 layer, paid once per weight
// update, i.e. per training step); N beyond the table falls back to evaluating per term.
constexpr int BASIS_TW = 2048;
__device__ __forceinline__ void fill_phase_table(double2* tw, int N) {
    if (N <= BASIS_TW) {
        for (int k = threadIdx.x; k < N; k += blockDim.x) {
            double s, co;
            sincospi(2.0 * k / N, &s, &co);
            tw[k] = make_double2(co, s);
        }
    }
    __syncthreads();
}
__device__ __forceinline__ double2 phase(const double2* tw, int ph, int N) {
    if (N <= BASIS_TW) return tw[ph];
    double s, co;
    sincospi(2.0 * ph / N, &s, &co);
    return make_double2(co, s);
}

__global__ void basis_stage_a(const float* __restrict__ dft, const uint8_t* __restrict__ gmap, int N1, int n2h,
                              double2* __restrict__ Z) {
    __shared__ double2 tw[BASIS_TW];
    fill_phase_table(tw, N1);
    const long total = (long)KN * N1 * n2h;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % n2h);
        const int y = (int)((i / n2h) % N1);
        const int g = (int)(i / ((long)n2h * N1));
        double re = 0.0, im = 0.0;
        int ph = 0;  // (r * y) mod N1
        for (int r = 0; r < N1; ++r) {
            if (gmap[(long)r * n2h + c] == g) {
                const double dr = dft[((long)r * n2h + c) * 2], di = dft[((long)r * n2h + c) * 2 + 1];
                const double2 w = phase(tw, ph, N1);
                re += dr * w.x - di * w.y;
                im += dr * w.y + di * w.x;
            }
            ph += y;
            if (ph >= N1) ph -= N1;
        }
        Z[i] = make_double2(re, im);
    }
}

// ---- basis: stage B  M[g][y][x] = 1/(N1 N2) sum_c wgt_c Re(Z[g][y][c] e^{+2 pi i c x / N2}) ---
// C2R semantics: only the real part of the self-conjugate columns contributes, weight 1; others 2.
__global__ void basis_stage_b(const double2* __restrict__ Z, float alpha, int cout, int cin, float* __restrict__ basis) {
    const int N1 = 3 * cout, N2 = 3 * cin, n2h = N2 / 2 + 1;
    __shared__ double2 tw[BASIS_TW];
    fill_phase_table(tw, N2);
    const long total = (long)KN * N1 * N2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % N2);
        const int y = (int)((i / N2) % N1);
        const int g = (int)(i / ((long)N2 * N1));
        const double2* z = Z + ((long)g * N1 + y) * n2h;
        double acc = 0.0;
        int ph = 0;  // (c * x) mod N2
        for (int c = 0; c < n2h; ++c) {
            const double2 w = phase(tw, ph, N2);
            const double wgt = (c == 0 || (N2 % 2 == 0 && c == N2 / 2)) ? 1.0 : 2.0;
            acc += wgt * (z[c].x * w.x - z[c].y * w.y);
            ph += x;
            if (ph >= N2) ph -= N2;
        }
        acc *= (double)alpha / ((double)N1 * (double)N2);
        // M[y = o*3+u][x = ci*3+v] -> basis[g][o][ci][u*3+v]   (reshape+permute at FDConv.py:683-684)
        const int o = y / 3, u = y % 3, ci = x / 3, v = x % 3;
        basis[(((long)g * cout + o) * cin + ci) * 9 + u * 3 + v] = (float)acc;
    }
}

// ---- per-sample weights -------------------------------------------------------------------
// Two launches per layer:
//   ksm_kernel       one workgroup of 16 waves for the whole batch: KSM_Global (fc -> BatchNorm -> StarReLU -> four
//                    sigmoid heads) and KSM_Local's LayerNorm.  Every dot product over the input channels is a
//                    wavefront reduction (lanes over cin, DPP/permute butterfly); BatchNorm sees the whole batch, so
//                    train mode (statistics over the batch of pooled vectors, FDConv.py:245-258 with nn.BatchNorm2d in
//                    train()) is the same kernel.  Result: the modulation vectors `mods` [B][ms] in global memory:
//                      ca[cin] | ln[cin] | fa[cout] | sa[9] | ka[4]      (ms = mods_stride(cin, cout))
//   aggregate_kernel one work item per (sample, o, ci): the 9 taps of W_b = sa*ca*fa*hr*sum_g ka_g*Basis_g, KSM_Local's
//                    k-tap conv along the channel axis formed on the way; cout*cin/256 x B workgroups.
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__host__ __device__ inline int mods_stride(int cin, int cout) { return (2 * cin + cout + 9 + KN + 3) & ~3; }

template <bool TRAIN>
__global__ __launch_bounds__(1024) void ksm_kernel(const float* __restrict__ pooled, int pooled_ld, sr_layer_params lp, int Btot,
                                                   int bpb, float momentum, float* __restrict__ mods, int staged) {
    extern __shared__ float sh[];
    const int cin = lp.cin, cout = lp.cout;
    const int ms = mods_stride(cin, cout);
    // this workgroup's samples: all of them in train mode (BatchNorm couples the batch), `bpb` per workgroup otherwise
    // (running statistics: samples are independent, and a kernel of one workgroup is a chain of dependent memory round
    // trips -- 184 us for 16 samples in one workgroup)
    const int b0 = blockIdx.x * bpb;
    const int B = min(bpb, Btot - b0);
    pooled += (long)b0 * pooled_ld;
    mods += (long)b0 * ms;
    float* z = sh;                 // [B][ATT] fc outputs, then StarReLU activations
    float* mean = z + B * ATT;     // [ATT]
    float* var = mean + ATT;       // [ATT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    // staged (train mode, one workgroup for the batch on an otherwise idle chip): pooled means and fc weights through LDS --
    // the loop below is otherwise a chain of B * 16 / 16 global round trips per wavefront (40 us per layer)
    float* ps = var + ATT;       // [B][cin]
    float* fw = ps + B * cin;    // [ATT][cin]
    if (staged) {
        for (int i = tid; i < B * cin; i += blockDim.x) ps[i] = pooled[(long)(i / cin) * pooled_ld + i % cin];
        for (int i = tid; i < ATT * cin; i += blockDim.x) fw[i] = lp.fc_w[i];
        __syncthreads();
    }
    // fc (no bias): one wavefront reduction per (sample, attention channel)
    for (int pair = wave; pair < B * ATT; pair += nw) {
        const int b = pair / ATT, k = pair - b * ATT;
        const float* g = staged ? ps + (long)b * cin : pooled + (long)b * pooled_ld;
        const float* w = staged ? fw + (long)k * cin : lp.fc_w + (long)k * cin;
        float s = 0.f;
        for (int i = lane; i < cin; i += 64) s += w[i] * g[i];
        s = wave_sum(s);
        if (lane == 0) z[pair] = s;
    }
    __syncthreads();
    // BatchNorm statistics: running (eval) or over the batch (train; biased variance normalises, the unbiased one goes
    // into the running estimate, as nn.BatchNorm2d does)
    if (tid < ATT) {
        if (TRAIN) {
            float m = 0.f;
            for (int b = 0; b < B; ++b) m += z[b * ATT + tid];
            m /= B;
            float v = 0.f;
            for (int b = 0; b < B; ++b) v += (z[b * ATT + tid] - m) * (z[b * ATT + tid] - m);
            mean[tid] = m;
            var[tid] = v / B;
            float* rm = const_cast<float*>(lp.bn_rm);
            float* rv = const_cast<float*>(lp.bn_rv);
            rm[tid] = (1.0f - momentum) * rm[tid] + momentum * m;
            rv[tid] = (1.0f - momentum) * rv[tid] + momentum * (v / (B - 1));
        } else {
            mean[tid] = lp.bn_rm[tid];
            var[tid] = lp.bn_rv[tid];
        }
    }
    __syncthreads();
    for (int i = tid; i < B * ATT; i += blockDim.x) {
        const int k = i % ATT;
        const float zn = (z[i] - mean[k]) / sqrtf(var[k] + 1e-5f) * lp.bn_w[k] + lp.bn_b[k];
        const float r = fmaxf(zn, 0.f);
        z[i] = lp.relu_s[0] * r * r + lp.relu_b[0];
    }
    __syncthreads();
    // heads: one work item per (sample, output); rows of 16 weights
    const int per = cin + cout + 9 + KN;
    for (int it = tid; it < B * per; it += blockDim.x) {
        const int b = it / per, j = it - b * per;
        const float *w, *bias;
        int row, at;
        float scale = 2.0f;
        if (j < cin) { row = j; w = lp.ch_w; bias = lp.ch_b; at = j; }
        else if (j < cin + cout) { row = j - cin; w = lp.fl_w; bias = lp.fl_b; at = 2 * cin + row; }
        else if (j < cin + cout + 9) { row = j - cin - cout; w = lp.sp_w; bias = lp.sp_b; at = 2 * cin + cout + row; }
        else { row = j - cin - cout - 9; w = lp.kn_w; bias = lp.kn_b; at = 2 * cin + cout + 9 + row; scale = 2.0f / KN; }
        float s = bias[row];
#pragma unroll
        for (int k = 0; k < ATT; ++k) s += w[row * ATT + k] * z[b * ATT + k];
        mods[(long)b * ms + at] = scale * sigmoidf(s);
    }
    // KSM_Local: LayerNorm over the channel axis (biased variance, eps 1e-5), one wavefront per sample
    for (int b = wave; b < B; b += nw) {
        const float* g = staged ? ps + (long)b * cin : pooled + (long)b * pooled_ld;
        float s = 0.f;
        for (int i = lane; i < cin; i += 64) s += g[i];
        const float m = wave_sum(s) / cin;
        float q = 0.f;
        for (int i = lane; i < cin; i += 64) q += (g[i] - m) * (g[i] - m);
        const float rstd = 1.0f / sqrtf(wave_sum(q) / cin + 1e-5f);
        for (int i = lane; i < cin; i += 64) mods[(long)b * ms + cin + i] = (g[i] - m) * rstd * lp.ln_w[i] + lp.ln_b[i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void aggregate_kernel(const float* __restrict__ mods, sr_layer_params lp, T* __restrict__ wdyn,
                                                        T* __restrict__ wdyn_lo, int f16 = 0) {
    const int cin = lp.cin, cout = lp.cout, taps = lp.loc_taps, pad = (taps - 1) / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cout * cin) return;
    const int b = blockIdx.y;
    const int o = idx / cin, ci = idx - o * cin;
    const float* md = mods + (long)b * mods_stride(cin, cout);
    const float* ln = md + cin;
    const float* sa = md + 2 * cin + cout;
    const float* ka = sa + 9;
    const float cf = md[ci] * md[2 * cin + o];  // ca * fa
    float lnw[15];
#pragma unroll
    for (int t = 0; t < 15; ++t) {
        const int j = ci + t - pad;
        lnw[t] = (t < taps && j >= 0 && j < cin) ? ln[j] : 0.f;
    }
    float ad[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) ad[tap] = 0.f;
#pragma unroll
    for (int q = 0; q < KN; ++q) {
        const float* bs = lp.basis + (((long)q * cout + o) * cin + ci) * 9;
        const float k = ka[q];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) ad[tap] += k * bs[tap];
    }
    const long bstride = 9L * cout * cin;
    float lo_sum = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* lw = lp.loc_w + (long)(o * 9 + tap) * taps;
        float logit = 0.f;
#pragma unroll
        for (int t = 0; t < 15; ++t)
            if (t < taps) logit += lw[t] * lnw[t];
        const float hr = 2.0f * sigmoidf(logit);
        const float w = sa[tap] * cf * ad[tap] * hr;
        const long at = b * bstride + ((long)tap * cout + o) * cin + ci;
        if (sizeof(T) == 2 && f16) {  // one fp16 value per weight (opt-in operand format of the banded conv): [B][10][cout][cin],
            // plane 9 = the sum over the taps of what the rounding lost (applied to the centre pixel by the conv)
            const uint16_t hb = f32_to_f16_bits(w);
            const long at16 = b * (10L * cout * cin) + ((long)tap * cout + o) * cin + ci;
            ((uint16_t*)wdyn)[at16] = hb;
            lo_sum += w - (float)__builtin_bit_cast(_Float16, hb);
            if (tap == 8) ((uint16_t*)wdyn)[b * (10L * cout * cin) + (9L * cout + o) * cin + ci] = f32_to_f16_bits(lo_sum);
            continue;
        }
        Elem<T>::st(wdyn + at, w);
        if (wdyn_lo) Elem<T>::st(wdyn_lo + at, w - Elem<T>::ld(wdyn + at));  // w = hi + lo, both in T (bf16: 16 significant bits)
    }
}

constexpr int KSM_MAX_B = 512;  // batch rows one ksm_kernel launch keeps in LDS (32 KiB of fc outputs)

size_t weightgen_ws_bytes(int B, int cin, int cout) { return align_up((size_t)B * mods_stride(cin, cout) * sizeof(float), 256); }

int weightgen_launch(int dtype, int B, const float* pooled, int pooled_ld, const sr_layer_params& lp, int train, float momentum,
                     void* wdyn, void* wdyn_lo, float* mods, hipStream_t st) {
    if (lp.loc_taps > 15) return SR_ERR_ARG;
    if (wdyn_lo == SR_WDYN_F16 && dtype != SR_BF16) return SR_ERR_ARG;
    if (train && (B < 2 || B > KSM_MAX_B)) return SR_ERR_ARG;  // BatchNorm needs more than one value per channel
    if (train) {
        size_t sh = ((size_t)B * ATT + 2 * ATT) * sizeof(float);
        const size_t sh_staged = sh + ((size_t)B * lp.cin + (size_t)ATT * lp.cin) * sizeof(float);
        const int staged = sh_staged <= 60 * 1024;
        if (staged) sh = sh_staged;
        ksm_kernel<true><<<1, B >= 4 ? 1024 : 256, sh, st>>>(pooled, pooled_ld, lp, B, B, momentum, mods, staged);
    } else {
        const size_t sh = (3 * ATT) * sizeof(float);
        ksm_kernel<false><<<B, 256, sh, st>>>(pooled, pooled_ld, lp, B, 1, momentum, mods, 0);
    }
    SR_LAUNCH_CHECK();
    dim3 grid(cdiv((long)lp.cout * lp.cin, 256), B);
    if (dtype == SR_F32)
        aggregate_kernel<float><<<grid, 256, 0, st>>>(mods, lp, (float*)wdyn, (float*)wdyn_lo);
    else
        if (wdyn_lo == SR_WDYN_F16) aggregate_kernel<bf16_t><<<grid, 256, 0, st>>>(mods, lp, (bf16_t*)wdyn, nullptr, 1);
        else aggregate_kernel<bf16_t><<<grid, 256, 0, st>>>(mods, lp, (bf16_t*)wdyn, (bf16_t*)wdyn_lo);
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

extern "C" size_t sr_fdconv_weightgen_workspace_bytes(int B, int cin, int cout) {
    if (B < 1 || cin < 1 || cout < 1) return 0;
    return weightgen_ws_bytes(B, cin, cout);
}

extern "C" int sr_fdconv_weightgen(int dtype, int B, const float* pooled, int pooled_ld, const sr_layer_params* lp, int train,
                                   float momentum, void* wdyn, void* wdyn_lo, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && B <= 65535 && pooled && wdyn && workspace && layer_ok_active(lp) && pooled_ld >= lp->cin);
    SR_REQUIRE(momentum >= 0.f && momentum <= 1.f);
    if (workspace_bytes < weightgen_ws_bytes(B, lp->cin, lp->cout)) return SR_ERR_WORKSPACE;
    return weightgen_launch(dtype, B, pooled, pooled_ld, *lp, train, momentum, wdyn, wdyn_lo, (float*)workspace, (hipStream_t)stream);
}
