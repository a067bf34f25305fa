// Backward of the FDConv weight path (the reference obtains it from torch autograd through FDConv.py:643-697).
//
// Forward (weightgen.hip), per sample b with pooled means p (cin):
//   z = fc p ; u = BN(z) ; a = s relu(u)^2 + b0 ; ca, fa, sa = 2 sigmoid(head(a)) ; ka = 2/KN sigmoid(kernel_fc(a))
//   ln = LayerNorm(p) ; logit[o*9+t][i] = sum_tau loc_w[o*9+t][tau] ln[i + tau - pad] ; R = 2 sigmoid(logit)
//   A[t][o][i] = sum_g ka_g Basis_g[o][i][t] ,  Basis_g = irfft2(alpha dft_weight [G == g])
//   W[t][o][i] = sa[t] ca[i] fa[o] R A
// Given dW [B][9][cout][cin] this file returns the gradients of every parameter on that path and of the pooled means.
//
// Kernels (one C entry point launches all of them):
//   wp_elem      per element: dA = dW M R (M = sa ca fa), dlogit = dW M A R (1 - R/2), and the reductions
//                dsa, dca, dfa (sums of dW R A times the two other factors) and dka_g = <dA, Basis_g>
//   wp_groupsum  S_g = sum_b ka[b][g] dA[b]   in the (3 cout, 3 cin) spatial layout of the spectrum's inverse transform
//   wp_dft_a/b   d dft_weight[r][c] = alpha w_c / (N1 N2) rfft2(S_{G(r,c)})[r][c]   (adjoint of the C2R transform: the
//                same two-stage direct DFT as basis_stage_a/b, run the other way, only the bins of each group kept)
//   wp_locw      d loc_w[q][tau] = sum_b sum_i dlogit[b][q][i] ln[b][i + tau - pad]
//   wp_ln        d ln[b][j] = sum_q sum_tau dlogit[b][q][j - tau + pad] loc_w[q][tau], LayerNorm backward -> dpooled (local
//                part), d ln_w, d ln_b
//   wp_global    one workgroup for the batch: heads, StarReLU, BatchNorm (batch or running statistics), fc backward ->
//                the KSM_Global parameter gradients and the global part of dpooled
#include <math.h>
#include <string.h>

#include "common.h"

namespace sr {

constexpr int WKN = 4;    // kernel_num
constexpr int WATT = 16;  // attention channels
__host__ __device__ inline int wp_mods_stride(int cin, int cout) { return (2 * cin + cout + 9 + WKN + 3) & ~3; }

struct WpGrads {
    float *dft, *fc_w, *bn_w, *bn_b, *relu_s, *relu_b, *ch_w, *ch_b, *fl_w, *fl_b, *sp_w, *sp_b, *kn_w, *kn_b, *loc_w, *ln_w, *ln_b;
};

__device__ __forceinline__ float wp_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// grid (cdiv(cin, 64), B, cdiv(cout, OPB)); 256 threads: lane <-> input channel, the four waves take output channels
constexpr int OPB = 8;
__global__ __launch_bounds__(256) void wp_elem(const float* __restrict__ dw, const float* __restrict__ mods, sr_layer_params lp,
                                               float* __restrict__ dA, float* __restrict__ dlogit, float* __restrict__ dca,
                                               float* __restrict__ dfa, float* __restrict__ dsa, float* __restrict__ dka) {
    __shared__ float red[4][64];
    __shared__ float red13[4][16];
    const int cin = lp.cin, cout = lp.cout, taps = lp.loc_taps, pad = (taps - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 64 + lane;
    const bool live = i < cin;
    const int ic = live ? i : cin - 1;
    const float* md = mods + (long)b * wp_mods_stride(cin, cout);
    const float* ln = md + cin;
    const float* fa = md + 2 * cin;
    const float* sa = fa + cout;
    const float* ka = sa + 9;
    const float cav = md[ic];
    float lnw[15];
#pragma unroll
    for (int t = 0; t < 15; ++t) {
        const int j = ic + t - pad;
        lnw[t] = (t < taps && j >= 0 && j < cin) ? ln[j] : 0.f;
    }
    float sav[9], kav[WKN];
#pragma unroll
    for (int t = 0; t < 9; ++t) sav[t] = sa[t];
#pragma unroll
    for (int g = 0; g < WKN; ++g) kav[g] = ka[g];
    float acc_ca = 0.f, acc_sa[9], acc_ka[WKN];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc_sa[t] = 0.f;
#pragma unroll
    for (int g = 0; g < WKN; ++g) acc_ka[g] = 0.f;
    const long n = 9L * cout * cin;
    const int o_end = min(cout, (int)(blockIdx.z + 1) * OPB);
    for (int o = blockIdx.z * OPB + wave; o < o_end; o += 4) {
        const float fav = fa[o];
        float bs[WKN][9];
#pragma unroll
        for (int g = 0; g < WKN; ++g)
#pragma unroll
            for (int t = 0; t < 9; ++t) bs[g][t] = lp.basis[(((long)g * cout + o) * cin + ic) * 9 + t];
        float acc_fa = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* lw = lp.loc_w + (long)(o * 9 + t) * taps;
            float logit = 0.f;
#pragma unroll
            for (int k = 0; k < 15; ++k)
                if (k < taps) logit += lw[k] * lnw[k];
            const float R = 2.0f * wp_sigmoid(logit);
            float A = 0.f;
#pragma unroll
            for (int g = 0; g < WKN; ++g) A += kav[g] * bs[g][t];
            const long at = ((long)t * cout + o) * cin + ic;
            const float d = live ? dw[(long)b * n + at] : 0.f;
            const float Q = d * R * A;
            const float M = sav[t] * cav * fav;
            const float dAv = d * M * R;
            acc_sa[t] += Q * cav * fav;
            acc_ca += Q * sav[t] * fav;
            acc_fa += Q * sav[t] * cav;
#pragma unroll
            for (int g = 0; g < WKN; ++g) acc_ka[g] += dAv * bs[g][t];
            if (live) {
                dA[(long)b * n + at] = dAv;
                dlogit[(long)b * n + ((long)(o * 9 + t)) * cin + i] = d * M * A * R * (1.0f - 0.5f * R);
            }
        }
        acc_fa = wp_wave_sum(acc_fa);
        if (lane == 0) atomicAdd(dfa + (long)b * cout + o, acc_fa);
    }
    red[wave][lane] = acc_ca;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float s = wp_wave_sum(acc_sa[t]);
        if (lane == 0) red13[wave][t] = s;
    }
#pragma unroll
    for (int g = 0; g < WKN; ++g) {
        const float s = wp_wave_sum(acc_ka[g]);
        if (lane == 0) red13[wave][9 + g] = s;
    }
    __syncthreads();
    if (wave == 0) {
        if (live) atomicAdd(dca + (long)b * cin + i, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        if (lane < 13) {
            const float s = red13[0][lane] + red13[1][lane] + red13[2][lane] + red13[3][lane];
            if (lane < 9) atomicAdd(dsa + (long)b * 12 + lane, s);
            else atomicAdd(dka + (long)b * WKN + (lane - 9), s);
        }
    }
}

// S[g][y = o*3+u][x = i*3+v] = sum_b ka[b][g] dA[b][t = u*3+v][o][i]
__global__ __launch_bounds__(256) void wp_groupsum(const float* __restrict__ dA, const float* __restrict__ mods, int B, int cin,
                                                   int cout, float* __restrict__ S) {
    const long n = 9L * cout * cin;
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= n) return;
    const int i = (int)(e % cin);
    const int o = (int)((e / cin) % cout);
    const int t = (int)(e / ((long)cin * cout));
    const int ms = wp_mods_stride(cin, cout);
    float acc[WKN] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        const float d = dA[(long)b * n + e];
        const float* ka = mods + (long)b * ms + 2 * cin + cout + 9;
#pragma unroll
        for (int g = 0; g < WKN; ++g) acc[g] += ka[g] * d;
    }
    const int N2 = 3 * cin;
    const long at = (long)(o * 3 + t / 3) * N2 + i * 3 + t % 3;
#pragma unroll
    for (int g = 0; g < WKN; ++g) S[(long)g * 9 * cout * cin + at] = acc[g];
}

constexpr int WP_TW = 2048;
__device__ __forceinline__ void wp_fill_tw(float2* tw, int N) {
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        double s, c;
        sincospi(2.0 * k / N, &s, &c);
        tw[k] = make_float2((float)c, (float)s);
    }
    __syncthreads();
}

// T[g][y][c] = sum_x S[g][y][x] e^{-2 pi i c x / N2}
__global__ __launch_bounds__(256) void wp_dft_a(const float* __restrict__ S, int N1, int N2, float2* __restrict__ T) {
    __shared__ float2 tw[WP_TW];
    wp_fill_tw(tw, N2);
    const int n2h = N2 / 2 + 1;
    const long total = (long)WKN * N1 * n2h;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % n2h);
        const long gy = idx / n2h;
        const float* row = S + gy * N2;
        float re = 0.f, im = 0.f;
        int ph = 0;
        for (int x = 0; x < N2; ++x) {
            const float v = row[x];
            const float2 w = tw[ph];
            re += v * w.x;
            im -= v * w.y;
            ph += c;
            if (ph >= N2) ph -= N2;
        }
        T[idx] = make_float2(re, im);
    }
}

// d dft[r][c] = alpha w_c / (N1 N2) sum_y T[G(r,c)][y][c] e^{-2 pi i r y / N1}
__global__ __launch_bounds__(256) void wp_dft_b(const float2* __restrict__ T, const uint8_t* __restrict__ gmap, float alpha, int N1,
                                                int N2, float* __restrict__ ddft) {
    __shared__ float2 tw[WP_TW];
    wp_fill_tw(tw, N1);
    const int n2h = N2 / 2 + 1;
    const long total = (long)N1 * n2h;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % n2h), r = (int)(idx / n2h);
        const int g = gmap[idx];
        const float2* col = T + (long)g * N1 * n2h + c;
        float re = 0.f, im = 0.f;
        int ph = 0;
        for (int y = 0; y < N1; ++y) {
            const float2 v = col[(long)y * n2h];
            const float2 w = tw[ph];
            re += v.x * w.x + v.y * w.y;
            im += v.y * w.x - v.x * w.y;
            ph += r;
            if (ph >= N1) ph -= N1;
        }
        const float wc = (c == 0 || (N2 % 2 == 0 && c == N2 / 2)) ? 1.0f : 2.0f;
        const float sc = alpha * wc / ((float)N1 * (float)N2);
        ddft[idx * 2] = sc * re;
        ddft[idx * 2 + 1] = sc * im;
    }
}

// d loc_w[q][tau]: one wavefront per q, lanes over the input channels
__global__ __launch_bounds__(256) void wp_locw(const float* __restrict__ dlogit, const float* __restrict__ mods, int B, int cin,
                                               int cout, int taps, float* __restrict__ dlocw) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= cout * 9) return;
    const int pad = (taps - 1) / 2, ms = wp_mods_stride(cin, cout);
    const long n = 9L * cout * cin;
    float acc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) acc[k] = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* ln = mods + (long)b * ms + cin;
        const float* dl = dlogit + (long)b * n + (long)q * cin;
        for (int i = lane; i < cin; i += 64) {
            const float d = dl[i];
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const int j = i + k - pad;
                if (k < taps && j >= 0 && j < cin) acc[k] += d * ln[j];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        if (k < taps) {
            const float s = wp_wave_sum(acc[k]);
            if (lane == 0) dlocw[(long)q * taps + k] = s;
        }
    }
}

// one workgroup per sample: d ln, LayerNorm backward
__global__ __launch_bounds__(256) void wp_ln(const float* __restrict__ dlogit, const float* __restrict__ pooled, int pooled_ld,
                                             sr_layer_params lp, float* __restrict__ dpooled, int dpooled_ld,
                                             float* __restrict__ dlnw, float* __restrict__ dlnb) {
    extern __shared__ float sh[];  // dnhat[cin] | nhat[cin] | red[8]
    const int cin = lp.cin, cout = lp.cout, taps = lp.loc_taps, pad = (taps - 1) / 2;
    float* dn = sh;
    float* nh = sh + cin;
    float* red = nh + cin;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pooled + (long)b * pooled_ld;
    const long n = 9L * cout * cin;
    // statistics of the sample (biased variance, eps 1e-5, as the forward)
    float s = 0.f;
    for (int i = tid; i < cin; i += 256) s += p[i];
    s = wp_wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / cin;
    __syncthreads();
    float v = 0.f;
    for (int i = tid; i < cin; i += 256) v += (p[i] - mean) * (p[i] - mean);
    v = wp_wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / cin + 1e-5f);
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
    for (int j = tid; j < cin; j += 256) {
        float acc = 0.f;
        for (int q = 0; q < cout * 9; ++q) {
            const float* dl = dlogit + (long)b * n + (long)q * cin;
            const float* lw = lp.loc_w + (long)q * taps;
            for (int k = 0; k < taps; ++k) {
                const int i = j - k + pad;  // output position whose window holds ln[j] at tap k
                if (i >= 0 && i < cin) acc += dl[i] * lw[k];
            }
        }
        const float nhat = (p[j] - mean) * rstd;
        atomicAdd(dlnw + j, acc * nhat);
        atomicAdd(dlnb + j, acc);
        const float d = acc * lp.ln_w[j];
        dn[j] = d;
        nh[j] = nhat;
        m1 += d;
        m2 += d * nhat;
    }
    m1 = wp_wave_sum(m1);
    m2 = wp_wave_sum(m2);
    if (lane == 0) {
        red[wave] = m1;
        red[4 + wave] = m2;
    }
    __syncthreads();
    const float a1 = (red[0] + red[1] + red[2] + red[3]) / cin, a2 = (red[4] + red[5] + red[6] + red[7]) / cin;
    for (int j = tid; j < cin; j += 256) dpooled[(long)b * dpooled_ld + j] = rstd * (dn[j] - a1 - nh[j] * a2);
}

// KSM_Global backward, one workgroup for the whole batch (BatchNorm couples the samples in train mode)
template <bool TRAIN>
__global__ __launch_bounds__(1024) void wp_global(const float* __restrict__ pooled, int pooled_ld, sr_layer_params lp, int B,
                                                  const float* __restrict__ mods, const float* __restrict__ dca,
                                                  const float* __restrict__ dfa, const float* __restrict__ dsa,
                                                  const float* __restrict__ dka, WpGrads g, float* __restrict__ dpooled,
                                                  int dpooled_ld) {
    extern __shared__ float sh[];
    const int cin = lp.cin, cout = lp.cout, ms = wp_mods_stride(cin, cout);
    float* zh = sh;                 // [B][ATT] normalised fc outputs
    float* rr = zh + B * WATT;      // relu(u)
    float* av = rr + B * WATT;      // StarReLU output a
    float* dz = av + B * WATT;      // da -> du -> dz
    float* stat = dz + B * WATT;    // mean[16] | rstd[16] | scratch[32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    for (int pair = wave; pair < B * WATT; pair += nw) {
        const int b = pair / WATT, k = pair - b * WATT;
        const float* p = pooled + (long)b * pooled_ld;
        const float* w = lp.fc_w + (long)k * cin;
        float s = 0.f;
        for (int i = lane; i < cin; i += 64) s += w[i] * p[i];
        s = wp_wave_sum(s);
        if (lane == 0) zh[pair] = s;
    }
    __syncthreads();
    if (tid < WATT) {
        float m, var;
        if (TRAIN) {
            m = 0.f;
            for (int b = 0; b < B; ++b) m += zh[b * WATT + tid];
            m /= B;
            var = 0.f;
            for (int b = 0; b < B; ++b) var += (zh[b * WATT + tid] - m) * (zh[b * WATT + tid] - m);
            var /= B;
        } else {
            m = lp.bn_rm[tid];
            var = lp.bn_rv[tid];
        }
        stat[tid] = m;
        stat[WATT + tid] = 1.0f / sqrtf(var + 1e-5f);
    }
    __syncthreads();
    const float s0 = lp.relu_s[0];
    for (int i = tid; i < B * WATT; i += nt) {
        const int k = i % WATT;
        const float zn = (zh[i] - stat[k]) * stat[WATT + k];
        zh[i] = zn;
        const float r = fmaxf(zn * lp.bn_w[k] + lp.bn_b[k], 0.f);
        rr[i] = r;
        av[i] = s0 * r * r + lp.relu_b[0];
    }
    __syncthreads();
    // pre-sigmoid gradients of the four heads, recomputed wherever they are used:
    //   ca, fa, sa = 2 sigmoid(v): d/dv = y (1 - y/2);  ka = 2/KN sigmoid(v): d/dv = y (1 - y KN / 2)
    const int per = cin + cout + 9 + WKN;
    auto dpre = [&](int b, int j) -> float {
        const float* md = mods + (long)b * ms;
        if (j < cin) { const float y = md[j]; return dca[(long)b * cin + j] * y * (1.0f - 0.5f * y); }
        if (j < cin + cout) { const int r = j - cin; const float y = md[2 * cin + r]; return dfa[(long)b * cout + r] * y * (1.0f - 0.5f * y); }
        if (j < cin + cout + 9) { const int r = j - cin - cout; const float y = md[2 * cin + cout + r]; return dsa[(long)b * 12 + r] * y * (1.0f - 0.5f * y); }
        const int r = j - cin - cout - 9;
        const float y = md[2 * cin + cout + 9 + r];
        return dka[(long)b * WKN + r] * y * (1.0f - 0.5f * WKN * y);
    };
    auto head = [&](int j, const float*& w, float*& gw, float*& gb, int& row) {
        if (j < cin) { row = j; w = lp.ch_w; gw = g.ch_w; gb = g.ch_b; }
        else if (j < cin + cout) { row = j - cin; w = lp.fl_w; gw = g.fl_w; gb = g.fl_b; }
        else if (j < cin + cout + 9) { row = j - cin - cout; w = lp.sp_w; gw = g.sp_w; gb = g.sp_b; }
        else { row = j - cin - cout - 9; w = lp.kn_w; gw = g.kn_w; gb = g.kn_b; }
    };
    // da[b][k] = sum_rows dpre[b][row] W[row][k]
    for (int i = tid; i < B * WATT; i += nt) {
        const int b = i / WATT, k = i - b * WATT;
        float s = 0.f;
        for (int j = 0; j < per; ++j) {
            const float* w;
            float *gw, *gb;
            int row;
            head(j, w, gw, gb, row);
            s += dpre(b, j) * w[row * WATT + k];
        }
        dz[i] = s;
    }
    // head weight / bias gradients
    for (int i = tid; i < per * WATT; i += nt) {
        const int j = i / WATT, k = i - j * WATT;
        const float* w;
        float *gw, *gb;
        int row;
        head(j, w, gw, gb, row);
        float s = 0.f, sb = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = dpre(b, j);
            s += d * av[b * WATT + k];
            sb += d;
        }
        gw[row * WATT + k] = s;
        if (k == 0) gb[row] = sb;
    }
    __syncthreads();
    // StarReLU: a = s r^2 + b0
    if (wave == 0) {
        float ds = 0.f, db = 0.f;
        for (int i = lane; i < B * WATT; i += 64) {
            ds += dz[i] * rr[i] * rr[i];
            db += dz[i];
        }
        ds = wp_wave_sum(ds);
        db = wp_wave_sum(db);
        if (lane == 0) {
            g.relu_s[0] = ds;
            g.relu_b[0] = db;
        }
    }
    __syncthreads();
    for (int i = tid; i < B * WATT; i += nt) dz[i] = dz[i] * 2.0f * s0 * rr[i];  // du (r = 0 where u <= 0)
    __syncthreads();
    if (tid < WATT) {
        const int k = tid;
        float dg = 0.f, db = 0.f;
        for (int b = 0; b < B; ++b) {
            dg += dz[b * WATT + k] * zh[b * WATT + k];
            db += dz[b * WATT + k];
        }
        g.bn_w[k] = dg;
        g.bn_b[k] = db;
        const float gam = lp.bn_w[k], rstd = stat[WATT + k];
        // d zhat = du gamma; train: dz = rstd (dzh - mean(dzh) - zhat mean(dzh zhat)); eval: dz = rstd dzh
        const float m1 = TRAIN ? gam * db / B : 0.f, m2 = TRAIN ? gam * dg / B : 0.f;
        for (int b = 0; b < B; ++b) dz[b * WATT + k] = rstd * (gam * dz[b * WATT + k] - m1 - zh[b * WATT + k] * m2);
    }
    __syncthreads();
    for (int i = tid; i < WATT * cin; i += nt) {
        const int k = i / cin, c = i - k * cin;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dz[b * WATT + k] * pooled[(long)b * pooled_ld + c];
        g.fc_w[i] = s;
    }
    for (int i = tid; i < B * cin; i += nt) {
        const int b = i / cin, c = i - b * cin;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WATT; ++k) s += dz[b * WATT + k] * lp.fc_w[k * cin + c];
        dpooled[(long)b * dpooled_ld + c] += s;  // the local (LayerNorm) part was written by wp_ln
    }
}

struct WpLayout {
    size_t dA, dlogit, S, T, red, total;
    size_t red_floats;
};
static WpLayout wp_layout(int B, int cin, int cout) {
    WpLayout l{};
    const size_t n = 9ul * cout * cin;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    l.dA = carve((size_t)B * n * 4);
    l.dlogit = carve((size_t)B * n * 4);
    l.S = carve((size_t)WKN * n * 4);
    l.T = carve((size_t)WKN * 3 * cout * (3 * cin / 2 + 1) * 8);
    l.red_floats = (size_t)B * (cin + cout + 12 + WKN);
    l.red = carve(l.red_floats * 4);
    l.total = off;
    return l;
}
}  // namespace sr

using namespace sr;

extern "C" size_t sr_fdconv_weightgen_backward_workspace_bytes(int B, int cin, int cout) {
    if (B < 1 || cin < 1 || cout < 1) return 0;
    return wp_layout(B, cin, cout).total;
}

extern "C" int sr_fdconv_weightgen_backward(int B, const float* pooled, int pooled_ld, const sr_layer_params* lp, int train,
                                            const float* mods, const float* dft_weight, const uint8_t* gmap, float alpha,
                                            const float* dw, const sr_weight_path_grads* grads, float* dpooled, int dpooled_ld,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    (void)dft_weight;  // the spectrum enters through lp->basis; its gradient does not depend on its value beyond that
    SR_REQUIRE(B > 0 && B <= 512 && pooled && lp && mods && gmap && dw && grads && dpooled && workspace);
    SR_REQUIRE(lp->cin > 0 && lp->cout > 0 && lp->cin <= 4096 && lp->cout <= 4096 && lp->loc_taps >= 1 && (lp->loc_taps & 1) && lp->loc_taps <= 15);
    SR_REQUIRE(lp->basis && lp->fc_w && lp->bn_w && lp->bn_b && lp->bn_rm && lp->bn_rv && lp->relu_s && lp->relu_b && lp->ch_w && lp->fl_w &&
               lp->sp_w && lp->kn_w && lp->loc_w && lp->ln_w && lp->ln_b);
    SR_REQUIRE(pooled_ld >= lp->cin && dpooled_ld >= lp->cin && (3 * lp->cin) % 2 == 0);
    SR_REQUIRE(!train || B >= 2);
    const float* const* gp = (const float* const*)grads;
    for (size_t i = 0; i < sizeof(sr_weight_path_grads) / sizeof(float*); ++i) SR_REQUIRE(gp[i] != nullptr);
    const int cin = lp->cin, cout = lp->cout;
    SR_REQUIRE(3 * cin <= WP_TW && 3 * cout <= WP_TW);
    const WpLayout l = wp_layout(B, cin, cout);
    if (workspace_bytes < l.total) return SR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* dA = (float*)(ws + l.dA);
    float* dlogit = (float*)(ws + l.dlogit);
    float* S = (float*)(ws + l.S);
    float2* T = (float2*)(ws + l.T);
    float* dca = (float*)(ws + l.red);
    float* dfa = dca + (size_t)B * cin;
    float* dsa = dfa + (size_t)B * cout;
    float* dka = dsa + (size_t)B * 12;
    WpGrads g;
    static_assert(sizeof(WpGrads) == sizeof(sr_weight_path_grads), "grads struct");
    ::memcpy(&g, grads, sizeof(g));
    const long n = 9L * cout * cin;
    const int N1 = 3 * cout, N2 = 3 * cin, n2h = N2 / 2 + 1;
    SR_HIP(hipMemsetAsync(dca, 0, l.red_floats * 4, st));
    SR_HIP(hipMemsetAsync(g.ln_w, 0, (size_t)cin * 4, st));
    SR_HIP(hipMemsetAsync(g.ln_b, 0, (size_t)cin * 4, st));
    wp_elem<<<dim3(cdiv(cin, 64), B, cdiv(cout, OPB)), 256, 0, st>>>(dw, mods, *lp, dA, dlogit, dca, dfa, dsa, dka);
    SR_LAUNCH_CHECK();
    wp_groupsum<<<cdiv(n, 256), 256, 0, st>>>(dA, mods, B, cin, cout, S);
    SR_LAUNCH_CHECK();
    const long na = (long)WKN * N1 * n2h, nb = (long)N1 * n2h;
    wp_dft_a<<<cdiv(na, 256), 256, 0, st>>>(S, N1, N2, T);
    SR_LAUNCH_CHECK();
    wp_dft_b<<<cdiv(nb, 256), 256, 0, st>>>(T, gmap, alpha, N1, N2, g.dft);
    SR_LAUNCH_CHECK();
    wp_locw<<<cdiv(cout * 9, 4), 256, 0, st>>>(dlogit, mods, B, cin, cout, lp->loc_taps, g.loc_w);
    SR_LAUNCH_CHECK();
    wp_ln<<<B, 256, (2 * (size_t)cin + 8) * 4, st>>>(dlogit, pooled, pooled_ld, *lp, dpooled, dpooled_ld, g.ln_w, g.ln_b);
    SR_LAUNCH_CHECK();
    const size_t shg = ((size_t)4 * B * WATT + 64) * 4;
    if (train) {
        SR_LDS_ATTR(shg, wp_global<true>);
        wp_global<true><<<1, 1024, shg, st>>>(pooled, pooled_ld, *lp, B, mods, dca, dfa, dsa, dka, g, dpooled, dpooled_ld);
    } else {
        SR_LDS_ATTR(shg, wp_global<false>);
        wp_global<false><<<1, 1024, shg, st>>>(pooled, pooled_ld, *lp, B, mods, dca, dfa, dsa, dka, g, dpooled, dpooled_ld);
    }
    SR_LAUNCH_CHECK();
    return SR_OK;
}
