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
//   wp_elem      one pass over dW: with dA = dW M R (M = sa ca fa) and dlogit = dW M A R (1 - R/2) per element, the reductions
//                dsa, dca, dfa (sums of dW R A times the two other factors), dka_g = <dA, Basis_g>,
//                S_g = sum_b ka[b][g] dA[b] in the (3 cout, 3 cin) spatial layout of the spectrum's inverse transform,
//                d loc_w[q][tau] = sum_b sum_i dlogit[b][q][i] ln[b][i + tau - pad], and
//                G[b][tau][i] = sum_q dlogit[b][q][i] loc_w[q][tau]
//   wp_dft_a/b   d dft_weight[r][c] = alpha w_c / (N1 N2) rfft2(S_{G(r,c)})[r][c]   (adjoint of the C2R transform: the
//                same two-stage direct DFT as basis_stage_a/b, run the other way, only the bins of each group kept)
//   wp_ln        d ln[b][j] = sum_tau G[b][tau][j - tau + pad], LayerNorm backward -> dpooled (local part), d ln_w, d ln_b
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

// ---- wp_elem: everything that needs the per-element cotangent dW, in ONE pass over it -----------------------------------
// grid (cdiv(cin, 64), cdiv(cout, 4), batch chunks); 256 threads: lane <-> input channel i, wave <-> output channel o.
// A thread keeps, for its (o, i) and the nine taps, the four basis values, the KSM_Local weight rows and the accumulators
// that run over the batch -- S_g (the spectrum's cotangent, sum_b ka[b][g] dA[b]) and d loc_w -- in registers; the
// per-sample reductions (dca, dfa, dsa, dka, and G[b][tau][i] = sum_q dlogit[b][q][i] loc_w[q][tau], from which wp_ln forms
// d ln) leave as one contiguous atomic instruction each.  (The first version wrote dA and dlogit to memory and reduced
// them in three more kernels: 520 us per layer, mostly in two single-workgroup-per-sample loops.)
template <int TAPS>
__global__ __launch_bounds__(256) void wp_elem(const float* __restrict__ dw, const float* __restrict__ mods, sr_layer_params lp, int B,
                                               int bchunk, float* __restrict__ S, float* __restrict__ dlocw, float* __restrict__ G,
                                               float* __restrict__ dca, float* __restrict__ dfa, float* __restrict__ dsk) {
    __shared__ float red[4][64];
    constexpr int PAD = (TAPS - 1) / 2;
    const int cin = lp.cin, cout = lp.cout;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const bool live = i < cin;
    const int ic = live ? i : cin - 1;
    const int o = blockIdx.y * 4 + wave;
    const bool olive = o < cout;          // wave-uniform
    const int oc = olive ? o : cout - 1;
    const int ms = wp_mods_stride(cin, cout);
    const long n = 9L * cout * cin;
    float bs[WKN][9], lw[9][TAPS];
#pragma unroll
    for (int g = 0; g < WKN; ++g)
#pragma unroll
        for (int t = 0; t < 9; ++t) bs[g][t] = lp.basis[(((long)g * cout + oc) * cin + ic) * 9 + t];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < TAPS; ++k) lw[t][k] = lp.loc_w[(long)(oc * 9 + t) * TAPS + k];
    float accS[WKN][9], accL[9][TAPS];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int g = 0; g < WKN; ++g) accS[g][t] = 0.f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) accL[t][k] = 0.f;
    }
    const int b0 = blockIdx.z * bchunk, b1 = min(B, b0 + bchunk);
    for (int b = b0; b < b1; ++b) {
        const float* md = mods + (long)b * ms;
        const float* ln = md + cin;
        const float cav = md[ic], fav = md[2 * cin + oc];
        float lnw[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const int j = ic + k - PAD;
            lnw[k] = (j >= 0 && j < cin) ? ln[j] : 0.f;
        }
        float sav[9], kav[WKN];
#pragma unroll
        for (int t = 0; t < 9; ++t) sav[t] = md[2 * cin + cout + t];
#pragma unroll
        for (int g = 0; g < WKN; ++g) kav[g] = md[2 * cin + cout + 9 + g];
        float a_ca = 0.f, a_fa = 0.f, a_sk[13], a_g[TAPS];
#pragma unroll
        for (int j = 0; j < 13; ++j) a_sk[j] = 0.f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) a_g[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float logit = 0.f;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) logit += lw[t][k] * lnw[k];
            const float R = 2.0f * wp_sigmoid(logit);
            float A = 0.f;
#pragma unroll
            for (int g = 0; g < WKN; ++g) A += kav[g] * bs[g][t];
            const float d = (live && olive) ? dw[(long)b * n + ((long)t * cout + oc) * cin + ic] : 0.f;
            const float Q = d * R * A;
            const float M = sav[t] * cav * fav;
            const float dAv = d * M * R;
            const float dl = d * M * A * R * (1.0f - 0.5f * R);
            a_sk[t] += Q * cav * fav;
            a_ca += Q * sav[t] * fav;
            a_fa += Q * sav[t] * cav;
#pragma unroll
            for (int g = 0; g < WKN; ++g) {
                a_sk[9 + g] += dAv * bs[g][t];
                accS[g][t] += kav[g] * dAv;
            }
#pragma unroll
            for (int k = 0; k < TAPS; ++k) {
                accL[t][k] += dl * lnw[k];
                a_g[k] += dl * lw[t][k];
            }
        }
        // per-sample reductions.  Over the four output channels of the workgroup through LDS, then one atomic per lane:
        // dca[b][i] and G[b][k][i] are contiguous over the lanes.
        float v = a_ca;
#pragma unroll
        for (int k = 0; k <= TAPS; ++k) {
            __syncthreads();
            red[wave][lane] = v;
            __syncthreads();
            if (wave == 0 && live) {
                const float s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
                if (k == 0) atomicAdd(dca + (long)b * cin + i, s);
                else atomicAdd(G + ((long)b * TAPS + (k - 1)) * cin + i, s);
            }
            if (k < TAPS) v = a_g[k];
        }
        // dfa[b][o], dsa[b][0..8] | dka[b][0..3]: sums over the lanes; the 13 + 1 results leave as one atomic instruction
        a_fa = wp_wave_sum(a_fa);
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const float s = wp_wave_sum(a_sk[j]);
            if (lane == j) mine = s;
        }
        if (olive) {
            if (lane < 13) atomicAdd(dsk + (long)b * 16 + lane, mine);
            if (lane == 13) atomicAdd(dfa + (long)b * cout + o, a_fa);
        }
    }
    if (!olive) return;
    // sums over the batch: S[g][y = o*3+u][x = i*3+v] (t = u*3+v), d loc_w[o*9+t][k]
    const int N2 = 3 * cin;
    if (live) {
#pragma unroll
        for (int g = 0; g < WKN; ++g)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                atomicAdd(S + (long)g * n + (long)(o * 3 + t / 3) * N2 + i * 3 + t % 3, accS[g][t]);
    }
    float mine = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const float s = wp_wave_sum(accL[t][k]);
            if (lane == t * TAPS + k) mine = s;
        }
    static_assert(9 * TAPS <= 64, "one atomic instruction per wave");
    if (lane < 9 * TAPS) atomicAdd(dlocw + (long)o * 9 * TAPS + lane, mine);
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

// one workgroup per sample: d ln[j] = sum_tau G[b][tau][j - tau + pad], then LayerNorm backward
__global__ __launch_bounds__(256) void wp_ln(const float* __restrict__ G, const float* __restrict__ pooled, int pooled_ld, sr_layer_params lp,
                                             float* __restrict__ dpooled, int dpooled_ld, int acc, float* __restrict__ dlnw,
                                             float* __restrict__ dlnb) {
    extern __shared__ float sh[];  // dnhat[cin] | nhat[cin] | red[8]
    const int cin = lp.cin, taps = lp.loc_taps, pad = (taps - 1) / 2;
    float* dn = sh;
    float* nh = sh + cin;
    float* red = nh + cin;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pooled + (long)b * pooled_ld;
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
        for (int k = 0; k < taps; ++k) {
            const int i = j - k + pad;  // output position whose window holds ln[j] at tap k
            if (i >= 0 && i < cin) acc += G[((long)b * taps + k) * cin + i];
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
    for (int j = tid; j < cin; j += 256) {
        float* d = dpooled + (long)b * dpooled_ld + j;
        *d = rstd * (dn[j] - a1 - nh[j] * a2) + (acc ? *d : 0.f);
    }
}

// KSM_Global backward, one workgroup for the whole batch (BatchNorm couples the samples in train mode).
// dpre: global scratch [B][per] for the pre-sigmoid gradients of the four heads (recomputing them inside the two loops that
// use them cost 3 dependent loads per term: 129 us per layer)
template <bool TRAIN, bool STAGED>
__global__ __launch_bounds__(1024) void wp_global(const float* __restrict__ pooled, int pooled_ld, sr_layer_params lp, int B,
                                                  const float* __restrict__ mods, const float* __restrict__ dca,
                                                  const float* __restrict__ dfa, const float* __restrict__ dsk, float* __restrict__ dpre_g,
                                                  WpGrads g, float* __restrict__ dpooled, int dpooled_ld) {
    extern __shared__ float sh[];
    const int cin = lp.cin, cout = lp.cout, ms = wp_mods_stride(cin, cout);
    float* zh = sh;                 // [B][ATT] normalised fc outputs
    float* rr = zh + B * WATT;      // relu(u)
    float* av = rr + B * WATT;      // StarReLU output a
    float* dz = av + B * WATT;      // da -> du -> dz
    float* stat = dz + B * WATT;    // mean[16] | rstd[16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    const int per = cin + cout + 9 + WKN;
    // STAGED: the pooled means, the fc weights, the four heads' weights and the pre-sigmoid gradients live in LDS.  This is
    // ONE workgroup on an otherwise idle chip: every global access is a full memory round trip, and the loops below are
    // chains of them (125 us per layer with everything read from global memory where it lies).
    float* ps = stat + 64;                         // [B][cin]
    float* fw = ps + (STAGED ? B * cin : 0);       // [ATT][cin]
    float* hw = fw + (STAGED ? WATT * cin : 0);    // [per][ATT]: channel_fc | filter_fc | spatial_fc | kernel_fc rows
    float* dpre = STAGED ? hw + per * WATT : dpre_g;  // [B][per]
    const float* pp = STAGED ? ps : pooled;
    const int pld = STAGED ? cin : pooled_ld;
    const float* fcw = STAGED ? fw : lp.fc_w;
    if (STAGED) {
        for (int i = tid; i < B * cin; i += nt) ps[i] = pooled[(long)(i / cin) * pooled_ld + i % cin];
        for (int i = tid; i < WATT * cin; i += nt) fw[i] = lp.fc_w[i];
        for (int i = tid; i < per * WATT; i += nt) {
            const int j = i / WATT;
            float v;
            if (j < cin) v = lp.ch_w[i];
            else if (j < cin + cout) v = lp.fl_w[i - cin * WATT];
            else if (j < cin + cout + 9) v = lp.sp_w[i - (cin + cout) * WATT];
            else v = lp.kn_w[i - (cin + cout + 9) * WATT];
            hw[i] = v;
        }
    }
    // pre-sigmoid gradients:  ca, fa, sa = 2 sigmoid(v): d/dv = y (1 - y/2);  ka = 2/KN sigmoid(v): d/dv = y (1 - y KN / 2)
    for (int it = tid; it < B * per; it += nt) {
        const int b = it / per, j = it - b * per;
        const float* md = mods + (long)b * ms;
        float v;
        if (j < cin) { const float y = md[j]; v = dca[(long)b * cin + j] * y * (1.0f - 0.5f * y); }
        else if (j < cin + cout) { const int r = j - cin; const float y = md[2 * cin + r]; v = dfa[(long)b * cout + r] * y * (1.0f - 0.5f * y); }
        else if (j < cin + cout + 9) { const int r = j - cin - cout; const float y = md[2 * cin + cout + r]; v = dsk[(long)b * 16 + r] * y * (1.0f - 0.5f * y); }
        else { const int r = j - cin - cout - 9; const float y = md[2 * cin + cout + 9 + r]; v = dsk[(long)b * 16 + 9 + r] * y * (1.0f - 0.5f * WKN * y); }
        dpre[it] = v;
    }
    __threadfence_block();
    __syncthreads();
    for (int pair = wave; pair < B * WATT; pair += nw) {
        const int b = pair / WATT, k = pair - b * WATT;
        const float* p = pp + (long)b * pld;
        const float* w = fcw + (long)k * cin;
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
    // the four heads as one [per][ATT] matrix: row j -> (weights, gradient destinations)
    auto head = [&](int j, const float*& w, float*& gw, float*& gb, int& row) {
        if (j < cin) { row = j; w = lp.ch_w; gw = g.ch_w; gb = g.ch_b; }
        else if (j < cin + cout) { row = j - cin; w = lp.fl_w; gw = g.fl_w; gb = g.fl_b; }
        else if (j < cin + cout + 9) { row = j - cin - cout; w = lp.sp_w; gw = g.sp_w; gb = g.sp_b; }
        else { row = j - cin - cout - 9; w = lp.kn_w; gw = g.kn_w; gb = g.kn_b; }
    };
    // da[b][k] = sum_rows dpre[b][row] W[row][k]: one wavefront per (b, k) pair, lanes over the rows
    for (int pair = wave; pair < B * WATT; pair += nw) {
        const int b = pair / WATT, k = pair - b * WATT;
        float s = 0.f;
        for (int j = lane; j < per; j += 64) {
            float wv;
            if (STAGED) wv = hw[j * WATT + k];
            else {
                const float* w;
                float *gw, *gb;
                int row;
                head(j, w, gw, gb, row);
                wv = w[row * WATT + k];
            }
            s += dpre[(long)b * per + j] * wv;
        }
        s = wp_wave_sum(s);
        if (lane == 0) dz[pair] = s;
    }
    // head weight / bias gradients: one thread per (row, k)
    for (int i = tid; i < per * WATT; i += nt) {
        const int j = i / WATT, k = i - j * WATT;
        const float* w;
        float *gw, *gb;
        int row;
        head(j, w, gw, gb, row);
        float s = 0.f, sb = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = dpre[(long)b * per + j];
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
        for (int b = 0; b < B; ++b) s += dz[b * WATT + k] * pp[(long)b * pld + c];
        g.fc_w[i] = s;
    }
    for (int i = tid; i < B * cin; i += nt) {
        const int b = i / cin, c = i - b * cin;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WATT; ++k) s += dz[b * WATT + k] * fcw[k * cin + c];
        dpooled[(long)b * dpooled_ld + c] += s;  // the local (LayerNorm) part was written by wp_ln
    }
}

struct WpLayout {
    size_t S, T, red, dpre, total;
    size_t S_floats, red_floats;
};
// S [4][9 cout cin] | T [4][3 cout][3 cin / 2 + 1] complex | reductions: G [B][taps][cin], dca [B][cin], dfa [B][cout],
// dsa|dka [B][16] (S and the reductions start from zero: one memset) | dpre [B][cin + cout + 13]
static WpLayout wp_layout(int B, int cin, int cout, int taps) {
    WpLayout l{};
    const size_t n = 9ul * cout * cin;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    l.S_floats = (size_t)WKN * n;
    l.S = carve(l.S_floats * 4);
    l.red_floats = (size_t)B * ((size_t)taps * cin + cin + cout + 16);
    l.red = carve(l.red_floats * 4);
    l.T = carve((size_t)WKN * 3 * cout * (3 * cin / 2 + 1) * 8);
    l.dpre = carve((size_t)B * (cin + cout + 9 + WKN) * 4);
    l.total = off;
    return l;
}
}  // namespace sr

using namespace sr;

extern "C" size_t sr_fdconv_weightgen_backward_workspace_bytes(int B, int cin, int cout) {
    if (B < 1 || cin < 1 || cout < 1) return 0;
    return wp_layout(B, cin, cout, 7).total;  // (taps <= 7: checked by the call)
}

extern "C" int sr_fdconv_weightgen_backward(int B, const float* pooled, int pooled_ld, const sr_layer_params* lp, int train,
                                            const float* mods, const float* dft_weight, const uint8_t* gmap, float alpha,
                                            const float* dw, const sr_weight_path_grads* grads, float* dpooled, int dpooled_ld,
                                            int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    (void)dft_weight;  // the spectrum enters through lp->basis; its gradient does not depend on its value beyond that
    SR_REQUIRE(B > 0 && B <= 512 && pooled && lp && mods && gmap && dw && grads && dpooled && workspace);
    SR_REQUIRE(lp->cin > 0 && lp->cout > 0 && lp->cin <= 4096 && lp->cout <= 4096 && lp->loc_taps >= 1 && (lp->loc_taps & 1) && lp->loc_taps <= 7);
    SR_REQUIRE(lp->basis && lp->fc_w && lp->bn_w && lp->bn_b && lp->bn_rm && lp->bn_rv && lp->relu_s && lp->relu_b && lp->ch_w && lp->fl_w &&
               lp->sp_w && lp->kn_w && lp->loc_w && lp->ln_w && lp->ln_b);
    SR_REQUIRE(pooled_ld >= lp->cin && dpooled_ld >= lp->cin && (3 * lp->cin) % 2 == 0);
    SR_REQUIRE(!train || B >= 2);
    const float* const* gp = (const float* const*)grads;
    for (size_t i = 0; i < sizeof(sr_weight_path_grads) / sizeof(float*); ++i) SR_REQUIRE(gp[i] != nullptr);
    const int cin = lp->cin, cout = lp->cout;
    SR_REQUIRE(3 * cin <= WP_TW && 3 * cout <= WP_TW);
    const int taps = lp->loc_taps;
    SR_REQUIRE(taps <= 7);  // KSM_Local's kernel: 5 taps for 16 <= cin < 512, 7 up to 2048 (FDConv.py:282-284)
    const WpLayout l = wp_layout(B, cin, cout, taps);
    if (workspace_bytes < l.total) return SR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* S = (float*)(ws + l.S);
    float2* T = (float2*)(ws + l.T);
    float* G = (float*)(ws + l.red);
    float* dca = G + (size_t)B * taps * cin;
    float* dfa = dca + (size_t)B * cin;
    float* dsk = dfa + (size_t)B * cout;
    float* dpre = (float*)(ws + l.dpre);
    WpGrads g;
    static_assert(sizeof(WpGrads) == sizeof(sr_weight_path_grads), "grads struct");
    ::memcpy(&g, grads, sizeof(g));
    const long n = 9L * cout * cin;
    const int N1 = 3 * cout, N2 = 3 * cin, n2h = N2 / 2 + 1;
    // S and the reductions are adjacent: one memset (l.red follows l.S up to alignment padding)
    SR_HIP(hipMemsetAsync(S, 0, (l.red - l.S) + l.red_floats * 4, st));
    const size_t n_locw = (size_t)cout * 9 * taps;
    if (g.ln_w == g.loc_w + n_locw && g.ln_b == g.ln_w + cin) {  // one flat gradient buffer (the Python host's): one memset
        SR_HIP(hipMemsetAsync(g.loc_w, 0, (n_locw + 2 * (size_t)cin) * 4, st));
    } else {
        SR_HIP(hipMemsetAsync(g.loc_w, 0, n_locw * 4, st));
        SR_HIP(hipMemsetAsync(g.ln_w, 0, (size_t)cin * 4, st));
        SR_HIP(hipMemsetAsync(g.ln_b, 0, (size_t)cin * 4, st));
    }
    // batch chunks: enough workgroups to fill the chip, as few atomic rounds on S / d loc_w as that allows
    const int wgs = cdiv(cin, 64) * cdiv(cout, 4);
    int nz = (512 + wgs - 1) / wgs;
    if (nz > B) nz = B;
    if (nz < 1) nz = 1;
    const int bchunk = cdiv(B, nz);
    const dim3 ge(cdiv(cin, 64), cdiv(cout, 4), cdiv(B, bchunk));
#define SR_WPE(TAPS_) wp_elem<TAPS_><<<ge, 256, 0, st>>>(dw, mods, *lp, B, bchunk, S, g.loc_w, G, dca, dfa, dsk)
    if (taps == 1) SR_WPE(1);
    else if (taps == 3) SR_WPE(3);
    else if (taps == 5) SR_WPE(5);
    else SR_WPE(7);
#undef SR_WPE
    SR_LAUNCH_CHECK();
    (void)n;
    const long na = (long)WKN * N1 * n2h, nb = (long)N1 * n2h;
    wp_dft_a<<<cdiv(na, 256), 256, 0, st>>>(S, N1, N2, T);
    SR_LAUNCH_CHECK();
    wp_dft_b<<<cdiv(nb, 256), 256, 0, st>>>(T, gmap, alpha, N1, N2, g.dft);
    SR_LAUNCH_CHECK();
    wp_ln<<<B, 256, (2 * (size_t)cin + 8) * 4, st>>>(G, pooled, pooled_ld, *lp, dpooled, dpooled_ld, accumulate, g.ln_w, g.ln_b);
    SR_LAUNCH_CHECK();
    const int per = cin + cout + 9 + WKN;
    const size_t sh_base = ((size_t)4 * B * WATT + 64 + 64) * 4;
    const size_t sh_staged = sh_base + ((size_t)B * cin + (size_t)WATT * cin + (size_t)per * WATT + (size_t)B * per) * 4;
    const bool staged = sh_staged <= 150 * 1024;
    const size_t shg = staged ? sh_staged : sh_base;
#define SR_WPG(TR_, ST_)                                                                                                            \
    do {                                                                                                                            \
        SR_LDS_ATTR(shg, wp_global<TR_, ST_>);                                                                                      \
        wp_global<TR_, ST_><<<1, 1024, shg, st>>>(pooled, pooled_ld, *lp, B, mods, dca, dfa, dsk, dpre, g, dpooled, dpooled_ld);     \
    } while (0)
    if (train) {
        if (staged) SR_WPG(true, true);
        else SR_WPG(true, false);
    } else {
        if (staged) SR_WPG(false, true);
        else SR_WPG(false, false);
    }
#undef SR_WPG
    SR_LAUNCH_CHECK();
    return SR_OK;
}
