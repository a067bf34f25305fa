// FBM spatial gates: g_k = 2*sigmoid(conv3x3(x; w_k) + b_k), k = 1..3  (FDConv.py:460-461, 417-419).
// Cin -> 3 output planes: a bandwidth-shaped kernel.  One workgroup = one 16x16 pixel tile; the
// halo tile of a few channels is staged in LDS plane-major so that the 9-tap reads of
// neighbouring lanes hit neighbouring banks.
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
    const int Hs = H >> in.ups, Ws = W >> in.ups;
    const float* wk[3] = {w0, w1, w2};
    float acc[3] = {0.f, 0.f, 0.f};
    for (int s = 0; s < in.nseg; ++s) {
        const View v = in.x[s];
        const T* base = (const T*)v.p + (long)b * Hs * Ws * v.ld + v.c0;
        for (int cc = 0; cc < v.C; cc += GCC) {
            const int nc = min(GCC, v.C - cc);
            __syncthreads();
            for (int it = threadIdx.x; it < (GT + 2) * (GT + 2) * nc; it += blockDim.x) {
                const int c = it % nc, pi = it / nc;
                const int py = pi / (GT + 2), px = pi % (GT + 2);
                const int h = ty0 + py - 1, w = tx0 + px - 1;
                float val = 0.f;
                if (h >= 0 && h < H && w >= 0 && w < W)
                    val = Elem<T>::ld(base + ((long)(h >> in.ups) * Ws + (w >> in.ups)) * v.ld + cc + c);
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

int fbm_gates_launch(int dtype, int B, int H, int W, const Concat& in, const sr_layer_params& lp, float* gates,
                     hipStream_t st) {
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
    Concat c;
    int rc = to_concat(in, &c);
    if (rc != SR_OK) return rc;
    SR_REQUIRE(c.ctot == lp->cin);
    SR_REQUIRE(!c.ups || (H % 2 == 0 && W % 2 == 0));
    for (int k = 0; k < 3; ++k) SR_REQUIRE(lp->gate_w[k] && lp->gate_b[k]);
    return fbm_gates_launch(dtype, B, H, W, c, *lp, gates, (hipStream_t)stream);
}
