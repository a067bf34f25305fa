// FBM low-pass plan shared by the generic and the power-of-two kernels.
#pragma once
#include <hip/hip_runtime.h>

struct sr_fbm_plan {
    int H, W;
    int nfW, nfH;
    int radW[24], radH[24];
    int Pw, Ph;            // channel pairs per workgroup in the row / column passes
    int n[3];              // kept full-spectrum columns for k=2,4,8
    float2* tw_w;          // device: exp(-2*pi*i*j/W)
    float2* tw_h;          // device: exp(-2*pi*i*j/H)
    int* cols;             // device: [n[0]] kept column index (mask k=2), ascending
    int* colinfo;          // device: [n[0]][4] = {variant, index in k=4 list or -1, index in k=8 list or -1, 0}
    int* cols4;            // device: [n[1]]
    int* cols8;            // device: [n[2]]
    float* rowmask;        // device: [3 masks][3 variants][H]
    void* dev_block;       // one allocation backing all of the above
    int half[3];           // kept columns of mask k are [0,half) u (W-half, W) when contiguous != 0
    int contiguous;
    // generic FFT, largest radix R an odd prime > 7 (it runs last in forward, first in inverse transforms): conjugate
    // pairs (u, R - u) of its outputs / operands at or beyond these indices are never used / all zero (fbm.hip)
    int ucutW, ucutH;      // forward along W / H: output pairs u >= ucut fall on columns / rows no mask keeps
    int tcutW[3];          // inverse along W, band k: operand pairs t >= tcut hold no kept column
    int tcutH[9];          // inverse along H, [mask k][variant]: operand pairs t >= tcut have zero row-mask weights
    // forward row transform of a nearest-2x upsampled source: one W/2-point transform per source row (fbm_rows_fwd_up)
    int nfWu, radWu[24], Pwu, ucutWu;
    int row_half[3];       // band k: every variant of its row mask lies inside [0, row_half) u (H - row_half, H) (fbm_fast.h prunes by it)
    // ---- real-basis tables of the MFMA path (fbm_mfma.hip); mf_ok == 0 when the shape / masks are not covered ----
    int mf_ok;
    int mf_jw;             // x-basis functions kept by mask k=2 (DC, dummy, then cos c / sin c pairs) = 2 * half[0]
    int mf_nw[3];          // prefix of them kept by mask k
    int mf_nh;             // y-basis functions with non-zero weight in mask k=2 (same ordering)
    int mf_nhk[3];         // prefix with non-zero symmetric weight per mask
    int mf_jw_fill, mf_nh_fill; // functions actually present in the analysis tables (rows beyond them are zero)
    int mf_ld_qw, mf_ld_ph; // padded k extents (multiples of 16) of the two synthesis tables
    void* mf_block;        // one device allocation: the tables below
    void* mf_qwt;          // fp16 [pad32(jw)][W]      q_j[w]        analysis along W
    void* mf_qw;           // fp16 [pad32(W)][pad16(jw)] q_j[w]      synthesis along W (band k uses the first nw[k] functions)
    void* mf_pht;          // fp16 [pad32(nh)][H]      p_n[h]        analysis along H
    void* mf_ph;           // fp16 [pad32(H)][pad16(nh)] 16 p_n[h]   synthesis along H (shared by the three bands)
    void* mf_qwf;          // fp16 folded synthesis along W (W % 64 == 0): tiles of columns 0 .. W/2 - 1, k = cos | sin halves; last tile = column W/2
    void* mf_win;          // float2 [3][32]: (s, d) weights of the 32 y-coefficient rows after the main k-steps of band k
    int mf_ksm[3];         // main k-steps of band k: y-coefficients 0 .. 16 * ksm - 1 enter with weight one
    int mf_nwin[3];        // window k-steps after them (0..2)
};

namespace sr {
int fbm_mfma_plan_build(sr_fbm_plan* pl, const unsigned char* rows /*[3][H]*/);  // SR_OK also when not eligible
void fbm_mfma_plan_free(sr_fbm_plan* pl);
}

