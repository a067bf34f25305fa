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
};

