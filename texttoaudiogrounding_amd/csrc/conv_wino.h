// Internals shared by conv_wino.hip (plane form: transform kernels + batched products) and conv_wino_fused.hip (one kernel per
// convolution: transforms inside the product kernel).  Not part of the C ABI.
#pragma once
#include "tag_common.h"

// What the output side of a Winograd convolution does besides y = A^T m A (the EPI forms of conv3x3_halo_kernel):
//   kind 0: BatchNorm batch statistics of y (partial rows [K | r | q][C] + counts) when a statistics buffer is given, else nothing
//   kind 1: rows [sum g | sum g xhat][C] of the BatchNorm+ReLU backward the gradient flows into (yref = that BatchNorm's input)
//   kind 2: the same sums for the BatchNorm+ReLU+pool+dropout backward of the block BELOW (yref unpooled, window ph x 2)
//   kind 3: inference -- out = pool(relu(y * scale + shift)), the raw y is never written
struct WinoEpi {
    const float* yref;
    const float* scale; const float* shift; const float* mean; const float* invstd;
    int ph; float wavg, wmax;   // kind 2 / 3: pool window ph x 2 and the weights of its average / maximum
    int Hf, Wf;                 // kind 2: yref is the UNPOOLED (B,Hf,Wf,C) tensor of the block below, H = Hf / ph, W = Wf / 2
    float drop_p; unsigned long long seed;
    int kind;
};

// ---- fused form (conv_wino_fused.hip) ----
// shapes the fused forward / dgrad kernel takes: 64-cout workgroup tiles, 8-channel K chunks
bool wino_fused_ok(int Cin, int Cout);
// partial-statistics rows of the fused kernel: one per 64-tile block
int wino_fused_rows(int B, int H, int W);
// y (or the pooled output, kind 3) = conv(prologue(x)) with U [16][Cin][Cout]; stats / epi as above (epi may be NULL = kind 0)
int wino_fused_run(const float* x, const float* U, int pro, const float* in_scale, const float* in_shift, float* y, float* stats,
                   const WinoEpi* epi, int B, int H, int W, int Cin, int Cout, hipStream_t st);
// shapes of the fused weight gradient: 64 x 64 (Cin x Cout) workgroup tiles
bool wino_fused_wgrad_ok(int Cin, int Cout);
size_t wino_fused_wgrad_ws_floats(int B, int H, int W, int Cin, int Cout);
int wino_fused_wgrad_run(const float* x, int pro, const float* in_scale, const float* in_shift, const float* dy, float* dw, int B,
                         int H, int W, int Cin, int Cout, float* ws, hipStream_t st);
