// Winograd F(2x2, 3x3) form of the 3x3 / stride 1 / padding 1 convolutions of the DEEP ConvBlocks (models/panns.py:29-38,
// 49-50: conv1 / conv2 of blocks 3 and 4 of Cnn8Rnn, models/audio_encoder.py:134-138): forward, dgrad and weight gradient of the
// training step and the inference forward, all fp32.
//
//   y  = A^T [ (G g G^T) (.) (B^T d B) ] A             per 2 x 2 output tile, 4 x 4 input window d, filter g (Lavin & Gray 2015)
//   dw = G^T [ sum over tiles (A dY A^T) (.) (B^T d B) ] G                       (the adjoint in g)
//
// 16 multiplies per 2 x 2 outputs and channel pair instead of 36: the contraction over the input channels becomes 16
// independent dense products  M_xi (T x Cout) = V_xi (T x Cin) . U_xi (Cin x Cout),  T = B * ceil(H/2) * ceil(W/2) tiles,
// with 2.25 x fewer MFMA FLOP than the direct kernels (csrc/conv.hip) spend -- those run at 0.86-0.88 of the fp32 MFMA peak on
// these layers, so executing fewer FLOP is the lever that is left.  Every operation is fp32 (transforms on the VALU, products on
// v_mfma_f32_32x32x2_f32 through the dense GEMM kernel of gemm.hip as ONE batched launch); the transform constants are 0, +-1
// and 1/2; against an fp64 convolution the result is at least as close as the direct kernel's on these layers (512 input
// channels, tools/wino_bench.py: 9.0e-7 max / 1.2e-7 rms of the output range against 2.0e-6 / 1.7e-7 -- sixteen chains of Cin
// products round less than one chain of 9 Cin).  oracle/tag_oracle.py winograd_conv3x3 / winograd_conv3x3_wgrad restate the
// algorithm stage by stage.  Kernels:
//   wino_pack_kernel     U = G g G^T of the filter (forward) and of the tap-flipped, channel-swapped filter (dgrad)
//   wino_input_kernel    x (B,H,W,Cin) -> V [16][T][Cin]: producer BatchNorm + ReLU prologue and zero padding applied on load
//                        (the prologue modes of conv3x3_halo_kernel), B^T d B on 4 channels per thread, 16-byte accesses
//   gemm_kernel<..,16>   (gemm.hip, tag_launch_gemm_batched) the 16 products, 128 x 128 tiles, 16-wide K chunk
//   wino_output_kernel   M -> y = A^T m A and, in the same pass, EITHER the BatchNorm batch statistics of y (pivoted partial
//                        rows [K | r | q] + counts, folded by tag_bn_stats_from_partials), OR the sums of the BatchNorm+ReLU
//                        backward the gradient flows into (rows [sum g | sum g xhat], tag_bn_grad_from_partials) -- the
//                        EPI == 0 / 1 epilogues of the direct kernel --, OR (dgrad of a block's first conv) the sums of the
//                        BatchNorm+ReLU+pool+dropout backward of the block below on the thread's own 2 x 2 tile (EPI == 2), OR
//                        (inference) BatchNorm(eval) + ReLU + avg/max pool of its own 2 x 2 tile, the direct kernel's EPI == 3
//   wino_dy_kernel, wino_wgrad_finish_kernel   the weight gradient: D = A dY A^T, 16 S products over S slices of the tile axis
//                        as one batched launch, the slices folded in a fixed order and G^T . G applied; the input planes can be
//                        the ones the forward launch of the same convolution left behind (v_keep / v_saved)
// The transform passes are HBM-bound (V and M are 4 x the activation each; 5-7 TB/s); per launch at B = 64 the 512 -> 512 layer
// takes 2.65-2.7 ms (forward / dgrad / weight gradient) against 4.4-4.5 ms direct.  Fixed summation order, no atomics.
#include "conv_wino.h"

int tag_launch_gemm_batched(const float* A, int lda, long sA, const float* B, int ldb, long sB, float* C, int ldc, long sC, int M,
                            int N, int K, int batch, hipStream_t st, int transA);

namespace {

__device__ __forceinline__ f32x4 wino_prologue(f32x4 v, int mode, f32x4 s, f32x4 t) {      // = apply_prologue of conv.hip
    if (mode == 1) {
        v.x = fmaxf(fmaf(v.x, s.x, t.x), 0.0f); v.y = fmaxf(fmaf(v.y, s.y, t.y), 0.0f);
        v.z = fmaxf(fmaf(v.z, s.z, t.z), 0.0f); v.w = fmaxf(fmaf(v.w, s.w, t.w), 0.0f);
    } else if (mode == 2) {
        v.x = fmaf(v.x > 0 ? v.x : 0.1f * v.x, s.x, t.x); v.y = fmaf(v.y > 0 ? v.y : 0.1f * v.y, s.y, t.y);
        v.z = fmaf(v.z > 0 ? v.z : 0.1f * v.z, s.z, t.z); v.w = fmaf(v.w > 0 ? v.w : 0.1f * v.w, s.w, t.w);
    } else if (mode == 3) {
        v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y); v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
    }
    return v;
}

// U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]: Uf[xi][ci][co] from w[co][ci][ky][kx] (forward), and
// Ud[xi][co][ci] from the tap-flipped filter w[co][ci][2-ky][2-kx] (dgrad: the transposed convolution as a convolution
// from the Cout-channel gradient to the Cin-channel gradient).  One thread per (ci, co) pair.
// tiled_f / tiled_d: the layout of the fused kernel (conv_wino_fused.hip) instead -- per (64-cout block, 8-channel K chunk) the
// LDS image [xi][n 64][k 8] of that kernel as one contiguous 32 KB block: forward index ((((co >> 6) (Cin / 8) + (ci >> 3)) 16 + xi)
// 64 + (co & 63)) 8 + (ci & 7); dgrad (K = co, N = ci) the same with the roles of ci and co exchanged.
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ Uf,
                                                        float* __restrict__ Ud, int Cin, int Cout, int tiled_f, int tiled_d) {
    const long n = (long)Cin * Cout;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int co = (int)(e / Cin), ci = (int)(e % Cin);
        float g[3][3];
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = w[e * 9 + k];
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            float* U = dir == 0 ? Uf : Ud;
            if (!U) continue;
            float p[4][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float g0 = dir == 0 ? g[0][c] : g[2][2 - c], g1 = dir == 0 ? g[1][c] : g[1][2 - c],
                            g2 = dir == 0 ? g[2][c] : g[0][2 - c];
                p[0][c] = g0;
                p[1][c] = 0.5f * (g0 + g1 + g2);
                p[2][c] = 0.5f * (g0 - g1 + g2);
                p[3][c] = g2;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u0 = p[r][0], u1 = 0.5f * (p[r][0] + p[r][1] + p[r][2]), u2 = 0.5f * (p[r][0] - p[r][1] + p[r][2]),
                            u3 = p[r][2];
                size_t base = dir == 0 ? (size_t)ci * Cout + co : (size_t)co * Cin + ci, plane = (size_t)n;
                if (dir == 0 && tiled_f) {
                    base = ((size_t)((co >> 6) * (Cin >> 3) + (ci >> 3)) * 16 * 64 + (co & 63)) * 8 + (ci & 7);
                    plane = 64 * 8;
                } else if (dir == 1 && tiled_d) {
                    base = ((size_t)((ci >> 6) * (Cout >> 3) + (co >> 3)) * 16 * 64 + (ci & 63)) * 8 + (co & 7);
                    plane = 64 * 8;
                }
                U[(size_t)(4 * r + 0) * plane + base] = u0;
                U[(size_t)(4 * r + 1) * plane + base] = u1;
                U[(size_t)(4 * r + 2) * plane + base] = u2;
                U[(size_t)(4 * r + 3) * plane + base] = u3;
            }
        }
    }
}

// V[xi = 4 r + s][t][c] = (B^T d B)[r][s],  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],  d = the 4 x 4 input window of
// tile t = (b, i, j): rows 2 i - 1 .. 2 i + 2, columns 2 j - 1 .. 2 j + 2 (zero outside the image, AFTER the prologue).
// A thread owns 4 consecutive channels of one tile; the C / 4 threads of a tile are neighbours (16-byte coalesced both ways).
template <int PRO>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift, float* __restrict__ V, int B, int H,
                                                         int W, int C, int th, int tw, long T, long Tpad) {
    const int cq = C >> 2;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long t = item / cq;
    const int q = (int)(item - t * cq);
    if (t >= Tpad) return;
    const size_t plane = (size_t)Tpad * C;
    if (t >= T) {                              // rows that only pad the K slices of the weight-gradient products: zero
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
            *reinterpret_cast<f32x4*>(V + (size_t)xi * plane + (size_t)t * C + 4 * q) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    const int j = (int)(t % tw);
    const long bi = t / tw;
    const int i = (int)(bi % th), b = (int)(bi / th);
    f32x4 rs = {1.0f, 1.0f, 1.0f, 1.0f}, rt = {0.0f, 0.0f, 0.0f, 0.0f};
    if (PRO != 0) {
        rs = *reinterpret_cast<const f32x4*>(in_scale + 4 * q);
        rt = *reinterpret_cast<const f32x4*>(in_shift + 4 * q);
    }
    const float* xb = x + (size_t)b * H * W * C + 4 * q;
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = 2 * i - 1 + r;
        const bool okh = (unsigned)h < (unsigned)H;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int w = 2 * j - 1 + s;
            const bool ok = okh && (unsigned)w < (unsigned)W;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ok) v = wino_prologue(*reinterpret_cast<const f32x4*>(xb + ((size_t)h * W + w) * C), PRO, rs, rt);
            d[r][s] = v;
        }
    }
    f32x4 tt[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tt[0][s] = d[0][s] - d[2][s];
        tt[1][s] = d[1][s] + d[2][s];
        tt[2][s] = d[2][s] - d[1][s];
        tt[3][s] = d[1][s] - d[3][s];
    }
    float* vp = V + (size_t)t * C + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * r + 0) * plane) = tt[r][0] - tt[r][2];
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * r + 1) * plane) = tt[r][1] + tt[r][2];
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * r + 2) * plane) = tt[r][2] - tt[r][1];
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * r + 3) * plane) = tt[r][1] - tt[r][3];
    }
}


// y (2 x 2 pixels of tile t) = A^T m A,  A^T = [[1,1,1,0],[0,1,-1,-1]],  m[r][s] = M[4 r + s][t][c].  A workgroup holds
// G = 256 / (C / 4) tile slots and walks ITERS tiles per slot (tile = (blockIdx.x * ITERS + it) * G + slot); every slot
// writes ONE partial row (row = blockIdx.x * G + slot) of
//   EPI == 0: [K | r | q][C] + count -- pivot K = the thread's first output, r = sum(y - K), q = sum((y - K)^2) over the
//             slot's pixels (tag_bn_stats_from_partials; the layout of the direct kernel's EPI == 0 and of conv_c1_fwd_rows);
//   EPI == 1: [sum g | sum g xhat][C], g = y where bn(yref) > 0 (tag_bn_grad_from_partials; the direct kernel's EPI == 1).
//   EPI == 3 (inference, BatchNorm in eval mode): no rows; the tile's 2 x 2 outputs ARE the pool windows (one 2 x 2 window, or two
//             1 x 2 windows), so out (B, H/ph, W/2, C) = avg/max pool(relu(y * scale + shift)) leaves directly -- the raw conv output
//             is never written (the expression and summation order of bnact_pool_fwd_kernel / the direct kernel's EPI == 3).
constexpr int WINO_ITERS = 8;
template <int EPI>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, float* __restrict__ y,
                                                          float* __restrict__ stats, WinoEpi epi, int B, int H, int W, int C,
                                                          int th, int tw, long T, int P) {
    const int cq = C >> 2, G = 256 / cq;
    const int slot = threadIdx.x / cq, q = threadIdx.x - slot * cq;
    const size_t plane = (size_t)T * C;
    float sk[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float cnt = 0.0f;
    bool have_pivot = false;
    f32x4 bsc = {0, 0, 0, 0}, bsh = {0, 0, 0, 0}, bmu = {0, 0, 0, 0}, bis = {0, 0, 0, 0};
    if (EPI == 1 || EPI == 2 || EPI == 3) {
        bsc = *reinterpret_cast<const f32x4*>(epi.scale + 4 * q); bsh = *reinterpret_cast<const f32x4*>(epi.shift + 4 * q);
    }
    if (EPI == 1 || EPI == 2) {
        bmu = *reinterpret_cast<const f32x4*>(epi.mean + 4 * q); bis = *reinterpret_cast<const f32x4*>(epi.invstd + 4 * q);
    }
    for (int it = 0; it < WINO_ITERS; ++it) {
        const long t = ((long)blockIdx.x * WINO_ITERS + it) * G + slot;
        if (t >= T) break;
        const int j = (int)(t % tw);
        const long bi = t / tw;
        const int i = (int)(bi % th), b = (int)(bi / th);
        const float* mp = M + (size_t)t * C + 4 * q;
        f32x4 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) m[r][s] = *reinterpret_cast<const f32x4*>(mp + (size_t)(4 * r + s) * plane);
        f32x4 u[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u[0][s] = m[0][s] + m[1][s] + m[2][s];
            u[1][s] = m[1][s] - m[2][s] - m[3][s];
        }
        if (EPI == 3) {
            const int Hp = H / epi.ph, Wp = W >> 1;
            f32x4 a[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f32x4 o0 = u[r][0] + u[r][1] + u[r][2], o1 = u[r][1] - u[r][2] - u[r][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[r][0][k] = fmaxf(fmaf(o0[k], bsc[k], bsh[k]), 0.0f);
                    a[r][1][k] = fmaxf(fmaf(o1[k], bsc[k], bsh[k]), 0.0f);
                }
            }
            if (j < Wp) {
                if (epi.ph == 2) {
                    if (i < Hp) {
                        f32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sum = ((a[0][0][k] + a[0][1][k]) + a[1][0][k]) + a[1][1][k];
                            const float mx = fmaxf(fmaxf(fmaxf(a[0][0][k], a[0][1][k]), a[1][0][k]), a[1][1][k]);
                            o[k] = sum * epi.wavg + mx * epi.wmax;
                        }
                        *reinterpret_cast<f32x4*>(y + (((size_t)b * Hp + i) * Wp + j) * C + 4 * q) = o;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int hp = 2 * i + r;
                        if (hp >= Hp) continue;
                        f32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            o[k] = (a[r][0][k] + a[r][1][k]) * epi.wavg + fmaxf(a[r][0][k], a[r][1][k]) * epi.wmax;
                        *reinterpret_cast<f32x4*>(y + (((size_t)b * Hp + hp) * Wp + j) * C + 4 * q) = o;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int h = 2 * i + a;
            if (h >= H) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int w = 2 * j + e;
                if (w >= W) continue;
                const f32x4 o = e == 0 ? u[a][0] + u[a][1] + u[a][2] : u[a][1] - u[a][2] - u[a][3];
                const size_t off = (((size_t)b * H + h) * W + w) * C + 4 * q;
                *reinterpret_cast<f32x4*>(y + off) = o;
                const float ov[4] = {o.x, o.y, o.z, o.w};
                if (EPI == 0 && stats) {
                    if (!have_pivot) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) sk[k] = ov[k];
                        have_pivot = true;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float dv = ov[k] - sk[k]; s1[k] += dv; s2[k] = fmaf(dv, dv, s2[k]); }
                    cnt += 1.0f;
                }
                if (EPI == 2) {
                    // the dgrad of a block's FIRST conv: o = dL/d(dropout(pool(relu(bn(yref))))) at pooled pixel (h, w) of the block
                    // BELOW; its BatchNorm+ReLU+pool+dropout backward needs sum(dz), sum(dz xhat) over the ph x 2 window of yref before
                    // any dy can be formed -- the arithmetic of conv3x3_halo_kernel's EPI == 2 / pool_bwd_reduce_kernel: undo the
                    // dropout (one hash per 4 channels), recompute a = bn(yref), the ReLU mask and the first-maximum arg-max
                    f32x4 g = o;
                    if (epi.drop_p > 0.0f) {
                        const uint64_t grp = ((uint64_t)((unsigned)b * (unsigned)H + (unsigned)h) * (unsigned)W + (unsigned)w) * (unsigned)cq
                                             + (unsigned)q;
                        const uint64_t bits = tag_keep4_bits(epi.seed, grp);
                        const float keep_scale = 1.0f / (1.0f - epi.drop_p);
                        const unsigned thr = tag_keep4_threshold(epi.drop_p);
#pragma unroll
                        for (int k = 0; k < 4; ++k) g[k] = tag_keep4(bits, k, thr) ? g[k] * keep_scale : 0.0f;
                    }
                    const float* yw = epi.yref + (((size_t)b * epi.Hf + (size_t)h * epi.ph) * epi.Wf + 2 * w) * C + 4 * q;
                    f32x4 vw[4];
                    vw[0] = *reinterpret_cast<const f32x4*>(yw);
                    vw[1] = *reinterpret_cast<const f32x4*>(yw + C);
                    if (epi.ph == 2) {
                        vw[2] = *reinterpret_cast<const f32x4*>(yw + (size_t)epi.Wf * C);
                        vw[3] = *reinterpret_cast<const f32x4*>(yw + (size_t)epi.Wf * C + C);
                    } else { vw[2] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; vw[3] = vw[2]; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float a4[4];
#pragma unroll
                        for (int u2 = 0; u2 < 4; ++u2) a4[u2] = fmaf(vw[u2][k], bsc[k], bsh[k]);
                        if (epi.ph != 2) { a4[2] = -INFINITY; a4[3] = -INFINITY; }
                        const float mx = fmaxf(fmaxf(a4[0], a4[1]), fmaxf(a4[2], a4[3]));
                        const float gw = g[k] * epi.wavg, gwm = g[k] * (epi.wavg + epi.wmax);
                        bool found = false;
#pragma unroll
                        for (int u2 = 0; u2 < 4; ++u2) {
                            const bool eq = a4[u2] == mx;
                            const bool hit = eq && !found;
                            found = found || eq;
                            const float dz = a4[u2] > 0.0f ? (hit ? gwm : gw) : 0.0f;
                            s1[k] += dz;
                            s2[k] = fmaf(dz, (vw[u2][k] - bmu[k]) * bis[k], s2[k]);
                        }
                    }
                }
                if (EPI == 1) {
                    const f32x4 yr = *reinterpret_cast<const f32x4*>(epi.yref + off);
                    const float yv[4] = {yr.x, yr.y, yr.z, yr.w};
                    const float sc[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, sh[4] = {bsh.x, bsh.y, bsh.z, bsh.w};
                    const float mu[4] = {bmu.x, bmu.y, bmu.z, bmu.w}, is[4] = {bis.x, bis.y, bis.z, bis.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float g = fmaf(yv[k], sc[k], sh[k]) > 0.0f ? ov[k] : 0.0f;
                        s1[k] += g;
                        s2[k] = fmaf(g, (yv[k] - mu[k]) * is[k], s2[k]);
                    }
                }
            }
        }
    }
    if (EPI == 3) return;
    const int prow = blockIdx.x * G + slot;
    if (slot >= G || prow >= P) return;
    if (EPI == 0 && stats) {
        float* ps = stats + (size_t)prow * 3 * C + 4 * q;
        *reinterpret_cast<f32x4*>(ps) = (f32x4){sk[0], sk[1], sk[2], sk[3]};
        *reinterpret_cast<f32x4*>(ps + C) = (f32x4){s1[0], s1[1], s1[2], s1[3]};
        *reinterpret_cast<f32x4*>(ps + 2 * C) = (f32x4){s2[0], s2[1], s2[2], s2[3]};
        if (q == 0) stats[(size_t)P * 3 * C + prow] = cnt;
    }
    if (EPI == 1 || EPI == 2) {
        float* ps = stats + (size_t)prow * 2 * C + 4 * q;
        *reinterpret_cast<f32x4*>(ps) = (f32x4){s1[0], s1[1], s1[2], s1[3]};
        *reinterpret_cast<f32x4*>(ps + C) = (f32x4){s2[0], s2[1], s2[2], s2[3]};
    }
}

// Weight gradient in the Winograd domain: dw = G^T [ sum_t (A dY_t A^T) (.) (B^T d_t B) ] G  (the adjoint of the forward form),
// A = [[1,0],[1,1],[1,-1],[0,-1]].  D[xi][t][c] = (A dY A^T)[r][s] of the 2 x 2 output-gradient tile t (zero outside the image;
// rows t >= T pad the K slices with zeros).  Same thread layout as wino_input_kernel.
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ D, int B, int H, int W,
                                                      int C, int th, int tw, long T, long Tpad) {
    const int cq = C >> 2;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long t = item / cq;
    const int q = (int)(item - t * cq);
    if (t >= Tpad) return;
    const size_t plane = (size_t)Tpad * C;
    float* dp = D + (size_t)t * C + 4 * q;
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    if (t >= T) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(dp + (size_t)xi * plane) = z;
        return;
    }
    const int j = (int)(t % tw);
    const long bi = t / tw;
    const int i = (int)(bi % th), b = (int)(bi / th);
    f32x4 g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int h = 2 * i + a, w = 2 * j + e;
            g[a][e] = (h < H && w < W) ? *reinterpret_cast<const f32x4*>(dy + (((size_t)b * H + h) * W + w) * C + 4 * q) : z;
        }
    f32x4 r[4][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        r[0][e] = g[0][e];
        r[1][e] = g[0][e] + g[1][e];
        r[2][e] = g[0][e] - g[1][e];
        r[3][e] = z - g[1][e];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        *reinterpret_cast<f32x4*>(dp + (size_t)(4 * k + 0) * plane) = r[k][0];
        *reinterpret_cast<f32x4*>(dp + (size_t)(4 * k + 1) * plane) = r[k][0] + r[k][1];
        *reinterpret_cast<f32x4*>(dp + (size_t)(4 * k + 2) * plane) = r[k][0] - r[k][1];
        *reinterpret_cast<f32x4*>(dp + (size_t)(4 * k + 3) * plane) = z - r[k][1];
    }
}

// dw (Cout,Cin,3,3) = G^T (sum over the S K-slices of Pp[xi][s][co][ci]) G, fixed summation order; one thread per (co, ci)
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ Pp, int S, int Cin, int Cout,
                                                                float* __restrict__ dw) {
    const long n = (long)Cin * Cout;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        float u[4][4];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            float a = 0.0f;
            for (int s = 0; s < S; ++s) a += Pp[((size_t)xi * S + s) * n + e];
            u[xi >> 2][xi & 3] = a;
        }
        float q[3][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float hs = 0.5f * (u[1][s] + u[2][s]), hd = 0.5f * (u[1][s] - u[2][s]);
            q[0][s] = u[0][s] + hs;
            q[1][s] = hd;
            q[2][s] = hs + u[3][s];
        }
        float* o = dw + e * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float hs = 0.5f * (q[a][1] + q[a][2]), hd = 0.5f * (q[a][1] - q[a][2]);
            o[3 * a + 0] = q[a][0] + hs;
            o[3 * a + 1] = hd;
            o[3 * a + 2] = hs + q[a][3];
        }
    }
}

// K slices of the weight-gradient products: (xi, slice) pairs are the batch entries of ONE launch, so a slice must be a whole
// number of 16-row K chunks and the planes are padded to S * kc rows
struct WinoWgGeom { int th, tw; long T, Tpad; int S, kc; };
inline WinoWgGeom wino_wg_geom(int B, int H, int W, int Cin, int Cout) {
    WinoWgGeom g;
    g.th = (H + 1) / 2; g.tw = (W + 1) / 2;
    g.T = (long)B * g.th * g.tw;
    const long tiles = (long)((Cin + 127) / 128) * ((Cout + 127) / 128) * 16;
    long S = (1024 + tiles - 1) / tiles;                       // ~1024 workgroups: one residency round of 4 per CU
    if (S > g.T / 512) S = g.T / 512;
    if (S < 1) S = 1;
    long kc = (g.T + S - 1) / S;
    kc = (kc + 15) / 16 * 16;
    g.S = (int)((g.T + kc - 1) / kc);
    g.kc = (int)kc;
    g.Tpad = (long)g.S * kc;
    return g;
}

struct WinoGeom { int th, tw; long T; int G, P; };
inline WinoGeom wino_geom(int B, int H, int W, int Cout) {
    WinoGeom g;
    g.th = (H + 1) / 2; g.tw = (W + 1) / 2;
    g.T = (long)B * g.th * g.tw;
    g.G = 256 / (Cout / 4);
    const long per_wg = (long)g.G * WINO_ITERS;
    g.P = (int)((g.T + per_wg - 1) / per_wg) * g.G;
    return g;
}
// channel counts the transform kernels take: C / 4 threads per tile must divide a 256-thread workgroup
inline bool wino_channels_ok(int C) { return C >= 32 && C <= 1024 && C % 4 == 0 && 256 % (C / 4) == 0; }

int wino_run(const float* x, const float* U, int pro, const float* s, const float* t, float* y, float* stats, const WinoEpi* epi,
             int B, int H, int W, int Cin, int Cout, float* ws, hipStream_t st, float* v_keep = nullptr) {
    if (wino_fused_ok(Cin, Cout)) {
        // one kernel, no planes (conv_wino_fused.hip); a caller that still asks for the transformed input gets it from the plane
        // form's input transform (the fused weight gradient does not need it)
        if (v_keep) {
            const WinoGeom gk = wino_geom(B, H, W, Cout);
            const int gk_in = (int)((gk.T * (Cin / 4) + 255) / 256);
            switch (pro) {
                case 0: hipLaunchKernelGGL(wino_input_kernel<0>, dim3(gk_in), dim3(256), 0, st, x, s, t, v_keep, B, H, W, Cin, gk.th, gk.tw, gk.T, gk.T); break;
                case 1: hipLaunchKernelGGL(wino_input_kernel<1>, dim3(gk_in), dim3(256), 0, st, x, s, t, v_keep, B, H, W, Cin, gk.th, gk.tw, gk.T, gk.T); break;
                case 2: hipLaunchKernelGGL(wino_input_kernel<2>, dim3(gk_in), dim3(256), 0, st, x, s, t, v_keep, B, H, W, Cin, gk.th, gk.tw, gk.T, gk.T); break;
                default: hipLaunchKernelGGL(wino_input_kernel<3>, dim3(gk_in), dim3(256), 0, st, x, s, t, v_keep, B, H, W, Cin, gk.th, gk.tw, gk.T, gk.T); break;
            }
        }
        return wino_fused_run(x, U, pro, s, t, y, stats, epi, B, H, W, Cin, Cout, st);
    }
    const WinoGeom g = wino_geom(B, H, W, Cout);
    // v_keep: the transformed input goes to a buffer of the caller's (16 T Cin floats) that outlives the call -- the weight
    // gradient of the same convolution multiplies the same planes (tag_conv3x3_wino_wgrad, v_saved); ws then holds the products only
    float* V = v_keep ? v_keep : ws;
    float* Mb = v_keep ? ws : ws + (size_t)16 * g.T * Cin;
    const long items = g.T * (Cin / 4);
    const int gin = (int)((items + 255) / 256);
    switch (pro) {
        case 0: hipLaunchKernelGGL(wino_input_kernel<0>, dim3(gin), dim3(256), 0, st, x, s, t, V, B, H, W, Cin, g.th, g.tw, g.T, g.T); break;
        case 1: hipLaunchKernelGGL(wino_input_kernel<1>, dim3(gin), dim3(256), 0, st, x, s, t, V, B, H, W, Cin, g.th, g.tw, g.T, g.T); break;
        case 2: hipLaunchKernelGGL(wino_input_kernel<2>, dim3(gin), dim3(256), 0, st, x, s, t, V, B, H, W, Cin, g.th, g.tw, g.T, g.T); break;
        default: hipLaunchKernelGGL(wino_input_kernel<3>, dim3(gin), dim3(256), 0, st, x, s, t, V, B, H, W, Cin, g.th, g.tw, g.T, g.T); break;
    }
    tag_launch_gemm_batched(V, Cin, g.T * Cin, U, Cout, (long)Cin * Cout, Mb, Cout, g.T * Cout, (int)g.T, Cout, Cin, 16, st, 0);
    const int gout = g.P / g.G;
    const WinoEpi none{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.0f, 0.0f, 0, 0, 0.0f, 0ull, 0};
    if (epi && epi->kind == 2)
        hipLaunchKernelGGL(wino_output_kernel<2>, dim3(gout), dim3(256), 0, st, Mb, y, stats, *epi, B, H, W, Cout, g.th, g.tw, g.T, g.P);
    else if (epi && epi->kind == 3)
        hipLaunchKernelGGL(wino_output_kernel<3>, dim3(gout), dim3(256), 0, st, Mb, y, stats, *epi, B, H, W, Cout, g.th, g.tw, g.T, g.P);
    else if (epi)
        hipLaunchKernelGGL(wino_output_kernel<1>, dim3(gout), dim3(256), 0, st, Mb, y, stats, *epi, B, H, W, Cout, g.th, g.tw, g.T, g.P);
    else
        hipLaunchKernelGGL(wino_output_kernel<0>, dim3(gout), dim3(256), 0, st, Mb, y, stats, none, B, H, W, Cout, g.th, g.tw, g.T, g.P);
    return 0;
}

}  // namespace

extern "C" int tag_conv3x3_wino_ok(int B, int H, int W, int Cin, int Cout) {
    // fused kernels (forward, dgrad and weight gradient all take the layer): channel counts that are multiples of 64; else the plane form
    const bool fused = Cin % 64 == 0 && Cout % 64 == 0 && Cin >= 64 && Cout >= 64 && Cin <= 1024 && Cout <= 1024;
    if (!(B > 0 && H > 0 && W > 0 && (fused || (wino_channels_ok(Cin) && wino_channels_ok(Cout) && Cin % 32 == 0)))) return 0;
    const WinoGeom g = wino_geom(B, H, W, Cout);
    // (the fused kernels address a tensor through a buffer descriptor: < 2^31 bytes; larger launches are cut by the caller)
    return g.T < (1L << 31) / 16 && (long)B * H * W < (1L << 31) && (long)B * H * W * (Cin > Cout ? Cin : Cout) < (1L << 29);
}

extern "C" int tag_pack_conv_weight_wino(const float* w, float* ufwd, float* udgrad, int Cin, int Cout, void* stream) {
    TAG_CHECK_ARG(w && (ufwd || udgrad) && Cin > 0 && Cout > 0);
    const long n = (long)Cin * Cout;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, ufwd,
                       udgrad, Cin, Cout, wino_fused_ok(Cin, Cout) ? 1 : 0, wino_fused_ok(Cout, Cin) ? 1 : 0);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t tag_conv3x3_wino_ws_bytes(int B, int H, int W, int Cin, int Cout) {
    if (wino_fused_ok(Cin > 0 ? Cin : 32, Cout)) return 64;          // the fused kernel has no planes
    const WinoGeom g = wino_geom(B, H, W, Cout);
    return (size_t)16 * g.T * ((size_t)Cin + Cout) * sizeof(float);
}

extern "C" int tag_conv3x3_wino_stats_rows(int B, int H, int W, int Cout) {
    if (wino_fused_ok(32, Cout)) return wino_fused_rows(B, H, W);
    if (!wino_channels_ok(Cout)) return 0;      // (every Cin tag_conv3x3_wino_ok accepts is a multiple of 32)
    return wino_geom(B, H, W, Cout).P;
}

// dgrad of a block's FIRST conv + the reduction half of the BatchNorm+ReLU+pool+dropout backward of the block BELOW it: the Winograd
// twin of tag_conv3x3_dgrad_poolsums (same arguments + ws; bnpart rows [P][2][Cout], P = tag_conv3x3_wino_stats_rows).
extern "C" int tag_conv3x3_wino_dgrad_poolsums(const float* dy, const float* u, float* dx, const float* yref, const float* bn_scale,
                                               const float* bn_shift, const float* bn_mean, const float* bn_invstd, float* bnpart,
                                               int B, int H, int W, int Cin, int Cout, int Hf, int Wf, int ph, int pw, int pool,
                                               float drop_p, uint64_t seed, void* ws, void* stream) {
    TAG_CHECK_ARG(dy && u && dx && yref && bn_scale && bn_shift && bn_mean && bn_invstd && bnpart && ws);
    TAG_CHECK_ARG(tag_conv3x3_wino_ok(B, H, W, Cin, Cout) && (long)B * Hf * Wf < (1L << 31));
    TAG_CHECK_ARG(pw == 2 && (ph == 1 || ph == 2) && H == Hf / ph && W == Wf / pw);
    TAG_CHECK_ARG((pool == 0 || pool == 2 || pool == 3) && drop_p >= 0.0f && drop_p < 1.0f);
    const float wavg = pool == 3 ? 0.0f : 1.0f / (float)(ph * pw), wmax = pool == 2 ? 0.0f : 1.0f;
    const WinoEpi epi{yref, bn_scale, bn_shift, bn_mean, bn_invstd, ph, wavg, wmax, Hf, Wf, drop_p, (unsigned long long)seed, 2};
    wino_run(dy, u, 0, nullptr, nullptr, dx, bnpart, &epi, B, H, W, Cin, Cout, static_cast<float*>(ws), as_stream(stream));
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t tag_conv3x3_wino_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout) {
    if (wino_fused_wgrad_ok(Cin, Cout)) return wino_fused_wgrad_ws_floats(B, H, W, Cin, Cout) * sizeof(float);
    const WinoWgGeom g = wino_wg_geom(B, H, W, Cin, Cout);
    return ((size_t)16 * g.Tpad * ((size_t)Cin + Cout) + (size_t)16 * g.S * Cin * Cout) * sizeof(float);
}

extern "C" int tag_conv3x3_wino_wgrad_can_reuse_v(int B, int H, int W, int Cin, int Cout) {
    if (wino_fused_wgrad_ok(Cin, Cout)) return 0;        // the fused weight gradient transforms its operands at staging: no planes
    const WinoWgGeom g = wino_wg_geom(B, H, W, Cin, Cout);
    return g.Tpad == g.T;                      // the K slices need no zero rows: the forward's planes are the operand as they are
}

extern "C" int tag_conv3x3_wino_wgrad(const float* x, int prologue, const float* in_scale, const float* in_shift, const float* dy,
                                      float* dw, int B, int H, int W, int Cin, int Cout, void* ws, const float* v_saved,
                                      void* stream) {
    TAG_CHECK_ARG((x || v_saved) && dy && dw && ws && tag_conv3x3_wino_ok(B, H, W, Cin, Cout));
    TAG_CHECK_ARG(v_saved == nullptr || tag_conv3x3_wino_wgrad_can_reuse_v(B, H, W, Cin, Cout));
    TAG_CHECK_ARG(prologue >= 0 && prologue <= 3 && (prologue == 0 || (in_scale && in_shift)));
    hipStream_t st = as_stream(stream);
    if (wino_fused_wgrad_ok(Cin, Cout) && x) {
        wino_fused_wgrad_run(x, prologue, in_scale, in_shift, dy, dw, B, H, W, Cin, Cout, static_cast<float*>(ws), st);
        TAG_LAUNCH_CHECK();
        return 0;
    }
    const WinoWgGeom g = wino_wg_geom(B, H, W, Cin, Cout);
    // v_saved: B^T prologue(x) B as the forward launch of this convolution left it (tag_conv3x3_wino_forward, v_keep): the input
    // transform (0.12-0.27 ms per layer at B = 64) is not repeated; ws then starts with the gradient planes
    float* D = v_saved ? static_cast<float*>(ws) : static_cast<float*>(ws) + (size_t)16 * g.Tpad * Cin;
    const float* V = v_saved ? v_saved : static_cast<const float*>(ws);
    float* Vw = static_cast<float*>(ws);
    float* Pp = D + (size_t)16 * g.Tpad * Cout;
    const int gin = (int)((g.Tpad * (Cin / 4) + 255) / 256), gdy = (int)((g.Tpad * (Cout / 4) + 255) / 256);
    if (!v_saved) switch (prologue) {
        case 0: hipLaunchKernelGGL(wino_input_kernel<0>, dim3(gin), dim3(256), 0, st, x, in_scale, in_shift, Vw, B, H, W, Cin, g.th, g.tw, g.T, g.Tpad); break;
        case 1: hipLaunchKernelGGL(wino_input_kernel<1>, dim3(gin), dim3(256), 0, st, x, in_scale, in_shift, Vw, B, H, W, Cin, g.th, g.tw, g.T, g.Tpad); break;
        case 2: hipLaunchKernelGGL(wino_input_kernel<2>, dim3(gin), dim3(256), 0, st, x, in_scale, in_shift, Vw, B, H, W, Cin, g.th, g.tw, g.T, g.Tpad); break;
        default: hipLaunchKernelGGL(wino_input_kernel<3>, dim3(gin), dim3(256), 0, st, x, in_scale, in_shift, Vw, B, H, W, Cin, g.th, g.tw, g.T, g.Tpad); break;
    }
    hipLaunchKernelGGL(wino_dy_kernel, dim3(gdy), dim3(256), 0, st, dy, D, B, H, W, Cout, g.th, g.tw, g.T, g.Tpad);
    // batch entry (xi, s): Pp[xi][s] (Cout x Cin) = D[xi][s kc .. (s+1) kc)^T . V[xi][the same rows]; the planes are contiguous, so
    // consecutive entries are kc rows apart in both operands
    tag_launch_gemm_batched(D, Cout, (long)g.kc * Cout, V, Cin, (long)g.kc * Cin, Pp, Cin, (long)Cin * Cout, Cout, Cin, g.kc,
                            16 * g.S, st, 1);
    const long n = (long)Cin * Cout;
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256)), dim3(256), 0, st, Pp, g.S, Cin, Cout,
                       dw);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_conv3x3_wino_forward(const float* x, const float* u, int prologue, const float* in_scale,
                                        const float* in_shift, float* y, float* stats, int B, int H, int W, int Cin, int Cout,
                                        void* ws, float* v_keep, void* stream) {
    TAG_CHECK_ARG(x && u && y && ws && tag_conv3x3_wino_ok(B, H, W, Cin, Cout));
    TAG_CHECK_ARG(prologue >= 0 && prologue <= 3 && (prologue == 0 || (in_scale && in_shift)));
    wino_run(x, u, prologue, in_scale, in_shift, y, stats, nullptr, B, H, W, Cin, Cout, static_cast<float*>(ws), as_stream(stream),
             v_keep);
    TAG_LAUNCH_CHECK();
    return 0;
}

// Inference forward of a conv + the rest of its ConvBlock stage with BatchNorm in eval mode (models/panns.py:49-60): the Winograd
// twin of tag_conv3x3_forward_bnrelu_pool_eval -- out (B, H/ph, W/2, Cout) = pool(relu(conv(prologue(x)) * bn_scale + bn_shift)).
extern "C" int tag_conv3x3_wino_forward_bnrelu_pool_eval(const float* x, const float* u, int prologue, const float* in_scale,
                                                         const float* in_shift, float* out, const float* bn_scale,
                                                         const float* bn_shift, int B, int H, int W, int Cin, int Cout, int ph, int pw,
                                                         int pool, void* ws, void* stream) {
    TAG_CHECK_ARG(x && u && out && bn_scale && bn_shift && ws && tag_conv3x3_wino_ok(B, H, W, Cin, Cout));
    TAG_CHECK_ARG(prologue >= 0 && prologue <= 3 && (prologue == 0 || (in_scale && in_shift)));
    TAG_CHECK_ARG(pw == 2 && (ph == 1 || ph == 2) && H / ph > 0 && W >= 2 && (pool == 0 || pool == 2 || pool == 3));
    const float wavg = pool == 3 ? 0.0f : 1.0f / (float)(ph * pw), wmax = pool == 2 ? 0.0f : 1.0f;
    const WinoEpi epi{nullptr, bn_scale, bn_shift, nullptr, nullptr, ph, wavg, wmax, 0, 0, 0.0f, 0ull, 3};
    wino_run(x, u, prologue, in_scale, in_shift, out, nullptr, &epi, B, H, W, Cin, Cout, static_cast<float*>(ws), as_stream(stream));
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_conv3x3_wino_dgrad_bnsums(const float* dy, const float* u, float* da, const float* yref, const float* bn_scale,
                                             const float* bn_shift, const float* bn_mean, const float* bn_invstd, float* bnpart,
                                             int B, int H, int W, int Cin, int Cout, void* ws, void* stream) {
    TAG_CHECK_ARG(dy && u && da && yref && bn_scale && bn_shift && bn_mean && bn_invstd && bnpart && ws);
    TAG_CHECK_ARG(tag_conv3x3_wino_ok(B, H, W, Cin, Cout));
    const WinoEpi epi{yref, bn_scale, bn_shift, bn_mean, bn_invstd, 0, 0.0f, 0.0f, 0, 0, 0.0f, 0ull, 1};
    wino_run(dy, u, 0, nullptr, nullptr, da, bnpart, &epi, B, H, W, Cin, Cout, static_cast<float*>(ws), as_stream(stream));
    TAG_LAUNCH_CHECK();
    return 0;
}
