// A4: bidirectional GRU recurrence, PyTorch gate order (r, z, n), h0 = 0, all T steps incl. padding
// (nn.GRU(512, 256, bidirectional=True, batch_first=True), models/audio_encoder.py:141,217;
//  CrnnEncoder nn.GRU(128,128) :58-63,75).
//
// The input projections (x W_ih^T + b_ih, both directions) are one MFMA GEMM done by the caller;
// this file is the serial part: per time step, gh = h_{t-1} W_hh^T + b_hh and the gate arithmetic,
// for both directions at once.  One launch per step (the recurrence is latency-bound: 250 dependent
// steps), each launch tiled (16 batch rows) x (16 hidden units) x direction with the K dimension
// split over the 4 waves of a workgroup on v_mfma_f32_16x16x4_f32 and reduced through LDS, so a
// step is ~130 workgroups of a few hundred cycles.  All launches of a sequence are issued
// back-to-back from one C call on the caller's stream.
#include "tag_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// wt[dir][g][k][j] = w_hh[dir][g*H + j][k]
__global__ __launch_bounds__(256) void gru_transpose_whh_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                int H) {
    const long n = (long)2 * 3 * H * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int j = (int)(i % H);
        long r = i / H;
        const int k = (int)(r % H);
        const int dg = (int)(r / H);   // dir*3 + g
        wt[i] = w[((size_t)dg * H + j) * H + k];
    }
}

// grid (H/16, ceil(B/16), 2).  HT > 0: compile-time hidden size -> the whole K slice of a wave is loaded up
// front (all loads in flight at once) before the MFMA chain; HT == 0: runtime loop.
template <int HT>
__global__ __launch_bounds__(256) void gru_fwd_step_kernel(const float* __restrict__ gi, const float* __restrict__ wt,
                                                           const float* __restrict__ b_hh, float* __restrict__ y,
                                                           float* __restrict__ gates, int B, int T, int Hrt, int step) {
    const int H = HT > 0 ? HT : Hrt;
    __shared__ float red[4][3][256];
    const int dir = blockIdx.z;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    if (step > 0) {
        const int kq = H / 4;                         // K range of this wave
        const int brow = b0 + li;
        const float* hp = y + (((size_t)(brow < B ? brow : 0) * T + tp) * 2 + dir) * H;
        const float* wbase = wt + (size_t)dir * 3 * H * H + j0 + li;
        if constexpr (HT > 0) {
            constexpr int NI = HT / 16;
            float av[NI], wv[3][NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int k = wid * kq + lk + 4 * i;
                av[i] = brow < B ? hp[k] : 0.0f;
#pragma unroll
                for (int g = 0; g < 3; ++g) wv[g][i] = wbase[((size_t)g * H + k) * H];
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wv[g][i], acc[g], 0, 0, 0);
        } else {
            for (int k = wid * kq + lk; k < (wid + 1) * kq; k += 4) {
                const float a = brow < B ? hp[k] : 0.0f;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float bw = wbase[((size_t)g * H + k) * H];
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[g], 0, 0, 0);
                }
            }
        }
    }
    // D layout: col = lane & 15 (unit), row = (lane >> 4) * 4 + reg (batch row)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][g][(lk * 4 + r) * 16 + li] = acc[g][r];
    __syncthreads();
    const int e = threadIdx.x;                        // element (row, unit) of the 16x16 tile
    const int row = e >> 4, col = e & 15;
    const int b = b0 + row, j = j0 + col;
    if (b >= B) return;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        gh[g] = red[0][g][e] + red[1][g][e] + red[2][g][e] + red[3][g][e] + b_hh[(size_t)dir * 3 * H + g * H + j];
    const float* gix = gi + (((size_t)b * T + t) * 2 + dir) * 3 * H;
    const float r = sigmoidf_(gix[j] + gh[0]);
    const float z = sigmoidf_(gix[H + j] + gh[1]);
    const float n = tanhf(gix[2 * H + j] + r * gh[2]);
    const float hprev = step > 0 ? y[(((size_t)b * T + tp) * 2 + dir) * H + j] : 0.0f;
    const float h = (1.0f - z) * n + z * hprev;
    y[(((size_t)b * T + t) * 2 + dir) * H + j] = h;
    if (gates) {
        float* gs = gates + (((size_t)b * T + t) * 2 + dir) * 4 * H;
        gs[j] = r; gs[H + j] = z; gs[2 * H + j] = n; gs[3 * H + j] = gh[2];
    }
}

// backward step: see header.  dhbuf (2, B, H) holds dh of the step processed just before.
template <int HT>
__global__ __launch_bounds__(256) void gru_bwd_step_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ gates,
                                                           const float* __restrict__ w_hh, float* __restrict__ dgi,
                                                           float* __restrict__ dgh, float* __restrict__ hprev_out,
                                                           float* __restrict__ dhbuf, int B, int T, int Hrt, int step) {
    const int H = HT > 0 ? HT : Hrt;
    __shared__ float red[4][256];
    const int dir = blockIdx.z;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    // backward visits time in the reverse of the forward order
    const int t = dir == 0 ? T - 1 - step : step;
    const int tp = dir == 0 ? t - 1 : t + 1;       // forward-order predecessor (h_{t-1})
    const int tn = dir == 0 ? t + 1 : t - 1;       // forward-order successor (processed one step ago)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (step > 0) {
        const int K = 3 * H, kq = K / 4;
        const int brow = b0 + li;
        const float* ap = dgh + (((size_t)(brow < B ? brow : 0) * T + tn) * 2 + dir) * K;
        const float* wbase = w_hh + (size_t)dir * K * H + j0 + li;
        if constexpr (HT > 0) {
            constexpr int NI = 3 * HT / 16;
            float av[NI], wv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int k = wid * kq + lk + 4 * i;
                av[i] = brow < B ? ap[k] : 0.0f;
                wv[i] = wbase[(size_t)k * H];
            }
#pragma unroll
            for (int i = 0; i < NI; i += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wv[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1], wv[i + 1], acc1, 0, 0, 0);
            }
        } else {
            for (int k = wid * kq + lk; k < (wid + 1) * kq; k += 8) {
                const float a0 = brow < B ? ap[k] : 0.0f;
                const float a1 = brow < B ? ap[k + 4] : 0.0f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wbase[(size_t)k * H], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wbase[(size_t)(k + 4) * H], acc1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][(lk * 4 + r) * 16 + li] = acc0[r] + acc1[r];
    __syncthreads();
    const int e = threadIdx.x;
    const int row = e >> 4, col = e & 15;
    const int b = b0 + row, j = j0 + col;
    if (b >= B) return;
    const size_t cell = ((size_t)b * T + t) * 2 + dir;
    float dh = dy[cell * H + j];
    if (step > 0) {
        const size_t ncell = ((size_t)b * T + tn) * 2 + dir;
        const float zn = gates[ncell * 4 * H + H + j];
        dh += red[0][e] + red[1][e] + red[2][e] + red[3][e] + dhbuf[((size_t)dir * B + b) * H + j] * zn;
    }
    dhbuf[((size_t)dir * B + b) * H + j] = dh;
    const float* gs = gates + cell * 4 * H;
    const float r = gs[j], z = gs[H + j], n = gs[2 * H + j], ghn = gs[3 * H + j];
    const bool has_prev = dir == 0 ? (t > 0) : (t < T - 1);
    const float hp = has_prev ? y[(((size_t)b * T + tp) * 2 + dir) * H + j] : 0.0f;
    const float dn = dh * (1.0f - z);
    const float dz = dh * (hp - n);
    const float dn_pre = dn * (1.0f - n * n);
    const float dz_pre = dz * z * (1.0f - z);
    const float dr_pre = dn_pre * ghn * r * (1.0f - r);
    float* gi_o = dgi + cell * 3 * H;
    float* gh_o = dgh + cell * 3 * H;
    gi_o[j] = dr_pre; gi_o[H + j] = dz_pre; gi_o[2 * H + j] = dn_pre;
    gh_o[j] = dr_pre; gh_o[H + j] = dz_pre; gh_o[2 * H + j] = dn_pre * r;
    hprev_out[cell * H + j] = hp;
}

}  // namespace

// ws: 2*3*H*H floats (transposed recurrent weights)
extern "C" int tag_gru_forward(const float* gi, const float* w_hh, const float* b_hh, float* y, float* gates,
                               float* ws, int B, int T, int H, void* stream) {
    TAG_CHECK_ARG(gi && w_hh && b_hh && y && ws && B > 0 && T > 0);
    TAG_CHECK_ARG(H % 16 == 0 && H >= 16);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(gru_transpose_whh_kernel, dim3(cdiv((long)6 * H * H, 256)), dim3(256), 0, st, w_hh, ws, H);
    TAG_LAUNCH_CHECK();
    const dim3 grid(H / 16, (B + 15) / 16, 2);
    for (int s = 0; s < T; ++s) {
        if (H == 256)
            hipLaunchKernelGGL(gru_fwd_step_kernel<256>, grid, dim3(256), 0, st, gi, ws, b_hh, y, gates, B, T, H, s);
        else if (H == 128)
            hipLaunchKernelGGL(gru_fwd_step_kernel<128>, grid, dim3(256), 0, st, gi, ws, b_hh, y, gates, B, T, H, s);
        else
            hipLaunchKernelGGL(gru_fwd_step_kernel<0>, grid, dim3(256), 0, st, gi, ws, b_hh, y, gates, B, T, H, s);
    }
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_gru_backward(const float* dy, const float* y, const float* gates, const float* w_hh, float* dgi,
                                float* dgh, float* hprev, float* scratch, int B, int T, int H, void* stream) {
    TAG_CHECK_ARG(dy && y && gates && w_hh && dgi && dgh && hprev && scratch && B > 0 && T > 0);
    TAG_CHECK_ARG(H % 16 == 0 && (3 * H) % 32 == 0);
    hipStream_t st = as_stream(stream);
    const dim3 grid(H / 16, (B + 15) / 16, 2);
    for (int s = 0; s < T; ++s) {
        if (H == 256)
            hipLaunchKernelGGL(gru_bwd_step_kernel<256>, grid, dim3(256), 0, st, dy, y, gates, w_hh, dgi, dgh, hprev,
                               scratch, B, T, H, s);
        else if (H == 128)
            hipLaunchKernelGGL(gru_bwd_step_kernel<128>, grid, dim3(256), 0, st, dy, y, gates, w_hh, dgi, dgh, hprev,
                               scratch, B, T, H, s);
        else
            hipLaunchKernelGGL(gru_bwd_step_kernel<0>, grid, dim3(256), 0, st, dy, y, gates, w_hh, dgi, dgh, hprev,
                               scratch, B, T, H, s);
    }
    TAG_LAUNCH_CHECK();
    return 0;
}
