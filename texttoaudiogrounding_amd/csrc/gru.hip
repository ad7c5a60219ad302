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
#include <stdlib.h>
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

// ------------------------------------------------------------------------------------------------------------
// Persistent variants: ONE launch for the whole sequence.  Workgroup (unit tile, batch tile, direction) keeps its
// slice of the recurrent weights in registers for all T steps; the 16 workgroups that share a batch tile exchange
// the new hidden state (backward: the new gate gradients) through 8-byte {epoch tag, value} granules written and
// read with agent-scope relaxed atomics (sc1: coherent across the 8 XCD L2s; the data IS the flag, so there is no
// fence and no separate counter -- cdna_hip_programming.md Guideline 16, form R2).  Two parities: a workgroup can
// be at most one step ahead of the slowest member of its group.  Every spin is bounded; a timeout raises *err.
// Residency: the grid is (H/16) x ceil(B/16) x 2 <= 256 one-per-CU workgroups (checked by the launcher).
// ------------------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
#define TAG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned GRU_SPIN_LIMIT = 1u << 21;

__device__ __forceinline__ u64 granule(unsigned epoch, float v) {
    return ((u64)epoch << 32) | (u64)__float_as_uint(v);
}

template <int HT>
__global__ __launch_bounds__(256) void gru_fwd_persistent_kernel(const float* __restrict__ gi,
                                                                 const float* __restrict__ wt,
                                                                 const float* __restrict__ b_hh, float* __restrict__ y,
                                                                 float* __restrict__ gates, u64* hx, unsigned* err,
                                                                 int B, int T) {
    constexpr int H = HT, NI = HT / 16, KQ = HT / 4;
    __shared__ float red[4][3][256];
    const int dir = blockIdx.z, j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int Bpad = gridDim.y * 16;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float wv[3][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int g = 0; g < 3; ++g)
            wv[g][i] = wt[(((size_t)dir * 3 + g) * H + wid * KQ + lk + 4 * i) * H + j0 + li];
    const int e = threadIdx.x, row = e >> 4, col = e & 15;
    const int b = b0 + row, j = j0 + col;
    const bool valid = b < B;
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = b_hh[(size_t)dir * 3 * H + g * H + j];
    u64* hxd = hx + (size_t)dir * 2 * Bpad * H;
    const bool arow = b0 + li < B;                    // rows >= B are never produced: do not wait for them
    float hprev = 0.0f;
    bool dead = false;
    const int t0 = dir == 0 ? 0 : T - 1;
    float gir = 0, giz = 0, gin = 0;
    if (valid) {
        const float* gx = gi + (((size_t)b * T + t0) * 2 + dir) * 3 * H;
        gir = gx[j]; giz = gx[H + j]; gin = gx[2 * H + j];
    }
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (s > 0) {
            const u64* src = hxd + (size_t)((s - 1) & 1) * Bpad * H + (size_t)(b0 + li) * H + wid * KQ + lk;
            float av[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) av[i] = 0.0f;
            unsigned spins = dead ? GRU_SPIN_LIMIT : 0;   // after one timeout never wait again (bounded total time)
            // granules still missing (bit i): a re-poll asks only for those -- the first sweep usually finds most producers
            // done, and re-reading everything (32 KB per workgroup per sweep, 128 workgroups) only queues behind the very
            // stores it is waiting for
            unsigned pend = arow ? ((NI >= 32) ? 0xffffffffu : ((1u << NI) - 1u)) : 0u;
            while (true) {
                u64 gq[NI];
                const unsigned want = pend;                // all loads of the sweep are issued before the first is looked at
#pragma unroll
                for (int i = 0; i < NI; ++i) gq[i] = (want & (1u << i)) ? __hip_atomic_load(src + 4 * i, TAG_RLX_AGENT) : 0ull;
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if ((want & (1u << i)) && (unsigned)(gq[i] >> 32) == (unsigned)s) {
                        av[i] = __uint_as_float((unsigned)gq[i]);
                        pend &= ~(1u << i);
                    }
                if (__all(pend == 0u)) break;
                if (++spins > GRU_SPIN_LIMIT) { if (lane == 0) atomicExch(err, 1u); dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wv[g][i], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wid][g][(lk * 4 + r) * 16 + li] = acc[g][r];
        __syncthreads();
        // prefetch the next step's input projections while the gates are computed
        float ngir = 0, ngiz = 0, ngin = 0;
        if (valid && s + 1 < T) {
            const int tn = dir == 0 ? t + 1 : t - 1;
            const float* gx = gi + (((size_t)b * T + tn) * 2 + dir) * 3 * H;
            ngir = gx[j]; ngiz = gx[H + j]; ngin = gx[2 * H + j];
        }
        if (valid) {
            float gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) gh[g] = red[0][g][e] + red[1][g][e] + red[2][g][e] + red[3][g][e] + bias[g];
            const float r = sigmoidf_(gir + gh[0]);
            const float z = sigmoidf_(giz + gh[1]);
            const float n = tanhf(gin + r * gh[2]);
            float h = (1.0f - z) * n + z * hprev;
            if (dead) h = __int_as_float(0x7fc00000);   // a timed-out exchange must not pass silently: NaN reaches the loss
            hprev = h;
            __hip_atomic_store(hxd + (size_t)(s & 1) * Bpad * H + (size_t)b * H + j, granule((unsigned)(s + 1), h),
                               TAG_RLX_AGENT);
            y[(((size_t)b * T + t) * 2 + dir) * H + j] = h;
            if (gates) {
                float* gs = gates + (((size_t)b * T + t) * 2 + dir) * 4 * H;
                gs[j] = r; gs[H + j] = z; gs[2 * H + j] = n; gs[3 * H + j] = gh[2];
            }
        }
        gir = ngir; giz = ngiz; gin = ngin;
        __syncthreads();                               // red[] is rewritten by the next step
    }
}

// Backward, K-partitioned: a workgroup multiplies ITS OWN 48 new gate-gradient columns (kept in LDS, never
// exchanged) with the matching 48 rows of W_hh for ALL H outputs and publishes the 16 x H partial products; the
// consumer of unit tile u' adds the 16 producers' partials of its 16 units in a fixed order.  Each thread polls
// exactly 16 granules per step (the per-thread (row, unit) element from 16 producers) instead of every lane
// sweeping 3H/16 of them.
template <int HT>
__global__ __launch_bounds__(256) void gru_bwd_persistent_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ gates,
                                                                 const float* __restrict__ w_hh, float* __restrict__ dgi,
                                                                 float* __restrict__ dgh, float* __restrict__ hprev_out,
                                                                 u64* gxch, unsigned* err, int B, int T) {
    constexpr int H = HT, NU = HT / 16;              // NU producers per batch tile
    constexpr int NT = HT / 64;                      // 16-wide output tiles per wave (4 waves cover H)
    __shared__ float At[16][48 + 1];                 // this step's dgh tile: [row][gate*16 + unit]
    const int dir = blockIdx.z, u = blockIdx.x, j0 = u * 16, bt = blockIdx.y, b0 = bt * 16;
    const int nbt = gridDim.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    // W_hh rows of the own 48 gate columns, all H outputs: wave w owns outputs [64w, 64w+64)
    float wv[NT][12];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            const int kk = 4 * ks + lk;              // 0..47 = gate*16 + unit
            wv[nt][ks] = w_hh[((size_t)dir * 3 * H + (kk >> 4) * H + j0 + (kk & 15)) * H + wid * (H / 4) + nt * 16 + li];
        }
    const int e = threadIdx.x, row = e >> 4, col = e & 15;
    const int b = b0 + row, j = j0 + col;
    const bool valid = b < B;
    // exchange layout: [parity][dir][bt][producer][row][H]
    const size_t slab = (size_t)16 * H;
    u64* xbase = gxch + ((size_t)dir * nbt + bt) * NU * slab;
    const size_t parity_stride = (size_t)2 * nbt * NU * slab;
    float dh_carry = 0.0f, z_next = 0.0f;
    bool dead = false;
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? T - 1 - s : s;
        const int tp = dir == 0 ? t - 1 : t + 1;
        float g_r = 0, g_z = 0, g_n = 0, g_hn = 0, dyv = 0, hp = 0;
        const size_t cell = ((size_t)(valid ? b : 0) * T + t) * 2 + dir;
        if (valid) {
            const float* gs = gates + cell * 4 * H;
            g_r = gs[j]; g_z = gs[H + j]; g_n = gs[2 * H + j]; g_hn = gs[3 * H + j];
            dyv = dy[cell * H + j];
            const bool has_prev = dir == 0 ? (t > 0) : (t < T - 1);
            hp = has_prev ? y[(((size_t)b * T + tp) * 2 + dir) * H + j] : 0.0f;
        }
        float msum = 0.0f;
        if (s > 0) {
            const u64* src = xbase + (size_t)((s - 1) & 1) * parity_stride + (size_t)row * H + j;
            float pv[NU];
#pragma unroll
            for (int p = 0; p < NU; ++p) pv[p] = 0.0f;
            unsigned spins = dead ? GRU_SPIN_LIMIT : 0;
            unsigned pend = valid ? ((NU >= 32) ? 0xffffffffu : ((1u << NU) - 1u)) : 0u;   // producers still missing
            while (true) {
                u64 gq[NU];
                const unsigned want = pend;
#pragma unroll
                for (int p = 0; p < NU; ++p)
                    gq[p] = (want & (1u << p)) ? __hip_atomic_load(src + (size_t)p * slab, TAG_RLX_AGENT) : 0ull;
#pragma unroll
                for (int p = 0; p < NU; ++p)
                    if ((want & (1u << p)) && (unsigned)(gq[p] >> 32) == (unsigned)s) {
                        pv[p] = __uint_as_float((unsigned)gq[p]);
                        pend &= ~(1u << p);
                    }
                if (__all(pend == 0u)) break;
                if (++spins > GRU_SPIN_LIMIT) { if (lane == 0) atomicExch(err, 1u); dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int p = 0; p < NU; ++p) msum += pv[p];
        }
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
        if (valid) {
            float dh = dyv;
            if (s > 0) dh += msum + dh_carry * z_next;
            if (dead) dh = __int_as_float(0x7fc00000);  // timed-out exchange: poison the gradients (loud, not silent)
            dh_carry = dh;
            z_next = g_z;
            const float dn = dh * (1.0f - g_z);
            const float dz = dh * (hp - g_n);
            const float dn_pre = dn * (1.0f - g_n * g_n);
            const float dz_pre = dz * g_z * (1.0f - g_z);
            const float dr_pre = dn_pre * g_hn * g_r * (1.0f - g_r);
            const float dnr = dn_pre * g_r;
            v0 = dr_pre; v1 = dz_pre; v2 = dnr;
            float* gi_o = dgi + cell * 3 * H;
            float* gh_o = dgh + cell * 3 * H;
            gi_o[j] = dr_pre; gi_o[H + j] = dz_pre; gi_o[2 * H + j] = dn_pre;
            gh_o[j] = dr_pre; gh_o[H + j] = dz_pre; gh_o[2 * H + j] = dnr;
            hprev_out[cell * H + j] = hp;
        }
        At[row][col] = v0; At[row][16 + col] = v1; At[row][32 + col] = v2;
        __syncthreads();
        if (s + 1 < T) {                             // the last step's products are never consumed
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                const float a = At[li][4 * ks + lk];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wv[nt][ks], acc[nt], 0, 0, 0);
            }
            u64* dst = xbase + (size_t)(s & 1) * parity_stride + (size_t)u * slab;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(dst + (size_t)(lk * 4 + r) * H + wid * (H / 4) + nt * 16 + li,
                                       granule((unsigned)(s + 1), acc[nt][r]), TAG_RLX_AGENT);
        }
        __syncthreads();                             // At is rewritten by the next step
    }
}

// ------------------------------------------------------------------------------------------------------------
// 4-row persistent variants (the default): the step time of the recurrence is the latency of one exchange
// (publish -> visible -> polled) plus the arithmetic between two exchanges, and the exchange grows with the bytes
// a workgroup has to sweep.  A workgroup therefore owns only GR = 4 batch rows x GU = 32 hidden units of one
// direction: it sweeps 4 x H granules per step (8 KB at H = 256, a quarter of the 16-row tiles above), B = 64 /
// H = 256 gives 16 x 8 x 2 = 256 workgroups = one per CU, and the matrix work of a step halves.
// The arithmetic runs on v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4x4 outer products per instruction, exact
// fp32 fma chains -- which has NO padding at 4 rows (the 16x16x4 form would idle 12 of its 16 rows).  Layout
// (probed, tools/mfma4x4_probe.hip): D[lane l][reg r] = A[lane 4*(l/4) + r] * B[lane l], i.e. block = l / 4,
// A row = reg, B column = l % 4.
//
// Forward: gh[4 rows][96 cols = 3 gates x 32 units] = h[4][H] W^T.  Wave w owns k in [w H/4, (w+1) H/4); one MFMA
// covers 4 consecutive k (block / 4) x 4 column blocks (block % 4), six MFMAs (c = 0..5) cover the 24 column blocks,
// so lane l needs h[row l%4][k0 + (l/16) NI + i] and keeps 96 weights in registers; the sweep itself is coalesced
// (64 lanes x 8 B of one row per load) and dealt to the lanes through a per-wave LDS tile.
// The 16 partial sums per output (4 k-phases x 4 waves) meet in LDS (double-buffered: one barrier per step).
// Backward (K-partitioned like the 16-row kernel): the workgroup multiplies ITS 96 new gate-gradient columns with
// the matching 96 rows of W_hh for all H outputs (wave w: outputs [64w, 64w+64), one MFMA per k) and publishes the
// 4 x H partial products; the owner of a unit tile adds the H/32 producers' partials in a fixed order.
// ------------------------------------------------------------------------------------------------------------
constexpr int GR = 4, GU = 32, GC = 3 * GU;          // rows, units, gate columns per workgroup

// Placement of the 4-row grid.  The NU = H / 32 workgroups that exchange with each other (same batch tile, same
// direction) are given linear ids that the dispatcher's round-robin (workgroup id mod 8 -> XCD) puts on ONE XCD whenever
// the number of groups is a multiple of 8 (B = 64: 32 groups, 4 per XCD, 32 workgroups = that XCD's 32 CUs).  Placement
// is VERIFIED, not assumed: every workgroup publishes the XCC id the hardware reports for it (write-through) and reads
// its group's ids; only a group that really shares an XCD publishes its granules with L2-resident stores (sc0: the XCD's
// L2 is the point of coherence of its CUs; the polls are sc1 loads, which bypass the L1 and are served by that L2) --
// an exchange then costs an L2 round trip instead of a trip through the fabric.  Any other placement (another dispatch
// order, a group count that is not a multiple of 8) keeps the write-through (sc1) stores, which are visible chip-wide.
struct P4Coord { int u, bt, dir, grp; };
__device__ __forceinline__ P4Coord p4_decode(int id, int NU, int nbt) {
    const int ngroups = nbt * 2;
    int grp, m;
    if ((ngroups & 7) == 0) {
        const int xcd = id & 7, slot = id >> 3, gpx = ngroups >> 3;
        m = slot / gpx;
        grp = xcd * gpx + slot % gpx;
    } else {
        grp = id / NU;
        m = id % NU;
    }
    return P4Coord{m, grp % nbt, grp / nbt, grp};
}
// true when all NU members of the group report the same XCC id (and fast publishing is allowed)
__device__ __forceinline__ bool p4_group_shares_xcd(u64* xid, int grp, int m, int NU, int allow) {
    __shared__ int same_flag;
    const unsigned my = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;   // hwreg(HW_REG_XCC_ID, 0, 4)
    if (threadIdx.x == 0) {
        same_flag = 0;
        __hip_atomic_store(xid + (size_t)grp * NU + m, (0x7fffffffull << 32) | (u64)(my + 1u), TAG_RLX_AGENT);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        bool ok = true;
        if ((int)threadIdx.x < NU) {
            ok = false;
            for (unsigned spins = 0; spins < GRU_SPIN_LIMIT; ++spins) {
                const u64 q = __hip_atomic_load(xid + (size_t)grp * NU + threadIdx.x, TAG_RLX_AGENT);
                if ((unsigned)(q >> 32) == 0x7fffffffu) { ok = (unsigned)q == my + 1u; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const bool all_ok = __all(ok);
        if (threadIdx.x == 0) same_flag = (all_ok && allow) ? 1 : 0;
    }
    __syncthreads();
    return same_flag != 0;
}
#define TAG_RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

// Gate functions of the 4-row forward kernel.  The recurrence is a chain of T dependent steps run by ONE wave per SIMD, so
// the ~100 dependent instructions of the library expf / tanhf / full-precision divide are paid in full on every step
// (measured: 410 of 3200 clocks per step).  Hardware forms instead: v_exp_f32 + v_rcp_f32 (1 ulp each); tanh switches to
// its odd series below |x| = 0.2, where 1 - 2 / (1 + e^2x) would cancel.  |error| <= 3e-7 absolute / 4e-7 relative on
// outputs in (0,1) / (-1,1) -- fp32 rounding class (the library forms are within 1-2 ulp = 1.2e-7).
__device__ __forceinline__ float sigmoid_hw(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_hw(float x) {
    const float t = x * x;
    const float small = x * fmaf(t, fmaf(t, fmaf(t, -17.0f / 315.0f, 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
    return fabsf(x) < 0.2f ? small : big;
}
// sum of x over the lanes l ^ 8 (DPP row_ror:8 inside each row of 16 lanes)
__device__ __forceinline__ float add_ror8(float x) {
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x128, 0xf, 0xf, false));
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int HT>
__global__ __launch_bounds__(256) void gru_fwd_p4_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                         const float* __restrict__ b_hh, float* __restrict__ y,
                                                         float* __restrict__ gates, u64* hx, u64* xid, unsigned* err, int B, int T,
                                                         int nbt, int allow_fast) {
    // Wave w owns 8 of the workgroup's 32 units (all three gates) over the WHOLE k range, so that a step has one barrier
    // and no reduction through LDS: MFMA blocks = 2 column blocks (8 units) x 8 k phases, lane l = 8 phase + unit offset;
    // the 8 phase partials of an output meet through v_permlane32_swap / v_permlane16_swap / DPP row_ror:8 (21 VALU
    // operations), which also leaves the three gates of ONE (row, unit) in one lane: lane row (l / 16) -> batch row
    // {0,2,1,3}, unit = 8 w + l % 8 (lanes l and l ^ 8 hold the same sums; the lower one writes).
    constexpr int H = HT, NK = HT / 8;               // k values per phase = A registers per lane
    // h_{t-1} of the 4 rows, as the sweep collected it: [row][k phase][NK + 4] -- the 4-word pad makes the 8 phases of a
    // ds_read_b128 land on the 8 distinct 16-byte bank groups (unpadded, phase stride NK words = every lane on ONE group)
    constexpr int PS = HT / 8 + 4;
    __shared__ __attribute__((aligned(16))) float hs[2][GR][8 * PS];
    __shared__ int dead_flag;
    const P4Coord co = p4_decode(blockIdx.x, H / GU, nbt);
    const int dir = co.dir, j0 = co.u * GU, b0 = co.bt * GR;
    const int Bpad = nbt * GR;
    if (threadIdx.x == 0) dead_flag = 0;
    const bool fast = p4_group_shares_xcd(xid, co.grp, co.u, H / GU, allow_fast);     // (contains the barriers)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int arow_i = lane & 3, ph = lane >> 3;
    const int j = j0 + 8 * wid + (lane & 7);                          // the unit of this lane's MFMA column AND of its gate element
    // B operand of MFMA (i, g): W_hh[g H + j][k = ph NK + i]
    float wv[NK][3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float* wrow = w_hh + ((size_t)dir * 3 * H + (size_t)g * H + j) * H + ph * NK;
#pragma unroll
        for (int i4 = 0; i4 < NK / 4; ++i4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 4 * i4);
            wv[4 * i4][g] = v[0]; wv[4 * i4 + 1][g] = v[1]; wv[4 * i4 + 2][g] = v[2]; wv[4 * i4 + 3][g] = v[3];
        }
    }
    const int grow = ((lane >> 4) & 1) * 2 + (lane >> 5);             // lane row 0,1,2,3 -> batch row 0,2,1,3
    const int b = b0 + grow;
    const bool gthr = (lane & 8) == 0, valid = gthr && b < B;
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = b_hh[(size_t)dir * 3 * H + g * H + j];
    u64* hxd = hx + (size_t)dir * 2 * Bpad * H;
    float hprev = 0.0f;
    bool dead = false;
    // Input projections of a step are loaded ONE STEP AHEAD, right after the sweep of the step before (vmcnt is one in-order
    // counter: a load issued before the sweep would sit in front of every poll, a load consumed in its own step would stall
    // the gate arithmetic for an HBM round trip).  Two register sets, the loop unrolled by two: a rotating copy would make
    // the compiler wait for the loads at the copy.
    float ga[3] = {0.0f, 0.0f, 0.0f}, gb[3] = {0.0f, 0.0f, 0.0f};
    auto load_gi = [&](int s_, float (&dst)[3]) {
        if (valid && s_ < T) {
            const float* gx = gi + (((size_t)b * T + (dir == 0 ? s_ : T - 1 - s_)) * 2 + dir) * 3 * H;
            dst[0] = gx[j]; dst[1] = gx[H + j]; dst[2] = gx[2 * H + j];
        }
    };
    load_gi(0, ga);
#ifdef TAG_GRU_PROF
    u64 pc[5] = {0, 0, 0, 0, 0}, p0 = __builtin_amdgcn_s_memtime();     // sweep, barrier, deal + MFMA, lane reduction, math + stores
#define PROF_MARK(i) { const u64 p1 = __builtin_amdgcn_s_memtime(); pc[i] += p1 - p0; p0 = p1; }
#else
#define PROF_MARK(i)
#endif
    auto step = [&](const int s, const float (&gcur)[3], float (&gnext)[3]) __attribute__((always_inline)) {
        const int t = dir == 0 ? s : T - 1 - s;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (s > 0) {
            {
                // sweep: wave w collects batch row w; every load instruction = 64 lanes x 8 B, contiguous (whole cache lines,
                // each requested once per workgroup)
                constexpr int NL = HT / 64;                  // granules per lane
                const u64* src = hxd + (size_t)((s - 1) & 1) * Bpad * H + (size_t)(b0 + wid) * H + lane;
                float hv[NL];
#pragma unroll
                for (int m = 0; m < NL; ++m) hv[m] = 0.0f;
                unsigned spins = dead ? GRU_SPIN_LIMIT : 0;
                unsigned pend = (b0 + wid < B) ? ((1u << NL) - 1u) : 0u;      // rows >= B are never produced: do not wait
                while (true) {
                    u64 gq[NL];
                    const unsigned want = pend;
#pragma unroll
                    for (int m = 0; m < NL; ++m) gq[m] = (want & (1u << m)) ? __hip_atomic_load(src + m * 64, TAG_RLX_AGENT) : 0ull;
#pragma unroll
                    for (int m = 0; m < NL; ++m)
                        if ((want & (1u << m)) && (unsigned)(gq[m] >> 32) == (unsigned)s) {
                            hv[m] = __uint_as_float((unsigned)gq[m]);
                            pend &= ~(1u << m);
                        }
                    if (__all(pend == 0u)) break;
                    if (++spins > GRU_SPIN_LIMIT) { if (lane == 0) { atomicExch(err, 1u); dead_flag = 1; } break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int m = 0; m < NL; ++m) hs[s & 1][wid][((m * 64 + lane) / NK) * PS + (m * 64 + lane) % NK] = hv[m];
            }
            load_gi(s + 1, gnext);
            PROF_MARK(0)
            __syncthreads();                              // hs[s & 1] complete; it is rewritten at step s + 2, after the barrier of s + 1
            PROF_MARK(1)
            dead = dead || dead_flag != 0;                // after one timeout never wait again (bounded total time)
            float av[NK];
#pragma unroll
            for (int i4 = 0; i4 < NK / 4; ++i4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&hs[s & 1][arow_i][ph * PS + 4 * i4]);
                av[4 * i4] = v[0]; av[4 * i4 + 1] = v[1]; av[4 * i4 + 2] = v[2]; av[4 * i4 + 3] = v[3];
            }
#pragma unroll
            for (int i = 0; i < NK; ++i)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i], wv[i][g], acc[g], 0, 0, 0);
        }
        else load_gi(1, gnext);
        PROF_MARK(2)
        float gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            // acc[g][r] = partial (this lane's k phase) of row r.  swap32 on (row 0, row 1): lanes < 32 then hold row 0's two
            // phase halves in the two results, lanes >= 32 row 1's; likewise rows 2, 3; swap16 folds the next phase bit and
            // leaves lane rows = batch rows {0, 2, 1, 3}; row_ror:8 folds the last bit.
            const u32x2_t s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[g][0]), __float_as_uint(acc[g][1]), false, false);
            const u32x2_t s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[g][2]), __float_as_uint(acc[g][3]), false, false);
            const float t01 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);
            const float t23 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
            const u32x2_t sq = __builtin_amdgcn_permlane16_swap(__float_as_uint(t01), __float_as_uint(t23), false, false);
            gh[g] = add_ror8(__uint_as_float(sq[0]) + __uint_as_float(sq[1])) + bias[g];
        }
        PROF_MARK(3)
        if (valid) {
#ifdef TAG_GRU_LIBM
            const float r = sigmoidf_(gcur[0] + gh[0]);
            const float z = sigmoidf_(gcur[1] + gh[1]);
            const float n = tanhf(gcur[2] + r * gh[2]);
#else
            const float r = sigmoid_hw(gcur[0] + gh[0]);
            const float z = sigmoid_hw(gcur[1] + gh[1]);
            const float n = tanh_hw(gcur[2] + r * gh[2]);
#endif
            float h = (1.0f - z) * n + z * hprev;
            if (dead) h = __int_as_float(0x7fc00000);   // a timed-out exchange must not pass silently: NaN reaches the loss
            hprev = h;
            u64* gdst = hxd + (size_t)(s & 1) * Bpad * H + (size_t)b * H + j;
            if (fast) __hip_atomic_store(gdst, granule((unsigned)(s + 1), h), TAG_RLX_WG);      // sc0: stays in this XCD's L2
            else __hip_atomic_store(gdst, granule((unsigned)(s + 1), h), TAG_RLX_AGENT);        // sc1: write-through
            y[(((size_t)b * T + t) * 2 + dir) * H + j] = h;
            if (gates) {
                float* gs = gates + (((size_t)b * T + t) * 2 + dir) * 4 * H;
                gs[j] = r; gs[H + j] = z; gs[2 * H + j] = n; gs[3 * H + j] = gh[2];
            }
        }
        PROF_MARK(4)
    };
    step(0, ga, gb);                                   // peeled: no sweep, and the weight loads are still in flight
    for (int s = 1; s < T; s += 2) {
        step(s, gb, ga);
        if (s + 1 < T) step(s + 1, ga, gb);
    }
#ifdef TAG_GRU_PROF
    if (blockIdx.x == 8 && lane == 0 && (wid == 0 || wid == 3)) {
        u64* o = reinterpret_cast<u64*>(err + 16) + (wid == 0 ? 0 : 5);
        for (int i = 0; i < 5; ++i) o[i] = pc[i];
    }
#endif
}

template <int HT>
__global__ __launch_bounds__(256) void gru_bwd_p4_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ gates, const float* __restrict__ w_hh,
                                                         float* __restrict__ dgi, float* __restrict__ dgh,
                                                         float* __restrict__ hprev_out, u64* gxch, u64* xid, unsigned* err,
                                                         int B, int T, int nbt, int allow_fast) {
    constexpr int H = HT, NU = HT / GU;              // NU producers per batch tile
    constexpr int AT_LD = GC + 4;                    // row stride of the gate-gradient tile (16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) float At[2][GR][AT_LD];
    const P4Coord co = p4_decode(blockIdx.x, NU, nbt);
    const int dir = co.dir, u = co.u, j0 = u * GU, bt = co.bt, b0 = bt * GR;
    const bool fast = p4_group_shares_xcd(xid, co.grp, co.u, NU, allow_fast);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool mwave = wid * 64 < H;                 // H = 128: two waves cover the outputs
    const int ocol = wid * 64 + lane;                // the output (hidden unit of h_{t-1}) this lane accumulates
    // W_hh rows of the own 96 gate columns (k = gate * 32 + unit), output column ocol
    float wv[GC];
    if (mwave) {
#pragma unroll
        for (int k = 0; k < GC; ++k)
            wv[k] = w_hh[((size_t)dir * 3 * H + (size_t)(k >> 5) * H + j0 + (k & 31)) * H + ocol];
    }
    const int e = threadIdx.x, grow = e >> 5, guu = e & 31;
    const int b = b0 + grow, j = j0 + guu;
    const bool gthr = e < GR * GU, valid = gthr && b < B;
    // exchange layout: [parity][dir][bt][producer][row][H]
    const size_t slab = (size_t)GR * H;
    u64* xbase = gxch + ((size_t)dir * nbt + bt) * NU * slab;
    const size_t parity_stride = (size_t)2 * nbt * NU * slab;
    float dh_carry = 0.0f, z_next = 0.0f;
    bool dead = false;
    // The saved gates / dy / h_{t-1} of a step are loaded ONE STEP AHEAD, right after the sweep of the step before (see the
    // forward kernel: vmcnt is one in-order counter); two register sets, the loop unrolled by two.
    struct StepIn { float r, z, n, hn, dyv, hp; };
    StepIn ia = {0, 0, 0, 0, 0, 0}, ib = {0, 0, 0, 0, 0, 0};
    auto load_in = [&](int s_, StepIn& d) {
        if (valid && s_ < T) {
            const int t_ = dir == 0 ? T - 1 - s_ : s_;
            const int tp_ = dir == 0 ? t_ - 1 : t_ + 1;
            const size_t cell = ((size_t)b * T + t_) * 2 + dir;
            const float* gs = gates + cell * 4 * H;
            d.r = gs[j]; d.z = gs[H + j]; d.n = gs[2 * H + j]; d.hn = gs[3 * H + j];
            d.dyv = dy[cell * H + j];
            const bool has_prev = dir == 0 ? (t_ > 0) : (t_ < T - 1);
            d.hp = has_prev ? y[(((size_t)b * T + tp_) * 2 + dir) * H + j] : 0.0f;
        }
    };
    load_in(0, ia);
    auto step = [&](const int s, const StepIn& in, StepIn& nxt) __attribute__((always_inline)) {
        const int t = dir == 0 ? T - 1 - s : s;
        float o_dr = 0, o_dz = 0, o_dn = 0, o_dnr = 0;
        if (gthr) {
            float msum = 0.0f;
            if (s > 0) {
                const u64* src = xbase + (size_t)((s - 1) & 1) * parity_stride + (size_t)grow * H + j;
                float pv[NU];
#pragma unroll
                for (int p = 0; p < NU; ++p) pv[p] = 0.0f;
                unsigned spins = dead ? GRU_SPIN_LIMIT : 0;
                unsigned pend = valid ? ((1u << NU) - 1u) : 0u;
                while (true) {
                    u64 gq[NU];
                    const unsigned want = pend;
#pragma unroll
                    for (int p = 0; p < NU; ++p)
                        gq[p] = (want & (1u << p)) ? __hip_atomic_load(src + (size_t)p * slab, TAG_RLX_AGENT) : 0ull;
#pragma unroll
                    for (int p = 0; p < NU; ++p)
                        if ((want & (1u << p)) && (unsigned)(gq[p] >> 32) == (unsigned)s) {
                            pv[p] = __uint_as_float((unsigned)gq[p]);
                            pend &= ~(1u << p);
                        }
                    if (__all(pend == 0u)) break;
                    if (++spins > GRU_SPIN_LIMIT) { if (lane == 0) atomicExch(err, 1u); dead = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int p = 0; p < NU; ++p) msum += pv[p];
            }
            load_in(s + 1, nxt);
            float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
            if (valid) {
                float dh = in.dyv;
                if (s > 0) dh += msum + dh_carry * z_next;
                if (dead) dh = __int_as_float(0x7fc00000);  // timed-out exchange: poison the gradients (loud, not silent)
                dh_carry = dh;
                z_next = in.z;
                const float dn = dh * (1.0f - in.z);
                const float dz = dh * (in.hp - in.n);
                const float dn_pre = dn * (1.0f - in.n * in.n);
                const float dz_pre = dz * in.z * (1.0f - in.z);
                const float dr_pre = dn_pre * in.hn * in.r * (1.0f - in.r);
                const float dnr = dn_pre * in.r;
                v0 = dr_pre; v1 = dz_pre; v2 = dnr;
                o_dr = dr_pre; o_dz = dz_pre; o_dn = dn_pre; o_dnr = dnr;
            }
            At[s & 1][grow][guu] = v0; At[s & 1][grow][GU + guu] = v1; At[s & 1][grow][2 * GU + guu] = v2;
        }
        __syncthreads();
        if (s + 1 < T && mwave) {                    // the last step's products are never consumed
            f32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            const float* arow = &At[s & 1][lane & 3][0];     // A operand: lane supplies row lane % 4 of its block
#pragma unroll
            for (int kq = 0; kq < GC / 4; ++kq) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 4 * kq);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q], wv[4 * kq + q], acc[q], 0, 0, 0);
            }
            u64* dst = xbase + (size_t)(s & 1) * parity_stride + (size_t)u * slab + ocol;
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const u64 gv = granule((unsigned)(s + 1), (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]));
                if (fast) __hip_atomic_store(dst + (size_t)r * H, gv, TAG_RLX_WG);
                else __hip_atomic_store(dst + (size_t)r * H, gv, TAG_RLX_AGENT);
            }
        }
        // the step's outputs leave AFTER the products are published: their issue time is off the recurrence's critical path
        if (valid) {
            const size_t cell = ((size_t)b * T + t) * 2 + dir;
            float* gi_o = dgi + cell * 3 * H;
            float* gh_o = dgh + cell * 3 * H;
            gi_o[j] = o_dr; gi_o[H + j] = o_dz; gi_o[2 * H + j] = o_dn;
            gh_o[j] = o_dr; gh_o[H + j] = o_dz; gh_o[2 * H + j] = o_dnr;
            hprev_out[cell * H + j] = in.hp;
        }
        // At is double-buffered: the tile of step s is rewritten at step s + 2, after the barrier of step s + 1
    };
    step(0, ia, ib);                                 // peeled: no sweep, the weight loads are still in flight
    for (int s = 1; s < T; s += 2) {
        step(s, ib, ia);
        if (s + 1 < T) step(s + 1, ia, ib);
    }
}

// The persistent kernels spin on their neighbours: every workgroup of the grid must be resident at once.
// (1) static check: grid <= occupancy(kernel) x CUs of THIS device (256 on a whole MI355X, fewer in a partitioned mode);
// (2) the launch itself is hipLaunchCooperativeKernel -- the runtime then guarantees co-residency (it refuses an
//     over-size grid with hipErrorCooperativeLaunchTooLarge and does not interleave the grid with other queues' work);
// (3) if either fails the launcher falls back to the per-step kernels (same arithmetic, one launch per time step);
// (4) every spin is bounded and a timeout raises the STICKY error word of the scratch (ops.check_async_errors()).
// TAG_GRU_COOP=0 (environment) switches (2) off -> plain launch, for A/B timing only.
template <class K>
bool gru_grid_fits(K kernel, int grid_blocks) {
    static int cus = 0;
    if (cus == 0) { cus = tag_device_cu_count(); if (cus <= 0) cus = 1; }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) return false;
    // one workgroup per CU is what the exchange latency was tuned for; more than one per CU is still correct as long
    // as all are resident, but keep a margin of one block per CU against the occupancy API's optimism (guide, residency)
    const int safe = per_cu > 1 ? per_cu - 1 : 1;
    return grid_blocks <= safe * cus;
}
// option gru_tile4 = 0 selects the 16-row persistent kernels (the round-2 form), for A/B timing
bool gru_tile4_enabled() {
    static int v = -1;
    if (v < 0) v = tag_option("gru_tile4") ? 1 : 0;
    return v == 1;
}
// option gru_xcd = 0: never use the L2-resident (same-XCD) publishing, for A/B timing
// The L2-resident publishing relies on gfx950 behaviour beyond the HIP memory model (workgroup-scope sc0 stores of one
// workgroup being served to agent-scope sc1 loads of another workgroup on the SAME XCD out of that XCD's L2; partition mode
// SPX, MTYPE_RW) and is gated by the run-time HW_REG_XCC_ID check in the kernels.  A stale read shows as a tag mismatch ->
// bounded spin -> sticky timeout word; the host then calls tag_gru_disable_xcd_fast() (ops._check_gru_word) and every later
// launch of the process publishes write-through (sc1), which is correct under every placement.
static int g_gru_xcd_fast = -1;
bool gru_xcd_fast_enabled() {
    if (g_gru_xcd_fast < 0) g_gru_xcd_fast = tag_option("gru_xcd") ? 1 : 0;
    return g_gru_xcd_fast == 1;
}
bool gru_coop_enabled() {
    static int v = -1;
    if (v < 0) v = tag_option("gru_coop") ? 1 : 0;
    return v == 1;
}
// returns hipSuccess when the persistent kernel was launched; anything else -> caller falls back to the step kernels
template <class K>
hipError_t gru_launch_persistent(K kernel, dim3 grid, void** args, hipStream_t st) {
    if (!gru_coop_enabled()) {
        return hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, dim3(256), args, 0, st);
    }
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), grid, dim3(256), args, 0, st);
    if (e != hipSuccess) (void)hipGetLastError();       // clear the sticky launch error; the caller falls back
    return e;
}

}  // namespace

// ws layout: [ 6*H*H floats: transposed recurrent weights | exchange granules 2*2*Bpad*3H u64 | err word ]
static size_t gru_ws_layout(int B, int H, size_t* off_x, size_t* off_err) {
    const size_t Bpad = (size_t)((B + 15) / 16) * 16;
    size_t o = (size_t)6 * H * H * sizeof(float);
    o = (o + 255) / 256 * 256;
    *off_x = o;
    // forward: [2 dirs][2 parities][Bpad][H] granules; backward: [2 parities][2 dirs][Bpad/16][H/16][16][H]
    o += (size_t)2 * 2 * Bpad * H * (size_t)(H / 16) * sizeof(u64);
    *off_err = o;
    return o + 256;
}
extern "C" size_t tag_gru_ws_bytes(int B, int T, int H) {
    (void)T;
    size_t a, b;
    return gru_ws_layout(B, H, &a, &b);
}

extern "C" int tag_gru_forward(const float* gi, const float* w_hh, const float* b_hh, float* y, float* gates,
                               void* ws, int B, int T, int H, void* stream) {
    TAG_CHECK_ARG(gi && w_hh && b_hh && y && ws && B > 0 && T > 0);
    TAG_CHECK_ARG(H % 16 == 0 && H >= 16);
    hipStream_t st = as_stream(stream);
    size_t off_x, off_err;
    gru_ws_layout(B, H, &off_x, &off_err);
    float* wt = static_cast<float*>(ws);
    const dim3 grid(H / 16, (B + 15) / 16, 2);
    const int nblk = (int)(grid.x * grid.y * grid.z);
    if ((H == 256 || H == 128) && T > 1 && gru_tile4_enabled()) {
        // 4-row tiles (default): one workgroup per (32 units, 4 rows, direction); reads w_hh as it lies
        u64* hx = reinterpret_cast<u64*>(static_cast<char*>(ws) + off_x);
        unsigned* err = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + off_err);
        int nbt = (B + GR - 1) / GR, allow_fast = gru_xcd_fast_enabled() ? 1 : 0;
        const dim3 g4((H / GU) * nbt * 2, 1, 1);
        u64* xid = hx + (size_t)2 * 2 * nbt * GR * H;          // behind the forward granules [2 dirs][2 parities][Bpad][H]
        void* args[] = {(void*)&gi, (void*)&w_hh, (void*)&b_hh, (void*)&y, (void*)&gates, (void*)&hx, (void*)&xid, (void*)&err,
                        (void*)&B, (void*)&T, (void*)&nbt, (void*)&allow_fast};
        const bool fits = H == 256 ? gru_grid_fits(gru_fwd_p4_kernel<256>, (int)(g4.x * g4.y * g4.z))
                                   : gru_grid_fits(gru_fwd_p4_kernel<128>, (int)(g4.x * g4.y * g4.z));
        if (fits) {
            if (hipMemsetAsync(hx, 0, off_err - off_x, st) != hipSuccess) { tag_set_error("memset failed"); return TAG_ELAUNCH; }
            const hipError_t e = H == 256 ? gru_launch_persistent(gru_fwd_p4_kernel<256>, g4, args, st)
                                          : gru_launch_persistent(gru_fwd_p4_kernel<128>, g4, args, st);
            if (e == hipSuccess) return 0;
        }
    }
    // the 16-row persistent kernels and the per-step kernels read the transposed recurrent weights
    hipLaunchKernelGGL(gru_transpose_whh_kernel, dim3(cdiv((long)6 * H * H, 256)), dim3(256), 0, st, w_hh, wt, H);
    TAG_LAUNCH_CHECK();
    if ((H == 256 || H == 128) && T > 1) {
        u64* hx = reinterpret_cast<u64*>(static_cast<char*>(ws) + off_x);
        unsigned* err = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + off_err);
        void* args[] = {(void*)&gi, (void*)&wt, (void*)&b_hh, (void*)&y, (void*)&gates, (void*)&hx, (void*)&err, (void*)&B, (void*)&T};
        const bool fits = H == 256 ? gru_grid_fits(gru_fwd_persistent_kernel<256>, nblk)
                                   : gru_grid_fits(gru_fwd_persistent_kernel<128>, nblk);
        if (fits) {
            // tags must not match any epoch (1..T) before the first write; the error word behind them is sticky
            if (hipMemsetAsync(hx, 0, off_err - off_x, st) != hipSuccess) { tag_set_error("memset failed"); return TAG_ELAUNCH; }
            const hipError_t e = H == 256 ? gru_launch_persistent(gru_fwd_persistent_kernel<256>, grid, args, st)
                                          : gru_launch_persistent(gru_fwd_persistent_kernel<128>, grid, args, st);
            if (e == hipSuccess) return 0;
        }
    }
    for (int s = 0; s < T; ++s) {
        if (H == 256)
            hipLaunchKernelGGL(gru_fwd_step_kernel<256>, grid, dim3(256), 0, st, gi, wt, b_hh, y, gates, B, T, H, s);
        else if (H == 128)
            hipLaunchKernelGGL(gru_fwd_step_kernel<128>, grid, dim3(256), 0, st, gi, wt, b_hh, y, gates, B, T, H, s);
        else
            hipLaunchKernelGGL(gru_fwd_step_kernel<0>, grid, dim3(256), 0, st, gi, wt, b_hh, y, gates, B, T, H, s);
    }
    TAG_LAUNCH_CHECK();
    return 0;
}

// scratch: tag_gru_ws_bytes(B,T,H) bytes (the per-step fallback uses the first 2*B*H floats as the running dh)
extern "C" int tag_gru_backward(const float* dy, const float* y, const float* gates, const float* w_hh, float* dgi,
                                float* dgh, float* hprev, void* scratch, int B, int T, int H, void* stream) {
    TAG_CHECK_ARG(dy && y && gates && w_hh && dgi && dgh && hprev && scratch && B > 0 && T > 0);
    TAG_CHECK_ARG(H % 16 == 0 && (3 * H) % 32 == 0);
    hipStream_t st = as_stream(stream);
    const dim3 grid(H / 16, (B + 15) / 16, 2);
    const int nblk = (int)(grid.x * grid.y * grid.z);
    if ((H == 256 || H == 128) && T > 1 && gru_tile4_enabled()) {
        size_t off_x, off_err;
        gru_ws_layout(B, H, &off_x, &off_err);
        u64* gx = reinterpret_cast<u64*>(static_cast<char*>(scratch) + off_x);
        unsigned* err = reinterpret_cast<unsigned*>(static_cast<char*>(scratch) + off_err);
        int nbt = (B + GR - 1) / GR, allow_fast = gru_xcd_fast_enabled() ? 1 : 0;
        const dim3 g4((H / GU) * nbt * 2, 1, 1);
        u64* xid = gx + (size_t)2 * 2 * nbt * (H / GU) * GR * H;   // behind [2 parities][2 dirs][nbt][H/32][4][H]
        void* args[] = {(void*)&dy, (void*)&y, (void*)&gates, (void*)&w_hh, (void*)&dgi, (void*)&dgh, (void*)&hprev, (void*)&gx,
                        (void*)&xid, (void*)&err, (void*)&B, (void*)&T, (void*)&nbt, (void*)&allow_fast};
        const bool fits = H == 256 ? gru_grid_fits(gru_bwd_p4_kernel<256>, (int)(g4.x * g4.y * g4.z))
                                   : gru_grid_fits(gru_bwd_p4_kernel<128>, (int)(g4.x * g4.y * g4.z));
        if (fits) {
            if (hipMemsetAsync(gx, 0, off_err - off_x, st) != hipSuccess) { tag_set_error("memset failed"); return TAG_ELAUNCH; }
            const hipError_t e = H == 256 ? gru_launch_persistent(gru_bwd_p4_kernel<256>, g4, args, st)
                                          : gru_launch_persistent(gru_bwd_p4_kernel<128>, g4, args, st);
            if (e == hipSuccess) return 0;
        }
    }
    if ((H == 256 || H == 128) && T > 1) {
        size_t off_x, off_err;
        gru_ws_layout(B, H, &off_x, &off_err);
        u64* gx = reinterpret_cast<u64*>(static_cast<char*>(scratch) + off_x);
        unsigned* err = reinterpret_cast<unsigned*>(static_cast<char*>(scratch) + off_err);
        void* args[] = {(void*)&dy, (void*)&y, (void*)&gates, (void*)&w_hh, (void*)&dgi, (void*)&dgh, (void*)&hprev, (void*)&gx,
                        (void*)&err, (void*)&B, (void*)&T};
        const bool fits = H == 256 ? gru_grid_fits(gru_bwd_persistent_kernel<256>, nblk)
                                   : gru_grid_fits(gru_bwd_persistent_kernel<128>, nblk);
        if (fits) {
            if (hipMemsetAsync(gx, 0, off_err - off_x, st) != hipSuccess) { tag_set_error("memset failed"); return TAG_ELAUNCH; }
            const hipError_t e = H == 256 ? gru_launch_persistent(gru_bwd_persistent_kernel<256>, grid, args, st)
                                          : gru_launch_persistent(gru_bwd_persistent_kernel<128>, grid, args, st);
            if (e == hipSuccess) return 0;
        }
    }
    float* dhbuf = static_cast<float*>(scratch);
    for (int s = 0; s < T; ++s) {
        if (H == 256)
            hipLaunchKernelGGL(gru_bwd_step_kernel<256>, grid, dim3(256), 0, st, dy, y, gates, w_hh, dgi, dgh, hprev,
                               dhbuf, B, T, H, s);
        else if (H == 128)
            hipLaunchKernelGGL(gru_bwd_step_kernel<128>, grid, dim3(256), 0, st, dy, y, gates, w_hh, dgi, dgh, hprev,
                               dhbuf, B, T, H, s);
        else
            hipLaunchKernelGGL(gru_bwd_step_kernel<0>, grid, dim3(256), 0, st, dy, y, gates, w_hh, dgi, dgh, hprev,
                               dhbuf, B, T, H, s);
    }
    TAG_LAUNCH_CHECK();
    return 0;
}

// after a timeout: never use the L2-resident publishing again in this process (returns the previous setting)
extern "C" int tag_gru_disable_xcd_fast(void) {
    const int was = gru_xcd_fast_enabled() ? 1 : 0;
    g_gru_xcd_fast = 0;
    return was;
}

// last persistent GRU launch that used this scratch timed out waiting for a neighbour (host-side check after a sync)
extern "C" int tag_gru_timed_out(const void* ws_host_copy_of_err_word) {
    return ws_host_copy_of_err_word && *static_cast<const unsigned*>(ws_host_copy_of_err_word) != 0;
}
