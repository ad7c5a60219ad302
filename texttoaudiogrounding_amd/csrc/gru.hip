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
bool gru_coop_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TAG_GRU_COOP"); v = (e && e[0] == '0') ? 0 : 1; }
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
    hipLaunchKernelGGL(gru_transpose_whh_kernel, dim3(cdiv((long)6 * H * H, 256)), dim3(256), 0, st, w_hh, wt, H);
    TAG_LAUNCH_CHECK();
    const dim3 grid(H / 16, (B + 15) / 16, 2);
    const int nblk = (int)(grid.x * grid.y * grid.z);
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

// last persistent GRU launch that used this scratch timed out waiting for a neighbour (host-side check after a sync)
extern "C" int tag_gru_timed_out(const void* ws_host_copy_of_err_word) {
    return ws_host_copy_of_err_word && *static_cast<const unsigned*>(ws_host_copy_of_err_word) != 0;
}
