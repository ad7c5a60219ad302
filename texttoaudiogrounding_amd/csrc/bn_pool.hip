// F3 / A1 / A2: BatchNorm statistics, BN+ReLU+(avg+max | LP) pool + dropout, and their backward.
// Reference: nn.BatchNorm2d + F.relu_ + F.avg_pool2d/F.max_pool2d (models/panns.py:46-62),
// bn0 over the mel axis (models/audio_encoder.py:188-190), F.dropout (:203-210),
// cdur_block / nn.LPPool2d (models/audio_encoder.py:16-22,39-49).
//
// All tensors channels-last (rows, C); every kernel is HBM-bound: one float4 per lane, channels
// innermost so a wave reads 1 KiB contiguous.  Per-channel reductions accumulate in fp64 per
// thread, are combined per block through LDS and written as per-block partials; a one-block
// finalize kernel sums the partials in a fixed order (deterministic, no atomics).
#include "tag_common.h"

// bf16 tensors: 8 channels per thread in the pool BACKWARD passes too (1) or only in the forward pass (0)
#ifndef TAG_POOL_BWD_NC8
#define TAG_POOL_BWD_NC8 1
#endif

namespace {

constexpr int RED_MAX_BLOCKS = 1024;

__device__ __forceinline__ float leaky01(float v) { return v > 0.0f ? v : 0.1f * v; }

// ------------------------------------------------------------------------------------------
// generic two-quantity per-channel reduction over (rows, C), C % 4 == 0, (C/4) | 256
// Functor: void operator()(long row, int c, float4& a, float4& b)   (c multiple of 4)
// partials: [nblk][2][C] doubles
// ------------------------------------------------------------------------------------------
template <class Fn>
__global__ __launch_bounds__(256) void reduce2_kernel(Fn fn, long rows, int C, double* __restrict__ partials) {
    extern __shared__ double sred[];   // [256][8]
    const int tpr = C >> 2;                 // threads per row
    const int rpi = 256 / tpr;              // rows per iteration
    const int c = (threadIdx.x % tpr) << 2;
    const int rsub = threadIdx.x / tpr;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    fn.prep(c);                             // per-channel constants -> registers (a thread keeps its channel quad)
    for (long r = (long)blockIdx.x * rpi + rsub; r < rows; r += (long)gridDim.x * rpi) {
        float4 a, b;
        fn(r, c, a, b);
        s1[0] += a.x; s1[1] += a.y; s1[2] += a.z; s1[3] += a.w;
        s2[0] += b.x; s2[1] += b.y; s2[2] += b.z; s2[3] += b.w;
    }
    double* mine = sred + threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { mine[j] = s1[j]; mine[4 + j] = s2[j]; }
    __syncthreads();
    if (rsub == 0) {
        for (int q = 1; q < rpi; ++q) {
            const double* o = sred + (threadIdx.x + q * tpr) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += o[j]; s2[j] += o[4 + j]; }
        }
        double* p = partials + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) { p[c + j] = s1[j]; p[C + c + j] = s2[j]; }
    }
}

// scalar fallback for C % 4 != 0 or C < 4 (CrnnEncoder's 1-channel BatchNorm): thread t owns
// channel t % C; requires C <= 256 and 256 % C == 0
template <class Fn1>
__global__ __launch_bounds__(256) void reduce2_scalar_kernel(Fn1 fn, long rows, int C,
                                                            double* __restrict__ partials) {
    __shared__ double sred[256][2];
    const int c = threadIdx.x % C;
    const int rpi = 256 / C;
    const int rsub = threadIdx.x / C;
    double s1 = 0, s2 = 0;
    for (long r = (long)blockIdx.x * rpi + rsub; r < rows; r += (long)gridDim.x * rpi) {
        float a, b;
        fn(r, c, a, b);
        s1 += a; s2 += b;
    }
    sred[threadIdx.x][0] = s1; sred[threadIdx.x][1] = s2;
    __syncthreads();
    if (rsub == 0) {
        for (int q = 1; q < rpi; ++q) { s1 += sred[threadIdx.x + q * C][0]; s2 += sred[threadIdx.x + q * C][1]; }
        double* p = partials + (size_t)blockIdx.x * 2 * C;
        p[c] = s1; p[C + c] = s2;
    }
}

struct StatsFn {
    const float* x; int C; int pre;
    __device__ void prep(int) {}
    __device__ void operator()(long r, int c, float4& a, float4& b) const {
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
        if (pre == 1) { v.x = leaky01(v.x); v.y = leaky01(v.y); v.z = leaky01(v.z); v.w = leaky01(v.w); }
        a = v;
        b = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
    }
};
struct StatsFn1 {
    const float* x; int C; int pre;
    __device__ void operator()(long r, int c, float& a, float& b) const {
        float v = x[(size_t)r * C + c];
        if (pre == 1) v = leaky01(v);
        a = v; b = v * v;
    }
};

// sum the per-block partials: FOLD_T threads = 16 channels x FOLD_P interleaved parts (4 independent loads in flight per
// thread and quantity), folded through LDS in a fixed order (deterministic).  One block per 16 channels used to walk up
// to 1024 partial rows with 16 threads per channel (64 dependent-latency loads each: 95 us per call, 617 us at C = 512);
// 64 parts x 4-way unrolled loads bring that to a few microseconds.
constexpr int FOLD_P = 64, FOLD_T = 16 * FOLD_P;
__device__ __forceinline__ void fold_partials(const double* __restrict__ partials, int nblk, int C, int c, int part,
                                              double& s1, double& s2) {
    __shared__ double sh[2][FOLD_P][17];
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (c < C) {
        int blk = part;
        for (; blk + 3 * FOLD_P < nblk; blk += 4 * FOLD_P) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] += partials[(size_t)(blk + u * FOLD_P) * 2 * C + c];
                b[u] += partials[(size_t)(blk + u * FOLD_P) * 2 * C + C + c];
            }
        }
        for (; blk < nblk; blk += FOLD_P) {
            a[0] += partials[(size_t)blk * 2 * C + c];
            b[0] += partials[(size_t)blk * 2 * C + C + c];
        }
    }
    sh[0][part][threadIdx.x & 15] = (a[0] + a[1]) + (a[2] + a[3]);
    sh[1][part][threadIdx.x & 15] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    s1 = 0; s2 = 0;
    if (part == 0)
        for (int q = 0; q < FOLD_P; ++q) { s1 += sh[0][q][threadIdx.x & 15]; s2 += sh[1][q][threadIdx.x & 15]; }
}

__global__ __launch_bounds__(FOLD_T) void bn_stats_finalize_kernel(const double* __restrict__ partials, int nblk,
                                                               long rows, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               float momentum, float* running_mean,
                                                               float* running_var, float* mean, float* invstd,
                                                               float* scale, float* shift) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    double s1, s2;
    fold_partials(partials, nblk, C, c, part, s1, s2);
    if (part != 0 || c >= C) return;
    const double m = s1 / (double)rows;
    double var = s2 / (double)rows - m * m;
    if (var < 0) var = 0;
    const double is = 1.0 / sqrt(var + (double)eps);
    if (mean) mean[c] = (float)m;
    if (invstd) invstd[c] = (float)is;
    const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    const float sc = (float)(g * is);
    if (scale) scale[c] = sc;
    if (shift) shift[c] = (float)((double)bt - m * (double)g * is);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// Batch statistics from the per-tile partials the conv kernels write in their epilogue: row p holds, per channel, a
// pivot mu_p (the tile mean as rounded in fp32), r_p = sum(y - mu_p) and q_p = sum((y - mu_p)^2); cnt[p] = pixels of
// the tile.  Tile sum s_p = n_p mu_p + r_p, tile M2 = q_p - r_p^2 / n_p; merged in fp64 (Chan et al.) about the pivot
// K = mu_0:  N var = sum M2_p + sum s_p^2 / n_p - S^2 / N.  Two stages: NCH row chunks -> 4 doubles per (chunk,
// channel), then one block per 16 channels folds the chunks and writes the BatchNorm outputs.
constexpr int STAT_MAX_CHUNKS = 256;
__device__ __forceinline__ void fold16(double (&v)[4], double (*sh)[16][17], int g, int cl) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k][g][cl] = v[k];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t = 0;
            for (int q = 0; q < 16; ++q) t += sh[k][q][cl];
            v[k] = t;
        }
    }
}
__global__ __launch_bounds__(256) void bn_stats_partials_reduce_kernel(const float* __restrict__ part,
                                                                      const float* __restrict__ cnt, int P, int C,
                                                                      int rows_per_chunk, double* __restrict__ out) {
    __shared__ double sh[4][16][17];
    const int cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, g = threadIdx.x >> 4;
    const int p0 = blockIdx.y * rows_per_chunk, p1 = min(P, p0 + rows_per_chunk);
    const double K = c < C ? (double)part[c] : 0.0;
    double v[4] = {0, 0, 0, 0};                       // sum about K, sum s^2/n, sum M2, n
    if (c < C)
        for (int p = p0 + g; p < p1; p += 16) {
            const double np = (double)cnt[p];
            if (np > 0) {
                const double rp = (double)part[(size_t)p * 3 * C + C + c];
                const double sp = np * ((double)part[(size_t)p * 3 * C + c] - K) + rp;
                v[0] += sp; v[1] += sp * sp / np; v[2] += (double)part[(size_t)p * 3 * C + 2 * C + c] - rp * rp / np; v[3] += np;
            }
        }
    fold16(v, sh, g, cl);
    if (g == 0 && c < C) {
        double* o = out + ((size_t)blockIdx.y * C + c) * 4;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
}
__global__ __launch_bounds__(256) void bn_stats_partials_finalize_kernel(const double* __restrict__ chunks, int nch,
                                                                        const float* __restrict__ part, int C,
                                                                        const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, float eps,
                                                                        float momentum, float* running_mean,
                                                                        float* running_var, float* mean, float* invstd,
                                                                        float* scale, float* shift) {
    __shared__ double sh[4][16][17];
    const int cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, g = threadIdx.x >> 4;
    double v[4] = {0, 0, 0, 0};
    if (c < C)
        for (int ch = g; ch < nch; ch += 16) {
            const double* o = chunks + ((size_t)ch * C + c) * 4;
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
    fold16(v, sh, g, cl);
    if (g != 0 || c >= C) return;
    const double a = v[0], bq = v[1], cq = v[2], n = v[3];
    const double m = (double)part[c] + a / n;
    double var = (cq + bq - a * a / n) / n;
    if (var < 0) var = 0;
    const double is = 1.0 / sqrt(var + (double)eps);
    if (mean) mean[c] = (float)m;
    if (invstd) invstd[c] = (float)is;
    const float gm = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    if (scale) scale[c] = (float)(gm * is);
    if (shift) shift[c] = (float)((double)bt - m * (double)gm * is);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) {
        const double unb = n > 1 ? var * n / (n - 1) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// dgamma / dbeta finalize: partial slot 0 = sum dz, slot 1 = sum dz*xhat
__global__ __launch_bounds__(FOLD_T) void bn_grad_finalize_kernel(const double* __restrict__ partials, int nblk, int C,
                                                              float* dgamma, float* dbeta) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    double s1, s2;
    fold_partials(partials, nblk, C, c, part, s1, s2);
    if (part != 0 || c >= C) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}

// dgamma / dbeta from the fp32 partial rows the dgrad conv epilogue writes (conv3x3_halo_kernel<.., EPI = 1>): row p =
// [sum g | sum g*xhat] over one 64-pixel wave tile.  Stage 1: (16 channels) x (row chunk) blocks fold their rows in fp64;
// stage 2: one block per 16 channels folds the chunks.  Fixed order throughout.
__global__ __launch_bounds__(256) void bn_grad_partials_reduce_kernel(const float* __restrict__ part, int P, int C,
                                                                     int rows_per_chunk, double* __restrict__ out) {
    __shared__ double sh[2][16][17];
    const int cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, g = threadIdx.x >> 4;
    const int p0 = blockIdx.y * rows_per_chunk, p1 = min(P, p0 + rows_per_chunk);
    double a = 0, b = 0;
    if (c < C)
        for (int p = p0 + g; p < p1; p += 16) {
            a += (double)part[(size_t)p * 2 * C + c];
            b += (double)part[(size_t)p * 2 * C + C + c];
        }
    sh[0][g][cl] = a; sh[1][g][cl] = b;
    __syncthreads();
    if (g == 0 && c < C) {
        double t1 = 0, t2 = 0;
        for (int q = 0; q < 16; ++q) { t1 += sh[0][q][cl]; t2 += sh[1][q][cl]; }
        out[((size_t)blockIdx.y * 2) * C + c] = t1;           // same [blk][2][C] layout as the reduce2 partials
        out[((size_t)blockIdx.y * 2 + 1) * C + c] = t2;
    }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    scale[c] = g * is;
    shift[c] = b - rm[c] * g * is;
}

__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, long n4, int C,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 s = *reinterpret_cast<const float4*>(scale + c);
        const float4 t = *reinterpret_cast<const float4*>(shift + c);
        v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y); v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

struct ParamGradFn {   // plain affine BN (bn0): a = dy, b = dy * xhat
    const float* x; const float* dy; const float* mean; const float* invstd; int C;
    float4 m, is;
    __device__ void prep(int c) {
        m = *reinterpret_cast<const float4*>(mean + c);
        is = *reinterpret_cast<const float4*>(invstd + c);
    }
    __device__ void operator()(long r, int c, float4& a, float4& b) const {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
        const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)r * C + c);
        a = g;
        b = make_float4(g.x * (v.x - m.x) * is.x, g.y * (v.y - m.y) * is.y, g.z * (v.z - m.z) * is.z,
                        g.w * (v.w - m.w) * is.w);
    }
};

// ------------------------------------------------------------------------------------------
// forward: act(bn(y)) -> pool -> dropout
// ------------------------------------------------------------------------------------------
// The bf16 instances of the pool forward and of the pool-backward reduction are software-pipelined: the loads of a thread's
// NEXT slot are issued before the arithmetic of the current one (two raw register sets, the loop unrolled by two so that no
// register rotation -- at which the compiler would wait for the loads -- is needed).  Those instances are latency-bound
// otherwise: 3-6 waves per SIMD, each alternating between one batch of loads and a few hundred dependent VALU operations
// (rocprofv3: VALU issue 0.15-0.22 per SIMD cycle at 2.1-3.1 TB/s; pipelined: forward 158 -> 128 us, reduction 229 -> 182 us
// on the first two blocks).  The fp32 instances already run at 5.6 TB/s and LOSE 3-7 % to the extra registers, and so does
// the backward apply pass in both storage types: they keep the plain loop.
template <class TS> constexpr bool pool_pipelined() { return Act<TS>::is_bf16; }
// Slot r = (b * Hs + hs) * Ws + ws of a thread that walks the slots with a fixed stride: the decomposition is carried from
// slot to slot (two 32-bit divisions per thread, none in the loop -- the 64-bit r / Ws, r % Ws of every load and finish were
// most of the VALU time of the bf16 pool passes).  Pixel and slot counts are < 2^31 (checked by the launchers).
struct SlotIx {
    int b, hs, ws;
    __device__ __forceinline__ void set(long r, int Hs, int Ws) {
        const unsigned ur = (unsigned)r, q = ur / (unsigned)Ws;
        ws = (int)(ur - q * (unsigned)Ws);
        b = (int)(q / (unsigned)Hs);
        hs = (int)(q - (unsigned)b * (unsigned)Hs);
    }
    // *this + d (d = the decomposed stride), with carries
    __device__ __forceinline__ SlotIx plus(const SlotIx& d, int Hs, int Ws) const {
        SlotIx o;
        o.ws = ws + d.ws;
        int carry = o.ws >= Ws ? 1 : 0;
        o.ws -= carry ? Ws : 0;
        o.hs = hs + d.hs + carry;
        carry = o.hs >= Hs ? 1 : 0;
        o.hs -= carry ? Hs : 0;
        o.b = b + d.b + carry;
        return o;
    }
};
template <int PH, int PW, class TS = float, int NC = 4>
__global__ __launch_bounds__(256) void bnact_pool_fwd_kernel(const TS* __restrict__ y,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             TS* __restrict__ out, int B, int H, int W, int C,
                                                             int act, int pool, float drop_p, uint64_t seed) {
    typedef typename ActN<TS, NC>::raw_t raw_t;
    const int Ho = H / PH, Wo = W / PW, CN = C / NC, rpi = 256 / CN;
    const int c = (threadIdx.x % CN) * NC, rsub = threadIdx.x / CN;
    const long slots = (long)B * Ho * Wo, stride = (long)gridDim.x * rpi;
    const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const unsigned keep_thr = tag_keep4_threshold(drop_p);
    float s[NC], t[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) { s[j] = scale ? scale[c + j] : 1.0f; t[j] = scale ? shift[c + j] : 0.0f; }
    auto load = [&](const SlotIx& ix, raw_t (&raw)[PH][PW]) {
        const unsigned px = ((unsigned)ix.b * H + ix.hs * PH) * W + ix.ws * PW;      // first input pixel of the window
        const TS* p = y + (size_t)px * C + c;
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw) raw[dh][dw] = ActN<TS, NC>::ldraw(p + (size_t)(dh * W + dw) * C);
    };
    auto finish = [&](long r, const raw_t (&raw)[PH][PW]) {
        const size_t e0 = (size_t)(unsigned)r * C + c;   // flat index of the thread's first output element
        float sum[NC], mx[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) { sum[j] = 0.0f; mx[j] = 0.0f; }
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw) {
                float v[NC];
                ActN<TS, NC>::unpack(raw[dh][dw], v);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    float a = fmaf(v[j], s[j], t[j]);
                    a = (act == 1) ? fmaxf(a, 0.0f) : leaky01(a);
                    if (pool != 1) {
                        sum[j] += a;
                        mx[j] = (dh == 0 && dw == 0) ? a : fmaxf(mx[j], a);
                    } else {
                        const float a2 = a * a;
                        sum[j] += a2 * a2;
                    }
                }
            }
        float rr[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            // pool: 0 avg+max | 1 LPPool(4) | 2 avg | 3 max  (models/panns.py:51-60)
            rr[j] = (pool == 0) ? sum[j] * (1.0f / (PH * PW)) + mx[j]
                  : (pool == 2) ? sum[j] * (1.0f / (PH * PW)) : (pool == 3) ? mx[j] : sqrtf(sqrtf(sum[j]));
        }
        if (drop_p > 0.0f) {
#pragma unroll
            for (int q = 0; q < NC / 4; ++q) {
                const uint64_t bits = tag_keep4_bits(seed, (uint64_t)(e0 >> 2) + q);
#pragma unroll
                for (int j = 0; j < 4; ++j) rr[4 * q + j] = tag_keep4(bits, j, keep_thr) ? rr[4 * q + j] * keep_scale : 0.0f;
            }
        }
        ActN<TS, NC>::st(out + e0, rr);
    };
    raw_t ra[PH][PW], rb[PH][PW];
    long r = (long)blockIdx.x * rpi + rsub;
    SlotIx ia, ib, dstep;
    ia.set(r < slots ? r : 0, Ho, Wo);
    dstep.set(stride, Ho, Wo);
    if constexpr (pool_pipelined<TS>()) {
        if (r < slots) load(ia, ra);
        for (; r < slots; r += 2 * stride) {
            const long r2 = r + stride, r3 = r2 + stride;
            ib = ia.plus(dstep, Ho, Wo);
            if (r2 < slots) load(ib, rb);
            finish(r, ra);
            ia = ib.plus(dstep, Ho, Wo);
            if (r3 < slots) load(ia, ra);
            if (r2 < slots) finish(r2, rb);
        }
    } else {
        for (; r < slots; r += stride) { load(ia, ra); finish(r, ra); ia = ia.plus(dstep, Ho, Wo); }
    }
}

// ------------------------------------------------------------------------------------------
// backward of relu(bn(y)) -> avg+max pool -> dropout.  A thread owns one pooling slot (4 channels),
// including the partial slots of the floor-dropped last row/column (dz = 0 there).
// MODE 0: accumulate (sum dz, sum dz*xhat); MODE 1: write dy.
// ------------------------------------------------------------------------------------------
template <int PH, int PW, class TS = float, int NC = 4>
struct PoolBwdCtx {
    const TS* y; const float* scale; const float* shift; const float* mean; const float* invstd;
    const TS* dout; int B, H, W, C; float drop_p; uint64_t seed;
    float wavg, wmax;                      // pool_type: 'avg+max' (1/(PH*PW), 1) | 'avg' (1/(PH*PW), 0) | 'max' (0, 1)
    float sv[NC], tv[NC], mv[NC], iv[NC];  // per-channel constants of this thread's NC channels
    __device__ void prep(int c) {
#pragma unroll
        for (int j = 0; j < NC; ++j) { sv[j] = scale[c + j]; tv[j] = shift[c + j]; mv[j] = mean[c + j]; iv[j] = invstd[c + j]; }
    }
    typedef typename ActN<TS, NC>::raw_t raw_t;
    struct Raw { raw_t v[PH][PW]; raw_t g; };
    // the loads of the slot (b, hs, ws, c..c+NC-1): positions outside the image load nothing (partial slots of the
    // floor-dropped last row / column), only full slots have an upstream gradient
    // 32-bit pixel indices (pixel counts < 2^31: launcher), one widening multiply per tensor, window positions at uniform offsets
    __device__ __forceinline__ size_t in_elem(int b, int hs, int ws, int c) const {
        return (size_t)(((unsigned)b * H + hs * PH) * W + ws * PW) * C + c;
    }
    __device__ __forceinline__ size_t out_elem(int b, int hs, int ws, int c) const {
        return (size_t)(((unsigned)b * (H / PH) + hs) * (W / PW) + ws) * C + c;
    }
    __device__ void load(int b, int hs, int ws, int c, Raw& raw) const {
        const int Ho = H / PH, Wo = W / PW;
        const TS* p = y + in_elem(b, hs, ws, c);
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw) {
                const int h = hs * PH + dh, w = ws * PW + dw;
                raw.v[dh][dw] = (h < H && w < W) ? ActN<TS, NC>::ldraw(p + (size_t)(dh * W + dw) * C) : ActN<TS, NC>::zero();
            }
        raw.g = (hs < Ho && ws < Wo) ? ActN<TS, NC>::ldraw(dout + out_elem(b, hs, ws, c)) : ActN<TS, NC>::zero();
    }
    // dz for the slot from its loaded values; ex[dh][dw] tells which positions exist
    __device__ void compute(const Raw& raw, int b, int hs, int ws, int c, float dz[PH][PW][NC], float xh[PH][PW][NC],
                            bool ex[PH][PW]) const {
        const int Ho = H / PH, Wo = W / PW;
        const bool full = hs < Ho && ws < Wo;
        float a[PH][PW][NC];
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw) {
                const int h = hs * PH + dh, w = ws * PW + dw;
                ex[dh][dw] = h < H && w < W;
                float vv[NC];
                ActN<TS, NC>::unpack(raw.v[dh][dw], vv);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    a[dh][dw][j] = fmaf(vv[j], sv[j], tv[j]);
                    xh[dh][dw][j] = (vv[j] - mv[j]) * iv[j];
                    dz[dh][dw][j] = 0.0f;
                }
            }
        if (!full) return;
        const size_t oi = out_elem(b, hs, ws, c);
        float g[NC];
        ActN<TS, NC>::unpack(raw.g, g);
        const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
        if (drop_p > 0.0f) {
            const unsigned keep_thr = tag_keep4_threshold(drop_p);
#pragma unroll
            for (int q = 0; q < NC / 4; ++q) {
                const uint64_t bits = tag_keep4_bits(seed, (uint64_t)(oi >> 2) + q);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[4 * q + j] = tag_keep4(bits, j, keep_thr) ? g[4 * q + j] * keep_scale : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            // The max-pool gradient goes to the first maximum of relu(a) in scan order (h then w), as ATen's max_pool2d picks it.
            // Where that maximum is positive it is the first position with a == max(a); where it is zero every dz of the slot
            // is zero anyway (a <= 0 everywhere), so the raw maximum decides: one compare per position, the first-hit
            // bookkeeping on the scalar unit, two selects.
            float m = a[0][0][j];
#pragma unroll
            for (int dh = 0; dh < PH; ++dh)
#pragma unroll
                for (int dw = 0; dw < PW; ++dw) m = fmaxf(m, a[dh][dw][j]);
            const float gw = g[j] * wavg, gwm = g[j] * (wavg + wmax);
            bool found = false;
#pragma unroll
            for (int dh = 0; dh < PH; ++dh)
#pragma unroll
                for (int dw = 0; dw < PW; ++dw) {
                    const bool eq = a[dh][dw][j] == m;
                    const bool hit = eq && !found;
                    found = found || eq;
                    dz[dh][dw][j] = a[dh][dw][j] > 0.0f ? (hit ? gwm : gw) : 0.0f;
                }
        }
    }
};

template <int PH, int PW, class TS = float, int NC = 4>
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(PoolBwdCtx<PH, PW, TS, NC> ctx, double* __restrict__ partials) {
    extern __shared__ double sred[];
    const int C = ctx.C, tpr = C / NC, rpi = 256 / tpr;
    const int c = (threadIdx.x % tpr) * NC, rsub = threadIdx.x / tpr;
    const int Ho = ctx.H / PH, Wo = ctx.W / PW;   // only full slots carry gradient
    const long slots = (long)ctx.B * Ho * Wo;
    double s1[NC], s2[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
    ctx.prep(c);
    typedef typename PoolBwdCtx<PH, PW, TS, NC>::Raw Raw;
    const long stride = (long)gridDim.x * rpi;
    auto load = [&](const SlotIx& ix, Raw& raw) { ctx.load(ix.b, ix.hs, ix.ws, c, raw); };
    auto finish = [&](const SlotIx& ix, const Raw& raw) {
        float dz[PH][PW][NC], xh[PH][PW][NC]; bool ex[PH][PW];
        ctx.compute(raw, ix.b, ix.hs, ix.ws, c, dz, xh, ex);
        // per-slot sums in fp32 (PH * PW terms), folded into the thread's fp64 running sums once per slot
        float f1[NC], f2[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) { f1[j] = 0.0f; f2[j] = 0.0f; }
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw)
#pragma unroll
                for (int j = 0; j < NC; ++j) { f1[j] += dz[dh][dw][j]; f2[j] = fmaf(dz[dh][dw][j], xh[dh][dw][j], f2[j]); }
#pragma unroll
        for (int j = 0; j < NC; ++j) { s1[j] += (double)f1[j]; s2[j] += (double)f2[j]; }
    };
    {
        Raw ra, rb;
        long r = (long)blockIdx.x * rpi + rsub;
        SlotIx ia, ib, ic, dstep;
        ia.set(r < slots ? r : 0, Ho, Wo);
        dstep.set(stride, Ho, Wo);
        if constexpr (pool_pipelined<TS>()) {
            if (r < slots) load(ia, ra);
            for (; r < slots; r += 2 * stride) {
                const long r2 = r + stride, r3 = r2 + stride;
                ib = ia.plus(dstep, Ho, Wo);
                if (r2 < slots) load(ib, rb);
                finish(ia, ra);
                ic = ib.plus(dstep, Ho, Wo);
                if (r3 < slots) load(ic, ra);
                if (r2 < slots) finish(ib, rb);
                ia = ic;
            }
        } else {
            for (; r < slots; r += stride) { load(ia, ra); finish(ia, ra); ia = ia.plus(dstep, Ho, Wo); }
        }
    }
    double* mine = sred + threadIdx.x * 2 * NC;
#pragma unroll
    for (int j = 0; j < NC; ++j) { mine[j] = s1[j]; mine[NC + j] = s2[j]; }
    __syncthreads();
    if (rsub == 0) {
        for (int q = 1; q < rpi; ++q) {
            const double* o = sred + (threadIdx.x + q * tpr) * 2 * NC;
#pragma unroll
            for (int j = 0; j < NC; ++j) { s1[j] += o[j]; s2[j] += o[NC + j]; }
        }
        double* p = partials + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int j = 0; j < NC; ++j) { p[c + j] = s1[j]; p[C + c + j] = s2[j]; }
    }
}

template <int PH, int PW, class TS = float, int NC = 4>
__global__ __launch_bounds__(256) void pool_bwd_apply_kernel(PoolBwdCtx<PH, PW, TS, NC> ctx, const float* __restrict__ gamma,
                                                             const float* __restrict__ dgamma,
                                                             const float* __restrict__ dbeta, int bn_train,
                                                             TS* __restrict__ dy) {
    const int C = ctx.C, tpr = C / NC, rpi = 256 / tpr;
    const int c = (threadIdx.x % tpr) * NC, rsub = threadIdx.x / tpr;
    const int Hs = (ctx.H + PH - 1) / PH, Ws = (ctx.W + PW - 1) / PW;
    const long slots = (long)ctx.B * Hs * Ws;
    const float invN = 1.0f / (float)((long)ctx.B * ctx.H * ctx.W);
    ctx.prep(c);
    float k0[NC], k1[NC], k2[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        k0[j] = gamma[c + j] * ctx.iv[j];
        k1[j] = bn_train ? dbeta[c + j] * invN : 0.0f;
        k2[j] = bn_train ? dgamma[c + j] * invN : 0.0f;
    }
    typedef typename PoolBwdCtx<PH, PW, TS, NC>::Raw Raw;
    const long stride = (long)gridDim.x * rpi;
    auto load = [&](const SlotIx& ix, Raw& raw) { ctx.load(ix.b, ix.hs, ix.ws, c, raw); };
    auto finish = [&](const SlotIx& ix, const Raw& raw) {
        float dz[PH][PW][NC], xh[PH][PW][NC]; bool ex[PH][PW];
        ctx.compute(raw, ix.b, ix.hs, ix.ws, c, dz, xh, ex);
        TS* p = dy + ctx.in_elem(ix.b, ix.hs, ix.ws, c);
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw) {
                if (!ex[dh][dw]) continue;
                float o[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) o[j] = k0[j] * (dz[dh][dw][j] - k1[j] - xh[dh][dw][j] * k2[j]);
                ActN<TS, NC>::st(p + (size_t)(dh * ctx.W + dw) * C, o);
            }
    };
    Raw ra;
    long r = (long)blockIdx.x * rpi + rsub;
    SlotIx ia, dstep;
    ia.set(r < slots ? r : 0, Hs, Ws);
    dstep.set(stride, Hs, Ws);
    for (; r < slots; r += stride) { load(ia, ra); finish(ia, ra); ia = ia.plus(dstep, Hs, Ws); }
}

// plain relu(bn(y)) backward (no pooling)
template <class TS = float>
struct BnReluBwdFnT {
    const TS* y; const float* scale; const float* shift; const float* mean; const float* invstd;
    const TS* da; int C;
    float4 s, t, m, is;
    __device__ void prep(int c) {
        s = *reinterpret_cast<const float4*>(scale + c);
        t = *reinterpret_cast<const float4*>(shift + c);
        m = *reinterpret_cast<const float4*>(mean + c);
        is = *reinterpret_cast<const float4*>(invstd + c);
    }
    __device__ void operator()(long r, int c, float4& a, float4& b) const {
        const f32x4 v = Act<TS>::ld4(y + (size_t)r * C + c);
        const f32x4 g = Act<TS>::ld4(da + (size_t)r * C + c);
        a.x = fmaf(v.x, s.x, t.x) > 0.0f ? g.x : 0.0f;
        a.y = fmaf(v.y, s.y, t.y) > 0.0f ? g.y : 0.0f;
        a.z = fmaf(v.z, s.z, t.z) > 0.0f ? g.z : 0.0f;
        a.w = fmaf(v.w, s.w, t.w) > 0.0f ? g.w : 0.0f;
        b = make_float4(a.x * (v.x - m.x) * is.x, a.y * (v.y - m.y) * is.y, a.z * (v.z - m.z) * is.z,
                        a.w * (v.w - m.w) * is.w);
    }
};

typedef BnReluBwdFnT<float> BnReluBwdFn;

template <class TS>
__global__ __launch_bounds__(256) void bnrelu_bwd_apply_kernel(BnReluBwdFnT<TS> fn, const float* __restrict__ gamma,
                                                               const float* __restrict__ dgamma,
                                                               const float* __restrict__ dbeta, int bn_train,
                                                               long rows, TS* __restrict__ dy) {
    const int C = fn.C, tpr = C >> 2, rpi = 256 / tpr;
    const int c = (threadIdx.x % tpr) << 2, rsub = threadIdx.x / tpr;
    const float invN = 1.0f / (float)rows;
    fn.prep(c);
    float k0[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        k0[j] = gamma[c + j] * fn.invstd[c + j];
        k1[j] = bn_train ? dbeta[c + j] * invN : 0.0f;
        k2[j] = bn_train ? dgamma[c + j] * invN : 0.0f;
    }
    const float mv[4] = {fn.m.x, fn.m.y, fn.m.z, fn.m.w}, iv[4] = {fn.is.x, fn.is.y, fn.is.z, fn.is.w};
    for (long r = (long)blockIdx.x * rpi + rsub; r < rows; r += (long)gridDim.x * rpi) {
        float4 dz, dzx;
        fn(r, c, dz, dzx);
        const f32x4 v = Act<TS>::ld4(fn.y + (size_t)r * C + c);
        const float vv[4] = {v.x, v.y, v.z, v.w}, dzv[4] = {dz.x, dz.y, dz.z, dz.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = k0[j] * (dzv[j] - k1[j] - (vv[j] - mv[j]) * iv[j] * k2[j]);
        Act<TS>::st4(dy + (size_t)r * C + c, (f32x4){o[0], o[1], o[2], o[3]});
    }
}

// ------------------------------------------------------------------------------------------
// CrnnEncoder (cdur_block = BN -> conv -> LeakyReLU, LPPool; models/audio_encoder.py:16-22,39-49):
// BatchNorm sits in FRONT of the conv, on v = pre(x) with pre = identity (after a pool) or leaky_relu(.,0.1)
// (after a conv).  Backward of u = bn(v): dv = g*invstd*(du - mean(du) - vhat*mean(du*vhat)), dx = dv*pre'(x).
// ------------------------------------------------------------------------------------------
struct BnActBwdFn {
    const float* x; const float* mean; const float* invstd; const float* du; int C; int pre;
    float4 m, is;
    __device__ void prep(int c) {
        m = *reinterpret_cast<const float4*>(mean + c);
        is = *reinterpret_cast<const float4*>(invstd + c);
    }
    __device__ void operator()(long r, int c, float4& a, float4& b) const {
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
        if (pre == 1) { v.x = leaky01(v.x); v.y = leaky01(v.y); v.z = leaky01(v.z); v.w = leaky01(v.w); }
        a = *reinterpret_cast<const float4*>(du + (size_t)r * C + c);
        b = make_float4(a.x * (v.x - m.x) * is.x, a.y * (v.y - m.y) * is.y, a.z * (v.z - m.z) * is.z,
                        a.w * (v.w - m.w) * is.w);
    }
};

__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(BnActBwdFn fn, const float* __restrict__ gamma,
                                                               const float* __restrict__ dgamma,
                                                               const float* __restrict__ dbeta, int bn_train,
                                                               long rows, float* __restrict__ dx) {
    const int C = fn.C, tpr = C >> 2, rpi = 256 / tpr;
    const int c = (threadIdx.x % tpr) << 2, rsub = threadIdx.x / tpr;
    const float invN = 1.0f / (float)rows;
    fn.prep(c);
    float k0[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        k0[j] = (gamma ? gamma[c + j] : 1.0f) * fn.invstd[c + j];
        k1[j] = bn_train ? dbeta[c + j] * invN : 0.0f;
        k2[j] = bn_train ? dgamma[c + j] * invN : 0.0f;
    }
    const float mv[4] = {fn.m.x, fn.m.y, fn.m.z, fn.m.w}, iv[4] = {fn.is.x, fn.is.y, fn.is.z, fn.is.w};
    for (long r = (long)blockIdx.x * rpi + rsub; r < rows; r += (long)gridDim.x * rpi) {
        const float4 xv = *reinterpret_cast<const float4*>(fn.x + (size_t)r * C + c);
        const float4 g = *reinterpret_cast<const float4*>(fn.du + (size_t)r * C + c);
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w}, gg[4] = {g.x, g.y, g.z, g.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = fn.pre == 1 ? leaky01(xx[j]) : xx[j];
            const float dv = k0[j] * (gg[j] - k1[j] - (v - mv[j]) * iv[j] * k2[j]);
            o[j] = (fn.pre == 1 && xx[j] <= 0.0f) ? 0.1f * dv : dv;
        }
        *reinterpret_cast<float4*>(dx + (size_t)r * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// backward of  out = dropout( LPPool4( leaky_relu(y, 0.1) ) ):  dy = dout * a^3 / out^3 * leaky'(y)
template <int PH, int PW>
__global__ __launch_bounds__(256) void lppool_leaky_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                               float* __restrict__ dy, int B, int H, int W, int C,
                                                               float drop_p, uint64_t seed) {
    const int Ho = H / PH, Wo = W / PW, C4 = C >> 2, rpi = 256 / C4;
    const int c = (threadIdx.x % C4) << 2, rsub = threadIdx.x / C4;
    const int Hs = (H + PH - 1) / PH, Ws = (W + PW - 1) / PW;
    const long slots = (long)B * Hs * Ws;
    const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (long r = (long)blockIdx.x * rpi + rsub; r < slots; r += (long)gridDim.x * rpi) {
        const int ws = (int)(r % Ws); long q = r / Ws;
        const int hs = (int)(q % Hs); const int b = (int)(q / Hs);
        const bool full = hs < Ho && ws < Wo;
        float a[PH][PW][4], yv[PH][PW][4], sum[4] = {0, 0, 0, 0};
        bool ex[PH][PW];
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw) {
                const int h = hs * PH + dh, w = ws * PW + dw;
                ex[dh][dw] = h < H && w < W;
                float4 v = make_float4(0, 0, 0, 0);
                if (ex[dh][dw]) v = *reinterpret_cast<const float4*>(y + (((size_t)b * H + h) * W + w) * C + c);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    yv[dh][dw][j] = vv[j];
                    a[dh][dw][j] = leaky01(vv[j]);
                    const float a2 = a[dh][dw][j] * a[dh][dw][j];
                    sum[j] += a2 * a2;
                }
            }
        float k[4] = {0, 0, 0, 0};
        if (full) {
            const size_t oi = (((size_t)b * Ho + hs) * Wo + ws) * C + c;
            const float4 g4 = *reinterpret_cast<const float4*>(dout + oi);
            float g[4] = {g4.x, g4.y, g4.z, g4.w};
            const uint64_t bits = drop_p > 0.0f ? tag_keep4_bits(seed, (uint64_t)(oi >> 2)) : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (drop_p > 0.0f) g[j] = tag_keep4(bits, j, tag_keep4_threshold(drop_p)) ? g[j] * keep_scale : 0.0f;
                const float out = sqrtf(sqrtf(sum[j]));
                k[j] = out > 0.0f ? g[j] / (out * out * out) : 0.0f;
            }
        }
#pragma unroll
        for (int dh = 0; dh < PH; ++dh)
#pragma unroll
            for (int dw = 0; dw < PW; ++dw) {
                if (!ex[dh][dw]) continue;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float aa = a[dh][dw][j];
                    const float da = k[j] * aa * aa * aa;
                    o[j] = yv[dh][dw][j] > 0.0f ? da : 0.1f * da;
                }
                const int h = hs * PH + dh, w = ws * PW + dw;
                *reinterpret_cast<float4*>(dy + (((size_t)b * H + h) * W + w) * C + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

__global__ void dropout_mask_kernel(uint64_t seed, long n, float p, uint8_t* mask, int pooled) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        mask[i] = (pooled ? tag_keep4(tag_keep4_bits(seed, (uint64_t)i >> 2), (int)(i & 3), tag_keep4_threshold(p))
                          : tag_keep(seed, (uint64_t)i, p)) ? 1 : 0;
}

// mean over W then dropout: x (rows, W, C) -> (rows, C)
template <class TS>
__global__ __launch_bounds__(256) void mean_w_fwd_kernel(const TS* __restrict__ x, long rows, int W, int C,
                                                         float drop_p, uint64_t seed, float* __restrict__ out) {
    const long total = rows * C;
    const float ks = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C; const int c = (int)(i % C);
        float s = 0.0f;
        for (int w = 0; w < W; ++w) s += Act<TS>::ld1(x + ((size_t)r * W + w) * C + c);
        s = s / (float)W;
        if (drop_p > 0.0f) s = tag_keep(seed, (uint64_t)i, drop_p) ? s * ks : 0.0f;
        out[i] = s;
    }
}
template <class TS>
__global__ __launch_bounds__(256) void mean_w_bwd_kernel(const float* __restrict__ dout, long rows, int W, int C,
                                                         float drop_p, uint64_t seed, TS* __restrict__ dx) {
    const long total = rows * C;
    const float ks = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C; const int c = (int)(i % C);
        float g = dout[i];
        if (drop_p > 0.0f) g = tag_keep(seed, (uint64_t)i, drop_p) ? g * ks : 0.0f;
        g = g / (float)W;
        for (int w = 0; w < W; ++w) Act<TS>::st1(dx + ((size_t)r * W + w) * C + c, g);
    }
}

int red_blocks(long rows, int C, int nc = 4) {
    const int rpi = 256 / (C / nc > 0 ? C / nc : 1);
    long nb = (rows + (long)rpi * 8 - 1) / ((long)rpi * 8);
    if (nb > RED_MAX_BLOCKS) nb = RED_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    return (int)nb;
}
// row-strided elementwise kernels: 256/(C/4) rows per block iteration, <= 8 iterations per thread at full size
int apply_blocks(long rows, int C, int nc = 4) {
    const int rpi = 256 / (C / nc);
    long nb = (rows + (long)rpi * 4 - 1) / ((long)rpi * 4);
    if (nb > 8192) nb = 8192;
    return (int)(nb < 1 ? 1 : nb);
}
bool vec_ok(int C) { return C % 4 == 0 && C >= 4 && (C >> 2) <= 256 && 256 % (C >> 2) == 0; }
// channels per thread of the pool passes: 8 for bf16 tensors when the channel count allows (16-byte accesses), else 4
template <class TS>
constexpr bool pool_nc8_type() { return Act<TS>::is_bf16; }
bool pool_nc8_ok(int C) { return C % 8 == 0 && (C >> 3) <= 256 && 256 % (C >> 3) == 0; }
int ew_blocks(long n) {
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

extern "C" size_t tag_bn_stats_ws_bytes(long rows, int C) {
    (void)rows;
    return (size_t)RED_MAX_BLOCKS * 2 * (size_t)C * sizeof(double);
}
extern "C" size_t tag_bn_backward_ws_bytes(long rows, int C) { return tag_bn_stats_ws_bytes(rows, C); }

extern "C" int tag_bn_stats(const float* x, long rows, int C, int pre_op, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var, float* mean,
                            float* invstd, float* scale, float* shift, void* ws, void* stream) {
    TAG_CHECK_ARG(x && ws && rows > 0 && C > 0);
    double* partials = static_cast<double*>(ws);
    int nblk;
    if (vec_ok(C)) {
        nblk = red_blocks(rows, C);
        hipLaunchKernelGGL(reduce2_kernel<StatsFn>, dim3(nblk), dim3(256), 256 * 8 * sizeof(double),
                           as_stream(stream), StatsFn{x, C, pre_op}, rows, C, partials);
    } else {
        TAG_CHECK_ARG(C <= 256 && 256 % C == 0);
        const int rpi = 256 / C;
        long nb = (rows + (long)rpi * 32 - 1) / ((long)rpi * 32);
        nblk = (int)(nb > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : (nb < 1 ? 1 : nb));
        hipLaunchKernelGGL(reduce2_scalar_kernel<StatsFn1>, dim3(nblk), dim3(256), 0, as_stream(stream),
                           StatsFn1{x, C, pre_op}, rows, C, partials);
    }
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(C, 16)), dim3(FOLD_T), 0, as_stream(stream), partials, nblk,
                       rows, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
    TAG_LAUNCH_CHECK();
    return 0;
}

static int stat_chunks(int P, int* rows_per_chunk) {
    int r = (P + STAT_MAX_CHUNKS - 1) / STAT_MAX_CHUNKS;
    if (r < 64) r = 64;
    *rows_per_chunk = r;
    return (P + r - 1) / r;
}
extern "C" size_t tag_bn_stats_from_partials_ws_bytes(int P, int C) {
    int r;
    return (size_t)stat_chunks(P, &r) * C * 4 * sizeof(double);
}
extern "C" int tag_bn_stats_from_partials(const float* partials, int P, int C, const float* gamma, const float* beta,
                                          float eps, float momentum, float* running_mean, float* running_var,
                                          float* mean, float* invstd, float* scale, float* shift, void* ws,
                                          void* stream) {
    TAG_CHECK_ARG(partials && ws && P > 0 && C > 0);
    int rpc;
    const int nch = stat_chunks(P, &rpc);
    double* chunks = static_cast<double*>(ws);
    hipLaunchKernelGGL(bn_stats_partials_reduce_kernel, dim3(cdiv(C, 16), nch), dim3(256), 0, as_stream(stream), partials,
                       partials + (size_t)P * 3 * C, P, C, rpc, chunks);
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_partials_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, as_stream(stream), chunks, nch,
                       partials, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* scale, float* shift,
                                  void* stream) {
    TAG_CHECK_ARG(running_mean && running_var && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 64)), dim3(64), 0, as_stream(stream), gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_affine_forward(const float* x, long rows, int C, const float* scale, const float* shift, float* y,
                                  void* stream) {
    TAG_CHECK_ARG(x && y && scale && shift && C % 4 == 0);
    const long n4 = rows * C / 4;
    hipLaunchKernelGGL(affine_kernel, dim3(ew_blocks(n4)), dim3(256), 0, as_stream(stream), x, n4, C, scale, shift, y);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_bn_param_grad(const float* x, const float* dy, long rows, int C, const float* mean,
                                 const float* invstd, float* dgamma, float* dbeta, void* ws, void* stream) {
    TAG_CHECK_ARG(x && dy && mean && invstd && dgamma && dbeta && ws && vec_ok(C));
    double* partials = static_cast<double*>(ws);
    const int nblk = red_blocks(rows, C);
    hipLaunchKernelGGL(reduce2_kernel<ParamGradFn>, dim3(nblk), dim3(256), 256 * 8 * sizeof(double),
                       as_stream(stream), ParamGradFn{x, dy, mean, invstd, C}, rows, C, partials);
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3(cdiv(C, 16)), dim3(FOLD_T), 0, as_stream(stream), partials, nblk, C,
                       dgamma, dbeta);
    TAG_LAUNCH_CHECK();
    return 0;
}

#define DISPATCH_POOL(PH_, PW_, ...)                       \
    if (ph == PH_ && pw == PW_) { constexpr int PH = PH_, PW = PW_; __VA_ARGS__; launched = true; }

template <class TS>
static int bnact_pool_forward_impl(const TS* y, const float* scale, const float* shift, TS* out, int B, int H, int W, int C,
                                   int ph, int pw, int act, int pool, float drop_p, uint64_t seed, void* stream) {
    TAG_CHECK_ARG(y && out && C % 4 == 0 && (act == 1 || act == 2) && pool >= 0 && pool <= 3);
    TAG_CHECK_ARG((scale == nullptr) == (shift == nullptr));
    TAG_CHECK_ARG(H / ph > 0 && W / pw > 0);
    TAG_CHECK_ARG(vec_ok(C));
    TAG_CHECK_ARG((long)B * H * W < (1L << 31));          // 32-bit pixel indices in the kernel
    bool launched = false;
    const bool nc8 = pool_nc8_type<TS>() && pool_nc8_ok(C);
    const int nb = apply_blocks((long)B * (H / ph) * (W / pw), C, nc8 ? 8 : 4);
#define POOL_FWD_BODY                                                                                              \
    if constexpr (pool_nc8_type<TS>()) {                                                                            \
        if (nc8) {                                                                                                  \
            hipLaunchKernelGGL((bnact_pool_fwd_kernel<PH, PW, TS, 8>), dim3(nb), dim3(256), 0, as_stream(stream), y, scale, \
                               shift, out, B, H, W, C, act, pool, drop_p, seed);                                    \
        } else {                                                                                                    \
            hipLaunchKernelGGL((bnact_pool_fwd_kernel<PH, PW, TS, 4>), dim3(nb), dim3(256), 0, as_stream(stream), y, scale, \
                               shift, out, B, H, W, C, act, pool, drop_p, seed);                                    \
        }                                                                                                           \
    } else {                                                                                                        \
        hipLaunchKernelGGL((bnact_pool_fwd_kernel<PH, PW, TS, 4>), dim3(nb), dim3(256), 0, as_stream(stream), y, scale, shift, \
                           out, B, H, W, C, act, pool, drop_p, seed);                                               \
    }
    DISPATCH_POOL(2, 2, POOL_FWD_BODY)
    DISPATCH_POOL(1, 2, POOL_FWD_BODY)
    DISPATCH_POOL(2, 4, POOL_FWD_BODY)
    DISPATCH_POOL(1, 4, POOL_FWD_BODY)
    DISPATCH_POOL(1, 1, POOL_FWD_BODY)
    DISPATCH_POOL(2, 1, POOL_FWD_BODY)
#undef POOL_FWD_BODY
    TAG_CHECK_ARG(launched);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_bnact_pool_forward(const float* y, const float* scale, const float* shift, float* out, int B,
                                      int H, int W, int C, int ph, int pw, int act, int pool, float drop_p,
                                      uint64_t seed, void* stream) {
    return bnact_pool_forward_impl<float>(y, scale, shift, out, B, H, W, C, ph, pw, act, pool, drop_p, seed, stream);
}
extern "C" int tag_bnact_pool_forward_bf16(const void* y, const float* scale, const float* shift, void* out, int B,
                                           int H, int W, int C, int ph, int pw, int act, int pool, float drop_p,
                                           uint64_t seed, void* stream) {
    return bnact_pool_forward_impl<bf16_t>(static_cast<const bf16_t*>(y), scale, shift, static_cast<bf16_t*>(out), B, H, W, C,
                                           ph, pw, act, pool, drop_p, seed, stream);
}

template <class TS>
static int bnrelu_pool_backward_impl(const TS* y, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const float* gamma, const TS* dout, TS* dy, float* dgamma,
                                     float* dbeta, int B, int H, int W, int C, int ph, int pw, int pool, float drop_p,
                                     uint64_t seed, int bn_train, void* ws, void* stream) {
    TAG_CHECK_ARG(y && scale && shift && mean && invstd && gamma && dout && dy && dgamma && dbeta && ws);
    TAG_CHECK_ARG(pool == 0 || pool == 2 || pool == 3);
    const float wavg = pool == 3 ? 0.0f : 1.0f / (float)(ph * pw), wmax = pool == 2 ? 0.0f : 1.0f;
    TAG_CHECK_ARG(vec_ok(C) && H / ph > 0 && W / pw > 0);
    TAG_CHECK_ARG((long)B * H * W < (1L << 31));          // 32-bit pixel indices in the kernels
    double* partials = static_cast<double*>(ws);
    const long slots = (long)B * (H / ph) * (W / pw);
    const bool nc8 = pool_nc8_type<TS>() && pool_nc8_ok(C) && TAG_POOL_BWD_NC8;
    const int nblk = red_blocks(slots, C, nc8 ? 8 : 4);
    const int nb = apply_blocks((long)B * ((H + ph - 1) / ph) * ((W + pw - 1) / pw), C, nc8 ? 8 : 4);
    bool launched = false;
#define POOL_BWD_NC(NC_)                                                                                           \
    {                                                                                                              \
        PoolBwdCtx<PH, PW, TS, NC_> ctx{y, scale, shift, mean, invstd, dout, B, H, W, C, drop_p, seed, wavg, wmax}; \
        hipLaunchKernelGGL((pool_bwd_reduce_kernel<PH, PW, TS, NC_>), dim3(nblk), dim3(256), 256 * 2 * NC_ * sizeof(double), \
                           as_stream(stream), ctx, partials);                                                      \
        hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3(cdiv(C, 16)), dim3(FOLD_T), 0, as_stream(stream), partials, nblk, \
                           C, dgamma, dbeta);                                                                      \
        hipLaunchKernelGGL((pool_bwd_apply_kernel<PH, PW, TS, NC_>), dim3(nb), dim3(256), 0, as_stream(stream), ctx, gamma, \
                           dgamma, dbeta, bn_train, dy);                                                           \
    }
#define POOL_BWD_BODY                                                                                              \
    if constexpr (pool_nc8_type<TS>()) {                                                                           \
        if (nc8) POOL_BWD_NC(8) else POOL_BWD_NC(4)                                                                \
    } else POOL_BWD_NC(4)
    DISPATCH_POOL(2, 2, POOL_BWD_BODY)
    DISPATCH_POOL(1, 2, POOL_BWD_BODY)
    DISPATCH_POOL(1, 1, POOL_BWD_BODY)
    DISPATCH_POOL(2, 1, POOL_BWD_BODY)
#undef POOL_BWD_BODY
#undef POOL_BWD_NC
    TAG_CHECK_ARG(launched);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_bnrelu_pool_backward(const float* y, const float* scale, const float* shift, const float* mean,
                                        const float* invstd, const float* gamma, const float* dout, float* dy,
                                        float* dgamma, float* dbeta, int B, int H, int W, int C, int ph, int pw,
                                        int pool, float drop_p, uint64_t seed, int bn_train, void* ws, void* stream) {
    return bnrelu_pool_backward_impl<float>(y, scale, shift, mean, invstd, gamma, dout, dy, dgamma, dbeta, B, H, W, C, ph, pw,
                                            pool, drop_p, seed, bn_train, ws, stream);
}
extern "C" int tag_bnrelu_pool_backward_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                                             const float* invstd, const float* gamma, const void* dout, void* dy,
                                             float* dgamma, float* dbeta, int B, int H, int W, int C, int ph, int pw,
                                             int pool, float drop_p, uint64_t seed, int bn_train, void* ws, void* stream) {
    return bnrelu_pool_backward_impl<bf16_t>(static_cast<const bf16_t*>(y), scale, shift, mean, invstd, gamma,
                                             static_cast<const bf16_t*>(dout), static_cast<bf16_t*>(dy), dgamma, dbeta, B, H, W,
                                             C, ph, pw, pool, drop_p, seed, bn_train, ws, stream);
}

// the APPLY half of tag_bnrelu_pool_backward alone: dgamma / dbeta already hold sum(dz * xhat) / sum(dz) (folded from the partial
// rows the dgrad conv that PRODUCED dout wrote in its epilogue: tag_conv3x3_dgrad_poolsums + tag_bn_grad_from_partials)
template <class TS>
static int bnrelu_pool_backward_apply_impl(const TS* y, const float* scale, const float* shift, const float* mean,
                                           const float* invstd, const float* gamma, const TS* dout, TS* dy,
                                           const float* dgamma, const float* dbeta, int B, int H, int W, int C, int ph, int pw,
                                           int pool, float drop_p, uint64_t seed, int bn_train, void* stream) {
    TAG_CHECK_ARG(y && scale && shift && mean && invstd && gamma && dout && dy && dgamma && dbeta);
    TAG_CHECK_ARG(pool == 0 || pool == 2 || pool == 3);
    const float wavg = pool == 3 ? 0.0f : 1.0f / (float)(ph * pw), wmax = pool == 2 ? 0.0f : 1.0f;
    TAG_CHECK_ARG(vec_ok(C) && H / ph > 0 && W / pw > 0);
    TAG_CHECK_ARG((long)B * H * W < (1L << 31));
    const bool nc8 = pool_nc8_type<TS>() && pool_nc8_ok(C) && TAG_POOL_BWD_NC8;
    const int nb = apply_blocks((long)B * ((H + ph - 1) / ph) * ((W + pw - 1) / pw), C, nc8 ? 8 : 4);
    bool launched = false;
#define POOL_APPLY_NC(NC_)                                                                                         \
    {                                                                                                              \
        PoolBwdCtx<PH, PW, TS, NC_> ctx{y, scale, shift, mean, invstd, dout, B, H, W, C, drop_p, seed, wavg, wmax}; \
        hipLaunchKernelGGL((pool_bwd_apply_kernel<PH, PW, TS, NC_>), dim3(nb), dim3(256), 0, as_stream(stream), ctx, gamma, \
                           dgamma, dbeta, bn_train, dy);                                                           \
    }
#define POOL_APPLY_BODY                                                                                            \
    if constexpr (pool_nc8_type<TS>()) {                                                                           \
        if (nc8) POOL_APPLY_NC(8) else POOL_APPLY_NC(4)                                                            \
    } else POOL_APPLY_NC(4)
    DISPATCH_POOL(2, 2, POOL_APPLY_BODY)
    DISPATCH_POOL(1, 2, POOL_APPLY_BODY)
    DISPATCH_POOL(1, 1, POOL_APPLY_BODY)
    DISPATCH_POOL(2, 1, POOL_APPLY_BODY)
#undef POOL_APPLY_BODY
#undef POOL_APPLY_NC
    TAG_CHECK_ARG(launched);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_bnrelu_pool_backward_apply(const float* y, const float* scale, const float* shift, const float* mean,
                                              const float* invstd, const float* gamma, const float* dout, float* dy,
                                              const float* dgamma, const float* dbeta, int B, int H, int W, int C, int ph,
                                              int pw, int pool, float drop_p, uint64_t seed, int bn_train, void* stream) {
    return bnrelu_pool_backward_apply_impl<float>(y, scale, shift, mean, invstd, gamma, dout, dy, dgamma, dbeta, B, H, W, C, ph,
                                                  pw, pool, drop_p, seed, bn_train, stream);
}
extern "C" int tag_bnrelu_pool_backward_apply_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                                                   const float* invstd, const float* gamma, const void* dout, void* dy,
                                                   const float* dgamma, const float* dbeta, int B, int H, int W, int C, int ph,
                                                   int pw, int pool, float drop_p, uint64_t seed, int bn_train, void* stream) {
    return bnrelu_pool_backward_apply_impl<bf16_t>(static_cast<const bf16_t*>(y), scale, shift, mean, invstd, gamma,
                                                   static_cast<const bf16_t*>(dout), static_cast<bf16_t*>(dy), dgamma, dbeta, B, H,
                                                   W, C, ph, pw, pool, drop_p, seed, bn_train, stream);
}

template <class TS>
static int bnrelu_backward_impl(const TS* y, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const float* gamma, const TS* da, TS* dy, float* dgamma,
                                float* dbeta, long rows, int C, int bn_train, void* ws, void* stream) {
    TAG_CHECK_ARG(y && scale && shift && mean && invstd && gamma && da && dy && dgamma && dbeta && ws && vec_ok(C));
    double* partials = static_cast<double*>(ws);
    const int nblk = red_blocks(rows, C);
    BnReluBwdFnT<TS> fn{y, scale, shift, mean, invstd, da, C};
    hipLaunchKernelGGL(reduce2_kernel<BnReluBwdFnT<TS>>, dim3(nblk), dim3(256), 256 * 8 * sizeof(double),
                       as_stream(stream), fn, rows, C, partials);
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3(cdiv(C, 16)), dim3(FOLD_T), 0, as_stream(stream), partials, nblk, C,
                       dgamma, dbeta);
    hipLaunchKernelGGL(bnrelu_bwd_apply_kernel<TS>, dim3(apply_blocks(rows, C)), dim3(256), 0, as_stream(stream), fn,
                       gamma, dgamma, dbeta, bn_train, rows, dy);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_bnrelu_backward(const float* y, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, const float* gamma, const float* da, float* dy,
                                   float* dgamma, float* dbeta, long rows, int C, int bn_train, void* ws,
                                   void* stream) {
    return bnrelu_backward_impl<float>(y, scale, shift, mean, invstd, gamma, da, dy, dgamma, dbeta, rows, C, bn_train, ws,
                                       stream);
}
extern "C" int tag_bnrelu_backward_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                                        const float* invstd, const float* gamma, const void* da, void* dy,
                                        float* dgamma, float* dbeta, long rows, int C, int bn_train, void* ws,
                                        void* stream) {
    return bnrelu_backward_impl<bf16_t>(static_cast<const bf16_t*>(y), scale, shift, mean, invstd, gamma,
                                        static_cast<const bf16_t*>(da), static_cast<bf16_t*>(dy), dgamma, dbeta, rows, C,
                                        bn_train, ws, stream);
}

extern "C" size_t tag_bn_grad_from_partials_ws_bytes(int P, int C) {
    int r;
    return (size_t)stat_chunks(P, &r) * 2 * C * sizeof(double);
}
extern "C" int tag_bn_grad_from_partials(const float* bnpart, int P, int C, float* dgamma, float* dbeta, void* ws,
                                         void* stream) {
    TAG_CHECK_ARG(bnpart && dgamma && dbeta && ws && P > 0 && C > 0);
    int rpc;
    const int nch = stat_chunks(P, &rpc);
    double* chunks = static_cast<double*>(ws);
    hipLaunchKernelGGL(bn_grad_partials_reduce_kernel, dim3(cdiv(C, 16), nch), dim3(256), 0, as_stream(stream), bnpart, P, C,
                       rpc, chunks);
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3(cdiv(C, 16)), dim3(FOLD_T), 0, as_stream(stream), chunks, nch, C, dgamma,
                       dbeta);
    TAG_LAUNCH_CHECK();
    return 0;
}
// the APPLY half of tag_bnrelu_backward alone: dgamma / dbeta already hold sum(g*xhat) / sum(g)
extern "C" int tag_bnrelu_backward_apply(const float* y, const float* scale, const float* shift, const float* mean,
                                         const float* invstd, const float* gamma, const float* da, float* dy,
                                         const float* dgamma, const float* dbeta, long rows, int C, int bn_train,
                                         void* stream) {
    TAG_CHECK_ARG(y && scale && shift && mean && invstd && gamma && da && dy && dgamma && dbeta && vec_ok(C));
    BnReluBwdFn fn{y, scale, shift, mean, invstd, da, C};
    hipLaunchKernelGGL(bnrelu_bwd_apply_kernel<float>, dim3(apply_blocks(rows, C)), dim3(256), 0, as_stream(stream), fn, gamma,
                       dgamma, dbeta, bn_train, rows, dy);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_bnrelu_backward_apply_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                                              const float* invstd, const float* gamma, const void* da, void* dy,
                                              const float* dgamma, const float* dbeta, long rows, int C, int bn_train,
                                              void* stream) {
    TAG_CHECK_ARG(y && scale && shift && mean && invstd && gamma && da && dy && dgamma && dbeta && vec_ok(C));
    BnReluBwdFnT<bf16_t> fn{static_cast<const bf16_t*>(y), scale, shift, mean, invstd, static_cast<const bf16_t*>(da), C};
    hipLaunchKernelGGL(bnrelu_bwd_apply_kernel<bf16_t>, dim3(apply_blocks(rows, C)), dim3(256), 0, as_stream(stream), fn,
                       gamma, dgamma, dbeta, bn_train, rows, static_cast<bf16_t*>(dy));
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_dropout_mask(uint64_t seed, long n, float p, uint8_t* mask, void* stream) {
    TAG_CHECK_ARG(mask && n > 0);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), seed, n, p, mask, 0);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_dropout_mask_pooled(uint64_t seed, long n, float p, uint8_t* mask, void* stream) {
    TAG_CHECK_ARG(mask && n > 0);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, as_stream(stream), seed, n, p, mask, 1);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_mean_w_forward(const float* x, long rows, int W, int C, float drop_p, uint64_t seed, float* out,
                                  void* stream) {
    TAG_CHECK_ARG(x && out && rows > 0 && W > 0 && C > 0);
    hipLaunchKernelGGL(mean_w_fwd_kernel<float>, dim3(ew_blocks(rows * C)), dim3(256), 0, as_stream(stream), x, rows, W, C,
                       drop_p, seed, out);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_mean_w_backward(const float* dout, long rows, int W, int C, float drop_p, uint64_t seed, float* dx,
                                   void* stream) {
    TAG_CHECK_ARG(dout && dx && rows > 0 && W > 0 && C > 0);
    hipLaunchKernelGGL(mean_w_bwd_kernel<float>, dim3(ew_blocks(rows * C)), dim3(256), 0, as_stream(stream), dout, rows, W,
                       C, drop_p, seed, dx);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_mean_w_forward_bf16(const void* x, long rows, int W, int C, float drop_p, uint64_t seed, float* out,
                                       void* stream) {
    TAG_CHECK_ARG(x && out && rows > 0 && W > 0 && C > 0);
    hipLaunchKernelGGL(mean_w_fwd_kernel<bf16_t>, dim3(ew_blocks(rows * C)), dim3(256), 0, as_stream(stream),
                       static_cast<const bf16_t*>(x), rows, W, C, drop_p, seed, out);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_mean_w_backward_bf16(const float* dout, long rows, int W, int C, float drop_p, uint64_t seed, void* dx,
                                        void* stream) {
    TAG_CHECK_ARG(dout && dx && rows > 0 && W > 0 && C > 0);
    hipLaunchKernelGGL(mean_w_bwd_kernel<bf16_t>, dim3(ew_blocks(rows * C)), dim3(256), 0, as_stream(stream), dout, rows, W,
                       C, drop_p, seed, static_cast<bf16_t*>(dx));
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_bn_act_backward(const float* x, int pre_op, const float* mean, const float* invstd,
                                   const float* gamma, const float* du, float* dx, float* dgamma, float* dbeta,
                                   long rows, int C, int bn_train, void* ws, void* stream) {
    TAG_CHECK_ARG(x && mean && invstd && du && dx && dgamma && dbeta && ws && vec_ok(C) && (pre_op == 0 || pre_op == 1));
    double* partials = static_cast<double*>(ws);
    const int nblk = red_blocks(rows, C);
    BnActBwdFn fn{x, mean, invstd, du, C, pre_op};
    hipLaunchKernelGGL(reduce2_kernel<BnActBwdFn>, dim3(nblk), dim3(256), 256 * 8 * sizeof(double), as_stream(stream),
                       fn, rows, C, partials);
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3(cdiv(C, 16)), dim3(FOLD_T), 0, as_stream(stream), partials, nblk, C,
                       dgamma, dbeta);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(apply_blocks(rows, C)), dim3(256), 0, as_stream(stream), fn, gamma,
                       dgamma, dbeta, bn_train, rows, dx);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_lppool_leaky_backward(const float* y, const float* dout, float* dy, int B, int H, int W, int C,
                                         int ph, int pw, float drop_p, uint64_t seed, void* stream) {
    TAG_CHECK_ARG(y && dout && dy && vec_ok(C) && H / ph > 0 && W / pw > 0);
    const int nb = apply_blocks((long)B * ((H + ph - 1) / ph) * ((W + pw - 1) / pw), C);
    bool launched = false;
    DISPATCH_POOL(2, 4, hipLaunchKernelGGL((lppool_leaky_bwd_kernel<PH, PW>), dim3(nb), dim3(256), 0, as_stream(stream),
                                           y, dout, dy, B, H, W, C, drop_p, seed))
    DISPATCH_POOL(1, 4, hipLaunchKernelGGL((lppool_leaky_bwd_kernel<PH, PW>), dim3(nb), dim3(256), 0, as_stream(stream),
                                           y, dout, dy, B, H, W, C, drop_p, seed))
    DISPATCH_POOL(2, 2, hipLaunchKernelGGL((lppool_leaky_bwd_kernel<PH, PW>), dim3(nb), dim3(256), 0, as_stream(stream),
                                           y, dout, dy, B, H, W, C, drop_p, seed))
    TAG_CHECK_ARG(launched);
    TAG_LAUNCH_CHECK();
    return 0;
}
