// BASELINE configs[3], the head the config text names literally: match.CrossAttention (models/match.py:63-88 in the
// reference) = nn.MultiheadAttention(embed_dim, heads, dropout, batch_first, kdim = vdim = kvdim) of every audio frame over
// the phrase tokens, residual + dropout, LayerNorm, Linear(E, 1), sigmoid -- forward and backward.
//
// The projections (q = audio Wq^T + bq over B*T rows; k, v over the B*L token rows; out_proj) are tag_gemm calls (MFMA);
// this file is what sits between them:
//   * mha_cross_{fwd,bwd}: scaled dot-product attention of one frame over L <= 32 tokens, all heads, one wave per
//     (clip, frame) (backward: per (clip, tile of frames) so that the token-side gradients dk / dv -- sums over frames --
//     leave as per-tile partials folded in a fixed order: deterministic, no atomics).  key_padding_mask = -inf on
//     tokens >= text_len; attention dropout (train) uses the counter-based mask of tag_common.h on the softmax weights.
//   * resln_head_{fwd,bwd}: z = audio + dropout(attn_out); LayerNorm(z) * gamma + beta; Linear(E,1); sigmoid -- one wave
//     per frame, E spread over the lanes.  Backward writes dz (shared by the two residual branches up to the dropout mask)
//     and the per-row terms of dW_linear / dgamma / dbeta, which the caller folds with tag_colsum (fixed order).
#include <stdlib.h>
#include "tag_common.h"

namespace {

constexpr int MAXL = 32;      // tokens per phrase
constexpr int QT = 8;         // frames per backward tile
constexpr int MAXE = 16;      // E <= 64 * MAXE

// sum over each aligned group of HL lanes (HL = 16, 32 or 64), result in every lane of the group
template <int HL>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = HL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Per-head dot products of two E-vectors held as NE register slices (element d = lane + 64 * i).
// dh >= 64 (multiple of 64): a head owns dh/64 consecutive slices  -> out[i] = the head's dot, replicated in its slices.
// dh  < 64 (16 or 32)      : a slice holds 64/dh heads             -> out[i] = the dot of the lane's own head.
template <int NE, int HL>
__device__ __forceinline__ void head_dots(const float (&a)[NE], const float (&b)[NE], int slices_per_head, float (&out)[NE]) {
    float p[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) p[i] = group_sum<HL>(a[i] * b[i]);
    if (HL < 64) {
#pragma unroll
        for (int i = 0; i < NE; ++i) out[i] = p[i];
        return;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        float s = 0.0f;
        const int h0 = (i / slices_per_head) * slices_per_head;
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (j >= h0 && j < h0 + slices_per_head) s += p[j];
        out[i] = s;
    }
}

template <int NE>
__device__ __forceinline__ void load_vec(float (&r)[NE], const float* p, int E, int lane) {
#pragma unroll
    for (int i = 0; i < NE; ++i) { const int d = lane + 64 * i; r[i] = d < E ? p[d] : 0.0f; }
}

// attn (B,T,H,L): softmax weights BEFORE dropout (what backward needs); ctx (B,T,E).
template <int NE, int HL>
__global__ __launch_bounds__(256) void mha_cross_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const long* __restrict__ klen,
                                                            float* __restrict__ attn, float* __restrict__ ctx, int B, int T,
                                                            int L, int E, int H, float scale, float drop_p, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * T) return;
    const int b = (int)(row / T);
    const int kl = (int)klen[b];
    const int dh = E / H, sph = dh >= 64 ? dh / 64 : 1;
    float qv[NE];
    load_vec<NE>(qv, q + row * E, E, lane);
    // scores of this lane's head(s) against every token: sc[k][i] is the (replicated) score of slice i's head
    float mx[NE], den[NE], acc[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { mx[i] = -3.0e38f; den[i] = 0.0f; acc[i] = 0.0f; }
    // pass 1: row maxima (scores recomputed in pass 2: L <= 32, E <= 1024 -- a few thousand FMAs per frame)
    for (int t = 0; t < L && t < kl; ++t) {
        float kv_[NE], s[NE];
        load_vec<NE>(kv_, k + ((long)b * L + t) * E, E, lane);
        head_dots<NE, HL>(qv, kv_, sph, s);
#pragma unroll
        for (int i = 0; i < NE; ++i) mx[i] = fmaxf(mx[i], s[i] * scale);
    }
    for (int t = 0; t < L && t < kl; ++t) {
        float kv_[NE], s[NE];
        load_vec<NE>(kv_, k + ((long)b * L + t) * E, E, lane);
        head_dots<NE, HL>(qv, kv_, sph, s);
#pragma unroll
        for (int i = 0; i < NE; ++i) den[i] += expf(s[i] * scale - mx[i]);
    }
    const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (int t = 0; t < L; ++t) {
        float a[NE];
        if (t < kl) {
            float kv_[NE], s[NE];
            load_vec<NE>(kv_, k + ((long)b * L + t) * E, E, lane);
            head_dots<NE, HL>(qv, kv_, sph, s);
#pragma unroll
            for (int i = 0; i < NE; ++i) a[i] = expf(s[i] * scale - mx[i]) / den[i];
        } else {
#pragma unroll
            // exp(-inf) of the key_padding_mask; a phrase with NO valid token (text_len 0) is a softmax over an all-masked
            // row: nn.MultiheadAttention yields NaN there (models/match.py:75-79) and so does this kernel -- loud, not 0
            for (int i = 0; i < NE; ++i) a[i] = kl > 0 ? 0.0f : __int_as_float(0x7fc00000);
        }
        float vv[NE];
        load_vec<NE>(vv, v + ((long)b * L + t) * E, E, lane);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int d = lane + 64 * i;
            if (d < E) {
                const int h = d / dh;
                if (d % dh == 0) attn[(row * H + h) * L + t] = a[i];
                float ad = a[i];
                if (drop_p > 0.0f) ad = tag_keep(seed, (uint64_t)((row * H + h) * L + t), drop_p) ? ad * keep_scale : 0.0f;
                acc[i] = fmaf(ad, vv[i], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) { const int d = lane + 64 * i; if (d < E) ctx[row * E + d] = acc[i]; }
}

// One wave per (clip, tile of QT frames).  dq (B,T,E) final; dk_p / dv_p (B,NT,L,E) per-tile partials.
template <int NE, int HL>
__global__ __launch_bounds__(64) void mha_cross_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ attn,
                                                           const float* __restrict__ dctx, const long* __restrict__ klen,
                                                           float* __restrict__ dq, float* __restrict__ dk_p,
                                                           float* __restrict__ dv_p, int B, int T, int L, int E, int H,
                                                           float scale, float drop_p, uint64_t seed, int NT) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x / NT, tile = blockIdx.x % NT;
    const int q0 = tile * QT, q1 = min(T, q0 + QT);
    const int kl = (int)klen[b];
    const int dh = E / H, sph = dh >= 64 ? dh / 64 : 1;
    const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float* dk_out = dk_p + ((size_t)b * NT + tile) * L * E;
    float* dv_out = dv_p + ((size_t)b * NT + tile) * L * E;
    for (int t = 0; t < L; ++t) {
        float kt[NE], vt[NE], dkacc[NE], dvacc[NE];
        load_vec<NE>(kt, k + ((size_t)b * L + t) * E, E, lane);
        load_vec<NE>(vt, v + ((size_t)b * L + t) * E, E, lane);
#pragma unroll
        for (int i = 0; i < NE; ++i) { dkacc[i] = 0.0f; dvacc[i] = 0.0f; }
        for (int qi = q0; qi < q1; ++qi) {
            const size_t row = (size_t)b * T + qi;
            float dc[NE], qv[NE];
            load_vec<NE>(dc, dctx + row * E, E, lane);
            load_vec<NE>(qv, q + row * E, E, lane);
            // softmax backward of token t needs sum_j a_j * dA_j over all tokens j of the same head
            float dot_all[NE], dA_t[NE], a_t[NE], ad_t[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) { dot_all[i] = 0.0f; dA_t[i] = 0.0f; a_t[i] = 0.0f; ad_t[i] = 0.0f; }
            for (int j = 0; j < L && j < kl; ++j) {
                float vj[NE], s[NE];
                load_vec<NE>(vj, v + ((size_t)b * L + j) * E, E, lane);
                head_dots<NE, HL>(dc, vj, sph, s);                  // d(attn after dropout)[j] per head
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int d = lane + 64 * i;
                    const int h = d < E ? d / dh : 0;
                    const float a = attn[(row * H + h) * L + j];
                    float g = s[i];                                  // through the dropout on the weights
                    float ad = a;
                    if (drop_p > 0.0f) {
                        const bool keep = tag_keep(seed, (uint64_t)((row * H + h) * L + j), drop_p);
                        g = keep ? g * keep_scale : 0.0f;
                        ad = keep ? a * keep_scale : 0.0f;
                    }
                    dot_all[i] = fmaf(a, g, dot_all[i]);
                    if (j == t) { dA_t[i] = g; a_t[i] = a; ad_t[i] = ad; }
                }
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int d = lane + 64 * i;
                if (d < E) {
                    const float ds = t < kl ? a_t[i] * (dA_t[i] - dot_all[i]) * scale : 0.0f;    // d score (incl. 1/sqrt(dh))
                    dkacc[i] = fmaf(ds, qv[i], dkacc[i]);
                    dvacc[i] = fmaf(ad_t[i], dc[i], dvacc[i]);
                    const float g = ds * kt[i];
                    if (t == 0) dq[row * E + d] = g;                 // same wave, program order: later tokens add
                    else dq[row * E + d] += g;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int d = lane + 64 * i;
            if (d < E) { dk_out[(size_t)t * E + d] = dkacc[i]; dv_out[(size_t)t * E + d] = dvacc[i]; }
        }
    }
}

// out[o][i] = sum_{t < NT} part[o][t][i]
__global__ __launch_bounds__(256) void mha_fold_tiles_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                             long outer, int NT, long inner) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < outer * inner; e += (long)gridDim.x * 256) {
        const long o = e / inner, i = e % inner;
        float s = 0.0f;
        for (int t = 0; t < NT; ++t) s += part[(o * NT + t) * inner + i];
        out[e] = s;
    }
}

// z = x + dropout(r); n = LayerNorm(z) * gamma + beta; sim = sigmoid(n . w + b)
template <int NE>
__global__ __launch_bounds__(256) void resln_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ sim, float* __restrict__ mu_out,
                                                             float* __restrict__ rstd_out, long rows, int E, float eps,
                                                             float drop_p, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float z[NE];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int d = lane + 64 * i;
        z[i] = 0.0f;
        if (d < E) {
            float rv = r[row * E + d];
            if (drop_p > 0.0f) rv = tag_keep(seed, (uint64_t)(row * E + d), drop_p) ? rv * keep_scale : 0.0f;
            z[i] = x[row * E + d] + rv;
            s += z[i];
        }
    }
    const float mu = wave_sum(s) / (float)E;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i) { const int d = lane + 64 * i; if (d < E) { const float c = z[i] - mu; q = fmaf(c, c, q); } }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
    float dot = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int d = lane + 64 * i;
        if (d < E) dot = fmaf(fmaf((z[i] - mu) * rstd, gamma[d], beta[d]), w[d], dot);
    }
    dot = wave_sum(dot) + bias[0];
    if (lane == 0) {
        sim[row] = 1.0f / (1.0f + expf(-dot));
        mu_out[row] = mu;
        rstd_out[row] = rstd;
    }
}

// dz (-> dx; dr = dz through the dropout mask); per-row terms gw = ds * n, gg = dn * xhat, gb = dn; ds_out = d logit
template <int NE>
__global__ __launch_bounds__(256) void resln_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ w, const float* __restrict__ mu_in,
                                                             const float* __restrict__ rstd_in, const float* __restrict__ sim,
                                                             const float* __restrict__ dsim, float* __restrict__ dx,
                                                             float* __restrict__ dr, float* __restrict__ gw,
                                                             float* __restrict__ gg, float* __restrict__ gb,
                                                             float* __restrict__ ds_out, long rows, int E, float drop_p,
                                                             uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const float mu = mu_in[row], rstd = rstd_in[row];
    const float p = sim[row];
    const float ds = dsim[row] * p * (1.0f - p);
    float xh[NE], dxh[NE], keep[NE];
    float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int d = lane + 64 * i;
        xh[i] = dxh[i] = 0.0f; keep[i] = 1.0f;
        if (d < E) {
            float rv = r[row * E + d];
            if (drop_p > 0.0f) {
                keep[i] = tag_keep(seed, (uint64_t)(row * E + d), drop_p) ? keep_scale : 0.0f;
                rv *= keep[i];
            }
            xh[i] = (x[row * E + d] + rv - mu) * rstd;
            const float dn = ds * w[d];
            gw[row * E + d] = ds * fmaf(xh[i], gamma[d], beta[d]);
            gg[row * E + d] = dn * xh[i];
            gb[row * E + d] = dn;
            dxh[i] = dn * gamma[d];
            m1 += dxh[i];
            m2 = fmaf(dxh[i], xh[i], m2);
        }
    }
    m1 = wave_sum(m1) / (float)E;
    m2 = wave_sum(m2) / (float)E;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int d = lane + 64 * i;
        if (d < E) {
            const float dz = rstd * (dxh[i] - m1 - xh[i] * m2);
            dx[row * E + d] = dz;
            dr[row * E + d] = dz * keep[i];
        }
    }
    if (lane == 0) ds_out[row] = ds;
}

}  // namespace

#define MHA_DISPATCH(E_, DH_, BODY)                                                                     \
    {                                                                                                   \
        const int ne__ = ((E_) + 63) / 64;                                                              \
        if ((DH_) >= 64) {                                                                              \
            if (ne__ <= 2) { constexpr int NE = 2, HL = 64; BODY; }                                     \
            else if (ne__ <= 4) { constexpr int NE = 4, HL = 64; BODY; }                                \
            else if (ne__ <= 8) { constexpr int NE = 8, HL = 64; BODY; }                                \
            else { constexpr int NE = 16, HL = 64; BODY; }                                              \
        } else if ((DH_) == 32) {                                                                       \
            if (ne__ <= 4) { constexpr int NE = 4, HL = 32; BODY; } else { constexpr int NE = 8, HL = 32; BODY; } \
        } else {                                                                                        \
            if (ne__ <= 4) { constexpr int NE = 4, HL = 16; BODY; } else { constexpr int NE = 8, HL = 16; BODY; } \
        }                                                                                               \
    }


// ------------------------------------------------------------------------------------------------------------------------
// The attention core on the matrix pipe (round 4; BASELINE configs[3] names "audio-text cross-attention as MFMA kernel").
// One wave per (clip, head, tile of 32 frames), exact-fp32 v_mfma_f32_32x32x2_f32 (an fmaf chain per output, like every
// fp32 kernel of the library).  The scores are formed TRANSPOSED, S^T[token][frame] = K_h Q_h^T, so that in the MFMA result
// layout (column = lane & 31, rows over the 16 registers and the two half-waves) a lane owns ONE frame and 16 of its 32
// tokens: the softmax over the tokens is a reduction over registers plus one lane ^ 32 exchange, and the probabilities
// P^T[token][frame] sit exactly where the B operand of the next product wants them -- step j of  ctx^T = V_h^T P^T  takes
// register j (k-slot = half-wave <-> token (j&3) + 8 (j>>2) + 4 half) with no data movement at all.  The same holds for
// dQ^T = K_h^T dS^T in the backward pass; the two token-side products (dV = Pd^T-as-(tokens x frames) dctx, dK = dS Q), whose
// contraction runs over the FRAMES (= lanes here), take their A operand through a wave-private 32 x 33 LDS transpose.
// Operand rows are read as contiguous half-rows (k-slot s of step j <-> column s * DH/2 + j: any bijection of the
// contraction index serves both operands alike).  32 + 32 MFMAs per wave forward, 128 backward.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int d_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }     // MFMA result row of register r

template <int HD>
__device__ __forceinline__ void load_half_row(float (&dst)[HD], const float* p, bool ok) {
#pragma unroll
    for (int j = 0; j < HD; j += 4) {
        const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(p + j) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        dst[j] = v.x; dst[j + 1] = v.y; dst[j + 2] = v.z; dst[j + 3] = v.w;
    }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_cross_fwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const long* __restrict__ klen,
                                                                 float* __restrict__ attn, float* __restrict__ ctx, int B, int T,
                                                                 int L, int E, int H, float scale, float drop_p, uint64_t seed) {
    constexpr int HD = DH / 2;
    const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const int NT = (T + 31) / 32;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)B * H * NT) return;
    const int tile = (int)(item % NT), h = (int)((item / NT) % H), b = (int)(item / ((long)NT * H));
    const int f = tile * 32 + n;                                   // this lane's frame (result column)
    const bool fok = f < T;
    const long row = (long)b * T + (fok ? f : T - 1);
    const int kl = (int)klen[b];
    // S^T = K_h Q_h^T: A = K (row = token n), B = Q^T (column = frame n); k-slot `half` of step j = column half * HD + j
    float ka[HD], qb[HD];
    load_half_row<HD>(ka, k + ((long)b * L + (n < L ? n : 0)) * E + h * DH + half * HD, n < L);
    load_half_row<HD>(qb, q + row * E + h * DH + half * HD, fok);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < HD; ++j) s = mfma_f32(ka[j], qb[j], s);
    // softmax over the tokens of this lane's frame: 16 registers here, 16 in lane ^ 32
    float p[16], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] = d_row(r, half) < kl ? s[r] * scale : -INFINITY;     // key_padding_mask
        mx = fmaxf(mx, p[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float den = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = expf(p[r] - mx); den += p[r]; }     // kl == 0: exp(-inf + inf) = NaN, as nn.MultiheadAttention
    den += __shfl_xor(den, 32, 64);
    const float inv = 1.0f / den, keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = d_row(r, half);
        p[r] *= inv;
        const long ai = (row * H + h) * L + t;
        if (fok && t < L) attn[ai] = p[r];                         // the weights BEFORE dropout (what backward needs)
        if (drop_p > 0.0f) p[r] = (t < L && tag_keep(seed, (uint64_t)ai, drop_p)) ? p[r] * keep_scale : 0.0f;
    }
    // ctx^T = V_h^T Pd^T, 32 channels at a time: A = V^T (row = channel, k-slot <-> token d_row(j, half)), B = register j of Pd^T
#pragma unroll
    for (int db = 0; db < DH / 32; ++db) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int t = d_row(j, half);
            const float a = t < L ? v[((long)b * L + t) * E + h * DH + db * 32 + n] : 0.0f;
            c = mfma_f32(a, p[j], c);
        }
        if (fok) {
#pragma unroll
            for (int g = 0; g < 4; ++g)                             // rows 8 g + 4 half .. + 3 = four consecutive channels of the frame
                *reinterpret_cast<f32x4*>(ctx + row * E + h * DH + db * 32 + 8 * g + 4 * half) =
                    (f32x4){c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
        }
    }
}

// dq (B,T,E) final; dk_p / dv_p (B,NT,L,E) per-tile partials (NT = tiles of 32 frames), folded by mha_fold_tiles_kernel.
template <int DH>
__global__ __launch_bounds__(256) void mha_cross_bwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const float* __restrict__ attn,
                                                                 const float* __restrict__ dctx, const long* __restrict__ klen,
                                                                 float* __restrict__ dq, float* __restrict__ dk_p,
                                                                 float* __restrict__ dv_p, int B, int T, int L, int E, int H,
                                                                 float scale, float drop_p, uint64_t seed) {
    constexpr int HD = DH / 2;
    __shared__ float xs[4][32 * 33];                               // wave-private transpose tile [token][frame]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
    const int NT = (T + 31) / 32;
    const long item = (long)blockIdx.x * 4 + wv;
    if (item >= (long)B * H * NT) return;
    const int tile = (int)(item % NT), h = (int)((item / NT) % H), b = (int)(item / ((long)NT * H));
    const int f0 = tile * 32, f = f0 + n;
    const bool fok = f < T;
    const long row = (long)b * T + (fok ? f : T - 1);
    const int kl = (int)klen[b];
    const float keep_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float* X = xs[wv];
    // d(dropped weights)^T[token][frame] = V_h dctx_h^T: same operand shapes as the forward scores
    f32x16 s;
    {
        float va[HD], gb[HD];
        load_half_row<HD>(va, v + ((long)b * L + (n < L ? n : 0)) * E + h * DH + half * HD, n < L);
        load_half_row<HD>(gb, dctx + row * E + h * DH + half * HD, fok);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < HD; ++j) s = mfma_f32(va[j], gb[j], s);
    }
    // softmax backward per frame (= per lane pair): ds = a (g - sum_t a g) scale; ad = the dropped weights the forward multiplied V with
    float ds[16], ad[16], dot = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = d_row(r, half);
        const long ai = (row * H + h) * L + t;
        const float a = (fok && t < L) ? attn[ai] : 0.0f;
        float g = s[r];
        ad[r] = a;
        if (drop_p > 0.0f) {
            const bool keep = t < L && tag_keep(seed, (uint64_t)ai, drop_p);
            g = keep ? g * keep_scale : 0.0f;
            ad[r] = keep ? a * keep_scale : 0.0f;
        }
        ds[r] = g;
        dot = fmaf(a, g, dot);
        s[r] = a;                                                   // keep a for the second half of the formula
    }
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = d_row(r, half) < kl ? s[r] * (ds[r] - dot) * scale : 0.0f;
    // dQ^T = K_h^T dS^T: A = K^T (row = channel, k-slot <-> token d_row(j, half)), B = register j of dS^T
#pragma unroll
    for (int db = 0; db < DH / 32; ++db) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int t = d_row(j, half);
            const float a = t < L ? k[((long)b * L + t) * E + h * DH + db * 32 + n] : 0.0f;
            c = mfma_f32(a, ds[j], c);
        }
        if (fok) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(dq + row * E + h * DH + db * 32 + 8 * g + 4 * half) =
                    (f32x4){c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
        }
    }
    // token-side products: contraction over the FRAMES.  A = M[token][frame] (through the transpose tile: k-slot `half` of step j
    // = frame half * 16 + j), B = rows of dctx / q (column = channel n)
    auto token_side = [&](const float (&m)[16], const float* __restrict__ rhs, float* __restrict__ out) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // earlier reads of X are done
#pragma unroll
        for (int r = 0; r < 16; ++r) X[d_row(r, half) * 33 + n] = fok ? m[r] : 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float am[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) am[j] = X[n * 33 + half * 16 + j];
#pragma unroll
        for (int db = 0; db < DH / 32; ++db) {
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int fr = f0 + half * 16 + j;
                const float bv = fr < T ? rhs[((long)b * T + fr) * E + h * DH + db * 32 + n] : 0.0f;
                c = mfma_f32(am[j], bv, c);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {                          // result row = token, column = channel n
                const int t = d_row(r, half);
                if (t < L) out[(((long)b * NT + tile) * L + t) * E + h * DH + db * 32 + n] = c[r];
            }
        }
    };
    token_side(ad, dctx, dv_p);
    token_side(ds, q, dk_p);
}

static bool mha_mfma_enabled() {
    static int v = -1;
    if (v < 0) v = tag_option("mha_mfma") ? 1 : 0;
    return v == 1;
}
static bool mha_mfma_ok(int E, int H) {
    const int dh = E / H;
    return mha_mfma_enabled() && (dh == 32 || dh == 64 || dh == 128) && E % 4 == 0;
}

static bool mha_shape_ok(int E, int H, int L) {
    if (E <= 0 || H <= 0 || E % H != 0 || L <= 0 || L > MAXL || E > 64 * MAXE) return false;
    const int dh = E / H;
    if (dh >= 64) return dh % 64 == 0;
    return (dh == 32 || dh == 16) && E <= 512;
}

extern "C" int tag_mha_cross_forward(const float* q, const float* k, const float* v, const long* klen, float* attn,
                                     float* ctx, int B, int T, int L, int E, int H, float drop_p, uint64_t seed,
                                     void* stream) {
    TAG_CHECK_ARG(q && k && v && klen && attn && ctx && B > 0 && T > 0);
    TAG_CHECK_ARG(mha_shape_ok(E, H, L) && drop_p >= 0.0f && drop_p < 1.0f);
    const int dh = E / H;
    const float scale = 1.0f / sqrtf((float)dh);
    if (mha_mfma_ok(E, H)) {                                       // the attention core on the matrix pipe (exact fp32 MFMA)
        const int g4 = cdiv((long)B * H * cdiv(T, 32), 4);
#define MFMA_FWD(DH_) hipLaunchKernelGGL((mha_cross_fwd_mfma_kernel<DH_>), dim3(g4), dim3(256), 0, as_stream(stream), q, k, v, klen, \
                                         attn, ctx, B, T, L, E, H, scale, drop_p, seed)
        if (dh == 32) MFMA_FWD(32); else if (dh == 64) MFMA_FWD(64); else MFMA_FWD(128);
#undef MFMA_FWD
        TAG_LAUNCH_CHECK();
        return 0;
    }
    const int grid = cdiv((long)B * T, 4);
    MHA_DISPATCH(E, dh, hipLaunchKernelGGL((mha_cross_fwd_kernel<NE, HL>), dim3(grid), dim3(256), 0, as_stream(stream), q, k,
                                           v, klen, attn, ctx, B, T, L, E, H, scale, drop_p, seed))
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t tag_mha_cross_backward_ws_bytes(int B, int T, int L, int E) {
    return (size_t)2 * B * cdiv(T, QT) * L * E * sizeof(float);
}

extern "C" int tag_mha_cross_backward(const float* q, const float* k, const float* v, const float* attn, const float* dctx,
                                      const long* klen, float* dq, float* dk, float* dv, int B, int T, int L, int E, int H,
                                      float drop_p, uint64_t seed, void* ws, void* stream) {
    TAG_CHECK_ARG(q && k && v && attn && dctx && klen && dq && dk && dv && ws && B > 0 && T > 0);
    TAG_CHECK_ARG(mha_shape_ok(E, H, L) && drop_p >= 0.0f && drop_p < 1.0f);
    const bool mfma = mha_mfma_ok(E, H);
    const int dh = E / H, NT = mfma ? cdiv(T, 32) : cdiv(T, QT);
    const float scale = 1.0f / sqrtf((float)dh);
    float* dk_p = static_cast<float*>(ws);
    float* dv_p = dk_p + (size_t)B * NT * L * E;
    if (mfma) {
        const int g4 = cdiv((long)B * H * NT, 4);
#define MFMA_BWD(DH_) hipLaunchKernelGGL((mha_cross_bwd_mfma_kernel<DH_>), dim3(g4), dim3(256), 0, as_stream(stream), q, k, v, attn, \
                                         dctx, klen, dq, dk_p, dv_p, B, T, L, E, H, scale, drop_p, seed)
        if (dh == 32) MFMA_BWD(32); else if (dh == 64) MFMA_BWD(64); else MFMA_BWD(128);
#undef MFMA_BWD
    } else
    MHA_DISPATCH(E, dh, hipLaunchKernelGGL((mha_cross_bwd_kernel<NE, HL>), dim3(B * NT), dim3(64), 0, as_stream(stream), q, k, v,
                                           attn, dctx, klen, dq, dk_p, dv_p, B, T, L, E, H, scale, drop_p, seed, NT))
    TAG_LAUNCH_CHECK();
    const long inner = (long)L * E;
    const int fb = cdiv((long)B * inner, 256) > 2048 ? 2048 : cdiv((long)B * inner, 256);
    hipLaunchKernelGGL(mha_fold_tiles_kernel, dim3(fb), dim3(256), 0, as_stream(stream), dk_p, dk, (long)B, NT, inner);
    hipLaunchKernelGGL(mha_fold_tiles_kernel, dim3(fb), dim3(256), 0, as_stream(stream), dv_p, dv, (long)B, NT, inner);
    TAG_LAUNCH_CHECK();
    return 0;
}

#define LN_DISPATCH(E_, BODY)                                                     \
    {                                                                             \
        const int ne__ = ((E_) + 63) / 64;                                        \
        if (ne__ <= 4) { constexpr int NE = 4; BODY; }                            \
        else if (ne__ <= 8) { constexpr int NE = 8; BODY; }                       \
        else { constexpr int NE = 16; BODY; }                                     \
    }

extern "C" int tag_resln_head_forward(const float* x, const float* r, const float* gamma, const float* beta, const float* w,
                                      const float* bias, float* sim, float* mu, float* rstd, long rows, int E, float eps,
                                      float drop_p, uint64_t seed, void* stream) {
    TAG_CHECK_ARG(x && r && gamma && beta && w && bias && sim && mu && rstd && rows > 0 && E > 0 && E <= 64 * MAXE);
    LN_DISPATCH(E, hipLaunchKernelGGL(resln_head_fwd_kernel<NE>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, r,
                                      gamma, beta, w, bias, sim, mu, rstd, rows, E, eps, drop_p, seed))
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_resln_head_backward(const float* x, const float* r, const float* gamma, const float* beta, const float* w,
                                       const float* mu, const float* rstd, const float* sim, const float* dsim, float* dx,
                                       float* dr, float* gw, float* gg, float* gb, float* ds, long rows, int E, float drop_p,
                                       uint64_t seed, void* stream) {
    TAG_CHECK_ARG(x && r && gamma && beta && w && mu && rstd && sim && dsim && dx && dr && gw && gg && gb && ds);
    TAG_CHECK_ARG(rows > 0 && E > 0 && E <= 64 * MAXE);
    LN_DISPATCH(E, hipLaunchKernelGGL(resln_head_bwd_kernel<NE>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, r,
                                      gamma, beta, w, mu, rstd, sim, dsim, dx, dr, gw, gg, gb, ds, rows, E, drop_p, seed))
    TAG_LAUNCH_CHECK();
    return 0;
}
