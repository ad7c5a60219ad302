// T1/T2 text encoder, M1/M2 frame x phrase heads, R1/L1 frame BCE, P1 post-processing.
// All HBM/latency-bound; one wave per embedding row, coalesced D-wide reads.
//   nn.Embedding + mean_with_lens      models/text_encoder.py:39-43,79-88; models/utils.py:33-58
//   match.ExpNegL2 / match.DotProduct  models/match.py:16-33, :43-60
//   FrameBceLoss                       losses.py:12-24 (+ run_strong.py:107-118 label alignment)
//   binarize/median/connect/regions    utils/eval_util.py:18-116, run_strong.py:203-252
#include "tag_common.h"

namespace {

constexpr int MAXD_PER_LANE = 16;   // D <= 1024

// ---------------------------------------------------------------- text encoder
__global__ __launch_bounds__(256) void embed_mean_fwd_kernel(const int64_t* __restrict__ text,
                                                             const int64_t* __restrict__ text_len,
                                                             const float* __restrict__ table,
                                                             float* __restrict__ token_emb, float* __restrict__ seq_emb,
                                                             int L, int D, int V) {
    const int b = blockIdx.x;
    const int len = (int)text_len[b];
    for (int d = threadIdx.x; d < D; d += 256) {
        float s = 0.0f;
        for (int l = 0; l < L; ++l) {
            int64_t tok = text[(size_t)b * L + l];
            tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
            const float v = table[(size_t)tok * D + d];
            if (token_emb) token_emb[((size_t)b * L + l) * D + d] = v;
            if (l < len) s += v;
        }
        seq_emb[(size_t)b * D + d] = s / (float)len;
    }
}
// Deterministic embedding-table gradient (no atomics): one workgroup per (clip, position) pair p.  The workgroup whose
// pair is the FIRST occurrence of its token id walks all later pairs with the same id in ascending order and adds
//   [l < text_len[b]] * dseq[b] / text_len[b]   (mean_with_lens backward)   and / or   dtok[b,l]   (token_emb backward)
// into registers, then adds the total to dtable[token] once -- every table row is owned by exactly one workgroup and
// summed in a fixed order, so a training step is bit-reproducible.  O(P^2) id compares for P = B*L pairs (P <= a few
// thousand on this path).  Out-of-range ids contribute nothing (tag_embed_check_ids raises the error flag for them).
__global__ __launch_bounds__(256) void embed_bwd_det_kernel(const float* __restrict__ dseq, const float* __restrict__ dtok,
                                                            const int64_t* __restrict__ text,
                                                            const int64_t* __restrict__ text_len,
                                                            float* __restrict__ dtable, int P, int L, int D, int V) {
    __shared__ int earlier;
    const int p = blockIdx.x;
    const int64_t tok = text[p];
    if (tok < 0 || tok >= V) return;
    if (threadIdx.x == 0) earlier = 0;
    __syncthreads();
    int found = 0;
    for (int q = threadIdx.x; q < p; q += 256) found |= (text[q] == tok);
    if (found) earlier = 1;                       // benign race: every writer stores 1
    __syncthreads();
    if (earlier) return;
    for (int d0 = 0; d0 < D; d0 += 256) {
        const int d = d0 + threadIdx.x;
        float acc = 0.0f;
        for (int q = p; q < P; ++q) {
            if (text[q] != tok) continue;         // uniform across the workgroup
            const int b = q / L, l = q - b * L;
            if (d < D) {
                if (dseq) {
                    const int len = (int)text_len[b];
                    if (l < len) acc += dseq[(size_t)b * D + d] / (float)len;
                }
                if (dtok) acc += dtok[(size_t)q * D + d];
            }
        }
        if (d < D) dtable[(size_t)tok * D + d] += acc;
    }
}
__global__ __launch_bounds__(256) void embed_check_ids_kernel(const int64_t* __restrict__ text, long n, int V,
                                                              int* __restrict__ err) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int64_t t = text[i];
        if (t < 0 || t >= V) atomicExch(err, 1);
    }
}

// ---------------------------------------------------------------- match heads
struct Row {   // one D-vector spread over a wave
    float v[MAXD_PER_LANE];
};
__device__ __forceinline__ void load_row(Row& r, const float* p, int D, int lane) {
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) {
        const int d = lane + 64 * i;
        r.v[i] = d < D ? p[d] : 0.0f;
    }
}
__device__ __forceinline__ float dot_rows(const Row& a, const Row& b) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) s = fmaf(a.v[i], b.v[i], s);
    return wave_sum(s);
}

// grid = (B, frame chunks), block = 256: wave w of chunk c handles t = 4c + w, 4c + w + 4 gridDim.y, ...
// group = phrases per clip (MultiTextBiEncoder): text row b pairs with audio clip b / group
__global__ __launch_bounds__(256) void match_fwd_kernel(const float* __restrict__ audio, const float* __restrict__ text,
                                                        float* __restrict__ sim, int kind, int l2norm, int scale,
                                                        int T, int D, int group) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    audio += (size_t)(b / group) * T * D - (size_t)b * T * D;
    Row tx;
    load_row(tx, text + (size_t)b * D, D, lane);
    float tinv = 1.0f;
    if (l2norm) {
        tinv = 1.0f / fmaxf(sqrtf(dot_rows(tx, tx)), 1e-12f);
#pragma unroll
        for (int i = 0; i < MAXD_PER_LANE; ++i) tx.v[i] *= tinv;
    }
    const float rs = scale ? 1.0f / sqrtf((float)D) : 1.0f;
    for (int t = blockIdx.y * 4 + wid; t < T; t += 4 * gridDim.y) {
        Row a;
        load_row(a, audio + ((size_t)b * T + t) * D, D, lane);
        if (l2norm) {
            const float ainv = 1.0f / fmaxf(sqrtf(dot_rows(a, a)), 1e-12f);
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) a.v[i] *= ainv;
        }
        float out;
        if (kind == 0) {
            const float s = dot_rows(a, tx) * rs;
            out = fminf(fmaxf(1.0f / (1.0f + expf(-s)), 1e-7f), 1.0f);
        } else {
            float d2 = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) { const float df = a.v[i] - tx.v[i]; d2 = fmaf(df, df, d2); }
            d2 = wave_sum(d2);
            out = expf(-sqrtf(d2));
        }
        if (lane == 0) sim[(size_t)b * T + t] = out;
    }
}

// backward: daudio rows directly; dtext accumulated per wave in registers, reduced through LDS
template <int NW>
__global__ __launch_bounds__(NW * 64) void match_bwd_kernel(const float* __restrict__ audio, const float* __restrict__ text,
                                                            const float* __restrict__ dsim, float* __restrict__ daudio,
                                                            float* __restrict__ dtext, int kind, int l2norm, int scale,
                                                            int T, int D) {
    __shared__ float red[NW][64 * MAXD_PER_LANE];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    Row traw, tx;
    load_row(traw, text + (size_t)b * D, D, lane);
    tx = traw;
    float tinv = 1.0f;
    if (l2norm) {
        tinv = 1.0f / fmaxf(sqrtf(dot_rows(traw, traw)), 1e-12f);
#pragma unroll
        for (int i = 0; i < MAXD_PER_LANE; ++i) tx.v[i] *= tinv;
    }
    const float rs = scale ? 1.0f / sqrtf((float)D) : 1.0f;
    Row dtn;   // gradient wrt the (normalised) text vector
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) dtn.v[i] = 0.0f;
    for (int t = wid; t < T; t += NW) {
        Row araw, a;
        load_row(araw, audio + ((size_t)b * T + t) * D, D, lane);
        a = araw;
        float ainv = 1.0f;
        if (l2norm) {
            ainv = 1.0f / fmaxf(sqrtf(dot_rows(araw, araw)), 1e-12f);
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) a.v[i] *= ainv;
        }
        const float g = dsim[(size_t)b * T + t];
        Row dan;   // gradient wrt the (normalised) audio row
        if (kind == 0) {
            const float s = dot_rows(a, tx) * rs;
            const float p = 1.0f / (1.0f + expf(-s));
            const float pass = (p >= 1e-7f && p <= 1.0f) ? 1.0f : 0.0f;   // clamp backward
            const float ds = g * pass * p * (1.0f - p) * rs;
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) { dan.v[i] = ds * tx.v[i]; dtn.v[i] = fmaf(ds, a.v[i], dtn.v[i]); }
        } else {
            float d2 = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) { const float df = a.v[i] - tx.v[i]; d2 = fmaf(df, df, d2); }
            d2 = wave_sum(d2);
            const float r = sqrtf(d2);
            const float out = expf(-r);
            const float k = r > 0.0f ? -g * out / r : 0.0f;   // d/d(diff) = dr * diff / r, dr = -out * g
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) {
                const float dd = k * (a.v[i] - tx.v[i]);
                dan.v[i] = dd;
                dtn.v[i] -= dd;
            }
        }
        if (l2norm) {   // u = x/||x||: dx = (du - u (u.du)) / ||x||
            const float ud = dot_rows(a, dan);
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) dan.v[i] = (dan.v[i] - a.v[i] * ud) * ainv;
        }
        float* o = daudio + ((size_t)b * T + t) * D;
#pragma unroll
        for (int i = 0; i < MAXD_PER_LANE; ++i) {
            const int d = lane + 64 * i;
            if (d < D) o[d] = dan.v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) red[wid][lane + 64 * i] = dtn.v[i];
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int i = 0; i < MAXD_PER_LANE; ++i) {
            float acc = red[0][lane + 64 * i];
#pragma unroll
            for (int w = 1; w < NW; ++w) acc += red[w][lane + 64 * i];      // fixed order: deterministic
            dtn.v[i] = acc;
        }
        if (l2norm) {
            const float ud = dot_rows(tx, dtn);
#pragma unroll
            for (int i = 0; i < MAXD_PER_LANE; ++i) dtn.v[i] = (dtn.v[i] - tx.v[i] * ud) * tinv;
        }
#pragma unroll
        for (int i = 0; i < MAXD_PER_LANE; ++i) {
            const int d = lane + 64 * i;
            if (d < D) dtext[(size_t)b * D + d] = dtn.v[i];
        }
    }
}

// ---------------------------------------------------------------- frame BCE
__device__ __forceinline__ int clamp_len(int64_t l, int Tt) { return (int)(l < 1 ? 1 : (l > Tt ? Tt : l)); }

__global__ __launch_bounds__(1024) void frame_bce_fwd_kernel(const float* __restrict__ sim, int ld_sim,
                                                             const float* __restrict__ label, int ld_label,
                                                             const int64_t* __restrict__ length, int B, int Tt,
                                                             float* __restrict__ loss) {
    __shared__ double sred[16];
    __shared__ double dred[16];
    double s = 0.0, den = 0.0;
    for (int i = threadIdx.x; i < B * Tt; i += 1024) {
        const int b = i / Tt, t = i % Tt;
        if (t < clamp_len(length[b], Tt)) {
            const float p = sim[(size_t)b * ld_sim + t], y = label[(size_t)b * ld_label + t];
            const float lp = fmaxf(logf(p), -100.0f), lq = fmaxf(logf(1.0f - p), -100.0f);
            s += (double)((y - 1.0f) * lq - y * lp);
        }
    }
    for (int b = threadIdx.x; b < B; b += 1024) den += (double)clamp_len(length[b], Tt);
    s = wave_sum_d(s);
    den = wave_sum_d(den);
    if ((threadIdx.x & 63) == 0) { sred[threadIdx.x >> 6] = s; dred[threadIdx.x >> 6] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double st = 0.0, dt = 0.0;
        for (int w = 0; w < 16; ++w) { st += sred[w]; dt += dred[w]; }
        loss[0] = (float)(st / dt);
    }
}
__global__ __launch_bounds__(256) void frame_bce_bwd_kernel(const float* __restrict__ sim, int ld_sim,
                                                            const float* __restrict__ label, int ld_label,
                                                            const int64_t* __restrict__ length, int B, int Tt,
                                                            const float* __restrict__ dloss, float* __restrict__ dsim) {
    __shared__ double dred[4];
    double den = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) den += (double)clamp_len(length[b], Tt);
    den = wave_sum_d(den);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = den;
    __syncthreads();
    const float k = dloss[0] / (float)(dred[0] + dred[1] + dred[2] + dred[3]);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * ld_sim; i += gridDim.x * 256) {
        const int b = i / ld_sim, t = i % ld_sim;
        float g = 0.0f;
        if (t < Tt && t < clamp_len(length[b], Tt)) {
            const float p = sim[(size_t)b * ld_sim + t], y = label[(size_t)b * ld_label + t];
            g = k * (p - y) / fmaxf((1.0f - p) * p, 1e-12f);
        }
        dsim[(size_t)b * ld_sim + t] = g;
    }
}

// ---------------------------------------------------------------- post-processing (integer work)
__device__ __forceinline__ int refl(int j, int T) {
    // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
    const int period = 2 * T;
    j %= period;
    if (j < 0) j += period;
    return j >= T ? period - 1 - j : j;
}
__global__ __launch_bounds__(64) void segments_kernel(const float* __restrict__ sim, int ld, int B, int T,
                                                      const double* __restrict__ thresholds, int NT, int window,
                                                      int n_connect, int64_t* __restrict__ regions,
                                                      int32_t* __restrict__ counts, int max_regions) {
    const int id = blockIdx.x * 64 + threadIdx.x;
    if (id >= B * NT) return;
    const int b = id / NT, ti = id % NT;
    const float* x = sim + (size_t)b * ld;
    const double th = thresholds[ti];
    int64_t* out = regions + (size_t)id * max_regions * 2;
    const int left = window / 2, need = window - window / 2;   // median of a 0/1 window = [#ones >= size - size/2]
    int n_out = 0;
    bool in_reg = false;
    int reg_start = 0;
    bool have_pending = false;
    int pend_start = 0, pend_end = 0;
    for (int i = 0; i <= T; ++i) {
        bool v = false;
        if (i < T) {
            if (window <= 1) {
                v = (double)x[i] > th;
            } else {
                int ones = 0;
                for (int j = i - left; j < i - left + window; ++j) ones += ((double)x[refl(j, T)] > th) ? 1 : 0;
                v = ones >= need;
            }
        }
        if (v && !in_reg) { in_reg = true; reg_start = i; }
        else if (!v && in_reg) {
            in_reg = false;
            const int s = reg_start, e = i;
            if (have_pending && s - pend_end <= n_connect) {
                pend_end = e;
            } else {
                if (have_pending && n_out < max_regions) { out[2 * n_out] = pend_start; out[2 * n_out + 1] = pend_end; ++n_out; }
                have_pending = true; pend_start = s; pend_end = e;
            }
        }
    }
    if (have_pending && n_out < max_regions) { out[2 * n_out] = pend_start; out[2 * n_out + 1] = pend_end; ++n_out; }
    counts[id] = n_out;
}

// ---------------------------------------------------------------- optimiser
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partials) {
    __shared__ double sred[4];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = g[i]; s += (double)v * v; }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ partials, int nblk, double* out) {
    __shared__ double sred[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sred[0] + sred[1] + sred[2] + sred[3];
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, const double* __restrict__ gnorm_sq,
                                                   float max_norm, float grad_scale) {
    // a non-finite gradient norm (e.g. a GRU exchange that timed out and poisoned its outputs with NaN) must not reach
    // the parameters or the moments: the whole step is skipped (the sticky error word tells the host why)
    if (gnorm_sq && !isfinite(gnorm_sq[0])) return;
    float coef = grad_scale;
    if (max_norm > 0.0f && gnorm_sq) {
        const float norm = (float)sqrt(gnorm_sq[0]) * grad_scale;
        const float c = max_norm / (norm + 1e-6f);
        coef *= c < 1.0f ? c : 1.0f;
    }
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= step_size * (mi / denom);
    }
}
// ---- weak supervision (MultiTextBiEncoder, models/audio_text_model.py:101-229): N phrases per clip ----
// backward of the grouped DotProduct head: one block per CLIP; daudio[b,t] = sum_n ds[n,t] text_n, dtext_n = sum_t ds a_t
constexpr int MAXG = 16;
template <int NG, int ND>      // N <= NG phrases per clip, D <= 64 * ND
__global__ __launch_bounds__(256) void match_group_bwd_kernel(const float* __restrict__ audio,
                                                              const float* __restrict__ text,
                                                              const float* __restrict__ dsim, float* __restrict__ daudio,
                                                              float* __restrict__ dtext, int scale, int T, int D, int N) {
    extern __shared__ float gsm[];                 // text rows [N][D], then the per-wave dtext partials [4][N][D]
    float* txs = gsm;
    float* red = gsm + (size_t)N * D;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < N * D; e += 256) txs[e] = text[(size_t)b * N * D + e];
    __syncthreads();
    const float rs = scale ? 1.0f / sqrtf((float)D) : 1.0f;
    float dt[NG][ND];
#pragma unroll
    for (int n = 0; n < NG; ++n)
#pragma unroll
        for (int i = 0; i < ND; ++i) dt[n][i] = 0.0f;
    for (int t = wid; t < T; t += 4) {
        float a[ND], da[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = lane + 64 * i;
            a[i] = d < D ? audio[((size_t)b * T + t) * D + d] : 0.0f;
            da[i] = 0.0f;
        }
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            if (n < N) {
                float tx[ND], sc = 0.0f;
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const int d = lane + 64 * i;
                    tx[i] = d < D ? txs[(size_t)n * D + d] : 0.0f;
                    sc = fmaf(a[i], tx[i], sc);
                }
                sc = wave_sum(sc) * rs;
                const float p = 1.0f / (1.0f + expf(-sc));
                const float pass = (p >= 1e-7f && p <= 1.0f) ? 1.0f : 0.0f;
                const float ds = dsim[((size_t)b * N + n) * T + t] * pass * p * (1.0f - p) * rs;
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    da[i] = fmaf(ds, tx[i], da[i]);
                    dt[n][i] = fmaf(ds, a[i], dt[n][i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = lane + 64 * i;
            if (d < D) daudio[((size_t)b * T + t) * D + d] = da[i];
        }
    }
#pragma unroll
    for (int n = 0; n < NG; ++n)
        if (n < N)
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int d = lane + 64 * i;
                if (d < D) red[((size_t)wid * N + n) * D + d] = dt[n][i];
            }
    __syncthreads();
    for (int e = threadIdx.x; e < N * D; e += 256)
        dtext[(size_t)b * N * D + e] = (red[e] + red[(size_t)N * D + e]) + (red[(size_t)2 * N * D + e] + red[(size_t)3 * N * D + e]);
}

// linear_softmax_with_lens (models/utils.py:75-76) over rows of frame probabilities: clip[r] = sum_{t<len} f^2 / sum f;
// len of row r = length[r / group]
__global__ __launch_bounds__(256) void linsoftmax_pool_fwd_kernel(const float* __restrict__ fs, const long* __restrict__ length,
                                                                  float* __restrict__ clip, long rows, int T, int group) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int len = (int)min((long)T, length[r / group]);
    float s1 = 0.0f, s2 = 0.0f;
    for (int t = lane; t < len; t += 64) { const float f = fs[r * T + t]; s1 += f; s2 = fmaf(f, f, s2); }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) clip[r] = s2 / s1;
}
__global__ __launch_bounds__(256) void linsoftmax_pool_bwd_kernel(const float* __restrict__ fs, const long* __restrict__ length,
                                                                  const float* __restrict__ dclip, float* __restrict__ dfs,
                                                                  long rows, int T, int group) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int len = (int)min((long)T, length[r / group]);
    float s1 = 0.0f, s2 = 0.0f;
    for (int t = lane; t < len; t += 64) { const float f = fs[r * T + t]; s1 += f; s2 = fmaf(f, f, s2); }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float g = dclip[r], inv = 1.0f / s1;
    for (int t = lane; t < T; t += 64) {
        const float f = fs[r * T + t];
        dfs[r * T + t] = t < len ? g * (2.0f * f * inv - s2 * inv * inv) : 0.0f;
    }
}

// ---- weak supervision, align-by-phrase (models/audio_text_model.py:907-976): sim_pooling.AudioMeanTextMean
// (models/sim_pooling.py:6-22) over the (B,B,T,N) matrix of align.DotProduct, and MaxMarginRankingLoss (losses.py:226-264)
// out[a,b] = mean_{n < tlen[b]} mean_{t < alen[a]} sim[a,b,t,n];  one wave per (a,b)
__global__ __launch_bounds__(256) void meanmean_pool_fwd_kernel(const float* __restrict__ sim, const long* __restrict__ alen,
                                                                const long* __restrict__ tlen, float* __restrict__ out, int B,
                                                                int T, int N) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long)B * B) return;
    const int a = (int)(r / B), b = (int)(r % B);
    const int al = (int)min((long)T, alen[a]), tl = (int)min((long)N, tlen[b]);
    float s = 0.0f;
    for (int e = lane; e < al * N; e += 64) { const int n = e % N; if (n < tl) s += sim[r * T * N + e]; }
    s = wave_sum(s);
    if (lane == 0) out[r] = s / ((float)al * (float)tl);
}
__global__ __launch_bounds__(256) void meanmean_pool_bwd_kernel(const float* __restrict__ dout, const long* __restrict__ alen,
                                                                const long* __restrict__ tlen, float* __restrict__ dsim, int B,
                                                                int T, int N) {
    const long total = (long)B * B * T * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % N);
        const int t = (int)((i / N) % T);
        const long r = i / ((long)N * T);
        const int a = (int)(r / B), b = (int)(r % B);
        const int al = (int)min((long)T, alen[a]), tl = (int)min((long)N, tlen[b]);
        dsim[i] = (t < al && n < tl) ? dout[r] / ((float)al * (float)tl) : 0.0f;
    }
}
// ------------------------------------------------------------------------------------------------------------------
// General similarity pooling (models/utils.py:22-105 *_with_lens; models/sim_pooling.py:6-204, all twelve reducers;
// MultiTextBiEncoder's pooling modes, models/audio_text_model.py:205-215): sim (R, T, N) -- R rows (clip, or (clip,
// caption) pairs), T frames, N tokens/phrases innermost.  Audio-axis reducer over the valid frames t < alen[r / a_div]:
//   0 mean   1 max (first maximum, as torch.max)   2 linear softmax sum f^2 / sum f   3 exp softmax sum softmax(f) f
// then (tmode >= 0) a text-axis reducer over the valid tokens n < tlen[r % t_mod]:
//   0 mean   1 sum   2 max (first)   3 mean + sum         (tmode = -1: no text reduction, out is (R, N))
// One wave per row, lanes over the frames; the backward recomputes the reducers and writes the whole dsim row
// (zeros outside the valid region): no atomics.
// ------------------------------------------------------------------------------------------------------------------
struct SeqPool { float out, s1; int arg; };     // s1: denominator (linear: sum f; exp: sum exp(f - max)); arg: first max
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
__device__ __forceinline__ SeqPool seq_pool(const float* f, int stride, int len, int mode, int lane) {
    SeqPool r{0.0f, 0.0f, 0};
    if (mode == 0) {
        float s = 0.0f;
        for (int t = lane; t < len; t += 64) s += f[(long)t * stride];
        r.out = wave_sum(s) / (float)len;
    } else if (mode == 2) {
        float s1 = 0.0f, s2 = 0.0f;
        for (int t = lane; t < len; t += 64) { const float v = f[(long)t * stride]; s1 += v; s2 = fmaf(v, v, s2); }
        r.s1 = wave_sum(s1);
        r.out = wave_sum(s2) / r.s1;
    } else {
        float m = -3.0e38f; int am = 0x7fffffff;
        for (int t = lane; t < len; t += 64) { const float v = f[(long)t * stride]; if (v > m) { m = v; am = t; } }
        wave_argmax(m, am);
        r.arg = am;
        r.out = m;
        if (mode == 3) {
            float e = 0.0f, ef = 0.0f;
            for (int t = lane; t < len; t += 64) { const float v = f[(long)t * stride]; const float x = expf(v - m); e += x; ef = fmaf(x, v, ef); }
            r.s1 = wave_sum(e);
            r.out = wave_sum(ef) / r.s1;
            r.arg = __float_as_int(m);               // the stabiliser, for the backward
        }
    }
    return r;
}
// d out / d f_t
__device__ __forceinline__ float seq_pool_grad(const SeqPool& r, float v, int t, int len, int mode) {
    if (mode == 0) return 1.0f / (float)len;
    if (mode == 1) return t == r.arg ? 1.0f : 0.0f;
    if (mode == 2) return (2.0f * v - r.out) / r.s1;
    return expf(v - __int_as_float(r.arg)) / r.s1 * (1.0f + v - r.out);
}
__global__ __launch_bounds__(256) void sim_pool_fwd_kernel(const float* __restrict__ sim, const long* __restrict__ alen,
                                                           const long* __restrict__ tlen, float* __restrict__ out, long R,
                                                           int T, int N, int a_div, int t_mod, int amode, int tmode) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int al = (int)min((long)T, alen[r / a_div]);
    const int tl = tmode >= 0 ? (int)min((long)N, tlen[r % t_mod]) : N;
    const float* base = sim + r * T * N;
    float acc = 0.0f, best = -3.0e38f;
    for (int n = 0; n < tl; ++n) {
        const float p = seq_pool(base + n, N, al, amode, lane).out;
        if (tmode < 0) { if (lane == 0) out[r * N + n] = p; }
        else if (tmode == 2) best = fmaxf(best, p);
        else acc += p;
    }
    if (tmode >= 0 && lane == 0)
        out[r] = tmode == 0 ? acc / (float)tl : tmode == 1 ? acc : tmode == 2 ? best : acc + acc / (float)tl;
}
__global__ __launch_bounds__(256) void sim_pool_bwd_kernel(const float* __restrict__ sim, const long* __restrict__ alen,
                                                           const long* __restrict__ tlen, const float* __restrict__ dout,
                                                           float* __restrict__ dsim, long R, int T, int N, int a_div, int t_mod,
                                                           int amode, int tmode) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int al = (int)min((long)T, alen[r / a_div]);
    const int tl = tmode >= 0 ? (int)min((long)N, tlen[r % t_mod]) : N;
    const float* base = sim + r * T * N;
    float* dbase = dsim + r * T * N;
    int argn = 0;
    if (tmode == 2) {                               // first maximum over the tokens
        float best = -3.0e38f;
        for (int n = 0; n < tl; ++n) { const float p = seq_pool(base + n, N, al, amode, lane).out; if (p > best) { best = p; argn = n; } }
    }
    for (int n = 0; n < N; ++n) {
        float dp = 0.0f;
        if (n < tl) {
            if (tmode < 0) dp = dout[r * N + n];
            else if (tmode == 0) dp = dout[r] / (float)tl;
            else if (tmode == 1) dp = dout[r];
            else if (tmode == 2) dp = n == argn ? dout[r] : 0.0f;
            else dp = dout[r] * (1.0f + 1.0f / (float)tl);
        }
        SeqPool sp{0.0f, 0.0f, 0};
        if (n < tl) sp = seq_pool(base + n, N, al, amode, lane);
        for (int t = lane; t < T; t += 64) {
            float g = 0.0f;
            if (n < tl && t < al) g = dp * seq_pool_grad(sp, base[(long)t * N + n], t, al, amode);
            dbase[(long)t * N + n] = g;
        }
    }
}
// ------------------------------------------------------------------------------------------------------------------
// EmbeddingAgg(aggregation="attention") = AttentionPooling (models/text_encoder.py:46-58): score_l = x_l . w + b,
// masked_fill(-1e10) beyond text_len, softmax over the tokens, out = sum_l weight_l x_l.  One wave per phrase.
// backward: dx_l = weight_l dout + ds_l w with ds_l = weight_l (x_l . dout - sum_j weight_j x_j . dout);
// gw (B,D) = sum_l ds_l x_l and gb (B) = sum_l ds_l are per-phrase terms folded by tag_colsum (fixed order).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attnpool_fwd_kernel(const float* __restrict__ x, const long* __restrict__ lens,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ weight, float* __restrict__ out, int B, int L,
                                                           int D) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int len = (int)lens[b];
    Row wr;
    load_row(wr, w, D, lane);
    // scores are recomputed in each of the three passes (L * D FMAs per pass: a few thousand) -- nothing round-trips
    auto score = [&](int l) {
        Row xr;
        load_row(xr, x + ((size_t)b * L + l) * D, D, lane);
        const float sc = dot_rows(xr, wr) + bias[0];
        return l >= len ? -1e10f : sc;
    };
    float mx = -3.0e38f;
    for (int l = 0; l < L; ++l) mx = fmaxf(mx, score(l));
    float den = 0.0f;
    for (int l = 0; l < L; ++l) den += expf(score(l) - mx);
    Row acc;
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) acc.v[i] = 0.0f;
    for (int l = 0; l < L; ++l) {
        const float wt = expf(score(l) - mx) / den;
        Row xr;
        load_row(xr, x + ((size_t)b * L + l) * D, D, lane);
#pragma unroll
        for (int i = 0; i < MAXD_PER_LANE; ++i) acc.v[i] = fmaf(wt, xr.v[i], acc.v[i]);
        if (lane == 0) weight[(size_t)b * L + l] = wt;
    }
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) { const int d = lane + 64 * i; if (d < D) out[(size_t)b * D + d] = acc.v[i]; }
}
__global__ __launch_bounds__(256) void attnpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ weight, const float* __restrict__ dout,
                                                           float* __restrict__ dx, float* __restrict__ gw,
                                                           float* __restrict__ gb, int B, int L, int D) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    Row wr, dr, gacc;
    load_row(wr, w, D, lane);
    load_row(dr, dout + (size_t)b * D, D, lane);
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) gacc.v[i] = 0.0f;
    float dot_all = 0.0f;
    for (int l = 0; l < L; ++l) {
        Row xr;
        load_row(xr, x + ((size_t)b * L + l) * D, D, lane);
        dot_all = fmaf(weight[(size_t)b * L + l], dot_rows(xr, dr), dot_all);
    }
    float gbs = 0.0f;
    for (int l = 0; l < L; ++l) {
        Row xr;
        load_row(xr, x + ((size_t)b * L + l) * D, D, lane);
        const float wt = weight[(size_t)b * L + l];
        const float ds = wt * (dot_rows(xr, dr) - dot_all);          // masked tokens: weight = 0 -> no gradient
        gbs += ds;
#pragma unroll
        for (int i = 0; i < MAXD_PER_LANE; ++i) {
            const int d = lane + 64 * i;
            if (d < D) dx[((size_t)b * L + l) * D + d] = fmaf(wt, dr.v[i], ds * wr.v[i]);
            gacc.v[i] = fmaf(ds, xr.v[i], gacc.v[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXD_PER_LANE; ++i) { const int d = lane + 64 * i; if (d < D) gw[(size_t)b * D + d] = gacc.v[i]; }
    if (lane == 0) gb[b] = gbs;
}

// ------------------------------------------------------------------------------------------------------------------
// BiEncoder(upsample=True) (models/audio_text_model.py:90-97): F.interpolate(frame_sim, T * ratio, mode="linear",
// align_corners=False).  out[i] = (1 - lam) x[i0] + lam x[i1], src = max((i + 0.5) / ratio - 0.5, 0), i0 = floor(src),
// i1 = min(i0 + 1, T - 1).  backward gathers: input t collects the outputs that read it (no atomics).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lin_src(int i, int ratio, int T, int& i0, int& i1, float& lam) {
    float src = ((float)i + 0.5f) / (float)ratio - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < T - 1 ? 1 : 0);
    lam = src - (float)i0;
}
__global__ __launch_bounds__(256) void upsample_lin_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long R,
                                                               int T, int ratio) {
    const long total = R * T * ratio;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / ((long)T * ratio);
        const int i = (int)(e % ((long)T * ratio));
        int i0, i1; float lam;
        lin_src(i, ratio, T, i0, i1, lam);
        out[e] = (1.0f - lam) * x[r * T + i0] + lam * x[r * T + i1];
    }
}
__global__ __launch_bounds__(256) void upsample_lin_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, long R,
                                                               int T, int ratio) {
    const long total = R * T;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / T;
        const int t = (int)(e % T);
        float g = 0.0f;
        const int lo = max(0, (t - 1) * ratio - 1), hi = min(T * ratio - 1, (t + 2) * ratio);
        for (int i = lo; i <= hi; ++i) {
            int i0, i1; float lam;
            lin_src(i, ratio, T, i0, i1, lam);
            const float d = dout[r * T * ratio + i];
            if (i0 == t) g = fmaf(1.0f - lam, d, g);
            if (i1 == t) g = fmaf(lam, d, g);
        }
        dx[e] = g;
    }
}

// loss = mean over i != j of relu(m - (x_ii - x_ij)) and relu(m - (x_ii - lam x_ji))   (fix_norm = True);
// fix_norm = False keeps the diagonal pairs (i,i) too: relu(m) and relu(m - (1 - lam) x_ii), mean over 2 n^2
__global__ __launch_bounds__(256) void maxmargin_fwd_kernel(const float* __restrict__ x, int n, float margin, float lam,
                                                            int fix_norm, float* __restrict__ loss) {
    __shared__ double sred[4];
    double s = 0.0;
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e / n, j = e % n;
        if (i == j && fix_norm) continue;
        const float d = x[i * n + i];
        s += (double)fmaxf(margin - (d - x[i * n + j]), 0.0f) + (double)fmaxf(margin - (d - lam * x[j * n + i]), 0.0f);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        loss[0] = (float)(((sred[0] + sred[1]) + (sred[2] + sred[3])) / (fix_norm ? 2.0 * n * (n - 1) : 2.0 * n * n));
}
// one wave per row i: off-diagonal dx_ij written by the owner of element (i,j); the diagonal collects its row's terms
__global__ __launch_bounds__(256) void maxmargin_bwd_kernel(const float* __restrict__ x, int n, float margin, float lam,
                                                            int fix_norm, const float* __restrict__ dloss,
                                                            float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float c = dloss[0] / (fix_norm ? 2.0f * n * (n - 1) : 2.0f * n * n);
    const float di = x[i * n + i];
    float dd = 0.0f;
    if (!fix_norm && lane == 0 && (margin - (di - lam * di)) > 0.0f) dd = -c * (1.0f - lam);   // pair (i,i), second term
    for (int j = lane; j < n; j += 64) {
        if (j == i) continue;
        const float xij = x[i * n + j], xji = x[j * n + i], dj = x[j * n + j];
        const float a1 = (margin - (di - xij)) > 0.0f ? 1.0f : 0.0f;          // term 1 of pair (i,j): depends on x_ii, x_ij
        const float a2 = (margin - (di - lam * xji)) > 0.0f ? 1.0f : 0.0f;    // term 2 of pair (i,j): depends on x_ii, x_ji
        const float b2 = (margin - (dj - lam * xij)) > 0.0f ? 1.0f : 0.0f;    // term 2 of pair (j,i): depends on x_jj, x_ij
        dx[i * n + j] = c * (a1 + lam * b2);
        dd -= c * (a1 + a2);
    }
    dd = wave_sum(dd);
    if (lane == 0) dx[i * n + i] = dd;
}

int sumsq_blocks(long n) {
    long nb = (n + 256 * 16 - 1) / (256 * 16);
    return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}


// ---------------------------------------------------------------- MultiTextBiEncoder with a cross-encoder: audio rows x N phrases
// out[(b*N + n)][j] = x[b][j] (the reference's unsqueeze(1).expand(-1, N, ...).reshape, models/audio_text_model.py:165-168)
__global__ __launch_bounds__(256) void group_expand_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long B, int N,
                                                               long R) {
    const long total = B * N * R;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long j = i % R, b = i / (R * N);
        out[i] = x[b * R + j];
    }
}
// dx[b][j] = sum over n (ascending: deterministic) of dout[(b*N + n)][j]
__global__ __launch_bounds__(256) void group_expand_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, long B, int N,
                                                               long R) {
    const long total = B * R;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long j = i % R, b = i / R;
        float s = 0.0f;
        for (int n = 0; n < N; ++n) s += dout[(b * N + n) * R + j];
        dx[i] = s;
    }
}
}  // namespace

extern "C" int tag_embed_mean_forward(const int64_t* text, const int64_t* text_len, const float* table,
                                      float* token_emb, float* seq_emb, int B, int L, int D, int V, void* stream) {
    TAG_CHECK_ARG(text && text_len && table && seq_emb && B > 0 && L > 0 && D > 0 && V > 0);
    hipLaunchKernelGGL(embed_mean_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), text, text_len, table,
                       token_emb, seq_emb, L, D, V);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_embed_mean_backward(const float* dseq, const int64_t* text, const int64_t* text_len, float* dtable,
                                       int B, int L, int D, int V, void* stream) {
    TAG_CHECK_ARG(dseq && text && text_len && dtable && B > 0 && L > 0 && D > 0 && V > 0);
    hipLaunchKernelGGL(embed_bwd_det_kernel, dim3(B * L), dim3(256), 0, as_stream(stream), dseq, (const float*)nullptr, text,
                       text_len, dtable, B * L, L, D, V);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_embed_tokens_backward(const float* dtok, const long* text, float* dtable, int B, int L, int D, int V,
                                         void* stream) {
    TAG_CHECK_ARG(dtok && text && dtable && B > 0 && L > 0 && D > 0 && V > 0);
    hipLaunchKernelGGL(embed_bwd_det_kernel, dim3(B * L), dim3(256), 0, as_stream(stream), (const float*)nullptr, dtok,
                       reinterpret_cast<const int64_t*>(text), (const int64_t*)nullptr, dtable, B * L, L, D, V);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_embed_check_ids(const int64_t* text, long n, int V, int* err_flag, void* stream) {
    TAG_CHECK_ARG(text && err_flag && n > 0 && V > 0);
    hipLaunchKernelGGL(embed_check_ids_kernel, dim3(cdiv(n, 256) > 64 ? 64 : cdiv(n, 256)), dim3(256), 0, as_stream(stream),
                       text, n, V, err_flag);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_match_forward(const float* audio, const float* text, float* sim, int kind, int l2norm, int scale,
                                 int B, int T, int D, void* stream) {
    TAG_CHECK_ARG(audio && text && sim && B > 0 && T > 0 && D > 0 && D <= 64 * MAXD_PER_LANE);
    TAG_CHECK_ARG(kind == 0 || kind == 1);
    int chunks = 1;                              // enough workgroups to fill the chip: ~4 frames per wave
    while (chunks < 16 && (long)B * chunks < 1024 && chunks * 16 < T) chunks *= 2;
    hipLaunchKernelGGL(match_fwd_kernel, dim3(B, chunks), dim3(256), 0, as_stream(stream), audio, text, sim, kind, l2norm,
                       scale, T, D, 1);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_match_backward(const float* audio, const float* text, const float* sim, const float* dsim,
                                  float* daudio, float* dtext, int kind, int l2norm, int scale, int B, int T, int D,
                                  void* stream) {
    (void)sim;
    TAG_CHECK_ARG(audio && text && dsim && daudio && dtext && B > 0 && T > 0 && D > 0 && D <= 64 * MAXD_PER_LANE);
    TAG_CHECK_ARG(kind == 0 || kind == 1);
    if (T >= 32)
        hipLaunchKernelGGL(match_bwd_kernel<8>, dim3(B), dim3(512), 0, as_stream(stream), audio, text, dsim, daudio, dtext,
                           kind, l2norm, scale, T, D);
    else
        hipLaunchKernelGGL(match_bwd_kernel<4>, dim3(B), dim3(256), 0, as_stream(stream), audio, text, dsim, daudio, dtext,
                           kind, l2norm, scale, T, D);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_frame_bce_forward(const float* sim, int ld_sim, const float* label, int ld_label,
                                     const int64_t* length, int B, int Tt, float* loss, void* stream) {
    TAG_CHECK_ARG(sim && label && length && loss && B > 0 && Tt > 0 && ld_sim >= Tt && ld_label >= Tt);
    hipLaunchKernelGGL(frame_bce_fwd_kernel, dim3(1), dim3(1024), 0, as_stream(stream), sim, ld_sim, label, ld_label,
                       length, B, Tt, loss);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_frame_bce_backward(const float* sim, int ld_sim, const float* label, int ld_label,
                                      const int64_t* length, int B, int Tt, const float* dloss, float* dsim,
                                      void* stream) {
    TAG_CHECK_ARG(sim && label && length && dloss && dsim && B > 0 && Tt > 0 && ld_sim >= Tt && ld_label >= Tt);
    hipLaunchKernelGGL(frame_bce_bwd_kernel, dim3(cdiv((long)B * ld_sim, 256)), dim3(256), 0, as_stream(stream), sim,
                       ld_sim, label, ld_label, length, B, Tt, dloss, dsim);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_segments(const float* sim, int ld, int B, int T, const double* thresholds, int NT, int window,
                            int n_connect, int64_t* regions, int32_t* counts, int max_regions, void* stream) {
    TAG_CHECK_ARG(sim && thresholds && regions && counts && B > 0 && T > 0 && NT > 0 && ld >= T);
    TAG_CHECK_ARG(window >= 1 && n_connect >= 0 && max_regions >= (T + 1) / 2);
    hipLaunchKernelGGL(segments_kernel, dim3(cdiv((long)B * NT, 64)), dim3(64), 0, as_stream(stream), sim, ld, B, T,
                       thresholds, NT, window, n_connect, regions, counts, max_regions);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t tag_sumsq_ws_bytes(long n) { return (size_t)sumsq_blocks(n) * sizeof(double); }
extern "C" int tag_sumsq(const float* g, long n, double* out, void* ws, void* stream) {
    TAG_CHECK_ARG(g && out && ws && n > 0);
    const int nblk = sumsq_blocks(n);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), g, n,
                       static_cast<double*>(ws));
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), static_cast<double*>(ws), nblk,
                       out);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, int step, const double* gnorm_sq, float max_norm, float grad_scale,
                             void* stream) {
    TAG_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((int)nb), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2,
                       eps, bc1, bc2_sqrt, gnorm_sq, max_norm, grad_scale);
    TAG_LAUNCH_CHECK();
    return 0;
}

/* ---- weak supervision: grouped heads ---- */
extern "C" int tag_match_group_forward(const float* audio, const float* text, float* sim, int scale, int B, int N, int T,
                                       int D, void* stream) {
    TAG_CHECK_ARG(audio && text && sim && B > 0 && N > 0 && T > 0 && D > 0 && D <= 64 * MAXD_PER_LANE);
    hipLaunchKernelGGL(match_fwd_kernel, dim3(B * N), dim3(256), 0, as_stream(stream), audio, text, sim, 0, 0, scale, T, D,
                       N);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_match_group_backward(const float* audio, const float* text, const float* dsim, float* daudio,
                                        float* dtext, int scale, int B, int N, int T, int D, void* stream) {
    TAG_CHECK_ARG(audio && text && dsim && daudio && dtext && B > 0 && N > 0 && N <= MAXG && T > 0 && D > 0);
    TAG_CHECK_ARG(D <= 64 * MAXD_PER_LANE);
    const size_t lds = (size_t)5 * N * D * sizeof(float);
    TAG_CHECK_ARG(lds <= 160 * 1024);
#define LAUNCH(NG, ND)                                                                                               \
    {                                                                                                                \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&match_group_bwd_kernel<NG, ND>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                       \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((match_group_bwd_kernel<NG, ND>), dim3(B), dim3(256), lds, as_stream(stream), audio, text, \
                           dsim, daudio, dtext, scale, T, D, N);                                                     \
    }
    const bool small_d = D <= 512;
    if (N <= 4) { if (small_d) LAUNCH(4, 8) else LAUNCH(4, 16) }
    else if (N <= 8) { if (small_d) LAUNCH(8, 8) else LAUNCH(8, 16) }
    else { if (small_d) LAUNCH(16, 8) else LAUNCH(16, 16) }
#undef LAUNCH
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_linear_softmax_pool_forward(const float* fs, const long* length, float* clip, long rows, int T,
                                               int group, void* stream) {
    TAG_CHECK_ARG(fs && length && clip && rows > 0 && T > 0 && group > 0);
    hipLaunchKernelGGL(linsoftmax_pool_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), fs, length, clip,
                       rows, T, group);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_linear_softmax_pool_backward(const float* fs, const long* length, const float* dclip, float* dfs,
                                                long rows, int T, int group, void* stream) {
    TAG_CHECK_ARG(fs && length && dclip && dfs && rows > 0 && T > 0 && group > 0);
    hipLaunchKernelGGL(linsoftmax_pool_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), fs, length, dclip,
                       dfs, rows, T, group);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_meanmean_pool_forward(const float* sim, const long* alen, const long* tlen, float* out, int B, int T,
                                         int N, void* stream) {
    TAG_CHECK_ARG(sim && alen && tlen && out && B > 0 && T > 0 && N > 0);
    hipLaunchKernelGGL(meanmean_pool_fwd_kernel, dim3(cdiv((long)B * B, 4)), dim3(256), 0, as_stream(stream), sim, alen, tlen,
                       out, B, T, N);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_meanmean_pool_backward(const float* dout, const long* alen, const long* tlen, float* dsim, int B, int T,
                                          int N, void* stream) {
    TAG_CHECK_ARG(dout && alen && tlen && dsim && B > 0 && T > 0 && N > 0);
    const long total = (long)B * B * T * N;
    hipLaunchKernelGGL(meanmean_pool_bwd_kernel, dim3(cdiv(total, 256) > 8192 ? 8192 : cdiv(total, 256)), dim3(256), 0,
                       as_stream(stream), dout, alen, tlen, dsim, B, T, N);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_attnpool_forward(const float* x, const long* lens, const float* w, const float* bias, float* weight,
                                    float* out, int B, int L, int D, void* stream) {
    TAG_CHECK_ARG(x && lens && w && bias && weight && out && B > 0 && L > 0 && D > 0 && D <= 64 * MAXD_PER_LANE);
    hipLaunchKernelGGL(attnpool_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), x, lens, w, bias, weight, out,
                       B, L, D);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_attnpool_backward(const float* x, const float* w, const float* weight, const float* dout, float* dx,
                                     float* gw, float* gb, int B, int L, int D, void* stream) {
    TAG_CHECK_ARG(x && w && weight && dout && dx && gw && gb && B > 0 && L > 0 && D > 0 && D <= 64 * MAXD_PER_LANE);
    hipLaunchKernelGGL(attnpool_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), x, w, weight, dout, dx, gw, gb,
                       B, L, D);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_upsample_linear_forward(const float* x, float* out, long R, int T, int ratio, void* stream) {
    TAG_CHECK_ARG(x && out && R > 0 && T > 0 && ratio >= 1);
    const long n = R * T * ratio;
    hipLaunchKernelGGL(upsample_lin_fwd_kernel, dim3(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256)), dim3(256), 0,
                       as_stream(stream), x, out, R, T, ratio);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_upsample_linear_backward(const float* dout, float* dx, long R, int T, int ratio, void* stream) {
    TAG_CHECK_ARG(dout && dx && R > 0 && T > 0 && ratio >= 1);
    const long n = R * T;
    hipLaunchKernelGGL(upsample_lin_bwd_kernel, dim3(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256)), dim3(256), 0,
                       as_stream(stream), dout, dx, R, T, ratio);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_group_expand_forward(const float* x, float* out, long B, int N, long R, void* stream) {
    TAG_CHECK_ARG(x && out && B > 0 && N > 0 && R > 0);
    const long n = B * N * R;
    hipLaunchKernelGGL(group_expand_fwd_kernel, dim3(cdiv(n, 256) > 8192 ? 8192 : cdiv(n, 256)), dim3(256), 0, as_stream(stream),
                       x, out, B, N, R);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_group_expand_backward(const float* dout, float* dx, long B, int N, long R, void* stream) {
    TAG_CHECK_ARG(dout && dx && B > 0 && N > 0 && R > 0);
    const long n = B * R;
    hipLaunchKernelGGL(group_expand_bwd_kernel, dim3(cdiv(n, 256) > 8192 ? 8192 : cdiv(n, 256)), dim3(256), 0, as_stream(stream),
                       dout, dx, B, N, R);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_sim_pool_forward(const float* sim, const long* alen, const long* tlen, float* out, long R, int T, int N,
                                    int a_div, int t_mod, int amode, int tmode, void* stream) {
    TAG_CHECK_ARG(sim && alen && out && R > 0 && T > 0 && N > 0 && a_div > 0 && amode >= 0 && amode <= 3);
    TAG_CHECK_ARG(tmode >= -1 && tmode <= 3 && (tmode < 0 || (tlen && t_mod > 0)));
    hipLaunchKernelGGL(sim_pool_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, as_stream(stream), sim, alen, tlen, out, R, T, N,
                       a_div, t_mod > 0 ? t_mod : 1, amode, tmode);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_sim_pool_backward(const float* sim, const long* alen, const long* tlen, const float* dout, float* dsim,
                                     long R, int T, int N, int a_div, int t_mod, int amode, int tmode, void* stream) {
    TAG_CHECK_ARG(sim && alen && dout && dsim && R > 0 && T > 0 && N > 0 && a_div > 0 && amode >= 0 && amode <= 3);
    TAG_CHECK_ARG(tmode >= -1 && tmode <= 3 && (tmode < 0 || (tlen && t_mod > 0)));
    hipLaunchKernelGGL(sim_pool_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, as_stream(stream), sim, alen, tlen, dout, dsim, R,
                       T, N, a_div, t_mod > 0 ? t_mod : 1, amode, tmode);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_maxmargin_forward(const float* x, int n, float margin, float lamda1, int fix_norm, float* loss,
                                     void* stream) {
    TAG_CHECK_ARG(x && loss && n > 1);
    hipLaunchKernelGGL(maxmargin_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, n, margin, lamda1, fix_norm, loss);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_maxmargin_backward(const float* x, int n, float margin, float lamda1, int fix_norm, const float* dloss,
                                      float* dx, void* stream) {
    TAG_CHECK_ARG(x && dloss && dx && n > 1);
    hipLaunchKernelGGL(maxmargin_bwd_kernel, dim3(cdiv(n, 4)), dim3(256), 0, as_stream(stream), x, n, margin, lamda1, fix_norm,
                       dloss, dx);
    TAG_LAUNCH_CHECK();
    return 0;
}
