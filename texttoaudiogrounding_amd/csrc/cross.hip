// BASELINE configs[3]: the cross-encoder (models/cross_encoder.py:5-79 in the reference) and the token-level DotProduct
// head it feeds (models/match.py:43-60 with text_level="token"), forward and backward.
//
// Seq2SeqAttention scores every (frame q, token k) pair with v . tanh(W [query_q ; kv_k] + b).  The reference
// materialises the (B, T*L, 2D) concatenation; here W = [Wq | Wk] is applied as two GEMMs (tag_gemm: aq = query Wq^T,
// ak = kv Wk^T + b) and these kernels do the rest per frame: broadcast-add + tanh + dot with v, the two -1e10 mask fills,
// softmax over the tokens and attn @ kv -- one wave per (clip, frame), the D axis spread over the lanes.
// Backward recomputes tanh; a wave owns a tile of 16 frames so that the token-side gradients (d ak, d kv, d v), which
// are sums over frames, leave as per-tile partials that a second kernel folds in a fixed order (no atomics).
#include "tag_common.h"

namespace {

constexpr int MAXL = 32;      // tokens per phrase
constexpr int QT = 8;         // frames per backward tile
constexpr float FILL = -1e10f;

// attn (B,T,L) softmax weights, ctx (B,T,Dk) = attn @ kv.  LM = compile-time bound on L (scores stay in registers)
template <int LM>
__global__ __launch_bounds__(256) void addattn_fwd_kernel(const float* __restrict__ aq, const float* __restrict__ ak,
                                                          const float* __restrict__ v, const float* __restrict__ kv,
                                                          const long* __restrict__ qlen, const long* __restrict__ klen,
                                                          float* __restrict__ attn, float* __restrict__ ctx, int B, int T,
                                                          int L, int Da, int Dk) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * T) return;
    const int b = (int)(row / T), q = (int)(row % T);
    const bool qok = q < qlen[b];
    const int kl = (int)klen[b];
    float sc[LM];                               // fully unrolled below: stays in registers
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < LM; ++k) {
        sc[k] = 0.0f;
        if (k < L) {
            float s = 0.0f;
            for (int d = lane; d < Da; d += 64) s = fmaf(v[d], tanhf(aq[row * Da + d] + ak[((long)b * L + k) * Da + d]), s);
            s = wave_sum(s);
            if (!qok || k >= kl) s = FILL;
            sc[k] = s;
            mx = fmaxf(mx, s);
        }
    }
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < LM; ++k)
        if (k < L) { sc[k] = expf(sc[k] - mx); den += sc[k]; }
    const float inv = 1.0f / den;
#pragma unroll
    for (int k = 0; k < LM; ++k)
        if (k < L) {
            sc[k] *= inv;
            if (lane == 0) attn[row * L + k] = sc[k];
        }
    for (int d = lane; d < Dk; d += 64) {
        float o = 0.0f;
#pragma unroll
        for (int k = 0; k < LM; ++k)
            if (k < L) o = fmaf(sc[k], kv[((long)b * L + k) * Dk + d], o);
        ctx[row * Dk + d] = o;
    }
}

// One wave per (clip, tile of QT frames).  daq (B,T,Da) is final; dak_p (B,NT,L,Da), dkv_p (B,NT,L,Dk), dv_p (B,NT,Da)
// are per-tile partial sums over the tile's frames.
template <int NDA, int NDK>     // Da <= 64*NDA, Dk <= 64*NDK
__global__ __launch_bounds__(64) void addattn_bwd_kernel(const float* __restrict__ aq, const float* __restrict__ ak,
                                                         const float* __restrict__ v, const float* __restrict__ kv,
                                                         const float* __restrict__ attn, const float* __restrict__ dctx,
                                                         const long* __restrict__ qlen, const long* __restrict__ klen,
                                                         float* __restrict__ daq, float* __restrict__ dak_p,
                                                         float* __restrict__ dkv_p, float* __restrict__ dv_p, int B, int T,
                                                         int L, int Da, int Dk, int NT) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x / NT, tile = blockIdx.x % NT;
    const int q0 = tile * QT, q1 = min(T, q0 + QT);
    const int kl = (int)klen[b], ql = (int)qlen[b];
    float vv[NDA], dvacc[NDA];
#pragma unroll
    for (int i = 0; i < NDA; ++i) { const int d = lane + 64 * i; vv[i] = d < Da ? v[d] : 0.0f; dvacc[i] = 0.0f; }
    float* dak_out = dak_p + ((size_t)b * NT + tile) * L * Da;
    float* dkv_out = dkv_p + ((size_t)b * NT + tile) * L * Dk;
    for (int k = 0; k < L; ++k) {
        float akv[NDA], kvv[NDK], dakacc[NDA], dkvacc[NDK];
#pragma unroll
        for (int i = 0; i < NDA; ++i) {
            const int d = lane + 64 * i;
            akv[i] = d < Da ? ak[((size_t)b * L + k) * Da + d] : 0.0f;
            dakacc[i] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NDK; ++i) {
            const int d = lane + 64 * i;
            kvv[i] = d < Dk ? kv[((size_t)b * L + k) * Dk + d] : 0.0f;
            dkvacc[i] = 0.0f;
        }
        for (int q = q0; q < q1; ++q) {
            const size_t row = (size_t)b * T + q;
            // d attn[q][j] for every token j (needed by the softmax backward of token k)
            float dcx[NDK];
#pragma unroll
            for (int i = 0; i < NDK; ++i) { const int d = lane + 64 * i; dcx[i] = d < Dk ? dctx[row * Dk + d] : 0.0f; }
            float dot_all = 0.0f, da_k = 0.0f;
            for (int j = 0; j < L; ++j) {
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < NDK; ++i) {
                    const int d = lane + 64 * i;
                    s = fmaf(dcx[i], d < Dk ? kv[((size_t)b * L + j) * Dk + d] : 0.0f, s);
                }
                s = wave_sum(s);
                dot_all = fmaf(attn[row * L + j], s, dot_all);
                if (j == k) da_k = s;
            }
            const float a_k = attn[row * L + k];
            // masked_fill blocks the gradient of filled scores
            const float dscore = (q < ql && k < kl) ? a_k * (da_k - dot_all) : 0.0f;
#pragma unroll
            for (int i = 0; i < NDA; ++i) {
                const int d = lane + 64 * i;
                if (d < Da) {
                    const float th = tanhf(aq[row * Da + d] + akv[i]);
                    const float gpre = dscore * vv[i] * (1.0f - th * th);
                    dakacc[i] += gpre;
                    dvacc[i] = fmaf(dscore, th, dvacc[i]);
                    // daq accumulates over the tokens: k = 0 writes, later tokens add (same wave, program order)
                    if (k == 0) daq[row * Da + d] = gpre;
                    else daq[row * Da + d] += gpre;
                }
            }
#pragma unroll
            for (int i = 0; i < NDK; ++i) dkvacc[i] = fmaf(a_k, dcx[i], dkvacc[i]);
        }
#pragma unroll
        for (int i = 0; i < NDA; ++i) { const int d = lane + 64 * i; if (d < Da) dak_out[(size_t)k * Da + d] = dakacc[i]; }
#pragma unroll
        for (int i = 0; i < NDK; ++i) { const int d = lane + 64 * i; if (d < Dk) dkv_out[(size_t)k * Dk + d] = dkvacc[i]; }
    }
#pragma unroll
    for (int i = 0; i < NDA; ++i) { const int d = lane + 64 * i; if (d < Da) dv_p[((size_t)b * NT + tile) * Da + d] = dvacc[i]; }
}

// out[o][i] = sum_{t < NT} part[o][t][i]   (inner = L*D elements)
__global__ __launch_bounds__(256) void fold_tiles_kernel(const float* __restrict__ part, float* __restrict__ out, long outer,
                                                         int NT, long inner) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < outer * inner; e += (long)gridDim.x * 256) {
        const long o = e / inner, i = e % inner;
        float s = 0.0f;
        for (int t = 0; t < NT; ++t) s += part[(o * NT + t) * inner + i];
        out[e] = s;
    }
}
// dv[d] = sum over all (b, tile) of dv_p, fp64, fixed order
__global__ __launch_bounds__(256) void fold_dv_kernel(const float* __restrict__ part, float* __restrict__ dv, long n, int Da) {
    // block = 16 columns x 16 row groups; rows folded in fp64 in a fixed order
    __shared__ double sh[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4, d = blockIdx.x * 16 + c;
    double s = 0.0;
    if (d < Da)
        for (long r = g; r < n; r += 16) s += (double)part[r * Da + d];
    sh[g][c] = s;
    __syncthreads();
    if (g == 0 && d < Da) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += sh[q][c];
        dv[d] = (float)t;
    }
}

// ---- gating and token-level dot product: rows of D spread over a wave ----
// out = x * g (elementwise); backward of out = x * g with g = sigmoid(z): dx (+)= dout * g, dz = dout * x * g (1 - g)
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
        reinterpret_cast<f32x4*>(out)[i] = x * y;
    }
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                       const float* __restrict__ g, float* __restrict__ dx, int accumulate,
                                                       float* __restrict__ dz, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 d = reinterpret_cast<const f32x4*>(dout)[i], xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 r = d * gv;
        if (accumulate) r += reinterpret_cast<const f32x4*>(dx)[i];
        reinterpret_cast<f32x4*>(dx)[i] = r;
        reinterpret_cast<f32x4*>(dz)[i] = d * xv * gv * (1.0f - gv);
    }
}
// sim[r] = sigmoid(sum_d a[r,d] b[r,d] * scale).clamp(1e-7, 1)
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ sim, long rows, int D, float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.0f;
    for (int d = lane; d < D; d += 64) s = fmaf(a[r * D + d], b[r * D + d], s);
    s = wave_sum(s) * scale;
    const float p = 1.0f / (1.0f + expf(-s));
    if (lane == 0) sim[r] = fminf(fmaxf(p, 1e-7f), 1.0f);
}
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ dsim, float* __restrict__ da,
                                                         float* __restrict__ db, long rows, int D, float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.0f;
    for (int d = lane; d < D; d += 64) s = fmaf(a[r * D + d], b[r * D + d], s);
    s = wave_sum(s) * scale;
    const float p = 1.0f / (1.0f + expf(-s));
    const float pass = (p >= 1e-7f && p <= 1.0f) ? 1.0f : 0.0f;           // clamp backward
    const float ds = dsim[r] * pass * p * (1.0f - p) * scale;
    for (int d = lane; d < D; d += 64) {
        da[r * D + d] = ds * b[r * D + d];
        db[r * D + d] = ds * a[r * D + d];
    }
}

// Token-level heads after a cross-encoder (models/match.py:16-33, 43-60 with text_level = "token"): one text vector per
// frame, rows r = (clip, frame).  kind 0: sigmoid(u.w * scale).clamp(1e-7, 1); kind 1: exp(-||u - w||); u, w = the rows,
// L2-normalised (F.normalize, eps 1e-12) when l2norm.  One wave per row; everything follows from aa, bb, ab.
struct RowPair {
    float ia, ib, uu, ww, uw;
};
__device__ __forceinline__ RowPair rowpair_sums(const float* a, const float* b, int D, int lane, int l2norm) {
    float aa = 0.0f, bb = 0.0f, ab = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float x = a[d], y = b[d];
        aa = fmaf(x, x, aa);
        bb = fmaf(y, y, bb);
        ab = fmaf(x, y, ab);
    }
    aa = wave_sum(aa);
    bb = wave_sum(bb);
    ab = wave_sum(ab);
    RowPair r;
    r.ia = l2norm ? 1.0f / fmaxf(sqrtf(aa), 1e-12f) : 1.0f;
    r.ib = l2norm ? 1.0f / fmaxf(sqrtf(bb), 1e-12f) : 1.0f;
    r.uu = aa * r.ia * r.ia;
    r.ww = bb * r.ib * r.ib;
    r.uw = ab * r.ia * r.ib;
    return r;
}
__global__ __launch_bounds__(256) void rowpair_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ sim, long rows, int D, int kind, int l2norm,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const RowPair q = rowpair_sums(a + r * D, b + r * D, D, lane, l2norm);
    float out;
    if (kind == 0) {
        const float p = 1.0f / (1.0f + expf(-q.uw * scale));
        out = fminf(fmaxf(p, 1e-7f), 1.0f);
    } else {
        float d2 = 0.0f;                                    // sum of squared differences directly (no cancellation)
        for (int d = lane; d < D; d += 64) { const float df = a[r * D + d] * q.ia - b[r * D + d] * q.ib; d2 = fmaf(df, df, d2); }
        out = expf(-sqrtf(wave_sum(d2)));
    }
    if (lane == 0) sim[r] = out;
}
__global__ __launch_bounds__(256) void rowpair_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ dsim, float* __restrict__ da,
                                                          float* __restrict__ db, long rows, int D, int kind, int l2norm,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* ar = a + r * D;
    const float* br = b + r * D;
    const RowPair q = rowpair_sums(ar, br, D, lane, l2norm);
    const float g = dsim[r];
    // du = cu_u * u + cu_w * w, dw = cw_u * u + cw_w * w
    float cu_u, cu_w, cw_u, cw_w;
    if (kind == 0) {
        const float p = 1.0f / (1.0f + expf(-q.uw * scale));
        const float pass = (p >= 1e-7f && p <= 1.0f) ? 1.0f : 0.0f;
        const float ds = g * pass * p * (1.0f - p) * scale;
        cu_u = 0.0f; cu_w = ds; cw_u = ds; cw_w = 0.0f;
    } else {
        float d2 = 0.0f;
        for (int d = lane; d < D; d += 64) { const float df = ar[d] * q.ia - br[d] * q.ib; d2 = fmaf(df, df, d2); }
        const float rr = sqrtf(wave_sum(d2));
        const float k = rr > 0.0f ? -g * expf(-rr) / rr : 0.0f;
        cu_u = k; cu_w = -k; cw_u = -k; cw_w = k;
    }
    // F.normalize backward: dx = (du - u (u.du)) / ||x||
    const float u_du = cu_u * q.uu + cu_w * q.uw, w_dw = cw_u * q.uw + cw_w * q.ww;
    for (int d = lane; d < D; d += 64) {
        const float u = ar[d] * q.ia, w = br[d] * q.ib;
        float du = cu_u * u + cu_w * w, dw = cw_u * u + cw_w * w;
        if (l2norm) { du = (du - u * u_du) * q.ia; dw = (dw - w * w_dw) * q.ib; }
        da[r * D + d] = du;
        db[r * D + d] = dw;
    }
}
}  // namespace

extern "C" int tag_addattn_forward(const float* aq, const float* ak, const float* v, const float* kv, const long* qlen,
                                   const long* klen, float* attn, float* ctx, int B, int T, int L, int Da, int Dk,
                                   void* stream) {
    TAG_CHECK_ARG(aq && ak && v && kv && qlen && klen && attn && ctx && B > 0 && T > 0 && L > 0 && L <= MAXL);
    TAG_CHECK_ARG(Da > 0 && Dk > 0);
#define LAUNCH(LM)                                                                                                    \
    hipLaunchKernelGGL(addattn_fwd_kernel<LM>, dim3(cdiv((long)B * T, 4)), dim3(256), 0, as_stream(stream), aq, ak, v, kv, \
                       qlen, klen, attn, ctx, B, T, L, Da, Dk);
    if (L <= 4) { LAUNCH(4) } else if (L <= 8) { LAUNCH(8) } else if (L <= 16) { LAUNCH(16) } else { LAUNCH(32) }
#undef LAUNCH
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t tag_addattn_backward_ws_bytes(int B, int T, int L, int Da, int Dk) {
    const size_t NT = (size_t)cdiv(T, QT);
    return (size_t)B * NT * ((size_t)L * Da + (size_t)L * Dk + Da) * sizeof(float);
}

extern "C" int tag_addattn_backward(const float* aq, const float* ak, const float* v, const float* kv, const float* attn,
                                    const float* dctx, const long* qlen, const long* klen, float* daq, float* dak,
                                    float* dkv, float* dv, int B, int T, int L, int Da, int Dk, void* ws, void* stream) {
    TAG_CHECK_ARG(aq && ak && v && kv && attn && dctx && qlen && klen && daq && dak && dkv && dv && ws);
    TAG_CHECK_ARG(B > 0 && T > 0 && L > 0 && L <= MAXL && Da > 0 && Dk > 0 && Da <= 1024 && Dk <= 1024);
    const int NT = cdiv(T, QT);
    float* dak_p = static_cast<float*>(ws);
    float* dkv_p = dak_p + (size_t)B * NT * L * Da;
    float* dv_p = dkv_p + (size_t)B * NT * L * Dk;
    hipStream_t st = as_stream(stream);
#define LAUNCH(NDA, NDK)                                                                                              \
    hipLaunchKernelGGL((addattn_bwd_kernel<NDA, NDK>), dim3(B * NT), dim3(64), 0, st, aq, ak, v, kv, attn, dctx, qlen, klen, \
                       daq, dak_p, dkv_p, dv_p, B, T, L, Da, Dk, NT);
    const int nda = cdiv(Da, 64), ndk = cdiv(Dk, 64);
    if (nda <= 1 && ndk <= 1) { LAUNCH(1, 1) }
    else if (nda <= 4 && ndk <= 4) { LAUNCH(4, 4) }
    else if (nda <= 8 && ndk <= 8) { LAUNCH(8, 8) }
    else { LAUNCH(16, 16) }
#undef LAUNCH
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(fold_tiles_kernel, dim3(cdiv((long)B * L * Da, 256)), dim3(256), 0, st, dak_p, dak, (long)B, NT,
                       (long)L * Da);
    hipLaunchKernelGGL(fold_tiles_kernel, dim3(cdiv((long)B * L * Dk, 256)), dim3(256), 0, st, dkv_p, dkv, (long)B, NT,
                       (long)L * Dk);
    hipLaunchKernelGGL(fold_dv_kernel, dim3(cdiv(Da, 16)), dim3(256), 0, st, dv_p, dv, (long)B * NT, Da);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_mul(const float* a, const float* b, float* out, long n, void* stream) {
    TAG_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0);
    const long n4 = n / 4;
    hipLaunchKernelGGL(mul_kernel, dim3(cdiv(n4, 256) > 4096 ? 4096 : cdiv(n4, 256)), dim3(256), 0, as_stream(stream), a, b,
                       out, n4);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_gate_backward(const float* dout, const float* x, const float* g, float* dx, int accumulate, float* dz,
                                 long n, void* stream) {
    TAG_CHECK_ARG(dout && x && g && dx && dz && n > 0 && n % 4 == 0);
    const long n4 = n / 4;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(cdiv(n4, 256) > 4096 ? 4096 : cdiv(n4, 256)), dim3(256), 0, as_stream(stream),
                       dout, x, g, dx, accumulate, dz, n4);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_rowdot_sigmoid_forward(const float* a, const float* b, float* sim, long rows, int D, int scale,
                                          void* stream) {
    TAG_CHECK_ARG(a && b && sim && rows > 0 && D > 0);
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), a, b, sim, rows, D,
                       scale ? 1.0f / sqrtf((float)D) : 1.0f);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_rowdot_sigmoid_backward(const float* a, const float* b, const float* dsim, float* da, float* db,
                                           long rows, int D, int scale, void* stream) {
    TAG_CHECK_ARG(a && b && dsim && da && db && rows > 0 && D > 0);
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), a, b, dsim, da, db, rows, D,
                       scale ? 1.0f / sqrtf((float)D) : 1.0f);
    TAG_LAUNCH_CHECK();
    return 0;
}


extern "C" int tag_rowpair_forward(const float* a, const float* b, float* sim, long rows, int D, int kind, int l2norm,
                                   int scale, void* stream) {
    TAG_CHECK_ARG(a && b && sim && rows > 0 && D > 0 && (kind == 0 || kind == 1));
    hipLaunchKernelGGL(rowpair_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), a, b, sim, rows, D, kind,
                       l2norm, scale ? 1.0f / sqrtf((float)D) : 1.0f);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_rowpair_backward(const float* a, const float* b, const float* dsim, float* da, float* db, long rows,
                                    int D, int kind, int l2norm, int scale, void* stream) {
    TAG_CHECK_ARG(a && b && dsim && da && db && rows > 0 && D > 0 && (kind == 0 || kind == 1));
    hipLaunchKernelGGL(rowpair_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), a, b, dsim, da, db, rows, D,
                       kind, l2norm, scale ? 1.0f / sqrtf((float)D) : 1.0f);
    TAG_LAUNCH_CHECK();
    return 0;
}
