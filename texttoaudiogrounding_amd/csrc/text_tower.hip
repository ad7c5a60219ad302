// Text tower of the inference path (models/hf_modeling_grounding.py:183-199 LaionClapEncoder = Hugging Face
// ClapTextModel + ClapProjectionLayer, a RoBERTa-style post-LayerNorm encoder), forward only.  The dense layers are
// tag_gemm calls (bias / GELU / tanh / ReLU epilogues); this file holds the three row kernels in between:
//   tag_roberta_embed_ln   position ids (cumsum of non-pad tokens) + word + type + position embeddings + LayerNorm
//   tag_add_layernorm      LayerNorm(x + residual)
//   tag_mha_small          per (sequence, head) softmax(q k^T / sqrt(d) + key mask) v for short sequences (L <= 64)
// Sequences are a handful of tokens (a phrase), so attention is one wave per (sequence, head) with K/V in LDS.
#include "tag_common.h"

namespace {

// one wave per row of D floats held as NV values per lane (D <= 64 * NV)
template <int NV>
__device__ __forceinline__ void ln_row(float (&v)[NV], int D, int lane, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, float* __restrict__ out) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (lane + 64 * i < D) ? v[i] : 0.0f;
    const float mean = wave_sum(s) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d = (lane + 64 * i < D) ? v[i] - mean : 0.0f;
        q = fmaf(d, d, q);
    }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) out[c] = (v[i] - mean) * inv * gamma[c] + beta[c];
    }
}

template <int NV>
__global__ __launch_bounds__(256) void roberta_embed_ln_kernel(const long* __restrict__ ids, const float* __restrict__ word,
                                                               const float* __restrict__ type0,
                                                               const float* __restrict__ pos,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               float* __restrict__ out, int B, int L, int D, int pad_id) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * L) return;
    const int b = (int)(row / L), l = (int)(row % L);
    // position id = (number of non-pad tokens in ids[b][0..l]) + pad_id for a non-pad token, pad_id for a pad token
    int cnt = 0;
    for (int i0 = 0; i0 <= l; i0 += 64) {
        const int i = i0 + lane;
        const bool np = i <= l && ids[(long)b * L + i] != pad_id;
        cnt += __popcll(__ballot(np));
    }
    const long id = ids[row];
    const int p = id != pad_id ? cnt + pad_id : pad_id;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? word[id * D + c] + type0[c] + pos[(long)p * D + c] : 0.0f;
    }
    ln_row<NV>(v, D, lane, gamma, beta, eps, out + row * D);
}

template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ out, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? x[row * D + c] + (res ? res[row * D + c] : 0.0f) : 0.0f;
    }
    ln_row<NV>(v, D, lane, gamma, beta, eps, out + row * D);
}

// qkv: (B*L, 3*D) rows = [q | k | v], D = heads * DH.  One wave per (b, head); lane i = query i (i < L <= 64).
template <int DH>
__global__ __launch_bounds__(64) void mha_small_kernel(const float* __restrict__ qkv, const long* __restrict__ mask,
                                                       float* __restrict__ out, int B, int L, int heads) {
    extern __shared__ float sm[];                 // K [L][DH], V [L][DH]
    float* Ks = sm;
    float* Vs = sm + L * DH;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int D = heads * DH, lane = threadIdx.x;
    const float* base = qkv + (size_t)b * L * 3 * D + h * DH;
    for (int e = lane; e < L * DH; e += 64) {
        const int j = e / DH, d = e % DH;
        Ks[e] = base[(size_t)j * 3 * D + D + d];
        Vs[e] = base[(size_t)j * 3 * D + 2 * D + d];
    }
    __syncthreads();
    if (lane >= L) return;
    float q[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) q[d] = base[(size_t)lane * 3 * D + d];
    const float scale = 1.0f / sqrtf((float)DH);
    // pass 1: max over the unmasked keys; pass 2: exp / sum and the weighted value sum
    float mx = -3.0e38f;
    for (int j = 0; j < L; ++j) {
        if (mask[(long)b * L + j] == 0) continue;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s = fmaf(q[d], Ks[j * DH + d], s);
        mx = fmaxf(mx, s * scale);
    }
    float o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.0f;
    float den = 0.0f;
    for (int j = 0; j < L; ++j) {
        if (mask[(long)b * L + j] == 0) continue;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s = fmaf(q[d], Ks[j * DH + d], s);
        const float p = expf(s * scale - mx);
        den += p;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = fmaf(p, Vs[j * DH + d], o[d]);
    }
    const float inv = 1.0f / den;
    float* dst = out + ((size_t)b * L + lane) * D + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = o[d] * inv;
}

}  // namespace

#define BY_NV(D, CALL)                          \
    if ((D) <= 64) { CALL(1) }                  \
    else if ((D) <= 256) { CALL(4) }            \
    else if ((D) <= 768) { CALL(12) }           \
    else { CALL(16) }

extern "C" int tag_roberta_embed_ln(const long* ids, const float* word, const float* type0, const float* pos,
                                    const float* gamma, const float* beta, float eps, float* out, int B, int L, int D,
                                    int pad_id, void* stream) {
    TAG_CHECK_ARG(ids && word && type0 && pos && gamma && beta && out && B > 0 && L > 0 && D > 0 && D <= 1024);
    const long rows = (long)B * L;
#define CALL(NV)                                                                                                   \
    hipLaunchKernelGGL(roberta_embed_ln_kernel<NV>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), ids, word, \
                       type0, pos, gamma, beta, eps, out, B, L, D, pad_id);
    BY_NV(D, CALL)
#undef CALL
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_add_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                                 float* out, long rows, int D, void* stream) {
    TAG_CHECK_ARG(x && gamma && beta && out && rows > 0 && D > 0 && D <= 1024);
#define CALL(NV)                                                                                                 \
    hipLaunchKernelGGL(add_layernorm_kernel<NV>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, res, gamma, \
                       beta, eps, out, rows, D);
    BY_NV(D, CALL)
#undef CALL
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_mha_small(const float* qkv, const long* mask, float* out, int B, int L, int heads, int dh,
                             void* stream) {
    TAG_CHECK_ARG(qkv && mask && out && B > 0 && heads > 0 && L > 0 && L <= 64);
    TAG_CHECK_ARG(dh == 16 || dh == 32 || dh == 64);
    const size_t lds = (size_t)2 * L * dh * sizeof(float);
    if (dh == 16)
        hipLaunchKernelGGL(mha_small_kernel<16>, dim3(B * heads), dim3(64), lds, as_stream(stream), qkv, mask, out, B, L, heads);
    else if (dh == 32)
        hipLaunchKernelGGL(mha_small_kernel<32>, dim3(B * heads), dim3(64), lds, as_stream(stream), qkv, mask, out, B, L, heads);
    else
        hipLaunchKernelGGL(mha_small_kernel<64>, dim3(B * heads), dim3(64), lds, as_stream(stream), qkv, mask, out, B, L, heads);
    TAG_LAUNCH_CHECK();
    return 0;
}
