// Dense fp32 GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), row-major, any transposition.
// Serves nn.Linear fc1 (models/audio_encoder.py:140,216), audio_proj/text_proj
// (models/audio_text_model.py:45-46,78-87), the GRU input projections (nn.GRU, :141,217), every
// backward GEMM of those, and align.DotProduct (models/align.py:14-31) through the fused
// sigmoid/clamp + (B,B,T,N) scatter epilogue.
//
// 128x128 (large problems) or 64x64 tiles, 4 waves x (2x2 | 1x1) 32x32 MFMA tiles, K chunks of 32 double-buffered in LDS
// (16 for the batched Winograd-domain products of conv_wino.hip: tag_launch_gemm_batched, blockIdx.y or a 1-D index = product).
// Both kinds of operand are COPIED into LDS with 16-byte writes (round 4: a ds_write_b32 waits for gaps in the co-resident
// workgroups' fp32 MFMA streams, tools/coissue_probe.hip, and the transposing store of a k-contiguous operand was four of them
// per float4): an mn-contiguous operand k-major (S[k][r]), a k-contiguous one row-major (S[r][32 k + 4 pad], the halo conv's
// conflict-free 144-byte rows).  A fragment of the row-major image is ONE ds_read_b128 = the lane's operands of four k-steps:
// k-step j of an 8-k group multiplies k = 8 g + 4 kl + j (kl = lane / 32); the k-major image is read at the same permuted rows.
#include <type_traits>
#include "tag_common.h"

namespace {

constexpr int GK = 32;

// element loader with tail handling: 4 consecutive elements starting at p, `n` of them valid
__device__ __forceinline__ float4 load4(const float* p, int n, bool aligned) {
    if (n >= 4 && aligned) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0, 0, 0, 0);
    if (n > 0) v.x = p[0];
    if (n > 1) v.y = p[1];
    if (n > 2) v.z = p[2];
    if (n > 3) v.w = p[3];
    return v;
}

// KC = operand is k-contiguous in memory (element (r,k) at base[r*ld + k]); else mn-contiguous
// (element (r,k) at base[k*ld + r]).  LDS image: S[r][LD] (KC) or S[k][LD] (else).  T = tile edge (64 or 128).
template <bool KC, int T, int GKT = GK>
struct Stage {
    static constexpr int LD = KC ? GKT + 4 : T;
    static constexpr int SZ = KC ? T * (GKT + 4) : GKT * T;     // floats per buffer
    static constexpr int NL = T * GKT / 1024;       // float4 per thread per chunk
    static constexpr int RQ = GKT / 4;              // float4 per row of a k-contiguous operand
    f32x4 reg[NL];
    // FAST: every tile and K chunk is whole and 16-byte aligned (checked by the launcher) -> plain float4 loads, no tail tests
    template <bool FAST>
    __device__ __forceinline__ void load(const float* base, int ld, int r0, int rmax, int k0, int kmax, bool al) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = threadIdx.x + 256 * i;
            float4 v;
            if (KC) {
                const int r = r0 + idx / RQ, k = k0 + (idx % RQ) * 4;
                if constexpr (FAST) v = *reinterpret_cast<const float4*>(base + (size_t)r * ld + k);
                else v = (r < rmax) ? load4(base + (size_t)r * ld + k, kmax - k, al) : make_float4(0, 0, 0, 0);
            } else {
                const int k = k0 + idx / (T / 4), r = r0 + (idx % (T / 4)) * 4;
                if constexpr (FAST) v = *reinterpret_cast<const float4*>(base + (size_t)k * ld + r);
                else v = (k < kmax) ? load4(base + (size_t)k * ld + r, rmax - r, al) : make_float4(0, 0, 0, 0);
            }
            reg[i] = (f32x4){v.x, v.y, v.z, v.w};
        }
    }
    __device__ __forceinline__ void store(float* s) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = threadIdx.x + 256 * i;
            if (KC) {
                const int r = idx / RQ, k = (idx % RQ) * 4;
                *reinterpret_cast<f32x4*>(s + r * LD + k) = reg[i];
            } else {
                const int k = idx / (T / 4), r = (idx % (T / 4)) * 4;
                *reinterpret_cast<f32x4*>(s + k * LD + r) = reg[i];
            }
        }
    }
};

// bf16-MFMA variant (tag_gemm_bf16): the LDS image is bf16, ROW-major S[r][32 k] with an 80-byte row stride (64 B of k + 16 B
// pad: the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots), so a fragment of v_mfma_f32_32x32x16_bf16 -- lane
// (kl, ml) = row ml, k = 8 kl .. 8 kl + 7 of a k16 step -- is ONE 16-byte LDS read.  Operands are rounded to bf16 (nearest-even)
// on their way into LDS.  A k-contiguous operand stores 4 k of one row as 8 bytes; an mn-contiguous one loads the same 4 rows
// at k and k + 1 and stores four (k, k+1) pairs.
// K chunk of the bf16 variant.  64 (144-byte rows, 4 MFMAs per wave of a 64-tile between two barriers instead of 2) was
// measured 8-20 % SLOWER at the step's shapes than 32 (tools/gemm_bench.py: 0.546 vs 0.50 ms for the nine GEMMs) -- the longer
// load -> convert -> store chain per chunk costs more than the halved barrier count buys -- so 32 stays.
constexpr int GKB = 32;
constexpr int BFROW = 2 * GKB + 16;        // 80-byte rows: (5 r) mod 16 is a bijection on the 16-byte slots of a lane group
template <bool KC, int T>
struct StageBF {
    static constexpr int NL = T * GKB / 1024;           // float4 per thread per chunk
    static constexpr int RQ = GKB / 4;                  // float4 per row of a k-contiguous operand
    f32x4 reg[NL];
    template <bool FAST>
    __device__ __forceinline__ void load(const float* base, int ld, int r0, int rmax, int k0, int kmax, bool al) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            float4 v;
            if (KC) {
                const int idx = threadIdx.x + 256 * i;
                const int r = r0 + idx / RQ, k = k0 + (idx % RQ) * 4;
                if constexpr (FAST) v = *reinterpret_cast<const float4*>(base + (size_t)r * ld + k);
                else v = (r < rmax) ? load4(base + (size_t)r * ld + k, kmax - k, al) : make_float4(0, 0, 0, 0);
            } else {
                const int idx2 = threadIdx.x + 256 * (i >> 1);
                const int k = k0 + 2 * (idx2 / (T / 4)) + (i & 1), r = r0 + (idx2 % (T / 4)) * 4;
                if constexpr (FAST) v = *reinterpret_cast<const float4*>(base + (size_t)k * ld + r);
                else v = (k < kmax) ? load4(base + (size_t)k * ld + r, rmax - r, al) : make_float4(0, 0, 0, 0);
            }
            reg[i] = (f32x4){v.x, v.y, v.z, v.w};
        }
    }
    __device__ __forceinline__ void store(unsigned char* s) const {
        if (KC) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = threadIdx.x + 256 * i;
                const int r = idx / RQ, k = (idx % RQ) * 4;
                *reinterpret_cast<tag_u32x2*>(s + r * BFROW + k * 2) =
                    (tag_u32x2){tag_pack_bf16(reg[i].x, reg[i].y), tag_pack_bf16(reg[i].z, reg[i].w)};
            }
        } else {
#pragma unroll
            for (int i = 0; i < NL; i += 2) {
                const int idx2 = threadIdx.x + 256 * (i >> 1);
                const int kp = idx2 / (T / 4), r = (idx2 % (T / 4)) * 4;
                unsigned char* d = s + r * BFROW + kp * 4;
                *reinterpret_cast<unsigned*>(d) = tag_pack_bf16(reg[i].x, reg[i + 1].x);
                *reinterpret_cast<unsigned*>(d + BFROW) = tag_pack_bf16(reg[i].y, reg[i + 1].y);
                *reinterpret_cast<unsigned*>(d + 2 * BFROW) = tag_pack_bf16(reg[i].z, reg[i + 1].z);
                *reinterpret_cast<unsigned*>(d + 3 * BFROW) = tag_pack_bf16(reg[i].w, reg[i + 1].w);
            }
        }
    }
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return fminf(fmaxf(1.0f / (1.0f + expf(-v)), 1e-7f), 1.0f);
    if (act == 3) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 4) return tanhf(v);
    if (act == 5) return 1.0f / (1.0f + expf(-v));
    return v;
}

struct Epilogue {
    const float* bias;
    int act;          // 0 none, 1 relu, 2 sigmoid().clamp(1e-7, 1), 3 gelu (erf form), 4 tanh, 5 sigmoid
    int accumulate;
    float alpha;      // scales the product before bias/act
    int scatter;      // 1: align layout: row = (b,t), col = (b2,n) -> out[b][b2][t][n]
    int sc_T, sc_N, sc_B;
};

typedef __bf16 gemm_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gemm_u32x4 __attribute__((ext_vector_type(4)));

// BF = false: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  BF = true (BASELINE configs[2] mode, tag_gemm_bf16): the same fp32
// tensors and the same LDS image, but the fragments are rounded to bf16 (nearest-even) on their way from LDS to the registers
// and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- what autocast does to nn.Linear / the GRU projections.
// GKT = K chunk per barrier of the fp32 form: 32 (default), or 16 for the batched Winograd-domain products -- half the LDS
// (36.8 KB at 128 x 128: THREE workgroups per CU) and a barrier per 32 MFMAs, the configuration of conv3x3_halo_kernel's
// half-tap weight stages, which is what keeps the matrix pipe fed there.
template <bool AKC, bool BKC, int T, bool BF = false, bool FAST = false, int GKT = GK>
__global__ __launch_bounds__(256, GKT == 16 ? 3 : 1) void gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                   int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                   Epilogue ep, bool a_al, bool b_al, int splits, int kchunk,
                                                   float* __restrict__ partial, long sA = 0, long sB = 0, long sC = 0,
                                                   int batch1d = 0) {
    // batched launches (tag_gemm_batched: the 16 Winograd-domain products of conv_wino.hip): blockIdx.y = batch entry
    if (gridDim.y > 1) { A += (size_t)blockIdx.y * sA; B += (size_t)blockIdx.y * sB; C += (size_t)blockIdx.y * sC; }
    constexpr int LDSA = Stage<AKC, T, GKT>::LD, LDSB = Stage<BKC, T, GKT>::LD;
    constexpr int KCH = BF ? GKB : GKT;    // K chunk per barrier
    constexpr int TT = T / 64;             // 32x32 MFMA tiles per wave per dimension (waves 2 x 2)
    constexpr int ASZ = BF ? T * BFROW / 4 : Stage<AKC, T, GKT>::SZ;                   // floats per buffer (16-byte aligned)
    constexpr int BSZ = BF ? T * BFROW / 4 : Stage<BKC, T, GKT>::SZ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][ASZ]
    float* Bs = smem + 2 * ASZ;            // [2][BSZ]
    const int n_tiles = (N + T - 1) / T, m_tiles = (M + T - 1) / T;
    int L;
    if (batch1d > 0) {
        // ONE-dimensional batched launch (the weight-gradient products: few tiles per entry, many entries): the XCD remap runs over
        // (entry, tile), so ALL tiles of an entry -- which read the same two K-slice slabs -- sit on one XCD and share its L2.  As a
        // 2-D launch of 16-tile rows every XCD ran 2 tiles of every entry and fetched both slabs again: 5.2 GB per launch from HBM
        // for 2.1 GB of operands (PMC).
        const int per = n_tiles * m_tiles;
        const int Lb = xcd_remap(blockIdx.x, per * batch1d);
        const int e = Lb / per;
        L = Lb - e * per;
        A += (size_t)e * sA; B += (size_t)e * sB; C += (size_t)e * sC;
    } else {
        L = xcd_remap(blockIdx.x, n_tiles * m_tiles * splits);
    }
    // split-K: all tiles of ONE K slice are neighbours -- they read the same A and B slices, and xcd_remap gives an XCD a
    // contiguous run of L, i.e. whole K slices whose operands are then fetched once into that XCD's L2 (the tile-major order
    // put the K slices of one tile side by side, which share nothing: 4 x the algorithmic fetch on the weight-gradient GEMMs)
    L = __builtin_amdgcn_readfirstlane(L);     // (a function of blockIdx: said explicitly, so that tile origins and K ranges live on the scalar unit)
    const int split = L / (n_tiles * m_tiles);
    L -= split * (n_tiles * m_tiles);
    const int n0 = __builtin_amdgcn_readfirstlane((L % n_tiles) * T), m0 = __builtin_amdgcn_readfirstlane((L / n_tiles) * T);
    const int kbeg = __builtin_amdgcn_readfirstlane(split * kchunk);
    const int kend = __builtin_amdgcn_readfirstlane((kbeg + kchunk < K) ? kbeg + kchunk : K);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm0 = (wid >> 1) * (T / 2), wn0 = (wid & 1) * (T / 2);
    const int kl = lane >> 5, ml = lane & 31;

    // PD register sets = PD K chunks requested ahead of the one being multiplied.  Measured at the step's shapes (B = 64,
    // tools/gemm_bench.py): PD = 3 (64-tiles) / 2 (128-tiles) is 0-17 % SLOWER than PD = 1 -- the loop is bound by its
    // per-chunk instruction and barrier overhead (TAG_GEMM_ABL), not by the latency of the loads -- so one chunk ahead stays.
    // The 16-wide chunk of the batched Winograd-domain products (4 workgroups per CU, 106 VGPRs): with a k-contiguous A operand TWO
    // chunks ahead is 5 % faster (2.99 -> 2.84 ms for the 16 products of the 512 -> 512 layer), with the m-contiguous A of the
    // weight-gradient products it is 17 % slower (2.77 -> 3.25 ms): one chunk there.
    constexpr int PD = (GKT == 16 && !BF && AKC) ? 2 : 1;
    typename std::conditional<BF, StageBF<AKC, T>, Stage<AKC, T, GKT>>::type sa[PD];
    typename std::conditional<BF, StageBF<BKC, T>, Stage<BKC, T, GKT>>::type sb[PD];
    f32x16 acc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int kiters = __builtin_amdgcn_readfirstlane((kend - kbeg + KCH - 1) / KCH);
    auto ld = [&](auto& ra, auto& rb, int chunk) {
        // FAST loads carry no bounds tests: past the last chunk the last one is requested again (and never used)
        const int c = FAST ? (chunk < kiters ? chunk : kiters - 1) : chunk;
        ra.template load<FAST>(A, lda, m0, M, kbeg + c * KCH, kend, a_al);
        rb.template load<FAST>(B, ldb, n0, N, kbeg + c * KCH, kend, b_al);
    };
    auto st = [&](const auto& ra, const auto& rb, int buf) {
        if constexpr (BF) {
            ra.store(reinterpret_cast<unsigned char*>(As + buf * ASZ));
            rb.store(reinterpret_cast<unsigned char*>(Bs + buf * BSZ));
        } else {
            ra.store(As + buf * ASZ);
            rb.store(Bs + buf * BSZ);
        }
    };
    if constexpr (FAST && !BF && PD == 1) {
        // Whole tiles on the exact-fp32 MFMA: beside it every VALU instruction and every load with a 64-bit vector address is matrix
        // time (tools/coissue2_probe.hip: ~3 and ~24 clocks; a scalar-base buffer load ~7).  The operands therefore come through
        // buffer descriptors at the TILE origin with one loop-invariant 32-bit vector offset per load and the K advance on the scalar
        // unit, and the loop body exists once per LDS stage so that every stage offset is an immediate of the LDS instruction: a chunk
        // carries no address arithmetic (it was 13-16 vector instructions, four of them 64-bit multiply-adds, beside 16 MFMAs).
        constexpr int NLA = Stage<AKC, T, GKT>::NL, NLB = Stage<BKC, T, GKT>::NL;
        constexpr int RQ = GKT / 4;
        auto uniform = [](const float* ptr) {          // (the tile origin IS wave-uniform; said explicitly, or the descriptor is built in a waterfall loop)
            const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
        };
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(uniform(AKC ? A + (size_t)m0 * lda : A + m0), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(uniform(BKC ? B + (size_t)n0 * ldb : B + n0), 0, 0x7FFFFFFF, 0x00020000);
        unsigned voa[NLA], vob[NLB];
        int wsa[NLA], wsb[NLB];                        // LDS offsets (floats) of the thread's pieces inside a stage
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int idx = threadIdx.x + 256 * i;
            if (AKC) { const int r = idx / RQ, k = (idx % RQ) * 4; voa[i] = (unsigned)((r * lda + k) * 4); wsa[i] = r * LDSA + k; }
            else { const int k = idx / (T / 4), r = (idx % (T / 4)) * 4; voa[i] = (unsigned)((k * lda + r) * 4); wsa[i] = k * LDSA + r; }
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int idx = threadIdx.x + 256 * i;
            if (BKC) { const int r = idx / RQ, k = (idx % RQ) * 4; vob[i] = (unsigned)((r * ldb + k) * 4); wsb[i] = r * LDSB + k; }
            else { const int k = idx / (T / 4), r = (idx % (T / 4)) * 4; vob[i] = (unsigned)((k * ldb + r) * 4); wsb[i] = k * LDSB + r; }
        }
        f32x4 ra[NLA], rb[NLB];
        auto ldf = [&](int chunk) {                    // (past the last chunk the last one is requested again and never used)
            const int k0 = kbeg + (chunk < kiters ? chunk : kiters - 1) * KCH;
            const int soa = AKC ? k0 * 4 : k0 * lda * 4, sob = BKC ? k0 * 4 : k0 * ldb * 4;
#pragma unroll
            for (int i = 0; i < NLA; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, voa[i], soa, 0));
#pragma unroll
            for (int i = 0; i < NLB; ++i) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, vob[i], sob, 0));
        };
        auto stf = [&](float* as, float* bs) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) *reinterpret_cast<f32x4*>(as + wsa[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < NLB; ++i) *reinterpret_cast<f32x4*>(bs + wsb[i]) = rb[i];
        };
        const int afr = AKC ? (wm0 + ml) * LDSA + kl * 4 : (kl * 4) * LDSA + wm0 + ml;
        const int bfr = BKC ? (wn0 + ml) * LDSB + kl * 4 : (kl * 4) * LDSB + wn0 + ml;
        auto step = [&](int it, auto stage) {
            constexpr int ST = decltype(stage)::value;
            const float* a = As + ST * ASZ + afr;
            const float* b = Bs + ST * BSZ + bfr;
            ldf(it + 1);
            __builtin_amdgcn_sched_barrier(0);
            auto frag_a = [&](int g, f32x4 (&v)[TT]) {
#pragma unroll
                for (int i = 0; i < TT; ++i) {
                    if (AKC) v[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDSA + 8 * g);
                    else v[i] = (f32x4){a[(8 * g) * LDSA + i * 32], a[(8 * g + 1) * LDSA + i * 32], a[(8 * g + 2) * LDSA + i * 32],
                                        a[(8 * g + 3) * LDSA + i * 32]};
                }
            };
            auto frag_b = [&](int g, f32x4 (&v)[TT]) {
#pragma unroll
                for (int j = 0; j < TT; ++j) {
                    if (BKC) v[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDSB + 8 * g);
                    else v[j] = (f32x4){b[(8 * g) * LDSB + j * 32], b[(8 * g + 1) * LDSB + j * 32], b[(8 * g + 2) * LDSB + j * 32],
                                        b[(8 * g + 3) * LDSB + j * 32]};
                }
            };
            f32x4 af[2][TT], bf[2][TT];
            frag_a(0, af[0]);
            frag_b(0, bf[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < GKT / 8; ++g) {
                if (g + 1 < GKT / 8) { frag_a(g + 1, af[(g + 1) & 1]); frag_b(g + 1, bf[(g + 1) & 1]); }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TT; ++i)
#pragma unroll
                        for (int j = 0; j < TT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][e], bf[g & 1][j][e], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it + 1 < kiters) stf(As + (1 - ST) * ASZ, Bs + (1 - ST) * BSZ);
            __syncthreads();
        };
        ldf(0);
        stf(As, Bs);
        __syncthreads();
        {
            int it = 0;
            for (; it + 1 < kiters; it += 2) {
                step(it, std::integral_constant<int, 0>{});
                step(it + 1, std::integral_constant<int, 1>{});
            }
            if (it < kiters) step(it, std::integral_constant<int, 0>{});
        }
    } else {
#pragma unroll
    for (int d = 0; d < PD; ++d) ld(sa[d], sb[d], d);
    st(sa[0], sb[0], 0);
    __syncthreads();
    for (int it0 = 0; it0 < kiters; it0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
        const int it = it0 + d;
        if (it >= kiters) break;
        const int buf = it & 1;
#ifndef TAG_GEMM_ABL     // ablation builds (tools/README.md): 1 = no staging after the first chunk, 2 = and no barrier
        ld(sa[d], sb[d], it + PD);                 // chunk `it` of this register set is already in LDS
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BF) {
            const unsigned char* a = reinterpret_cast<const unsigned char*>(As + buf * ASZ) + (wm0 + ml) * BFROW + kl * 16;
            const unsigned char* b = reinterpret_cast<const unsigned char*>(Bs + buf * BSZ) + (wn0 + ml) * BFROW + kl * 16;
#pragma unroll
            for (int ks = 0; ks < GKB / 16; ++ks) {
                gemm_u32x4 af[TT], bfr[TT];
#pragma unroll
                for (int i = 0; i < TT; ++i) af[i] = *reinterpret_cast<const gemm_u32x4*>(a + i * 32 * BFROW + ks * 32);
#pragma unroll
                for (int j = 0; j < TT; ++j) bfr[j] = *reinterpret_cast<const gemm_u32x4*>(b + j * 32 * BFROW + ks * 32);
#pragma unroll
                for (int i = 0; i < TT; ++i)
#pragma unroll
                    for (int j = 0; j < TT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gemm_bf16x8, af[i]),
                                                                            __builtin_bit_cast(gemm_bf16x8, bfr[j]), acc[i][j],
                                                                            0, 0, 0);
            }
        } else {
        // fragment of 8-k group g: four operands per lane and 32-row tile (k = 8 g + 4 kl + j, j = 0..3)
        const float* a = As + buf * ASZ + (AKC ? (wm0 + ml) * LDSA + kl * 4 : (kl * 4) * LDSA + wm0 + ml);
        const float* b = Bs + buf * BSZ + (BKC ? (wn0 + ml) * LDSB + kl * 4 : (kl * 4) * LDSB + wn0 + ml);
        auto frag_a = [&](int g, f32x4 (&v)[TT]) {
#pragma unroll
            for (int i = 0; i < TT; ++i) {
                if (AKC) v[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDSA + 8 * g);
                else v[i] = (f32x4){a[(8 * g) * LDSA + i * 32], a[(8 * g + 1) * LDSA + i * 32], a[(8 * g + 2) * LDSA + i * 32],
                                    a[(8 * g + 3) * LDSA + i * 32]};
            }
        };
        auto frag_b = [&](int g, f32x4 (&v)[TT]) {
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                if (BKC) v[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDSB + 8 * g);
                else v[j] = (f32x4){b[(8 * g) * LDSB + j * 32], b[(8 * g + 1) * LDSB + j * 32], b[(8 * g + 2) * LDSB + j * 32],
                                    b[(8 * g + 3) * LDSB + j * 32]};
            }
        };
        f32x4 af[2][TT], bf[2][TT];
        frag_a(0, af[0]);
        frag_b(0, bf[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GKT / 8; ++g) {
            if (g + 1 < GKT / 8) { frag_a(g + 1, af[(g + 1) & 1]); frag_b(g + 1, bf[(g + 1) & 1]); }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TT; ++i)
#pragma unroll
                    for (int j = 0; j < TT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][e], bf[g & 1][j][e], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef TAG_GEMM_ABL
        if (it + 1 < kiters) st(sa[(d + 1) % PD], sb[(d + 1) % PD], buf ^ 1);
#endif
#if !defined(TAG_GEMM_ABL) || TAG_GEMM_ABL < 2
        __syncthreads();
#endif
        }
    }
    }
    if constexpr (FAST && GKT == 16 && !BF) {
        // The batched Winograd-domain products (whole tiles, no bias / activation / split-K): 16-byte stores after a 4 x 4 transpose
        // inside each lane quad (two DPP exchange rounds, as conv3x3_halo_kernel's epilogue) -- a global_store_dword is one of the
        // instruction forms that wait for gaps in the co-resident waves' MFMA streams (tools/coissue_probe.hip), and a 128-tile has
        // 64 of them per lane: the ablation without stores says they cost 10 % of this kernel (2.25 -> 2.03 ms for the 16 products
        // of the 512 -> 512 layer).
        const int c4 = lane & 3;
        const bool odd = c4 & 1, hi = c4 & 2;
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                const int nq = n0 + wn0 + j * 32 + (ml & ~3);
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float x0 = acc[i][j][4 * rq], x1 = acc[i][j][4 * rq + 1], x2 = acc[i][j][4 * rq + 2], x3 = acc[i][j][4 * rq + 3];
                    const float s01 = odd ? x0 : x1, s23 = odd ? x2 : x3;
                    const float r01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
                    const float r23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
                    if (odd) { x0 = r01; x2 = r23; } else { x1 = r01; x3 = r23; }
                    const float s02 = hi ? x0 : x2, s13 = hi ? x1 : x3;
                    const float r02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
                    const float r13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
                    if (hi) { x0 = r02; x1 = r13; } else { x2 = r02; x3 = r13; }
                    // x_t = column nq + t of row (register 4 rq + c4) = c4 + 8 rq + 4 kl of the 32-row tile
                    const int m = m0 + wm0 + i * 32 + c4 + 8 * rq + 4 * kl;
                    *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + nq) = (f32x4){x0, x1, x2, x3};
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int n = n0 + wn0 + j * 32 + ml;
            if (n >= N) continue;
            const float bv = ep.bias ? ep.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                if (m >= M) continue;
                if (splits > 1) {            // raw K-slice partial; the epilogue runs in splitk_reduce_kernel
                    partial[((size_t)split * M + m) * N + n] = acc[i][j][r];
                    continue;
                }
                float v = acc[i][j][r] * ep.alpha + bv;
                size_t o;
                if (ep.scatter) {
                    const int bb = m / ep.sc_T, t = m % ep.sc_T, b2 = n / ep.sc_N, nn = n % ep.sc_N;
                    o = (((size_t)bb * ep.sc_B + b2) * ep.sc_T + t) * ep.sc_N + nn;
                } else {
                    o = (size_t)m * ldc + n;
                }
                if (ep.accumulate) v += C[o];
                v = apply_act(v, ep.act);
                C[o] = v;
            }
        }
}

// deterministic split-K combine + epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N,
                                                            float* __restrict__ C, int ldc, Epilogue ep) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = 0.0f;
        for (int sp = 0; sp < splits; ++sp) s += partial[(size_t)sp * total + i];
        float v = s * ep.alpha + (ep.bias ? ep.bias[n] : 0.0f);
        const size_t o = (size_t)m * ldc + n;
        if (ep.accumulate) v += C[o];
        v = apply_act(v, ep.act);
        C[o] = v;
    }
}

// K slices for a problem whose tile count cannot fill the chip (weight-gradient GEMMs: small MxN, K = B*T)
int gemm_splits(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (tiles >= 384 || K < 1024) return 1;
    long s = (768 + tiles - 1) / tiles;
    const long maxs = K / 256;
    if (s > maxs) s = maxs;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
}

template <bool AKC, bool BKC, int T, bool BF = false, int GKT = GK>
void launch_gemm_t(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                   Epilogue ep, bool a_al, bool b_al, int splits, float* partial, hipStream_t st, int batch = 1, long sA = 0,
                   long sB = 0, long sC = 0, bool one_d = false) {
    constexpr int ASZ = BF ? T * BFROW / 4 : Stage<AKC, T, GKT>::SZ;
    constexpr int BSZ = BF ? T * BFROW / 4 : Stage<BKC, T, GKT>::SZ;
    const size_t lds = (size_t)(2 * ASZ + 2 * BSZ) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<AKC, BKC, T, BF, false, GKT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = ((M + T - 1) / T) * ((N + T - 1) / T) * splits;
    constexpr int KCH = BF ? GKB : GKT;
    int kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + KCH - 1) / KCH * KCH;
    // whole tiles, whole K chunks, no empty K slice, 16-byte aligned rows: the loader without tail handling
    // (... and, for the buffer-load form of the fp32 kernel, 32-bit byte offsets from a tile's origin)
    const size_t lim = 0x7FFFFFFFull;
    const bool off32 = (AKC ? ((size_t)T * lda + K) * 4 < lim : (size_t)K * lda * 4 < lim) &&
                       (BKC ? ((size_t)T * ldb + K) * 4 < lim : (size_t)K * ldb * 4 < lim);
    const bool fast = a_al && b_al && M % T == 0 && N % T == 0 && K % KCH == 0 && (long)(splits - 1) * kchunk < K && off32;
    if (fast) {
        static bool fast_attr_set = false;
        if (!fast_attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<AKC, BKC, T, BF, true, GKT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            fast_attr_set = true;
        }
        hipLaunchKernelGGL((gemm_kernel<AKC, BKC, T, BF, true, GKT>), one_d ? dim3(grid * batch) : dim3(grid, batch), dim3(256), lds, st,
                           A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, splits, kchunk, partial, sA, sB, sC, one_d ? batch : 0);
    } else {
        hipLaunchKernelGGL((gemm_kernel<AKC, BKC, T, BF, false, GKT>), one_d ? dim3(grid * batch) : dim3(grid, batch), dim3(256), lds, st,
                           A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, splits, kchunk, partial, sA, sB, sC, one_d ? batch : 0);
    }
    if (splits > 1) {
        long nb = ((long)M * N + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)nb), dim3(256), 0, st, partial, splits, M, N, C, ldc, ep);
    }
}

int launch_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc, int M,
                int N, int K, Epilogue ep, float* ws, hipStream_t st, bool bf = false) {
    const int splits = (ws && !ep.scatter) ? gemm_splits(M, N, K) : 1;
    const bool a_al = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (lda % 4 == 0);
    const bool b_al = (reinterpret_cast<uintptr_t>(B) % 16 == 0) && (ldb % 4 == 0);
    // A stored (M,K) -> k-contiguous; transA: stored (K,M) -> m-contiguous
    // B stored (K,N) -> n-contiguous; transB: stored (N,K) -> k-contiguous
    const bool akc = !transA, bkc = transB != 0;
    // 128x128 tiles only when they still make >= 4 residency rounds: below that the 64-tiles (4 workgroups per CU, two rounds
    // whose epilogue stores overlap the next tiles' products) are 3-8 % faster at the step's shapes (tools/gemm_bench.py)
    static const int big_min = tag_option("gemm_big_min");
    const bool big = splits == 1 && (long)((M + 127) / 128) * ((N + 127) / 128) >= big_min;
#define GEMM_DISPATCH(AK, BK_)                                                                                        \
    if (bf) {                                                                                                         \
        if (big) launch_gemm_t<AK, BK_, 128, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st);  \
        else launch_gemm_t<AK, BK_, 64, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, splits, ws, st);       \
    } else if (big) launch_gemm_t<AK, BK_, 128>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st);     \
    else launch_gemm_t<AK, BK_, 64>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, splits, ws, st);
    if (akc && bkc) { GEMM_DISPATCH(true, true) }
    else if (akc && !bkc) { GEMM_DISPATCH(true, false) }
    else if (!akc && bkc) { GEMM_DISPATCH(false, true) }
    else { GEMM_DISPATCH(false, false) }
#undef GEMM_DISPATCH
    return 0;
}

// column sums: partial over row blocks then finalize
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ld, long M, int N,
                                                             double* __restrict__ partials) {
    // block handles a 64-column stripe (blockIdx.y) and a strided set of rows
    __shared__ double sred[4][64];
    const int col = blockIdx.y * 64 + (threadIdx.x & 63);
    const int rsub = threadIdx.x >> 6;
    double s = 0;
    if (col < N)
        for (long r = (long)blockIdx.x * 4 + rsub; r < M; r += (long)gridDim.x * 4) s += x[(size_t)r * ld + col];
    sred[rsub][threadIdx.x & 63] = s;
    __syncthreads();
    if (rsub == 0 && col < N)
        partials[(size_t)blockIdx.x * N + col] = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] +
                                                 sred[3][threadIdx.x];
}
// 16 columns x 16 row groups per block, folded through LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ partials, int nblk, int N,
                                                           float* __restrict__ out) {
    __shared__ double sh[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    double s = 0;
    if (c < N)
        for (int b = g; b < nblk; b += 16) s += partials[(size_t)b * N + c];
    sh[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < N) {
        double t = 0;
        for (int q = 0; q < 16; ++q) t += sh[q][cl];
        out[c] = (float)t;
    }
}
int colsum_blocks(long M) {
    long nb = (M + 63) / 64;
    return (int)(nb > 256 ? 256 : (nb < 1 ? 1 : nb));
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        dx[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

// row-wise L2 normalisation x / max(||x||, 1e-12)  (F.normalize) for align.DotProduct(l2norm=True)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                          int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.0f;
    for (int d = lane; d < D; d += 64) { const float v = x[row * D + d]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int d = lane; d < D; d += 64) y[row * D + d] = x[row * D + d] * inv;
}

// dx = (du - u (u.du)) / max(||x||, 1e-12), u = x / max(||x||, 1e-12)   (backward of F.normalize)
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ du,
                                                              float* __restrict__ dx, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float ss = 0.0f, xd = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float v = x[row * D + d];
        ss = fmaf(v, v, ss);
        xd = fmaf(v, du[row * D + d], xd);
    }
    ss = wave_sum(ss);
    xd = wave_sum(xd);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    const float ud = xd * inv;                        // u . du
    for (int d = lane; d < D; d += 64) dx[row * D + d] = (du[row * D + d] - x[row * D + d] * inv * ud) * inv;
}

// ds[(b,t)][(b2,n)] = dout[b][b2][t][n] * p (1-p) * alpha * [p >= 1e-7], p = out (sigmoid, clamp(1e-7, 1))
__global__ __launch_bounds__(256) void align_dscore_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                           float* __restrict__ ds, float alpha, int B, int T, int N) {
    const long total = (long)B * B * T * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % N);
        long r = i / N;
        const int t = (int)(r % T); r /= T;
        const int b2 = (int)(r % B);
        const int b = (int)(r / B);
        const float p = out[i];
        const float g = p > 1e-7f ? dout[i] * p * (1.0f - p) * alpha : 0.0f;
        ds[((size_t)b * T + t) * ((size_t)B * N) + (size_t)b2 * N + n] = g;
    }
}

}  // namespace

extern "C" size_t tag_gemm_ws_bytes(int M, int N, int K) {
    const int s = gemm_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int tag_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
                        int M, int N, int K, const float* bias, int act, int accumulate, void* ws, void* stream) {
    TAG_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N);
    TAG_CHECK_ARG(act == 0 || act == 1 || act == 3 || act == 4 || act == 5);
    Epilogue ep{bias, act, accumulate, 1.0f, 0, 1, 1, 1};
    launch_gemm(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, ep, static_cast<float*>(ws), as_stream(stream));
    TAG_LAUNCH_CHECK();
    return 0;
}

// the same GEMM with both operands rounded to bf16 (nearest-even) and fp32 accumulation (BASELINE configs[2] mode)
extern "C" int tag_gemm_bf16(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
                             int M, int N, int K, const float* bias, int act, int accumulate, void* ws, void* stream) {
    TAG_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N);
    TAG_CHECK_ARG(act == 0 || act == 1 || act == 3 || act == 4 || act == 5);
    Epilogue ep{bias, act, accumulate, 1.0f, 0, 1, 1, 1};
    launch_gemm(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, ep, static_cast<float*>(ws), as_stream(stream), true);
    TAG_LAUNCH_CHECK();
    return 0;
}

// `batch` independent products C_b = A_b x B_b of one shape in ONE launch (blockIdx.y = b): A_b (M,K) row-major, B_b (K,N)
// row-major, C_b (M,N); strides in floats.  128 x 128 tiles when M, N allow it (the launch as a whole makes many residency
// rounds), no split-K, no epilogue.  Serves the Winograd-domain products of conv_wino.hip.
// transA: A_b stored (K,M) (m-contiguous) -- the weight-gradient products of conv_wino.hip, whose K runs over the tiles.
int tag_launch_gemm_batched(const float* A, int lda, long sA, const float* B, int ldb, long sB, float* C, int ldc, long sC, int M,
                            int N, int K, int batch, hipStream_t st, int transA) {
    Epilogue ep{nullptr, 0, 0, 1.0f, 0, 1, 1, 1};
    const bool a_al = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (lda % 4 == 0) && (sA % 4 == 0);
    const bool b_al = (reinterpret_cast<uintptr_t>(B) % 16 == 0) && (ldb % 4 == 0) && (sB % 4 == 0);
    constexpr int gkt = 16;         // K chunk of the batched products (32, the dense GEMM's chunk, measured 15 % slower: docs/experiments_r05.md)
    if (transA) {
        if (M >= 128 && N >= 128 && gkt == 16 && K % 16 == 0)
            launch_gemm_t<false, false, 128, false, 16>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st, batch, sA, sB, sC,
                                                        true);
        else if (M >= 128 && N >= 128)
            launch_gemm_t<false, false, 128>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st, batch, sA, sB, sC, true);
        else
            launch_gemm_t<false, false, 64>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st, batch, sA, sB, sC, true);
        return 0;
    }
    if (M >= 128 && N >= 128 && gkt == 16 && K % 16 == 0)
        launch_gemm_t<true, false, 128, false, 16>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st, batch, sA, sB, sC);
    else if (M >= 128 && N >= 128)
        launch_gemm_t<true, false, 128>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st, batch, sA, sB, sC);
    else
        launch_gemm_t<true, false, 64>(A, lda, B, ldb, C, ldc, M, N, K, ep, a_al, b_al, 1, nullptr, st, batch, sA, sB, sC);
    return 0;
}

extern "C" size_t tag_colsum_ws_bytes(long M, int N) { return (size_t)colsum_blocks(M) * N * sizeof(double); }

extern "C" int tag_colsum(const float* x, int ld, long M, int N, float* out, void* ws, void* stream) {
    TAG_CHECK_ARG(x && out && ws && M > 0 && N > 0 && ld >= N);
    const int nblk = colsum_blocks(M);
    double* partials = static_cast<double*>(ws);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, cdiv(N, 64)), dim3(256), 0, as_stream(stream), x, ld, M, N,
                       partials);
    TAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 16)), dim3(256), 0, as_stream(stream), partials, nblk, N, out);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_relu_backward(const float* y, const float* dy, float* dx, long n, void* stream) {
    TAG_CHECK_ARG(y && dy && dx && n > 0);
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)nb), dim3(256), 0, as_stream(stream), y, dy, dx, n);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_align_dot_forward(const float* audio, const float* text, float* out, int l2norm, int scaled, int B,
                                     int T, int N, int D, float* ws, void* stream) {
    TAG_CHECK_ARG(audio && text && out && B > 0 && T > 0 && N > 0 && D > 0);
    const float* a = audio;
    const float* t = text;
    if (l2norm) {
        TAG_CHECK_ARG(ws != nullptr);
        float* an = ws;
        float* tn = ws + (size_t)B * T * D;
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cdiv((long)B * T, 4)), dim3(256), 0, as_stream(stream), audio, an,
                           (long)B * T, D);
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cdiv((long)B * N, 4)), dim3(256), 0, as_stream(stream), text, tn,
                           (long)B * N, D);
        TAG_LAUNCH_CHECK();
        a = an;
        t = tn;
    }
    Epilogue ep{nullptr, 2, 0, scaled ? 1.0f / sqrtf((float)D) : 1.0f, 1, T, N, B};
    // score[(b,t)][(b2,n)] = audio (B*T, D) x text^T (stored (B*N, D): k-contiguous -> transB)
    launch_gemm(a, D, 0, t, D, 1, out, 0, B * T, B * N, D, ep, nullptr, as_stream(stream));
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_l2norm_rows_forward(const float* x, float* y, long rows, int D, void* stream) {
    TAG_CHECK_ARG(x && y && rows > 0 && D > 0);
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, y, rows, D);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_l2norm_rows_backward(const float* x, const float* du, float* dx, long rows, int D, void* stream) {
    TAG_CHECK_ARG(x && du && dx && rows > 0 && D > 0);
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, du, dx, rows, D);
    TAG_LAUNCH_CHECK();
    return 0;
}
extern "C" int tag_align_dot_dscore(const float* out, const float* dout, float* ds, int scaled, int B, int T, int N,
                                    int D, void* stream) {
    TAG_CHECK_ARG(out && dout && ds && B > 0 && T > 0 && N > 0 && D > 0);
    const long total = (long)B * B * T * N;
    long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(align_dscore_kernel, dim3((int)nb), dim3(256), 0, as_stream(stream), out, dout, ds,
                       scaled ? 1.0f / sqrtf((float)D) : 1.0f, B, T, N);
    TAG_LAUNCH_CHECK();
    return 0;
}
