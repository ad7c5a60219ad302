// A1 (alternative arithmetic): the same 3x3 / stride 1 / pad 1 convolution as conv.hip (models/panns.py:25-33,49-50),
// forward + dgrad, with every fp32 operand split EXACTLY into three bf16 terms (8 + 8 + 8 mantissa bits:
// v = hi + mid + lo, by truncation, no residual) and the products formed on the bf16 matrix pipe
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate), which issues 16x the MACs per cycle of the fp32 MFMA:
//     a*b = (ah+am+al)(bh+bm+bl) ~= ah*bl + al*bh + am*bm + ah*bm + am*bh + ah*bh          (NP = 6 products)
// The three dropped terms (am*bl, al*bm, al*bl) are <= 2^-23 |a*b|, i.e. at the fp32 rounding level; NP = 9 keeps
// them (every partial product exact), NP = 1 is plain bf16.  NOT the default arithmetic of the library: the
// exact-fp32 kernels of conv.hip are; this path is selected explicitly (tag_conv3x3_forward_x3).
//
// Workgroup = 128 output pixels (TH x TW rectangle of one image) x BN couts, 4 waves.  Per 32-channel chunk the
// (TH+2) x (TW+2) input patch is staged once in LDS as three bf16 planes, pixel-major with an 80-byte pixel stride
// (64 B of channels + 16 B pad): a ds_read_b128 lane group (16 lanes = 16 pixels that are distinct mod 16, same
// channel octet) then covers all 16 sixteen-byte slots of the 256-B bank row -> conflict-free, and the 9 taps are
// immediate offsets.  The MFMA row -> pixel map is permuted so that each ds_read_b128 lane group
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}) reads 16 consecutive pixels (two rows r, r+2 of a 12-wide patch for TW=8).
// Weights are pre-split and pre-swizzled into MFMA B-fragment order (1 KB per fragment) and stream
// L2 -> registers directly (no LDS, no barrier), two k16 steps ahead of their use.
#include <stdlib.h>
#include "tag_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 32;            // channels per LDS chunk (2 k16 MFMA steps)
constexpr int PIXB = 80;          // bytes per pixel per plane in LDS

template <int TW>
struct X3Geom {
    static constexpr int TH = 128 / TW, PH = TH + 2;
    static constexpr int S = TW == 8 ? 12 : TW + 2;               // patch row stride in pixels
    static constexpr int PP = PH * S;
    static constexpr int PLANE = PP * PIXB;                       // bytes per plane
    static constexpr int ITEMS = (PP * 8 + 255) / 256;            // (pixel, channel quad) items per thread
    static constexpr int LDS_BYTES = 3 * PLANE + 2 * 512 * 4;     // + producer scale/shift table
};

// MFMA row i (0..31) -> pixel index inside the 32-pixel block, chosen so that the ds_read_b128 lane groups
// {0-3,12-15,20-27} and {4-11,16-19,28-31} map to pixels 0..15 and 16..31
__device__ __forceinline__ int row_to_pix(int i) {
    return (int)((0xED6360u >> (3 * (i >> 2))) & 7u) * 4 + (i & 3);       // quad map [0,4,5,1,6,2,3,7]
}

// pixel index m (0..127) of the tile -> (ty, tx)
template <int TW>
__device__ __forceinline__ void pix_to_yx(int m, int& ty, int& tx) {
    if (TW == 8) {
        const int blk = m >> 5, qq = m & 31, r = qq >> 3;
        ty = blk * 4 + ((r & 1) * 2 + (r >> 1));                  // rows 0,2,1,3 of the block
        tx = qq & 7;
    } else {
        ty = m / TW;
        tx = m % TW;
    }
}

__device__ __forceinline__ float prologue1(float v, int mode, float s, float t) {
    if (mode == 1) return fmaxf(fmaf(v, s, t), 0.0f);
    if (mode == 2) return fmaf(v > 0 ? v : 0.1f * v, s, t);
    if (mode == 3) return fmaf(v, s, t);
    return v;
}

// exact three-way split of two floats into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void split_pack(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ab = __float_as_uint(a), bb = __float_as_uint(b);
    const float a1 = a - __uint_as_float(ab & 0xffff0000u), b1 = b - __uint_as_float(bb & 0xffff0000u);
    const unsigned a1b = __float_as_uint(a1), b1b = __float_as_uint(b1);
    const float a2 = a1 - __uint_as_float(a1b & 0xffff0000u), b2 = b1 - __uint_as_float(b1b & 0xffff0000u);
    h = __builtin_amdgcn_perm(bb, ab, 0x07060302u);
    m = __builtin_amdgcn_perm(b1b, a1b, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(b2), __float_as_uint(a2), 0x07060302u);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MB = 32-pixel MFMA blocks per wave (4: wave = 128 px x 32 co, workgroup 128 x 128; 2: wave = 64 px x 32 co,
// workgroup 128 x 64).  NP = products per fp32 multiply (6 default, 9 exact, 1 plain bf16).
template <int MB, int PRO, int TW, int NP>
__global__ __launch_bounds__(256, 2) void conv3x3_x3_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift, float* __restrict__ y,
                                                            int B, int H, int W, int Cin, int Cout) {
    using G = X3Geom<TW>;
    constexpr int BN_ = MB == 4 ? 128 : 64;
    constexpr int NSPL = NP == 1 ? 1 : 3;                         // planes actually read
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Ss = reinterpret_cast<float*>(smem + 3 * G::PLANE);    // [2][Cin]

    const int n_tiles = Cout / BN_;
    const int row_tiles = (H + G::TH - 1) / G::TH;
    const int m_tiles = B * row_tiles;
    const int L = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int n0 = (L % n_tiles) * BN_;
    const int mt = L / n_tiles;
    const int img = mt / row_tiles, h0 = (mt % row_tiles) * G::TH;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = MB == 4 ? 0 : (wid >> 1);                      // M group of the wave
    const int wn = MB == 4 ? wid : (wid & 1);                     // 32-cout block of the wave
    const int kl = lane >> 5, ml = lane & 31;
    if (PRO != 0)
        for (int c = tid; c < Cin; c += 256) { Ss[c] = in_scale[c]; Ss[Cin + c] = in_shift[c]; }

    // ---- patch staging geometry: item = (patch pixel, channel quad) ----
    const int q = tid & 7;
    unsigned poff[G::ITEMS];
    unsigned pvalid = 0, pexist = 0;
#pragma unroll
    for (int i = 0; i < G::ITEMS; ++i) {
        const int idx = tid + 256 * i;
        const int pp = idx >> 3;                                  // 0 .. PH*(TW+2)-1 (dense numbering of real patch pixels)
        const int pr = pp / (TW + 2), pc = pp - pr * (TW + 2);
        const int h = h0 - 1 + pr, w = pc - 1;
        const bool ex = pp < G::PH * (TW + 2);
        const bool ok = ex & ((unsigned)h < (unsigned)H) & ((unsigned)w < (unsigned)W);
        pexist |= (unsigned)ex << i;
        pvalid |= (unsigned)ok << i;
        const long pix = ok ? ((long)img * H + h) * W + w : (long)img * H * W;
        poff[i] = (unsigned)((pix * Cin + q * 4) * 4);
    }
    // per-lane LDS byte offset of tap (ky=0,kx=0) for each MFMA block, channel octet kl
    unsigned abase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        int ty, tx;
        pix_to_yx<TW>((wm * MB + i) * 32 + row_to_pix(ml), ty, tx);
        abase[i] = (unsigned)((ty * G::S + tx) * PIXB + kl * 16);
    }
    const int KK = Cin / 16, NBK = Cout / 32;
    const u32x4* wlane = wp + (size_t)(n0 / 32 + wn) * 192 + lane;        // + ((tap*KK + kk) * NBK) * 192 + s*64

    f32x4 ra[G::ITEMS];
    auto issue_patch = [&](int cc) {
        const unsigned coff = (unsigned)(cc * KC * 4);
#pragma unroll
        for (int i = 0; i < G::ITEMS; ++i)
            ra[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(x) + (poff[i] + coff));
    };
    auto store_patch = [&](int cc) {
        f32x4 rs = {1.0f, 1.0f, 1.0f, 1.0f}, rt = {0.0f, 0.0f, 0.0f, 0.0f};
        if (PRO != 0) {
            rs = *reinterpret_cast<const f32x4*>(Ss + cc * KC + q * 4);
            rt = *reinterpret_cast<const f32x4*>(Ss + Cin + cc * KC + q * 4);
        }
#pragma unroll
        for (int i = 0; i < G::ITEMS; ++i) {
            if (!((pexist >> i) & 1u)) continue;
            const int pp = (tid + 256 * i) >> 3;
            const int pr = pp / (TW + 2), pc = pp - pr * (TW + 2);
            f32x4 v;
            v.x = prologue1(ra[i].x, PRO, rs.x, rt.x);
            v.y = prologue1(ra[i].y, PRO, rs.y, rt.y);
            v.z = prologue1(ra[i].z, PRO, rs.z, rt.z);
            v.w = prologue1(ra[i].w, PRO, rs.w, rt.w);
            if (!((pvalid >> i) & 1u)) v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            unsigned h0_, m0_, l0_, h1_, m1_, l1_;
            split_pack(v.x, v.y, h0_, m0_, l0_);
            split_pack(v.z, v.w, h1_, m1_, l1_);
            unsigned char* dst = smem + (pr * G::S + pc) * PIXB + q * 8;
            *reinterpret_cast<u32x2*>(dst) = (u32x2){h0_, h1_};
            if (NSPL == 3) {
                *reinterpret_cast<u32x2*>(dst + G::PLANE) = (u32x2){m0_, m1_};
                *reinterpret_cast<u32x2*>(dst + 2 * G::PLANE) = (u32x2){l0_, l1_};
            }
        }
    };

    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // weight fragments: ring of 3 k16 steps (18 steps per chunk = 6 turns of the ring), loaded 2 steps ahead
    u32x4 bq[3][NSPL];
    auto issue_b = [&](int cc, int step, int slot) {              // step = tap*2 + ks inside chunk cc
        const int tap = step >> 1, ks = step & 1;
        const u32x4* p = wlane + ((size_t)(tap * KK + cc * 2 + ks) * NBK) * 192;
#pragma unroll
        for (int s = 0; s < NSPL; ++s) bq[slot][s] = p[s * 64];
    };

    const int cchunks = Cin / KC;
    issue_patch(0);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    __syncthreads();                                              // Ss visible
    store_patch(0);
    __syncthreads();

    // one half-step = 2 MFMA blocks x NP products on one (tap, k16) slice; HS half-steps per step, 18 steps per chunk.
    // A fragments of half-step i+1 are read from LDS while the MFMAs of half-step i run; the scheduler is pinned
    // (sched_group_barrier) to 1 ds_read per 2 MFMAs, otherwise it sinks every load next to its use.
    constexpr int HS = MB / 2, NH = 18 * HS;
    constexpr int PA[9] = {2, 2, 1, 0, 2, 1, 0, 1, 0};            // smallest partial products first
    constexpr int PB[9] = {2, 1, 2, 2, 0, 1, 1, 0, 0};
    constexpr int P0 = NP == 1 ? 8 : 9 - NP;
    auto load_a = [&](int hidx, u32x4 (&af)[2][NSPL]) {
        const int step = hidx / HS, hb = (hidx % HS) * 2;
        const int tap = step >> 1, ks = step & 1;
        const int tapoff = ((tap / 3) * G::S + (tap % 3)) * PIXB + ks * 32;
#pragma unroll
        for (int o = 0; o < NSPL; ++o) {                          // in order of first use: hi, lo, mid
            const int sp = NSPL == 1 ? 0 : (o == 0 ? 0 : 3 - o);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i][sp] = *reinterpret_cast<const u32x4*>(smem + abase[hb + i] + (tapoff + sp * G::PLANE));
        }
    };
    for (int cc = 0; cc < cchunks; ++cc) {
        const bool more = cc + 1 < cchunks;
        if (more) issue_patch(cc + 1);
        u32x4 afb[2][2][NSPL];
        load_a(0, afb[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hidx = 0; hidx < NH; ++hidx) {
            const int step = hidx / HS, hb = (hidx % HS) * 2, slot = step % 3;
            const bool newstep = hidx % HS == 0;
            if (newstep) {                                        // weights of step+2 (possibly of the next chunk)
                if (step + 2 < 18) issue_b(cc, step + 2, (step + 2) % 3);
                else if (more) issue_b(cc + 1, step + 2 - 18, (step + 2) % 3);
            }
            if (hidx + 1 < NH) load_a(hidx + 1, afb[(hidx + 1) & 1]);
#pragma unroll
            for (int p = P0; p < 9; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[hb + i] = mfma_bf16(afb[hidx & 1][i][PA[p]], bq[slot][PB[p]], acc[hb + i]);
            // pin: (2 MFMA, 1 VMEM read)* then (2 MFMA, 1 DS read)*
            constexpr int NM = 2 * (9 - P0), NR = 2 * NSPL;
#pragma unroll
            for (int g = 0; g < NM / 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (g < NSPL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (g < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            __syncthreads();                                      // every wave is done reading the patch
            store_patch(cc + 1);
            __syncthreads();
        }
    }

    // ---- epilogue: D col = lane&31 (cout), D row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> pixel via row_to_pix ----
    const int n = n0 + wn * 32 + ml;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int ty, tx;
            pix_to_yx<TW>((wm * MB + i) * 32 + row_to_pix((r & 3) + 8 * (r >> 2) + 4 * kl), ty, tx);
            const int h = h0 + ty;
            if (h < H) y[(((size_t)img * H + h) * W + tx) * Cout + n] = acc[i][r];
        }
}

// (Cout,Cin,3,3) fp32 -> split bf16 planes in B-fragment order, for forward (K = Cin, N = Cout) and dgrad
// (K = Cout, N = Cin, taps mirrored).  Fragment (tap, kk, nb, s) = 64 lanes x 16 B at
// ((((tap*K/16 + kk) * N/32 + nb) * 3 + s) * 64 + lane); lane holds k = kk*16 + 8*(lane>>5) + e, n = nb*32 + (lane&31).
__global__ __launch_bounds__(256) void pack_weight_x3_kernel(const float* __restrict__ w, u32x4* __restrict__ wf,
                                                             u32x4* __restrict__ wd, int Cin, int Cout) {
    const long per_dir = (long)9 * (Cin / 16) * (Cout / 32) * 64;         // == 9 * (Cout/16) * (Cin/32) * 64
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < 2 * per_dir; idx += (long)gridDim.x * 256) {
        const bool dg = idx >= per_dir;
        long r = dg ? idx - per_dir : idx;
        const int K = dg ? Cout : Cin, N = dg ? Cin : Cout;
        const int lane = (int)(r & 63); r >>= 6;
        const int nb = (int)(r % (N / 32)); r /= (N / 32);
        const int kk = (int)(r % (K / 16));
        const int tap = (int)(r / (K / 16));
        const int nn = nb * 32 + (lane & 31), k0 = kk * 16 + 8 * (lane >> 5);
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + 2 * e + j;
                v[j] = dg ? w[((size_t)k * Cin + nn) * 9 + (8 - tap)] : w[((size_t)nn * Cin + k) * 9 + tap];
            }
            split_pack(v[0], v[1], h[e], m[e], l[e]);
        }
        u32x4* dst = (dg ? wd : wf) + ((((size_t)tap * (K / 16) + kk) * (N / 32) + nb) * 3) * 64 + lane;
        dst[0] = (u32x4){h[0], h[1], h[2], h[3]};
        dst[64] = (u32x4){m[0], m[1], m[2], m[3]};
        dst[128] = (u32x4){l[0], l[1], l[2], l[3]};
    }
}

static int x3_products() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TAG_X3_PRODUCTS");
        v = e ? atoi(e) : 6;
        if (v != 1 && v != 6 && v != 9) v = 6;
    }
    return v;
}

template <int MB, int TW, int NP>
void launch_x3(const float* x, const u32x4* wp, int pro, const float* s, const float* t, float* y, int B, int H, int W,
               int Cin, int Cout, hipStream_t st) {
    using G = X3Geom<TW>;
    constexpr int BN_ = MB == 4 ? 128 : 64;
    const int grid = B * ((H + G::TH - 1) / G::TH) * (Cout / BN_);
    const size_t lds = G::LDS_BYTES;
#define LAUNCH_PRO(P)                                                                                               \
    {                                                                                                               \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<MB, P, TW, NP>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((conv3x3_x3_kernel<MB, P, TW, NP>), dim3(grid), dim3(256), lds, st, x, wp, s, t, y, B, H, \
                           W, Cin, Cout);                                                                           \
    }
    switch (pro) {
        case 0: LAUNCH_PRO(0) break;
        case 1: LAUNCH_PRO(1) break;
        case 2: LAUNCH_PRO(2) break;
        default: LAUNCH_PRO(3) break;
    }
#undef LAUNCH_PRO
}

template <int MB, int NP>
void launch_x3_w(const float* x, const u32x4* wp, int pro, const float* s, const float* t, float* y, int B, int H, int W,
                 int Cin, int Cout, hipStream_t st) {
    if (W == 8) launch_x3<MB, 8, NP>(x, wp, pro, s, t, y, B, H, W, Cin, Cout, st);
    else if (W == 16) launch_x3<MB, 16, NP>(x, wp, pro, s, t, y, B, H, W, Cin, Cout, st);
    else if (W == 32) launch_x3<MB, 32, NP>(x, wp, pro, s, t, y, B, H, W, Cin, Cout, st);
    else launch_x3<MB, 64, NP>(x, wp, pro, s, t, y, B, H, W, Cin, Cout, st);
}

}  // namespace

extern "C" size_t tag_conv3x3_x3_pack_bytes(int Cin, int Cout) { return (size_t)9 * Cin * Cout * 3 * 2; }

extern "C" int tag_pack_conv_weight_x3(const float* w, void* wfwd, void* wdgrad, int Cin, int Cout, void* stream) {
    TAG_CHECK_ARG(w && wfwd && wdgrad && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0);
    const long n = (long)2 * 9 * (Cin / 16) * (Cout / 32) * 64;
    hipLaunchKernelGGL(pack_weight_x3_kernel, dim3(cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256)), dim3(256), 0,
                       as_stream(stream), w, reinterpret_cast<u32x4*>(wfwd), reinterpret_cast<u32x4*>(wdgrad), Cin, Cout);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_conv3x3_forward_x3(const float* x, const void* wpack, int prologue, const float* in_scale,
                                      const float* in_shift, float* y, int B, int H, int W, int Cin, int Cout,
                                      void* stream) {
    TAG_CHECK_ARG(x && wpack && y && B > 0 && H > 0);
    TAG_CHECK_ARG(W == 8 || W == 16 || W == 32 || W == 64);
    TAG_CHECK_ARG(Cin % 32 == 0 && Cout % 64 == 0 && Cin <= 512);
    TAG_CHECK_ARG((long)B * H * W * Cin * 4 < (1L << 32));      // 32-bit byte offsets inside the kernel
    TAG_CHECK_ARG(prologue >= 0 && prologue <= 3);
    TAG_CHECK_ARG(prologue == 0 || (in_scale && in_shift));
    hipStream_t st = as_stream(stream);
    const u32x4* wp = reinterpret_cast<const u32x4*>(wpack);
    const int np = x3_products();
#define BY_NP(MB)                                                                                        \
    if (np == 6) launch_x3_w<MB, 6>(x, wp, prologue, in_scale, in_shift, y, B, H, W, Cin, Cout, st);     \
    else if (np == 9) launch_x3_w<MB, 9>(x, wp, prologue, in_scale, in_shift, y, B, H, W, Cin, Cout, st); \
    else launch_x3_w<MB, 1>(x, wp, prologue, in_scale, in_shift, y, B, H, W, Cin, Cout, st);
    if (Cout % 128 == 0) { BY_NP(4) } else { BY_NP(2) }
#undef BY_NP
    TAG_LAUNCH_CHECK();
    return 0;
}
