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

// schedule pinning of the wgrad MFMA phase: 0 = fences between plane-steps only (measured best: 213 TFLOP/s),
// 1 = + sched_group_barrier interleave of reads and MFMAs (206), 2 = interleave without fences
#ifndef TAG_WX3_PIN
#define TAG_WX3_PIN 0
#endif
// k16 steps per chunk of the bf16-storage wgrad kernel (chunk = 16 x this many pixels)
#ifndef TAG_WX3_KS16
#define TAG_WX3_KS16 4
#endif
// 32-pixel MFMA blocks per wave of the bf16-storage launches on the 64-cout layers: 2 = 128 px x 64 co tiles, 4 = 256 px x 64 co.
// Measured at B = 64: 4 is SLOWER (conv forward+dgrad 5.24 vs 4.84 ms per step: the prologue variants need 176 VGPRs and lose
// a residency level), so 2 stays the default.
#ifndef TAG_X3_BF16_MB64
#define TAG_X3_BF16_MB64 2
#endif
// bf16-storage launches: stage the output tile through LDS and store 16-byte pieces (1) or store 4 bytes per lane directly (0)
#ifndef TAG_X3_LDS_EPI
#define TAG_X3_LDS_EPI 1
#endif
// ablation of the forward/dgrad kernel for tools/conv_bf16_bench.py (never set in the product build): 1 = no output stores,
// 2 = no statistics epilogue, 3 = no MFMAs (and hence no operand reads), 4 = 1 + 2, 5 = A fragments read from LDS only once per
// chunk, 6 = weight fragments loaded only in the prologue, 7 = 5 + 6 (MFMAs with no operand traffic)
#ifndef TAG_X3_ABL
#define TAG_X3_ABL 0
#endif
// depth of the weight-fragment ring of the ONE-product (plain bf16) forward/dgrad kernels (3, 6, 9 or 18)
#ifndef TAG_X3_RING1
#define TAG_X3_RING1 9
#endif

// -DTAG_X3_PROF (tools/conv_x3_prof.py, never in the product build): s_memtime deltas of the phases of ONE workgroup's wave 0
#ifdef TAG_X3_PROF
__device__ unsigned long long tag_x3_prof[10];
extern "C" int tag_debug_get_x3_prof(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(tag_x3_prof), 80) == hipSuccess ? 0 : -1; }
#define XP_MARK(i) { const unsigned long long p1_ = __builtin_amdgcn_s_memtime(); xpc[i] += p1_ - xp0; xp0 = p1_; }
#else
#define XP_MARK(i)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 32;            // channels per LDS chunk (2 k16 MFMA steps)
constexpr int PIXB = 80;          // bytes per pixel per plane in LDS

template <int TW, int TP = 128>                                   // TP = output pixels per workgroup tile
struct X3Geom {
    static constexpr int TH = TP / TW, PH = TH + 2;
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

// exact three-way split of two floats into packed bf16 pairs (low half = first value).  RNE = true (plain-bf16 mode,
// one product): only the hi plane is used and it is rounded to nearest-even instead of truncated.
__device__ __forceinline__ unsigned bf16_rne_bits(unsigned u) { return u + 0x7fffu + ((u >> 16) & 1u); }
template <bool RNE = false>
__device__ __forceinline__ void split_pack(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ab = __float_as_uint(a), bb = __float_as_uint(b);
    if (RNE) {
        h = __builtin_amdgcn_perm(bf16_rne_bits(bb), bf16_rne_bits(ab), 0x07060302u);
        m = l = 0u;
        return;
    }
    const float a1 = a - __uint_as_float(ab & 0xffff0000u), b1 = b - __uint_as_float(bb & 0xffff0000u);
    const unsigned a1b = __float_as_uint(a1), b1b = __float_as_uint(b1);
    const float a2 = a1 - __uint_as_float(a1b & 0xffff0000u), b2 = b1 - __uint_as_float(b1b & 0xffff0000u);
    h = __builtin_amdgcn_perm(bb, ab, 0x07060302u);
    m = __builtin_amdgcn_perm(b1b, a1b, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(b2), __float_as_uint(a2), 0x07060302u);
}

// value of lane ^ 1: DPP quad_perm [1,0,3,2] -- one VALU operation (__shfl_xor compiles to ds_bpermute_b32: an LDS-crossbar
// round trip per call; the bf16 epilogue makes 32-64 of them per lane and spent 10 k clocks per tile there, tools/conv_x3_prof.py)
__device__ __forceinline__ float lane_xor1(float x) {
    return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0xB1, 0xf, 0xf, true));
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MB = 32-pixel MFMA blocks per wave (4: wave = 128 px x 32 co, workgroup 128 x 128; 2: wave = 64 px x 32 co,
// workgroup 128 x 64).  NP = products per fp32 multiply (6 default, 9 exact, 1 plain bf16).
// TS = activation storage of x and y: float, or bf16_t (BASELINE configs[2]: bf16 tensors; NP == 1 only) -- then the
// patch is staged as (pixel, channel OCTET) items of 16 B, copied to LDS as they are when there is no prologue, and the
// output is rounded to bf16 (nearest-even) and stored as channel pairs.
// Epilogue operands of the bf16 dgrad launches (EPI == 1), as in conv.hip: the tensor whose BatchNorm+ReLU the gradient
// flows into next; the epilogue then writes the per-tile sums of g = da*[bn(yref) > 0] and g*xhat instead of output statistics.
struct BnBwdEpiX {
    const bf16_t* yref;     // (B,H,W,Cout) raw conv output saved by the forward pass (= BatchNorm input), bf16
    const float* scale;     // gamma * invstd
    const float* shift;     // beta - mean * gamma * invstd
    const float* mean;
    const float* invstd;
    // EPI == 2 only (conv.hip, EPI == 2): the gradient flows into relu(bn(yref)) -> ph x 2 pool -> dropout; yref is the UNPOOLED
    // (B,Hf,Wf,Cout) tensor
    int Hf, Wf, ph;
    float wavg, wmax, drop_p;
    unsigned long long seed;
};

// WM = waves along M: 1 (4 waves side by side along N: tile MB*32 px x 128 co) or 2 (2 x 2 waves: tile 2*MB*32 px x 64 co).
// MB = 2, WM = 2 is the 128 px x 64 co tile of the 64-cout layers; the bf16-storage launches take MB = 4, WM = 2
// (256 px x 64 co) there: with one product per operand pair a 128 x 64 tile is 2 chunks x 18 short steps of work behind a
// full HBM round trip (MFMA busy 0.17-0.20), the larger tile halves the fixed cost per output and the weight re-reads.
// (EPI == 2 on the 64-cout tiles asks for the 4 waves per SIMD the instance's main loop runs at -- 117 VGPRs -- so that the pool-sum
// epilogue, which alone would take 150, is fitted into that budget instead of costing the whole kernel a wave of occupancy)
template <int MB, int PRO, int TW, int NP, class TS = float, int EPI = 0, int WM = (MB == 4 ? 1 : 2)>
__global__ __launch_bounds__(256, (EPI == 2 && MB == 2) ? 4 : (EPI == 2 ? 3 : 2)) void conv3x3_x3_kernel(const TS* __restrict__ x, const u32x4* __restrict__ wp,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift, TS* __restrict__ y,
                                                            float* __restrict__ stats, BnBwdEpiX epi, int B, int H, int W,
                                                            int Cin, int Cout) {
    using G = X3Geom<TW, WM * MB * 32>;
    constexpr bool HS16 = Act<TS>::is_bf16;
    static_assert(!HS16 || NP == 1, "bf16 storage goes with the one-product arithmetic");
    constexpr int BN_ = (4 / WM) * 32;
    constexpr int NSPL = NP == 1 ? 1 : 3;                         // planes actually read
    constexpr int QPP = HS16 ? 4 : 8;                             // staging items per pixel: octets (16 B bf16) | quads (16 B fp32)
    constexpr int QSH = HS16 ? 2 : 3;
    constexpr int ITEMS = (G::PH * (TW + 2) * QPP + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Ss = reinterpret_cast<float*>(smem + NSPL * G::PLANE); // [2][Cin]  (one-product kernels allocate ONE plane)

    const int n_tiles = Cout / BN_;
    const int row_tiles = (H + G::TH - 1) / G::TH;
    const int m_tiles = B * row_tiles;
    const int L = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int n0 = (L % n_tiles) * BN_;
    const int mt = L / n_tiles;
    const int img = mt / row_tiles, h0 = (mt % row_tiles) * G::TH;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = WM == 1 ? 0 : (wid >> 1);                      // M group of the wave
    const int wn = WM == 1 ? wid : (wid & 1);                     // 32-cout block of the wave
    const int kl = lane >> 5, ml = lane & 31;
    if (PRO != 0)
        for (int c = tid; c < Cin; c += 256) { Ss[c] = in_scale[c]; Ss[Cin + c] = in_shift[c]; }

    // ---- patch staging geometry: item = (patch pixel, channel quad | octet) ----
    const int q = tid & (QPP - 1);
    unsigned poff[ITEMS];
    unsigned pvalid = 0, pexist = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = tid + 256 * i;
        const int pp = idx >> QSH;                                // 0 .. PH*(TW+2)-1 (dense numbering of real patch pixels)
        const int pr = pp / (TW + 2), pc = pp - pr * (TW + 2);
        const int h = h0 - 1 + pr, w = pc - 1;
        const bool ex = pp < G::PH * (TW + 2);
        const bool ok = ex & ((unsigned)h < (unsigned)H) & ((unsigned)w < (unsigned)W);
        pexist |= (unsigned)ex << i;
        pvalid |= (unsigned)ok << i;
        const long pix = ok ? (long)h * W + w : 0;        // inside the image: 32-bit byte offsets hold any batch (round 4)
        poff[i] = HS16 ? (unsigned)((pix * Cin + q * 8) * 2) : (unsigned)((pix * Cin + q * 4) * 4);
    }
    const char* ximg = reinterpret_cast<const char*>(x) + (size_t)img * H * W * Cin * sizeof(TS);   // wave-uniform 64-bit image base
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

    u32x4 ra[ITEMS];                                              // 16 B per item: 4 fp32 or 8 bf16
    auto issue_patch = [&](int cc) {
        const unsigned coff = (unsigned)(cc * KC * (HS16 ? 2 : 4));
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            ra[i] = *reinterpret_cast<const u32x4*>(ximg + (poff[i] + coff));
    };
    auto store_patch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            if (!((pexist >> i) & 1u)) continue;
            const int pp = (tid + 256 * i) >> QSH;
            const int pr = pp / (TW + 2), pc = pp - pr * (TW + 2);
            const bool valid = (pvalid >> i) & 1u;
            if constexpr (HS16) {
                u32x4 o = ra[i];
                if (PRO != 0) {
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(Ss + cc * KC + q * 8);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(Ss + cc * KC + q * 8 + 4);
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(Ss + Cin + cc * KC + q * 8);
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(Ss + Cin + cc * KC + q * 8 + 4);
                    o.x = tag_pack_bf16(prologue1(tag_bf16_lo(ra[i].x), PRO, s0.x, t0.x), prologue1(tag_bf16_hi(ra[i].x), PRO, s0.y, t0.y));
                    o.y = tag_pack_bf16(prologue1(tag_bf16_lo(ra[i].y), PRO, s0.z, t0.z), prologue1(tag_bf16_hi(ra[i].y), PRO, s0.w, t0.w));
                    o.z = tag_pack_bf16(prologue1(tag_bf16_lo(ra[i].z), PRO, s1.x, t1.x), prologue1(tag_bf16_hi(ra[i].z), PRO, s1.y, t1.y));
                    o.w = tag_pack_bf16(prologue1(tag_bf16_lo(ra[i].w), PRO, s1.z, t1.z), prologue1(tag_bf16_hi(ra[i].w), PRO, s1.w, t1.w));
                }
                if (!valid) o = (u32x4){0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(smem + (pr * G::S + pc) * PIXB + q * 16) = o;
            } else {
                f32x4 rs = {1.0f, 1.0f, 1.0f, 1.0f}, rt = {0.0f, 0.0f, 0.0f, 0.0f};
                if (PRO != 0) {
                    rs = *reinterpret_cast<const f32x4*>(Ss + cc * KC + q * 4);
                    rt = *reinterpret_cast<const f32x4*>(Ss + Cin + cc * KC + q * 4);
                }
                const f32x4 rv = __builtin_bit_cast(f32x4, ra[i]);
                f32x4 v;
                v.x = prologue1(rv.x, PRO, rs.x, rt.x);
                v.y = prologue1(rv.y, PRO, rs.y, rt.y);
                v.z = prologue1(rv.z, PRO, rs.z, rt.z);
                v.w = prologue1(rv.w, PRO, rs.w, rt.w);
                if (!valid) v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                unsigned h0_, m0_, l0_, h1_, m1_, l1_;
                split_pack<NP == 1>(v.x, v.y, h0_, m0_, l0_);
                split_pack<NP == 1>(v.z, v.w, h1_, m1_, l1_);
                unsigned char* dst = smem + (pr * G::S + pc) * PIXB + q * 8;
                *reinterpret_cast<u32x2*>(dst) = (u32x2){h0_, h1_};
                if (NSPL == 3) {
                    *reinterpret_cast<u32x2*>(dst + G::PLANE) = (u32x2){m0_, m1_};
                    *reinterpret_cast<u32x2*>(dst + 2 * G::PLANE) = (u32x2){l0_, l1_};
                }
            }
        }
    };

    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // weight fragments: ring of RING k16 steps (18 steps per chunk = a whole number of turns), loaded RING-1 steps ahead.
    // With six products a step lasts ~770 MFMA cycles and 2 steps of distance cover the L2 latency; with ONE product a step is
    // 128 cycles, so the one-product kernels keep 8 steps (~1000 cycles) in flight.
    constexpr int RING = NP == 1 ? TAG_X3_RING1 : 3;
    static_assert(18 % RING == 0, "ring must divide the 18 steps of a chunk");
    u32x4 bq[RING][NSPL];
    auto issue_b = [&](int cc, int step, int slot) {              // step = tap*2 + ks inside chunk cc
        const int tap = step >> 1, ks = step & 1;
        const u32x4* p = wlane + ((size_t)(tap * KK + cc * 2 + ks) * NBK) * 192;
#pragma unroll
        for (int s = 0; s < NSPL; ++s) bq[slot][s] = p[s * 64];
    };

    const int cchunks = Cin / KC;
#ifdef TAG_X3_PROF   // 0 prologue, 1 MFMA loop, 2 barrier, 3 store_patch, 4 barrier, 5 pre-epilogue barrier, 6 pack + LDS write, 7 16-B stores, 8 statistics
    unsigned long long xpc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xp0 = __builtin_amdgcn_s_memtime();
#endif
    issue_patch(0);
#pragma unroll
    for (int st = 0; st < RING - 1; ++st) issue_b(0, st, st);
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
    XP_MARK(0)
    for (int cc = 0; cc < cchunks; ++cc) {
        const bool more = cc + 1 < cchunks;
        if (more) issue_patch(cc + 1);
        u32x4 afb[2][2][NSPL];
        load_a(0, afb[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hidx = 0; hidx < NH; ++hidx) {
            const int step = hidx / HS, hb = (hidx % HS) * 2, slot = step % RING;
            const bool newstep = hidx % HS == 0;
            if (newstep) {                                        // weights of step+RING-1 (possibly of the next chunk)
                constexpr int D = RING - 1;
                if (TAG_X3_ABL != 6 && TAG_X3_ABL != 7) {
                    if (step + D < 18) issue_b(cc, step + D, (step + D) % RING);
                    else if (more) issue_b(cc + 1, step + D - 18, (step + D) % RING);
                }
            }
            if (hidx + 1 < NH && ((TAG_X3_ABL != 5 && TAG_X3_ABL != 7) || hidx == 0)) load_a(hidx + 1, afb[(hidx + 1) & 1]);
#pragma unroll
            for (int p = P0; p < 9; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (TAG_X3_ABL != 3) acc[hb + i] = mfma_bf16(afb[hidx & 1][i][PA[p]], bq[slot][PB[p]], acc[hb + i]);
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
        XP_MARK(1)
        if (more) {
            __syncthreads();                                      // every wave is done reading the patch
            XP_MARK(2)
            store_patch(cc + 1);
            XP_MARK(3)
            __syncthreads();
            XP_MARK(4)
        }
    }

    // ---- epilogue: D col = lane&31 (cout), D row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> pixel via row_to_pix ----
    const int n = n0 + wn * 32 + ml;
    bool okrow[MB][16];
    // bf16 storage: the tile goes through LDS ([pixel][BN_ channels], the patch buffer is free by now) and leaves as 16-byte
    // pieces, consecutive threads = consecutive channel octets of a pixel (128 / 256 contiguous bytes per pixel, whole rows of the
    // 64-cout layers) -- the direct form below writes 4 bytes per lane and costs 18 % of the one-product kernel's time
    // (tools/conv_bf16_bench.py, ablation 1)
    constexpr bool LDS_EPI = HS16 && (TAG_X3_LDS_EPI != 0);
    constexpr int OROWB = BN_ * 2 + 16;                           // bytes per pixel row of the staged tile
    if constexpr (LDS_EPI) __syncthreads();                       // every wave is done reading the patch
    XP_MARK(5)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int ty, tx;
            const int pm = (wm * MB + i) * 32 + row_to_pix((r & 3) + 8 * (r >> 2) + 4 * kl);
            pix_to_yx<TW>(pm, ty, tx);
            const int h = h0 + ty;
            okrow[i][r] = h < H;
            if constexpr (LDS_EPI) {
                // every lane stores its own element as 2 bytes: no partner lane, no select, no exec-masked branch per element
                *reinterpret_cast<unsigned short*>(smem + pm * OROWB + (wn * 32 + ml) * 2) =
                    (unsigned short)tag_pack_bf16(acc[i][r], 0.0f);
            } else if constexpr (HS16) {
                // channel pairs: even lanes store (n, n+1) of the even rows r, odd lanes (n-1, n) of the odd rows
                const float other = lane_xor1(acc[i][r]);
                const bool mine = ((r ^ ml) & 1) == 0;
                const unsigned w2 = (ml & 1) ? tag_pack_bf16(other, acc[i][r]) : tag_pack_bf16(acc[i][r], other);
                if (TAG_X3_ABL != 1 && TAG_X3_ABL != 4 && mine && h < H)
                    *reinterpret_cast<unsigned*>(y + (((size_t)img * H + h) * W + tx) * Cout + (n & ~1)) = w2;
            } else {
                if (h < H) y[(((size_t)img * H + h) * W + tx) * Cout + n] = acc[i][r];
            }
        }
    XP_MARK(6)
    if constexpr (LDS_EPI) {
        __syncthreads();
        constexpr int TP = WM * MB * 32, PPP = BN_ / 8;           // pixels per tile, 16-byte pieces per pixel
#pragma unroll
        for (int k = 0; k < TP * PPP / 256; ++k) {
            const int piece = tid + 256 * k;
            const int pm = piece / PPP, c8 = piece % PPP;
            int ty, tx;
            pix_to_yx<TW>(pm, ty, tx);
            const int h = h0 + ty;
            if (TAG_X3_ABL != 1 && TAG_X3_ABL != 4 && h < H)
                *reinterpret_cast<u32x4*>(y + (((size_t)img * H + h) * W + tx) * Cout + n0 + c8 * 8) =
                    *reinterpret_cast<const u32x4*>(smem + pm * OROWB + c8 * 16);
        }
    }
    // ---- EPI == 2 (bf16 storage; the dgrad launch of a block's FIRST conv, whose output is the gradient of the pooled output of the
    // block below): the reduction half of the backward of relu(bn(yref)) -> avg/max pool (ph x 2) -> dropout (conv.hip, EPI == 2;
    // what pool_bwd_reduce_kernel computes in a pass of its own).  The output tile is still in LDS as the bf16 values the apply pass
    // will read back from HBM: a thread takes (pixel, channel octet) pieces of it -- its octet is FIXED (256 % pieces-per-pixel == 0),
    // so the 8 + 8 running sums stay in registers over its pieces --, undoes the dropout (two hashes per piece), loads the ph x 2
    // window of yref as 16-byte pieces, recomputes a = bn(yref), ReLU mask and first-maximum arg-max, and accumulates sum(dz),
    // sum(dz * xhat); the threads of an octet are folded through LDS in a fixed order: ONE partial row per workgroup m-tile. ----
    if constexpr (EPI == 2) {
        static_assert(LDS_EPI, "the pool-backward sums read the staged bf16 tile");
        constexpr int TP = WM * MB * 32, PPP = BN_ / 8, NPC = TP * PPP / 256;
        const int c8 = tid % PPP, nb = n0 + c8 * 8;
        float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = epi.scale[nb + e]; sh[e] = epi.shift[nb + e]; mu[e] = epi.mean[nb + e]; is[e] = epi.invstd[nb + e];
            s1[e] = 0.0f; s2[e] = 0.0f;
        }
        const int Hf = epi.Hf, Wf = epi.Wf, ph = epi.ph;
        const size_t rowf = (size_t)Wf * Cout;
        const bf16_t* ybase = epi.yref + (size_t)img * Hf * rowf + nb;
        const bool drop = epi.drop_p > 0.0f;
        const float keep_scale = drop ? 1.0f / (1.0f - epi.drop_p) : 1.0f;
        const unsigned keep_thr = tag_keep4_threshold(epi.drop_p);
        auto bfe = [](const u32x4& w, int e) {                    // element e (0..7) of 8 packed bf16 (e is a compile-time constant)
            const unsigned d = w[e >> 1];
            return (e & 1) ? tag_bf16_hi(d) : tag_bf16_lo(d);
        };
        // pieces per iteration: 2 (8 window loads in flight) where the main loop's own register count leaves room for them at the
        // same occupancy (128-cout tiles: 143 VGPRs, 3 waves per SIMD up to 170), 1 for the 64-cout tiles (117 VGPRs: 4 waves up to 128)
        constexpr int PB = MB == 4 ? 2 : 1;
#pragma unroll 1
        for (int k2 = 0; k2 < NPC; k2 += PB) {                    // (a real loop, not unrolled: registers)
            u32x4 vw[PB][4], gq[PB];
            int hh[PB], ww[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int piece = tid + 256 * (k2 + u);
                const int pm = piece / PPP;
                int ty, tx;
                pix_to_yx<TW>(pm, ty, tx);
                hh[u] = h0 + ty; ww[u] = tx;
                const int hc = hh[u] < H ? hh[u] : H - 1;
                gq[u] = *reinterpret_cast<const u32x4*>(smem + pm * OROWB + c8 * 16);
                if (hh[u] >= H) gq[u] = (u32x4){0u, 0u, 0u, 0u};
                const bf16_t* pw = ybase + (size_t)(hc * ph) * rowf + (size_t)(2 * tx) * Cout;
                vw[u][0] = *reinterpret_cast<const u32x4*>(pw);
                vw[u][1] = *reinterpret_cast<const u32x4*>(pw + Cout);
                if (ph == 2) {
                    vw[u][2] = *reinterpret_cast<const u32x4*>(pw + rowf);
                    vw[u][3] = *reinterpret_cast<const u32x4*>(pw + rowf + Cout);
                } else { vw[u][2] = (u32x4){0u, 0u, 0u, 0u}; vw[u][3] = vw[u][2]; }
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                uint64_t bits[2] = {~0ull, ~0ull};
                if (drop) {
                    const size_t oi = (((size_t)img * H + hh[u]) * W + ww[u]) * Cout + nb;
                    bits[0] = tag_keep4_bits(epi.seed, (uint64_t)(oi >> 2));
                    bits[1] = tag_keep4_bits(epi.seed, (uint64_t)(oi >> 2) + 1);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float g = bfe(gq[u], e);
                    if (drop) g = tag_keep4(bits[e >> 2], e & 3, keep_thr) ? g * keep_scale : 0.0f;
                    float v[4], a[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[k] = bfe(vw[u][k], e); a[k] = fmaf(v[k], sc[e], sh[e]); }
                    if (ph != 2) { a[2] = -INFINITY; a[3] = -INFINITY; }
                    const float mx = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
                    const float gw = g * epi.wavg, gwm = g * (epi.wavg + epi.wmax);
                    bool found = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool eq = a[k] == mx;
                        const bool hit = eq && !found;
                        found = found || eq;
                        const float dz = a[k] > 0.0f ? (hit ? gwm : gw) : 0.0f;
                        s1[e] += dz;
                        s2[e] = fmaf(dz, (v[k] - mu[e]) * is[e], s2[e]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                    // the next pair's loads stay behind this pair's arithmetic (registers)
        }
        __syncthreads();                                          // every thread is done reading the staged tile
        float* red = reinterpret_cast<float*>(smem);              // [256][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
        __syncthreads();
        if (tid < BN_) {
            const int oc = tid >> 3, e = tid & 7;
            float a = 0.0f, b = 0.0f;
            for (int j = 0; j < 256 / PPP; ++j) { a += red[(j * PPP + oc) * 16 + e]; b += red[(j * PPP + oc) * 16 + 8 + e]; }
            float* ps = stats + (size_t)mt * 2 * Cout;
            ps[n0 + tid] = a;
            ps[Cout + n0 + tid] = b;
        }
    }
    XP_MARK(7)
    // ---- EPI == 1: the reduction half of the BatchNorm+ReLU backward this gradient flows into (conv.hip, EPI == 1): per wave
    // M-group and channel sum(g) and sum(g * xhat), g taken from the fp32 accumulators; rows [prow][2][Cout] ----
    if (EPI == 1) {
        constexpr int MG = WM;
        const int prow = mt * MG + wm;
        float* ps = stats + (size_t)prow * 2 * Cout;
        const float sc = epi.scale[n], sh = epi.shift[n], mu = epi.mean[n], is = epi.invstd[n];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            float yv[16];                                         // 16 loads in flight, then their arithmetic
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int ty, tx;
                pix_to_yx<TW>((wm * MB + i) * 32 + row_to_pix((r & 3) + 8 * (r >> 2) + 4 * kl), ty, tx);
                const int h = h0 + ty, hc = h < H ? h : H - 1;
                yv[r] = Act<bf16_t>::ld1(epi.yref + (((size_t)img * H + hc) * W + tx) * Cout + n);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g = (okrow[i][r] && fmaf(yv[r], sc, sh) > 0.0f) ? acc[i][r] : 0.0f;
                s1 += g;
                s2 = fmaf(g, (yv[r] - mu) * is, s2);
            }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (kl == 0) { ps[n] = s1; ps[Cout + n] = s2; }
    }
    // ---- fused BatchNorm statistics (see conv.hip): one partial row per wave M-group (MB * 32 pixels) ----
    if (EPI == 0 && stats && TAG_X3_ABL != 2 && TAG_X3_ABL != 4) {
        constexpr int MG = WM;
        const int prow = mt * MG + wm;
        float* ps = stats + (size_t)prow * 3 * Cout;
        float cnt = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { cnt += okrow[i][r] ? 1.0f : 0.0f; s1 += okrow[i][r] ? acc[i][r] : 0.0f; }
        cnt += __shfl_xor(cnt, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        const float mu = s1 / fmaxf(cnt, 1.0f);
        float r1 = 0.0f, q = 0.0f;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = okrow[i][r] ? acc[i][r] - mu : 0.0f;
                r1 += d;
                q = fmaf(d, d, q);
            }
        r1 += __shfl_xor(r1, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (kl == 0) { ps[n] = mu; ps[Cout + n] = r1; ps[2 * Cout + n] = q; }
        if (n0 == 0 && wn == 0 && lane == 0) stats[(size_t)m_tiles * MG * 3 * Cout + prow] = cnt;
    }
#ifdef TAG_X3_PROF
    XP_MARK(8)
    if (blockIdx.x == 1500 && tid == 0) for (int i = 0; i < 9; ++i) tag_x3_prof[i] = xpc[i];
#endif
}

// (Cout,Cin,3,3) fp32 -> split bf16 planes in B-fragment order, for forward (K = Cin, N = Cout) and dgrad
// (K = Cout, N = Cin, taps mirrored).  Fragment (tap, kk, nb, s) = 64 lanes x 16 B at
// ((((tap*K/16 + kk) * N/32 + nb) * 3 + s) * 64 + lane); lane holds k = kk*16 + 8*(lane>>5) + e, n = nb*32 + (lane&31).
__global__ __launch_bounds__(256) void pack_weight_x3_kernel(const float* __restrict__ w, u32x4* __restrict__ wf,
                                                             u32x4* __restrict__ wd, int Cin, int Cout, int rne) {
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
            if (rne) split_pack<true>(v[0], v[1], h[e], m[e], l[e]);
            else split_pack<false>(v[0], v[1], h[e], m[e], l[e]);
        }
        u32x4* dst = (dg ? wd : wf) + ((((size_t)tap * (K / 16) + kk) * (N / 32) + nb) * 3) * 64 + lane;
        dst[0] = (u32x4){h[0], h[1], h[2], h[3]};
        dst[64] = (u32x4){m[0], m[1], m[2], m[3]};
        dst[128] = (u32x4){l[0], l[1], l[2], l[3]};
    }
}


// ------------------------------------------------------------------------------------------
// wgrad with the same arithmetic: partial[split][tap][ci][co] = sum_{pixels of the split} X[p + tap][ci] * dY[p][co].
// GEMM with M = ci, N = co, K = pixels: both operands are consumed K-major while LDS (like HBM) holds them
// pixel-major, so the fragments are fetched with the gfx950 transposing read ds_read_b64_tr_b16 (a 16-lane group
// reads 4 pixels x 16 channels and each lane receives the 4 pixels of its channel).  A workgroup owns 64 ci x 64 co of
// all nine taps (4 waves x 32x32x9 = 144 accumulator registers) and walks DOWN one column strip of one image in
// chunks of 32 pixels (CH rows x CW columns): the input rows live in a ring of 2*CH+2 row slots, so each chunk
// stages only its CH new rows (not the whole (CH+2)-row halo patch), into slots the running chunk does not read --
// one barrier per chunk, staging of chunk c+1 overlaps the MFMAs of chunk c.
// LDS image: X planes [split 3][ci block 2][ring row][PW pixels][32 ch bf16 = 64 B]; dY planes [2 buffers][split 3]
// [co block 2][32 pixels][64 B]: the 4 pixels x 64 B of a transposing read are contiguous -> conflict-free.
// ------------------------------------------------------------------------------------------
// KS = k16 MFMA steps per chunk (chunk = 16 KS pixels): 2 for the split arithmetics; the one-product kernels take 4 (a chunk
// then carries 36 MFMAs per wave instead of 18 between two barriers / two rounds of index arithmetic -- with one product
// per operand pair the 32-pixel chunk was barrier- and VALU-bound: MFMA busy 0.28).  NSP = split planes held in LDS.
template <int TW, int KS = 2, int NSP = 3>
struct WX3Geom {
    static constexpr int CPX = 16 * KS;
    static constexpr int CW = TW >= 32 ? 32 : TW, CH = CPX / CW, PW = CW + 2, R = 2 * CH + 2;
    static constexpr int XROWB = PW * 64;                                       // bytes of one ring row in one plane
    static constexpr int XPL = R * XROWB + ((R * XROWB) % 256 == 0 ? 128 : 0);  // plane stride = 128 (mod 256)
    static constexpr int YPL = CPX * 64;
    static constexpr int XBYTES = 2 * NSP * XPL, YBYTES = 2 * NSP * YPL;        // per dY buffer
    static constexpr int LDS_BYTES = XBYTES + 2 * YBYTES;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 lds_tr_read(const unsigned char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(u32x2, v);
}

template <int TW, int PRO, int NP, class TS = float>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_x3_kernel(const TS* __restrict__ x,
                                                                  const float* __restrict__ in_scale,
                                                                  const float* __restrict__ in_shift,
                                                                  const TS* __restrict__ dy,
                                                                  float* __restrict__ partial, int B, int H, int W,
                                                                  int Cin, int Cout, int splits, int chunks_per_split) {
    constexpr int NSPL = NP == 1 ? 1 : 3;
    constexpr bool HS16 = Act<TS>::is_bf16;                      // bf16 tensors: items are (pixel, channel OCTET) of 16 B
    constexpr int KS = HS16 ? TAG_WX3_KS16 : 2;                  // k16 steps per chunk (fp32 staging registers do not fit more)
    using G = WX3Geom<TW, KS, NSPL>;
    constexpr int CW = G::CW, CH = G::CH, PW = G::PW, R = G::R;
    static_assert(!HS16 || NP == 1, "bf16 storage goes with the one-product arithmetic");
    constexpr int QPP = HS16 ? 8 : 16, QSH = HS16 ? 3 : 4;       // items per pixel (64 channels)
    constexpr int XITEMS = (CH * PW * QPP + 255) / 256, DITEMS = G::CPX * QPP / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xs = smem;
    unsigned char* Ys = smem + G::XBYTES;

    const int ci_tiles = Cin / 64, co_tiles = Cout / 64;
    int L = xcd_remap(blockIdx.x, ci_tiles * co_tiles * splits);
    const int cot = L % co_tiles; L /= co_tiles;
    const int cit = L % ci_tiles; L /= ci_tiles;
    const int split = L;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int rb_per_img = (H + CH - 1) / CH, cb_per_row = TW / CW;
    const int chunks_total = B * rb_per_img * cb_per_row;
    const int cbeg = split * chunks_per_split;
    int cend = cbeg + chunks_per_split;
    if (cend > chunks_total) cend = chunks_total;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = wid >> 1, wj = wid & 1;                       // ci / co 32-block of the wave
    const int kl = lane >> 5, half = (lane >> 4) & 1, li = lane & 15;

    // ---- staging geometry: item = (pixel, channel quad | octet); LDS block = 32 channels = 64 B per pixel ----
    const int quad = tid & (QPP - 1);
    const int ca = ci0 + quad * (HS16 ? 8 : 4), cb = co0 + quad * (HS16 ? 8 : 4);
    const unsigned qoff = HS16 ? (unsigned)(quad >> 2) : (unsigned)(quad >> 3);
    const unsigned qbyte = HS16 ? (unsigned)((quad & 3) * 16) : (unsigned)((quad & 7) * 8);
    f32x4 rs = {1.0f, 1.0f, 1.0f, 1.0f}, rt = {0.0f, 0.0f, 0.0f, 0.0f}, rs1 = rs, rt1 = rt;
    if (PRO != 0) {
        rs = *reinterpret_cast<const f32x4*>(in_scale + ca); rt = *reinterpret_cast<const f32x4*>(in_shift + ca);
        if (HS16) { rs1 = *reinterpret_cast<const f32x4*>(in_scale + ca + 4); rt1 = *reinterpret_cast<const f32x4*>(in_shift + ca + 4); }
    }

    u32x4 rx[XITEMS], rd[DITEMS];
    unsigned xok = 0, dok = 0;
    // chunk index -> (img, h0, w0); row blocks run fastest so that consecutive chunks walk down a strip.  The main loop touches
    // chunks c, c+1, c+2 per iteration: their origins are kept in a 3-entry window advanced by increments (two runtime integer
    // divisions per call were a visible share of the one-product kernel's issue time); anything else divides.
    auto origin_div = [&](int c, int& img, int& h0, int& w0) {
        const int rbk = c % rb_per_img; const int t = c / rb_per_img;
        const int cbk = t % cb_per_row; img = t / cb_per_row;
        h0 = rbk * CH; w0 = cbk * CW;
    };
    int ob = -4, o_img[3] = {0, 0, 0}, o_h[3] = {0, 0, 0}, o_w[3] = {0, 0, 0};
    auto origin_next = [&](int img, int h0, int w0, int& img2, int& h2, int& w2) {
        h2 = h0 + CH; w2 = w0; img2 = img;
        if (h2 >= rb_per_img * CH) {
            h2 = 0; w2 = w0 + CW;
            if (w2 >= TW) { w2 = 0; img2 = img + 1; }
        }
    };
    auto window_set = [&](int c) {
        ob = c;
        origin_div(c, o_img[0], o_h[0], o_w[0]);
        origin_next(o_img[0], o_h[0], o_w[0], o_img[1], o_h[1], o_w[1]);
        origin_next(o_img[1], o_h[1], o_w[1], o_img[2], o_h[2], o_w[2]);
    };
    auto window_advance = [&]() {
        ++ob;
        o_img[0] = o_img[1]; o_h[0] = o_h[1]; o_w[0] = o_w[1];
        o_img[1] = o_img[2]; o_h[1] = o_h[2]; o_w[1] = o_w[2];
        origin_next(o_img[1], o_h[1], o_w[1], o_img[2], o_h[2], o_w[2]);
    };
    auto origin = [&](int c, int& img, int& h0, int& w0) {
        const int d = c - ob;
        if (d == 0) { img = o_img[0]; h0 = o_h[0]; w0 = o_w[0]; }
        else if (d == 1) { img = o_img[1]; h0 = o_h[1]; w0 = o_w[1]; }
        else if (d == 2) { img = o_img[2]; h0 = o_h[2]; w0 = o_w[2]; }
        else origin_div(c, img, h0, w0);
    };
    // ---- chunk-invariant staging geometry of this thread's items, computed ONCE (the per-chunk form re-derived row / column /
    // 64-bit addresses of every item at every chunk: 619 VALU instructions per 64-pixel chunk against 36 MFMAs -- the staging
    // phase was 52-57 % of the one-product kernel's time, tools/conv_x3_prof.py).  Per chunk an item now costs two adds and
    // two compares for its validity and one 32-bit add for its address.
    constexpr unsigned ESZ = HS16 ? 2u : 4u;                     // bytes per stored element of x / dy
    int x_pr[XITEMS], x_pc[XITEMS];
    unsigned x_rel[XITEMS], x_dst0[XITEMS], x_ex = 0;
#pragma unroll
    for (int i = 0; i < XITEMS; ++i) {
        const int pp = (tid + 256 * i) >> QSH;
        x_pr[i] = pp / PW; x_pc[i] = pp - x_pr[i] * PW;
        x_ex |= (unsigned)(pp < CH * PW) << i;
        x_rel[i] = (unsigned)((x_pr[i] * W + x_pc[i]) * Cin + ca) * ESZ;          // byte offset relative to pixel (first_row, w0 - 1)
        x_dst0[i] = qoff * (unsigned)G::XPL + (unsigned)x_pc[i] * 64u + qbyte;
    }
    unsigned d_rel[DITEMS], d_dst0[DITEMS];
    int d_kh[DITEMS];
#pragma unroll
    for (int i = 0; i < DITEMS; ++i) {
        const int k = (tid + 256 * i) >> QSH;
        d_kh[i] = k / CW;
        d_rel[i] = (unsigned)((d_kh[i] * W + k % CW) * Cout + cb) * ESZ;           // relative to pixel (h0, w0)
        d_dst0[i] = qoff * (unsigned)G::YPL + (unsigned)k * 64u + qbyte;
    }
    auto load_rows = [&](int img, int first_row, int w0) {       // CH input rows [first_row, first_row+CH), PW columns
        const char* xb = reinterpret_cast<const char*>(x) + (size_t)img * H * W * Cin * ESZ;     // wave-uniform image base
        const int roff = ((first_row * W + w0 - 1) * Cin) * (int)ESZ;                           // may be negative (halo rows)
        xok = 0;
#pragma unroll
        for (int i = 0; i < XITEMS; ++i) {
            const unsigned ok = ((x_ex >> i) & 1u) & (unsigned)((unsigned)(first_row + x_pr[i]) < (unsigned)H) &
                                (unsigned)((unsigned)(w0 - 1 + x_pc[i]) < (unsigned)W);
            xok |= ok << i;
            const unsigned off = ok ? (unsigned)(roff + (int)x_rel[i]) : (unsigned)ca * ESZ;
            rx[i] = *reinterpret_cast<const u32x4*>(xb + off);
        }
    };
    auto store_rows = [&](int first_row, int last_wanted) {      // rows > last_wanted are not stored (priming overshoot)
        const int slot_first = (first_row + 4 * R) % R;          // wave-uniform
#pragma unroll
        for (int i = 0; i < XITEMS; ++i) {
            if (!((x_ex >> i) & 1u)) continue;
            const int row = first_row + x_pr[i];
            if (row > last_wanted) continue;
            int slot = slot_first + x_pr[i];
            slot = slot >= R ? slot - R : slot;
            const bool valid = (xok >> i) & 1u;
            unsigned char* dst = Xs + x_dst0[i] + (unsigned)slot * (unsigned)(PW * 64);
            if constexpr (HS16) {
                u32x4 o = rx[i];
                if (PRO != 0) {
                    o.x = tag_pack_bf16(prologue1(tag_bf16_lo(rx[i].x), PRO, rs.x, rt.x), prologue1(tag_bf16_hi(rx[i].x), PRO, rs.y, rt.y));
                    o.y = tag_pack_bf16(prologue1(tag_bf16_lo(rx[i].y), PRO, rs.z, rt.z), prologue1(tag_bf16_hi(rx[i].y), PRO, rs.w, rt.w));
                    o.z = tag_pack_bf16(prologue1(tag_bf16_lo(rx[i].z), PRO, rs1.x, rt1.x), prologue1(tag_bf16_hi(rx[i].z), PRO, rs1.y, rt1.y));
                    o.w = tag_pack_bf16(prologue1(tag_bf16_lo(rx[i].w), PRO, rs1.z, rt1.z), prologue1(tag_bf16_hi(rx[i].w), PRO, rs1.w, rt1.w));
                }
                if (!valid) o = (u32x4){0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(dst) = o;
            } else {
                const f32x4 rv = __builtin_bit_cast(f32x4, rx[i]);
                f32x4 v;
                v.x = prologue1(rv.x, PRO, rs.x, rt.x);
                v.y = prologue1(rv.y, PRO, rs.y, rt.y);
                v.z = prologue1(rv.z, PRO, rs.z, rt.z);
                v.w = prologue1(rv.w, PRO, rs.w, rt.w);
                if (!valid) v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                unsigned h0_, m0_, l0_, h1_, m1_, l1_;
                split_pack<NP == 1>(v.x, v.y, h0_, m0_, l0_);
                split_pack<NP == 1>(v.z, v.w, h1_, m1_, l1_);
                *reinterpret_cast<u32x2*>(dst) = (u32x2){h0_, h1_};
                if (NSPL == 3) {
                    *reinterpret_cast<u32x2*>(dst + 2 * G::XPL) = (u32x2){m0_, m1_};
                    *reinterpret_cast<u32x2*>(dst + 4 * G::XPL) = (u32x2){l0_, l1_};
                }
            }
        }
    };
    auto load_dy = [&](int img, int h0, int w0) {
        const char* db = reinterpret_cast<const char*>(dy) + (size_t)img * H * W * Cout * ESZ;
        const unsigned roff = (unsigned)((h0 * W + w0) * Cout) * ESZ;
        dok = 0;
#pragma unroll
        for (int i = 0; i < DITEMS; ++i) {
            const unsigned ok = (unsigned)(h0 + d_kh[i] < H);
            dok |= ok << i;
            rd[i] = *reinterpret_cast<const u32x4*>(db + (ok ? roff + d_rel[i] : (unsigned)cb * ESZ));
        }
    };
    auto store_dy = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DITEMS; ++i) {
            const bool valid = (dok >> i) & 1u;
            unsigned char* dst = Ys + buf * G::YBYTES + d_dst0[i];
            if constexpr (HS16) {
                *reinterpret_cast<u32x4*>(dst) = valid ? rd[i] : (u32x4){0u, 0u, 0u, 0u};
            } else {
                f32x4 v = __builtin_bit_cast(f32x4, rd[i]);
                if (!valid) v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                unsigned h0_, m0_, l0_, h1_, m1_, l1_;
                split_pack<NP == 1>(v.x, v.y, h0_, m0_, l0_);
                split_pack<NP == 1>(v.z, v.w, h1_, m1_, l1_);
                *reinterpret_cast<u32x2*>(dst) = (u32x2){h0_, h1_};
                if (NSPL == 3) {
                    *reinterpret_cast<u32x2*>(dst + 2 * G::YPL) = (u32x2){m0_, m1_};
                    *reinterpret_cast<u32x2*>(dst + 4 * G::YPL) = (u32x2){l0_, l1_};
                }
            }
        }
    };
    // regular staging of chunk c = its CH new rows (h0+1 .. h0+CH) + its dy tile
    auto issue_chunk = [&](int c) {
        int img, h0, w0;
        origin(c, img, h0, w0);
        load_rows(img, h0 + 1, w0);
        load_dy(img, h0, w0);
    };
    auto store_chunk = [&](int c) {
        int img, h0, w0;
        origin(c, img, h0, w0);
        store_rows(h0 + 1, h0 + CH);
        store_dy((c - cbeg) & 1);
    };
    // first chunk of a strip: rows h0-1 .. h0 are not in the ring yet (synchronous; once per strip)
    auto prime = [&](int c) {
        int img, h0, w0;
        origin(c, img, h0, w0);
        constexpr int NPRIME = (2 + CH - 1) / CH;
#pragma unroll
        for (int k = NPRIME; k >= 1; --k) {
            load_rows(img, h0 + 1 - k * CH, w0);
            store_rows(h0 + 1 - k * CH, h0);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // per-lane fragment bases (bytes): 16-lane group = (half, kl); lane li: pixel row (li>>2), channel quad (li&3)
    const unsigned lq = (unsigned)(half * 32 + (li & 3) * 8);
    const unsigned va = (unsigned)(wi * G::XPL) + lq + (unsigned)((CW == 8 ? (li >> 2) : (kl * 8 + (li >> 2))) * 64);
    const unsigned vb = (unsigned)(wj * G::YPL) + lq + (unsigned)((kl * 8 + (li >> 2)) * 64);

    // MFMA phase of one chunk as 18 plane-steps q = ((s*3 + ky)*3 + o): the A fragments (3 taps kx) of ONE split plane
    // (o = 0 lo, 1 mid, 2 hi) are live at a time and feed every product that uses that plane (lo: a_l*b_h; mid:
    // a_m*b_m, a_m*b_h; hi: a_h*b_l, a_h*b_m, a_h*b_h; NP = 9 adds a_l*b_l, a_l*b_m, a_m*b_l); the fragments of
    // plane-step q+1 are read while the MFMAs of q run (pinned with sched_group_barrier).
    auto mma_chunk = [&](int c) {
        int img, h0, w0;
        origin(c, img, h0, w0);
        const int slot0 = (h0 - 1 + 4 * R) % R;
        unsigned rowb[CH + 2];
#pragma unroll
        for (int j = 0; j < CH + 2; ++j) {
            int sl = slot0 + j;
            sl = sl >= R ? sl - R : sl;
            rowb[j] = (unsigned)(sl * G::XROWB);
        }
        const unsigned char* yb = Ys + ((c - cbeg) & 1) * G::YBYTES + vb;
        constexpr int NQ = (NSPL == 1 ? 3 : 9) * KS;
        auto plane_of = [](int q) { return NSPL == 1 ? 0 : 2 - (q % 3); };          // lo (2), mid (1), hi (0)
        auto load_a = [&](int q, u32x4 (&af)[3]) {
            const int g = NSPL == 1 ? q : q / 3, s = g / 3, ky = g % 3, sp = plane_of(q);
            unsigned arow;                                        // k16 step s = pixels 16 s .. 16 s + 15 of the chunk
            if (CW == 8) arow = kl ? rowb[2 * s + 1 + ky] : rowb[2 * s + ky];
            else if (CW == 16) arow = rowb[s + ky];
            else arow = rowb[(s >> 1) + ky];
            const unsigned char* xa = Xs + va + arow;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int imm = sp * 2 * G::XPL + (kx + (CW == 32 ? (s & 1) * 16 : 0)) * 64;
                const u32x2 t0 = lds_tr_read(xa + imm);
                const u32x2 t1 = lds_tr_read(xa + imm + 256);
                af[kx] = (u32x4){t0.x, t0.y, t1.x, t1.y};
            }
        };
        // dY fragments of k16 step s: two live sets (this step's and the next one's, read while this step's MFMAs run)
        u32x4 bf[2][NSPL];
        auto load_b = [&](int s, u32x4 (&b)[NSPL]) {
#pragma unroll
            for (int sp = 0; sp < NSPL; ++sp) {
                const u32x2 t0 = lds_tr_read(yb + sp * 2 * G::YPL + s * 1024);
                const u32x2 t1 = lds_tr_read(yb + sp * 2 * G::YPL + s * 1024 + 256);
                b[sp] = (u32x4){t0.x, t0.y, t1.x, t1.y};
            }
        };
        load_b(0, bf[0]);
        u32x4 afb[2][3];
        load_a(0, afb[0]);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int QPS = NQ / KS;                              // plane-steps per k16 step
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int g = NSPL == 1 ? q : q / 3, s = g / 3, ky = g % 3, sp = plane_of(q);
            if (q % QPS == 0 && s + 1 < KS) load_b(s + 1, bf[(s + 1) & 1]);
            if (q + 1 < NQ) load_a(q + 1, afb[(q + 1) & 1]);
            // b planes paired with this a plane, smallest product first
            int nb = 0, bl[3] = {0, 0, 0};
            if (NP == 1) { nb = 1; bl[0] = 0; }
            else if (NP == 6) {
                if (sp == 2) { nb = 1; bl[0] = 0; }
                else if (sp == 1) { nb = 2; bl[0] = 1; bl[1] = 0; }
                else { nb = 3; bl[0] = 2; bl[1] = 1; bl[2] = 0; }
            } else { nb = 3; bl[0] = 2; bl[1] = 1; bl[2] = 0; }
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (b >= nb) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc[ky * 3 + kx] = mfma_bf16(afb[q & 1][kx], bf[s & 1][bl[b]], acc[ky * 3 + kx]);
            }
            // pin: the 6 transposing reads of the next plane-step are spread over this step's MFMAs
            const int nm = 3 * nb;
            if (TAG_WX3_PIN && q + 1 < NQ) {
                if (nm >= 6) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                }
            }
            if (TAG_WX3_PIN != 2) __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto fresh = [&](int c) { int i_, h_, w_; origin(c, i_, h_, w_); return h_ == 0; };   // first chunk of a strip
#ifdef TAG_X3_PROF   // 0 prologue, 1 store next chunk + issue loads, 2 MFMA chunk, 3 barrier, 4 new-strip rebuild, 5 epilogue
    unsigned long long xpc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xp0 = __builtin_amdgcn_s_memtime();
#endif
    if (cbeg < cend) {
        window_set(cbeg);
        prime(cbeg);
        issue_chunk(cbeg);
        store_chunk(cbeg);
        __syncthreads();
        if (cbeg + 1 < cend && !fresh(cbeg + 1)) issue_chunk(cbeg + 1);
    }
    XP_MARK(0)
    for (int c = cbeg; c < cend; ++c) {
        const bool next = c + 1 < cend;
        const bool next_fresh = next && fresh(c + 1);
        if (next && !next_fresh) {
#ifdef TAG_X3_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // profile build only: separates the wait for the loads from the stores
            XP_MARK(6)
#endif
            store_chunk(c + 1);                                   // into ring slots / dy buffer chunk c does not read
            XP_MARK(7)
            if (c + 2 < cend && !fresh(c + 2)) issue_chunk(c + 2);
        }
        XP_MARK(1)
        mma_chunk(c);
        XP_MARK(2)
        __syncthreads();
        XP_MARK(3)
        if (next_fresh) {                                         // new strip: rebuild the ring (rare)
            prime(c + 1);
            issue_chunk(c + 1);
            store_chunk(c + 1);
            __syncthreads();
            if (c + 2 < cend && !fresh(c + 2)) issue_chunk(c + 2);
        }
        window_advance();
        XP_MARK(4)
    }
    // partial[split][tap][ci][co]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* out = partial + ((size_t)split * 9 + t) * Cin * Cout;
        const int co = co0 + wj * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            out[(size_t)ci * Cout + co] = acc[t][r];
        }
    }
#ifdef TAG_X3_PROF
    XP_MARK(5)
    if (blockIdx.x == 300 && tid == 0) { for (int i = 0; i < 8; ++i) tag_x3_prof[i] = xpc[i]; tag_x3_prof[8] = (unsigned long long)(cend - cbeg); }
#endif
}

// products per fp32 multiply: 6 (default), 9 (every partial product), 1 (plain bf16, hi plane rounded to nearest);
// 0 = the process default (option x3_products, else 6)
static int x3_products(int requested) {
    if (requested == 1 || requested == 6 || requested == 9) return requested;
    static int v = -1;
    if (v < 0) {
        v = tag_option("x3_products");
        if (v != 1 && v != 6 && v != 9) v = 6;
    }
    return v;
}

template <int MB, int TW, int NP, class TS = float, int WM = (MB == 4 ? 1 : 2)>
void launch_x3(const TS* x, const u32x4* wp, int pro, const float* s, const float* t, TS* y, float* stats, int B,
               int H, int W, int Cin, int Cout, hipStream_t st, const BnBwdEpiX* epi = nullptr) {
    using G = X3Geom<TW, WM * MB * 32>;
    constexpr int BN_ = (4 / WM) * 32;
    const int grid = B * ((H + G::TH - 1) / G::TH) * (Cout / BN_);
    size_t lds = (NP == 1 ? 1 : 3) * G::PLANE + 2 * 512 * 4;
    if (Act<TS>::is_bf16 && TAG_X3_LDS_EPI) {                     // the staged output tile reuses the patch buffer
        const size_t ot = (size_t)(WM * MB * 32) * (BN_ * 2 + 16);
        if (ot > lds) lds = ot;
    }
#define LAUNCH_PRO(P)                                                                                               \
    {                                                                                                               \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<MB, P, TW, NP, TS, 0, WM>),         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((conv3x3_x3_kernel<MB, P, TW, NP, TS, 0, WM>), dim3(grid), dim3(256), lds, st, x, wp, s, t, y, stats, \
                           BnBwdEpiX{}, B, H, W, Cin, Cout);                                                        \
    }
    if constexpr (NP == 1 && Act<TS>::is_bf16 && TAG_X3_LDS_EPI) {
        if (epi && epi->ph > 0) {                                 // dgrad + the pool-backward sums of the block below (EPI == 2)
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<MB, 0, TW, NP, TS, 2, WM>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_set = true;
            }
            hipLaunchKernelGGL((conv3x3_x3_kernel<MB, 0, TW, NP, TS, 2, WM>), dim3(grid), dim3(256), lds, st, x, wp, s, t, y, stats,
                               *epi, B, H, W, Cin, Cout);
            return;
        }
    }
    if constexpr (NP == 1 && Act<TS>::is_bf16) {
        if (epi) {                                                // dgrad + BatchNorm-backward sums (prologue 0 only)
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<MB, 0, TW, NP, TS, 1, WM>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_set = true;
            }
            hipLaunchKernelGGL((conv3x3_x3_kernel<MB, 0, TW, NP, TS, 1, WM>), dim3(grid), dim3(256), lds, st, x, wp, s, t, y, stats,
                               *epi, B, H, W, Cin, Cout);
            return;
        }
    }
    switch (pro) {
        case 0: LAUNCH_PRO(0) break;
        case 1: LAUNCH_PRO(1) break;
        case 2: LAUNCH_PRO(2) break;
        default: LAUNCH_PRO(3) break;
    }
#undef LAUNCH_PRO
}

template <int MB, int NP, class TS = float, int WM = (MB == 4 ? 1 : 2)>
void launch_x3_w(const TS* x, const u32x4* wp, int pro, const float* s, const float* t, TS* y, float* stats, int B,
                 int H, int W, int Cin, int Cout, hipStream_t st, const BnBwdEpiX* epi = nullptr) {
    if (W == 8) launch_x3<MB, 8, NP, TS, WM>(x, wp, pro, s, t, y, stats, B, H, W, Cin, Cout, st, epi);
    else if (W == 16) launch_x3<MB, 16, NP, TS, WM>(x, wp, pro, s, t, y, stats, B, H, W, Cin, Cout, st, epi);
    else if (W == 32) launch_x3<MB, 32, NP, TS, WM>(x, wp, pro, s, t, y, stats, B, H, W, Cin, Cout, st, epi);
    else launch_x3<MB, 64, NP, TS, WM>(x, wp, pro, s, t, y, stats, B, H, W, Cin, Cout, st, epi);
}

template <int TW, int NP, class TS = float>
void launch_wgrad_x3(const TS* x, int pro, const float* s, const float* t, const TS* dy, float* partial, int B,
                     int H, int W, int Cin, int Cout, int splits, int cps, hipStream_t st) {
    using G = WX3Geom<TW, Act<TS>::is_bf16 ? TAG_WX3_KS16 : 2, NP == 1 ? 1 : 3>;
    const int grid = (Cin / 64) * (Cout / 64) * splits;
    const size_t lds = G::LDS_BYTES;
#define LAUNCH_PRO(P)                                                                                                \
    {                                                                                                                \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_x3_kernel<TW, P, NP, TS>),        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((conv3x3_wgrad_x3_kernel<TW, P, NP, TS>), dim3(grid), dim3(256), lds, st, x, s, t, dy, partial, \
                           B, H, W, Cin, Cout, splits, cps);                                                         \
    }
    switch (pro) {
        case 0: LAUNCH_PRO(0) break;
        case 1: LAUNCH_PRO(1) break;
        case 2: LAUNCH_PRO(2) break;
        default: LAUNCH_PRO(3) break;
    }
#undef LAUNCH_PRO
}

template <int NP, class TS = float>
void launch_wgrad_x3_w(const TS* x, int pro, const float* s, const float* t, const TS* dy, float* partial, int B,
                       int H, int W, int Cin, int Cout, int splits, int cps, hipStream_t st) {
    if (W == 8) launch_wgrad_x3<8, NP, TS>(x, pro, s, t, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
    else if (W == 16) launch_wgrad_x3<16, NP, TS>(x, pro, s, t, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
    else if (W == 32) launch_wgrad_x3<32, NP, TS>(x, pro, s, t, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
    else launch_wgrad_x3<64, NP, TS>(x, pro, s, t, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
}

}  // namespace

// conv_rows.hip: the row-streaming kernel that takes the bf16-storage launches of the layers with Cin <= 128
bool tag_conv_rows_takes(int H, int W, int Cin, int Cout, int prologue);
int tag_conv_rows_partial_rows(int B, int H, int W, int Cin, int Cout);
int tag_conv_rows_launch(const bf16_t* x, const void* wpack, int prologue, const float* in_scale, const float* in_shift,
                         bf16_t* y, float* stats, int epi_kind, const bf16_t* yref, const float* bn_scale, const float* bn_shift,
                         const float* bn_mean, const float* bn_invstd, int B, int H, int W, int Cin, int Cout, hipStream_t st);

// conv_wgrad_dma.hip: the bf16-storage weight gradient with both operands brought in by LDS-DMA
bool tag_wgrad_dma_takes(int prologue);
int tag_wgrad_dma_launch(const bf16_t* x, int prologue, const float* in_scale, const float* in_shift, const bf16_t* dy,
                         float* partial, int B, int H, int W, int Cin, int Cout, int splits, int cps, hipStream_t st);

extern "C" size_t tag_conv3x3_x3_pack_bytes(int Cin, int Cout) { return (size_t)9 * Cin * Cout * 3 * 2; }

extern "C" int tag_pack_conv_weight_x3(const float* w, void* wfwd, void* wdgrad, int Cin, int Cout, int products,
                                       void* stream) {
    TAG_CHECK_ARG(w && wfwd && wdgrad && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0);
    const long n = (long)2 * 9 * (Cin / 16) * (Cout / 32) * 64;
    hipLaunchKernelGGL(pack_weight_x3_kernel, dim3(cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256)), dim3(256), 0,
                       as_stream(stream), w, reinterpret_cast<u32x4*>(wfwd), reinterpret_cast<u32x4*>(wdgrad), Cin, Cout,
                       x3_products(products) == 1 ? 1 : 0);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_conv3x3_x3_stats_rows(int B, int H, int W, int Cout) {
    if (!(W == 8 || W == 16 || W == 32 || W == 64)) return 0;
    const int th = 128 / W;
    return B * ((H + th - 1) / th) * (Cout % 128 == 0 ? 1 : 2);
}

// rows of the statistics / BatchNorm-backward partials written by the bf16-STORAGE launch of this shape: one per (strip, wave
// M-group) where the row-streaming kernel of conv_rows.hip takes the layer, one per tile M-group of the tile kernel otherwise
extern "C" int tag_conv3x3_x3_bf16_stats_rows(int B, int H, int W, int Cin, int Cout, int prologue) {
    if (!(W == 8 || W == 16 || W == 32 || W == 64)) return 0;
    if (tag_conv_rows_takes(H, W, Cin, Cout, prologue)) return tag_conv_rows_partial_rows(B, H, W, Cin, Cout);
    const int th = (Cout % 128 == 0 ? 128 : 64 * TAG_X3_BF16_MB64) / W;
    return B * ((H + th - 1) / th) * (Cout % 128 == 0 ? 1 : 2);
}

extern "C" int tag_conv3x3_forward_x3(const float* x, const void* wpack, int prologue, const float* in_scale,
                                      const float* in_shift, float* y, float* stats, int B, int H, int W, int Cin,
                                      int Cout, int products, void* stream) {
    TAG_CHECK_ARG(x && wpack && y && B > 0 && H > 0);
    TAG_CHECK_ARG(W == 8 || W == 16 || W == 32 || W == 64);
    TAG_CHECK_ARG(Cin % 32 == 0 && Cout % 64 == 0 && Cin <= 512);
    TAG_CHECK_ARG((long)B * H * W < (1L << 31) && (long)H * W * Cin * 4 < (1L << 32));      // 32-bit byte offsets inside one image
    TAG_CHECK_ARG(prologue >= 0 && prologue <= 3);
    TAG_CHECK_ARG(prologue == 0 || (in_scale && in_shift));
    hipStream_t st = as_stream(stream);
    const u32x4* wp = reinterpret_cast<const u32x4*>(wpack);
    const int np = x3_products(products);
#define BY_NP(MB)                                                                                        \
    if (np == 6) launch_x3_w<MB, 6>(x, wp, prologue, in_scale, in_shift, y, stats, B, H, W, Cin, Cout, st);     \
    else if (np == 9) launch_x3_w<MB, 9>(x, wp, prologue, in_scale, in_shift, y, stats, B, H, W, Cin, Cout, st); \
    else launch_x3_w<MB, 1>(x, wp, prologue, in_scale, in_shift, y, stats, B, H, W, Cin, Cout, st);
    if (Cout % 128 == 0) { BY_NP(4) } else { BY_NP(2) }
#undef BY_NP
    TAG_LAUNCH_CHECK();
    return 0;
}

// bf16 activation storage (BASELINE configs[2]): x and y are bf16 tensors, weights the one-product (rounded) pack
extern "C" int tag_conv3x3_forward_x3_bf16(const void* x, const void* wpack, int prologue, const float* in_scale,
                                           const float* in_shift, void* y, float* stats, int B, int H, int W, int Cin,
                                           int Cout, void* stream) {
    TAG_CHECK_ARG(x && wpack && y && B > 0 && H > 0);
    TAG_CHECK_ARG(W == 8 || W == 16 || W == 32 || W == 64);
    TAG_CHECK_ARG(Cin % 32 == 0 && Cout % 64 == 0 && Cin <= 512);
    TAG_CHECK_ARG((long)B * H * W < (1L << 31) && (long)H * W * Cin * 2 < (1L << 32));
    TAG_CHECK_ARG(prologue >= 0 && prologue <= 3);
    TAG_CHECK_ARG(prologue == 0 || (in_scale && in_shift));
    hipStream_t st = as_stream(stream);
    const u32x4* wp = reinterpret_cast<const u32x4*>(wpack);
    const bf16_t* xi = static_cast<const bf16_t*>(x);
    bf16_t* yo = static_cast<bf16_t*>(y);
    if (tag_conv_rows_takes(H, W, Cin, Cout, prologue)) {
        const int rc = tag_conv_rows_launch(xi, wpack, prologue, in_scale, in_shift, yo, stats, stats ? 1 : 0, nullptr, nullptr,
                                            nullptr, nullptr, nullptr, B, H, W, Cin, Cout, st);
        if (rc != 0) return rc;
        TAG_LAUNCH_CHECK();
        return 0;
    }
    if (Cout % 128 == 0) launch_x3_w<4, 1, bf16_t>(xi, wp, prologue, in_scale, in_shift, yo, stats, B, H, W, Cin, Cout, st);
    else launch_x3_w<TAG_X3_BF16_MB64, 1, bf16_t, 2>(xi, wp, prologue, in_scale, in_shift, yo, stats, B, H, W, Cin, Cout, st);
    TAG_LAUNCH_CHECK();
    return 0;
}

// bf16 dgrad + the reduction half of the BatchNorm+ReLU backward its output flows into (the bf16 twin of
// tag_conv3x3_dgrad_bnsums): bnpart rows [P][2][Cout], P = tag_conv3x3_x3_stats_rows; fold with tag_bn_grad_from_partials.
extern "C" int tag_conv3x3_dgrad_bnsums_bf16(const void* dy, const void* wpack, void* da, const void* yref,
                                             const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                             const float* bn_invstd, float* bnpart, int B, int H, int W, int Cin, int Cout,
                                             void* stream) {
    TAG_CHECK_ARG(dy && wpack && da && yref && bn_scale && bn_shift && bn_mean && bn_invstd && bnpart && B > 0 && H > 0);
    TAG_CHECK_ARG(W == 8 || W == 16 || W == 32 || W == 64);
    TAG_CHECK_ARG(Cin % 32 == 0 && Cout % 64 == 0 && Cin <= 512);
    TAG_CHECK_ARG((long)B * H * W < (1L << 31) && (long)H * W * Cin * 2 < (1L << 32));
    hipStream_t st = as_stream(stream);
    const u32x4* wp = reinterpret_cast<const u32x4*>(wpack);
    const bf16_t* xi = static_cast<const bf16_t*>(dy);
    bf16_t* yo = static_cast<bf16_t*>(da);
    if (tag_conv_rows_takes(H, W, Cin, Cout, 0)) {
        const int rc = tag_conv_rows_launch(xi, wpack, 0, nullptr, nullptr, yo, bnpart, 2, static_cast<const bf16_t*>(yref), bn_scale,
                                            bn_shift, bn_mean, bn_invstd, B, H, W, Cin, Cout, st);
        if (rc != 0) return rc;
        TAG_LAUNCH_CHECK();
        return 0;
    }
    const BnBwdEpiX epi{static_cast<const bf16_t*>(yref), bn_scale, bn_shift, bn_mean, bn_invstd, 0, 0, 0, 0.0f, 0.0f, 0.0f, 0ull};
    if (Cout % 128 == 0) launch_x3_w<4, 1, bf16_t>(xi, wp, 0, nullptr, nullptr, yo, bnpart, B, H, W, Cin, Cout, st, &epi);
    else launch_x3_w<TAG_X3_BF16_MB64, 1, bf16_t, 2>(xi, wp, 0, nullptr, nullptr, yo, bnpart, B, H, W, Cin, Cout, st, &epi);
    TAG_LAUNCH_CHECK();
    return 0;
}

// bf16 twin of tag_conv3x3_dgrad_poolsums (conv.hip): partial rows P = tag_conv3x3_dgrad_poolsums_bf16_rows (one per workgroup
// m-tile), 0 = this shape is not served (the row-streaming kernel of conv_rows.hip takes it, or the staged-tile epilogue is off):
// the caller keeps the two-pass pool backward.
extern "C" int tag_conv3x3_dgrad_poolsums_bf16_rows(int B, int H, int W, int Cin, int Cout) {
    if (!(W == 8 || W == 16 || W == 32 || W == 64) || !TAG_X3_LDS_EPI) return 0;
    if (Cin % 32 != 0 || Cout % 64 != 0 || Cin > 512) return 0;
    if (tag_conv_rows_takes(H, W, Cin, Cout, 0)) return 0;
    const int th = (Cout % 128 == 0 ? 128 : 64 * TAG_X3_BF16_MB64) / W;
    return B * ((H + th - 1) / th);
}
extern "C" int tag_conv3x3_dgrad_poolsums_bf16(const void* dy, const void* wpack, void* dx, const void* yref,
                                               const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                               const float* bn_invstd, float* bnpart, int B, int H, int W, int Cin, int Cout,
                                               int Hf, int Wf, int ph, int pw, int pool, float drop_p, uint64_t seed, void* stream) {
    TAG_CHECK_ARG(dy && wpack && dx && yref && bn_scale && bn_shift && bn_mean && bn_invstd && bnpart && B > 0 && H > 0);
    TAG_CHECK_ARG(tag_conv3x3_dgrad_poolsums_bf16_rows(B, H, W, Cin, Cout) > 0);
    TAG_CHECK_ARG((long)B * H * W < (1L << 31) && (long)H * W * Cin * 2 < (1L << 32) && (long)B * Hf * Wf < (1L << 31));
    TAG_CHECK_ARG(pw == 2 && (ph == 1 || ph == 2) && H == Hf / ph && W == Wf / pw);
    TAG_CHECK_ARG((pool == 0 || pool == 2 || pool == 3) && drop_p >= 0.0f && drop_p < 1.0f);
    hipStream_t st = as_stream(stream);
    const u32x4* wp = reinterpret_cast<const u32x4*>(wpack);
    const bf16_t* xi = static_cast<const bf16_t*>(dy);
    bf16_t* yo = static_cast<bf16_t*>(dx);
    const float wavg = pool == 3 ? 0.0f : 1.0f / (float)(ph * pw), wmax = pool == 2 ? 0.0f : 1.0f;
    const BnBwdEpiX epi{static_cast<const bf16_t*>(yref), bn_scale, bn_shift, bn_mean, bn_invstd, Hf, Wf, ph, wavg, wmax, drop_p,
                        (unsigned long long)seed};
    if (Cout % 128 == 0) launch_x3_w<4, 1, bf16_t>(xi, wp, 0, nullptr, nullptr, yo, bnpart, B, H, W, Cin, Cout, st, &epi);
    else launch_x3_w<TAG_X3_BF16_MB64, 1, bf16_t, 2>(xi, wp, 0, nullptr, nullptr, yo, bnpart, B, H, W, Cin, Cout, st, &epi);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t tag_conv3x3_wgrad_x3_ws_bytes(int B, int H, int W, int Cin, int Cout) {
    int cps;
    const int s32 = tag_wgrad_alltaps_splits(B, H, W, Cin, Cout, &cps, 32), s64 = tag_wgrad_alltaps_splits(B, H, W, Cin, Cout, &cps, 64);
    return (size_t)(s32 > s64 ? s32 : s64) * 9 * Cin * Cout * sizeof(float);
}

extern "C" int tag_conv3x3_wgrad_x3(const float* x, int prologue, const float* in_scale, const float* in_shift,
                                    const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int products,
                                    void* ws, void* stream) {
    TAG_CHECK_ARG(x && dy && dw && ws && B > 0 && H > 0);
    TAG_CHECK_ARG(W == 8 || W == 16 || W == 32 || W == 64);
    TAG_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0 && prologue >= 0 && prologue <= 3);
    TAG_CHECK_ARG(prologue == 0 || (in_scale && in_shift));
    const long M = (long)B * H * W;
    TAG_CHECK_ARG(M < (1L << 31) && (long)H * W * Cin * 4 < (1L << 32) && (long)H * W * Cout * 4 < (1L << 32));   // offsets inside one image
    float* partial = static_cast<float*>(ws);
    hipStream_t st = as_stream(stream);
    int cps;
    const int np = x3_products(products);
    const int sp = tag_wgrad_alltaps_splits(B, H, W, Cin, Cout, &cps, 32);
    if (np == 6) launch_wgrad_x3_w<6>(x, prologue, in_scale, in_shift, dy, partial, B, H, W, Cin, Cout, sp, cps, st);
    else if (np == 9) launch_wgrad_x3_w<9>(x, prologue, in_scale, in_shift, dy, partial, B, H, W, Cin, Cout, sp, cps, st);
    else launch_wgrad_x3_w<1>(x, prologue, in_scale, in_shift, dy, partial, B, H, W, Cin, Cout, sp, cps, st);
    TAG_LAUNCH_CHECK();
    return tag_launch_wgrad_reduce(partial, sp, Cin, Cout, dw, st);
}

extern "C" int tag_conv3x3_wgrad_x3_bf16(const void* x, int prologue, const float* in_scale, const float* in_shift,
                                         const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, void* ws,
                                         void* stream) {
    TAG_CHECK_ARG(x && dy && dw && ws && B > 0 && H > 0);
    TAG_CHECK_ARG(W == 8 || W == 16 || W == 32 || W == 64);
    TAG_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0 && prologue >= 0 && prologue <= 3);
    TAG_CHECK_ARG(prologue == 0 || (in_scale && in_shift));
    const long M = (long)B * H * W;
    TAG_CHECK_ARG(M < (1L << 31) && (long)H * W * Cin * 2 < (1L << 32) && (long)H * W * Cout * 2 < (1L << 32));   // 32-bit byte offsets inside one image (bf16 storage)
    float* partial = static_cast<float*>(ws);
    hipStream_t st = as_stream(stream);
    int cps;
    const int sp = tag_wgrad_alltaps_splits(B, H, W, Cin, Cout, &cps, 16 * TAG_WX3_KS16);
    if (TAG_WX3_KS16 == 4 && tag_wgrad_dma_takes(prologue)) {
        const int rc = tag_wgrad_dma_launch(static_cast<const bf16_t*>(x), prologue, in_scale, in_shift, static_cast<const bf16_t*>(dy),
                                            partial, B, H, W, Cin, Cout, sp, cps, st);
        if (rc != 0) return rc;
        TAG_LAUNCH_CHECK();
        return tag_launch_wgrad_reduce(partial, sp, Cin, Cout, dw, st);
    }
    launch_wgrad_x3_w<1, bf16_t>(static_cast<const bf16_t*>(x), prologue, in_scale, in_shift, static_cast<const bf16_t*>(dy),
                                 partial, B, H, W, Cin, Cout, sp, cps, st);
    TAG_LAUNCH_CHECK();
    return tag_launch_wgrad_reduce(partial, sp, Cin, Cout, dw, st);
}
