// A1, bf16 mode (BASELINE configs[2]), layers with Cin <= 128: the 3x3 / stride 1 / pad 1 convolution of models/panns.py:25-33,49-50
// (forward and dgrad) as a ROW-STREAMING kernel with the weights stationary in registers.
//
// Why another kernel: the tile kernel of conv_x3.hip pays a full HBM round trip, a weight stream from L2 and a halo re-read
// per 128-pixel tile; on the wide, shallow layers of conv_block1/2 (64 -> 64 at 1001 x 64, 128 -> 128 at 500 x 32 ...) a tile
// is 36-144 MFMAs of work behind that fixed cost and the kernels ran at 0.23-0.39 of the bf16 MFMA peak, 2.6x their HBM
// time.  These layers are HBM-bound (64 -> 64: 256 B of activations per pixel against 74 kFLOP), so the structure here is a
// stream: a workgroup walks DOWN a strip of rows of one image,
//   * the 9 * Cin x 32-cout weight slice of every wave lives in REGISTERS for the whole strip (Cin = 64: 36 fragments = 144
//     VGPRs; Cin = 128: 72 fragments = 288 VGPRs, one wave per SIMD) -- no weight traffic at all inside the loop;
//   * input rows arrive by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction) into a RING of row slots, PF steps
//     ahead of their use: every input row is read from HBM once per strip (no halo re-read between steps; 2 rows per
//     strip), nothing passes through registers, and the loads stay in flight across the step barrier (counted vmcnt);
//   * the MFMA phase of a step reads A fragments with ds_read_b128 from the ring: the LDS image of a row is pixel-major
//     (Cin * 2 bytes per pixel) with the 16-byte channel octets of pixel p XOR-swizzled by a function of p -- applied on the
//     SOURCE address of the DMA (whose LDS side is lane-linear) and on the read address -- so that the 16 pixels of a
//     ds_read_b128 lane group fall on 16 different bank groups for every tap shift;
//   * zero padding costs nothing: one pixel slot of zeros sits between consecutive row slots (left / right halo) and rows
//     outside the image read a permanently zero slot (a wave-uniform base select);
//   * the producer BatchNorm + ReLU (prologue 1) is applied IN PLACE in LDS by the wave that loaded the piece, one step
//     before its first use, beside the other waves' MFMAs;
//   * the output tile leaves through a wave-private LDS staging tile as 16-byte stores;
//   * the BatchNorm statistics (forward) / BatchNorm-backward sums (dgrad) accumulate in registers over the WHOLE strip:
//     one partial row per (workgroup, wave M-group) instead of one per 64-pixel tile (49 MB of partials per launch on
//     conv_block1 before).
// Every vector-memory instruction inside the loop (DMA, output stores) is inline asm and counted by hand: hipcc would
// drain the DMA queue (vmcnt(0)) at every barrier and at every use of an ordinary load otherwise.  gfx9 returns loads and
// stores of one wave in issue order (one counter), so `s_waitcnt vmcnt(N)` with N = the operations issued after the row
// group of interest is exact; every step issues the same number of operations (rows past the strip are clamped, not skipped).
#include <stdlib.h>
#include "tag_common.h"

#ifndef TAG_ROWS_PF
#define TAG_ROWS_PF 2          // row groups in flight ahead of the step that lands next
#endif
#ifndef TAG_ROWS_NACC
#define TAG_ROWS_NACC 2        // accumulator sets the k loop alternates between (dependent-MFMA latency)
#endif
#ifndef TAG_ROWS_NA
#define TAG_ROWS_NA 6          // A fragments in flight: the fragment of product f + NA - 1 is read while product f multiplies
#endif

// -DTAG_ROWS_PROF (tools/run_rows_prof.sh, never in the product build): s_memtime deltas of the phases of ONE workgroup's wave 0
#ifdef TAG_ROWS_PROF
__device__ unsigned long long tag_rows_prof[12];
extern "C" int tag_debug_get_rows_prof(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(tag_rows_prof), 96) == hipSuccess ? 0 : -1; }
#define RP_MARK(i) { const unsigned long long p1_ = __builtin_amdgcn_s_memtime(); rpc[i] += p1_ - rp0; rp0 = p1_; }
#else
#define RP_MARK(i)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MFMA row i (0..31) -> pixel of the 32-pixel block such that the ds_read_b128 lane groups {0-3,12-15,20-27} and
// {4-11,16-19,28-31} each read 16 CONSECUTIVE pixels (as conv_x3.hip)
__device__ __forceinline__ int row_to_pix(int i) {
    return (int)((0xED6360u >> (3 * (i >> 2))) & 7u) * 4 + (i & 3);
}

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1024).  M0 = destination (wave-uniform).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gstore16(void* gdst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(gdst), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void lds_fence_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct RowsEpi {            // EPI == 2: the tensor whose BatchNorm + ReLU the gradient flows into next
    const bf16_t* yref;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
};

// EPI == 2 keeps a ring of yref tiles in LDS as well: its prefetch depth is one step shorter and the output staging tile
// lives inside the yref tile the wave has just consumed, so that two workgroups still fit a CU (75 KB each)
template <int TW, int CIN, int WN, int EPI>
struct RowsGeom {
    static constexpr int WM = 4 / WN;                       // waves along M
    static constexpr int SP = WM * 32;                      // output pixels per step
    static_assert(SP % TW == 0, "a step is a whole number of rows");
    static constexpr int RS = SP / TW;                      // rows per step
    static constexpr int PF = EPI == 2 ? TAG_ROWS_PF - 1 : TAG_ROWS_PF;
    static_assert(PF >= 1, "at least one row group in flight");
    static constexpr int D = (PF + 2) * RS + 2;             // ring slots (+ slot D = zeros)
    static constexpr int PIXB = CIN * 2;                    // bytes per pixel
    static constexpr int RSB = (TW + 1) * PIXB;             // slot stride: one zero pixel between rows
    static constexpr int RING = (D + 1) * RSB + PIXB;
    static constexpr int PR = TW * PIXB / 1024;             // DMA pieces per row
    static constexpr int KX = RS * PR / 4;                  // pieces per wave per step
    static_assert(RS * PR % 4 == 0 && KX >= 1, "pieces divide over the 4 waves");
    static constexpr int PPP = 1024 / PIXB, LPP = PIXB / 16; // pixels per piece, lanes per pixel
    static constexpr int YPIX = WN * 64;                    // bytes per pixel of the yref tile (the workgroup's couts)
    static constexpr int YB = SP * YPIX, KY = YB / 1024 / 4, DY = PF + 2;
    static constexpr int STGP = EPI == 2 ? YPIX : 80;       // pixel stride of the wave's output staging tile
    static constexpr int STG = EPI == 2 ? 0 : 32 * 80;      // wave-private output staging (EPI 2: inside the yref tile)
    static constexpr int OFF_STG = (RING + 15) & ~15;
    static constexpr int OFF_SS = OFF_STG + 4 * STG;        // [2][CIN] floats
    static constexpr int OFF_Y = OFF_SS + 2 * CIN * 4;
    static constexpr int KS = CIN / 16, NF = 9 * KS;
    static constexpr int LDS_BYTES = OFF_Y + (EPI == 2 ? DY * YB : 0);
    static __device__ __forceinline__ int swz(int pslot) { return CIN == 64 ? ((pslot >> 1) & 7) : (pslot & 15); }
};

// EPI: 0 none, 1 BatchNorm statistics of y (rows [P][3][Cout] + counts [P]), 2 BatchNorm-backward sums (rows [P][2][Cout]).
// P = B * NS * WM (one row per workgroup M-group; the n-tiles of a strip write disjoint channel columns of the same row).
template <int TW, int CIN, int WN, int PRO, int EPI>
__global__ __launch_bounds__(256, (CIN == 64 ? 2 : 1)) void conv3x3_rows_kernel(
    const bf16_t* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, bf16_t* __restrict__ y, float* __restrict__ stats, RowsEpi epi, int B, int H, int Cout,
    int NS) {
    using G = RowsGeom<TW, CIN, WN, EPI>;
    constexpr int WM = G::WM, RS = G::RS, D = G::D, PF = G::PF, KS = G::KS, NF = G::NF, KX = G::KX, KY = G::KY;
    constexpr int NACC = TAG_ROWS_NACC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;           // LDS byte address of the array (LDS pointers are 32-bit offsets)

#ifdef TAG_ROWS_PROF
    const unsigned long long rp_t0 = __builtin_amdgcn_s_memtime(), rp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % WN, wm = wid / WN;
    const int ml = lane & 31, kl = lane >> 5;

    // ---- workgroup -> (image, strip, n-tile); n-tiles of a strip are neighbours (they share the input rows in one L2) ----
    const int NT = Cout / (WN * 32);
    int L = xcd_remap(blockIdx.x, B * NS * NT);
    const int nt = L % NT; L /= NT;
    const int strip = L % NS, img = L / NS;
    const int steps_img = H / RS;
    const int t_beg = (int)((long)steps_img * strip / NS), t_end = (int)((long)steps_img * (strip + 1) / NS);
    const int T = t_end - t_beg;
    const int r0 = t_beg * RS;                              // first output row; input row of ring index u: r0 - 1 + u
    const int n0 = nt * WN * 32;
    if (T <= 0) return;

    // ---- weights: this wave's 32-cout slice of all 9 taps, in registers for the whole strip ----
    u32x4 bq[NF];
    {
        const int NBK = Cout / 32;
        const u32x4* wl = wp + (size_t)(n0 / 32 + wn) * 192 + lane;      // + ((tap*KS + ks) * NBK) * 192
#pragma unroll
        for (int f = 0; f < NF; ++f) bq[f] = wl[(size_t)f * NBK * 192];
    }

    // ---- zero pixels between the row slots, the zero row, the prologue table ----
    for (int i = tid; i < (D + 2) * (G::PIXB / 16); i += 256) {
        const int s = i / (G::PIXB / 16), c = i % (G::PIXB / 16);
        *reinterpret_cast<u32x4*>(smem + s * G::RSB + c * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < G::RSB / 16; i += 256)
        *reinterpret_cast<u32x4*>(smem + D * G::RSB + i * 16) = (u32x4){0u, 0u, 0u, 0u};
    float* Ss = reinterpret_cast<float*>(smem + G::OFF_SS);
    if (PRO != 0)
        for (int c = tid; c < CIN; c += 256) { Ss[c] = in_scale[c]; Ss[CIN + c] = in_shift[c]; }

    // ---- DMA geometry of this lane: piece e -> (row of the group, piece of the row); the lane's pixel and channel octet ----
    const char* ximg = reinterpret_cast<const char*>(x) + (size_t)img * H * TW * G::PIXB;
    const int lpix = lane / G::LPP, lchunk = lane % G::LPP;
    auto src_off = [&](int q) {                             // byte offset inside a row of the 16 B this lane fetches for piece q
        const int p = q * G::PPP + lpix;
        return (unsigned)(p * G::PIXB + ((lchunk ^ G::swz(p + 1)) << 4));
    };
    auto issue_row_piece = [&](int u, int q) {              // ring index u (input row r0 - 1 + u, clamped into the image)
        int ri = r0 - 1 + u;
        ri = ri < 0 ? 0 : (ri >= H ? H - 1 : ri);
        const unsigned dst = lds0 + (unsigned)((u % D) * G::RSB + G::PIXB + q * 1024);
        glds16(ximg + (size_t)ri * TW * G::PIXB + src_off(q), __builtin_amdgcn_readfirstlane(dst));
    };
    auto issue_group = [&](int g) {                         // rows u = g*RS + 2 .. g*RS + RS + 1 (group of step g)
#pragma unroll
        for (int jj = 0; jj < KX; ++jj) {
            const int e = wid + 4 * jj;
            issue_row_piece(g * RS + 2 + e / G::PR, e % G::PR);
        }
    };
    // BatchNorm + ReLU of the producer, in place, on the pieces this wave loaded
    auto transform_piece = [&](int u, int q) {
        unsigned char* pp = smem + (u % D) * G::RSB + G::PIXB + q * 1024 + lane * 16;
        const int p = q * G::PPP + lpix;
        const int c0 = (lchunk ^ G::swz(p + 1)) * 8;
        const u32x4 raw = *reinterpret_cast<const u32x4*>(pp);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(Ss + c0), s1 = *reinterpret_cast<const f32x4*>(Ss + c0 + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(Ss + CIN + c0), t1 = *reinterpret_cast<const f32x4*>(Ss + CIN + c0 + 4);
        u32x4 o;
        o.x = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.x), s0.x, t0.x), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.x), s0.y, t0.y), 0.0f));
        o.y = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.y), s0.z, t0.z), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.y), s0.w, t0.w), 0.0f));
        o.z = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.z), s1.x, t1.x), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.z), s1.y, t1.y), 0.0f));
        o.w = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.w), s1.z, t1.z), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.w), s1.w, t1.w), 0.0f));
        *reinterpret_cast<u32x4*>(pp) = o;
    };
    auto transform_group = [&](int g) {
#pragma unroll
        for (int jj = 0; jj < KX; ++jj) {
            const int e = wid + 4 * jj;
            transform_piece(g * RS + 2 + e / G::PR, e % G::PR);
        }
    };
    // yref tile of step g (EPI == 2): SP pixels x YPIX bytes, linear
    const char* yimg = reinterpret_cast<const char*>(epi.yref) + (size_t)img * H * TW * Cout * 2;
    auto issue_yref = [&](int g) {
        if constexpr (EPI == 2) {
            constexpr int YL = G::YPIX / 16;                // lanes per pixel
#pragma unroll
            for (int jj = 0; jj < KY; ++jj) {
                const int e = wid + 4 * jj;                 // piece of the tile: pixels e * (64 / YL) ...
                const int sp = e * (64 / YL) + lane / YL;
                int row = r0 + g * RS + sp / TW;
                row = row >= H ? H - 1 : row;
                const unsigned dst = lds0 + (unsigned)(G::OFF_Y + (g % G::DY) * G::YB + e * 1024);
                glds16(yimg + ((size_t)row * TW + sp % TW) * Cout * 2 + n0 * 2 + (lane % YL) * 16,
                       __builtin_amdgcn_readfirstlane(dst));
            }
        }
    };

    // ---- A-fragment addresses of this lane: pixel slot p' = column + kx (slot 0 = the zero pixel left of the row) ----
    const int sp_lane = wm * 32 + row_to_pix(ml);           // pixel of the step this lane's MFMA row is
    const int rin = sp_lane / TW, col = sp_lane % TW;        // row inside the step (wave-uniform unless TW == 16), column
    unsigned ta[3][KS];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ps = col + kx;
            ta[kx][ks] = (unsigned)(ps * G::PIXB + (((2 * ks + kl) ^ G::swz(ps)) << 4));
        }

    // ---- epilogue constants ----
    const int n = n0 + wn * 32 + ml;
    unsigned char* stg0 = smem + G::OFF_STG + wid * G::STG;                   // EPI != 2
    float e_sc = 0.0f, e_sh = 0.0f, e_mu = 0.0f, e_is = 0.0f;
    if (EPI == 2) { e_sc = epi.scale[n]; e_sh = epi.shift[n]; e_mu = epi.mean[n]; e_is = epi.invstd[n]; }
    float st_mu = 0.0f, st_a = 0.0f, st_b = 0.0f;           // EPI 1: pivot, sum(y - pivot), sum (y - pivot)^2; EPI 2: sum g, sum g*xhat

    // ---- prime the ring: every row up to the group of step PF, the yref tiles of steps 0..PF ----
    {
        constexpr int U0 = (PF + 1) * RS + 2;
        for (int e = wid; e < U0 * G::PR; e += 4) issue_row_piece(e / G::PR, e % G::PR);
        for (int g = 0; g <= PF; ++g) issue_yref(g);
        wait_vmcnt<0>();
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(bq[f]));      // hipcc's own wait for the weight loads goes HERE
        lds_fence_barrier();                                // zero fill, table and every primed row are visible
        if (PRO != 0) {
            for (int e = wid; e < (RS + 2) * G::PR; e += 4) transform_piece(e / G::PR, e % G::PR);   // rows of step 0
        }
    }

    constexpr int MST = 2;                                  // output stores per wave per step
    constexpr int KOPS = KX + (EPI == 2 ? KY : 0);          // DMA instructions per wave per step
    constexpr int NWAIT = MST + PF * (KOPS + MST);          // operations issued after the group that must have landed

#ifdef TAG_ROWS_PROF   // 0 prologue, 1 barrier, 2 DMA issue, 3 MFMA phase, 4 pack + staging + statistics, 5 output stores, 6 wait for the DMA, 7 transform
    unsigned long long rpc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rp0 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < T; ++t) {
        RP_MARK(t == 0 ? 0 : 7)
        lds_fence_barrier();                                // rows of step t transformed and visible; step t-1's rows are free
        RP_MARK(1)
        issue_group(t + PF + 1);
        issue_yref(t + PF + 1);
        RP_MARK(2)

        // ---- row bases of this step: ring indices u0 .. u0 + RS + 1 ----
        const int u0 = t * RS;
        unsigned rb[RS + 2];
#pragma unroll
        for (int j = 0; j < RS + 2; ++j) {
            const int ri = r0 - 1 + u0 + j;
            rb[j] = (unsigned)(((unsigned)ri < (unsigned)H ? (u0 + j) % D : D) * G::RSB);
        }
        f32x16 acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
        // 9 * KS MFMAs; the A fragment of product i + 2 is read while product i multiplies
        auto a_addr = [&](int f) {
            const int tap = f / KS, ks = f % KS, ky = tap / 3, kx = tap % 3;
            unsigned base;
            if (TW >= 32) base = rb[(wm * 32 / TW) % RS + ky];              // wave-uniform row of the step
            else base = rin ? rb[1 + ky] : rb[ky];                          // TW == 16: the block spans two rows
            return base + ta[kx][ks];
        };
        // LDS latency under 8 waves per CU is several MFMA issue times: NA - 1 reads stay in flight (with 2 the MFMA of every
        // product waited on a read issued one product earlier: 0.32 MFMA busy)
        constexpr int NA = TAG_ROWS_NA;
        u32x4 af[NA];
#pragma unroll
        for (int f = 0; f < NA - 1; ++f) af[f] = *reinterpret_cast<const u32x4*>(smem + a_addr(f));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (f + NA - 1 < NF) af[(f + NA - 1) % NA] = *reinterpret_cast<const u32x4*>(smem + a_addr(f + NA - 1));
            acc[f % NACC] = mfma_bf16(af[f % NA], bq[f], acc[f % NACC]);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // pin: 1 MFMA, 1 address add, 1 LDS read
            if (f + NA - 1 < NF) {
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef TAG_ROWS_PROF
        asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[NACC - 1][0]));     // the last products have retired
#endif
        RP_MARK(3)
        if (NACC == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
        }

        // ---- epilogue of the step: D col = lane & 31 (cout), D row = (r&3) + 8 (r>>2) + 4 kl -> pixel via row_to_pix ----
        // staging tile of the wave: [32 pixels][32 couts] bf16; EPI 2: the wave's own 32 px x 64 B window of the yref tile of
        // this step, overwritten once its values are in registers (no other wave touches that window)
        unsigned char* stg = EPI == 2 ? smem + G::OFF_Y + (t % G::DY) * G::YB + wm * 32 * G::YPIX + wn * 64 : stg0;
        float yv[EPI == 2 ? 16 : 1];
        if (EPI == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pm = row_to_pix((r & 3) + 8 * (r >> 2) + 4 * kl);
                yv[r] = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(stg + pm * G::STGP + ml * 2) << 16);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pm = row_to_pix((r & 3) + 8 * (r >> 2) + 4 * kl);
            *reinterpret_cast<unsigned short*>(stg + pm * G::STGP + ml * 2) = (unsigned short)tag_pack_bf16(acc[0][r], 0.0f);
        }
        if (EPI == 1) {
            if (t == 0) {                                   // pivot = mean of the wave's first tile
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[0][r];
                s += __shfl_xor(s, 32, 64);
                st_mu = s * (1.0f / 32.0f);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[0][r] - st_mu;
                st_a += d;
                st_b = fmaf(d, d, st_b);
            }
        }
        if (EPI == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g = fmaf(yv[r], e_sc, e_sh) > 0.0f ? acc[0][r] : 0.0f;
                st_a += g;
                st_b = fmaf(g, (yv[r] - e_mu) * e_is, st_b);
            }
        }
        RP_MARK(4)
        // the wave's 32 px x 32 co tile = 128 pieces of 16 B: two per lane, 4 lanes per pixel
#pragma unroll
        for (int k = 0; k < MST; ++k) {
            const int piece = lane + 64 * k, pm = piece >> 2, c4 = piece & 3;
            const int sp = wm * 32 + pm;
            const int row = r0 + t * RS + sp / TW;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + pm * G::STGP + c4 * 16);
            gstore16(y + (((size_t)img * H + row) * TW + sp % TW) * Cout + n0 + wn * 32 + c4 * 8, v);
        }
        RP_MARK(5)
        // ---- the group of step t + 1 (issued PF + 1 steps ago): landed -> producer BatchNorm + ReLU in place ----
        wait_vmcnt<NWAIT>();
        RP_MARK(6)
        if (PRO != 0 && t + 1 < T) transform_group(t + 1);
    }
#ifdef TAG_ROWS_PROF
    RP_MARK(7)
    if (blockIdx.x == 77 && tid == 0) {
        for (int i = 0; i < 8; ++i) tag_rows_prof[i] = rpc[i];
        tag_rows_prof[8] = (unsigned long long)T;
        tag_rows_prof[9] = __builtin_amdgcn_s_memtime() - rp_t0;             // whole workgroup, shader clocks
        tag_rows_prof[10] = __builtin_amdgcn_s_memrealtime() - rp_r0;        // the same in 100 MHz ticks
    }
#endif
    wait_vmcnt<0>();                                        // no DMA may still target this workgroup's LDS

    // ---- one partial row per (image, strip, wave M-group); n-tiles and waves write disjoint channels ----
    if (EPI == 1) {
        const int prow = (img * NS + strip) * WM + wm;
        const int P = B * NS * WM;
        st_a += __shfl_xor(st_a, 32, 64);
        st_b += __shfl_xor(st_b, 32, 64);
        float* ps = stats + (size_t)prow * 3 * Cout;
        if (kl == 0) { ps[n] = st_mu; ps[Cout + n] = st_a; ps[2 * Cout + n] = st_b; }
        if (nt == 0 && wn == 0 && lane == 0) stats[(size_t)P * 3 * Cout + prow] = (float)(T * 32);
    }
    if (EPI == 2) {
        const int prow = (img * NS + strip) * WM + wm;
        st_a += __shfl_xor(st_a, 32, 64);
        st_b += __shfl_xor(st_b, 32, 64);
        float* ps = stats + (size_t)prow * 2 * Cout;
        if (kl == 0) { ps[n] = st_a; ps[Cout + n] = st_b; }
    }
}

// strips per image: about two workgroups per CU in one residency round, strips of at least 8 steps
int rows_strips(int B, int H, int RS, int NT) {
    static int cus = 0;
    if (!cus) { cus = tag_device_cu_count(); if (cus <= 0) cus = 256; }
    const int steps = H / RS;
    int ns = (2 * cus + B * NT - 1) / (B * NT);
    if (ns > steps / 8) ns = steps / 8;
    return ns < 1 ? 1 : ns;
}

bool rows_c128() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TAG_CONV_ROWS_C128"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
struct RowsCfg { int TW, CIN, WN; };
// the (width, Cin, Cout) shapes this kernel takes: weights of a 32-cout slice fit the register file (Cin <= 128)
bool rows_cfg(int H, int W, int Cin, int Cout, RowsCfg* c) {
    int wn = 0;
    if (W == 64 && Cin == 64 && Cout == 64) wn = 2;
    else if (W == 32 && Cin == 64 && Cout % 128 == 0) wn = 4;
    // Cin = 128 (288 weight registers: ONE wave per SIMD, every phase of a step serialised) measured 0.66-0.96 of the tile
    // kernel's speed: those layers stay on the tile kernel unless TAG_CONV_ROWS_C128=1
    else if (rows_c128() && W == 32 && Cin == 128 && Cout == 64) wn = 2;
    else if (rows_c128() && W == 32 && Cin == 128 && Cout % 128 == 0) wn = 4;
    else if (rows_c128() && W == 16 && Cin == 128 && Cout % 128 == 0) wn = 4;
    if (!wn) return false;
    const int rs = (4 / wn) * 32 / W;
    if (H % rs != 0 || H / rs < 2) return false;
    if (c) { c->TW = W; c->CIN = Cin; c->WN = wn; }
    return true;
}

template <int TW, int CIN, int WN, int PRO, int EPI>
void launch_rows(const bf16_t* x, const u32x4* wp, const float* s, const float* t, bf16_t* y, float* stats, const RowsEpi& epi,
                 int B, int H, int Cout, hipStream_t st) {
    using G = RowsGeom<TW, CIN, WN, EPI>;
    const int NT = Cout / (WN * 32), NS = rows_strips(B, H, G::RS, NT);
    const int lds = G::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rows_kernel<TW, CIN, WN, PRO, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_rows_kernel<TW, CIN, WN, PRO, EPI>), dim3(B * NS * NT), dim3(256), lds, st, x, wp, s, t, y, stats,
                       epi, B, H, Cout, NS);
}

template <int TW, int CIN, int WN>
void launch_rows_pe(int pro, int epi_kind, const bf16_t* x, const u32x4* wp, const float* s, const float* t, bf16_t* y,
                    float* stats, const RowsEpi& epi, int B, int H, int Cout, hipStream_t st) {
    if (epi_kind == 2) launch_rows<TW, CIN, WN, 0, 2>(x, wp, s, t, y, stats, epi, B, H, Cout, st);
    else if (pro == 0 && epi_kind == 0) launch_rows<TW, CIN, WN, 0, 0>(x, wp, s, t, y, stats, epi, B, H, Cout, st);
    else if (pro == 0) launch_rows<TW, CIN, WN, 0, 1>(x, wp, s, t, y, stats, epi, B, H, Cout, st);
    else if (epi_kind == 0) launch_rows<TW, CIN, WN, 1, 0>(x, wp, s, t, y, stats, epi, B, H, Cout, st);
    else launch_rows<TW, CIN, WN, 1, 1>(x, wp, s, t, y, stats, epi, B, H, Cout, st);
}

}  // namespace

// TAG_CONV_ROWS=0 (environment) or tag_conv_rows_enable(0) keeps every layer on the tile kernel of conv_x3.hip (A/B timing)
static int g_rows_on = -1;
static bool rows_enabled() {
    if (g_rows_on < 0) { const char* e = getenv("TAG_CONV_ROWS"); g_rows_on = (e && e[0] == '0') ? 0 : 1; }
    return g_rows_on == 1;
}
extern "C" int tag_conv_rows_enable(int on) {
    const int was = rows_enabled() ? 1 : 0;
    g_rows_on = on ? 1 : 0;
    return was;
}

// does the bf16-storage launch of this shape go to the row-streaming kernel?  (prologue 0 / 1 only)
bool tag_conv_rows_takes(int H, int W, int Cin, int Cout, int prologue) {
    return rows_enabled() && prologue <= 1 && rows_cfg(H, W, Cin, Cout, nullptr);
}
// partial rows its epilogues write (statistics: [P][3][Cout] + [P]; BatchNorm-backward sums: [P][2][Cout])
int tag_conv_rows_partial_rows(int B, int H, int W, int Cin, int Cout) {
    RowsCfg c;
    if (!rows_cfg(H, W, Cin, Cout, &c)) return 0;
    const int wm = 4 / c.WN, rs = wm * 32 / W;
    return B * rows_strips(B, H, rs, Cout / (c.WN * 32)) * wm;
}

// epi_kind 0 none, 1 statistics, 2 BatchNorm-backward sums (yref... of *epi)
int tag_conv_rows_launch(const bf16_t* x, const void* wpack, int prologue, const float* in_scale, const float* in_shift,
                         bf16_t* y, float* stats, int epi_kind, const bf16_t* yref, const float* bn_scale, const float* bn_shift,
                         const float* bn_mean, const float* bn_invstd, int B, int H, int W, int Cin, int Cout, hipStream_t st) {
    RowsCfg c;
    if (!rows_cfg(H, W, Cin, Cout, &c)) return TAG_EINVAL;
    const u32x4* wp = reinterpret_cast<const u32x4*>(wpack);
    const RowsEpi epi{yref, bn_scale, bn_shift, bn_mean, bn_invstd};
#define ROWS_CASE(TW_, CIN_, WN_)                                                                                          \
    if (c.TW == TW_ && c.CIN == CIN_ && c.WN == WN_) {                                                                     \
        launch_rows_pe<TW_, CIN_, WN_>(prologue, epi_kind, x, wp, in_scale, in_shift, y, stats, epi, B, H, Cout, st);      \
        return 0;                                                                                                          \
    }
    ROWS_CASE(64, 64, 2)
    ROWS_CASE(32, 64, 4)
    ROWS_CASE(32, 128, 2)
    ROWS_CASE(32, 128, 4)
    ROWS_CASE(16, 128, 4)
#undef ROWS_CASE
    return TAG_EINVAL;
}
