// A1, bf16 mode (BASELINE configs[2]), the 64-channel layers: the 3x3 / stride 1 / pad 1 convolution of models/panns.py:25-33,49-50
// (forward and dgrad) as a ROW-STREAMING kernel: weights stationary in registers, input rows by LDS-DMA into a ring, and the
// eight waves of a workgroup split into two groups that alternate between a matrix phase and a memory / epilogue phase.
//
// Why another kernel: the tile kernel of conv_x3.hip pays a full HBM round trip, a weight stream from L2 and a halo re-read
// per 128-pixel tile; on the wide, shallow layers of conv_block1/2 (64 -> 64 at 1001 x 64, 64 -> 128 at 500 x 32) a tile is
// 36-72 MFMAs of work behind that fixed cost and the kernels ran at 0.23-0.34 of the bf16 MFMA peak, 2.6x their HBM time.
// These layers are HBM-bound (64 -> 64: 256 B of activations per pixel against 74 kFLOP), so the structure is a stream: a
// workgroup walks DOWN a strip of rows of one image,
//   * the 9 * 64 x 32-cout weight slice of every wave lives in REGISTERS for the whole strip (36 fragments = 144 VGPRs):
//     no weight traffic at all inside the loop;
//   * input rows arrive by LDS-DMA (global_load_lds_dwordx4) into a RING of row slots, 2 PF phases ahead of their use: every
//     input row is read from HBM once per strip (no halo re-read between rows; 2 rows per strip), nothing passes through
//     registers, and the loads stay in flight across the phase barriers (counted vmcnt);
//   * the LDS image of a row is OCTET-PLANE major: plane o holds the 16-byte channel octet o of every pixel of the row,
//     pixels consecutive, behind 16 bytes of zeros.  A ds_read_b128 lane group (16 lanes = 16 consecutive pixels, one octet)
//     reads 256 contiguous bytes (conflict-free for every tap shift), the address of (tap, k-step) is an IMMEDIATE offset
//     from one lane base per input row (3 address adds per 36 MFMAs), the zeros in front of a plane are the left halo of its
//     row and the right halo of the plane before it, and a DMA instruction = one plane of one row (the lanes gather their
//     pixel's octet from the 128-byte pixel records: same cache lines as a linear copy, 8 passes over them through L1);
//   * rows outside the image read a permanently zero slot (a wave-uniform base select);
//   * the producer BatchNorm + ReLU (prologue 1) is applied IN PLACE in LDS by the wave that loaded the plane (its channel
//     octet is wave-uniform), one phase before the row's first use;
//   * the output tile leaves through a wave-private LDS window as 16-byte stores; for the dgrad with BatchNorm-backward sums
//     that window is the wave's own 32 px x 32 co tile of yref, fetched by its own DMA (no barrier) and overwritten once read;
//   * the BatchNorm statistics (forward) / BatchNorm-backward sums (dgrad) accumulate in registers over the WHOLE strip:
//     one partial row per (strip, group, wave M-block) instead of one per 64-pixel tile (49 MB of partials per launch on
//     conv_block1 before).
// Ping-pong: waves w and w + 4 share a SIMD.  Group 0 (waves 0-3) owns the even rows of the strip, group 1 the odd rows; in
// every phase one group issues the 36 MFMAs of its row while the other runs the tail of its previous row (wait for a DMA,
// prologue transform, pack, statistics, stores, next DMA), one s_barrier per phase.  With two independent 4-wave workgroups
// per CU instead (the first form of this file) both waves of a SIMD drifted into the same phase: the matrix phase took 2.3x
// its issue time, the pipe idled during the tails, and the step was bound by instruction issue (3700-4150 clocks per row
// pair against 2 x 1152 of MFMA issue) at a power-limited 1.3-1.8 GHz.
// Every vector-memory instruction inside the loop (DMA, output stores) is inline asm and counted by hand: hipcc would
// drain the DMA queue (vmcnt(0)) at every barrier and at every use of an ordinary load otherwise.  gfx9 returns the loads and
// stores of one wave in issue order (one counter), so `s_waitcnt vmcnt(N)` with N = the operations issued after the DMA of
// interest is exact; every tail issues the same number of operations (rows past the strip are clamped, not skipped).
#include <stdlib.h>
#include "tag_common.h"

#ifndef TAG_ROWS_PF
#define TAG_ROWS_PF 2          // tails of its group a DMA is issued ahead of the tail that consumes it (>= 2)
#endif
#ifndef TAG_ROWS_DMA_SADDR
#define TAG_ROWS_DMA_SADDR 0   // 1: SGPR-base form of the DMA (glds16s).  Measured: plane DMA 370-374 vs 374 us (nothing), the yref gather
#endif                         // of the dgrad launches 484 vs 405 us (worse): the partner group's real MFMA stream has gaps, the probe's has none
#ifndef TAG_ROWS_NACC
#define TAG_ROWS_NACC 2        // accumulator sets the k loop alternates between (dependent-MFMA latency)
#endif
#ifndef TAG_ROWS_M16
#define TAG_ROWS_M16 0         // 1: v_mfma_f32_16x16x32_bf16 (2 x 2 blocks per wave tile), 0: v_mfma_f32_32x32x16_bf16.  Same FLOP rate, same
#endif                         // operand traffic per FLOP here (an A fragment feeds two products); tools/mfma_peak.hip: the 16x16x32
                               // shape sustains 2.01-2.05 PFLOP/s on random operands against 1.81-1.85 for 32x32x16 (fewer
                               // accumulator reads and writes per FLOP) -- in a power-limited kernel that is clock
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
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MFMA row i (0..31) -> pixel of the 32-pixel block such that the ds_read_b128 lane groups {0-3,12-15,20-27} and
// {4-11,16-19,28-31} each read 16 CONSECUTIVE pixels (as conv_x3.hip)
__device__ __forceinline__ int row_to_pix(int i) {
    return (int)((0xED6360u >> (3 * (i >> 2))) & 7u) * 4 + (i & 3);
}

// LDS-DMA: the active lanes copy 16 B each from their global address to LDS [dst + 16 * lane).  M0 = destination (wave-uniform).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same with a wave-uniform base in an SGPR pair and a 32-bit per-lane offset.  tools/coissue_probe.hip: beside another wave's
// bf16 MFMA stream on the same SIMD the 64-bit-VGPR-address form above issues once per ~2500 clocks, this form once per ~95
// (an idle SIMD: 84-97 either way) -- and the ping-pong groups of this kernel issue their DMA exactly while the partner group
// streams MFMAs.
__device__ __forceinline__ void glds16s(const void* ubase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(ubase), "s"(lds_dst) : "memory");
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

template <int TW, int CIN, int WN, int EPI>
struct RowsGeom {
    static constexpr int WM = 4 / WN;                       // waves along M inside a group
    static_assert(WM * 32 == TW, "a group covers one image row per phase");
    static constexpr int PF = TAG_ROWS_PF;
    static_assert(PF >= 2, "the DMA of a row is issued at least two tails ahead");
    static constexpr int D = 2 * PF + 4;                    // ring slots (+ slot D = zeros)
    static constexpr int PIXB = CIN * 2;                    // bytes per pixel in HBM
    static constexpr int NPL = CIN / 8;                     // octet planes per row
    static constexpr int PLB = (TW + 1) * 16;               // plane of a row: 16 B of zeros + TW pixels x 16 B
    static constexpr int SLOT = NPL * PLB;
    static constexpr int RING = (D + 1) * SLOT + 16;        // + the zeros behind the last plane
    static constexpr int KX = NPL / 4;                      // DMA instructions (planes) per wave per tail
    static_assert(NPL % 4 == 0, "planes divide over the 4 waves of a group");
    static constexpr int WINB = 32 * 64;                    // wave window: [32 pixels][32 couts] bf16 (yref tile / output staging)
    static constexpr int NWIN = EPI == 2 ? PF : 1;
    static constexpr int KY = EPI == 2 ? WINB / 1024 : 0;   // yref DMA instructions per wave per tail
    static constexpr int OFF_WIN = (RING + 15) & ~15;
    static constexpr int OFF_SS = OFF_WIN + 8 * NWIN * WINB;     // [2][CIN] floats
    static constexpr int LDS_BYTES = OFF_SS + 2 * CIN * 4;
    static constexpr int KS = CIN / 16, NF = 9 * KS;
    static constexpr int MST = 2;                           // output stores per wave per tail
    // every tail issues MST stores, then KX row planes, then KY yref pieces; the row planes are consumed PF matrix phases of the
    // wave later, the yref pieces PF tails later: operations issued after them at the respective wait
    static constexpr int NWAIT_X = KY + (PF - 1) * (MST + KX + KY);
    static constexpr int NWAIT_Y = (PF - 1) * (MST + KX + KY);
};

// EPI: 0 none, 1 BatchNorm statistics of y (rows [P][3][Cout] + counts [P]), 2 BatchNorm-backward sums (rows [P][2][Cout]).
// P = B * NS * 2 * WM: one row per (image, strip, group, wave M-block); the n-tiles of a strip write disjoint channel columns.
template <int TW, int CIN, int WN, int PRO, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_rows_kernel(
    const bf16_t* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, bf16_t* __restrict__ y, float* __restrict__ stats, RowsEpi epi, int B, int H, int Cout,
    int NS) {
    using G = RowsGeom<TW, CIN, WN, EPI>;
    constexpr int WM = G::WM, D = G::D, PF = G::PF, KS = G::KS, NF = G::NF, KX = G::KX, NPL = G::NPL;
    constexpr int NACC = TAG_ROWS_NACC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;           // LDS byte address of the array (LDS pointers are 32-bit offsets)
#ifdef TAG_ROWS_PROF
    const unsigned long long rp_t0 = __builtin_amdgcn_s_memtime(), rp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, w4 = wid & 3;                 // waves w and w + 4 share a SIMD: one of each group
    const int wn = w4 % WN, wm = w4 / WN;
    const int ml = lane & 31, kl = lane >> 5;

    // ---- workgroup -> (image, strip, n-tile); n-tiles of a strip are neighbours (they share the input rows in one L2) ----
    const int NT = Cout / (WN * 32);
    int L = xcd_remap(blockIdx.x, B * NS * NT);
    const int nt = L % NT; L /= NT;
    const int strip = L % NS, img = L / NS;
    const int r0 = (int)((long)H * strip / NS), r1 = (int)((long)H * (strip + 1) / NS);
    const int T = r1 - r0;                                  // output rows r0 .. r1 - 1; input row of ring index u: r0 - 1 + u
    const int n0 = nt * WN * 32;
    if (T <= 0) return;

    // ---- weights: this wave's 32-cout slice of all 9 taps, in registers for the whole strip ----
    u32x4 bq[NF];
    {
        const int NBK = Cout / 32;
        const u32x4* wl = wp + (size_t)(n0 / 32 + wn) * 192 + lane;      // + ((tap*KS + ks) * NBK) * 192
#if TAG_ROWS_M16
        // 16x16x32 B fragment (tap, k32, ni): lane (column n15 = lane & 15, k octet kq = lane >> 4) holds k = 32 k32 + 8 kq .. + 7 of cout
        // ni * 16 + n15 = the 16 bytes the 32x32x16 pack keeps at fragment (tap, kk16 = 2 k32 + (kq >> 1)), lane (kq & 1) * 32 + ni * 16 + n15
        const int kq = lane >> 4, n15 = lane & 15;
#pragma unroll
        for (int f = 0; f < NF; ++f) {                          // f = (tap * KS/2 + k32) * 2 + ni
            const int ni = f & 1, k32 = (f >> 1) % (KS / 2), tap = (f >> 1) / (KS / 2);
            bq[f] = wp[((size_t)(tap * KS + 2 * k32 + (kq >> 1)) * NBK + n0 / 32 + wn) * 192 + (kq & 1) * 32 + ni * 16 + n15];
        }
        (void)wl;
#else
#pragma unroll
        for (int f = 0; f < NF; ++f) bq[f] = wl[(size_t)f * NBK * 192];
#endif
    }

    // ---- zeros everywhere (plane heads, the zero slot), the prologue table ----
    for (int i = tid; i < G::RING / 16; i += 512) *reinterpret_cast<u32x4*>(smem + i * 16) = (u32x4){0u, 0u, 0u, 0u};
    float* Ss = reinterpret_cast<float*>(smem + G::OFF_SS);
    if (PRO != 0)
        for (int c = tid; c < CIN; c += 512) { Ss[c] = in_scale[c]; Ss[CIN + c] = in_shift[c]; }
    lds_fence_barrier();                                    // no DMA may land before the zero fill is done

    // ---- DMA of plane o of ring row u (input row r0 - 1 + u, clamped into the image): lane = pixel ----
    const char* ximg = reinterpret_cast<const char*>(x) + (size_t)img * H * TW * G::PIXB;
    auto issue_plane = [&](int u, int o) {
        int ri = r0 - 1 + u;
        ri = ri < 0 ? 0 : (ri >= H ? H - 1 : ri);
        const unsigned dst = lds0 + (unsigned)((u % D) * G::SLOT + o * G::PLB + 16);
#ifdef TAG_ROWS_LINEAR_DMA   // timing experiment only (WRONG data): every DMA instruction reads 1 KB of consecutive bytes
        if (TW == 64 || lane < TW)
            glds16(ximg + (size_t)ri * TW * G::PIXB + lane * 16 + o * TW * 16, __builtin_amdgcn_readfirstlane(dst));
#else
        if (TW == 64 || lane < TW) {
#if TAG_ROWS_DMA_SADDR
            glds16s(ximg + (size_t)ri * TW * G::PIXB + o * 16, (unsigned)(lane * G::PIXB), __builtin_amdgcn_readfirstlane(dst));
#else
            glds16(ximg + (size_t)ri * TW * G::PIXB + lane * G::PIXB + o * 16, __builtin_amdgcn_readfirstlane(dst));
#endif
        }
#endif
    };
    // BatchNorm + ReLU of the producer, in place: the 8 channels of the plane are wave-uniform
    auto transform_plane = [&](int u, int o) {
        if (TW == 64 || lane < TW) {
            unsigned char* pp = smem + (u % D) * G::SLOT + o * G::PLB + 16 + lane * 16;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(pp);
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(Ss + o * 8), s1 = *reinterpret_cast<const f32x4*>(Ss + o * 8 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(Ss + CIN + o * 8), t1 = *reinterpret_cast<const f32x4*>(Ss + CIN + o * 8 + 4);
            u32x4 v;
            v.x = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.x), s0.x, t0.x), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.x), s0.y, t0.y), 0.0f));
            v.y = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.y), s0.z, t0.z), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.y), s0.w, t0.w), 0.0f));
            v.z = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.z), s1.x, t1.x), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.z), s1.y, t1.y), 0.0f));
            v.w = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.w), s1.z, t1.z), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.w), s1.w, t1.w), 0.0f));
            *reinterpret_cast<u32x4*>(pp) = v;
        }
    };
    // the planes this wave transforms in every matrix phase are always o = w4 + 4 k: their scale / shift stay in registers
    float tsc[KX][8], tsh[KX][8];
    if (PRO != 0) {
#pragma unroll
        for (int k = 0; k < KX; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) { tsc[k][i] = in_scale[(w4 + 4 * k) * 8 + i]; tsh[k][i] = in_shift[(w4 + 4 * k) * 8 + i]; }
    }
    // the wave's own yref tile of output row rho (EPI 2): 32 px x 64 B, lane-linear = [pixel][4 x 16 B]
    unsigned char* win0 = smem + G::OFF_WIN + wid * G::NWIN * G::WINB;
    const char* yimg = reinterpret_cast<const char*>(epi.yref) + (size_t)img * H * TW * Cout * 2;
    auto issue_yref = [&](int rho, int window) {
        if constexpr (EPI == 2) {
            int row = r0 + rho;
            row = row >= H ? H - 1 : row;
#pragma unroll
            for (int k = 0; k < G::KY; ++k) {
                [[maybe_unused]] const int px = wm * 32 + k * 16 + (lane >> 2);
                const unsigned dst = lds0 + (unsigned)(G::OFF_WIN + (wid * G::NWIN + window) * G::WINB + k * 1024);
#if TAG_ROWS_DMA_SADDR
                glds16s(yimg + ((size_t)row * TW + wm * 32 + k * 16) * Cout * 2 + (n0 + wn * 32) * 2,
                        (unsigned)((lane >> 2) * Cout * 2 + (lane & 3) * 16), __builtin_amdgcn_readfirstlane(dst));
#else
                glds16(yimg + ((size_t)row * TW + px) * Cout * 2 + (n0 + wn * 32) * 2 + (lane & 3) * 16,
                       __builtin_amdgcn_readfirstlane(dst));
#endif
            }
        }
    };

    // ---- A-fragment lane base: plane kl, pixel slot = column (the slot in front of column 0 is the plane's zero head) ----
#if TAG_ROWS_M16
    // 16x16x32: lane = (pixel lane & 15 of a 16-pixel block, k octet lane >> 4 of a 32-channel group); block mi at + mi * 256 bytes
    const unsigned abase = (unsigned)((lane >> 4) * G::PLB + (wm * 32 + (lane & 15)) * 16);   // + slot + (k32 * 4 * PLB + mi * 256 + kx * 16)
    constexpr int NCH = 2;                                  // couts per lane: ni * 16 + (lane & 15)
    // result element e = (mi * 2 + ni) * 4 + r: pixel mi * 16 + 4 (lane >> 4) + r of the wave's 32, cout ni * 16 + (lane & 15) of its 32
    auto PIX = [&](int e) { return (e >> 3) * 16 + 4 * (lane >> 4) + (e & 3); };
    auto CHL = [&](int e) { return ((e >> 2) & 1) * 16 + (lane & 15); };
    auto CHI = [](int e) { return (e >> 2) & 1; };
#else
    const int col = wm * 32 + row_to_pix(ml);
    const unsigned abase = (unsigned)(kl * G::PLB + col * 16);   // + slot base + (ks * 2 * PLB + kx * 16) as an immediate
    constexpr int NCH = 1;
    // result element e = register e of the 32x32 tile: D col = lane & 31 (cout), D row = (e&3) + 8 (e>>2) + 4 kl -> pixel via row_to_pix
    auto PIX = [&](int e) { return row_to_pix((e & 3) + 8 * (e >> 2) + 4 * kl); };
    auto CHL = [&](int e) { return ml; };
    auto CHI = [](int e) { return 0; };
#endif

    // ---- epilogue constants (per cout of this lane) ----
    float e_sc[NCH], e_sh[NCH], e_mu[NCH], e_is[NCH], st_mu[NCH], st_a[NCH], st_b[NCH];   // EPI 1: pivot, sum(y - pivot), sum (y - pivot)^2
#pragma unroll                                                                            // EPI 2: sum g, sum g * xhat
    for (int c = 0; c < NCH; ++c) {
        const int n = n0 + wn * 32 + CHL(c * 4);
        e_sc[c] = e_sh[c] = e_mu[c] = e_is[c] = st_mu[c] = st_a[c] = st_b[c] = 0.0f;
        if (EPI == 2) { e_sc[c] = epi.scale[n]; e_sh[c] = epi.shift[n]; e_mu[c] = epi.mean[n]; e_is[c] = epi.invstd[n]; }
    }

    // Ring index u = input row r0 - 1 + u.  The matrix phase of output row rho (phase rho) reads u = rho, rho + 1, rho + 2 and,
    // in the issue shadow of its MFMAs, applies the producer BatchNorm + ReLU to row rho + 3 (first read in phase rho + 1); the
    // tail of row rho (phase rho + 1, beside the OTHER group's matrix phase) issues the DMA of row rho + 2 PF + 3, which the same
    // wave transforms PF of its matrix phases later (phase rho + 2 PF).
    // ---- prime: ring rows 0 .. 2 + 2 PF (everything a tail before phase 1 would have issued), this wave's first PF yref tiles ----
    {
        constexpr int U0 = 3 + 2 * PF;
        for (int e = wid; e < U0 * NPL; e += 8) issue_plane(e / NPL, e % NPL);
#pragma unroll
        for (int j = 0; j < PF; ++j) issue_yref(grp + 2 * j, j);
        wait_vmcnt<0>();
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(bq[f]));      // hipcc's own wait for the weight loads goes HERE
        lds_fence_barrier();                                // every primed row is visible to every wave
        if (PRO != 0)
            for (int e = wid; e < 3 * NPL; e += 8) transform_plane(e / NPL, e % NPL);   // the rows of output row 0
    }
#ifdef TAG_ROWS_PROF   // 0 barriers, 1 MFMA phase (+ transform), 2 wait for the DMA, 3 fold, 4 pack + window + statistics, 5 output stores, 6 DMA issue
    unsigned long long rpc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rp0 = __builtin_amdgcn_s_memtime();
#endif
    lds_fence_barrier();                                    // start of phase 0
    if (grp == 1) lds_fence_barrier();                      // group 1 runs one phase behind
    RP_MARK(0)

    int j = 0;                                              // this wave's row counter (rho = grp + 2 j)
    for (int rho = grp; rho < T; rho += 2, ++j) {
        // ================= matrix phase of output row rho: input ring rows rho, rho + 1, rho + 2 =================
        unsigned va[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ri = r0 - 1 + rho + ky;
            va[ky] = abase + (unsigned)(((unsigned)ri < (unsigned)H ? (rho + ky) % D : D) * G::SLOT);
        }
        // ring row rho + 3 (issued PF tails ago by this wave) has landed: its planes are transformed between the MFMAs below
        wait_vmcnt<G::NWAIT_X>();
        RP_MARK(2)
        u32x4 traw[KX], tout[KX];
        unsigned char* tpp[KX];
        if (PRO != 0) {
#pragma unroll
            for (int k = 0; k < KX; ++k) {
                tpp[k] = smem + ((rho + 3) % D) * G::SLOT + (w4 + 4 * k) * G::PLB + 16 + (TW == 64 ? lane : (lane & (TW - 1))) * 16;
                traw[k] = *reinterpret_cast<const u32x4*>(tpp[k]);
            }
        }
        // one packed dword (two channels) of one plane of the producer transform, placed behind product f
        constexpr int TF0 = 6;
        auto transform_step = [&](int f) {
            if (PRO != 0 && f >= TF0 && f < TF0 + 4 * KX) {
                const int k = (f - TF0) / 4, d = (f - TF0) % 4;
                const unsigned w = traw[k][d];
                tout[k][d] = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(w), tsc[k][2 * d], tsh[k][2 * d]), 0.0f),
                                           fmaxf(fmaf(tag_bf16_hi(w), tsc[k][2 * d + 1], tsh[k][2 * d + 1]), 0.0f));
                __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                if (d == 3) {
                    if (TW == 64 || lane < TW) *reinterpret_cast<u32x4*>(tpp[k]) = tout[k];
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
        };
        float val[16];                                      // the wave's 32 px x 32 co results, element e <-> (PIX(e), CHL(e))
        constexpr int NA = TAG_ROWS_NA;
#if TAG_ROWS_M16
        // 9 taps x KS/2 k32 steps x (2 A fragments, 4 MFMAs on 4 independent accumulators): NA - 1 A reads stay in flight
        f32x4 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        constexpr int NAF = 9 * (KS / 2) * 2;               // A fragments per row: index a = (tap * KS/2 + k32) * 2 + mi
        auto a_ptr = [&](int a) {
            const int mi = a & 1, k32 = (a >> 1) % (KS / 2), tap = (a >> 1) / (KS / 2), ky = tap / 3, kx = tap % 3;
            return smem + va[ky] + (k32 * 4 * G::PLB + mi * 256 + kx * 16);
        };
        u32x4 af[NA];
#pragma unroll
        for (int a = 0; a < NA - 1; ++a) af[a] = *reinterpret_cast<const u32x4*>(a_ptr(a));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NAF; ++a) {
            if (a + NA - 1 < NAF) af[(a + NA - 1) % NA] = *reinterpret_cast<const u32x4*>(a_ptr(a + NA - 1));
            const int mi = a & 1, fb = (a >> 1) * 2;        // B fragments fb (ni = 0), fb + 1 (ni = 1)
            acc[mi][0] = mfma16_bf16(af[a % NA], bq[fb], acc[mi][0]);
            acc[mi][1] = mfma16_bf16(af[a % NA], bq[fb + 1], acc[mi][1]);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // pin: 2 MFMAs, 1 LDS read
            if (a + NA - 1 < NAF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            transform_step(a);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 16; ++e) val[e] = acc[e >> 3][(e >> 2) & 1][e & 3];
#else
        f32x16 acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
        auto a_ptr = [&](int f) {
            const int tap = f / KS, ks = f % KS, ky = tap / 3, kx = tap % 3;
            return smem + va[ky] + (ks * 2 * G::PLB + kx * 16);
        };
        // NA - 1 reads stay in flight: LDS latency under 8 waves per CU is several MFMA issue times
        u32x4 af[NA];
#pragma unroll
        for (int f = 0; f < NA - 1; ++f) af[f] = *reinterpret_cast<const u32x4*>(a_ptr(f));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (f + NA - 1 < NF) af[(f + NA - 1) % NA] = *reinterpret_cast<const u32x4*>(a_ptr(f + NA - 1));
            acc[f % NACC] = mfma_bf16(af[f % NA], bq[f], acc[f % NACC]);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // pin: 1 MFMA, 1 LDS read, (one dword pair of the transform)
            if (f + NA - 1 < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            transform_step(f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 16; ++e) val[e] = acc[0][e] + (NACC == 2 ? acc[NACC - 1][e] : 0.0f);
#endif
#ifdef TAG_ROWS_PROF
        asm volatile("s_nop 0" :: "v"(val[0]), "v"(val[15]));     // the last products have retired
#endif
        RP_MARK(1)
        lds_fence_barrier();                                // end of phase rho
        RP_MARK(0)

        // ================= tail of output row rho (phase rho + 1; the other group multiplies) =================
        // the yref tile of row rho was issued PF tails ago by this wave
        if (EPI == 2) wait_vmcnt<G::NWAIT_Y>();
        RP_MARK(2)
        RP_MARK(3)
        // Window = [32 pixels][32 couts] bf16: EPI 2 reads the yref values first, then the packed outputs overwrite them (no other
        // wave touches the window)
        unsigned char* win = win0 + (EPI == 2 ? (j % PF) * G::WINB : 0);
        float yv[EPI == 2 ? 16 : 1];
        if (EPI == 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                yv[e] = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(win + PIX(e) * 64 + CHL(e) * 2) << 16);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
            *reinterpret_cast<unsigned short*>(win + PIX(e) * 64 + CHL(e) * 2) = (unsigned short)tag_pack_bf16(val[e], 0.0f);
        if (EPI == 1) {
            if (j == 0) {                                   // pivot per cout = mean of the wave's first tile
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    float sm = 0.0f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) sm += CHI(e) == c ? val[e] : 0.0f;
                    if (NCH == 2) sm += __shfl_xor(sm, 16, 64);
                    sm += __shfl_xor(sm, 32, 64);
                    st_mu[c] = sm * (1.0f / 32.0f);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = val[e] - st_mu[CHI(e)];
                st_a[CHI(e)] += d;
                st_b[CHI(e)] = fmaf(d, d, st_b[CHI(e)]);
            }
        }
        if (EPI == 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = CHI(e);
                const float g = fmaf(yv[e], e_sc[c], e_sh[c]) > 0.0f ? val[e] : 0.0f;
                st_a[c] += g;
                st_b[c] = fmaf(g, (yv[e] - e_mu[c]) * e_is[c], st_b[c]);
            }
        }
        RP_MARK(4)
        // the wave's 32 px x 32 co tile = 128 pieces of 16 B: two per lane, 4 lanes per pixel (the window is lane-linear)
#pragma unroll
        for (int k = 0; k < G::MST; ++k) {
            const int piece = lane + 64 * k, pm = piece >> 2, c4 = piece & 3;
            const u32x4 v = *reinterpret_cast<const u32x4*>(win + piece * 16);
            gstore16(y + (((size_t)img * H + r0 + rho) * TW + wm * 32 + pm) * Cout + n0 + wn * 32 + c4 * 8, v);
        }
        RP_MARK(5)
        // the DMA this group owes the ring (row rho + 2 PF + 3) and this wave's yref tile PF rows ahead (into the window just used)
#pragma unroll
        for (int k = 0; k < KX; ++k) issue_plane(rho + 2 * PF + 3, w4 + 4 * k);
        issue_yref(rho + 2 * PF, j % PF);
        RP_MARK(6)
        lds_fence_barrier();                                // end of phase rho + 1
        RP_MARK(0)
    }
    // both groups pass T + 1 phase barriers: group g has passed g + 2 * its rows
    if (((T + 1 - grp) & 1) != 0) lds_fence_barrier();
    wait_vmcnt<0>();                                        // no DMA may still target this workgroup's LDS
#ifdef TAG_ROWS_PROF
    if (blockIdx.x == 77 && tid == 0) {
        for (int i = 0; i < 8; ++i) tag_rows_prof[i] = rpc[i];
        tag_rows_prof[8] = (unsigned long long)j;
        tag_rows_prof[9] = __builtin_amdgcn_s_memtime() - rp_t0;             // whole workgroup, shader clocks
        tag_rows_prof[10] = __builtin_amdgcn_s_memrealtime() - rp_r0;        // the same in 100 MHz ticks
    }
#endif

    // ---- one partial row per (image, strip, group, wave M-block); n-tiles and waves write disjoint channels ----
    if (EPI != 0) {
        const int prow = ((img * NS + strip) * 2 + grp) * WM + wm;
        const int P = B * NS * 2 * WM;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (NCH == 2) { st_a[c] += __shfl_xor(st_a[c], 16, 64); st_b[c] += __shfl_xor(st_b[c], 16, 64); }
            st_a[c] += __shfl_xor(st_a[c], 32, 64);
            st_b[c] += __shfl_xor(st_b[c], 32, 64);
            const int n = n0 + wn * 32 + CHL(c * 4);
            const bool writer = NCH == 2 ? lane < 16 : kl == 0;
            if (EPI == 1) {
                float* ps = stats + (size_t)prow * 3 * Cout;
                if (writer) { ps[n] = st_mu[c]; ps[Cout + n] = st_a[c]; ps[2 * Cout + n] = st_b[c]; }
            } else {
                float* ps = stats + (size_t)prow * 2 * Cout;
                if (writer) { ps[n] = st_a[c]; ps[Cout + n] = st_b[c]; }
            }
        }
        if (EPI == 1 && nt == 0 && wn == 0 && lane == 0) stats[(size_t)P * 3 * Cout + prow] = (float)(j * 32);
    }
}

// strips per image: one 8-wave workgroup per CU in one residency round, strips of at least 16 rows
int rows_strips(int B, int H, int NT) {
    static int cus = 0;
    if (!cus) { cus = tag_device_cu_count(); if (cus <= 0) cus = 256; }
    int ns = (cus + B * NT - 1) / (B * NT);
    if (ns > H / 16) ns = H / 16;
    return ns < 1 ? 1 : ns;
}

struct RowsCfg { int TW, CIN, WN; };
// the (width, Cin, Cout) shapes this kernel takes: a group covers a whole row (W = 64: 2 x 2 waves; W = 32: 1 x 4 waves) and
// the weights of a 32-cout slice fit the register file beside two waves per SIMD (Cin = 64: 144 VGPRs).  The Cin = 128 layers
// (288 weight registers = ONE wave per SIMD, every phase of a step serialised) measured 0.66-0.96 of the tile kernel's speed
// in the first form of this file and stay on the tile kernel.
bool rows_cfg(int H, int W, int Cin, int Cout, RowsCfg* c) {
    int wn = 0;
    if (W == 64 && Cin == 64 && Cout == 64) wn = 2;
    else if (W == 32 && Cin == 64 && Cout % 128 == 0) wn = 4;
    if (!wn || H < 2) return false;
    if (c) { c->TW = W; c->CIN = Cin; c->WN = wn; }
    return true;
}

template <int TW, int CIN, int WN, int PRO, int EPI>
void launch_rows(const bf16_t* x, const u32x4* wp, const float* s, const float* t, bf16_t* y, float* stats, const RowsEpi& epi,
                 int B, int H, int Cout, hipStream_t st) {
    using G = RowsGeom<TW, CIN, WN, EPI>;
    const int NT = Cout / (WN * 32), NS = rows_strips(B, H, NT);
    const int lds = G::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rows_kernel<TW, CIN, WN, PRO, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_rows_kernel<TW, CIN, WN, PRO, EPI>), dim3(B * NS * NT), dim3(512), lds, st, x, wp, s, t, y, stats,
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
    if (g_rows_on < 0) g_rows_on = tag_option("conv_rows") ? 1 : 0;
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
    return B * rows_strips(B, H, Cout / (c.WN * 32)) * 2 * (4 / c.WN);
}

// epi_kind 0 none, 1 statistics, 2 BatchNorm-backward sums (yref, bn_*)
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
#undef ROWS_CASE
    return TAG_EINVAL;
}
