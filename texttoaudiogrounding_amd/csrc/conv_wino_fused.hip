// Winograd F(2x2,3x3) convolution as ONE kernel per launch (round 6): the input transform B^T d B is formed on the way into LDS,
// the 16 products M_xi = V_xi . U_xi run into 16 accumulator sets that never leave the registers, and the output transform
// A^T m A + the epilogue forms of conv3x3_halo_kernel (BatchNorm statistics / BatchNorm-backward sums / pool-backward sums /
// inference pooling) are applied from a parked output tile.  Replaces the plane form of conv_wino.hip (wino_input_kernel ->
// batched gemm_kernel<..,16> -> wino_output_kernel: V and M planes of 4 x the activation each, written and read once per launch:
// 47 of the 86.5 GB per training step in round 5) for the convs of models/panns.py:29-38,49-50 in Cnn8Rnn blocks 3-4
// (models/audio_encoder.py:134-138).  Same arithmetic: fp32 transforms on the VALU with constants 0, +-1, 1/2, products on
// v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain).
//
// wino_fused_kernel<PRO, EPI>: 512 threads (8 waves, two per SIMD) own 64 tiles x 64 couts x 16 xi:
//   * wave (h, wm, wn): xi half h (transform rows 2h, 2h+1), 32 couts x 32 tiles -> 8 accumulator sets of 16 registers = 128 AGPRs;
//     A operand = U (rows = couts), B operand = V (columns = tiles): a lane then holds 4 CONSECUTIVE couts of one tile per register
//     quad, so the parked tile is written with 16-byte LDS stores and no lane transposes.
//   * K chunk = 8 input channels.  Per chunk every thread stages ONE (tile, channel quad, window column s) item of V -- 4 global loads
//     (the window column's 4 rows; the addresses are fixed for the whole K loop, only the channel offset moves), producer prologue +
//     zero padding, the column transform locally and the row transform with ONE quad-permute DPP exchange per value, 4 ds_write_b128
//     -- and ONE 4 x 4 (k x n) block of U per xi: 4 global loads, the transpose is register naming, 4 ds_write_b128.
//   * LDS image of both operands: [xi][row 64][k 8] with the 16-byte slot XOR-ed by (row >> 3) & 1 and planes 2080 B apart: the
//     fragment of four k-steps is ONE conflict-free ds_read_b128 per operand, and the 8 lanes of a ds_write_b128 group (same row,
//     different xi / slot) cover 128 contiguous bytes.  Two stages of (V + U) = 133 KB; one barrier per chunk.
//   * pipeline: global loads of chunk c+2 | transform + LDS writes of chunk c+1 | MFMAs of chunk c, the staging work cut into
//     slices between the 8 MFMA groups of a chunk (sched_barrier keeps the slices where they are).  Beside the fp32 MFMA every
//     VALU instruction is matrix time (tools/coissue2_probe.hip), so the loop carries the MINIMUM of them: 16 + 16 for the two
//     transforms of a thread's 16 values (the row transform is one in-place v_fmac_f32_dpp per value), nothing for addresses (the
//     loop body exists once per pipeline stage, the four operand buffers lie [V0 | V1 | U0 | U1] and every stage offset is an
//     immediate of the LDS instruction), no register copies (transforms written element by element, file built without the SLP
//     vectoriser: csrc/Makefile).
//   * epilogue: each wave reduces its 8 sets to its share of the 2 x 2 outputs (A^T . A is linear, the two xi halves add), parks
//     it in LDS [half][pixel of the tile][tile][cout]; then thread = (pixel, cout quad): the two halves are added, y leaves with
//     fully coalesced 16-byte stores and the epilogue sums are formed per thread, folded over the workgroup through LDS in a fixed
//     order: ONE partial row per 64-tile block (P = number of blocks).
// Workgroup order: groups of 8 tile blocks x all cout blocks, tile block fastest, an XCD gets a contiguous run (xcd_remap): the
// 32 workgroups resident on an XCD share 4 cout blocks of U and 8 tile blocks of x in its L2.
//
// wino_fused_wgrad_kernel: dw = G^T [ sum_t (A dY A^T) (.) (B^T d B) ] G with BOTH transforms formed at staging: 64 ci x 64 co x
// 16 xi accumulators per workgroup, K = tiles in chunks of 8, split over S slices of the tile axis; partial 3 x 3 filters per
// (slice, xi half) are folded in a fixed order by wino_fused_wgrad_finish_kernel.
#include <type_traits>
#include "conv_wino.h"

#ifndef TAG_WG_ABL
#define TAG_WG_ABL 0        // weight-gradient ablations: bit 0 no tile advance, bit 1 no gradient transform, bit 2 no input transform
#endif
#ifndef TAG_WF_ABL
#define TAG_WF_ABL 0        // ablation builds (tools/wino_fused_abl.sh; results wrong by construction): bit 0 no x loads in the K loop,
#endif                      // bit 1 no U loads, bit 2 no transform VALU (raw rows stored), bit 3 no LDS stores (values kept alive), bit 4 no barrier, bit 5 no epilogue, bit 6 no epilogue sums, bit 7 no output stores, bits 8 / 9 x / U loads always of chunk 0

namespace {

constexpr int FPL = 64 * 8 + 8;               // floats per xi plane of an operand buffer (2080 B: planes rotate by 32 B mod 128)
constexpr int FBUF = 16 * FPL;                // one operand buffer (V or U) of one pipeline stage
constexpr int FPARK_LD = 68;                  // floats per (pixel, tile) row of the parked tile (272 B: conflict-free 16-byte stores)
constexpr int FPARK = 2 * 4 * 64 * FPARK_LD;  // [xi half][pixel of the 2 x 2 tile][tile][cout]
constexpr int FRED = 32 * 16 * 8 + 32;        // per-slot epilogue sums + counts
constexpr int FLDS = FPARK + FRED;            // 155,776 B
constexpr int FSS = 4 * FBUF;                 // producer scale / shift (2 x Cin floats) behind the operand buffers
constexpr int FGM = 8;                        // tile blocks per workgroup-order group
static_assert(FSS + 2048 <= FLDS, "scale/shift staging must fit");

template <int PRO>
__device__ __forceinline__ f32x4 fused_prologue(f32x4 v, f32x4 s, f32x4 t) {      // = apply_prologue of conv.hip
    if (PRO == 1) {
        v.x = fmaxf(fmaf(v.x, s.x, t.x), 0.0f); v.y = fmaxf(fmaf(v.y, s.y, t.y), 0.0f);
        v.z = fmaxf(fmaf(v.z, s.z, t.z), 0.0f); v.w = fmaxf(fmaf(v.w, s.w, t.w), 0.0f);
    } else if (PRO == 2) {
        v.x = fmaf(v.x > 0 ? v.x : 0.1f * v.x, s.x, t.x); v.y = fmaf(v.y > 0 ? v.y : 0.1f * v.y, s.y, t.y);
        v.z = fmaf(v.z > 0 ? v.z : 0.1f * v.z, s.z, t.z); v.w = fmaf(v.w > 0 ? v.w : 0.1f * v.w, s.w, t.w);
    } else if (PRO == 3) {
        v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y); v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
    }
    return v;
}

// v[k] += f * (the value v[k] has in lane {2, 2, 1, 1}[s] / {0, 0, 3, 3}[s] of the quad), in place, as ONE v_fmac_f32_dpp per value:
// for the same expression the compiler selects v_mov_b32_dpp + v_fma_f32 / v_pk_fma_f32 (it has no DPP form of the three-address
// fma), and beside the fp32 MFMA every VALU instruction is matrix time.  All four values of a 16-byte LDS store go through ONE asm
// statement (an earlier per-value form cost three register copies per value on the way to the store: 96 v_mov per chunk in the
// weight-gradient kernel; this one 3-4 per chunk).  s_nop 1: a DPP read needs two wait states after the VALU write of its source,
// and the hazard recogniser does not look inside inline assembly.
#define TAG_QUAD_FMAC(NAME, PERM)                                                                                             \
    __device__ __forceinline__ void NAME(const float (&t)[4], float f, f32x4& v) {                                           \
        float a = t[0], b = t[1], c = t[2], d = t[3];                                                                        \
        asm("s_nop 1\n\t"                                                                                                    \
            "v_fmac_f32_dpp %0, %0, %4 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                       \
            "v_fmac_f32_dpp %1, %1, %4 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                       \
            "v_fmac_f32_dpp %2, %2, %4 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                       \
            "v_fmac_f32_dpp %3, %3, %4 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1"                            \
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                                                             \
            : "v"(f));                                                                                                       \
        v = (f32x4){a, b, c, d};                                                                                             \
    }
TAG_QUAD_FMAC(quad_fmac_2211, "[2,2,1,1]")
TAG_QUAD_FMAC(quad_fmac_0033, "[0,0,3,3]")
#undef TAG_QUAD_FMAC

// the wave's share of the 2 x 2 outputs from its 8 accumulator sets (xi = 4 r + s, r = 2 HH + {0, 1}), register quad rq
template <int HH>
__device__ __forceinline__ void fused_out_quad(const f32x16 (&acc)[8], int rq, f32x4 (&o)[4]) {
    f32x4 p0[4], p1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float m0 = acc[s][4 * rq + k], m1 = acc[4 + s][4 * rq + k];
            if (HH == 0) { p0[s][k] = m0 + m1; p1[s][k] = m1; }          // rows r = 0, 1 of  [[1,1,1,0],[0,1,-1,-1]]
            else { p0[s][k] = m0; p1[s][k] = -m0 - m1; }                 // rows r = 2, 3
        }
    o[0] = p0[0] + p0[1] + p0[2];
    o[1] = p0[1] - p0[2] + p0[3];            // (column 3 of V is stored negated: p[3] = -(A^T m)[.][3])
    o[2] = p1[0] + p1[1] + p1[2];
    o[3] = p1[1] - p1[2] + p1[3];
}

template <int PRO, int EPI>
__global__ __launch_bounds__(512) void wino_fused_kernel(const float* __restrict__ x, const float* __restrict__ U,
                                                         const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                         float* __restrict__ y, float* __restrict__ stats, WinoEpi epi, int B, int H,
                                                         int W, int Cin, int Cout, int th, int tw, long T, int m_tiles, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef TAG_WF_PROF
    const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
    // ---- workgroup -> (tile block mt, cout block nt)
    int mt, nt;
    {
        const int L = xcd_remap(blockIdx.x, m_tiles * n_tiles);
        const int gsz = FGM * n_tiles, g = L / gsz, rem = L - g * gsz;
        const int mcount = (m_tiles - g * FGM) < FGM ? (m_tiles - g * FGM) : FGM;
        nt = rem / mcount;
        mt = g * FGM + (rem - nt * mcount);
    }
    const long m0 = (long)mt * 64;
    const int n0 = nt * 64;
    const int nch = Cin >> 3;

    // ---- V staging role: (tile tl, channel quad q of the chunk, window column s).  Buffer loads: descriptor + chunk offset in
    // SGPRs, ONE 32-bit VGPR offset per window row that never changes (tools/coissue2_probe.hip: a global load with a 64-bit VGPR
    // address costs the fp32 MFMA stream 24 clocks, the scalar-base form 7); rows outside the image get an offset beyond the
    // descriptor's range and come back as zeros -- no mask arithmetic when there is no producer prologue.
    const int s = tid & 3, q = (tid >> 2) & 1, tl = tid >> 3;
    // V[r][s] = tt[r][s] + fb tt[r][{2,2,1,1}[s]]; column s = 3 is stored NEGATED (tt3 - tt1: one fma per value for every lane), which
    // negates the products M[r][3] exactly -- the output transform adds them instead of subtracting
    const float fb = s == 1 ? 1.0f : -1.0f;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)B * H * W * Cin * 4), 0x00020000);
    unsigned xo[4];
    float okb[4];                                      // PRO == 1: upper clamp of the row (+inf inside the image, 0 outside)
    bool okr[4];
    {
        // (32-bit arithmetic: T < 2^27 (tag_conv3x3_wino_ok); a 64-bit division is several hundred clocks, and with one workgroup
        // per CU nothing hides a workgroup's set-up)
        const unsigned t = (unsigned)m0 + (unsigned)tl;
        const bool okt = t < (unsigned)T;
        const unsigned tt_ = okt ? t : 0u;
        const unsigned bi = tt_ / (unsigned)tw;
        const int j = (int)(tt_ - bi * (unsigned)tw);
        const int b = (int)(bi / (unsigned)th), i = (int)(bi - (unsigned)b * (unsigned)th);
        const int w = 2 * j - 1 + s;
        const bool okw = okt && (unsigned)w < (unsigned)W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = 2 * i - 1 + r;
            okr[r] = okw && (unsigned)h < (unsigned)H;
            okb[r] = okr[r] ? INFINITY : 0.0f;
            xo[r] = okr[r] ? (unsigned)(((((size_t)b * H + h) * W + w) * Cin + 4 * q) * sizeof(float)) : 0x80000000u;
        }
    }
    const int vw0 = s * FPL + tl * 8 + ((q ^ ((tl >> 3) & 1)) << 2);         // + 4 r FPL

    // ---- U staging role.  U is packed per (cout block, chunk) as the LDS image itself, [xi][n 64][k 8] (wino_pack_fused_kernel): a
    // thread copies pieces tid + 512 i of the 32 KB block: plane 4 i + (tid >> 7), row n = (tid & 127) >> 1, 16-byte slot tid & 1
    // (the k quad that belongs there is slot ^ ((n >> 3) & 1)) -- coalesced loads, conflict-free stores, no arithmetic.
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, (int)((size_t)16 * Cin * Cout * 4), 0x00020000);
    const int un = (tid & 127) >> 1, uslot = tid & 1;
    const unsigned uo0 = (unsigned)((((tid >> 7) * 64 + un) * 8 + ((uslot ^ ((un >> 3) & 1)) << 2)) * sizeof(float));   // + i * 4 planes
    const int uw0 = (tid >> 7) * FPL + un * 8 + (uslot << 2);                                                            // + i * 4 FPL
    const unsigned ublk = (unsigned)nt * (unsigned)nch;                     // block index of chunk 0

    // ---- MFMA role
    const int hh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int kl = lane >> 5, ml = lane & 31;
    const int fsw = (kl ^ ((ml >> 3) & 1)) << 2;
    const int aoff = 2 * FBUF + 8 * hh * FPL + (wm * 32 + ml) * 8 + fsw;      // U fragment (A operand); LDS = [V0 | V1 | U0 | U1]
    const int boff = 8 * hh * FPL + (wn * 32 + ml) * 8 + fsw;                 // V fragment (B operand)

    f32x16 acc[8];
    f32x4 xr[4], ur[4];
    auto load_x = [&](int c, int r) {
        xr[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[r], c * 32, 0));
    };
    auto load_u = [&](int c, int i) {
        ur[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uo0 + i * (4 * 64 * 8 * 4), (ublk + c) * (16 * 64 * 8 * 4), 0));
    };
    f32x4 d[4], vv;
    float tt[4][4];
    f32x4 psc = {1.0f, 1.0f, 1.0f, 1.0f}, psh = {0.0f, 0.0f, 0.0f, 0.0f};
    auto ld_ss = [&](int c) {
        if (PRO != 0) {
            psc = *reinterpret_cast<const f32x4*>(smem + FSS + c * 8 + 4 * q);
            psh = *reinterpret_cast<const f32x4*>(smem + FSS + Cin + c * 8 + 4 * q);
        }
    };
    auto x_row = [&](int r) {                          // producer prologue + zero padding of window row r
        if (TAG_WF_ABL & 4) return;
        if (PRO == 0) {
            d[r] = xr[r];                              // (rows outside the image were loaded as zeros)
        } else if (PRO == 1) {
            // relu(bn(x)) inside the image, 0 outside: the median of (bn(x), 0, +inf | 0) -- one operation for both
#pragma unroll
            for (int k = 0; k < 4; ++k) d[r][k] = __builtin_amdgcn_fmed3f(fmaf(xr[r][k], psc[k], psh[k]), 0.0f, okb[r]);
        } else {
            const f32x4 v = fused_prologue<PRO>(xr[r], psc, psh);
            d[r] = okr[r] ? v : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto x_col = [&](int r) {                          // column transform (B^T d) of the thread's window column
        if (TAG_WF_ABL & 4) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                  // (element by element: a packed add would put two values into a register PAIR,
            if (r == 0) tt[0][k] = d[0][k] - d[2][k];  //  and the row transform's in-place DPP instruction then needs copies out of it)
            if (r == 1) tt[1][k] = d[1][k] + d[2][k];
            if (r == 2) tt[2][k] = d[2][k] - d[1][k];
            if (r == 3) tt[3][k] = d[1][k] - d[3][k];
        }
    };
    auto v_row = [&](int r) {                          // row transform (. B): one quad-permute exchange per value
        if (TAG_WF_ABL & 4) return;
        quad_fmac_2211(tt[r], fb, vv);
    };
    auto v_write = [&](float* Vb, int r) {
#if TAG_WF_ABL & 4
        vv = xr[r];
#endif
#if TAG_WF_ABL & 8
        asm volatile("" ::"v"(vv));
#else
        *reinterpret_cast<f32x4*>(Vb + vw0 + 4 * r * FPL) = vv;
#endif
    };
    auto u_write = [&](float* Ub, int i) {
#if TAG_WF_ABL & 8
        asm volatile("" ::"v"(ur[i]));
#else
        *reinterpret_cast<f32x4*>(Ub + uw0 + 4 * i * FPL) = ur[i];
#endif
    };
    auto lx = [&](int c, int r) {
#if TAG_WF_ABL & 256
        load_x(0, r);                                  // ablation: always chunk 0 (cache-resident: issue cost without latency)
#elif !(TAG_WF_ABL & 1)
        load_x(c, r);
#endif
    };
    auto lu = [&](int c, int j) {
#if TAG_WF_ABL & 512
        load_u(0, j);
#elif !(TAG_WF_ABL & 2)
        load_u(c, j);
#endif
    };

#ifdef TAG_WF_PROF
    const unsigned long long pt1 = __builtin_readcyclecounter();
#endif
    // ---- prologue of the pipeline: chunk 0 into buffer 0, chunk 1 into the registers.  BOTH chunks are requested before anything
    // waits (the accumulators are not live yet, so chunk 1 has registers to land in): one exposed memory latency per workgroup
    // instead of two -- with one workgroup per CU nothing else covers it, and a 64-channel layer has only 8 chunks to amortise it over
    {
        const int c1 = nch > 1 ? 1 : 0;
        f32x4 x1[4], u1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { load_x(0, r); load_u(0, r); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x1[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[r], c1 * 32, 0));
            u1[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uo0 + r * (4 * 64 * 8 * 4), (ublk + c1) * (16 * 64 * 8 * 4), 0));
        }
        // (behind the requests, not in front of them: the scale / shift round trip and the zero fill run under the chunks' latency)
        if (PRO != 0) {
            for (int i = tid; i < Cin; i += 512) { smem[FSS + i] = in_scale[i]; smem[FSS + Cin + i] = in_shift[i]; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        __syncthreads();                               // (the producer scale / shift are in LDS)
        ld_ss(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) x_row(r);
#pragma unroll
        for (int r = 0; r < 4; ++r) x_col(r);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v_row(r); v_write(smem, r); u_write(smem + 2 * FBUF, r); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { xr[r] = x1[r]; ur[r] = u1[r]; }
        ld_ss(c1);
    }
    __syncthreads();

    // ---- K loop.  A chunk is 8 MFMA groups M_j (4 dependent MFMAs into accumulator set j, back to back) and 8 staging slices S_j
    // (chunk c + 1 from the registers into the other buffer, loads of chunk c + 2), one barrier.  On gfx950 NOTHING of the
    // staging hides behind an fp32 MFMA -- v_mfma_f32_32x32x2_f32 runs at the vector rate and streams its 16 accumulator registers
    // through the register file on every instruction; tools/coissue2_probe.hip: beside it every VALU operation costs ~3 clocks of
    // matrix time, a ds_write_b128 ~8, a scalar-base load ~7 (a 64-bit-address load 24, LDS-DMA 14-80), an LDS read ~1, in EVERY
    // order (fillers between groups, behind every MFMA, the two waves of a SIMD in complementary phases) -- so the loop is built
    // for the fewest instructions, not for overlap: bare MFMA loop 1.95 ms, staging +0.85 ms (round-6 first form) for the
    // 512 -> 512 layer.
#ifdef TAG_WF_PROF
    const unsigned long long pt2 = __builtin_readcyclecounter();
#endif
    // The loop body exists twice, once per pipeline stage: the stage offsets are then immediates of the LDS instructions (the four
    // operand buffers are laid out [V0 | V1 | U0 | U1] so that every toggle, 33,280 B, fits the 16-bit offset field) and a chunk
    // costs no address arithmetic -- it was 5 vector adds per wave and chunk, each ~3 clocks of matrix time.
    f32x4 afP, bfP, afQ, bfQ;
#define WF_M4(J, A_, B_)                                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                            \
        acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[e], B_[e], acc[J], 0, 0, 0)
#define WF_FRAG(A_, B_, J)                                                                                   \
    A_ = *reinterpret_cast<const f32x4*>(cur + aoff + (J) * FPL);                                            \
    B_ = *reinterpret_cast<const f32x4*>(cur + boff + (J) * FPL)
#define WF_SB __builtin_amdgcn_sched_barrier(0)
    auto chunk = [&](int c, auto stage) {
        constexpr int ST = decltype(stage)::value;
        const float* cur = smem + ST * FBUF;           // V of this chunk; its U is 2 FBUF further (aoff)
        float* nxt = smem + (1 - ST) * FBUF;
        const int c2 = c + 2 < nch ? c + 2 : nch - 1;
        WF_SB;
        WF_FRAG(afP, bfP, 0); WF_SB;
        WF_FRAG(afQ, bfQ, 1); WF_M4(0, afP, bfP); WF_SB; x_row(0); x_row(1); WF_SB;
        WF_FRAG(afP, bfP, 2); WF_M4(1, afQ, bfQ); WF_SB; x_row(2); x_row(3); x_col(0); x_col(1); x_col(2); x_col(3); WF_SB;
        WF_FRAG(afQ, bfQ, 3); WF_M4(2, afP, bfP); WF_SB; v_row(0); v_write(nxt, 0); v_row(1); v_write(nxt, 1); lx(c2, 0); lx(c2, 1); WF_SB;
        WF_FRAG(afP, bfP, 4); WF_M4(3, afQ, bfQ); WF_SB; v_row(2); v_write(nxt, 2); v_row(3); v_write(nxt, 3); lx(c2, 2); lx(c2, 3); WF_SB;
        WF_FRAG(afQ, bfQ, 5); WF_M4(4, afP, bfP); WF_SB; u_write(nxt + 2 * FBUF, 0); u_write(nxt + 2 * FBUF, 1); lu(c2, 0); lu(c2, 1); WF_SB;
        WF_FRAG(afP, bfP, 6); WF_M4(5, afQ, bfQ); WF_SB; u_write(nxt + 2 * FBUF, 2); u_write(nxt + 2 * FBUF, 3); lu(c2, 2); lu(c2, 3); WF_SB;
        WF_FRAG(afQ, bfQ, 7); WF_M4(6, afP, bfP); WF_SB; ld_ss(c2); WF_SB;
        WF_M4(7, afQ, bfQ); WF_SB;
#if !(TAG_WF_ABL & 16)
        __syncthreads();
#endif
    };
    {
        int c = 0;
        for (; c + 1 < nch; c += 2) {
            chunk(c, std::integral_constant<int, 0>{});
            chunk(c + 1, std::integral_constant<int, 1>{});
        }
        if (c < nch) chunk(c, std::integral_constant<int, 0>{});      // odd chunk count: the last chunk is in stage 0
    }
#undef WF_M4
#undef WF_FRAG
#undef WF_SB

#ifdef TAG_WF_PROF
    const unsigned long long pt3 = __builtin_readcyclecounter();
#endif
#if TAG_WF_ABL & 32
    if (acc[0][0] != 12345.678f) return;               // ablation: no epilogue at all (prologue + K loop only)
#endif
    // ---- output transform: each wave parks its share of the 2 x 2 outputs  [half][pixel 2a+e][tile][cout]
    {
        float* park = smem + hh * (4 * 64 * FPARK_LD) + (wn * 32 + ml) * FPARK_LD + wm * 32 + 4 * kl;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 o[4];
            if (hh == 0) fused_out_quad<0>(acc, rq, o); else fused_out_quad<1>(acc, rq, o);
#pragma unroll
            for (int ae = 0; ae < 4; ++ae) *reinterpret_cast<f32x4*>(park + ae * 64 * FPARK_LD + 8 * rq) = o[ae];
        }
    }
    __syncthreads();

#ifdef TAG_WF_PROF
    const unsigned long long pt4 = __builtin_readcyclecounter();
#endif
    // ---- thread = (pixel, cout quad): add the halves, store, epilogue sums
    const int cq = tid & 15, slot = tid >> 4;
    const float* pk = smem + 4 * cq;
    constexpr int HALF = 4 * 64 * FPARK_LD;
    int tb[2], ti[2], tj[2];
    bool tok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned t = (unsigned)m0 + (unsigned)(slot + 32 * u);
        tok[u] = t < (unsigned)T;
        const unsigned t2 = tok[u] ? t : 0u;
        const unsigned bi = t2 / (unsigned)tw;
        tj[u] = (int)(t2 - bi * (unsigned)tw);
        tb[u] = (int)(bi / (unsigned)th);
        ti[u] = (int)(bi - (unsigned)tb[u] * (unsigned)th);
    }
    const int co = n0 + 4 * cq;

    if (EPI == 3) {
        // inference: out (B, H/ph, W/2, C) = avg/max pool(relu(y * scale + shift)) of the tile's own 2 x 2 outputs (the expression and
        // summation order of bnact_pool_fwd_kernel / wino_output_kernel<3>)
        const f32x4 bsc = *reinterpret_cast<const f32x4*>(epi.scale + co), bsh = *reinterpret_cast<const f32x4*>(epi.shift + co);
        const int Hp = H / epi.ph, Wp = W >> 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile = slot + 32 * u;
            f32x4 a[4];
#pragma unroll
            for (int ae = 0; ae < 4; ++ae) {
                const float* p = pk + (ae * 64 + tile) * FPARK_LD;
                const f32x4 o = *reinterpret_cast<const f32x4*>(p) + *reinterpret_cast<const f32x4*>(p + HALF);
#pragma unroll
                for (int k = 0; k < 4; ++k) a[ae][k] = fmaxf(fmaf(o[k], bsc[k], bsh[k]), 0.0f);
            }
            if (!tok[u] || tj[u] >= Wp) continue;
            if (epi.ph == 2) {
                if (ti[u] < Hp) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sum = ((a[0][k] + a[1][k]) + a[2][k]) + a[3][k];
                        const float mx = fmaxf(fmaxf(fmaxf(a[0][k], a[1][k]), a[2][k]), a[3][k]);
                        o[k] = sum * epi.wavg + mx * epi.wmax;
                    }
                    *reinterpret_cast<f32x4*>(y + (((size_t)tb[u] * Hp + ti[u]) * Wp + tj[u]) * Cout + co) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int hp = 2 * ti[u] + r;
                    if (hp >= Hp) continue;
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = (a[2 * r][k] + a[2 * r + 1][k]) * epi.wavg + fmaxf(a[2 * r][k], a[2 * r + 1][k]) * epi.wmax;
                    *reinterpret_cast<f32x4*>(y + (((size_t)tb[u] * Hp + hp) * Wp + tj[u]) * Cout + co) = o;
                }
            }
        }
        return;
    }

#if TAG_WF_ABL & 64
    const bool want = false;                           // ablation: no epilogue sums
#else
    const bool want = (EPI == 0) ? (stats != nullptr) : true;
#endif
    f32x4 bsc = {0, 0, 0, 0}, bsh = {0, 0, 0, 0}, bmu = {0, 0, 0, 0}, bis = {0, 0, 0, 0};
    if (EPI == 1 || EPI == 2) {
        bsc = *reinterpret_cast<const f32x4*>(epi.scale + co); bsh = *reinterpret_cast<const f32x4*>(epi.shift + co);
        bmu = *reinterpret_cast<const f32x4*>(epi.mean + co); bis = *reinterpret_cast<const f32x4*>(epi.invstd + co);
    }
    f32x4 piv = {0, 0, 0, 0};
    if (EPI == 0 && want) piv = *reinterpret_cast<const f32x4*>(pk) + *reinterpret_cast<const f32x4*>(pk + HALF);   // pixel (0,0) of tile 0
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, cnt = 0.0f;
    size_t tbase[2];                                   // offset of the tile's first output pixel (the 64-bit products once per tile)
#pragma unroll
    for (int u = 0; u < 2; ++u) tbase[u] = (((size_t)tb[u] * H + 2 * ti[u]) * W + 2 * tj[u]) * Cout + co;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int ae = it >> 1, u = it & 1, a = ae >> 1, e = ae & 1;
        const int tile = slot + 32 * u;
        const float* p = pk + (ae * 64 + tile) * FPARK_LD;
        const f32x4 o = *reinterpret_cast<const f32x4*>(p) + *reinterpret_cast<const f32x4*>(p + HALF);
        const int h = 2 * ti[u] + a, w = 2 * tj[u] + e;
        if (!tok[u] || h >= H || w >= W) continue;
        const size_t off = tbase[u] + (size_t)((a * W + e) * Cout);
#if TAG_WF_ABL & 128
        if (o[0] == 12345.678f)                        // ablation: no output stores
#endif
        *reinterpret_cast<f32x4*>(y + off) = o;
        if (EPI == 0 && want) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float dv = o[k] - piv[k]; s1[k] += dv; s2[k] = fmaf(dv, dv, s2[k]); }
            cnt += 1.0f;
        }
        if (EPI == 1) {
            const f32x4 yr = *reinterpret_cast<const f32x4*>(epi.yref + off);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = fmaf(yr[k], bsc[k], bsh[k]) > 0.0f ? o[k] : 0.0f;
                s1[k] += g;
                s2[k] = fmaf(g, (yr[k] - bmu[k]) * bis[k], s2[k]);
            }
        }
        if (EPI == 2) {
            // o = dL/d(dropout(pool(relu(bn(yref))))) at pooled pixel (h, w) of the block BELOW: the arithmetic of
            // wino_output_kernel<2> / conv3x3_halo_kernel's EPI == 2 (dropout undone with one hash per 4 channels, a = bn(yref), ReLU
            // mask and first-maximum arg-max recomputed from the ph x 2 window)
            f32x4 g = o;
            if (epi.drop_p > 0.0f) {
                const uint64_t grp = ((uint64_t)((unsigned)tb[u] * (unsigned)H + (unsigned)h) * (unsigned)W + (unsigned)w)
                                     * (unsigned)(Cout >> 2) + (unsigned)(co >> 2);
                const uint64_t bits = tag_keep4_bits(epi.seed, grp);
                const float keep_scale = 1.0f / (1.0f - epi.drop_p);
                const unsigned thr = tag_keep4_threshold(epi.drop_p);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = tag_keep4(bits, k, thr) ? g[k] * keep_scale : 0.0f;
            }
            const float* yw = epi.yref + (((size_t)tb[u] * epi.Hf + (size_t)h * epi.ph) * epi.Wf + 2 * w) * Cout + co;
            f32x4 vw[4];
            vw[0] = *reinterpret_cast<const f32x4*>(yw);
            vw[1] = *reinterpret_cast<const f32x4*>(yw + Cout);
            if (epi.ph == 2) {
                vw[2] = *reinterpret_cast<const f32x4*>(yw + (size_t)epi.Wf * Cout);
                vw[3] = *reinterpret_cast<const f32x4*>(yw + (size_t)epi.Wf * Cout + Cout);
            } else { vw[2] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; vw[3] = vw[2]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a4[4];
#pragma unroll
                for (int u2 = 0; u2 < 4; ++u2) a4[u2] = fmaf(vw[u2][k], bsc[k], bsh[k]);
                if (epi.ph != 2) { a4[2] = -INFINITY; a4[3] = -INFINITY; }
                const float mx = fmaxf(fmaxf(a4[0], a4[1]), fmaxf(a4[2], a4[3]));
                const float gw = g[k] * epi.wavg, gwm = g[k] * (epi.wavg + epi.wmax);
                bool found = false;
#pragma unroll
                for (int u2 = 0; u2 < 4; ++u2) {
                    const bool eq = a4[u2] == mx;
                    const bool hit = eq && !found;
                    found = found || eq;
                    const float dz = a4[u2] > 0.0f ? (hit ? gwm : gw) : 0.0f;
                    s1[k] += dz;
                    s2[k] = fmaf(dz, (vw[u2][k] - bmu[k]) * bis[k], s2[k]);
                }
            }
        }
    }
    if (!want) return;
    // ---- fold the 32 slots in a fixed order: ONE partial row per tile block
    float* red = smem + FPARK;
    *reinterpret_cast<f32x4*>(red + (slot * 16 + cq) * 8) = (f32x4){s1[0], s1[1], s1[2], s1[3]};
    *reinterpret_cast<f32x4*>(red + (slot * 16 + cq) * 8 + 4) = (f32x4){s2[0], s2[1], s2[2], s2[3]};
    if (EPI == 0 && cq == 0) red[32 * 16 * 8 + slot] = cnt;
    __syncthreads();
    const int P = m_tiles;
    if (tid < 128) {
        float a = 0.0f;
        for (int sl = 0; sl < 32; ++sl) a += red[sl * 128 + tid];
        const int c = n0 + 4 * (tid >> 3) + (tid & 3), which = (tid >> 2) & 1;
        if (EPI == 0) stats[((size_t)mt * 3 + 1 + which) * Cout + c] = a;
        else stats[((size_t)mt * 2 + which) * Cout + c] = a;
    }
    if (EPI == 0) {
        if (slot == 0) *reinterpret_cast<f32x4*>(stats + (size_t)mt * 3 * Cout + co) = piv;
        if (tid == 128 && nt == 0) {
            float a = 0.0f;
            for (int sl = 0; sl < 32; ++sl) a += red[32 * 16 * 8 + sl];
            stats[(size_t)P * 3 * Cout + mt] = a;
        }
    }
#ifdef TAG_WF_PROF
    if (tid == 0 && (blockIdx.x == 1000 || blockIdx.x == 1001)) {
        const unsigned long long pt5 = __builtin_readcyclecounter();
        printf("wf prof block %d nch %d: setup %llu  prologue %llu  loop %llu (%llu per chunk)  park %llu  phase3+fold %llu  total %llu clocks\n", (int)blockIdx.x, nch,
               pt1 - pt0, pt2 - pt1, pt3 - pt2, (pt3 - pt2) / nch, pt4 - pt3, pt5 - pt4, pt5 - pt0);
    }
#endif
}

template <int PRO, int EPI>
void launch_fused(const float* x, const float* U, const float* s, const float* t, float* y, float* stats, const WinoEpi& epi, int B,
                  int H, int W, int Cin, int Cout, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fused_kernel<PRO, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, FLDS * (int)sizeof(float));
        attr_set = true;
    }
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long T = (long)B * th * tw;
    const int m_tiles = (int)((T + 63) / 64), n_tiles = Cout / 64;
    hipLaunchKernelGGL((wino_fused_kernel<PRO, EPI>), dim3(m_tiles * n_tiles), dim3(512), FLDS * sizeof(float), st, x, U, s, t, y, stats,
                       epi, B, H, W, Cin, Cout, th, tw, T, m_tiles, n_tiles);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient.  dU_xi (Cin x Cout) = sum over tiles V_xi[t][ci] D_xi[t][co],  V = B^T d B of the (prologued) input window,
// D = A dY A^T of the 2 x 2 gradient tile, dw = G^T dU G -- the arithmetic of wino_input_kernel / wino_dy_kernel /
// wino_wgrad_finish_kernel with BOTH transforms formed at staging and the 16 dU_xi kept in registers.  A workgroup owns 64 ci x
// 64 co x 16 xi over ONE slice of the tile axis (K chunks of 8 tiles; S slices so that the launch is one workgroup per CU);
// wave (h, wm, wn) = xi half, 32 ci x 32 co.  A wave stages ONE tile per chunk (lane = channel quad x window column s), so the
// tile geometry -- image, tile row / column, row validity, row offsets -- is wave-uniform and lives on the scalar unit, advanced
// by 8 tiles per chunk without a division; only the column validity needs a vector compare.  LDS images are k-major,
// [xi][tile 8][channel 64] (16-byte stores from the channel-quad threads; fragments are four ds_read_b32 per operand and group
// instead of one ds_read_b128 -- reads cost the MFMA stream ~1 clock each, a transposing store would be 32-bit LDS writes, the
// one form that waits for gaps in the matrix pipe).  Column 3 of both V and D is stored negated (their product is unchanged);
// row 3 of D too (-g1 would be an instruction per value; the products of row 3 come out negated and the fold subtracts them).
// Each wave folds its 8 sets to its share of the 3 x 3 filter and writes it to part[slice][half][co][ci][9];
// wino_fused_wgrad_finish_kernel<Q> adds the 2 S shares in a fixed order.
struct WgGeom { int th, tw; long T; int nci, nco, S, cps; };
inline WgGeom wg_geom(int B, int H, int W, int Cin, int Cout) {
    WgGeom g;
    g.th = (H + 1) / 2; g.tw = (W + 1) / 2;
    g.T = (long)B * g.th * g.tw;
    g.nci = Cin / 64; g.nco = Cout / 64;
    const long chunks = (g.T + 7) / 8;
    long S = 256 / ((long)g.nci * g.nco);
    if (S < 1) S = 1;
    if (S > chunks) S = chunks;
    const long cps = (chunks + S - 1) / S;
    g.cps = (int)cps;
    g.S = (int)((chunks + cps - 1) / cps);
    return g;
}


template <int PRO>
__global__ __launch_bounds__(512) void wino_fused_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                               const float* __restrict__ in_shift, const float* __restrict__ dy,
                                                               float* __restrict__ part, int B, int H, int W, int Cin, int Cout,
                                                               int th, int tw, int nci, int nco, int S, int cps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int sl, cib, cob;
    {
        const int per = nci * nco;
        const int L = xcd_remap(blockIdx.x, per * S);
        sl = L / per;
        const int rem = L - sl * per;
        cib = rem / nco;
        cob = rem - cib * nco;
    }
    // ---- staging role: tile = wave (8 tiles per chunk), lane = (channel quad, window column s)
    const int s = tid & 3, quad = (tid >> 2) & 15;
    const float fb = s == 1 ? 1.0f : -1.0f;                       // V[r][s] = tt[r][s] + fb tt[r][{2,2,1,1}[s]]  (column 3 negated)
    const float cb = s == 1 ? 1.0f : (s == 2 ? -1.0f : 0.0f);     // D[r][s] = R[r][s & 1] + cb R[r][{0,0,3,3}[s] & 1] (column 3 negated)
    // descriptors: x one pixel early, so that the window column 2 j - 1 of a tile is a non-negative offset from (row, 2 j); a row
    // outside the image gets the same descriptor with ZERO records (a scalar select: every load of the row comes back as zeros)
    const int xbytes = (int)(((size_t)B * H * W + 1) * Cin * 4), dbytes = (int)((size_t)B * H * W * Cout * 4);
    const unsigned xvo = (unsigned)((s * Cin + cib * 64 + 4 * quad) * 4);
    const unsigned dvo = (unsigned)(((s & 1) * Cout + cob * 64 + 4 * quad) * 4);
    const int vw0 = s * FPL + wave * 64 + 4 * quad;               // + 4 r FPL  (V image; the D image is 2 FBUF further)
    // tile of this wave in the chunk being LOADED (uniform): image b, tile row i, tile column j
    int tb, ti, tj;
    {
        const unsigned t = (unsigned)sl * (unsigned)cps * 8u + (unsigned)wave;
        const unsigned bi = t / (unsigned)tw;
        tj = (int)(t - bi * (unsigned)tw);
        tb = (int)(bi / (unsigned)th);
        ti = (int)(bi - (unsigned)tb * (unsigned)th);
        tb = __builtin_amdgcn_readfirstlane(tb); ti = __builtin_amdgcn_readfirstlane(ti); tj = __builtin_amdgcn_readfirstlane(tj);
    }
    auto advance = [&]() {                            // + 8 tiles
#if TAG_WG_ABL & 1
        return;                                        // ablation: the same tile every chunk (no scalar geometry update)
#endif
        tj += 8;
        while (tj >= tw) { tj -= tw; ++ti; }
        while (ti >= th) { ti -= th; ++tb; }
    };
    f32x4 psc = {1.0f, 1.0f, 1.0f, 1.0f}, psh = {0.0f, 0.0f, 0.0f, 0.0f};
    if (PRO != 0) {
        psc = *reinterpret_cast<const f32x4*>(in_scale + cib * 64 + 4 * quad);
        psh = *reinterpret_cast<const f32x4*>(in_shift + cib * 64 + 4 * quad);
    }

    // ---- MFMA role
    const int hh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int kl = lane >> 5, ml = lane & 31;
    const int aoff = 8 * hh * FPL + kl * 64 + wm * 32 + ml;             // V fragment (A operand: rows = ci); + xi FPL + 128 e
    const int boff = 2 * FBUF + 8 * hh * FPL + kl * 64 + wn * 32 + ml;  // D fragment (B operand: columns = co); LDS = [V0 | V1 | D0 | D1]

    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    f32x4 xr[4], gr[2];
    float okn[4] = {0.0f, 0.0f, 0.0f, 0.0f}, okc[4];   // PRO != 0: upper clamp (+inf | 0) of the rows in flight / being transformed
    unsigned xcol = 0x80000000u, gcol = 0x80000000u;   // vector offsets of the chunk being loaded (out of range = column outside the image)
    float xclamp = 0.0f;
    auto col_offsets = [&]() {                         // once per chunk, before its loads
        const bool cx = (unsigned)(2 * tj - 1 + s) < (unsigned)W;
        xcol = cx ? xvo : 0x80000000u;
        gcol = (2 * tj + (s & 1)) < W ? dvo : 0x80000000u;
        if (PRO != 0) xclamp = cx ? INFINITY : 0.0f;
    };
    auto load_x = [&](int r) {                         // window row r of the tile (tb, ti, tj)
        const int h = 2 * ti - 1 + r;
        const bool rv = tb < B && (unsigned)h < (unsigned)H;
        // (32-bit scalar arithmetic: the tensor is < 2^31 bytes.)  A row outside the image is a WAVE-uniform condition: it selects an
        // EMPTY descriptor on the scalar unit (every load comes back as zeros); only the column test is a vector select, once per
        // chunk (xcol), not once per row
        const int so = (int)(((unsigned)(tb * H + h) * (unsigned)W + 2u * (unsigned)tj) * (unsigned)(Cin * 4));     // (unused when !rv)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) - Cin, 0, rv ? xbytes : 0, 0x00020000);
        xr[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xcol, so, 0));
        if (PRO != 0) okn[r] = rv ? xclamp : 0.0f;
    };
    auto load_g = [&](int a) {                         // gradient row 2 ti + a, column 2 tj + (s & 1)
        const int h = 2 * ti + a;
        const bool rv = tb < B && h < H;
        const int so = (int)(((unsigned)(tb * H + h) * (unsigned)W + 2u * (unsigned)tj) * (unsigned)(Cout * 4));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, rv ? dbytes : 0, 0x00020000);
        gr[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, gcol, so, 0));
    };
    f32x4 d[4], vv;
    float tt[4][4], R[4][4];
    auto x_row = [&](int r) {
        if (PRO == 0) {
            d[r] = xr[r];
        } else if (PRO == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[r][k] = __builtin_amdgcn_fmed3f(fmaf(xr[r][k], psc[k], psh[k]), 0.0f, okc[r]);
        } else {
            const f32x4 v = fused_prologue<PRO>(xr[r], psc, psh);
            d[r] = okc[r] > 0.0f ? v : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto x_col = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tt[0][k] = d[0][k] - d[2][k];
            tt[1][k] = d[1][k] + d[2][k];
            tt[2][k] = d[2][k] - d[1][k];
            tt[3][k] = d[1][k] - d[3][k];
        }
    };
    auto v_write = [&](float* Vb, int r) {
#if TAG_WG_ABL & 4
        vv = xr[r];                                    // ablation: no input transform
#else
        quad_fmac_2211(tt[r], fb, vv);
#endif
        *reinterpret_cast<f32x4*>(Vb + vw0 + 4 * r * FPL) = vv;
    };
    auto g_rows = [&]() {                              // R = A g (the thread's gradient column)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R[0][k] = gr[0][k];
            R[1][k] = gr[0][k] + gr[1][k];
            R[2][k] = gr[0][k] - gr[1][k];
            R[3][k] = gr[1][k];                        // (row 3 of D is stored NEGATED: no instruction here, a sign in the fold below)
        }
    };
    auto d_write = [&](float* Db, int r) {
#if TAG_WG_ABL & 2
        vv = gr[r & 1];                                // ablation: no gradient transform
#else
        quad_fmac_0033(R[r], cb, vv);
#endif
        *reinterpret_cast<f32x4*>(Db + vw0 + 4 * r * FPL) = vv;
    };
    auto rotate_ok = [&]() {
        if (PRO != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) okc[r] = okn[r];
        }
    };

    // ---- pipeline prologue: chunk 0 into buffer 0, chunk 1 into the registers (both requested before anything waits)
    {
        f32x4 x0[4], g0[2];
        float ok0[4];
        col_offsets();
#pragma unroll
        for (int r = 0; r < 4; ++r) load_x(r);
        load_g(0); load_g(1);
        advance();
#pragma unroll
        for (int r = 0; r < 4; ++r) { x0[r] = xr[r]; ok0[r] = okn[r]; }
        g0[0] = gr[0]; g0[1] = gr[1];
        col_offsets();
#pragma unroll
        for (int r = 0; r < 4; ++r) load_x(r);         // chunk 1 (stays in xr / gr / okn for the first iteration)
        load_g(0); load_g(1);
        advance();
        f32x4 x1[4], g1[2];
        float ok1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { x1[r] = xr[r]; ok1[r] = okn[r]; xr[r] = x0[r]; okc[r] = ok0[r]; }
        g1[0] = gr[0]; g1[1] = gr[1]; gr[0] = g0[0]; gr[1] = g0[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) x_row(r);
        x_col();
        g_rows();
#pragma unroll
        for (int r = 0; r < 4; ++r) { v_write(smem, r); d_write(smem + 2 * FBUF, r); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { xr[r] = x1[r]; okn[r] = ok1[r]; }
        gr[0] = g1[0]; gr[1] = g1[1];
    }
    __syncthreads();

    // Fragment addresses: the planes are 2080 B apart, i.e. 8 x 256 B + 32 B -- the 32 J bytes cannot ride in the offset field of
    // ds_read2st64_b32 (units of 256 B), and the compiler re-added them to the base for every plane of every chunk (10 vector adds
    // per wave and chunk).  The 16 plane bases are formed ONCE and made opaque, so that they stay in registers.
    int aJ[8], bJ[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        aJ[j] = aoff + j * FPL;
        bJ[j] = boff + j * FPL;
        asm volatile("" : "+v"(aJ[j]), "+v"(bJ[j]));
    }
#define WG_M4(J)                                                                                              \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                             \
        acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[aJ[J] + 128 * e], cur[bJ[J] + 128 * e], acc[J], 0, 0, 0)
#define WG_SB __builtin_amdgcn_sched_barrier(0)
    // (two copies of the chunk body, one per pipeline stage: stage offsets as LDS-instruction immediates, as in the forward kernel)
    auto chunk = [&](auto stage) {
        constexpr int ST = decltype(stage)::value;
        const float* cur = smem + ST * FBUF;
        float* nxt = smem + (1 - ST) * FBUF;
        WG_SB;
        WG_M4(0); WG_SB; rotate_ok(); x_row(0); x_row(1); WG_SB;
        WG_M4(1); WG_SB; x_row(2); x_row(3); x_col(); WG_SB;
        WG_M4(2); WG_SB; v_write(nxt, 0); v_write(nxt, 1); col_offsets(); load_x(0); load_x(1); WG_SB;
        WG_M4(3); WG_SB; v_write(nxt, 2); v_write(nxt, 3); load_x(2); load_x(3); WG_SB;
        WG_M4(4); WG_SB; g_rows(); d_write(nxt + 2 * FBUF, 0); d_write(nxt + 2 * FBUF, 1); WG_SB;
        WG_M4(5); WG_SB; d_write(nxt + 2 * FBUF, 2); d_write(nxt + 2 * FBUF, 3); load_g(0); load_g(1); advance(); WG_SB;
        WG_M4(6); WG_SB;
        WG_M4(7); WG_SB;
        __syncthreads();
    };
    {
        int c = 0;
        for (; c + 1 < cps; c += 2) {
            chunk(std::integral_constant<int, 0>{});
            chunk(std::integral_constant<int, 1>{});
        }
        if (c < cps) chunk(std::integral_constant<int, 0>{});
    }
#undef WG_M4
#undef WG_SB

    // ---- dw share of this wave: G^T (rows of its xi half) . G, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float* pb = part + ((size_t)(sl * 2 + hh) * Cout + cob * 64 + wn * 32 + ml) * Cin * 9 + (size_t)(cib * 64 + wm * 32 + 4 * kl) * 9;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        float o[4][9];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float P[3][4];
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const float m0 = acc[sx][4 * rq + k], m1 = acc[4 + sx][4 * rq + k];
                if (hh == 0) { P[0][sx] = m0 + 0.5f * m1; P[1][sx] = 0.5f * m1; P[2][sx] = 0.5f * m1; }      // transform rows r = 0, 1
                else { P[0][sx] = 0.5f * m0; P[1][sx] = -0.5f * m0; P[2][sx] = 0.5f * m0 - m1; }              // rows r = 2, 3 (m1 = -dU of row 3)
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float hs = 0.5f * (P[a][1] + P[a][2]), hd = 0.5f * (P[a][1] - P[a][2]);
                o[k][3 * a + 0] = P[a][0] + hs;
                o[k][3 * a + 1] = hd;
                o[k][3 * a + 2] = hs + P[a][3];
            }
        }
        float* q = pb + (size_t)(8 * rq) * 9;          // 4 consecutive ci x 9 taps = 36 floats
        const float* of = &o[0][0];
#pragma unroll
        for (int v4 = 0; v4 < 9; ++v4) *reinterpret_cast<f32x4*>(q + 4 * v4) = (f32x4){of[4 * v4], of[4 * v4 + 1], of[4 * v4 + 2], of[4 * v4 + 3]};
    }
}

// dw (Cout,Cin,3,3) = the 2 S shares added in a fixed order; 4 consecutive floats per thread group.  The small filters have the
// MOST shares (64 x 64: 512 shares of 9216 float4): there Q = 4 or 16 threads split the share axis of one float4 into contiguous
// runs (each run added in order, the Q run sums added in order through LDS) so that the launch still fills the chip -- a single
// thread per float4 left 36 workgroups walking 512 dependent loads each.
template <int Q>
__global__ __launch_bounds__(256) void wino_fused_wgrad_finish_kernel(const float* __restrict__ part, int nshare, long n4, float* __restrict__ dw) {
    constexpr int E = 256 / Q;
    __shared__ f32x4 run[Q > 1 ? 256 : 1];
    const int el = threadIdx.x % E, q = threadIdx.x / E;
    const int per = (nshare + Q - 1) / Q;
    const int p0 = q * per, p1 = (p0 + per < nshare) ? p0 + per : nshare;
    for (long e0 = (long)blockIdx.x * E; e0 < n4; e0 += (long)gridDim.x * E) {
        const long e = e0 + el;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        if (e < n4)
            for (int p = p0; p < p1; ++p) a += *reinterpret_cast<const f32x4*>(part + ((size_t)p * n4 + e) * 4);
        if (Q == 1) {
            if (e < n4) *reinterpret_cast<f32x4*>(dw + e * 4) = a;
        } else {
            run[threadIdx.x] = a;
            __syncthreads();
            if (q == 0 && e < n4) {
#pragma unroll
                for (int k = 1; k < Q; ++k) a += run[k * E + el];
                *reinterpret_cast<f32x4*>(dw + e * 4) = a;
            }
            __syncthreads();
        }
    }
}

}  // namespace

bool wino_fused_ok(int Cin, int Cout) { return Cin >= 8 && Cin % 8 == 0 && Cin <= 1024 && Cout >= 64 && Cout % 64 == 0; }

int wino_fused_rows(int B, int H, int W) {
    const long T = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
    return (int)((T + 63) / 64);
}

int wino_fused_run(const float* x, const float* U, int pro, const float* s, const float* t, float* y, float* stats, const WinoEpi* epi,
                   int B, int H, int W, int Cin, int Cout, hipStream_t st) {
    const WinoEpi none{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.0f, 0.0f, 0, 0, 0.0f, 0ull, 0};
    const int kind = epi ? epi->kind : 0;
    const WinoEpi& e = epi ? *epi : none;
#define FUSED_CASE(P_, E_) launch_fused<P_, E_>(x, U, s, t, y, stats, e, B, H, W, Cin, Cout, st); return 0
    if (kind == 0) {
        switch (pro) { case 0: FUSED_CASE(0, 0); case 1: FUSED_CASE(1, 0); case 2: FUSED_CASE(2, 0); default: FUSED_CASE(3, 0); }
    }
    if (kind == 3) {
        switch (pro) { case 0: FUSED_CASE(0, 3); case 1: FUSED_CASE(1, 3); case 2: FUSED_CASE(2, 3); default: FUSED_CASE(3, 3); }
    }
    if (pro != 0) return -1;                      // the gradient launches have no producer prologue
    if (kind == 1) { FUSED_CASE(0, 1); }
    FUSED_CASE(0, 2);
#undef FUSED_CASE
}

bool wino_fused_wgrad_ok(int Cin, int Cout) { return Cin >= 64 && Cin % 64 == 0 && Cout >= 64 && Cout % 64 == 0; }

size_t wino_fused_wgrad_ws_floats(int B, int H, int W, int Cin, int Cout) {
    const WgGeom g = wg_geom(B, H, W, Cin, Cout);
    return (size_t)2 * g.S * Cin * Cout * 9;
}

int wino_fused_wgrad_run(const float* x, int pro, const float* s, const float* t, const float* dy, float* dw, int B, int H, int W,
                         int Cin, int Cout, float* ws, hipStream_t st) {
    const WgGeom g = wg_geom(B, H, W, Cin, Cout);
    const int grid = g.nci * g.nco * g.S;
    const size_t lds = (size_t)4 * FBUF * sizeof(float);
#define WG_CASE(P_)                                                                                                          \
    {                                                                                                                        \
        static bool attr_set = false;                                                                                        \
        if (!attr_set) {                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fused_wgrad_kernel<P_>),                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
            attr_set = true;                                                                                                 \
        }                                                                                                                    \
        hipLaunchKernelGGL((wino_fused_wgrad_kernel<P_>), dim3(grid), dim3(512), lds, st, x, s, t, dy, ws, B, H, W, Cin, Cout, g.th,  \
                           g.tw, g.nci, g.nco, g.S, g.cps);                                                                  \
    }
    switch (pro) {
        case 0: WG_CASE(0) break;
        case 1: WG_CASE(1) break;
        case 2: WG_CASE(2) break;
        default: WG_CASE(3) break;
    }
#undef WG_CASE
    const long n4 = (long)Cin * Cout * 9 / 4;
    const auto blocks = [&](int e) { const long b = cdiv(n4, (long)e); return dim3((unsigned)(b > 4096 ? 4096 : b)); };
    if (n4 >= 131072 || g.S < 8) hipLaunchKernelGGL(wino_fused_wgrad_finish_kernel<1>, blocks(256), dim3(256), 0, st, ws, 2 * g.S, n4, dw);
    else if (n4 >= 32768 || g.S < 32) hipLaunchKernelGGL(wino_fused_wgrad_finish_kernel<4>, blocks(64), dim3(256), 0, st, ws, 2 * g.S, n4, dw);
    else hipLaunchKernelGGL(wino_fused_wgrad_finish_kernel<16>, blocks(16), dim3(256), 0, st, ws, 2 * g.S, n4, dw);
    return 0;
}
