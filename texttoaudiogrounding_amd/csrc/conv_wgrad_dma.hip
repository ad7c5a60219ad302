// A1 backward, bf16 mode (BASELINE configs[2]): the weight gradient of the 3x3 convolution (models/panns.py:25-33,49-50),
//     partial[split][tap][ci][co] = sum over the pixels p of the split of  prologue(X)[p + tap][ci] * dY[p][co],
// with both operands brought in by LDS-DMA.  Same decomposition, same LDS images, same MFMA phase (transposing reads
// ds_read_b64_tr_b16, 9 taps x 16 accumulator registers per wave) and the same deterministic split reduction as
// conv3x3_wgrad_x3_kernel<.., bf16_t> of conv_x3.hip -- what changes is how a 64-pixel chunk gets into LDS.  There every
// thread loaded its (pixel, channel octet) items into registers, applied the producer BatchNorm + ReLU and stored them with
// ds_write_b128: per chunk ~600 VALU / address instructions against 36 MFMAs per wave, and the phase clocks read MFMA 37-40 %,
// staging 43-46 %, waiting 8 % (MFMA busy 0.43-0.45, 0.36 of the bf16 peak on every layer).  Here
//   * the CH new input rows of a chunk (one contiguous block of ring slots per 32-channel half) and its dY tile arrive by
//     global_load_lds_dwordx4, 16 pixels x 64 B per wave instruction, TWO chunks ahead of the MFMAs that read them; image
//     borders are lanes that fetch from a 16-byte zero page instead (no masks, no branches);
//   * the producer BatchNorm + ReLU, where the layer has one, is applied in place in LDS by the wave that loaded the piece
//     (border pixels keep their zeros);
//   * every vector-memory instruction of the loop is inline asm and counted by hand (one in-order counter per wave on gfx9:
//     `s_waitcnt vmcnt(5)` = "everything but the newest chunk has landed"), one barrier per chunk.
// A strip change (the split walks from the bottom of one column strip to the top of the next) drains the pipeline and primes
// the ring again: rare (one in 32-500 chunks).
#include <stdlib.h>
#include "tag_common.h"

// tools/run_wdma_prof.sh: s_memtime deltas of the phases of ONE workgroup's wave 0
#ifdef TAG_WDMA_PROF
__device__ unsigned long long tag_wdma_prof[12];
extern "C" int tag_debug_get_wdma_prof(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(tag_wdma_prof), 96) == hipSuccess ? 0 : -1; }
#define WP_MARK(i) { const unsigned long long p1_ = __builtin_amdgcn_s_memtime(); wpc[i] += p1_ - wp0; wp0 = p1_; }
#else
#define WP_MARK(i)
#endif

__device__ const unsigned tag_zero_page[64] = {0};          // source of every border lane

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2 lds_tr_read(const unsigned char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(u32x2, v);
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void lds_fence_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int TW>
struct WDGeom {
    static constexpr int KS = 4, CPX = 16 * KS;                                 // k16 steps / pixels per chunk
    static constexpr int CW = TW >= 32 ? 32 : TW, CH = CPX / CW, PW = CW + 2;
    static constexpr int R = 4 * CH;                                            // ring rows: chunk c reads CH + 2, two more chunks land
    static constexpr int XROWB = PW * 64;                                       // bytes of one ring row in one 32-channel half
    static constexpr int XPL = R * XROWB + ((R * XROWB) % 256 == 0 ? 128 : 0);  // half stride = 128 (mod 256)
    static constexpr int YPL = CPX * 64;                                        // one 32-cout half of a dY tile
    static constexpr int NPX = (CH * PW + 15) / 16;                             // DMA pieces of a row block per half
    static constexpr int NPIECE = 2 * NPX + 2 * (CPX / 16);                     // per chunk: X halves + dY halves
    static constexpr int PER_WAVE = (NPIECE + 3) / 4;                           // every wave issues this many (the surplus: dummies)
    static constexpr int OFF_Y = 2 * XPL, YBYTES = 2 * YPL, NYB = 3;
    static constexpr int OFF_DUMMY = OFF_Y + NYB * YBYTES;
    static constexpr int OFF_SS = OFF_DUMMY + 1024;                             // [2][64] floats
    static constexpr int LDS_BYTES = OFF_SS + 2 * 64 * 4;
};

template <int TW, int PRO>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_dma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ in_scale,
                                                                   const float* __restrict__ in_shift,
                                                                   const bf16_t* __restrict__ dy, float* __restrict__ partial,
                                                                   int B, int H, int W, int Cin, int Cout, int splits,
                                                                   int chunks_per_split) {
    using G = WDGeom<TW>;
    constexpr int KS = G::KS, CW = G::CW, CH = G::CH, PW = G::PW, R = G::R, NPX = G::NPX, NPD = G::CPX / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xs = smem;
    unsigned char* Ys = smem + G::OFF_Y;
    float* Ss = reinterpret_cast<float*>(smem + G::OFF_SS);
    const unsigned lds0 = (unsigned)(size_t)smem;

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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wid >> 1, wj = wid & 1;                       // ci / co 32-block of the wave
    const int kl = lane >> 5, half = (lane >> 4) & 1, li = lane & 15;

    if (PRO != 0)
        for (int c = tid; c < 64; c += 256) { Ss[c] = in_scale[ci0 + c]; Ss[64 + c] = in_shift[ci0 + c]; }

    // ---- DMA pieces of a chunk.  Piece e < 2 NPX: half b = e / NPX of the row block, pixels 16 q .. 16 q + 15 of the block
    // (pixel t -> block row t / PW, column t % PW); 2 NPX <= e < NPIECE: half b of the dY tile, pixels 16 q ..; e >= NPIECE: dummy.
    const int lpix = lane >> 2, lch = lane & 3;
    const char* zero = reinterpret_cast<const char*>(tag_zero_page);
    auto issue_x_piece = [&](int img, int hblk, int w0, int b, int q) {         // block = input rows hblk + 1 .. hblk + CH
        const int t = q * 16 + lpix;
        const int r = t / PW, p = t - r * PW;
        const int row = hblk + 1 + r, col = w0 - 1 + p;
        const bool ok = ((unsigned)row < (unsigned)H) & ((unsigned)col < (unsigned)W);
        const char* src = ok ? reinterpret_cast<const char*>(x) + ((((size_t)img * H + row) * W + col) * Cin + ci0 + b * 32) * 2 + lch * 16
                             : zero + lch * 16;
        const int slot0 = ((hblk % R) + R) % R;                                 // slot of row r = (r - 1) mod R; hblk is a multiple of CH
        const unsigned dst = lds0 + (unsigned)(b * G::XPL + slot0 * G::XROWB + q * 1024);
        if (t < CH * PW) glds16(src, __builtin_amdgcn_readfirstlane(dst));
    };
    auto transform_x_piece = [&](int img, int hblk, int w0, int b, int q) {
        const int t = q * 16 + lpix;
        const int r = t / PW, p = t - r * PW;
        const int row = hblk + 1 + r, col = w0 - 1 + p;
        const bool ok = ((unsigned)row < (unsigned)H) & ((unsigned)col < (unsigned)W) & (t < CH * PW);
        const int slot0 = ((hblk % R) + R) % R;
        unsigned char* pp = Xs + b * G::XPL + slot0 * G::XROWB + q * 1024 + lane * 16;
        if (ok) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(pp);
            const int c0 = b * 32 + lch * 8;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(Ss + c0), s1 = *reinterpret_cast<const f32x4*>(Ss + c0 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(Ss + 64 + c0), t1 = *reinterpret_cast<const f32x4*>(Ss + 64 + c0 + 4);
            u32x4 v;
            v.x = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.x), s0.x, t0.x), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.x), s0.y, t0.y), 0.0f));
            v.y = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.y), s0.z, t0.z), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.y), s0.w, t0.w), 0.0f));
            v.z = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.z), s1.x, t1.x), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.z), s1.y, t1.y), 0.0f));
            v.w = tag_pack_bf16(fmaxf(fmaf(tag_bf16_lo(raw.w), s1.z, t1.z), 0.0f), fmaxf(fmaf(tag_bf16_hi(raw.w), s1.w, t1.w), 0.0f));
            *reinterpret_cast<u32x4*>(pp) = v;
        }
    };
    auto issue_y_piece = [&](int img, int h0, int w0, int b, int q, int buf) {
        const int k = q * 16 + lpix;                                            // pixel of the chunk: row k / CW, column k % CW
        const int row = h0 + k / CW, col = w0 + k % CW;
        const char* src = row < H ? reinterpret_cast<const char*>(dy) + ((((size_t)img * H + row) * W + col) * Cout + co0 + b * 32) * 2 + lch * 16
                                  : zero + lch * 16;
        const unsigned dst = lds0 + (unsigned)(G::OFF_Y + buf * G::YBYTES + b * G::YPL + q * 1024);
        glds16(src, __builtin_amdgcn_readfirstlane(dst));
    };
    // every wave issues PER_WAVE pieces per chunk (in-order vmcnt: the count must not depend on the wave)
    auto issue_chunk = [&](int img, int h0, int w0, int buf) {
#pragma unroll
        for (int jj = 0; jj < G::PER_WAVE; ++jj) {
            const int e = wid + 4 * jj;
            if (e < 2 * NPX) issue_x_piece(img, h0, w0, e / NPX, e % NPX);
            else if (e < G::NPIECE) issue_y_piece(img, h0, w0, (e - 2 * NPX) / NPD, (e - 2 * NPX) % NPD, buf);
            else glds16(zero + lch * 16, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)G::OFF_DUMMY));
        }
    };
    auto issue_prime = [&](int img, int h0, int w0) {            // rows h0 - CH + 1 .. h0 of a strip's first chunk (h0 = 0: row 0 + zeros)
#pragma unroll
        for (int jj = 0; jj < (2 * NPX + 3) / 4; ++jj) {
            const int e = wid + 4 * jj;
            if (e < 2 * NPX) issue_x_piece(img, h0 - CH, w0, e / NPX, e % NPX);
        }
    };
    auto transform_chunk = [&](int img, int h0, int w0) {
        if (PRO != 0) {
#pragma unroll
            for (int jj = 0; jj < G::PER_WAVE; ++jj) {
                const int e = wid + 4 * jj;
                if (e < 2 * NPX) transform_x_piece(img, h0, w0, e / NPX, e % NPX);
            }
        }
    };
    auto transform_prime = [&](int img, int h0, int w0) {
        if (PRO != 0) {
#pragma unroll
            for (int jj = 0; jj < (2 * NPX + 3) / 4; ++jj) {
                const int e = wid + 4 * jj;
                if (e < 2 * NPX) transform_x_piece(img, h0 - CH, w0, e / NPX, e % NPX);
            }
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

    // MFMA phase of one chunk: 3 * KS plane-steps (k16 step s, tap row ky), the three kx fragments of the next step are read
    // while this step's MFMAs run; the dY fragment of a k16 step feeds all nine taps
    auto mma_chunk = [&](int h0, int buf) {
        const int slot0 = (h0 - 2 + 4 * R) % R;                  // slot of input row h0 - 1
        unsigned rowb[CH + 2];
#pragma unroll
        for (int j = 0; j < CH + 2; ++j) {
            int sl = slot0 + j;
            sl = sl >= R ? sl - R : sl;
            rowb[j] = (unsigned)(sl * G::XROWB);
        }
        const unsigned char* yb = Ys + buf * G::YBYTES + vb;
        constexpr int NQ = 3 * KS;
        auto load_a = [&](int q, u32x4 (&af)[3]) {
            const int s = q / 3, ky = q % 3;
            unsigned arow;                                        // k16 step s = pixels 16 s .. 16 s + 15 of the chunk
            if (CW == 8) arow = kl ? rowb[2 * s + 1 + ky] : rowb[2 * s + ky];
            else if (CW == 16) arow = rowb[s + ky];
            else arow = rowb[(s >> 1) + ky];
            const unsigned char* xa = Xs + va + arow;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int imm = (kx + (CW == 32 ? (s & 1) * 16 : 0)) * 64;
                const u32x2 t0 = lds_tr_read(xa + imm);
                const u32x2 t1 = lds_tr_read(xa + imm + 256);
                af[kx] = (u32x4){t0.x, t0.y, t1.x, t1.y};
            }
        };
        u32x4 bf[2];
        auto load_b = [&](int s, u32x4& b) {
            const u32x2 t0 = lds_tr_read(yb + s * 1024);
            const u32x2 t1 = lds_tr_read(yb + s * 1024 + 256);
            b = (u32x4){t0.x, t0.y, t1.x, t1.y};
        };
        load_b(0, bf[0]);
        u32x4 afb[2][3];
        load_a(0, afb[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int s = q / 3, ky = q % 3;
            if (q % 3 == 0 && s + 1 < KS) load_b(s + 1, bf[(s + 1) & 1]);
            if (q + 1 < NQ) load_a(q + 1, afb[(q + 1) & 1]);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                acc[ky * 3 + kx] = mfma_bf16(afb[q & 1][kx], bf[s & 1], acc[ky * 3 + kx]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // chunk index -> (img, h0, w0): row blocks run fastest, so consecutive chunks walk down a column strip
    auto origin_div = [&](int c, int& img, int& h0, int& w0) {
        const int rbk = c % rb_per_img; const int t = c / rb_per_img;
        const int cbk = t % cb_per_row; img = t / cb_per_row;
        h0 = rbk * CH; w0 = cbk * CW;
    };
#ifdef TAG_WDMA_PROF   // 0 segment start (prime, drain), 1 DMA issue, 2 MFMA chunk, 3 wait for the DMA, 4 transform, 5 barrier, 6 epilogue
    unsigned long long wpc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wp0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int KW = G::PER_WAVE;
    int c = cbeg, nb = 0;                                        // nb: running chunk counter -> dY buffer nb % 3
    while (c < cend) {
        // ---- a segment: chunks c .. cs - 1 walk down one column strip ----
        int img, h0, w0;
        origin_div(c, img, h0, w0);
        int cs = c + (rb_per_img - h0 / CH);                     // first chunk of the next strip
        if (cs > cend) cs = cend;
        lds_fence_barrier();                                     // Ss visible / the previous segment's reads are done
        issue_prime(img, h0, w0);
        issue_chunk(img, h0, w0, nb % 3);
        const bool two = c + 1 < cs;
        if (two) issue_chunk(img, h0 + CH, w0, (nb + 1) % 3);
        if (two) wait_vmcnt<KW>(); else wait_vmcnt<0>();
        transform_prime(img, h0, w0);
        transform_chunk(img, h0, w0);
        lds_fence_barrier();
        WP_MARK(0)
        for (int cc = c; cc < cs; ++cc, ++nb, h0 += CH) {
            const bool n2 = cc + 2 < cs, n1 = cc + 1 < cs;
            if (n2) issue_chunk(img, h0 + 2 * CH, w0, (nb + 2) % 3);     // its slots were last read by chunk cc - 1
            WP_MARK(1)
            mma_chunk(h0, nb % 3);
            WP_MARK(2)
            if (n1) {
                if (n2) wait_vmcnt<KW>(); else wait_vmcnt<0>();          // chunk cc + 1 has landed
                WP_MARK(3)
                transform_chunk(img, h0 + CH, w0);
                WP_MARK(4)
            }
            lds_fence_barrier();
            WP_MARK(5)
        }
        c = cs;
    }
    wait_vmcnt<0>();
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
#ifdef TAG_WDMA_PROF
    WP_MARK(6)
    if (blockIdx.x == 300 && tid == 0) { for (int i = 0; i < 8; ++i) tag_wdma_prof[i] = wpc[i]; tag_wdma_prof[8] = (unsigned long long)(cend - cbeg); }
#endif
}

template <int TW>
void launch_wgrad_dma(const bf16_t* x, int pro, const float* s, const float* t, const bf16_t* dy, float* partial, int B, int H,
                      int W, int Cin, int Cout, int splits, int cps, hipStream_t st) {
    using G = WDGeom<TW>;
    const int grid = (Cin / 64) * (Cout / 64) * splits;
    const int lds = G::LDS_BYTES;
#define LAUNCH_PRO(P)                                                                                                \
    {                                                                                                                \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_dma_kernel<TW, P>),               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);                              \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((conv3x3_wgrad_dma_kernel<TW, P>), dim3(grid), dim3(256), lds, st, x, s, t, dy, partial,  \
                           B, H, W, Cin, Cout, splits, cps);                                                         \
    }
    if (pro == 0) LAUNCH_PRO(0) else LAUNCH_PRO(1)
#undef LAUNCH_PRO
}

}  // namespace

// Measured at B = 64 (tools/conv_wgrad_bench.py, interleaved rounds): the layers WITHOUT a producer prologue (conv1 of blocks
// 2-4: the input is the pooled activation) are 5-15 % faster here (190 -> 165, 158 -> 148, 310 -> 295 us); the layers WITH the
// BatchNorm + ReLU prologue are 3-5 % slower (the in-LDS transform costs an LDS round trip the register-staged kernel does not
// make) -- both kernels sit at 0.92-1.02 PFLOP/s, which is where the part's power limit puts an LDS-fed bf16 MFMA loop on random
// operands (tools/mfma_peak.hip: 1.88 PFLOP/s with NOTHING but MFMAs).  Default: prologue 0 here, prologue 1 on the staged kernel.
// TAG_WGRAD_DMA=0 / tag_wgrad_dma_enable(0): never; =2 / enable(2): also the prologue-1 layers.
static int g_wdma_on = -1;
bool tag_wgrad_dma_takes(int prologue) {
    if (g_wdma_on < 0) g_wdma_on = tag_option("wgrad_dma");
    return (g_wdma_on == 1 && prologue == 0) || (g_wdma_on >= 2 && prologue <= 1);
}
extern "C" int tag_wgrad_dma_enable(int on) {
    (void)tag_wgrad_dma_takes(0);
    const int was = g_wdma_on;
    g_wdma_on = on;
    return was;
}

int tag_wgrad_dma_launch(const bf16_t* x, int prologue, const float* in_scale, const float* in_shift, const bf16_t* dy,
                         float* partial, int B, int H, int W, int Cin, int Cout, int splits, int cps, hipStream_t st) {
    if (W == 8) launch_wgrad_dma<8>(x, prologue, in_scale, in_shift, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
    else if (W == 16) launch_wgrad_dma<16>(x, prologue, in_scale, in_shift, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
    else if (W == 32) launch_wgrad_dma<32>(x, prologue, in_scale, in_shift, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
    else if (W == 64) launch_wgrad_dma<64>(x, prologue, in_scale, in_shift, dy, partial, B, H, W, Cin, Cout, splits, cps, st);
    else return TAG_EINVAL;
    return 0;
}
