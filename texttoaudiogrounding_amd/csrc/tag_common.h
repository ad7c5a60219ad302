// Shared helpers for the gfx950 kernels of libtag_hip.so.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tag_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void tag_set_error(const char* fmt, ...);
// developer switch (tag_lib.hip; set with tag_set_option before the first launch that reads it)
int tag_option(const char* name);

#define TAG_CHECK_ARG(cond)                                                              \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            tag_set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond); \
            return TAG_EINVAL;                                                           \
        }                                                                                \
    } while (0)

#define TAG_LAUNCH_CHECK()                                                                       \
    do {                                                                                         \
        hipError_t e__ = hipGetLastError();                                                      \
        if (e__ != hipSuccess) {                                                                 \
            tag_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                                                     \
        }                                                                                        \
    } while (0)

// conv.hip internals shared with conv_x3.hip (not part of the C ABI)
int tag_wgrad_alltaps_splits(int B, int H, int W, int Cin, int Cout, int* chunks_per_split, int chunk_px = 32);
int tag_launch_wgrad_reduce(const float* partial, int splits, int Cin, int Cout, float* dw, hipStream_t st);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- counter-based keep-mask for dropout: splitmix64 of (seed, index) -> uniform in [0,1) ----
__host__ __device__ __forceinline__ uint64_t tag_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// keep-probability test on a 24-bit uniform; identical in every kernel that needs the same mask
__host__ __device__ __forceinline__ bool tag_keep(uint64_t seed, uint64_t idx, float p) {
    uint64_t h = tag_mix64(seed * 0xD1342543DE82EF95ull + idx);
    float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    return u >= p;
}

// Dropout of the POOLED activations (bn_pool.hip: one mask for the forward, the backward sums and the backward apply):
// ONE splitmix64 per 4 consecutive elements and a 16-bit uniform each -- element i is kept iff
//   ((mix64(seed * 0xD1342543DE82EF95 + (i >> 2)) >> 16 (i & 3)) & 0xFFFF) >= ceil(p * 2^16)        (u / 2^16 >= p).
// One hash per element (two 64-bit multiplies of quarter-rate integer ops each) was a quarter of the VALU time of the bf16
// pool passes, which are VALU-bound.  oracle/tag_oracle.py dropout_keep_mask4 restates it.
__host__ __device__ __forceinline__ uint64_t tag_keep4_bits(uint64_t seed, uint64_t group) {
    return tag_mix64(seed * 0xD1342543DE82EF95ull + group);
}
__host__ __device__ __forceinline__ unsigned tag_keep4_threshold(float p) { return (unsigned)ceilf(p * 65536.0f); }
__host__ __device__ __forceinline__ bool tag_keep4(uint64_t bits, int j, unsigned thr) {
    return (unsigned)((bits >> (16 * j)) & 0xFFFFull) >= thr;
}

// ---- wave / block reductions (wave = 64) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; give each XCD a contiguous
// chunk of the logical tile order so neighbouring tiles share an L2 (speed only, bijective).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, i = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

// ---- activation storage types: fp32 (exact path) or bf16 (BASELINE configs[2]: bf16 tensors, fp32 arithmetic) ----
// bf16 values travel as raw 16-bit patterns; conversion to fp32 is a shift, rounding to bf16 is round-to-nearest-even.
typedef unsigned short bf16_t;
typedef unsigned tag_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned tag_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned tag_bf16_rne(float f) {           // -> bits in the LOW half
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two floats -> packed bf16 pair (low half = first value), round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 (the integer
// form above costs ~10 VALU operations per pair, which the bf16 epilogues / prologues pay 30-60 times per lane and tile)
typedef __bf16 tag_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tag_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned tag_pack_bf16(float lo, float hi) {
    const tag_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, tag_bf16x2));
}
__device__ __forceinline__ float tag_bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float tag_bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

template <class T> struct Act;
template <> struct Act<float> {
    static constexpr bool is_bf16 = false;
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Act<bf16_t> {
    static constexpr bool is_bf16 = true;
    static __device__ __forceinline__ f32x4 ld4(const bf16_t* p) {       // 4 channels = one 8-byte load
        const tag_u32x2 w = *reinterpret_cast<const tag_u32x2*>(p);
        return (f32x4){tag_bf16_lo(w.x), tag_bf16_hi(w.x), tag_bf16_lo(w.y), tag_bf16_hi(w.y)};
    }
    static __device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
        *reinterpret_cast<tag_u32x2*>(p) = (tag_u32x2){tag_pack_bf16(v.x, v.y), tag_pack_bf16(v.z, v.w)};
    }
    static __device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((unsigned)*p << 16); }
    static __device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)tag_bf16_rne(v); }
};

// NC consecutive channels of one pixel <-> NC floats: 16-byte accesses for fp32 (NC = 4) and for bf16 with NC = 8 (the bf16
// BatchNorm / pool passes take 8 channels per thread: an 8-byte access moves data at 0.55-0.7 of the 16-byte rate)
template <class TS, int NC> struct ActN;
// raw_t / ldraw / unpack: the load on its own (issued one slot ahead by the software-pipelined pool passes) and the conversion
template <> struct ActN<float, 4> {
    typedef f32x4 raw_t;
    static __device__ __forceinline__ raw_t ldraw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ raw_t zero() { return (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
    static __device__ __forceinline__ void unpack(raw_t q, float (&v)[4]) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
    }
};
template <> struct ActN<bf16_t, 4> {
    typedef tag_u32x2 raw_t;
    static __device__ __forceinline__ raw_t ldraw(const bf16_t* p) { return *reinterpret_cast<const tag_u32x2*>(p); }
    static __device__ __forceinline__ raw_t zero() { return (tag_u32x2){0u, 0u}; }
    static __device__ __forceinline__ void unpack(raw_t w, float (&v)[4]) {
        v[0] = tag_bf16_lo(w.x); v[1] = tag_bf16_hi(w.x); v[2] = tag_bf16_lo(w.y); v[3] = tag_bf16_hi(w.y);
    }
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[4]) {
        const tag_u32x2 w = *reinterpret_cast<const tag_u32x2*>(p);
        v[0] = tag_bf16_lo(w.x); v[1] = tag_bf16_hi(w.x); v[2] = tag_bf16_lo(w.y); v[3] = tag_bf16_hi(w.y);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[4]) {
        *reinterpret_cast<tag_u32x2*>(p) = (tag_u32x2){tag_pack_bf16(v[0], v[1]), tag_pack_bf16(v[2], v[3])};
    }
};
template <> struct ActN<bf16_t, 8> {
    typedef tag_u32x4 raw_t;
    static __device__ __forceinline__ raw_t ldraw(const bf16_t* p) { return *reinterpret_cast<const tag_u32x4*>(p); }
    static __device__ __forceinline__ raw_t zero() { return (tag_u32x4){0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ void unpack(raw_t w, float (&v)[8]) {
        v[0] = tag_bf16_lo(w.x); v[1] = tag_bf16_hi(w.x); v[2] = tag_bf16_lo(w.y); v[3] = tag_bf16_hi(w.y);
        v[4] = tag_bf16_lo(w.z); v[5] = tag_bf16_hi(w.z); v[6] = tag_bf16_lo(w.w); v[7] = tag_bf16_hi(w.w);
    }
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
        const tag_u32x4 w = *reinterpret_cast<const tag_u32x4*>(p);
        v[0] = tag_bf16_lo(w.x); v[1] = tag_bf16_hi(w.x); v[2] = tag_bf16_lo(w.y); v[3] = tag_bf16_hi(w.y);
        v[4] = tag_bf16_lo(w.z); v[5] = tag_bf16_hi(w.z); v[6] = tag_bf16_lo(w.w); v[7] = tag_bf16_hi(w.w);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
        *reinterpret_cast<tag_u32x4*>(p) = (tag_u32x4){tag_pack_bf16(v[0], v[1]), tag_pack_bf16(v[2], v[3]),
                                                       tag_pack_bf16(v[4], v[5]), tag_pack_bf16(v[6], v[7])};
    }
};

