// Shared helpers for the gfx950 kernels of libtag_hip.so.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tag_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void tag_set_error(const char* fmt, ...);

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
int tag_wgrad_alltaps_splits(int B, int H, int W, int Cin, int Cout, int* chunks_per_split);
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
