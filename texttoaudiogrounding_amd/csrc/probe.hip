// Measurement aid of bench.py (not on the data path): the rate the matrix pipe SUSTAINS on this part, on this box, right now.
// A whole chip of register-resident MFMA work (no LDS, no memory) with either constant operands (the datapath barely toggles:
// the datasheet rate) or eight operand pairs of random bits rotating from MFMA to MFMA as a real kernel's do (the part runs
// into its power limit and the shader clock drops).  bench.py launches it for ~50 ms beside each timed leg and writes the rate
// into the JSON line as `measured_ceiling`, so that a conv family's fraction can be read against what THIS box could deliver
// at all, and box-to-box differences of the power-limited legs explain themselves in the driver's record.
// Workgroup 0 also reports the shader clock it ran at: s_memtime (shader clocks) over s_memrealtime (100 MHz).
#include "tag_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix32(unsigned z) {
    z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
    return z;
}

constexpr int NACC = 4;        // independent accumulators per wave (4 waves per workgroup, one workgroup per CU)

// KIND 0: bf16 32x32x16, random operands | 1: bf16, constant operands | 2: f32 32x32x2, random | 3: f32, constant
template <int KIND>
__global__ __launch_bounds__(256) void mfma_probe_kernel(unsigned long long* __restrict__ clocks, int iters, unsigned seed) {
    constexpr bool BF = KIND < 2, RANDOM = (KIND & 1) == 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    u32x4 av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // random sign + mantissa, exponents 0x3c..0x3f (|v| in [2^-7, 2)); constant: the same pattern in every lane and pair
            const unsigned ra = RANDOM ? mix32(seed + (blockIdx.x * 256 + threadIdx.x) * 64 + u * 8 + k) : 0x12345678u;
            const unsigned rb = mix32(ra + 0x9e3779b9u);
            if (BF) {
                av[u][k] = (ra & 0x81ff81ffu) | 0x3c003c00u | ((ra >> 3) & 0x01800180u);
                bv[u][k] = (rb & 0x81ff81ffu) | 0x3c003c00u | ((rb >> 3) & 0x01800180u);
            } else {
                av[u][k] = (ra & 0x81ffffffu) | 0x3c000000u | ((ra >> 3) & 0x01800000u);
                bv[u][k] = (rb & 0x81ffffffu) | 0x3c000000u | ((rb >> 3) & 0x01800000u);
            }
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (BF)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[(u + i) & 7]),
                                                                     __builtin_bit_cast(bf16x8, bv[(u + 3 * i) & 7]), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[(u + i) & 7][0]),
                                                                  __uint_as_float(bv[(u + 3 * i) & 7][0]), acc[i], 0, 0, 0);
            }
        if ((it & 255) == 255) {                     // keep the sums bounded: the accumulators decay
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) clocks[2] = (unsigned long long)s;      // keeps the accumulators alive
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = __builtin_amdgcn_s_memtime() - t0;
        clocks[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// Packed-fp32 VALU loop (v_pk_fma_f32 on 16 independent float2 accumulators, registers only), 4 waves per workgroup: the partner of
// tools/hybrid_probe.py, which runs it on a second stream BESIDE a real MFMA convolution to see what the two pipes deliver together.
__global__ __launch_bounds__(256) void valu_probe_kernel(unsigned long long* __restrict__ clocks, int iters, unsigned seed) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x2 acc[16], a[4], b[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x2){0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned ra = mix32(seed + threadIdx.x * 16 + u), rb = mix32(ra + 0x9e3779b9u);
        a[u] = (f32x2){__uint_as_float((ra & 0x81ffffffu) | 0x3c000000u), __uint_as_float((rb & 0x81ffffffu) | 0x3c000000u)};
        b[u] = (f32x2){__uint_as_float((rb & 0x80ffffffu) | 0x3c000000u), __uint_as_float((ra & 0x80ffffffu) | 0x3c000000u)};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[u & 15]) : "v"(a[u & 3]), "v"(b[(u >> 2) & 3]));
        if ((it & 63) == 63) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= 0.001f;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    if (s == 12345.678f) clocks[2] = (unsigned long long)s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = __builtin_amdgcn_s_memtime() - t0;
        clocks[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

}  // namespace

// kind 4 of the probe family: the VALU loop above; FLOP of a launch = workgroups * 4 waves * iters * 32 * 256
extern "C" int tag_valu_probe(int iters, int workgroups, unsigned seed, void* clocks, void* stream) {
    TAG_CHECK_ARG(iters > 0 && workgroups > 0 && clocks);
    hipLaunchKernelGGL(valu_probe_kernel, dim3(workgroups), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<unsigned long long*>(clocks), iters, seed);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" double tag_mfma_probe_flop(int kind, int iters, int workgroups) {
    const double per_mfma = kind < 2 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    return (double)workgroups * 4.0 * (double)iters * 8.0 * NACC * per_mfma;
}

extern "C" int tag_mfma_probe(int kind, int iters, int workgroups, unsigned seed, void* clocks, void* stream) {
    TAG_CHECK_ARG(kind >= 0 && kind < 4 && iters > 0 && workgroups > 0 && clocks);
    hipStream_t st = as_stream(stream);
    unsigned long long* c = reinterpret_cast<unsigned long long*>(clocks);
    switch (kind) {
        case 0: hipLaunchKernelGGL((mfma_probe_kernel<0>), dim3(workgroups), dim3(256), 0, st, c, iters, seed); break;
        case 1: hipLaunchKernelGGL((mfma_probe_kernel<1>), dim3(workgroups), dim3(256), 0, st, c, iters, seed); break;
        case 2: hipLaunchKernelGGL((mfma_probe_kernel<2>), dim3(workgroups), dim3(256), 0, st, c, iters, seed); break;
        default: hipLaunchKernelGGL((mfma_probe_kernel<3>), dim3(workgroups), dim3(256), 0, st, c, iters, seed); break;
    }
    TAG_LAUNCH_CHECK();
    return 0;
}
