// Sustained MFMA rate of register-resident operands (no LDS, no memory): what the matrix pipe delivers on this part once
// a whole chip of it runs for milliseconds -- the practical ceiling beside the datasheet peaks bench.py prices against.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/bin/mfma_peak tools/mfma_peak.hip && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BF>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
    bf16x8 ab, bb;
#pragma unroll
    for (int k = 0; k < 8; ++k) { ab[k] = (__bf16)(a + k); bb[k] = (__bf16)(b - k); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;              // keep the accumulators alive
}

// The same loop with EIGHT different operand pairs of random bits (normal exponents) rotating from MFMA to MFMA, as the
// operands of a real kernel do: the datapath toggles fully and the part runs into its POWER limit -- the sustained rate of
// random bf16 operands, with nothing else running (no LDS, no HBM), is the practical ceiling of every conv / GEMM kernel here.
__device__ __forceinline__ unsigned mix32(unsigned z) {
    z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
    return z;
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_random(float* out, int iters, unsigned seed) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
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
            // two bf16 per dword: random sign + mantissa, exponent 0x3c..0x3f (|v| in [2^-7, 2)): no overflow in the accumulators
            const unsigned ra = mix32(seed + (blockIdx.x * 256 + threadIdx.x) * 64 + u * 8 + k);
            const unsigned rb = mix32(ra + 0x9e3779b9u);
            av[u][k] = (ra & 0x81ff81ffu) | 0x3c003c00u | ((ra >> 3) & 0x01800180u);
            bv[u][k] = (rb & 0x81ff81ffu) | 0x3c003c00u | ((rb >> 3) & 0x01800180u);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[(u + i) & 7]),
                                                                 __builtin_bit_cast(bf16x8, bv[(u + 3 * i) & 7]), acc[i], 0, 0, 0);
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
    if (s == 12345.678f) out[0] = s;
}
// the 16x16x32 shape of the same pipe on random operands (is the energy per FLOP shape-dependent?)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_random16(float* out, int iters, unsigned seed) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned ra = mix32(seed + (blockIdx.x * 256 + threadIdx.x) * 64 + u * 8 + k);
            const unsigned rb = mix32(ra + 0x9e3779b9u);
            av[u][k] = (ra & 0x81ff81ffu) | 0x3c003c00u | ((ra >> 3) & 0x01800180u);
            bv[u][k] = (rb & 0x81ff81ffu) | 0x3c003c00u | ((rb >> 3) & 0x01800180u);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[(u + i) & 7]),
                                                                 __builtin_bit_cast(bf16x8, bv[(u + 3 * i) & 7]), acc[i], 0, 0, 0);
        if ((it & 255) == 255) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] *= 0.001f;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
// fp32 32x32x2 on random operands (one float per lane and operand): the fp32 path's practical ceiling
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_random_f32(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const unsigned ra = mix32(seed + (blockIdx.x * 256 + threadIdx.x) * 64 + u * 8);
        const unsigned rb = mix32(ra + 0x9e3779b9u);
        av[u] = __uint_as_float((ra & 0x81ffffffu) | 0x3c000000u | ((ra >> 3) & 0x01800000u));   // |v| in [2^-7, 2), random mantissa
        bv[u] = __uint_as_float((rb & 0x81ffffffu) | 0x3c000000u | ((rb >> 3) & 0x01800000u));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u + i) & 7], bv[(u + 3 * i) & 7], acc[i], 0, 0, 0);
        if ((it & 255) == 255) {
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
    if (s == 12345.678f) out[0] = s;
}
template <int NACC>
static void run_random_f32(int blocks_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop_random_f32<NACC>), dim3(grid), dim3(256), 0, 0, out, iters / 4, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop_random_f32<NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("%-34s %d workgroup(s)/CU, %d accumulators/wave: %7.2f ms  %8.1f TFLOP/s\n", "f32 32x32x2, RANDOM operands", blocks_per_cu, NACC, ms,
           flops / ms * 1e-9);
    hipFree(out);
}

template <int NACC>
static void run_random16(int blocks_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop_random16<NACC>), dim3(grid), dim3(256), 0, 0, out, iters / 4, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop_random16<NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 2.0 * 16 * 16 * 32;
    printf("%-34s %d workgroup(s)/CU, %d accumulators/wave: %7.2f ms  %8.1f TFLOP/s\n", "bf16 16x16x32, RANDOM operands", blocks_per_cu, NACC, ms,
           flops / ms * 1e-9);
    hipFree(out);
}

template <int NACC>
static void run_random(int blocks_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop_random<NACC>), dim3(grid), dim3(256), 0, 0, out, iters / 4, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop_random<NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 16;
    printf("%-34s %d workgroup(s)/CU, %d accumulators/wave: %7.2f ms  %8.1f TFLOP/s\n", "bf16 32x32x16, RANDOM operands", blocks_per_cu, NACC, ms,
           flops / ms * 1e-9);
    hipFree(out);
}

template <int NACC, bool BF>
static void run(const char* name, int blocks_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<NACC, BF>), dim3(grid), dim3(256), 0, 0, out, iters / 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, BF>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_mfma = BF ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    const double flops = (double)grid * 4 * iters * 8 * NACC * flop_per_mfma;
    printf("%-34s %d workgroup(s)/CU, %d accumulators/wave: %7.2f ms  %8.1f TFLOP/s\n", name, blocks_per_cu, NACC, ms,
           flops / ms * 1e-9);
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        run<4, false>("v_mfma_f32_32x32x2_f32", bpc, 20000 / bpc);
        run<4, true>("v_mfma_f32_32x32x16_bf16", bpc, 40000 / bpc);
    }
    run<2, true>("v_mfma_f32_32x32x16_bf16", 2, 20000);
    run<1, true>("v_mfma_f32_32x32x16_bf16", 4, 10000);
    run_random<4>(1, 40000);
    run_random<4>(2, 20000);
    run_random<2>(2, 40000);
    run_random<4>(1, 160000);                      // ~70 ms: the clock has settled at the power limit
    run_random_f32<4>(1, 20000);
    run_random_f32<4>(2, 10000);
    run_random_f32<4>(1, 80000);                   // ~70 ms
    run_random16<8>(1, 40000);
    run_random16<8>(2, 20000);
    return 0;
}
