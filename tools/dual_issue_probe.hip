// Can a CU run the exact-fp32 MFMA and packed fp32 VALU FMAs AT THE SAME TIME, and what does the board do (power, clock)?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) stream v_mfma_f32_32x32x2_f32 on register operands, waves 4-7 (the
// same SIMDs) stream v_pk_fma_f32 on 16 independent float2 accumulators -- each alone, then both.  Random fp32 operands.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dual_issue_probe tools/dual_issue_probe.hip && /tmp/dual_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mix32(unsigned z) {
    z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
    return z;
}
__device__ __forceinline__ float rnd(unsigned s) {           // |v| in [2^-7, 2), random mantissa and sign
    const unsigned r = mix32(s);
    return __uint_as_float((r & 0x81ffffffu) | 0x3c000000u | ((r >> 3) & 0x01800000u));
}

// MODE bit 0: the MFMA waves run, bit 1: the VALU waves run.  clocks[wave of block 0][2] = shader clocks, 100 MHz ticks
template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* clocks, float* sink, int mf_iters, int va_iters) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    if (wid < 4) {
        if (!(MODE & 1)) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        float av[8], bv[8];
        for (int u = 0; u < 8; ++u) { av[u] = rnd(blockIdx.x * 4096 + threadIdx.x * 8 + u); bv[u] = rnd(77777 + blockIdx.x * 4096 + threadIdx.x * 8 + u); }
        for (int it = 0; it < mf_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u & 7], bv[(3 * u) & 7], acc[u & 3], 0, 0, 0);
            if ((it & 255) == 255)
                for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        if (!(MODE & 2)) return;
        f32x2 acc[16], a[4], b[4];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x2){0.0f, 0.0f};
        for (int u = 0; u < 4; ++u) {
            a[u] = (f32x2){rnd(threadIdx.x * 16 + u), rnd(threadIdx.x * 16 + 4 + u)};
            b[u] = (f32x2){rnd(9999 + threadIdx.x * 16 + u) * 0.5f, rnd(9999 + threadIdx.x * 16 + 4 + u) * 0.5f};
        }
        for (int it = 0; it < va_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[u & 15]) : "v"(a[u & 3]), "v"(b[(u >> 2) & 3]));
            if ((it & 63) == 63)
                for (int i = 0; i < 16; ++i) acc[i] *= 0.001f;
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    }
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && lane == 0) {
        clocks[2 * wid] = __builtin_amdgcn_s_memtime() - t0;
        clocks[2 * wid + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE>
static void run(const char* name, int mf_iters, int va_iters) {
    unsigned long long* clocks;
    float* sink;
    hipMalloc(&clocks, 16 * 8);
    hipMalloc(&sink, 4);
    hipMemset(clocks, 0, 16 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, clocks, sink, mf_iters / 8, va_iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, clocks, sink, mf_iters, va_iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost);
    // per-wave rates from the wave's own clocks (block 0): FLOP per shader clock per SIMD, and the clock it saw
    const double mf_flop = (double)mf_iters * 16 * 2.0 * 32 * 32 * 2, va_flop = (double)va_iters * 32 * 64 * 2 * 2.0;
    printf("%-22s kernel %7.2f ms |", name, ms);
    if (MODE & 1) printf(" MFMA wave: %5.1f FLOP/clk/SIMD, %7.1f us, sclk %4.0f MHz, chip %6.1f TFLOP/s |", mf_flop / (double)h[0],
                         h[1] / 100.0, (double)h[0] / h[1] * 100.0, mf_flop * 4 * 256 / (h[1] / 100.0 * 1e-6) / 1e12);
    if (MODE & 2) printf(" VALU wave: %5.1f FLOP/clk/SIMD, %7.1f us, sclk %4.0f MHz, chip %6.1f TFLOP/s |", va_flop / (double)h[8],
                         h[9] / 100.0, (double)h[8] / h[9] * 100.0, va_flop * 4 * 256 / (h[9] / 100.0 * 1e-6) / 1e12);
    printf("\n");
    hipFree(clocks); hipFree(sink);
}

int main() {
    // ~60 ms each: 16 MFMAs x 64 clocks = 1024 clocks per MFMA iteration; 32 pk_fma x 4 clocks = 128 clocks per VALU iteration
    const int mf = 140000, va = 1120000;
    for (int rep = 0; rep < 2; ++rep) {
        run<1>("MFMA fp32 alone", mf, va);
        run<2>("v_pk_fma_f32 alone", mf, va);
        run<3>("both, same SIMDs", mf, va);
    }
    run<3>("both, VALU half work", mf, va / 2);
    return 0;
}
