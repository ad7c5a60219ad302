// micro-benchmark: fp32 MFMA issue rate with/without the LDS fragment-read pattern of the conv kernels
// hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o /tmp/mfma_ubench && /tmp/mfma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[2 * 32 * 129 + 2 * 32 * 128];
    for (int i = threadIdx.x; i < 2 * 32 * 129 + 2 * 32 * 128; i += 256) lds[i] = (float)(i % 7) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int kl = lane >> 5, ml = lane & 31;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float a0 = lane * 1e-3f, b0 = wid * 1e-3f;
    const float* pa = lds + kl * 129 + (wid >> 1) * 64 + ml;
    const float* pb = lds + 2 * 32 * 129 + kl * 128 + (wid & 1) * 64 + ml;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[a], 0, 0, 0);
        } else {
            float af[3][2], bf[3][2];
#pragma unroll
            for (int s0 = 0; s0 < 2; ++s0) {
                af[s0][0] = pa[2 * s0 * 129]; af[s0][1] = pa[2 * s0 * 129 + 32];
                bf[s0][0] = pb[2 * s0 * 128]; bf[s0][1] = pb[2 * s0 * 128 + 32];
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int cur = s % 3, nxt = (s + 2) % 3;
                if (s + 2 < 16) {
                    af[nxt][0] = pa[2 * (s + 2) * 129]; af[nxt][1] = pa[2 * (s + 2) * 129 + 32];
                    bf[nxt][0] = pb[2 * (s + 2) * 128]; bf[nxt][1] = pb[2 * (s + 2) * 128 + 32];
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][0], bf[cur][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][0], bf[cur][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][1], bf[cur][0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][1], bf[cur][1], acc[3], 0, 0, 0);
                if (s + 2 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            if (MODE == 2) __syncthreads();
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, int blocks) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16 * (MODE == 0 ? NACC : 4) * 4096.0;
    printf("%-44s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    run<0, 4>("MFMA only, 4 acc, 1 WG/CU", 256);
    run<0, 4>("MFMA only, 4 acc, 2 WG/CU", 512);
    run<0, 2>("MFMA only, 2 acc, 2 WG/CU", 512);
    run<1, 4>("MFMA + LDS frag reads (PF2), 1 WG/CU", 256);
    run<1, 4>("MFMA + LDS frag reads (PF2), 2 WG/CU", 512);
    run<2, 4>("MFMA + LDS reads + barrier/chunk, 2 WG/CU", 512);
    run<2, 4>("MFMA + LDS reads + barrier/chunk, 1 WG/CU", 256);
    run<1, 4>("MFMA + LDS frag reads (PF2), 3 WG/CU", 768);
    run<2, 4>("MFMA + LDS reads + barrier/chunk, 3 WG/CU", 768);
    return 0;
}
