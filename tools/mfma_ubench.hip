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

// MODE 3: the half-tap pipeline of conv3x3_halo_kernel on synthetic addresses: per half (8 k-steps x 4 MFMAs) 2 A reads (b128) +
// B reads (b32) from LDS, 2 global float4 loads from a buffer far larger than L2, their ds_write_b128 in the middle of the next half,
// one barrier per half.  MODE 4: the same without the global loads (stores write registers).  MODE 5: no stores either.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// MODE 6: the weight halves by LDS-DMA (issued at the start of a half into the other buffer, vmcnt(0) before the barrier)
template <int MODE>
__global__ __launch_bounds__(256, 3) void k2(const float* __restrict__ g, float* out, int iters, long gstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [A: 180 x 36] [B: 2 x 16 x 128]
    for (int i = threadIdx.x; i < 180 * 36 + 2 * 16 * 128; i += 256) lds[i] = (float)(i % 7) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, tid = threadIdx.x;
    const int kl = lane >> 5, ml = lane & 31;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float* As = lds; float* Bs = lds + 180 * 36;
    const float* gp = g + (gstride == 4096 ? (long)((blockIdx.x & 3) * 256 + tid) * 4 : ((long)blockIdx.x * 256 + tid) * 4);   // 4096: 1 MB shared by all
    f32x4 rh[2][2];
    rh[0][0] = rh[0][1] = rh[1][0] = rh[1][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int p0 = 11 + (wid >> 1) * 40 + (ml & 7) + (ml >> 3) * 10, p1 = p0 + 20;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (MODE == 6) {
                const unsigned d = (unsigned)(size_t)(Bs + (half ^ 1) * 16 * 128) + __builtin_amdgcn_readfirstlane(wid) * 1024;
                glds16(gp + ((long)(it * 2 + half) % 64) * gstride, __builtin_amdgcn_readfirstlane(d));
                glds16(gp + ((long)(it * 2 + half) % 64) * gstride + 1024, __builtin_amdgcn_readfirstlane(d + 4096));
            }
            if (MODE == 3 || MODE == 7) {
                rh[half][0] = *reinterpret_cast<const f32x4*>(gp + ((long)(it * 2 + half) % 64) * gstride);
                rh[half][1] = *reinterpret_cast<const f32x4*>(gp + ((long)(it * 2 + half) % 64) * gstride + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* a0 = As + (p0 + (it % 3)) * 36 + half * 16 + kl * 4;
            const float* a1 = As + (p1 + (it % 3)) * 36 + half * 16 + kl * 4;
            const float* b = Bs + half * 16 * 128 + kl * 4 * 128 + (wid & 1) * 64 + ml;
            f32x4 af[2][2];
            af[0][0] = *reinterpret_cast<const f32x4*>(a0); af[0][1] = *reinterpret_cast<const f32x4*>(a1);
            float bf[3][2];
            for (int s0 = 0; s0 < 2; ++s0) for (int j = 0; j < 2; ++j) bf[s0][j] = b[s0 * 128 + j * 32];
            af[1][0] = *reinterpret_cast<const f32x4*>(a0 + 8); af[1][1] = *reinterpret_cast<const f32x4*>(a1 + 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int cur = ks % 3, nxt = (ks + 2) % 3;
                if (ks + 2 < 8) { const int kn = ks + 2; for (int j = 0; j < 2; ++j) bf[nxt][j] = b[((kn >> 2) * 8 + (kn & 3)) * 128 + j * 32]; }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks >> 2][i][ks & 3], bf[cur][j], acc[i * 2 + j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 3 && MODE != 5 && MODE != 6) {
                    float* d = Bs + (half ^ 1) * 16 * 128;
                    *reinterpret_cast<f32x4*>(d + tid * 4) = rh[half ^ 1][0];
                    *reinterpret_cast<f32x4*>(d + 1024 + tid * 4) = rh[half ^ 1][1];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE != 7 || half == 1) __syncthreads();          // MODE 7: one barrier per 64 MFMAs (full-tap weight buffers)
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run2(const char* name, int blocks, int shared_rows = 0) {
    float *out, *g;
    const long gstride = shared_rows ? 4096 : (long)blocks * 256 * 4 + 4096;
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&g, (size_t)gstride * 64 * 4 + 65536);
    (void)hipMemset(g, 0, (size_t)gstride * 64 * 4 + 65536);
    const int iters = 1000;
    const size_t ldsb = (180 * 36 + 2 * 16 * 128) * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<MODE>), dim3(blocks), dim3(256), ldsb, 0, g, out, 10, gstride);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k2<MODE>), dim3(blocks), dim3(256), ldsb, 0, g, out, iters, gstride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-52s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
    (void)hipFree(out); (void)hipFree(g);
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
    run2<5>("half-tap pipeline, no stores, 3 WG/CU", 768);
    run2<4>("half-tap pipeline + LDS stores, 3 WG/CU", 768);
    run2<3>("half-tap pipeline + global loads + stores, 3 WG/CU", 768);
    run2<3>("half-tap pipeline + global loads + stores, 16 rounds", 768 * 16);
    run2<7>("loads + stores, barrier per 64 MFMAs, 3 WG/CU x16", 768 * 16);
    run2<7>("loads + stores, barrier per 64 MFMAs, 2 WG/CU x16", 512 * 16);
    run2<3>("loads + stores, barrier per 32 MFMAs, 2 WG/CU x16", 512 * 16);
    run2<6>("half-tap pipeline + LDS-DMA weights, 3 WG/CU", 768);
    run2<6>("half-tap pipeline + LDS-DMA weights, 16 rounds", 768 * 16);
    run2<6>("... DMA from an L2-resident 1 MB, 16 rounds", 768 * 16, 1);
    run2<3>("... loads from an L2-resident 1 MB, 3 WG/CU", 768, 1);
    run2<3>("... loads from an L2-resident 1 MB, 16 rounds", 768 * 16, 1);
    return 0;
}
