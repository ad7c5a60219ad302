// What does non-MFMA work cost beside fp32 MFMA groups when TWO waves share a SIMD and meet at a barrier per "chunk"?
// (the regime of wino_fused_kernel: 8 waves per workgroup, one workgroup per CU, 8 groups of 4 MFMAs per wave and chunk)
//   hipcc --offload-arch=gfx950 -O3 tools/coissue2_probe.hip -o tools/bin/coissue2_probe && tools/bin/coissue2_probe
// Per wave and chunk: 8 x [ 4 x v_mfma_f32_32x32x2_f32 ][ NF filler instructions ].
//   DEP   1: the 4 MFMAs of a group accumulate into ONE register set (dependent), 0: into four sets
//   ORDER 0: every wave runs M F M F ..; 1: waves 4-7 (the second wave of each SIMD) run F M F M .. (complementary)
//         2: fillers spread behind every single MFMA (M f M f ..), same total
//   FILL  0: v_fma_f32   1: ds_write_b128   2: global_load_dwordx4 (L2-resident)   3: ds_read_b128
// Prints clocks per MFMA and SIMD (ideal 64 = pipe saturated by the two waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FILL>
__device__ __forceinline__ void filler(float (&v)[8], int i, float c1, float c2, float* lds, const float* g, f32x4& sink, const float* gbase = nullptr, float* ldsbase = nullptr) {
    if (FILL == 0) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(c1), "v"(c2));
    } else if (FILL == 1) {
        f32x4 w = {v[0], v[1], v[2], v[3]};
        asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)lds + (unsigned)(i & 7) * 16384u), "v"(w) : "memory");
    } else if (FILL == 2) {
        f32x4 r;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(g + (size_t)(i & 7) * 4096) : "memory");
        sink = r;        // (never waited for inside the loop: s_waitcnt only at the end)
    } else if (FILL == 5) {          // LDS-DMA, wave-uniform SGPR base + 32-bit lane offset; destination = M0 (wave-uniform) + 16 B * lane
        unsigned keep;
        const unsigned voff = (unsigned)(threadIdx.x * 16u + (unsigned)(i & 7) * 16384u);
        const unsigned dst = (unsigned)(size_t)ldsbase + (threadIdx.x >> 6) * 1024u + (unsigned)(i & 7) * 16384u;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
    } else if (FILL == 6) {          // global load, SGPR base + 32-bit lane offset
        f32x4 r;
        const unsigned voff = (unsigned)(threadIdx.x * 16u + (unsigned)(i & 7) * 16384u);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(gbase) : "memory");
        sink = r;
    } else if (FILL == 4) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 w = {v[(2 * i) & 7], v[(2 * i + 1) & 7]}, cc1 = {c1, c1}, cc2 = {c2, c2};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w) : "v"(cc1), "v"(cc2));
        v[(2 * i) & 7] = w[0]; v[(2 * i + 1) & 7] = w[1];
    } else {
        f32x4 r;
        asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((unsigned)(size_t)lds + (unsigned)(i & 7) * 16384u) : "memory");
        sink = r;
    }
}

template <int DEP, int ORDER, int FILL, int NF>
__global__ __launch_bounds__(512) void k(float* out, const float* g, int iters, int barrier) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* lds = smem + threadIdx.x * 4;
    const float* gp = g + threadIdx.x * 4;
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float a0 = lane * 1e-3f, b0 = wave * 1e-3f, c1 = 0.999f, c2 = 1e-3f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    f32x4 sink = {0, 0, 0, 0};
    const bool second = wave >= 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            if (ORDER == 1 && second) {
#pragma unroll
                for (int i = 0; i < NF; ++i) filler<FILL>(v, i, c1, c2, lds, gp, sink, g, smem);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ORDER == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int a = DEP ? j : (4 * (j & 1) + e);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[a], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = e * NF / 4; i < (e + 1) * NF / 4; ++i) filler<FILL>(v, i, c1, c2, lds, gp, sink, g, smem);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (DEP == 2) {
                // the same FLOP on v_mfma_f32_16x16x4_f32: 8 MFMAs per group, 2 k-steps into each of the 4 quads (16 x 16 blocks) of set j
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        f32x4 t = {acc[j][4 * qd], acc[j][4 * qd + 1], acc[j][4 * qd + 2], acc[j][4 * qd + 3]};
                        t = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, t, 0, 0, 0);
                        acc[j][4 * qd] = t[0]; acc[j][4 * qd + 1] = t[1]; acc[j][4 * qd + 2] = t[2]; acc[j][4 * qd + 3] = t[3];
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ORDER == 1 && second)) {
#pragma unroll
                    for (int i = 0; i < NF; ++i) filler<FILL>(v, i, c1, c2, lds, gp, sink, g, smem);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int a = DEP ? j : (4 * (j & 1) + e);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[a], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ORDER == 1 && second)) {
#pragma unroll
                    for (int i = 0; i < NF; ++i) filler<FILL>(v, i, c1, c2, lds, gp, sink, g, smem);
                }
            }
        }
        if (FILL >= 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (barrier) __syncthreads();
    }
    float s = sink[0] + sink[1];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float* d_out;
static float* d_g;

template <int DEP, int ORDER, int FILL, int NF>
void run(int barrier) {
    const int iters = 2000, lds = 140 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<DEP, ORDER, FILL, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<DEP, ORDER, FILL, NF>), dim3(256), dim3(512), lds, 0, d_out, d_g, 200, barrier);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<DEP, ORDER, FILL, NF>), dim3(256), dim3(512), lds, 0, d_out, d_g, iters, barrier);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x 8 groups x 4 MFMAs per iteration
    const double ns_per_mfma = ms * 1e6 / ((double)iters * 64);
    printf("DEP %d ORDER %d FILL %d NF %2d barrier %d: %7.2f ns per MFMA slot = %6.1f clk @2.4GHz  (%.1f TF/s)\n", DEP, ORDER, FILL, NF, barrier,
           ns_per_mfma, ns_per_mfma * 2.4, 256.0 * 4 * 4096 / ns_per_mfma * 1e-3);
}

template <int DEP, int ORDER, int FILL>
void sweep(int barrier) {
    run<DEP, ORDER, FILL, 0>(barrier);
    run<DEP, ORDER, FILL, 4>(barrier);
    run<DEP, ORDER, FILL, 8>(barrier);
    run<DEP, ORDER, FILL, 16>(barrier);
    run<DEP, ORDER, FILL, 32>(barrier);
}

int main() {
    hipMalloc(&d_out, 256 * 512 * sizeof(float));
    hipMalloc(&d_g, (512 * 4 + 8 * 4096 + 64) * sizeof(float));
    hipMemset(d_g, 0, (512 * 4 + 8 * 4096 + 64) * sizeof(float));
    printf("== VALU fillers\n");
    sweep<1, 0, 0>(1); sweep<1, 1, 0>(1); sweep<1, 2, 0>(1); sweep<0, 0, 0>(1); sweep<0, 2, 0>(1);
    printf("== ds_write_b128 fillers\n");
    run<1, 0, 1, 2>(1); run<1, 0, 1, 4>(1); run<1, 1, 1, 2>(1); run<1, 1, 1, 4>(1); run<0, 2, 1, 4>(1);
    printf("== global_load_dwordx4 fillers\n");
    run<1, 0, 2, 2>(1); run<1, 0, 2, 4>(1); run<1, 1, 2, 2>(1); run<1, 1, 2, 4>(1); run<0, 2, 2, 4>(1);
    printf("== ds_read_b128 fillers\n");
    run<1, 0, 3, 2>(1); run<1, 0, 3, 4>(1); run<1, 1, 3, 2>(1); run<1, 1, 3, 4>(1);
    printf("== v_mfma_f32_16x16x4_f32 (DEP 2), VALU / ds_write / global_load fillers\n");
    sweep<2, 0, 0>(1); sweep<2, 1, 0>(1);
    run<2, 0, 1, 4>(1); run<2, 0, 2, 4>(1); run<2, 1, 2, 4>(1);
    printf("== v_pk_fma_f32 fillers (two FMAs per lane each)\n");
    sweep<1, 0, 4>(1); run<2, 0, 4, 16>(1);
    printf("== LDS-DMA (FILL 5) and saddr global loads (FILL 6)\n");
    run<1, 0, 5, 2>(1); run<1, 0, 5, 4>(1); run<1, 1, 5, 4>(1); run<1, 0, 5, 8>(1);
    run<1, 0, 6, 2>(1); run<1, 0, 6, 4>(1); run<1, 1, 6, 4>(1);
    printf("== no barrier\n");
    run<1, 0, 0, 0>(0); run<1, 0, 0, 16>(0); run<1, 1, 0, 16>(0); run<1, 0, 0, 32>(0);
    return 0;
}
