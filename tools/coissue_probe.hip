// How fast does a wave issue VALU / LDS-write / LDS-read instructions while ANOTHER wave of the same SIMD streams MFMAs?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run an MFMA stream (or idle), waves 4-7 (same SIMDs) run a timed loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/coissue_probe tools/coissue_probe.hip && tools/bin/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_s(const void* gbase, unsigned voff, unsigned lds_dst) {      // SGPR base + 32-bit VGPR offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
// MF: 0 idle partner, 1 fp32 32x32x2 stream, 2 bf16 32x32x16 stream.  WORK: 0 independent v_fma, 1 ds_write_b32, 2 ds_read_b32
template <int MF, int WORK, int PRIO = 0>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, float* sink, int mf_iters, int work_iters) {
    __shared__ float lds[8192];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wid < 4) {
        if (MF == 0) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        const float a = lane * 1e-3f, b = 1.0f - lane * 1e-3f;
        bf16x8 ab, bb;
        for (int k = 0; k < 8; ++k) { ab[k] = (__bf16)(a + k); bb[k] = (__bf16)(b - k); }
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        f32x4_ acc4[8];
        for (int i = 0; i < 8; ++i) acc4[i] = (f32x4_){0.0f, 0.0f, 0.0f, 0.0f};
        const float* pr = lds + wid * 128 + lane;
        for (int it = 0; it < mf_iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (MF == 1) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
                else if (MF == 2) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[u & 3], 0, 0, 0);
                else if (MF == 3) {                              // the halo kernel's pattern: an LDS operand read + wait per 4 MFMAs
                    float av = a;
                    if ((u & 3) == 0) av = pr[(u >> 2) * 64];
                    acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[u & 3], 0, 0, 0);
                } else acc4[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u & 7], 0, 0, 0);
            }
        for (int i = 0; i < 8; ++i) acc[0][0] += acc4[i][0];
        float s = 0.0f;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 12345.678f) sink[0] = s;
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);                // does a higher issue priority get the starved instruction forms through?
    // let the MFMA stream get going
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
    float v[16];
    f32x4 vq[4] = {};
    for (int i = 0; i < 16; ++i) v[i] = lane * 0.001f + i;
    int su = __builtin_amdgcn_readfirstlane(work_iters);      // wave-uniform chain for the SALU / SMEM probes
    const float* __restrict__ csink = sink;
    float* p = lds + (wid - 4) * 2048 + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < work_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (WORK == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            if (WORK == 1) p[i * 64] = v[i];
            if (WORK == 2) v[i] += p[i * 64];
            if (WORK == 3 && i < 4) *reinterpret_cast<f32x4*>(lds + (wid - 4) * 2048 + (i * 64 + lane) * 4) = (f32x4){v[i], v[i + 1], v[i + 2], v[i + 3]};
            if (WORK == 4 && i < 4) glds16(sink + 64 + (i * 64 + lane) * 4, __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + (wid - 4) * 2048 + i * 256) & 0xffffu));
            if (WORK == 5 && i < 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(sink + 64 + ((it & 3) * 256 + i * 64 + lane) * 4); v[i] += t.x; }
            if (WORK == 6 && i < 4) sink[64 + 4096 + (wid * 4 + i) * 64 + lane] = v[i];
            if (WORK == 9 && i < 4) asm volatile("global_store_dword %0, %1, %2" :: "v"((unsigned)(((wid * 4 + i) * 64 + lane) * 4)), "v"(v[i]), "s"(sink + 64 + 4096) : "memory");
            if (WORK == 10 && i < 4) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"((unsigned)(((wid * 4 + i) * 64 + lane) * 16)), "v"((f32x4){v[i], v[i + 1], v[i + 2], v[i + 3]}), "s"(sink + 64 + 4096) : "memory");
            if (WORK == 11 && i < 4) *reinterpret_cast<f32x4*>(sink + 64 + 4096 + ((wid * 4 + i) * 64 + lane) * 4) = (f32x4){v[i], v[i + 1], v[i + 2], v[i + 3]};
            if (WORK == 12 && i < 4) glds16_s(sink + 64, (unsigned)((i * 64 + lane) * 16), __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + (wid - 4) * 2048 + i * 256) & 0xffffu));
            if (WORK == 13 && i < 4) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)(lds + (wid - 4) * 2048 + i * 64 + lane) & 0xffffu), "v"(v[i]) : "memory");
            if (WORK == 14) { su = su * 3 + 7 + i; su ^= su >> 3; }
            if (WORK == 15 && i < 4) { su += reinterpret_cast<const int*>(csink)[64 + ((su + i) & 1023)]; }
            if (WORK == 16) { unsigned u_ = __float_as_uint(v[i]); u_ = u_ * 2654435761u + 12345u; v[i] = __uint_as_float((u_ & 0x007fffffu) | 0x3f800000u); }
            if (WORK == 17 && i < 4) { f32x4 t_; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t_) : "v"(sink + 64 + ((it & 3) * 256 + i * 64 + lane) * 4) : "memory"); vq[i] = t_; }
            if (WORK == 18 && i < 4) { f32x4 t_; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t_) : "v"((unsigned)(((it & 3) * 256 + i * 64 + lane) * 16)), "s"(sink + 64) : "memory"); vq[i] = t_; }
            if (WORK == 19 && i < 4) v[i] += __shfl_xor(v[i], 32, 64);
            if (WORK == 7 && i < 4) *reinterpret_cast<float2*>(lds + (wid - 4) * 2048 + (i * 64 + lane) * 2) = make_float2(v[i], v[i + 1]);
            if (WORK == 8 && i < 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(lds + (wid - 4) * 2048 + (i * 64 + lane) * 4); v[i] += t.x; }
        }
        if (WORK == 17 || WORK == 18) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); v[0] += vq[0].x + vq[1].x + vq[2].x + vq[3].x; }
        if (WORK == 4 || WORK == 5 || WORK == 6 || (WORK >= 9 && WORK <= 12)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (WORK != 0) __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f || su == 123457) sink[1] = s + lds[lane] + su;
    if (lane == 0 && wid == 4 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MF, int WORK, int PRIO = 0>
static void run(const char* name) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 8); hipMalloc(&sink, 65536); hipMemset(sink, 0, 65536);
    const int work_iters = 200;
    hipLaunchKernelGGL((probe<MF, WORK, PRIO>), dim3(256), dim3(512), 0, 0, out, sink, 4000, work_iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    const int per = WORK >= 3 ? 4 : 16;
    printf("%-52s %8.1f clk per instruction (%d instructions)\n", name, (double)h / (work_iters * per), work_iters * per);
    hipFree(out); hipFree(sink);
}

int main() {
    run<0, 0>("v_fma, partner idle");
    run<1, 0>("v_fma beside fp32 32x32x2 MFMA stream");
    run<2, 0>("v_fma beside bf16 32x32x16 MFMA stream");
    run<0, 1>("ds_write_b32, partner idle");
    run<1, 1>("ds_write_b32 beside fp32 MFMA stream");
    run<2, 1>("ds_write_b32 beside bf16 MFMA stream");
    run<0, 2>("ds_read_b32 (+add), partner idle");
    run<1, 2>("ds_read_b32 (+add) beside fp32 MFMA stream");
    run<2, 2>("ds_read_b32 (+add) beside bf16 MFMA stream");
    run<3, 1>("ds_write_b32 beside fp32 MFMA + LDS-read stream");
    run<4, 1>("ds_write_b32 beside fp32 16x16x4 MFMA stream");
    run<0, 3>("ds_write_b128, partner idle");
    run<1, 3>("ds_write_b128 beside fp32 MFMA stream");
    run<3, 3>("ds_write_b128 beside fp32 MFMA + LDS-read stream");
    run<0, 5>("global_load_dwordx4, partner idle");
    run<1, 5>("global_load_dwordx4 beside fp32 MFMA stream");
    run<0, 6>("global_store_dword, partner idle");
    run<1, 6>("global_store_dword beside fp32 MFMA stream");
    run<2, 6>("global_store_dword beside bf16 MFMA stream");
    run<1, 9>("global_store_dword (SGPR base) beside fp32 MFMA");
    run<2, 9>("global_store_dword (SGPR base) beside bf16 MFMA");
    run<1, 10>("global_store_dwordx4 (SGPR base) beside fp32 MFMA");
    run<2, 10>("global_store_dwordx4 (SGPR base) beside bf16 MFMA");
    run<1, 11>("global_store_dwordx4 (64-bit VGPR addr) beside fp32");
    run<2, 11>("global_store_dwordx4 (64-bit VGPR addr) beside bf16");
    run<0, 12>("global_load_lds_dwordx4 (SGPR base), partner idle");
    run<1, 12>("global_load_lds_dwordx4 (SGPR base) beside fp32");
    run<2, 12>("global_load_lds_dwordx4 (SGPR base) beside bf16");
    run<0, 13>("single ds_write_b32, partner idle");
    run<1, 13>("single ds_write_b32 beside fp32 MFMA");
    run<2, 13>("single ds_write_b32 beside bf16 MFMA");
    run<0, 14>("5 dependent SALU ops, partner idle");
    run<1, 14>("5 dependent SALU ops beside fp32 MFMA");
    run<2, 14>("5 dependent SALU ops beside bf16 MFMA");
    run<3, 14>("5 dependent SALU ops beside fp32 MFMA + LDS-read stream");
    run<1, 14, 3>("5 dependent SALU ops beside fp32 MFMA, s_setprio 3");
    run<1, 13, 3>("single ds_write_b32 beside fp32 MFMA, s_setprio 3");
    run<1, 6, 3>("global_store_dword beside fp32 MFMA, s_setprio 3");
    run<2, 4, 3>("global_load_lds_dwordx4 beside bf16 MFMA, s_setprio 3");
    run<0, 15>("dependent s_load_dword (+2 SALU), partner idle");
    run<1, 15>("dependent s_load_dword (+2 SALU) beside fp32 MFMA");
    run<0, 16>("v_mul_lo_u32 + 3 int VALU, partner idle");
    run<1, 16>("v_mul_lo_u32 + 3 int VALU beside fp32 MFMA");
    run<0, 17>("global_load_dwordx4 (64-bit VGPR addr), partner idle");
    run<1, 17>("global_load_dwordx4 (64-bit VGPR addr) beside fp32");
    run<2, 17>("global_load_dwordx4 (64-bit VGPR addr) beside bf16");
    run<2, 18>("global_load_dwordx4 (SGPR base) beside bf16");
    run<0, 19>("__shfl_xor 32 (ds_bpermute) + add, partner idle");
    run<1, 19>("__shfl_xor 32 (ds_bpermute) + add beside fp32 MFMA");
    run<0, 7>("ds_write_b64, partner idle");
    run<1, 7>("ds_write_b64 beside fp32 MFMA stream");
    run<0, 8>("ds_read_b128, partner idle");
    run<1, 8>("ds_read_b128 beside fp32 MFMA stream");
    run<2, 4>("global_load_lds_dwordx4 beside bf16 MFMA stream");
    run<0, 4>("global_load_lds_dwordx4, partner idle");
    run<1, 4>("global_load_lds_dwordx4 beside fp32 MFMA stream");
    run<3, 4>("global_load_lds_dwordx4 beside fp32 MFMA + LDS-read");
    return 0;
}
