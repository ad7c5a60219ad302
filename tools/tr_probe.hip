// Probe of ds_read_b64_tr_b16 lane semantics on gfx950: LDS holds element index i at element i; lane l passes the
// address of element A(l); prints what each lane receives.  hipcc --offload-arch=gfx950 tools/tr_probe.hip -o tools/bin/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(s16x4* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) short sm[];
    for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int a = l * 4;                                   // mode 0: linear
    if (mode == 1) a = (l & 15) * 4 + (l >> 4) * 1024;  // groups far apart
    if (mode == 2) a = ((l & 3) * 4) + ((l >> 2) & 3) * 100 * 4 + (l >> 4) * 2048;   // rows (i>>2) at a 100-quad stride
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + a));
    out[l] = v;
}
int main() {
    s16x4* d;
    hipMalloc(&d, 64 * sizeof(s16x4));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, mode);
        s16x4 h[64];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l].x, h[l].y, h[l].z, h[l].w);
    }
    return 0;
}
