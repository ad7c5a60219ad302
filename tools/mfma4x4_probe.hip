// Operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, found empirically:
//   pass 1: a = lane + 1, b = 1      -> D[lane][r] = a of the lane that supplied A for (block, row)
//   pass 2: a = 1,        b = lane+1 -> D[lane][r] = b of the lane that supplied B for (block, col)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma4x4_probe tools/mfma4x4_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x4 z = {0, 0, 0, 0};
    f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);
    f32x4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { out[(l * 4 + r) * 2] = d1[r]; out[(l * 4 + r) * 2 + 1] = d2[r]; }
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 2 * sizeof(float));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[64 * 4 * 2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int a_lane = (int)h[(l * 4 + r) * 2] - 1, b_lane = (int)h[(l * 4 + r) * 2 + 1] - 1;
            // hypothesis: D[lane l][reg r] = A[lane 4*(l/4) + r] * B[lane l]   (block = l/4, row = reg, col = l%4)
            if (a_lane != 4 * (l / 4) + r || b_lane != l) ok = 0;
            if (l < 8 || l == 63) printf("lane %2d reg %d: A from lane %2d, B from lane %2d\n", l, r, a_lane, b_lane);
        }
    printf("hypothesis D[l][r] = A[4*(l/4)+r] * B[l]: %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
