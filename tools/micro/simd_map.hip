// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: simd_id bits [5:4], cu_id [11:8], wave_id [3:0])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 0, 0, d);
    unsigned h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; ++b) {
        printf("wg %2d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd%u cu%u slot%u", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
        printf("\n");
    }
    return 0;
}
