// What ds_read_b64_tr_b16 returns: LDS holds element index values (u16), every lane supplies its own byte address, the 4 result
// elements of every lane are printed.  hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read.hip -o build/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* addr, uint16_t* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(size_t)(__attribute__((address_space(3))) uint16_t*)lds + addr[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    uint32_t* da; uint16_t* dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&dout, 256 * 2);
    for (int pat = 0; pat < 4; ++pat) {
        uint32_t ha[64];
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) ha[l] = l * 8;                                   // lane l -> elements 4l .. 4l+3
            if (pat == 1) ha[l] = (l & 15) * 128 + (l >> 4) * 8;           // rows of 64 elements: row = l&15, col = 4*(l>>4)
            if (pat == 2) ha[l] = (l & 15) * 128 + (l >> 4) * 32;          // row = l&15, col = 16*(l>>4)
            if (pat == 3) ha[l] = ((l & 3) * 4 + ((l >> 2) & 3)) * 128 + (l >> 4) * 8;
        }
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
        uint16_t ho[256];
        hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        printf("pattern %d (lane: addr-elem -> result elems)\n", pat);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: a=%4u -> %4u %4u %4u %4u%s", l, ha[l] / 2, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3], (l & 1) ? "\n" : "   |");
    }
    return 0;
}
