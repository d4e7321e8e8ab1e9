// LDS cycles of ds_read_b64_tr_b16 under different lane-address patterns (one workgroup of 4 waves, 256 reads per wave, wall clock
// of the whole loop from s_memtime).  hipcc --offload-arch=gfx950 -O2 tools/micro/tr_conflict.hip -o build/tr_conflict
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k(const uint32_t* addr, uint64_t* out, int wide) {
    __shared__ uint16_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(size_t)(__attribute__((address_space(3))) uint16_t*)lds + addr[threadIdx.x & 63];
    uint2 acc = {0, 0};
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        uint2 v0, v1, v2, v3;
        if (wide) {
            uint4 w0, w1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=v"(w0), "=v"(w1) : "v"(a) : "memory");
            acc.x ^= w0.x ^ w1.x; acc.y ^= w0.w ^ w1.w;
        } else {
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\tds_read_b64_tr_b16 %2, %4 offset:4096\n\tds_read_b64_tr_b16 %3, %4 offset:4608\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");
            acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x; acc.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y;
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc.x == 0x12345678u) out[1] = acc.y;
}
int main() {
    uint32_t* da; uint64_t* dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&dout, 16);
    const char* names[] = {"canonical 4x16 blocks (lds[(l&15)*4.. contiguous per group)", "rows of 128 B, no swizzle", "rows of 128 B, current swizzle s = (row>>3&3)<<1 | row>>1&1",
                           "rows of 128 B, s = (b1<<1)|(b3<<2)", "rows of 128 B, s = (b1<<1)|(b3<<2)|b2", "b128 fragment reads, current swizzle (reference: 4 cycles each)"};
    for (int pat = 0; pat < 6; ++pat) {
        uint32_t ha[64];
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, m = l & 15;
            const int row = 8 * g + (m >> 2), piece = m & 3;   // 4 q rows x 16 d columns per 16-lane group, d block nd = 0
            int s = 0;
            if (pat == 2) s = (((row >> 3) & 3) << 1) | ((row >> 1) & 1);
            if (pat == 3) s = (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
            if (pat == 4) s = (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2) | ((row >> 2) & 1);
            if (pat == 0) ha[l] = l * 8;
            else if (pat == 5) { const int c = m, r = (c >> 2) * 8 + (c & 3); const int sw = ((c >> 2) << 1) | ((c >> 1) & 1); ha[l] = r * 128 + ((g ^ sw) << 4); }
            else ha[l] = row * 128 + (((piece >> 1) ^ s) << 4) + (piece & 1) * 8;
        }
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
        uint64_t best = ~0ull;
        for (int r = 0; r < 5; ++r) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, da, dout, pat == 5 ? 1 : 0);
            uint64_t h[2]; hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
            if (h[0] < best) best = h[0];
        }
        const int per_wave = pat == 5 ? 128 : 256;
        printf("%-75s %6llu ticks for %d reads per wave x 4 waves\n", names[pat], (unsigned long long)best, per_wave);
    }
    return 0;
}
