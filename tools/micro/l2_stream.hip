// Micro-benchmark (tools/micro/run_l2_stream.sh): how fast can ONE CU pull bytes out of the L2 / Infinity Cache / HBM
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into an LDS ring, counted vmcnt
//   mode 1: global_load_dwordx4 into VGPRs (xor-reduced so that the loads stay), 8 in flight per wave
//   mode 2: global_load_dwordx4 -> ds_write_b128 (register-staged fill of the same LDS ring)
// Every workgroup (256 threads) streams `iters` slices of 16 KiB, cycling through a footprint of F bytes from its own offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NS>
__global__ __launch_bounds__(256) void stream_dma(const char* __restrict__ base, uint32_t F, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 4096u;
    uint32_t off = (uint32_t)(((uint64_t)blockIdx.x * 2654435761ull) % (F / 16384)) * 16384u;
    const uint32_t voff = lane * 16u;
    auto issue = [&](int stage) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const char* sb = base + off + wave * 4096u + p * 1024u;
            const uint32_t dst = lds0 + stage * 16384u + p * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sb) : "memory");
        }
        off += 16384u;
        if (off >= F) off = 0;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue(p);
    int st = NS - 1;
    for (int t = 0; t < iters; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 4) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(st);
        st = st + 1 == NS ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && iters < 0) sink[0] = *(uint32_t*)smem;
}

__global__ __launch_bounds__(256) void stream_vgpr(const char* __restrict__ base, uint32_t F, int iters, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t off = (uint32_t)(((uint64_t)blockIdx.x * 2654435761ull) % (F / 16384)) * 16384u;
    u32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < iters; t += 2) {
        u32x4 v[8];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int p = 0; p < 4; ++p) v[s * 4 + p] = *reinterpret_cast<const u32x4*>(base + off + wave * 4096u + p * 1024u + lane * 16u);
            off += 16384u;
            if (off >= F) off = 0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[threadIdx.x] = 1;
}

template <int NS>
__global__ __launch_bounds__(256) void stream_staged(const char* __restrict__ base, uint32_t F, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t off = (uint32_t)(((uint64_t)blockIdx.x * 2654435761ull) % (F / 16384)) * 16384u;
    u32x4 v[4];
    auto load = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const u32x4*>(base + off + wave * 4096u + p * 1024u + lane * 16u);
        off += 16384u;
        if (off >= F) off = 0;
    };
    load();
    int st = 0;
    for (int t = 0; t < iters; ++t) {
        u32x4 w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) w[p] = v[p];
        load();   // next slice in flight while this one is written
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(smem + st * 16384 + wave * 4096 + p * 1024 + lane * 16) = w[p];
        __syncthreads();
        st = st + 1 == NS ? 0 : st + 1;
    }
    if (threadIdx.x == 0 && iters < 0) sink[0] = *(uint32_t*)smem;
}

int main(int argc, char** argv) {
    const int ncu = 256;
    char* buf;
    uint32_t* sink;
    const size_t cap = 1ull << 30;
    (void)hipMalloc(&buf, cap);
    (void)hipMemset(buf, 1, cap);
    (void)hipMalloc(&sink, 4096);
    (void)hipFuncSetAttribute((const void*)stream_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)stream_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 512;   // 8 MiB per workgroup
    printf("%-44s %10s %10s %10s\n", "mode / footprint / workgroups per CU", "us", "TB/s", "B/clk/CU@2.4");
    for (size_t F : {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
        for (int wpc : {1, 2, 4}) {
            for (int mode = 0; mode < 5; ++mode) {
                const int grid = ncu * wpc;
                auto run = [&]() {
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(stream_dma<4>, dim3(grid), dim3(256), 65536, 0, buf, (uint32_t)F, iters, sink); break;
                        case 1: hipLaunchKernelGGL(stream_dma<8>, dim3(grid), dim3(256), wpc > 1 ? 65536 : 131072, 0, buf, (uint32_t)F, iters, sink); break;
                        case 2: hipLaunchKernelGGL(stream_vgpr, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)F, iters, sink); break;
                        case 3: hipLaunchKernelGGL(stream_staged<2>, dim3(grid), dim3(256), 32768, 0, buf, (uint32_t)F, iters, sink); break;
                        default: hipLaunchKernelGGL(stream_staged<2>, dim3(grid), dim3(256), 32768, 0, buf, (uint32_t)F, iters, sink); break;
                    }
                };
                if (mode == 4) continue;
                if (mode == 0 && wpc > 2) continue;   // the 4-stage ring needs 64 KiB: 2 per CU
                if (mode == 1 && wpc > 1) continue;   // 8-stage ring: 128 KiB, one per CU
                run();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < 5; ++r) run();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / 5, bytes = (double)grid * iters * 16384.0;
                const char* names[] = {"lds-dma ring4", "lds-dma ring8", "vgpr loads x8", "staged ds_write", ""};
                char label[96];
                snprintf(label, sizeof label, "%-16s F=%4zu MiB  %d wg/CU", names[mode], F >> 20, wpc);
                printf("%-44s %10.1f %10.2f %10.1f\n", label, us, bytes / us / 1e6, bytes / us / 1e6 * 1e12 / 256 / 2.4e9);
            }
        }
    }
    return 0;
}
