// Does an XCD's L2 keep lines across kernel launches of one stream?  Workgroup b (XCD b & 7) reads the 1 MiB region of its XCD;
// the same kernel is launched four times back to back.  rocprofv3 --pmc FETCH_SIZE: a later launch that fetches ~nothing found
// its lines still in the L2 (run: tools/micro/run_l2_persist.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_region(const char* base, unsigned* sink) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;            // 32 workgroups per XCD, 32 KiB each
    const char* p = base + ((size_t)xcd << 20) + ((size_t)idx << 15);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= *reinterpret_cast<const u32x4*>(p + i * 4096 + threadIdx.x * 16);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void write_region(char* base) {   // the producer case: lines WRITTEN by the previous kernel (same XCD mapping)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    char* p = base + ((size_t)xcd << 20) + ((size_t)idx << 15);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(p + i * 4096 + threadIdx.x * 16) = u32x4{1u, 2u, 3u, (unsigned)i};
}
int main(int argc, char** argv) {
    char* buf; unsigned* sink;
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0 hipMalloc (coarse-grained), 1 fine-grained, 2 uncached
    if (mode == 0) (void)hipMalloc(&buf, 64 << 20);
    else (void)hipExtMallocWithFlags((void**)&buf, 64 << 20, mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    (void)hipMemset(buf, 1, 64 << 20); (void)hipMalloc(&sink, 4096);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(read_region, dim3(256), dim3(256), 0, 0, buf, sink);
    hipLaunchKernelGGL(write_region, dim3(256), dim3(256), 0, 0, buf + (32 << 20));
    hipLaunchKernelGGL(read_region, dim3(256), dim3(256), 0, 0, buf + (32 << 20), sink);
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
