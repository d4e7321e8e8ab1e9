// The 256 x 256 8-phase GEMM main loop (splice_amd/csrc/gemm8p.h; persistent workgroups, one continuous K-tile stream) against the
// 128 x 128 one-barrier tile of gemm.h on the batched ViT shapes: bit comparison + timing, plain bf16 store epilogue.  Also timed:
// the 8-phase kernel WITHOUT its output stores (what the stores cost: they are not overlapped -- vmcnt retires in order, so the
// first counted DMA wait behind an epilogue drains its stores) and with staggered workgroup starts (no gain).
// Results: profiles/r04_gemm8p_micro.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I splice_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/gemm8p.hip -o build/gemm8p
//   timeout 120 build/gemm8p        (a mis-counted barrier hangs the workgroup: always run under timeout)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm8p.h"

int g_splice_prof_open = 0;
bool splice_prof_take(hipEvent_t*, hipEvent_t*) { return false; }

template <int BM, int BN, int NS>
__global__ __launch_bounds__(256) void tile_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K,
                                                   int gm, bf16_t* C, int ldc) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    grouped_tile(t, tiles_m, tiles_n, gm, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    GemmTile<BM, BN, true, false, 2> tile;
    tile.run_glds(A, lda, B, ldb, M, N, K, m0, n0, smem, 0);
    tile.for_each_cols(m0, n0, [&](int row, int col, const f32x4& v) {
        if (row < M && col < N) *reinterpret_cast<uint2*>(C + (size_t)row * ldc + col) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    });
}

// plain bf16 store of the accumulators: two adjacent 16-column fragments of a row leave as one 16-byte store per lane
// (v_permlane16_swap: lanes 16 apart exchange their 4-column groups, so a lane ends up with 8 consecutive columns)
__device__ __forceinline__ void store_tile_bf16(const Gemm8p& g, bf16_t* C, int ldc, int M, int N, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const int grp = lane >> 4;
    const int c8 = ((grp & 1) << 4) | ((grp >> 1) << 3);   // {0, 16, 8, 24}[grp]
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const f32x4 v0 = g.acc[mh * 4 + i][nh * 2 + 0], v1 = g.acc[mh * 4 + i][nh * 2 + 1];
                uint2 a = uint2{pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3])}, b = uint2{pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
                auto rx = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
                auto ry = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
                const int row = m0 + mh * 128 + wr * 64 + i * 16 + (lane & 15), col = n0 + nh * 128 + wc * 32 + c8;
                if (row < M && col + 7 < N) *reinterpret_cast<uint4*>(C + (size_t)row * ldc + col) = uint4{rx[0], ry[0], rx[1], ry[1]};
            }
}

template <int MODE>   // 0: store the tiles; 1: timing experiment -- no output stores (accumulators kept alive)
__global__ __launch_bounds__(512) void k8p_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K,
                                                  int gm, bf16_t* C, int ldc) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int count = (ntiles - (int)blockIdx.x + G - 1) / G;   // tiles blockIdx.x, blockIdx.x + G, ...
    if (MODE == 2) {
        // workgroups with one tile fewer than the busiest ones start late, spread over one tile period: their output-store bursts
        // (and, in order behind them, their counted DMA waits) no longer coincide with everybody else's
        const int maxc = (ntiles + G - 1) / G, nfull = ntiles - (maxc - 1) * G;   // workgroups 0 .. nfull-1 have maxc tiles
        if (count < maxc && G > nfull) {
            const long long period = (long long)(K / 64) * 3600;   // ~1.5 us per K tile at 2.4 GHz
            const long long wait = period * ((int)blockIdx.x - nfull) / (G - nfull);
            const long long t0 = clock64();
            while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
        }
    }
    Gemm8p g;
    g.run_tiles(A, lda, B, ldb, M, N, K, count,
                [&](int r, int& m0, int& n0) {
                    const int t = xcd_remap(blockIdx.x + r * G, ntiles);   // G % 8 == 0: every tile of this workgroup maps to its own XCD's chunk
                    int tm, tn;
                    grouped_tile(t, tiles_m, tiles_n, gm, tm, tn);
                    m0 = tm * 256; n0 = tn * 256;
                },
                [&](int m0, int n0) {
                    if (MODE != 1) store_tile_bf16(g, C, ldc, M, N, m0, n0);
                    else {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(g.acc[i][j]));
                    }
                }, smem);
}

static int group_height(int tm, int tn, int BM, int BN) {
    int gm = 1;
    while ((gm + 1) * (gm + 1) * BM <= (tm * tn / 8 + 1) * BN && gm + 1 <= tm) ++gm;
    return gm;
}
template <class F>
static float time_us(F&& launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) return -1.f;
    return ms / reps * 1e3f;
}
static unsigned short f2bf_h(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int shapes[][3] = {{512, 512, 256}, {704, 520, 128}, {3000, 2304, 768}, {12800, 3072, 768}, {12800, 2304, 768}, {6400, 3072, 768}, {12800, 768, 3072}, {6400, 2304, 768}, {25600, 3072, 768},
                             {4096, 4096, 4096}};
    hipFuncSetAttribute((const void*)k8p_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, Gemm8p::LDS_BYTES);
    hipFuncSetAttribute((const void*)k8p_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, Gemm8p::LDS_BYTES);
    hipFuncSetAttribute((const void*)k8p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, Gemm8p::LDS_BYTES);
    hipFuncSetAttribute((const void*)tile_kernel<128, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 64 * 2);
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
        srand(1);
        for (auto& x : ha) x = f2bf_h((rand() % 2001 - 1000) * 1e-3f);
        for (auto& x : hb) x = f2bf_h((rand() % 2001 - 1000) * 1e-3f);
        bf16_t *A, *B, *C, *C2;
        hipMalloc(&A, ha.size() * 2); hipMalloc(&B, hb.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&C2, (size_t)M * N * 2);
        hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        hipMemset(C2, 0xFF, (size_t)M * N * 2);
        const double fl = 2.0 * M * N * K;
        const int tm1 = (M + 127) / 128, tn1 = (N + 127) / 128, tm2 = (M + 255) / 256, tn2 = (N + 255) / 256;
        const int gm1 = group_height(tm1, tn1, 128, 128), gm2 = group_height(tm2, tn2, 256, 256);
        const float t1 = time_us([&] { hipLaunchKernelGGL((tile_kernel<128, 128, 2>), dim3(tm1 * tn1), dim3(256), 2 * 256 * 64 * 2, 0, A, K, B, K, M, N, K, gm1, C, N); }, 20);
        const float t3 = time_us([&] { hipLaunchKernelGGL(k8p_kernel<1>, dim3(tm2 * tn2 < 256 ? tm2 * tn2 : 256), dim3(512), Gemm8p::LDS_BYTES, 0, A, K, B, K, M, N, K, gm2, C2, N); }, 20);
        const float t4 = time_us([&] { hipLaunchKernelGGL(k8p_kernel<2>, dim3(tm2 * tn2 < 256 ? tm2 * tn2 : 256), dim3(512), Gemm8p::LDS_BYTES, 0, A, K, B, K, M, N, K, gm2, C2, N); }, 20);
        const float t2 = time_us([&] { hipLaunchKernelGGL(k8p_kernel<0>, dim3(tm2 * tn2 < 256 ? tm2 * tn2 : 256), dim3(512), Gemm8p::LDS_BYTES, 0, A, K, B, K, M, N, K, gm2, C2, N); }, 20);
        std::vector<unsigned short> h1((size_t)M * N), h2((size_t)M * N);
        hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), C2, h2.size() * 2, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < h1.size(); ++i) bad += h1[i] != h2[i];
        printf("M %5d N %4d K %4d  128x128: %8.1f us %7.1f TF | 256x256 8-phase: %8.1f us %7.1f TF (%d tiles; without stores %.1f us; staggered starts %.1f us)  mismatches %zu\n", M, N, K, t1, fl / t1 * 1e-6, t2,
               fl / t2 * 1e-6, tm2 * tn2, t3, t4, bad);
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(C); hipFree(C2);
    }
    return 0;
}
