// Main-loop comparison of GemmTile workgroup shapes (4 waves 128 x 128 / 8 waves 256 x 128 / 8 waves 256 x 256) on the batched ViT
// GEMM shapes, plain bf16 store epilogue.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I splice_amd/csrc
//   -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/gemm_tiles.hip -o build/gemm_tiles   (run: build/gemm_tiles)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm.h"

int g_splice_prof_open = 0;
bool splice_prof_take(hipEvent_t*, hipEvent_t*) { return false; }

template <int BM, int BN, int NS, int WGM>
__global__ __launch_bounds__(128 * WGM) void tile_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K,
                                                         int gm, bf16_t* C, int ldc) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    grouped_tile(t, tiles_m, tiles_n, gm, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    GemmTile<BM, BN, true, false, WGM> tile;
    if (NS == 2) tile.run_glds(A, lda, B, ldb, M, N, K, m0, n0, smem, 0);
    else tile.template run_ring<(NS < 3 ? 3 : NS)>(A, lda, B, ldb, M, N, K, m0, n0, smem, 0);
    tile.for_each_cols(m0, n0, [&](int row, int col, const f32x4& v) {
        if (row < M && col < N) {
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(C + (size_t)row * ldc + col) = o;
        }
    });
}

template <int BM, int BN, int NS, int WGM>
float run(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int reps) {
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    int gm = 1;
    while ((gm + 1) * (gm + 1) * BM <= (tm * tn / 8 + 1) * BN && gm + 1 <= tm) ++gm;
    const size_t lds = (size_t)NS * (BM + BN) * 64 * 2;
    hipFuncSetAttribute((const void*)tile_kernel<BM, BN, NS, WGM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_kernel<BM, BN, NS, WGM>), dim3(tm * tn), dim3(128 * WGM), lds, 0, A, K, B, K, M, N, K, gm, C, N);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tile_kernel<BM, BN, NS, WGM>), dim3(tm * tn), dim3(128 * WGM), lds, 0, A, K, B, K, M, N, K, gm, C, N);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) return -1.f;
    return ms / reps * 1e3f;
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main() {
    const int shapes[][3] = {{12800, 3072, 768}, {12800, 2304, 768}, {12800, 768, 3072}, {12800, 768, 768}, {6400, 3072, 768}, {6400, 768, 3072}, {6274, 2304, 768}, {25600, 3072, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
        srand(1);
        for (auto& x : ha) x = f2bf((rand() % 2001 - 1000) * 1e-3f);
        for (auto& x : hb) x = f2bf((rand() % 2001 - 1000) * 1e-3f);
        bf16_t *A, *B, *C, *C2;
        hipMalloc(&A, ha.size() * 2); hipMalloc(&B, hb.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&C2, (size_t)M * N * 2);
        hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        const double fl = 2.0 * M * N * K;
        auto rep = [&](const char* name, float us, bf16_t* out) {
            std::vector<unsigned short> h1((size_t)M * N), h2((size_t)M * N);
            hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost);
            hipMemcpy(h2.data(), out, h2.size() * 2, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < h1.size(); ++i) bad += h1[i] != h2[i];
            printf("M %5d N %4d K %4d  %-22s %8.1f us  %7.1f TF  mismatches vs 128x128: %zu\n", M, N, K, name, us, fl / us * 1e-6, bad);
        };
        float t;
        t = run<128, 128, 2, 2>(A, B, C, M, N, K, 20); rep("128x128 4w NS2", t, C);
        t = run<128, 128, 3, 2>(A, B, C2, M, N, K, 20); rep("128x128 4w NS3", t, C2);
        t = run<256, 128, 2, 4>(A, B, C2, M, N, K, 20); rep("256x128 8w NS2", t, C2);
        t = run<256, 128, 3, 4>(A, B, C2, M, N, K, 20); rep("256x128 8w NS3", t, C2);
        t = run<256, 256, 2, 4>(A, B, C2, M, N, K, 20); rep("256x256 8w NS2", t, C2);
        t = run<128, 256, 2, 2>(A, B, C2, M, N, K, 20); rep("128x256 4w NS2", t, C2);
        hipFree(A); hipFree(B); hipFree(C); hipFree(C2);
    }
    return 0;
}
