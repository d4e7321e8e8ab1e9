// Which dependency pattern of the attention loop breaks the MFMA / VALU overlap that tools/micro/issue_model.hip measures for independent fillers?
// One wave per SIMD (256-thread workgroups, one per CU).  Per iteration 16 x v_mfma_f32_32x32x16_bf16 in two accumulator chains; after each MFMA:
//   mode 0: nothing                                   mode 1: 2 v_exp + 2 v_add on PRIVATE registers (the issue_model case)
//   mode 2: 2 v_exp + 2 v_add whose SOURCES are registers of an accumulator written by MFMAs of the previous iteration (softmax on S)
//   mode 3: mode 2 + 1 v_cvt_pk whose result is the B operand of a LATER MFMA (P -> PV)
//   mode 4: mode 1 + 1.5 LDS reads whose results are A operands of later MFMAs (K / V fragments)
//   mode 5: all of it (2 + 3 + 4)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0x3c003c00u + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[2], sprev[2];
    for (int j = 0; j < 16; ++j) { acc[0][j] = acc[1][j] = 0.f; sprev[0][j] = -1.f - j * 0.01f; sprev[1][j] = -2.f - j * 0.01f; }
    u32x4 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = u32x4{0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + i};
    for (int i = 0; i < 4; ++i) b[i] = u32x4{0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u + i};
    float x[32], sum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 32; ++i) x[i] = -1.0f - i * 0.01f - lane * 0.001f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m & 7]), __builtin_bit_cast(bf16x8, b[m & 3]), acc[m & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1 || MODE == 4) {
                x[2 * m] = __builtin_amdgcn_exp2f(x[2 * m]); x[2 * m + 1] = __builtin_amdgcn_exp2f(x[2 * m + 1]);
                sum[0] += x[2 * m]; sum[1] += x[2 * m + 1];
                asm volatile("" : "+v"(sum[0]), "+v"(sum[1]));
            }
            if (MODE == 2 || MODE == 3 || MODE == 5) {
                const float e0 = __builtin_amdgcn_exp2f(sprev[m >> 3][(2 * m) & 15]), e1 = __builtin_amdgcn_exp2f(sprev[m >> 3][(2 * m + 1) & 15]);
                sum[0] += e0; sum[1] += e1;
                x[2 * m] = e0; x[2 * m + 1] = e1;
                asm volatile("" : "+v"(sum[0]), "+v"(sum[1]));
                if (MODE == 3 || MODE == 5) {
                    unsigned pk;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(e0), "v"(e1));
                    b[(m + 2) & 3][m & 3] = pk;   // B operand of the MFMA two slots on
                }
            }
            if (MODE == 4 || MODE == 5) {
                a[(m + 4) & 7] = *reinterpret_cast<const u32x4*>(lds + ((lane * 4 + m * 256) & 8191));
                if (m & 1) a[(m + 5) & 7][0] = lds[(lane + m * 64) & 8191];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the accumulators of this iteration are the scores of the next
        sprev[0] = acc[0]; sprev[1] = acc[1];
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[0][j] = -1.f; acc[1][j] = -2.f; }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = sum[0] + sum[1] + sum[2] + sum[3];
    for (int i = 0; i < 32; ++i) r += x[i];
    for (int j = 0; j < 16; ++j) r += acc[0][j] + acc[1][j] + sprev[0][j] + sprev[1][j];
    out[blockIdx.x * 512 + threadIdx.x] = r + a[0][0] + b[0][0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    printf("%-66s", name);
    for (int threads : {256, 512}) {   // one / two waves per SIMD; the SLOWEST wave of a workgroup counts
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        static long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double c = 0;
        for (int i = 0; i < 256; ++i) { long long mx = 0; for (int w = 0; w < threads / 64; ++w) mx = h[i * 8 + w] > mx ? h[i * 8 + w] : mx; c += mx; }
        printf("  %d w/SIMD: %6.1f cycles per MFMA per SIMD", threads / 256, c / 256 / iters / 16 / (threads / 256));
    }
    printf("\n");
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8);
    run<0>("0 MFMA only (two chains)", out, cyc);
    run<1>("1 + 2 exp + 2 add, private registers", out, cyc);
    run<2>("2 + 2 exp + 2 add on the PREVIOUS iteration's accumulators", out, cyc);
    run<3>("3 = 2 + cvt_pk into the B operand of the MFMA two slots on", out, cyc);
    run<4>("4 = 1 + 1.5 LDS reads into A operands of later MFMAs", out, cyc);
    run<5>("5 = 2 + 3 + 4", out, cyc);
    return 0;
}
