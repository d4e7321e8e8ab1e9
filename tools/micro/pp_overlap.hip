// Does one wave's VALU / LDS stream overlap with its SIMD partner's MFMA stream?  (round 5, design input of attn_pp.h)
// 512-thread workgroups, one per CU: waves 0-3 play role A, waves 4-7 role B (w and w + 4 share a SIMD, tools/micro/simd_map.hip).
// Per iteration: A issues NA MFMAs (16x16x32 bf16, 8 independent accumulators), B issues its stream, then one s_barrier.
// Printed: cycles per iteration (s_memtime of wave 0) for A alone, B alone, both, and for the SAME work interleaved inside every wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NA>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[8], const bf16x8& a, const bf16x8& b) {
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
}
// B streams: 1 = 32 v_exp_f32 + 32 v_add_f32, 2 = 96 v_fma_f32, 3 = 24 LDS reads (8 b128 + 16 b64), 4 = 32 v_exp only, 5 = 64 v_add only
template <int MB>
__device__ __forceinline__ void b_block(float (&x)[32], float& sum, const float* lds, int lane) {
    if (MB == 1 || MB == 4) {
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
    }
    if (MB == 1 || MB == 5) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(x[i])); }
        if (MB == 5) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(sum)); }
        }
    }
    if (MB == 2) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(sum));
    }
    if (MB == 3) {
        typedef __attribute__((ext_vector_type(4))) float f4;
        typedef __attribute__((ext_vector_type(2))) float f2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f4 v = *reinterpret_cast<const f4*>(lds + i * 1024 + lane * 4);
            x[i * 4] += v[0]; x[i * 4 + 1] += v[1]; x[i * 4 + 2] += v[2]; x[i * 4 + 3] += v[3];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f2 v = *reinterpret_cast<const f2*>(lds + 8192 + i * 512 + lane * 2);
            x[i * 2] += v[0]; x[i * 2 + 1] += v[1];
        }
    }
}

// MODE: 0 = roles split over the wave groups (A on waves 0-3, B on waves 4-7), 1 = every wave runs A then B back to back (no split),
// 2 = every wave runs A and B INTERLEAVED (one B chunk after each group of MFMAs)
template <int NA, int MB, int MODE, bool RUN_A, bool RUN_B>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    __shared__ float lds[16384 + 64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(0.5f - lane * 0.002f); }
    float x[32], sum = 0.f;
    for (int i = 0; i < 32; ++i) x[i] = -1.0f - i * 0.01f - lane * 0.001f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            if (w < 4) { if (RUN_A) mfma_block<NA>(acc, a, b); }
            else { if (RUN_B) b_block<MB>(x, sum, lds, lane); }
        } else if (MODE == 1) {
            if (RUN_A) mfma_block<NA / 2>(acc, a, b);
            if (RUN_B && (it & 1) == (w >> 2)) b_block<MB>(x, sum, lds, lane);
        } else {
            // interleaved inside every wave: NA/2 MFMAs per wave per iteration, B every other iteration per wave group (same totals per SIMD)
            if (RUN_B && (it & 1) == (w >> 2)) {
                if (MB == 1) {
#pragma unroll
                    for (int i = 0; i < NA / 2; ++i) {
                        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
                        x[2 * i] = __builtin_amdgcn_exp2f(x[2 * i]);
                        x[2 * i + 1] = __builtin_amdgcn_exp2f(x[2 * i + 1]);
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(x[2 * i]));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(x[2 * i + 1]));
                    }
                } else { mfma_block<NA / 2>(acc, a, b); b_block<MB>(x, sum, lds, lane); }
            } else if (RUN_A) mfma_block<NA / 2>(acc, a, b);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = sum;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 32; ++i) r += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NA, int MB, int MODE, bool RA, bool RB>
double run(float* out, long long* cyc, int iters) {
    hipLaunchKernelGGL((k<NA, MB, MODE, RA, RB>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NA, MB, MODE, RA, RB>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) { long long mx = 0; for (int w = 0; w < 8; ++w) mx = h[i * 8 + w] > mx ? h[i * 8 + w] : mx; c += mx; }
    printf("  %7.1f us  %7.0f memtime ticks/iter", ms * 1e3, c / 256 / iters);
    return ms;
}

template <int MB>
void suite(const char* name, float* out, long long* cyc, int iters) {
    printf("B stream = %s\n", name);
    printf(" split roles, A only (32 MFMA)   :"); run<32, MB, 0, true, false>(out, cyc, iters); printf("\n");
    printf(" split roles, B only             :"); run<32, MB, 0, false, true>(out, cyc, iters); printf("\n");
    printf(" split roles, A || B             :"); run<32, MB, 0, true, true>(out, cyc, iters); printf("\n");
    printf(" every wave A then B (halved)    :"); run<32, MB, 1, true, true>(out, cyc, iters); printf("\n");
    printf(" every wave A, B interleaved     :"); run<32, MB, 2, true, true>(out, cyc, iters); printf("\n");
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 4000;
    suite<1>("32 v_exp + 32 v_add", out, cyc, iters);
    suite<4>("32 v_exp", out, cyc, iters);
    suite<5>("64 v_add", out, cyc, iters);
    suite<2>("96 v_fma", out, cyc, iters);
    suite<3>("24 LDS reads (16 KB)", out, cyc, iters);
    return 0;
}
