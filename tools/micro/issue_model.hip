// gfx950 issue model for the attention kernels (round 5): what does a VALU instruction cost next to MFMAs?
//  (1) VALU streams alone: cycles per instruction per SIMD with 1 / 2 / 4 waves per SIMD issuing independent ops;
//  (2) one MFMA followed by F independent filler instructions, repeated: cycles per MFMA for F = 0 .. 8, for both MFMA shapes,
//      fillers = v_fma_f32 / v_exp_f32 / v_cvt_pk_bf16_f32 / ds_read_b128, with 1 and 2 waves per SIMD.
// Cycles = s_memtime span of the SLOWEST wave of a workgroup (one tick per shader cycle), 256 workgroups, one per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { F_FMA = 0, F_EXP = 1, F_CVT = 2, F_LDS = 3, F_ADD = 4, F_PKADD = 5, F_MOV64 = 6 };

template <int T>
__device__ __forceinline__ void filler(float& x, float& y, unsigned& u, const float* lds, f32x4& ld) {
    if (T == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (T == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (T == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
    if (T == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (T == F_LDS) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float*)lds));
}

template <int T, int NOPS>
__global__ __launch_bounds__(1024) void valu_stream(float* out, long long* cyc, int iters) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float x[16], y = 1.0001f;
    unsigned u = 0;
    f32x4 ld = {0, 0, 0, 0};
    double d[8];
    for (int i = 0; i < 16; ++i) x[i] = -1.0f - 0.01f * i - threadIdx.x * 1e-4f;
    for (int i = 0; i < 8; ++i) d[i] = i + threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NOPS; ++i) {
            if (T == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
            else if (T == F_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i & 7]) : "v"(d[(i + 4) & 7]));
            else filler<T>(x[i & 15], y, u, lds + (threadIdx.x & 63) * 4, ld);
        }
        if (T == F_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = y + u + ld[0];
    for (int i = 0; i < 16; ++i) r += x[i];
    for (int i = 0; i < 8; ++i) r += (float)d[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int BIG, int T, int F>   // BIG: 0 = 16x16x32, 1 = 32x32x16
__global__ __launch_bounds__(512) void mfma_fill(float* out, long long* cyc, int iters) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 acc[8];
    f32x16 big[4];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f - threadIdx.x * 0.002f); }
    float x[16], y = 1.0001f;
    unsigned u = 0;
    f32x4 ld = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) x[i] = -1.0f - 0.01f * i - threadIdx.x * 1e-4f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (BIG) big[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[m & 3], 0, 0, 0);
            else acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < F; ++f) filler<T>(x[(m * F + f) & 15], y, u, lds + (threadIdx.x & 63) * 4, ld);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (T == F_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = y + u + ld[0];
    for (int i = 0; i < 16; ++i) r += x[i];
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    for (int i = 0; i < 4; ++i) r += big[i][0] + big[i][15];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

static float* g_out; static long long* g_cyc;
template <class K>
double cycles(K kern, int threads, int iters) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, g_out, g_cyc, iters);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, g_out, g_cyc, iters);
    hipDeviceSynchronize();
    static long long h[256 * 16]; hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) {   // slowest wave of each workgroup (arbitration favours the oldest wave: wave 0 alone says nothing)
        long long mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[i * 16 + w] > mx ? h[i * 16 + w] : mx;
        c += mx;
    }
    return c / 256 / iters;
}
template <int T>
void valu_row(const char* name) {
    const int iters = 2000;
    printf("%-22s", name);
    for (int waves : {4, 8, 16}) printf("  %d w/SIMD: %5.2f cyc/instr/SIMD", waves / 4, cycles(valu_stream<T, 64>, waves * 64, iters) / 64 / (waves / 4));
    printf("\n");
}
template <int BIG, int T>
void fill_row(const char* name) {
    const int iters = 1000;
    for (int threads : {256, 512}) {
        printf("%-28s %d w/SIMD, cyc per MFMA (per wave) F=0..8:", name, threads / 256);
        printf(" %5.1f", cycles(mfma_fill<BIG, T, 0>, threads, iters) / 16);
        printf(" %5.1f", cycles(mfma_fill<BIG, T, 1>, threads, iters) / 16);
        printf(" %5.1f", cycles(mfma_fill<BIG, T, 2>, threads, iters) / 16);
        printf(" %5.1f", cycles(mfma_fill<BIG, T, 3>, threads, iters) / 16);
        printf(" %5.1f", cycles(mfma_fill<BIG, T, 4>, threads, iters) / 16);
        printf(" %5.1f", cycles(mfma_fill<BIG, T, 6>, threads, iters) / 16);
        printf(" %5.1f", cycles(mfma_fill<BIG, T, 8>, threads, iters) / 16);
        printf("\n");
    }
}
int main() {
    hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_cyc, 256 * 16 * 8);
    printf("== VALU streams alone (64 independent-ish ops per iteration; cycles per instruction per SIMD)\n");
    valu_row<F_FMA>("v_fma_f32");
    valu_row<F_ADD>("v_add_f32");
    valu_row<F_EXP>("v_exp_f32");
    valu_row<F_CVT>("v_cvt_pk_bf16_f32");
    valu_row<F_PKADD>("v_pk_add_f32");
    valu_row<F_MOV64>("v_mov_b64");
    valu_row<F_LDS>("ds_read_b128");
    printf("== MFMA + F fillers after each MFMA (cycles per MFMA of a wave, slowest wave; with 2 w/SIMD the SIMD retires two MFMAs in that time)\n");
    fill_row<0, F_FMA>("16x16x32 + v_fma_f32");
    fill_row<0, F_EXP>("16x16x32 + v_exp_f32");
    fill_row<0, F_CVT>("16x16x32 + v_cvt_pk");
    fill_row<0, F_LDS>("16x16x32 + ds_read_b128");
    fill_row<1, F_FMA>("32x32x16 + v_fma_f32");
    fill_row<1, F_EXP>("32x32x16 + v_exp_f32");
    fill_row<1, F_CVT>("32x32x16 + v_cvt_pk");
    fill_row<1, F_LDS>("32x32x16 + ds_read_b128");
    return 0;
}
