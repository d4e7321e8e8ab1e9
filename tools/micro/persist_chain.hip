// VERDICT r3 #2c: "measure, don't price" a small-grid persistent kernel against a chain of launches.
// Model of the generator's deep scales (planes of 28 x 28 and below, <= 104 workgroups per kernel, every stage a global
// dependency: a train-mode BatchNorm needs the whole plane of the convolution in front of it): S dependent stages on G
// workgroups; in every stage a workgroup reads FOUR 4-KB chunks that other workgroups (other CUs, other XCDs) wrote in the
// previous stage, does a little arithmetic and writes its own 4-KB chunk.
//   A  S launches, replayed as one hipGraph (what the step does today)
//   B  ONE persistent launch, a grid barrier per stage: plain stores -> agent-scope release fence -> arrive on a monotonic
//      counter -> relaxed polling -> agent-scope acquire fence (CDNA4 guide, Guideline 16, counter form)
//   C  the same with write-through (sc0 sc1) stores and sc1 loads instead of the two fences (Guideline 16, R1)
// Outputs are compared bit for bit with A.  Always run under `timeout` (a lost arrival would spin forever).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/persist_chain.hip -o build/persist_chain && timeout 120 build/persist_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f4;
constexpr int CHUNK = 1024;   // floats per workgroup and stage (4 KB): 256 threads x float4

__device__ __forceinline__ f4 ld_plain(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void ld4_sc1(const float* p0, const float* p1, const float* p2, const float* p3, f4 (&v)[4]) {   // four loads in flight, one wait
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void st_plain(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
__device__ __forceinline__ void st_wt(float* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

template <bool SC1>
__device__ __forceinline__ void stage_body(const float* in, float* out, int w, int G, int stage) {
    const int t = threadIdx.x;
    const int src[4] = {w, (w + 1) % G, (w + G / 2) % G, (w * 7 + 3 + stage) % G};
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 v[4];
    if (SC1) ld4_sc1(in + (size_t)src[0] * CHUNK + t * 4, in + (size_t)src[1] * CHUNK + t * 4, in + (size_t)src[2] * CHUNK + t * 4, in + (size_t)src[3] * CHUNK + t * 4, v);
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ld_plain(in + (size_t)src[k] * CHUNK + t * 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k] * (0.25f + 0.001f * k);
    acc = acc * 0.999f + 0.0005f;
    float* q = out + (size_t)w * CHUNK + t * 4;
    if (SC1) st_wt(q, acc); else st_plain(q, acc);
}

__global__ __launch_bounds__(256) void stage_kernel(const float* in, float* out, int G, int stage) { stage_body<false>(in, out, blockIdx.x, G, stage); }

template <bool SC1>
__global__ __launch_bounds__(256) void persistent_kernel(float* a, float* b, int G, int S, unsigned* counter) {
    const int w = blockIdx.x;
    for (int s = 0; s < S; ++s) {
        stage_body<SC1>(s & 1 ? b : a, s & 1 ? a : b, w, G, s);
        // ---- grid barrier
        if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
        __syncthreads();
        if (threadIdx.x == 0) {
            if (!SC1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(s + 1) * (unsigned)G;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000u) break;   // bounded: a bug shows as a wrong result, not as a hung GPU
            }
            if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

int main() {
    const int S = 32, REPS = 50;
    const int grids[] = {16, 32, 64, 104, 256};
    hipStream_t st;
    hipStreamCreate(&st);
    for (int G : grids) {
        const size_t n = (size_t)G * CHUNK;
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
        float *a, *b;
        unsigned* cnt;
        hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&cnt, 64);
        auto reset = [&] { hipMemcpyAsync(a, h.data(), n * 4, hipMemcpyHostToDevice, st); hipMemsetAsync(b, 0, n * 4, st); hipMemsetAsync(cnt, 0, 64, st); hipStreamSynchronize(st); };
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        // ---- A: graph of S launches
        reset();
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int s = 0; s < S; ++s) hipLaunchKernelGGL(stage_kernel, dim3(G), dim3(256), 0, st, s & 1 ? b : a, s & 1 ? a : b, G, s);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        std::vector<float> ref(n), got(n);
        hipMemcpy(ref.data(), a, n * 4, hipMemcpyDeviceToHost);   // S even: the result is in a
        hipEventRecord(e0, st);
        for (int r = 0; r < REPS; ++r) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float msA; hipEventElapsedTime(&msA, e0, e1);
        // ---- B, C: persistent
        float msP[2]; size_t bad[2];
        for (int v = 0; v < 2; ++v) {
            reset();
            auto launch = [&] {
                hipMemsetAsync(cnt, 0, 4, st);
                if (v == 0) hipLaunchKernelGGL(persistent_kernel<false>, dim3(G), dim3(256), 0, st, a, b, G, S, cnt);
                else hipLaunchKernelGGL(persistent_kernel<true>, dim3(G), dim3(256), 0, st, a, b, G, S, cnt);
            };
            launch(); hipStreamSynchronize(st);
            hipMemcpy(got.data(), a, n * 4, hipMemcpyDeviceToHost);
            bad[v] = 0;
            for (size_t i = 0; i < n; ++i) bad[v] += got[i] != ref[i];
            hipEventRecord(e0, st);
            for (int r = 0; r < REPS; ++r) launch();   // (each repetition continues from the previous result: timing only)
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            hipEventElapsedTime(&msP[v], e0, e1);
        }
        printf("G %3d workgroups, %d stages: graph of launches %6.2f us/stage | persistent, fences %6.2f us/stage (mismatches %zu) | persistent, sc1 %6.2f us/stage (mismatches %zu)\n",
               G, S, msA / REPS / S * 1e3f, msP[0] / REPS / S * 1e3f, bad[0], msP[1] / REPS / S * 1e3f, bad[1]);
        fflush(stdout);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
        hipFree(a); hipFree(b); hipFree(cnt);
    }
    return 0;
}
