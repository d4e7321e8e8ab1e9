// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "splice_hip.h"

typedef uint16_t bf16_t;  // raw bf16 storage

// ---- every kernel launch of the library goes through SPLICE_LAUNCH.  While bench.py's roofline leg has a kernel family
// armed (splice_prof_begin) and one of its host calls is open (SpliceProfScope), the launch goes out as hipExtLaunchKernelGGL
// with a start / stop event pair of its own: the pair carries the kernel's begin and end time stamps (what rocprofv3's
// kernel trace reports), not the cost of two event records queued around it.  Otherwise: a plain launch.
bool splice_prof_take(hipEvent_t* start, hipEvent_t* stop, const char* kernel_name);
// algorithmic work of the NEXT launch that takes an event pair (conv launchers: FLOPs and bytes of the layer); consumed by that launch
void splice_prof_note(double flops, double bytes);
extern int g_splice_prof_open;   // > 0: a scope of the armed family is open (prof.hip)
struct SpliceProfScope {
    bool on;
    explicit SpliceProfScope(int which);
    ~SpliceProfScope();
};
// Scratch builds only (make DEV=1, -DSPLICE_DEV_SWITCHES; bench.py refuses such a library): a launch inside a SPLICE_DEV_REGION(id) whose bit is set
// in the environment mask SPLICE_DEV_SKIP is NOT issued once two steps have run (buffers then hold finite data of those steps) -- TIMING ONLY, the
// results are garbage: what the step loses when a region's launches vanish is that region's cost on the critical path (tools/critical_path_map.sh).
#ifdef SPLICE_DEV_SWITCHES
extern unsigned long long g_splice_dev_skip;   // prof.hip
extern int g_splice_dev_region, g_splice_dev_steps;
struct SpliceDevRegion {
    int prev;
    explicit SpliceDevRegion(int id) : prev(g_splice_dev_region) { g_splice_dev_region = id; }
    ~SpliceDevRegion() { g_splice_dev_region = prev; }
};
#define SPLICE_DEV_REGION(id) SpliceDevRegion splice_dev_region_##id(id)
#define SPLICE_DEV_SKIPPED() (g_splice_dev_steps > 2 && ((g_splice_dev_skip >> g_splice_dev_region) & 1ull))
#else
#define SPLICE_DEV_REGION(id) do { } while (0)
#define SPLICE_DEV_SKIPPED() false
#endif
#define SPLICE_LAUNCH(kernel, grid, block, lds, stream, ...)                                                       \
    do {                                                                                                           \
        if (SPLICE_DEV_SKIPPED()) break;                                                                           \
        hipEvent_t pa_, pb_;                                                                                       \
        if (g_splice_prof_open > 0 && splice_prof_take(&pa_, &pb_, #kernel))                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, pa_, pb_, 0, __VA_ARGS__);                     \
        else                                                                                                       \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                     \
    } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Write-through stores (sc0 sc1) for kernel OUTPUTS that leave as FULL cache lines (the staged GEMM tiles, LayerNorm rows).
// A kernel's dirty L2 lines are written back when it ends, before the next kernel of the stream may start; with several
// hundred short launches per step that drain sits on the critical chain.  Written through, the lines go to memory while the
// kernel is still computing.  Measured (tools/ab_so.sh, same box, alternating): GEMM + LayerNorm +1.1 % steps/s at one pair
// per GPU, +0.7 % at eight.  NOT for partial-line stores: the attention outputs (8 bytes per lane, one row per lane) lose
// 1.7 % this way, the generator's scalar stores 1.5 %; `nt` on the GEMM tiles loses 2 %.  WT_STORES 0 = plain stores.
#ifndef WT_STORES
#define WT_STORES 1
#endif
template <class V>
__device__ __forceinline__ void st_out(V* p, const V& v) {
    static_assert(sizeof(V) == 16 || sizeof(V) == 8 || sizeof(V) == 4, "4-, 8- or 16-byte values");
#if WT_STORES
    if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(__builtin_bit_cast(u32x2, v)) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(__builtin_bit_cast(unsigned int, v)) : "memory");
#else
    *p = v;
#endif
}


// fp32 -> bf16, round-to-nearest-even, on gfx950's hardware converter (v_cvt_pk_bf16_f32: one instruction per
// PAIR; the bit-twiddling form costs ~6 VALU ops per element and dominated the attention / epilogue VALU time)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-form GELU (torch.nn.GELU default) x * Phi(x) and its derivative Phi(x) + x phi(x), WITHOUT transcendentals: both
// Phi(x) - 1/2 and gelu'(x) - 1/2 are odd functions, fitted on [-4, 4] by odd polynomials in x (least squares on Chebyshev
// nodes; max error in fp32 Horner form 6e-6 / 5e-5 absolute -- the outputs are rounded to bf16, eps 4e-3) and saturated
// outside (Phi(4) = 1 - 3e-5).  10 / 11 full-rate FMAs instead of the Abramowitz-Stegun form's rcp + exp + 14 ops: the GELU
// epilogues of fc1 / fc2^T cost 21 % / 33 % of those GEMMs at M = 1600 (tools/gelu_cost.py).
__device__ __forceinline__ float gelu_cdf_poly(float xc) {   // Phi(xc) for xc in [-4, 4]
    const float u = xc * xc;
    float p = 7.804400182e-11f;
    p = __builtin_fmaf(p, u, -6.827484800e-09f);
    p = __builtin_fmaf(p, u, 2.666929504e-07f);
    p = __builtin_fmaf(p, u, -6.221946023e-06f);
    p = __builtin_fmaf(p, u, 9.829076589e-05f);
    p = __builtin_fmaf(p, u, -1.130963792e-03f);
    p = __builtin_fmaf(p, u, 9.869961999e-03f);
    p = __builtin_fmaf(p, u, -6.640202552e-02f);
    p = __builtin_fmaf(p, u, 3.989198506e-01f);
    return __builtin_fmaf(xc, p, 0.5f);
}
// Outside [-4, 4] the fitted values at the clamp stand in (Phi(4) = 1 - 2.8e-5, Phi(-4) = 2.8e-5: the fit meets the true function
// there to 4e-6), and the negative tail multiplies the CLAMPED argument: gelu(x <= -4) = -4 Phi(-4) = -1.1e-4 (true: -1.3e-4 ... 0)
// -- errors bounded by 1.3e-4 absolute / 3e-5 relative, no compare + select per element (round 4: the saturating selects were
// 4 of the 16 vector instructions per element of the fc1 epilogue).
__device__ __forceinline__ float gelu_f(float x) {
    const float m = fmaxf(x, -4.0f);
    return m * gelu_cdf_poly(fminf(m, 4.0f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float xc = fminf(fmaxf(x, -4.0f), 4.0f), u = xc * xc;
    float p = -5.066447262e-11f;
    p = __builtin_fmaf(p, u, 4.789254326e-09f);
    p = __builtin_fmaf(p, u, -2.026289394e-07f);
    p = __builtin_fmaf(p, u, 5.107016932e-06f);
    p = __builtin_fmaf(p, u, -8.623141184e-05f);
    p = __builtin_fmaf(p, u, 1.038558665e-03f);
    p = __builtin_fmaf(p, u, -9.190188721e-03f);
    p = __builtin_fmaf(p, u, 5.937872082e-02f);
    p = __builtin_fmaf(p, u, -2.656380534e-01f);
    p = __builtin_fmaf(p, u, 7.978171706e-01f);
    const float d = __builtin_fmaf(xc, p, 0.5f);
    // Outside [-4, 4] the derivative of gelu_f AS IMPLEMENTED: 0 below -4 (the forward is constant there), Phi(4) above 4 (the forward is x * Phi(4)).
    // Round 4 returned the fit's values at the clamp (-5e-4 / 1 + 5e-4): a small systematic gradient on every strongly negative fc1
    // pre-activation; the two selects sit in the backward epilogue only, not in the fc1-forward hot spot the diet was for.
    return x < -4.0f ? 0.0f : (x > 4.0f ? 0.99997f : d);
}

// D = A(16xK32) * B(K32x16) + C on one wave.  Operand layout (gfx950, 16x16x32 bf16):
//   A: lane l holds A[i = l&15][k = (l>>4)*8 .. +7]      (8 contiguous k)
//   B: lane l holds B[k = (l>>4)*8 .. +7][n = l&15]
//   C/D: lane l, reg r holds C[row = (l>>4)*4 + r][col = l&15]
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// XCD-aware bijective remap of a linear workgroup id: consecutive logical tiles land
// on the same XCD (block b is dispatched to XCD b % 8), so neighbours share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Tile order inside each XCD's contiguous chunk: super-rows of `gm` tile-rows walked column-major, so the tiles an
// XCD runs concurrently form a near-square block of the output (fewest distinct A and B panels per private L2).
__device__ __forceinline__ void grouped_tile(int t, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
    const int width = gm * tiles_n;
    const int gid = t / width, first = gid * gm;
    const int gsz = (tiles_m - first) < gm ? (tiles_m - first) : gm;
    const int r = t - gid * width;
    tm = first + r % gsz;
    tn = r / gsz;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
