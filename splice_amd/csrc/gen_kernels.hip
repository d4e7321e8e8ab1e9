// Kernels of the skip-U-Net generator (K13-K18 of SURVEY.md; models/unet/skip.py:42-102,
// models/unet/common.py:11-124): fp32 NCHW, convolutions as implicit GEMM on the exact-f32
// matrix cores (v_mfma_f32_16x16x4_f32: bit-equal to an fmaf chain, so generator parity with
// the fp32 oracle is rounding-order only), train-mode BatchNorm with per-instance statistics
// (the reference calls netG once per image with batch 1), LeakyReLU(0.2), bilinear x2
// up-sampling written straight into the concat buffer, sigmoid head.  Every reduction runs
// in a fixed order (no float atomics): replicas are bit-reproducible.
#include "gen_kernels.h"
#include <atomic>
#ifndef CONV_KB
#define CONV_KB 6   // k steps per LDS fragment batch of conv_igemm_body at one output-channel fragment per workgroup (0: the whole tile at once); 3 at 2 or 4 fragments
#endif
#include <cstdlib>

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    // A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][n = l&15]
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16-byte accesses at 4-byte alignment (gfx950's global accesses need dword alignment only): runs of 4 consecutive pixels of odd-sized planes
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// run k of this thread inside the segment [lo, hi): pixels i .. i + 3, i = lo + 4 (threadIdx.x + 256 k); pixels behind `hi` read as 0
__device__ __forceinline__ void ld_run(const float* __restrict__ p, int i, int hi, float (&v)[4]) {
    if (i + 3 < hi) {
        const f4u t = *(const f4u*)(p + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = i + j < hi ? p[i + j] : 0.f;
    }
}
__device__ __forceinline__ void st_run(float* __restrict__ p, int i, int hi, const float (&v)[4]) {
    if (i + 3 < hi) {
        f4u t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        *(f4u*)(p + i) = t;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j < hi) p[i + j] = v[j];
    }
}

// ---------------------------------------------------------------------------------------
// Implicit-GEMM convolution.  M = output pixels of one image (64 per workgroup),
// N = output channels (16*FN per workgroup), K = (channel, ky, kx) in the weight's own
// memory order, consumed CK channels at a time.  TRANSPOSED = data-gradient form:
// out[c][iy][ix] = sum_{n,ky,kx} in[n][oy][ox] w[n][c][ky][kx] with oy*stride + ky - pad = iy.
// The gather offsets of a thread do not depend on the channel tile, so they are computed once;
// the next tile's operands are fetched into registers while the current one feeds the MFMAs.
// NG = 2: an 8-wave workgroup -- the second wave group takes the other half of every K tile's MFMA steps (and half of
// the gather), its accumulators are added through LDS at the end.  For the layers with at most ~2 workgroups per CU the
// run time is the serial K walk of one workgroup (fp32 MFMA: 32 cycles per 16x16x4 step), and this halves it.
template <int KS, int FN, int CK>
struct ConvTile {   // LDS geometry of one instantiation (shared by the kernel wrappers that carve the LDS)
    static constexpr int T = KS * KS, KT = CK * T, BM = 64;
    static constexpr int LDA = KS == 7 ? BM + 1 : BM + 16;   // 80: k-rows 16 banks apart -> conflict-free ds_read_b32 (7x7: 65, the 196-row tile must fit 64 KB)
    static constexpr int LDW = KT + 2;                        // 2*odd -> conflict-free
    static constexpr int A_FLOATS = KT * LDA, W_FLOATS = 16 * FN * LDW;
};
// bx / by / bz: the block coordinates of a launch of this convolution alone (m tile, n tile * ksplit + slice, image); As / Ws:
// ConvTile<..>::A_FLOATS / W_FLOATS floats of LDS.  Called from conv_igemm_kernel (one convolution per launch) and from
// conv_pair_kernel (two independent convolutions -- e.g. the 1x1 skip branch and the 3x3 stride-2 encoder convolution of one
// scale, which read the same input -- sharing one launch: a launch less on a latency-bound chain).
template <int KS, bool TRANSPOSED, int FN, int CK, int NG>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, int bx, int by, int bz, float* As, float* Ws) {
    constexpr int T = KS * KS;
    constexpr int KT = CK * T;          // k extent of one LDS tile (multiple of 4)
    constexpr int BM = 64;
    constexpr int LDA = ConvTile<KS, FN, CK>::LDA;
    constexpr int LDW = ConvTile<KS, FN, CK>::LDW;
    constexpr int BN = 16 * FN;
    constexpr int NTH = 256 * NG;
    constexpr int NWV = 4 * NG;                     // waves
    constexpr int NA = KT / NWV;                    // gathered elements per thread per tile
    constexpr int NW = (BN * KT + NTH - 1) / NTH;   // weight elements per thread per tile
    constexpr int KSTEPS = KT / 4 / NG;             // MFMA k steps per wave group per tile
    static_assert(KT % (4 * NG) == 0 && ((LDW / 2) & 1) == 1, "tile shape");
    static_assert(NG == 1 || KT * LDA >= FN * 4 * 256, "accumulator exchange reuses the A tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: what depends only on it stays out of the VGPRs)
    const int pw = wave & 3, grp = wave >> 2;       // pixel fragment / wave group
    const int img = bz;
    const int m0 = bx * BM;
    const int HWo = a.Ho * a.Wo;
    const float* in = a.in + (size_t)img * a.in_nstride;
    // independent images (several pairs optimised side by side): image n convolves with ITS OWN parameter arena
    const float* wgt = a.w + (size_t)img * a.p_nstride;
    const float* bias = a.bias ? a.bias + (size_t)img * a.p_nstride : nullptr;
    const int pl = tid & 63;
    const int p = m0 + pl;
    const bool pvalid = p < HWo;
    const int oy = pvalid ? p / a.Wo : 0, ox = pvalid ? p % a.Wo : 0;
    // split-K: blockIdx.y = n_tile * ksplit + slice; each slice reduces its own channel range and writes a raw
    // partial tile that conv_splitk_reduce_kernel sums in slice order (tiny deep layers: few tiles, long reductions)
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    const int kslice = by % ksplit;
    const int n0 = (by / ksplit) * BN;
    const int cper = ((a.Cin + ksplit - 1) / ksplit + CK - 1) / CK * CK;
    const int cbeg = kslice * cper;
    const int Kc = min(a.Cin, cbeg + cper);  // reduction channels [cbeg, Kc)
    // ---- per-thread gather descriptors (k = wave + NWV*i is wave-uniform)
    int a_off[NA], a_cl[NA];
    unsigned long long a_ok = 0;   // (7x7: 49 gathered elements per thread)
    static_assert(NA <= 64, "gather mask");
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int k = wave + NWV * i;
        const int cl = k / T, tap = k % T;
        const int ky = tap / KS, kx = tap % KS;
        int sy, sx;
        bool ok;
        if (!TRANSPOSED) {
            sy = oy * a.stride + ky - a.pad;
            sx = ox * a.stride + kx - a.pad;
            if (a.reflect) {   // nn.ReflectionPad2d in front of the convolution (models/unet/common.py:113-118): mirror without the edge
                sy = sy < 0 ? -sy : (sy >= a.Hi ? 2 * (a.Hi - 1) - sy : sy);
                sx = sx < 0 ? -sx : (sx >= a.Wi ? 2 * (a.Wi - 1) - sx : sx);
            }
            ok = sy >= 0 && sy < a.Hi && sx >= 0 && sx < a.Wi;
        } else {
            const int ty = oy + a.pad - ky, tx = ox + a.pad - kx;
            ok = ty >= 0 && tx >= 0;
            if (a.stride == 2) {
                ok = ok && !(ty & 1) && !(tx & 1);
                sy = ty >> 1; sx = tx >> 1;
            } else {
                sy = ty; sx = tx;
            }
            ok = ok && sy < a.Hi && sx < a.Wi;
        }
        ok = ok && pvalid;
        a_cl[i] = cl;
        a_off[i] = ok ? (int)(cl * a.in_cstride) + sy * a.Wi + sx : 0;
        if (ok) a_ok |= 1ull << i;
    }
    static_assert(NW <= 64, "weight mask");
    int w_off[NW], w_cl[NW], w_lds[NW];
    unsigned long long w_ok = 0;
#pragma unroll
    for (int t = 0; t < NW; ++t) {
        const int e = tid + NTH * t;
        const int j = e / KT, k = e % KT;
        const int cl = k / T, tap = k % T;
        const bool ok = e < BN * KT && (n0 + j) < a.Cout;
        w_cl[t] = cl;
        w_off[t] = ok ? (int)((size_t)(n0 + j) * a.w_jstride + (size_t)cl * a.w_cstride + tap) : 0;
        w_lds[t] = e < BN * KT ? j * LDW + k : -1;
        if (ok) w_ok |= 1ull << t;
    }
    float av[NA], wv[NW];
    // Full channel tiles are fetched with RAW BUFFER loads: a thread's byte offsets are fixed for the whole kernel (an element that is
    // padding / out of the image / out of the output-channel range carries bit 31 = out of the descriptor's range, and the hardware
    // returns 0 for it), the channel tile enters as the SCALAR offset -- no predicate, no branch, no address arithmetic per load
    // (the guarded global loads compiled to one exec-masked basic block per element: ~120 instructions and 13 branches per tile in
    // front of 18 MFMAs).  Same values, same order.  The last, partial channel tile of a range keeps the guarded form.
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wgt), 0, 0x7FFFFFFF, 0x00020000);
    int a_vo[NA], w_vo[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_vo[i] = ((a_ok >> i) & 1ull) ? a_off[i] * 4 : (int)0x80000000;
#pragma unroll
    for (int t = 0; t < NW; ++t) w_vo[t] = ((w_ok >> t) & 1ull) ? w_off[t] * 4 : (int)0x80000000;
    auto fetch = [&](int c0) {
        if (c0 + CK <= Kc) {
            const int so_a = __builtin_amdgcn_readfirstlane(c0 * (int)a.in_cstride * 4), so_w = __builtin_amdgcn_readfirstlane(c0 * (int)a.w_cstride * 4);
#pragma unroll
            for (int i = 0; i < NA; ++i) av[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, a_vo[i], so_a, 0));
#pragma unroll
            for (int t = 0; t < NW; ++t) wv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, w_vo[t], so_w, 0));
            return;
        }
        const float* inc = in + (size_t)c0 * a.in_cstride;
        const float* wc = wgt + (size_t)c0 * a.w_cstride;
#pragma unroll
        for (int i = 0; i < NA; ++i) av[i] = a_vo[i] >= 0 && (c0 + a_cl[i] < Kc) ? inc[a_vo[i] >> 2] : 0.f;
#pragma unroll
        for (int t = 0; t < NW; ++t) wv[t] = w_vo[t] >= 0 && (c0 + w_cl[t] < Kc) ? wc[w_vo[t] >> 2] : 0.f;
    };
    f32x4 acc[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(cbeg);
    // LDS addresses as ONE base per access family + compile-time offsets (the instruction's offset field).  The bases are re-defined (an empty asm) in
    // every trip: hoisted out of the loop as invariants, the 18 + 18 + 18 sums each sat in a register of their own (167 -> 188 VGPRs with the batches)
    // (integer indices, not pointers: a pointer that went through an asm loses its LDS address space and the accesses become flat)
    int a_wr = wave * LDA + pl;
    int a_rd = (grp * KSTEPS * 4 + (lane >> 4)) * LDA + pw * 16 + (lane & 15);
    int w_rd = (lane & 15) * LDW + grp * KSTEPS * 4 + (lane >> 4);
    for (int c0 = cbeg; c0 < Kc; c0 += CK) {
        asm volatile("" : "+v"(a_wr), "+v"(a_rd), "+v"(w_rd));
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) As[a_wr + NWV * i * LDA] = av[i];
#pragma unroll
        for (int t = 0; t < NW; ++t)
            if (w_lds[t] >= 0) Ws[w_lds[t]] = wv[t];
        __syncthreads();
        if (c0 + CK < Kc) fetch(c0 + CK);
        // The fragments of KB k steps are read from LDS as one batch, and the NEXT batch is on its way while this one feeds the MFMAs (round 5:
        // the compiler's own order was read -> wait -> MFMA per k step, an exposed LDS round trip in front of every one of the 18 MFMAs of a tile;
        // same operands in the same order, same bits).
        constexpr int KB0 = FN == 1 ? CONV_KB : 3;
        constexpr int KB = KB0 > 0 ? (KB0 < KSTEPS ? KB0 : KSTEPS) : KSTEPS;
        float af[2][KB], bf[2][FN][KB];
        auto frag = [&](int buf, int k0) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                if (k0 + t >= KSTEPS) break;
                af[buf][t] = As[a_rd + (k0 + t) * 4 * LDA];
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[buf][j][t] = Ws[w_rd + j * 16 * LDW + (k0 + t) * 4];
            }
        };
        frag(0, 0);
#pragma unroll
        for (int k0 = 0, cur = 0; k0 < KSTEPS; k0 += KB, cur ^= 1) {
            if (k0 + KB < KSTEPS) frag(cur ^ 1, k0 + KB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                if (k0 + t >= KSTEPS) break;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    acc[j] = mfma4(af[cur][t], bf[cur][j][t], acc[j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (NG == 2) {   // second wave group -> first, through the (now idle) A tile
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) As[(j * 4 + r) * 256 + (tid & 255)] = acc[j][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] += As[(j * 4 + r) * 256 + tid];
    }
    // epilogue: acc[j][r] = out[n = n0 + j*16 + (lane&15)][pixel = m0 + pw*16 + (lane>>4)*4 + r]
    if (ksplit > 1) {
        float* wsp = a.ws + (((size_t)kslice * a.N + img) * a.Cout) * HWo;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + j * 16 + (lane & 15);
            if (n >= a.Cout) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pp = m0 + pw * 16 + (lane >> 4) * 4 + r;
                if (pp < HWo) wsp[(size_t)n * HWo + pp] = acc[j][r];
            }
        }
        return;
    }
    float* out = a.out + (size_t)img * a.out_nstride;
    // bias and (accumulate) the previous values are loaded up front from clamped addresses: a guarded load per element
    // would be one memory round trip per element
    float bj[FN], prev[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int nc = min(n0 + j * 16 + (lane & 15), a.Cout - 1);
        bj[j] = bias ? bias[nc] : 0.f;
        if (a.accumulate) {
#pragma unroll
            for (int r = 0; r < 4; ++r) prev[j][r] = out[(size_t)nc * a.out_cstride + min(m0 + pw * 16 + (lane >> 4) * 4 + r, HWo - 1)];
        }
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + j * 16 + (lane & 15);
        if (n >= a.Cout) continue;
        const float b = bj[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pp = m0 + pw * 16 + (lane >> 4) * 4 + r;
            if (pp < HWo) {
                float v = acc[j][r] + b;
                if (a.act == 1) v = 1.0f / (1.0f + __expf(-v));
                out[(size_t)n * a.out_cstride + pp] = a.accumulate ? prev[j][r] + v : v;
            }
        }
    }
}

template <int KS, bool TRANSPOSED, int FN, int CK, int NG>
__global__ __launch_bounds__(256 * NG) void conv_igemm_kernel(ConvArgs a) {
    __shared__ float As[ConvTile<KS, FN, CK>::A_FLOATS];
    __shared__ float Ws[ConvTile<KS, FN, CK>::W_FLOATS];
    conv_igemm_body<KS, TRANSPOSED, FN, CK, NG>(a, blockIdx.x, blockIdx.y, blockIdx.z, As, Ws);
}
// Two independent convolutions in one launch: blocks [0, na) run convolution A (KS = KSA ...), the rest convolution B; each with
// the block coordinates of its own grid (gxa = A's gridDim.x, gxb = B's), blockIdx.z = image for both.  Each keeps the wave
// groups (NGA / NGB) of its own launch -- the surplus waves of the smaller one leave at once (a barrier does not wait for
// waves that have ended) -- so a convolution's bits do not depend on whether it was paired.  LDS is the larger footprint.
template <int KSA, bool TRA, int FNA, int CKA, int NGA, int KSB, bool TRB, int FNB, int CKB, int NGB>
__global__ __launch_bounds__(256 * (NGA > NGB ? NGA : NGB)) void conv_pair_kernel(ConvArgs a, ConvArgs b, int na, int gxa, int gxb) {
    using TA = ConvTile<KSA, FNA, CKA>;
    using TB = ConvTile<KSB, FNB, CKB>;
    constexpr int AF = TA::A_FLOATS > TB::A_FLOATS ? TA::A_FLOATS : TB::A_FLOATS;
    constexpr int WF = TA::W_FLOATS > TB::W_FLOATS ? TA::W_FLOATS : TB::W_FLOATS;
    __shared__ float As[AF];
    __shared__ float Ws[WF];
    const int bid = blockIdx.x;
    if (bid < na) {
        if (NGA < NGB && threadIdx.x >= 256 * NGA) return;
        conv_igemm_body<KSA, TRA, FNA, CKA, NGA>(a, bid % gxa, bid / gxa, blockIdx.z, As, Ws);
    } else {
        if (NGB < NGA && threadIdx.x >= 256 * NGB) return;
        conv_igemm_body<KSB, TRB, FNB, CKB, NGB>(b, (bid - na) % gxb, (bid - na) / gxb, blockIdx.z, As, Ws);
    }
}

// out = (accumulate ? out : 0) + bias + sum_slices ws   (slice order fixed)
__global__ void conv_splitk_reduce_kernel(ConvArgs a, int ksplit) {
    const int HWo = a.Ho * a.Wo;
    const size_t per = (size_t)a.N * a.Cout * HWo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        const int pp = i % HWo;
        const int n = (i / HWo) % a.Cout;
        const int img = i / ((size_t)HWo * a.Cout);
        float v = a.bias ? a.bias[(size_t)img * a.p_nstride + n] : 0.f;
        float* q = a.out + (size_t)img * a.out_nstride + (size_t)n * a.out_cstride + pp;
        const float prev = a.accumulate ? *q : 0.f;
        int k = 0;
        for (; k + 3 < ksplit; k += 4) {   // four slices in flight, added in slice order
            const float t0 = a.ws[(size_t)k * per + i], t1 = a.ws[(size_t)(k + 1) * per + i], t2 = a.ws[(size_t)(k + 2) * per + i],
                        t3 = a.ws[(size_t)(k + 3) * per + i];
            v += t0; v += t1; v += t2; v += t3;
        }
        for (; k < ksplit; ++k) v += a.ws[(size_t)k * per + i];
        if (a.act == 1) v = 1.0f / (1.0f + __expf(-v));
        *q = a.accumulate ? prev + v : v;
    }
}

// launch policy of one convolution (KS x KS filter, CK channels per K tile): output-channel fragments per workgroup, 8-wave
// workgroups (NG = 2), split-K; the grid is (mt, ny, N)
struct ConvPolicy { int fn_run, ng, ksplit, mt, ny; };
static ConvPolicy conv_policy(const ConvArgs& a, int KS, int CK) {
    const int HWo = a.Ho * a.Wo;
    const int mt = cdiv(HWo, 64);
    // output-channel fragments per workgroup, tuned in-step with alternating runs: the 64 / 128-channel layers of the deep
    // scales run fastest with ONE 16-channel fragment per workgroup (FN = 4: +2.2 %, FN = 2: +0.6 %) -- these launches are
    // latency-bound and more, smaller workgroups shorten them; the 32-channel layers are indifferent (2 kept)
    const int fn = (KS < 5 && a.Cout > 16 && a.Cout <= 32) ? 2 : 1;
    const int nt = cdiv(a.Cout, 16 * fn);
    // Several images per launch (pairs side by side, or the crops of one pair): the chip is full anyway, and a workgroup that owns
    // 32 output channels gathers its input tile once instead of twice (same-box A/B at 4 / 8 pairs per GPU: 2 fragments -1.0 % /
    // -0.75 % step time, 4 fragments -0.3 % / 0; profiles/r03_conv_fn_ab.txt).  Which output channels share a workgroup does not
    // touch any sum, so a pair's bits do not depend on it -- the split-K / wave-group policy below (which does change the
    // summation order) keeps using the one-image fragment count.
    constexpr int batch_fn = 2;
    constexpr int batch_min = 4;
    int fn_run = (KS < 5 && fn == 1 && a.Cout >= 64 && a.N >= batch_min && (batch_fn == 2 || batch_fn == 4)) ? batch_fn : fn;
    // Big planes (round 5; the reference's default 855 .. 900 crops, 448^2, 512^2): thousands of pixel tiles per layer fill the chip whatever the
    // channel split, and every 16-channel fragment a workgroup does NOT own is a second gather of the same input tile by another workgroup
    static const int big_fn = getenv("SPLICE_CONV_BIG_FN") ? atoi(getenv("SPLICE_CONV_BIG_FN")) : 4;   // same-box A/B at 900 x 1200: 0 -> 10.14, 2 -> 10.12, 4 -> 10.10 ms per step
    constexpr int big_mt = 512;
    if (KS < 5 && (big_fn == 2 || big_fn == 4) && mt >= big_mt && a.Cout > 16) {
        const int want = a.Cout > 32 ? big_fn : 2;
        if (want > fn_run) fn_run = want;
    }
    const int nt_run = cdiv(a.Cout, 16 * fn_run);
    // launch policy (split-K, 8-wave workgroups) from the workgroups of ONE image when the images are independent pairs:
    // split-K changes the summation order, and a pair's result must not depend on how many pairs share the launch
    const int npol = a.p_nstride ? 1 : a.N;
    const int wgs = mt * nt * npol;
    int ksplit = 1;
    const int ktiles = cdiv(a.Cin, CK);
    if (a.ws && wgs < 128 && ktiles >= 4 && (size_t)a.N * a.Cout * HWo * 16 <= a.ws_floats) {
        // down to ONE channel tile per slice: in-step (cold caches, latency-bound) more, shorter workgroups win 0.5 % over
        // two tiles per slice, although the second tile's loads would overlap the first one's MFMAs
        ksplit = cdiv(256, wgs);   // (targets of 128 / 384 / 512 workgroups lose 0.3-0.6 %, splitting grids of up to 256 loses 0.8 %)
        if (ksplit > ktiles) ksplit = ktiles;
        if (ksplit > 16) ksplit = 16;
        if (ksplit < 2) ksplit = 1;
    }
    // 8-wave workgroups while the chip holds at most ~2 workgroups per CU (the serial K walk is the run time then)
    // (K tile of 72: 9 steps per wave group; the A tile is big enough for the exchange)
    const bool can8 = KS == 3 && CK == 8;
    const bool ng2 = can8 && (long)mt * nt * ksplit * npol <= 2048 && cdiv(a.Cin, CK) >= 2;
    ConvPolicy p;
    p.fn_run = KS >= 5 ? 1 : fn_run;
    p.ng = ng2 ? 2 : 1;
    p.ksplit = ksplit;
    p.mt = mt;
    p.ny = (KS >= 5 ? cdiv(a.Cout, 16) : nt_run) * ksplit;
    return p;
}
static void conv_splitk_reduce_launch(const ConvArgs& a, int ksplit, hipStream_t s) {
    const size_t per = (size_t)a.N * a.Cout * a.Ho * a.Wo;
    size_t g = (per + 255) / 256;
    if (g > 1024) g = 1024;
    SPLICE_LAUNCH(conv_splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, a, ksplit);
}

// roofline leg: the layer's algorithmic work rides with the launch (prof.hip): 2 x outputs x reduction length FLOPs; input + output + weight bytes
static inline void conv_note_work(const ConvArgs& a) {
    if (g_splice_prof_open <= 0) return;
    const double outs = (double)a.N * a.Cout * a.Ho * a.Wo, red = (double)a.Cin * a.ks * a.ks;
    splice_prof_note(2.0 * outs * red, 4.0 * (outs + (double)a.N * a.Cin * a.Hi * a.Wi + (double)a.Cout * red * (a.p_nstride ? a.N : 1)));
}
template <int KS, bool TR, int CK>
static void conv_launch_fn(ConvArgs a, hipStream_t s, int* ksplit_out) {
    const ConvPolicy pol = conv_policy(a, KS, CK);
    conv_note_work(a);
    a.ksplit = pol.ksplit;
    const dim3 grid(pol.mt, pol.ny, a.N);
    constexpr bool CAN8 = KS == 3 && CK == 8;
    if constexpr (KS >= 5) {   // 5x5 / 7x7 (the inversion experiment's generator): one fragment per workgroup, 4 waves
        SPLICE_LAUNCH((conv_igemm_kernel<KS, TR, 1, CK, 1>), grid, dim3(256), 0, s, a);
    } else {
        if (pol.ng == 2) {
            if constexpr (CAN8) {
                if (pol.fn_run == 1) SPLICE_LAUNCH((conv_igemm_kernel<KS, TR, 1, CK, 2>), grid, dim3(512), 0, s, a);
                else if (pol.fn_run == 2) SPLICE_LAUNCH((conv_igemm_kernel<KS, TR, 2, CK, 2>), grid, dim3(512), 0, s, a);
                else SPLICE_LAUNCH((conv_igemm_kernel<KS, TR, 4, CK, 2>), grid, dim3(512), 0, s, a);
            }
        } else {
            if (pol.fn_run == 1) SPLICE_LAUNCH((conv_igemm_kernel<KS, TR, 1, CK, 1>), grid, dim3(256), 0, s, a);
            else if (pol.fn_run == 2) SPLICE_LAUNCH((conv_igemm_kernel<KS, TR, 2, CK, 1>), grid, dim3(256), 0, s, a);
            else SPLICE_LAUNCH((conv_igemm_kernel<KS, TR, 4, CK, 1>), grid, dim3(256), 0, s, a);
        }
    }
    if (ksplit_out) *ksplit_out = pol.ksplit;
    if (pol.ksplit > 1 && !a.defer_reduce) conv_splitk_reduce_launch(a, pol.ksplit, s);
}

// channels per K tile as conv_launch picks them (deeper tiles where the reduction is long)
// (K tiles of 16 channels = 144 k-values for the 3x3 layers with >= 64 / 128 input channels -- half the barrier rounds, twice the
// gathered loads in flight per round -- measured +2.0 % / +1.3 % step time at one pair per GPU, +2.4 % at eight: not kept,
// profiles/r04_gen_ab.txt)
static inline int conv_ck(const ConvArgs& a) { return a.ks == 3 ? (a.Cin >= 32 ? 8 : 4) : a.ks == 1 ? (a.Cin >= 64 ? 32 : 16) : 4; }

// Two INDEPENDENT convolutions in one launch (conv_pair_kernel).  Forward: a = the 1x1 skip convolution of a scale, b = its 3x3
// stride-2 encoder convolution (same input, models/unet/skip.py:60-66); backward: two 1x1 data-gradient convolutions (the skip
// branch's and the deeper scale's last decoder convolution).  Each keeps the launch policy, the split-K workspace and the BITS of
// its own launch, so pairing is a pure launch-count choice: it pays on the latency-bound chains of few images (-0.9 % step time at
// one pair per GPU) and not where the launches fill the chip (+0.2 % at eight: the pair runs at the larger register / LDS
// footprint), hence only below SPLICE_CONV_PAIR_MAXN images (profiles/r04_gen_ab.txt).  Combinations outside the instantiated set,
// reflection padding or 5x5 / 7x7 filters fall back to two launches.  Returns through ksplit_a / ksplit_b what conv_launch would.
template <int KSA, bool TRA, int FNA, int CKA, int KSB, bool TRB, int FNB, int CKB, int NGB>
static void conv_pair_go(const ConvArgs& a, const ConvArgs& b, const ConvPolicy& pa, const ConvPolicy& pb, hipStream_t s) {
    const int na = pa.mt * pa.ny, nb = pb.mt * pb.ny;
    SPLICE_LAUNCH((conv_pair_kernel<KSA, TRA, FNA, CKA, 1, KSB, TRB, FNB, CKB, NGB>), dim3(na + nb, 1, a.N), dim3(256 * NGB), 0, s, a, b, na, pa.mt, pb.mt);
}
int conv_pair_launch(ConvArgs a, ConvArgs b, hipStream_t s, int* ksplit_a, int* ksplit_b) {
    static const int pair_on = getenv("SPLICE_CONV_PAIR") ? atoi(getenv("SPLICE_CONV_PAIR")) : 1;
    constexpr int pair_maxn = 4;
    const int cka = conv_ck(a), ckb = conv_ck(b);
    bool done = false;
    if (pair_on && a.N == b.N && a.N < pair_maxn && !a.reflect && !b.reflect && a.ks == 1 && a.stride == 1 && (b.stride == 1 || b.stride == 2) &&
        a.transposed == b.transposed && (size_t)a.Cin * a.in_cstride <= 0x7fffffffULL && (size_t)b.Cin * b.in_cstride <= 0x7fffffffULL) {
        const ConvPolicy pa = conv_policy(a, 1, cka), pb = conv_policy(b, b.ks, ckb);
        a.ksplit = pa.ksplit; b.ksplit = pb.ksplit;
#define PAIR(TR_, FNA_, CKA_, KSB_, FNB_, CKB_, NG_)                                                                                         \
    if (!done && a.transposed == (TR_ ? 1 : 0) && pa.ng == 1 && pa.fn_run == FNA_ && cka == CKA_ && b.ks == KSB_ && pb.fn_run == FNB_ && ckb == CKB_ && pb.ng == NG_) { \
        conv_pair_go<1, TR_, FNA_, CKA_, KSB_, TR_, FNB_, CKB_, NG_>(a, b, pa, pb, s);                                                       \
        done = true;                                                                                                                          \
    }
        // forward: skip (Cout = 4: one fragment) || 3x3 encoder convolution
        PAIR(false, 1, 16, 3, 1, 4, 1) PAIR(false, 1, 16, 3, 2, 4, 1)
        PAIR(false, 1, 16, 3, 1, 8, 1) PAIR(false, 1, 16, 3, 2, 8, 1) PAIR(false, 1, 16, 3, 1, 8, 2) PAIR(false, 1, 16, 3, 2, 8, 2)
        PAIR(false, 1, 32, 3, 1, 8, 1) PAIR(false, 1, 32, 3, 2, 8, 1) PAIR(false, 1, 32, 3, 1, 8, 2) PAIR(false, 1, 32, 3, 2, 8, 2)
        // backward: skip data gradient (4 -> Cin channels) || 1x1 data gradient of the deeper scale's decoder output convolution
        PAIR(true, 1, 16, 1, 1, 16, 1) PAIR(true, 1, 16, 1, 2, 16, 1) PAIR(true, 2, 16, 1, 1, 16, 1) PAIR(true, 2, 16, 1, 2, 16, 1)
        PAIR(true, 1, 16, 1, 1, 32, 1) PAIR(true, 1, 16, 1, 2, 32, 1) PAIR(true, 2, 16, 1, 1, 32, 1) PAIR(true, 2, 16, 1, 2, 32, 1)
#undef PAIR
        if (done) {
            if (ksplit_a) *ksplit_a = pa.ksplit;
            if (ksplit_b) *ksplit_b = pb.ksplit;
            if (pa.ksplit > 1 && !a.defer_reduce) conv_splitk_reduce_launch(a, pa.ksplit, s);
            if (pb.ksplit > 1 && !b.defer_reduce) conv_splitk_reduce_launch(b, pb.ksplit, s);
            return SPLICE_OK;
        }
    }
    int rc = conv_launch(a, s, ksplit_a);
    if (rc != SPLICE_OK) return rc;
    return conv_launch(b, s, ksplit_b);
}

// ---------------------------------------------------------------------------------------
// 3x3 stride-1 convolution of BIG planes (round 5: the reference's default 855 .. 900 crops, 448^2, 512^2): a 2-D pixel tile with the input halo
// staged in LDS.  conv_igemm_body gathers 64 pixels x 72 k-values per channel tile for 18 MFMAs per wave -- every input element is fetched 9 times
// per 16 output channels, each behind its own descriptor -- and the instruction stream around the MFMAs costs twice the matrix pipe's time on
// these layers (profiles/r05_conv_big_planes.txt).  Here a workgroup owns 4 RW rows x 64 columns of output (RW = 1 is what runs: 4 rows); per channel
// chunk it stages the (4 RW + 2) x (64 + 2) input patch ONCE (padding / reflection resolved while staging, row-contiguous loads), a wave owns RW rows =
// 4 RW pixel fragments, and one weight fragment + one address add serve 4 RW MFMAs per output-channel fragment.  The k order inside a chunk, the chunk size (conv_ck) and the operand layout are those
// of conv_igemm_body: the same bits (tools/gen_bits.py under SPLICE_CONV_TILE=0 / 1).
constexpr int CT_TW = 64, CT_PW = CT_TW + 2;
// RW = output rows per wave (1: the 4-row tile, the default on every plane -- 118 .. 133 VGPRs, 3 - 4 waves per SIMD; 2: an 8-row tile with 1.25 x instead of
// 1.5 x the halo and 167 .. 183 VGPRs, measured slower everywhere: SPLICE_CONV_TILE_RW1_MAX)
template <bool TRANSPOSED, int FN, int CK, int RW>
__global__ __launch_bounds__(256) void conv3x3_tile_kernel(ConvArgs a, int tiles_x) {
    constexpr int CT_TH = 4 * RW, CT_PH = CT_TH + 2, CT_PLANE = CT_PH * CT_PW, NF = 4 * RW;   // NF = 16-pixel fragments per wave
    constexpr int KT = CK * 9, KSTEPS = KT / 4, LDW = KT + 2, BN = 16 * FN;
    constexpr int PE = CK * CT_PLANE;                 // patch elements per chunk
    constexpr int NP = (PE + 255) / 256;              // ... per thread
    constexpr int NW = (BN * KT + 255) / 256;         // weight elements per thread per chunk
    __shared__ float Ps[PE];
    __shared__ float Ws[BN * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.z;
    const int x0 = (blockIdx.x % tiles_x) * CT_TW, y0 = (blockIdx.x / tiles_x) * CT_TH;
    const int n0 = blockIdx.y * BN;
    const float* in = a.in + (size_t)img * a.in_nstride;
    const float* wgt = a.w + (size_t)img * a.p_nstride;
    const float* bias = a.bias ? a.bias + (size_t)img * a.p_nstride : nullptr;
    // source coordinates of the patch origin: forward sy = oy - pad + ky, data gradient sy = oy + pad - ky (ky = 0 .. 2)
    const int sy0 = TRANSPOSED ? y0 + a.pad - 2 : y0 - a.pad, sx0 = TRANSPOSED ? x0 + a.pad - 2 : x0 - a.pad;
    // ---- staging descriptors: patch element e = tid + 256 j = (channel, patch row, patch column); fixed for the kernel, the chunk enters as the scalar offset
    int p_vo[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int e = tid + 256 * j;
        const int c = e / CT_PLANE, rem = e % CT_PLANE, r = rem / CT_PW, x = rem % CT_PW;
        int sy = sy0 + r, sx = sx0 + x;
        if (!TRANSPOSED && a.reflect) {   // nn.ReflectionPad2d in front of the convolution: mirror without the edge
            sy = sy < 0 ? -sy : (sy >= a.Hi ? 2 * (a.Hi - 1) - sy : sy);
            sx = sx < 0 ? -sx : (sx >= a.Wi ? 2 * (a.Wi - 1) - sx : sx);
        }
        const bool ok = e < PE && sy >= 0 && sy < a.Hi && sx >= 0 && sx < a.Wi;
        p_vo[j] = ok ? (int)(c * a.in_cstride + (size_t)sy * a.Wi + sx) * 4 : (int)0x80000000;
    }
    int w_vo[NW], w_lds[NW];
#pragma unroll
    for (int t = 0; t < NW; ++t) {
        const int e = tid + 256 * t;
        const int j = e / KT, k = e % KT;
        const int cl = k / 9, tap = k % 9;
        const bool ok = e < BN * KT && (n0 + j) < a.Cout;
        w_vo[t] = ok ? (int)((size_t)(n0 + j) * a.w_jstride + (size_t)cl * a.w_cstride + tap) * 4 : (int)0x80000000;
        w_lds[t] = e < BN * KT ? j * LDW + k : -1;
    }
    // ---- fragment addressing: lane (pixel i = lane & 15, k group g = lane >> 4); k = kk * 4 + g = (channel, ky, kx) -> offset inside the patch
    int koff[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        const int k = kk * 4 + (lane >> 4);
        const int c = k / 9, tap = k % 9, ky = tap / 3, kx = tap % 3;
        koff[kk] = c * CT_PLANE + (TRANSPOSED ? 2 - ky : ky) * CT_PW + (TRANSPOSED ? 2 - kx : kx);
    }
    int a_rd = (RW * wave) * CT_PW + (lane & 15);                 // pixel (row 2 wave, column lane & 15) of the tile, in patch coordinates without the tap
    int w_rd = (lane & 15) * LDW + (lane >> 4);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wgt), 0, 0x7FFFFFFF, 0x00020000);
    const int Kc = a.Cin;
    float pv[NP], wv[NW];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
        const int so_a = __builtin_amdgcn_readfirstlane(c0 * (int)a.in_cstride * 4), so_w = __builtin_amdgcn_readfirstlane(c0 * (int)a.w_cstride * 4);
        if (c0 + CK <= Kc) {
#pragma unroll
            for (int j = 0; j < NP; ++j) pv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, p_vo[j], so_a, 0));
#pragma unroll
            for (int t = 0; t < NW; ++t) wv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, w_vo[t], so_w, 0));
        } else {   // the last, partial chunk of the reduction: channels behind Kc read as 0
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const bool cok = c0 + (tid + 256 * j) / CT_PLANE < Kc;
                pv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, cok ? p_vo[j] : (int)0x80000000, so_a, 0));
            }
#pragma unroll
            for (int t = 0; t < NW; ++t) {
                const bool cok = c0 + ((tid + 256 * t) % KT) / 9 < Kc;
                wv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, cok ? w_vo[t] : (int)0x80000000, so_w, 0));
            }
        }
    };
    const int fn_live = __builtin_amdgcn_readfirstlane(min(FN, (a.Cout - n0 + 15) / 16));   // fragments of this workgroup that hold output channels
    f32x4 acc[NF][FN];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int c0 = 0; c0 < Kc; c0 += CK) {
        asm volatile("" : "+v"(a_rd), "+v"(w_rd));   // (LDS bases re-defined per trip: base + instruction offset instead of one register per address)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (tid + 256 * j < PE) Ps[tid + 256 * j] = pv[j];
#pragma unroll
        for (int t = 0; t < NW; ++t)
            if (w_lds[t] >= 0) Ws[w_lds[t]] = wv[t];
        __syncthreads();
        if (c0 + CK < Kc) fetch(c0 + CK);
        // fragments of k step kk + 1 are on their way while step kk feeds 8 x FN MFMAs
        float af[2][NF], bf[2][FN];
        auto frag = [&](int buf, int kk) __attribute__((always_inline)) {
            const int ao = a_rd + koff[kk];
#pragma unroll
            for (int f = 0; f < NF; ++f) af[buf][f] = Ps[ao + (f >> 2) * CT_PW + (f & 3) * 16];
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[buf][j] = Ws[w_rd + j * 16 * LDW + kk * 4];
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (kk + 1 < KSTEPS) frag((kk + 1) & 1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (j > 0 && j >= fn_live) continue;   // a 16-channel fragment wholly behind Cout (36 = 32 + 4, 68, 132 output channels): no MFMAs for it
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f][j] = mfma4(af[kk & 1][f], bf[kk & 1][j], acc[f][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: acc[f][j][r] = out[n = n0 + j*16 + (lane & 15)][row y0 + 2 wave + (f >> 2)][column x0 + (f & 3) * 16 + (lane >> 4) * 4 + r]
    float* out = a.out + (size_t)img * a.out_nstride;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + j * 16 + (lane & 15);
        if (n >= a.Cout) continue;
        const float b = bias ? bias[n] : 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int row = y0 + RW * wave + (f >> 2), col = x0 + (f & 3) * 16 + (lane >> 4) * 4;
            if (row >= a.Ho || col >= a.Wo) continue;
            float* q = out + (size_t)n * a.out_cstride + (size_t)row * a.Wo + col;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[f][j][r] + b;   // (no activation in this kernel: the generator's only sigmoid sits behind a 1x1 convolution; conv_tile_ok)
            }
            if (a.accumulate) {
                float pr[4];
                ld_run(q, 0, a.Wo - col, pr);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = pr[r] + v[r];
            }
            st_run(q, 0, a.Wo - col, v);
        }
    }
}
static bool conv_tile_ok(const ConvArgs& a) {
    static const int on = getenv("SPLICE_CONV_TILE") ? atoi(getenv("SPLICE_CONV_TILE")) : 1;
    // planes above 40000 pixels (same-box A/B, ms per step at min 65536 / 40000 / 12000: one pair at 224^2 3.630 / 3.609 / 3.717, eight pairs 16.40 / 16.25 / 16.12,
    // 900 x 1200 9.42 / 9.23 / 9.41: the 112^2 planes are 28 tiles -- too few for one image, and the policy must not depend on the images per launch)
    static const int min_px = getenv("SPLICE_CONV_TILE_MIN") ? atoi(getenv("SPLICE_CONV_TILE_MIN")) : 40000;
    return on && a.ks == 3 && a.stride == 1 && !a.act && a.Wo >= 64 && (long long)a.Ho * a.Wo > min_px && !(a.reflect && a.transposed) &&
           (size_t)a.Cin * a.in_cstride <= 0x1fffffffULL;
}
template <bool TR, int CK>
static void conv_tile_launch(const ConvArgs& a, hipStream_t s) {
    conv_note_work(a);
    constexpr int rw1_max = 0x7fffffff;   // planes up to this many pixels take the 4-row tile: all of them (same-box A/B against the 8-row tile: 900 x 1200 9.20 -> 9.07 ms, one pair at 224^2 3.594 -> 3.574, eight pairs 16.33 -> 16.30: 118 .. 133 instead of 167 .. 183 VGPRs)
    const bool rw1 = (long long)a.Ho * a.Wo <= rw1_max;
    const int tiles_x = cdiv(a.Wo, CT_TW), tiles_y = cdiv(a.Ho, rw1 ? 4 : 8);
    if (rw1) {
        if (a.Cout <= 16) SPLICE_LAUNCH((conv3x3_tile_kernel<TR, 1, CK, 1>), dim3(tiles_x * tiles_y, 1, a.N), dim3(256), 0, s, a, tiles_x);
        else SPLICE_LAUNCH((conv3x3_tile_kernel<TR, 2, CK, 1>), dim3(tiles_x * tiles_y, cdiv(a.Cout, 32), a.N), dim3(256), 0, s, a, tiles_x);
        return;
    }
    if (a.Cout <= 16) SPLICE_LAUNCH((conv3x3_tile_kernel<TR, 1, CK, 2>), dim3(tiles_x * tiles_y, 1, a.N), dim3(256), 0, s, a, tiles_x);
    else SPLICE_LAUNCH((conv3x3_tile_kernel<TR, 2, CK, 2>), dim3(tiles_x * tiles_y, cdiv(a.Cout, 32), a.N), dim3(256), 0, s, a, tiles_x);
}

int conv_launch(const ConvArgs& a, hipStream_t s, int* ksplit_out) {
    if (a.ks != 1 && a.ks != 3 && a.ks != 5 && a.ks != 7) return SPLICE_ERR_ARG;
    if (a.reflect && a.transposed) return SPLICE_ERR_ARG;   // the data gradient of a reflection-padded conv goes through conv_reflect_dgrad_launch
    if (a.ks >= 5) {
        if (a.ks == 5) { if (a.transposed) conv_launch_fn<5, true, 4>(a, s, ksplit_out); else conv_launch_fn<5, false, 4>(a, s, ksplit_out); }
        else { if (a.transposed) conv_launch_fn<7, true, 4>(a, s, ksplit_out); else conv_launch_fn<7, false, 4>(a, s, ksplit_out); }
        return SPLICE_OK;
    }
    if (a.stride != 1 && a.stride != 2) return SPLICE_ERR_ARG;
    if ((size_t)a.Cin * a.in_cstride > 0x7fffffffULL) return SPLICE_ERR_ARG;   // 32-bit gather offsets
    // deeper channel tiles where the reduction is long (fewer barrier rounds on the small, deep layers)
    if (conv_tile_ok(a)) {   // big planes: the LDS-halo tile (no split-K there)
        if (a.Cin >= 32) { if (a.transposed) conv_tile_launch<true, 8>(a, s); else conv_tile_launch<false, 8>(a, s); }
        else { if (a.transposed) conv_tile_launch<true, 4>(a, s); else conv_tile_launch<false, 4>(a, s); }
        if (ksplit_out) *ksplit_out = 1;
        return SPLICE_OK;
    }
    if (a.ks == 3) {
        if (a.Cin >= 32) { if (a.transposed) conv_launch_fn<3, true, 8>(a, s, ksplit_out); else conv_launch_fn<3, false, 8>(a, s, ksplit_out); }
        else { if (a.transposed) conv_launch_fn<3, true, 4>(a, s, ksplit_out); else conv_launch_fn<3, false, 4>(a, s, ksplit_out); }
    } else {
        if (a.Cin >= 64) { if (a.transposed) conv_launch_fn<1, true, 32>(a, s, ksplit_out); else conv_launch_fn<1, false, 32>(a, s, ksplit_out); }
        else { if (a.transposed) conv_launch_fn<1, true, 16>(a, s, ksplit_out); else conv_launch_fn<1, false, 16>(a, s, ksplit_out); }
    }
    return SPLICE_OK;
}

// Data gradient of a reflection-padded convolution: y = conv(pad_reflect(x)), so dL/dx = fold(dL/dx_pad) where dL/dx_pad is the
// plain transposed convolution on the PADDED domain (Hi + 2p) x (Wi + 2p) -- conv_launch in data-gradient form with pad 0 --
// and fold adds every padded position into the interior pixel it mirrors (separable: up to 3 source rows x 3 source
// columns per pixel, fixed order).
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dpad, float* __restrict__ dx, size_t dx_nstride, size_t dx_cstride,
                                                           int C, int H, int W, int p, int accumulate) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const int c = blockIdx.y, img = blockIdx.z;
    const float* src = dpad + ((size_t)img * C + c) * Hp * Wp;
    float* dst = dx + (size_t)img * dx_nstride + (size_t)c * dx_cstride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int y = i / W, x = i % W;
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;                               // top border mirrors rows 1..p
        if (y <= H - 2 && y >= H - 1 - p) ys[ny++] = p + 2 * (H - 1) - y;     // bottom border mirrors rows H-1-p..H-2
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x <= W - 2 && x >= W - 1 - p) xs[nx++] = p + 2 * (W - 1) - x;
        float acc = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) acc += src[(size_t)ys[a] * Wp + xs[b]];
        dst[i] = accumulate ? dst[i] + acc : acc;
    }
}
// a: the layer in data-gradient form as for the zero-padded case (in = dy, out = d_in [N][Cout][Ho][Wo], pad = the layer's
// padding); pad_scratch: N * Cout * (Ho + 2 pad) * (Wo + 2 pad) floats
int conv_reflect_dgrad_launch(ConvArgs a, float* pad_scratch, hipStream_t s) {
    const int p = a.pad, Hp = a.Ho + 2 * p, Wp = a.Wo + 2 * p;
    float* out = a.out;
    const size_t out_ns = a.out_nstride, out_cs = a.out_cstride;
    const int acc = a.accumulate, H = a.Ho, W = a.Wo;
    a.out = pad_scratch; a.out_nstride = (size_t)a.Cout * Hp * Wp; a.out_cstride = (size_t)Hp * Wp;
    a.Ho = Hp; a.Wo = Wp; a.pad = 0; a.accumulate = 0; a.reflect = 0; a.transposed = 1; a.ws = nullptr;
    const int rc = conv_launch(a, s);
    if (rc != SPLICE_OK) return rc;
    int gx = cdiv(H * W, 256);
    if (gx > 64) gx = 64;
    SPLICE_LAUNCH(reflect_fold_kernel, dim3(gx, a.Cout, a.N), dim3(256), 0, s, pad_scratch, out, out_ns, out_cs, a.Cout, H, W, p, acc);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
// Weight gradient: dW[n][c][tap] = sum_{img,pixel} dy[n][pixel] * x[c][tap-shifted pixel].
// Workgroup = (pixel chunk, 4-or-16-channel K tile); MFMA reduces over pixels (64 per LDS fill, the next
// fill prefetched into registers); the partial tile goes to ws[chunk][n][c][tap] and ONE
// wgrad_reduce_all_kernel per backward sums every layer's chunks in a fixed order (bit-reproducible).
constexpr int WG_PC = 64;                      // pixels per LDS fill
constexpr int WG_LD = WG_PC + 2;               // 66 = 2*33
constexpr int WG_XS_FLOATS = 3 * 16 * WG_LD;   // [k][pixel], largest variant (3x3: 36 -> 48 k rows)
constexpr int WG_DS_FLOATS = 8 * 16 * WG_LD;   // [n][pixel], largest variant (128 output channels)
// ROWS (5x5 / 7x7 filters): a K tile is 4 channels x ONE filter row (KS taps), ktile = channel_tile * KS + ky -- keeps the
// tile at 20 / 28 k-values instead of 100 / 196.
template <int KS, int NI, bool ROWS = false>   // NI = 16-row fragments of output channels
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs& a, int chunk, int ktile_in, float* Xs, float* Ds) {
    constexpr int T = ROWS ? KS : KS * KS;     // taps per channel inside one K tile
    constexpr int CK = (ROWS || KS == 3) ? 4 : 16;
    constexpr int KT = CK * T;                 // 36 / 16 / 20 / 28
    const int ktile = ROWS ? ktile_in / KS : ktile_in;
    const int krow = ROWS ? ktile_in % KS : 0;
    constexpr int NJ = (KT + 15) / 16;         // 3 / 1
    constexpr int PC = WG_PC;
    constexpr int LD = WG_LD;
    constexpr int NQ = (NI * NJ + 3) / 4;      // fragment pairs per wave
    constexpr int NA = KT / 4;                 // gathered x elements per thread per fill
    constexpr int ND = NI * 16 * PC / 256;     // dy elements per thread per fill
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = ktile * CK;
    const int HWo = a.Ho * a.Wo;
    const int chunks_per_img = a.chunks_per_img;
    const int img = chunk / chunks_per_img, ch_in_img = chunk % chunks_per_img;
    const float* x = a.x + (size_t)img * a.x_nstride + (size_t)c0 * a.x_cstride;
    const float* dy = a.dy + (size_t)img * a.dy_nstride;
    f32x4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < (NJ * 16 - KT) * LD; e += 256) Xs[KT * LD + e] = 0.f;   // k-padding rows stay zero
    const int pl = tid & 63;
    const int p_begin = ch_in_img * a.pix_per_chunk;
    const int p_end = min(p_begin + a.pix_per_chunk, HWo);
    // per-thread gather descriptors (k = wave + 4*i is wave-uniform): channel + tap displacement
    int g_coff[NA], g_dy[NA], g_dx[NA];
    unsigned g_cok = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int k = wave + 4 * i;
        const int cl = k / T, tap = k % T;
        g_coff[i] = (int)(cl * a.x_cstride);
        g_dy[i] = (ROWS ? krow : tap / KS) - a.pad;
        g_dx[i] = (ROWS ? tap : tap % KS) - a.pad;
        if (c0 + cl < a.Cin) g_cok |= 1u << i;
    }
    float xv[NA], dv[ND];
    // raw buffer loads (see conv_igemm_body): an element outside the image / the channel range / the chunk gets an offset with bit 31
    // set and reads as 0 -- no exec-masked block per element.  The output-gradient offsets of a thread are fixed, the fill enters as
    // the scalar offset.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, 0x7FFFFFFF, 0x00020000);
    int d_vo[ND];
#pragma unroll
    for (int t = 0; t < ND; ++t) {
        const int e = tid + 256 * t;
        const int n = e / PC, q = e % PC;
        d_vo[t] = n < a.Cout ? (int)((n * a.dy_cstride + q) * 4) : (int)0x80000000;
    }
    auto fetch = [&](int pb) {
        const int p = pb + pl;
        const bool pvalid = p < p_end;
        const int oy = pvalid ? p / a.Wo : 0, ox = pvalid ? p % a.Wo : 0;
        const int by = oy * a.stride, bx = ox * a.stride;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int sy = by + g_dy[i], sx = bx + g_dx[i];
            if (a.reflect) {
                sy = sy < 0 ? -sy : (sy >= a.Hi ? 2 * (a.Hi - 1) - sy : sy);
                sx = sx < 0 ? -sx : (sx >= a.Wi ? 2 * (a.Wi - 1) - sx : sx);
            }
            const bool ok = pvalid && ((g_cok >> i) & 1u) && (unsigned)sy < (unsigned)a.Hi && (unsigned)sx < (unsigned)a.Wi;
            const int vo = ok ? (g_coff[i] + sy * a.Wi + sx) * 4 : (int)0x80000000;
            xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, 0, 0));
        }
        if (pb + PC <= p_end) {   // a full fill: fixed per-thread offsets + the fill as the scalar offset
            const int so = __builtin_amdgcn_readfirstlane(pb * 4);
#pragma unroll
            for (int t = 0; t < ND; ++t) dv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, d_vo[t], so, 0));
        } else {
#pragma unroll
            for (int t = 0; t < ND; ++t) {
                const int e = tid + 256 * t;
                const int n = e / PC, q = e % PC;
                const int pp = pb + q;
                dv[t] = (n < a.Cout && pp < p_end) ? dy[(size_t)n * a.dy_cstride + pp] : 0.f;
            }
        }
    };
    fetch(p_begin);
    // fragment reads in batches of WB pixel steps, the next batch in flight under this one's MFMAs (round 5, as in conv_igemm_body: the compiler's
    // order was read -> wait -> MFMA per step); one LDS base per operand + compile-time offsets, re-defined per fill so that they are not hoisted
    // into a register each.  Same operands, same order, same bits.
    constexpr int WB = 4, NB = PC / 4 / WB;
    [[maybe_unused]] int lds_rd = (lane & 15) * LD + (lane >> 4);
    for (int pb = p_begin; pb < p_end; pb += PC) {
        if constexpr (NI < 4) asm volatile("" : "+v"(lds_rd));
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) Xs[(wave + 4 * i) * LD + pl] = xv[i];
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int e = tid + 256 * t;
            Ds[(e / PC) * LD + (e % PC)] = dv[t];
        }
        __syncthreads();
        if (pb + PC < p_end) fetch(pb + PC);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pair = wave + 4 * q;
            if (pair < NI * NJ) {
                const int fi = pair / NJ, fj = pair % NJ;
                if constexpr (NI >= 4) {   // the 64- and 128-channel bodies sit at the 2-waves-per-SIMD register limit: they keep the plain order
#pragma unroll
                    for (int s4 = 0; s4 < PC / 4; ++s4) {
                        const float av = Ds[(fi * 16 + (lane & 15)) * LD + s4 * 4 + (lane >> 4)];
                        const float bv = Xs[(fj * 16 + (lane & 15)) * LD + s4 * 4 + (lane >> 4)];
                        acc[q] = mfma4(av, bv, acc[q]);
                    }
                } else {
                    const int d_rd = lds_rd + fi * 16 * LD, x_rd = lds_rd + fj * 16 * LD;
                    float af[2][WB], bf[2][WB];
#pragma unroll
                    for (int t = 0; t < WB; ++t) { af[0][t] = Ds[d_rd + t * 4]; bf[0][t] = Xs[x_rd + t * 4]; }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (b + 1 < NB) {
#pragma unroll
                            for (int t = 0; t < WB; ++t) { af[(b + 1) & 1][t] = Ds[d_rd + ((b + 1) * WB + t) * 4]; bf[(b + 1) & 1][t] = Xs[x_rd + ((b + 1) * WB + t) * 4]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < WB; ++t) acc[q] = mfma4(af[b & 1][t], bf[b & 1][t], acc[q]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    // partial[chunk][n][c][tap]: layout identical to the weight tensor
    constexpr int TT = KS * KS;   // taps of the full filter (the layout of the partial = the weight tensor's)
    float* ws = a.ws + (size_t)chunk * a.Cout * a.Cin * TT;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int pair = wave + 4 * q;
        if (pair < NI * NJ) {
            const int fi = pair / NJ, fj = pair % NJ;
            const int k = fj * 16 + (lane & 15);
            const int cl = k / T, tap = k % T;
            if (k < KT && c0 + cl < a.Cin) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = fi * 16 + (lane >> 4) * 4 + r;
                    if (n < a.Cout) ws[((size_t)n * a.Cin + c0 + cl) * TT + (ROWS ? krow * KS + tap : tap)] = acc[q][r];
                }
            }
        }
    }
}

// dw[layer][i] (+)= sum_chunk ws[layer][chunk][i] for every conv layer of a backward in one launch
// n_img > 1 with p_nstride > 0: independent images -- blockIdx.y = image, which sums only ITS chunks (chunk index = image *
// chunks_per_image + k) into its own gradient arena
// one chain of a layer's chunk sum: 4 independent partial sums in a FIXED association order, 16 loads in flight
template <class Load>
__device__ __forceinline__ float wgrad_chunk_sum(int chunks, Load&& ld) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 15 < chunks; c += 16) {   // 16 loads in flight, added in the order of the 4-wide loop below (same bits)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld(c + u);
#pragma unroll
        for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
    }
    for (; c + 3 < chunks; c += 4) {
        s0 += ld(c);
        s1 += ld(c + 1);
        s2 += ld(c + 2);
        s3 += ld(c + 3);
    }
    for (; c < chunks; ++c) s0 += ld(c);
    return (s0 + s1) + (s2 + s3);   // chunks == 0: an exact-zero gradient range (BN-fed conv bias)
}
__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(WgradReduceAll d, const float* __restrict__ ws, float* __restrict__ grads,
                                                               int accumulate, int n_img, size_t p_nstride) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= d.total) return;
    int l = 0;
#pragma unroll 1
    while (l + 1 < d.count && gid >= d.prefix[l + 1]) ++l;
    const bool vec = d.vec[l] != 0;
    const int i = (int)(gid - d.prefix[l]) * (vec ? 4 : 1);
    const int n = d.n[l];
    int chunks = d.chunks[l];
    const float* p = ws + d.ws_off[l] + i;
    if (p_nstride) {
        chunks /= n_img;
        p += (size_t)blockIdx.y * chunks * n;
        grads += (size_t)blockIdx.y * p_nstride;
    }
    float* q = grads + d.dw_off[l] + i;
    if (vec) {
        // four neighbouring elements per thread through 16-byte loads (round 4: the reduction streams 15 MB of partials per image and
        // was latency-bound on 4-byte accesses); every element still sums its chunks in the order of the scalar path: same bits
        float4 s = {0.f, 0.f, 0.f, 0.f};
        float4 a0 = s, a1 = s, a2 = s, a3 = s;
        int c = 0;
        for (; c + 7 < chunks; c += 8) {   // (8 x 16 B in flight; chains by c mod 4 as in wgrad_chunk_sum)
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(c + u) * n);
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
                a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w;
                a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
                a2.x += v[u + 2].x; a2.y += v[u + 2].y; a2.z += v[u + 2].z; a2.w += v[u + 2].w;
                a3.x += v[u + 3].x; a3.y += v[u + 3].y; a3.z += v[u + 3].z; a3.w += v[u + 3].w;
            }
        }
        for (; c + 3 < chunks; c += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)c * n), v1 = *reinterpret_cast<const float4*>(p + (size_t)(c + 1) * n),
                         v2 = *reinterpret_cast<const float4*>(p + (size_t)(c + 2) * n), v3 = *reinterpret_cast<const float4*>(p + (size_t)(c + 3) * n);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; c < chunks; ++c) {
            const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)c * n);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
        s.x = (a0.x + a1.x) + (a2.x + a3.x); s.y = (a0.y + a1.y) + (a2.y + a3.y);
        s.z = (a0.z + a1.z) + (a2.z + a3.z); s.w = (a0.w + a1.w) + (a2.w + a3.w);
        float4* q4 = reinterpret_cast<float4*>(q);
        if (accumulate) { const float4 o = *q4; s.x = o.x + s.x; s.y = o.y + s.y; s.z = o.z + s.z; s.w = o.w + s.w; }
        *q4 = s;
        return;
    }
    const float s = wgrad_chunk_sum(chunks, [&](int c) { return p[(size_t)c * n]; });
    *q = accumulate ? *q + s : s;
}

int wgrad_chunks(int N, int Ho, int Wo, int* pix_per_chunk, int* chunks_per_img) {
    const int HWo = Ho * Wo;
    // tuned in-step with alternating runs (512 / 256 / 64; 1024 or 256 for the big planes and 128 or 512 for the middle ones lose 0.2-0.4 %)
    int ppc = 512;
    if (HWo <= 1024) ppc = 256;
    if (HWo <= 256) ppc = 64;
    // big planes (round 5: the reference's default 900 x 900 crops): at 512 pixels a plane of 0.81 MP made 1582 chunks, every one writing a full
    // weight-shaped partial -- 0.84 ms of wgrad_reduce_all per step just to stream them back.  At most 128 chunks per image: a workgroup walks
    // more 64-pixel fills before it writes (planes up to 256 x 256 keep their 512-pixel chunks, so the 224 x 224 configs keep their bits).
    if (HWo > 128 * 512) ppc = (cdiv(HWo, 128) + 63) / 64 * 64;
    *pix_per_chunk = ppc;
    *chunks_per_img = cdiv(HWo, ppc);
    return N * *chunks_per_img;
}

// Every conv layer's weight gradient of a backward in ONE launch: the layers are independent of each other (each needs
// only its own input and output gradient, all in place once the dgrad chain has finished), so instead of ~30 small
// serial kernels the workgroups of all layers fill the chip together.  Workgroup -> (layer, pixel chunk, channel tile)
// through the prefix table of the by-value descriptor array; the (filter size, output-channel fragments) variant is a
// workgroup-uniform switch.
// Two instantiations: BIG = false runs the layers with <= 32 output channels (1 or 2 fragments: ~60 VGPRs, 21 KB of LDS,
// several workgroups per CU -- these are the layers with thousands of workgroups), BIG = true the 64 / 128-channel ones
// (up to 229 VGPRs).  One kernel for everything ran the small layers at the big variant's occupancy.
template <bool BIG>
__global__ __launch_bounds__(256, 2) void conv_wgrad_batched_kernel(WgradBatch b) {
    __shared__ float Xs[WG_XS_FLOATS];
    __shared__ float Ds[(BIG ? 8 : 2) * 16 * WG_LD];
    int l = 0;
#pragma unroll 1
    while (l + 1 < b.count && blockIdx.x >= b.d[l + 1].wg_begin) ++l;
    const WgradDesc& d = b.d[l];
    WgradArgs a;
    a.x = d.x; a.dy = d.dy; a.ws = d.ws;
    a.x_nstride = d.x_nstride; a.x_cstride = d.x_cstride; a.dy_nstride = d.dy_nstride; a.dy_cstride = d.dy_cstride;
    a.N = 0; a.Cin = d.Cin; a.Hi = d.Hi; a.Wi = d.Wi; a.Cout = d.Cout; a.Ho = d.Ho; a.Wo = d.Wo;
    a.ks = d.ks; a.stride = d.stride; a.pad = d.pad; a.pix_per_chunk = d.pix_per_chunk; a.chunks_per_img = d.chunks_per_img;
    a.reflect = (int)d.reflect;
    const int local = blockIdx.x - d.wg_begin;
    const int chunk = local % d.chunks, ktile = local / d.chunks;
    if (BIG) {
        switch (d.variant) {
            case 2: conv_wgrad_body<1, 4>(a, chunk, ktile, Xs, Ds); break;
            case 3: conv_wgrad_body<1, 8>(a, chunk, ktile, Xs, Ds); break;
            case 6: conv_wgrad_body<3, 4>(a, chunk, ktile, Xs, Ds); break;
            case 7: conv_wgrad_body<3, 8>(a, chunk, ktile, Xs, Ds); break;
            case 10: conv_wgrad_body<5, 4, true>(a, chunk, ktile, Xs, Ds); break;
            case 11: conv_wgrad_body<5, 8, true>(a, chunk, ktile, Xs, Ds); break;
            case 14: conv_wgrad_body<7, 4, true>(a, chunk, ktile, Xs, Ds); break;
            default: conv_wgrad_body<7, 8, true>(a, chunk, ktile, Xs, Ds); break;
        }
    } else {
        switch (d.variant) {
            case 0: conv_wgrad_body<1, 1>(a, chunk, ktile, Xs, Ds); break;
            case 1: conv_wgrad_body<1, 2>(a, chunk, ktile, Xs, Ds); break;
            case 4: conv_wgrad_body<3, 1>(a, chunk, ktile, Xs, Ds); break;
            case 5: conv_wgrad_body<3, 2>(a, chunk, ktile, Xs, Ds); break;
            case 8: conv_wgrad_body<5, 1, true>(a, chunk, ktile, Xs, Ds); break;
            case 9: conv_wgrad_body<5, 2, true>(a, chunk, ktile, Xs, Ds); break;
            case 12: conv_wgrad_body<7, 1, true>(a, chunk, ktile, Xs, Ds); break;
            default: conv_wgrad_body<7, 2, true>(a, chunk, ktile, Xs, Ds); break;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Weight gradient of the 3x3 stride-1 layers of BIG planes (round 5), the counterpart of conv3x3_tile_kernel: a workgroup = (strip of 4 x 64-pixel tiles,
// 8-channel K tile).  Per tile the (4 + 2) x (64 + 2) x 8 input patch is staged in LDS once (interior tiles: fixed per-thread offsets + the tile as the
// scalar offset; border tiles resolve padding / reflection per element); a wave owns one pixel row, takes its output-gradient operand STRAIGHT from
// global memory (16-byte loads: lane (n, g) holds dy[n][16 q + 4 g .. + 3], four MFMA steps per load) and its input operand from the patch at
// [per-lane (channel, tap) offset + 4 g] + [compile-time pixel offset]; the 72 (channel, tap) columns are 5 fragments, the accumulators stay in registers
// over the whole strip, the four waves' sums are added in wave order through LDS at the end.  conv_wgrad_body fills LDS with 64 pixels x 36 k-values and
// 64 x Cout gradients for 16 MFMA steps per fragment pair (and leaves one wave idle at 16 output channels).
constexpr int WT_CK = 8, WT_NJ = 5, WT_PH = 6, WT_PLANE = WT_PH * CT_PW, WT_PE = WT_CK * WT_PLANE, WT_NP = (WT_PE + 255) / 256;
template <int NI>
__device__ __forceinline__ void conv_wgrad_tile_body(const WgradDesc& d, int chunk, int ktile, int nb /* first output channel of this workgroup */, float* Ps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = d.Cin, Cout = d.Cout, Hi = d.Hi, Wi = d.Wi, Ho = d.Ho, Wo = d.Wo, pad = d.pad;
    const int img = chunk / d.chunks_per_img, strip = chunk % d.chunks_per_img;
    const int tiles_x = (Wo + CT_TW - 1) / CT_TW, tiles_y = (Ho + 3) / 4, T = tiles_x * tiles_y;
    const int t_begin = (int)((long long)strip * T / d.chunks_per_img), t_end = (int)((long long)(strip + 1) * T / d.chunks_per_img);
    const int c0 = ktile * WT_CK;
    const float* x = d.x + (size_t)img * d.x_nstride + (size_t)c0 * d.x_cstride;
    const float* dy = d.dy + (size_t)img * d.dy_nstride;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, 0x7FFFFFFF, 0x00020000);
    // patch element e = tid + 256 j = (channel, patch row, patch column), relative to the tile's patch origin
    int p_rel[WT_NP];
#pragma unroll
    for (int j = 0; j < WT_NP; ++j) {
        const int e = tid + 256 * j;
        const int c = e / WT_PLANE, rem = e % WT_PLANE, r = rem / CT_PW, xx = rem % CT_PW;
        p_rel[j] = (e < WT_PE && c0 + c < Cin) ? (int)(c * d.x_cstride + (unsigned)(r * Wi + xx)) * 4 : (int)0x80000000;
    }
    // input operand: column j = jf * 16 + (lane & 15) = (channel, tap) -> offset inside the patch (+ this wave's row, + the lane's pixel group)
    int b_base[WT_NJ];
#pragma unroll
    for (int jf = 0; jf < WT_NJ; ++jf) {
        const int j = jf * 16 + (lane & 15);
        const int cl = j / 9, tap = j % 9;
        b_base[jf] = (j < WT_CK * 9 ? cl * WT_PLANE + (tap / 3) * CT_PW + tap % 3 : 0) + wave * CT_PW + 4 * (lane >> 4);
    }
    int dy_off[NI];
#pragma unroll
    for (int fi = 0; fi < NI; ++fi) {
        const int n = nb + fi * 16 + (lane & 15);
        dy_off[fi] = n < Cout ? (int)(n * d.dy_cstride) * 4 : (int)0x80000000;
    }
    f32x4 acc[NI][WT_NJ];
#pragma unroll
    for (int fi = 0; fi < NI; ++fi)
#pragma unroll
        for (int jf = 0; jf < WT_NJ; ++jf) acc[fi][jf] = f32x4{0.f, 0.f, 0.f, 0.f};
    float pv[WT_NP];
    float dyn[NI][4][4], dyc[NI][4][4];
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int y0 = ty * 4, x0 = tx * CT_TW, sy0 = y0 - pad, sx0 = x0 - pad;
        const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + WT_PH <= Hi && sx0 + CT_PW <= Wi;
        if (interior) {
            const int so = __builtin_amdgcn_readfirstlane((sy0 * Wi + sx0) * 4);
#pragma unroll
            for (int j = 0; j < WT_NP; ++j) pv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, p_rel[j], so, 0));
        } else {
#pragma unroll
            for (int j = 0; j < WT_NP; ++j) {
                const int e = tid + 256 * j;
                const int c = e / WT_PLANE, rem = e % WT_PLANE, r = rem / CT_PW, xx = rem % CT_PW;
                int sy = sy0 + r, sx = sx0 + xx;
                if (d.reflect) {
                    sy = sy < 0 ? -sy : (sy >= Hi ? 2 * (Hi - 1) - sy : sy);
                    sx = sx < 0 ? -sx : (sx >= Wi ? 2 * (Wi - 1) - sx : sx);
                }
                const bool ok = p_rel[j] >= 0 && sy >= 0 && sy < Hi && sx >= 0 && sx < Wi;
                pv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (int)(c * d.x_cstride + (unsigned)(sy * Wi + sx)) * 4 : (int)0x80000000, 0, 0));
            }
        }
        // this wave's row of the output gradient: 16-byte runs (4-byte aligned); pixels behind the row end / rows behind the plane read as 0
        const int row = y0 + wave;
        const int col0 = x0 + 4 * (lane >> 4);
#pragma unroll
        for (int fi = 0; fi < NI; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = col0 + 16 * q;
                const bool ok = row < Ho && col < Wo && dy_off[fi] >= 0;
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? dy_off[fi] + (row * Wo + col) * 4 : (int)0x80000000, 0, 0));
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) dyn[fi][q][tt] = col + tt < Wo ? __uint_as_float(v[tt]) : 0.f;
            }
    };
    if (t_begin < t_end) fetch(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("" : "+v"(b_base[0]), "+v"(b_base[1]), "+v"(b_base[2]), "+v"(b_base[3]), "+v"(b_base[4]));   // (LDS bases re-defined per trip: base + instruction offset)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WT_NP; ++j)
            if (tid + 256 * j < WT_PE) Ps[tid + 256 * j] = pv[j];
#pragma unroll
        for (int fi = 0; fi < NI; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) dyc[fi][q][tt] = dyn[fi][q][tt];
        __syncthreads();
        if (t + 1 < t_end) fetch(t + 1);
        float bv[2][WT_NJ];
#pragma unroll
        for (int jf = 0; jf < WT_NJ; ++jf) bv[0][jf] = Ps[b_base[jf]];
#pragma unroll
        for (int st = 0; st < 16; ++st) {   // pixel step: column 16 (st / 4) + 4 g + st % 4 of the wave's row
            if (st + 1 < 16) {
#pragma unroll
                for (int jf = 0; jf < WT_NJ; ++jf) bv[(st + 1) & 1][jf] = Ps[b_base[jf] + 16 * ((st + 1) >> 2) + ((st + 1) & 3)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int fi = 0; fi < NI; ++fi)
#pragma unroll
                for (int jf = 0; jf < WT_NJ; ++jf) acc[fi][jf] = mfma4(dyc[fi][st >> 2][st & 3], bv[st & 1][jf], acc[fi][jf]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- the four waves' sums, added in wave order (fixed: bit-reproducible), then the partial of this (chunk, K tile): layout of the weight tensor
    __syncthreads();
    constexpr int NV = NI * WT_NJ * 4;
    if (wave > 0) {
#pragma unroll
        for (int fi = 0; fi < NI; ++fi)
#pragma unroll
            for (int jf = 0; jf < WT_NJ; ++jf)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ps[((wave - 1) * NV + (fi * WT_NJ + jf) * 4 + r) * 64 + lane] = acc[fi][jf][r];
    }
    __syncthreads();
    if (wave > 0) return;
    float* ws = d.ws + (size_t)chunk * Cout * Cin * 9;
#pragma unroll
    for (int fi = 0; fi < NI; ++fi)
#pragma unroll
        for (int jf = 0; jf < WT_NJ; ++jf) {
            const int j = jf * 16 + (lane & 15);
            const int cl = j / 9, tap = j % 9;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[fi][jf][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += Ps[(w * NV + (fi * WT_NJ + jf) * 4 + r) * 64 + lane];
                const int n = nb + fi * 16 + (lane >> 4) * 4 + r;
                if (j < WT_CK * 9 && c0 + cl < Cin && n < Cout) ws[((size_t)n * Cin + c0 + cl) * 9 + tap] = v;
            }
        }
}
constexpr int WT_LDS_FLOATS = 3 * 2 * WT_NJ * 4 * 64 > WT_PE ? 3 * 2 * WT_NJ * 4 * 64 : WT_PE;
__global__ __launch_bounds__(256) void conv_wgrad_tile_kernel(WgradBatch b) {
    __shared__ float Ps[WT_LDS_FLOATS];
    int l = 0;
#pragma unroll 1
    while (l + 1 < b.count && blockIdx.x >= b.d[l + 1].wg_begin) ++l;
    const WgradDesc& d = b.d[l];
    const int local = blockIdx.x - d.wg_begin;
    // (chunk, K tile, block of 32 output channels): layers with 64 / 128 output channels are 2 / 4 passes over the same patches
    const int chunk = local % d.chunks, kt_all = local / d.chunks, ktiles_c = (d.Cin + WT_CK - 1) / WT_CK;
    const int ktile = kt_all % ktiles_c, nb = (kt_all / ktiles_c) * 32;
    if (d.variant == 1) conv_wgrad_tile_body<1>(d, chunk, ktile, nb, Ps);
    else conv_wgrad_tile_body<2>(d, chunk, ktile, nb, Ps);
}
static bool wgrad_tile_ok(const WgradArgs& a) {
    static const int on = getenv("SPLICE_WGRAD_TILE") ? atoi(getenv("SPLICE_WGRAD_TILE")) : 1;
    constexpr int min_px = 40000;
    return on && a.ks == 3 && a.stride == 1 && a.Wo >= 64 && (long long)a.Ho * a.Wo > min_px && a.Hi == a.Ho && a.Wi == a.Wo && a.pad == 1 &&
           (size_t)a.Cin * a.x_cstride <= 0x1fffffffULL && (size_t)a.Cout * a.dy_cstride <= 0x1fffffffULL;   // (32-bit byte offsets)
}

// append one layer to a batch (partial sums only: ws gets chunks * Cout*Cin*ks*ks floats); returns the number of chunks
int conv_wgrad_add(WgradBatchPair* pair, WgradArgs a, int* chunks_out) {
    const bool tile = wgrad_tile_ok(a);
    WgradBatch* b = tile ? &pair->tile : a.Cout > 32 ? &pair->big : &pair->small;
    if (a.Cout > 128 || (a.ks != 1 && a.ks != 3 && a.ks != 5 && a.ks != 7) || b->count >= WGRAD_BATCH_MAX) return SPLICE_ERR_ARG;
    if ((size_t)a.Cin * a.x_cstride > 0x7fffffffULL || a.x_nstride > 0xffffffffULL || a.dy_nstride > 0xffffffffULL) return SPLICE_ERR_ARG;
    if (a.Hi > 65535 || a.Wi > 65535 || a.Cin > 65535) return SPLICE_ERR_ARG;
    const int chunks = wgrad_chunks(a.N, a.Ho, a.Wo, &a.pix_per_chunk, &a.chunks_per_img);
    const int CK = tile ? WT_CK : a.ks == 1 ? 16 : 4;
    const int ktiles = cdiv(a.Cin, CK) * (a.ks >= 5 ? a.ks : 1) * (tile ? cdiv(a.Cout, 32) : 1);   // 5x5 / 7x7: one K tile per (channel tile, filter row); tile kernel: x blocks of 32 output channels
    const int ni = cdiv(a.Cout, 16);
    WgradDesc& d = b->d[b->count++];
    d.x = a.x; d.dy = a.dy; d.ws = a.ws;
    d.x_nstride = (uint32_t)a.x_nstride; d.x_cstride = (uint32_t)a.x_cstride; d.dy_nstride = (uint32_t)a.dy_nstride; d.dy_cstride = (uint32_t)a.dy_cstride;
    d.Cin = (uint16_t)a.Cin; d.Cout = (uint16_t)a.Cout; d.Hi = (uint16_t)a.Hi; d.Wi = (uint16_t)a.Wi; d.Ho = (uint16_t)a.Ho; d.Wo = (uint16_t)a.Wo;
    d.ks = (uint8_t)a.ks; d.stride = (uint8_t)a.stride; d.pad = (uint8_t)a.pad;
    d.variant = (uint8_t)((a.ks == 3 ? 4 : a.ks == 5 ? 8 : a.ks == 7 ? 12 : 0) + (ni <= 1 ? 0 : ni <= 2 ? 1 : ni <= 4 ? 2 : 3));
    if (tile) d.variant = (uint8_t)(ni < 2 ? 1 : 2);   // 1 or 2 fragments of output channels per workgroup
    d.reflect = (uint32_t)(a.reflect ? 1 : 0);
    d.pix_per_chunk = (uint16_t)a.pix_per_chunk; d.chunks_per_img = (uint16_t)a.chunks_per_img;
    d.wg_begin = (uint32_t)b->total_wgs; d.chunks = (uint32_t)chunks;
    b->total_wgs += chunks * ktiles;
    if (chunks_out) *chunks_out = chunks;
    return SPLICE_OK;
}
int conv_wgrad_batched_launch(const WgradBatchPair& p, hipStream_t s) {
    if (p.big.count > 0) SPLICE_LAUNCH(conv_wgrad_batched_kernel<true>, dim3((unsigned)p.big.total_wgs), dim3(256), 0, s, p.big);
    if (p.small.count > 0) SPLICE_LAUNCH(conv_wgrad_batched_kernel<false>, dim3((unsigned)p.small.total_wgs), dim3(256), 0, s, p.small);
    if (p.tile.count > 0) SPLICE_LAUNCH(conv_wgrad_tile_kernel, dim3((unsigned)p.tile.total_wgs), dim3(256), 0, s, p.tile);
    return SPLICE_OK;
}

int wgrad_reduce_all_launch(const WgradReduceAll& d0, const float* ws, float* grads, int accumulate, hipStream_t s, int n_img, size_t p_nstride) {
    if (d0.count < 1 || d0.count > WGRAD_MAX_LAYERS) return SPLICE_ERR_ARG;
    // work items: four elements per thread wherever a layer's ranges are 16-byte aligned (every layer but the 3-channel head's bias)
    WgradReduceAll d = d0;
    static const int vec_on = getenv("SPLICE_WGRAD_REDUCE_VEC") ? atoi(getenv("SPLICE_WGRAD_REDUCE_VEC")) : 1;
    const bool base_ok = vec_on && !((reinterpret_cast<size_t>(ws) | reinterpret_cast<size_t>(grads)) & 15) && p_nstride % 4 == 0;
    d.prefix[0] = 0;
    for (int i = 0; i < d.count; ++i) {
        const bool v = base_ok && d.n[i] % 4 == 0 && d.ws_off[i] % 4 == 0 && d.dw_off[i] % 4 == 0;
        d.vec[i] = v ? 1 : 0;
        d.prefix[i + 1] = d.prefix[i] + (v ? d.n[i] / 4 : d.n[i]);
    }
    d.total = d.prefix[d.count];
    SPLICE_LAUNCH(wgrad_reduce_all_kernel, dim3((unsigned)((d.total + 255) / 256), p_nstride ? n_img : 1), dim3(256), 0, s, d, ws, grads, accumulate,
                       n_img, p_nstride);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
// block-wide fixed-order sum of two values (256 threads)
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[4 + w] = b; }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
}

// Plane reductions run in two deterministic stages so that planes with few channels still fill the
// chip: stage 1 = PB workgroups per (image, channel) plane write partials; stage 2 = the consuming
// element-wise kernel recombines the <= 64 partials in a fixed order in its prologue.
constexpr int MAX_PB = 64;
// Planes of more than 64 x 1024 pixels (every 448^2 / 512^2 / 900^2 layer; round 5): up to MAX_PB_V segments of <= BN_V_CH x 1024 pixels whose
// length is a multiple of 4, so that a thread owns runs of 4 consecutive pixels and moves them as ONE 16-byte access (4-byte aligned:
// gfx950's global accesses need dword alignment only, so odd plane sizes -- the random 855 .. 900 crops -- need no peeling).  A segment
// count above MAX_PB is what selects that layout everywhere (seg_len, the combines); the 224^2 layers keep theirs, bit for bit.
constexpr int MAX_PB_V = 256;
constexpr int BN_V_CH = 5;   // 16-byte runs a thread holds: 5 x 4 x 256 = 5120 pixels per segment at most (1.31 M pixel planes)
__host__ __device__ inline int seg_len(int HW, int PB) {
    const int s = (HW + PB - 1) / PB;
    return PB > MAX_PB ? (s + 3) & ~3 : s;
}
// fixed-order sum of component `comp` of the PB per-segment pairs of a plane by one wave (PB <= 64: lane b holds segment b, as before)
__device__ __forceinline__ float part_sum(const float* __restrict__ pp, int PB, int comp, int lane) {
    if (PB <= MAX_PB) return wave_sum(lane < PB ? pp[2 * lane + comp] : 0.f);
    float a = 0.f;
    for (int b = lane; b < PB; b += 64) a += pp[2 * b + comp];
    return wave_sum(a);
}

// x2 bilinear (align_corners=False) source coordinates of output o, and one upsampled value from a plane p[h][w] (global
// or LDS): the expression order of upsample2x_fwd_kernel
__device__ __forceinline__ void up_coord(int o, int n, int& i0, int& i1, float& lam) {
    float src = ((float)o + 0.5f) * 0.5f - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    lam = src - (float)i0;
}
template <class P>
__device__ __forceinline__ float up_value(P p, int h, int w, int oy, int ox) {
    int y0, y1, x0, x1;
    float ly, lx;
    up_coord(oy, h, y0, y1, ly);
    up_coord(ox, w, x0, x1, lx);
    const float top = p[y0 * w + x0] * (1.f - lx) + p[y0 * w + x1] * lx;
    const float bot = p[y1 * w + x0] * (1.f - lx) + p[y1 * w + x1] * lx;
    return top * (1.f - ly) + bot * ly;
}
__device__ __forceinline__ void up_adjoint_weights(int m, int n, int No, float (&wt)[4]) {
    wt[0] = m > 0 ? 0.25f : 0.f;                  // o = 2m-1 (odd output of input m-1, upper neighbour = m)
    wt[1] = m > 0 ? 0.75f : 1.0f;                 // o = 2m   (source m - 1/4, clamped to 0 at the border)
    wt[2] = m < n - 1 ? 0.75f : 1.0f;             // o = 2m+1 (source m + 1/4, upper neighbour clamped to n-1)
    wt[3] = m < n - 1 ? 0.25f : 0.f;              // o = 2m+2 (even output of input m+1, lower neighbour = m)
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (2 * m - 1 + t >= No) wt[t] = 0.f;
}
// adjoint of the above for input pixel (my, mx), read from the output-sized gradient plane p[Ho][Wo] (upsample2x_bwd_kernel)
template <class P>
__device__ __forceinline__ float up_adjoint_value(P p, int h, int w, int Ho, int Wo, int my, int mx) {
    float wy[4], wx[4];
    up_adjoint_weights(my, h, Ho, wy);
    up_adjoint_weights(mx, w, Wo, wx);
    float t[4][4];
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
        const int oy = min(max(2 * my - 1 + ty, 0), Ho - 1);
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) t[ty][tx] = p[oy * Wo + min(max(2 * mx - 1 + tx, 0), Wo - 1)];
    }
    float acc = 0.f;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
        float row = 0.f;
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) row += wx[tx] != 0.f ? wx[tx] * t[ty][tx] : 0.f;
        acc += wy[ty] != 0.f ? wy[ty] * row : 0.f;
    }
    return acc;
}
// stage 1 of the BN statistics: per-segment (count, mean, M2), two-pass inside the segment
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* y, size_t nstride, int C, int HW, int PB,
                                                               float* __restrict__ part /* [N][C][PB][2] */, BnUpsample up) {
    __shared__ float red[8];
    const int pb = blockIdx.x, c = blockIdx.y, img = blockIdx.z;
    const int seg = seg_len(HW, PB), lo = pb * seg, hi = min(lo + seg, HW);
    const float* p = y + (size_t)img * nstride + (size_t)c * HW;
    float s = 0.f, dummy = 0.f;
    // segments of up to 1024 elements (every plane up to 256 x 256) stay in registers between the two passes: the second pass
    // used to re-read them (a dependent L2 round trip per workgroup); same values, same order of additions, same bits
    const bool in_regs = seg <= 1024;
    float keep[4] = {0.f, 0.f, 0.f, 0.f};
    if (up.src && c >= up.c0) {   // upsampled channel: the values are produced here (and stored: the apply kernel and the backward read them)
        const float* sp = up.src + (size_t)img * up.src_ns + (size_t)(c - up.c0) * up.h * up.w;
        float* yo = const_cast<float*>(p);
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lo + threadIdx.x + k * 256;
                if (i < hi) {
                    const float v = up_value(sp, up.h, up.w, i / up.Wo, i % up.Wo);
                    yo[i] = v;
                    keep[k] = v;
                    s += v;
                }
            }
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += 256) {
                const float v = up_value(sp, up.h, up.w, i / up.Wo, i % up.Wo);
                yo[i] = v;
                s += v;
            }
        }
    } else if (in_regs) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lo + threadIdx.x + k * 256;
            keep[k] = i < hi ? p[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lo + threadIdx.x + k * 256 < hi) s += keep[k];
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) s += p[i];
    }
    block_sum2(s, dummy, red);
    const int cnt = hi - lo;
    const float m = cnt > 0 ? s / (float)cnt : 0.f;
    float sq = 0.f;
    dummy = 0.f;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lo + threadIdx.x + k * 256 < hi) { const float d = keep[k] - m; sq = __builtin_fmaf(d, d, sq); }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) { const float d = p[i] - m; sq = __builtin_fmaf(d, d, sq); }
    }
    block_sum2(sq, dummy, red);
    if (threadIdx.x == 0) {
        float* o = part + (((size_t)img * C + c) * PB + pb) * 2;
        o[0] = m;
        o[1] = sq;
    }
}

// the same stage for the big planes (PB > MAX_PB): the segment sits in registers as BN_V_CH runs of 4 pixels per thread, loaded (or, for an
// upsampled channel, produced and stored) with 16-byte accesses that are all in flight before the first sum
__global__ __launch_bounds__(256) void bn_stats_partial_v_kernel(const float* y, size_t nstride, int C, int HW, int PB,
                                                                 float* __restrict__ part /* [N][C][PB][2] */, BnUpsample up) {
    __shared__ float red[8];
    const int pb = blockIdx.x, c = blockIdx.y, img = blockIdx.z;
    const int seg = seg_len(HW, PB), lo = pb * seg, hi = min(lo + seg, HW);
    const float* p = y + (size_t)img * nstride + (size_t)c * HW;
    float keep[BN_V_CH][4];
    if (up.src && c >= up.c0) {
        const float* sp = up.src + (size_t)img * up.src_ns + (size_t)(c - up.c0) * up.h * up.w;
        float* yo = const_cast<float*>(p);
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k) {
            const int i = lo + 4 * (threadIdx.x + 256 * k);
            int oy = i / up.Wo, ox = i - oy * up.Wo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                keep[k][j] = i + j < hi ? up_value(sp, up.h, up.w, oy, ox) : 0.f;
                if (++ox == up.Wo) { ox = 0; ++oy; }
            }
            if (i < hi) st_run(yo, i, hi, keep[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k) ld_run(p, lo + 4 * (threadIdx.x + 256 * k), hi, keep[k]);
    }
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int k = 0; k < BN_V_CH; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += keep[k][j];   // (pixels behind `hi` are zeros)
    block_sum2(s, dummy, red);
    const int cnt = hi - lo;
    const float m = cnt > 0 ? s / (float)cnt : 0.f;
    float sq = 0.f;
    dummy = 0.f;
#pragma unroll
    for (int k = 0; k < BN_V_CH; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lo + 4 * ((int)threadIdx.x + 256 * k) + j < hi) { const float d = keep[k][j] - m; sq = __builtin_fmaf(d, d, sq); }
    block_sum2(sq, dummy, red);
    if (threadIdx.x == 0) {
        float* o = part + (((size_t)img * C + c) * PB + pb) * 2;
        o[0] = m;
        o[1] = sq;
    }
}

// Chan et al. pairwise combination of the <= 64 segment statistics by ONE wave: lane b holds segment b, a fixed
// binary tree (lane l absorbs lane l + off, off = 32 .. 1) leaves the plane's (mean, M2) in lane 0.  Deterministic, and
// ~100 cycles instead of a 49-step dependent chain in front of every workgroup of the apply kernel.
__device__ __forceinline__ void bn_combine_wave_raw(const float* part, int PB, int HW, float& mean, float& M2out);
__device__ __forceinline__ void bn_combine_wave(const float* part, int PB, int HW, float eps, float& mean, float& rstd) {
    float M2;
    bn_combine_wave_raw(part, PB, HW, mean, M2);
    rstd = rsqrtf(M2 / (float)HW + eps);
}
// (mean, M2) of one plane from its <= 64 segment statistics; every lane returns lane 0's result
__device__ __forceinline__ void bn_combine_wave_raw(const float* part, int PB, int HW, float& mean, float& M2out) {
    const int lane = threadIdx.x & 63;
    const int seg = seg_len(HW, PB), lo = lane * seg;
    int cnt = lane < PB ? min(lo + seg, HW) - lo : 0;
    cnt = cnt > 0 ? cnt : 0;
    float n = (float)cnt, m = cnt > 0 ? part[2 * lane] : 0.f, M2 = cnt > 0 ? part[2 * lane + 1] : 0.f;
    if (PB > MAX_PB) {   // big planes: lane l first merges its G consecutive segments l G .. l G + G - 1 in order, then the same tree
        const int G = (PB + 63) >> 6;
        n = 0.f; m = 0.f; M2 = 0.f;
        for (int g = 0; g < G; ++g) {
            const int b = lane * G + g, blo = b * seg;
            const int bc = b < PB ? min(blo + seg, HW) - blo : 0;
            if (bc > 0) {
                const float nb = (float)bc, mb = part[2 * b], Mb = part[2 * b + 1];
                const float nt = n + nb, d = mb - m, w = nb / nt;
                m += d * w;
                M2 += Mb + d * d * n * w;
                n = nt;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float nb = __shfl_down(n, off, 64), mb = __shfl_down(m, off, 64), Mb = __shfl_down(M2, off, 64);
        const float nt = n + nb;
        if (nb > 0.f) {
            const float d = mb - m, w = nb / nt;
            m += d * w;
            M2 += Mb + d * d * n * w;
            n = nt;
        }
    }
    mean = __shfl(m, 0, 64);
    M2out = __shfl(M2, 0, 64);
}
// batch statistics (nn.BatchNorm2d over a batch of N images, models/unet/common.py:95-96 when netG is fed n_crops > 1
// crops at once): Chan merge of the N planes' (mean, M2) in image order
__device__ __forceinline__ void bn_combine_batch(const float* part_c0 /* image 0, channel c */, size_t img_stride, int N, int PB, int HW, float eps,
                                                 float& mean, float& rstd) {
    float n = 0.f, m = 0.f, M2 = 0.f;
    for (int i = 0; i < N; ++i) {
        float mb, Mb;
        bn_combine_wave_raw(part_c0 + (size_t)i * img_stride, PB, HW, mb, Mb);
        const float nb = (float)HW, nt = n + nb, d = mb - m, w = nb / nt;
        m += d * w;
        M2 += Mb + d * d * n * w;
        n = nt;
    }
    mean = m;
    rstd = rsqrtf(M2 / n + eps);
}

// the affine map of the big-plane forward with its two roundings pinned (one fma each), so that the backward can re-form the pre-activation
// value from y, mean, rstd, gamma, beta to the bit and read the activation's sign off it instead of loading the activated tensor
__device__ __forceinline__ float bn_shift(float beta, float mean, float sc) { return __builtin_fmaf(-mean, sc, beta); }
__device__ __forceinline__ float bn_affine(float y, float sc, float sh) { return __builtin_fmaf(y, sc, sh); }
// stage 2 + apply: a = act(gamma * (y - mean) * rstd + beta), written to a channel slice of `out`
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ y, size_t y_nstride, float* __restrict__ out,
                                                     size_t out_nstride, int C, int HW, int PB, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ part, float eps,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, float slope, size_t p_nstride, int batch) {
    __shared__ float st[2];
    const int c = blockIdx.y, img = blockIdx.z;
    gamma += (size_t)img * p_nstride; beta += (size_t)img * p_nstride;
    if (threadIdx.x < 64) {
        float m, r;
        if (batch) bn_combine_batch(part + (size_t)c * PB * 2, (size_t)C * PB * 2, gridDim.z, PB, HW, eps, m, r);
        else bn_combine_wave(part + ((size_t)img * C + c) * PB * 2, PB, HW, eps, m, r);
        if (threadIdx.x == 0) {
            st[0] = m; st[1] = r;
            if (blockIdx.x == 0) { mean_o[img * C + c] = m; rstd_o[img * C + c] = r; }
        }
    }
    __syncthreads();
    const float sc = gamma[c] * st[1];
    const float sh = beta[c] - st[0] * sc;
    const float* p = y + (size_t)img * y_nstride + (size_t)c * HW;
    float* q = out + (size_t)img * out_nstride + (size_t)c * HW;
    if (PB > MAX_PB) {   // big planes: the workgroup's own segment as 16-byte runs, all loads in flight before the first store
        const int seg = seg_len(HW, PB), lo = blockIdx.x * seg, hi = min(lo + seg, HW);
        float v[BN_V_CH][4];
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k) ld_run(p, lo + 4 * (threadIdx.x + 256 * k), hi, v[k]);
        const float shv = bn_shift(beta[c], st[0], sc);
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k) {
            const int i = lo + 4 * (threadIdx.x + 256 * k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = bn_affine(v[k][j], sc, shv);   // (the backward of these planes re-forms t from y to get the activation's sign)
                v[k][j] = t > 0.f ? t : t * slope;
            }
            if (i < hi) st_run(q, i, hi, v[k]);
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float v = p[i] * sc + sh;
        q[i] = v > 0.f ? v : v * slope;
    }
}

// BN backward stage 1: per-segment s1 = sum dz, s2 = sum dz * xhat, dz = da * act'(a)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ da, size_t da_nstride, const float* __restrict__ aout,
                                                             size_t a_nstride, const float* __restrict__ y, size_t y_nstride, int C, int HW,
                                                             int PB, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float slope, float* __restrict__ part /* [N][C][PB][2] */,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, size_t p_nstride) {
    __shared__ float red[8];
    const int pb = blockIdx.x, c = blockIdx.y, img = blockIdx.z;
    const int seg = seg_len(HW, PB), lo = pb * seg, hi = min(lo + seg, HW);
    const float m = mean[img * C + c], r = rstd[img * C + c];
    const float* pd = da + (size_t)img * da_nstride + (size_t)c * HW;
    const float* pa = aout + (size_t)img * a_nstride + (size_t)c * HW;
    const float* py = y + (size_t)img * y_nstride + (size_t)c * HW;
    float s1 = 0.f, s2 = 0.f;
    if (PB > MAX_PB) {
        const bool act = slope != 1.0f, from_y = act && beta != nullptr;   // the activation's sign from y (bn_affine): one tensor less to read
        float d[BN_V_CH][4], a[BN_V_CH][4], yy[BN_V_CH][4];
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k) {
            const int i = lo + 4 * (threadIdx.x + 256 * k);
            ld_run(pd, i, hi, d[k]);
            ld_run(py, i, hi, yy[k]);
            if (act && !from_y) ld_run(pa, i, hi, a[k]);
        }
        float sc = 0.f, sh = 0.f;
        if (from_y) { sc = gamma[(size_t)img * p_nstride + c] * r; sh = bn_shift(beta[(size_t)img * p_nstride + c], m, sc); }
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // (pixels behind `hi`: dz = 0 adds nothing to either sum)
                float dz = d[k][j];
                const float av = from_y ? bn_affine(yy[k][j], sc, sh) : a[k][j];
                if (act && !(av > 0.f)) dz *= slope;
                s1 += dz;
                s2 += dz * (yy[k][j] - m) * r;
            }
    } else
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float dz = pd[i];
        if (slope != 1.0f && !(pa[i] > 0.f)) dz *= slope;
        s1 += dz;
        s2 += dz * (py[i] - m) * r;
    }
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) {
        float* o = part + (((size_t)img * C + c) * PB + pb) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

// stage 2 + apply: dy = gamma * rstd * (dz - s1/HW - xhat * s2/HW); also dgamma/dbeta (sum over images)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ da, size_t da_nstride, const float* __restrict__ aout,
                                                           size_t a_nstride, const float* __restrict__ y, size_t y_nstride,
                                                           float* __restrict__ dy, size_t dy_nstride, int C, int HW, int N, int PB,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float slope, const float* __restrict__ part,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, size_t p_nstride, int batch,
                                                           const float* __restrict__ beta) {
    __shared__ float st[2];
    const int c = blockIdx.y, img = blockIdx.z;
    gamma += (size_t)img * p_nstride;
    if (threadIdx.x < 64) {   // one wave: lane k holds segment k (PB <= 64), fixed-order tree sums
        const int lane = threadIdx.x;
        const float* pp = part + ((size_t)img * C + c) * PB * 2;
        const float a = part_sum(pp, PB, 0, lane), b = part_sum(pp, PB, 1, lane);
        if (lane == 0) { st[0] = a; st[1] = b; }
        if (p_nstride) {   // independent images: every image owns its parameter gradients (the sums of the N = 1 path: 0 + x)
            if (blockIdx.x == 0 && lane == 0) {
                float* dg = dgamma + (size_t)img * p_nstride + c;
                float* db = dbeta + (size_t)img * p_nstride + c;
                const float g = 0.f + b, be = 0.f + a;
                *dg = accumulate ? *dg + g : g;
                *db = accumulate ? *db + be : be;
            }
        } else if (img == 0 && blockIdx.x == 0) {   // parameter gradients: sum over the images in order
            float g = 0.f, be = 0.f;
            for (int n = 0; n < N; ++n) {
                const float* pn = part + ((size_t)n * C + c) * PB * 2;
                be += part_sum(pn, PB, 0, lane);
                g += part_sum(pn, PB, 1, lane);
            }
            if (lane == 0) {
                dgamma[c] = accumulate ? dgamma[c] + g : g;
                dbeta[c] = accumulate ? dbeta[c] + be : be;
            }
        }
    }
    if (batch && threadIdx.x < 64) {   // batch statistics: the two sums run over every image of the batch (image order)
        const int lane = threadIdx.x;
        float a = 0.f, b = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* pn = part + ((size_t)n * C + c) * PB * 2;
            a += part_sum(pn, PB, 0, lane);
            b += part_sum(pn, PB, 1, lane);
        }
        if (lane == 0) { st[0] = a; st[1] = b; }
    }
    __syncthreads();
    const float m = mean[img * C + c], r = rstd[img * C + c];
    const float cnt = batch ? (float)HW * (float)N : (float)HW;
    const float k1 = st[0] / cnt, k2 = st[1] / cnt;
    const float gr = gamma[c] * r;
    const float* pd = da + (size_t)img * da_nstride + (size_t)c * HW;
    const float* pa = aout + (size_t)img * a_nstride + (size_t)c * HW;
    const float* py = y + (size_t)img * y_nstride + (size_t)c * HW;
    float* po = dy + (size_t)img * dy_nstride + (size_t)c * HW;
    if (PB > MAX_PB) {
        const int seg = seg_len(HW, PB), lo = blockIdx.x * seg, hi = min(lo + seg, HW);
        const bool act = slope != 1.0f, from_y = act && beta != nullptr;
        float d[BN_V_CH][4], a[BN_V_CH][4], yy[BN_V_CH][4];
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k) {
            const int i = lo + 4 * (threadIdx.x + 256 * k);
            ld_run(pd, i, hi, d[k]);
            ld_run(py, i, hi, yy[k]);
            if (act && !from_y) ld_run(pa, i, hi, a[k]);
        }
        const float sc = gr, sh = from_y ? bn_shift(beta[(size_t)img * p_nstride + c], m, sc) : 0.f;
#pragma unroll
        for (int k = 0; k < BN_V_CH; ++k) {
            const int i = lo + 4 * (threadIdx.x + 256 * k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dz = d[k][j];
                const float av = from_y ? bn_affine(yy[k][j], sc, sh) : a[k][j];
                if (act && !(av > 0.f)) dz *= slope;
                d[k][j] = gr * (dz - k1 - (yy[k][j] - m) * r * k2);
            }
            if (i < hi) st_run(po, i, hi, d[k]);
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float dz = pd[i];
        if (slope != 1.0f && !(pa[i] > 0.f)) dz *= slope;
        po[i] = gr * (dz - k1 - (py[i] - m) * r * k2);
    }
}

// db[c] = sum over images and pixels of dy[c]
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dy, size_t nstride, int HW, int N, float* __restrict__ db,
                                                          int accumulate) {
    __shared__ float red[8];
    const int c = blockIdx.x;
    float s = 0.f, dummy = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* p = dy + (size_t)n * nstride + (size_t)c * HW;
        for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    block_sum2(s, dummy, red);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + s : s;
}

// Small planes (<= BN_SMALL_HW pixels: every layer from the 56x56 scale down at 224^2): ONE workgroup owns a whole
// (image, channel) plane, so statistics + apply are a single launch (the plane is re-read from L1/L2).  These layers are
// pure launch latency -- a kernel boundary costs more than the work.
// pins a value in a register as the rounded fp32 number it is: the compiler cannot fuse the multiply that produced it into an
// add that consumes it (-ffp-contract=fast works across statements), which keeps a fused kernel bit-identical to the two kernels
// it replaces, where the value went through memory
__device__ __forceinline__ void rounded(float& x) { asm volatile("" : "+v"(x)); }
// The BatchNorm backward's per-element arithmetic, written once with every rounding pinned (products that feed a sum are rounded
// or fused EXPLICITLY), so that the stand-alone kernels and the chained forms (BnPre) cannot be contracted differently:
//   dz = da * (a > 0 ? 1 : slope);  xhat = (y - mean) rstd;  s1 += dz;  s2 = fma(dz, xhat, s2);
//   dy = gamma rstd ((dz - s1/n) - xhat (s2/n))
__device__ __forceinline__ float bn_dz(float d, float a, float slope, bool act) {
    if (act && !(a > 0.f)) { d *= slope; rounded(d); }
    return d;
}
__device__ __forceinline__ float bn_xhat(float y, float m, float r) {
    float x = (y - m) * r;
    rounded(x);
    return x;
}
__device__ __forceinline__ float bn_dy(float gr, float dz, float k1, float xh, float k2) {
    float t = xh * k2;
    rounded(t);
    float g = gr * ((dz - k1) - t);
    rounded(g);
    return g;
}
constexpr int BN_SMALL_HW = 4096;
constexpr int BN_MAX_BATCH = 8;               // images per batch-statistics call (n_crops)
constexpr int BN_SMALL_PER = BN_SMALL_HW / 256;   // most plane elements a thread keeps in registers (kernels are instantiated for 1, 4 and 16:
                                                  // a 7x7 plane running the 16-element code fetched ten times the instructions it executed)
constexpr int BN_UP_SRC = 34 * 34;            // low-resolution plane of a fused upsampling staged in LDS (else read from global)
// slabs != null: the plane is first formed as bias + sum of the feeding convolution's split-K slabs (slice order) and stored to y
template <int PER>   // plane elements a thread keeps in registers: 1, 4 or 16 (planes of <= 256, <= 1024, <= 4096 pixels)
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* y, size_t y_nstride, float* __restrict__ out,
                                                           size_t out_nstride, int C, int HW, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, float slope, const float* __restrict__ slabs,
                                                           int ksplit, const float* __restrict__ bias, float* __restrict__ y_out, BnUpsample up,
                                                           size_t p_nstride, int batch, BnPre pre) {
    __shared__ float red[8];
    __shared__ float up_src_s[BN_UP_SRC];
    const int c = blockIdx.x, img = blockIdx.y;
    gamma += (size_t)img * p_nstride; beta += (size_t)img * p_nstride;
    if (bias) bias += (size_t)img * p_nstride;
    const float* p = y + (size_t)img * y_nstride + (size_t)c * HW;
    float* q = out + (size_t)img * out_nstride + (size_t)c * HW;
    // loads are issued branch-free (wave-uniform guards only; a thread past the end of the plane re-reads element 0 and
    // is masked afterwards): lane-guarded loads compile to load + s_waitcnt per element, i.e. one memory round trip each
    float v[PER];
    float s = 0.f, dummy = 0.f;
    if (pre.y && c < pre.C) {
        // ---- the skip branch's own BatchNorm + LeakyReLU on this plane (BnPre), in front of the concat's: the plane is formed
        // (from the skip convolution's output, or from its split-K slabs + bias), normalised with ITS statistics, activated and
        // stored into the concat buffer (the backward reads it there); v then holds the concat BatchNorm's input
        const float* py1 = pre.y + (size_t)img * pre.y_ns + (size_t)c * HW;
        if (pre.slabs) {
            const size_t per = (size_t)gridDim.y * pre.C * HW;
            const float* sp = pre.slabs + ((size_t)img * pre.C + c) * HW;
            float* yo = const_cast<float*>(py1);
            const float b = pre.bias ? pre.bias[(size_t)img * p_nstride + c] : 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) v[k] = b;
            for (int ks = 0; ks < pre.ksplit; ks += 4) {   // slice order, four slices of loads in flight (as the slabs path below)
                float t[4][PER];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ks + u >= pre.ksplit) continue;
                    const float* sk = sp + (size_t)(ks + u) * per;
#pragma unroll
                    for (int k = 0; k < PER; ++k) {
                        if (k * 256 >= HW) continue;
                        const int i = threadIdx.x + k * 256;
                        t[u][k] = sk[i < HW ? i : 0];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ks + u >= pre.ksplit) continue;
#pragma unroll
                    for (int k = 0; k < PER; ++k)
                        if (k * 256 < HW) v[k] += t[u][k];
                }
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + k * 256;
                if (i < HW) yo[i] = v[k]; else v[k] = 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + k * 256;
                v[k] = k * 256 < HW ? py1[i < HW ? i : 0] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (threadIdx.x + k * 256 >= HW) v[k] = 0.f;
        }
        float s1 = 0.f, d1 = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) s1 += v[k];
        block_sum2(s1, d1, red);
        const float m1 = s1 / (float)HW;
        float q1 = 0.f;
        d1 = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float d = threadIdx.x + k * 256 < HW ? v[k] - m1 : 0.f;
            q1 += d * d;
        }
        block_sum2(q1, d1, red);
        const float r1 = rsqrtf(q1 / (float)HW + eps);
        if (threadIdx.x == 0) { pre.mean[img * pre.C + c] = m1; pre.rstd[img * pre.C + c] = r1; }
        const float sc1 = pre.gamma[(size_t)img * p_nstride + c] * r1;
        const float sh1 = pre.beta[(size_t)img * p_nstride + c] - m1 * sc1;
        float* yo2 = const_cast<float*>(p);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            const float t = v[k] * sc1 + sh1;
            v[k] = i < HW ? (t > 0.f ? t : t * pre.slope) : 0.f;
            rounded(v[k]);   // the value the stand-alone kernels hand over through memory: nothing may be contracted across it
            if (i < HW) yo2[i] = v[k];
            s += v[k];
        }
    } else if (slabs) {
        const size_t per = (size_t)gridDim.y * C * HW;
        const float* sp = slabs + ((size_t)img * C + c) * HW;
        float* yo = y_out + (size_t)img * y_nstride + (size_t)c * HW;
        const float b = bias ? bias[c] : 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = b;
        for (int ks = 0; ks < ksplit; ks += 4) {   // slice order (as conv_splitk_reduce_kernel); four slices of loads in flight
            float t[4][PER];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ks + u >= ksplit) continue;
                const float* sk = sp + (size_t)(ks + u) * per;
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    if (k * 256 >= HW) continue;
                    const int i = threadIdx.x + k * 256;
                    t[u][k] = sk[i < HW ? i : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ks + u >= ksplit) continue;
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if (k * 256 < HW) v[k] += t[u][k];
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < HW) yo[i] = v[k]; else v[k] = 0.f;
            s += v[k];
        }
    } else if (up.src && c >= up.c0) {
        // upsampled channel of the concat: the values are produced here from the low-resolution plane (staged through LDS
        // when it fits) and stored into y for the backward -- the upsampling is not a launch of its own
        const float* sp = up.src + (size_t)img * up.src_ns + (size_t)(c - up.c0) * up.h * up.w;
        float* yo = const_cast<float*>(p);
        const bool in_lds = up.h * up.w <= BN_UP_SRC;
        if (in_lds) {
            for (int e = threadIdx.x; e < up.h * up.w; e += 256) up_src_s[e] = sp[e];
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            v[k] = 0.f;
            if (k * 256 < HW) {
                const int ii = i < HW ? i : HW - 1;
                const int oy = ii / up.Wo, ox = ii - oy * up.Wo;
                v[k] = in_lds ? up_value((const float*)up_src_s, up.h, up.w, oy, ox) : up_value(sp, up.h, up.w, oy, ox);
                if (i < HW) yo[i] = v[k];
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (threadIdx.x + k * 256 >= HW) v[k] = 0.f;
            s += v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            v[k] = k * 256 < HW ? p[i < HW ? i : 0] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (threadIdx.x + k * 256 >= HW) v[k] = 0.f;
            s += v[k];
        }
    }
    block_sum2(s, dummy, red);
    float m = s / (float)HW;
    float sq = 0.f;
    dummy = 0.f;
    float cnt = (float)HW;
    if (batch) {
        // batch statistics: every workgroup of channel c walks all N planes in image order (same bits in each); the other
        // planes are re-read (small, L2-resident); slabs / fused upsampling are not combined with this mode
        const int N = gridDim.y;
        float tot = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* pn = y + (size_t)n * y_nstride + (size_t)c * HW;
            float sn = 0.f, d2 = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + k * 256;
                if (k * 256 < HW) sn += i < HW ? pn[i] : 0.f;
            }
            block_sum2(sn, d2, red);
            tot += sn;
        }
        cnt = (float)HW * (float)N;
        m = tot / cnt;
        for (int n = 0; n < N; ++n) {
            const float* pn = y + (size_t)n * y_nstride + (size_t)c * HW;
            float qn = 0.f, d2 = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + k * 256;
                if (k * 256 < HW) { const float d = i < HW ? pn[i] - m : 0.f; qn += d * d; }
            }
            block_sum2(qn, d2, red);
            sq += qn;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float d = threadIdx.x + k * 256 < HW ? v[k] - m : 0.f;
            sq += d * d;
        }
        block_sum2(sq, dummy, red);
    }
    const float r = rsqrtf(sq / cnt + eps);
    if (threadIdx.x == 0) { mean_o[img * C + c] = m; rstd_o[img * C + c] = r; }
    const float sc = gamma[c] * r;
    const float sh = beta[c] - m * sc;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < HW) {
            const float t = v[k] * sc + sh;
            q[i] = t > 0.f ? t : t * slope;
        }
    }
}
// one workgroup per (channel, image); each plane is read once and kept in registers between the reduction and the apply
// pass.  dgamma / dbeta need the sums of EVERY image: the workgroup of image 0 recomputes the other images' two sums
// (reads only) and adds them in image order -- deterministic, no second launch, no cross-workgroup wait.
// sl.slabs != null: the output gradient of this plane was left as split-K slabs by the data-gradient convolution that produces it
// (BnSlabs): d = (accumulate ? the value at pd : 0) + (0 + slab 0 + slab 1 + ...), the arithmetic of conv_splitk_reduce_kernel
template <int PER>
__device__ __forceinline__ void bn_small_bwd_sums(const float* pd, const float* pa, const float* py, int HW, float m, float r, float slope,
                                                  float (&dz)[PER], float (&xh)[PER], float& s1, float& s2, float* red,
                                                  const float* sp = nullptr, size_t per = 0, int ksplit = 0, int sl_acc = 0) {
    s1 = 0.f; s2 = 0.f;
    // branch-free loads first (see bn_small_fwd_kernel), arithmetic after
    const bool act = slope != 1.0f;
    float vd[PER], va[PER], vy[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (k * 256 >= HW) continue;
        const int i = threadIdx.x + k * 256;
        const int j = i < HW ? i : 0;
        vd[k] = (!sp || sl_acc) ? pd[j] : 0.f;
        vy[k] = py[j];
        va[k] = act ? pa[j] : 1.f;
    }
    if (sp) {
        float v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = 0.f;
        for (int ks = 0; ks < ksplit; ks += 4) {   // slice order, four slices of loads in flight
            float t[4][PER];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ks + u >= ksplit) continue;
                const float* sk = sp + (size_t)(ks + u) * per;
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    if (k * 256 >= HW) continue;
                    const int i = threadIdx.x + k * 256;
                    t[u][k] = sk[i < HW ? i : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ks + u >= ksplit) continue;
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if (k * 256 < HW) v[k] += t[u][k];
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (k * 256 < HW) vd[k] = sl_acc ? vd[k] + v[k] : v[k];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * 256;
        float d = 0.f, x = 0.f;
        if (k * 256 < HW) {
            d = bn_dz(vd[k], va[k], slope, act);
            x = bn_xhat(vy[k], m, r);
            if (i >= HW) { d = 0.f; x = 0.f; }
        }
        dz[k] = d; xh[k] = x;
        s1 += d;
        s2 = __builtin_fmaf(d, x, s2);
    }
    block_sum2(s1, s2, red);
}
template <int PER>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ da, size_t da_nstride, const float* __restrict__ aout,
                                                           size_t a_nstride, const float* __restrict__ y, size_t y_nstride,
                                                           float* __restrict__ dy, size_t dy_nstride, int C, int HW, int N,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float slope, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate, BnUpsample up, size_t p_nstride, int batch, BnPre pre,
                                                           BnSlabs sl) {
    __shared__ float red[8];
    __shared__ float up_grad_s[BN_SMALL_HW];   // fused upsampling adjoint: this plane's input gradient
    const int c = blockIdx.x, img = blockIdx.y;
    gamma += (size_t)img * p_nstride;
    float dz[PER], xh[PER];
    float s1, s2;
    const bool chained = pre.y && c < pre.C;   // workgroup-uniform: a skip-branch BatchNorm sits in front of this channel (BnPre)
    {
        const float m = mean[img * C + c], r = rstd[img * C + c];
        float b1 = 0.f, b2 = 0.f;   // batch statistics: sums over every image of the batch, in image order; the own plane last (dz / xh stay)
        if (batch) {
            float t1s[BN_MAX_BATCH], t2s[BN_MAX_BATCH];
#pragma unroll
            for (int n = 0; n < BN_MAX_BATCH; ++n) {
                t1s[n] = 0.f; t2s[n] = 0.f;
                if (n < N && n != img)
                    bn_small_bwd_sums<PER>(da + (size_t)n * da_nstride + (size_t)c * HW, aout + (size_t)n * a_nstride + (size_t)c * HW,
                                      y + (size_t)n * y_nstride + (size_t)c * HW, HW, mean[n * C + c], rstd[n * C + c], slope, dz, xh, t1s[n], t2s[n], red);
            }
            bn_small_bwd_sums<PER>(da + (size_t)img * da_nstride + (size_t)c * HW, aout + (size_t)img * a_nstride + (size_t)c * HW,
                              y + (size_t)img * y_nstride + (size_t)c * HW, HW, m, r, slope, dz, xh, s1, s2, red);
#pragma unroll
            for (int n = 0; n < BN_MAX_BATCH; ++n)
                if (n < N) { b1 += n == img ? s1 : t1s[n]; b2 += n == img ? s2 : t2s[n]; }
        } else {
            bn_small_bwd_sums<PER>(da + (size_t)img * da_nstride + (size_t)c * HW, aout + (size_t)img * a_nstride + (size_t)c * HW,
                              y + (size_t)img * y_nstride + (size_t)c * HW, HW, m, r, slope, dz, xh, s1, s2, red,
                              sl.slabs ? sl.slabs + ((size_t)img * C + c) * HW : nullptr, (size_t)gridDim.y * C * HW, sl.ksplit, sl.accumulate);
        }
        const float cnt = batch ? (float)HW * (float)N : (float)HW;
        const float k1 = (batch ? b1 : s1) / cnt, k2 = (batch ? b2 : s2) / cnt;
        if (batch) { s1 = b1; s2 = b2; }   // the parameter gradients are exactly these sums
        const float gr = gamma[c] * r;
        float* po = dy + (size_t)img * dy_nstride + (size_t)c * HW;
        const bool through_adjoint = up.d_src && c >= up.c0;   // workgroup-uniform
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < HW) {
                const float gv = bn_dy(gr, dz[k], k1, xh[k], k2);
                if (through_adjoint) up_grad_s[i] = gv; else if (chained) dz[k] = gv; else po[i] = gv;
            } else if (chained) dz[k] = 0.f;
        }
        if (chained) {
            // ---- the adjoint of the skip branch's BatchNorm + LeakyReLU behind the concat's, on the same plane: dz holds the
            // gradient w.r.t. the activated skip plane a (= this BatchNorm's input y, in the concat buffer)
            const float m1 = pre.mean[img * pre.C + c], r1 = pre.rstd[img * pre.C + c];
            const float* pa1 = y + (size_t)img * y_nstride + (size_t)c * HW;                   // a = act(bn1(y1)): sign selects the LeakyReLU branch
            const float* py1 = pre.y + (size_t)img * pre.y_ns + (size_t)c * HW;
            float va[PER], vy[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if (k * 256 >= HW) continue;
                const int i = threadIdx.x + k * 256, j = i < HW ? i : 0;
                va[k] = pa1[j];
                vy[k] = py1[j];
            }
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + k * 256;
                float d = 0.f, x = 0.f;
                if (k * 256 < HW && i < HW) {
                    d = bn_dz(dz[k], va[k], pre.slope, true);
                    x = bn_xhat(vy[k], m1, r1);
                }
                dz[k] = d; xh[k] = x;
                t1 += d;
                t2 = __builtin_fmaf(d, x, t2);
            }
            block_sum2(t1, t2, red);
            const float j1 = t1 / (float)HW, j2 = t2 / (float)HW;
            const float gr1 = pre.gamma[(size_t)img * p_nstride + c] * r1;
            float* pd1 = pre.dy + (size_t)img * pre.y_ns + (size_t)c * HW;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + k * 256;
                if (i < HW) pd1[i] = bn_dy(gr1, dz[k], j1, xh[k], j2);
            }
            if (threadIdx.x == 0) {
                float* dg = pre.dgamma + (size_t)img * p_nstride + c;
                float* db = pre.dbeta + (size_t)img * p_nstride + c;
                *dg = accumulate ? *dg + t2 : t2;
                *db = accumulate ? *db + t1 : t1;
            }
        }
        if (through_adjoint) {
            // this channel is an upsampled one: nobody but the upsampling's adjoint reads its input gradient, so it goes
            // from LDS straight into the gradient of the low-resolution plane (the arithmetic of upsample2x_bwd_kernel)
            __syncthreads();
            float* qd = up.d_src + (size_t)img * up.d_src_ns + (size_t)(c - up.c0) * up.h * up.w;
            for (int e = threadIdx.x; e < up.h * up.w; e += 256)
                qd[e] = up_adjoint_value((const float*)up_grad_s, up.h, up.w, up.Ho, up.Wo, e / up.w, e % up.w);
        }
    }
    if (p_nstride) {   // independent images: every image owns its parameter gradients
        if (threadIdx.x == 0) {
            float* dg = dgamma + (size_t)img * p_nstride + c;
            float* db = dbeta + (size_t)img * p_nstride + c;
            *dg = accumulate ? *dg + s2 : s2;
            *db = accumulate ? *db + s1 : s1;
        }
        return;
    }
    if (img != 0) return;
    float g = s2, be = s1;
    for (int n = 1; n < (batch ? 0 : N); ++n) {
        float t1, t2;
        bn_small_bwd_sums<PER>(da + (size_t)n * da_nstride + (size_t)c * HW, aout + (size_t)n * a_nstride + (size_t)c * HW,
                          y + (size_t)n * y_nstride + (size_t)c * HW, HW, mean[n * C + c], rstd[n * C + c], slope, dz, xh, t1, t2, red);
        be += t1;
        g += t2;
    }
    if (threadIdx.x == 0) {
        dgamma[c] = accumulate ? dgamma[c] + g : g;
        dbeta[c] = accumulate ? dbeta[c] + be : be;
    }
}

// ---- middle planes (BN_SMALL_HW < pixels <= BN_MID_HW: the 112 x 112 planes of a 224 x 224 image) --------------------------------
// The two-stage form costs two launches of ~5 us each per BatchNorm in every direction.  One workgroup of 1024 threads owns a
// whole (image, channel) plane, as for the small planes, with the plane staged in LDS instead of registers (a register tile of 49
// elements per thread would be 18 k instructions of straight-line code): statistics + apply in ONE launch (round 4: 6 BatchNorms
// per image and direction at 224 x 224).  Same arithmetic as the small-plane kernels up to the order of the block-wide sums.
constexpr int BN_MID_HW = 16384;        // floats of LDS plane (64 KB)
constexpr int BN_MID_THREADS = 1024;
__device__ __forceinline__ void block_sum2_1024(float& a, float& b, float* red /* 32 floats */) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[16 + w] = b; }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { sa += red[k]; sb += red[16 + k]; }   // fixed order
    a = sa; b = sb;
}
constexpr int BN_MID_SRC = 4096;        // low-resolution source plane of a fused upsampling, staged behind the plane (else read from global)
// plane elements in groups of 4 (16-byte accesses, every thread's loads of a pass in flight together); tail elements one by one
template <class F4, class F1>
__device__ __forceinline__ void bn_mid_for(int HW, bool vec, F4&& f4, F1&& f1) {
    const int n4 = vec ? HW >> 2 : 0;
    for (int i = threadIdx.x; i < n4; i += BN_MID_THREADS) f4(i);
    for (int i = n4 * 4 + threadIdx.x; i < HW; i += BN_MID_THREADS) f1(i);
}
__global__ __launch_bounds__(BN_MID_THREADS) void bn_mid_fwd_kernel(const float* y, size_t y_nstride, float* __restrict__ out, size_t out_nstride, int C, int HW,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                    float* __restrict__ mean_o, float* __restrict__ rstd_o, float slope, BnUpsample up,
                                                                    size_t p_nstride, BnPre pre) {
    extern __shared__ __attribute__((aligned(16))) float bn_mid_plane[];   // HW floats (+ the upsampling source)
    __shared__ float red[32];
    const int c = blockIdx.x, img = blockIdx.y;
    gamma += (size_t)img * p_nstride; beta += (size_t)img * p_nstride;
    const float* p = y + (size_t)img * y_nstride + (size_t)c * HW;
    float* q = out + (size_t)img * out_nstride + (size_t)c * HW;
    const bool vec = !(HW & 3) && !((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(q)) & 15);
    float s = 0.f, dummy = 0.f;
    if (pre.y && c < pre.C) {
        // the skip branch's own BatchNorm + LeakyReLU in front of the concat's, on the same plane (BnPre; see bn_small_fwd_kernel).
        // Same access pattern and summation order as a launch of this kernel on the skip unit alone: the same bits either way.
        const float* py1 = pre.y + (size_t)img * pre.y_ns + (size_t)c * HW;
        const bool vec1 = vec && !(reinterpret_cast<size_t>(py1) & 15);
        float s1 = 0.f, d1 = 0.f;
        bn_mid_for(HW, vec1,
                   [&](int i) { const float4 v = reinterpret_cast<const float4*>(py1)[i]; reinterpret_cast<float4*>(bn_mid_plane)[i] = v; s1 += (v.x + v.y) + (v.z + v.w); },
                   [&](int i) { const float v = py1[i]; bn_mid_plane[i] = v; s1 += v; });
        block_sum2_1024(s1, d1, red);
        const float m1 = s1 / (float)HW;
        float q1 = 0.f;
        d1 = 0.f;
        bn_mid_for(HW, vec1,
                   [&](int i) { const float4 v = reinterpret_cast<const float4*>(bn_mid_plane)[i]; const float a = v.x - m1, b = v.y - m1, cc = v.z - m1, d = v.w - m1; q1 += (a * a + b * b) + (cc * cc + d * d); },
                   [&](int i) { const float d = bn_mid_plane[i] - m1; q1 += d * d; });
        block_sum2_1024(q1, d1, red);
        const float r1 = rsqrtf(q1 / (float)HW + eps);
        if (threadIdx.x == 0) { pre.mean[img * pre.C + c] = m1; pre.rstd[img * pre.C + c] = r1; }
        const float sc1 = pre.gamma[(size_t)img * p_nstride + c] * r1;
        const float sh1 = pre.beta[(size_t)img * p_nstride + c] - m1 * sc1;
        float* yo2 = const_cast<float*>(p);
        auto act1 = [&](float x) { const float t = x * sc1 + sh1; float a = t > 0.f ? t : t * pre.slope; rounded(a); return a; };
        bn_mid_for(HW, vec1,
                   [&](int i) {
                       const float4 v = reinterpret_cast<const float4*>(bn_mid_plane)[i];
                       const float4 a = float4{act1(v.x), act1(v.y), act1(v.z), act1(v.w)};
                       reinterpret_cast<float4*>(bn_mid_plane)[i] = a;
                       reinterpret_cast<float4*>(yo2)[i] = a;
                       s += (a.x + a.y) + (a.z + a.w);
                   },
                   [&](int i) { const float a = act1(bn_mid_plane[i]); bn_mid_plane[i] = a; yo2[i] = a; s += a; });
    } else if (up.src && c >= up.c0) {   // upsampled channel of the concat: produced here, stored into y for the backward
        const float* sp = up.src + (size_t)img * up.src_ns + (size_t)(c - up.c0) * up.h * up.w;
        float* yo = const_cast<float*>(p);
        const int hw = up.h * up.w;
        const bool in_lds = hw <= BN_MID_SRC;
        float* srcs = bn_mid_plane + ((HW + 3) & ~3);
        if (in_lds) {
            for (int e = threadIdx.x; e < hw; e += BN_MID_THREADS) srcs[e] = sp[e];
            __syncthreads();
        }
        for (int i = threadIdx.x; i < HW; i += BN_MID_THREADS) {
            const int oy = i / up.Wo, ox = i - oy * up.Wo;
            const float v = in_lds ? up_value((const float*)srcs, up.h, up.w, oy, ox) : up_value(sp, up.h, up.w, oy, ox);
            yo[i] = v;
            bn_mid_plane[i] = v;
            s += v;
        }
    } else {
        bn_mid_for(HW, vec,
                   [&](int i) { const float4 v = reinterpret_cast<const float4*>(p)[i]; reinterpret_cast<float4*>(bn_mid_plane)[i] = v; s += (v.x + v.y) + (v.z + v.w); },
                   [&](int i) { const float v = p[i]; bn_mid_plane[i] = v; s += v; });
    }
    block_sum2_1024(s, dummy, red);
    const float m = s / (float)HW;
    float sq = 0.f;
    dummy = 0.f;
    bn_mid_for(HW, vec,
               [&](int i) { const float4 v = reinterpret_cast<const float4*>(bn_mid_plane)[i]; const float a = v.x - m, b = v.y - m, cc = v.z - m, d = v.w - m; sq += (a * a + b * b) + (cc * cc + d * d); },
               [&](int i) { const float d = bn_mid_plane[i] - m; sq += d * d; });
    block_sum2_1024(sq, dummy, red);
    const float r = rsqrtf(sq / (float)HW + eps);
    if (threadIdx.x == 0) { mean_o[img * C + c] = m; rstd_o[img * C + c] = r; }
    const float sc = gamma[c] * r;
    const float sh = beta[c] - m * sc;
    auto act = [&](float x) { const float t = x * sc + sh; return t > 0.f ? t : t * slope; };
    bn_mid_for(HW, vec,
               [&](int i) { const float4 v = reinterpret_cast<const float4*>(bn_mid_plane)[i]; reinterpret_cast<float4*>(q)[i] = float4{act(v.x), act(v.y), act(v.z), act(v.w)}; },
               [&](int i) { q[i] = act(bn_mid_plane[i]); });
}
// backward: dz = da * act'(a) staged in LDS while s1 = sum dz and s2 = sum dz xhat are taken; second pass forms
// dy = gamma rstd (dz - s1/HW - xhat s2/HW) (xhat from a second read of y), or -- for an upsampled channel -- sends it through
// the x2 bilinear adjoint out of LDS.  Per-image parameter gradients (independent generators) or a single image.
__global__ __launch_bounds__(BN_MID_THREADS) void bn_mid_bwd_kernel(const float* __restrict__ da, size_t da_nstride, const float* __restrict__ aout, size_t a_nstride,
                                                                    const float* __restrict__ y, size_t y_nstride, float* __restrict__ dy, size_t dy_nstride, int C, int HW,
                                                                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    float slope, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, BnUpsample up,
                                                                    size_t p_nstride, BnPre pre) {
    extern __shared__ __attribute__((aligned(16))) float bn_mid_plane[];
    __shared__ float red[32];
    const int c = blockIdx.x, img = blockIdx.y;
    gamma += (size_t)img * p_nstride;
    const float m = mean[img * C + c], r = rstd[img * C + c];
    const float* pd = da + (size_t)img * da_nstride + (size_t)c * HW;
    const float* pa = aout + (size_t)img * a_nstride + (size_t)c * HW;
    const float* py = y + (size_t)img * y_nstride + (size_t)c * HW;
    float* po = dy + (size_t)img * dy_nstride + (size_t)c * HW;
    const bool act = slope != 1.0f;
    const bool vec = !(HW & 3) && !((reinterpret_cast<size_t>(pd) | reinterpret_cast<size_t>(pa) | reinterpret_cast<size_t>(py) | reinterpret_cast<size_t>(po)) & 15);
    float s1 = 0.f, s2 = 0.f;
    auto one = [&](float d, float a, float yv) {
        d = bn_dz(d, a, slope, act);
        s1 += d;
        s2 = __builtin_fmaf(d, bn_xhat(yv, m, r), s2);
        return d;
    };
    bn_mid_for(HW, vec,
               [&](int i) {
                   const float4 d = reinterpret_cast<const float4*>(pd)[i], yv = reinterpret_cast<const float4*>(py)[i];
                   const float4 a = act ? reinterpret_cast<const float4*>(pa)[i] : float4{1.f, 1.f, 1.f, 1.f};
                   reinterpret_cast<float4*>(bn_mid_plane)[i] = float4{one(d.x, a.x, yv.x), one(d.y, a.y, yv.y), one(d.z, a.z, yv.z), one(d.w, a.w, yv.w)};
               },
               [&](int i) { bn_mid_plane[i] = one(pd[i], act ? pa[i] : 1.f, py[i]); });
    block_sum2_1024(s1, s2, red);
    const float k1 = s1 / (float)HW, k2 = s2 / (float)HW;
    const float gr = gamma[c] * r;
    const bool through_adjoint = up.d_src && c >= up.c0;   // workgroup-uniform
    auto grad = [&](float dz, float yv) { return bn_dy(gr, dz, k1, bn_xhat(yv, m, r), k2); };
    if (pre.y && c < pre.C) {
        // the adjoint of the skip branch's BatchNorm + LeakyReLU behind the concat's (BnPre): the gradient w.r.t. the activated skip
        // plane a (= this BatchNorm's input y) stays in LDS.  Access pattern and summation order of a launch on the skip unit alone.
        const float m1 = pre.mean[img * pre.C + c], r1 = pre.rstd[img * pre.C + c];
        const float* py1 = pre.y + (size_t)img * pre.y_ns + (size_t)c * HW;
        float* pd1 = pre.dy + (size_t)img * pre.y_ns + (size_t)c * HW;
        const bool vec1 = vec && !((reinterpret_cast<size_t>(py1) | reinterpret_cast<size_t>(pd1)) & 15);
        float t1 = 0.f, t2 = 0.f;
        auto one1 = [&](float dzc, float a, float y1) {
            const float d = bn_dz(grad(dzc, a), a, pre.slope, true);
            t1 += d;
            t2 = __builtin_fmaf(d, bn_xhat(y1, m1, r1), t2);
            return d;
        };
        bn_mid_for(HW, vec1,
                   [&](int i) {
                       const float4 dzc = reinterpret_cast<const float4*>(bn_mid_plane)[i], a = reinterpret_cast<const float4*>(py)[i], y1 = reinterpret_cast<const float4*>(py1)[i];
                       reinterpret_cast<float4*>(bn_mid_plane)[i] = float4{one1(dzc.x, a.x, y1.x), one1(dzc.y, a.y, y1.y), one1(dzc.z, a.z, y1.z), one1(dzc.w, a.w, y1.w)};
                   },
                   [&](int i) { bn_mid_plane[i] = one1(bn_mid_plane[i], py[i], py1[i]); });
        block_sum2_1024(t1, t2, red);
        const float j1 = t1 / (float)HW, j2 = t2 / (float)HW;
        const float gr1 = pre.gamma[(size_t)img * p_nstride + c] * r1;
        auto grad1 = [&](float dz, float y1) { return bn_dy(gr1, dz, j1, bn_xhat(y1, m1, r1), j2); };
        bn_mid_for(HW, vec1,
                   [&](int i) {
                       const float4 dz = reinterpret_cast<const float4*>(bn_mid_plane)[i], y1 = reinterpret_cast<const float4*>(py1)[i];
                       reinterpret_cast<float4*>(pd1)[i] = float4{grad1(dz.x, y1.x), grad1(dz.y, y1.y), grad1(dz.z, y1.z), grad1(dz.w, y1.w)};
                   },
                   [&](int i) { pd1[i] = grad1(bn_mid_plane[i], py1[i]); });
        if (threadIdx.x == 0) {
            float* dg = pre.dgamma + (size_t)img * p_nstride + c;
            float* db = pre.dbeta + (size_t)img * p_nstride + c;
            *dg = accumulate ? *dg + t2 : t2;
            *db = accumulate ? *db + t1 : t1;
        }
    } else if (through_adjoint) {
        for (int i = threadIdx.x; i < HW; i += BN_MID_THREADS) bn_mid_plane[i] = grad(bn_mid_plane[i], py[i]);
        __syncthreads();
        float* qd = up.d_src + (size_t)img * up.d_src_ns + (size_t)(c - up.c0) * up.h * up.w;
        for (int e = threadIdx.x; e < up.h * up.w; e += BN_MID_THREADS)
            qd[e] = up_adjoint_value((const float*)bn_mid_plane, up.h, up.w, up.Ho, up.Wo, e / up.w, e % up.w);
    } else {
        bn_mid_for(HW, vec,
                   [&](int i) {
                       const float4 dz = reinterpret_cast<const float4*>(bn_mid_plane)[i], yv = reinterpret_cast<const float4*>(py)[i];
                       reinterpret_cast<float4*>(po)[i] = float4{grad(dz.x, yv.x), grad(dz.y, yv.y), grad(dz.z, yv.z), grad(dz.w, yv.w)};
                   },
                   [&](int i) { po[i] = grad(bn_mid_plane[i], py[i]); });
    }
    if (threadIdx.x == 0) {
        float* dg = dgamma + (size_t)img * p_nstride + c;
        float* db = dbeta + (size_t)img * p_nstride + c;
        *dg = accumulate ? *dg + s2 : s2;
        *db = accumulate ? *db + s1 : s1;
    }
}
static void bn_mid_allow_lds() {   // > 48 KB of dynamic LDS has to be allowed once per kernel and device
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<unsigned long long> done{0};
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute((const void*)bn_mid_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (BN_MID_HW + BN_MID_SRC) * 4);
        (void)hipFuncSetAttribute((const void*)bn_mid_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BN_MID_HW * 4);
        done.fetch_or(bit, std::memory_order_relaxed);
    }
}
// one image per parameter set (a single image, or independent generators) and per-image statistics: what the mid kernels cover
bool bn_pre_supported(int HW, int N, size_t p_nstride, int batch);
static inline bool bn_mid_ok(int HW, int N, size_t p_nstride, int batch) {
    constexpr int on = 1;
    return on && HW > BN_SMALL_HW && HW <= BN_MID_HW && !batch && (N == 1 || p_nstride);
}

// the instantiation whose register tile just covers the plane (same arithmetic in the same order: the surplus elements of a
// bigger tile only ever added zeros)
#define BN_SMALL_DISPATCH(HW_, KERNEL, GRID, STREAM, ...)                                                         \
    do {                                                                                                          \
        if ((HW_) <= 256) SPLICE_LAUNCH(KERNEL<1>, GRID, dim3(256), 0, STREAM, __VA_ARGS__);                  \
        else if ((HW_) <= 1024) SPLICE_LAUNCH(KERNEL<4>, GRID, dim3(256), 0, STREAM, __VA_ARGS__);            \
        else SPLICE_LAUNCH(KERNEL<16>, GRID, dim3(256), 0, STREAM, __VA_ARGS__);                              \
    } while (0)
// 512-pixel segments (tuned in-step with alternating runs: 1024 +0.45 %, 256 / 384 +0.1 %, 2048 +1.3 %)
static inline int plane_blocks(int HW) { int b = cdiv(HW, 512); return b < 1 ? 1 : (b > MAX_PB ? MAX_PB : b); }
int bn_part_floats(int N, int C) { return N * C * MAX_PB_V * 2; }
// big planes (round 5): segments of <= 4096 pixels moved in 16-byte runs; SPLICE_BN_VEC=0 keeps the 64-segment scalar layout everywhere
static inline bool bn_vec_ok(int HW) {
    constexpr int on = 1;
    return on && HW > MAX_PB * 1024 && (long long)HW <= (long long)MAX_PB_V * BN_V_CH * 1024;
}
static inline int bn_plane_blocks(int HW) {
    if (!bn_vec_ok(HW)) return plane_blocks(HW);
    const int b = cdiv(HW, 4096);
    return b <= MAX_PB ? MAX_PB + 1 : (b > MAX_PB_V ? MAX_PB_V : b);
}
// (chained or not, the skip branch's BatchNorm yields the same bits -- shared arithmetic helpers with pinned roundings,
// tests/test_generator_gpu.py::test_launch_count_forms_are_bit_neutral -- so this is a pure launch-count choice.  With the chained
// skip convolution's split-K slabs summed inside the concat kernel it wins at every batch size: -1.3 % step time at one pair per
// GPU, -1.0 % at four, -0.6 % at eight; profiles/r04_gen_ab.txt.  SPLICE_BN_CHAIN_MAXN limits it to fewer images per launch.)
bool bn_pre_supported(int HW, int N, size_t p_nstride, int batch) {
    static const int on = getenv("SPLICE_BN_CHAIN") ? atoi(getenv("SPLICE_BN_CHAIN")) : 1;
    constexpr int maxn = 1 << 30;
    return on && N < maxn && !batch && (N == 1 || p_nstride) && (HW <= BN_SMALL_HW || bn_mid_ok(HW, N, p_nstride, batch));
}
// the BatchNorm backward of a plane of this size can sum the split-K slabs of the data-gradient convolution that feeds it while it
// loads the plane (one-launch small-plane kernel, every image's workgroup reads only its own plane)
bool bn_bwd_takes_slabs(int HW, int N, size_t p_nstride, int batch) {
    static const int on = getenv("SPLICE_BN_BWD_SLABS") ? atoi(getenv("SPLICE_BN_BWD_SLABS")) : 1;
    return on && HW <= BN_SMALL_HW && !batch && (N == 1 || p_nstride);
}
int bn_fwd_launch(const float* y, size_t y_nstride, float* out, size_t out_nstride, int N, int C, int HW, const float* gamma,
                  const float* beta, float eps, float* part, float* mean, float* rstd, float slope, hipStream_t s, const BnUpsample* up, size_t p_nstride,
                  int batch, const BnPre* pre) {
    const BnUpsample u = up ? *up : BnUpsample{};
    const BnPre pr = pre ? *pre : BnPre{};
    if (batch && (N > BN_MAX_BATCH || u.src || p_nstride)) return SPLICE_ERR_ARG;
    if (pre && !bn_pre_supported(HW, N, p_nstride, batch)) return SPLICE_ERR_ARG;
    if (HW <= BN_SMALL_HW) {
        BN_SMALL_DISPATCH(HW, bn_small_fwd_kernel, dim3(C, N), s, y, y_nstride, out, out_nstride, C, HW, gamma, beta, eps, mean, rstd, slope,
                          (const float*)nullptr, 0, (const float*)nullptr, (float*)nullptr, u, p_nstride, batch, pr);
        return SPLICE_OK;
    }
    if (bn_mid_ok(HW, N, p_nstride, batch)) {
        bn_mid_allow_lds();
        SPLICE_LAUNCH(bn_mid_fwd_kernel, dim3(C, N), dim3(BN_MID_THREADS), (size_t)(((HW + 3) & ~3) + (u.src ? BN_MID_SRC : 0)) * 4, s, y, y_nstride, out, out_nstride, C, HW, gamma, beta, eps, mean, rstd, slope, u, p_nstride, pr);
        return SPLICE_OK;
    }
    const int PB = bn_plane_blocks(HW);
    if (PB > MAX_PB) SPLICE_LAUNCH(bn_stats_partial_v_kernel, dim3(PB, C, N), dim3(256), 0, s, y, y_nstride, C, HW, PB, part, u);
    else SPLICE_LAUNCH(bn_stats_partial_kernel, dim3(PB, C, N), dim3(256), 0, s, y, y_nstride, C, HW, PB, part, u);
    SPLICE_LAUNCH(bn_act_kernel, dim3(PB, C, N), dim3(256), 0, s, y, y_nstride, out, out_nstride, C, HW, PB, gamma, beta, part, eps, mean, rstd, slope, p_nstride, batch);
    return SPLICE_OK;
}
// (middle planes fuse it too when bn_bwd_launch takes the one-launch form: the caller passes what bn_bwd_launch will see)
bool bn_bwd_fuses_upsample(int HW, int h, int w) { return HW <= BN_SMALL_HW && h > 0 && w > 0; }
bool bn_bwd_fuses_upsample_ex(int HW, int h, int w, int N, size_t p_nstride, int batch) {
    return h > 0 && w > 0 && (HW <= BN_SMALL_HW || bn_mid_ok(HW, N, p_nstride, batch));
}
int bn_small_hw() { return BN_SMALL_HW; }
int bn_fwd_slabs_launch(const float* slabs, int ksplit, const float* bias, float* y, size_t y_nstride, float* out, size_t out_nstride, int N,
                        int C, int HW, const float* gamma, const float* beta, float eps, float* mean, float* rstd, float slope, hipStream_t s, size_t p_nstride) {
    if (HW > BN_SMALL_HW || ksplit < 2 || !slabs) return SPLICE_ERR_ARG;
    BN_SMALL_DISPATCH(HW, bn_small_fwd_kernel, dim3(C, N), s, (const float*)y, y_nstride, out, out_nstride, C, HW, gamma, beta, eps, mean, rstd,
                      slope, slabs, ksplit, bias, y, BnUpsample{}, p_nstride, 0, BnPre{});
    return SPLICE_OK;
}
int bn_bwd_launch(const float* da, size_t da_nstride, const float* aout, size_t a_nstride, const float* y, size_t y_nstride, float* dy,
                  size_t dy_nstride, int N, int C, int HW, const float* gamma, const float* mean, const float* rstd, float slope,
                  float* part, float* dgamma, float* dbeta, int accumulate, hipStream_t s, const BnUpsample* up, size_t p_nstride, int batch, const BnPre* pre,
                  const BnSlabs* slabs, const float* beta) {
    if (batch && (N > BN_MAX_BATCH || p_nstride)) return SPLICE_ERR_ARG;
    if (pre && !bn_pre_supported(HW, N, p_nstride, batch)) return SPLICE_ERR_ARG;
    if (slabs && slabs->slabs && !bn_bwd_takes_slabs(HW, N, p_nstride, batch)) return SPLICE_ERR_ARG;
    const BnPre pr = pre ? *pre : BnPre{};
    if (HW <= BN_SMALL_HW) {
        BnUpsample u = up ? *up : BnUpsample{};
        if (!bn_bwd_fuses_upsample(HW, u.h, u.w)) u.d_src = nullptr;
        const BnSlabs sl = slabs ? *slabs : BnSlabs{};
        BN_SMALL_DISPATCH(HW, bn_small_bwd_kernel, dim3(C, N), s, da, da_nstride, aout, a_nstride, y, y_nstride, dy, dy_nstride, C, HW, N,
                          gamma, mean, rstd, slope, dgamma, dbeta, accumulate, u, p_nstride, batch, pr, sl);
        return SPLICE_OK;
    }
    if (bn_mid_ok(HW, N, p_nstride, batch)) {
        BnUpsample u = up ? *up : BnUpsample{};
        if (!(u.h > 0 && u.w > 0)) u.d_src = nullptr;
        bn_mid_allow_lds();
        SPLICE_LAUNCH(bn_mid_bwd_kernel, dim3(C, N), dim3(BN_MID_THREADS), (size_t)HW * 4, s, da, da_nstride, aout, a_nstride, y, y_nstride, dy, dy_nstride, C, HW, gamma, mean,
                      rstd, slope, dgamma, dbeta, accumulate, u, p_nstride, pr);
        return SPLICE_OK;
    }
    const int PB = bn_plane_blocks(HW);
    // big planes: the activation's sign is re-formed from y (needs beta; batch statistics keep reading the activated tensor: their mean / rstd arrays
    // are per image while the forward normalised with the batch's)
    static const int sign_from_y = getenv("SPLICE_BN_SIGN_FROM_Y") ? atoi(getenv("SPLICE_BN_SIGN_FROM_Y")) : 1;
    const float* be = (sign_from_y && !batch && PB > MAX_PB) ? beta : nullptr;
    SPLICE_LAUNCH(bn_bwd_partial_kernel, dim3(PB, C, N), dim3(256), 0, s, da, da_nstride, aout, a_nstride, y, y_nstride, C, HW, PB, mean, rstd, slope, part, gamma, be, p_nstride);
    SPLICE_LAUNCH(bn_bwd_apply_kernel, dim3(PB, C, N), dim3(256), 0, s, da, da_nstride, aout, a_nstride, y, y_nstride, dy,
                       dy_nstride, C, HW, N, PB, gamma, mean, rstd, slope, part, dgamma, dbeta, accumulate, p_nstride, batch, be);
    return SPLICE_OK;
}
__global__ void fill_zero_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
int fill_zero_launch(float* p, int n, hipStream_t s) {
    SPLICE_LAUNCH(fill_zero_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, n);
    return SPLICE_OK;
}
int channel_sum_launch(const float* dy, size_t nstride, int N, int C, int HW, float* db, int accumulate, hipStream_t s) {
    SPLICE_LAUNCH(channel_sum_kernel, dim3(C), dim3(256), 0, s, dy, nstride, HW, N, db, accumulate);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
// bilinear x2 (align_corners=False) of in[C][h][w] -> the top-left Ho x Wo window of the 2h x 2w
// result (Concat's centre-crop offset is always 0 here: models/unet/common.py:24-37).
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ in, size_t in_nstride, float* __restrict__ out,
                                                             size_t out_nstride, int C, int h, int w, int Ho, int Wo) {
    const int c = blockIdx.y, img = blockIdx.z;
    const float* p = in + (size_t)img * in_nstride + (size_t)c * h * w;
    float* q = out + (size_t)img * out_nstride + (size_t)c * Ho * Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Ho * Wo; i += gridDim.x * 256) {
        q[i] = up_value(p, h, w, i / Wo, i % Wo);
    }
}
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dout, size_t dout_nstride, float* __restrict__ din,
                                                             size_t din_nstride, int C, int h, int w, int Ho, int Wo) {
    const int c = blockIdx.y, img = blockIdx.z;
    const float* p = dout + (size_t)img * dout_nstride + (size_t)c * Ho * Wo;
    float* q = din + (size_t)img * din_nstride + (size_t)c * h * w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < h * w; i += gridDim.x * 256) {
        // adjoint weights of the x2 bilinear (align_corners=False) in closed form: output o = 2m-1+t, t = 0..3, reads input m
        // with weight {1/4, 3/4, 3/4, 1/4}; at the borders the clamped source index folds the missing neighbour's share
        // in (o = 0 and o = 2n-1 read their input pixel with weight 1), and outputs outside the Ho x Wo window do not exist.
        // All 16 taps are loaded unconditionally from clamped coordinates (a zero weight marks the taps that do not exist).
        q[i] = up_adjoint_value(p, h, w, Ho, Wo, i / w, i % w);
    }
}
int upsample2x_fwd_launch(const float* in, size_t in_nstride, float* out, size_t out_nstride, int N, int C, int h, int w, int Ho, int Wo, hipStream_t s) {
    SPLICE_LAUNCH(upsample2x_fwd_kernel, dim3(plane_blocks(Ho * Wo), C, N), dim3(256), 0, s, in, in_nstride, out, out_nstride, C, h, w, Ho, Wo);
    return SPLICE_OK;
}
int upsample2x_bwd_launch(const float* dout, size_t dout_nstride, float* din, size_t din_nstride, int N, int C, int h, int w, int Ho, int Wo, hipStream_t s) {
    SPLICE_LAUNCH(upsample2x_bwd_kernel, dim3(cdiv(h * w, 256), C, N), dim3(256), 0, s, dout, dout_nstride, din, din_nstride, C, h, w, Ho, Wo);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
// dpre = dout * s * (1 - s) for the sigmoid head (s = generator output)
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ sout, float* __restrict__ dpre, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float sv = sout[i];
        dpre[i] = dout[i] * sv * (1.f - sv);
    }
}
// dpre = dout * s * (1 - s) and the per-segment sums of dpre for the head bias gradient: block (pb, c) handles segment pb of
// channel c over every image (gridDim.z > 1: independent images -- blockIdx.z = image, partials per image).  The partials are laid
// out [image][segment][channel] = the [chunk][element] layout of the weight-gradient partials, so the backward's ONE
// wgrad_reduce_all launch sums them with everything else (round 4: the bias had a reduce launch of its own).
__global__ __launch_bounds__(256) void sigmoid_bwd_bias_kernel(const float* __restrict__ dout, const float* __restrict__ sout,
                                                               float* __restrict__ dpre, int N, int C, int HW, int PB,
                                                               float* __restrict__ part) {
    __shared__ float red[8];
    const int pb = blockIdx.x, c = blockIdx.y;
    const int seg = seg_len(HW, PB), lo = pb * seg, hi = min(lo + seg, HW);
    float acc = 0.f, dummy = 0.f;
    const int n_lo = gridDim.z > 1 ? blockIdx.z : 0, n_hi = gridDim.z > 1 ? blockIdx.z + 1 : N;
    part += (size_t)(gridDim.z > 1 ? blockIdx.z : 0) * C * PB;
    for (int n = n_lo; n < n_hi; ++n) {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            const float sv = sout[base + i];
            const float d = dout[base + i] * sv * (1.f - sv);
            dpre[base + i] = d;
            acc += d;
        }
    }
    block_sum2(acc, dummy, red);
    if (threadIdx.x == 0) part[pb * C + c] = acc;
}
// returns the number of partial "chunks" per parameter (for the reduce entry): PB, or N * PB for independent images
int sigmoid_bwd_bias_launch(const float* dout, const float* sout, float* dpre, int N, int C, int HW, float* part, hipStream_t s, size_t p_nstride, int* chunks) {
    const int PB = plane_blocks(HW);
    const int nz = p_nstride ? N : 1;
    SPLICE_LAUNCH(sigmoid_bwd_bias_kernel, dim3(PB, C, nz), dim3(256), 0, s, dout, sout, dpre, N, C, HW, PB, part);
    if (chunks) *chunks = PB * nz;
    return SPLICE_OK;
}
int sigmoid_bias_part_floats(int N, int C) { return N * C * MAX_PB; }
int sigmoid_bwd_launch(const float* dout, const float* sout, float* dpre, size_t n, hipStream_t s) {
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    SPLICE_LAUNCH(sigmoid_bwd_kernel, dim3((unsigned)g), dim3(256), 0, s, dout, sout, dpre, n);
    return SPLICE_OK;
}

// one element of the update; contraction is OFF so that the vector body and the scalar tail of the kernel round alike (an
// element's result must not depend on where in an arena it sits: P pairs per step == P single runs, bit for bit)
__device__ __forceinline__ void adam_update(float& pi, float& gi, float& mi_, float& vi_, float g2i, bool has_g2, float b1, float b2, float eps, float lr,
                                            float bc1, float bc2_sqrt, int zero_grad) {
#pragma clang fp contract(off)
    if (has_g2) gi += g2i;   // second gradient arena (the B-crop plan): g = g + g2, as a separate add would leave it
    const float mi = b1 * mi_ + (1.f - b1) * gi;
    const float vi = b2 * vi_ + (1.f - b2) * gi * gi;
    mi_ = mi;
    vi_ = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    if (zero_grad) gi = 0.f;
}
// ---------------------------------------------------------------------------------------
// Fused multi-tensor Adam over the flat parameter arena (K19; torch.optim.Adam as configured by
// util/util.py:28-32: no weight decay / amsgrad, eps added after sqrt(v_hat)).  Also clears the
// gradient (optimizer.zero_grad, train.py:56) when zero_grad != 0.
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                            float b1, float b2, float eps, float bc1, float bc2_sqrt, int zero_grad, const int* __restrict__ step_ptr,
                            const float* __restrict__ g2) {
    if (step_ptr) {   // step count lives on the device (graph replay): bias corrections computed here
        const float t = (float)*step_ptr;
        bc1 = 1.0f - powf(b1, t);
        bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    }
    // one float4 per thread and (at the generator's size) ONE pass: the five operand vectors of an element group are a single
    // memory round trip; the kernel is the last node of the step's critical chain
    auto upd = [&](float& pi, float& gi, float& mi_, float& vi_, float g2i) {
        adam_update(pi, gi, mi_, vi_, g2i, g2 != nullptr, b1, b2, eps, lr, bc1, bc2_sqrt, zero_grad);
    };
    const size_t n4 = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v) |
                        reinterpret_cast<size_t>(g2)) & 15) ? 0 : n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<float4*>(g)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 g2v = g2 ? reinterpret_cast<const float4*>(g2)[i] : float4{0.f, 0.f, 0.f, 0.f};
        upd(pv.x, gv.x, mv.x, vv.x, g2v.x); upd(pv.y, gv.y, mv.y, vv.y, g2v.y); upd(pv.z, gv.z, mv.z, vv.z, g2v.z); upd(pv.w, gv.w, mv.w, vv.w, g2v.w);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
        if (g2 || zero_grad) reinterpret_cast<float4*>(g)[i] = gv;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
        upd(pi, gi, mi, vi, g2 ? g2[i] : 0.f);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (g2 || zero_grad) g[i] = gi;
    }
}
int adam_launch(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step, int zero_grad, hipStream_t s) {
    if (step < 1) return SPLICE_ERR_ARG;
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2 = 1.0f - powf(b2, (float)step);
    size_t g_ = (n / 4 + 255) / 256 + 1;
    if (g_ > 2048) g_ = 2048;
    SPLICE_LAUNCH(adam_kernel, dim3((unsigned)g_), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, bc1, sqrtf(bc2), zero_grad, (const int*)nullptr, (const float*)nullptr);
    return SPLICE_OK;
}
// same, the step count t (>= 1) read from device memory at execution time
int adam_launch_dev(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, const int* step_dev,
                    int zero_grad, hipStream_t s, const float* g2) {
    size_t g_ = (n / 4 + 255) / 256 + 1;
    if (g_ > 2048) g_ = 2048;
    SPLICE_LAUNCH(adam_kernel, dim3((unsigned)g_), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, 1.f, 1.f, zero_grad, step_dev, g2);
    return SPLICE_OK;
}
// ---------------------------------------------------------------------------------------
// nn.BatchNorm2d bookkeeping (models/unet/common.py:95-96: every netG call in train mode moves running_mean / running_var
// with momentum 0.1; nothing ever reads them, but they are part of netG.state_dict()).  One launch covers every
// BatchNorm of up to RUNSTAT_MAX_PLANS generator calls IN CALL ORDER: thread (bn, channel) walks the plans in order (the
// updates of one buffer do not commute exactly), and the images of a plan in order unless they are independent
// generators (blockIdx.y = image = its own buffer arena).  var_unbiased is rebuilt from the saved rstd.
__global__ __launch_bounds__(256) void bn_running_update_kernel(RunStatTable t, float* __restrict__ running, size_t r_nstride, float momentum, float eps) {
    const int bn = blockIdx.x, c = threadIdx.x;
    if (c >= t.C[bn]) return;
    for (int p = 0; p < t.n_plans; ++p) {
        const bool indep = t.indep[p] != 0;
        if (indep && (int)blockIdx.y >= t.N[p]) continue;
        if (!indep && blockIdx.y != 0) continue;
        const int n_lo = indep ? blockIdx.y : 0, n_hi = indep ? blockIdx.y + 1 : t.N[p];
        float* arena = running + (indep ? (size_t)blockIdx.y * r_nstride : 0) + t.r_off[bn];
        const float hw = (float)t.HW[p][bn];
        const float unbias = hw > 1.f ? hw / (hw - 1.f) : 1.f;
        for (int n = n_lo; n < n_hi; ++n) {
            const float m = t.mean[p][bn][n * t.C[bn] + c], r = t.rstd[p][bn][n * t.C[bn] + c];
            const float var = fmaxf(1.0f / (r * r) - eps, 0.f) * unbias;
            arena[c] = (1.f - momentum) * arena[c] + momentum * m;
            arena[t.C[bn] + c] = (1.f - momentum) * arena[t.C[bn] + c] + momentum * var;
        }
    }
}
int bn_running_update_launch(const RunStatTable& t, float* running, size_t r_nstride, float momentum, float eps, int max_images, hipStream_t s) {
    if (t.n_plans < 1 || t.n_plans > RUNSTAT_MAX_PLANS || t.n_bn < 1 || t.n_bn > RUNSTAT_MAX_BN) return SPLICE_ERR_ARG;
    // one thread per channel: the concat BatchNorm of an architecture within arch_check has up to 128 + 128 = 256 channels
    SPLICE_LAUNCH(bn_running_update_kernel, dim3(t.n_bn, max_images), dim3(256), 0, s, t, running, r_nstride, momentum, eps);
    return SPLICE_OK;
}

__global__ void set_int_kernel(int* p, int v) { *p = v; }
int set_int_launch(int* p, int v, hipStream_t s) {
    SPLICE_LAUNCH(set_int_kernel, dim3(1), dim3(1), 0, s, p, v);
    return SPLICE_OK;
}
