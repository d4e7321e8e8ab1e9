// bf16 "NT" GEMM tile engine for gfx950:  C[M][N] = sum_k A[m][k] * B[n][k]   (fp32 accumulate)
//
// Both operands are K-contiguous (activations [rows][K]; weights pre-packed [N][K]), so
// every MFMA fragment is one 16-byte LDS read.  Workgroup = 256 threads = 4 waves in a
// 2x2 grid; each wave owns a (BM/2)x(BN/2) sub-tile built from 16x16x32 bf16 MFMAs.
// K is consumed in BK=64 slices staged through LDS with an XOR swizzle of the 16-byte
// chunks (chunk ^= row&7) that makes the ds_read_b128 fragment reads conflict-free;
// slices arrive by LDS-DMA into a 2-stage ring (run_glds) or a 4-stage ring (run_ring).
//
// Ragged M / N are handled by clamping the load row and predicating the epilogue, so any
// M, N >= 1 works; K must be a multiple of 64 and lda/ldb multiples of 8 (16-byte rows).
#pragma once
#include <atomic>
#include "common.h"

constexpr int GEMM_BK = 64;

// SWAP = true computes the transposed fragments (MFMA operands exchanged): each lane then owns 4 consecutive
// COLUMNS of one output row, so the epilogue moves 8/16-byte vectors instead of scalar elements.
// FP8 = true: the operands are e4m3 bytes (OCP fp8, the gfx950 MFMA format) instead of bf16 pairs.  The byte geometry of a
// slice is unchanged (rows of 128 B, XOR-swizzled 16-B chunks, the same LDS-DMA), so the loaders run as they are with
// K, lda, ldb given in 2-byte units; only the contraction differs: a 16-byte fragment now holds 16 k-values = TWO
// k-values where it fed one bf16 MFMA -- half the LDS and HBM bytes per FLOP.  A and B use the same k permutation inside a
// slice, so the sum is unaffected.
// The fp8 contraction runs on the block-scaled K = 128 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, both operands e4m3, every
// block scale 2^0 = E8M0 127): twice the issue rate of the non-scaled 16x16x32 fp8 form, which runs at the bf16 rate.  A lane's
// 32 k-values of a slice are its two 16-byte LDS fragments (the kk = 0 / 1 chunks the bf16 path feeds to two MFMAs); A and B
// are assembled the same way, so whatever k order the instruction uses inside a lane's 32 bytes, the products pair up.  The
// per-token / per-channel scales stay in the epilogue (fp32): block scales of 1 make the instruction a plain fp8 dot product.
typedef int mx8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16_fp8_mx(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, f32x4 c) {
    const mx8_t av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const mx8_t bv = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// WGM = waves along M (the wave grid is WGM x 2): 2 = the 256-thread workgroup; 4 = 512 threads, e.g. a 256 x 256 tile of 64 x 128
// wave tiles -- half the L2 -> LDS bytes per FLOP of the 128 x 128 tile
template <int BM, int BN, bool SWAP = false, bool FP8 = false, int WGM = 2>
struct GemmTile {
    static constexpr int NW = 2 * WGM;      // waves per workgroup
    static constexpr int WROWS = BM / WGM;  // rows of a wave's sub-tile
    static constexpr int FM = WROWS / 16;   // 16-row fragments per wave along M
    static constexpr int FN = BN / 32;
    static constexpr int A_LOADS = BM * 8 / (64 * NW);  // 16-byte chunks per thread per slice
    static constexpr int B_LOADS = BN * 8 / (64 * NW);
    static constexpr int LDS_ELEMS = (BM + BN) * GEMM_BK;
    f32x4 acc[FM][FN];

    // the MFMAs of one BK slice staged at As / Bs
    __device__ __forceinline__ void slice_mfma(const bf16_t* As, const bf16_t* Bs, int wm, int wn, int frow, int fchunk) {
        if constexpr (FP8) {
            u32x4 af[FM][2], bf[FN][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int chunk = kk * 4 + fchunk;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = wm * WROWS + i * 16 + frow;
                    af[i][kk] = *reinterpret_cast<const u32x4*>(As + row * GEMM_BK + ((chunk ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = wn * (BN / 2) + j * 16 + frow;
                    bf[j][kk] = *reinterpret_cast<const u32x4*>(Bs + row * GEMM_BK + ((chunk ^ (row & 7)) << 3));
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = SWAP ? mfma16_fp8_mx(bf[j][0], bf[j][1], af[i][0], af[i][1], acc[i][j]) : mfma16_fp8_mx(af[i][0], af[i][1], bf[j][0], bf[j][1], acc[i][j]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 af[FM], bf[FN];
                const int chunk = kk * 4 + fchunk;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = wm * WROWS + i * 16 + frow;
                    af[i] = *reinterpret_cast<const u32x4*>(As + row * GEMM_BK + ((chunk ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = wn * (BN / 2) + j * 16 + frow;
                    bf[j] = *reinterpret_cast<const u32x4*>(Bs + row * GEMM_BK + ((chunk ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = SWAP ? mfma16(bf[j], af[i], acc[i][j]) : mfma16(af[i], bf[j], acc[i][j]);
            }
        }
    }

    // Direct-to-LDS staging (global_load_lds, 16 B per lane): no VGPR round trip, no ds_write pass.  A wave-instruction fills 8 rows x 128 B of LDS linearly (dest = wave base + lane*16), so the
    // XOR swizzle is applied on the SOURCE chunk each lane fetches.  Two LDS stages; the loads of slice t+1 are
    // issued right after the barrier that publishes slice t and fly during its MFMAs (one barrier per slice).
    __device__ __forceinline__ void run_glds(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                             int M, int N, int K, int m0, int n0, bf16_t* smem, int kbeg = 0) {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef __attribute__((address_space(1))) const void glb_void;
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations (M0) stay on the SALU
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
        // per-lane BYTE offsets (32-bit, fixed for the whole K loop) from a uniform base that the SALU advances:
        // global_load_lds in its scalar-base + vector-offset form, no per-slice address VALU
        uint32_t ao[A_LOADS], bo[B_LOADS];
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int gr = m0 + wave * 8 + 8 * NW * i + lrow;
            gr = gr < M ? gr : M - 1;
            ao[i] = ((uint32_t)gr * (uint32_t)lda + gchunk * 8) * 2u;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            int gr = n0 + wave * 8 + 8 * NW * i + lrow;
            gr = gr < N ? gr : N - 1;
            bo[i] = ((uint32_t)gr * (uint32_t)ldb + gchunk * 8) * 2u;
        }
        auto issue = [&](int k0, int stage) {
            bf16_t* st = smem + stage * LDS_ELEMS;
            const char* ab = reinterpret_cast<const char*>(A + kbeg + k0);
            const char* bb = reinterpret_cast<const char*>(B + kbeg + k0);
            asm volatile("" : "+s"(ab), "+s"(bb));   // keep the bases scalar (the loop optimiser would fold them into 8 vector pointers)
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i)
                __builtin_amdgcn_global_load_lds((glb_void*)(ab + ao[i]), (lds_void*)(st + (wave * 8 + 8 * NW * i) * GEMM_BK), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i)
                __builtin_amdgcn_global_load_lds((glb_void*)(bb + bo[i]), (lds_void*)(st + BM * GEMM_BK + (wave * 8 + 8 * NW * i) * GEMM_BK), 16, 0, 0);
        };
        issue(0, 0);
        const int frow = lane & 15, fchunk = lane >> 4;
        int it = 0;
        for (int k0 = 0; k0 < K; k0 += GEMM_BK, ++it) {
            __syncthreads();   // slice `it` has landed (the barrier drains the LDS-DMA queue); slice it-1 fully consumed
            if (k0 + GEMM_BK < K) issue(k0 + GEMM_BK, (it + 1) & 1);
            const bf16_t* As = smem + (it & 1) * LDS_ELEMS;
            const bf16_t* Bs = As + BM * GEMM_BK;
            slice_mfma(As, Bs, wm, wn, frow, fchunk);
        }
    }

    // Deep LDS-DMA ring (NS stages, NS-1 slices in flight) for the latency-bound shapes: with one workgroup per CU
    // and a short slice (a 64x64 tile issues 8 MFMAs per wave per slice) a 2-stage ring exposes the full
    // global->LDS latency (~0.7-1.5 us under load) on EVERY slice.  The DMA is issued from inline asm in its
    // scalar-base + 32-bit-lane-offset form (no address VALU, and hipcc's scoreboard -- which would drain the
    // queue with vmcnt(0) at every barrier / first LDS read -- does not see it); the waits are counted by hand:
    // before slice t is read, only the NS-2 younger slices may still be in flight.
    template <int NL>
    static __device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory"); }
    static __device__ __forceinline__ void dma16(uint32_t lds_byte, uint32_t voff, const void* sbase) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(voff), "s"(sbase) : "memory");
    }
    template <int NS>
    __device__ __forceinline__ void run_ring(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                             int M, int N, int K, int m0, int n0, bf16_t* smem, int kbeg = 0) {
        static_assert(NS >= 3, "use run_glds for the 2-stage form");
        constexpr int NL = A_LOADS + B_LOADS;
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
        uint32_t ao[A_LOADS], bo[B_LOADS];
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int gr = m0 + wave * 8 + 8 * NW * i + lrow;
            gr = gr < M ? gr : M - 1;
            ao[i] = ((uint32_t)gr * (uint32_t)lda + gchunk * 8) * 2u;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            int gr = n0 + wave * 8 + 8 * NW * i + lrow;
            gr = gr < N ? gr : N - 1;
            bo[i] = ((uint32_t)gr * (uint32_t)ldb + gchunk * 8) * 2u;
        }
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) bf16_t*)smem + (uint32_t)wave * (8 * GEMM_BK * 2);
        auto issue = [&](int sl, int stage) {
            const uint32_t st = lds0 + (uint32_t)stage * (LDS_ELEMS * 2);
            const bf16_t* ab = A + kbeg + sl * GEMM_BK;
            const bf16_t* bb = B + kbeg + sl * GEMM_BK;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) dma16(st + i * (8 * NW * GEMM_BK * 2), ao[i], ab);
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) dma16(st + (BM + 8 * NW * i) * (GEMM_BK * 2), bo[i], bb);
        };
        const int nsl = K / GEMM_BK;
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nsl) issue(p, p);
        const int frow = lane & 15, fchunk = lane >> 4;
        int stage = 0, nstage = NS - 1;   // stage of slice t, stage of slice t + NS - 1
        for (int t = 0; t < nsl; ++t) {
            if (t + NS - 2 < nsl) wait_dma<(NS - 2) * NL>(); else wait_dma<0>();
            __builtin_amdgcn_s_barrier();   // slice t landed for every wave; the stage of slice t-1 is free again
            if (t + NS - 1 < nsl) issue(t + NS - 1, nstage);
            const bf16_t* As = smem + stage * LDS_ELEMS;
            const bf16_t* Bs = As + BM * GEMM_BK;
            slice_mfma(As, Bs, wm, wn, frow, fchunk);
            stage = stage + 1 == NS ? 0 : stage + 1;
            nstage = nstage + 1 == NS ? 0 : nstage + 1;
        }
    }

    // run_ring + the squared norms of the operand rows, taken from the bf16 fragments on their way to the MFMAs (no extra memory
    // traffic): ssa[i] / ssb[j] = sum over k of the squares of A row (wm * WROWS + i * 16 + (lane & 15)) / B row
    // (wn * BN / 2 + j * 16 + (lane & 15)), complete in every lane.  A row's sum is formed by the same instruction sequence
    // whether the row is an A row or a B row (key self-similarity: K K^T with both operands the same matrix).
    static __device__ __forceinline__ float sumsq8(const u32x4& v, float acc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float lo = __uint_as_float(v[q] << 16), hi = __uint_as_float(v[q] & 0xFFFF0000u);
            acc = __builtin_fmaf(lo, lo, acc);
            acc = __builtin_fmaf(hi, hi, acc);
        }
        return acc;
    }
    template <int NS>
    __device__ __forceinline__ void run_ring_norms(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K,
                                                   int m0, int n0, bf16_t* smem, float (&ssa)[FM], float (&ssb)[FN]) {
        static_assert(NS >= 3 && !FP8 && !SWAP, "bf16, unswapped, ring form");
        constexpr int NL = A_LOADS + B_LOADS;
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            ssa[i] = 0.f;
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) ssb[j] = 0.f;
        const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
        uint32_t ao[A_LOADS], bo[B_LOADS];
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int gr = m0 + wave * 8 + 8 * NW * i + lrow;
            gr = gr < M ? gr : M - 1;
            ao[i] = ((uint32_t)gr * (uint32_t)lda + gchunk * 8) * 2u;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            int gr = n0 + wave * 8 + 8 * NW * i + lrow;
            gr = gr < N ? gr : N - 1;
            bo[i] = ((uint32_t)gr * (uint32_t)ldb + gchunk * 8) * 2u;
        }
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) bf16_t*)smem + (uint32_t)wave * (8 * GEMM_BK * 2);
        auto issue = [&](int sl, int stage) {
            const uint32_t st = lds0 + (uint32_t)stage * (LDS_ELEMS * 2);
            const bf16_t* ab = A + sl * GEMM_BK;
            const bf16_t* bb = B + sl * GEMM_BK;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) dma16(st + i * (8 * NW * GEMM_BK * 2), ao[i], ab);
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) dma16(st + (BM + 8 * NW * i) * (GEMM_BK * 2), bo[i], bb);
        };
        const int nsl = K / GEMM_BK;
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nsl) issue(p, p);
        const int frow = lane & 15, fchunk = lane >> 4;
        int stage = 0, nstage = NS - 1;
        for (int t = 0; t < nsl; ++t) {
            if (t + NS - 2 < nsl) wait_dma<(NS - 2) * NL>(); else wait_dma<0>();
            __builtin_amdgcn_s_barrier();
            if (t + NS - 1 < nsl) issue(t + NS - 1, nstage);
            const bf16_t* As = smem + stage * LDS_ELEMS;
            const bf16_t* Bs = As + BM * GEMM_BK;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 af[FM], bf[FN];
                const int chunk = kk * 4 + fchunk;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = wm * WROWS + i * 16 + frow;
                    af[i] = *reinterpret_cast<const u32x4*>(As + row * GEMM_BK + ((chunk ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = wn * (BN / 2) + j * 16 + frow;
                    bf[j] = *reinterpret_cast<const u32x4*>(Bs + row * GEMM_BK + ((chunk ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < FM; ++i) ssa[i] = sumsq8(af[i], ssa[i]);
#pragma unroll
                for (int j = 0; j < FN; ++j) ssb[j] = sumsq8(bf[j], ssb[j]);
            }
            stage = stage + 1 == NS ? 0 : stage + 1;
            nstage = nstage + 1 == NS ? 0 : nstage + 1;
        }
        // a lane holds the k-values of its 16-byte chunk column (fchunk): the four lane groups of a row complete the sum
#pragma unroll
        for (int i = 0; i < FM; ++i) { ssa[i] += __shfl_xor(ssa[i], 16, 64); ssa[i] += __shfl_xor(ssa[i], 32, 64); }
#pragma unroll
        for (int j = 0; j < FN; ++j) { ssb[j] += __shfl_xor(ssb[j], 16, 64); ssb[j] += __shfl_xor(ssb[j], 32, 64); }
    }

    // SWAP layout: f(row, col0, v) where v[r] is C[row][col0 + r].
    template <class F>
    __device__ __forceinline__ void for_each_cols(int m0, int n0, F&& f) {
        static_assert(SWAP, "for_each_cols needs the swapped fragment layout");
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                f(m0 + wm * WROWS + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, acc[i][j]);
    }

    // Visit every accumulator fragment: f(row0, col, v) where v[r] is C[row0 + r][col].
    template <class F>
    __device__ __forceinline__ void for_each(int m0, int n0, F&& f) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                f(m0 + wm * WROWS + i * 16 + (lane >> 4) * 4, n0 + wn * (BN / 2) + j * 16 + (lane & 15), acc[i][j]);
    }
};

// Slabs a split-K GEMM (plain fp32 output, e.ksplit = ks) leaves for its consumer: ks for the few-row shapes (real split-K:
// parallelism), 1 from GEMM_VSPLIT_ROWS rows on (the workgroups keep the slab sums apart internally: same bits, see the kernel).
constexpr int GEMM_VSPLIT_ROWS = 2401;   // = the row count from which the dispatcher uses the 2-stage pipelines
static inline int gemm_splitk_slabs(int M, int ks) { return (ks > 1 && M >= GEMM_VSPLIT_ROWS) ? 1 : ks; }

// Fused epilogue description shared by every ViT GEMM: the public splice_gemm_epilogue.
typedef splice_gemm_epilogue GemmEpi;
enum : unsigned {
    EPI_BIAS = SPLICE_EPI_BIAS, EPI_RESID = SPLICE_EPI_RESID, EPI_OUT_F32 = SPLICE_EPI_OUT_F32,
    EPI_OUT_BF = SPLICE_EPI_OUT_BF, EPI_OUT_T = SPLICE_EPI_OUT_T, EPI_GELU = SPLICE_EPI_GELU,
    EPI_GELU_GRAD = SPLICE_EPI_GELU_GRAD, EPI_COLS_F32 = SPLICE_EPI_COLS_F32, EPI_ALPHA = SPLICE_EPI_ALPHA,
    EPI_ROWDOT = SPLICE_EPI_ROWDOT, EPI_SCALE_RC = SPLICE_EPI_SCALE_RC, EPI_OUT_F8 = SPLICE_EPI_OUT_F8,
    EPI_OUT_F8T = SPLICE_EPI_OUT_F8T
};

template <unsigned FLAGS>
__device__ __forceinline__ void gemm_epilogue(const GemmEpi& e, int M, int N, int row0, int col, f32x4 v) {
    if (col >= N || row0 >= M) return;
    float b = 0.f;
    if (FLAGS & EPI_BIAS) b = e.bias[col];
    float x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        x[r] = v[r];
        if (FLAGS & EPI_ALPHA) x[r] *= e.alpha;
        x[r] += b;
    }
    if (FLAGS & EPI_RESID) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < M) {
                const int rr = e.resid_mod ? (row0 + r) % e.resid_mod : (row0 + r);
                x[r] += e.resid[(size_t)rr * e.ldr + col];
            }
    }
    if (FLAGS & EPI_GELU_GRAD) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < M) x[r] *= gelu_grad_f(bf2f(e.aux[(size_t)(row0 + r) * e.ldaux + col]));
    }
    if (FLAGS & EPI_GELU) {
        if (e.out_pre) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r < M) e.out_pre[(size_t)(row0 + r) * e.ldp + col] = f2bf(x[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = gelu_f(x[r]);
    }
    if (FLAGS & EPI_OUT_F32) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < M) e.out_f32[(size_t)(row0 + r) * e.ldo + col] = x[r];
    }
    if (FLAGS & EPI_OUT_BF) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < M) e.out_bf[(size_t)(row0 + r) * e.ldbf + col] = f2bf(x[r]);
    }
    if (FLAGS & EPI_OUT_T) {
        bf16_t* p = e.out_bf_t + (size_t)col * e.ldt + row0;
        if (row0 + 3 < M) {
            uint2 pk = {pack2bf(x[0], x[1]), pack2bf(x[2], x[3])};
            *reinterpret_cast<uint2*>(p) = pk;  // row0 % 4 == 0 and ldt % 4 == 0 -> 8-byte aligned
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r < M) p[r] = f2bf(x[r]);
        }
    }
    if (FLAGS & EPI_COLS_F32) {
        if (col >= e.col_lo && col < e.col_hi) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r < M) e.out_f32_cols[(size_t)(row0 + r) * e.ld_cols + (col - e.col_lo)] = x[r];
        }
    }
}

// Epilogue for the swapped fragment layout: lane holds C[row][col0 .. col0+3].
// PRE: the bias / residual / GELU' operands were loaded beforehand (gemm_nt_kernel issues the loads of every
// fragment before the first one is consumed); otherwise they are loaded here, one fragment at a time.
template <unsigned FLAGS, bool PRE = false>
__device__ __forceinline__ void gemm_epilogue_cols(const GemmEpi& e, int M, int N, int row, int col0, f32x4 v, float4 pre_bias = float4{},
                                                   float4 pre_resid = float4{}, uint2 pre_aux = uint2{}) {
    if (row >= M || col0 >= N) return;
    const bool full = PRE || col0 + 3 < N;
    float x[4] = {v[0], v[1], v[2], v[3]};
    if (FLAGS & EPI_ALPHA) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] *= e.alpha;
    }
    if (FLAGS & EPI_SCALE_RC) {   // de-quantisation of an fp8 product: per-row scale of A x per-column scale of B
        const float rs = e.row_scale ? e.row_scale[row] : 1.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] *= rs * e.col_scale[min(col0 + r, N - 1)];
    }
    if (FLAGS & EPI_BIAS) {
        if (full) {
            const float4 b = PRE ? pre_bias : *reinterpret_cast<const float4*>(e.bias + col0);
            x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (col0 + r < N) x[r] += e.bias[col0 + r];
        }
    }
    if (FLAGS & EPI_RESID) {
        const int rr = PRE ? 0 : (e.resid_mod ? row % e.resid_mod : row);
        const float* p = e.resid + (size_t)rr * e.ldr + col0;
        if (PRE || (full && !(e.ldr & 3))) {
            const float4 b = PRE ? pre_resid : *reinterpret_cast<const float4*>(p);
            x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (col0 + r < N) x[r] += p[r];
        }
    }
    if (FLAGS & EPI_GELU_GRAD) {
        const bf16_t* p = e.aux + (size_t)row * e.ldaux + col0;
        if (PRE || (full && !(e.ldaux & 3))) {
            const uint2 u = PRE ? pre_aux : *reinterpret_cast<const uint2*>(p);
            x[0] *= gelu_grad_f(bf2f((bf16_t)(u.x & 0xFFFF))); x[1] *= gelu_grad_f(bf2f((bf16_t)(u.x >> 16)));
            x[2] *= gelu_grad_f(bf2f((bf16_t)(u.y & 0xFFFF))); x[3] *= gelu_grad_f(bf2f((bf16_t)(u.y >> 16)));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (col0 + r < N) x[r] *= gelu_grad_f(bf2f(p[r]));
        }
    }
    auto store_bf = [&](bf16_t* base, int ld) {
        bf16_t* p = base + (size_t)row * ld + col0;
        if (full && !(ld & 3)) *reinterpret_cast<uint2*>(p) = uint2{pack2bf(x[0], x[1]), pack2bf(x[2], x[3])};
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (col0 + r < N) p[r] = f2bf(x[r]);
        }
    };
    if (FLAGS & EPI_GELU) {
        if (e.out_pre && row >= e.pre_row_lo) store_bf(e.out_pre, e.ldp);   // only gradient-carrying rows need it
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = gelu_f(x[r]);
    }
    if (FLAGS & EPI_OUT_F32) {
        float* p = e.out_f32 + (size_t)row * e.ldo + col0;
        if (full && !(e.ldo & 3)) *reinterpret_cast<float4*>(p) = float4{x[0], x[1], x[2], x[3]};
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (col0 + r < N) p[r] = x[r];
        }
    }
    if (FLAGS & EPI_OUT_BF) store_bf(e.out_bf, e.ldbf);
    if (FLAGS & EPI_OUT_T) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (col0 + r < N) e.out_bf_t[(size_t)(col0 + r) * e.ldt + row] = f2bf(x[r]);   // 16 lanes -> 32 contiguous bytes
    }
    if (FLAGS & EPI_COLS_F32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = col0 + r;
            if (c < N && c >= e.col_lo && c < e.col_hi) e.out_f32_cols[(size_t)row * e.ld_cols + (c - e.col_lo)] = x[r];
        }
    }
}

// ---- bf16 outputs through LDS ---------------------------------------------------------------------------------------
// In the swapped fragment layout a lane owns 4 consecutive columns of one row: a direct bf16 store is 8 bytes per lane and
// reaches memory as 32-byte row segments (2-byte elements for the transposed copy).  With P pairs per step the QKV launch
// (bf16 + transposed bf16) and fc1 (GELU + saved pre-activation) lost a third of their time to that.  The tile is staged
// in the ring buffers, idle after the K loop, and leaves as 16 bytes per lane along full rows (256 threads x 16 B = 4 KB of
// complete 128 / 256-byte lines per instruction); one pass per output tensor.
// Value of one fragment BEFORE the activation: alpha / fp8 scales / bias / residual / GELU' applied; the fp32 outputs (which
// keep their direct 16-byte stores) are written here.
template <unsigned FLAGS, bool PRE, bool STORE_F32 = true>
__device__ __forceinline__ f32x4 gemm_frag_value(const GemmEpi& e, int M, int N, int row, int col0, f32x4 v, float4 pre_bias, float4 pre_resid,
                                                 uint2 pre_aux) {
    const bool inb = row < M && col0 < N;
    const int rowc = min(row, M - 1), colc = min(col0, N - 4);   // N % 4 == 0 on this path
    float x[4] = {v[0], v[1], v[2], v[3]};
    if (FLAGS & EPI_ALPHA) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] *= e.alpha;
    }
    if (FLAGS & EPI_SCALE_RC) {
        const float rs = e.row_scale ? e.row_scale[rowc] : 1.0f;
        const float4 cs = *reinterpret_cast<const float4*>(e.col_scale + colc);
        x[0] *= rs * cs.x; x[1] *= rs * cs.y; x[2] *= rs * cs.z; x[3] *= rs * cs.w;
    }
    if (FLAGS & EPI_BIAS) {
        const float4 b = PRE ? pre_bias : *reinterpret_cast<const float4*>(e.bias + colc);
        x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
    }
    if (FLAGS & EPI_RESID) {
        const int rr = e.resid_mod ? rowc % e.resid_mod : rowc;
        const float4 b = PRE ? pre_resid : *reinterpret_cast<const float4*>(e.resid + (size_t)rr * e.ldr + colc);
        x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
    }
    if (FLAGS & EPI_GELU_GRAD) {
        const uint2 u = PRE ? pre_aux : *reinterpret_cast<const uint2*>(e.aux + (size_t)rowc * e.ldaux + colc);
        x[0] *= gelu_grad_f(bf2f((bf16_t)(u.x & 0xFFFF))); x[1] *= gelu_grad_f(bf2f((bf16_t)(u.x >> 16)));
        x[2] *= gelu_grad_f(bf2f((bf16_t)(u.y & 0xFFFF))); x[3] *= gelu_grad_f(bf2f((bf16_t)(u.y >> 16)));
    }
    if (inb) {
        if ((FLAGS & EPI_OUT_F32) && STORE_F32) *reinterpret_cast<float4*>(e.out_f32 + (size_t)row * e.ldo + col0) = float4{x[0], x[1], x[2], x[3]};
        if (FLAGS & EPI_COLS_F32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = col0 + r;
                if (c >= e.col_lo && c < e.col_hi) e.out_f32_cols[(size_t)row * e.ld_cols + (c - e.col_lo)] = x[r];
            }
        }
    }
    return f32x4{x[0], x[1], x[2], x[3]};
}
// one staged pass: every wave writes its fragments (bf16) into cs [BM][BN + 8] (or transposed [BN][BM + 8]), then the
// workgroup stores the tile with 16-byte chunks along the rows of `out` (row stride ld); rows < row_lo are not stored
template <int BM, int BN, bool TRANSPOSED>
__device__ __forceinline__ void gemm_stage_store(const f32x4 (&acc)[BM / 32][BN / 32], bf16_t* cs, bf16_t* out, int ld, int M, int N, int m0, int n0,
                                                 int row_lo) {
    constexpr int FMt = BM / 32, FNt = BN / 32;
    constexpr int R = TRANSPOSED ? BN : BM, Cc = TRANSPOSED ? BM : BN, PITCH = Cc + 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    __syncthreads();   // the ring (or the previous pass) is no longer read
#pragma unroll
    for (int i = 0; i < FMt; ++i)
#pragma unroll
        for (int j = 0; j < FNt; ++j) {
            const int lr = wm * (BM / 2) + i * 16 + (lane & 15), lc = wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
            const f32x4 v = acc[i][j];
            if (!TRANSPOSED) {
                *reinterpret_cast<uint2*>(cs + lr * PITCH + lc) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[(lc + r) * PITCH + lr] = f2bf(v[r]);
            }
        }
    __syncthreads();
    // tile coordinates in the OUTPUT's frame: rows r0.., columns c0.. ; limits Rm, Cm
    const int r0 = TRANSPOSED ? n0 : m0, c0 = TRANSPOSED ? m0 : n0, Rm = TRANSPOSED ? N : M, Cm = TRANSPOSED ? M : N;
    constexpr int CH = Cc / 8;   // 16-byte chunks per tile row
    for (int c = threadIdx.x; c < R * CH; c += 256) {
        const int r = c / CH, k = (c % CH) * 8;
        const int gr = r0 + r, gc = c0 + k;
        if (gr >= Rm || gc >= Cm) continue;
        if (!TRANSPOSED && gr < row_lo) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(cs + r * PITCH + k);
        bf16_t* q = out + (size_t)gr * ld + gc;
        if (gc + 7 < Cm && !(ld & 7)) st_out(reinterpret_cast<u32x4*>(q), v);
        else {
            const bf16_t* h = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (gc + t < Cm) q[t] = h[t];
        }
    }
}

// e4m3 output tile: cs bytes [BM][BN + 16]; a lane's 4 columns are one dword, the tile leaves as 16 bytes per lane along rows
__device__ __forceinline__ uint32_t pack4_e4m3_sat(const f32x4& v) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0], -448.f), 448.f), fminf(fmaxf(v[1], -448.f), 448.f), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2], -448.f), 448.f), fminf(fmaxf(v[3], -448.f), 448.f), w, true);
    return (uint32_t)w;
}
template <int BM, int BN>
__device__ __forceinline__ void gemm_stage_store_f8(const f32x4 (&acc)[BM / 32][BN / 32], uint8_t* cs, uint8_t* out, int ld, int M, int N, int m0, int n0) {
    constexpr int FMt = BM / 32, FNt = BN / 32, PITCH = BN + 16, CH = BN / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FMt; ++i)
#pragma unroll
        for (int j = 0; j < FNt; ++j) {
            const int lr = wm * (BM / 2) + i * 16 + (lane & 15), lc = wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<uint32_t*>(cs + lr * PITCH + lc) = pack4_e4m3_sat(acc[i][j]);
        }
    __syncthreads();
    for (int c = threadIdx.x; c < BM * CH; c += 256) {
        const int r = c / CH, k = (c % CH) * 16;
        const int gr = m0 + r, gc = n0 + k;
        if (gr >= M || gc >= N) continue;   // N % 16 == 0: a chunk is inside or outside as a whole
        st_out(reinterpret_cast<u32x4*>(out + (size_t)gr * ld + gc), *reinterpret_cast<const u32x4*>(cs + r * PITCH + k));
    }
}

// the transposed e4m3 tile: cs bytes [BN][BM + 16] (row = output column), stored as 16 bytes per lane along the rows of out [N][ld]
template <int BM, int BN>
__device__ __forceinline__ void gemm_stage_store_f8t(const f32x4 (&acc)[BM / 32][BN / 32], uint8_t* cs, uint8_t* out, int ld, int M, int N, int m0, int n0) {
    constexpr int FMt = BM / 32, FNt = BN / 32, PITCH = BM + 16, CH = BM / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FMt; ++i)
#pragma unroll
        for (int j = 0; j < FNt; ++j) {
            const int lr = wm * (BM / 2) + i * 16 + (lane & 15), lc = wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
            const uint32_t w = pack4_e4m3_sat(acc[i][j]);
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[(lc + r) * PITCH + lr] = (uint8_t)(w >> (8 * r));
        }
    __syncthreads();
    for (int c = threadIdx.x; c < BN * CH; c += 256) {
        const int r = c / CH, k = (c % CH) * 16;
        const int gr = n0 + r, gc = m0 + k;
        if (gr >= N || gc >= M) continue;
        uint8_t* q = out + (size_t)gr * ld + gc;
        if (gc + 15 < M) st_out(reinterpret_cast<u32x4*>(q), *reinterpret_cast<const u32x4*>(cs + r * PITCH + k));
        else
            for (int t = 0; t < 16 && gc + t < M; ++t) q[t] = cs[r * PITCH + k + t];
    }
}

// the same for an fp32 output tile: cs [BM][BN + 4] floats; a lane's float4 becomes part of a full 256-byte row line
template <int BM, int BN>
__device__ __forceinline__ void gemm_stage_store_f32(const f32x4 (&acc)[BM / 32][BN / 32], float* cs, float* out, int ld, int M, int N, int m0, int n0) {
    constexpr int FMt = BM / 32, FNt = BN / 32, PITCH = BN + 4, CH = BN / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FMt; ++i)
#pragma unroll
        for (int j = 0; j < FNt; ++j) {
            const int lr = wm * (BM / 2) + i * 16 + (lane & 15), lc = wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<f32x4*>(cs + lr * PITCH + lc) = acc[i][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < BM * CH; c += 256) {
        const int r = c / CH, k = (c % CH) * 4;
        const int gr = m0 + r, gc = n0 + k;
        if (gr >= M || gc >= N) continue;   // N % 4 == 0: a chunk is inside or outside as a whole
        st_out(reinterpret_cast<f32x4*>(out + (size_t)gr * ld + gc), *reinterpret_cast<const f32x4*>(cs + r * PITCH + k));
    }
}

template <int BM, int BN, unsigned FLAGS, int NS, bool FP8 = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                                      int ldb, int M, int N, int K, int gm, int ksplit, GemmEpi e) {
    // (argument order: what the tile mapping and the K loop need comes first, as scalars -- the leading kernel arguments are
    // preloaded into SGPRs by the dispatcher (-amdgpu-kernarg-preload-count), the epilogue description is fetched meanwhile)
    extern __shared__ __attribute__((aligned(16))) bf16_t gemm_smem[];   // NS * LDS_ELEMS
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int nwg = gridDim.x;
    int t = xcd_remap(blockIdx.x, nwg);
    int kbeg = 0;
    if (ksplit > 1) {   // split-K (plain fp32 output only): slice s of a tile writes slab s; the consumer sums the slabs in order
        const int slice = t % ksplit;
        t /= ksplit;
        K /= ksplit;
        kbeg = slice * K;
        e.out_f32 += (size_t)slice * e.slab_stride;
    }
    int tm, tn;
    grouped_tile(t, tiles_m, tiles_n, gm, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    GemmTile<BM, BN, true, FP8> tile;
    bool walked = false;
    if constexpr (FLAGS == EPI_OUT_F32 && NS == 2) {
        if (ksplit < -1) {
            // "virtual" split-K for many rows: ONE workgroup walks the whole K but keeps the -ksplit slab sums apart and adds them in
            // slab order -- bit for bit what the real split-K (needed by the few-row shapes for parallelism) plus the consumer's
            // in-order slab sum produce, without writing and re-reading the slabs.  A pair's gradient does not depend on which
            // form a batch size selects.
            const int vs = -ksplit, Kp = K / vs;
            f32x4 tot[GemmTile<BM, BN>::FM][GemmTile<BM, BN>::FN];
            for (int sl = 0; sl < vs; ++sl) {
                if (sl) __syncthreads();   // the previous walk's last slice is still being read
                tile.run_glds(A, lda, B, ldb, M, N, Kp, m0, n0, gemm_smem, sl * Kp);
#pragma unroll
                for (int i = 0; i < GemmTile<BM, BN>::FM; ++i)
#pragma unroll
                    for (int j = 0; j < GemmTile<BM, BN>::FN; ++j) tot[i][j] = sl ? tot[i][j] + tile.acc[i][j] : tile.acc[i][j];
            }
#pragma unroll
            for (int i = 0; i < GemmTile<BM, BN>::FM; ++i)
#pragma unroll
                for (int j = 0; j < GemmTile<BM, BN>::FN; ++j) tile.acc[i][j] = tot[i][j];
            walked = true;
        }
    }
    if (walked) {
    } else if (NS == 2) tile.run_glds(A, lda, B, ldb, M, N, K, m0, n0, gemm_smem, kbeg);
    else tile.template run_ring<(NS < 3 ? 3 : NS)>(A, lda, B, ldb, M, N, K, m0, n0, gemm_smem, kbeg);
    // Epilogue operands: lane-guarded loads inside the per-fragment epilogue compile to load + s_waitcnt per fragment,
    // one memory round trip each (4 ... 16 per lane, the residual usually an L2 miss).  With N and the leading dimensions
    // multiples of 4 every fragment's operands are loaded first, from clamped (always valid) addresses.
    constexpr bool HAS_OPERANDS = (FLAGS & (EPI_BIAS | EPI_RESID | EPI_GELU_GRAD)) != 0;
    constexpr bool STAGE_F32_FITS = (size_t)NS * GemmTile<BM, BN>::LDS_ELEMS * 2 >= (size_t)BM * (BN + 4) * 4;
    const bool pre_ok = HAS_OPERANDS && !(N & 3) && (!(FLAGS & EPI_RESID) || !(e.ldr & 3)) && (!(FLAGS & EPI_GELU_GRAD) || !(e.ldaux & 3));
    // staged bf16 outputs (see gemm_stage_store): every vector access of gemm_frag_value must be aligned
    const bool stage_ok = (FLAGS & EPI_OUT_BF) && !(N & 3) && (!(FLAGS & EPI_RESID) || !(e.ldr & 3)) && (!(FLAGS & EPI_GELU_GRAD) || !(e.ldaux & 3)) &&
                          (!(FLAGS & EPI_OUT_F32) || !(e.ldo & 3)) && (!(FLAGS & EPI_OUT_T) || !(e.ldt & 7));
    if constexpr ((FLAGS & EPI_OUT_F8) != 0 && (FLAGS & EPI_OUT_BF) == 0) {
        // e4m3 output (fc1 of the fp8 MLP): [bf16 pre-activation of the gradient-carrying rows,] GELU, saturated e4m3 tile
        constexpr int FMt = GemmTile<BM, BN>::FM, FNt = GemmTile<BM, BN>::FN;
        static_assert((size_t)NS * GemmTile<BM, BN>::LDS_ELEMS >= (size_t)BM * (BN + 8), "the ring must hold one staged output tile");
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
        float4 pb[FNt];
#pragma unroll
        for (int j = 0; j < FNt; ++j)
            if (FLAGS & EPI_BIAS) pb[j] = *reinterpret_cast<const float4*>(e.bias + min(n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, N - 4));
#pragma unroll
        for (int i = 0; i < FMt; ++i)
#pragma unroll
            for (int j = 0; j < FNt; ++j)
                tile.acc[i][j] = gemm_frag_value<FLAGS, true>(e, M, N, m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4,
                                                              tile.acc[i][j], pb[j], float4{}, uint2{});
        if (FLAGS & EPI_GELU) {
            if (e.out_pre) gemm_stage_store<BM, BN, false>(tile.acc, gemm_smem, e.out_pre, e.ldp, M, N, m0, n0, e.pre_row_lo);
#pragma unroll
            for (int i = 0; i < FMt; ++i)
#pragma unroll
                for (int j = 0; j < FNt; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile.acc[i][j][r] = gelu_f(tile.acc[i][j][r]);
        }
        gemm_stage_store_f8<BM, BN>(tile.acc, reinterpret_cast<uint8_t*>(gemm_smem), e.out_f8, e.ld8, M, N, m0, n0);
    } else if ((FLAGS & EPI_OUT_BF) && stage_ok) {
        constexpr int FMt = GemmTile<BM, BN>::FM, FNt = GemmTile<BM, BN>::FN;
        static_assert((size_t)NS * GemmTile<BM, BN>::LDS_ELEMS >= (size_t)BM * (BN + 8) && (size_t)NS * GemmTile<BM, BN>::LDS_ELEMS >= (size_t)BN * (BM + 8),
                      "the ring must hold one staged output tile");
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
        float4 pb[FNt], pr[FMt][FNt];
        uint2 pa[FMt][FNt];
        if (HAS_OPERANDS) {   // operands of every fragment before the first use (clamped addresses)
#pragma unroll
            for (int j = 0; j < FNt; ++j) {
                const int colc = min(n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, N - 4);
                if (FLAGS & EPI_BIAS) pb[j] = *reinterpret_cast<const float4*>(e.bias + colc);
#pragma unroll
                for (int i = 0; i < FMt; ++i) {
                    const int rowc = min(m0 + wm * (BM / 2) + i * 16 + (lane & 15), M - 1);
                    const int rr = e.resid_mod ? rowc % e.resid_mod : rowc;
                    if (FLAGS & EPI_RESID) pr[i][j] = *reinterpret_cast<const float4*>(e.resid + (size_t)rr * e.ldr + colc);
                    if (FLAGS & EPI_GELU_GRAD) pa[i][j] = *reinterpret_cast<const uint2*>(e.aux + (size_t)rowc * e.ldaux + colc);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FMt; ++i)
#pragma unroll
            for (int j = 0; j < FNt; ++j)
                tile.acc[i][j] = gemm_frag_value<FLAGS, HAS_OPERANDS>(e, M, N, m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4,
                                                                      tile.acc[i][j], pb[j], pr[i][j], pa[i][j]);
        if (FLAGS & EPI_GELU) {
            if (e.out_pre) gemm_stage_store<BM, BN, false>(tile.acc, gemm_smem, e.out_pre, e.ldp, M, N, m0, n0, e.pre_row_lo);   // only gradient-carrying rows need it
#pragma unroll
            for (int i = 0; i < FMt; ++i)
#pragma unroll
                for (int j = 0; j < FNt; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile.acc[i][j][r] = gelu_f(tile.acc[i][j][r]);
        }
        gemm_stage_store<BM, BN, false>(tile.acc, gemm_smem, e.out_bf, e.ldbf, M, N, m0, n0, 0);
        if (FLAGS & EPI_OUT_T) gemm_stage_store<BM, BN, true>(tile.acc, gemm_smem, e.out_bf_t, e.ldt, M, N, m0, n0, 0);
        if constexpr ((FLAGS & EPI_OUT_F8) != 0) gemm_stage_store_f8<BM, BN>(tile.acc, reinterpret_cast<uint8_t*>(gemm_smem), e.out_f8, e.ld8, M, N, m0, n0);
        if constexpr ((FLAGS & EPI_OUT_F8T) != 0) gemm_stage_store_f8t<BM, BN>(tile.acc, reinterpret_cast<uint8_t*>(gemm_smem), e.out_f8_t, e.ldt8, M, N, m0, n0);
        if (FLAGS & EPI_ROWDOT) {   // the row dots below use the bf16-ROUNDED result: round the accumulators in place
#pragma unroll
            for (int i = 0; i < FMt; ++i)
#pragma unroll
                for (int j = 0; j < FNt; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile.acc[i][j][r] = bf2f(f2bf(tile.acc[i][j][r]));
        }
    } else if (STAGE_F32_FITS && (FLAGS & EPI_OUT_F32) && !(FLAGS & (EPI_OUT_BF | EPI_COLS_F32)) && !(N & 3) && !(e.ldo & 3) && (!(FLAGS & EPI_RESID) || !(e.ldr & 3))) {
        // fp32 output (proj / fc2 forward, the dgrads into LayerNorm backward): staged like the bf16 tiles (where the ring holds the tile)
        constexpr int FMt = GemmTile<BM, BN>::FM, FNt = GemmTile<BM, BN>::FN;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
        float4 pb[FNt], pr[FMt][FNt];
        if (HAS_OPERANDS) {
#pragma unroll
            for (int j = 0; j < FNt; ++j) {
                const int colc = min(n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, N - 4);
                if (FLAGS & EPI_BIAS) pb[j] = *reinterpret_cast<const float4*>(e.bias + colc);
#pragma unroll
                for (int i = 0; i < FMt; ++i) {
                    const int rowc = min(m0 + wm * (BM / 2) + i * 16 + (lane & 15), M - 1);
                    const int rr = e.resid_mod ? rowc % e.resid_mod : rowc;
                    if (FLAGS & EPI_RESID) pr[i][j] = *reinterpret_cast<const float4*>(e.resid + (size_t)rr * e.ldr + colc);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FMt; ++i)
#pragma unroll
            for (int j = 0; j < FNt; ++j)
                tile.acc[i][j] = gemm_frag_value<FLAGS, HAS_OPERANDS, false>(e, M, N, m0 + wm * (BM / 2) + i * 16 + (lane & 15),
                                                                             n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, tile.acc[i][j], pb[j], pr[i][j], uint2{});
        if constexpr (STAGE_F32_FITS) gemm_stage_store_f32<BM, BN>(tile.acc, reinterpret_cast<float*>(gemm_smem), e.out_f32, e.ldo, M, N, m0, n0);
    } else if (pre_ok) {
        constexpr int FMt = GemmTile<BM, BN>::FM, FNt = GemmTile<BM, BN>::FN;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
        float4 pb[FNt], pr[FMt][FNt];
        uint2 pa[FMt][FNt];
        int colc[FNt];
#pragma unroll
        for (int j = 0; j < FNt; ++j) {
            colc[j] = min(n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, N - 4);
            if (FLAGS & EPI_BIAS) pb[j] = *reinterpret_cast<const float4*>(e.bias + colc[j]);
        }
#pragma unroll
        for (int i = 0; i < FMt; ++i) {
            const int rowc = min(m0 + wm * (BM / 2) + i * 16 + (lane & 15), M - 1);
            const int rr = e.resid_mod ? rowc % e.resid_mod : rowc;
#pragma unroll
            for (int j = 0; j < FNt; ++j) {
                if (FLAGS & EPI_RESID) pr[i][j] = *reinterpret_cast<const float4*>(e.resid + (size_t)rr * e.ldr + colc[j]);
                if (FLAGS & EPI_GELU_GRAD) pa[i][j] = *reinterpret_cast<const uint2*>(e.aux + (size_t)rowc * e.ldaux + colc[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < FMt; ++i)
#pragma unroll
            for (int j = 0; j < FNt; ++j)
                gemm_epilogue_cols<FLAGS, true>(e, M, N, m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4,
                                                tile.acc[i][j], pb[j], pr[i][j], pa[i][j]);
    } else {
        tile.for_each_cols(m0, n0, [&](int row, int col0, f32x4 v) { gemm_epilogue_cols<FLAGS>(e, M, N, row, col0, v); });
    }
    if (FLAGS & EPI_ROWDOT) {
        // 64-column row dots of the bf16-rounded result against rd_other (BN == 64: one workgroup = one 64-column block):
        // lane partial over its 4 columns x 2 fragments -> 4 lane groups (shuffles) -> the 2 waves of a row (LDS)
        static_assert(!(FLAGS & EPI_ROWDOT) || BN == 64, "EPI_ROWDOT needs the 64-column tile");
        constexpr int FMt = GemmTile<BM, BN>::FM, FNt = GemmTile<BM, BN>::FN;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
        float* red = reinterpret_cast<float*>(gemm_smem);   // [2 wn][BM] (the K loop is over)
        __syncthreads();
        uint2 ov[FMt][FNt];   // every fragment's operand is loaded before the first use (clamped row: the loads are unconditional)
#pragma unroll
        for (int i = 0; i < FMt; ++i)
#pragma unroll
            for (int j = 0; j < FNt; ++j)
                ov[i][j] = *reinterpret_cast<const uint2*>(e.rd_other + (size_t)min(m0 + wm * (BM / 2) + i * 16 + (lane & 15), M - 1) * e.ld_rd + n0 +
                                                           wn * (BN / 2) + j * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int i = 0; i < FMt; ++i) {
            const int lrow = wm * (BM / 2) + i * 16 + (lane & 15);
            const int row = m0 + lrow;
            float part = 0.f;
            if (row < M) {
#pragma unroll
                for (int j = 0; j < FNt; ++j) {
                    const uint2 o = ov[i][j];
                    const f32x4 v = tile.acc[i][j];
                    part += bf2f(f2bf(v[0])) * bf2f((bf16_t)(o.x & 0xFFFF)) + bf2f(f2bf(v[1])) * bf2f((bf16_t)(o.x >> 16)) +
                            bf2f(f2bf(v[2])) * bf2f((bf16_t)(o.y & 0xFFFF)) + bf2f(f2bf(v[3])) * bf2f((bf16_t)(o.y >> 16));
                }
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (lane < 16) red[wn * BM + lrow] = part;
        }
        __syncthreads();
        if (threadIdx.x < BM) {
            const int row = m0 + threadIdx.x;
            if (row < M)
                e.rowdot[((size_t)(row / e.rd_rows) * (N / 64) + n0 / 64) * e.rd_rows + row % e.rd_rows] = red[threadIdx.x] + red[BM + threadIdx.x];
        }
    }
}

template <int BM, int BN, unsigned FLAGS, int NS, bool FP8 = false>
static inline void launch_gemm_nt(hipStream_t s, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K,
                                  const GemmEpi& e) {
    const int tm = cdiv(M, BM), tn = cdiv(N, BN);
    int ksplit = (FLAGS == EPI_OUT_F32 && e.ksplit > 1 && K % (e.ksplit * GEMM_BK) == 0) ? e.ksplit : 1;
    if (NS == 2 && ksplit > 1 && gemm_splitk_slabs(M, ksplit) == 1) ksplit = -ksplit;   // virtual: one slab, the split happens inside the workgroup
    const int grid = tm * tn * (ksplit > 1 ? ksplit : 1);
    // group height ~ sqrt(tiles per XCD), weighted by the tile aspect so the block is square in elements
    int gm = 1;
    while ((gm + 1) * (gm + 1) * BM <= (tm * tn / 8 + 1) * BN && gm + 1 <= tm) ++gm;
    constexpr size_t lds = (size_t)NS * GemmTile<BM, BN>::LDS_ELEMS * sizeof(bf16_t);
    if (lds > 65536) {   // > 64 KB of dynamic LDS has to be allowed once per kernel AND per device (the attribute lives on the
        int dev = 0;     // device's copy of the function: a process that drives several GPUs sets it on each)
        (void)hipGetDevice(&dev);
        static std::atomic<unsigned long long> done_mask{0};
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, FLAGS, NS, FP8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            done_mask.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    SPLICE_LAUNCH((gemm_nt_kernel<BM, BN, FLAGS, NS, FP8>), dim3(grid), dim3(256), lds, s, A, lda, B, ldb, M, N, K, gm, ksplit, e);
}
