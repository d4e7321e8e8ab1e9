// 256 x 256 bf16 "NT" GEMM tile on an 8-phase schedule for gfx950 (the batched ViT shapes: M >= 6400 rows, N >= 2304):
//   C[M][N] = sum_k A[m][k] * B[n][k], fp32 accumulate, same MFMA (16x16x32 bf16, swapped operands), same k order per
//   accumulator and therefore the SAME BITS as the 128 x 128 / 128 x 64 / 64 x 64 tiles of gemm.h.
//
// Why a second main loop: the one-barrier-per-slice loops of gemm.h stop at ~35 % of the bf16 MFMA peak on these shapes -- all
// waves of a workgroup load, wait, then compute, and the L2 -> LDS feed of a 128 x 128 tile (32 KB per 16 MFMAs per wave) is
// exposed (profiles/r03_gemm_ablation.txt: +48 % with the DMA compiled out).  Here (CDNA4 guide, "256^2 8-phase"):
//   * 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 32 accumulator fragments; half the LDS-DMA bytes per FLOP;
//   * a K tile (BK = 64) is FOUR half-tiles of 16 KB: B rows 0..127 (q = 0), A rows 0..127 (q = 1), B rows 128..255 (q = 2),
//     A rows 128..255 (q = 3).  A wave's 128 rows are rows wr*64..+63 of BOTH A halves and its 64 columns are columns wc*32..+31
//     of BOTH B halves, so every half-tile is read in exactly one phase: q0 + q1 in phase 1, q2 in phase 2, q3 in phase 3,
//     nothing in phase 4 (the B fragments of phase 1 stay in registers for quadrant (1, 0));
//   * one half-tile is staged per phase by LDS-DMA, SEVEN half-tiles ahead of its first reader, into an 8-slot ring (128 KB);
//     the only DMA wait is a counted vmcnt(6) in the last phase of a K tile (three half-tiles stay in flight across it);
//   * each phase is [fragment reads + DMA issue] barrier [16 MFMAs] barrier, and the two wave halves (wr = 0 / 1: one wave of
//     each per SIMD) run one barrier apart: while one half's MFMAs own the matrix pipe, the other half reads and stages.
// Hazards (guide, "Read a staged buffer one phase AFTER the wait that retires it"): RAW -- the counted wait sits before the
// first barrier of phase 4, the first read of that K tile is in the next phase; WAR -- slot of q = p - 1 is re-staged in phase
// p: q1, q2, q3 were last read two phases earlier; q0 (read in phase 1, re-staged in phase 2) has its four reads retired by an
// lgkmcnt(8) BEFORE phase 1's first barrier.
#pragma once
#include <type_traits>
#include "gemm.h"

struct Gemm8p {
    static constexpr int BM = 256, BN = 256, NTHREADS = 512;
    static constexpr int SLOT_ELEMS = 128 * GEMM_BK;          // one half-tile: 128 rows x 64 k (16 KB)
    static constexpr int LDS_BYTES = 8 * SLOT_ELEMS * 2;      // 8-slot ring = 128 KB
    f32x4 acc[8][4];                                          // [mh * 4 + i][nh * 2 + j]
    u32x4 af[4][2];                                           // A fragments of the current 64-row sub-tile: [i][kk]
    u32x4 bfr[2][2][2];                                       // B fragments of both 32-column sub-tiles: [nh][j][kk]

    template <int N>
    static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
    template <int N>
    static __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
    static __device__ __forceinline__ void dma16(uint32_t lds_byte, uint32_t voff, const void* sbase) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(voff), "s"(sbase) : "memory");
    }

    // origin of one output tile; the per-lane LDS-DMA source offsets are formed at issue time (3 VALU per DMA piece) instead of
    // living in 8 VGPRs per tile: the accumulators leave little room
    struct Src { int m0, n0; };
    // PERSISTENT form: the workgroup walks `count` output tiles (tile r at coords(r) -> m0, n0) as ONE continuous stream of K
    // tiles: the half-tile prefetch runs on into the next output tile's first two K tiles while the current tile finishes, so
    // neither the DMA ring's fill nor the output stores of a tile (epi(m0, n0): registers -> global, asynchronous) leave the
    // matrix pipe idle for longer than the epilogue's own instructions.  K % 128 == 0, K >= 128.
    // (vmcnt also counts the epilogue's stores: a counted wait behind them can only wait LONGER than needed -- loads retire in
    // order among themselves -- never shorter.)
    template <class Coords, class Epi>
    __device__ __forceinline__ void run_tiles(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K, int count,
                                              Coords&& coords, Epi&& epi, bf16_t* smem) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wr = wave >> 2, wc = wave & 3;
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) bf16_t*)smem + (uint32_t)wave * 1024u;
        const int nt = K / GEMM_BK;
        // a wave-instruction fills 8 rows x 128 B (dest = piece base + lane * 16); wave w moves pieces w and w + 8 of a 128-row
        // half-tile; the XOR swizzle of the 16-byte chunks (chunk ^= row & 7) is applied on the source side; rows past the
        // matrix are clamped (their products land in accumulator rows / columns the epilogue does not store)
        const int prow = wave * 8 + (lane >> 3);                              // row of this lane's chunk inside piece 0
        const uint32_t pcol = (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);   // byte offset of its (swizzled) source chunk
        auto issue = [&](const Src& sc, int tile, auto qc, auto slotc) {
            constexpr int q = decltype(qc)::value, slot = decltype(slotc)::value;
            constexpr bool isA = q == 1 || q == 3;
            constexpr int h = q >> 1;   // q0 = B half 0, q1 = A half 0, q2 = B half 1, q3 = A half 1
            const bf16_t* src = (isA ? A : B) + (size_t)tile * GEMM_BK;
            const uint32_t d = lds0 + (uint32_t)slot * (SLOT_ELEMS * 2);
            const int r0 = (isA ? sc.m0 : sc.n0) + h * 128 + prow, lim = (isA ? M : N) - 1;
            const uint32_t ld2 = (uint32_t)(isA ? lda : ldb) * 2u;
            dma16(d, (uint32_t)min(r0, lim) * ld2 + pcol, src);
            dma16(d + 8192u, (uint32_t)min(r0 + 64, lim) * ld2 + pcol, src);
        };
        const int frow = lane & 15, fchunk = lane >> 4;
        const bf16_t* arow = smem + (wr * 64 + frow) * GEMM_BK;
        const bf16_t* brow = smem + (wc * 32 + frow) * GEMM_BK;
        const int ck0 = ((0 * 4 + fchunk) ^ (frow & 7)) << 3, ck1 = ((1 * 4 + fchunk) ^ (frow & 7)) << 3;
        auto read_a = [&](auto slotc) {
            constexpr int slot = decltype(slotc)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *reinterpret_cast<const u32x4*>(arow + slot * SLOT_ELEMS + i * 16 * GEMM_BK + ck0);
                af[i][1] = *reinterpret_cast<const u32x4*>(arow + slot * SLOT_ELEMS + i * 16 * GEMM_BK + ck1);
            }
        };
        auto read_b = [&](auto slotc, auto nhc) {
            constexpr int slot = decltype(slotc)::value, nh = decltype(nhc)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bfr[nh][j][0] = *reinterpret_cast<const u32x4*>(brow + slot * SLOT_ELEMS + j * 16 * GEMM_BK + ck0);
                bfr[nh][j][1] = *reinterpret_cast<const u32x4*>(brow + slot * SLOT_ELEMS + j * 16 * GEMM_BK + ck1);
            }
        };
        auto quadrant = [&](auto mhc, auto nhc) {
            constexpr int mh = decltype(mhc)::value, nh = decltype(nhc)::value;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[mh * 4 + i][nh * 2 + j] = mfma16(bfr[nh][j][kk], af[i][kk], acc[mh * 4 + i][nh * 2 + j]);
            __builtin_amdgcn_s_setprio(0);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        // one K tile = 4 phases.  s0 / t0: source set and K tile of the half-tile staged in phase 1 (q3), s1 / t1: of the three
        // staged in phases 2 .. 4 (q0 .. q2); on0 / on1: whether they exist
        auto ktile = [&](auto parc, const Src& s0, int t0, bool on0, const Src& s1, int t1, bool on1, int waitn) {
            constexpr int PAR = decltype(parc)::value;   // waitn: vmcnt of phase 4 -- 6, 0, or -1 (no wait); wave-uniform
            using S0 = std::integral_constant<int, PAR * 4 + 0>; using S1 = std::integral_constant<int, PAR * 4 + 1>;
            using S2 = std::integral_constant<int, PAR * 4 + 2>; using S3 = std::integral_constant<int, PAR * 4 + 3>;
            using O3 = std::integral_constant<int, (1 - PAR) * 4 + 3>;
            // ---- phase 1: B sub-tile 0 (4 reads, first), A sub-tile 0 (8 reads); stage q3 of the next K tile (slot of the previous tile's q3)
            read_b(S0{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            read_a(S1{});
            __builtin_amdgcn_sched_barrier(0);
            if (on0) issue(s0, t0, I3{}, O3{});
            wait_lgkm<8>();                       // the 4 B reads are back: slot q0 may be re-staged in the next phase
            __builtin_amdgcn_s_barrier();
            wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            quadrant(I0{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2: B sub-tile 1; stage q0 of the K tile after next (slot of this tile's q0)
            read_b(S2{}, I1{});
            __builtin_amdgcn_sched_barrier(0);
            if (on1) issue(s1, t1, I0{}, S0{});
            __builtin_amdgcn_s_barrier();
            wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            quadrant(I0{}, I1{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3: A sub-tile 1; stage q1
            read_a(S3{});
            __builtin_amdgcn_sched_barrier(0);
            if (on1) issue(s1, t1, I1{}, S1{});
            __builtin_amdgcn_s_barrier();
            wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            quadrant(I1{}, I1{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 4: no reads; stage q2; the counted wait that retires the next K tile
            if (on1) issue(s1, t1, I2{}, S2{});
            if (waitn == 6) wait_vm<6>();
            else if (waitn == 0) wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            quadrant(I1{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        if (count < 1) return;
        Src cur, nxt;
        coords(0, cur.m0, cur.n0);
        nxt = cur;
        if (count > 1) coords(1, nxt.m0, nxt.n0);
        // ---- prologue: half-tiles 0 .. 6 (K tile 0 complete, q0..q2 of K tile 1), K tile 0 landed
        issue(cur, 0, I0{}, I0{}); issue(cur, 0, I1{}, I1{}); issue(cur, 0, I2{}, I2{}); issue(cur, 0, I3{}, I3{});
        issue(cur, 1, I0{}, std::integral_constant<int, 4>{}); issue(cur, 1, I1{}, std::integral_constant<int, 5>{}); issue(cur, 1, I2{}, std::integral_constant<int, 6>{});
        wait_vm<6>();
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave half runs one barrier behind the first
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ONE loop over pairs of K tiles of the whole tile stream (two instances of the K-tile body in the kernel): K tile u of
        // the current output tile while u < nt, else K tile u - nt of the next one (staged only if there is a next one)
        int r = 0, t = 0;
        const int pairs = count * (nt >> 1);
#pragma unroll 1
        for (int it = 0; it < pairs; ++it) {
            const bool more = r + 1 < count;
            const bool in2 = t + 2 < nt, in3 = t + 3 < nt;
            const Src s2 = in2 ? cur : nxt, s3 = in3 ? cur : nxt;
            const int k2 = in2 ? t + 2 : 0, k3 = in3 ? t + 3 : 1;
            const bool on2 = in2 || more, on3 = in3 || more;
            ktile(I0{}, cur, t + 1, true, s2, k2, on2, on2 ? 6 : 0);
            ktile(I1{}, s2, k2, on2, s3, k3, on3, on3 ? 6 : -1);
            t += 2;
            if (t == nt) {
                epi(cur.m0, cur.n0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                cur = nxt;
                ++r;
                t = 0;
                if (r + 1 < count) coords(r + 1, nxt.m0, nxt.n0);
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the barrier count of the two halves
    }
};

// ---- product kernel: the persistent 8-phase tile behind two ViT forward epilogues ------------------------------------------------
//   EPI_BIAS | EPI_OUT_BF              QKV projection (layers that need no transposed / fp32 copy)
//   EPI_BIAS | EPI_GELU | EPI_OUT_BF   fc1: + the bf16 pre-activation of the gradient-carrying rows (e.out_pre, rows >= e.pre_row_lo)
// Two adjacent 16-column fragments of a row leave as ONE 16-byte store per lane: v_permlane16_swap exchanges the 4-column groups of
// lanes 16 apart, so a lane ends up with 8 consecutive columns (the epilogue has no LDS to stage through: the ring is already
// receiving the next tile).  N % 256 == 0 (whole tiles along N; ragged M is masked), K % 128 == 0.
template <unsigned FLAGS>
__global__ __launch_bounds__(512) void gemm8p_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int M, int N, int K, int gm,
                                                     GemmEpi e) {
    static_assert(FLAGS == (EPI_BIAS | EPI_OUT_BF) || FLAGS == (EPI_BIAS | EPI_GELU | EPI_OUT_BF) || FLAGS == (EPI_BIAS | EPI_RESID | EPI_OUT_F32),
                  "epilogues of the 8-phase tile");
    extern __shared__ __attribute__((aligned(16))) bf16_t gemm8p_smem[];
    const int tiles_n = N / 256, tiles_m = (M + 255) / 256, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int count = (ntiles - (int)blockIdx.x + G - 1) / G;   // tiles blockIdx.x, blockIdx.x + G, ... (G % 8 == 0: all on this workgroup's XCD)
    Gemm8p g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const int grp = lane >> 4;
    const int c8 = ((grp & 1) << 4) | ((grp >> 1) << 3);   // first of the lane's 8 columns after the swap: {0, 16, 8, 24}[grp]
    g.run_tiles(A, lda, B, ldb, M, N, K, count,
                [&](int r, int& m0, int& n0) {
                    const int t = xcd_remap(blockIdx.x + r * G, ntiles);
                    int tm, tn;
                    grouped_tile(t, tiles_m, tiles_n, gm, tm, tn);
                    m0 = tm * 256; n0 = tn * 256;
                },
                [&](int m0, int n0) {
                    if constexpr ((FLAGS & EPI_OUT_F32) != 0) {
                        // bias + residual -> fp32 (proj / fc2 forward, e.resid_mod == 0): a lane's 4 columns are one 16-byte access; the
                        // residual rows of a 64 x 32 quadrant are requested together, ahead of their use
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) {
                            const int colf = n0 + nh * 128 + wc * 32 + grp * 4;
                            const float4 b0 = *reinterpret_cast<const float4*>(e.bias + colf), b1 = *reinterpret_cast<const float4*>(e.bias + colf + 16);
#pragma unroll
                            for (int mh = 0; mh < 2; ++mh) {
                                float4 r0[4], r1[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int row = min(m0 + mh * 128 + wr * 64 + i * 16 + (lane & 15), M - 1);
                                    r0[i] = *reinterpret_cast<const float4*>(e.resid + (size_t)row * e.ldr + colf);
                                    r1[i] = *reinterpret_cast<const float4*>(e.resid + (size_t)row * e.ldr + colf + 16);
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int row = m0 + mh * 128 + wr * 64 + i * 16 + (lane & 15);
                                    const f32x4 v0 = g.acc[mh * 4 + i][nh * 2 + 0], v1 = g.acc[mh * 4 + i][nh * 2 + 1];
                                    if (row < M) {
                                        float* q = e.out_f32 + (size_t)row * e.ldo + colf;
                                        *reinterpret_cast<float4*>(q) = float4{v0[0] + b0.x + r0[i].x, v0[1] + b0.y + r0[i].y, v0[2] + b0.z + r0[i].z, v0[3] + b0.w + r0[i].w};
                                        *reinterpret_cast<float4*>(q + 16) = float4{v1[0] + b1.x + r1[i].x, v1[1] + b1.y + r1[i].y, v1[2] + b1.z + r1[i].z, v1[3] + b1.w + r1[i].w};
                                    }
                                }
                            }
                        }
                        return;
                    }
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        const int colf = n0 + nh * 128 + wc * 32 + grp * 4;   // this lane's 4 columns of fragment j = 0 (j = 1: + 16)
                        const float4 b0 = *reinterpret_cast<const float4*>(e.bias + colf), b1 = *reinterpret_cast<const float4*>(e.bias + colf + 16);
                        const int col = n0 + nh * 128 + wc * 32 + c8;
#pragma unroll
                        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                f32x4 v0 = g.acc[mh * 4 + i][nh * 2 + 0], v1 = g.acc[mh * 4 + i][nh * 2 + 1];
                                v0[0] += b0.x; v0[1] += b0.y; v0[2] += b0.z; v0[3] += b0.w;
                                v1[0] += b1.x; v1[1] += b1.y; v1[2] += b1.z; v1[3] += b1.w;
                                const int row = m0 + mh * 128 + wr * 64 + i * 16 + (lane & 15);
                                auto store8 = [&](bf16_t* base, int ld, const f32x4& x0, const f32x4& x1, bool on) {
                                    const uint2 a = uint2{pack2bf(x0[0], x0[1]), pack2bf(x0[2], x0[3])}, b = uint2{pack2bf(x1[0], x1[1]), pack2bf(x1[2], x1[3])};
                                    const auto rx = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
                                    const auto ry = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
                                    if (on) *reinterpret_cast<uint4*>(base + (size_t)row * ld + col) = uint4{rx[0], ry[0], rx[1], ry[1]};
                                };
                                if constexpr ((FLAGS & EPI_GELU) != 0) {
                                    if (e.out_pre) store8(e.out_pre, e.ldp, v0, v1, row < M && row >= e.pre_row_lo);   // (wave-uniform pointer test; the swap runs in every lane)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) { v0[r] = gelu_f(v0[r]); v1[r] = gelu_f(v1[r]); }
                                }
                                store8(e.out_bf, e.ldbf, v0, v1, row < M);
                            }
                    }
                },
                gemm8p_smem);
}

// quantisation of the persistent grid: 256 workgroups walk ceil(tiles / 256) rounds; the tile pays where the last round is
// reasonably full (tools/micro/gemm8p.hip, profiles/r04_gemm8p_micro.txt: 450 and 600 tiles win, 300 loses to the 128 x 128 tile)
// the operands' alignment the epilogue needs (16-byte accesses)
static inline bool gemm8p_operands_ok(int N, int K, const GemmEpi& e, unsigned flags) {
    if (N % 256 || K % 128 || K < 128 || (reinterpret_cast<size_t>(e.bias) & 15)) return false;
    if (flags & EPI_OUT_F32)
        return !e.resid_mod && e.ldr % 4 == 0 && e.ldo % 4 == 0 && !(reinterpret_cast<size_t>(e.resid) & 15) && !(reinterpret_cast<size_t>(e.out_f32) & 15);
    if (e.ldbf % 8 || (reinterpret_cast<size_t>(e.out_bf) & 15)) return false;
    if ((flags & EPI_GELU) && e.out_pre && (e.ldp % 8 || (reinterpret_cast<size_t>(e.out_pre) & 15))) return false;
    return true;
}
static inline bool gemm8p_shape_ok(int M, int N, int K, int lda, int ldb, const GemmEpi& e, unsigned flags) {
    if (!gemm8p_operands_ok(N, K, e, flags) || M < 256) return false;
    const long tiles = (long)((M + 255) / 256) * (N / 256);
    // N = 768 (proj / fc2 forward): three column tiles only -- from 128 tiles on, one round of the chip (the stand-alone loop beats
    // the 128 x 64 tiles already on 150 of the 256 CUs: 871 against 770 TFLOP/s at 12800 x 768 x 3072)
    // (round 6: proj, K = 768, moves 100 MB of fp32 residual in + out per call at eight pairs and is HBM-bound on 150 workgroups -- but sending it back to
    // the many-workgroup 128 x 64 tile was slower in the step at 8 / 16 / 32 pairs: -0.3 / -0.7 / -1.3 %, profiles/r06_proj_tile_ab.txt)
    if ((flags & EPI_OUT_F32) && tiles >= 128 && tiles <= 256) return true;
    // ... and beyond one round the same fill is as good as it was inside one: 300 tiles (16 pairs) = two rounds at 59 % -- the fill eight pairs run at (150 tiles).
    // Round 6: 467 -> 497 pair-steps/s at 16 pairs per GPU against the 128 x 64 tile the 75 % rule below sent them to (profiles/r06_gemm8p_fill_ab.txt)
    if ((flags & EPI_OUT_F32) && tiles > 256 && tiles * 100 >= ((tiles + 255) / 256) * 256 * 55) return true;
    if (tiles < 200) return false;
    const long rounds = (tiles + 255) / 256;
    return tiles * 100 >= rounds * 256 * 75;
}
template <unsigned FLAGS>
static inline void launch_gemm8p(hipStream_t s, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e) {
    const int tm = (M + 255) / 256, tn = N / 256, tiles = tm * tn;
    int gm = 1;
    while ((gm + 1) * (gm + 1) <= (tiles / 8 + 1) && gm + 1 <= tm) ++gm;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<unsigned long long> done_mask{0};
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, Gemm8p::LDS_BYTES);
        done_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    SPLICE_LAUNCH((gemm8p_kernel<FLAGS>), dim3(tiles < 256 ? tiles : 256), dim3(512), Gemm8p::LDS_BYTES, s, A, lda, B, ldb, M, N, K, gm, e);
}
