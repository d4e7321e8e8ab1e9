// Instantiations + runtime dispatch of the bf16 NT GEMM (gemm.h) for the epilogues the
// DINO-ViT forward / dgrad path uses (K3, K5, K7, K8 of SURVEY.md section 2b).
#include "kernels.h"
#include "gemm8p.h"

static int g_force_tile = 0;   // 0 auto; tile + 10 * ring: tile 1 = 128x128, 2 = 128x64, 3 = 64x64; ring 0 = 2 stages, 1 = 4 stages, 2 = 3 stages (64x64 only) (tools/gemm_bench.py); 5 = the persistent 256x256 8-phase tile where it exists (gemm8p.h), else automatic
void gemm_force_tile(int t) { g_force_tile = t; }

template <unsigned FLAGS>
static int dispatch_tile(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e,
                         hipStream_t s) {
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    const long t12864 = (long)cdiv(M, 128) * cdiv(N, 64);
    // the batched forward projections with whole 256-column tiles (QKV of the layers without side outputs, fc1) on the persistent
    // 256 x 256 8-phase tile where its grid quantisation is good (gemm8p.h; same bits as every other tile)
    if constexpr (FLAGS == (EPI_BIAS | EPI_OUT_BF) || FLAGS == (EPI_BIAS | EPI_GELU | EPI_OUT_BF) || FLAGS == (EPI_BIAS | EPI_RESID | EPI_OUT_F32)) {
        constexpr int on8p = 7;   // bit 0: QKV, bit 1: fc1, bit 2: proj / fc2 forward
        const int bit = (FLAGS & EPI_OUT_F32) ? 4 : (FLAGS & EPI_GELU) ? 2 : 1;
        const bool want = g_force_tile == 5 ? gemm8p_operands_ok(N, K, e, FLAGS) : (!g_force_tile && (on8p & bit) && gemm8p_shape_ok(M, N, K, lda, ldb, e, FLAGS));
        if (want) {
            launch_gemm8p<FLAGS>(s, A, lda, B, ldb, M, N, K, e);
            return SPLICE_OK;
        }
    }
    int tile, ring;
    if (g_force_tile && g_force_tile != 5 && !(FLAGS & EPI_ROWDOT)) { tile = g_force_tile % 10; ring = g_force_tile / 10; }
    else {
        // measured on MI355X (tools/gemm_bench.py): with N <= 768 the 64x64 tile wins (more workgroups on the long-K shapes);
        // otherwise the biggest tile that still yields ~2 workgroups per CU (M = 1600: 128x64 beats 128x128 by 10 %, M = 800: 64x64)
        constexpr int t2min = 300;   // (400 -> 300, end of round 3: -0.4 % step time at one pair, equal at two / eight; profiles/r03_gemm_t1min_sweep.txt)
        constexpr int t2ring = 3;
        // (t1min 420 -> 230 at the end of round 3: with the QKV epilogue down to one output the one-pair forward GEMMs of QKV / fc1
        // (234 / 312 tiles of 128 x 128 at 1600 rows) beat their 128 x 64 ring form: -1.9 % step time at one pair, -0.6 % at two,
        // unchanged at eight; profiles/r03_gemm_t1min_sweep.txt)
#ifndef SPLICE_T1MIN   // (compile-time defaults; variant builds for tools/ab_libs.sh override them with -D)
#define SPLICE_T1MIN 230
#endif
#ifndef SPLICE_T96
#define SPLICE_T96 6
#endif
        constexpr int t1min = SPLICE_T1MIN;
        // rows of a P-pair batch (tools/gemm_sweep.py, r2): with N <= 768 the 64x64 tile stops winning at ~4800 rows (fc2 at
        // M = 6400: 688 TF on 128x64 vs 554; M = 12800: 872 vs 656); N >= 2304 takes 128x128 from 3200 rows on (t1min 640 -> 420)
        // (in-step at 2 / 3 / 4 pairs per GPU -- 3200 ... 6400 rows -- the 128x64 tile already wins from the first row count without
        // a ring on: +0.9 / +0.3 / +0.8 % pair-steps/s against the stand-alone sweep's 4800, tools/gemm_thresh_sweep.sh)
        constexpr int bigm = 2401;
        tile = (FLAGS & EPI_ROWDOT) ? 3 : N <= 768 ? (M >= bigm ? 2 : 3) : t128 >= t1min ? 1 : t12864 >= t2min ? 2 : 3;
        // few workgroups walking a long K (fc2, the fc1 / qkv dgrads): the per-slice DMA latency is exposed with 2 stages,
        // the 4-stage ring keeps 3 slices in flight (fc1T 800x768x3072: 23.4 -> 15.3 us); elsewhere its LDS footprint costs occupancy
        const int ks = (FLAGS == EPI_OUT_F32 && e.ksplit > 1 && K % (e.ksplit * GEMM_BK) == 0) ? e.ksplit : 1;
        constexpr int ringk = 768;
        constexpr int ringwg = 640;
        // (the rings pay while a launch is a single wave of latency-bound workgroups: M <= 2400 rows; beyond that the plain
        // 2-stage pipeline is faster -- fc2 at M = 3200: 557 vs 460 TF, 128x64 at M = 6400: 688 vs 626)
        ring = (tile == 3 && M <= 2400 && K / ks >= (ks > 1 ? 768 : ringk) && (long)cdiv(M, 64) * cdiv(N, 64) * ks <= ringwg) ? 1 : 0;
        if (tile == 2 && t2ring && M <= 2400) ring = t2ring == 3 ? 2 : 1;
        // the short-K ring shapes (proj, projT: 12 slices) run the 3-stage form -- its own instantiation, so profiles keep
        // them apart from the long-K launches of the same epilogue (fc2)
        constexpr int shortns = 3;
        if (ring && tile == 3 && ks == 1 && K < 1536 && shortns == 3) ring = 2;
    }
    // (round 6: hipBLASLt runs fc1 at 1600 rows on 128 x 160 tiles -- 260 tiles, one round of the chip -- and is 10 % faster there than our 128 x 128
    // (312 tiles).  The same shape on THIS tile engine (wave tile 64 x 80, five fragment columns) measured 17.1 us against 14.6 stand-alone and +0.5 % step
    // time in the alternating in-step A/B: not kept.  profiles/r06_gemm_t160.txt)
    if constexpr ((FLAGS & EPI_ROWDOT) != 0) {   // only instantiated for the 64-column tile
        if (ring == 2) launch_gemm_nt<64, 64, FLAGS, 3>(s, A, lda, B, ldb, M, N, K, e);
        else if (ring) launch_gemm_nt<64, 64, FLAGS, 4>(s, A, lda, B, ldb, M, N, K, e);
        else launch_gemm_nt<64, 64, FLAGS, 2>(s, A, lda, B, ldb, M, N, K, e);
    } else {
        if constexpr (FLAGS == (EPI_BIAS | EPI_RESID | EPI_OUT_F32)) {   // proj forward at M = 2 x 800 (K = 768): one wave of 64x96 tiles (200 workgroups) instead of 300 64x64 ones, in-step -0.4 %;
            // the same tile for fc2 (K = 3072) measured equal or worse (t96: 0 off, 3 both, 5 long-K only, 6 short-K only)
            constexpr int t96 = SPLICE_T96;
            const long wg64 = (long)cdiv(M, 64) * cdiv(N, 64), wg96 = (long)cdiv(M, 64) * (N / 96);
            if (t96 && !(t96 == 5 && K < 1536) && !(t96 == 6 && K >= 1536) && tile == 3 && N % 96 == 0 && wg64 > 256 && wg96 <= 256) {
                if (K >= 1536 && t96 != 3) launch_gemm_nt<64, 96, FLAGS, 4>(s, A, lda, B, ldb, M, N, K, e);
                else launch_gemm_nt<64, 96, FLAGS, 3>(s, A, lda, B, ldb, M, N, K, e);
                return SPLICE_OK;
            }
        }
        if (tile == 3 && ring == 2) { launch_gemm_nt<64, 64, FLAGS, 3>(s, A, lda, B, ldb, M, N, K, e); return SPLICE_OK; }
        if (tile == 2 && ring == 2) { launch_gemm_nt<128, 64, FLAGS, 3>(s, A, lda, B, ldb, M, N, K, e); return SPLICE_OK; }
        switch (tile * 2 + (ring ? 1 : 0)) {
            case 2: launch_gemm_nt<128, 128, FLAGS, 2>(s, A, lda, B, ldb, M, N, K, e); break;
            case 3: launch_gemm_nt<128, 128, FLAGS, 4>(s, A, lda, B, ldb, M, N, K, e); break;
            case 4: launch_gemm_nt<128, 64, FLAGS, 2>(s, A, lda, B, ldb, M, N, K, e); break;
            case 5: launch_gemm_nt<128, 64, FLAGS, 4>(s, A, lda, B, ldb, M, N, K, e); break;
            case 6: launch_gemm_nt<64, 64, FLAGS, 2>(s, A, lda, B, ldb, M, N, K, e); break;
            default: launch_gemm_nt<64, 64, FLAGS, 4>(s, A, lda, B, ldb, M, N, K, e); break;
        }
    }
    return SPLICE_OK;
}

int gemm_nt_launch(unsigned flags, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K,
                   const GemmEpi& e, hipStream_t s) {
    if (M < 1 || N < 1 || K < GEMM_BK || K % GEMM_BK || lda % 8 || ldb % 8) return SPLICE_ERR_ARG;
    // the LDS-DMA uses 32-bit byte offsets from the operand bases
    if ((size_t)M * lda * sizeof(bf16_t) >= (1ull << 32) || (size_t)N * ldb * sizeof(bf16_t) >= (1ull << 32)) return SPLICE_ERR_ARG;
    if ((flags & EPI_OUT_T) && (e.ldt % 4)) return SPLICE_ERR_ARG;
    if ((flags & EPI_ROWDOT) && (N % 64 || !e.rd_other || !e.rowdot || e.rd_rows < 1 || e.ld_rd % 4)) return SPLICE_ERR_ARG;
#define CASE(F) case (F): return dispatch_tile<(F)>(A, lda, B, ldb, M, N, K, e, s)
    switch (flags) {
        CASE(EPI_BIAS | EPI_OUT_BF | EPI_OUT_T);                  // qkv
        CASE(EPI_BIAS | EPI_OUT_BF | EPI_OUT_T | EPI_COLS_F32);   // qkv, layer 11 (fp32 keys too)
        CASE(EPI_BIAS | EPI_OUT_F32);                             // keys-only projection
        CASE(EPI_BIAS | EPI_OUT_BF);
        CASE(EPI_BIAS | EPI_RESID | EPI_OUT_F32);                 // proj / fc2 / patch-embed
        CASE(EPI_BIAS | EPI_GELU | EPI_OUT_BF);                   // fc1
        CASE(EPI_OUT_F32);                                        // dgrad -> LN backward
        CASE(EPI_OUT_F32 | EPI_ALPHA);
        CASE(EPI_OUT_BF);                                         // dgrad -> next GEMM
        CASE(EPI_OUT_BF | EPI_OUT_T);                             // proj dgrad -> attention backward
        CASE(EPI_OUT_BF | EPI_OUT_T | EPI_ROWDOT);                // same + delta = rowsum(dO * O) per head
        CASE(EPI_OUT_BF | EPI_ROWDOT);                            // proj dgrad of the ViT engine: dO and delta (the attention backward transposes in LDS)
        CASE(EPI_GELU_GRAD | EPI_OUT_BF);                         // fc2 dgrad
        default: return SPLICE_ERR_ARG;
    }
#undef CASE
}

// ---- e4m3 operands (BASELINE configs[4]): the QKV projection of the fp8 ViT mode.  Same tile engine, loaders fed with the
// byte geometry in 2-byte units; tile choice as for bf16 (the LDS footprint per slice is the same, the work per slice doubles).
template <unsigned FLAGS>
static int dispatch_fp8(const uint8_t* A, int lda, const uint8_t* B, int ldb, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
    const bf16_t* a = reinterpret_cast<const bf16_t*>(A);
    const bf16_t* b = reinterpret_cast<const bf16_t*>(B);
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t12864 = (long)cdiv(M, 128) * cdiv(N, 64);
    constexpr int t1min8 = 230;   // (as the bf16 dispatcher: -1.3 % step time at one pair; profiles/r03_gemm_t1min_sweep.txt)
    constexpr int t2min8 = 300;
    if (N <= 768) {   // fc2: as the bf16 dispatcher -- 64x64 tiles on the deep ring for the few-row shapes, 128x64 from the batched row counts on
        if (M >= 2401) launch_gemm_nt<128, 64, FLAGS, 2, true>(s, a, lda / 2, b, ldb / 2, M, N, K / 2, e);
        else launch_gemm_nt<64, 64, FLAGS, 4, true>(s, a, lda / 2, b, ldb / 2, M, N, K / 2, e);
    } else if (t128 >= t1min8) launch_gemm_nt<128, 128, FLAGS, 2, true>(s, a, lda / 2, b, ldb / 2, M, N, K / 2, e);
    else if (t12864 >= t2min8) launch_gemm_nt<128, 64, FLAGS, 3, true>(s, a, lda / 2, b, ldb / 2, M, N, K / 2, e);
    else launch_gemm_nt<64, 64, FLAGS, 3, true>(s, a, lda / 2, b, ldb / 2, M, N, K / 2, e);
    return SPLICE_OK;
}
int gemm_nt_fp8_launch(unsigned flags, const uint8_t* A, int lda, const uint8_t* B, int ldb, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
    if (M < 1 || N < 1 || K < 128 || K % 128 || lda % 16 || ldb % 16) return SPLICE_ERR_ARG;
    if ((size_t)M * lda >= (1ull << 32) || (size_t)N * ldb >= (1ull << 32)) return SPLICE_ERR_ARG;
    if (!(flags & EPI_SCALE_RC) || !e.col_scale) return SPLICE_ERR_ARG;   // (row_scale may be NULL: unscaled e4m3 activations)
    if ((flags & EPI_OUT_T) && (e.ldt % 4)) return SPLICE_ERR_ARG;
    if ((flags & EPI_OUT_F8) && (!e.out_f8 || N % 16 || e.ld8 % 16 || (e.out_pre && e.ldp % 8))) return SPLICE_ERR_ARG;
    if ((flags & EPI_OUT_F8T) && (!e.out_f8_t || e.ldt8 % 16)) return SPLICE_ERR_ARG;
    // e4m3 copies beside the bf16 outputs leave through the staged-tile path only: its alignment conditions must hold
    if ((flags & (EPI_OUT_F8 | EPI_OUT_F8T)) && (flags & EPI_OUT_BF) && (N % 16 || e.ldbf % 8 || ((flags & EPI_OUT_T) && e.ldt % 8))) return SPLICE_ERR_ARG;
#define CASE8(F) case (F): return dispatch_fp8<(F)>(A, lda, B, ldb, M, N, K, e, s)
    switch (flags) {
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_OUT_BF);                              // qkv
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_OUT_BF | EPI_OUT_F8 | EPI_OUT_F8T);   // qkv + e4m3 copies for the fp8 attention
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_OUT_BF | EPI_OUT_T);                  // qkv with a transposed bf16 copy
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_OUT_BF | EPI_OUT_T | EPI_COLS_F32);   // qkv, last layer
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_OUT_BF | EPI_OUT_T | EPI_OUT_F8 | EPI_OUT_F8T);                  // qkv + e4m3 copies for the fp8 attention
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_OUT_BF | EPI_OUT_T | EPI_COLS_F32 | EPI_OUT_F8 | EPI_OUT_F8T);   // the same, last layer
        CASE8(EPI_SCALE_RC | EPI_OUT_F32);
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_GELU | EPI_OUT_F8);                   // fc1: e4m3 GELU output (+ bf16 pre-activation)
        CASE8(EPI_SCALE_RC | EPI_BIAS | EPI_RESID | EPI_OUT_F32);                 // fc2
        default: return SPLICE_ERR_ARG;
    }
#undef CASE8
}
