// Instantiations + runtime dispatch of the bf16 NT GEMM (gemm.h) for the epilogues the
// DINO-ViT forward / dgrad path uses (K3, K5, K7, K8 of SURVEY.md section 2b).
#include "kernels.h"

static int g_force_tile = 0;   // 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64 (tools/gemm_bench.py)
void gemm_force_tile(int t) { g_force_tile = t; }

template <unsigned FLAGS>
static int dispatch_tile(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const GemmEpi& e,
                         hipStream_t s) {
    // 256 CUs: prefer the biggest tile that still yields >= ~1 workgroup per CU.
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    const long t12864 = (long)cdiv(M, 128) * cdiv(N, 64);
    if (g_force_tile == 1) { launch_gemm_nt<128, 128, FLAGS>(s, A, lda, B, ldb, M, N, K, e); return SPLICE_OK; }
    if (g_force_tile == 2) { launch_gemm_nt<128, 64, FLAGS>(s, A, lda, B, ldb, M, N, K, e); return SPLICE_OK; }
    if (g_force_tile == 3) { launch_gemm_nt<64, 64, FLAGS>(s, A, lda, B, ldb, M, N, K, e); return SPLICE_OK; }
    // measured on MI355X (tools/gemm_bench.py): with N <= 768 the 64x64 tile wins (more workgroups on the long-K shapes)
    if (N <= 768) { launch_gemm_nt<64, 64, FLAGS>(s, A, lda, B, ldb, M, N, K, e); return SPLICE_OK; }
    if (t128 >= 224) launch_gemm_nt<128, 128, FLAGS>(s, A, lda, B, ldb, M, N, K, e);
    else if (t12864 >= 200) launch_gemm_nt<128, 64, FLAGS>(s, A, lda, B, ldb, M, N, K, e);
    else launch_gemm_nt<64, 64, FLAGS>(s, A, lda, B, ldb, M, N, K, e);
    return SPLICE_OK;
}

int gemm_nt_launch(unsigned flags, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K,
                   const GemmEpi& e, hipStream_t s) {
    if (M < 1 || N < 1 || K < GEMM_BK || K % GEMM_BK || lda % 8 || ldb % 8) return SPLICE_ERR_ARG;
    if ((flags & EPI_OUT_T) && (e.ldt % 4)) return SPLICE_ERR_ARG;
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
        CASE(EPI_GELU_GRAD | EPI_OUT_BF);                         // fc2 dgrad
        default: return SPLICE_ERR_ARG;
    }
#undef CASE
}
