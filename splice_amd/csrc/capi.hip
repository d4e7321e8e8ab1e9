// extern "C" op-level entry points of include/splice_hip.h (thin argument checks + launch).
#include <stdarg.h>
#include <stdio.h>

#include "kernels.h"

static thread_local char g_err[512] = "";

void splice_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int finish(int rc, const char* what) {
    if (rc != SPLICE_OK) {
        splice_set_error("%s: invalid argument", what);
        return rc;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        splice_set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
        return SPLICE_ERR_HIP;
    }
    return SPLICE_OK;
}

#define ST(s) ((hipStream_t)(s))

extern "C" {

int splice_version(void) { return 100; }
int splice_dev_switches(void) {
#ifdef SPLICE_DEV_SWITCHES
    return 1;
#else
    return 0;
#endif
}
const char* splice_last_error(void) { return g_err; }

int splice_gemm_nt_bf16(unsigned flags, const splice_bf16* A, int lda, const splice_bf16* B, int ldb, int M, int N, int K,
                        const splice_gemm_epilogue* epi, splice_stream_t stream) {
    if (!A || !B || !epi) return finish(SPLICE_ERR_ARG, "splice_gemm_nt_bf16");
    return finish(gemm_nt_launch(flags, A, lda, B, ldb, M, N, K, *epi, ST(stream)), "splice_gemm_nt_bf16");
}

int splice_gemm_nt_fp8(unsigned flags, const uint8_t* A, int lda, const uint8_t* B, int ldb, int M, int N, int K,
                       const splice_gemm_epilogue* epi, splice_stream_t stream) {
    if (!A || !B || !epi) return finish(SPLICE_ERR_ARG, "splice_gemm_nt_fp8");
    return finish(gemm_nt_fp8_launch(flags, A, lda, B, ldb, M, N, K, *epi, ST(stream)), "splice_gemm_nt_fp8");
}
int splice_quantize_rows_fp8(const float* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int cols, splice_stream_t stream) {
    if (!x || !q || !scale) return finish(SPLICE_ERR_ARG, "splice_quantize_rows_fp8");
    return finish(quantize_rows_fp8_launch(x, ldx, q, ldq, scale, rows, cols, ST(stream)), "splice_quantize_rows_fp8");
}

int splice_gemm_splitk_slabs(int M, int ksplit) { return gemm_splitk_slabs(M, ksplit); }

/* benchmarking hook: force the GEMM tile (0 auto, 1 128x128, 2 128x64, 3 64x64) */
int splice_gemm_force_tile(int tile) { gemm_force_tile(tile); return SPLICE_OK; }
int splice_attention_variant(int variant) { attn_set_variant(variant); return SPLICE_OK; }
int splice_attention_qfold(int on) { attn_set_qfold(on); return SPLICE_OK; }
int splice_attention_bwd_variant(int variant) { attn_set_bwd_variant(variant); return SPLICE_OK; }

int splice_layernorm_fwd(const float* x, const float* gamma, const float* beta, splice_bf16* y, float* mean, float* rstd,
                         int rows, int D, float eps, splice_stream_t stream) {
    return finish(layernorm_fwd_launch(x, gamma, beta, y, mean, rstd, rows, D, eps, ST(stream)), "splice_layernorm_fwd");
}
int splice_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* g_in, float* g_out, splice_bf16* g_out_bf, int rows, int D, splice_stream_t stream) {
    return finish(layernorm_bwd_launch(dy, x, gamma, mean, rstd, g_in, g_out, g_out_bf, rows, D, ST(stream)), "splice_layernorm_bwd");
}

int splice_attention_fwd(const splice_bf16* qkv, const splice_bf16* qkvT, int ldt, int B, int T, int Tld, int D, int H,
                         float scale, splice_bf16* out, float* lse, splice_stream_t stream) {
    AttnArgs a = {};
    a.qkv = qkv; a.qkvT = qkvT; a.ldt = ldt; a.B = B; a.T = T; a.Tld = Tld; a.D = D; a.H = H; a.scale = scale;
    a.out = out; a.lse = lse; a.qfold = attn_qfold_hook();
    return finish(attn_fwd_launch(&a, ST(stream)), "splice_attention_fwd");
}
int splice_attention_fwd_fp8(const uint8_t* qkv8, const uint8_t* qkvT8, int ldt8, int B, int T, int Tld, int D, int H, float scale,
                             splice_bf16* out, float* lse, splice_stream_t stream) {
    if (!qkv8 || !qkvT8 || !out || !lse) return finish(SPLICE_ERR_ARG, "splice_attention_fwd_fp8");
    AttnArgs a = {};
    a.qkv8 = qkv8; a.qkvT8 = qkvT8; a.ldt8 = ldt8; a.ldt = 4; a.B = B; a.T = T; a.Tld = Tld; a.D = D; a.H = H; a.scale = scale;
    a.out = out; a.lse = lse;
    return finish(attn_fwd_launch(&a, ST(stream)), "splice_attention_fwd_fp8");
}
int splice_attention_bwd(const splice_bf16* qkv, const splice_bf16* qkvT, int ldt, int B, int T, int Tld, int D, int H,
                         float scale, const splice_bf16* out, const float* lse, const splice_bf16* dout,
                         const splice_bf16* doutT, float* delta, splice_bf16* dqkv, splice_stream_t stream) {
    AttnArgs a = {};
    a.qkv = qkv; a.qkvT = qkvT; a.ldt = ldt; a.B = B; a.T = T; a.Tld = Tld; a.D = D; a.H = H; a.scale = scale;
    a.out = const_cast<splice_bf16*>(out); a.lse = const_cast<float*>(lse);
    a.dout = dout; a.doutT = doutT; a.delta = delta; a.dqkv = dqkv; a.qfold = attn_qfold_hook();
    return finish(attn_bwd_launch(&a, ST(stream)), "splice_attention_bwd");
}
int splice_attention_probs(const splice_bf16* qkv, int B, int T, int Tld, int D, int H, float scale, const float* lse,
                           float* probs, splice_stream_t stream) {
    AttnArgs a = {};
    a.qkv = qkv; a.B = B; a.T = T; a.Tld = Tld; a.D = D; a.H = H; a.scale = scale; a.lse = const_cast<float*>(lse);
    return finish(attn_probs_launch(&a, probs, ST(stream)), "splice_attention_probs");
}

int splice_augment_structure(const float* img, float* out, float* scratch, int H, int W, int flip, int n_ops, const int* order,
                             const float* factors, float blur_sigma, splice_stream_t stream) {
    return finish(augment_structure_launch(img, out, scratch, H, W, flip, n_ops, order, factors, blur_sigma, ST(stream)), "splice_augment_structure");
}

size_t splice_keys_selfsim_ws_bytes(int T, int D) { return selfsim_ws_bytes(T, D); }
int splice_keys_selfsim_fwd(const float* K, int ldk, int T, int D, float eps, float* S, void* ws, splice_stream_t stream) {
    SelfSimWs w;
    selfsim_ws_carve(ws, T, D, &w);
    return finish(selfsim_fwd_launch(K, ldk, T, D, eps, S, w, ST(stream)), "splice_keys_selfsim_fwd");
}
int splice_keys_selfsim_bwd(const float* dS, const float* S, int T, int D, float eps, float* dK, int lddk, int accumulate,
                            void* ws, splice_stream_t stream) {
    SelfSimWs w;
    selfsim_ws_carve(ws, T, D, &w);
    return finish(selfsim_bwd_launch(dS, S, T, D, eps, dK, lddk, accumulate, w, ST(stream)), "splice_keys_selfsim_bwd");
}
int splice_mse(const float* a, int lda, const float* b, int ldb, int rows, int cols, float weight, float* loss_accum,
               float* grad, int ldg, splice_stream_t stream) {
    return finish(mse_launch(a, lda, b, ldb, rows, cols, weight, loss_accum, grad, ldg, ST(stream)), "splice_mse");
}

int splice_patchify(const float* img, splice_bf16* patches, int B, int H, int W, int p, int Tld, int normalize,
                    splice_stream_t stream) {
    return finish(patchify_launch(img, patches, B, H, W, p, Tld, normalize, ST(stream)), "splice_patchify");
}
int splice_unpatchify(const float* dpatches, float* dimg, int B, int H, int W, int p, int Tld, int normalize,
                      splice_stream_t stream) {
    return finish(unpatchify_launch(dpatches, dimg, B, H, W, p, Tld, normalize, ST(stream)), "splice_unpatchify");
}
int splice_cast_f32_bf16(const float* x, splice_bf16* y, size_t n, splice_stream_t stream) {
    return finish(cast_f32_bf16_launch(x, y, n, ST(stream)), "splice_cast_f32_bf16");
}
int splice_cast_bf16_f32(const splice_bf16* x, float* y, size_t n, splice_stream_t stream) {
    return finish(cast_bf16_f32_launch(x, y, n, ST(stream)), "splice_cast_bf16_f32");
}
int splice_resize_bilinear_fwd(const float* in, float* out, int planes, int h, int w, int oh, int ow, splice_stream_t stream) {
    return finish(resize_bilinear_fwd_launch(in, out, planes, h, w, oh, ow, ST(stream)), "splice_resize_bilinear_fwd");
}
int splice_resize_bilinear_bwd(const float* dout, float* din, int planes, int h, int w, int oh, int ow, splice_stream_t stream) {
    return finish(resize_bilinear_bwd_launch(dout, din, planes, h, w, oh, ow, ST(stream)), "splice_resize_bilinear_bwd");
}
int splice_transpose_f32_bf16(const float* x, splice_bf16* y, int rows, int cols, int ldy, splice_stream_t stream) {
    return finish(transpose_f32_to_bf16_launch(x, y, rows, cols, ldy, ST(stream)), "splice_transpose_f32_bf16");
}
}
