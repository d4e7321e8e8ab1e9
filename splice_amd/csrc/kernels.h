// Internal launcher declarations (one per hand-written gfx950 kernel family).
// The public C ABI is include/splice_hip.h, implemented in capi.hip / engine.hip.
#pragma once
#include "common.h"
#include "gemm.h"

// ---- attention.hip -------------------------------------------------------------------
struct AttnArgs {
    const bf16_t* qkv;     // [B*Tld][3D]
    const bf16_t* qkvT;    // [3D][ldt]
    int ldt;
    int B, T, Tld, D, H;
    float scale;
    bf16_t* out;           // [B*Tld][D]
    float* lse;            // [B][H][Tld]
    // backward only
    const bf16_t* dout;    // [B*Tld][D]
    const bf16_t* doutT;   // [D][ldt]
    float* delta;          // [B][H][Tld]  rowsum(dO * O)
    int delta_ready;       // != 0: the caller already filled delta (the ViT engine: proj-dgrad GEMM epilogue)
    bf16_t* dqkv;          // [B*Tld][3D]
    // e4m3 forward (BASELINE configs[4]; null = bf16 forward): the same qkv as unscaled e4m3 bytes, row-major and transposed
    const uint8_t* qkv8;   // [B*Tld][3D]
    const uint8_t* qkvT8;  // [3D][ldt8]
    int ldt8;
    int qfold;             // != 0: the q columns of qkv hold q * scale * log2(e) (the ViT engine's packing of the QKV projection); scores are exponents as they leave the MFMA
};
int attn_fwd_launch(const AttnArgs* a, hipStream_t s);
int attn_bwd_launch(const AttnArgs* a, hipStream_t s);
int attn_probs_launch(const AttnArgs* a, float* probs, hipStream_t s);
void attn_set_variant(int v);   // benchmarking hook
void attn_set_bwd_variant(int v);
int attn_qfold_hook();
void attn_set_qfold(int on);    // benchmarking hook: the stand-alone entry points take pre-scaled q

// ---- vit_cls.hip: the tail of the top block on the [CLS] rows only ---------------------------
int attn_cls_fwd_launch(const bf16_t* qkv, const bf16_t* qkvT, int ldt, int B, int T, int Tld, int D, int H, float scale, bf16_t* out, float* probs,
                        hipStream_t s);
int attn_cls_bwd_launch(const bf16_t* qkv, const bf16_t* qkvT, int ldt, int B, int T, int Tld, int D, int H, float scale, const float* probs,
                        const float* dout_slabs, int n_slabs, size_t slab_stride, bf16_t* dqkv, hipStream_t s);
int ln_rows_fwd_launch(float* x, size_t xs, const float* gamma, const float* beta, bf16_t* y, size_t ys, float* mean, float* rstd, size_t ss, int rows,
                       int D, float eps, const float* slabs, int n_slabs, size_t slab_stride, const float* bias, const float* resid, size_t rs, hipStream_t s);
int ln_rows_bwd_launch(float* dy, size_t dys, const float* x, size_t xs, const float* gamma, const float* mean, const float* rstd, size_t ss, float* g,
                       bf16_t* g_bf, int rows, int D, int n_slabs, size_t slab_stride, hipStream_t s);
int rows_finish_launch(int mode, const float* slabs, int n_slabs, size_t slab_stride, int rows, int N, const float* bias, const float* resid, size_t rs,
                       float* out_f32, size_t os, bf16_t* out_bf, bf16_t* pre_bf, const bf16_t* aux, size_t ps, int pre_lo, hipStream_t s);

// ---- gemm.hip --------------------------------------------------------------------------
// Runtime dispatch over the instantiated (tile, epilogue-flag) combinations.
int gemm_nt_launch(unsigned flags, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K,
                   const GemmEpi& e, hipStream_t s);

void gemm_force_tile(int t);   // benchmarking hook: 0 auto, 1 128x128, 2 128x64, 3 64x64
// e4m3 operands (bytes; lda / ldb / K in elements = bytes): flags must include EPI_SCALE_RC
int gemm_nt_fp8_launch(unsigned flags, const uint8_t* A, int lda, const uint8_t* B, int ldb, int M, int N, int K, const GemmEpi& e, hipStream_t s);

// ---- vit_ops.hip -----------------------------------------------------------------------
// LayerNorm over the last dim (D % 256 == 0 handled generally), one wave per row.
int layernorm_fwd_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                         int rows, int D, float eps, hipStream_t s);
// fp8 (e4m3) operand path: LayerNorm output quantised per token (y bytes [rows][D], yscale[rows]: y_true ~= y * yscale[row])
int layernorm_fwd_fp8_launch(const float* x, const float* gamma, const float* beta, uint8_t* y, float* yscale, float* mean, float* rstd,
                             int rows, int D, float eps, hipStream_t s);
int quantize_rows_fp8_launch(const float* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int cols, hipStream_t s);
int quantize_rows_bf16_fp8_launch(const bf16_t* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int cols, hipStream_t s);
int quantize_keys_fp8_launch(const bf16_t* k, int ldk, size_t k_pstride, uint8_t* q, int ldq, size_t q_pstride, float* qnorm, int T, int Tp, int D,
                             int pairs, hipStream_t s);
// g_out = g_in + LN_backward(dy); also emits bf16(g_out).  dy is fp32 [rows][D].
int layernorm_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* g_in, float* g_out, bf16_t* g_out_bf, int rows, int D, hipStream_t s);
// same with dy given as n_slabs split-K slabs (dy + s * slab_stride), summed in order
int layernorm_bwd_slabs_launch(const float* dy, int n_slabs, size_t slab_stride, const float* x, const float* gamma, const float* mean,
                               const float* rstd, const float* g_in, float* g_out, bf16_t* g_out_bf, int rows, int D, hipStream_t s);
// image [B][3][H][W] fp32 -> patch matrix bf16 [B*Tld][3*p*p] (row b*Tld+1+patch; row 0 and
// rows >= T are zero), k index = c*p*p + py*p + px (Conv2d weight order).  Optional
// ImageNet normalisation (x-mean)/std fused in.
int patchify_launch(const float* img, bf16_t* patches, int B, int H, int W, int p, int Tld, int normalize, hipStream_t s);
// d(patch matrix) fp32 [B*Tld][3*p*p] -> d(image) [B][3][H][W] (divides by std when normalize).
int unpatchify_launch(const float* dpatches, float* dimg, int B, int H, int W, int p, int Tld, int normalize, hipStream_t s);
int cast_f32_bf16_launch(const float* x, bf16_t* y, size_t n, hipStream_t s);
int cast_bf16_f32_launch(const bf16_t* x, float* y, size_t n, hipStream_t s);
int transpose_f32_to_bf16_launch(const float* x, bf16_t* y, int rows, int cols, int ldy, hipStream_t s);  // y[c][r]
int fill_f32_launch(float* x, float v, size_t n, hipStream_t s);
// kernel-based copy / zero (graph-capture safe replacements of hipMemcpyAsync / hipMemsetAsync, bytes % 4 == 0)
int dev_copy_launch(void* dst, const void* src, size_t bytes, hipStream_t s);
int dev_zero_launch(void* dst, size_t bytes, hipStream_t s);
// torchvision-0.10 tensor Resize = F.interpolate(bilinear, align_corners=False, no antialias) on [N][C][h][w]
// (util/losses.py:20,77-78); backward is the exact adjoint in gather form (deterministic).
int resize_bilinear_fwd_launch(const float* in, float* out, int planes, int h, int w, int oh, int ow, hipStream_t s);
int resize_bilinear_bwd_launch(const float* dout, float* din, int planes, int h, int w, int oh, int ow, hipStream_t s);
int add_f32_launch(float* y, const float* x, size_t n, hipStream_t s);  // y += x

// ---- augment.hip ----------------------------------------------------------------------
int augment_structure_launch(const float* img, float* out, float* scratch, int H, int W, int flip, int n_ops, const int* order,
                             const float* factors, float blur_sigma, hipStream_t s);

// ---- selfsim.hip -----------------------------------------------------------------------
// Cosine self-similarity of the rows of K (fp32 [T][ldk], D columns), models/extractor.py:4-9.
struct SelfSimWs {          // carved from caller-provided scratch (selfsim_ws_bytes)
    bf16_t* kbf;            // [Tp][D]   bf16 keys (rows >= T zero), Tp = roundup(T, 64)
    bf16_t* kbfT;           // [D][Tp]
    float* norm;            // [Tp]      L2 norms of the bf16-rounded rows
    bf16_t* wmat;           // [Tp][Tp]  backward weights  (dS + dS^T) / max(n_i n_j, eps)
    float* rowdot;          // [Tp]      sum_j (dS+dS^T)_ij S_ij / n_i^2
    int Tp;
};
size_t selfsim_ws_bytes(int T, int D);
void selfsim_ws_carve(void* base, int T, int D, SelfSimWs* ws);
// S = cos-sim(K) -> fp32 [T][T]; fills ws.kbf / kbfT / norm (needed by the backward).
int selfsim_fwd_launch(const float* K, int ldk, int T, int D, float eps, float* S, const SelfSimWs& ws, hipStream_t s);
// dK fp32 [T][lddk] (+= if accumulate) from dS fp32 [T][T] (any, not nec. symmetric); needs the
// ws state left by selfsim_fwd_launch on the same K, and S.
int selfsim_bwd_launch(const float* dS, const float* S, int T, int D, float eps, float* dK, int lddk, int accumulate,
                       const SelfSimWs& ws, hipStream_t s);
// ---- fused + batched structure loss of the optimisation step (blockIdx.y = pair); see selfsim.hip
struct SelfSimBatch {
    int T, Tp, D, pairs;
    int ldk, ldt;                   // leading dimensions of the key matrices below
    size_t k_pstride, kT_pstride;   // element strides between the pairs' problems
    const bf16_t* k_tgt;            // target keys (A' passes)     bf16 [T][ldk], pair p at + p * k_pstride
    const bf16_t* k_x;              // generated keys (x' passes)
    const bf16_t* kT_x;             // transposed generated keys   bf16 [D][ldt], pair p at + p * kT_pstride
    float* S_tgt;                   // [pairs][T][T] fp32 (upper-triangular 64x64 tiles valid)
    bf16_t* wmat;                   // [pairs][Tp][Tp]
    float *norm_tgt, *norm_x;       // [pairs][Tp]
    float* rpart;                   // [pairs][Tp/64][Tp] per-tile partial row dots
    float* loss_part;               // pair p's per-tile loss partials at + p * part_pstride (>= nt(nt+1)/2 floats, summed by the caller)
    size_t part_pstride;
    float* dk;                      // d keys fp32: pair p at + p * dk_pstride, [T][lddk]
    size_t dk_pstride; int lddk;
    float eps, e_scale, loss_scale; // e_scale = 4 lambda / T^2 ; loss_scale = 1 / T^2
    int fp8;                        // != 0: the two Gram matrices run on the fp8 MFMA from per-row quantised keys (D % 128 == 0)
    uint8_t *k8_tgt, *k8_x;         // [pairs][Tp][D] e4m3 rows (workspace)
    float *qnorm_tgt, *qnorm_x;     // [pairs][Tp] norms of the quantised rows
};
size_t selfsim_batch_ws_bytes(int T, int D, int pairs);
void selfsim_batch_carve(void* base, int T, int D, int pairs, SelfSimBatch* b);   // fills T, Tp, D, pairs and the workspace pointers
int selfsim_target_launch(const SelfSimBatch& b, hipStream_t s);   // S* with the row norms taken inside the Gram kernel (1 launch; 2 on the fp8 path)
int selfsim_loss_launch(const SelfSimBatch& b, hipStream_t s);     // fused S / loss / W (norms in-kernel) + dK (2 launches; 3 on the fp8 path)
int mse_batched_launch(const float* a, int lda, size_t a_ps, const float* b, int ldb, size_t b_ps, int rows, int cols, float loss_weight,
                       float grad_weight, float* part, size_t part_ps, float* grad, int ldg, size_t g_ps, int pairs, hipStream_t s);
// 2-D strided MSE: loss_accum[0] += weight * mean((a-b)^2); grad (optional) = weight * 2 (a-b) / (rows*cols)
int mse_launch(const float* a, int lda, const float* b, int ldb, int rows, int cols, float weight, float* loss_accum,
               float* grad, int ldg, hipStream_t s);
// same with separate scales: loss_accum[0] += loss_weight * mean(d^2); grad = grad_weight * 2 d / n
// partial sums only (no atomics): part = SPLICE_MSE_PARTIALS zeroed floats, summed in index order by the caller
#define SPLICE_MSE_PARTIALS 1024
int mse_partials_launch(const float* a, int lda, const float* b, int ldb, int rows, int cols, float loss_weight, float grad_weight,
                        float* part, float* grad, int ldg, hipStream_t s);
int mse2_launch(const float* a, int lda, const float* b, int ldb, int rows, int cols, float loss_weight, float grad_weight,
                float* loss_accum, float* grad, int ldg, hipStream_t s);
