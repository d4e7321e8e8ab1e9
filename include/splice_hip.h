/* splice_hip.h -- C ABI of libsplice_hip.so: the MI355X (gfx950) implementation of the
 * Splice per-pair optimisation hot path.
 *
 * The reference (omerbt/Splice) has NO native/FFI boundary: its hot path is the Python API
 * of models/extractor.py, models/model.py, models/unet/skip.py, models/unet/common.py, util/losses.py, util/util.py
 * and the loop of train.py:51-80, executed as stock ATen ops.  This header is the boundary
 * a maintainer binds instead (ctypes stub: splice_amd/_lib.py).  Each
 * entry point cites the reference code it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  All pointers are DEVICE
 *     pointers owned by the caller unless stated otherwise; the library never frees them.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and returns 0 (SPLICE_OK) or a negative error code; the message
 *     is available from splice_last_error() (thread-local).  No exception crosses the ABI.
 *   - bf16 tensors are raw uint16 (splice_bf16).  "Tld" is the per-pass row stride of a
 *     token matrix (multiple of 32, >= T); rows T..Tld-1 of a pass are padding.
 *   - one handle <-> one device <-> one stream at a time; handles are not thread-safe;
 *     handles on different GPUs are independent (one process per GPU, no collectives).
 */
#ifndef SPLICE_HIP_H
#define SPLICE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t splice_bf16;
typedef void* splice_stream_t;

#define SPLICE_OK 0
#define SPLICE_ERR_ARG -1
#define SPLICE_ERR_HIP -2
#define SPLICE_ERR_STATE -3
#define SPLICE_ERR_NOMEM -4

int splice_version(void);
/* 1 when the library was built with the timing-only experiment switches (make DEV=1: SPLICE_STEP_ABLATE skips work, results
 * are garbage); 0 for the product build.  bench.py refuses to run on a 1. */
int splice_dev_switches(void);
const char* splice_last_error(void);

/* ------------------------------------------------------------------ op level: ViT GEMMs
 * C[M][N] = sum_k A[m][k] * B[n][k]  (bf16 in, fp32 accumulate) + fused epilogue.
 * Replaces the nn.Linear / Conv2d(patch-embed) call sites inside the DINO ViT that the
 * reference reaches through self.model(input_img) (models/extractor.py:83,91,99) and their
 * autograd dgrads (train.py:78).  K % 64 == 0, lda/ldb % 8 == 0. */
typedef struct splice_gemm_epilogue {
    const float* bias;        /* [N] */
    const float* resid;       /* fp32 [*][ldr]; row = resid_mod ? row % resid_mod : row */
    int ldr, resid_mod;
    float* out_f32;           /* [M][ldo] */
    int ldo;
    splice_bf16* out_bf;      /* [M][ldbf] */
    int ldbf;
    splice_bf16* out_bf_t;    /* transposed [N][ldt] (ldt % 4 == 0) */
    int ldt;
    splice_bf16* out_pre;     /* [M][ldp] value before GELU (may be NULL); written for rows >= pre_row_lo */
    int ldp, pre_row_lo;
    const splice_bf16* aux;   /* [M][ldaux] pre-GELU activations for SPLICE_EPI_GELU_GRAD */
    int ldaux;
    float* out_f32_cols;      /* fp32 copy of columns [col_lo, col_hi): [M][ld_cols] */
    int ld_cols, col_lo, col_hi;
    float alpha;
    int ksplit;               /* > 1 with flags == SPLICE_EPI_OUT_F32 only: K is cut into ksplit slices, slice s writes */
    long long slab_stride;    /* out_f32 + s * slab_stride; the consumer adds the slabs in order (deterministic split-K).
                               * With splice_gemm_splitk_slabs(M, ksplit) == 1 (many rows) the slices are summed IN the kernel,
                               * in the same order -- the same bits -- and only slab 0 is written */
    /* SPLICE_EPI_ROWDOT (with OUT_BF): rowdot[(row / rd_rows) * (N/64) * rd_rows + (col/64) * rd_rows + row % rd_rows]
     * = sum over the 64 columns [col, col+64) of bf16(C[row][c]) * rd_other[row][c] -- the attention backward's
     * delta = rowsum(dO * O) per (pass, head, query), formed where dO is produced (proj dgrad).  N % 64 == 0. */
    const splice_bf16* rd_other;   /* [M][ld_rd] */
    int ld_rd, rd_rows;
    float* rowdot;
    /* SPLICE_EPI_SCALE_RC (fp8 operands): C[m][n] *= row_scale[m] * col_scale[n] before the bias -- the per-token scale of the
     * quantised activations times the per-output-channel scale of the quantised weights */
    const float* row_scale;        /* [M]; NULL = 1 (unscaled e4m3 activations) */
    const float* col_scale;        /* [N] */
    /* SPLICE_EPI_OUT_F8: the result as e4m3 bytes [M][ld8] (saturated at +-448, no scale: the format's relative precision holds
     * over 2^-9 .. 448) -- the GELU output of fc1 as the operand of the fp8 fc2.  N % 16 == 0, ld8 % 16 == 0. */
    uint8_t* out_f8;
    int ld8;
    /* SPLICE_EPI_OUT_F8T: the same e4m3 bytes transposed, [N][ldt8] (ldt8 % 16 == 0) -- the token-contiguous operands of the
     * fp8 attention (V^T tiles), written next to the row-major copy by the QKV projection */
    uint8_t* out_f8_t;
    int ldt8;
} splice_gemm_epilogue;

enum {
    SPLICE_EPI_BIAS = 1, SPLICE_EPI_RESID = 2, SPLICE_EPI_OUT_F32 = 4, SPLICE_EPI_OUT_BF = 8,
    SPLICE_EPI_OUT_T = 16, SPLICE_EPI_GELU = 32, SPLICE_EPI_GELU_GRAD = 64,
    SPLICE_EPI_COLS_F32 = 128, SPLICE_EPI_ALPHA = 256, SPLICE_EPI_ROWDOT = 512, SPLICE_EPI_SCALE_RC = 1024, SPLICE_EPI_OUT_F8 = 2048,
    SPLICE_EPI_OUT_F8T = 4096
};

int splice_gemm_nt_bf16(unsigned flags, const splice_bf16* A, int lda, const splice_bf16* B, int ldb,
                        int M, int N, int K, const splice_gemm_epilogue* epi, splice_stream_t stream);

/* The same product with e4m3 (OCP fp8) operands on the gfx950 block-scaled K = 128 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4
 * with unit block scales: twice the bf16 issue rate, half the operand bytes): A [M][K] and B [N][K] are bytes, lda / ldb in
 * bytes (multiples of 16), K % 128 == 0.  BASELINE configs[4] ("fp8 MFMA attention + self-sim path"): the QKV, fc1 and fc2
 * projections of a ViT context in fp8 mode (splice_vit_ctx_set_fp8).  flags must contain SPLICE_EPI_SCALE_RC. */
int splice_gemm_nt_fp8(unsigned flags, const uint8_t* A, int lda, const uint8_t* B, int ldb, int M, int N, int K,
                       const splice_gemm_epilogue* epi, splice_stream_t stream);
/* Row-wise e4m3 quantisation: q[r][:] = fp8(x[r][:] * 448 / amax_r), scale[r] = amax_r / 448 (x ~= q * scale[r]).
 * x fp32 [rows][ldx] (cols % 8 == 0), q bytes [rows][ldq].  Packs the frozen weights per output channel. */
int splice_quantize_rows_fp8(const float* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int cols, splice_stream_t stream);
/* number of slabs a split-K call with M rows leaves for its consumer: ksplit (few rows) or 1 (the sum formed in the kernel) */
int splice_gemm_splitk_slabs(int M, int ksplit);
/* benchmarking / test hook (tools/gemm_bench.py, tests/test_ops_gpu.py): force the tile shape, 0 = automatic; tile + 10 * ring with tile 1..3 =
 * the one-barrier tiles 128x128, 128x64, 64x64 and ring 0 / 1 / 2 = 2 / 4 / 3 LDS stages; 5 = the 8-phase 256x256 tile (gemm8p.h) wherever its operand and
 * epilogue constraints hold (N % 256 == 0, K % 128 == 0, bias [+GELU] -> bf16 or bias + residual -> fp32) */
int splice_gemm_force_tile(int tile);
/* test / benchmarking hook: attention forward launch form, 0 = automatic.  bf16: 41 / 42 / 48 = the 32x32x16 kernel with 4 / 2 / 8 waves per workgroup
 * (same bits from all three, tests/test_ops_gpu.py); e4m3 forward: queries per wave / 16 + 10 * (two wave groups) */
int splice_attention_variant(int variant);
/* test / benchmarking hook: != 0 -> the stand-alone attention entry points take q columns pre-multiplied by scale * log2(e) and scale = ln 2
 * (what the ViT engine runs: attn_fwd_x32_kernel<., FOLD>, attn_bwd_x32_kernel); 0 -> plain q and scale = d^-1/2 */
int splice_attention_qfold(int on);
/* test / benchmarking hook: attention backward form, 0 automatic; 1 = the 16x16x32 halves (what plain q gets in any case); pre-scaled q only:
 * 2 / 3 = the 32x32x16 halves in one / two launches, 4 = one launch of two-wave workgroups (same bits from 2, 3 and 4) */
int splice_attention_bwd_variant(int variant);

/* LayerNorm(D, eps) of the DINO blocks (eps 1e-6), fp32 in -> bf16 out, and its dgrad
 * accumulated into the fp32 residual-gradient stream: g_out = g_in + dLN(dy). */
int splice_layernorm_fwd(const float* x, const float* gamma, const float* beta, splice_bf16* y,
                         float* mean, float* rstd, int rows, int D, float eps, splice_stream_t stream);
int splice_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                         const float* rstd, const float* g_in, float* g_out, splice_bf16* g_out_bf,
                         int rows, int D, splice_stream_t stream);

/* Multi-head self-attention (head dim 64), DINO Attention.forward: softmax(q k^T d^-1/2) v.
 * qkv [B*Tld][3D] bf16 (layout of models/extractor.py:136-151).  qkvT / ldt (the transpose [3D][ldt]) are not read any more --
 * V^T comes out of the V tile inside LDS (ds_read_b64_tr_b16) -- and may be NULL / 0; the parameters stay for ABI stability.
 * out [B*Tld][D] bf16; lse [B][H][Tld] fp32 (saved for the backward). */
int splice_attention_fwd(const splice_bf16* qkv, const splice_bf16* qkvT, int ldt, int B, int T, int Tld,
                         int D, int H, float scale, splice_bf16* out, float* lse, splice_stream_t stream);
/* The forward with e4m3 operands (BASELINE configs[4]): Q K^T and P V on the fp8 MFMA from UNSCALED e4m3 copies of qkv --
 * qkv8 bytes [B*Tld][3D] row-major and qkvT8 bytes [3D][ldt8] transposed (ldt8 % 16 == 0), as the fp8 QKV projection writes them
 * (SPLICE_EPI_OUT_F8 | SPLICE_EPI_OUT_F8T).  Output / lse as splice_attention_fwd; the backward runs on the bf16 tensors. */
int splice_attention_fwd_fp8(const uint8_t* qkv8, const uint8_t* qkvT8, int ldt8, int B, int T, int Tld, int D, int H, float scale,
                             splice_bf16* out, float* lse, splice_stream_t stream);
/* dqkv [B*Tld][3D] bf16 from dout [B*Tld][D]; delta is [B][H][Tld] fp32 scratch (rowsum(dO * O), formed by the call).
 * qkvT / ldt / doutT are not read any more (the kernels take the transposed operands out of the token tiles with
 * ds_read_b64_tr_b16) and may be NULL / 0: the parameters stay for ABI stability. */
int splice_attention_bwd(const splice_bf16* qkv, const splice_bf16* qkvT, int ldt, int B, int T, int Tld,
                         int D, int H, float scale, const splice_bf16* out, const float* lse,
                         const splice_bf16* dout, const splice_bf16* doutT, float* delta,
                         splice_bf16* dqkv, splice_stream_t stream);
/* attention probabilities [B][H][T][T] fp32 (the tensors models/extractor.py:44-45 hooks). */
int splice_attention_probs(const splice_bf16* qkv, int B, int T, int Tld, int D, int H, float scale,
                           const float* lse, float* probs, splice_stream_t stream);

/* dino_structure_transforms (data/transforms.py:30-37) on a device image fp32 [3][H][W] in [0,1]: horizontal flip,
 * ColorJitter (order[k] in {0 brightness, 1 contrast, 2 saturation, 3 hue}: the op order of this draw, n_ops = 0 for
 * none; factors[4] indexed by op) and GaussianBlur(3) (blur_sigma <= 0 for none); torchvision-0.10 tensor arithmetic.
 * out != img; scratch >= 3*H*W + 256 floats.  Replaces the per-step PIL pipeline of data/Dataset.py:62-70. */
int splice_augment_structure(const float* img, float* out, float* scratch, int H, int W, int flip, int n_ops, const int* order,
                             const float* factors, float blur_sigma, splice_stream_t stream);

/* attn_cosine_sim (models/extractor.py:4-9) on K fp32 [T][ldk] (D columns): S fp32 [T][T].
 * `ws` is caller scratch of splice_keys_selfsim_ws_bytes(T, D) bytes; the backward needs the
 * state the forward left in it. */
size_t splice_keys_selfsim_ws_bytes(int T, int D);
int splice_keys_selfsim_fwd(const float* K, int ldk, int T, int D, float eps, float* S, void* ws,
                            splice_stream_t stream);
int splice_keys_selfsim_bwd(const float* dS, const float* S, int T, int D, float eps, float* dK, int lddk,
                            int accumulate, void* ws, splice_stream_t stream);

/* F.mse_loss(a, b) (util/losses.py:82,93,104) on 2-D strided fp32 views:
 * loss_accum[0] += weight * mean((a-b)^2); grad (optional) = weight * 2 (a-b) / (rows*cols).
 * Workgroup partials are added in a fixed order (no float atomics) through a scratch line owned by the (device, stream)
 * pair of the call: calls on different streams or devices are independent. */
int splice_mse(const float* a, int lda, const float* b, int ldb, int rows, int cols, float weight,
               float* loss_accum, float* grad, int ldg, splice_stream_t stream);

/* patch-embed operand builders (Conv2d(3,D,p,p) as a GEMM; optional fused ImageNet Normalize,
 * util/losses.py:19) and helpers */
int splice_patchify(const float* img, splice_bf16* patches, int B, int H, int W, int p, int Tld,
                    int normalize, splice_stream_t stream);
int splice_unpatchify(const float* dpatches, float* dimg, int B, int H, int W, int p, int Tld,
                      int normalize, splice_stream_t stream);
int splice_cast_f32_bf16(const float* x, splice_bf16* y, size_t n, splice_stream_t stream);
int splice_cast_bf16_f32(const splice_bf16* x, float* y, size_t n, splice_stream_t stream);
int splice_transpose_f32_bf16(const float* x, splice_bf16* y, int rows, int cols, int ldy, splice_stream_t stream);
/* transforms.Resize on tensors (util/losses.py:20; torchvision 0.10: bilinear, align_corners=False,
 * no antialias) on `planes` images [h][w] -> [oh][ow], and its exact adjoint (deterministic gather). */
int splice_resize_bilinear_fwd(const float* in, float* out, int planes, int h, int w, int oh, int ow, splice_stream_t stream);
int splice_resize_bilinear_bwd(const float* dout, float* din, int planes, int h, int w, int oh, int ow, splice_stream_t stream);

/* ------------------------------------------------------------------ ViT engine (handle level)
 * Replaces VitExtractor.__init__ + the hooked self.model(img) forwards of
 * models/extractor.py:19-103 and the autograd backward through them (train.py:78).
 * Weights are the DINO state-dict entries (public checkpoint key names), frozen and
 * re-packed to bf16 ([out][in] and [in][out]) inside the handle. */
int splice_vit_create(int patch, int dim, int depth, int heads, void** out_handle);
void splice_vit_destroy(void* vit);
int splice_vit_set_param(void* vit, const char* name, const float* data, long long numel, splice_stream_t stream);
/* The factor the stored q columns of every layer's qkv carry (d^-1/2 * log2(e)).  splice_vit_read_tensor
 * kinds 1 and 3 divide it out of the copy; kind 7 (= kind 1 as stored) and the zero-copy splice_vit_get_tensor pointers do not -- attention entry
 * points fed with stored q take scale = ln 2 instead of d^-1/2. */
float splice_vit_qscale(void* vit);
int splice_vit_params_complete(void* vit);
/* BASELINE configs[4] fp8 path: prepares e4m3 copies of the QKV / fc1 / fc2 weights (per-output-channel scales).  A context
 * opts in with splice_vit_ctx_set_fp8: its three big forward projections then run on the fp8 MFMA (e4m3 LayerNorm outputs with
 * per-token scales, e4m3 GELU output), de-quantised in the GEMM epilogues.  dim % 128 == 0.  The backward (dgrad through bf16
 * weights) is unchanged. */
int splice_vit_enable_fp8(void* vit, splice_stream_t stream);
/* A context = one (batch, image shape): owns the activations of the last forward.
 * pos_TD: position table for this token grid, fp32 [T][dim] (interpolate_pos_encoding done
 * once per shape by the host). need_grad != 0 also allocates the backward workspace. */
int splice_vit_ctx_create(void* vit, int B, int H, int W, const float* pos_TD, int need_grad,
                          splice_stream_t stream, void** out_ctx);
void splice_vit_ctx_destroy(void* ctx);
int splice_vit_ctx_info(void* ctx, int* T, int* Tld, int* rows);
/* on != 0: behind the QKV projection of the TOP block only the [CLS] row of every pass is computed -- all the Splice losses
 * read of that block besides its keys (util/losses.py:90: [CLS] of block 11).  Then kind-0 rows other than [CLS] are
 * undefined for layer depth-1, and d_block[depth-1] must be zero outside the [CLS] rows.  splice_step_create switches its
 * contexts to this mode; the extractor API (models/extractor.py:81-103 hands out every token) never does. */
int splice_vit_ctx_set_top_cls_only(void* ctx, int on);
/* precision of THIS context's forward: 0 bf16 (default); 1 = QKV / fc1 / fc2 projections on the fp8 MFMA; 3 = those + the attention
 * forward (Q K^T, P V) on the fp8 MFMA.  Needs splice_vit_enable_fp8 on the engine.  The backward is bf16 in every mode. */
int splice_vit_ctx_set_fp8(void* ctx, int mode);
/* img fp32 [B][3][H][W]; normalize != 0 fuses transforms.Normalize (util/losses.py:19). */
int splice_vit_forward(void* ctx, const float* img, int normalize, splice_stream_t stream);
/* same; passes [0, grad_pass_begin) are no-grad targets (util/losses.py:79,91,101 `with torch.no_grad()`):
 * tensors only a backward would read are not stored for them */
int splice_vit_forward_ex(void* ctx, const float* img, int normalize, int grad_pass_begin, splice_stream_t stream);
/* same, restricted to passes (images) [pass_begin, pass_end) of the ctx batch; img is still the full [B] batch.
 * The forward is row-parallel over the token matrix: disjoint pass ranges of one ctx may run concurrently on
 * different streams (splice_step_run computes the no-grad target passes beside the generator forward). */
int splice_vit_forward_passes(void* ctx, const float* img, int normalize, int grad_pass_begin, int pass_begin, int pass_end,
                              splice_stream_t stream);
/* kind 0: block output l fp32 [rows][D] (models/extractor.py:56-60) | 1: raw qkv l bf16
 * [rows][3D] (:68-72) | 2: attention output l bf16 [rows][D] | 3: last-layer qkv fp32
 * [rows][3D] | 4: lse l fp32 [B][H][Tld] | 5: embedded tokens fp32 [rows][D] | 6: raw qkv l transposed, bf16 [3D][rows] | 7: as 1.
 * ZERO-COPY: the pointers are the engine's own buffers, so the q columns of kinds 1 / 3 / 6 / 7 carry the factor splice_vit_qscale()
 * (a C caller that wants the reference's q divides it out, or calls splice_vit_read_tensor, whose copies of kinds 1 and 3
 * are un-scaled -- in bf16 for kind 1: one more rounding; Python's VitExtractor reads kind 7 and divides in fp32). */
int splice_vit_get_tensor(void* ctx, int kind, int layer, void** out_ptr);
int splice_vit_read_tensor(void* ctx, int kind, int layer, void* dst, size_t bytes, splice_stream_t stream);
/* dgrad-only backward over passes [pass_begin, pass_end): gradients may be injected at
 * any block output (d_block[l], fp32 [rows][D]), any raw qkv (d_qkv[l], fp32 [rows][3D])
 * or its key columns only (d_keys[l], fp32 [rows][D]); arrays of `depth` pointers, NULL
 * entries / NULL arrays allowed.  d_img fp32 [B][3][H][W] (selected passes written). */
int splice_vit_backward(void* ctx, int pass_begin, int pass_end, const float* const* d_block,
                        const float* const* d_qkv, const float* const* d_keys, float* d_img,
                        int normalize, splice_stream_t stream);

/* ------------------------------------------------------------------ generator engine
 * The reference's define_G() -> skip() U-Net (models/networks.py:56-58, models/unet/skip.py:4-102,
 * models/unet/common.py:11-124), forward + full backward, fp32.  Parameters / gradients are
 * ONE flat fp32 arena in netG.parameters() order; splice_gen_tensor_info gives the per-tensor
 * (state_dict name, offset, numel) table.  A plan = (N side-by-side generator calls, H, W). */
int splice_gen_create(void** out_handle);
/* Any skip() the kernels cover (models/unet/skip.py:4-11): e.g. the feature-inversion net of inversion.py:21-25 -- 6 scales,
 * 32 noise input channels, filter_size_down = filter_size_up = {7,7,5,5,3,3}, pad = 'reflection'.  Same engine, same plans,
 * same flat parameter arena (state_dict names follow the reference's numbering, incl. the ".1" Conv2d index behind a
 * ReflectionPad2d).  Fixed: stride-2 'stride' down-sampling, bilinear up-sampling, LeakyReLU(0.2), 1x1 skip and post-up
 * filters, sigmoid head, every scale with skip channels.  NULL = the default architecture of define_G. */
#define SPLICE_GEN_MAX_SCALES 6
typedef struct splice_gen_arch {
    int n_scales, in_channels, out_channels;
    int down[SPLICE_GEN_MAX_SCALES], up[SPLICE_GEN_MAX_SCALES], skip[SPLICE_GEN_MAX_SCALES];   /* channels (<= 128) */
    int filter_down[SPLICE_GEN_MAX_SCALES], filter_up[SPLICE_GEN_MAX_SCALES];                  /* 1 / 3 / 5 / 7 */
    int filter_skip;                                                                          /* 1 */
    int reflect;                                                                              /* pad = 'reflection' (else 'zero') */
} splice_gen_arch;
int splice_gen_create_arch(const splice_gen_arch* arch, void** out_handle);
void splice_gen_destroy(void* gen);
long long splice_gen_param_count(void* gen);
int splice_gen_num_tensors(void* gen);
int splice_gen_tensor_info(void* gen, int i, const char** name, long long* offset, long long* numel);
int splice_gen_plan_create(void* gen, int N, int H, int W, int need_grad, void** out_plan);
void splice_gen_plan_destroy(void* plan);
/* y = netG(x) per image (models/model.py:15-23): x,y fp32 [N][3][H][W] */
int splice_gen_forward(void* plan, const float* params, const float* x, float* y, splice_stream_t stream);
/* same, without the plan's private copies of x and y (two launches): the caller keeps both buffers unchanged until the
 * matching splice_gen_backward has run (splice_step_run owns its staged inputs and outputs) */
int splice_gen_forward_borrowed(void* plan, const float* params, const float* x, float* y, splice_stream_t stream);
/* parameter gradients of the last forward from dy = dL/dy (autograd backward through netG,
 * train.py:78); grads overwritten, or accumulated when accumulate != 0 */
int splice_gen_backward(void* plan, const float* params, const float* dy, float* grads, int accumulate,
                        splice_stream_t stream);
/* BatchNorm buffers of netG.state_dict() ("<bn>.running_mean", "<bn>.running_var"; fp32 arena in state_dict order) and their
 * train-mode update from the batch statistics of the LAST forward of each listed plan, applied in the order given (the
 * order of the netG calls).  Plans of independent images (arena stride set) update arena n for image n. */
long long splice_gen_buffer_count(void* gen);
int splice_gen_num_buffers(void* gen);
int splice_gen_buffer_info(void* gen, int i, const char** name, long long* offset, long long* numel);
int splice_gen_running_stats_update(void* const* plans, int n_plans, float* running, long long running_stride, float momentum,
                                    splice_stream_t stream);
/* torch.optim.Adam(lr, betas) step (util/util.py:28-32, train.py:79) fused over the arena;
 * step counts from 1; zero_grad != 0 also clears grads (optimizer.zero_grad, train.py:56) */
int splice_adam_step(float* params, float* grads, float* m, float* v, long long n, float lr, float beta1,
                     float beta2, float eps, int step, int zero_grad, splice_stream_t stream);

/* live timing of one kernel family (bench.py roofline leg, prof.hip): while a family is armed its kernels are launched with
 * a start / stop event pair each (hipExtLaunchKernelGGL: the kernel's own begin / end time stamps, as rocprofv3 reports them).
 * which 1 = fc1 GEMM fwd, 2 = qkv GEMM fwd (layers 0..depth-2), 3 = attention fwd, 4 = fc2 GEMM fwd, 5 = split-K dgrad GEMMs
 * (fc1^T, qkv^T), 6 = attention bwd, 7 = generator (all kernels of splice_gen_forward* / splice_gen_backward), 8 = key
 * self-similarity loss kernels, 9 = proj GEMM fwd.  end: summed kernel time, host calls of the family, kernels launched. */
int splice_prof_begin(int which);
int splice_prof_end(float* total_ms, int* launches);
int splice_prof_end_ex(float* total_ms, int* calls, int* kernels);
/* as splice_prof_end_ex, plus one text line per distinct kernel of the family in `detail` (may be NULL):
 * "kernel\tlaunches\ttotal ms\talgorithmic FLOPs\talgorithmic bytes\n", longest first (the work columns are what the launchers noted: the convolutions) */
int splice_prof_end_detail(float* total_ms, int* calls, int* kernels, char* detail, int detail_len);
int splice_prof_active(void);
int splice_vit_ctx_dims(void* ctx, int* B, int* H, int* W, int* D, int* depth, int* heads, int* patch);
int splice_gen_plan_dims(void* plan, int* N, int* H, int* W, long long* nparams);
/* stride > 0: the N images of the plan are INDEPENDENT generators (P image pairs optimised side by side: the reference
 * runs train.py once per pair): image n reads params + n*stride, its gradient goes to grads + n*stride, BatchNorm and
 * launch policies are per image -- a pair's result is bit-identical to its N = 1 run.  0 (default): one generator. */
int splice_gen_plan_set_arena_stride(void* plan, long long stride);
/* on != 0: the plan is ONE netG call on a batch of N <= 8 images (n_crops > 1: models/model.py:15, data/transforms.py:19-27):
 * BatchNorm statistics over the whole batch as nn.BatchNorm2d takes them; gradients summed over the images.  Default 0:
 * N separate batch-1 calls. */
int splice_gen_plan_set_batch_stats(void* plan, int on);
/* re-target a plan to a smaller input without reallocating (per-step random crop sizes,
 * data/transforms.py:21-22) */
int splice_gen_plan_resize(void* plan, int H, int W);

/* ------------------------------------------------------------------ fused optimisation step
 * train.py:51-80 for P image pairs side by side (P = 1: the reference's one pair per process): Model.forward
 * (models/model.py:12-25), LossG.forward incl. its lambda schedule (util/losses.py:34-72), loss.backward(),
 * optimizer.step().  Built on a B = 4P ViT context [A'_0.. | B'_0.. | x'_0.. | y'_0..] (X' = T(X crop), x = G(A crop),
 * y = G(B crop)), two generator plans of P independent images (A crops, B crops) and, for the every-75th-step
 * entire-image branch, a B = 2P context and a third plan.  Pairs share only the frozen ViT; a pair's results do not
 * depend on P (bit-identical to its P = 1 run). */
typedef struct splice_step_config {
    int crop_h, crop_w;          /* (maximum) size of the global crops fed to G */
    int vit_h, vit_w;            /* after global_transform's Resize (== crop when it is the identity) */
    int ent_h, ent_w;            /* entire structure image (0 = branch disabled) */
    int ent_vit_h, ent_vit_w;    /* its size after Resize */
    float lambda_global_cls, lambda_global_ssim, lambda_global_identity, lambda_entire_cls, lambda_entire_ssim;
    int entire_every, cls_warmup;
    float lr, beta1, beta2, eps;
    int pairs;                   /* P (0 or 1: one pair) */
    long long arena_stride;      /* P > 1: floats between the pairs' parameter / gradient / Adam-moment arenas (>= param count);
                                  * the generator plans must have been given the same stride (splice_gen_plan_set_arena_stride) */
    int fp8_selfsim;             /* != 0: the key self-similarity Gram matrices on the fp8 MFMA (per-row e4m3 keys; dim % 128 == 0) */
    int top_cls_only;            /* != 0: the step switches its ViT contexts to splice_vit_ctx_set_top_cls_only (the Python engine's default;
                                  * results differ from the full top block by rounding only) */
    int n_crops;                 /* > 1 (with pairs <= 1): global_{A,B}_crops_n_crops of conf/default/config.yaml -- the step works on
                                  * n_crops crops of ONE pair: A_crop / B_crop are [n_crops][3][h][w], the generator plans hold n_crops
                                  * images in batch-statistics mode (splice_gen_plan_set_batch_stats: netG sees the stacked crops,
                                  * data/transforms.py:27), each loss term is summed over the crops (util/losses.py:75-105), one arena */
    int n_crops_b;               /* 0: global_B_crops_n_crops == n_crops.  > 0: the B-crop plan holds n_crops_b images and n_crops counts
                                  * the A crops only -- the reference zips the crop lists (util/losses.py:76,87,98): structure term over the
                                  * A crops, identity term over the B crops, appearance term over min(n_crops, n_crops_b) pairs */
} splice_step_config;
/* gen_plan_a / gen_plan_b: N = P plans at the crop size for the A and the B crops; gen_plan_entire: N = P at the entire size
 * (NULL with ent_h == 0).  Contexts: need_grad, B = 4P / 2P. */
int splice_step_create(const splice_step_config* cfg, void* vit_ctx_global, void* vit_ctx_entire, void* gen_plan_a, void* gen_plan_b,
                       void* gen_plan_entire, void** out_handle);
void splice_step_destroy(void* step);
/* params / grads / m / v: [P][arena_stride] (one flat arena for P = 1); A_crop, B_crop: [P][3][h][w] at the current crop
 * sizes; A_entire: [P][3][ent_h][ent_w] or NULL (needed when step_idx % entire_every == 0).
 * losses_out: device fp32 [P][8], per pair {loss, loss_global_ssim, loss_entire_ssim, loss_entire_cls, loss_global_cls,
 * loss_global_id_B, 0, 0} -- the keys of the dict LossG.forward returns.  Written by the step's own loss kernel; the
 * captured hipGraph is bound to the arena pointers AND to losses_out, so keep passing the same buffers (a different
 * pointer re-captures; two consecutive steps with identical pointers and crop sizes are needed before a graph is used). */
int splice_step_run(void* step, float* params, float* grads, float* m, float* v, const float* A_crop, const float* B_crop,
                    const float* A_entire, int step_idx, float* losses_out, splice_stream_t stream);
/* generator outputs of the last step: which 0 x_global [P][3][a_h][a_w], 1 x_entire, 2 y_global */
int splice_step_output(void* step, int which, float** out_ptr);
/* graph bookkeeping of this handle: out[0] = captures that updated a retired executable in place (hipGraphExecUpdate; executables are pooled per
 * step configuration and never destroyed while the process runs), out[1] = updates the runtime refused, out[2] = executables instantiated */
int splice_step_graph_stats(void* step, long long* out);
/* 1 (default): capture the step's launch sequence once per regime into a hipGraph and replay it;
 * 0: launch every kernel eagerly (also used automatically while splice_prof_begin is armed) */
int splice_step_use_graph(void* step, int on);
/* 1 = the independent chains of a step (target passes beside the generator, the two ViT backward chains) run on two
 * streams (default), 0 = one stream; results are bit-identical either way */
int splice_step_use_overlap(void* step, int on);
/* per-step crop sizes (<= creation size) of the A and the B crops (the reference draws them independently,
 * data/Dataset.py:66-67); shared by all pairs of the batch */
int splice_step_set_crops(void* step, int a_h, int a_w, int b_h, int b_w);
/* skip_adam != 0: stop after backward -- `grads` = gradient of this step's loss (+= previous content when accumulate != 0),
 * parameters untouched; several losses (the same crops at several ViT input scales) are summed that way before ONE
 * splice_adam_step */
int splice_step_set_mode(void* step, int skip_adam, int accumulate);
/* Run part of a step: phases = mask of 1 generator forward (+ input staging), 2 ViT forward / losses / ViT backward down to
 * the gradient of the generated images, 4 generator backward (+ Adam unless splice_step_set_mode disables it); default 7.
 * leader != NULL (phases must be 2; another step handle with the same image shapes): this handle reads the leader's staged
 * inputs and generator outputs and ADDS its image gradients to the leader's.  The several-scales step (the reference has
 * one scale, util/losses.py:22-27; BASELINE configs[4] asks for three) is leader 1|2, followers 2, leader 4: one generator
 * pass per step whatever the number of scales. */
int splice_step_set_phases(void* step, int phases, void* leader);
/* BatchNorm running statistics of netG (models/unet/common.py:95-96, momentum 0.1): when `running` is set every step
 * applies the updates of its generator calls in the reference's order (A_global, A on entire steps, B_global) to the
 * caller's buffer arena(s): layout splice_gen_buffer_info, pair p at running + p * stride.  NULL = not tracked. */
int splice_step_set_running_stats(void* step, float* running, long long stride);

#ifdef __cplusplus
}
#endif
#endif /* SPLICE_HIP_H */
