// DINO-ViT feature-extraction engine: frozen pre-packed weights + batched forward and
// dgrad-only backward, built from the kernels of gemm.h / attention.hip / vit_ops.hip.
//
// Replaces what the reference does with forward hooks on a torch.hub model
// (models/extractor.py:19-103): one forward over a BATCH of images (the reference loops
// batch-1 calls, util/losses.py:76-81) keeps, per layer, exactly the tensors its hooks
// expose -- block output, raw qkv, attention output -- plus what the backward needs; the
// backward propagates gradients injected at any block output / qkv back to the image and
// never forms weight gradients (the reference accumulates and discards them).
#include <string.h>

#include <string>
#include <vector>

#include "kernels.h"

void splice_set_error(const char* fmt, ...);
extern "C" int splice_vit_params_complete(void* vit);

#define HIPCHK(x)                                                                                 \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            splice_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            return SPLICE_ERR_HIP;                                                                \
        }                                                                                         \
    } while (0)
#define RC(x)                                                                                     \
    do {                                                                                          \
        int rc_ = (x);                                                                            \
        if (rc_ != SPLICE_OK) {                                                                   \
            splice_set_error("%s:%d %s failed (%d)", __FILE__, __LINE__, #x, rc_);                \
            return rc_;                                                                           \
        }                                                                                         \
    } while (0)

// (live timing of one kernel family for the roofline leg: SpliceProfScope, prof.hip)

struct Linear {
    bf16_t* w = nullptr;    // [out][in]
    bf16_t* wT = nullptr;   // [in][out]
    float* b = nullptr;     // [out]
    int out = 0, in = 0;
    uint8_t* w8 = nullptr;  // [out][in] e4m3, quantised per output channel (fp8 mode: qkv, fc1, fc2)
    float* w8_scale = nullptr;   // [out]
};
struct LayerW {
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    Linear qkv, proj, fc1, fc2;
};

struct SpliceVit {
    int patch = 0, dim = 0, depth = 0, heads = 0, hidden = 0;
    int device = 0;
    std::vector<LayerW> layers;
    Linear pe;              // patch-embed as [dim][3*p*p]
    float* cls = nullptr;   // [dim]
    float* pos = nullptr;   // [pos_tokens][dim]  (trained grid, incl. cls row)
    int pos_tokens = 0;
    float *norm_g = nullptr, *norm_b = nullptr;
    std::vector<void*> allocs;
    int n_set = 0;
    int fp8 = 0;            // e4m3 copies of the qkv / fc1 / fc2 weights exist (splice_vit_enable_fp8); contexts opt in (splice_vit_ctx_set_fp8)
    // q folding (round 5): the q rows of every QKV projection (weight rows and bias entries [0, dim)) are packed pre-multiplied by
    // d^-1/2 * log2(e), so the stored q IS the base-2 softmax exponent's left factor: the attention kernels skip the scale-and-shift
    // FMA per score (attn_x32.h) and every kernel that takes "scale" gets ln 2 instead of d^-1/2 (scale * q.k == ln2 * q'.k).
    // The dgrad uses the same packed weights, so the attention backward hands it dL/dq' (the same formulas with ln 2 for the scale).
    // What leaves the engine is unscaled again: splice_vit_read_tensor divides the q columns, gradient seeds for qkv are divided on the way in.
    int qfold = 1;
    float qscale = 1.0f;    // 0.125 * log2(e) when folding
};
static const float kLn2 = 0.6931471805599453f, kLog2e = 1.4426950408889634f;

struct SpliceVitCtx {
    SpliceVit* vit = nullptr;
    int B = 0, H = 0, W = 0, T = 0, Tld = 0, rows = 0, need_grad = 0;
    int grad_pass_begin = 0;               // passes below this index never run a backward (targets): skip backward-only saves
    float* pos_eff = nullptr;              // [Tld][D]
    bf16_t* patches = nullptr;             // [rows][3pp]
    std::vector<float*> xs;                // depth+1 x [rows][D]   residual stream (block outputs)
    std::vector<float*> xmid;              // depth   x [rows][D]
    std::vector<float*> mean1, rstd1, mean2, rstd2;  // depth x [rows]
    std::vector<bf16_t*> qkv, attn_out, hpre;  // depth
    bf16_t* qkvT_last = nullptr;   // [3D][rows] transpose of the LAST layer's qkv (the [CLS]-row kernels of vit_cls.hip read token-contiguous K / V);
                                   // the attention kernels proper transpose inside LDS and need no such copy
    std::vector<float*> lse;               // depth x [B][H][Tld]
    float* qkv_last_f32 = nullptr;         // [rows][3D]
    bf16_t* ln_out = nullptr;              // [rows][D]   transient
    int fp8 = 0;                           // bit 0: this context's QKV / fc1 / fc2 forward projections run on the fp8 MFMA; bit 1: so does its attention forward
    uint8_t* ln_out8 = nullptr;            // [rows][D]   e4m3 LayerNorm output (fp8 mode)
    float* ln_scale = nullptr;             // [rows]      its per-token scales
    uint8_t* hact8 = nullptr;              // [rows][4D]  e4m3 GELU output of fc1 = operand of fc2 (fp8 mode; unscaled)
    uint8_t *qkv8 = nullptr, *qkvT8 = nullptr;   // [rows][3D] / [3D][rows] unscaled e4m3 copies of the current layer's qkv (fp8 attention forward)
    bf16_t* hact = nullptr;                // [rows][4D]  transient
    // backward temporaries (sized for all rows)
    float* g = nullptr;                    // [rows][D]
    bf16_t* g_bf = nullptr;
    bf16_t* dh = nullptr;                  // [rows][4D]
    float* dln = nullptr;                  // [rows][D]
    bf16_t *dout = nullptr, *dqkv = nullptr;
    float* delta = nullptr;
    float* dpatches = nullptr;             // [rows][3pp]
    std::vector<void*> allocs;
    int forward_done = 0;
    // top block only where the Splice losses read it (vit_cls.hip): behind the last QKV projection only the [CLS] row of a pass goes on
    int top_cls_only = 0;
    bf16_t *cls_attn = nullptr, *cls_ln = nullptr, *cls_h = nullptr, *cls_dh = nullptr, *cls_dout = nullptr;   // [B][D] / [B][4D], one row per pass
    float *cls_probs = nullptr, *cls_dln = nullptr;                                                            // [B][H][Tld], [B][D]
    float* cls_slabs = nullptr;            // [16][B][4D] split-K slabs of the M = passes GEMMs of that tail
};

template <class T>
static int dev_alloc(std::vector<void*>& list, T** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(T) + 256) != hipSuccess) {
        splice_set_error("hipMalloc of %zu bytes failed", n * sizeof(T));
        return SPLICE_ERR_NOMEM;
    }
    list.push_back(q);
    *p = (T*)q;
    return SPLICE_OK;
}

static int pack_linear(SpliceVit* v, Linear& L, const float* w, int out, int in, hipStream_t s) {
    L.out = out; L.in = in;
    RC(dev_alloc(v->allocs, &L.w, (size_t)out * in));
    RC(dev_alloc(v->allocs, &L.wT, (size_t)out * in));
    RC(cast_f32_bf16_launch(w, L.w, (size_t)out * in, s));
    RC(transpose_f32_to_bf16_launch(w, L.wT, out, in, out, s));  // wT[in][out]
    return SPLICE_OK;
}
__global__ void scale_f32_kernel(float* dst, const float* src, size_t n, size_t n_scaled, float f) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = i < n_scaled ? src[i] * f : src[i];
}
// the QKV projection with its q rows scaled (SpliceVit::qfold): fp32 product first, ONE rounding to bf16
static int pack_qkv(SpliceVit* v, Linear& L, const float* w, int D, hipStream_t s) {
    if (!v->qfold) return pack_linear(v, L, w, 3 * D, D, s);
    float* tmp = nullptr;
    const size_t n = (size_t)3 * D * D;
    HIPCHK(hipMalloc((void**)&tmp, n * sizeof(float)));
    hipLaunchKernelGGL(scale_f32_kernel, dim3(1024), dim3(256), 0, s, tmp, w, n, (size_t)D * D, v->qscale);
    const int rc = pack_linear(v, L, tmp, 3 * D, D, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(tmp);
    return rc;
}
static int copy_vec(SpliceVit* v, float** dst, const float* src, size_t n, hipStream_t s) {
    RC(dev_alloc(v->allocs, dst, n));
    HIPCHK(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SPLICE_OK;
}

// add fp32 src[rows][lds] (cols columns) into bf16 dst[rows][ldd] columns [col0, col0+cols)
// (the first qcols source columns are multiplied by qf: gradient seeds for q arrive as dL/dq, the engine carries dL/dq')
__global__ void add_f32_into_bf16_kernel(bf16_t* dst, int ldd, int col0, const float* src, int lds, int rows, int cols, int qcols, float qf) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = i / cols, c = i % cols;
        bf16_t* p = dst + (size_t)r * ldd + col0 + c;
        const float v = src[(size_t)r * lds + c];
        *p = f2bf(bf2f(*p) + (c < qcols ? v * qf : v));
    }
}
// g (+)= add ; g_bf = bf16(g).  add may be null (then just refresh g_bf); init: g = add (or 0).
__global__ void grad_stream_kernel(float* g, bf16_t* g_bf, const float* add, size_t n, int init) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = init ? 0.f : g[i];
        if (add) v += add[i];
        g[i] = v;
        g_bf[i] = f2bf(v);
    }
}
__global__ void scale_cols_bf16_kernel(bf16_t* x, int ld, int rows, int cols, float f) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        bf16_t* p = x + (i / cols) * ld + i % cols;
        *p = f2bf(bf2f(*p) * f);
    }
}
__global__ void scale_cols_f32_kernel(float* x, int ld, int rows, int cols, float f) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[(i / cols) * ld + i % cols] *= f;
}
static inline unsigned grid_n(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b > 2048 ? 2048 : (b ? b : 1)); }

extern "C" {

int splice_vit_create(int patch, int dim, int depth, int heads, void** out) {
    if (!out || (patch != 8 && patch != 16) || dim % 64 || dim / 64 != heads || depth < 1 || dim > 1024) {
        splice_set_error("splice_vit_create: unsupported config patch=%d dim=%d depth=%d heads=%d (head dim must be 64)", patch, dim, depth, heads);
        return SPLICE_ERR_ARG;
    }
    SpliceVit* v = new SpliceVit();
    v->patch = patch; v->dim = dim; v->depth = depth; v->heads = heads; v->hidden = 4 * dim;
    v->qfold = 1;   // (the plain packing of rounds 1-4 is gone with its kernels; the field stays: the kernels below are written for either)
    v->qscale = v->qfold ? 0.125f * kLog2e : 1.0f;
    (void)hipGetDevice(&v->device);
    v->layers.resize(depth);
    *out = v;
    return SPLICE_OK;
}

void splice_vit_destroy(void* h) {
    SpliceVit* v = (SpliceVit*)h;
    if (!v) return;
    for (void* p : v->allocs) (void)hipFree(p);
    delete v;
}

// One DINO state-dict entry (public checkpoint key names), fp32 on the device.
int splice_vit_set_param(void* h, const char* name, const float* data, long long numel, splice_stream_t stream) {
    SpliceVit* v = (SpliceVit*)h;
    hipStream_t s = (hipStream_t)stream;
    if (!v || !name || !data) return SPLICE_ERR_ARG;
    const int D = v->dim, Hd = v->hidden, pp3 = 3 * v->patch * v->patch;
    std::string n(name);
    auto expect = [&](long long want) {
        if (numel != want) { splice_set_error("splice_vit_set_param(%s): numel %lld, expected %lld", name, numel, want); return false; }
        return true;
    };
    if (n == "cls_token") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &v->cls, data, D, s)); }
    else if (n == "pos_embed") {
        if (numel % D) return SPLICE_ERR_ARG;
        v->pos_tokens = (int)(numel / D);
        RC(copy_vec(v, &v->pos, data, numel, s));
    }
    else if (n == "patch_embed.proj.weight") { if (!expect((long long)D * pp3)) return SPLICE_ERR_ARG; RC(pack_linear(v, v->pe, data, D, pp3, s)); }
    else if (n == "patch_embed.proj.bias") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &v->pe.b, data, D, s)); }
    else if (n == "norm.weight") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &v->norm_g, data, D, s)); }
    else if (n == "norm.bias") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &v->norm_b, data, D, s)); }
    else if (n.rfind("blocks.", 0) == 0) {
        const size_t dot = n.find('.', 7);
        if (dot == std::string::npos) return SPLICE_ERR_ARG;
        const int li = atoi(n.substr(7, dot - 7).c_str());
        if (li < 0 || li >= v->depth) { splice_set_error("splice_vit_set_param: layer %d out of range", li); return SPLICE_ERR_ARG; }
        LayerW& L = v->layers[li];
        const std::string r = n.substr(dot + 1);
        if (r == "norm1.weight") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.ln1_g, data, D, s)); }
        else if (r == "norm1.bias") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.ln1_b, data, D, s)); }
        else if (r == "norm2.weight") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.ln2_g, data, D, s)); }
        else if (r == "norm2.bias") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.ln2_b, data, D, s)); }
        else if (r == "attn.qkv.weight") { if (!expect(3LL * D * D)) return SPLICE_ERR_ARG; RC(pack_qkv(v, L.qkv, data, D, s)); }
        else if (r == "attn.qkv.bias") { if (!expect(3 * D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.qkv.b, data, 3 * D, s)); if (v->qfold) hipLaunchKernelGGL(scale_f32_kernel, dim3(8), dim3(256), 0, s, L.qkv.b, L.qkv.b, (size_t)3 * D, (size_t)D, v->qscale); }
        else if (r == "attn.proj.weight") { if (!expect((long long)D * D)) return SPLICE_ERR_ARG; RC(pack_linear(v, L.proj, data, D, D, s)); }
        else if (r == "attn.proj.bias") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.proj.b, data, D, s)); }
        else if (r == "mlp.fc1.weight") { if (!expect((long long)Hd * D)) return SPLICE_ERR_ARG; RC(pack_linear(v, L.fc1, data, Hd, D, s)); }
        else if (r == "mlp.fc1.bias") { if (!expect(Hd)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.fc1.b, data, Hd, s)); }
        else if (r == "mlp.fc2.weight") { if (!expect((long long)Hd * D)) return SPLICE_ERR_ARG; RC(pack_linear(v, L.fc2, data, D, Hd, s)); }
        else if (r == "mlp.fc2.bias") { if (!expect(D)) return SPLICE_ERR_ARG; RC(copy_vec(v, &L.fc2.b, data, D, s)); }
        else { splice_set_error("splice_vit_set_param: unknown key %s", name); return SPLICE_ERR_ARG; }
    } else { splice_set_error("splice_vit_set_param: unknown key %s", name); return SPLICE_ERR_ARG; }
    v->n_set++;
    return SPLICE_OK;
}

// BASELINE configs[4] ("fp8 MFMA attention + self-sim path"): prepares the e4m3 copies of the QKV, fc1 and fc2 weights
// (quantised once per output channel).  A context created afterwards runs those three forward projections on the fp8 MFMA
// when it opts in (splice_vit_ctx_set_fp8) -- the mode is a property of the CONTEXT, so engines that share one frozen ViT
// keep their own precision.  Needs dim % 128 == 0.  The dgrad path (bf16 weights) is unchanged.
int splice_vit_enable_fp8(void* h, splice_stream_t stream) {
    SpliceVit* v = (SpliceVit*)h;
    if (!v || !splice_vit_params_complete(h) || v->dim % 128) { splice_set_error("splice_vit_enable_fp8: incomplete weights or dim %% 128 != 0"); return SPLICE_ERR_STATE; }
    hipStream_t s = (hipStream_t)stream;
    for (auto& L : v->layers) {
        for (Linear* lin : {&L.qkv, &L.fc1, &L.fc2}) {
            if (lin->w8) continue;
            RC(dev_alloc(v->allocs, &lin->w8, (size_t)lin->out * lin->in));
            RC(dev_alloc(v->allocs, &lin->w8_scale, (size_t)lin->out));
            RC(quantize_rows_bf16_fp8_launch(lin->w, lin->in, lin->w8, lin->in, lin->w8_scale, lin->out, lin->in, s));
        }
    }
    v->fp8 = 1;
    return SPLICE_OK;
}

/* The factor the stored q columns carry (d^-1/2 log2(e), or 1 when SPLICE_VIT_QFOLD=0): splice_vit_read_tensor kinds 1 / 3 divide it out,
 * kind 7 and the zero-copy splice_vit_get_tensor pointers do not; attention entry points fed with stored q take scale = ln 2. */
float splice_vit_qscale(void* h) { return h ? ((SpliceVit*)h)->qscale : 1.0f; }

int splice_vit_params_complete(void* h) {
    SpliceVit* v = (SpliceVit*)h;
    if (!v) return 0;
    bool ok = v->cls && v->pos && v->pe.w && v->pe.b;
    for (auto& L : v->layers)
        ok = ok && L.ln1_g && L.ln1_b && L.ln2_g && L.ln2_b && L.qkv.w && L.qkv.b && L.proj.w && L.proj.b && L.fc1.w &&
             L.fc1.b && L.fc2.w && L.fc2.b;
    return ok ? 1 : 0;
}

// pos_eff_host-independent: caller passes the (already interpolated) position table for this
// image shape as fp32 [T][D] on the device (row 0 = cls position); K20 of SURVEY.md is done
// once per shape by the host (splice_amd/extractor.py).
int splice_vit_ctx_create(void* h, int B, int H, int W, const float* pos_TD, int need_grad, splice_stream_t stream, void** out);
}

__global__ void pos_eff_kernel(float* pos_eff, const float* pos, const float* cls, const float* pe_bias, int T, int Tld, int D) {
    const size_t n = (size_t)Tld * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i / D, d = i % D;
        float v;
        if (t == 0) v = cls[d] + pos[d] - pe_bias[d];          // GEMM adds the bias back on the zero patch row
        else if (t < T) v = pos[(size_t)t * D + d];
        else v = -pe_bias[d];                                  // padding tokens are exactly zero
        pos_eff[i] = v;
    }
}

extern "C" {

int splice_vit_ctx_create(void* h, int B, int H, int W, const float* pos_TD, int need_grad, splice_stream_t stream, void** out) {
    SpliceVit* v = (SpliceVit*)h;
    hipStream_t s = (hipStream_t)stream;
    if (!v || !out || B < 1 || !splice_vit_params_complete(h)) {
        splice_set_error("splice_vit_ctx_create: bad handle / incomplete weights / B<1");
        return SPLICE_ERR_STATE;
    }
    const int p = v->patch, D = v->dim, Hd = v->hidden, pp3 = 3 * p * p;
    if (H < p || W < p) return SPLICE_ERR_ARG;
    SpliceVitCtx* c = new SpliceVitCtx();
    c->vit = v; c->B = B; c->H = H; c->W = W; c->need_grad = need_grad;
    c->T = 1 + (H / p) * (W / p);
    c->Tld = (c->T + 31) / 32 * 32;
    c->rows = B * c->Tld;
    const size_t rows = c->rows;
    const int L = v->depth;
    int rc = SPLICE_OK;
#define A(ptr, n) if (rc == SPLICE_OK) rc = dev_alloc(c->allocs, &(ptr), (size_t)(n))
    A(c->pos_eff, (size_t)c->Tld * D);
    A(c->patches, rows * pp3);
    c->xs.resize(L + 1); c->xmid.resize(L); c->mean1.resize(L); c->rstd1.resize(L); c->mean2.resize(L); c->rstd2.resize(L);
    c->qkv.resize(L); c->attn_out.resize(L); c->hpre.resize(L); c->lse.resize(L);
    for (int l = 0; l <= L; ++l) A(c->xs[l], rows * D);
    for (int l = 0; l < L; ++l) {
        A(c->xmid[l], rows * D);
        A(c->mean1[l], rows); A(c->rstd1[l], rows); A(c->mean2[l], rows); A(c->rstd2[l], rows);
        A(c->qkv[l], rows * 3 * D); A(c->attn_out[l], rows * D);
        A(c->hpre[l], rows * Hd);
        A(c->lse[l], (size_t)B * v->heads * c->Tld);
    }
    A(c->qkv_last_f32, rows * 3 * D);
    A(c->qkvT_last, rows * 3 * D);
    A(c->ln_out, rows * D);
    A(c->cls_attn, (size_t)B * D); A(c->cls_ln, (size_t)B * D); A(c->cls_h, (size_t)B * Hd); A(c->cls_probs, (size_t)B * v->heads * c->Tld);
    A(c->cls_slabs, (size_t)16 * B * Hd);
    if (need_grad) { A(c->cls_dh, (size_t)B * Hd); A(c->cls_dout, (size_t)B * D); A(c->cls_dln, (size_t)B * D); }
    A(c->hact, rows * Hd);
    if (need_grad) {
        A(c->g, rows * D); A(c->g_bf, rows * D); A(c->dh, rows * Hd); A(c->dln, rows * D * 4);   // dln: up to 4 split-K slabs
        A(c->dout, rows * D); A(c->dqkv, rows * 3 * D);
        A(c->delta, (size_t)B * v->heads * c->Tld);
        A(c->dpatches, rows * pp3);
    }
#undef A
    if (rc != SPLICE_OK) {
        for (void* q : c->allocs) (void)hipFree(q);
        delete c;
        return rc;
    }
    SPLICE_LAUNCH(pos_eff_kernel, dim3(grid_n((size_t)c->Tld * D)), dim3(256), 0, s, c->pos_eff, pos_TD, v->cls, v->pe.b, c->T, c->Tld, D);
    *out = c;
    return SPLICE_OK;
}

void splice_vit_ctx_destroy(void* ctx) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c) return;
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
}

int splice_vit_ctx_info(void* ctx, int* T, int* Tld, int* rows) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c) return SPLICE_ERR_ARG;
    if (T) *T = c->T;
    if (Tld) *Tld = c->Tld;
    if (rows) *rows = c->rows;
    return SPLICE_OK;
}

// on != 0: behind the QKV projection of the TOP block only the [CLS] row of every pass is computed (attention for one query,
// proj / LayerNorm / fc1 / fc2 on one row) -- all the Splice losses read of that block besides its keys (util/losses.py:90).
// Block-output rows other than [CLS] are then undefined for layer depth-1, and a gradient injected at that block output
// must be zero outside the [CLS] rows.  The fused step switches its contexts to this mode; the extractor API never does.
int splice_vit_ctx_set_top_cls_only(void* ctx, int on) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c) return SPLICE_ERR_ARG;
    c->top_cls_only = on ? 1 : 0;
    return SPLICE_OK;
}

// mode bit 0: this context's QKV, fc1 and fc2 forward projections run on the fp8 MFMA (e4m3 LayerNorm outputs with per-token
// scales, e4m3 GELU output, e4m3 weights with per-channel scales); bit 1 (needs bit 0): its attention forward too (Q K^T and
// P V from unscaled e4m3 copies of q, k, v written by the QKV projection).  0 = bf16.  Needs splice_vit_enable_fp8 on the engine.
int splice_vit_ctx_set_fp8(void* ctx, int mode) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c || mode < 0 || mode > 3 || mode == 2) return SPLICE_ERR_ARG;
    if (mode && !c->vit->fp8) { splice_set_error("splice_vit_ctx_set_fp8: the engine has no e4m3 weights (splice_vit_enable_fp8 first)"); return SPLICE_ERR_STATE; }
    const size_t rows = c->rows;
    if (mode && !c->ln_out8) {
        RC(dev_alloc(c->allocs, &c->ln_out8, rows * c->vit->dim));
        RC(dev_alloc(c->allocs, &c->ln_scale, rows));
        RC(dev_alloc(c->allocs, &c->hact8, rows * c->vit->hidden));
    }
    if ((mode & 2) && !c->qkv8) {
        RC(dev_alloc(c->allocs, &c->qkv8, rows * 3 * c->vit->dim));
        RC(dev_alloc(c->allocs, &c->qkvT8, rows * 3 * c->vit->dim));
    }
    c->fp8 = mode;
    return SPLICE_OK;
}

int splice_vit_ctx_dims(void* ctx, int* B, int* H, int* W, int* D, int* depth, int* heads, int* patch) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c) return SPLICE_ERR_ARG;
    if (B) *B = c->B;
    if (H) *H = c->H;
    if (W) *W = c->W;
    if (D) *D = c->vit->dim;
    if (depth) *depth = c->vit->depth;
    if (heads) *heads = c->vit->heads;
    if (patch) *patch = c->vit->patch;
    return SPLICE_OK;
}

// Forward over the ctx's batch.  img: fp32 [B][3][H][W]; normalize != 0 applies the ImageNet
// Normalize of util/losses.py:19 on the fly (input in [0,1]); == 0 expects a normalised image
// (what VitExtractor receives, models/extractor.py:81).
int splice_vit_forward_ex(void* ctx, const float* img, int normalize, int grad_pass_begin, splice_stream_t stream) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c) return SPLICE_ERR_ARG;
    return splice_vit_forward_passes(ctx, img, normalize, grad_pass_begin, 0, c->B, stream);
}
int splice_vit_forward(void* ctx, const float* img, int normalize, splice_stream_t stream) {
    return splice_vit_forward_ex(ctx, img, normalize, 0, stream);
}
// grad_pass_begin: passes [0, grad_pass_begin) are targets that will never be differentiated -- tensors only the
// backward reads (the pre-GELU activations) are not stored for them.
// [pass_begin, pass_end): the passes (images) of the ctx batch this call computes.  Every kernel of the forward is
// row-parallel over the token matrix, so disjoint pass ranges of ONE ctx may run concurrently on different
// streams (the step runs the target passes beside the generator forward); img is always the full [B] batch.
int splice_vit_forward_passes(void* ctx, const float* img, int normalize, int grad_pass_begin, int pass_begin, int pass_end,
                              splice_stream_t stream) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c || !img || grad_pass_begin < 0 || grad_pass_begin > c->B || pass_begin < 0 || pass_end > c->B || pass_begin >= pass_end)
        return SPLICE_ERR_ARG;
    c->grad_pass_begin = grad_pass_begin;
    SpliceVit* v = c->vit;
    hipStream_t s = (hipStream_t)stream;
    const int D = v->dim, Hd = v->hidden, pp3 = 3 * v->patch * v->patch, L = v->depth;
    const int Bp = pass_end - pass_begin, R = Bp * c->Tld;
    const size_t r0 = (size_t)pass_begin * c->Tld;
    bf16_t* patches = c->patches + r0 * pp3;
    bf16_t* ln_out = c->ln_out + r0 * D;
    bf16_t* hact = c->hact + r0 * Hd;
    RC(patchify_launch(img + (size_t)pass_begin * 3 * c->H * c->W, patches, Bp, c->H, c->W, v->patch, c->Tld, normalize, s));
    {
        GemmEpi e = {};
        e.bias = v->pe.b; e.resid = c->pos_eff; e.ldr = D; e.resid_mod = c->Tld; e.out_f32 = c->xs[0] + r0 * D; e.ldo = D;
        RC(gemm_nt_launch(EPI_BIAS | EPI_RESID | EPI_OUT_F32, patches, pp3, v->pe.w, pp3, R, D, pp3, e, s));
    }
    for (int l = 0; l < L; ++l) {
        const LayerW& W = v->layers[l];
        float* x_in = c->xs[l] + r0 * D;
        float* x_mid = c->xmid[l] + r0 * D;
        const bool fp8 = (c->fp8 & 1) && c->ln_out8 && W.qkv.w8;
        const bool fp8_attn = fp8 && (c->fp8 & 2) && c->qkv8;
        {
            SpliceProfScope ps(10); SPLICE_DEV_REGION(1);
            if (fp8) RC(layernorm_fwd_fp8_launch(x_in, W.ln1_g, W.ln1_b, c->ln_out8 + r0 * D, c->ln_scale + r0, c->mean1[l] + r0, c->rstd1[l] + r0, R, D, 1e-6f, s));
            else RC(layernorm_fwd_launch(x_in, W.ln1_g, W.ln1_b, ln_out, c->mean1[l] + r0, c->rstd1[l] + r0, R, D, 1e-6f, s));
        }
        {
            GemmEpi e = {};
            e.bias = W.qkv.b; e.out_bf = c->qkv[l] + r0 * 3 * D; e.ldbf = 3 * D;
            unsigned fl = EPI_BIAS | EPI_OUT_BF;
            if (l == L - 1) {   // + the transposed copy for the [CLS]-row kernels and the fp32 keys of the loss terms
                fl |= EPI_OUT_T | EPI_COLS_F32;
                e.out_bf_t = c->qkvT_last + r0; e.ldt = c->rows;
                e.out_f32_cols = c->qkv_last_f32 + r0 * 3 * D; e.ld_cols = 3 * D; e.col_lo = 0; e.col_hi = 3 * D;
            }
            SpliceProfScope ps(l == L - 1 ? 0 : 2); SPLICE_DEV_REGION(2);
            if (fp8) {   // e4m3 LayerNorm output (per-token scale) x e4m3 weights (per-channel scale) on the fp8 MFMA
                e.row_scale = c->ln_scale + r0; e.col_scale = W.qkv.w8_scale;
                if (fp8_attn && !(l == L - 1 && c->top_cls_only)) {   // e4m3 copies of q, k, v for the fp8 attention forward of this layer
                    fl |= EPI_OUT_F8 | EPI_OUT_F8T;
                    e.out_f8 = c->qkv8 + r0 * 3 * D; e.ld8 = 3 * D; e.out_f8_t = c->qkvT8 + r0; e.ldt8 = c->rows;
                }
                RC(gemm_nt_fp8_launch(fl | EPI_SCALE_RC, c->ln_out8 + r0 * D, D, W.qkv.w8, D, R, 3 * D, D, e, s));
            } else {
                RC(gemm_nt_launch(fl, ln_out, D, W.qkv.w, D, R, 3 * D, D, e, s));
            }
        }
        if (l == L - 1 && c->top_cls_only) {
            SPLICE_DEV_REGION(19);
            // the tail of the top block on the [CLS] row of every pass only (vit_cls.hip): M = passes, rows Tld apart
            const int rs = c->Tld * D, hs = c->Tld * Hd;   // element strides between the [CLS] rows of consecutive passes
            bf16_t* cattn = c->cls_attn + (size_t)pass_begin * D;
            bf16_t* cln = c->cls_ln + (size_t)pass_begin * D;
            bf16_t* ch = c->cls_h + (size_t)pass_begin * Hd;
            RC(attn_cls_fwd_launch(c->qkv[l] + r0 * 3 * D, c->qkvT_last + r0, c->rows, Bp, c->T, c->Tld, D, v->heads, v->qfold ? kLn2 : 0.125f, cattn,
                                   c->cls_probs + (size_t)pass_begin * v->heads * c->Tld, s));
            // The GEMMs of this tail have M = passes rows: their run time is the serial K walk of a workgroup, not the rows, so
            // they run split-K over many workgroups (plain fp32 slabs) and the next kernel of the chain sums the slabs.
            const size_t sstr = (size_t)c->B * Hd;                  // slab stride (floats): room for the widest output
            float* slabs = c->cls_slabs + (size_t)pass_begin * Hd;  // this call's rows inside every slab
            auto ks_of = [](int K) { int n = K / 64, k = 16; while (k > 1 && n % k) --k; return k; };
            {
                GemmEpi e = {};
                e.out_f32 = slabs; e.ldo = D; e.ksplit = ks_of(D); e.slab_stride = (long long)sstr;
                RC(gemm_nt_launch(EPI_OUT_F32, cattn, D, W.proj.w, D, Bp, D, D, e, s));
                // x_mid rows = bias + x_in rows + slabs, then LayerNorm of those rows
                RC(ln_rows_fwd_launch(x_mid, rs, W.ln2_g, W.ln2_b, cln, D, c->mean2[l] + r0, c->rstd2[l] + r0, c->Tld, Bp, D, 1e-6f, slabs, e.ksplit, sstr,
                                      W.proj.b, x_in, rs, s));
            }
            {
                GemmEpi e = {};
                e.out_f32 = slabs; e.ldo = Hd; e.ksplit = ks_of(D); e.slab_stride = (long long)sstr;
                RC(gemm_nt_launch(EPI_OUT_F32, cln, D, W.fc1.w, D, Bp, Hd, D, e, s));
                const int lo = c->grad_pass_begin - pass_begin;   // first row (= pass) whose pre-activation is kept
                RC(rows_finish_launch(1, slabs, e.ksplit, sstr, Bp, Hd, W.fc1.b, nullptr, 0, nullptr, 0, ch, c->need_grad ? c->hpre[l] + r0 * Hd : nullptr, nullptr, hs,
                                      lo > 0 ? lo : 0, s));
            }
            {
                GemmEpi e = {};
                e.out_f32 = slabs; e.ldo = D; e.ksplit = ks_of(Hd); e.slab_stride = (long long)sstr;
                RC(gemm_nt_launch(EPI_OUT_F32, ch, Hd, W.fc2.w, Hd, Bp, D, Hd, e, s));
                RC(rows_finish_launch(0, slabs, e.ksplit, sstr, Bp, D, W.fc2.b, x_mid, rs, c->xs[l + 1] + r0 * D, rs, nullptr, nullptr, nullptr, 0, 0, s));
            }
            continue;
        }
        {
            AttnArgs a = {};
            a.qkv = c->qkv[l] + r0 * 3 * D; a.B = Bp; a.T = c->T; a.Tld = c->Tld; a.D = D; a.H = v->heads;
            a.scale = v->qfold ? kLn2 : 0.125f; a.qfold = v->qfold; a.out = c->attn_out[l] + r0 * D; a.lse = c->lse[l] + (size_t)pass_begin * v->heads * c->Tld;
            if (fp8_attn) { a.qkv8 = c->qkv8 + r0 * 3 * D; a.qkvT8 = c->qkvT8 + r0; a.ldt8 = c->rows; }
            SpliceProfScope ps(3); SPLICE_DEV_REGION(3);
            RC(attn_fwd_launch(&a, s));
        }
        {
            GemmEpi e = {};
            e.bias = W.proj.b; e.resid = x_in; e.ldr = D; e.out_f32 = x_mid; e.ldo = D;
            SpliceProfScope ps(9); SPLICE_DEV_REGION(4);
            RC(gemm_nt_launch(EPI_BIAS | EPI_RESID | EPI_OUT_F32, c->attn_out[l] + r0 * D, D, W.proj.w, D, R, D, D, e, s));
        }
        {
            SpliceProfScope ps(10); SPLICE_DEV_REGION(1);
            if (fp8) RC(layernorm_fwd_fp8_launch(x_mid, W.ln2_g, W.ln2_b, c->ln_out8 + r0 * D, c->ln_scale + r0, c->mean2[l] + r0, c->rstd2[l] + r0, R, D, 1e-6f, s));
            else RC(layernorm_fwd_launch(x_mid, W.ln2_g, W.ln2_b, ln_out, c->mean2[l] + r0, c->rstd2[l] + r0, R, D, 1e-6f, s));
        }
        {
            GemmEpi e = {};
            e.bias = W.fc1.b; e.out_bf = hact; e.ldbf = Hd; e.out_pre = c->need_grad ? c->hpre[l] + r0 * Hd : nullptr; e.ldp = Hd;
            const long lo = (long)c->grad_pass_begin * c->Tld - (long)r0;   // first local row whose pre-activation is kept
            e.pre_row_lo = lo > 0 ? (int)lo : 0;
            SpliceProfScope ps(1); SPLICE_DEV_REGION(5);
            if (fp8) {   // e4m3 LayerNorm output x e4m3 weights; GELU(x) leaves as e4m3 (the operand of fc2), the pre-activation as bf16
                e.out_bf = nullptr; e.out_f8 = c->hact8 + r0 * Hd; e.ld8 = Hd;
                e.row_scale = c->ln_scale + r0; e.col_scale = W.fc1.w8_scale;
                RC(gemm_nt_fp8_launch(EPI_SCALE_RC | EPI_BIAS | EPI_GELU | EPI_OUT_F8, c->ln_out8 + r0 * D, D, W.fc1.w8, D, R, Hd, D, e, s));
            } else {
                RC(gemm_nt_launch(EPI_BIAS | EPI_GELU | EPI_OUT_BF, ln_out, D, W.fc1.w, D, R, Hd, D, e, s));
            }
        }
        {
            GemmEpi e = {};
            e.bias = W.fc2.b; e.resid = x_mid; e.ldr = D; e.out_f32 = c->xs[l + 1] + r0 * D; e.ldo = D;
            SpliceProfScope ps(4); SPLICE_DEV_REGION(6);
            if (fp8) {
                e.row_scale = nullptr; e.col_scale = W.fc2.w8_scale;
                RC(gemm_nt_fp8_launch(EPI_SCALE_RC | EPI_BIAS | EPI_RESID | EPI_OUT_F32, c->hact8 + r0 * Hd, Hd, W.fc2.w8, Hd, R, D, Hd, e, s));
            } else {
                RC(gemm_nt_launch(EPI_BIAS | EPI_RESID | EPI_OUT_F32, hact, Hd, W.fc2.w, Hd, R, D, Hd, e, s));
            }
        }
    }
    c->forward_done = 1;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { splice_set_error("splice_vit_forward: %s", hipGetErrorString(e)); return SPLICE_ERR_HIP; }
    return SPLICE_OK;
}

// Device pointers of the tensors the reference's hooks expose (valid until the next forward):
// kind 0: block output l  fp32 [rows][D]      (models/extractor.py:56-60; l = depth-1 gives the CLS source)
// kind 1: raw qkv      l  bf16 [rows][3D]     (models/extractor.py:68-72)
// kind 2: attention out l bf16 [rows][D]      (pre-projection; the hooked 'patch_imd' is proj of it)
// kind 3: last-layer qkv  fp32 [rows][3D]
// kind 4: lse          l  fp32 [B][H][Tld]
// kind 5: embedded tokens fp32 [rows][D]
// kind 6: raw qkv l, transposed  bf16 [3D][rows]   (row = feature, column = pass * Tld + token)
int splice_vit_get_tensor(void* ctx, int kind, int layer, void** out) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c || !out || layer < 0 || layer >= c->vit->depth) return SPLICE_ERR_ARG;
    switch (kind) {
        case 0: *out = c->xs[layer + 1]; break;
        case 1: *out = c->qkv[layer]; break;
        case 2: *out = c->attn_out[layer]; break;
        case 3: *out = c->qkv_last_f32; break;
        case 4: *out = c->lse[layer]; break;
        case 5: *out = c->xs[0]; break;
        case 6: if (layer != c->vit->depth - 1) return SPLICE_ERR_ARG; *out = c->qkvT_last; break;   // only the last layer keeps a transposed copy
        case 7: *out = c->qkv[layer]; break;   // as kind 1; splice_vit_read_tensor copies it AS STORED (q columns pre-scaled by splice_vit_qscale)
        default: return SPLICE_ERR_ARG;
    }
    return SPLICE_OK;
}

// Copy one of the tensors above into caller memory (device to device), `bytes` from its start.
int splice_vit_read_tensor(void* ctx, int kind, int layer, void* dst, size_t bytes, splice_stream_t stream) {
    void* src = nullptr;
    RC(splice_vit_get_tensor(ctx, kind, layer, &src));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (c->vit->qfold && (kind == 1 || kind == 3)) {   // the caller sees q, not q * d^-1/2 log2(e)
        const int D = c->vit->dim;
        const size_t rows = bytes / ((size_t)3 * D * (kind == 1 ? 2 : 4));
        if (kind == 1) hipLaunchKernelGGL(scale_cols_bf16_kernel, dim3(grid_n(rows * D)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst, 3 * D, (int)rows, D, 1.0f / c->vit->qscale);
        else hipLaunchKernelGGL(scale_cols_f32_kernel, dim3(grid_n(rows * D)), dim3(256), 0, (hipStream_t)stream, (float*)dst, 3 * D, (int)rows, D, 1.0f / c->vit->qscale);
    }
    return SPLICE_OK;
}

// dgrad-only backward over passes [pass_begin, pass_end) of the last forward.
//   d_block[l]  fp32 [rows][D]   (full-batch layout) or NULL : gradient w.r.t. block output l
//   d_qkv[l]    fp32 [rows][3D]  or NULL                    : gradient w.r.t. the raw qkv of layer l
//   d_keys[l]   fp32 [rows][D]   or NULL                    : gradient w.r.t. the key columns only
// (arrays may be NULL as a whole).  d_img fp32 [B][3][H][W]: only the selected passes are written.
// Rows of padding tokens must carry zero gradient.
int splice_vit_backward(void* ctx, int pass_begin, int pass_end, const float* const* d_block, const float* const* d_qkv,
                        const float* const* d_keys, float* d_img, int normalize, splice_stream_t stream) {
    SpliceVitCtx* c = (SpliceVitCtx*)ctx;
    if (!c || !c->need_grad || !c->forward_done || pass_begin < c->grad_pass_begin || pass_end > c->B || pass_begin >= pass_end || !d_img) {
        splice_set_error("splice_vit_backward: bad ctx / range / no forward / ctx created without need_grad");
        return SPLICE_ERR_STATE;
    }
    SpliceVit* v = c->vit;
    hipStream_t s = (hipStream_t)stream;
    const int D = v->dim, Hd = v->hidden, pp3 = 3 * v->patch * v->patch, L = v->depth;
    const size_t r0 = (size_t)pass_begin * c->Tld;
    const int R = (pass_end - pass_begin) * c->Tld;
    const int Bp = pass_end - pass_begin;
    float* g = c->g + r0 * D;
    bf16_t* g_bf = c->g_bf + r0 * D;
    bool g_live = false;  // gradient stream known non-zero
    // the two long-K dgrad GEMMs (N = D) have too few tiles for the chip: split K in three (tools/gemm_splitk_bench.py) or two,
    // LayerNorm-backward adds the slabs
    const size_t slab = (size_t)c->rows * D;
    // The split is chosen from the tiles of ONE pass, not of the range: split-K changes the summation order, and the
    // gradient of an image must not depend on how many images (pairs) share the backward.
    const int tiles1 = cdiv(c->Tld, 64) * cdiv(D, 64);
    const int ks = (tiles1 * 3 <= 640 && D % 192 == 0) ? 3 : (tiles1 <= 640 && D % 128 == 0) ? 2 : 1;
    for (int l = L - 1; l >= 0; --l) {
        const LayerW& W = v->layers[l];
        const float* db = d_block ? d_block[l] : nullptr;
        const float* dq = d_qkv ? d_qkv[l] : nullptr;
        const float* dk = d_keys ? d_keys[l] : nullptr;
        if (db) {
            SPLICE_LAUNCH(grad_stream_kernel, dim3(grid_n((size_t)R * D)), dim3(256), 0, s, g, g_bf, db + r0 * D, (size_t)R * D, g_live ? 0 : 1);
            g_live = true;
        }
        if (!g_live && !dq && !dk) continue;
        bf16_t* dqkv = c->dqkv + r0 * 3 * D;
        const float* g_after_mlp = nullptr;  // g_in for LN1 backward
        if (g_live && l == L - 1 && c->top_cls_only) {
            SPLICE_DEV_REGION(20);
            // the forward ran the tail of this block on the [CLS] rows only: so does the backward (the gradient injected at the
            // block output must be zero outside the [CLS] rows -- the Splice appearance term, util/losses.py:90)
            const int rs = c->Tld * D, hs = c->Tld * Hd;
            bf16_t* cdh = c->cls_dh + (size_t)pass_begin * Hd;
            const size_t sstr = (size_t)c->B * Hd;
            float* slabs = c->cls_slabs + (size_t)pass_begin * Hd;
            auto ks_of = [](int K) { int n = K / 64, k = 16; while (k > 1 && n % k) --k; return k; };
            {   // fc2^T: split-K slabs, then dh = bf16(sum * gelu'(pre))
                GemmEpi e = {};
                e.out_f32 = slabs; e.ldo = Hd; e.ksplit = ks_of(D); e.slab_stride = (long long)sstr;
                RC(gemm_nt_launch(EPI_OUT_F32, g_bf, rs, W.fc2.wT, D, Bp, Hd, D, e, s));
                RC(rows_finish_launch(2, slabs, e.ksplit, sstr, Bp, Hd, nullptr, nullptr, 0, nullptr, 0, cdh, nullptr, c->hpre[l] + r0 * Hd, hs, 0, s));
            }
            {   // fc1^T: slabs summed by the LayerNorm backward of the [CLS] rows
                GemmEpi e = {};
                e.out_f32 = slabs; e.ldo = D; e.ksplit = ks_of(Hd); e.slab_stride = (long long)sstr;
                RC(gemm_nt_launch(EPI_OUT_F32, cdh, Hd, W.fc1.wT, Hd, Bp, D, Hd, e, s));
                RC(ln_rows_bwd_launch(slabs, D, c->xmid[l] + r0 * D, rs, W.ln2_g, c->mean2[l] + r0, c->rstd2[l] + r0, c->Tld, g, g_bf, Bp, D, e.ksplit, sstr, s));
            }
            {   // proj^T: slabs summed by the single-query attention backward
                GemmEpi e = {};
                e.out_f32 = slabs; e.ldo = D; e.ksplit = ks_of(D); e.slab_stride = (long long)sstr;
                RC(gemm_nt_launch(EPI_OUT_F32, g_bf, rs, W.proj.wT, D, Bp, D, D, e, s));
                RC(attn_cls_bwd_launch(c->qkv[l] + r0 * 3 * D, c->qkvT_last + r0, c->rows, Bp, c->T, c->Tld, D, v->heads, v->qfold ? kLn2 : 0.125f,
                                       c->cls_probs + (size_t)pass_begin * v->heads * c->Tld, slabs, e.ksplit, sstr, dqkv, s));
            }
            g_after_mlp = g;
        } else if (g_live) {
            // MLP branch
            {
                GemmEpi e = {};
                e.aux = c->hpre[l] + r0 * Hd; e.ldaux = Hd; e.out_bf = c->dh + r0 * Hd; e.ldbf = Hd;
                SpliceProfScope ps(11); SPLICE_DEV_REGION(7);
                RC(gemm_nt_launch(EPI_GELU_GRAD | EPI_OUT_BF, g_bf, D, W.fc2.wT, D, R, Hd, D, e, s));
            }
            {
                GemmEpi e = {};
                e.out_f32 = c->dln + r0 * D; e.ldo = D; e.ksplit = ks; e.slab_stride = (long long)slab;
                SpliceProfScope ps(5); SPLICE_DEV_REGION(8);
                RC(gemm_nt_launch(EPI_OUT_F32, c->dh + r0 * Hd, Hd, W.fc1.wT, Hd, R, D, Hd, e, s));
            }
            {
                SpliceProfScope ps(10); SPLICE_DEV_REGION(9);
                RC(layernorm_bwd_slabs_launch(c->dln + r0 * D, gemm_splitk_slabs(R, ks), slab, c->xmid[l] + r0 * D, W.ln2_g, c->mean2[l] + r0, c->rstd2[l] + r0, g, g, g_bf, R, D, s));
            }
            // attention branch
            {
                GemmEpi e = {};
                e.out_bf = c->dout + r0 * D; e.ldbf = D;   // (no transposed copy: the attention backward reads dO^T out of the dO tile)
                // delta = rowsum(dO * O) per (pass, head, query) for the attention backward, formed where dO is produced
                e.rd_other = c->attn_out[l] + r0 * D; e.ld_rd = D; e.rd_rows = c->Tld;
                e.rowdot = c->delta + (size_t)pass_begin * v->heads * c->Tld;
                SpliceProfScope ps(11); SPLICE_DEV_REGION(10);
                RC(gemm_nt_launch(EPI_OUT_BF | EPI_ROWDOT, g_bf, D, W.proj.wT, D, R, D, D, e, s));
            }
            {
                AttnArgs a = {};
                a.qkv = c->qkv[l] + r0 * 3 * D; a.B = Bp; a.T = c->T; a.Tld = c->Tld;
                a.D = D; a.H = v->heads; a.scale = v->qfold ? kLn2 : 0.125f; a.qfold = v->qfold; a.out = c->attn_out[l] + r0 * D;
                a.lse = c->lse[l] + (size_t)pass_begin * v->heads * c->Tld;
                a.dout = c->dout + r0 * D; a.doutT = nullptr; a.delta = c->delta + (size_t)pass_begin * v->heads * c->Tld; a.delta_ready = 1; a.dqkv = dqkv;
                SpliceProfScope ps(6); SPLICE_DEV_REGION(11);
                RC(attn_bwd_launch(&a, s));
            }
            g_after_mlp = g;
        } else {
            RC(dev_zero_launch(dqkv, (size_t)R * 3 * D * sizeof(bf16_t), s));
        }
        if (dq) SPLICE_LAUNCH(add_f32_into_bf16_kernel, dim3(grid_n((size_t)R * 3 * D)), dim3(256), 0, s, dqkv, 3 * D, 0, dq + r0 * 3 * D, 3 * D, R, 3 * D, v->qfold ? D : 0, 1.0f / v->qscale);
        if (dk) SPLICE_LAUNCH(add_f32_into_bf16_kernel, dim3(grid_n((size_t)R * D)), dim3(256), 0, s, dqkv, 3 * D, D, dk + r0 * D, D, R, D, 0, 1.0f);
        {
            GemmEpi e = {};
            e.out_f32 = c->dln + r0 * D; e.ldo = D; e.ksplit = ks; e.slab_stride = (long long)slab;
            SpliceProfScope ps(5); SPLICE_DEV_REGION(12);
            RC(gemm_nt_launch(EPI_OUT_F32, dqkv, 3 * D, W.qkv.wT, 3 * D, R, D, 3 * D, e, s));
        }
        {
            SpliceProfScope ps(10); SPLICE_DEV_REGION(9);
            RC(layernorm_bwd_slabs_launch(c->dln + r0 * D, gemm_splitk_slabs(R, ks), slab, c->xs[l] + r0 * D, W.ln1_g, c->mean1[l] + r0, c->rstd1[l] + r0, g_after_mlp, g, g_bf, R, D, s));
        }
        g_live = true;
    }
    if (!g_live) {
        RC(dev_zero_launch(d_img + (size_t)pass_begin * 3 * c->H * c->W, (size_t)Bp * 3 * c->H * c->W * sizeof(float), s));
        return SPLICE_OK;
    }
    {
        GemmEpi e = {};
        e.out_f32 = c->dpatches + r0 * pp3; e.ldo = pp3;
        RC(gemm_nt_launch(EPI_OUT_F32, g_bf, D, v->pe.wT, D, R, pp3, D, e, s));
    }
    RC(unpatchify_launch(c->dpatches + r0 * pp3, d_img + (size_t)pass_begin * 3 * c->H * c->W, Bp, c->H, c->W, v->patch, c->Tld, normalize, s));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { splice_set_error("splice_vit_backward: %s", hipGetErrorString(e)); return SPLICE_ERR_HIP; }
    return SPLICE_OK;
}
}
