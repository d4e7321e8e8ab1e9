// One Splice optimisation step (train.py:51-80) as a single host call: zero_grad, Model.forward
// (models/model.py:12-25), LossG.forward with its lambda schedule (util/losses.py:34-72),
// backward and Adam -- with the de-duplicated ViT plan of SURVEY.md section 7: the four global
// passes [T(A_crop), T(B_crop), T(G(A_crop)), T(G(B_crop))] run as ONE batched forward (the
// reference runs six batch-1 forwards, x' and B' twice), the backward covers only the two
// generated images and only data gradients.  Nothing is cached across steps.
//
// P image pairs side by side (cfg.pairs): the reference optimises one pair per process (train.py:34-49); pairs share
// nothing but the frozen ViT, so P of them can ride the SAME launches -- the ViT context holds 4P passes
// [A'_0..A'_{P-1} | B'_0.. | x'_0.. | y'_0..], every GEMM / attention / LayerNorm launch covers P times the rows, the
// generator plans hold P independent generators (own parameter / gradient / Adam arenas, own BatchNorm statistics), the
// loss kernels take the pair as a grid dimension.  Every per-pair quantity is computed exactly as in a P = 1 step (no
// policy depends on P), so a pair's trajectory is bit-identical whichever batch it rides in.
#include <vector>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <deque>
#include <array>
#include <map>
#include <mutex>

#include "gen_kernels.h"
#include "kernels.h"

void splice_set_error(const char* fmt, ...);
extern "C" {
int splice_vit_ctx_info(void* ctx, int* T, int* Tld, int* rows);
int splice_vit_ctx_dims(void* ctx, int* B, int* H, int* W, int* D, int* depth, int* heads, int* patch);
int splice_vit_ctx_set_top_cls_only(void* ctx, int on);
int splice_vit_forward(void* ctx, const float* img, int normalize, splice_stream_t stream);
int splice_vit_forward_ex(void* ctx, const float* img, int normalize, int grad_pass_begin, splice_stream_t stream);
int splice_vit_get_tensor(void* ctx, int kind, int layer, void** out);
int splice_vit_backward(void* ctx, int pass_begin, int pass_end, const float* const* d_block, const float* const* d_qkv,
                        const float* const* d_keys, float* d_img, int normalize, splice_stream_t stream);
int splice_gen_forward_borrowed(void* plan, const float* params, const float* x, float* y, splice_stream_t stream);
int splice_gen_running_stats_update(void* const* plans, int n_plans, float* running, long long running_stride, float momentum,
                                    splice_stream_t stream);
int splice_vit_forward_passes(void* ctx, const float* img, int normalize, int grad_pass_begin, int pass_begin, int pass_end,
                              splice_stream_t stream);
int splice_gen_plan_dims(void* plan, int* N, int* H, int* W, long long* nparams);
int splice_gen_plan_resize(void* plan, int H, int W);
int splice_gen_forward(void* plan, const float* params, const float* x, float* y, splice_stream_t stream);
int splice_gen_backward(void* plan, const float* params, const float* dy, float* grads, int accumulate, splice_stream_t stream);
int splice_adam_step(float* params, float* grads, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                     int step, int zero_grad, splice_stream_t stream);
int splice_prof_active(void);
}

#define RC(x)                                                                                     \
    do {                                                                                          \
        int rc_ = (x);                                                                            \
        if (rc_ != SPLICE_OK) {                                                                   \
            splice_set_error("%s:%d %s failed (%d)", __FILE__, __LINE__, #x, rc_);                \
            return rc_;                                                                           \
        }                                                                                         \
    } while (0)
#define HIPCHK(x)                                                                                 \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            splice_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            return SPLICE_ERR_HIP;                                                                \
        }                                                                                         \
    } while (0)

enum { L_TOTAL = 0, L_GLOBAL_SSIM = 1, L_ENTIRE_SSIM = 2, L_ENTIRE_CLS = 3, L_GLOBAL_CLS = 4, L_GLOBAL_ID = 5 };

struct VitView {
    void* ctx = nullptr;
    int B = 0, H = 0, W = 0, D = 0, depth = 0, heads = 0, patch = 0, T = 0, Tld = 0, rows = 0;
    float* d_block = nullptr;   // [rows][D]
    float* d_keys = nullptr;    // [rows][D]
    float* imgs = nullptr;      // [B][3][H][W]
    float* d_imgs = nullptr;    // [B][3][H][W]
    std::vector<const float*> pb, pk;   // per-layer pointer tables
};

// signature of a captured graph: what hipGraphExecUpdate needs to be equal between a capture and the executable it updates (see drop_graphs)
struct GraphSig {
    unsigned long long h = 1469598103934665603ull;   // FNV-1a over (device, node types / kernel functions, edges)
    size_t nodes = 0, edges = 0;
    void mix(unsigned long long v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } }
    bool operator<(const GraphSig& o) const { return h != o.h ? h < o.h : nodes != o.nodes ? nodes < o.nodes : edges < o.edges; }
};

struct SpliceStep {
    splice_step_config cfg;
    int P = 1;                   // pairs optimised side by side -- or, in crops mode, the n_crops global crops of ONE pair (max of the two counts)
    int Pa = 1, Pb = 1;          // images of the A-crop / B-crop plans: P in pairs mode; global_A_crops_n_crops / global_B_crops_n_crops in crops mode
    int Pe = 1;                  // images of the entire-image branch: P in pairs mode, 1 in crops mode (netG(A) is one image there)
    int crops_mode = 0;          // the P slots are crops of one pair: one generator (batch-statistics plans), losses summed over the crops
    size_t astride = 0;          // floats between the pairs' parameter / gradient / moment arenas (P > 1, pairs mode)
    VitView vg, ve;
    void *plan_a = nullptr, *plan_b = nullptr, *plan_e = nullptr;   // generator plans: P independent images each
    int cropb_h = 0, cropb_w = 0;
    long long nparams = 0;
    float* gen_in = nullptr;     // [P][3][ha][wa]   A crops            (staged copies: a captured graph only sees own buffers)
    float* in_b = nullptr;       // [P][3][hb][wb]   B crops
    float* ent_in = nullptr;     // [P][3][He][We]   entire structure images
    float* gen_out = nullptr;    // [P][3][ha][wa]   x_global
    float* gen_out_b = nullptr;  // [P][3][hb][wb]   y_global
    float *d_gen_out = nullptr, *d_gen_out_b = nullptr;
    float* ent_out = nullptr;    // [P][3][He][We]
    float* d_ent_out = nullptr;
    void* ssim_ws = nullptr;     // SelfSimBatch workspace (shared by the global and the entire-image term: they run back to back)
    float* losses = nullptr;     // per pair: [8] raw per-term losses, then [8][lp] workgroup partials of each term
    size_t lp = 0, lstride = 0;  // partial slots per term; floats per pair
    std::vector<void*> allocs;
    int max_crop_h = 0, max_crop_w = 0;
    int* dev_t = nullptr;        // Adam step count on the device
    hipStream_t own_stream = nullptr;
    // Cross-stream events, a RING of sets: an eager step records its fork / join events twice and the next step records them again
    // ~2 ms later, while waits on the previous record may still be queued; every record site has an event of its own and a set is
    // reused only EV_RING steps later (the runtime's completion handler crashed in the release chain of such marker commands once per
    // ~30 train_model runs, see drop_graphs)
    enum { EV_IN, EV_OUT, EV_FORK, EV_JOIN, EV_GB, EV_FORK2, EV_JOIN2, EV_KINDS };
    static constexpr int EV_RING = 4;
    hipEvent_t evs[EV_RING][EV_KINDS] = {};
    int ev_slot = 0;
    hipEvent_t ev(int kind) const { return evs[ev_slot][kind]; }
    hipStream_t side_stream = nullptr;               // target-pass ViT forward beside the generator forward
    int overlap = 1;                                 // SPLICE_STEP_OVERLAP=0 serialises (debugging)
    int ablate = 0;                                  // always 0 in the product build.  Scratch builds (-DSPLICE_DEV_SWITCHES) read the SPLICE_STEP_ABLATE bitmask, TIMING experiments only (results are garbage):
                                                     // 1 skip G fwd, 2 skip G bwd, 4 skip ViT bwd, 8 skip target ViT fwd, 16 skip generated ViT fwd, 32 skip the y' chain of the ViT bwd
    struct StepGraph { hipGraphExec_t ex; GraphSig sig; };
    std::map<int, StepGraph> graphs;
    long graph_updates = 0, graph_update_refusals = 0, graph_instantiations = 0;
    void* graph_ptrs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // arenas + the caller's losses / running buffers a captured graph is bound to
    float* losses_out = nullptr;                     // this step's destination of the [P][8] loss values (written by total_loss_kernel)
    int graph_crops[4] = {0, 0, 0, 0};
    float* grads_b = nullptr;                        // gradient arena(s) of the B-crop plan (added to `grads` inside Adam)
    int shape_repeats = 0;                           // consecutive steps with the same arenas and crop sizes
    int repeat_step = -1;                            // step_idx the count above was last advanced / reset for (several partial-phase calls of ONE step count once)
    int use_graph = 1;
    int dbg_sync = 0;
    int ssim_id_on = 0;          // lambda_global_ssim / lambda_global_identity switched on (util/losses.py:35-37)
    int skip_adam = 0, accumulate = 0;   // splice_step_set_mode: leave the summed gradient in `grads` (optionally += ) and do not update
    float* running = nullptr;    // BatchNorm running statistics arena(s) of the caller (null: not tracked)
    long long running_stride = 0;
    int phases = 7;              // splice_step_set_phases: 1 generator forward, 2 ViT forward / losses / ViT backward, 4 generator backward (+ Adam)
    SpliceStep* leader = nullptr;   // the handle whose staged inputs / generator outputs this one reads and whose image gradients it adds to
};

static void drop_graphs(SpliceStep* st);
static bool create_events(SpliceStep* st) {
    for (auto& set : st->evs)
        for (hipEvent_t& e : set)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    return true;
}

template <class T>
static int salloc(SpliceStep* st, T** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(T) + 256) != hipSuccess) {
        splice_set_error("splice_step: hipMalloc of %zu bytes failed", n * sizeof(T));
        return SPLICE_ERR_NOMEM;
    }
    st->allocs.push_back(q);
    *p = (T*)q;
    return SPLICE_OK;
}

static int view_init(SpliceStep* st, VitView& v, void* ctx, int want_B) {
    v.ctx = ctx;
    RC(splice_vit_ctx_dims(ctx, &v.B, &v.H, &v.W, &v.D, &v.depth, &v.heads, &v.patch));
    RC(splice_vit_ctx_info(ctx, &v.T, &v.Tld, &v.rows));
    // Of the top block the losses read the keys and the [CLS] row (util/losses.py:90); cfg.top_cls_only skips the rest of it
    // (vit_cls.hip): +2.7 % / +5.2 % pair-steps/s at 4 / 8 pairs per GPU, +2.7 % at one pair once the few-row kernels of the tail
    // request their operands in one round trip, DESIGN.md section 8.
    RC(splice_vit_ctx_set_top_cls_only(ctx, st->cfg.top_cls_only));
    if (v.B != want_B) { splice_set_error("splice_step_create: ViT context has batch %d, need %d", v.B, want_B); return SPLICE_ERR_ARG; }
    RC(salloc(st, &v.d_block, (size_t)v.rows * v.D));
    RC(salloc(st, &v.d_keys, (size_t)v.rows * v.D));
    RC(salloc(st, &v.imgs, (size_t)v.B * 3 * v.H * v.W));
    RC(salloc(st, &v.d_imgs, (size_t)v.B * 3 * v.H * v.W));
    // rows no loss kernel ever writes (padding tokens, non-[CLS] rows of d_block) must read as zero gradient
    if (hipMemset(v.d_block, 0, (size_t)v.rows * v.D * sizeof(float)) != hipSuccess || hipMemset(v.d_keys, 0, (size_t)v.rows * v.D * sizeof(float)) != hipSuccess)
        return SPLICE_ERR_HIP;
    v.pb.assign(v.depth, nullptr);
    v.pk.assign(v.depth, nullptr);
    v.pb[v.depth - 1] = v.d_block;
    v.pk[v.depth - 1] = v.d_keys;
    return SPLICE_OK;
}

// blockIdx.x = pair.  raw_k = fixed-order sum of term k's workgroup partials (no float atomics anywhere: replicas are
// bit-reproducible); total = sum_k lambda_k * raw_k   (util/losses.py:53-71)
// n_slots > 1 (crops mode, grid 1): the terms of the n_crops crops are added in crop order (util/losses.py:75-82 `loss +=`).
__global__ __launch_bounds__(320) void total_loss_kernel(float* lbase, size_t lstride, int lp, float w_ssim, float w_essim, float w_ecls, float w_cls,
                                                         float w_id, float* out8, int n_slots) {
    __shared__ float raw[8];
    float* l = lbase + (size_t)blockIdx.x * lstride;
    const int k = 1 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave k-1 owns term k (5 waves)
    float tot = 0.f;
    for (int slot = 0; slot < n_slots; ++slot) {
        const float* part = l + (size_t)slot * lstride + 8 + (size_t)k * lp;
        float acc = 0.f;
        for (int i = lane; i < lp; i += 64) acc += part[i];
        tot += wave_sum(acc);
    }
    const float acc = tot;
    if (lane == 0) raw[k] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t <= 5; ++t) l[t] = raw[t];
        l[L_TOTAL] = w_ssim * raw[L_GLOBAL_SSIM] + w_essim * raw[L_ENTIRE_SSIM] + w_ecls * raw[L_ENTIRE_CLS] + w_cls * raw[L_GLOBAL_CLS] + w_id * raw[L_GLOBAL_ID];
        if (out8) {   // the caller's losses buffer, written here instead of by a copy behind the step
            float* o = out8 + (size_t)blockIdx.x * 8;
            o[0] = l[L_TOTAL];
            for (int t = 1; t <= 5; ++t) o[t] = raw[t];
            o[6] = 0.f; o[7] = 0.f;
        }
    }
}
// All per-step inputs in ONE eager launch in front of the graph replay (three copies + the Adam step count were four
// launches with ~10-30 us of host/queue gaps between them): up to three fp32 buffers and one int.
struct StageArgs { const float* src[3]; float* dst[3]; unsigned long long n[3]; int* ip; int iv; };
__global__ __launch_bounds__(256) void stage_inputs_kernel(StageArgs a) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!a.src[k]) continue;
        const size_t n4 = a.n[k] / 4;   // vector body + scalar tail
        const float4* s4 = reinterpret_cast<const float4*>(a.src[k]);
        float4* d4 = reinterpret_cast<float4*>(a.dst[k]);
        const bool vec = ((reinterpret_cast<size_t>(a.src[k]) | reinterpret_cast<size_t>(a.dst[k])) & 15) == 0;
        if (vec) {
            for (size_t i = gid; i < n4; i += stride) d4[i] = s4[i];
            for (size_t i = n4 * 4 + gid; i < a.n[k]; i += stride) a.dst[k][i] = a.src[k][i];
        } else {
            for (size_t i = gid; i < a.n[k]; i += stride) a.dst[k][i] = a.src[k][i];
        }
    }
    if (gid == 0) *a.ip = a.iv;
}
static float* loss_part(SpliceStep* st, int slot) { return st->losses + 8 + (size_t)slot * st->lp; }   // pair 0; pair p at + p * lstride

// `n_img` images [3][h][w] -> [3][oh][ow] each (contiguous batches on both sides)
static int place_images(const float* src, int h, int w, float* dst, int oh, int ow, int n_img, hipStream_t s) {
    if (src == dst) return SPLICE_OK;   // identity Resize with the producer writing the ViT context's image slot itself (step_ptrs)
    if (h == oh && w == ow) {   // Resize returns its input when the shorter edge already matches
        RC(dev_copy_launch(dst, src, (size_t)n_img * 3 * h * w * sizeof(float), s));
        return SPLICE_OK;
    }
    return resize_bilinear_fwd_launch(src, dst, 3 * n_img, h, w, oh, ow, s);
}
static int unplace_grads(const float* dsrc, int oh, int ow, float* ddst, int h, int w, int n_img, hipStream_t s) {
    if (dsrc == ddst) return SPLICE_OK;
    if (h == oh && w == ow) {
        RC(dev_copy_launch(ddst, dsrc, (size_t)n_img * 3 * h * w * sizeof(float), s));
        return SPLICE_OK;
    }
    return resize_bilinear_bwd_launch(dsrc, ddst, 3 * n_img, h, w, oh, ow, s);
}

// Where the step's images live.  With an identity Resize (crop size == ViT input size: the benchmark's 224 x 224 pair, and
// every size-matched crop of train_model) the staged inputs, the generator outputs and their gradients ARE the image slots
// of the ViT context -- the generator reads and writes them in place, no copy kernels on either chain (4 + 2 launches per
// step, 3 of them on the critical chain).  Otherwise the private buffers + resize_bilinear_* as before.
struct StepPtrs { float *a_in, *b_in, *e_in, *x, *y, *xe, *dx, *dy, *dxe; };
static StepPtrs step_ptrs(SpliceStep* st) {
    const splice_step_config& c = st->cfg;
    VitView& vg = st->vg;
    const size_t vimg = (size_t)3 * vg.H * vg.W;
    const int pA = 0, pB = st->Pa, pX = st->Pa + st->Pb, pY = 2 * st->Pa + st->Pb;
    const bool ida = c.crop_h == vg.H && c.crop_w == vg.W, idb = st->cropb_h == vg.H && st->cropb_w == vg.W;
    StepPtrs p;
    p.a_in = ida ? vg.imgs + pA * vimg : st->gen_in;
    p.x = ida ? vg.imgs + pX * vimg : st->gen_out;
    p.dx = ida ? vg.d_imgs + pX * vimg : st->d_gen_out;
    p.b_in = idb ? vg.imgs + pB * vimg : st->in_b;
    p.y = idb ? vg.imgs + pY * vimg : st->gen_out_b;
    p.dy = idb ? vg.d_imgs + pY * vimg : st->d_gen_out_b;
    p.e_in = st->ent_in; p.xe = st->ent_out; p.dxe = st->d_ent_out;
    if (st->plan_e) {
        VitView& ve = st->ve;
        if (c.ent_h == ve.H && c.ent_w == ve.W) {
            const size_t eimg = (size_t)3 * ve.H * ve.W;
            p.e_in = ve.imgs; p.xe = ve.imgs + st->Pe * eimg; p.dxe = ve.d_imgs + st->Pe * eimg;
        }
    }
    return p;
}

// fp32 keys of pass b (view into the last layer's raw qkv): pointer + leading dimension 3D
static const float* keys_ptr(const VitView& v, const float* qkv_last, int pass) { return qkv_last + (size_t)pass * v.Tld * 3 * v.D + v.D; }

// The structure term of P pairs (util/losses.py:74-83): targets = passes [pass_tgt, pass_tgt + P), generated = passes
// [pass_x, pass_x + P) of view v.  `b` is carved over st->ssim_ws for v's token count.
static int ssim_batch(SpliceStep* st, VitView& v, int pass_tgt, int pass_x, int count, float lambda, int slot, SelfSimBatch* b) {
    bf16_t *qkv = nullptr, *qkvT = nullptr;
    RC(splice_vit_get_tensor(v.ctx, 1, v.depth - 1, (void**)&qkv));
    RC(splice_vit_get_tensor(v.ctx, 6, v.depth - 1, (void**)&qkvT));
    selfsim_batch_carve(st->ssim_ws, v.T, v.D, count, b);
    const size_t pass_rows = (size_t)v.Tld * 3 * v.D;
    b->ldk = 3 * v.D; b->ldt = v.rows; b->k_pstride = pass_rows; b->kT_pstride = (size_t)v.Tld;
    b->k_tgt = qkv + (size_t)pass_tgt * pass_rows + v.D;
    b->k_x = qkv + (size_t)pass_x * pass_rows + v.D;
    b->kT_x = qkvT + (size_t)v.D * v.rows + (size_t)pass_x * v.Tld;
    b->loss_part = loss_part(st, slot); b->part_pstride = st->lstride;
    b->dk = v.d_keys + (size_t)pass_x * v.Tld * v.D; b->dk_pstride = (size_t)v.Tld * v.D; b->lddk = v.D;
    b->eps = 1e-8f;
    b->fp8 = st->cfg.fp8_selfsim && v.D % 128 == 0;
    b->loss_scale = 1.0f / ((float)v.T * (float)v.T);
    b->e_scale = 4.0f * lambda * b->loss_scale;
    return SPLICE_OK;
}

extern "C" {

int splice_step_create(const splice_step_config* cfg, void* vit_ctx_global, void* vit_ctx_entire, void* gen_plan_a, void* gen_plan_b,
                       void* gen_plan_entire, void** out) {
    if (!cfg || !vit_ctx_global || !gen_plan_a || !gen_plan_b || !out) return SPLICE_ERR_ARG;
    SpliceStep* st = new SpliceStep();
    st->cfg = *cfg;
    if ((cfg->n_crops > 1 || cfg->n_crops_b > 1) && cfg->pairs > 1) { splice_set_error("splice_step_create: n_crops > 1 and pairs > 1 cannot be combined"); delete st; return SPLICE_ERR_ARG; }
    st->crops_mode = cfg->n_crops > 1 || cfg->n_crops_b > 1;
    // crops mode: the reference zips the crop lists (util/losses.py:76,87,98) -- the structure term runs over the A crops, the
    // identity term over the B crops, the appearance term over min(nA, nB) pairs -- while netG's BatchNorm sees every crop of
    // its call (models/model.py:15-23): the two plans may hold different numbers of images
    const int nb_crops = cfg->n_crops_b > 0 ? cfg->n_crops_b : cfg->n_crops;
    if (st->crops_mode && (nb_crops < 1 || nb_crops > 8 || cfg->n_crops > 8)) { splice_set_error("splice_step_create: 1..8 crops per side"); delete st; return SPLICE_ERR_ARG; }
    const int Pa = st->Pa = st->crops_mode ? cfg->n_crops : cfg->pairs > 1 ? cfg->pairs : 1;
    const int Pb = st->Pb = st->crops_mode ? nb_crops : Pa;
    const int P = st->P = Pa > Pb ? Pa : Pb;
    const int Pe = st->Pe = st->crops_mode ? 1 : P;
    st->max_crop_h = cfg->crop_h; st->max_crop_w = cfg->crop_w;
    st->cropb_h = cfg->crop_h; st->cropb_w = cfg->crop_w;
    int rc = SPLICE_OK;
    auto fail = [&](int code) { for (void* q : st->allocs) (void)hipFree(q); delete st; return code; };
    if ((rc = view_init(st, st->vg, vit_ctx_global, 2 * (Pa + Pb))) != SPLICE_OK) return fail(rc);
    if (st->vg.H != cfg->vit_h || st->vg.W != cfg->vit_w) { splice_set_error("splice_step_create: global ViT context shape mismatch"); return fail(SPLICE_ERR_ARG); }
    int n, h, w;
    for (int k = 0; k < 2; ++k) {
        void* plan = k ? gen_plan_b : gen_plan_a;
        const int want = k ? Pb : Pa;
        if ((rc = splice_gen_plan_dims(plan, &n, &h, &w, &st->nparams)) != SPLICE_OK) return fail(rc);
        if (n != want || h != cfg->crop_h || w != cfg->crop_w) { splice_set_error("splice_step_create: the crop generator plans must hold %d / %d image(s) at the crop size", Pa, Pb); return fail(SPLICE_ERR_ARG); }
    }
    st->plan_a = gen_plan_a; st->plan_b = gen_plan_b;
    if (P > 1 && !st->crops_mode) {
        if (cfg->arena_stride < st->nparams) { splice_set_error("splice_step_create: pairs > 1 needs arena_stride >= the parameter count"); return fail(SPLICE_ERR_ARG); }
        st->astride = (size_t)cfg->arena_stride;
    }
    const size_t crop = (size_t)3 * cfg->crop_h * cfg->crop_w;
    for (float** q : {&st->gen_in, &st->in_b, &st->gen_out, &st->gen_out_b, &st->d_gen_out, &st->d_gen_out_b})
        if ((rc = salloc(st, q, P * crop)) != SPLICE_OK) return fail(rc);
    if ((rc = salloc(st, &st->grads_b, st->astride ? P * st->astride : (size_t)st->nparams)) != SPLICE_OK) return fail(rc);
    // the backward writes nparams floats per pair; the <= 63 floats of padding between two arenas are read by the whole-range
    // add / Adam and must never hold garbage
    if (hipMemset(st->grads_b, 0, (st->astride ? P * st->astride : (size_t)st->nparams) * sizeof(float)) != hipSuccess) return fail(SPLICE_ERR_HIP);
    int Tmax = st->vg.T;
    if (cfg->ent_h > 0) {
        if (!vit_ctx_entire || !gen_plan_entire) { splice_set_error("splice_step_create: entire-image branch needs its ViT context and generator plan"); return fail(SPLICE_ERR_ARG); }
        if ((rc = view_init(st, st->ve, vit_ctx_entire, 2 * Pe)) != SPLICE_OK) return fail(rc);
        if (st->ve.H != cfg->ent_vit_h || st->ve.W != cfg->ent_vit_w) { splice_set_error("splice_step_create: entire ViT context shape mismatch"); return fail(SPLICE_ERR_ARG); }
        if ((rc = splice_gen_plan_dims(gen_plan_entire, &n, &h, &w, nullptr)) != SPLICE_OK) return fail(rc);
        if (n != Pe || h != cfg->ent_h || w != cfg->ent_w) { splice_set_error("splice_step_create: the entire generator plan must hold %d image(s) at the entire-image size", Pe); return fail(SPLICE_ERR_ARG); }
        st->plan_e = gen_plan_entire;
        const size_t ent = (size_t)3 * cfg->ent_h * cfg->ent_w;
        for (float** q : {&st->ent_in, &st->ent_out, &st->d_ent_out})
            if ((rc = salloc(st, q, Pe * ent)) != SPLICE_OK) return fail(rc);
        if (st->ve.T > Tmax) Tmax = st->ve.T;
    }
    char* wsb = nullptr;
    if ((rc = salloc(st, &wsb, selfsim_batch_ws_bytes(Tmax, st->vg.D, P))) != SPLICE_OK) return fail(rc);
    st->ssim_ws = wsb;
    {   // per-term partial slots: the MSE kernels use up to SPLICE_MSE_PARTIALS workgroups, the structure term one per upper-triangular tile
        const size_t nt = (Tmax + 63) / 64, tri = nt * (nt + 1) / 2;
        st->lp = tri > SPLICE_MSE_PARTIALS ? tri : SPLICE_MSE_PARTIALS;
        st->lstride = 8 + 8 * st->lp;
    }
    if ((rc = salloc(st, &st->losses, P * st->lstride)) != SPLICE_OK) return fail(rc);
    if ((rc = salloc(st, &st->dev_t, 4)) != SPLICE_OK) return fail(rc);
    if (const char* e = getenv("SPLICE_STEP_GRAPH")) st->use_graph = atoi(e);
    if (const char* e = getenv("SPLICE_STEP_SYNC")) st->dbg_sync = atoi(e);
    if (const char* e = getenv("SPLICE_STEP_OVERLAP")) st->overlap = atoi(e);
#ifdef SPLICE_DEV_SWITCHES   // scratch builds only (make DEV=1): the work-skipping timing switch is not part of the product library
    if (const char* e = getenv("SPLICE_STEP_ABLATE")) st->ablate = atoi(e);
#endif
    if (hipStreamCreateWithFlags(&st->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&st->side_stream, hipStreamNonBlocking) != hipSuccess ||
        !create_events(st)) {
        splice_set_error("splice_step_create: stream/event creation failed");
        return fail(SPLICE_ERR_HIP);
    }
    *out = st;
    return SPLICE_OK;
}

void splice_step_destroy(void* h) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st) return;
    drop_graphs(st);
    if (st->own_stream) { (void)hipStreamSynchronize(st->own_stream); (void)hipStreamDestroy(st->own_stream); }
    if (st->side_stream) { (void)hipStreamSynchronize(st->side_stream); (void)hipStreamDestroy(st->side_stream); }
    for (auto& set : st->evs)
        for (hipEvent_t e : set)
            if (e) (void)hipEventDestroy(e);
    for (void* q : st->allocs) (void)hipFree(q);
    delete st;
}

// Pointers to the generator outputs of the last step (device, valid until the next run):
// which 0: x_global [P][3][a_h][a_w], 1: x_entire [P][3][ent_h][ent_w], 2: y_global [P][3][b_h][b_w]
int splice_step_output(void* h, int which, float** out) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st || !out) return SPLICE_ERR_ARG;
    const StepPtrs p = step_ptrs(st);
    *out = which == 0 ? p.x : which == 1 ? p.xe : which == 2 ? p.y : nullptr;
    return *out ? SPLICE_OK : SPLICE_ERR_STATE;
}

// Crop sizes of the NEXT steps (<= the creation size): the reference's data feed draws size ~ U(min_cover*h, h) per step
// and image (data/transforms.py:21-22, data/Dataset.py:66-67); the ViT input size (after Resize) is unchanged for square
// crops.  All pairs of a batch share the two sizes.
int splice_step_set_crops(void* h, int a_h, int a_w, int b_h, int b_w) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st) return SPLICE_ERR_ARG;
    if (a_h > st->max_crop_h || a_w > st->max_crop_w || b_h > st->max_crop_h || b_w > st->max_crop_w) { splice_set_error("splice_step_set_crops: larger than the creation size"); return SPLICE_ERR_ARG; }
    RC(splice_gen_plan_resize(st->plan_a, a_h, a_w));
    RC(splice_gen_plan_resize(st->plan_b, b_h, b_w));
    st->cfg.crop_h = a_h; st->cfg.crop_w = a_w; st->cropb_h = b_h; st->cropb_w = b_w;
    return SPLICE_OK;
}

// skip_adam != 0: the step stops after backward -- `grads` holds the gradient of this step's loss (+= its previous content
// when accumulate != 0) and the parameters are untouched; the caller sums several losses that way (e.g. the same crops seen
// at several ViT input scales) and applies splice_adam_step once.
int splice_step_set_mode(void* h, int skip_adam, int accumulate) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st || (accumulate && !skip_adam)) return SPLICE_ERR_ARG;
    if (st->skip_adam != (skip_adam ? 1 : 0) || st->accumulate != (accumulate ? 1 : 0)) drop_graphs(st);
    st->skip_adam = skip_adam ? 1 : 0; st->accumulate = accumulate ? 1 : 0;
    return SPLICE_OK;
}
// Run only part of a step: phases is a mask of 1 = generator forward (and input staging), 2 = ViT forward, losses and ViT backward
// down to the gradient of the generated images, 4 = generator backward (+ Adam unless splice_step_set_mode says otherwise).
// With `leader` (another step handle, same image shapes; phases must be 2) this handle is a FOLLOWER: it reads the leader's
// staged inputs and generator outputs instead of running the generator, and ADDS its image gradients to the leader's.  That
// is the several-scales step (BASELINE configs[4]) with one generator pass: leader 1|2, followers 2, leader 4 -- the generator
// backward is linear in the image gradient, so backpropagating the sum equals summing the backpropagations.
int splice_step_set_phases(void* h, int phases, void* leader) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st || phases <= 0 || phases > 7 || leader == h || (leader && phases != 2) || (leader && ((SpliceStep*)leader)->leader)) return SPLICE_ERR_ARG;
    if (st->leader != (SpliceStep*)leader) drop_graphs(st);
    st->phases = phases;
    st->leader = (SpliceStep*)leader;
    return SPLICE_OK;
}

// BatchNorm running statistics (models/unet/common.py:95-96): when set, every step applies the momentum-0.1 update of its
// netG calls in the reference's order (A_global, A on entire steps, B_global; models/model.py:15-23) to the caller's
// buffer arena(s) (layout: splice_gen_buffer_info; pair p at running + p * stride).  NULL switches tracking off.
int splice_step_set_running_stats(void* h, float* running, long long stride) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st || (running && st->P > 1 && stride <= 0)) return SPLICE_ERR_ARG;
    st->running = running; st->running_stride = stride;
    return SPLICE_OK;
}

// ---- the launch sequence of one step (capturable: no allocation, no host sync, internal pointers only)
static int step_body(SpliceStep* st, float* params, float* grads, float* m, float* v, bool ssim_on, bool entire, hipStream_t s) {
    const splice_step_config& c = st->cfg;
    VitView& vg = st->vg;
    const int P = st->P, Pa = st->Pa, Pb = st->Pb, Pc = Pa < Pb ? Pa : Pb;   // Pc: pairs of the appearance term (zip of x' and B')
    const float l_ssim = ssim_on ? c.lambda_global_ssim : 0.f, l_id = ssim_on ? c.lambda_global_identity : 0.f;
    const float l_cls = c.lambda_global_cls;
    const float l_essim = entire ? c.lambda_entire_ssim : 0.f, l_ecls = entire ? c.lambda_entire_cls : 0.f;
    const size_t vimg = (size_t)3 * vg.H * vg.W;
    const StepPtrs ip = step_ptrs(st);
    // a follower (splice_step_set_phases) sees the leader's images and adds its image gradients to the leader's
    const StepPtrs src = st->leader ? step_ptrs(st->leader) : ip;
    const bool do_gf = (st->phases & 1) != 0, do_v = (st->phases & 2) != 0, do_gb = (st->phases & 4) != 0;
    const float* A_crop = src.a_in;
    const float* B_crop = src.b_in;
    const float* A_entire = src.e_in;
    auto to_leader = [&](float* lead, const float* own, size_t n, hipStream_t q) -> int {   // d(leader's image) += d(this scale's view of it)
        return st->leader ? add_f32_launch(lead, own, n, q) : SPLICE_OK;
    };
    // pass layout of the global context: [0, Pa) A'   [Pa, Pa + Pb) B'   then x' = G(A crops) (Pa passes), y' = G(B crops) (Pb passes)
    const int pA = 0, pB = Pa, pX = Pa + Pb, pY = 2 * Pa + Pb, pEnd = 2 * (Pa + Pb);
    // ---- the no-grad target passes A', B' (util/losses.py:79,91,101) do not depend on the generator: their ViT forward
    // runs on a side stream beside the generator forward (hundreds of small latency-bound launches that leave most
    // CUs idle); inside a capture this becomes a fork/join of the graph.  The instrumented (profiling) path stays serial.
    const bool overlap = st->overlap && !splice_prof_active();
    hipStream_t s2 = overlap ? st->side_stream : s;
    if (overlap) {
        HIPCHK(hipEventRecord(st->ev(SpliceStep::EV_FORK), s));
        HIPCHK(hipStreamWaitEvent(s2, st->ev(SpliceStep::EV_FORK), 0));
    }
    // global_transform (Resize -> Normalize; the Normalize is fused into the ViT patch gather).  The generator runs as one
    // plan per crop kind (A crops | B crops: the reference draws their sizes independently, data/Dataset.py:66-67);
    // G(B_crop) goes first on the side stream so that it runs beside G(A_crop) instead of behind it
    if (do_gf && !(st->ablate & 1)) {
        RC(splice_gen_forward_borrowed(st->plan_b, params, B_crop, ip.y, s2));
        if (overlap) HIPCHK(hipEventRecord(st->ev(SpliceStep::EV_GB), s2));
    }
    float *blk_g = nullptr, *qkv_g = nullptr;
    RC(splice_vit_get_tensor(vg.ctx, 0, vg.depth - 1, (void**)&blk_g));
    RC(splice_vit_get_tensor(vg.ctx, 3, vg.depth - 1, (void**)&qkv_g));
    const size_t passD = (size_t)vg.Tld * vg.D;
    SelfSimBatch sb = {};
    if (do_v) {
        RC(place_images(A_crop, c.crop_h, c.crop_w, vg.imgs + pA * vimg, vg.H, vg.W, Pa, s2));
        RC(place_images(B_crop, st->cropb_h, st->cropb_w, vg.imgs + pB * vimg, vg.H, vg.W, Pb, s2));
        if (!(st->ablate & 8)) RC(splice_vit_forward_passes(vg.ctx, vg.imgs, 1, pX, pA, pX, s2));
        // everything of the loss stage that does not need the generated images also runs here, off the critical path
        RC(dev_zero_launch(st->losses, (size_t)P * st->lstride * sizeof(float), s2));
        RC(dev_zero_launch(vg.d_block + pX * passD, (size_t)(Pa + Pb) * passD * sizeof(float), s2));
        RC(dev_zero_launch(vg.d_keys + pX * passD, (size_t)(Pa + Pb) * passD * sizeof(float), s2));
        if (l_ssim > 0.f) {   // target self-similarity S* of A' (util/losses.py:79)
            RC(ssim_batch(st, vg, pA, pX, Pa, l_ssim, L_GLOBAL_SSIM, &sb));
            RC(selfsim_target_launch(sb, s2));
        }
    }
    if (overlap) HIPCHK(hipEventRecord(st->ev(SpliceStep::EV_JOIN), s2));
    // ---- Model.forward: x_global = G(A_crop), y_global = G(B_crop) [, x_entire = G(A)]
    if (do_gf && !(st->ablate & 1)) {
        RC(splice_gen_forward_borrowed(st->plan_a, params, A_crop, ip.x, s));
        if (overlap) HIPCHK(hipStreamWaitEvent(s, st->ev(SpliceStep::EV_GB), 0));
    }
    if (do_v) {
        RC(place_images(src.x, c.crop_h, c.crop_w, vg.imgs + pX * vimg, vg.H, vg.W, Pa, s));
        RC(place_images(src.y, st->cropb_h, st->cropb_w, vg.imgs + pY * vimg, vg.H, vg.W, Pb, s));
        if (!(st->ablate & 16)) RC(splice_vit_forward_passes(vg.ctx, vg.imgs, 1, pX, pX, pEnd, s));
    }
    if (overlap) HIPCHK(hipStreamWaitEvent(s, st->ev(SpliceStep::EV_JOIN), 0));
    // ---- losses on the global batch
    if (do_v) {
        if (l_ssim > 0.f) RC(selfsim_loss_launch(sb, s));
        if (l_cls > 0.f)   // [CLS] of block 11, before the final LayerNorm (util/losses.py:90-93)
            RC(mse_batched_launch(blk_g + pX * passD, vg.D, passD, blk_g + pB * passD, vg.D, passD, 1, vg.D, 1.0f, l_cls, loss_part(st, L_GLOBAL_CLS),
                                  st->lstride, vg.d_block + pX * passD, vg.D, passD, Pc, s));
        if (l_id > 0.f)    // keys of y' against keys of B' (util/losses.py:96-105): mean over h*T*d = T*D
            RC(mse_batched_launch(keys_ptr(vg, qkv_g, pY), 3 * vg.D, 3 * passD, keys_ptr(vg, qkv_g, pB), 3 * vg.D, 3 * passD, vg.T, vg.D, 1.0f, l_id,
                                  loss_part(st, L_GLOBAL_ID), st->lstride, vg.d_keys + pY * passD, vg.D, passD, Pb, s));
    }
    // ---- entire-image branch (every entire_every-th step): passes [0, Pe) A_entire', [Pe, 2 Pe) x_entire'
    const int Pe = st->Pe;
    if (entire) {
        VitView& ve = st->ve;
        const size_t eimg = (size_t)3 * ve.H * ve.W;
        if (do_gf) RC(splice_gen_forward_borrowed(st->plan_e, params, A_entire, ip.xe, s));
        if (do_v) {
            RC(place_images(A_entire, c.ent_h, c.ent_w, ve.imgs, ve.H, ve.W, Pe, s));
            RC(place_images(src.xe, c.ent_h, c.ent_w, ve.imgs + Pe * eimg, ve.H, ve.W, Pe, s));
            RC(splice_vit_forward_ex(ve.ctx, ve.imgs, 1, Pe, s));
            float* blk_e = nullptr;
            RC(splice_vit_get_tensor(ve.ctx, 0, ve.depth - 1, (void**)&blk_e));
            const size_t epassD = (size_t)ve.Tld * ve.D;
            RC(dev_zero_launch(ve.d_block + Pe * epassD, (size_t)Pe * epassD * sizeof(float), s));
            RC(dev_zero_launch(ve.d_keys + Pe * epassD, (size_t)Pe * epassD * sizeof(float), s));
            if (l_essim > 0.f) {
                SelfSimBatch se = {};
                RC(ssim_batch(st, ve, 0, Pe, Pe, l_essim, L_ENTIRE_SSIM, &se));
                RC(selfsim_target_launch(se, s));
                RC(selfsim_loss_launch(se, s));
            }
            if (l_ecls > 0.f)   // target is the B_global crop's CLS (util/losses.py:60; with n_crops > 1 the zip pairs x_entire with the FIRST crop)
                RC(mse_batched_launch(blk_e + Pe * epassD, ve.D, epassD, blk_g + pB * passD, vg.D, passD, 1, ve.D, 1.0f, l_ecls, loss_part(st, L_ENTIRE_CLS),
                                      st->lstride, ve.d_block + Pe * epassD, ve.D, epassD, Pe, s));
        }
    }
    // ---- backward (train.py:78): ViT dgrad for the generated images only, then the generator
    // the x' and y' passes are independent chains until the generator: one per stream (every launch of a
    // dependent chain pays ~8 us of fixed latency; two chains in flight hide each other's)
    bool loss_summed = false;
    auto sum_losses = [&](hipStream_t q) {
        if (!do_v) return;
        SPLICE_LAUNCH(total_loss_kernel, dim3(st->crops_mode ? 1 : P), dim3(320), 0, q, st->losses, st->lstride, (int)st->lp, l_ssim, l_essim, l_ecls, l_cls, l_id,
                           st->losses_out, st->crops_mode ? P : 1);
    };
    auto track_running = [&](hipStream_t q) -> int {   // BatchNorm running statistics in the reference's call order
        if (!st->running || (st->ablate & 1) || !do_gf) return SPLICE_OK;
        void* plans[3];
        int np = 0;
        plans[np++] = st->plan_a;
        if (entire) plans[np++] = st->plan_e;
        plans[np++] = st->plan_b;
        return splice_gen_running_stats_update(plans, np, st->running, st->running_stride, 0.1f, q);
    };
    const float* adam_g2 = nullptr;
    if (!(st->ablate & 4)) {   // (the same launches whether or not the second stream is used: results are bit-identical)
        if (overlap) {
            HIPCHK(hipEventRecord(st->ev(SpliceStep::EV_FORK2), s));
            HIPCHK(hipStreamWaitEvent(s2, st->ev(SpliceStep::EV_FORK2), 0));
        }
        if (do_v) {
            if (!(st->ablate & 32)) RC(splice_vit_backward(vg.ctx, pY, pEnd, vg.pb.data(), nullptr, vg.pk.data(), vg.d_imgs, 1, s2));
            RC(unplace_grads(vg.d_imgs + pY * vimg, vg.H, vg.W, ip.dy, st->cropb_h, st->cropb_w, Pb, s2));
            RC(to_leader(src.dy, ip.dy, (size_t)Pb * 3 * st->cropb_h * st->cropb_w, s2));
        }
        // each chain continues into its own generator plan (own gradient arena: no cross-chain accumulation)
        if (do_gb && !(st->ablate & 2)) RC(splice_gen_backward(st->plan_b, params, ip.dy, st->grads_b, 0, s2));
        if (overlap) {
            // the reported loss values and the BatchNorm bookkeeping depend on nothing downstream: they run at the tail of
            // the side chain instead of between the generator backward and Adam on the critical one
            sum_losses(s2);
            loss_summed = true;
            RC(track_running(s2));
            HIPCHK(hipEventRecord(st->ev(SpliceStep::EV_JOIN2), s2));
        }
        if (do_v) {
            RC(splice_vit_backward(vg.ctx, pX, pY, vg.pb.data(), nullptr, vg.pk.data(), vg.d_imgs, 1, s));
            RC(unplace_grads(vg.d_imgs + pX * vimg, vg.H, vg.W, ip.dx, c.crop_h, c.crop_w, Pa, s));
            RC(to_leader(src.dx, ip.dx, (size_t)Pa * 3 * c.crop_h * c.crop_w, s));
        }
        if (do_gb && !(st->ablate & 2)) RC(splice_gen_backward(st->plan_a, params, ip.dx, grads, st->accumulate, s));
        if (overlap) HIPCHK(hipStreamWaitEvent(s, st->ev(SpliceStep::EV_JOIN2), 0));
        // grads = g(A) + g(B): folded into the Adam kernel on ordinary steps; a separate add when the entire-image branch
        // still has to accumulate into the sum (same association order either way)
        if (do_gb && !(st->ablate & 2)) {
            if (entire || st->skip_adam) RC(add_f32_launch(grads, st->grads_b, st->astride ? P * st->astride : (size_t)st->nparams, s));
            else adam_g2 = st->grads_b;
        }
    }
    if (entire) {
        VitView& ve = st->ve;
        const size_t eimg = (size_t)3 * ve.H * ve.W;
        if (do_v) {
            RC(splice_vit_backward(ve.ctx, Pe, 2 * Pe, ve.pb.data(), nullptr, ve.pk.data(), ve.d_imgs, 1, s));
            RC(unplace_grads(ve.d_imgs + Pe * eimg, ve.H, ve.W, ip.dxe, c.ent_h, c.ent_w, Pe, s));
            RC(to_leader(src.dxe, ip.dxe, (size_t)Pe * 3 * c.ent_h * c.ent_w, s));
        }
        if (do_gb) RC(splice_gen_backward(st->plan_e, params, ip.dxe, grads, 1, s));
    }
    if (!loss_summed) { sum_losses(s); RC(track_running(s)); }
    // ---- optimizer.step() (train.py:79) over every pair's arena; Adam's step count (>= 1) is read from the device at execution time
    SPLICE_DEV_REGION(21);
    if (do_gb && !st->skip_adam) RC(adam_launch_dev(params, grads, m, v, st->astride ? P * st->astride : (size_t)st->nparams, c.lr, c.beta1, c.beta2, c.eps, st->dev_t, 0, s, adam_g2));
    return SPLICE_OK;
}

// Graph executables are NEVER destroyed while the process lives (round 6).  History: hipGraphExecDestroy behind a stream synchronize freed objects
// under the runtime's asynchronous completion-handler thread -- a segmentation fault inside that thread once per ~15 train_model runs of 2000 steps
// (profiles/r04_graph_drop_crash.txt); round 5 moved dropped executables to a process-wide pool and updated them in place (hipGraphExecUpdate), but
// keyed the pool by a few configuration words and still sent executables whose update the runtime refused to a delayed destroy once 64 had piled up
// (VERDICT r5 "What's weak" #5).  Now the pool is keyed by what an in-place update actually requires to be equal:
//     the device, the node sequence (type and, for kernel nodes, the kernel FUNCTION of every node in the capture's order) and the edge list
// hashed from the captured graph itself.  Anything else a step can differ in (pointers, grids, lambdas, crop sizes) is kernel parameters, which the
// update rewrites.  Two captures with the same signature are the same launch sequence with the same dependencies, so an update of a pooled
// executable is not expected to be refused; if the runtime refuses one anyway it is parked for good (counted in splice_step_graph_stats[1]) --
// a bounded leak of one executable per refusal instead of a destroy.  The number of executables a process owns is bounded by
// (distinct launch sequences it has run) x (handles alive at once).
namespace {
std::mutex g_pool_mu;
std::map<GraphSig, std::vector<hipGraphExec_t>> g_spare;   // guarded by g_pool_mu
std::vector<hipGraphExec_t> g_parked;                       // executables whose update the runtime refused: kept, never launched again, never destroyed
}
static bool graph_signature(hipGraph_t g, GraphSig* sig) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    size_t nn = 0, ne = 0;
    if (hipGraphGetNodes(g, nullptr, &nn) != hipSuccess || hipGraphGetEdges(g, nullptr, nullptr, &ne) != hipSuccess) return false;
    std::vector<hipGraphNode_t> nodes(nn), from(ne), to(ne);
    if (nn && hipGraphGetNodes(g, nodes.data(), &nn) != hipSuccess) return false;
    if (ne && hipGraphGetEdges(g, from.data(), to.data(), &ne) != hipSuccess) return false;
    GraphSig s;
    s.nodes = nn; s.edges = ne;
    s.mix((unsigned long long)dev);
    std::map<hipGraphNode_t, unsigned> index;
    for (size_t i = 0; i < nn; ++i) {
        index[nodes[i]] = (unsigned)i;
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) return false;
        s.mix((unsigned long long)ty);
        if (ty == hipGraphNodeTypeKernel) {
            hipKernelNodeParams kp;
            if (hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess) return false;
            s.mix((unsigned long long)(uintptr_t)kp.func);
            s.mix(((unsigned long long)kp.blockDim.x << 32) ^ ((unsigned long long)kp.blockDim.y << 16) ^ kp.blockDim.z);   // (a workgroup shape is part of the kernel choice here)
        }
    }
    // edges as index pairs, order-independent (sum of per-edge hashes): the runtime may list them in any order
    unsigned long long eh = 0;
    for (size_t i = 0; i < ne; ++i) {
        GraphSig e1;
        e1.mix(index[from[i]]); e1.mix(index[to[i]]);
        eh += e1.h;
    }
    s.mix(eh);
    *sig = s;
    return true;
}
static void drop_graphs(SpliceStep* st) {
    if (st->graphs.empty()) return;
    // (no synchronize needed here: nothing is destroyed; the next user's update waits for the device)
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& kv : st->graphs) g_spare[kv.second.sig].push_back(kv.second.ex);
    st->graphs.clear();
}

// One step of every pair.  step_idx is the reference's data step counter (0-based, data/Dataset.py:57,63).
// params/grads/m/v: the generator arenas, [P][arena_stride] (one flat arena when P = 1).  A_crop / B_crop: [P][3][h][w]
// at the current crop sizes; A_entire [P][3][ent_h][ent_w] (may be NULL on steps where step_idx % entire_every != 0).
// losses_out: device fp32 [P][8], per pair {loss, loss_global_ssim, loss_entire_ssim, loss_entire_cls, loss_global_cls,
// loss_global_id_B, 0, 0} (inactive terms are 0).
// The launch sequence (~500 kernels) is captured once per regime (first step / ordinary / entire-image) into a hipGraph
// and replayed; inputs are staged into handle-owned buffers first so the graph only ever sees the same pointers.
int splice_step_run(void* h, float* params, float* grads, float* m, float* v, const float* A_crop, const float* B_crop,
                    const float* A_entire, int step_idx, float* losses_out, splice_stream_t stream) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st || !params || !grads || !m || !v || !A_crop || !B_crop || step_idx < 0) return SPLICE_ERR_ARG;
#ifdef SPLICE_DEV_SWITCHES
    ++g_splice_dev_steps;
#endif
    hipStream_t caller = (hipStream_t)stream;
    const splice_step_config& c = st->cfg;
    const int P = st->P;
    st->ev_slot = (st->ev_slot + 1) % SpliceStep::EV_RING;   // this call's set of cross-stream events
    // ---- lambda schedule (util/losses.py:34-44)
    if (step_idx == c.cls_warmup) st->ssim_id_on = 1;
    const bool entire = c.ent_h > 0 && c.entire_every > 0 && (step_idx % c.entire_every == 0);
    if (entire && !A_entire) { splice_set_error("splice_step_run: step %d needs the entire structure image", step_idx); return SPLICE_ERR_ARG; }
    // Graphs pay off only while the launch sequence repeats: with random crop sizes (data/transforms.py:21) nearly every
    // step has new shapes, and re-capturing + instantiating ~600 nodes costs as much as the step itself (9.9 vs 5.8 ms
    // measured).  So a step whose arenas / crop sizes differ from the previous step's runs eagerly (same kernels, same
    // results) and a graph is captured only once the shapes have repeated (below).
    {
        void* ptrs[6] = {params, grads, m, v, losses_out, st->running};
        st->losses_out = losses_out;
        const int crops[4] = {c.crop_h, c.crop_w, st->cropb_h, st->cropb_w};
        if (memcmp(ptrs, st->graph_ptrs, sizeof(ptrs)) || memcmp(crops, st->graph_crops, sizeof(crops))) {
            drop_graphs(st);
            memcpy(st->graph_ptrs, ptrs, sizeof(ptrs));
            memcpy(st->graph_crops, crops, sizeof(crops));
            st->shape_repeats = 0;
            st->repeat_step = step_idx;
        } else if (step_idx != st->repeat_step) {
            // a step split into several calls (splice_step_set_phases: MultiScaleEngine runs phases 1|2, then 4, with identical
            // pointers and crop sizes) is ONE step: its later calls inherit the first call's eager / graph decision instead
            // of looking like a repeat and capturing a graph that the next step's new crop sizes drop again
            if (st->shape_repeats < 2) ++st->shape_repeats;
            st->repeat_step = step_idx;
        }
    }
    // (from the THIRD consecutive step with identical shapes on: under random crop sizes two equal steps in a row happen ~14 times
    // per 2000 steps, each time capturing ~600 nodes for ONE replay before the next size drops the graph again)
    const bool graph = st->use_graph && !splice_prof_active() && st->shape_repeats >= 2;
    const bool own = graph;
    hipStream_t s = caller;
    if (own) {   // graphs cannot be captured on the legacy default stream: run on the handle's own stream, fenced by events
        s = st->own_stream;
        HIPCHK(hipEventRecord(st->ev(SpliceStep::EV_IN), caller));
        HIPCHK(hipStreamWaitEvent(s, st->ev(SpliceStep::EV_IN), 0));
    }
    if (st->leader) {
        const SpliceStep* ld = st->leader;
        if (ld->P != P || ld->Pa != st->Pa || ld->Pb != st->Pb || ld->Pe != st->Pe || ld->cfg.crop_h != c.crop_h || ld->cfg.crop_w != c.crop_w ||
            ld->cropb_h != st->cropb_h || ld->cropb_w != st->cropb_w || ld->cfg.ent_h != c.ent_h || ld->cfg.ent_w != c.ent_w) {
            splice_set_error("splice_step_run: a follower's images must have the leader's shapes");
            return SPLICE_ERR_ARG;
        }
    }
    // ---- stage the inputs (eager; with the generator forward: the other phases of the step see the same staged images)
    if (st->phases & 1) {
        StageArgs sa = {};
        const StepPtrs ip = step_ptrs(st);
        sa.src[0] = A_crop; sa.dst[0] = ip.a_in; sa.n[0] = (size_t)st->Pa * 3 * c.crop_h * c.crop_w;
        sa.src[1] = B_crop; sa.dst[1] = ip.b_in; sa.n[1] = (size_t)st->Pb * 3 * st->cropb_h * st->cropb_w;
        if (entire) { sa.src[2] = A_entire; sa.dst[2] = ip.e_in; sa.n[2] = (size_t)st->Pe * 3 * c.ent_h * c.ent_w; }
        sa.ip = st->dev_t; sa.iv = step_idx + 1;
        SPLICE_LAUNCH(stage_inputs_kernel, dim3(128 * (P > 4 ? 4 : P)), dim3(256), 0, s, sa);
    }
    // Steps whose shapes differ from the previous step's (the reference's random crop sizes) are launched eagerly.  Capturing EVERY such step into a
    // rotation of executables updated in place was built in round 5, is bit-equal and SLOWER (272.1 -> 265.9 steps/s: recording ~600 nodes +
    // hipGraphExecUpdate costs more host time than ~600 launches; profiles/r05_graph_every_step.txt); it left the library in round 6.
    if (!graph) {
        RC(step_body(st, params, grads, m, v, st->ssim_id_on != 0, entire, s));
    } else {
        const int variant = (st->ssim_id_on ? 1 : 0) | (entire ? 2 : 0) | (st->phases << 2);
        auto it = st->graphs.find(variant);
        if (it == st->graphs.end()) {
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int rc = step_body(st, params, grads, m, v, st->ssim_id_on != 0, entire, s);
            const hipError_t ee = hipStreamEndCapture(s, &g);
            if (rc != SPLICE_OK || ee != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                if (rc == SPLICE_OK) splice_set_error("splice_step_run: graph capture failed (%s)", hipGetErrorString(ee));
                return rc != SPLICE_OK ? rc : SPLICE_ERR_HIP;
            }
            GraphSig sig;
            const bool have_sig = graph_signature(g, &sig);
            hipGraphExec_t ex = nullptr;
            hipGraphExec_t spare = nullptr;
            if (have_sig) {
                std::lock_guard<std::mutex> lk(g_pool_mu);
                auto sp = g_spare.find(sig);
                if (sp != g_spare.end() && !sp->second.empty()) { spare = sp->second.back(); sp->second.pop_back(); }
            }
            if (spare) {   // a retired executable of exactly this launch sequence on this device: update it in place
                hipGraphNode_t bad_node = nullptr;
                hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
                // (its last launch -- possibly by another handle -- is long finished when its regime captures again; the device-wide wait is for the
                // case of a worker that destroys and re-creates engines back to back)
                (void)hipDeviceSynchronize();
                if (hipGraphExecUpdate(spare, g, &bad_node, &res) == hipSuccess && res == hipGraphExecUpdateSuccess) {
                    ex = spare;
                    ++st->graph_updates;
                } else {
                    (void)hipGetLastError();
                    std::lock_guard<std::mutex> lk(g_pool_mu);
                    g_parked.push_back(spare);   // kept for good: no hipGraphExecDestroy in a running process
                    ++st->graph_update_refusals;
                }
            }
            if (!ex) {
                const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
                if (ei != hipSuccess) { (void)hipGraphDestroy(g); splice_set_error("splice_step_run: hipGraphInstantiate: %s", hipGetErrorString(ei)); return SPLICE_ERR_HIP; }
                ++st->graph_instantiations;
            }
            (void)hipGraphDestroy(g);
            if (!have_sig) { sig = GraphSig(); sig.mix((unsigned long long)(uintptr_t)ex); }   // (signature unavailable: a key of its own, never shared)
            it = st->graphs.emplace(variant, SpliceStep::StepGraph{ex, sig}).first;
        }
        HIPCHK(hipGraphLaunch(it->second.ex, s));
    }
    if (own) {
        HIPCHK(hipEventRecord(st->ev(SpliceStep::EV_OUT), s));
        HIPCHK(hipStreamWaitEvent(caller, st->ev(SpliceStep::EV_OUT), 0));
    }
    if (st->dbg_sync) HIPCHK(hipStreamSynchronize(s));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { splice_set_error("splice_step_run: %s", hipGetErrorString(e)); return SPLICE_ERR_HIP; }
    return SPLICE_OK;
}

// out[0..2] = captures that updated a pooled executable in place / updates the runtime refused (executable parked for good) / executables instantiated, by this handle
int splice_step_graph_stats(void* h, long long* out) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st || !out) return SPLICE_ERR_ARG;
    out[0] = st->graph_updates; out[1] = st->graph_update_refusals; out[2] = st->graph_instantiations;
    return SPLICE_OK;
}

// 1 = replay captured hipGraphs (default), 0 = launch every kernel eagerly
int splice_step_use_graph(void* h, int on) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st) return SPLICE_ERR_ARG;
    st->use_graph = on ? 1 : 0;
    return SPLICE_OK;
}
// 1 = the independent chains of a step run on two streams (default), 0 = everything on one stream.  The results are
// bit-identical either way (every kernel is deterministic and owns its outputs); captured graphs are dropped.
int splice_step_use_overlap(void* h, int on) {
    SpliceStep* st = (SpliceStep*)h;
    if (!st) return SPLICE_ERR_ARG;
    if (st->overlap != (on ? 1 : 0)) drop_graphs(st);
    st->overlap = on ? 1 : 0;
    return SPLICE_OK;
}
}
