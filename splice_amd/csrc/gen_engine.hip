// Generator engine: the reference's define_G() -> skip() network (models/networks.py:56-58,
// models/unet/skip.py:4-102) as a static plan of HIP kernel launches over pre-allocated
// buffers, forward and full backward (dgrad + wgrad + bias/BN grads).  Parameters and
// gradients live in ONE flat fp32 arena in the reference's parameters() order so the
// optimiser is a single fused launch and the Python facade can expose per-tensor views.
//
// "N" is the number of independent generator calls processed side by side (e.g. G(A_crop)
// and G(B_crop), models/model.py:15-23): BatchNorm statistics are per call (the reference
// runs batch 1), parameter gradients are summed over calls.
#include <string>
#include <vector>

#include "gen_kernels.h"

int dev_copy_launch(void* dst, const void* src, size_t bytes, hipStream_t s);

void splice_set_error(const char* fmt, ...);

#define RC(x)                                                                                     \
    do {                                                                                          \
        int rc_ = (x);                                                                            \
        if (rc_ != SPLICE_OK) {                                                                   \
            splice_set_error("%s:%d %s failed (%d)", __FILE__, __LINE__, #x, rc_);                \
            return rc_;                                                                           \
        }                                                                                         \
    } while (0)

// Architecture of a skip() network (models/unet/skip.py:4-11): define_G builds the default one; the feature-inversion
// experiment (inversion.py:21-25) asks for 6 scales, 7/7/5/5/3/3 filters, reflection padding and 32 input channels.
constexpr int MAXS = SPLICE_GEN_MAX_SCALES;
typedef splice_gen_arch GenArch;
static GenArch default_arch() {
    GenArch a = {};
    a.n_scales = 5; a.in_channels = 3; a.out_channels = 3; a.filter_skip = 1; a.reflect = 0;
    const int ch[5] = {16, 32, 64, 128, 128};
    for (int i = 0; i < 5; ++i) { a.down[i] = ch[i]; a.up[i] = ch[i]; a.skip[i] = 4; a.filter_down[i] = 3; a.filter_up[i] = 3; }
    return a;
}
static const float BN_EPS = 1e-5f;
static const float LRELU = 0.2f;

struct ParamTable {
    struct Entry { std::string name; size_t off, numel; };
    std::vector<Entry> entries;
    size_t total = 0;
    size_t add(const std::string& name, size_t numel) {
        entries.push_back({name, total, numel});
        total += numel;
        return entries.back().off;
    }
};

// conv(+bias) -> train-mode BN -> LeakyReLU unit (or BN alone when ks == 0)
struct Unit {
    int ks = 0, stride = 1, Cin = 0, Cout = 0, Hi = 0, Wi = 0, Ho = 0, Wo = 0;
    size_t w_off = 0, b_off = 0, g_off = 0, be_off = 0;   // offsets into the parameter arena
    size_t r_off = 0;                                     // offset of running_mean in the buffer arena (running_var at + Cout)
    bool has_bn = true;
    float slope = LRELU;
    // tensors (device): input, conv output y, activation output a (+ strides, may be channel slices)
    const float* in = nullptr; size_t in_ns = 0;
    float* y = nullptr; size_t y_ns = 0;
    float* out = nullptr; size_t out_ns = 0;
    float *mean = nullptr, *rstd = nullptr, *s1 = nullptr, *s2 = nullptr;
    // gradients
    float* d_out = nullptr; size_t d_out_ns = 0;   // grad w.r.t. `out` (provided by the consumer)
    float* dy = nullptr;                            // grad w.r.t. y (same shape/stride as y)
    float* d_in = nullptr; size_t d_in_ns = 0;      // grad w.r.t. `in` (null: not needed)
    int d_in_accumulate = 0;
    bool own_out = false;
    size_t wg_off = 0;   // this layer's region of the wgrad partial workspace (floats)
};

struct SpliceGen {
    GenArch arch;
    ParamTable table;
    ParamTable buffers;   // BatchNorm running statistics in state_dict order: "<bn>.running_mean" (C floats) then "<bn>.running_var" (C floats)
};

struct SpliceGenPlan {
    SpliceGen* gen = nullptr;
    int N = 0, H = 0, W = 0, need_grad = 0, maxH = 0, maxW = 0;
    size_t p_nstride = 0;                 // > 0: the N images are independent generators -- image n uses params / grads + n * p_nstride
    int batch_stats = 0;                  // != 0: ONE netG call on a batch of N images -- BatchNorm statistics over the whole batch
    int h[MAXS + 1], w[MAXS + 1];         // spatial size at scale i (h[0] = H)
    std::vector<void*> allocs;
    // per scale
    Unit u_skip[MAXS], u_da[MAXS], u_db[MAXS], u_cat[MAXS], u_up3[MAXS], u_up1[MAXS];
    float* cat[MAXS]; float* d_cat[MAXS]; // [N][skip+k][h][w]
    int kch[MAXS];
    bool chain[MAXS];                     // scale i: the skip branch's BatchNorm runs inside the concat BatchNorm's kernels (BnPre), forward and backward
    float* skip_ws[MAXS];                 // chained skip convolution: its split-K slabs stay here until the concat BatchNorm sums them (conv_ws is reused in between)
    size_t skip_ws_floats[MAXS];
    int skip_ks[MAXS];                    // slabs left by the last forward (<= 1: none, u_skip.y is complete)
    struct { const float* target = nullptr; const float* slabs = nullptr; int ksplit = 0, accumulate = 0; } pend;   // backward: slabs of the next BatchNorm's output gradient
    float* pad_scratch = nullptr;         // reflection padding: padded-domain data gradient of one layer (largest layer)
    float* head_y = nullptr;              // unused (sigmoid fused)
    size_t head_w = 0, head_b = 0;
    float* d_head_pre = nullptr;          // [N][3][H][W]
    float* d_u0 = nullptr;                // grad w.r.t. u_0 (scale-0 output)
    float* wgrad_ws = nullptr;
    float* conv_ws = nullptr;             // split-K scratch of the small deep convolutions
    size_t conv_ws_floats = 0;
    size_t head_wg_off = 0, head_bias_off = 0;
    WgradReduceAll red;                   // filled during a backward, consumed by its single reduce launch
    WgradBatchPair wg;                    // every layer's weight-gradient work of a backward: launched after the dgrad chain
    float* out_copy = nullptr;            // generator output kept for the sigmoid backward
    float* x_copy = nullptr;              // private copy of the input (the caller may free x after forward)
    const float* x_in = nullptr;          // input of the last forward: x_copy, or the caller's buffer (splice_gen_forward_borrowed)
    const float* y_saved = nullptr;       // output of the last forward: out_copy, or the caller's buffer
    int forward_saved = 0;
};

static void build_table(const GenArch& A, ParamTable& t, size_t* offs /* [MAXS][6 units][4] */, size_t* head, ParamTable* bufs = nullptr,
                        size_t* roffs /* [MAXS][6] */ = nullptr) {
    // a conv is nn.Sequential([ReflectionPad2d,] Conv2d): the Conv2d is child "1" behind a padder, child "0" otherwise (models/unet/common.py:113-124)
    const std::string cv = A.reflect ? ".1" : ".0";
    auto conv = [&](const std::string& n, int co, int ci, int k, size_t* o) {
        o[0] = t.add(n + ".weight", (size_t)co * ci * k * k);
        o[1] = t.add(n + ".bias", co);
    };
    size_t* cur_r = nullptr;   // where the next bn() records its running-buffer offset
    auto bn = [&](const std::string& n, int c, size_t* o) {
        o[2] = t.add(n + ".weight", c);
        o[3] = t.add(n + ".bias", c);
        if (bufs) {
            const size_t r = bufs->add(n + ".running_mean", c);
            bufs->add(n + ".running_var", c);
            if (cur_r) *cur_r = r;
        }
    };
    std::string prefix[MAXS];
    for (int i = 1; i < A.n_scales; ++i) prefix[i] = prefix[i - 1] + "1.1.7.";
    // registration order is recursive (skip.py:46-99): skip, down-a, down-b of scale i, then all of scale
    // i+1, then cat-BN, up3, up1 of scale i.
    struct Rec {
        static void go(const GenArch& A, const std::string& cv, int i, int cin, ParamTable& t, size_t* offs, std::string* prefix,
                       decltype(conv)& conv, decltype(bn)& bn, size_t*& cur_r, size_t* roffs) {
            const std::string p = prefix[i];
            size_t* o = offs + (size_t)i * 6 * 4;
            auto at = [&](int unit) { cur_r = roffs ? roffs + i * 6 + unit : nullptr; };
            conv(p + "1.0.1" + cv, A.skip[i], cin, A.filter_skip, o + 0 * 4); at(0); bn(p + "1.0.2", A.skip[i], o + 0 * 4);
            conv(p + "1.1.1" + cv, A.down[i], cin, A.filter_down[i], o + 1 * 4); at(1); bn(p + "1.1.2", A.down[i], o + 1 * 4);
            conv(p + "1.1.4" + cv, A.down[i], A.down[i], A.filter_down[i], o + 2 * 4); at(2); bn(p + "1.1.5", A.down[i], o + 2 * 4);
            int k = A.down[i];
            if (i < A.n_scales - 1) { go(A, cv, i + 1, A.down[i], t, offs, prefix, conv, bn, cur_r, roffs); k = A.up[i + 1]; }
            at(3); bn(p + "2", A.skip[i] + k, o + 3 * 4);
            conv(p + "3" + cv, A.up[i], A.skip[i] + k, A.filter_up[i], o + 4 * 4); at(4); bn(p + "4", A.up[i], o + 4 * 4);
            conv(p + "6" + cv, A.up[i], A.up[i], 1, o + 5 * 4); at(5); bn(p + "7", A.up[i], o + 5 * 4);
        }
    };
    Rec::go(A, cv, 0, A.in_channels, t, offs, prefix, conv, bn, cur_r, roffs);
    conv("9" + cv, A.out_channels, A.up[0], 1, head);
}

template <class T>
static int palloc(SpliceGenPlan* p, T** ptr, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(T) + 256) != hipSuccess) {
        splice_set_error("generator plan: hipMalloc of %zu bytes failed", n * sizeof(T));
        return SPLICE_ERR_NOMEM;
    }
    p->allocs.push_back(q);
    *ptr = (T*)q;
    return SPLICE_OK;
}

// allocate a unit's tensors for the plan's MAXIMUM size (dims currently set on the unit)
static int unit_alloc(SpliceGenPlan* p, Unit& u, bool own_out) {
    const size_t N = p->N;
    const size_t plane = (size_t)u.Cout * u.Ho * u.Wo;
    if (u.ks) RC(palloc(p, &u.y, N * plane));
    RC(palloc(p, &u.mean, N * u.Cout)); RC(palloc(p, &u.rstd, N * u.Cout));
    RC(palloc(p, &u.s1, (size_t)bn_part_floats((int)N, u.Cout)));
    if (own_out) RC(palloc(p, &u.out, N * plane));
    if (p->need_grad) {
        RC(palloc(p, &u.dy, N * plane));
        if (own_out) RC(palloc(p, &u.d_out, N * plane));
    }
    u.own_out = own_out;
    return SPLICE_OK;
}

// (re)derive every size-dependent field for an H x W input (<= the size the plan was created for);
// buffers are laid out compactly inside their maximum-size allocations.
static void plan_configure(SpliceGenPlan* p, int H, int W) {
    p->H = H; p->W = W;
    p->h[0] = H; p->w[0] = W;
    const GenArch& A = p->gen->arch;
    const int S = A.n_scales;
    for (int i = 1; i <= S; ++i) { p->h[i] = (p->h[i - 1] + 1) / 2; p->w[i] = (p->w[i - 1] + 1) / 2; }
    const bool ng = p->need_grad;
    for (int i = 0; i < S; ++i) {
        const int cin = i == 0 ? A.in_channels : A.down[i - 1];
        const int hi = p->h[i], wi = p->w[i], hd = p->h[i + 1], wd = p->w[i + 1];
        const size_t catC = A.skip[i] + p->kch[i];
        const size_t cat_ns = catC * hi * wi;
        Unit &sk = p->u_skip[i], &da = p->u_da[i], &db = p->u_db[i], &ct = p->u_cat[i], &u3 = p->u_up3[i], &u1 = p->u_up1[i];
        auto dims = [](Unit& u, int Hi, int Wi, int Ho, int Wo) {
            u.Hi = Hi; u.Wi = Wi; u.Ho = Ho; u.Wo = Wo;
            u.y_ns = (size_t)u.Cout * Ho * Wo;
            if (u.own_out) { u.out_ns = u.y_ns; u.d_out_ns = u.y_ns; }
        };
        // the skip branch's BatchNorm rides in the concat BatchNorm's kernels where those own whole planes (BnPre)
        p->chain[i] = A.skip[i] > 0 && bn_pre_supported(hi * wi, p->N, p->p_nstride, p->batch_stats);
        p->skip_ks[i] = 1;
        dims(sk, hi, wi, hi, wi); dims(da, hi, wi, hd, wd); dims(db, hd, wd, hd, wd);
        dims(ct, hi, wi, hi, wi); dims(u3, hi, wi, hi, wi); dims(u1, hi, wi, hi, wi);
        sk.out = p->cat[i]; sk.out_ns = cat_ns;
        if (ng) { sk.d_out = p->d_cat[i]; sk.d_out_ns = cat_ns; }
        ct.in = p->cat[i]; ct.in_ns = cat_ns;
        if (ng) { ct.d_in = p->d_cat[i]; ct.d_in_ns = cat_ns; }
        da.in_ns = (size_t)cin * hi * wi; sk.in_ns = da.in_ns;
        db.in = da.out; db.in_ns = da.out_ns;
        u3.in = ct.out; u3.in_ns = ct.out_ns;
        u1.in = u3.out; u1.in_ns = u3.out_ns;
        if (ng) {
            db.d_in = da.d_out; db.d_in_ns = da.d_out_ns;
            u3.d_in = ct.d_out; u3.d_in_ns = ct.d_out_ns;
            u1.d_in = u3.d_out; u1.d_in_ns = u3.d_out_ns;
        }
    }
    for (int i = 1; i < S; ++i) {   // x_{i+1} = db_i.out ; d x_{i+1} = db_i.d_out
        p->u_skip[i].in = p->u_db[i - 1].out; p->u_da[i].in = p->u_db[i - 1].out;
        if (ng) {
            p->u_da[i].d_in = p->u_db[i - 1].d_out; p->u_da[i].d_in_ns = p->u_db[i - 1].d_out_ns; p->u_da[i].d_in_accumulate = 0;  // first writer
            p->u_skip[i].d_in = p->u_db[i - 1].d_out; p->u_skip[i].d_in_ns = p->u_db[i - 1].d_out_ns; p->u_skip[i].d_in_accumulate = 1;
        }
    }
    p->forward_saved = 0;
}

// the convolution of a unit as ConvArgs (ws / ws_floats: its split-K workspace)
static ConvArgs unit_conv_args(const SpliceGenPlan* p, const Unit& u, const float* params, float* ws, size_t ws_floats) {
    ConvArgs a = {};
    a.in = u.in; a.w = params + u.w_off; a.bias = params + u.b_off; a.out = u.y;
    a.in_nstride = u.in_ns; a.in_cstride = (size_t)u.Hi * u.Wi; a.out_nstride = u.y_ns; a.out_cstride = (size_t)u.Ho * u.Wo;
    a.w_jstride = (size_t)u.Cin * u.ks * u.ks; a.w_cstride = (size_t)u.ks * u.ks; a.p_nstride = p->p_nstride;
    a.N = p->N; a.Cin = u.Cin; a.Hi = u.Hi; a.Wi = u.Wi; a.Cout = u.Cout; a.Ho = u.Ho; a.Wo = u.Wo;
    a.ks = u.ks; a.stride = u.stride; a.pad = (u.ks - 1) / 2; a.reflect = p->gen->arch.reflect && u.ks > 1;
    a.ws = ws; a.ws_floats = ws_floats;
    // small planes: a split-K convolution leaves its slabs for the BatchNorm kernel, which adds them while it loads the plane
    a.defer_reduce = !p->batch_stats && u.Ho * u.Wo <= bn_small_hw();
    return a;
}
// the skip unit of a scale as the BnPre of that scale's concat BatchNorm
static BnPre skip_pre(const SpliceGenPlan* p, int i, const float* params, float* grads) {
    const Unit& sk = p->u_skip[i];
    BnPre pr;
    pr.y = sk.y; pr.y_ns = sk.y_ns;
    pr.gamma = params + sk.g_off; pr.beta = params + sk.be_off;
    pr.mean = sk.mean; pr.rstd = sk.rstd; pr.slope = sk.slope; pr.C = sk.Cout;
    pr.dy = sk.dy;
    if (grads) { pr.dgamma = grads + sk.g_off; pr.dbeta = grads + sk.be_off; }
    else if (p->skip_ks[i] > 1) { pr.slabs = p->skip_ws[i]; pr.ksplit = p->skip_ks[i]; pr.bias = params + sk.b_off; }
    return pr;
}
// BatchNorm + activation of a unit behind its convolution (ksplit > 1 with a deferred reduction: the slabs at `slabs` are summed here)
static int unit_bn_forward(const SpliceGenPlan* p, const Unit& u, const float* params, const float* y, size_t y_ns, const float* slabs, int ksplit,
                           bool deferred, hipStream_t s, const BnUpsample* up, const BnPre* pre = nullptr) {
    SPLICE_DEV_REGION(13);
    const int N = p->N;
    if (deferred && ksplit > 1) {
        RC(bn_fwd_slabs_launch(slabs, ksplit, params + u.b_off, u.y, u.y_ns, u.out, u.out_ns, N, u.Cout, u.Ho * u.Wo, params + u.g_off,
                               params + u.be_off, BN_EPS, u.mean, u.rstd, u.slope, s, p->p_nstride));
        return SPLICE_OK;
    }
    if (p->batch_stats && up) {   // batch statistics: the upsampled channels are materialised first (no fusion with the statistics pass)
        RC(upsample2x_fwd_launch(up->src, up->src_ns, const_cast<float*>(y) + (size_t)up->c0 * u.Ho * u.Wo, y_ns, N, u.Cout - up->c0, up->h, up->w, up->Ho, up->Wo, s));
        up = nullptr;
    }
    RC(bn_fwd_launch(y, y_ns, u.out, u.out_ns, N, u.Cout, u.Ho * u.Wo, params + u.g_off, params + u.be_off, BN_EPS, u.s1, u.mean, u.rstd, u.slope, s, up,
                     p->p_nstride, p->batch_stats, pre));
    return SPLICE_OK;
}
static int unit_forward(const SpliceGenPlan* p, const Unit& u, const float* params, hipStream_t s, const BnUpsample* up = nullptr, const BnPre* pre = nullptr) {
    if (!u.ks) return unit_bn_forward(p, u, params, u.in, u.in_ns, nullptr, 1, false, s, up, pre);
    const ConvArgs a = unit_conv_args(p, u, params, p->conv_ws, p->conv_ws_floats);
    int ksplit = 1;
    RC(conv_launch(a, s, &ksplit));
    return unit_bn_forward(p, u, params, u.y, u.y_ns, p->conv_ws, ksplit, a.defer_reduce != 0, s, up);
}
// two units that read the same input and do not depend on each other (the skip branch and the first encoder convolution of a
// scale): their convolutions share one launch (conv_pair_launch); a = the 1x1 unit
// a_bn_later: a's BatchNorm runs inside a later kernel (BnPre of the concat BatchNorm): a split-K convolution then sums its slabs
// itself (same order as the BatchNorm kernel would: same bits) -- the slab workspace is reused long before that kernel runs
static int unit_pair_forward(SpliceGenPlan* p, int scale, const Unit& ua, const Unit& ub, const float* params, hipStream_t s, bool a_bn_later) {
    const size_t half = p->conv_ws_floats / 2;
    ConvArgs a = unit_conv_args(p, ua, params, p->conv_ws, half);
    if (a_bn_later) {   // the slabs must outlive the deeper scales' convolutions: own workspace (none: summed here, same order, same bits)
        if (p->skip_ws[scale] && (size_t)a.N * a.Cout * a.Ho * a.Wo * 16 <= p->skip_ws_floats[scale]) { a.ws = p->skip_ws[scale]; a.ws_floats = p->skip_ws_floats[scale]; }
        else a.defer_reduce = 0;
    }
    const ConvArgs b = unit_conv_args(p, ub, params, p->conv_ws + half, half);
    int ksa = 1, ksb = 1;
    RC(conv_pair_launch(a, b, s, &ksa, &ksb));
    p->skip_ks[scale] = a_bn_later && a.defer_reduce ? ksa : 1;
    if (!a_bn_later) RC(unit_bn_forward(p, ua, params, ua.y, ua.y_ns, a.ws, ksa, a.defer_reduce != 0, s, nullptr));
    RC(unit_bn_forward(p, ub, params, ub.y, ub.y_ns, b.ws, ksb, b.defer_reduce != 0, s, nullptr));
    return SPLICE_OK;
}

// backward of one unit, first part: BatchNorm backward (consumes u.d_out, leaves dy), the conv bias gradient (exact zero) and the
// weight-gradient work item
// bn_done: the BatchNorm backward of this unit has already run inside another kernel (BnPre)
static int unit_backward_bn(const SpliceGenPlan* p, const Unit& u, const float* params, float* grads, int acc, hipStream_t s,
                            const BnUpsample* up = nullptr, const BnPre* pre = nullptr, bool bn_done = false) {
    const int N = p->N, HW = u.Ho * u.Wo;
    const float* y = u.ks ? u.y : u.in;
    const size_t y_ns = u.ks ? u.y_ns : u.in_ns;
    float* dy = u.ks ? u.dy : u.d_in;          // BN-only unit: dy IS the input gradient
    const size_t dy_ns = u.ks ? u.y_ns : u.d_in_ns;
    auto& pend = const_cast<SpliceGenPlan*>(p)->pend;
    if (pend.target && (bn_done || pend.target != u.d_out)) {
        splice_set_error("generator backward: split-K slabs pending for a gradient that the next BatchNorm does not read");
        return SPLICE_ERR_STATE;
    }
    if (!bn_done) {
        BnSlabs sl;
        if (pend.target) { sl.slabs = pend.slabs; sl.ksplit = pend.ksplit; sl.accumulate = pend.accumulate; pend.target = nullptr; }
        SPLICE_DEV_REGION(14);
        RC(bn_bwd_launch(u.d_out, u.d_out_ns, u.out, u.out_ns, y, y_ns, dy, dy_ns, N, u.Cout, HW, params + u.g_off, u.mean, u.rstd, u.slope,
                         u.s1, grads + u.g_off, grads + u.be_off, acc, s, p->batch_stats ? nullptr : up, p->p_nstride, p->batch_stats, pre, &sl, params + u.be_off));
    }
    if (!u.ks) return SPLICE_OK;
    // The bias of a conv that feeds a train-mode BatchNorm has an analytically ZERO gradient (BN subtracts the
    // per-channel mean, sum_p dy = 0); the reference's autograd returns fp32 rounding noise there.  We write the
    // exact value and skip the reduction.
    {
        WgradReduceAll& r = const_cast<SpliceGenPlan*>(p)->red;
        const int li = r.count++;
        r.n[li] = u.Cout; r.chunks[li] = 0; r.ws_off[li] = 0; r.dw_off[li] = (long long)u.b_off;
    }
    {
        WgradArgs a = {};
        a.x = u.in; a.dy = u.dy;
        a.x_nstride = u.in_ns; a.x_cstride = (size_t)u.Hi * u.Wi; a.dy_nstride = u.y_ns; a.dy_cstride = (size_t)HW;
        a.N = N; a.Cin = u.Cin; a.Hi = u.Hi; a.Wi = u.Wi; a.Cout = u.Cout; a.Ho = u.Ho; a.Wo = u.Wo;
        a.ks = u.ks; a.stride = u.stride; a.pad = (u.ks - 1) / 2; a.reflect = p->gen->arch.reflect && u.ks > 1;
        a.ws = p->wgrad_ws + u.wg_off;
        int chunks = 0;
        RC(conv_wgrad_add(&const_cast<SpliceGenPlan*>(p)->wg, a, &chunks));
        WgradReduceAll& r = const_cast<SpliceGenPlan*>(p)->red;
        const int li = r.count++;
        r.n[li] = u.Cout * u.Cin * u.ks * u.ks; r.chunks[li] = chunks; r.ws_off[li] = (long long)u.wg_off; r.dw_off[li] = (long long)u.w_off;
    }
    return SPLICE_OK;
}
// the data-gradient convolution of a unit (dy -> d_in) in data-gradient form; accumulate: 1 = add into d_in
static ConvArgs unit_dgrad_args(const SpliceGenPlan* p, const Unit& u, const float* params, int accumulate, float* ws, size_t ws_floats) {
    const int HW = u.Ho * u.Wo;
    ConvArgs a = {};
    a.in = u.dy; a.w = params + u.w_off; a.bias = nullptr; a.out = u.d_in;
    a.in_nstride = u.y_ns; a.in_cstride = (size_t)HW; a.out_nstride = u.d_in_ns; a.out_cstride = (size_t)u.Hi * u.Wi;
    a.w_jstride = (size_t)u.ks * u.ks; a.w_cstride = (size_t)u.Cin * u.ks * u.ks; a.p_nstride = p->p_nstride;
    a.N = p->N; a.Cin = u.Cout; a.Hi = u.Ho; a.Wi = u.Wo; a.Cout = u.Cin; a.Ho = u.Hi; a.Wo = u.Wi;
    a.ks = u.ks; a.stride = u.stride; a.pad = (u.ks - 1) / 2; a.transposed = 1; a.accumulate = accumulate;
    a.ws = ws; a.ws_floats = ws_floats;
    return a;
}
// A split-K data gradient whose result is read by exactly one kernel -- the small-plane BatchNorm backward of the unit that
// produced this unit's input, which runs next -- leaves its slabs to that kernel (BnSlabs) instead of launching the reduction.
// last_writer: no other convolution adds to d_in after this one.
static bool dgrad_may_defer(const SpliceGenPlan* p, const Unit& u, bool last_writer) {
    return last_writer && u.ks && u.d_in && !(p->gen->arch.reflect && u.ks > 1) && u.d_in_ns == (size_t)u.Cin * u.Hi * u.Wi &&
           bn_bwd_takes_slabs(u.Hi * u.Wi, p->N, p->p_nstride, p->batch_stats);
}
static void dgrad_note_slabs(const SpliceGenPlan* p, const ConvArgs& a, int ksplit) {
    if (!a.defer_reduce || ksplit <= 1) return;
    auto& pend = const_cast<SpliceGenPlan*>(p)->pend;
    pend.target = a.out; pend.slabs = a.ws; pend.ksplit = ksplit; pend.accumulate = a.accumulate;
}
// second part: the data gradient (if the unit's input needs one)
static int unit_backward_dgrad(const SpliceGenPlan* p, const Unit& u, const float* params, int accumulate, hipStream_t s, bool last_writer = false) {
    if (!u.ks || !u.d_in) return SPLICE_OK;
    ConvArgs a = unit_dgrad_args(p, u, params, accumulate, p->conv_ws, p->conv_ws_floats);
    if (p->gen->arch.reflect && u.ks > 1) { RC(conv_reflect_dgrad_launch(a, p->pad_scratch, s)); return SPLICE_OK; }
    a.defer_reduce = dgrad_may_defer(p, u, last_writer);
    int ks = 1;
    RC(conv_launch(a, s, &ks));
    dgrad_note_slabs(p, a, ks);
    return SPLICE_OK;
}
// backward of one unit: consumes u.d_out, produces parameter grads and (optionally) u.d_in
// last_writer: the unit's data gradient is the only (or the last) contribution to d_in and the BatchNorm backward that reads d_in runs next
static int unit_backward(const SpliceGenPlan* p, const Unit& u, const float* params, float* grads, int acc, hipStream_t s,
                         const BnUpsample* up = nullptr, const BnPre* pre = nullptr, bool bn_done = false, bool last_writer = false) {
    RC(unit_backward_bn(p, u, params, grads, acc, s, up, pre, bn_done));
    return unit_backward_dgrad(p, u, params, u.d_in_accumulate, s, last_writer);
}

static size_t wgrad_ws_need(const SpliceGenPlan* p, const Unit& u) {
    if (!u.ks) return 0;
    // upper bound over every size <= the plan maximum (the pixels-per-chunk choice is not monotone in size)
    const size_t chunks = (size_t)p->N * ((u.Ho * u.Wo + 63) / 64);
    return chunks * u.Cout * u.Cin * u.ks * u.ks;
}

extern "C" {

static int arch_check(const GenArch& a) {
    if (a.n_scales < 1 || a.n_scales > MAXS || a.in_channels < 1 || a.in_channels > 128 || a.out_channels < 1 || a.out_channels > 16 || a.filter_skip != 1) return 0;
    for (int i = 0; i < a.n_scales; ++i) {
        if (a.down[i] < 1 || a.down[i] > 128 || a.up[i] < 1 || a.up[i] > 128 || a.skip[i] < 1 || a.skip[i] > 128) return 0;
        for (int k : {a.filter_down[i], a.filter_up[i]})
            if (k != 1 && k != 3 && k != 5 && k != 7) return 0;
    }
    return 1;
}
int splice_gen_create_arch(const splice_gen_arch* arch, void** out) {
    if (!out) return SPLICE_ERR_ARG;
    SpliceGen* g = new SpliceGen();
    g->arch = arch ? *arch : default_arch();
    if (!arch_check(g->arch)) {
        splice_set_error("splice_gen_create_arch: unsupported skip() architecture (1..%d scales, channels <= 128, filters 1/3/5/7, 1x1 skip filter, skip channels > 0)", MAXS);
        delete g;
        return SPLICE_ERR_ARG;
    }
    size_t offs[MAXS * 6 * 4], head[4], roffs[MAXS * 6];
    build_table(g->arch, g->table, offs, head, &g->buffers, roffs);
    *out = g;
    return SPLICE_OK;
}
int splice_gen_create(void** out) { return splice_gen_create_arch(nullptr, out); }
void splice_gen_destroy(void* h) { delete (SpliceGen*)h; }

long long splice_gen_param_count(void* h) { return h ? (long long)((SpliceGen*)h)->table.total : -1; }
int splice_gen_num_tensors(void* h) { return h ? (int)((SpliceGen*)h)->table.entries.size() : -1; }
// name / offset / numel of parameter tensor i, in the reference's netG.parameters() order
int splice_gen_tensor_info(void* h, int i, const char** name, long long* offset, long long* numel) {
    SpliceGen* g = (SpliceGen*)h;
    if (!g || i < 0 || i >= (int)g->table.entries.size()) return SPLICE_ERR_ARG;
    if (name) *name = g->table.entries[i].name.c_str();
    if (offset) *offset = (long long)g->table.entries[i].off;
    if (numel) *numel = (long long)g->table.entries[i].numel;
    return SPLICE_OK;
}

int splice_gen_plan_create(void* h, int N, int H, int W, int need_grad, void** out) {
    SpliceGen* g = (SpliceGen*)h;
    if (!g || !out) return SPLICE_ERR_ARG;
    const GenArch& A = g->arch;
    const int S = A.n_scales, min_hw = (1 << S) + 1;
    if (N < 1 || H < min_hw || W < min_hw) {
        splice_set_error("splice_gen_plan_create: need N>=1 and H,W >= %d (train-mode BatchNorm needs >1 value per channel at the deepest scale)", min_hw);
        return SPLICE_ERR_ARG;
    }
    SpliceGenPlan* p = new SpliceGenPlan();
    p->gen = g; p->N = N; p->H = H; p->W = W; p->maxH = H; p->maxW = W; p->need_grad = need_grad;
    p->h[0] = H; p->w[0] = W;
    for (int i = 1; i <= S; ++i) { p->h[i] = (p->h[i - 1] + 1) / 2; p->w[i] = (p->w[i - 1] + 1) / 2; }
    size_t offs[MAXS * 6 * 4], head[4], roffs[MAXS * 6];
    ParamTable tmp, tmpb;
    build_table(A, tmp, offs, head, &tmpb, roffs);
    p->head_w = head[0]; p->head_b = head[1];
    int rc = SPLICE_OK;
    auto fail = [&]() { for (void* q : p->allocs) (void)hipFree(q); delete p; return rc; };
    size_t ws_need = 0, pad_need = 0;
    int max_c = A.in_channels;
    for (int i = 0; i < S && rc == SPLICE_OK; ++i) {
        const int cin = i == 0 ? A.in_channels : A.down[i - 1];
        const int hi = p->h[i], wi = p->w[i], hd = p->h[i + 1], wd = p->w[i + 1];
        const int k = i < S - 1 ? A.up[i + 1] : A.down[i];
        p->kch[i] = k;
        const size_t catC = A.skip[i] + k;
        if ((int)catC > max_c) max_c = (int)catC;
        if (A.reflect) {   // largest padded-domain gradient: N * Cin * (H + 2 pad) * (W + 2 pad) over the layers with a filter > 1
            const size_t pd = A.filter_down[i] / 2, pu = A.filter_up[i] / 2;
            const size_t cand[3] = {(size_t)cin * (hi + 2 * pd) * (wi + 2 * pd), (size_t)A.down[i] * (hd + 2 * pd) * (wd + 2 * pd), catC * (hi + 2 * pu) * (wi + 2 * pu)};
            for (size_t c : cand) if (N * c > pad_need) pad_need = N * c;
        }
        p->skip_ws[i] = nullptr; p->skip_ws_floats[i] = 0;
        if (hi * wi <= bn_small_hw() && A.skip[i] > 0) {   // split-K slabs of a chained skip convolution (up to 16 slices)
            p->skip_ws_floats[i] = (size_t)16 * N * A.skip[i] * hi * wi;
            if ((rc = palloc(p, &p->skip_ws[i], p->skip_ws_floats[i])) != SPLICE_OK) break;
        }
        if ((rc = palloc(p, &p->cat[i], (size_t)N * catC * hi * wi)) != SPLICE_OK) break;
        if (need_grad && (rc = palloc(p, &p->d_cat[i], (size_t)N * catC * hi * wi)) != SPLICE_OK) break;
        size_t* o = offs + (size_t)i * 6 * 4;
        auto setp = [&](Unit& u, int idx) { u.w_off = o[idx * 4 + 0]; u.b_off = o[idx * 4 + 1]; u.g_off = o[idx * 4 + 2]; u.be_off = o[idx * 4 + 3]; u.r_off = roffs[i * 6 + idx]; };
        auto mk = [&](Unit& u, int idx, int ks, int stride, int ci, int co, int Hi, int Wi, int Ho, int Wo, bool own) {
            u.ks = ks; u.stride = stride; u.Cin = ci; u.Cout = co; u.Hi = Hi; u.Wi = Wi; u.Ho = Ho; u.Wo = Wo;
            setp(u, idx);
            return unit_alloc(p, u, own);
        };
        if ((rc = mk(p->u_skip[i], 0, A.filter_skip, 1, cin, A.skip[i], hi, wi, hi, wi, false)) != SPLICE_OK) break;
        if ((rc = mk(p->u_da[i], 1, A.filter_down[i], 2, cin, A.down[i], hi, wi, hd, wd, true)) != SPLICE_OK) break;
        if ((rc = mk(p->u_db[i], 2, A.filter_down[i], 1, A.down[i], A.down[i], hd, wd, hd, wd, true)) != SPLICE_OK) break;
        p->u_cat[i].slope = 1.0f;
        if ((rc = mk(p->u_cat[i], 3, 0, 1, (int)catC, (int)catC, hi, wi, hi, wi, true)) != SPLICE_OK) break;
        if ((rc = mk(p->u_up3[i], 4, A.filter_up[i], 1, (int)catC, A.up[i], hi, wi, hi, wi, true)) != SPLICE_OK) break;
        if ((rc = mk(p->u_up1[i], 5, 1, 1, A.up[i], A.up[i], hi, wi, hi, wi, true)) != SPLICE_OK) break;
        for (Unit* u : {&p->u_skip[i], &p->u_da[i], &p->u_db[i], &p->u_up3[i], &p->u_up1[i]}) { u->wg_off = ws_need; ws_need += wgrad_ws_need(p, *u); }
    }
    if (rc != SPLICE_OK) return fail();
    {
        p->head_wg_off = ws_need;
        ws_need += (size_t)N * ((H * W + 63) / 64) * A.out_channels * A.up[0];
        p->head_bias_off = ws_need;   // per-segment partials of the head bias gradient: reduced with the weight-gradient partials
        ws_need += (size_t)sigmoid_bias_part_floats(N, A.out_channels);
    }
    if ((rc = palloc(p, &p->x_copy, (size_t)N * A.in_channels * H * W)) != SPLICE_OK) return fail();
    if (pad_need && need_grad && (rc = palloc(p, &p->pad_scratch, pad_need)) != SPLICE_OK) return fail();
    p->conv_ws_floats = (size_t)16 * N * max_c * 8192;   // split-K is only chosen for layers with < 128 tiles (<= 8192 pixels)
    if ((rc = palloc(p, &p->conv_ws, p->conv_ws_floats)) != SPLICE_OK) return fail();
    if (need_grad) {
        if ((rc = palloc(p, &p->wgrad_ws, ws_need)) != SPLICE_OK) return fail();
        if ((rc = palloc(p, &p->d_head_pre, (size_t)N * A.out_channels * H * W)) != SPLICE_OK) return fail();
        if ((rc = palloc(p, &p->out_copy, (size_t)N * A.out_channels * H * W)) != SPLICE_OK) return fail();
    }
    plan_configure(p, H, W);
    *out = p;
    return SPLICE_OK;
}

// Re-target a plan to a smaller input (H, W <= the creation size) without reallocating: the data
// feed draws a different crop size every step (data/transforms.py:21-22).
int splice_gen_plan_resize(void* plan, int H, int W) {
    SpliceGenPlan* p = (SpliceGenPlan*)plan;
    const int min_hw = p ? (1 << p->gen->arch.n_scales) + 1 : 33;
    if (!p || H < min_hw || W < min_hw || H > p->maxH || W > p->maxW) {
        splice_set_error("splice_gen_plan_resize: %dx%d outside [%d, plan maximum]", H, W, min_hw);
        return SPLICE_ERR_ARG;
    }
    if (H != p->H || W != p->W) plan_configure(p, H, W);
    return SPLICE_OK;
}

// stride > 0: the plan's N images are INDEPENDENT generators (several pairs optimised side by side, train.py:34-49 run P
// times): image n reads its parameters at params + n * stride and its gradients go to grads + n * stride; every launch
// policy is then taken from one image, so a pair's results do not depend on how many pairs share the launches.
// stride == 0 (default): one generator applied to N images, gradients summed over the images.
int splice_gen_plan_set_arena_stride(void* plan, long long stride) {
    SpliceGenPlan* p = (SpliceGenPlan*)plan;
    if (!p || stride < 0 || (stride > 0 && stride < (long long)p->gen->table.total)) {
        splice_set_error("splice_gen_plan_set_arena_stride: stride must be 0 or >= the parameter count");
        return SPLICE_ERR_ARG;
    }
    p->p_nstride = (size_t)stride;
    plan_configure(p, p->H, p->W);   // (the BatchNorm launch forms depend on it)
    return SPLICE_OK;
}

// on != 0: the plan stands for ONE netG call on a batch of N images (netG(A_global) with n_crops > 1 crops, models/model.py:15
// + data/transforms.py:19-27): train-mode BatchNorm takes its statistics over the whole batch (N <= 8), gradients are
// summed over the images.  0 (default): N separate batch-1 calls (per-image statistics).
int splice_gen_plan_set_batch_stats(void* plan, int on) {
    SpliceGenPlan* p = (SpliceGenPlan*)plan;
    if (!p || (on && (p->N > 8 || p->p_nstride))) {
        splice_set_error("splice_gen_plan_set_batch_stats: batch statistics need N <= 8 images of one generator");
        return SPLICE_ERR_ARG;
    }
    p->batch_stats = on ? 1 : 0;
    plan_configure(p, p->H, p->W);
    return SPLICE_OK;
}

int splice_gen_plan_dims(void* plan, int* N, int* H, int* W, long long* nparams) {
    SpliceGenPlan* p = (SpliceGenPlan*)plan;
    if (!p) return SPLICE_ERR_ARG;
    if (N) *N = p->N;
    if (H) *H = p->H;
    if (W) *W = p->W;
    if (nparams) *nparams = (long long)p->gen->table.total;
    return SPLICE_OK;
}

// BatchNorm running statistics (nn.BatchNorm2d, momentum 0.1, train mode: models/unet/common.py:95-96) of the LAST forward
// of each plan, applied in the order given (= the order of the netG calls they stand for: models/model.py:15-23 calls
// A_global, A, B_global).  `running`: buffer arena(s) in state_dict order (splice_gen_buffer_info); plans whose images
// are independent generators update arena n = image n at + n * running_stride.
long long splice_gen_buffer_count(void* h) { return h ? (long long)((SpliceGen*)h)->buffers.total : -1; }
int splice_gen_num_buffers(void* h) { return h ? (int)((SpliceGen*)h)->buffers.entries.size() : -1; }
int splice_gen_buffer_info(void* h, int i, const char** name, long long* offset, long long* numel) {
    SpliceGen* g = (SpliceGen*)h;
    if (!g || i < 0 || i >= (int)g->buffers.entries.size()) return SPLICE_ERR_ARG;
    if (name) *name = g->buffers.entries[i].name.c_str();
    if (offset) *offset = (long long)g->buffers.entries[i].off;
    if (numel) *numel = (long long)g->buffers.entries[i].numel;
    return SPLICE_OK;
}
int splice_gen_running_stats_update(void* const* plans, int n_plans, float* running, long long running_stride, float momentum,
                                    splice_stream_t stream) {
    if (!plans || n_plans < 1 || n_plans > RUNSTAT_MAX_PLANS || !running) return SPLICE_ERR_ARG;
    RunStatTable t = {};
    t.n_plans = n_plans; t.n_bn = 6 * ((SpliceGenPlan*)plans[0])->gen->arch.n_scales;
    int max_images = 1;
    for (int k = 0; k < n_plans; ++k) {
        SpliceGenPlan* p = (SpliceGenPlan*)plans[k];
        if (!p) return SPLICE_ERR_ARG;
        t.N[k] = p->batch_stats ? 1 : p->N; t.indep[k] = p->p_nstride ? 1 : 0;   // a batch call is ONE update with the batch statistics
        if (t.indep[k] && p->N > max_images) max_images = p->N;
        int bn = 0;
        for (int i = 0; i < p->gen->arch.n_scales; ++i)
            for (Unit* u : {&p->u_skip[i], &p->u_da[i], &p->u_db[i], &p->u_cat[i], &p->u_up3[i], &p->u_up1[i]}) {
                if (k == 0) { t.C[bn] = u->Cout; t.r_off[bn] = (int)u->r_off; }
                t.HW[k][bn] = u->Ho * u->Wo * (p->batch_stats ? p->N : 1); t.mean[k][bn] = u->mean; t.rstd[k][bn] = u->rstd;
                ++bn;
            }
    }
    RC(bn_running_update_launch(t, running, (size_t)running_stride, momentum, BN_EPS, max_images, (hipStream_t)stream));
    return SPLICE_OK;
}

void splice_gen_plan_destroy(void* plan) {
    SpliceGenPlan* p = (SpliceGenPlan*)plan;
    if (!p) return;
    for (void* q : p->allocs) (void)hipFree(q);
    delete p;
}

static int scale_forward(SpliceGenPlan* p, int i, const float* params, hipStream_t s) {
    RC(unit_pair_forward(p, i, p->u_skip[i], p->u_da[i], params, s, p->chain[i]));
    RC(unit_forward(p, p->u_db[i], params, s));
    const float* deep = p->u_db[i].out;
    size_t deep_ns = p->u_db[i].out_ns;
    const GenArch& A = p->gen->arch;
    const int SKIPC = A.skip[i];
    if (i < A.n_scales - 1) {
        RC(scale_forward(p, i + 1, params, s));
        deep = p->u_up1[i + 1].out; deep_ns = p->u_up1[i + 1].out_ns;
    }
    const int hi = p->h[i], wi = p->w[i];
    // nn.Upsample(x2, bilinear) of the deeper branch into channels SKIPC.. of the concat: produced inside the concat's
    // BatchNorm kernels (bn_fwd_launch with a BnUpsample), not by a launch of its own
    BnUpsample up;
    up.src = deep; up.src_ns = deep_ns; up.c0 = SKIPC; up.h = p->h[i + 1]; up.w = p->w[i + 1]; up.Ho = hi; up.Wo = wi;
    const BnPre pre = skip_pre(p, i, params, nullptr);
    RC(unit_forward(p, p->u_cat[i], params, s, &up, p->chain[i] ? &pre : nullptr));
    RC(unit_forward(p, p->u_up3[i], params, s));
    RC(unit_forward(p, p->u_up1[i], params, s));
    return SPLICE_OK;
}

// x [N][3][H][W] in [0,1] -> y [N][3][H][W] in (0,1)   (netG(input), models/model.py:15-23)
static int gen_forward_impl(void* plan, const float* params, const float* x, float* y, bool borrowed, splice_stream_t stream) {
    SpliceGenPlan* p = (SpliceGenPlan*)plan;
    if (!p || !params || !x || !y) return SPLICE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    SpliceProfScope prof_scope(7); SPLICE_DEV_REGION(15);
    if (borrowed) {
        p->x_in = x;
    } else {
        RC(dev_copy_launch(p->x_copy, x, (size_t)p->N * p->gen->arch.in_channels * p->H * p->W * sizeof(float), s));
        p->x_in = p->x_copy;
    }
    p->u_skip[0].in = p->x_in; p->u_da[0].in = p->x_in;
    RC(scale_forward(p, 0, params, s));
    {
        const Unit& u = p->u_up1[0];
        ConvArgs a = {};
        a.in = u.out; a.w = params + p->head_w; a.bias = params + p->head_b; a.out = y;
        const int OC = p->gen->arch.out_channels, U0 = p->gen->arch.up[0];
        a.in_nstride = u.out_ns; a.in_cstride = (size_t)p->H * p->W; a.out_nstride = (size_t)OC * p->H * p->W; a.out_cstride = (size_t)p->H * p->W;
        a.w_jstride = U0; a.w_cstride = 1; a.p_nstride = p->p_nstride;
        a.N = p->N; a.Cin = U0; a.Hi = p->H; a.Wi = p->W; a.Cout = OC; a.Ho = p->H; a.Wo = p->W; a.ks = 1; a.stride = 1; a.pad = 0; a.act = 1;
        RC(conv_launch(a, s));
    }
    if (p->need_grad) {
        if (borrowed) {
            p->y_saved = y;
        } else {
            RC(dev_copy_launch(p->out_copy, y, (size_t)p->N * p->gen->arch.out_channels * p->H * p->W * sizeof(float), s));
            p->y_saved = p->out_copy;
        }
        p->forward_saved = 1;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { splice_set_error("splice_gen_forward: %s", hipGetErrorString(e)); return SPLICE_ERR_HIP; }
    return SPLICE_OK;
}
int splice_gen_forward(void* plan, const float* params, const float* x, float* y, splice_stream_t stream) {
    return gen_forward_impl(plan, params, x, y, false, stream);
}
// x and y stay untouched by the caller until the matching splice_gen_backward has run: no private copies (two launches)
int splice_gen_forward_borrowed(void* plan, const float* params, const float* x, float* y, splice_stream_t stream) {
    return gen_forward_impl(plan, params, x, y, true, stream);
}

// head_done: the caller has already run the backward of u_up1[i] (paired with the shallower scale's skip branch)
static int scale_backward(SpliceGenPlan* p, int i, const float* params, float* grads, int acc, hipStream_t s, bool head_done = false) {
    // u_up1[i].d_out holds d u_i
    // (up1 -> up3 -> concat: each data gradient is read by the next BatchNorm backward only, which sums its split-K slabs)
    if (!head_done) RC(unit_backward(p, p->u_up1[i], params, grads, acc, s, nullptr, nullptr, false, true));
    RC(unit_backward(p, p->u_up3[i], params, grads, acc, s, nullptr, nullptr, false, true));
    const int hi = p->h[i], wi = p->w[i];
    const GenArch& A = p->gen->arch;
    const int SKIPC = A.skip[i];
    Unit& deep = i < A.n_scales - 1 ? p->u_up1[i + 1] : p->u_db[i];
    // small planes: the upsampled channels' gradient goes through the adjoint inside the concat's BatchNorm backward
    BnUpsample up;
    up.d_src = deep.d_out; up.d_src_ns = deep.d_out_ns; up.c0 = SKIPC; up.h = p->h[i + 1]; up.w = p->w[i + 1]; up.Ho = hi; up.Wo = wi;
    const BnPre pre = skip_pre(p, i, params, grads);
    const bool chained = p->chain[i];
    RC(unit_backward(p, p->u_cat[i], params, grads, acc, s, &up, chained ? &pre : nullptr));   // -> d_cat[i] (chained: the skip channels' gradient goes on into u_skip[i].dy)
    if (p->batch_stats || !bn_bwd_fuses_upsample_ex(hi * wi, up.h, up.w, p->N, p->p_nstride, p->batch_stats))
        RC(upsample2x_bwd_launch(p->d_cat[i] + (size_t)SKIPC * hi * wi, p->u_skip[i].d_out_ns, deep.d_out, deep.d_out_ns, p->N, p->kch[i],
                                 p->h[i + 1], p->w[i + 1], hi, wi, s));
    // The skip branch's backward depends on nothing deeper: where its input needs a gradient (every scale but the first) and a
    // deeper scale exists, it runs NOW, and its 1x1 data gradient shares a launch with the 1x1 data gradient of the deeper
    // scale's decoder output unit (two independent convolutions, conv_pair_launch).  It is then the FIRST writer of d x_i and
    // the encoder convolution's data gradient adds to it (a + b = b + a: the same bits as the other order).
    const bool skip_early = i < A.n_scales - 1 && p->u_skip[i].d_in != nullptr ;
    if (skip_early) {
        const Unit &sk = p->u_skip[i], &h1 = p->u_up1[i + 1];
        RC(unit_backward_bn(p, sk, params, grads, acc, s, nullptr, nullptr, chained));
        RC(unit_backward_bn(p, h1, params, grads, acc, s));
        const size_t half = p->conv_ws_floats / 2;
        ConvArgs db = unit_dgrad_args(p, h1, params, h1.d_in_accumulate, p->conv_ws + half, half);
        db.defer_reduce = dgrad_may_defer(p, h1, true);   // read by up3_{i+1}'s BatchNorm backward, the next launch
        int ksa = 1, ksb = 1;
        RC(conv_pair_launch(unit_dgrad_args(p, sk, params, 0, p->conv_ws, half), db, s, &ksa, &ksb));
        dgrad_note_slabs(p, db, ksb);
    }
    if (i < A.n_scales - 1) RC(scale_backward(p, i + 1, params, grads, acc, s, skip_early));   // leaves d x_{i+1} in u_db[i].d_out
    RC(unit_backward(p, p->u_db[i], params, grads, acc, s, nullptr, nullptr, false, true));
    RC(unit_backward_bn(p, p->u_da[i], params, grads, acc, s));
    // (with the skip branch's data gradient already in d x_i, the encoder's is the last writer and u_db[i-1]'s BatchNorm backward runs next)
    RC(unit_backward_dgrad(p, p->u_da[i], params, skip_early ? 1 : p->u_da[i].d_in_accumulate, s, skip_early));
    if (!skip_early) RC(unit_backward(p, p->u_skip[i], params, grads, acc, s, nullptr, nullptr, chained));
    return SPLICE_OK;
}

// dy: grad w.r.t. the generator output [N][3][H][W]; grads: flat arena (overwritten, or += when accumulate)
int splice_gen_backward(void* plan, const float* params, const float* dy, float* grads, int accumulate, splice_stream_t stream) {
    SpliceGenPlan* p = (SpliceGenPlan*)plan;
    if (!p || !p->need_grad || !p->forward_saved || !params || !dy || !grads) {
        splice_set_error("splice_gen_backward: plan without need_grad / no forward / null argument");
        return SPLICE_ERR_STATE;
    }
    hipStream_t s = (hipStream_t)stream;
    SpliceProfScope prof_scope(7); SPLICE_DEV_REGION(16);
    const int OC = p->gen->arch.out_channels, U0 = p->gen->arch.up[0];
    const size_t npix = (size_t)p->N * OC * p->H * p->W;
    p->red.count = 0;
    p->pend.target = nullptr;   // (a backward that failed half-way must not leave slabs pending for this one)
    p->wg.small.count = p->wg.small.total_wgs = 0;
    p->wg.big.count = p->wg.big.total_wgs = 0;
    p->wg.tile.count = p->wg.tile.total_wgs = 0;
    const Unit& u = p->u_up1[0];
    const int HW = p->H * p->W;
    (void)npix;
    {
        int chunks = 0;
        RC(sigmoid_bwd_bias_launch(dy, p->y_saved, p->d_head_pre, p->N, OC, HW, p->wgrad_ws + p->head_bias_off, s, p->p_nstride, &chunks));
        WgradReduceAll& r = p->red;
        const int li = r.count++;
        r.n[li] = OC; r.chunks[li] = chunks; r.ws_off[li] = (long long)p->head_bias_off; r.dw_off[li] = (long long)p->head_b;
    }
    {
        WgradArgs a = {};
        a.x = u.out; a.dy = p->d_head_pre;
        a.x_nstride = u.out_ns; a.x_cstride = (size_t)HW; a.dy_nstride = (size_t)OC * HW; a.dy_cstride = (size_t)HW;
        a.N = p->N; a.Cin = U0; a.Hi = p->H; a.Wi = p->W; a.Cout = OC; a.Ho = p->H; a.Wo = p->W; a.ks = 1; a.stride = 1; a.pad = 0;
        a.ws = p->wgrad_ws + p->head_wg_off;
        int chunks = 0;
        RC(conv_wgrad_add(&p->wg, a, &chunks));
        WgradReduceAll& r = p->red;
        const int li = r.count++;
        r.n[li] = OC * U0; r.chunks[li] = chunks; r.ws_off[li] = (long long)p->head_wg_off; r.dw_off[li] = (long long)p->head_w;
    }
    {
        ConvArgs a = {};
        a.in = p->d_head_pre; a.w = params + p->head_w; a.out = u.d_out;
        a.in_nstride = (size_t)OC * HW; a.in_cstride = (size_t)HW; a.out_nstride = u.d_out_ns; a.out_cstride = (size_t)HW;
        a.w_jstride = 1; a.w_cstride = U0; a.p_nstride = p->p_nstride;
        a.N = p->N; a.Cin = OC; a.Hi = p->H; a.Wi = p->W; a.Cout = U0; a.Ho = p->H; a.Wo = p->W; a.ks = 1; a.stride = 1; a.pad = 0; a.transposed = 1;
        RC(conv_launch(a, s));
    }
    RC(scale_backward(p, 0, params, grads, accumulate, s));
    if (p->pend.target) { splice_set_error("splice_gen_backward: split-K slabs left without a consumer"); return SPLICE_ERR_STATE; }
    SPLICE_DEV_REGION(17);
    RC(conv_wgrad_batched_launch(p->wg, s));   // all layers' partials, one launch
    {   // one deterministic reduction of every layer's per-chunk weight-gradient partials
        WgradReduceAll& r = p->red;
        r.prefix[0] = 0;
        for (int i = 0; i < r.count; ++i) r.prefix[i + 1] = r.prefix[i] + r.n[i];
        r.total = r.prefix[r.count];
        RC(wgrad_reduce_all_launch(r, p->wgrad_ws, grads, accumulate, s, p->N, p->p_nstride));
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { splice_set_error("splice_gen_backward: %s", hipGetErrorString(e)); return SPLICE_ERR_HIP; }
    return SPLICE_OK;
}

// torch.optim.Adam step over a flat arena (util/util.py:28-32); zero_grad != 0 also clears g.
int splice_adam_step(float* params, float* grads, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                     int step, int zero_grad, splice_stream_t stream) {
    if (!params || !grads || !m || !v || n < 1) return SPLICE_ERR_ARG;
    RC(adam_launch(params, grads, m, v, (size_t)n, lr, beta1, beta2, eps, step, zero_grad, (hipStream_t)stream));
    return SPLICE_OK;
}
}
