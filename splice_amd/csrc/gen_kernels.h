// Launchers of the generator kernels (gen_kernels.hip).
#pragma once
#include "common.h"

struct ConvArgs {
    const float* in;      // [N][*][Hi][Wi]; channel c at in + img*in_nstride + c*in_cstride
    const float* w;       // element (col n, red. channel c, tap) at n*w_jstride + c*w_cstride + tap
    const float* bias;    // [Cout] or null
    float* out;           // [N][*][Ho][Wo]; channel n at out + img*out_nstride + n*out_cstride
    size_t in_nstride, in_cstride, out_nstride, out_cstride;
    size_t w_jstride, w_cstride;
    size_t p_nstride;     // > 0: independent images -- image n reads w / bias at + n * p_nstride (its own parameter arena)
    int N, Cin, Hi, Wi, Cout, Ho, Wo;   // Cin = reduction channels, Cout = output columns
    int ks, stride, pad;
    int reflect;      // forward only: reflection padding (nn.ReflectionPad2d) instead of zero padding
    int act;          // 1 = sigmoid
    int transposed;   // data-gradient form
    int accumulate;   // out += result
    float* ws;        // optional split-K scratch (ws_floats floats); null disables split-K
    size_t ws_floats;
    int ksplit;       // set by the launcher
    int defer_reduce; // split-K: leave the raw slabs in ws (ksplit_out tells how many); the consumer adds bias + slabs in slice order
};
int conv_launch(const ConvArgs& a, hipStream_t s, int* ksplit_out = nullptr);
// two independent convolutions (a: 1x1, stride 1) in ONE launch where an instantiation exists, else two launches; same results
// as two conv_launch calls up to the summation order of `a` (it adopts b's wave groups)
int conv_pair_launch(ConvArgs a, ConvArgs b, hipStream_t s, int* ksplit_a = nullptr, int* ksplit_b = nullptr);
// data gradient of a reflection-padded convolution (transposed conv on the padded domain + mirror fold); see gen_kernels.hip
int conv_reflect_dgrad_launch(ConvArgs a, float* pad_scratch, hipStream_t s);

struct WgradArgs {
    const float* x;       // layer input  [N][*][Hi][Wi]
    const float* dy;      // grad of the conv output [N][*][Ho][Wo]
    float* ws;            // scratch: chunks * Cout*Cin*ks*ks floats
    size_t x_nstride, x_cstride, dy_nstride, dy_cstride;
    int N, Cin, Hi, Wi, Cout, Ho, Wo, ks, stride, pad;
    int pix_per_chunk, chunks_per_img;   // filled by the launcher
    int reflect;                         // reflection padding of the forward convolution
};
int wgrad_chunks(int N, int Ho, int Wo, int* pix_per_chunk, int* chunks_per_img);
constexpr int WGRAD_BATCH_MAX = 32;
struct WgradDesc {           // compact per-layer descriptor of the batched weight-gradient launch (72 bytes)
    const float* x; const float* dy; float* ws;
    uint32_t x_nstride, x_cstride, dy_nstride, dy_cstride;
    uint16_t Cin, Cout, Hi, Wi, Ho, Wo;
    uint8_t ks, stride, pad, variant;
    uint16_t pix_per_chunk, chunks_per_img;
    uint32_t wg_begin, chunks;
    uint32_t reflect;
};
struct WgradBatch { int count; int total_wgs; WgradDesc d[WGRAD_BATCH_MAX]; };   // by-value kernel argument
struct WgradBatchPair { WgradBatch small, big, tile; };   // layers with <= 32 / > 32 output channels (different kernel occupancy); 3x3 layers of big planes (conv_wgrad_tile_kernel)
int conv_wgrad_add(WgradBatchPair* b, WgradArgs a, int* chunks_out);       // queue one layer; partials go to a.ws
int conv_wgrad_batched_launch(const WgradBatchPair& b, hipStream_t s);     // one launch per class for every queued layer

constexpr int WGRAD_MAX_LAYERS = 64;   // 26 conv weights + 25 exact-zero bias ranges (chunks == 0)
struct WgradReduceAll {      // by-value kernel argument: one entry per conv layer of a backward
    long long prefix[WGRAD_MAX_LAYERS + 1];   // prefix sums of n (output elements), prefix[count] = total
    long long ws_off[WGRAD_MAX_LAYERS];       // offset of the layer's partials in the workspace (floats)
    long long dw_off[WGRAD_MAX_LAYERS];       // offset of the layer's weight gradient in the arena
    int n[WGRAD_MAX_LAYERS], chunks[WGRAD_MAX_LAYERS];
    long long total;
    int count;
    unsigned char vec[WGRAD_MAX_LAYERS];      // set by wgrad_reduce_all_launch: the layer is reduced four elements per thread (16-byte accesses); prefix then counts work items
};
// p_nstride > 0: independent images -- image n's chunks are summed into grads + n * p_nstride
int wgrad_reduce_all_launch(const WgradReduceAll& d, const float* ws, float* grads, int accumulate, hipStream_t s, int n_img = 1, size_t p_nstride = 0);

// train-mode BatchNorm (+LeakyReLU when slope != 1) forward: statistics in two deterministic stages
// (per-segment partials, recombined in the apply kernel); `part` = bn_part_floats(N, C) floats of scratch.
int bn_part_floats(int N, int C);
// Optional fusion of the decoder's x2 bilinear upsampling (models/unet/skip.py: nn.Upsample in front of the concat's
// BatchNorm) into the BatchNorm kernels of the concat unit: channels >= c0 of the normalised tensor ARE the upsampling of
// `src`.  Forward: they are computed on the fly (and stored into y for the backward) instead of by a launch of their
// own; backward (planes <= bn_small_hw() only): their input gradient goes straight through the adjoint into d_src.
struct BnUpsample {
    const float* src = nullptr; size_t src_ns = 0;   // [N][C - c0][h][w]
    float* d_src = nullptr; size_t d_src_ns = 0;
    int c0 = 0, h = 0, w = 0, Ho = 0, Wo = 0;
};
// A BatchNorm + LeakyReLU that sits in FRONT of the first `C` channels of a concat BatchNorm's input (the skip branch of a scale:
// models/unet/skip.py:60-62 feeding the Concat + BatchNorm of :75-81).  Both BatchNorms of such a channel are reductions over the
// SAME plane, so the workgroup that owns the plane in the concat's one-launch BatchNorm kernels runs the skip branch's first
// (forward: statistics, normalise, activate, store into the concat buffer; backward: the adjoint behind the concat's) -- the skip
// branch's BatchNorm has no launch of its own in either direction (round 4).
struct BnPre {
    const float* y = nullptr; size_t y_ns = 0;      // the skip convolution's output [N][C][HW] (forward: input; slabs != null: formed and stored here)
    const float* slabs = nullptr; int ksplit = 0;    // split-K slabs of that convolution (deferred reduction), with its bias
    const float* bias = nullptr;
    const float* gamma = nullptr; const float* beta = nullptr;   // the skip BatchNorm's parameters (image n at + n * p_nstride)
    float* mean = nullptr; float* rstd = nullptr;    // its saved statistics [N][C]
    float slope = 0.2f;
    int C = 0;
    // backward only
    float* dy = nullptr;                             // gradient w.r.t. the skip convolution's output [N][C][HW]
    float* dgamma = nullptr; float* dbeta = nullptr;
};
// The output gradient of a BatchNorm, left as split-K slabs by the data-gradient convolution that produces it (deferred reduction,
// backward direction): d_out = (accumulate ? what the first writer stored at d_out : 0) + sum_k slabs[k], formed by the BatchNorm
// backward while it loads the plane -- the arithmetic and order of conv_splitk_reduce_kernel, which then has no launch (round 4).
struct BnSlabs {
    const float* slabs = nullptr;   // [ksplit][N][C][HW]
    int ksplit = 0, accumulate = 0;
};
bool bn_bwd_takes_slabs(int HW, int N, size_t p_nstride, int batch);
bool bn_pre_supported(int HW, int N, size_t p_nstride, int batch);   // the concat BatchNorm of this size runs as ONE launch that can host a BnPre
bool bn_bwd_fuses_upsample(int HW, int h, int w);
bool bn_bwd_fuses_upsample_ex(int HW, int h, int w, int N, size_t p_nstride, int batch);   // incl. the one-launch form of the middle planes
int bn_fwd_launch(const float* y, size_t y_nstride, float* out, size_t out_nstride, int N, int C, int HW, const float* gamma,
                  const float* beta, float eps, float* part, float* mean, float* rstd, float slope, hipStream_t s, const BnUpsample* up = nullptr,
                  size_t p_nstride = 0, int batch = 0, const BnPre* pre = nullptr);   // batch != 0: statistics over all N images (nn.BatchNorm2d on a batch), N <= 8
// same, fused with the split-K reduction of the convolution that feeds it (small planes only: HW <= bn_small_hw()):
// y = bias + sum_k slabs[k] is formed, stored (the backward reads it) and normalised in one launch
int bn_small_hw();
int bn_fwd_slabs_launch(const float* slabs, int ksplit, const float* bias, float* y, size_t y_nstride, float* out, size_t out_nstride, int N,
                        int C, int HW, const float* gamma, const float* beta, float eps, float* mean, float* rstd, float slope, hipStream_t s,
                        size_t p_nstride = 0);
int bn_bwd_launch(const float* da, size_t da_nstride, const float* aout, size_t a_nstride, const float* y, size_t y_nstride, float* dy,
                  size_t dy_nstride, int N, int C, int HW, const float* gamma, const float* mean, const float* rstd, float slope,
                  float* part, float* dgamma, float* dbeta, int accumulate, hipStream_t s, const BnUpsample* up = nullptr, size_t p_nstride = 0,
                  int batch = 0, const BnPre* pre = nullptr, const BnSlabs* slabs = nullptr, const float* beta = nullptr);
int fill_zero_launch(float* p, int n, hipStream_t s);
int channel_sum_launch(const float* dy, size_t nstride, int N, int C, int HW, float* db, int accumulate, hipStream_t s);
int upsample2x_fwd_launch(const float* in, size_t in_nstride, float* out, size_t out_nstride, int N, int C, int h, int w, int Ho, int Wo, hipStream_t s);
int upsample2x_bwd_launch(const float* dout, size_t dout_nstride, float* din, size_t din_nstride, int N, int C, int h, int w, int Ho, int Wo, hipStream_t s);
int sigmoid_bwd_launch(const float* dout, const float* sout, float* dpre, size_t n, hipStream_t s);
// sigmoid backward of the [N][C][HW] head + per-channel sums of the result (the head's bias gradient) in two
// deterministic stages; part = C * 64 floats of scratch
int sigmoid_bwd_bias_launch(const float* dout, const float* sout, float* dpre, int N, int C, int HW, float* part, hipStream_t s, size_t p_nstride, int* chunks);
int sigmoid_bias_part_floats(int N, int C);   // floats of `part`: [image][segment][channel], summed by wgrad_reduce_all_launch
int adam_launch(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step, int zero_grad, hipStream_t s);
int adam_launch_dev(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, const int* step_dev,
                    int zero_grad, hipStream_t s, const float* g2 = nullptr);
int set_int_launch(int* p, int v, hipStream_t s);
// BatchNorm running statistics of up to 3 generator calls (in call order) in one launch; see gen_kernels.hip
constexpr int RUNSTAT_MAX_PLANS = 3, RUNSTAT_MAX_BN = 36;
struct RunStatTable {
    int n_plans, n_bn;
    int C[RUNSTAT_MAX_BN], r_off[RUNSTAT_MAX_BN];      // channels; offset of running_mean in the buffer arena (running_var follows at + C)
    int N[RUNSTAT_MAX_PLANS], indep[RUNSTAT_MAX_PLANS];
    int HW[RUNSTAT_MAX_PLANS][RUNSTAT_MAX_BN];
    const float* mean[RUNSTAT_MAX_PLANS][RUNSTAT_MAX_BN];   // [N][C] batch statistics saved by the forward
    const float* rstd[RUNSTAT_MAX_PLANS][RUNSTAT_MAX_BN];
};
int bn_running_update_launch(const RunStatTable& t, float* running, size_t r_nstride, float momentum, float eps, int max_images, hipStream_t s);
