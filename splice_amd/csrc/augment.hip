// Device-side structure-image augmentations of the reference's data feed (data/transforms.py:30-37):
// RandomHorizontalFlip, ColorJitter(brightness, contrast, saturation, hue) with its random op order, GaussianBlur(3).
// The reference runs them on PIL images on the host every step; here the image stays in HBM and the whole jitter is
// at most two launches (the contrast op needs the mean grey level of the image as it is at that point of the chain, so
// the op list is cut there), the blur one more.  Arithmetic = torchvision-0.10's tensor code path
// (functional_tensor.adjust_* / gaussian_blur); splice_amd/augment.py holds the same ops as torch code and the tests
// compare the two.
#include "kernels.h"

namespace {
constexpr int AUG_MAX_PART = 256;

__device__ __forceinline__ float grey(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float blend(float a, float b, float ratio) { return clamp01(ratio * a + (1.0f - ratio) * b); }

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float f) {
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const bool eq = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eq ? 1.0f : maxc);
    const float crd = eq ? 1.0f : cr;
    const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
    const float hr = maxc == r ? bc - gc : 0.f;
    const float hg = (maxc == g && maxc != r) ? 2.0f + rc - bc : 0.f;
    const float hb = (maxc != g && maxc != r) ? 4.0f + gc - rc : 0.f;
    float h = fmodf((hr + hg + hb) / 6.0f + 1.0f, 1.0f);
    h = h + f;
    h = h - floorf(h);   // python's % 1.0 (result in [0, 1))
    const float v = maxc;
    const float i6 = floorf(h * 6.0f);
    const float fr = h * 6.0f - i6;
    const int i = ((int)i6) % 6;
    const float p = clamp01(v * (1.0f - s)), q = clamp01(v * (1.0f - s * fr)), t = clamp01(v * (1.0f - s * (1.0f - fr)));
    switch (i) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

struct AugOps { int n; int op[4]; float f[4]; };   // op: 0 brightness, 1 contrast, 2 saturation, 3 hue

// out = ops(flip? mirror(in) : in); ops[0] may be a contrast (its mean = fixed-order sum of `mean_part` / (H*W));
// grey_part != null: per-workgroup sums of the OUTPUT's grey level (for a following contrast op).
__global__ __launch_bounds__(256) void aug_ops_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int flip,
                                                      AugOps ops, const float* __restrict__ mean_part, int n_mean_part,
                                                      float* __restrict__ grey_part) {
    __shared__ float red[4];
    __shared__ float mean_s;
    const int HW = H * W;
    if (mean_part) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < n_mean_part; i += 256) acc += mean_part[i];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) mean_s = ((red[0] + red[1]) + (red[2] + red[3])) / (float)HW;
        __syncthreads();
    }
    const float mean = mean_part ? mean_s : 0.f;
    float gsum = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const int y = i / W, x = i % W;
        const int src = flip ? y * W + (W - 1 - x) : i;
        float r = in[src], g = in[HW + src], b = in[2 * HW + src];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= ops.n) break;
            const float f = ops.f[k];
            switch (ops.op[k]) {
                case 0: r = clamp01(f * r); g = clamp01(f * g); b = clamp01(f * b); break;
                case 1: r = blend(r, mean, f); g = blend(g, mean, f); b = blend(b, mean, f); break;
                case 2: { const float gr = grey(r, g, b); r = blend(r, gr, f); g = blend(g, gr, f); b = blend(b, gr, f); break; }
                default: hue_shift(r, g, b, f); break;
            }
        }
        out[i] = r; out[HW + i] = g; out[2 * HW + i] = b;
        gsum += grey(r, g, b);
    }
    if (grey_part) {
        gsum = wave_sum(gsum);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsum;
        __syncthreads();
        if (threadIdx.x == 0) grey_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// 3x3 separable Gaussian (weights wc centre, ws side), reflect padding, all three channels
__global__ __launch_bounds__(256) void aug_blur3_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, float wc, float ws) {
    const int HW = H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * HW; i += gridDim.x * 256) {
        const int c = i / HW, p = i % HW, y = p / W, x = p % W;
        const int ym = y > 0 ? y - 1 : (H > 1 ? 1 : 0), yp = y < H - 1 ? y + 1 : (H > 1 ? H - 2 : 0);
        const int xm = x > 0 ? x - 1 : (W > 1 ? 1 : 0), xp = x < W - 1 ? x + 1 : (W > 1 ? W - 2 : 0);
        const float* q = in + (size_t)c * HW;
        const float r0 = ws * q[ym * W + xm] + wc * q[ym * W + x] + ws * q[ym * W + xp];
        const float r1 = ws * q[y * W + xm] + wc * q[y * W + x] + ws * q[y * W + xp];
        const float r2 = ws * q[yp * W + xm] + wc * q[yp * W + x] + ws * q[yp * W + xp];
        out[i] = ws * r0 + wc * r1 + ws * r2;
    }
}
}  // namespace

// img / out: fp32 [3][H][W] in [0,1] (out != img); scratch: >= 3*H*W + 256 floats.  order / factors: the ColorJitter draw
// (n_ops = 0: no jitter; order[k] in {0 brightness, 1 contrast, 2 saturation, 3 hue}, factors indexed BY OP as torchvision
// returns them).  blur_sigma <= 0: no blur.  The result is always left in `out`.
int augment_structure_launch(const float* img, float* out, float* scratch, int H, int W, int flip, int n_ops, const int* order,
                             const float* factors, float blur_sigma, hipStream_t s) {
    if (!img || !out || !scratch || H < 1 || W < 1 || n_ops < 0 || n_ops > 4 || out == img) return SPLICE_ERR_ARG;
    const int HW = H * W;
    int grid = cdiv(HW, 256);
    if (grid > AUG_MAX_PART) grid = AUG_MAX_PART;
    float* tmp = scratch;              // [3][H][W]
    float* part = scratch + 3 * HW;    // [AUG_MAX_PART]
    const bool blur = blur_sigma > 0.f;
    // stages of the jitter: ops before the contrast | contrast and everything after it
    int cpos = -1;
    for (int k = 0; k < n_ops; ++k) {
        if (order[k] < 0 || order[k] > 3) return SPLICE_ERR_ARG;
        if (order[k] == 1) cpos = k;
    }
    AugOps first = {}, second = {};
    const int n_first = cpos < 0 ? n_ops : cpos;
    first.n = n_first;
    for (int k = 0; k < n_first; ++k) { first.op[k] = order[k]; first.f[k] = factors[order[k]]; }
    if (cpos >= 0) {
        second.n = n_ops - cpos;
        for (int k = cpos; k < n_ops; ++k) { second.op[k - cpos] = order[k]; second.f[k - cpos] = factors[order[k]]; }
    }
    // destination juggling so that the final result lands in `out`: stages write tmp / out alternately
    const int n_stage = 1 + (cpos >= 0 ? 1 : 0) + (blur ? 1 : 0);
    float* dst = (n_stage & 1) ? out : tmp;
    SPLICE_LAUNCH(aug_ops_kernel, dim3(grid), dim3(256), 0, s, img, dst, H, W, flip, first, (const float*)nullptr, 0,
                       cpos >= 0 ? part : (float*)nullptr);
    const float* cur = dst;
    if (cpos >= 0) {
        dst = cur == out ? tmp : out;
        SPLICE_LAUNCH(aug_ops_kernel, dim3(grid), dim3(256), 0, s, cur, dst, H, W, 0, second, (const float*)part, grid, (float*)nullptr);
        cur = dst;
    }
    if (blur) {
        const float e = expf(-0.5f / (blur_sigma * blur_sigma));
        const float wc = 1.0f / (1.0f + 2.0f * e), ws = e / (1.0f + 2.0f * e);
        dst = cur == out ? tmp : out;
        SPLICE_LAUNCH(aug_blur3_kernel, dim3(cdiv(3 * HW, 256)), dim3(256), 0, s, cur, dst, H, W, wc, ws);
        cur = dst;
    }
    return cur == out ? SPLICE_OK : SPLICE_ERR_STATE;
}
