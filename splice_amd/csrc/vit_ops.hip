// Row-wise and element-wise kernels of the ViT path (all HBM-bound): LayerNorm fwd/bwd
// (K4 in SURVEY.md), patchify / un-patchify around the patch-embed GEMM (K2+K3), casts.
#include "kernels.h"

// ---------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row cached in registers (D <= 64*4*MAXV).
template <int MAXV>  // float4 vectors per lane
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                            int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;  // float4 per row
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        v[i] = idx < nv ? xr[idx] : float4{0.f, 0.f, 0.f, 0.f};
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    if (lane == 0 && mean_o) { mean_o[row] = mean; rstd_o[row] = rstd; }
    uint2* yr = reinterpret_cast<uint2*>(y + (size_t)row * D);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = g4[idx], b = b4[idx];
            yr[idx] = uint2{pack2bf((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y),
                            pack2bf((v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w)};
        }
    }
}

int layernorm_fwd_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                         int rows, int D, float eps, hipStream_t s) {
    if (D % 4 || D > 64 * 4 * 4) return SPLICE_ERR_ARG;
    hipLaunchKernelGGL(layernorm_fwd_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, D, eps);
    return SPLICE_OK;
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dy * gamma
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i, const float* __restrict__ g_in,
                                                            float* __restrict__ g_out, bf16_t* __restrict__ g_out_bf,
                                                            int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    const float4* dr = reinterpret_cast<const float4*>(dy + (size_t)row * D);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float mean = mean_i[row], rstd = rstd_i[row];
    float4 xh[MAXV], dh[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 xv = xr[idx], dv = dr[idx], g = g4[idx];
            xh[i] = float4{(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
            dh[i] = float4{dv.x * g.x, dv.y * g.y, dv.z * g.z, dv.w * g.w};
            s1 += dh[i].x + dh[i].y + dh[i].z + dh[i].w;
            s2 += dh[i].x * xh[i].x + dh[i].y * xh[i].y + dh[i].z * xh[i].z + dh[i].w * xh[i].w;
        } else {
            xh[i] = float4{0.f, 0.f, 0.f, 0.f};
            dh[i] = xh[i];
        }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    const float4* gi = g_in ? reinterpret_cast<const float4*>(g_in + (size_t)row * D) : nullptr;
    float4* go = reinterpret_cast<float4*>(g_out + (size_t)row * D);
    uint2* gb = g_out_bf ? reinterpret_cast<uint2*>(g_out_bf + (size_t)row * D) : nullptr;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            float4 r = float4{rstd * (dh[i].x - m1 - xh[i].x * m2), rstd * (dh[i].y - m1 - xh[i].y * m2),
                              rstd * (dh[i].z - m1 - xh[i].z * m2), rstd * (dh[i].w - m1 - xh[i].w * m2)};
            if (gi) { const float4 a = gi[idx]; r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w; }
            go[idx] = r;
            if (gb) gb[idx] = uint2{pack2bf(r.x, r.y), pack2bf(r.z, r.w)};
        }
    }
}

int layernorm_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* g_in, float* g_out, bf16_t* g_out_bf, int rows, int D, hipStream_t s) {
    if (D % 4 || D > 64 * 4 * 4) return SPLICE_ERR_ARG;
    hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, dy, x, gamma, mean, rstd, g_in, g_out, g_out_bf, rows, D);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_istd[3] = {1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f};

// One thread per (row, c, py) -> p contiguous pixels (p = 8 or 16): coalesced-ish reads of
// image rows, 16/32-byte writes of the patch matrix.
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ patches, int B, int H, int W, int p,
                                int Tld, int normalize) {
    const int gw = W / p, gh = H / p, T = 1 + gw * gh;
    const int Kp = 3 * p * p;
    const size_t total = (size_t)B * Tld * 3 * p;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int py = idx % p;
    const int c = (idx / p) % 3;
    const size_t row = idx / (3 * p);
    const int t = row % Tld, b = row / Tld;
    bf16_t* dst = patches + row * Kp + c * p * p + py * p;
    if (t == 0 || t >= T) {
        for (int i = 0; i < p; ++i) dst[i] = 0;
        return;
    }
    const int pi = t - 1, gy = pi / gw, gx = pi % gw;
    const float* src = img + (((size_t)b * 3 + c) * H + gy * p + py) * W + gx * p;
    const float mu = normalize ? c_mean[c] : 0.f, is = normalize ? c_istd[c] : 1.f;
    for (int i = 0; i < p; ++i) dst[i] = f2bf((src[i] - mu) * is);
}

int patchify_launch(const float* img, bf16_t* patches, int B, int H, int W, int p, int Tld, int normalize, hipStream_t s) {
    const size_t total = (size_t)B * Tld * 3 * p;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, patches, B, H, W, p, Tld, normalize);
    return SPLICE_OK;
}

// every pixel belongs to exactly one patch: gather form, one thread per pixel.
__global__ void unpatchify_kernel(const float* __restrict__ dp, float* __restrict__ dimg, int B, int H, int W, int p,
                                  int Tld, int normalize) {
    const size_t total = (size_t)B * 3 * H * W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = idx % W, y = (idx / W) % H, c = (idx / ((size_t)W * H)) % 3, b = idx / ((size_t)3 * W * H);
    const int gw = W / p, gh = H / p;
    float v = 0.f;
    if (x < gw * p && y < gh * p) {
        const int t = 1 + (y / p) * gw + (x / p);
        v = dp[((size_t)b * Tld + t) * (3 * p * p) + c * p * p + (y % p) * p + (x % p)];
        if (normalize) v *= c_istd[c];
    }
    dimg[idx] = v;
}

int unpatchify_launch(const float* dpatches, float* dimg, int B, int H, int W, int p, int Tld, int normalize, hipStream_t s) {
    const size_t total = (size_t)B * 3 * H * W;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpatches, dimg, B, H, W, p, Tld, normalize);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = f2bf(x[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = bf2f(x[i]);
}
__global__ void fill_f32_kernel(float* x, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = v;
}
__global__ void add_f32_kernel(float* y, const float* x, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] += x[i];
}
__global__ void transpose_f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int rows, int cols, int ldy) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? x[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) y[(size_t)c * ldy + r] = f2bf(tile[tx][i]);
    }
}

static inline unsigned grid_for(size_t n) { size_t g = (n + 255) / 256; return (unsigned)(g > 4096 ? 4096 : (g ? g : 1)); }

int cast_f32_bf16_launch(const float* x, bf16_t* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    return SPLICE_OK;
}
int cast_bf16_f32_launch(const bf16_t* x, float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    return SPLICE_OK;
}
int fill_f32_launch(float* x, float v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, v, n);
    return SPLICE_OK;
}
int add_f32_launch(float* y, const float* x, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, n);
    return SPLICE_OK;
}
int transpose_f32_to_bf16_launch(const float* x, bf16_t* y, int rows, int cols, int ldy, hipStream_t s) {
    hipLaunchKernelGGL(transpose_f32_to_bf16_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, s, x, y, rows, cols, ldy);
    return SPLICE_OK;
}
