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
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4 v[MAXV], gv[MAXV], bv[MAXV];
    float sum = 0.f;
    // row, gamma and beta are all requested up front, from clamped addresses under wave-uniform guards (lane-guarded
    // loads would each wait for their own round trip)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (64 * i >= nv) continue;
        const int j = min(lane + 64 * i, nv - 1);
        v[i] = xr[j]; gv[i] = g4[j]; bv[i] = b4[j];
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (!(64 * i < nv && lane + 64 * i < nv)) v[i] = float4{0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = gv[i], b = bv[i];
            st_out(yr + idx, uint2{pack2bf((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y),
                                   pack2bf((v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w)});
        }
    }
}

int layernorm_fwd_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                         int rows, int D, float eps, hipStream_t s) {
    if (D % 4 || D > 64 * 4 * 4) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(layernorm_fwd_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, D, eps);
    return SPLICE_OK;
}

// ---- fp8 (e4m3, OCP) operand path: the LayerNorm output quantised per TOKEN where it is produced (the row is in
// registers: its amax is one wave reduction), q = fp8(y * 448 / amax), scale = amax / 448
__device__ __forceinline__ uint32_t pack4fp8(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_fwd_fp8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, uint8_t* __restrict__ y, float* __restrict__ yscale,
                                                                float* __restrict__ mean_o, float* __restrict__ rstd_o, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4 v[MAXV], gv[MAXV], bv[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (64 * i >= nv) continue;
        const int j = min(lane + 64 * i, nv - 1);
        v[i] = xr[j]; gv[i] = g4[j]; bv[i] = b4[j];
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (!(64 * i < nv && lane + 64 * i < nv)) v[i] = float4{0.f, 0.f, 0.f, 0.f};
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (lane + 64 * i < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    if (lane == 0 && mean_o) { mean_o[row] = mean; rstd_o[row] = rstd; }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (lane + 64 * i < nv) {
            const float4 g = gv[i], b = bv[i];
            v[i] = float4{(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y, (v[i].z - mean) * rstd * g.z + b.z,
                          (v[i].w - mean) * rstd * g.w + b.w};
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
        }
    }
    amax = fmaxf(wave_max(amax), 1e-20f);
    const float q = 448.0f / amax;
    if (lane == 0) yscale[row] = amax / 448.0f;
    uint32_t* yr = reinterpret_cast<uint32_t*>(y + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) yr[lane + 64 * i] = pack4fp8(v[i].x * q, v[i].y * q, v[i].z * q, v[i].w * q);
}
int layernorm_fwd_fp8_launch(const float* x, const float* gamma, const float* beta, uint8_t* y, float* yscale, float* mean, float* rstd,
                             int rows, int D, float eps, hipStream_t s) {
    if (D % 4 || D > 64 * 4 * 4) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(layernorm_fwd_fp8_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, gamma, beta, y, yscale, mean, rstd, rows, D, eps);
    return SPLICE_OK;
}
// q[r][:] = fp8(x[r][:] * 448 / amax_r), scale[r] = amax_r / 448; x fp32 or bf16 rows; one wave per row
template <class T>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const T* __restrict__ x, int ldx, uint8_t* __restrict__ q, int ldq, float* __restrict__ scale,
                                                                float* __restrict__ qnorm, int rows, int cols, size_t x_pstride, size_t q_pstride, int rows_alloc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_alloc) return;
    const T* xr = x + (size_t)blockIdx.y * x_pstride + (size_t)row * ldx;
    auto ld = [&](int c) -> float {
        if constexpr (sizeof(T) == 2) return bf2f(xr[c]);
        else return xr[c];
    };
    float amax = 0.f;
    if (row < rows)
        for (int c = lane; c < cols; c += 64) amax = fmaxf(amax, fabsf(ld(c)));
    amax = fmaxf(wave_max(amax), 1e-20f);
    const float sc = 448.0f / amax;
    uint32_t* qr = reinterpret_cast<uint32_t*>(q + (size_t)blockIdx.y * q_pstride + (size_t)row * ldq);
    float sq = 0.f;
    for (int c4 = lane; c4 < cols / 4; c4 += 64) {
        uint32_t pk = 0;
        if (row < rows) {
            pk = pack4fp8(ld(4 * c4) * sc, ld(4 * c4 + 1) * sc, ld(4 * c4 + 2) * sc, ld(4 * c4 + 3) * sc);
            const float f0 = __builtin_amdgcn_cvt_f32_fp8((int)pk, 0), f1 = __builtin_amdgcn_cvt_f32_fp8((int)pk, 1),
                        f2 = __builtin_amdgcn_cvt_f32_fp8((int)pk, 2), f3 = __builtin_amdgcn_cvt_f32_fp8((int)pk, 3);
            sq += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
        }
        qr[c4] = pk;
    }
    sq = wave_sum(sq);
    if (lane == 0) {
        if (scale) scale[(size_t)blockIdx.y * rows_alloc + row] = row < rows ? amax / 448.0f : 0.f;
        if (qnorm) qnorm[(size_t)blockIdx.y * rows_alloc + row] = sqrtf(sq);   // L2 norm of the QUANTISED row (in quantised units)
    }
}
int quantize_rows_fp8_launch(const float* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int cols, hipStream_t s) {
    if (cols % 4 || ldq % 4) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(quantize_rows_fp8_kernel<float>, dim3(cdiv(rows, 4), 1), dim3(256), 0, s, x, ldx, q, ldq, scale, (float*)nullptr, rows, cols, (size_t)0, (size_t)0, rows);
    return SPLICE_OK;
}
// bf16 rows -> fp8 rows of `pairs` problems at once (+ the norm of every quantised row): the keys of the structure loss
int quantize_keys_fp8_launch(const bf16_t* k, int ldk, size_t k_pstride, uint8_t* q, int ldq, size_t q_pstride, float* qnorm, int T, int Tp, int D,
                             int pairs, hipStream_t s) {
    if (D % 4 || ldq % 4) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(quantize_rows_fp8_kernel<bf16_t>, dim3(cdiv(Tp, 4), pairs), dim3(256), 0, s, k, ldk, q, ldq, (float*)nullptr, qnorm, T, D, k_pstride, q_pstride, Tp);
    return SPLICE_OK;
}
int quantize_rows_bf16_fp8_launch(const bf16_t* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int cols, hipStream_t s) {
    if (cols % 4 || ldq % 4) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(quantize_rows_fp8_kernel<bf16_t>, dim3(cdiv(rows, 4), 1), dim3(256), 0, s, x, ldx, q, ldq, scale, (float*)nullptr, rows, cols, (size_t)0, (size_t)0, rows);
    return SPLICE_OK;
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dy * gamma
constexpr int LN_MAX_SLABS = 4;
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i, const float* __restrict__ g_in,
                                                            float* __restrict__ g_out, bf16_t* __restrict__ g_out_bf,
                                                            int rows, int D, int n_slabs, size_t slab_stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    const float4* dr = reinterpret_cast<const float4*>(dy + (size_t)row * D);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* gi = g_in ? reinterpret_cast<const float4*>(g_in + (size_t)row * D) : nullptr;
    const float mean = mean_i[row], rstd = rstd_i[row];
    // every load of the row is issued before the first use: the guards below are wave-uniform (a lane past the end of
    // the row re-reads element nv-1 and is masked afterwards), so nothing waits between the loads -- lane-guarded loads
    // compile to one load + s_waitcnt per element and made this kernel a chain of ~20 serial memory round trips
    float4 xv[MAXV], gv[MAXV], dv[MAXV], ev[LN_MAX_SLABS - 1][MAXV], av[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (64 * i >= nv) continue;
        const int j = min(lane + 64 * i, nv - 1);
        xv[i] = xr[j]; gv[i] = g4[j]; dv[i] = dr[j];
#pragma unroll
        for (int sl = 1; sl < LN_MAX_SLABS; ++sl)
            if (sl < n_slabs) ev[sl - 1][i] = reinterpret_cast<const float4*>(dy + (size_t)sl * slab_stride + (size_t)row * D)[j];
        if (gi) av[i] = gi[j];
    }
    float4 xh[MAXV], dh[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const bool in = 64 * i < nv && lane + 64 * i < nv;
        if (64 * i < nv) {
            float4 d = dv[i];
#pragma unroll
            for (int sl = 1; sl < LN_MAX_SLABS; ++sl)   // split-K slabs of the producing GEMM, summed in slab order
                if (sl < n_slabs) { const float4 e = ev[sl - 1][i]; d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
            const float4 xv_ = xv[i], g = gv[i];
            xh[i] = float4{(xv_.x - mean) * rstd, (xv_.y - mean) * rstd, (xv_.z - mean) * rstd, (xv_.w - mean) * rstd};
            dh[i] = float4{d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w};
        }
        if (!in) { xh[i] = float4{0.f, 0.f, 0.f, 0.f}; dh[i] = xh[i]; }
        s1 += dh[i].x + dh[i].y + dh[i].z + dh[i].w;
        s2 += dh[i].x * xh[i].x + dh[i].y * xh[i].y + dh[i].z * xh[i].z + dh[i].w * xh[i].w;
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    float4* go = reinterpret_cast<float4*>(g_out + (size_t)row * D);
    uint2* gb = g_out_bf ? reinterpret_cast<uint2*>(g_out_bf + (size_t)row * D) : nullptr;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            float4 r = float4{rstd * (dh[i].x - m1 - xh[i].x * m2), rstd * (dh[i].y - m1 - xh[i].y * m2),
                              rstd * (dh[i].z - m1 - xh[i].z * m2), rstd * (dh[i].w - m1 - xh[i].w * m2)};
            if (gi) { const float4 a = av[i]; r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w; }
            st_out(go + idx, r);
            if (gb) st_out(gb + idx, uint2{pack2bf(r.x, r.y), pack2bf(r.z, r.w)});
        }
    }
}

int layernorm_bwd_slabs_launch(const float* dy, int n_slabs, size_t slab_stride, const float* x, const float* gamma, const float* mean,
                               const float* rstd, const float* g_in, float* g_out, bf16_t* g_out_bf, int rows, int D, hipStream_t s) {
    if (D % 4 || D > 64 * 4 * 4 || n_slabs < 1 || n_slabs > LN_MAX_SLABS) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(layernorm_bwd_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, dy, x, gamma, mean, rstd, g_in, g_out, g_out_bf, rows, D,
                       n_slabs, slab_stride);
    return SPLICE_OK;
}
int layernorm_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* g_in, float* g_out, bf16_t* g_out_bf, int rows, int D, hipStream_t s) {
    return layernorm_bwd_slabs_launch(dy, 1, 0, x, gamma, mean, rstd, g_in, g_out, g_out_bf, rows, D, s);
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
    SPLICE_LAUNCH(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, patches, B, H, W, p, Tld, normalize);
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
    SPLICE_LAUNCH(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpatches, dimg, B, H, W, p, Tld, normalize);
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
    SPLICE_LAUNCH(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    return SPLICE_OK;
}
int cast_bf16_f32_launch(const bf16_t* x, float* y, size_t n, hipStream_t s) {
    SPLICE_LAUNCH(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    return SPLICE_OK;
}
int fill_f32_launch(float* x, float v, size_t n, hipStream_t s) {
    SPLICE_LAUNCH(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, v, n);
    return SPLICE_OK;
}
int add_f32_launch(float* y, const float* x, size_t n, hipStream_t s) {
    SPLICE_LAUNCH(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, n);
    return SPLICE_OK;
}
int transpose_f32_to_bf16_launch(const float* x, bf16_t* y, int rows, int cols, int ldy, hipStream_t s) {
    SPLICE_LAUNCH(transpose_f32_to_bf16_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, s, x, y, rows, cols, ldy);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
// K1: bilinear resize (align_corners=False, no antialias) and its adjoint.
__device__ __forceinline__ void rs_coord(int o, float scale, int n, int& i0, int& i1, float& lam) {
    float src = ((float)o + 0.5f) * scale - 0.5f;   // scale = in / out
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i0 = i0 < n - 1 ? i0 : n - 1;
    i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    lam = src - (float)i0;
    lam = lam > 1.f ? 1.f : lam;
}
__global__ void resize_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int oh, int ow,
                                  float sy, float sx) {
    const size_t n = (size_t)planes * oh * ow;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ox = i % ow, oy = (i / ow) % oh;
        const size_t pl = i / ((size_t)ow * oh);
        const float* p = in + pl * h * w;
        int y0, y1, x0, x1;
        float ly, lx;
        rs_coord(oy, sy, h, y0, y1, ly);
        rs_coord(ox, sx, w, x0, x1, lx);
        const float top = p[y0 * w + x0] * (1.f - lx) + p[y0 * w + x1] * lx;
        const float bot = p[y1 * w + x0] * (1.f - lx) + p[y1 * w + x1] * lx;
        out[i] = top * (1.f - ly) + bot * ly;
    }
}
// gather adjoint: input pixel m receives from the output pixels whose (i0, i1) include m
__global__ void resize_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int planes, int h, int w, int oh, int ow,
                                  float sy, float sx) {
    const size_t n = (size_t)planes * h * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int mx = i % w, my = (i / w) % h;
        const size_t pl = i / ((size_t)w * h);
        const float* p = dout + pl * oh * ow;
        // candidate output range: src(o) in (m-1, m+1)  ->  o in ((m-1+0.5)/s - 0.5, (m+1+0.5)/s - 0.5)
        int oy_lo = (int)floorf(((float)my - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)my + 1.5f) / sy - 0.5f) + 1;
        int ox_lo = (int)floorf(((float)mx - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)mx + 1.5f) / sx - 0.5f) + 1;
        oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
        oy_hi = oy_hi > oh - 1 ? oh - 1 : oy_hi; ox_hi = ox_hi > ow - 1 ? ow - 1 : ox_hi;
        if (my == 0) oy_lo = 0;           // clamped sources (src < 0) all land on row/col 0
        if (mx == 0) ox_lo = 0;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int a0, a1; float l;
            rs_coord(oy, sy, h, a0, a1, l);
            float wy = 0.f;
            if (a0 == my) wy += 1.f - l;
            if (a1 == my) wy += l;
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int b0, b1; float lx;
                rs_coord(ox, sx, w, b0, b1, lx);
                float wx = 0.f;
                if (b0 == mx) wx += 1.f - lx;
                if (b1 == mx) wx += lx;
                if (wx != 0.f) row += wx * p[oy * ow + ox];
            }
            acc += wy * row;
        }
        din[i] = acc;
    }
}
int resize_bilinear_fwd_launch(const float* in, float* out, int planes, int h, int w, int oh, int ow, hipStream_t s) {
    const size_t n = (size_t)planes * oh * ow;
    SPLICE_LAUNCH(resize_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, planes, h, w, oh, ow, (float)h / (float)oh, (float)w / (float)ow);
    return SPLICE_OK;
}
int resize_bilinear_bwd_launch(const float* dout, float* din, int planes, int h, int w, int oh, int ow, hipStream_t s) {
    const size_t n = (size_t)planes * h * w;
    SPLICE_LAUNCH(resize_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, dout, din, planes, h, w, oh, ow, (float)h / (float)oh, (float)w / (float)ow);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
// Kernel-based device copy / zero (bytes % 4 == 0).  Used inside the step's captured hipGraph instead of
// hipMemcpyAsync / hipMemsetAsync: on ROCm 7.2 memcpy/memset graph nodes raced with the neighbouring kernel
// nodes of a purely linear captured chain (intermittent garbage on replay); kernel nodes alone replay cleanly.
__global__ void dev_copy_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void dev_zero_kernel(uint32_t* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = 0u;
}
int dev_copy_launch(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes & 3) return SPLICE_ERR_ARG;
    const size_t n = bytes >> 2;
    if (!n) return SPLICE_OK;
    SPLICE_LAUNCH(dev_copy_kernel, dim3(grid_for(n)), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, n);
    return SPLICE_OK;
}
int dev_zero_launch(void* dst, size_t bytes, hipStream_t s) {
    if (bytes & 3) return SPLICE_ERR_ARG;
    const size_t n = bytes >> 2;
    if (!n) return SPLICE_OK;
    SPLICE_LAUNCH(dev_zero_kernel, dim3(grid_for(n)), dim3(256), 0, s, (uint32_t*)dst, n);
    return SPLICE_OK;
}
