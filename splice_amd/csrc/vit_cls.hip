// The top ViT block computed only where the Splice losses read it (DESIGN.md section 10 of round 1).
//
// util/losses.py uses two things of the last block: the KEYS of every token (qkv of layer 11: structure + identity terms) and
// the [CLS] row of its output (appearance term, util/losses.py:90).  Everything of layer 11 behind the QKV projection --
// attention, proj, LayerNorm, fc1, fc2 -- is therefore needed for ONE query row per pass; the reference computes all T rows
// because a hooked nn.Module cannot do less.  The fused step runs that tail on the [CLS] rows only:
//   forward : attention of the [CLS] query over all keys (this file) -> proj / fc1 / fc2 GEMMs with M = passes on strided rows
//             (the GEMM takes any row stride) -> LayerNorm of the strided rows (this file);
//   backward: the same chain on one row per pass; the attention backward for a single query is a rank-1 update of dK / dV
//             plus one dQ row (this file).
// The façade (VitExtractor) keeps the full layer: its callers may read any token of block 11.
#include "kernels.h"

#define LOG2E 1.4426950408889634f
constexpr int CLS_NV = 26;   // token-contiguous 16-byte vectors a thread prefetches: 26 x 32 tokens covers Tld = 800 (ViT-B/8 @ 224)

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
// acc + a.lo * b.lo + a.hi * b.hi on two packed bf16 pairs (v_dot2c_f32_bf16): one instruction where the unpack-and-FMA form
// needs four -- these kernels run once per launch, their size is instruction fetch time on the critical chain
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, a), __builtin_bit_cast(bf16x2v, b), acc, false);
}

// grid (H, B): softmax(q_cls . k_j * scale) v over the T keys of one (pass, head), fp32 statistics, bf16 probabilities in the
// second product (as the full attention kernel).
//   qkv  bf16 [B*Tld][3D]; qkvT bf16 [3D][ldt] (token-contiguous V for the second product)
//   out  bf16 [B][D] (compact: one row per pass); probs fp32 [B][H][Tld] (saved for the backward; 0 beyond T)
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ qkvT, int ldt, int T, int Tld, int D,
                                                           float scale, bf16_t* __restrict__ out, float* __restrict__ probs) {
    extern __shared__ __attribute__((aligned(16))) float cls_smem[];   // [Tld] scores / probabilities | [Tld/2] packed bf16 probabilities | [256] partial o | [4]
    float* p = cls_smem;
    uint32_t* pb = reinterpret_cast<uint32_t*>(p + Tld);
    float* po = reinterpret_cast<float*>(pb + Tld / 2);
    float* red = po + 256;
    const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x;
    const size_t row0 = (size_t)b * Tld;
    // Everything that does not depend on a result is requested up front (the kernel is a chain of memory round trips on the
    // critical path of the step): the query row, the first CLS_NV value vectors of this thread's (d, token quarter), four key rows.
    const int d = threadIdx.x & 63, part = threadIdx.x >> 6;
    const bf16_t* vt = qkvT + (size_t)(2 * D + h * 64 + d) * ldt + row0;
    u32x4 qv[8], vpre[CLS_NV], kv[4][8];
    {
        const u32x4* qr = reinterpret_cast<const u32x4*>(qkv + row0 * 3 * D + h * 64);
#pragma unroll
        for (int c = 0; c < 8; ++c) qv[c] = qr[c];
    }
#pragma unroll
    for (int i = 0; i < CLS_NV; ++i) vpre[i] = *reinterpret_cast<const u32x4*>(vt + min(8 * part + 32 * i, Tld - 8));
    auto load_keys = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32x4* kr = reinterpret_cast<const u32x4*>(qkv + (row0 + min(j0 + 256 * u, T - 1)) * 3 * D + D + h * 64);
#pragma unroll
            for (int c = 0; c < 8; ++c) kv[u][c] = kr[c];
        }
    };
    load_keys(threadIdx.x);
    float mx = -1e30f;
    for (int j0 = threadIdx.x; j0 < T; j0 += 4 * 256) {
        if (j0 != (int)threadIdx.x) load_keys(j0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 256 * u;
            if (j >= T) break;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) s = dot2bf(qv[c][e], kv[u][c][e], s);
            s *= scale;
            p[j] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = block_max256(mx, red);
    float sum = 0.f;
    for (int j = threadIdx.x; j < Tld; j += 256) {
        const float e = j < T ? __builtin_amdgcn_exp2f((p[j] - mx) * LOG2E) : 0.f;
        p[j] = e;
        sum += e;
    }
    sum = block_sum256(sum, red);
    const float inv = 1.0f / sum;
    float* pg = probs + ((size_t)b * H + h) * Tld;
    for (int j2 = threadIdx.x; j2 < Tld / 2; j2 += 256) {   // two tokens per thread: fp32 for the backward, packed bf16 for the product
        const float v0 = p[2 * j2] * inv, v1 = p[2 * j2 + 1] * inv;
        *reinterpret_cast<float2*>(pg + 2 * j2) = float2{v0, v1};
        pb[j2] = pack2bf(v0, v1);
    }
    __syncthreads();
    // o[d] = sum_j p_j v_j[d]: thread = (d, quarter of the tokens), token-contiguous reads from the transposed copy;
    // 8 tokens (16 bytes) per load; p is zero beyond T and the padding columns are finite
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < CLS_NV; ++i) {
        const int j = 8 * part + 32 * i;
        if (j < Tld) {
            const u32x4 pv = *reinterpret_cast<const u32x4*>(pb + j / 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = dot2bf(pv[e], vpre[i][e], acc);
        }
    }
#pragma unroll 2
    for (int j = 8 * part + 32 * CLS_NV; j < Tld; j += 32) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(vt + j);
        const u32x4 pv = *reinterpret_cast<const u32x4*>(pb + j / 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = dot2bf(pv[e], v[e], acc);
    }
    po[part * 64 + d] = acc;
    __syncthreads();
    if (threadIdx.x < 64) out[(size_t)b * D + h * 64 + threadIdx.x] = f2bf((po[threadIdx.x] + po[64 + threadIdx.x]) + (po[128 + threadIdx.x] + po[192 + threadIdx.x]));
}

// grid (H, B, nz): backward of the above for dO given at the [CLS] query only.  Writes the WHOLE [Tld] x (q | k | v head slices) block
// of dqkv bf16 [B*Tld][3D]: dq on the [CLS] row (zero elsewhere), dk_j = ds_j q_cls, dv_j = p_j dO, zero rows beyond T.
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ qkvT, int ldt, int T, int Tld, int D,
                                                           float scale, const float* __restrict__ probs, const float* __restrict__ dout /* n_slabs x [*][D] fp32 */,
                                                           int n_slabs, size_t slab_stride, bf16_t* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float cls_smem[];   // [Tld] dP / ds | [Tld/2] packed bf16 ds | [64] q | [64] dO | [32] packed dO | [256] partial dq | [4]
    float* ds = cls_smem;
    uint32_t* dsb = reinterpret_cast<uint32_t*>(ds + Tld);
    float* q = reinterpret_cast<float*>(dsb + Tld / 2);
    float* dO = q + 64;
    uint32_t* dOb = reinterpret_cast<uint32_t*>(dO + 64);
    float* pq = reinterpret_cast<float*>(dOb + 32);
    float* red = pq + 256;
    // blockIdx.z: the token rows this workgroup WRITES (307 KB of dqkv rows per (pass, head) are the bulk of the kernel: spread over
    // gridDim.z workgroups; each recomputes the cheap reductions, z = 0 also forms the dq row).  Same values whatever gridDim.z is.
    const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x, z = blockIdx.z, nz = gridDim.z;
    const size_t row0 = (size_t)b * Tld;
    const float* pg = probs + ((size_t)b * H + h) * Tld;
    if (threadIdx.x < 64) {
        q[threadIdx.x] = bf2f(qkv[row0 * 3 * D + h * 64 + threadIdx.x]);
        // the proj^T GEMM ran split-K over many workgroups (M = passes: the K walk is its whole run time): slabs summed here, in
        // order, all of them requested first
        float sv[16];
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) sv[sl] = dout[(size_t)min(sl, n_slabs - 1) * slab_stride + (size_t)b * D + h * 64 + threadIdx.x];
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl)
            if (sl < n_slabs) v += sv[sl];
        const bf16_t vb = f2bf(v);   // (bf16 like the full path's dout)
        dO[threadIdx.x] = bf2f(vb);
        reinterpret_cast<bf16_t*>(dOb)[threadIdx.x] = vb;
    }
    // (requests that depend on nothing computed here go out first: see the forward)
    const int d = threadIdx.x & 63, part = threadIdx.x >> 6;
    const bf16_t* kt = qkvT + (size_t)(D + h * 64 + d) * ldt + row0;
    u32x4 kpre[CLS_NV], vv[4][8];
    float pj4[4];
    if (z == 0) {
#pragma unroll
        for (int i = 0; i < CLS_NV; ++i) kpre[i] = *reinterpret_cast<const u32x4*>(kt + min(8 * part + 32 * i, Tld - 8));
    }
    auto load_values = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jc = min(j0 + 256 * u, T - 1);
            const u32x4* vr = reinterpret_cast<const u32x4*>(qkv + (row0 + jc) * 3 * D + 2 * D + h * 64);
#pragma unroll
            for (int c = 0; c < 8; ++c) vv[u][c] = vr[c];
            pj4[u] = pg[jc];
        }
    };
    load_values(threadIdx.x);
    __syncthreads();
    u32x4 dov[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dov[c] = *reinterpret_cast<const u32x4*>(dOb + 4 * c);
    // dP_j = dO . v_j ; delta = sum_j p_j dP_j
    float dl = 0.f;
    for (int j0 = threadIdx.x; j0 < T; j0 += 4 * 256) {
        if (j0 != (int)threadIdx.x) load_values(j0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 256 * u;
            if (j >= T) break;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) s = dot2bf(dov[c][e], vv[u][c][e], s);
            ds[j] = s;
            dl += pj4[u] * s;
        }
    }
    const float delta = block_sum256(dl, red);
    // ds_j = p_j (dP_j - delta) * scale ; rows: dk_j = ds_j q, dv_j = p_j dO, dq_j = 0 (j > 0)
    for (int j = threadIdx.x; j < Tld; j += 256) {
        const float pj = j < T ? pg[j] : 0.f;
        const float dsj = j < T ? pj * (ds[j] - delta) * scale : 0.f;
        reinterpret_cast<bf16_t*>(dsb)[j] = f2bf(dsj);
        if ((j >> 8) % nz != z) continue;   // another workgroup writes these rows
        bf16_t* r = dqkv + (row0 + j) * 3 * D + h * 64;
        u32x4* rq = reinterpret_cast<u32x4*>(r);
        u32x4* rk = reinterpret_cast<u32x4*>(r + D);
        u32x4* rv = reinterpret_cast<u32x4*>(r + 2 * D);
#pragma unroll 2
        for (int c = 0; c < 8; ++c) {   // 16-byte stores: 8 per 64-wide head slice
            if (j != 0) rq[c] = u32x4{0u, 0u, 0u, 0u};
            rk[c] = u32x4{pack2bf(dsj * q[8 * c], dsj * q[8 * c + 1]), pack2bf(dsj * q[8 * c + 2], dsj * q[8 * c + 3]),
                          pack2bf(dsj * q[8 * c + 4], dsj * q[8 * c + 5]), pack2bf(dsj * q[8 * c + 6], dsj * q[8 * c + 7])};
            rv[c] = u32x4{pack2bf(pj * dO[8 * c], pj * dO[8 * c + 1]), pack2bf(pj * dO[8 * c + 2], pj * dO[8 * c + 3]),
                          pack2bf(pj * dO[8 * c + 4], pj * dO[8 * c + 5]), pack2bf(pj * dO[8 * c + 6], pj * dO[8 * c + 7])};
        }
    }
    if (z != 0) return;
    __syncthreads();
    // dq_cls[d] = sum_j ds_j k_j[d]  (bf16 ds, as the full kernel packs it; ds is zero beyond T)
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < CLS_NV; ++i) {
        const int j = 8 * part + 32 * i;
        if (j < Tld) {
            const u32x4 dv4 = *reinterpret_cast<const u32x4*>(dsb + j / 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = dot2bf(dv4[e], kpre[i][e], acc);
        }
    }
#pragma unroll 2
    for (int j = 8 * part + 32 * CLS_NV; j < Tld; j += 32) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(kt + j);
        const u32x4 dv4 = *reinterpret_cast<const u32x4*>(dsb + j / 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = dot2bf(dv4[e], v[e], acc);
    }
    pq[part * 64 + d] = acc;
    __syncthreads();
    if (threadIdx.x < 64)
        dqkv[row0 * 3 * D + h * 64 + threadIdx.x] = f2bf((pq[threadIdx.x] + pq[64 + threadIdx.x]) + (pq[128 + threadIdx.x] + pq[192 + threadIdx.x]));
}

int attn_cls_fwd_launch(const bf16_t* qkv, const bf16_t* qkvT, int ldt, int B, int T, int Tld, int D, int H, float scale, bf16_t* out, float* probs,
                        hipStream_t s) {
    if (D / 64 != H || D % 64) return SPLICE_ERR_ARG;
    const size_t lds = (size_t)(Tld + Tld / 2 + 256 + 8) * sizeof(float);
    SPLICE_LAUNCH(attn_cls_fwd_kernel, dim3(H, B), dim3(256), lds, s, qkv, qkvT, ldt, T, Tld, D, scale, out, probs);
    return SPLICE_OK;
}
int attn_cls_bwd_launch(const bf16_t* qkv, const bf16_t* qkvT, int ldt, int B, int T, int Tld, int D, int H, float scale, const float* probs,
                        const float* dout_slabs, int n_slabs, size_t slab_stride, bf16_t* dqkv, hipStream_t s) {
    if (D / 64 != H || D % 64 || n_slabs < 1 || n_slabs > 16) return SPLICE_ERR_ARG;
    const size_t lds = (size_t)(Tld + Tld / 2 + 64 + 64 + 32 + 256 + 8) * sizeof(float);
    const int nz = Tld > 768 ? 4 : Tld > 256 ? 2 : 1;   // one 256-row slice of the store loop per workgroup at ViT-B/8 @ 224
    SPLICE_LAUNCH(attn_cls_bwd_kernel, dim3(H, B, nz), dim3(256), lds, s, qkv, qkvT, ldt, T, Tld, D, scale, probs, dout_slabs, n_slabs, slab_stride, dqkv);
    return SPLICE_OK;
}

// ---- LayerNorm on a few strided rows (one wave per row): row r at x + r * xs; statistics at [r * ss]
// slabs != null: the row is first FORMED as bias + resid_row + sum of the n_slabs split-K slabs of the producing GEMM
// (slabs[s * slab_stride + row * D + c], in slab order) and stored to x -- the M = passes GEMMs of the [CLS] tail run
// split-K over many workgroups because their run time is the serial K walk, not the rows.
// Both kernels sit on the critical chain with a few rows of work: their run time is the number of dependent memory round trips
// plus the instruction fetch of code that runs exactly once.  One WORKGROUP per row, three columns per thread: every operand
// (all split-K slabs included) is requested before the first use -- one round trip -- and the body stays a few hundred instructions.
constexpr int LNR_MAXC = 3;    // columns per thread: D <= 768 (ViT-S / ViT-B)
constexpr int LNR_SB = 16;     // split-K slabs requested per round trip
__global__ __launch_bounds__(256) void ln_rows_fwd_kernel(float* __restrict__ x, size_t xs, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16_t* __restrict__ y, size_t ys, float* __restrict__ mean_o, float* __restrict__ rstd_o, size_t ss,
                                                          int rows, int D, float eps, const float* __restrict__ slabs, int n_slabs, size_t slab_stride,
                                                          const float* __restrict__ bias, const float* __restrict__ resid, size_t rs) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    float* xr = x + (size_t)row * xs;
    float v[LNR_MAXC], gv[LNR_MAXC], bv[LNR_MAXC];
#pragma unroll
    for (int i = 0; i < LNR_MAXC; ++i) {
        const int c = min((int)threadIdx.x + 256 * i, D - 1);   // clamped: unconditional loads, surplus threads are dropped below
        gv[i] = gamma[c]; bv[i] = beta[c];
        v[i] = slabs ? bias[c] + resid[(size_t)row * rs + c] : xr[c];
    }
    if (slabs) {
        for (int s0 = 0; s0 < n_slabs; s0 += LNR_SB) {   // LNR_SB slabs requested at a time, added in slab order
            float t[LNR_SB][LNR_MAXC];
#pragma unroll
            for (int k = 0; k < LNR_SB; ++k) {
                const int sl = min(s0 + k, n_slabs - 1);
#pragma unroll
                for (int i = 0; i < LNR_MAXC; ++i) t[k][i] = slabs[(size_t)sl * slab_stride + (size_t)row * D + min((int)threadIdx.x + 256 * i, D - 1)];
            }
#pragma unroll
            for (int k = 0; k < LNR_SB; ++k)
                if (s0 + k < n_slabs) {
#pragma unroll
                    for (int i = 0; i < LNR_MAXC; ++i) v[i] += t[k][i];
                }
        }
#pragma unroll
        for (int i = 0; i < LNR_MAXC; ++i)
            if ((int)threadIdx.x + 256 * i < D) xr[threadIdx.x + 256 * i] = v[i];
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LNR_MAXC; ++i)
        if ((int)threadIdx.x + 256 * i < D) sum += v[i];
    const float mean = block_sum256(sum, red) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LNR_MAXC; ++i)
        if ((int)threadIdx.x + 256 * i < D) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = rsqrtf(block_sum256(sq, red) / (float)D + eps);
    if (threadIdx.x == 0) { mean_o[(size_t)row * ss] = mean; rstd_o[(size_t)row * ss] = rstd; }
    bf16_t* yr = y + (size_t)row * ys;
#pragma unroll
    for (int i = 0; i < LNR_MAXC; ++i)
        if ((int)threadIdx.x + 256 * i < D) yr[threadIdx.x + 256 * i] = f2bf((v[i] - mean) * rstd * gv[i] + bv[i]);
}
// g (strided like x, in place) += LN_backward(dy); g_bf = bf16(g).  dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)), dxhat = dy gamma
// n_slabs > 1: dy is given as split-K slabs (dy + s * slab_stride), summed in place into slab 0 first
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(float* __restrict__ dy, size_t dys, const float* __restrict__ x, size_t xs,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                          size_t ss, float* __restrict__ g, bf16_t* __restrict__ g_bf, int rows, int D, int n_slabs,
                                                          size_t slab_stride) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * xs;
    float* dr = dy + (size_t)row * dys;
    float* gr = g + (size_t)row * xs;
    bf16_t* gb = g_bf + (size_t)row * xs;
    const float mean = mean_i[(size_t)row * ss], rstd = rstd_i[(size_t)row * ss];
    float v[LNR_MAXC], gv[LNR_MAXC], xv[LNR_MAXC], g0[LNR_MAXC];
#pragma unroll
    for (int i = 0; i < LNR_MAXC; ++i) {
        const int c = min((int)threadIdx.x + 256 * i, D - 1);
        v[i] = dr[c]; gv[i] = gamma[c]; xv[i] = xr[c]; g0[i] = gr[c];
    }
    if (n_slabs > 1) {
        for (int s0 = 1; s0 < n_slabs; s0 += LNR_SB) {
            float t[LNR_SB][LNR_MAXC];
#pragma unroll
            for (int k = 0; k < LNR_SB; ++k) {
                const int sl = min(s0 + k, n_slabs - 1);
#pragma unroll
                for (int i = 0; i < LNR_MAXC; ++i) t[k][i] = dr[(size_t)sl * slab_stride + min((int)threadIdx.x + 256 * i, D - 1)];
            }
#pragma unroll
            for (int k = 0; k < LNR_SB; ++k)
                if (s0 + k < n_slabs) {
#pragma unroll
                    for (int i = 0; i < LNR_MAXC; ++i) v[i] += t[k][i];
                }
        }
#pragma unroll
        for (int i = 0; i < LNR_MAXC; ++i)
            if ((int)threadIdx.x + 256 * i < D) dr[threadIdx.x + 256 * i] = v[i];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LNR_MAXC; ++i)
        if ((int)threadIdx.x + 256 * i < D) {
            const float dh = v[i] * gv[i], xh = (xv[i] - mean) * rstd;
            s1 += dh;
            s2 += dh * xh;
        }
    s1 = block_sum256(s1, red) / (float)D;
    s2 = block_sum256(s2, red) / (float)D;
#pragma unroll
    for (int i = 0; i < LNR_MAXC; ++i)
        if ((int)threadIdx.x + 256 * i < D) {
            const float dh = v[i] * gv[i], xh = (xv[i] - mean) * rstd;
            const float o = g0[i] + rstd * (dh - s1 - xh * s2);
            gr[threadIdx.x + 256 * i] = o;
            gb[threadIdx.x + 256 * i] = f2bf(o);
        }
}
int ln_rows_fwd_launch(float* x, size_t xs, const float* gamma, const float* beta, bf16_t* y, size_t ys, float* mean, float* rstd, size_t ss, int rows,
                       int D, float eps, const float* slabs, int n_slabs, size_t slab_stride, const float* bias, const float* resid, size_t rs, hipStream_t s) {
    if (D > 256 * LNR_MAXC || D < 1 || rows < 1) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(ln_rows_fwd_kernel, dim3(rows), dim3(256), 0, s, x, xs, gamma, beta, y, ys, mean, rstd, ss, rows, D, eps, slabs, n_slabs, slab_stride,
                       bias, resid, rs);
    return SPLICE_OK;
}
int ln_rows_bwd_launch(float* dy, size_t dys, const float* x, size_t xs, const float* gamma, const float* mean, const float* rstd, size_t ss, float* g,
                       bf16_t* g_bf, int rows, int D, int n_slabs, size_t slab_stride, hipStream_t s) {
    if (D > 256 * LNR_MAXC || D < 1 || rows < 1) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(ln_rows_bwd_kernel, dim3(rows), dim3(256), 0, s, dy, dys, x, xs, gamma, mean, rstd, ss, g, g_bf, rows, D, n_slabs, slab_stride);
    return SPLICE_OK;
}

// ---- finishers of the split-K [CLS]-row GEMMs: out = f(bias + sum of slabs); one thread per element (rows x N is tiny)
// mode 0: out_f32[row * os + n] = v + resid[row * rs + n]                      (fc2: the block output rows)
// mode 1: pre = v; out_bf[row * N + n] = gelu(v); pre_bf[row * ps + n] = bf16(v) for rows >= pre_lo   (fc1)
// mode 2: out_bf[row * N + n] = v * gelu'(aux[row * ps + n])  (no bias)        (fc2^T)
__global__ __launch_bounds__(256) void rows_finish_kernel(int mode, const float* __restrict__ slabs, int n_slabs, size_t slab_stride, int rows, int N,
                                                          const float* __restrict__ bias, const float* __restrict__ resid, size_t rs, float* __restrict__ out_f32,
                                                          size_t os, bf16_t* __restrict__ out_bf, bf16_t* __restrict__ pre_bf, const bf16_t* __restrict__ aux,
                                                          size_t ps, int pre_lo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * N) return;
    const int row = i / N, n = i % N;
    float v = (mode != 2 && bias) ? bias[n] : 0.f;
    for (int s0 = 0; s0 < n_slabs; s0 += 8) {   // eight slabs requested at a time, added in slab order
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = slabs[(size_t)min(s0 + k, n_slabs - 1) * slab_stride + (size_t)row * N + n];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < n_slabs) v += t[k];
    }
    if (mode == 0) {
        out_f32[(size_t)row * os + n] = v + resid[(size_t)row * rs + n];
    } else if (mode == 1) {
        if (pre_bf && row >= pre_lo) pre_bf[(size_t)row * ps + n] = f2bf(v);
        out_bf[(size_t)row * N + n] = f2bf(gelu_f(v));
    } else {
        out_bf[(size_t)row * N + n] = f2bf(v * gelu_grad_f(bf2f(aux[(size_t)row * ps + n])));
    }
}
int rows_finish_launch(int mode, const float* slabs, int n_slabs, size_t slab_stride, int rows, int N, const float* bias, const float* resid, size_t rs,
                       float* out_f32, size_t os, bf16_t* out_bf, bf16_t* pre_bf, const bf16_t* aux, size_t ps, int pre_lo, hipStream_t s) {
    SPLICE_LAUNCH(rows_finish_kernel, dim3(cdiv(rows * N, 256)), dim3(256), 0, s, mode, slabs, n_slabs, slab_stride, rows, N, bias, resid, rs, out_f32, os, out_bf,
                       pre_bf, aux, ps, pre_lo);
    return SPLICE_OK;
}
