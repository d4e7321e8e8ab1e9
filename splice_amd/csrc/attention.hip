// Fused multi-head self-attention (head dim 64) for the DINO ViT, forward + backward,
// gfx950.  Replaces the DINO `Attention.forward` the reference reaches through
// `self.model(input_img)` (models/extractor.py:83,91,99): softmax(q k^T / sqrt(d)) v,
// without materialising the [h,T,T] probability tensors the reference's hooks keep.
//
// All products are "swapped" 16x16x32 bf16 MFMAs so that every lane owns ONE query (or
// one key) column of the score tile: softmax statistics are per-lane scalars, the
// probabilities feed the second MFMA straight from registers (no LDS, no shuffles beyond
// a 4-lane-group reduce), and the k-index of the second product is permuted consistently
// on both operands (lane group g, slot j  <->  key g*4+j | 16+g*4+(j-4)).
//
// Inputs: qkv  bf16 [B*Tld][3D]  (row = b*Tld + token; q | k | v column blocks, heads
//                                 contiguous inside each block -- the layout
//                                 models/extractor.py:136-151 reshapes)
//         qkvT bf16 [3D][ldt]    the same matrix transposed (written by the QKV GEMM
//                                 epilogue) -- supplies the token-contiguous operands.
// Tokens t >= T inside a pass are padding: masked as keys, harmless as queries.
#include "kernels.h"


#define NEG_BIG (-1.0e30f)

__device__ __forceinline__ uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 ld8x2(const bf16_t* p) {  // tokens [0,4) and [16,20) relative to p
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 16);
    return uint4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ uint4 pack8(const f32x4& a, const f32x4& b) {
    return uint4{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
}
__device__ __forceinline__ void st4bf(bf16_t* p, const f32x4& v, float s) {
    *reinterpret_cast<uint2*>(p) = uint2{pack2bf(v[0] * s, v[1] * s), pack2bf(v[2] * s, v[3] * s)};
}
__device__ __forceinline__ float group4_max(float v) {  // across the 4 lane groups (same lane&15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------
// forward: workgroup = 4 waves, wave = QB blocks of 16 queries; loop over 32-key tiles.
template <int QB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int ld = 3 * a.D;
    const int qbase = blockIdx.x * (64 * QB) + wave * (16 * QB);
    if (qbase >= a.Tld) return;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.Tld * ld;
    uint4 qf[QB][2];
    int qidx[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int q = qbase + qb * 16 + c;
        qidx[qb] = q;
        q = q < a.Tld ? q : a.Tld - 1;
        const bf16_t* p = qkv_b + (size_t)q * ld + h * 64 + g * 8;
        qf[qb][0] = ld16(p);
        qf[qb][1] = ld16(p + 32);
    }
    float m[QB], l[QB];
    f32x4 o[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = NEG_BIG;
        l[qb] = 0.f;
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) o[qb][nd] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bf16_t* kbase = qkv_b + a.D + h * 64 + g * 8;
    const bf16_t* vT = a.qkvT + (size_t)(2 * a.D + h * 64 + c) * a.ldt + (size_t)b * a.Tld + g * 4;
    for (int kt = 0; kt < a.Tld; kt += 32) {
        uint4 kf[2][2], vf[4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const bf16_t* p = kbase + (size_t)(kt + nb * 16 + c) * ld;
            kf[nb][0] = ld16(p);
            kf[nb][1] = ld16(p + 32);
        }
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) vf[nd] = ld8x2(vT + (size_t)(nd * 16) * a.ldt + kt);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x4 s[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                s[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                s[nb] = mfma16(kf[nb][0], qf[qb][0], s[nb]);
                s[nb] = mfma16(kf[nb][1], qf[qb][1], s[nb]);
            }
            float mx = NEG_BIG;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + nb * 16 + g * 4 + r;
                    const float v = key < a.T ? s[nb][r] * a.scale : NEG_BIG;
                    s[nb][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = group4_max(mx);
            const float mn = fmaxf(m[qb], mx);
            const float alpha = __expf(m[qb] - mn);
            m[qb] = mn;
            float ps = 0.f;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(s[nb][r] - mn);
                    s[nb][r] = p;
                    ps += p;
                }
            l[qb] = l[qb] * alpha + ps;
            const uint4 pb = pack8(s[0], s[1]);
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][nd][r] *= alpha;
                o[qb][nd] = mfma16(vf[nd], pb, o[qb][nd]);
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lt = group4_sum(l[qb]);
        const int q = qidx[qb];
        if (q < a.Tld) {
            const float inv = 1.0f / lt;
            bf16_t* op = a.out + ((size_t)b * a.Tld + q) * a.D + h * 64 + g * 4;
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) st4bf(op + nd * 16, o[qb][nd], inv);
            if (g == 0) a.lse[((size_t)b * a.H + h) * a.Tld + q] = m[qb] + __logf(lt);
        }
    }
}

// ---------------------------------------------------------------------------------------
// delta[b][h][q] = sum_d dO[q][h*64+d] * O[q][h*64+d]
__global__ void attn_delta_kernel(AttnArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over B*Tld*H
    const int total = a.B * a.Tld * a.H;
    if (idx >= total) return;
    const int h = idx % a.H, row = idx / a.H;
    const int b = row / a.Tld, q = row % a.Tld;
    const bf16_t* po = a.out + (size_t)row * a.D + h * 64;
    const bf16_t* pd = a.dout + (size_t)row * a.D + h * 64;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 x = ld16(po + i * 8), y = ld16(pd + i * 8);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc += bf2f((bf16_t)(xs[j] & 0xFFFF)) * bf2f((bf16_t)(ys[j] & 0xFFFF));
            acc += bf2f((bf16_t)(xs[j] >> 16)) * bf2f((bf16_t)(ys[j] >> 16));
        }
    }
    a.delta[((size_t)b * a.H + h) * a.Tld + q] = acc;
}

// ---------------------------------------------------------------------------------------
// backward, dK / dV: wave owns 16 keys (lane&15 = key), loops over 32-query tiles.
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int ld = 3 * a.D;
    const int kbase_idx = blockIdx.x * 64 + wave * 16;
    if (kbase_idx >= a.Tld) return;
    const int key = kbase_idx + c;
    const int keyc = key < a.Tld ? key : a.Tld - 1;
    const bool key_valid = key < a.T;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.Tld * ld;
    uint4 kf[2], vf[2];
    {
        const bf16_t* pk = qkv_b + (size_t)keyc * ld + a.D + h * 64 + g * 8;
        const bf16_t* pv = qkv_b + (size_t)keyc * ld + 2 * a.D + h * 64 + g * 8;
        kf[0] = ld16(pk); kf[1] = ld16(pk + 32);
        vf[0] = ld16(pv); vf[1] = ld16(pv + 32);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) { dk[nd] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[nd] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const bf16_t* qrow = qkv_b + h * 64 + g * 8;                                   // + q*ld
    const bf16_t* dorow = a.dout + (size_t)b * a.Tld * a.D + h * 64 + g * 8;       // + q*D
    const bf16_t* qT = a.qkvT + (size_t)(h * 64 + c) * a.ldt + (size_t)b * a.Tld + g * 4;
    const bf16_t* doT = a.doutT + (size_t)(h * 64 + c) * a.ldt + (size_t)b * a.Tld + g * 4;
    const float* lse = a.lse + ((size_t)b * a.H + h) * a.Tld;
    const float* dl = a.delta + ((size_t)b * a.H + h) * a.Tld;
    for (int qt = 0; qt < a.Tld; qt += 32) {
        f32x4 s[2], dp[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const bf16_t* pq = qrow + (size_t)(qt + qb * 16 + c) * ld;
            const bf16_t* pd = dorow + (size_t)(qt + qb * 16 + c) * a.D;
            const uint4 q0 = ld16(pq), q1 = ld16(pq + 32), d0 = ld16(pd), d1 = ld16(pd + 32);
            s[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            s[qb] = mfma16(q0, kf[0], s[qb]);      // S[q][key]: rows q = qb*16+g*4+r, col = key
            s[qb] = mfma16(q1, kf[1], s[qb]);
            dp[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[qb] = mfma16(d0, vf[0], dp[qb]);    // dP[q][key]
            dp[qb] = mfma16(d1, vf[1], dp[qb]);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const f32x4 ls = *reinterpret_cast<const f32x4*>(lse + qt + qb * 16 + g * 4);
            const f32x4 de = *reinterpret_cast<const f32x4*>(dl + qt + qb * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = key_valid ? __expf(s[qb][r] * a.scale - ls[r]) : 0.f;
                s[qb][r] = p;
                dp[qb][r] = p * (dp[qb][r] - de[r]);
            }
        }
        const uint4 pb = pack8(s[0], s[1]), dsb = pack8(dp[0], dp[1]);
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) {
            dv[nd] = mfma16(ld8x2(doT + (size_t)(nd * 16) * a.ldt + qt), pb, dv[nd]);   // dV^T[d][key]
            dk[nd] = mfma16(ld8x2(qT + (size_t)(nd * 16) * a.ldt + qt), dsb, dk[nd]);   // dK^T[d][key]
        }
    }
    if (key < a.Tld) {
        bf16_t* pk = a.dqkv + ((size_t)b * a.Tld + key) * ld + a.D + h * 64 + g * 4;
        bf16_t* pv = a.dqkv + ((size_t)b * a.Tld + key) * ld + 2 * a.D + h * 64 + g * 4;
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) {
            st4bf(pk + nd * 16, dk[nd], a.scale);
            st4bf(pv + nd * 16, dv[nd], 1.0f);
        }
    }
}

// backward, dQ: wave owns 16 queries (lane&15 = query), loops over 32-key tiles.
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int ld = 3 * a.D;
    const int qbase = blockIdx.x * 64 + wave * 16;
    if (qbase >= a.Tld) return;
    const int q = qbase + c;
    const int qc = q < a.Tld ? q : a.Tld - 1;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.Tld * ld;
    uint4 qf[2], dof[2];
    {
        const bf16_t* pq = qkv_b + (size_t)qc * ld + h * 64 + g * 8;
        const bf16_t* pd = a.dout + ((size_t)b * a.Tld + qc) * a.D + h * 64 + g * 8;
        qf[0] = ld16(pq); qf[1] = ld16(pq + 32);
        dof[0] = ld16(pd); dof[1] = ld16(pd + 32);
    }
    const float lse_q = a.lse[((size_t)b * a.H + h) * a.Tld + qc];
    const float del_q = a.delta[((size_t)b * a.H + h) * a.Tld + qc];
    f32x4 dq[4];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) dq[nd] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* krow = qkv_b + a.D + h * 64 + g * 8;
    const bf16_t* vrow = qkv_b + 2 * a.D + h * 64 + g * 8;
    const bf16_t* kT = a.qkvT + (size_t)(a.D + h * 64 + c) * a.ldt + (size_t)b * a.Tld + g * 4;
    for (int kt = 0; kt < a.Tld; kt += 32) {
        f32x4 s[2], dp[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const bf16_t* pk = krow + (size_t)(kt + nb * 16 + c) * ld;
            const bf16_t* pv = vrow + (size_t)(kt + nb * 16 + c) * ld;
            const uint4 k0 = ld16(pk), k1 = ld16(pk + 32), v0 = ld16(pv), v1 = ld16(pv + 32);
            s[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            s[nb] = mfma16(k0, qf[0], s[nb]);      // S^T[key][q]
            s[nb] = mfma16(k1, qf[1], s[nb]);
            dp[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[nb] = mfma16(v0, dof[0], dp[nb]);   // dP^T[key][q]
            dp[nb] = mfma16(v1, dof[1], dp[nb]);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + nb * 16 + g * 4 + r;
                const float p = key < a.T ? __expf(s[nb][r] * a.scale - lse_q) : 0.f;
                dp[nb][r] = p * (dp[nb][r] - del_q);
            }
        const uint4 dsb = pack8(dp[0], dp[1]);
#pragma unroll
        for (int nd = 0; nd < 4; ++nd)
            dq[nd] = mfma16(ld8x2(kT + (size_t)(nd * 16) * a.ldt + kt), dsb, dq[nd]);   // dQ^T[d][q]
    }
    if (q < a.Tld) {
        bf16_t* p = a.dqkv + ((size_t)b * a.Tld + q) * ld + h * 64 + g * 4;
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) st4bf(p + nd * 16, dq[nd], a.scale);
    }
}

// attention probabilities of one layer, fp32 [B][H][T][T] (models/extractor.py:44-45,97-103);
// only materialised when the API asks for them.
__global__ void attn_probs_kernel(AttnArgs a, float* probs) {
    const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int ld = 3 * a.D;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.Tld * ld;
    const bf16_t* pq = qkv_b + (size_t)q * ld + h * 64;
    const float lse = a.lse[((size_t)b * a.H + h) * a.Tld + q];
    float qv[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) qv[d] = bf2f(pq[d]);
    float* out = probs + (((size_t)b * a.H + h) * a.T + q) * a.T;
    for (int key = threadIdx.x; key < a.T; key += blockDim.x) {
        const bf16_t* pk = qkv_b + (size_t)key * ld + a.D + h * 64;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) s += qv[d] * bf2f(pk[d]);
        out[key] = __expf(s * a.scale - lse);
    }
}

// ---------------------------------------------------------------------------------------
int attn_fwd_launch(const AttnArgs* a, hipStream_t s) {
    if (a->Tld % 32 || a->D % 64 || a->D / 64 != a->H || a->ldt % 4) return SPLICE_ERR_ARG;
    constexpr int QB = 2;
    dim3 grid(cdiv(a->Tld, 64 * QB), a->H, a->B);
    hipLaunchKernelGGL(attn_fwd_kernel<QB>, grid, dim3(256), 0, s, *a);
    return SPLICE_OK;
}

int attn_bwd_launch(const AttnArgs* a, hipStream_t s) {
    if (a->Tld % 32 || a->D % 64 || a->D / 64 != a->H || a->ldt % 4) return SPLICE_ERR_ARG;
    const int total = a->B * a->Tld * a->H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, *a);
    dim3 grid(cdiv(a->Tld, 64), a->H, a->B);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(256), 0, s, *a);
    hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(256), 0, s, *a);
    return SPLICE_OK;
}

int attn_probs_launch(const AttnArgs* a, float* probs, hipStream_t s) {
    dim3 grid(a->T, a->H, a->B);
    hipLaunchKernelGGL(attn_probs_kernel, grid, dim3(128), 0, s, *a, probs);
    return SPLICE_OK;
}
