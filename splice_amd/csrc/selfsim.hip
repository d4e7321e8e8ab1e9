// Key self-similarity (K11/K12 of SURVEY.md): S_ij = K_i.K_j / max(|K_i||K_j|, eps),
// models/extractor.py:4-9 (attn_cosine_sim) applied to the concatenated layer-11 keys
// (models/extractor.py:158-163), forward and backward, plus the strided MSE used by the
// three losses (util/losses.py:82,93,104).
//
// Forward  = row prep (bf16 cast + norms + transposed copy) -> K K^T on the MFMA tile
//            engine with the normalising epilogue.
// Backward = E = dS + dS^T;  W_ij = E_ij / c_ij;  r_i = sum_j [n_i n_j > eps] E_ij S_ij / n_i^2;
//            dK = W K - diag(r) K   (second MFMA GEMM, K-dim = tokens, via the transposed copy).
#include <map>
#include <mutex>
#include <utility>

#include "kernels.h"

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
#define RC_SS(x) do { const int rc_ = (x); if (rc_ != SPLICE_OK) return rc_; } while (0)
// both GEMMs are ~170 workgroups walking 12-13 K slices: latency-bound, so they use the 4-stage LDS-DMA ring (64 KB)
constexpr int SS_RING = 4;
constexpr size_t SS_LDS = (size_t)SS_RING * GemmTile<64, 64>::LDS_ELEMS * sizeof(bf16_t);

size_t selfsim_ws_bytes(int T, int D) {
    const size_t Tp = round_up(T, 64);
    return Tp * D * 2 * 2 + Tp * 4 * 2 + Tp * Tp * 2 + 256;
}
void selfsim_ws_carve(void* base, int T, int D, SelfSimWs* ws) {
    const size_t Tp = round_up(T, 64);
    char* p = (char*)base;
    ws->Tp = (int)Tp;
    ws->kbf = (bf16_t*)p; p += Tp * D * 2;
    ws->kbfT = (bf16_t*)p; p += Tp * D * 2;
    ws->wmat = (bf16_t*)p; p += Tp * Tp * 2;
    ws->norm = (float*)p; p += Tp * 4;
    ws->rowdot = (float*)p;
}

// one wave per row: bf16 copy, norm of the ROUNDED row (so the diagonal is exactly cos=1)
__global__ __launch_bounds__(256) void selfsim_prep_kernel(const float* __restrict__ K, int ldk, int T, int D, SelfSimWs ws) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= ws.Tp) return;
    float sq = 0.f;
    for (int d = lane; d < D; d += 64) {
        const bf16_t h = row < T ? f2bf(K[(size_t)row * ldk + d]) : (bf16_t)0;
        ws.kbf[(size_t)row * D + d] = h;
        ws.kbfT[(size_t)d * ws.Tp + row] = h;
        const float f = bf2f(h);
        sq += f * f;
    }
    sq = wave_sum(sq);
    if (lane == 0) ws.norm[row] = sqrtf(sq);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void selfsim_gemm_kernel(const bf16_t* __restrict__ Kb, int T, int D, const float* __restrict__ norm,
                                                           float eps, float* __restrict__ S) {
    extern __shared__ __attribute__((aligned(16))) bf16_t selfsim_smem[];   // SS_RING stages
    const int tiles_n = (T + BN - 1) / BN;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    GemmTile<BM, BN> tile;
    tile.template run_ring<SS_RING>(Kb, D, Kb, D, T, T, D, m0, n0, selfsim_smem);
    tile.for_each(m0, n0, [&](int row0, int col, f32x4 v) {
        if (col >= T) return;
        const float nj = norm[col];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < T) S[(size_t)(row0 + r) * T + col] = v[r] / fmaxf(norm[row0 + r] * nj, eps);
    });
}

int selfsim_fwd_launch(const float* K, int ldk, int T, int D, float eps, float* S, const SelfSimWs& ws, hipStream_t s) {
    if (D % 64) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(selfsim_prep_kernel, dim3(cdiv(ws.Tp, 4)), dim3(256), 0, s, K, ldk, T, D, ws);
    const int grid = cdiv(T, 64) * cdiv(T, 64);
    SPLICE_LAUNCH((selfsim_gemm_kernel<64, 64>), dim3(grid), dim3(256), SS_LDS, s, ws.kbf, T, D, ws.norm, eps, S);
    return SPLICE_OK;
}

// one block per row i: W_i. = (dS_i. + dS_.i) / c_i. ; r_i
__global__ __launch_bounds__(256) void selfsim_wmat_kernel(const float* __restrict__ dS, const float* __restrict__ S, int T, float eps,
                                                           SelfSimWs ws) {
    const int i = blockIdx.x;
    __shared__ float red[4];
    bf16_t* wrow = ws.wmat + (size_t)i * ws.Tp;
    if (i >= T) {
        for (int j = threadIdx.x; j < ws.Tp; j += 256) wrow[j] = 0;
        if (threadIdx.x == 0) ws.rowdot[i] = 0.f;
        return;
    }
    const float ni = ws.norm[i];
    float acc = 0.f;
    for (int j = threadIdx.x; j < ws.Tp; j += 256) {
        float w = 0.f;
        if (j < T) {
            const float e = dS[(size_t)i * T + j] + dS[(size_t)j * T + i];
            const float nn = ni * ws.norm[j];
            w = e / fmaxf(nn, eps);
            if (nn > eps) acc += e * S[(size_t)i * T + j];
        }
        wrow[j] = f2bf(w);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws.rowdot[i] = (red[0] + red[1] + red[2] + red[3]) / fmaxf(ni * ni, 1e-30f);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void selfsim_bwd_gemm_kernel(SelfSimWs ws, int T, int D, float* __restrict__ dK, int lddk,
                                                               int accumulate) {
    extern __shared__ __attribute__((aligned(16))) bf16_t selfsim_smem[];
    const int tiles_n = (D + BN - 1) / BN;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    GemmTile<BM, BN> tile;
    tile.template run_ring<SS_RING>(ws.wmat, ws.Tp, ws.kbfT, ws.Tp, T, D, ws.Tp, m0, n0, selfsim_smem);
    tile.for_each(m0, n0, [&](int row0, int col, f32x4 v) {
        if (col >= D) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + r;
            if (row < T) {
                const float g = v[r] - ws.rowdot[row] * bf2f(ws.kbf[(size_t)row * D + col]);
                float* p = dK + (size_t)row * lddk + col;
                *p = accumulate ? *p + g : g;
            }
        }
    });
}

int selfsim_bwd_launch(const float* dS, const float* S, int T, int D, float eps, float* dK, int lddk, int accumulate,
                       const SelfSimWs& ws, hipStream_t s) {
    if (D % 64) return SPLICE_ERR_ARG;
    SPLICE_LAUNCH(selfsim_wmat_kernel, dim3(ws.Tp), dim3(256), 0, s, dS, S, T, eps, ws);
    const int grid = cdiv(T, 64) * cdiv(D, 64);
    SPLICE_LAUNCH((selfsim_bwd_gemm_kernel<64, 64>), dim3(grid), dim3(256), SS_LDS, s, ws, T, D, dK, lddk, accumulate);
    return SPLICE_OK;
}

// ---------------------------------------------------------------------------------------
// mean((a-b)^2) and its gradient.  The per-workgroup partial sums go to part[blockIdx.x] (no float atomics: the
// caller adds them in index order -- mse_sum_kernel below, or the step engine's total_loss_kernel).
constexpr int MSE_MAX_WG = SPLICE_MSE_PARTIALS;
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int rows,
                                                  int cols, float wmean, float gmean, float* __restrict__ part, float* __restrict__ grad,
                                                  int ldg) {
    const size_t n = (size_t)rows * cols;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = i / cols, c = i % cols;
        const float d = a[(size_t)r * lda + c] - b[(size_t)r * ldb + c];
        acc += d * d;
        if (grad) grad[(size_t)r * ldg + c] = 2.0f * gmean * d;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * wmean;
}
// fixed-order sum of up to MSE_MAX_WG partials (entries the launch did not write must be zero)
__device__ __forceinline__ float mse_partials_sum(const float* part, float* red /* 4 floats of LDS */) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < MSE_MAX_WG; i += 256) acc += part[i];
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void mse_sum_kernel(float* __restrict__ part, float* __restrict__ loss_accum) {
    __shared__ float red[4];
    const float t = mse_partials_sum(part, red);
    __syncthreads();
    for (int i = threadIdx.x; i < MSE_MAX_WG; i += 256) part[i] = 0.f;   // ready for the next call
    if (threadIdx.x == 0) loss_accum[0] += t;
}

// part: SPLICE_MSE_PARTIALS floats, all zero on entry except what this launch writes; the caller sums them
int mse_partials_launch(const float* a, int lda, const float* b, int ldb, int rows, int cols, float loss_weight, float grad_weight,
                        float* part, float* grad, int ldg, hipStream_t s) {
    const size_t n = (size_t)rows * cols;
    if (!n || !part) return SPLICE_ERR_ARG;
    size_t g = (n + 255) / 256;
    if (g > MSE_MAX_WG) g = MSE_MAX_WG;
    SPLICE_LAUNCH(mse_kernel, dim3((unsigned)g), dim3(256), 0, s, a, lda, b, ldb, rows, cols, loss_weight / (float)n,
                       grad_weight / (float)n, part, grad, ldg);
    return SPLICE_OK;
}
// stand-alone form: loss_accum[0] += loss_weight * mean(d^2).  The partials line is scoped to (device, stream): the sum
// kernel launched right behind the partials kernel on the same stream consumes and re-zeroes it, so calls on one stream
// are ordered by the stream and calls on different streams / devices never share a line.
static float* mse_scratch_for(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, float*> lines;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    float*& line = lines[{dev, s}];
    if (!line) {
        if (hipMalloc(&line, MSE_MAX_WG * sizeof(float)) != hipSuccess) { line = nullptr; return nullptr; }
        if (hipMemset(line, 0, MSE_MAX_WG * sizeof(float)) != hipSuccess) { (void)hipFree(line); line = nullptr; return nullptr; }
    }
    return line;
}
int mse2_launch(const float* a, int lda, const float* b, int ldb, int rows, int cols, float loss_weight, float grad_weight,
                float* loss_accum, float* grad, int ldg, hipStream_t s) {
    float* scratch = mse_scratch_for(s);
    if (!scratch) return SPLICE_ERR_HIP;
    const int rc = mse_partials_launch(a, lda, b, ldb, rows, cols, loss_weight, grad_weight, scratch, grad, ldg, s);
    if (rc != SPLICE_OK) return rc;
    SPLICE_LAUNCH(mse_sum_kernel, dim3(1), dim3(256), 0, s, scratch, loss_accum);
    return SPLICE_OK;
}
int mse_launch(const float* a, int lda, const float* b, int ldb, int rows, int cols, float weight, float* loss_accum,
               float* grad, int ldg, hipStream_t s) {
    return mse2_launch(a, lda, b, ldb, rows, cols, weight, weight, loss_accum, grad, ldg, s);
}

// =======================================================================================
// Fused, batched structure loss of the optimisation step (util/losses.py:74-83 for P pairs side by side; blockIdx.y =
// pair).  The step never needs S, S* or dS as such, only
//     loss = mean((S - S*)^2),   W = (dS + dS^T) / c,   r_i = sum_j [n_i n_j > eps] (dS + dS^T)_ij S_ij / n_i^2
// and both S and S* are symmetric, so dS = 2 lambda (S - S*) / T^2 is too:
//   * the keys are read where the layer-11 QKV GEMM left them (bf16 [rows][3D] and its transpose): no prep / cast pass;
//   * only the upper-triangular 64x64 tiles are computed; a tile (m < n) emits W_mn and its mirror W_nm, counts twice in
//     the loss, and gives row sums to r (rows of block m) and column sums (rows of block n);
//   * the MSE, the dS seed and the W matrix come out of the GEMM epilogue: 3 launches (norms, this one, dK) where the
//     unfused path has 5 (prep, S GEMM, MSE, W, dK), and 2.5 T^2 fp32 matrices less traffic.
// Every sum runs in a fixed order (per-tile partials, slots written exactly once): bit-reproducible.
__global__ __launch_bounds__(256) void selfsim_rownorm_kernel(const bf16_t* __restrict__ k, int ldk, size_t k_pstride, int T, int Tp, int D,
                                                              float* __restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Tp) return;
    float sq = 0.f;
    if (row < T) {
        const u32x4* p = reinterpret_cast<const u32x4*>(k + (size_t)blockIdx.y * k_pstride + (size_t)row * ldk);
        for (int c = lane; c < D / 8; c += 64) {
            const u32x4 v = p[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = __uint_as_float(v[q] << 16), hi = __uint_as_float(v[q] & 0xFFFF0000u);
                sq += lo * lo + hi * hi;
            }
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) norm[(size_t)blockIdx.y * Tp + row] = sqrtf(sq);
}

__device__ __forceinline__ void tri_tile(int t, int nt, int& tm, int& tn) {
    tm = 0;
#pragma unroll 1
    while (t >= nt - tm) { t -= nt - tm; ++tm; }
    tn = tm + t;
}

// upper-triangular tiles of S* = cos-sim(target keys) into the full-layout fp32 [T][T] buffer of each pair
// FP8: the Gram matrix on the fp8 MFMA from per-row quantised keys (k8, e4m3; BASELINE configs[4]).  The cosine is scale
// invariant per row, so the quantisation scales cancel against the norms of the QUANTISED rows (qnorm) exactly.
// Row / column norms of a tile from the sums of squares run_ring_norms leaves in every lane (bf16 path: the norms come out of
// the Gram kernels themselves -- the separate row-norm launch in front of each of them is gone, round 4)
template <int FMt, int FNt>
__device__ __forceinline__ void tile_norms(const float (&ssa)[FMt], const float (&ssb)[FNt], float (&nr)[FMt][4], float (&nc)[FNt]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < FMt; ++i) {
        const float na = sqrtf(ssa[i]);            // A row i * 16 + (lane & 15) of this wave's sub-tile
#pragma unroll
        for (int r = 0; r < 4; ++r) nr[i][r] = __shfl(na, (lane >> 4) * 4 + r, 64);   // the accumulator rows of this lane
    }
#pragma unroll
    for (int j = 0; j < FNt; ++j) nc[j] = sqrtf(ssb[j]);   // B row = output column j * 16 + (lane & 15)
}

template <bool FP8>
__global__ __launch_bounds__(256) void selfsim_tgt_kernel(SelfSimBatch b) {
    extern __shared__ __attribute__((aligned(16))) bf16_t selfsim_smem[];
    const int T = b.T, nt = (T + 63) / 64, pair = blockIdx.y;
    int tm, tn;
    tri_tile(xcd_remap(blockIdx.x, gridDim.x), nt, tm, tn);
    const int m0 = tm * 64, n0 = tn * 64;
    float* S = b.S_tgt + (size_t)pair * T * T;
    GemmTile<64, 64, false, FP8> tile;
    if constexpr (FP8) {
        const float* norm = b.qnorm_tgt + (size_t)pair * b.Tp;
        const bf16_t* K8 = reinterpret_cast<const bf16_t*>(b.k8_tgt + (size_t)pair * b.Tp * b.D);
        tile.template run_ring<SS_RING>(K8, b.D / 2, K8, b.D / 2, T, T, b.D / 2, m0, n0, selfsim_smem);
        tile.for_each(m0, n0, [&](int row0, int col, f32x4 v) {
            if (col >= T) return;
            const float nj = norm[col];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r < T) S[(size_t)(row0 + r) * T + col] = v[r] / fmaxf(norm[row0 + r] * nj, 1e-30f);
        });
    } else {
        const bf16_t* K = b.k_tgt + (size_t)pair * b.k_pstride;
        float ssa[2], ssb[2], nr[2][4], nc[2];
        tile.template run_ring_norms<SS_RING>(K, b.ldk, K, b.ldk, T, T, b.D, m0, n0, selfsim_smem, ssa, ssb);
        tile_norms<2, 2>(ssa, ssb, nr, nc);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row0 = m0 + wm * 32 + i * 16 + (lane >> 4) * 4, col = n0 + wn * 32 + j * 16 + (lane & 15);
                if (col >= T) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + r < T) S[(size_t)(row0 + r) * T + col] = tile.acc[i][j][r] / fmaxf(nr[i][r] * nc[j], b.eps);
            }
    }
}

template <bool FP8>
__global__ __launch_bounds__(256) void selfsim_loss_kernel(SelfSimBatch b) {
    extern __shared__ __attribute__((aligned(16))) bf16_t selfsim_smem[];
    const int T = b.T, Tp = b.Tp, nt = Tp / 64, pair = blockIdx.y;
    const int tile_id = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tri_tile(tile_id, nt, tm, tn);
    const int m0 = tm * 64, n0 = tn * 64;
    const bool offdiag = tm != tn;
    const bf16_t* K = b.k_x + (size_t)pair * b.k_pstride;
    const float* norm = b.norm_x + (size_t)pair * Tp;
    const float* St = b.S_tgt + (size_t)pair * T * T;
    bf16_t* W = b.wmat + (size_t)pair * Tp * Tp;
    GemmTile<64, 64, false, FP8> tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    constexpr int FM = 2, FN = 2;
    // operands of every fragment first (clamped, always-valid addresses), arithmetic after
    float nr[FM][4], nc[FN], st[FM][FN][4], qr[FM][4], qc[FN];
    if constexpr (FP8) {
        const bf16_t* K8 = reinterpret_cast<const bf16_t*>(b.k8_x + (size_t)pair * Tp * b.D);
        tile.template run_ring<SS_RING>(K8, b.D / 2, K8, b.D / 2, T, T, b.D / 2, m0, n0, selfsim_smem);
        const float* qnorm = b.qnorm_x + (size_t)pair * Tp;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nr[i][r] = norm[min(m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, T - 1)];
                qr[i][r] = qnorm[min(m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, T - 1)];
            }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            nc[j] = norm[min(n0 + wn * 32 + j * 16 + (lane & 15), T - 1)];
            qc[j] = qnorm[min(n0 + wn * 32 + j * 16 + (lane & 15), T - 1)];
        }
    } else {
        // bf16: the key norms are by-products of this kernel's own operand stream (no row-norm launch in front of it); the
        // diagonal tiles publish them for the dK kernel
        float ssa[FM], ssb[FN];
        tile.template run_ring_norms<SS_RING>(K, b.ldk, K, b.ldk, T, T, b.D, m0, n0, selfsim_smem, ssa, ssb);
        tile_norms<FM, FN>(ssa, ssb, nr, nc);
        if (!offdiag && wn == 0 && lane < 16) {
#pragma unroll
            for (int i = 0; i < FM; ++i) b.norm_x[(size_t)pair * Tp + m0 + wm * 32 + i * 16 + lane] = sqrtf(ssa[i]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) qr[i][r] = nr[i][r];
#pragma unroll
        for (int j = 0; j < FN; ++j) qc[j] = nc[j];
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[i][j][r] = St[(size_t)min(m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, T - 1) * T + min(n0 + wn * 32 + j * 16 + (lane & 15), T - 1)];
    float lsum = 0.f, rs[FM][4], cs[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j) cs[j] = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row0 = m0 + wm * 32 + i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int col = n0 + wn * 32 + j * 16 + (lane & 15);
            float w4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = row0 + r < T && col < T;
                const float nn = nr[i][r] * nc[j];
                const float c = fmaxf(nn, b.eps);
                const float s = FP8 ? tile.acc[i][j][r] / fmaxf(qr[i][r] * qc[j], 1e-30f) : tile.acc[i][j][r] / c;
                const float d = valid ? s - st[i][j][r] : 0.f;
                lsum += d * d;
                const float e = b.e_scale * d;                 // (dS + dS^T)_ij = 4 lambda d / T^2
                w4[r] = e / c;
                const float rd = nn > b.eps ? e * s : 0.f;
                rs[i][r] += rd;
                cs[j] += rd;
                W[(size_t)(row0 + r) * Tp + col] = f2bf(w4[r]);
            }
            if (offdiag)   // mirror tile: W[col][row0 .. row0+3], 8 contiguous bytes
                *reinterpret_cast<uint2*>(W + (size_t)col * Tp + row0) = uint2{pack2bf(w4[0], w4[1]), pack2bf(w4[2], w4[3])};
        }
    }
    // ---- reductions through LDS (the K loop is over): row sums over the 2 waves wn, column sums over the 2 waves wm
    float* red = reinterpret_cast<float*>(selfsim_smem);   // [2][64] row sums | [2][64] column sums | [4] loss
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = rs[i][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if ((lane & 15) == 0) red[wn * 64 + wm * 32 + i * 16 + (lane >> 4) * 4 + r] = v;
        }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        float v = cs[j];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lane < 16) red[128 + wm * 64 + wn * 32 + j * 16 + lane] = v;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[256 + wave] = lsum;
    __syncthreads();
    float* rpart = b.rpart + (size_t)pair * nt * Tp;
    if (threadIdx.x < 64) rpart[(size_t)tn * Tp + m0 + threadIdx.x] = red[threadIdx.x] + red[64 + threadIdx.x];
    else if (threadIdx.x < 128 && offdiag) rpart[(size_t)tm * Tp + n0 + threadIdx.x - 64] = red[128 + threadIdx.x - 64] + red[192 + threadIdx.x - 64];
    if (threadIdx.x == 0)
        b.loss_part[(size_t)pair * b.part_pstride + tile_id] = ((red[256] + red[257]) + (red[258] + red[259])) * (offdiag ? 2.0f : 1.0f) * b.loss_scale;
}

// dK = W K - diag(r) K for every pair: dk[pair][T][lddk] fp32
__global__ __launch_bounds__(256) void selfsim_dk_kernel(SelfSimBatch b) {
    extern __shared__ __attribute__((aligned(16))) bf16_t selfsim_smem[];
    __shared__ float rdot[64];
    const int T = b.T, Tp = b.Tp, D = b.D, nt = Tp / 64, pair = blockIdx.y;
    const int tiles_n = D / 64;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (t / tiles_n) * 64, n0 = (t % tiles_n) * 64;
    const float* norm = b.norm_x + (size_t)pair * Tp;
    if (threadIdx.x < 64) {
        const int row = m0 + threadIdx.x;
        const float* rp = b.rpart + (size_t)pair * nt * Tp + row;
        float acc = 0.f;
        for (int k = 0; k < nt; ++k) acc += rp[(size_t)k * Tp];
        const float ni = norm[row];
        rdot[threadIdx.x] = row < T ? acc / fmaxf(ni * ni, 1e-30f) : 0.f;
    }
    __syncthreads();   // (the ring's raw s_barrier does not wait for LDS stores)
    const bf16_t* W = b.wmat + (size_t)pair * Tp * Tp;
    const bf16_t* KT = b.kT_x + (size_t)pair * b.kT_pstride;
    const bf16_t* K = b.k_x + (size_t)pair * b.k_pstride;
    float* dK = b.dk + (size_t)pair * b.dk_pstride;
    GemmTile<64, 64> tile;
    tile.template run_ring<SS_RING>(W, Tp, KT, b.ldt, T, D, Tp, m0, n0, selfsim_smem);
    tile.for_each(m0, n0, [&](int row0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + r;
            if (row < T) dK[(size_t)row * b.lddk + col] = v[r] - rdot[row - m0] * bf2f(K[(size_t)row * b.ldk + col]);
        }
    });
}

int selfsim_norms_launch(const bf16_t* k, int ldk, size_t k_pstride, int T, int D, float* norm, int pairs, hipStream_t s) {
    if (D % 64 || ldk % 8) return SPLICE_ERR_ARG;
    const int Tp = round_up(T, 64);
    SPLICE_LAUNCH(selfsim_rownorm_kernel, dim3(cdiv(Tp, 4), pairs), dim3(256), 0, s, k, ldk, k_pstride, T, Tp, D, norm);
    return SPLICE_OK;
}
int selfsim_target_launch(const SelfSimBatch& b, hipStream_t s) {
    SpliceProfScope prof_scope(8); SPLICE_DEV_REGION(18);
    const int nt = b.Tp / 64;
    if (b.fp8) {
        if (b.D % 128) return SPLICE_ERR_ARG;
        RC_SS(quantize_keys_fp8_launch(b.k_tgt, b.ldk, b.k_pstride, b.k8_tgt, b.D, (size_t)b.Tp * b.D, b.qnorm_tgt, b.T, b.Tp, b.D, b.pairs, s));
        SPLICE_LAUNCH(selfsim_tgt_kernel<true>, dim3(nt * (nt + 1) / 2, b.pairs), dim3(256), SS_LDS, s, b);
        return SPLICE_OK;
    }
    SPLICE_LAUNCH(selfsim_tgt_kernel<false>, dim3(nt * (nt + 1) / 2, b.pairs), dim3(256), SS_LDS, s, b);
    return SPLICE_OK;
}
int selfsim_loss_launch(const SelfSimBatch& b, hipStream_t s) {
    SpliceProfScope prof_scope(8); SPLICE_DEV_REGION(18);
    const int nt = b.Tp / 64;
    if (nt * (nt + 1) / 2 > (int)b.part_pstride) return SPLICE_ERR_ARG;
    if (b.fp8) {
        if (b.D % 128) return SPLICE_ERR_ARG;
        RC_SS(selfsim_norms_launch(b.k_x, b.ldk, b.k_pstride, b.T, b.D, b.norm_x, b.pairs, s));   // true norms: W and r are in key units either way
        RC_SS(quantize_keys_fp8_launch(b.k_x, b.ldk, b.k_pstride, b.k8_x, b.D, (size_t)b.Tp * b.D, b.qnorm_x, b.T, b.Tp, b.D, b.pairs, s));
        SPLICE_LAUNCH(selfsim_loss_kernel<true>, dim3(nt * (nt + 1) / 2, b.pairs), dim3(256), SS_LDS, s, b);
    } else {
        SPLICE_LAUNCH(selfsim_loss_kernel<false>, dim3(nt * (nt + 1) / 2, b.pairs), dim3(256), SS_LDS, s, b);
    }
    SPLICE_LAUNCH(selfsim_dk_kernel, dim3(nt * (b.D / 64), b.pairs), dim3(256), SS_LDS, s, b);
    return SPLICE_OK;
}
size_t selfsim_batch_ws_bytes(int T, int D, int pairs) {
    const size_t Tp = round_up(T, 64), nt = Tp / 64;
    return (size_t)pairs * (4 * Tp * 4 + nt * Tp * 4 + Tp * Tp * 2 + (size_t)T * T * 4 + 2 * Tp * D) + 2048;
}
void selfsim_batch_carve(void* base, int T, int D, int pairs, SelfSimBatch* b) {
    const size_t Tp = round_up(T, 64), nt = Tp / 64;
    char* p = (char*)base;
    b->T = T; b->Tp = (int)Tp; b->D = D; b->pairs = pairs;
    b->S_tgt = (float*)p; p += (size_t)pairs * T * T * 4; p = (char*)(((size_t)p + 255) & ~(size_t)255);
    b->wmat = (bf16_t*)p; p += (size_t)pairs * Tp * Tp * 2;
    b->norm_tgt = (float*)p; p += (size_t)pairs * Tp * 4;
    b->norm_x = (float*)p; p += (size_t)pairs * Tp * 4;
    b->qnorm_tgt = (float*)p; p += (size_t)pairs * Tp * 4;
    b->qnorm_x = (float*)p; p += (size_t)pairs * Tp * 4;
    b->rpart = (float*)p; p += (size_t)pairs * nt * Tp * 4; p = (char*)(((size_t)p + 255) & ~(size_t)255);
    b->k8_tgt = (uint8_t*)p; p += (size_t)pairs * Tp * D;
    b->k8_x = (uint8_t*)p;
}

// ---- batched strided MSE (blockIdx.y = pair): the [CLS] and key-identity terms of P pairs in one launch each
__global__ __launch_bounds__(256) void mse_batched_kernel(const float* __restrict__ a, int lda, size_t a_ps, const float* __restrict__ bb, int ldb,
                                                          size_t b_ps, int rows, int cols, float wmean, float gmean, float* __restrict__ part,
                                                          size_t part_ps, float* __restrict__ grad, int ldg, size_t g_ps) {
    const int pair = blockIdx.y;
    a += (size_t)pair * a_ps; bb += (size_t)pair * b_ps; part += (size_t)pair * part_ps;
    if (grad) grad += (size_t)pair * g_ps;
    const size_t n = (size_t)rows * cols;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = i / cols, c = i % cols;
        const float d = a[(size_t)r * lda + c] - bb[(size_t)r * ldb + c];
        acc += d * d;
        if (grad) grad[(size_t)r * ldg + c] = 2.0f * gmean * d;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * wmean;
}
int mse_batched_launch(const float* a, int lda, size_t a_ps, const float* b, int ldb, size_t b_ps, int rows, int cols, float loss_weight,
                       float grad_weight, float* part, size_t part_ps, float* grad, int ldg, size_t g_ps, int pairs, hipStream_t s) {
    const size_t n = (size_t)rows * cols;
    if (!n || !part || pairs < 1) return SPLICE_ERR_ARG;
    size_t g = (n + 255) / 256;
    if (g > MSE_MAX_WG) g = MSE_MAX_WG;
    SPLICE_LAUNCH(mse_batched_kernel, dim3((unsigned)g, pairs), dim3(256), 0, s, a, lda, a_ps, b, ldb, b_ps, rows, cols, loss_weight / (float)n,
                       grad_weight / (float)n, part, part_ps, grad, ldg, g_ps);
    return SPLICE_OK;
}
