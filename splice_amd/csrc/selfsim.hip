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
    hipLaunchKernelGGL(selfsim_prep_kernel, dim3(cdiv(ws.Tp, 4)), dim3(256), 0, s, K, ldk, T, D, ws);
    const int grid = cdiv(T, 64) * cdiv(T, 64);
    hipLaunchKernelGGL((selfsim_gemm_kernel<64, 64>), dim3(grid), dim3(256), SS_LDS, s, ws.kbf, T, D, ws.norm, eps, S);
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
    hipLaunchKernelGGL(selfsim_wmat_kernel, dim3(ws.Tp), dim3(256), 0, s, dS, S, T, eps, ws);
    const int grid = cdiv(T, 64) * cdiv(D, 64);
    hipLaunchKernelGGL((selfsim_bwd_gemm_kernel<64, 64>), dim3(grid), dim3(256), SS_LDS, s, ws, T, D, dK, lddk, accumulate);
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
    hipLaunchKernelGGL(mse_kernel, dim3((unsigned)g), dim3(256), 0, s, a, lda, b, ldb, rows, cols, loss_weight / (float)n,
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
    hipLaunchKernelGGL(mse_sum_kernel, dim3(1), dim3(256), 0, s, scratch, loss_accum);
    return SPLICE_OK;
}
int mse_launch(const float* a, int lda, const float* b, int ldb, int rows, int cols, float weight, float* loss_accum,
               float* grad, int ldg, hipStream_t s) {
    return mse2_launch(a, lda, b, ldb, rows, cols, weight, weight, loss_accum, grad, ldg, s);
}
