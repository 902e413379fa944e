// K21: the weight of torch.nn.utils.spectral_norm — one power iteration (v <- normalize(W^T u), u <- normalize(W v)),
// sigma = u . (W v), W / sigma — and its backward, as four + two launches instead of the framework's ~20 (two gemv, norms, clamps,
// divisions, dot, clones; and the autograd of the division / dot / mv): every convolution of the reference's generator,
// discriminator and of the feature producers of netCorr is wrapped in it (normalization.py:21-61 `get_nonspade_norm_layer`,
// architecture.py:41-52), 24 layers per step of the module scope = ~2 ms of launch-bound glue.  gfx950.
//
// W is the weight viewed as [R = out channels][K = everything else] (dim = 0, contiguous).  u [R], v [K] are the module's buffers,
// updated IN PLACE exactly as the framework does (DataParallel replicas share their storage).  Memory-bound on three reads of W.
#include "common.h"

namespace cocos {

constexpr int SN_ROWS = 32;      // rows per workgroup of W^T u (R / 32 x K / 256 workgroups, one partial row of t each: the first
                                 // version, 64 rows per thread and one workgroup reducing the slices, took 58 us per layer — slower
                                 // than the framework's gemv chain; round 3 had 8 rows and atomics: not reproducible run to run)

__device__ __forceinline__ float sn_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// part[blockIdx.y][k] = sum over the workgroup's 32 rows of u[r] W[r][k]     grid (ceil(K / 256), ceil(R / 32))
// Deterministic: every partial is written once and sn_v_kernel adds them in a fixed order — the buffers u / v, and
// with them W / sigma, are bit-reproducible run to run and rank to rank, as torch's mv is.
__global__ __launch_bounds__(256) void sn_wtu_partial_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                             float* __restrict__ part, int R, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * SN_ROWS;
    if (k >= K) return;
    float acc = 0.f;
#pragma unroll 8
    for (int j = 0; j < SN_ROWS; ++j)
        if (r0 + j < R) acc += u[r0 + j] * W[(size_t)(r0 + j) * K + k];
    part[(size_t)blockIdx.y * K + k] = acc;
}

// v = normalize(t), t[k] = sum of the RS partial rows      ONE workgroup of 1024 threads, K <= 16 * 1024
__global__ __launch_bounds__(1024) void sn_v_kernel(const float* __restrict__ part, int RS, float* __restrict__ v, int K, float eps) {
    __shared__ float red[16];
    float t[16];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = j * 1024 + threadIdx.x;
        float a = 0.f;
        if (k < K)
            for (int r = 0; r < RS; ++r) a += part[(size_t)r * K + k];
        t[j] = a;
        ss += t[j] * t[j];
    }
    const float nrm = sqrtf(sn_block_sum(ss, red));
    const float d = fmaxf(nrm, eps);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = j * 1024 + threadIdx.x;
        if (k < K) v[k] = t[j] / d;
    }
}

// s[r] = W[r] . v        one workgroup per row
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ v, float* __restrict__ s,
                                                    int R, int K) {
    __shared__ float red[4];
    const float* w = W + (size_t)blockIdx.x * K;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) acc += w[k] * v[k];
    const float tot = sn_block_sum(acc, red);
    if (threadIdx.x == 0) s[blockIdx.x] = tot;
}

// sigma from s (every workgroup recomputes it: R values), optionally u <- normalize(s) (workgroup 0 writes), then Wsn = W / sigma
__global__ __launch_bounds__(256) void sn_apply_kernel(const float* __restrict__ W, const float* __restrict__ s, float* __restrict__ u,
                                                       float* __restrict__ wsn, float* __restrict__ sigma_out, int R, size_t n,
                                                       float eps, int update_u, unsigned* __restrict__ amax_cell) {
    __shared__ float red[4];
    float a = 0.f;
    if (update_u) {
        for (int r = threadIdx.x; r < R; r += 256) a += s[r] * s[r];
    } else {
        for (int r = threadIdx.x; r < R; r += 256) a += u[r] * s[r];
    }
    const float tot = sn_block_sum(a, red);
    float sigma;
    if (update_u) {
        const float d = fmaxf(sqrtf(tot), eps);           // u = s / d, sigma = u . s = |s|^2 / d
        sigma = tot / d;
        if (blockIdx.x == 0)
            for (int r = threadIdx.x; r < R; r += 256) u[r] = s[r] / d;
    } else {
        sigma = tot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *sigma_out = sigma;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float q = W[i] / sigma;
        wsn[i] = q;
        m = fmaxf(m, fabsf(q));
    }
    if (amax_cell) {          // max|W / sigma| for the consumer's f16 split (one atomic per workgroup, <= 512 of them: finding 14)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(amax_cell, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    }
}

// backward:  dW = G / sigma - (sum(G o W) / sigma^2) u v^T      (W the ORIGINAL weight; u, v the vectors sigma was taken with)
__global__ __launch_bounds__(256) void sn_gw_partial_kernel(const float* __restrict__ G, const float* __restrict__ W,
                                                            float* __restrict__ part, size_t n) {
    __shared__ float red[4];
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += G[i] * W[i];
    const float tot = sn_block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ u,
                                                           const float* __restrict__ v, const float* __restrict__ part, int nparts,
                                                           const float* __restrict__ sigma_p, float* __restrict__ dW, int K, size_t n) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
    const float gw = sn_block_sum(a, red);
    const float sigma = *sigma_p, c = gw / (sigma * sigma);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / (size_t)K), k = (int)(i - (size_t)r * K);
        dW[i] = G[i] / sigma - c * u[r] * v[k];
    }
}

}  // namespace cocos

static_assert(cocos::SN_ROWS == 32, "cocos_spectral_weight_workspace_floats assumes 32 rows per partial");
extern "C" long long cocos_spectral_weight_workspace_floats(int R, int K) {
    if (R < 1 || K < 1) return 0;
    return (long long)K * ((R + 31) / 32) + R;     // the partial rows of t = W^T u, then s = W v
}

// wsn = W / sigma with sigma = u . (W v); power_iteration != 0: first v <- normalize(W^T u), u <- normalize(W v) in place (one
// iteration: torch.nn.utils.spectral_norm's n_power_iterations = 1).  sigma_out: device float (kept for the backward).
extern "C" int cocos_spectral_weight_fwd(const float* W, float* u, float* v, float* wsn, float* sigma_out, float* amax_inout_dev,
                                         float* workspace, int R, int K, float eps, int power_iteration, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(W && u && v && wsn && sigma_out && workspace, COCOS_ERR_INVALID, "spectral_weight_fwd: null pointer");
    COCOS_REQUIRE(R >= 1 && K >= 1 && eps >= 0.f, COCOS_ERR_INVALID, "spectral_weight_fwd: bad dims R=%d K=%d", R, K);
    COCOS_REQUIRE(K <= 16 * 1024 && (long long)R * K < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "spectral_weight_fwd: matrix too large");
    hipStream_t st = as_stream(stream);
    const int RS = (R + SN_ROWS - 1) / SN_ROWS;
    float* t = workspace;
    float* s = workspace + (size_t)K * RS;
    if (power_iteration) {
        hipLaunchKernelGGL(sn_wtu_partial_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)RS), dim3(256), 0, st, W, u, t, R, K);
        hipLaunchKernelGGL(sn_v_kernel, dim3(1), dim3(1024), 0, st, t, RS, v, K, eps);
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3((unsigned)R), dim3(256), 0, st, W, v, s, R, K);
    const size_t n = (size_t)R * K;
    const size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    const size_t cap = amax_inout_dev ? 512 : 2048;
    hipLaunchKernelGGL(sn_apply_kernel, dim3((unsigned)(blocks > cap ? cap : blocks)), dim3(256), 0, st, W, s, u, wsn, sigma_out, R, n,
                       eps, power_iteration ? 1 : 0, reinterpret_cast<unsigned*>(amax_inout_dev));
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// dW = G / sigma - (sum(G o W) / sigma^2) u v^T;  workspace: 1024 floats
extern "C" int cocos_spectral_weight_bwd(const float* G, const float* W, const float* u, const float* v, const float* sigma_dev,
                                         float* dW, float* workspace, int R, int K, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(G && W && u && v && sigma_dev && dW && workspace, COCOS_ERR_INVALID, "spectral_weight_bwd: null pointer");
    COCOS_REQUIRE(R >= 1 && K >= 1 && (long long)R * K < 0x7fffffffLL, COCOS_ERR_INVALID, "spectral_weight_bwd: bad dims R=%d K=%d", R, K);
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)R * K;
    size_t nparts = (n + 256 * 8 - 1) / (256 * 8);
    if (nparts > 1024) nparts = 1024;
    hipLaunchKernelGGL(sn_gw_partial_kernel, dim3((unsigned)nparts), dim3(256), 0, st, G, W, workspace, n);
    size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, G, u, v, workspace,
                       (int)nparts, sigma_dev, dW, K, n);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
