// K15: the row reductions of the contextual loss (gfx950).
//
// SURVEY.md §8(f) rank 3 — `ContextualLoss_forward.forward` (models/networks/ContextualLoss.py:93-137) after its
// cosine-similarity matmul (:121, our K3 `cocos_corr_materialize`):
//     d      = 1 - cos                                   [rows = positions of X, cols = positions of Y]     :121
//     d_norm = d / (min_j d + eps)                       eps = 1e-3                                         :125
//     w      = exp((1 - d_norm) / h)                                                                        :128
//     A      = w / sum_j w                                                                                  :129
//     cx_i   = max_j A_ij                                (then CX = mean_i cx_i, loss = -log CX: host side) :132-133
// With m = max_j cos (so min_j d = 1 - m) and s = 1 / (h (1 - m + eps)) every row is a softmax of cos * s whose
// temperature depends on its own maximum, and   cx_i = 1 / sum_j exp((cos_ij - m_i) s_i).
// The framework formulation makes five passes over the [N, N] matrix and keeps four of them for autograd; here a row
// is read ONCE into registers (one wave per row, cols <= 4096: 64 floats per lane), reduced with wave shuffles, and
// the backward recomputes the row statistics from the same single read:
//     Z = sum_j e_j,  e_j = exp((c_j - m) s),  cx = 1/Z,  ds/dm = h s^2
//     dcx/dc_j = -(s e_j)/Z^2                                                     (j != argmax)
//     dcx/dc_j* = -(s e_j*)/Z^2 - (1/Z^2) sum_j e_j ((c_j - m) h s^2 - s)         (through m and s; ties: first index)
// HBM-bound: 4 B/element forward, 8 B/element backward.
#include "common.h"

namespace cocos {

constexpr int CX_VPT = 16;   // float4 per lane: rows up to 4096 columns

__device__ __forceinline__ float cx_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float cx_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int cx_wave_min_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

template <bool BWD>
__global__ __launch_bounds__(256) void contextual_rows_kernel(const float* __restrict__ cosm,   // [rows, cols]
                                                              const float* __restrict__ dcx,    // [rows] (BWD)
                                                              float* __restrict__ outp,         // cx [rows] | dcos [rows, cols]
                                                              int64_t rows, int cols, float h, float eps) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row = (int64_t)blockIdx.x * 4 + sub;
    if (row >= rows) return;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(cosm + row * cols, (size_t)cols * 4);
    const bool vec = (cols % 4 == 0) && (reinterpret_cast<uintptr_t>(cosm) & 15u) == 0;
    float c[CX_VPT][4];
#pragma unroll
    for (int u = 0; u < CX_VPT; ++u) {
        const int q = u * 64 + lane;           // float4 index in the row
        if (vec) {
            const f32x4 x = buf_load4(rs, q * 4 < cols ? (unsigned)q * 16u : kBufOob);
#pragma unroll
            for (int e = 0; e < 4; ++e) c[u][e] = x[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[u][e] = buf_load1(rs, (q * 4 + e) < cols ? (unsigned)(q * 4 + e) * 4u : kBufOob);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q * 4 + e >= cols) c[u][e] = -INFINITY;         // padding never wins the max, adds exp(-inf) = 0
    }
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < CX_VPT; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, c[u][e]);
    m = cx_wave_max(m);
    const float s = 1.0f / (h * (1.0f - m + eps));
    const float s2 = s * kLog2e;
    float z = 0.f, t = 0.f;        // Z and sum_j e_j (c_j - m)
#pragma unroll
    for (int u = 0; u < CX_VPT; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dlt = c[u][e] - m;
            const float ex = fast_exp2(dlt * s2);                 // exp(-inf) = 0 for the padding
            c[u][e] = ex;
            z += ex;
            if (BWD) t += (dlt > -INFINITY) ? ex * dlt : 0.f;
        }
    z = cx_wave_sum(z);
    if (!BWD) {
        if (lane == 0) outp[row] = 1.0f / z;
        return;
    }
    t = cx_wave_sum(t);
    // first column that holds the maximum (e == 1 there: exp(0)); torch.min / torch.max route the gradient to one index
    int jstar = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < CX_VPT; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c[u][e] == 1.0f) jstar = min(jstar, (u * 64 + lane) * 4 + e);
    jstar = cx_wave_min_int(jstar);
    const float g = dcx[row];
    const float k = -g / (z * z);                                    // d cx / d Z
    const float extra = k * (t * h * s * s - s * z);                  // through m and s(m), lands on column j*
    float* orow = outp + row * cols;
#pragma unroll
    for (int u = 0; u < CX_VPT; ++u) {
        const int q = u * 64 + lane;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = k * s * c[u][e] + ((q * 4 + e) == jstar ? extra : 0.f);
        if (vec) {
            if (q * 4 < cols) *reinterpret_cast<f32x4*>(orow + (size_t)q * 4) = f32x4{o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q * 4 + e < cols) orow[q * 4 + e] = o[e];
        }
    }
}

}  // namespace cocos

// cx[i] = max_j A_ij of the contextual affinity built from the cosine matrix `cosm` [rows, cols] (see header);
// 1 <= cols <= 4096.
extern "C" int cocos_contextual_rows_fwd(const float* cosm, float* cx, long long rows, int cols, float h, float eps,
                                         cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(cosm && cx, COCOS_ERR_INVALID, "contextual_rows_fwd: null pointer");
    COCOS_REQUIRE(rows >= 1 && cols >= 1 && h > 0.f && eps > 0.f, COCOS_ERR_INVALID,
                  "contextual_rows_fwd: bad arguments rows=%lld cols=%d h=%g eps=%g", rows, cols, (double)h, (double)eps);
    COCOS_REQUIRE(cols <= 4096, COCOS_ERR_UNSUPPORTED, "contextual_rows_fwd: cols=%d > 4096", cols);
    COCOS_REQUIRE((rows + 3) / 4 <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "contextual_rows_fwd: too many rows");
    hipLaunchKernelGGL((contextual_rows_kernel<false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream),
                       cosm, nullptr, cx, (int64_t)rows, cols, h, eps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// dcos[i,j] = dcx[i] * d cx_i / d cos_ij (see header), recomputed from one read of the row.
extern "C" int cocos_contextual_rows_bwd(const float* cosm, const float* dcx, float* dcos, long long rows, int cols,
                                         float h, float eps, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(cosm && dcx && dcos, COCOS_ERR_INVALID, "contextual_rows_bwd: null pointer");
    COCOS_REQUIRE(rows >= 1 && cols >= 1 && h > 0.f && eps > 0.f, COCOS_ERR_INVALID,
                  "contextual_rows_bwd: bad arguments rows=%lld cols=%d h=%g eps=%g", rows, cols, (double)h, (double)eps);
    COCOS_REQUIRE(cols <= 4096, COCOS_ERR_UNSUPPORTED, "contextual_rows_bwd: cols=%d > 4096", cols);
    COCOS_REQUIRE((rows + 3) / 4 <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "contextual_rows_bwd: too many rows");
    hipLaunchKernelGGL((contextual_rows_kernel<true>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream),
                       cosm, dcx, dcos, (int64_t)rows, cols, h, eps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
