// K4: row softmax over a materialised [rows, cols] matrix and its backward (gfx950).
//
// Replaces F.softmax(f_WTA.squeeze(), dim=-1) (correspondence.py:307, and the transposed
// variants at :338/:351 once the caller hands in the transposed matrix) on the materialised
// fallback path.  HBM-bound: each row is read once and written once — the row lives in registers
// between the max, the sum and the normalise (wavefront reductions over 64 lanes, no re-read).
//
// One WAVE per row when cols <= 4096 (64 floats per lane, 4 rows per workgroup, no LDS, no
// barrier); one WORKGROUP per row up to 16384 columns; a 3-pass streaming kernel beyond that.
#include "common.h"

namespace cocos {

constexpr int SM_VPT = 16;   // float4 per thread held in registers

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// TPR = threads per row (64 or 256).  BWD = false: p = softmax(s).  BWD = true: ds = p*(dp - sum(p*dp)).
template <int TPR, bool BWD>
__global__ __launch_bounds__(256) void row_softmax_reg_kernel(const float* __restrict__ in0,
                                                              const float* __restrict__ in1,
                                                              float* __restrict__ outp,
                                                              int64_t rows, int cols) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int tr = tid % TPR;                               // thread index inside the row
    // wave-uniform by construction; readfirstlane makes that provable so the buffer descriptors
    // below stay in SGPRs (no waterfall loop around every load, guide T20)
    const int sub = (TPR == 256) ? 0 : __builtin_amdgcn_readfirstlane(tid / TPR);
    const int64_t row = (int64_t)blockIdx.x * (256 / TPR) + sub;
    const bool live = row < rows;
    const int64_t rclamp = live ? row : rows - 1;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in0 + rclamp * cols, (size_t)cols * 4);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc((BWD ? in1 : in0) + rclamp * cols, (size_t)cols * 4);
    const bool vec = (cols % 4 == 0);
    const int nvec = (cols + 3) / 4;

    f32x4 a[SM_VPT], g[BWD ? SM_VPT : 1];
#pragma unroll
    for (int u = 0; u < SM_VPT; ++u) {
        const int q = u * TPR + tr;   // float4 index in the row
        f32x4 x, y;
        if (vec) {
            const unsigned off = q < nvec ? (unsigned)q * 16u : kBufOob;
            x = buf_load4(r0, off);
            if (BWD) y = buf_load4(r1, off);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned off = (q * 4 + e) < cols ? (unsigned)(q * 4 + e) * 4u : kBufOob;
                x[e] = buf_load1(r0, off);
                if (BWD) y[e] = buf_load1(r1, off);
            }
        }
        if (!BWD) {   // padding must not win the max nor add to the sum
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q * 4 + e >= cols) x[e] = -INFINITY;
        }
        a[u] = x;
        if (BWD) g[u] = y;
    }

    auto block_reduce = [&](float v, bool is_max) {
        v = is_max ? wave_max(v) : wave_sum(v);
        if (TPR == 256) {
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = v;
            __syncthreads();
            v = is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))
                       : (red[0] + red[1]) + (red[2] + red[3]);
        }
        return v;
    };

    if (!BWD) {
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < SM_VPT; ++u)
            m = fmaxf(m, fmaxf(fmaxf(a[u].x, a[u].y), fmaxf(a[u].z, a[u].w)));
        m = block_reduce(m, true);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < SM_VPT; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[u][e] = fast_exp2((a[u][e] - m) * kLog2e);
                sum += a[u][e];
            }
        }
        sum = block_reduce(sum, false);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int u = 0; u < SM_VPT; ++u) a[u] *= inv;
    } else {
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < SM_VPT; ++u)
            dot += a[u].x * g[u].x + a[u].y * g[u].y + a[u].z * g[u].z + a[u].w * g[u].w;
        dot = block_reduce(dot, false);
#pragma unroll
        for (int u = 0; u < SM_VPT; ++u) a[u] = a[u] * (g[u] - dot);
    }

    if (!live) return;
    float* orow = outp + row * cols;
#pragma unroll
    for (int u = 0; u < SM_VPT; ++u) {
        const int q = u * TPR + tr;
        if (vec) {
            if (q < nvec) *reinterpret_cast<f32x4*>(orow + (size_t)q * 4) = a[u];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q * 4 + e < cols) orow[q * 4 + e] = a[u][e];
        }
    }
}

// Streaming fallback for very long rows (> 16384 columns): 3 passes, one workgroup per row.
template <bool BWD>
__global__ __launch_bounds__(256) void row_softmax_stream_kernel(const float* __restrict__ in0,
                                                                 const float* __restrict__ in1,
                                                                 float* __restrict__ outp,
                                                                 int cols) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const float* x = in0 + (size_t)blockIdx.x * cols;
    const float* y = BWD ? in1 + (size_t)blockIdx.x * cols : nullptr;
    float* o = outp + (size_t)blockIdx.x * cols;
    auto block_reduce = [&](float v, bool is_max) {
        v = is_max ? wave_max(v) : wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))
                      : (red[0] + red[1]) + (red[2] + red[3]);
    };
    if (!BWD) {
        float m = -INFINITY;
        for (int j = tid; j < cols; j += 256) m = fmaxf(m, x[j]);
        m = block_reduce(m, true);
        float sum = 0.f;
        for (int j = tid; j < cols; j += 256) sum += fast_exp2((x[j] - m) * kLog2e);
        sum = block_reduce(sum, false);
        const float inv = 1.0f / sum;
        for (int j = tid; j < cols; j += 256) o[j] = fast_exp2((x[j] - m) * kLog2e) * inv;
    } else {
        float dot = 0.f;
        for (int j = tid; j < cols; j += 256) dot += x[j] * y[j];
        dot = block_reduce(dot, false);
        for (int j = tid; j < cols; j += 256) o[j] = x[j] * (y[j] - dot);
    }
}

template <bool BWD>
static int launch_row_softmax(const float* in0, const float* in1, float* outp, int64_t rows,
                              int cols, hipStream_t s) {
    COCOS_REQUIRE(rows >= 1 && cols >= 1, COCOS_ERR_INVALID, "row_softmax: bad dims rows=%lld cols=%d",
                  (long long)rows, cols);
    if (cols <= 64 * SM_VPT * 4) {
        const int64_t nblk = (rows + 3) / 4;
        COCOS_REQUIRE(nblk <= 0x7fffffff, COCOS_ERR_UNSUPPORTED, "row_softmax: too many rows");
        hipLaunchKernelGGL((row_softmax_reg_kernel<64, BWD>), dim3((unsigned)nblk), dim3(256), 0, s,
                           in0, in1, outp, rows, cols);
    } else if (cols <= 256 * SM_VPT * 4) {
        COCOS_REQUIRE(rows <= 0x7fffffff, COCOS_ERR_UNSUPPORTED, "row_softmax: too many rows");
        hipLaunchKernelGGL((row_softmax_reg_kernel<256, BWD>), dim3((unsigned)rows), dim3(256), 0, s,
                           in0, in1, outp, rows, cols);
    } else {
        COCOS_REQUIRE(rows <= 0x7fffffff, COCOS_ERR_UNSUPPORTED, "row_softmax: too many rows");
        hipLaunchKernelGGL((row_softmax_stream_kernel<BWD>), dim3((unsigned)rows), dim3(256), 0, s,
                           in0, in1, outp, cols);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

}  // namespace cocos

extern "C" int cocos_row_softmax_fwd(const float* s, float* p, int64_t rows, int cols,
                                     cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(s && p, COCOS_ERR_INVALID, "row_softmax_fwd: null pointer");
    return launch_row_softmax<false>(s, nullptr, p, rows, cols, as_stream(stream));
}

extern "C" int cocos_row_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows,
                                     int cols, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(p && dp && ds, COCOS_ERR_INVALID, "row_softmax_bwd: null pointer");
    return launch_row_softmax<true>(p, dp, ds, rows, cols, as_stream(stream));
}
