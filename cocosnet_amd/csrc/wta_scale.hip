// K8: WTA_scale (+ the 1/temperature that follows it) on a materialised [rows, cols] correlation (gfx950).
//
// Replaces the autograd.Function at correspondence.py:38-77 as applied at :300-303, and folds the
// `/temperature` of :304 into the same pass:
//   fwd: y = (x == rowmax(x) ? x : x*scale) * post          mask bit = (x == rowmax(x))
//   bwd: dx = dy * post * (mask ? 1 : 1e-4)                  (the reference's hard-coded 1e-4, :72 —
//                                                             NOT the derivative of the forward)
// The reference keeps the input and a float mask for backward (2 x 512 MiB at B=8, HW=4096); here the
// mask is one BIT per element, produced by the wavefront ballot (64 columns -> one 64-bit word), so the
// backward reads dy + 1/32 of it.  HBM-bound: fwd = 1 read + 1 write (the second sweep over the row
// for the write hits L2: a row is <= 64 KB), bwd = 1 read + 1 write.
//
// One workgroup (4 waves) per row; a wave owns 64 consecutive columns per step, so every load/store
// instruction covers one contiguous 256-byte segment.
#include "common.h"

namespace cocos {

__device__ __forceinline__ float wta_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void wta_scale_fwd_kernel(const float* __restrict__ x,
                                                            float* __restrict__ y,
                                                            unsigned long long* __restrict__ mask,
                                                            int cols, int words, float scale, float post) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const float* xr = x + row * cols;
    float* yr = y + row * cols;
    unsigned long long* mr = mask + row * words;

    float m = -INFINITY;
    for (int c = tid; c < cols; c += 256) m = fmaxf(m, xr[c]);
    m = wta_wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

    const int steps = (cols + 255) / 256;
    for (int s = 0; s < steps; ++s) {
        const int c = s * 256 + tid;
        const bool in = c < cols;
        const float v = in ? xr[c] : 0.f;
        const bool is_max = in && (v == m);          // NaN rows: no maxima, exactly like torch's `==`
        const unsigned long long bits = __ballot(is_max);
        if (in) yr[c] = (is_max ? v : v * scale) * post;
        if ((tid & 63) == 0 && (c >> 6) < words) mr[c >> 6] = bits;
    }
}

__global__ __launch_bounds__(256) void wta_scale_bwd_kernel(const float* __restrict__ dy,
                                                            const unsigned long long* __restrict__ mask,
                                                            float* __restrict__ dx, int cols, int words,
                                                            float post, float off_grad) {
    const int tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const float* gr = dy + row * cols;
    float* dr = dx + row * cols;
    const unsigned long long* mr = mask + row * words;
    for (int c = tid; c < cols; c += 256) {
        const unsigned long long bits = mr[c >> 6];             // wave-uniform address
        const bool is_max = (bits >> (c & 63)) & 1ull;
        dr[c] = gr[c] * post * (is_max ? 1.0f : off_grad);
    }
}

}  // namespace cocos

extern "C" long long cocos_wta_scale_mask_bytes(long long rows, int cols) {
    if (rows < 1 || cols < 1) return -1;
    return rows * (long long)((cols + 63) / 64) * 8;
}

extern "C" int cocos_wta_scale_fwd(const float* x, float* y, void* mask, long long rows, int cols,
                                   float scale, float post_scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && y && mask, COCOS_ERR_INVALID, "wta_scale_fwd: null pointer");
    COCOS_REQUIRE(rows >= 1 && cols >= 1 && rows <= 0x7fffffffLL, COCOS_ERR_INVALID,
                  "wta_scale_fwd: bad dims rows=%lld cols=%d", rows, cols);
    hipLaunchKernelGGL(wta_scale_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), x, y,
                       static_cast<unsigned long long*>(mask), cols, (cols + 63) / 64, scale, post_scale);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_wta_scale_bwd(const float* dy, const void* mask, float* dx, long long rows, int cols,
                                   float post_scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(dy && dx && mask, COCOS_ERR_INVALID, "wta_scale_bwd: null pointer");
    COCOS_REQUIRE(rows >= 1 && cols >= 1 && rows <= 0x7fffffffLL, COCOS_ERR_INVALID,
                  "wta_scale_bwd: bad dims rows=%lld cols=%d", rows, cols);
    hipLaunchKernelGGL(wta_scale_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), dy,
                       static_cast<const unsigned long long*>(mask), dx, cols, (cols + 63) / 64, post_scale,
                       1e-4f /* correspondence.py:72 */);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
