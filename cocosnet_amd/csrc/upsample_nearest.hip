// K11: nearest-neighbour x`d` up-sampling of the warped image and its backward (gfx950).
//
// Replaces self.upsampling (nn.Upsample(scale_factor=down), mode 'nearest'; correspondence.py:188 used at :327):
//   fwd: y[b,c,Y,X] = x[b,c,Y/d,X/d]
//   bwd: dx[b,c,y,x] = sum_{dy,dx < d} dy[b,c,y*d+dy,x*d+dx]
// Tiny and HBM-bound (8 x 3 x 256 x 256 floats at the benchmark shape); it exists because the generic framework
// kernels take 15 us (fwd) and 35 us (bwd) for 6 MB.  One thread per OUTPUT float4 (fwd) / per input element (bwd).
#include "common.h"

namespace cocos {

__global__ __launch_bounds__(256) void upsample_nearest_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   int planes, int h, int w, int d) {
    const int W = w * d, H = h * d;
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;            // float4 index in y (W % 4 == 0)
    const size_t n4 = (size_t)planes * H * (W / 4);
    if (i4 >= n4) return;
    const int X4 = (int)(i4 % (W / 4));
    const size_t r = i4 / (W / 4);
    const int Y = (int)(r % H);
    const size_t p = r / H;
    const float* xr = x + (p * h + Y / d) * w;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = xr[(X4 * 4 + e) / d];
    *reinterpret_cast<f32x4*>(y + i4 * 4) = v;
}

__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                   int planes, int h, int w, int d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)planes * h * w;
    if (i >= n) return;
    const int xx = (int)(i % w);
    const size_t r = i / w;
    const int yy = (int)(r % h);
    const size_t p = r / h;
    const int W = w * d;
    const float* g = dy + ((p * h + yy) * d) * (size_t)W + (size_t)xx * d;
    float acc = 0.f;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) acc += g[(size_t)a * W + b];
    dx[i] = acc;
}

}  // namespace cocos

extern "C" int cocos_upsample_nearest_fwd(const float* x, float* y, int planes, int h, int w, int scale,
                                          cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && y, COCOS_ERR_INVALID, "upsample_nearest_fwd: null pointer");
    COCOS_REQUIRE(planes >= 1 && h >= 1 && w >= 1 && scale >= 1, COCOS_ERR_INVALID,
                  "upsample_nearest_fwd: bad dims planes=%d h=%d w=%d scale=%d", planes, h, w, scale);
    COCOS_REQUIRE((w * scale) % 4 == 0 && aligned16(y), COCOS_ERR_UNSUPPORTED,
                  "upsample_nearest_fwd: output width %d must be a multiple of 4 and y 16-byte aligned", w * scale);
    const size_t n4 = (size_t)planes * h * scale * (w * scale / 4);
    COCOS_REQUIRE((n4 + 255) / 256 <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "upsample_nearest_fwd: tensor too large");
    hipLaunchKernelGGL(upsample_nearest_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       x, y, planes, h, w, scale);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_upsample_nearest_bwd(const float* dy, float* dx, int planes, int h, int w, int scale,
                                          cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(dy && dx, COCOS_ERR_INVALID, "upsample_nearest_bwd: null pointer");
    COCOS_REQUIRE(planes >= 1 && h >= 1 && w >= 1 && scale >= 1, COCOS_ERR_INVALID,
                  "upsample_nearest_bwd: bad dims planes=%d h=%d w=%d scale=%d", planes, h, w, scale);
    const size_t n = (size_t)planes * h * w;
    COCOS_REQUIRE((n + 255) / 256 <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "upsample_nearest_bwd: tensor too large");
    hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       dy, dx, planes, h, w, scale);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
