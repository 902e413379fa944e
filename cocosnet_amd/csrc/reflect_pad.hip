// K18: nn.ReflectionPad2d(p) and its backward (gfx950) — in front of every 3x3 convolution of the ResidualBlocks
// (correspondence.py:19,:23) and of SPADE / SPADEResnetBlock (normalization.py:118,:146; architecture.py:30).
//   fwd: y[b,c,Y,X] = x[b,c,r(Y - p, H), r(X - p, W)],   r(i, n) = i < 0 ? -i : (i >= n ? 2(n-1) - i : i)      (p < H, W)
//   bwd: dx[b,c,y,x] = sum of dy over the (at most 3 x 3) padded pixels that read (y, x): a GATHER, no atomics
//        (the framework's backward scatters with atomics: 67 us per call at [8,407,64,64] against ~12 us of HBM time)
// Streaming, HBM-bound: 8 B/element each way.
#include "common.h"

namespace cocos {

__device__ __forceinline__ int rp_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// grid (planes, row chunks of 32), block (64 columns, 4 rows): no divisions in the index arithmetic
__global__ __launch_bounds__(256) void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                              int p) {
    const int Ho = H + 2 * p, Wo = W + 2 * p;
    const float* xp = x + (size_t)blockIdx.x * H * W;
    float* yp = y + (size_t)blockIdx.x * Ho * Wo;
    const int y1 = min(Ho, (int)(blockIdx.y + 1) * 32);
    for (int Y = blockIdx.y * 32 + threadIdx.y; Y < y1; Y += 4) {
        const float* row = xp + (size_t)rp_reflect(Y - p, H) * W;
        for (int X = threadIdx.x; X < Wo; X += 64) yp[(size_t)Y * Wo + X] = row[rp_reflect(X - p, W)];
    }
}

// the padded coordinates that read input coordinate i of an axis of length n: i + p always; p - i for 1 <= i <= p;
// 2(n-1) - i + p for n-1-p <= i <= n-2
__device__ __forceinline__ int rp_sources(int i, int n, int p, int (&src)[3]) {
    int k = 0;
    src[k++] = i + p;
    if (i >= 1 && i <= p) src[k++] = p - i;
    if (i >= n - 1 - p && i <= n - 2) src[k++] = 2 * (n - 1) - i + p;
    return k;
}

__global__ __launch_bounds__(256) void reflect_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W,
                                                              int p) {
    const int Ho = H + 2 * p, Wo = W + 2 * p;
    const float* d = dy + (size_t)blockIdx.x * Ho * Wo;
    float* o = dx + (size_t)blockIdx.x * H * W;
    const int y1 = min(H, (int)(blockIdx.y + 1) * 32);
    for (int yy = blockIdx.y * 32 + threadIdx.y; yy < y1; yy += 4) {
        int sy[3];
        const int ny = rp_sources(yy, H, p, sy);
        for (int xx = threadIdx.x; xx < W; xx += 64) {
            int sx[3];
            const int nx = rp_sources(xx, W, p, sx);
            float acc = 0.f;
            for (int a = 0; a < ny; ++a)
                for (int b = 0; b < nx; ++b) acc += d[(size_t)sy[a] * Wo + sx[b]];
            o[(size_t)yy * W + xx] = acc;
        }
    }
}

}  // namespace cocos

extern "C" int cocos_reflect_pad2d_fwd(const float* x, float* y, long long planes, int H, int W, int pad, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && y, COCOS_ERR_INVALID, "reflect_pad2d_fwd: null pointer");
    COCOS_REQUIRE(planes >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W, COCOS_ERR_INVALID,
                  "reflect_pad2d_fwd: bad dims planes=%lld H=%d W=%d pad=%d (pad must be smaller than H and W)", planes, H, W, pad);
    COCOS_REQUIRE(planes <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "reflect_pad2d_fwd: too many planes");
    hipLaunchKernelGGL(reflect_pad_fwd_kernel, dim3((unsigned)planes, (unsigned)((H + 2 * pad + 31) / 32)), dim3(64, 4), 0,
                       as_stream(stream), x, y, H, W, pad);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_reflect_pad2d_bwd(const float* dy, float* dx, long long planes, int H, int W, int pad, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(dy && dx, COCOS_ERR_INVALID, "reflect_pad2d_bwd: null pointer");
    COCOS_REQUIRE(planes >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad < H && pad < W, COCOS_ERR_INVALID,
                  "reflect_pad2d_bwd: bad dims planes=%lld H=%d W=%d pad=%d", planes, H, W, pad);
    COCOS_REQUIRE(planes <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "reflect_pad2d_bwd: too many planes");
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3((unsigned)planes, (unsigned)((H + 31) / 32)), dim3(64, 4), 0,
                       as_stream(stream), dy, dx, H, W, pad);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
