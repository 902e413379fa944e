// K14: the value tensor of the first row pass, built in one kernel (gfx950).
//
// Replaces (correspondence.py:314 / :318-319 and the mask branch :331-334, as restated in hot_path.py)
//     ref     = F.avg_pool2d(ref_img, down)                               [B, Ci, h, w]
//     ref_seg = F.interpolate(ref_seg_map, scale_factor=1/down, 'nearest') [B, Cs, h, w]   (src = dst * down)
//     V       = torch.cat((ref, ref_seg), dim=1)                           [B, Ci+Cs, h, w]
// three framework kernels (8 + 27 + 12 us at the benchmark shape) and a 20 MB intermediate that is written and read
// again by the cat.  HBM-bound: of the one-hot label map only every down-th row is touched, and of those rows every
// byte crosses HBM once (64-byte bursts hold 4 of the wanted floats): 79 MB + 6 MB in, 20 MB out.
// One thread per OUTPUT element: a wave reads 64 x 16-byte-strided floats = 1 KB of a label row / writes 256
// contiguous bytes; the image channels (3 of 154) take their d x d window as d float4 loads when d == 4.
#include <algorithm>

#include "common.h"

namespace cocos {

__global__ __launch_bounds__(256) void warp_values_kernel(const float* __restrict__ img, const float* __restrict__ seg,
                                                          float* __restrict__ out, int Ci, int Cs, int h, int w, int d,
                                                          size_t n, bool vec4, unsigned* __restrict__ amax) {
  float vmax = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  const int W = w * d, H = h * d;
  // four output elements per thread and iteration, their loads issued before any is used (round 6: with one strided 4-byte load in
  // flight per thread the kernel was latency-bound — 2.1 TB/s of 64-byte bursts over every d-th row of the label map)
  constexpr int U = 4;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * U) {
    float v[U];
    const float* pimg[U];
    bool is_img[U], live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * stride;
      live[u] = i < n;
      const size_t ii = live[u] ? i : 0;
      const int x = (int)(ii % w);
      size_t r = ii / w;
      const int y = (int)(r % h);
      r /= h;
      const int c = (int)(r % (Ci + Cs));
      const size_t b = r / (Ci + Cs);
      is_img[u] = c < Ci;
      pimg[u] = nullptr;
      v[u] = 0.f;
      if (!is_img[u]) {                                     // nearest: source pixel (y*d, x*d)
        v[u] = seg[((b * Cs + (c - Ci)) * H + (size_t)y * d) * W + (size_t)x * d];
      } else {
        pimg[u] = img + ((b * Ci + c) * H + (size_t)y * d) * W + (size_t)x * d;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (is_img[u]) {                                      // mean of the d x d window
        const float* p = pimg[u];
        float acc = 0.f;
        if (vec4) {                                         // d == 4, rows 16-byte aligned
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + (size_t)a * W);
            acc += (q[0] + q[1]) + (q[2] + q[3]);
          }
        } else {
          for (int a = 0; a < d; ++a)
            for (int e = 0; e < d; ++e) acc += p[(size_t)a * W + e];
        }
        v[u] = acc / (float)(d * d);
      }
      if (live[u]) {
        out[i0 + (size_t)u * stride] = v[u];
        vmax = fmaxf(vmax, fabsf(v[u]));
      }
    }
  }
  // max|V| as a by-product (the K2 forward's f16 split of V wants it): one same-address atomic per workgroup, the
  // grid of the _amax entry point is capped accordingly
  if (amax) {
      __shared__ float red[4];
      vmax = wave_max_dpp(vmax);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
      __syncthreads();
      if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
  }
}

}  // namespace cocos

// out [B, Ci+Cs, H/down, W/down]: channels [0,Ci) = down x down mean of img [B,Ci,H,W]; channels [Ci,Ci+Cs) =
// seg [B,Cs,H,W] sampled at (y*down, x*down).  Either part may be absent (Ci == 0 / Cs == 0 with a NULL pointer).
static int warp_values_impl(const float* img, const float* seg, float* out, int B, int Ci, int Cs, int H, int W, int down,
                            float* amax_inout_dev, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(out && (img || Ci == 0) && (seg || Cs == 0), COCOS_ERR_INVALID, "warp_values: null pointer");
    COCOS_REQUIRE(B >= 1 && Ci >= 0 && Cs >= 0 && Ci + Cs >= 1 && H >= 1 && W >= 1 && down >= 1, COCOS_ERR_INVALID,
                  "warp_values: bad dims B=%d Ci=%d Cs=%d H=%d W=%d down=%d", B, Ci, Cs, H, W, down);
    COCOS_REQUIRE(H % down == 0 && W % down == 0, COCOS_ERR_UNSUPPORTED,
                  "warp_values: %dx%d is not a multiple of down=%d", H, W, down);
    const size_t n = (size_t)B * (Ci + Cs) * (H / down) * (W / down);
    COCOS_REQUIRE((n + 255) / 256 <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "warp_values: tensor too large");
    const bool vec4 = down == 4 && W % 4 == 0 && img && aligned16(img);
    // (4 elements per thread and trip; more than ~1000 workgroups of strided streams at once was SLOWER: 61 us at 4096 against 35)
    const size_t blocks = std::min<size_t>(1024, (n + 1023) / 1024);
    hipLaunchKernelGGL(warp_values_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), img, seg,
                       out, Ci, Cs, H / down, W / down, down, n, vec4, reinterpret_cast<unsigned*>(amax_inout_dev));
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_warp_values(const float* img, const float* seg, float* out, int B, int Ci, int Cs, int H, int W,
                                 int down, cocos_stream_t stream) {
    return warp_values_impl(img, seg, out, B, Ci, Cs, H, W, down, nullptr, stream);
}

// Same, and *amax_inout_dev = max(*amax_inout_dev, max|out|) (a cell holding a finite value >= 0, e.g. zero).
extern "C" int cocos_warp_values_amax(const float* img, const float* seg, float* out, int B, int Ci, int Cs, int H, int W,
                                      int down, float* amax_inout_dev, cocos_stream_t stream) {
    COCOS_REQUIRE(amax_inout_dev, COCOS_ERR_INVALID, "warp_values_amax: null amax cell");
    return warp_values_impl(img, seg, out, B, Ci, Cs, H, W, down, amax_inout_dev, stream);
}
