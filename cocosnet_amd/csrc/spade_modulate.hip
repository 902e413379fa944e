// K17: SPADE modulation + LeakyReLU behind ANY parameter-free norm (gfx950).          [SURVEY §8(f) rank 1]
//
// The non-PONO branch of SPADE (normalization.py:93-101: InstanceNorm2d | SynchronizedBatchNorm2d | BatchNorm2d, all
// affine=False) followed by `normalized * (1 + gamma) + beta` (:148) and the LeakyReLU(0.2) of
// SPADEResnetBlock.actvn (architecture.py:88-95).  The statistics of those norms couple whole planes / the whole batch
// (and, for sync-batch, all ranks): they stay where they are (torch's instance norm, cocosnet_amd.dist.SyncBatchNorm2d);
// what this kernel takes is everything after them — four framework kernels forward (1+gamma, mul, add, leaky_relu:
// 40 B/element) and their autograd (three saved intermediates) become one pass each way:
//   fwd: z = xh*(1+gamma) + beta;  y = z > 0 ? z : slope*z                                    16 B/element
//   bwd: dz = dy*(z > 0 ? 1 : slope);  dxh = dz*(1+gamma);  dgamma = dz*xh;  dbeta = dz        28 B/element (z recomputed)
// Pure streaming: HBM-bound.  (With --PONO the whole thing incl. the norm is K9, pono_spade.hip.)
#include "common.h"

namespace cocos {

template <bool BWD>
__global__ __launch_bounds__(256) void spade_modulate_kernel(const float* __restrict__ xh, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ dy,
                                                             float* __restrict__ o0 /* y | dxh */, float* __restrict__ dg,
                                                             float* __restrict__ db, size_t n, float slope, bool vec) {
    const size_t stride = (size_t)gridDim.x * 256;
    if (vec) {
        const size_t n4 = n / 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const f32x4 x = reinterpret_cast<const f32x4*>(xh)[i], g = reinterpret_cast<const f32x4*>(gamma)[i],
                        b = reinterpret_cast<const f32x4*>(beta)[i];
            f32x4 z = x * (g + 1.0f) + b;
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = z[e] > 0.f ? z[e] : slope * z[e];
                reinterpret_cast<f32x4*>(o0)[i] = z;
            } else {
                f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = z[e] > 0.f ? d[e] : slope * d[e];
                if (o0) reinterpret_cast<f32x4*>(o0)[i] = d * (g + 1.0f);
                if (dg) reinterpret_cast<f32x4*>(dg)[i] = d * x;
                if (db) reinterpret_cast<f32x4*>(db)[i] = d;
            }
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float x = xh[i], g = gamma[i], z = x * (g + 1.0f) + beta[i];
            if (!BWD) {
                o0[i] = z > 0.f ? z : slope * z;
            } else {
                const float d = z > 0.f ? dy[i] : slope * dy[i];
                if (o0) o0[i] = d * (g + 1.0f);
                if (dg) dg[i] = d * x;
                if (db) db[i] = d;
            }
        }
    }
}

static unsigned sm_blocks(size_t n) {
    const size_t want = (n / 4 + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
}

}  // namespace cocos

// y = leaky_relu(xh * (1 + gamma) + beta, slope) for n elements (xh = the already normalised activations)
extern "C" int cocos_spade_modulate_fwd(const float* xh, const float* gamma, const float* beta, float* y, long long n,
                                        float slope, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(xh && gamma && beta && y, COCOS_ERR_INVALID, "spade_modulate_fwd: null pointer");
    COCOS_REQUIRE(n >= 1, COCOS_ERR_INVALID, "spade_modulate_fwd: n=%lld", n);
    const bool vec = n % 4 == 0 && aligned16(xh) && aligned16(gamma) && aligned16(beta) && aligned16(y);
    hipLaunchKernelGGL((spade_modulate_kernel<false>), dim3(sm_blocks((size_t)n)), dim3(256), 0, as_stream(stream), xh, gamma,
                       beta, nullptr, y, nullptr, nullptr, (size_t)n, slope, vec);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// gradients of the above w.r.t. xh, gamma, beta (any of the three outputs may be NULL)
extern "C" int cocos_spade_modulate_bwd(const float* xh, const float* gamma, const float* beta, const float* dy, float* dxh,
                                        float* dgamma, float* dbeta, long long n, float slope, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(xh && gamma && beta && dy, COCOS_ERR_INVALID, "spade_modulate_bwd: null pointer");
    COCOS_REQUIRE(n >= 1, COCOS_ERR_INVALID, "spade_modulate_bwd: n=%lld", n);
    bool vec = n % 4 == 0 && aligned16(xh) && aligned16(gamma) && aligned16(beta) && aligned16(dy);
    for (const float* p : {dxh, dgamma, dbeta}) vec = vec && (!p || aligned16(p));
    hipLaunchKernelGGL((spade_modulate_kernel<true>), dim3(sm_blocks((size_t)n)), dim3(256), 0, as_stream(stream), xh, gamma,
                       beta, dy, dxh, dgamma, dbeta, (size_t)n, slope, vec);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
