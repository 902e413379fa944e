// The head of the first row pass: what sits between its output o = softmax(f) @ [exemplar | ref_seg]  [B, Ci+Cs, h*w] and the two
// tensors the losses see (gfx950, round 6):
//     warp_out  = nearest x`d` up-sampling of o[:, :Ci]      (correspondence.py:188 at :327)
//     warp_mask = o[:, Ci:]                                   (:334, a view)
// Forward: the up-sampling straight from the channel slice (the strided slice used to be copied first).
// Backward: ONE kernel for everything autograd ran between the two loss gradients and the K2 / K19 backward —
//     d o[:, :Ci] = d x d window sums of d warp_out           (was cocos_upsample_nearest_bwd)
//     d o[:, Ci:] = d warp_mask                               (was cocos_concat2_amax: copy + max|.|)
//     max|d o|                                                (the scale source of the backward's f16 split)
//     D[b, n] = sum_c d o[b,c,n] * o[b,c,n]  in fp64          (was cocos_rowdot_f64: the D of the softmax backward)
// four launches and three passes over the 20 MB gradient at the benchmark shape, now one pass.
#include "common.h"

namespace cocos {

__global__ __launch_bounds__(256) void warp_head_fwd_kernel(const float* __restrict__ o, float* __restrict__ y, int Ci, int C,
                                                            int h, int w, int d, size_t n4) {
    const int W = w * d, H = h * d;
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;            // float4 index in y (W % 4 == 0)
    if (i4 >= n4) return;
    const int X4 = (int)(i4 % (W / 4));
    size_t r = i4 / (W / 4);
    const int Y = (int)(r % H);
    r /= H;
    const int c = (int)(r % Ci);
    const size_t b = r / Ci;
    const float* xr = o + ((b * C + c) * h + Y / d) * (size_t)w;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = xr[(X4 * 4 + e) / d];
    *reinterpret_cast<f32x4*>(y + i4 * 4) = v;
}

// Workgroup = 64 positions (16 lanes x 4) x 16 channel groups, grid (N / 64, B): 512 workgroups at the benchmark shape (two per CU;
// with rowdot_f64's 128 x 8 decomposition the kernel was latency-bound at one workgroup per CU: 23 us for 66 MB).
template <int D>
__global__ __launch_bounds__(256) void warp_head_bwd_kernel(const float* __restrict__ g_img, const float* __restrict__ g_mask,
                                                            const float* __restrict__ o, float* __restrict__ dout,
                                                            float* __restrict__ drow, unsigned* __restrict__ amax, int Ci, int Cs,
                                                            int h, int w, int dd) {
    __shared__ double red[16][64];
    __shared__ float redm[4];
    const int d = D ? D : dd;
    const int N = h * w, C = Ci + Cs;
    const int q4 = threadIdx.x & 15, cg = threadIdx.x >> 4;
    const int i0 = blockIdx.x * 64 + q4 * 4, b = blockIdx.y;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    float vmax = 0.f;
    if (i0 < N) {                                          // N % 4 == 0, w % 4 == 0: a quad lies in one image row
        const int yy = i0 / w, xx = i0 - yy * w;
        const int W = w * d;
        for (int c = cg; c < C; c += 16) {
            f32x4 g;
            if (c < Ci) {
                const float* p = g_img + (((size_t)b * Ci + c) * h + yy) * d * (size_t)W + (size_t)xx * d;
                g = f32x4{0.f, 0.f, 0.f, 0.f};
                if (D == 4) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(p + (size_t)a * W + 4 * e);
                            g[e] += (t[0] + t[1]) + (t[2] + t[3]);
                        }
                } else {
                    for (int a = 0; a < d; ++a)
                        for (int e = 0; e < 4; ++e)
                            for (int j = 0; j < d; ++j) g[e] += p[(size_t)a * W + e * d + j];
                }
            } else {
                g = *reinterpret_cast<const f32x4*>(g_mask + ((size_t)b * Cs + (c - Ci)) * N + i0);
            }
            const size_t off = ((size_t)b * C + c) * N + i0;
            *reinterpret_cast<f32x4*>(dout + off) = g;
            const f32x4 y = *reinterpret_cast<const f32x4*>(o + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] += (double)g[e] * (double)y[e];
                vmax = fmaxf(vmax, fabsf(g[e]));
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[cg][q4 * 4 + e] = acc[e];
    vmax = wave_max_dpp(vmax);
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = blockIdx.x * 64 + threadIdx.x;
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
        if (i < N) drow[(size_t)b * N + i] = (float)t;
    }
    if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]))));
}

}  // namespace cocos

// y [B, Ci, h*down, w*down] = nearest up-sampling of channels [0, Ci) of o [B, C, h, w] (Ci <= C; (w * down) % 4 == 0).
extern "C" int cocos_warp_head_fwd(const float* o, float* y, int B, int Ci, int C, int h, int w, int down, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(o && y, COCOS_ERR_INVALID, "warp_head_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && Ci >= 1 && Ci <= C && h >= 1 && w >= 1 && down >= 1, COCOS_ERR_INVALID,
                  "warp_head_fwd: bad dims B=%d Ci=%d C=%d h=%d w=%d down=%d", B, Ci, C, h, w, down);
    COCOS_REQUIRE((w * down) % 4 == 0 && aligned16(y), COCOS_ERR_UNSUPPORTED,
                  "warp_head_fwd: output width %d must be a multiple of 4 and y 16-byte aligned", w * down);
    const size_t n4 = (size_t)B * Ci * h * down * (w * down / 4);
    COCOS_REQUIRE((n4 + 255) / 256 <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "warp_head_fwd: tensor too large");
    hipLaunchKernelGGL(warp_head_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), o, y, Ci, C, h, w,
                       down, n4);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// g_img [B, Ci, h*down, w*down] (d loss / d warp_out), g_mask [B, Cs, h, w] (d loss / d warp_mask), o [B, Ci+Cs, h, w] ->
// dout [B, Ci+Cs, h, w], drow [B, h*w] = sum_c dout * o (fp64 accumulation), *amax_inout = max(*amax_inout, max|dout|)
// (a cell holding a finite value >= 0).  w % 4 == 0; all tensors 16-byte aligned.
extern "C" int cocos_warp_head_bwd(const float* g_img, const float* g_mask, const float* o, float* dout, float* drow,
                                   float* amax_inout_dev, int B, int Ci, int Cs, int h, int w, int down, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(o && dout && drow && amax_inout_dev && (g_img || Ci == 0) && (g_mask || Cs == 0), COCOS_ERR_INVALID,
                  "warp_head_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && Ci >= 0 && Cs >= 0 && Ci + Cs >= 1 && h >= 1 && w >= 1 && down >= 1, COCOS_ERR_INVALID,
                  "warp_head_bwd: bad dims B=%d Ci=%d Cs=%d h=%d w=%d down=%d", B, Ci, Cs, h, w, down);
    COCOS_REQUIRE(w % 4 == 0, COCOS_ERR_UNSUPPORTED, "warp_head_bwd: grid width %d must be a multiple of 4", w);
    for (const void* p : {(const void*)g_img, (const void*)g_mask, (const void*)o, (const void*)dout})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "warp_head_bwd: tensors must be 16-byte aligned");
    const int N = h * w;
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)B);
    unsigned* cell = reinterpret_cast<unsigned*>(amax_inout_dev);
    if (down == 4)
        hipLaunchKernelGGL(warp_head_bwd_kernel<4>, grid, dim3(256), 0, as_stream(stream), g_img, g_mask, o, dout, drow, cell, Ci, Cs,
                           h, w, down);
    else
        hipLaunchKernelGGL(warp_head_bwd_kernel<0>, grid, dim3(256), 0, as_stream(stream), g_img, g_mask, o, dout, drow, cell, Ci, Cs,
                           h, w, down);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
