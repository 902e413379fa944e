// K10: split an fp32 tensor into two f16 planes, x*scale ~= hi + lo, for the split-precision (f16x3)
// correlation kernels (gfx950).  hi = rn_f16(x*scale), lo = rn_f16(x*scale - hi): together 22 mantissa
// bits, so   a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi   (three v_mfma_f32_32x32x16_f16, fp32 accumulate)
// reproduces the fp32 product to ~2^-22 relative — fp32-class accuracy at 16/3 the fp32-MFMA rate.
// `scale` is a power of two chosen by the caller so that the lo plane stays in f16's normal range
// (unit-norm columns: scale = 16) — exact, undone in the consumer's logit scale.
//
//   transpose = 0:  x [B,C,N] -> hi, lo [B,C,N]   (channel-major: the V operand, key-contiguous)
//   transpose = 1:  x [B,C,N] -> hi, lo [B,N,C]   (position-major: Q/K operands want 8 consecutive
//                                                  channels of one position per lane = one 16-byte load)
// HBM-bound: 4 B read + 4 B written per element.
#include "common.h"

namespace cocos {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

__device__ __forceinline__ float scale_from_amax(float amax);

__global__ __launch_bounds__(256) void split_f16_flat_kernel(const float* __restrict__ x,
                                                             _Float16* __restrict__ hi,
                                                             _Float16* __restrict__ lo, size_t n4, size_t n,
                                                             float scale, const float* __restrict__ amax_dev,
                                                             float* __restrict__ scale_out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (amax_dev) {
        scale = scale_from_amax(*amax_dev);
        if (scale_out && i == 0) *scale_out = scale;
    }
    if (i < n4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4) * scale;
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { _Float16 a, b; split1(v[e], a, b); h[e] = a; l[e] = b; }
        *reinterpret_cast<f16x4*>(hi + i * 4) = h;
        *reinterpret_cast<f16x4*>(lo + i * 4) = l;
    } else {                // tail (n % 4 elements), or every element when the pointers are not 16-byte aligned
        const size_t j = n4 * 4 + (i - n4);
        if (j < n) split1(x[j] * scale, hi[j], lo[j]);
    }
}

// Scale from a device-side |x| maximum: the power of two that brings it into [2^9, 2^10) (1 if amax is 0
// or not finite) — gradients have no a-priori magnitude, and the f16 planes need one.
__device__ __forceinline__ float scale_from_amax(float amax) {
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
    int e;
    frexpf(amax, &e);                 // amax = m * 2^e, m in [0.5, 1)
    return ldexpf(1.0f, 10 - e);      // amax * scale in [2^9, 2^10)
}

// 64 channels x 64 positions per workgroup through an LDS transpose; rows of the output are Cpad halfs
// (channels C..Cpad-1 zero) so that a consumer can read whole 16-channel MFMA k-steps
__device__ __forceinline__ void split_f16_transpose_tile(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                         _Float16* __restrict__ lo, int C, int N, int Cpad, float scale,
                                                         const float* __restrict__ amax_dev, float* __restrict__ scale_out, int b) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* xb = x + (size_t)b * C * N;
    if (amax_dev) {
        scale = scale_from_amax(*amax_dev);
        if (scale_out && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && b == 0) *scale_out = scale;
    }
    {
        const int q = tid & 15, r = tid >> 4;   // 16 position quads x 16 rows, 4 sweeps
        const bool vec = (N & 3) == 0 && n0 + 64 <= N && (reinterpret_cast<uintptr_t>(xb) & 15) == 0;    // (workgroup-uniform)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = c0 + r + 16 * u, n = n0 + 4 * q;
            if (vec) {          // whole, aligned quads: one 16-byte load per thread and sweep (round 4: was four scalar loads)
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ch < C) v = *reinterpret_cast<const f32x4*>(xb + (size_t)ch * N + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[r + 16 * u][4 * q + e] = v[e] * scale;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    tile[r + 16 * u][4 * q + e] = (ch < C && n + e < N) ? xb[(size_t)ch * N + n + e] * scale : 0.f;
            }
        }
    }
    __syncthreads();
    const int p = tid >> 2, cq = tid & 3;       // position, 16-channel chunk
    const int n = n0 + p;
    if (n >= N) return;
    _Float16* hrow = hi + ((size_t)b * N + n) * Cpad;
    _Float16* lrow = lo + ((size_t)b * N + n) * Cpad;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int cbase = cq * 16 + g * 8;
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 a, b; split1(tile[cbase + e][p], a, b); h[e] = a; l[e] = b; }
        // channels >= C were loaded as zeros: they ARE the padding
        if (c0 + cbase + 8 <= Cpad && (Cpad % 8) == 0) {
            *reinterpret_cast<f16x8*>(hrow + c0 + cbase) = h;
            *reinterpret_cast<f16x8*>(lrow + c0 + cbase) = l;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + cbase + e < Cpad) { hrow[c0 + cbase + e] = h[e]; lrow[c0 + cbase + e] = l[e]; }
        }
    }
}

__global__ __launch_bounds__(256) void split_f16_transpose_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                                  _Float16* __restrict__ lo, int C, int N, int Cpad, float scale,
                                                                  const float* __restrict__ amax_dev,
                                                                  float* __restrict__ scale_out) {
    split_f16_transpose_tile(x, hi, lo, C, N, Cpad, scale, amax_dev, scale_out, blockIdx.z);
}

// Two tensors of one shape in one launch (grid z = 2 B): the K2 / K19 backward splits d out and v back to back.
struct SplitPair {
    const float* x[2];
    _Float16 *hi[2], *lo[2];
    const float* amax[2];
    float* scale_out[2];
};
__global__ __launch_bounds__(256) void split_f16_transpose_pair_kernel(const SplitPair sp, int B, int C, int N, int Cpad) {
    const bool second = (int)blockIdx.z >= B;          // (workgroup-uniform)
    split_f16_transpose_tile(second ? sp.x[1] : sp.x[0], second ? sp.hi[1] : sp.hi[0], second ? sp.lo[1] : sp.lo[0], C, N, Cpad, 1.0f,
                             second ? sp.amax[1] : sp.amax[0], second ? sp.scale_out[1] : sp.scale_out[0],
                             (int)blockIdx.z - (second ? B : 0));
}

}  // namespace cocos

namespace cocos {

// rows x cols (row-major) -> planes rows x cols_pad, zero beyond cols: the register-resident operand of the K0
// streaming kernel (a weight matrix, k-contiguous, rows padded to whole 16-wide MFMA k-steps)
__global__ __launch_bounds__(256) void split_f16_rows_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                             _Float16* __restrict__ lo, int rows, int cols,
                                                             int cols_pad, float scale,
                                                             const float* __restrict__ amax_dev,
                                                             float* __restrict__ scale_out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (amax_dev) {
        scale = scale_from_amax(*amax_dev);
        if (scale_out && i == 0) *scale_out = scale;
    }
    if (i >= (size_t)rows * cols_pad) return;
    const int r = (int)(i / cols_pad), c = (int)(i - (size_t)r * cols_pad);
    _Float16 a = (_Float16)0.f, b = (_Float16)0.f;
    if (c < cols) split1(x[(size_t)r * cols + c] * scale, a, b);
    hi[i] = a;
    lo[i] = b;
}

// bit (c >> 5) of *mask |= (some element of channel c of a channel-major f16 plane [B,C,N] is non-zero): one workgroup per
// (b, c) row.  Used on the LO plane of the values: label / mask channels are exact in f16, their lo plane is all zero and
// the K2 split kernels then skip it for whole 32-channel blocks (corr_fused_fwd_f16x3.hip, corr_fused_bwd_f16x3.hip).
__global__ __launch_bounds__(256) void f16_plane_block_mask_kernel(const _Float16* __restrict__ plane, int C, int N,
                                                                   unsigned* __restrict__ mask) {
    const int row = blockIdx.x, c = row % C;
    const unsigned short* p = reinterpret_cast<const unsigned short*>(plane) + (size_t)row * N;
    unsigned any = 0;
    const bool vec = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0);
    if (vec) {
        for (int i = threadIdx.x; i < N / 8; i += 256) {
            const u32x4 v = reinterpret_cast<const u32x4*>(p)[i];
            any |= (v.x | v.y | v.z | v.w) & 0x7fff7fffu;           // -0 counts as zero
        }
    } else {
        for (int i = threadIdx.x; i < N; i += 256) any |= p[i] & 0x7fffu;
    }
    if (__builtin_amdgcn_ballot_w64(any != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(mask, 1u << (c >> 5));
}

}  // namespace cocos

static int split_f16_launch(const float* x, void* hi, void* lo, int B, int C, int N, int Cpad, int transpose,
                            float scale, const float* amax_dev, float* scale_out, hipStream_t s) {
    using namespace cocos;
    _Float16* h = static_cast<_Float16*>(hi);
    _Float16* l = static_cast<_Float16*>(lo);
    if (!transpose) {
        const size_t n = (size_t)B * C * N;
        const bool vec = aligned16(x) && (reinterpret_cast<uintptr_t>(hi) & 7u) == 0 &&
                         (reinterpret_cast<uintptr_t>(lo) & 7u) == 0;
        const size_t n4 = vec ? n / 4 : 0;
        const size_t blocks = (n4 + (n - 4 * n4) + 255) / 256;
        COCOS_REQUIRE(blocks <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "split_f16: tensor too large");
        hipLaunchKernelGGL(split_f16_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, h, l, n4, n, scale,
                           amax_dev, scale_out);
    } else {
        COCOS_REQUIRE((Cpad + 63) / 64 <= 65535, COCOS_ERR_UNSUPPORTED, "split_f16: C too large");
        COCOS_REQUIRE(aligned16(hi) && aligned16(lo), COCOS_ERR_INVALID, "split_f16: planes must be 16-byte aligned");
        hipLaunchKernelGGL(split_f16_transpose_kernel, dim3((N + 63) / 64, (Cpad + 63) / 64, B), dim3(256), 0, s, x,
                           h, l, C, N, Cpad, scale, amax_dev, scale_out);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_split_f16(const float* x, void* hi, void* lo, int B, int C, int N, int transpose,
                               float scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && hi && lo, COCOS_ERR_INVALID, "split_f16: null pointer");
    COCOS_REQUIRE(B >= 1 && C >= 1 && N >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "split_f16: bad dims B=%d C=%d N=%d", B, C, N);
    return split_f16_launch(x, hi, lo, B, C, N, C, transpose, scale, nullptr, nullptr, as_stream(stream));
}

extern "C" int cocos_split_f16_ex(const float* x, void* hi, void* lo, int B, int C, int N, int Cpad,
                                  int transpose, float scale, const float* amax_dev, float* scale_out_dev,
                                  cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && hi && lo, COCOS_ERR_INVALID, "split_f16_ex: null pointer");
    COCOS_REQUIRE(B >= 1 && C >= 1 && N >= 1 && B <= 65535 && Cpad >= C, COCOS_ERR_INVALID,
                  "split_f16_ex: bad dims B=%d C=%d N=%d Cpad=%d", B, C, N, Cpad);
    COCOS_REQUIRE(transpose || Cpad == C, COCOS_ERR_INVALID, "split_f16_ex: padding only with transpose");
    return split_f16_launch(x, hi, lo, B, C, N, Cpad, transpose, scale, amax_dev, scale_out_dev, as_stream(stream));
}

// cocos_split_f16_ex(transpose = 1, device-side scales) for TWO tensors of one shape [B,C,N] in one launch.
extern "C" int cocos_split_f16_transpose_pair(const float* x0, void* hi0, void* lo0, const float* amax0_dev, float* scale0_out_dev,
                                              const float* x1, void* hi1, void* lo1, const float* amax1_dev, float* scale1_out_dev,
                                              int B, int C, int N, int Cpad, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x0 && hi0 && lo0 && amax0_dev && scale0_out_dev && x1 && hi1 && lo1 && amax1_dev && scale1_out_dev, COCOS_ERR_INVALID,
                  "split_f16_transpose_pair: null pointer");
    COCOS_REQUIRE(B >= 1 && C >= 1 && N >= 1 && 2 * B <= 65535 && Cpad >= C && (Cpad + 63) / 64 <= 65535, COCOS_ERR_INVALID,
                  "split_f16_transpose_pair: bad dims B=%d C=%d N=%d Cpad=%d", B, C, N, Cpad);
    for (const void* p : {(const void*)hi0, (const void*)lo0, (const void*)hi1, (const void*)lo1})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "split_f16_transpose_pair: planes must be 16-byte aligned");
    SplitPair sp;
    sp.x[0] = x0; sp.x[1] = x1;
    sp.hi[0] = static_cast<_Float16*>(hi0); sp.hi[1] = static_cast<_Float16*>(hi1);
    sp.lo[0] = static_cast<_Float16*>(lo0); sp.lo[1] = static_cast<_Float16*>(lo1);
    sp.amax[0] = amax0_dev; sp.amax[1] = amax1_dev;
    sp.scale_out[0] = scale0_out_dev; sp.scale_out[1] = scale1_out_dev;
    hipLaunchKernelGGL(split_f16_transpose_pair_kernel, dim3((N + 63) / 64, (Cpad + 63) / 64, 2 * B), dim3(256), 0, as_stream(stream), sp,
                       B, C, N, Cpad);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// 2-D form with padded rows: x [rows][cols] -> hi, lo [rows][cols_pad] (zero beyond cols), x*scale ~= hi + lo with
// the scale chosen as in cocos_split_f16_ex (amax_dev != NULL: power of two from *amax_dev, written to scale_out_dev).
extern "C" int cocos_split_f16_rows(const float* x, void* hi, void* lo, int rows, int cols, int cols_pad, float scale,
                                    const float* amax_dev, float* scale_out_dev, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && hi && lo, COCOS_ERR_INVALID, "split_f16_rows: null pointer");
    COCOS_REQUIRE(rows >= 1 && cols >= 1 && cols_pad >= cols, COCOS_ERR_INVALID,
                  "split_f16_rows: bad dims rows=%d cols=%d cols_pad=%d", rows, cols, cols_pad);
    const size_t n = (size_t)rows * cols_pad;
    COCOS_REQUIRE((n + 255) / 256 <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "split_f16_rows: tensor too large");
    hipLaunchKernelGGL(split_f16_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                       static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), rows, cols, cols_pad, scale, amax_dev,
                       scale_out_dev);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_f16_plane_block_mask(const void* plane, int B, int C, int N, unsigned* mask_inout_dev,
                                          cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(plane && mask_inout_dev, COCOS_ERR_INVALID, "f16_plane_block_mask: null pointer");
    COCOS_REQUIRE(B >= 1 && C >= 1 && C <= 1024 && N >= 1 && (long long)B * C <= 0x7fffffffll, COCOS_ERR_INVALID,
                  "f16_plane_block_mask: bad dims B=%d C=%d N=%d (C <= 1024)", B, C, N);
    hipLaunchKernelGGL(f16_plane_block_mask_kernel, dim3((unsigned)(B * C)), dim3(256), 0, as_stream(stream),
                       static_cast<const _Float16*>(plane), C, N, mask_inout_dev);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
