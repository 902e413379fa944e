// Operand-plane preparation folded into the kernels that already touch the data (round 3, VERDICT r2 "weak 3": 24.5 % of
// the benchmark step went into cocos_split_f16 / cocos_absmax / cocos_f16_plane_block_mask launches).  gfx950.
//
//   cocos_concat2_amax          torch.cat((a, b), dim=1) of two gradients + max|.| of the result in the same pass: the
//                               backward of the row pass's output split (hot_path._SplitChannels) — the concatenated
//                               gradient is the `dout` of the K2 backward, whose f16 split needs that maximum
//   cocos_split_f16_chan_mask   cocos_split_f16_ex(transpose = 0) + cocos_f16_plane_block_mask of its lo plane in one
//                               launch (the V operand of the K2 forward)
//   cocos_proj_weight_planes    BOTH plane sets of a 1x1-projection weight (K0, correspondence.py:181-182 at :272, :282)
//                               in one launch: [Cout][KpadIn] for y = W x and the transposed [Cin][KpadOut] for dx = W^T dy
#include <algorithm>

#include "common.h"

namespace cocos {

typedef _Float16 pp_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pp_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void pp_split1(float x, _Float16& hi, _Float16& lo) {     // as split_f16.hip: hi rounded to nearest
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
__device__ __forceinline__ float pp_scale_from_amax(float amax) {                    // as split_f16.hip
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
    int e;
    frexpf(amax, &e);
    return ldexpf(1.0f, 10 - e);
}
// one same-address atomic per WORKGROUP (they serialise at the memory side, ~10 ns each: grids are capped at 512)
__device__ __forceinline__ void pp_block_amax(float m, unsigned* out) {
    __shared__ float red[4];
    m = wave_max_dpp(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// out[b] = [a[b] (na4 float4) | b[b] (nb4 float4)], float4 granularity; grid-stride over B * (na4 + nb4)
__global__ __launch_bounds__(256) void concat2_amax_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ bsrc,
                                                           f32x4* __restrict__ out, size_t na4, size_t nb4, size_t total4,
                                                           unsigned* __restrict__ amax) {
    const size_t per = na4 + nb4, stride = (size_t)gridDim.x * 256;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const size_t b = i / per, r = i - b * per;
        const f32x4 v = r < na4 ? a[b * na4 + r] : bsrc[b * nb4 + (r - na4)];
        out[i] = v;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (amax) pp_block_amax(m, amax);
}

// x [B,C,N] -> hi, lo [B,C,N] scaled by the power of two from *amax_dev; *mask |= 1 << (c >> 5) for every channel c whose
// lo plane has a non-zero element.  N % 4 == 0.  A set bit is looked up (plain load) before the atomic: for a general V
// every block gets its bit from the first few waves and the other ~20 000 waves only read.
__global__ __launch_bounds__(256) void split_f16_chan_mask_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                                  _Float16* __restrict__ lo, size_t n4, int C, int N4,
                                                                  const float* __restrict__ amax_dev,
                                                                  float* __restrict__ scale_out,
                                                                  unsigned* __restrict__ mask) {
    const float scale = pp_scale_from_amax(*amax_dev);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (scale_out && i == 0) *scale_out = scale;
    if (i >= n4) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4) * scale;
    pp_f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { _Float16 p, q; pp_split1(v[e], p, q); h[e] = p; l[e] = q; }
    *reinterpret_cast<pp_f16x4*>(hi + i * 4) = h;
    *reinterpret_cast<pp_f16x4*>(lo + i * 4) = l;
    if (mask) {
        typedef unsigned int pp_u32x2 __attribute__((ext_vector_type(2)));
        const pp_u32x2 w = __builtin_bit_cast(pp_u32x2, l);
        if (((w.x | w.y) & 0x7fff7fffu) != 0u) {                      // -0 counts as zero
            const unsigned bit = 1u << (((i / (size_t)N4) % (size_t)C) >> 5);
            if ((__builtin_nontemporal_load(mask) & bit) == 0u) atomicOr(mask, bit);
        }
    }
}

// W [Cout][Cin] (row-major) -> rows planes [Cout][KpIn] (zero for columns >= Cin) and transposed planes [Cin][KpOut]
// (zero for columns >= Cout), both scale * W with the power-of-two scale from *amax_dev.  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void proj_weight_planes_kernel(const float* __restrict__ w, _Float16* __restrict__ rh,
                                                                 _Float16* __restrict__ rl, _Float16* __restrict__ th,
                                                                 _Float16* __restrict__ tl, int Cout, int Cin, int KpIn,
                                                                 int KpOut, const float* __restrict__ amax_dev,
                                                                 float* __restrict__ scale_out) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;       // rows = output channels, columns = input channels
    const float scale = pp_scale_from_amax(*amax_dev);
    if (scale_out && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *scale_out = scale;
    {
        const int q = tid & 63, r = tid >> 6;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int rr = r0 + r + 4 * u, cc = c0 + q;
            tile[r + 4 * u][q] = (rr < Cout && cc < Cin) ? w[(size_t)rr * Cin + cc] * scale : 0.f;
        }
    }
    __syncthreads();
    // rows planes: thread (row = tid >> 2, 16-column chunk = tid & 3)
    {
        const int row = r0 + (tid >> 2), cq = tid & 3;
        if (row < Cout) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int cb = cq * 16 + g * 8;
                pp_f16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 p, q; pp_split1(tile[tid >> 2][cb + e], p, q); h[e] = p; l[e] = q; }
                if (c0 + cb + 8 <= KpIn) {                       // KpIn % 8 == 0 (launcher)
                    *reinterpret_cast<pp_f16x8*>(rh + (size_t)row * KpIn + c0 + cb) = h;
                    *reinterpret_cast<pp_f16x8*>(rl + (size_t)row * KpIn + c0 + cb) = l;
                }
            }
        }
    }
    // transposed planes: thread (input channel = tid >> 2, 16-output-channel chunk = tid & 3)
    if (th) {
        const int col = c0 + (tid >> 2), rq = tid & 3;
        if (col < Cin) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int rb = rq * 16 + g * 8;
                pp_f16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 p, q; pp_split1(tile[rb + e][tid >> 2], p, q); h[e] = p; l[e] = q; }
                if (r0 + rb + 8 <= KpOut) {
                    *reinterpret_cast<pp_f16x8*>(th + (size_t)col * KpOut + r0 + rb) = h;
                    *reinterpret_cast<pp_f16x8*>(tl + (size_t)col * KpOut + r0 + rb) = l;
                }
            }
        }
    }
}

// out[i] = sum_s x[s][i]: the sum over the leading dimension of [S][n] partial tiles (weight-gradient partials of the
// general split GEMMs) — one thread per float4 column group, S small
__global__ __launch_bounds__(256) void sum_leading_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ out, int S,
                                                          size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = x[i];
    for (int s = 1; s < S; ++s) acc += x[(size_t)s * n4 + i];
    out[i] = acc;
}
__global__ __launch_bounds__(256) void sum_leading_tail_kernel(const float* __restrict__ x, float* __restrict__ out, int S,
                                                               size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = x[i];
    for (int s = 1; s < S; ++s) acc += x[(size_t)s * n + i];
    out[i] = acc;
}

// db[c] = sum over b, n of dy[b][c][n]  (the bias gradient of a convolution: dy.sum((0, 2, 3))).  Workgroup (c, s) sums slice s
// of channel c's B rows of N values — 16-byte loads when N allows — into out[s * C + c]; with one slice that is db itself,
// otherwise the [S][C] partials are summed by a second launch of the same kernel (B = S, N = 1).  One workgroup per channel
// (the first version) left a 64-channel, 134 MB gradient to 64 workgroups: 2.9 ms per step of the module scope.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dy, float* __restrict__ out, int B, int C,
                                                          int N, int per_slice) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * per_slice, n1 = min(N, n0 + per_slice);
    float acc = 0.f;
    if (((N | per_slice) & 3) == 0) {
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const f32x4* p = reinterpret_cast<const f32x4*>(dy + ((size_t)b * C + c) * N);
            for (int i = (n0 >> 2) + threadIdx.x; i < (n1 >> 2); i += 256) a4 += p[i];
        }
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else {
        for (int b = 0; b < B; ++b) {
            const float* p = dy + ((size_t)b * C + c) * N;
            for (int i = n0 + threadIdx.x; i < n1; i += 256) acc += p[i];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)s * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// slices per channel: enough workgroups for 256 CUs (>= 2048 in all), at least 4096 values per slice and batch row
static int channel_sum_slice_count(int C, long long N) {
    long long s = (2048 + C - 1) / C;
    const long long cap = (N + 4095) / 4096;
    if (s > cap) s = cap;
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}

// the four statistic gradients of K6's backward from its row / column sums (box3_unfold.hip; ops._Box3Logits.backward)
__global__ __launch_bounds__(256) void box3_stat_grads_kernel(const float* __restrict__ r1, const float* __restrict__ r2,
                                                              const float* __restrict__ c1, const float* __restrict__ c2,
                                                              const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ dmu, float* __restrict__ dnu,
                                                              float* __restrict__ da, float* __restrict__ db, size_t n,
                                                              float kcs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float av = a[i], bv = b[i];
    dmu[i] = -kcs * av * r1[i];
    dnu[i] = -kcs * bv * c1[i];
    da[i] = r2[i] / av;
    db[i] = c2[i] / bv;
}

}  // namespace cocos

extern "C" int cocos_sum_leading(const float* x, float* out, int S, long long n, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && out && S >= 1 && n >= 1, COCOS_ERR_INVALID, "sum_leading: bad arguments");
    if (n % 4 == 0 && aligned16(x) && aligned16(out)) {
        const size_t n4 = (size_t)n / 4;
        hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(out), S, n4);
    } else {
        hipLaunchKernelGGL(sum_leading_tail_kernel, dim3((unsigned)(((size_t)n + 255) / 256)), dim3(256), 0,
                           as_stream(stream), x, out, S, (size_t)n);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_channel_sum_slices(int C, long long N) { return C >= 1 && N >= 1 ? cocos::channel_sum_slice_count(C, N) : 0; }

extern "C" int cocos_channel_sum(const float* dy, float* db, float* partials, int B, int C, long long N, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(dy && db && B >= 1 && C >= 1 && N >= 1 && N <= 0x7fffffffLL, COCOS_ERR_INVALID, "channel_sum: bad arguments");
    const int S = partials ? channel_sum_slice_count(C, N) : 1;       // partials: cocos_channel_sum_slices(C, N) * C floats, or NULL
    const bool ok16 = (N % 4 == 0) && aligned16(dy);                   // an odd per_slice switches the kernel to 4-byte loads
    if (S == 1) {
        hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C, 1), dim3(256), 0, as_stream(stream), dy, db, B, C, (int)N,
                           ok16 ? (int)N : (int)N | 1);
    } else {
        const int per = (int)((((N + S - 1) / S) + 3) & ~3LL);
        hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C, (unsigned)S), dim3(256), 0, as_stream(stream), dy, partials, B, C,
                           (int)N, ok16 ? per : per | 1);
        hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C, 1), dim3(256), 0, as_stream(stream), partials, db, S, C, 1, 1);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_box3_stat_grads(const float* r1, const float* r2, const float* c1, const float* c2, const float* a,
                                     const float* b, float* dmu, float* dnu, float* da, float* db, long long n,
                                     float k_unfolded, float scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(r1 && r2 && c1 && c2 && a && b && dmu && dnu && da && db && n >= 1, COCOS_ERR_INVALID,
                  "box3_stat_grads: bad arguments");
    hipLaunchKernelGGL(box3_stat_grads_kernel, dim3((unsigned)(((size_t)n + 255) / 256)), dim3(256), 0, as_stream(stream), r1,
                       r2, c1, c2, a, b, dmu, dnu, da, db, (size_t)n, k_unfolded * scale);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_concat2_amax(const float* a, const float* b, float* out, int B, long long na, long long nb,
                                  float* amax_inout_dev, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(a && b && out, COCOS_ERR_INVALID, "concat2_amax: null pointer");
    COCOS_REQUIRE(B >= 1 && na >= 4 && nb >= 4 && na % 4 == 0 && nb % 4 == 0, COCOS_ERR_UNSUPPORTED,
                  "concat2_amax: per-sample sizes must be positive multiples of 4 (na=%lld nb=%lld)", na, nb);
    COCOS_REQUIRE(aligned16(a) && aligned16(b) && aligned16(out), COCOS_ERR_INVALID,
                  "concat2_amax: pointers must be 16-byte aligned");
    const size_t total4 = (size_t)B * (size_t)((na + nb) / 4);
    const unsigned blocks = (unsigned)std::min<size_t>(512, (total4 + 255) / 256);
    hipLaunchKernelGGL(concat2_amax_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const f32x4*>(a), reinterpret_cast<const f32x4*>(b), reinterpret_cast<f32x4*>(out),
                       (size_t)(na / 4), (size_t)(nb / 4), total4, reinterpret_cast<unsigned*>(amax_inout_dev));
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_split_f16_chan_mask(const float* x, void* hi, void* lo, int B, int C, int N, const float* amax_dev,
                                         float* scale_out_dev, unsigned* mask_inout_dev, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && hi && lo && amax_dev, COCOS_ERR_INVALID, "split_f16_chan_mask: null pointer");
    COCOS_REQUIRE(B >= 1 && C >= 1 && C <= 1024 && N >= 4 && N % 4 == 0, COCOS_ERR_UNSUPPORTED,
                  "split_f16_chan_mask: needs N %% 4 == 0, C <= 1024 (B=%d C=%d N=%d)", B, C, N);
    COCOS_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(hi) & 7u) == 0 && (reinterpret_cast<uintptr_t>(lo) & 7u) == 0,
                  COCOS_ERR_INVALID, "split_f16_chan_mask: x must be 16-byte, the planes 8-byte aligned");
    const size_t n4 = (size_t)B * C * (N / 4);
    COCOS_REQUIRE((n4 + 255) / 256 <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "split_f16_chan_mask: tensor too large");
    hipLaunchKernelGGL(split_f16_chan_mask_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                       static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), n4, C, N / 4, amax_dev, scale_out_dev,
                       mask_inout_dev);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_proj_weight_planes(const float* w, void* rows_hi, void* rows_lo, void* t_hi, void* t_lo, int Cout,
                                        int Cin, int KpadIn, int KpadOut, const float* amax_dev, float* scale_out_dev,
                                        cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(w && rows_hi && rows_lo && amax_dev, COCOS_ERR_INVALID, "proj_weight_planes: null pointer");
    COCOS_REQUIRE((t_hi == nullptr) == (t_lo == nullptr), COCOS_ERR_INVALID,
                  "proj_weight_planes: the transposed planes come as a hi/lo pair");
    COCOS_REQUIRE(Cout >= 1 && Cin >= 1 && KpadIn >= Cin && KpadOut >= Cout && KpadIn % 8 == 0 && KpadOut % 8 == 0,
                  COCOS_ERR_INVALID, "proj_weight_planes: bad dims Cout=%d Cin=%d KpadIn=%d KpadOut=%d", Cout, Cin, KpadIn,
                  KpadOut);
    for (const void* p : {(const void*)rows_hi, (const void*)rows_lo, (const void*)t_hi, (const void*)t_lo})
        COCOS_REQUIRE(!p || aligned16(p), COCOS_ERR_INVALID, "proj_weight_planes: planes must be 16-byte aligned");
    const dim3 grid((KpadIn + 63) / 64, (KpadOut + 63) / 64);
    hipLaunchKernelGGL(proj_weight_planes_kernel, grid, dim3(256), 0, as_stream(stream), w,
                       static_cast<_Float16*>(rows_hi), static_cast<_Float16*>(rows_lo), static_cast<_Float16*>(t_hi),
                       static_cast<_Float16*>(t_lo), Cout, Cin, KpadIn, KpadOut, amax_dev, scale_out_dev);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
