// K12: per-position statistics of the zero-padded 3x3-unfolded, centred feature vectors WITHOUT unfolding, and
// their backward (gfx950) — the mean mu[p] and the scale a[p] = 1 / (||U_p - mu_p||_2 + eps) that K6 needs to
// turn the boxed K = 256 correlation into the reference's match_kernel = 3 cosine:
//     U_p = F.unfold(x, 3, padding=1)[:, :, p]  (2304 = 9*C entries, zeros outside the map; correspondence.py:276)
//     mu[p]  = sum(U_p) / 2304                = box3(sum_c x)[p] / 2304                 (:277-278, PONO_C)
//     |U_p - mu|^2 = sum(U_p^2) - 2304 mu^2   = box3(sum_c x^2)[p] - 2304 mu[p]^2       (:279-280)
// box3 = zero-padded 3x3 box SUM of a per-position map (self-adjoint).  The reference materialises U
// ([B,2304,HW]: 288 MiB per tensor at HW = 4096); in PyTorch the statistics alone are ~60 small launches and
// ~10 passes over x (0.7 ms per step at B=8); here: one pass over x forward, one read + one write backward.
//   fwd:  x [B,C,h,w] -> mu, a, nrm [B,h*w]
//   bwd:  dx[c,p] = g1[p] + 2 x[c,p] g2[p],  g1 = box3(t1), g2 = box3(t2),
//         t2 = dv = (d nrm)/(2 nrm) with d nrm = -da a^2 (0 where the clamp at 0 is active),  t1 = (dmu - 2*2304*mu*dv)/2304
#include "common.h"

namespace cocos {

// per-position channel sums: WG = 64 positions (16 quads) x 16 channel groups
__global__ __launch_bounds__(256) void unfold3_sums_kernel(const float* __restrict__ x, float* __restrict__ s1,
                                                           float* __restrict__ s2, int C, int N,
                                                           unsigned* __restrict__ amax) {
    __shared__ __attribute__((aligned(16))) float red[2 * 16 * 64];
    const int tid = threadIdx.x, pq = tid & 15, cg = tid >> 4;
    const int b = blockIdx.y, n = blockIdx.x * 64 + pq * 4;
    const float* xb = x + (size_t)b * C * N;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
    float vmax = 0.f;
    const bool vec = (N % 4 == 0) && n < N;
#pragma unroll 4
    for (int c = cg; c < C; c += 16) {      // (unrolled: four independent 16-byte loads in flight per thread, round 4)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec) v = *reinterpret_cast<const f32x4*>(xb + (size_t)c * N + n);
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < N) v[e] = xb[(size_t)c * N + n + e];
        a1 += v;
        a2 += v * v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    *reinterpret_cast<f32x4*>(red + cg * 64 + pq * 4) = a1;
    *reinterpret_cast<f32x4*>(red + 1024 + cg * 64 + pq * 4) = a2;
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, p = tid & 63;
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc += red[which * 1024 + g * 64 + p];
        const int nn = blockIdx.x * 64 + p;
        if (nn < N) (which ? s2 : s1)[(size_t)b * N + nn] = acc;
    }
    // max|x| as a by-product (the scale source of the f16 split of x for the correlation GEMM that follows): one
    // same-address atomic per workgroup
    if (amax) {
        __shared__ float wred[4];
        vmax = wave_max_dpp(vmax);
        if ((tid & 63) == 0) wred[tid >> 6] = vmax;
        __syncthreads();
        if (tid == 0) atomicMax(amax, __float_as_uint(fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]))));
    }
}

__device__ __forceinline__ float box3_at(const float* __restrict__ m, int y, int x, int h, int w) {
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) acc += m[yy * w + xx];
        }
    return acc;
}

__global__ __launch_bounds__(256) void unfold3_finish_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                                             float* __restrict__ mu, float* __restrict__ a,
                                                             float* __restrict__ nrm, int h, int w, float kc, float eps) {
    const int N = h * w, p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= N) return;
    const int y = p / w, x = p % w;
    const float m = box3_at(s1 + (size_t)b * N, y, x, h, w) / kc;
    const float v = fmaxf(box3_at(s2 + (size_t)b * N, y, x, h, w) - kc * m * m, 0.f);
    const float r = sqrtf(v);
    mu[(size_t)b * N + p] = m;
    nrm[(size_t)b * N + p] = r;
    a[(size_t)b * N + p] = 1.0f / (r + eps);
}

// the same for two tensors of one shape in one launch (grid z): K25 leaves the sums of theta and phi together
struct Unfold3FinishPair {
    const float *s1[2], *s2[2];
    float *mu[2], *a[2], *nrm[2];
};
__global__ __launch_bounds__(256) void unfold3_finish_pair_kernel(const Unfold3FinishPair fp, int h, int w, float kc, float eps) {
    const bool second = blockIdx.z != 0;
    const float* __restrict__ s1 = second ? fp.s1[1] : fp.s1[0];
    const float* __restrict__ s2 = second ? fp.s2[1] : fp.s2[0];
    float* __restrict__ mu = second ? fp.mu[1] : fp.mu[0];
    float* __restrict__ a = second ? fp.a[1] : fp.a[0];
    float* __restrict__ nrm = second ? fp.nrm[1] : fp.nrm[0];
    const int N = h * w, p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= N) return;
    const int y = p / w, x = p % w;
    const float m = box3_at(s1 + (size_t)b * N, y, x, h, w) / kc;
    const float v = fmaxf(box3_at(s2 + (size_t)b * N, y, x, h, w) - kc * m * m, 0.f);
    const float r = sqrtf(v);
    mu[(size_t)b * N + p] = m;
    nrm[(size_t)b * N + p] = r;
    a[(size_t)b * N + p] = 1.0f / (r + eps);
}

// the backward's two maps for two tensors of one shape in one launch
struct Unfold3MapsPair {
    const float *mu[2], *a[2], *nrm[2], *dmu[2], *da[2];
    float *g1[2], *g2[2];
};
__device__ __forceinline__ void unfold3_t(const float* mu, const float* a, const float* nrm, const float* dmu, const float* da, int q,
                                          float kc, float& t1, float& t2);
__global__ __launch_bounds__(256) void unfold3_bwd_maps_pair_kernel(const Unfold3MapsPair mp, int h, int w, float kc) {
    const int z = blockIdx.z != 0 ? 1 : 0;
    const float* mu = z ? mp.mu[1] : mp.mu[0];
    const float* a = z ? mp.a[1] : mp.a[0];
    const float* nrm = z ? mp.nrm[1] : mp.nrm[0];
    const float* dmu = z ? mp.dmu[1] : mp.dmu[0];
    const float* da = z ? mp.da[1] : mp.da[0];
    float* g1 = z ? mp.g1[1] : mp.g1[0];
    float* g2 = z ? mp.g2[1] : mp.g2[0];
    const int N = h * w, p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= N) return;
    const size_t o = (size_t)b * N;
    const int y = p / w, x = p % w;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                float t1, t2;
                unfold3_t(mu + o, a + o, nrm + o, dmu ? dmu + o : nullptr, da ? da + o : nullptr, yy * w + xx, kc, t1, t2);
                s1 += t1;
                s2 += t2;
            }
        }
    g1[o + p] = s1;
    g2[o + p] = s2;
}

// t1, t2 of one position (see header)
__device__ __forceinline__ void unfold3_t(const float* mu, const float* a, const float* nrm, const float* dmu,
                                          const float* da, int q, float kc, float& t1, float& t2) {
    const float r = nrm[q], aa = a[q];
    const float dn = da ? -da[q] * aa * aa : 0.f;
    const float dv = r > 0.f ? dn / (2.f * r) : 0.f;          // clamp(min=0) active (constant patch): no gradient
    t2 = dv;
    t1 = ((dmu ? dmu[q] : 0.f) - 2.f * kc * mu[q] * dv) / kc;
}

__global__ __launch_bounds__(256) void unfold3_bwd_maps_kernel(const float* __restrict__ mu, const float* __restrict__ a,
                                                               const float* __restrict__ nrm,
                                                               const float* __restrict__ dmu,
                                                               const float* __restrict__ da, float* __restrict__ g1,
                                                               float* __restrict__ g2, int h, int w, float kc) {
    const int N = h * w, p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= N) return;
    const size_t o = (size_t)b * N;
    const int y = p / w, x = p % w;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                float t1, t2;
                unfold3_t(mu + o, a + o, nrm + o, dmu ? dmu + o : nullptr, da ? da + o : nullptr, yy * w + xx, kc, t1, t2);
                s1 += t1;
                s2 += t2;
            }
        }
    g1[o + p] = s1;
    g2[o + p] = s2;
}

__global__ __launch_bounds__(256) void unfold3_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ g1,
                                                                const float* __restrict__ g2, float* __restrict__ dx,
                                                                int C, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // element index inside one sample
    const int b = blockIdx.y;
    const size_t per = (size_t)C * N;
    if (i >= per) return;
    const int p = (int)(i % N);
    const size_t o = (size_t)b * per + i;
    dx[o] = g1[(size_t)b * N + p] + 2.f * x[o] * g2[(size_t)b * N + p];
}

// the same four elements at a time (N % 4 == 0, 16-byte aligned tensors): round 4 — the scalar version moved 64 MB at 2.2 TB/s
__global__ __launch_bounds__(256) void unfold3_bwd_apply4_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ g1,
                                                                 const f32x4* __restrict__ g2, f32x4* __restrict__ dx,
                                                                 size_t per4, int N4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // quad index inside one sample
    const int b = blockIdx.y;
    if (i >= per4) return;
    const int p = (int)(i % (size_t)N4);
    const size_t o = (size_t)b * per4 + i;
    const f32x4 a = g1[(size_t)b * N4 + p], m = g2[(size_t)b * N4 + p], v = x[o];
    dx[o] = a + 2.f * v * m;
}

}  // namespace cocos

static int unfold3_stats_fwd_impl(const float* x, float* mu, float* a, float* nrm, float* ws, int B, int C, int h, int w,
                                  float k_unfolded, float eps, float* amax_inout, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && mu && a && nrm && ws, COCOS_ERR_INVALID, "unfold3_stats_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && h >= 1 && w >= 1 && k_unfolded > 0.f, COCOS_ERR_INVALID,
                  "unfold3_stats_fwd: bad dims B=%d C=%d h=%d w=%d", B, C, h, w);
    const int N = h * w;
    hipStream_t s = as_stream(stream);
    float* s1 = ws;
    float* s2 = ws + (size_t)B * N;
    hipLaunchKernelGGL(unfold3_sums_kernel, dim3((N + 63) / 64, B), dim3(256), 0, s, x, s1, s2, C, N,
                       reinterpret_cast<unsigned*>(amax_inout));
    hipLaunchKernelGGL(unfold3_finish_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, s1, s2, mu, a, nrm, h, w,
                       k_unfolded, eps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_unfold3_stats_fwd(const float* x, float* mu, float* a, float* nrm, float* ws /* 2*B*h*w */,
                                       int B, int C, int h, int w, float k_unfolded, float eps, cocos_stream_t stream) {
    return unfold3_stats_fwd_impl(x, mu, a, nrm, ws, B, C, h, w, k_unfolded, eps, nullptr, stream);
}

// Same, and *amax_inout_dev = max(*amax_inout_dev, max|x|) (a cell holding a finite value >= 0): x is read once for both.
extern "C" int cocos_unfold3_stats_fwd_amax(const float* x, float* mu, float* a, float* nrm, float* ws, int B, int C, int h,
                                            int w, float k_unfolded, float eps, float* amax_inout_dev,
                                            cocos_stream_t stream) {
    COCOS_REQUIRE(amax_inout_dev, COCOS_ERR_INVALID, "unfold3_stats_fwd_amax: null amax cell");
    return unfold3_stats_fwd_impl(x, mu, a, nrm, ws, B, C, h, w, k_unfolded, eps, amax_inout_dev, stream);
}

extern "C" int cocos_unfold3_stats_bwd(const float* x, const float* mu, const float* a, const float* nrm,
                                       const float* dmu /* nullable */, const float* da /* nullable */, float* dx,
                                       float* ws /* 2*B*h*w */, int B, int C, int h, int w, float k_unfolded,
                                       cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && mu && a && nrm && dx && ws, COCOS_ERR_INVALID, "unfold3_stats_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && h >= 1 && w >= 1 && k_unfolded > 0.f, COCOS_ERR_INVALID,
                  "unfold3_stats_bwd: bad dims B=%d C=%d h=%d w=%d", B, C, h, w);
    const int N = h * w;
    const size_t per = (size_t)C * N;
    COCOS_REQUIRE((per + 255) / 256 <= 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "unfold3_stats_bwd: tensor too large");
    hipStream_t s = as_stream(stream);
    float* g1 = ws;
    float* g2 = ws + (size_t)B * N;
    hipLaunchKernelGGL(unfold3_bwd_maps_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, mu, a, nrm, dmu, da, g1, g2, h,
                       w, k_unfolded);
    if (N % 4 == 0 && aligned16(x) && aligned16(dx) && aligned16(ws) && ((size_t)B * N) % 4 == 0)
        hipLaunchKernelGGL(unfold3_bwd_apply4_kernel, dim3((unsigned)((per / 4 + 255) / 256), B), dim3(256), 0, s,
                           reinterpret_cast<const f32x4*>(x), reinterpret_cast<const f32x4*>(g1), reinterpret_cast<const f32x4*>(g2),
                           reinterpret_cast<f32x4*>(dx), per / 4, N / 4);
    else
        hipLaunchKernelGGL(unfold3_bwd_apply_kernel, dim3((unsigned)((per + 255) / 256), B), dim3(256), 0, s, x, g1, g2, dx, C, N);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// Only the two per-position maps of the backward (ws = [g1 | g2], B*h*w floats each): dx[c,p] = g1[p] + 2 x[c,p] g2[p] is then
// applied by the consumer — round 6: K24 (cocos_proj_bwd_input_f16x3, mode 1) folds it into the projection's backward.
extern "C" int cocos_unfold3_stats_bwd_maps(const float* mu, const float* a, const float* nrm, const float* dmu /* nullable */,
                                            const float* da /* nullable */, float* ws /* 2*B*h*w */, int B, int h, int w,
                                            float k_unfolded, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(mu && a && nrm && ws, COCOS_ERR_INVALID, "unfold3_stats_bwd_maps: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && k_unfolded > 0.f, COCOS_ERR_INVALID,
                  "unfold3_stats_bwd_maps: bad dims B=%d h=%d w=%d", B, h, w);
    const int N = h * w;
    hipLaunchKernelGGL(unfold3_bwd_maps_kernel, dim3((N + 255) / 256, B), dim3(256), 0, as_stream(stream), mu, a, nrm, dmu, da, ws,
                       ws + (size_t)B * N, h, w, k_unfolded);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// mu, a, nrm [B,h*w] of two tensors from their per-position channel sums s1 = sum_c x, s2 = sum_c x^2 [B,h*w] (K25 computes
// them in the projection's epilogue) — cocos_unfold3_stats_fwd's second half, for a pair, in one launch.
extern "C" int cocos_unfold3_stats_finish_pair(const float* s1_0, const float* s2_0, float* mu0, float* a0, float* nrm0,
                                               const float* s1_1, const float* s2_1, float* mu1, float* a1, float* nrm1, int B, int h,
                                               int w, float k_unfolded, float eps, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(s1_0 && s2_0 && mu0 && a0 && nrm0 && s1_1 && s2_1 && mu1 && a1 && nrm1, COCOS_ERR_INVALID,
                  "unfold3_stats_finish_pair: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && k_unfolded > 0.f, COCOS_ERR_INVALID,
                  "unfold3_stats_finish_pair: bad dims B=%d h=%d w=%d", B, h, w);
    Unfold3FinishPair fp;
    fp.s1[0] = s1_0; fp.s1[1] = s1_1; fp.s2[0] = s2_0; fp.s2[1] = s2_1;
    fp.mu[0] = mu0; fp.mu[1] = mu1; fp.a[0] = a0; fp.a[1] = a1; fp.nrm[0] = nrm0; fp.nrm[1] = nrm1;
    hipLaunchKernelGGL(unfold3_finish_pair_kernel, dim3((h * w + 255) / 256, B, 2), dim3(256), 0, as_stream(stream), fp, h, w, k_unfolded,
                       eps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// cocos_unfold3_stats_bwd_maps for two tensors of one shape in one launch: ws_i = [g1 | g2] (B*h*w floats each).
extern "C" int cocos_unfold3_stats_bwd_maps_pair(const float* mu0, const float* a0, const float* nrm0, const float* dmu0, const float* da0,
                                                 float* ws0, const float* mu1, const float* a1, const float* nrm1, const float* dmu1,
                                                 const float* da1, float* ws1, int B, int h, int w, float k_unfolded,
                                                 cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(mu0 && a0 && nrm0 && ws0 && mu1 && a1 && nrm1 && ws1, COCOS_ERR_INVALID, "unfold3_stats_bwd_maps_pair: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && k_unfolded > 0.f, COCOS_ERR_INVALID,
                  "unfold3_stats_bwd_maps_pair: bad dims B=%d h=%d w=%d", B, h, w);
    const size_t BN = (size_t)B * h * w;
    Unfold3MapsPair mp;
    mp.mu[0] = mu0; mp.mu[1] = mu1; mp.a[0] = a0; mp.a[1] = a1; mp.nrm[0] = nrm0; mp.nrm[1] = nrm1;
    mp.dmu[0] = dmu0; mp.dmu[1] = dmu1; mp.da[0] = da0; mp.da[1] = da1;
    mp.g1[0] = ws0; mp.g1[1] = ws1; mp.g2[0] = ws0 + BN; mp.g2[1] = ws1 + BN;
    hipLaunchKernelGGL(unfold3_bwd_maps_pair_kernel, dim3((h * w + 255) / 256, B, 2), dim3(256), 0, as_stream(stream), mp, h, w, k_unfolded);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
