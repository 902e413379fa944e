// K1: centre + L2-normalise the correlation operands, and its backward (gfx950).
//
// Replaces correspondence.py:277-280 (theta) and :287-289 (phi):
//     x = x - x.mean(dim=dim_mean, keepdim=True)          dim_mean = 1 if PONO_C else -1
//     x = x / (torch.norm(x, 2, 1, keepdim=True) + sys.float_info.epsilon)
// The reference does this in 4-5 elementwise/reduction launches per tensor, each a full HBM
// pass (plus autograd copies); here it is one pass-structured kernel per tensor: HBM-bound,
// x is read once from HBM (re-reads of the 64-position column block hit L1/L2).
//
// Layout: x, y are channel-major [B, K, N]; a workgroup owns 64 consecutive positions, its 4
// waves split the K channels (wave w takes k = w, w+4, ...), lanes run along positions so every
// global access is a coalesced 256-byte row segment.
#include "common.h"

namespace cocos {

constexpr int CN_POS = 64;   // positions per workgroup (one per lane)

// Sum the 4 per-wave partials of up to 3 quantities; every thread gets the totals.
template <int NQ>
__device__ __forceinline__ void reduce_waves(float (&v)[NQ], float* red /*[NQ][4][64]*/, int wave,
                                             int lane) {
    __syncthreads();   // protect `red` from the previous use
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[(q * 4 + wave) * CN_POS + lane] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        v[q] = red[(q * 4 + 0) * CN_POS + lane] + red[(q * 4 + 1) * CN_POS + lane] +
               red[(q * 4 + 2) * CN_POS + lane] + red[(q * 4 + 3) * CN_POS + lane];
}

// mean over N for each (b, k) row:  out[b*K + k] = mean_n f(b,k,n)
// MODE 0: f = x.   MODE 1 (backward): f = u_n * dy - (a_n / nrm_n) * y  with u = 1/(nrm+eps).
template <int MODE>
__global__ __launch_bounds__(256) void row_mean_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ y,
                                                       const float* __restrict__ nrm,
                                                       const float* __restrict__ acol,
                                                       float* __restrict__ out, int K, int N,
                                                       float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x;            // b*K + k
    const int b = row / K;
    const float* xr = x + (size_t)row * N;
    const float* yr = (MODE == 1) ? y + (size_t)row * N : nullptr;
    float acc = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        if (MODE == 0) {
            acc += xr[n];
        } else {
            const float nn = nrm[(size_t)b * N + n];
            const float u = 1.0f / (nn + eps);
            const float g = (nn > 0.f) ? acol[(size_t)b * N + n] / nn : 0.f;
            acc += u * xr[n] - g * yr[n];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = (red[0] + red[1] + red[2] + red[3]) / (float)N;
}

// Forward.  PONO_C: mean over K per position (computed here); else mean over N per channel
// (precomputed in row_mean[b*K + k]).
template <bool PONO_C>
__global__ __launch_bounds__(256) void center_l2norm_fwd_kernel(const float* __restrict__ x,
                                                                float* __restrict__ y,
                                                                float* __restrict__ norm_out,
                                                                const float* __restrict__ row_mean,
                                                                int K, int N, float eps, bool center) {
    __shared__ float red[3 * 4 * CN_POS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int n = blockIdx.x * CN_POS + lane;
    const bool ok = n < N;
    const float* xb = x + (size_t)b * K * N;
    float* yb = y + (size_t)b * K * N;

    float mean = 0.f;
    if (PONO_C && center) {   // !center: plain L2 normalisation (util.feature_normalize)
        float part[1] = {0.f};
        if (ok)
            for (int k = wave; k < K; k += 4) part[0] += xb[(size_t)k * N + n];
        reduce_waves<1>(part, red, wave, lane);
        mean = part[0] / (float)K;
    }
    float ss[1] = {0.f};
    if (ok)
        for (int k = wave; k < K; k += 4) {
            const float m = PONO_C ? mean : row_mean[(size_t)b * K + k];
            const float d = xb[(size_t)k * N + n] - m;
            ss[0] += d * d;
        }
    reduce_waves<1>(ss, red, wave, lane);
    const float nrm = sqrtf(ss[0]);
    const float u = 1.0f / (nrm + eps);
    if (ok) {
        for (int k = wave; k < K; k += 4) {
            const float m = PONO_C ? mean : row_mean[(size_t)b * K + k];
            yb[(size_t)k * N + n] = (xb[(size_t)k * N + n] - m) * u;
        }
        if (wave == 0) norm_out[(size_t)b * N + n] = nrm;
    }
}

// Backward.  With u = 1/(nrm+eps), a = sum_k dy*y:   dxc = u*dy - (a/nrm)*y   (second term 0
// when nrm == 0, matching the zero sub-gradient torch.norm uses there), then the centring
// backward  dx = dxc - mean(dxc)  over the same axis as the forward mean.
//   PONO_C: everything in this kernel.   else: this kernel only writes a[b,n] to `acol`
//   (pass 1); row_mean_kernel<1> then gives mean_n(dxc) per channel and pass 2 finishes.
template <bool PONO_C, int PASS>
__global__ __launch_bounds__(256) void center_l2norm_bwd_kernel(
    const float* __restrict__ y, const float* __restrict__ nrm_in, const float* __restrict__ dy,
    float* __restrict__ dx, float* __restrict__ acol, const float* __restrict__ row_mean, int K,
    int N, float eps, bool center) {
    __shared__ float red[3 * 4 * CN_POS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int n = blockIdx.x * CN_POS + lane;
    const bool ok = n < N;
    const float* yb = y + (size_t)b * K * N;
    const float* dyb = dy + (size_t)b * K * N;
    float* dxb = dx + (size_t)b * K * N;

    const float nrm = ok ? nrm_in[(size_t)b * N + n] : 1.f;
    const float u = 1.0f / (nrm + eps);

    if (PONO_C || PASS == 1) {
        float s[3] = {0.f, 0.f, 0.f};   // sum dy*y, sum dy, sum y
        if (ok)
            for (int k = wave; k < K; k += 4) {
                const float yy = yb[(size_t)k * N + n], dd = dyb[(size_t)k * N + n];
                s[0] += dd * yy;
                s[1] += dd;
                s[2] += yy;
            }
        reduce_waves<3>(s, red, wave, lane);
        if (!PONO_C) {
            if (ok && wave == 0) acol[(size_t)b * N + n] = s[0];
            return;
        }
        const float g = (nrm > 0.f) ? s[0] / nrm : 0.f;
        const float mean_dxc = center ? (u * s[1] - g * s[2]) / (float)K : 0.f;
        if (ok)
            for (int k = wave; k < K; k += 4)
                dxb[(size_t)k * N + n] =
                    u * dyb[(size_t)k * N + n] - g * yb[(size_t)k * N + n] - mean_dxc;
    } else {   // !PONO_C, pass 2
        const float g = (ok && nrm > 0.f) ? acol[(size_t)b * N + n] / nrm : 0.f;
        if (ok)
            for (int k = wave; k < K; k += 4)
                dxb[(size_t)k * N + n] = u * dyb[(size_t)k * N + n] - g * yb[(size_t)k * N + n] -
                                         row_mean[(size_t)b * K + k];
    }
}

// ---- register-resident PONO_C variants (K <= 512, K % 16 == 0, N % 4 == 0) ------------------------
// A workgroup owns 64 consecutive positions; thread (pq = tid & 15, cg = tid >> 4) keeps the float4
// of positions 4pq..4pq+3 for its K/16 channels in registers, so x (and dy, y) cross HBM exactly once;
// the per-position sums over channels go through a 16 x 64 LDS reduction.
constexpr int CNR_MAXI = 32;   // K / 16 <= 32

template <int NQ>
__device__ __forceinline__ void reduce_cg(f32x4 (&v)[NQ], float* red /*[NQ][16][64]*/, int cg, int pq) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(red + (q * 16 + cg) * 64 + pq * 4) = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 16; ++g) acc += *reinterpret_cast<const f32x4*>(red + (q * 16 + g) * 64 + pq * 4);
        v[q] = acc;
    }
}

template <int NI>   // NI = K / 16
__global__ __launch_bounds__(256) void center_l2norm_fwd_reg_kernel(const float* __restrict__ x,
                                                                    float* __restrict__ y,
                                                                    float* __restrict__ norm_out, int K,
                                                                    int N, float eps, bool center) {
    __shared__ __attribute__((aligned(16))) float red[16 * 64];
    const int tid = threadIdx.x, pq = tid & 15, cg = tid >> 4;
    const int b = blockIdx.y, n = blockIdx.x * 64 + pq * 4;
    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(x + (size_t)b * K * N, (size_t)K * N * 4);
    const bool ok = n < N;   // N % 4 == 0: a float4 is entirely in or out
    f32x4 v[NI];
    f32x4 s[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        v[i] = buf_load4(x_rs, ok ? (unsigned)((cg + 16 * i) * N + n) * 4u : kBufOob);
        s[0] += v[i];
    }
    if (center) reduce_cg<1>(s, red, cg, pq);          // uniform branch (kernel argument)
    const f32x4 mean = s[0] * (center ? 1.0f / (float)K : 0.0f);
    f32x4 ss[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        v[i] -= mean;
        ss[0] += v[i] * v[i];
    }
    reduce_cg<1>(ss, red, cg, pq);
    f32x4 nrm, u;
#pragma unroll
    for (int e = 0; e < 4; ++e) { nrm[e] = sqrtf(ss[0][e]); u[e] = 1.0f / (nrm[e] + eps); }
    if (!ok) return;
    float* yb = y + (size_t)b * K * N;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        *reinterpret_cast<f32x4*>(yb + (size_t)(cg + 16 * i) * N + n) = v[i] * u;
    if (cg == 0) *reinterpret_cast<f32x4*>(norm_out + (size_t)b * N + n) = nrm;
}

template <int NI>
__global__ __launch_bounds__(256) void center_l2norm_bwd_reg_kernel(const float* __restrict__ y,
                                                                    const float* __restrict__ nrm_in,
                                                                    const float* __restrict__ dy,
                                                                    float* __restrict__ dx, int K, int N,
                                                                    float eps, bool center,
                                                                    unsigned* __restrict__ dx_amax) {
    __shared__ __attribute__((aligned(16))) float red[3 * 16 * 64];
    const int tid = threadIdx.x, pq = tid & 15, cg = tid >> 4;
    const int b = blockIdx.y, n = blockIdx.x * 64 + pq * 4;
    const __amdgpu_buffer_rsrc_t y_rs = make_rsrc(y + (size_t)b * K * N, (size_t)K * N * 4);
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(dy + (size_t)b * K * N, (size_t)K * N * 4);
    const bool ok = n < N;
    f32x4 yy[NI], gg[NI];
    f32x4 s[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const unsigned off = ok ? (unsigned)((cg + 16 * i) * N + n) * 4u : kBufOob;
        yy[i] = buf_load4(y_rs, off);
        gg[i] = buf_load4(g_rs, off);
        s[0] += gg[i] * yy[i];
        s[1] += gg[i];
        s[2] += yy[i];
    }
    reduce_cg<3>(s, red, cg, pq);
    if (!ok && !dx_amax) return;
    const f32x4 nrm = ok ? *reinterpret_cast<const f32x4*>(nrm_in + (size_t)b * N + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    f32x4 u, g, m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = 1.0f / (nrm[e] + eps);
        g[e] = nrm[e] > 0.f ? s[0][e] / nrm[e] : 0.f;
        m[e] = center ? (u[e] * s[1][e] - g[e] * s[2][e]) / (float)K : 0.f;
    }
    float* dxb = dx + (size_t)b * K * N;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 d = u * gg[i] - g * yy[i] - m;
        if (ok) *reinterpret_cast<f32x4*>(dxb + (size_t)(cg + 16 * i) * N + n) = d;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
    }
    // max|dx| as a by-product (the consumer, K0's backward, needs it for the scale of its f16 split and would
    // otherwise read all of dx once more): non-negative floats order like their bit patterns
    if (dx_amax) {
        // one same-address atomic per WORKGROUP: they serialise at the memory side (~10 ns each) — one per wave
        // (2048 per launch) put 16 us on the critical path, 512 of them hide under the kernel
        const float wmax = wave_max_dpp(ok ? amax : 0.f);
        __syncthreads();                                  // (red[] was last read in reduce_cg)
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmax;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(dx_amax, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    }
}

// ---- K1 forward that emits the f16 hi/lo OPERAND PLANES of the split-precision correlation kernels itself (round 3).
// The fp32 result qn / kn used to be written (4 B/element) only to be read again by two cocos_split_f16 launches per
// tensor (position-major planes for the forward, channel-major ones for the backward): 24 B/element of traffic and six
// launches per step for what is 12 B/element here.  Thread (pq, cg) owns 16 CONSECUTIVE channels 16cg..16cg+15 of
// positions 4pq..4pq+3, so a position-major row piece (16 channels = 32 bytes per plane) leaves as two 16-byte stores
// and a channel-major piece (4 positions of one channel) as one 8-byte store.  plane_scale * y = hi + lo, hi rounded to
// nearest exactly like split_f16.hip (the planes are bit-identical to the ones the split kernels would have made).
typedef _Float16 cn_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 cn_f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void center_l2norm_fwd_planes_kernel(
    const float* __restrict__ x, float* __restrict__ norm_out, _Float16* __restrict__ ph, _Float16* __restrict__ pl,
    _Float16* __restrict__ ch, _Float16* __restrict__ cl, int N, float eps, bool center, float plane_scale) {
    constexpr int K = 256, NI = 16;
    __shared__ __attribute__((aligned(16))) float red[16 * 64];
    const int tid = threadIdx.x, pq = tid & 15, cg = tid >> 4;
    const int b = blockIdx.y, n = blockIdx.x * 64 + pq * 4;
    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(x + (size_t)b * K * N, (size_t)K * N * 4);
    const bool ok = n < N;   // N % 4 == 0
    f32x4 v[NI];
    f32x4 s[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        v[i] = buf_load4(x_rs, ok ? (unsigned)((cg * 16 + i) * N + n) * 4u : kBufOob);
        s[0] += v[i];
    }
    if (center) reduce_cg<1>(s, red, cg, pq);
    const f32x4 mean = s[0] * (center ? 1.0f / (float)K : 0.0f);
    f32x4 ss[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        v[i] -= mean;
        ss[0] += v[i] * v[i];
    }
    reduce_cg<1>(ss, red, cg, pq);
    f32x4 nrm, u;
#pragma unroll
    for (int e = 0; e < 4; ++e) { nrm[e] = sqrtf(ss[0][e]); u[e] = 1.0f / (nrm[e] + eps); }
    if (!ok) return;
    if (cg == 0) *reinterpret_cast<f32x4*>(norm_out + (size_t)b * N + n) = nrm;
    // hi / lo of the 16 channels x 4 positions of this thread (y rounded exactly as the fp32 kernel rounds it, then times
    // the power-of-two plane scale: the planes are bit-identical to cocos_split_f16 of that kernel's output)
    _Float16 hh[NI][4], ll[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = (v[i][e] * u[e]) * plane_scale;
            hh[i][e] = (_Float16)y;
            ll[i][e] = (_Float16)(y - (float)hh[i][e]);
        }
    if (ch) {      // channel-major planes [B,256,N]
        _Float16* chb = ch + (size_t)b * K * N;
        _Float16* clb = cl + (size_t)b * K * N;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const size_t off = (size_t)(cg * 16 + i) * N + n;
            *reinterpret_cast<cn_f16x4*>(chb + off) = cn_f16x4{hh[i][0], hh[i][1], hh[i][2], hh[i][3]};
            *reinterpret_cast<cn_f16x4*>(clb + off) = cn_f16x4{ll[i][0], ll[i][1], ll[i][2], ll[i][3]};
        }
    }
    // position-major planes [B,N,256]: 32 bytes per plane and position
    _Float16* phb = ph + ((size_t)b * N + n) * K + cg * 16;
    _Float16* plb = pl + ((size_t)b * N + n) * K + cg * 16;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            cn_f16x8 a, c;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = hh[8 * g + j][e]; c[j] = ll[8 * g + j][e]; }
            *reinterpret_cast<cn_f16x8*>(phb + (size_t)e * K + 8 * g) = a;
            *reinterpret_cast<cn_f16x8*>(plb + (size_t)e * K + 8 * g) = c;
        }
}

// Backward of the planes flavour: y is read back from the channel-major planes, y = (hi + lo) / plane_scale (22 mantissa
// bits: the gradient's own rounding is coarser), everything else as center_l2norm_bwd_reg_kernel<16>.
__global__ __launch_bounds__(256) void center_l2norm_bwd_planes_kernel(
    const _Float16* __restrict__ ch, const _Float16* __restrict__ cl, const float* __restrict__ nrm_in,
    const float* __restrict__ dy, float* __restrict__ dx, int N, float eps, bool center, float inv_plane_scale,
    unsigned* __restrict__ dx_amax) {
    constexpr int K = 256, NI = 16;
    __shared__ __attribute__((aligned(16))) float red[3 * 16 * 64];
    const int tid = threadIdx.x, pq = tid & 15, cg = tid >> 4;
    const int b = blockIdx.y, n = blockIdx.x * 64 + pq * 4;
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(dy + (size_t)b * K * N, (size_t)K * N * 4);
    const __amdgpu_buffer_rsrc_t h_rs = make_rsrc(ch + (size_t)b * K * N, (size_t)K * N * 2);
    const __amdgpu_buffer_rsrc_t l_rs = make_rsrc(cl + (size_t)b * K * N, (size_t)K * N * 2);
    const bool ok = n < N;
    f32x4 yy[NI], gg[NI];
    f32x4 s[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const unsigned e0 = (unsigned)((cg + 16 * i) * N + n);
        gg[i] = buf_load4(g_rs, ok ? e0 * 4u : kBufOob);
        typedef unsigned int cn_u32x2 __attribute__((ext_vector_type(2)));
        const cn_u32x2 hw = __builtin_amdgcn_raw_buffer_load_b64(h_rs, (int)(ok ? e0 * 2u : kBufOob), 0, 0);
        const cn_u32x2 lw = __builtin_amdgcn_raw_buffer_load_b64(l_rs, (int)(ok ? e0 * 2u : kBufOob), 0, 0);
        const cn_f16x4 h4 = __builtin_bit_cast(cn_f16x4, hw), l4 = __builtin_bit_cast(cn_f16x4, lw);
#pragma unroll
        for (int e = 0; e < 4; ++e) yy[i][e] = ((float)h4[e] + (float)l4[e]) * inv_plane_scale;
        s[0] += gg[i] * yy[i];
        s[1] += gg[i];
        s[2] += yy[i];
    }
    reduce_cg<3>(s, red, cg, pq);
    if (!ok && !dx_amax) return;
    const f32x4 nrm = ok ? *reinterpret_cast<const f32x4*>(nrm_in + (size_t)b * N + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    f32x4 u, g, m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = 1.0f / (nrm[e] + eps);
        g[e] = nrm[e] > 0.f ? s[0][e] / nrm[e] : 0.f;
        m[e] = center ? (u[e] * s[1][e] - g[e] * s[2][e]) / (float)K : 0.f;
    }
    float* dxb = dx + (size_t)b * K * N;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 d = u * gg[i] - g * yy[i] - m;
        if (ok) *reinterpret_cast<f32x4*>(dxb + (size_t)(cg + 16 * i) * N + n) = d;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
    }
    if (dx_amax) {
        const float wmax = wave_max_dpp(ok ? amax : 0.f);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmax;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(dx_amax, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    }
}

template <bool BWD>
static bool launch_reg_variant(const float* a, const float* nrm_in, const float* dy, float* out,
                               float* norm_out, int B, int K, int N, float eps, bool center, hipStream_t s,
                               unsigned* dx_amax = nullptr) {
    if (K % 16 != 0 || K / 16 > CNR_MAXI || N % 4 != 0 || !aligned16(a) || !aligned16(out)) return false;
    const dim3 grid((N + 63) / 64, B);
#define COCOS_NI(NI)                                                                                     \
    case NI:                                                                                             \
        if (BWD) hipLaunchKernelGGL(center_l2norm_bwd_reg_kernel<NI>, grid, dim3(256), 0, s, a, nrm_in, dy, out, K, N, eps, center, dx_amax); \
        else hipLaunchKernelGGL(center_l2norm_fwd_reg_kernel<NI>, grid, dim3(256), 0, s, a, out, norm_out, K, N, eps, center);      \
        return true;
    switch (K / 16) {
        COCOS_NI(1) COCOS_NI(2) COCOS_NI(4) COCOS_NI(8) COCOS_NI(16) COCOS_NI(32)
        default: return false;
    }
#undef COCOS_NI
}

}  // namespace cocos

extern "C" int cocos_center_l2norm_fwd(const float* x, float* y, float* norm, float* row_ws, int B,
                                       int K, int N, int center_over_channels, float eps,
                                       cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && y && norm, COCOS_ERR_INVALID, "center_l2norm_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && K >= 1 && N >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "center_l2norm_fwd: bad dims B=%d K=%d N=%d", B, K, N);
    hipStream_t s = as_stream(stream);
    const dim3 grid((N + CN_POS - 1) / CN_POS, B);
    COCOS_REQUIRE(center_over_channels >= 0 && center_over_channels <= 2, COCOS_ERR_INVALID,
                  "center_l2norm_fwd: mode %d (0 positions, 1 channels, 2 none)", center_over_channels);
    const bool center = center_over_channels != COCOS_CENTER_NONE;
    if (center_over_channels) {
        if (!launch_reg_variant<false>(x, nullptr, nullptr, y, norm, B, K, N, eps, center, s))
            hipLaunchKernelGGL(center_l2norm_fwd_kernel<true>, grid, dim3(256), 0, s, x, y, norm,
                               (const float*)nullptr, K, N, eps, center);
    } else {
        COCOS_REQUIRE(row_ws, COCOS_ERR_INVALID,
                      "center_l2norm_fwd: row_ws [B*K] required when centring over positions");
        hipLaunchKernelGGL(row_mean_kernel<0>, dim3(B * K), dim3(256), 0, s, x,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           row_ws, K, N, eps);
        hipLaunchKernelGGL(center_l2norm_fwd_kernel<false>, grid, dim3(256), 0, s, x, y, norm,
                           (const float*)row_ws, K, N, eps, true);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_absmax_accumulate(const float* x, long long n, float* inout_dev, cocos_stream_t stream);

static int center_l2norm_bwd_impl(const float* y, const float* norm, const float* dy, float* dx, float* col_ws,
                                  float* row_ws, int B, int K, int N, int center_over_channels, float eps,
                                  float* dx_amax_inout, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(y && norm && dy && dx, COCOS_ERR_INVALID, "center_l2norm_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && K >= 1 && N >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "center_l2norm_bwd: bad dims B=%d K=%d N=%d", B, K, N);
    hipStream_t s = as_stream(stream);
    const dim3 grid((N + CN_POS - 1) / CN_POS, B);
    COCOS_REQUIRE(center_over_channels >= 0 && center_over_channels <= 2, COCOS_ERR_INVALID,
                  "center_l2norm_bwd: mode %d (0 positions, 1 channels, 2 none)", center_over_channels);
    const bool center = center_over_channels != COCOS_CENTER_NONE;
    bool amax_done = false;
    if (center_over_channels) {
        if (launch_reg_variant<true>(y, norm, dy, dx, nullptr, B, K, N, eps, center, s,
                                     reinterpret_cast<unsigned*>(dx_amax_inout)))
            amax_done = true;
        else
            hipLaunchKernelGGL((center_l2norm_bwd_kernel<true, 1>), grid, dim3(256), 0, s, y, norm,
                               dy, dx, (float*)nullptr, (const float*)nullptr, K, N, eps, center);
    } else {
        COCOS_REQUIRE(col_ws && row_ws, COCOS_ERR_INVALID,
                      "center_l2norm_bwd: col_ws [B*N] and row_ws [B*K] required");
        hipLaunchKernelGGL((center_l2norm_bwd_kernel<false, 1>), grid, dim3(256), 0, s, y, norm,
                           dy, dx, col_ws, (const float*)nullptr, K, N, eps, true);
        hipLaunchKernelGGL(row_mean_kernel<1>, dim3(B * K), dim3(256), 0, s, dy, y, norm,
                           (const float*)col_ws, row_ws, K, N, eps);
        hipLaunchKernelGGL((center_l2norm_bwd_kernel<false, 2>), grid, dim3(256), 0, s, y, norm,
                           dy, dx, col_ws, (const float*)row_ws, K, N, eps, true);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    if (dx_amax_inout && !amax_done)         // shapes the fused kernel does not take: one separate pass
        return cocos_absmax_accumulate(dx, (long long)B * K * N, dx_amax_inout, stream);
    return COCOS_OK;
}

extern "C" int cocos_center_l2norm_bwd(const float* y, const float* norm, const float* dy,
                                       float* dx, float* col_ws, float* row_ws, int B, int K,
                                       int N, int center_over_channels, float eps,
                                       cocos_stream_t stream) {
    return center_l2norm_bwd_impl(y, norm, dy, dx, col_ws, row_ws, B, K, N, center_over_channels, eps, nullptr,
                                  stream);
}

// Same, and on return *dx_amax_inout = max(*dx_amax_inout, max|dx|) (cell must hold a finite value >= 0): the scale
// source of K0's backward, produced while dx is written instead of by another pass over it.
extern "C" int cocos_center_l2norm_bwd_amax(const float* y, const float* norm, const float* dy, float* dx,
                                            float* col_ws, float* row_ws, int B, int K, int N,
                                            int center_over_channels, float eps, float* dx_amax_inout,
                                            cocos_stream_t stream) {
    COCOS_REQUIRE(dx_amax_inout, COCOS_ERR_INVALID, "center_l2norm_bwd_amax: null amax cell");
    return center_l2norm_bwd_impl(y, norm, dy, dx, col_ws, row_ws, B, K, N, center_over_channels, eps,
                                  dx_amax_inout, stream);
}

// K1 forward for the split-precision correlation kernels (K == 256, N % 4 == 0; PONO_C centring or none): writes the row
// norms and the operand planes of plane_scale * y directly — position-major [B,N,256] (pos_hi / pos_lo: always) and
// channel-major [B,256,N] (chan_hi / chan_lo: nullable, wanted by the backward) — and NO fp32 y.
extern "C" int cocos_center_l2norm_fwd_planes(const float* x, float* norm, void* pos_hi, void* pos_lo, void* chan_hi,
                                              void* chan_lo, int B, int K, int N, int center_over_channels, float eps,
                                              float plane_scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && norm && pos_hi && pos_lo, COCOS_ERR_INVALID, "center_l2norm_fwd_planes: null pointer");
    COCOS_REQUIRE((chan_hi == nullptr) == (chan_lo == nullptr), COCOS_ERR_INVALID,
                  "center_l2norm_fwd_planes: channel-major planes come as a hi/lo pair");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && plane_scale > 0.f, COCOS_ERR_INVALID,
                  "center_l2norm_fwd_planes: bad dims B=%d N=%d", B, N);
    COCOS_REQUIRE(K == 256 && N % 4 == 0 && (center_over_channels == 1 || center_over_channels == 2), COCOS_ERR_UNSUPPORTED,
                  "center_l2norm_fwd_planes: needs K == 256, N %% 4 == 0, centring over channels or none (K=%d N=%d mode=%d)",
                  K, N, center_over_channels);
    for (const void* p : {(const void*)x, (const void*)norm, (const void*)pos_hi, (const void*)pos_lo})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "center_l2norm_fwd_planes: pointers must be 16-byte aligned");
    COCOS_REQUIRE(!chan_hi || ((reinterpret_cast<uintptr_t>(chan_hi) | reinterpret_cast<uintptr_t>(chan_lo)) & 7u) == 0,
                  COCOS_ERR_INVALID, "center_l2norm_fwd_planes: channel-major planes must be 8-byte aligned");
    hipLaunchKernelGGL(center_l2norm_fwd_planes_kernel, dim3((N + 63) / 64, B), dim3(256), 0, as_stream(stream), x, norm,
                       static_cast<_Float16*>(pos_hi), static_cast<_Float16*>(pos_lo), static_cast<_Float16*>(chan_hi),
                       static_cast<_Float16*>(chan_lo), N, eps, center_over_channels == 1, plane_scale);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// ... and its backward: y comes from the channel-major planes; *dx_amax_inout (nullable) as in cocos_center_l2norm_bwd_amax.
extern "C" int cocos_center_l2norm_bwd_planes(const void* chan_hi, const void* chan_lo, const float* norm, const float* dy,
                                              float* dx, int B, int K, int N, int center_over_channels, float eps,
                                              float plane_scale, float* dx_amax_inout, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(chan_hi && chan_lo && norm && dy && dx, COCOS_ERR_INVALID, "center_l2norm_bwd_planes: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && plane_scale > 0.f, COCOS_ERR_INVALID,
                  "center_l2norm_bwd_planes: bad dims B=%d N=%d", B, N);
    COCOS_REQUIRE(K == 256 && N % 4 == 0 && (center_over_channels == 1 || center_over_channels == 2), COCOS_ERR_UNSUPPORTED,
                  "center_l2norm_bwd_planes: needs K == 256, N %% 4 == 0, centring over channels or none (K=%d N=%d mode=%d)",
                  K, N, center_over_channels);
    COCOS_REQUIRE(aligned16(dy) && aligned16(dx) && aligned16(norm), COCOS_ERR_INVALID,
                  "center_l2norm_bwd_planes: dy, dx, norm must be 16-byte aligned");
    hipLaunchKernelGGL(center_l2norm_bwd_planes_kernel, dim3((N + 63) / 64, B), dim3(256), 0, as_stream(stream),
                       static_cast<const _Float16*>(chan_hi), static_cast<const _Float16*>(chan_lo), norm, dy, dx, N, eps,
                       center_over_channels == 1, 1.0f / plane_scale, reinterpret_cast<unsigned*>(dx_amax_inout));
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
