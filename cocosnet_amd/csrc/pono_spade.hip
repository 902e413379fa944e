// K9: PONO + SPADE modulation + LeakyReLU in one pass, and its backward (gfx950).   [SURVEY §8(f) rank 1]
//
// Replaces, for `--PONO` networks, the chain
//     PositionalNorm2d      normalization.py:63-68    xn = (x - mean_C) / sqrt(var_C(unbiased) + 1e-5)
//     SPADE.forward         normalization.py:148-151  z  = xn * (1 + gamma) + beta
//     actvn (LeakyReLU 0.2) architecture.py:88-95     y  = z > 0 ? z : slope * z      (slope 1: no activation)
// which PyTorch runs as ~9 elementwise/reduction launches (each a full HBM pass over [B,C,H,W], plus the
// autograd copies of xn, 1+gamma, z).  Here: forward = 3 reads + 1 write, backward = 4 reads + 3 writes,
// nothing saved between them except the inputs (statistics are recomputed from x, which the backward
// reads anyway).  Purely HBM-bound: 16 B/element forward, 28 B/element backward.
//
// Layout: x, gamma, beta, y are [B, C, N] (N = H*W positions, contiguous).  The normalisation runs over
// C at fixed position, i.e. across rows that are N floats apart, so a workgroup owns 32 consecutive
// positions (8 float4 "quads") and splits the channels 32 ways: thread (pq = tid & 7, cg = tid >> 3)
// keeps the float4 of its quad for channels cg, cg+32, ... in registers — x crosses HBM once, every
// access is a 128-byte row segment, and the per-position sums over channels go through one 32x32 LDS
// reduction per statistic.  C % 32 != 0, C > 1024 or N % 4 != 0 take the generic kernel (one lane per
// position, three sweeps over the column; the re-reads hit L2).
#include "common.h"

namespace cocos {

constexpr int PS_POS = 32;    // positions per workgroup
constexpr int PS_CG = 32;     // channel groups per workgroup

// Sum NQ float4 quantities over the 32 channel groups; every thread gets the totals of its quad.
template <int NQ>
__device__ __forceinline__ void ps_reduce(f32x4 (&v)[NQ], float* red /*[NQ][32][32]*/, int cg, int pq) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(red + (q * PS_CG + cg) * PS_POS + pq * 4) = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < PS_CG; ++g)
            acc += *reinterpret_cast<const f32x4*>(red + (q * PS_CG + g) * PS_POS + pq * 4);
        v[q] = acc;
    }
}

// mean over C and 1/sqrt(unbiased var + eps) of the register-resident column block; x becomes xn
template <int NI>
__device__ __forceinline__ void ps_normalise(f32x4 (&x)[NI], float* red, int cg, int pq, int C, float eps) {
    f32x4 s[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) s[0] += x[i];
    ps_reduce<1>(s, red, cg, pq);
    const f32x4 mean = s[0] * (1.0f / (float)C);
    f32x4 ss[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        x[i] -= mean;
        ss[0] += x[i] * x[i];
    }
    ps_reduce<1>(ss, red, cg, pq);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = 1.0f / sqrtf(ss[0][e] / (float)(C - 1) + eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) x[i] *= r;
    // r is needed again by the backward: leave it where the caller can find it
    *reinterpret_cast<f32x4*>(red + 2 * PS_CG * PS_POS + pq * 4) = r;   // same value from all 32 groups
    __syncthreads();
}

template <int NI, bool BWD>   // NI = C / 32
__global__ __launch_bounds__(256) void pono_spade_reg_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ dout,
                                                             float* __restrict__ out0,    // y | dx
                                                             float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int C, int N,
                                                             float eps, float slope,
                                                             float* __restrict__ amax /* nullable: per-workgroup maxima: forward [nwg] max|y|; backward [2][nwg] max|dgamma|, max|dbeta| */) {
    __shared__ __attribute__((aligned(16))) float red[2 * PS_CG * PS_POS + PS_POS];
    __shared__ float redm[2][4];
    const int tid = threadIdx.x, pq = tid & 7, cg = tid >> 3;
    const int b = blockIdx.y, n = blockIdx.x * PS_POS + pq * 4;
    const size_t sample = (size_t)b * C * N;
    const size_t bytes = (size_t)C * N * 4;
    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(x + sample, bytes);
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(gamma + sample, bytes);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(beta + sample, bytes);
    const __amdgpu_buffer_rsrc_t d_rs = make_rsrc((BWD ? dout : x) + sample, bytes);
    const bool ok = n < N;   // N % 4 == 0: a float4 is entirely in or out

    f32x4 xn[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
        xn[i] = buf_load4(x_rs, ok ? (unsigned)((cg + PS_CG * i) * N + n) * 4u : kBufOob);
    ps_normalise<NI>(xn, red, cg, pq, C, eps);

    // max|.| of the tensors the next convolutions split (round 6): one value per workgroup (reduced by amax_finish_kernel in the same
    // entry point — same-address atomics from every wave serialise on the memory side); every lane stays to the end
    const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
    auto put_amax = [&](int idx, float v) {
        v = wave_max_dpp(v);
        if ((tid & 63) == 0) redm[idx][tid >> 6] = v;
    };
    auto flush_amax = [&](int n) {      // after the put_amax calls of this workgroup (uniform)
        __syncthreads();
        if (tid < n) amax[(size_t)tid * nwg + wg] = fmaxf(fmaxf(redm[tid][0], redm[tid][1]), fmaxf(redm[tid][2], redm[tid][3]));
    };
    if (!BWD) {
        float* yb = out0 + sample;
        float vm = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned off = ok ? (unsigned)((cg + PS_CG * i) * N + n) * 4u : kBufOob;
            const f32x4 g = buf_load4(g_rs, off), bt = buf_load4(b_rs, off);
            f32x4 z = xn[i] * (1.0f + g) + bt;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = z[e] > 0.f ? z[e] : z[e] * slope;
            if (ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vm = fmaxf(vm, fabsf(z[e]));
                *reinterpret_cast<f32x4*>(yb + (size_t)(cg + PS_CG * i) * N + n) = z;
            }
        }
        if (amax) { put_amax(0, vm); flush_amax(1); }
    } else {
        const f32x4 r = *reinterpret_cast<const f32x4*>(red + 2 * PS_CG * PS_POS + pq * 4);
        f32x4 dxn[NI];
        f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // sum dxn, sum dxn*xn
        float vmg = 0.f, vmb = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned off = ok ? (unsigned)((cg + PS_CG * i) * N + n) * 4u : kBufOob;
            const f32x4 g = buf_load4(g_rs, off), bt = buf_load4(b_rs, off), dy = buf_load4(d_rs, off);
            const f32x4 z = xn[i] * (1.0f + g) + bt;
            f32x4 dz;
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = z[e] > 0.f ? dy[e] : dy[e] * slope;
            if (ok) {
                const size_t o = sample + (size_t)(cg + PS_CG * i) * N + n;
                const f32x4 dgv = dz * xn[i];
                if (dgamma) *reinterpret_cast<f32x4*>(dgamma + o) = dgv;
                if (dbeta) *reinterpret_cast<f32x4*>(dbeta + o) = dz;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vmg = fmaxf(vmg, fabsf(dgv[e]));
                    vmb = fmaxf(vmb, fabsf(dz[e]));
                }
            }
            dxn[i] = dz * (1.0f + g);
            s[0] += dxn[i];
            s[1] += dxn[i] * xn[i];
        }
        if (amax) {
            put_amax(0, dgamma ? vmg : 0.f);
            put_amax(1, dbeta ? vmb : 0.f);
            flush_amax(2);
        }
        if (!out0) return;                       // uniform: kernel argument
        ps_reduce<2>(s, red, cg, pq);
        if (!ok) return;
        const f32x4 m1 = s[0] * (1.0f / (float)C), m2 = s[1] * (1.0f / (float)(C - 1));
        float* dxb = out0 + sample;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<f32x4*>(dxb + (size_t)(cg + PS_CG * i) * N + n) = r * (dxn[i] - m1 - xn[i] * m2);
    }
}

// Generic shapes: one lane per position, sweeps over the channel column (coalesced across lanes).
template <bool BWD>
__global__ __launch_bounds__(256) void pono_spade_generic_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 const float* __restrict__ dout,
                                                                 float* __restrict__ out0,
                                                                 float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int C, int N,
                                                                 float eps, float slope) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const size_t base = (size_t)blockIdx.y * C * N + n;
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += x[base + (size_t)c * N];
    const float mean = sum / (float)C;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = x[base + (size_t)c * N] - mean;
        ss += d * d;
    }
    const float r = 1.0f / sqrtf(ss / (float)(C - 1) + eps);
    if (!BWD) {
        for (int c = 0; c < C; ++c) {
            const size_t o = base + (size_t)c * N;
            const float z = (x[o] - mean) * r * (1.0f + gamma[o]) + beta[o];
            out0[o] = z > 0.f ? z : z * slope;
        }
        return;
    }
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * N;
        const float xn = (x[o] - mean) * r, g1 = 1.0f + gamma[o];
        const float z = xn * g1 + beta[o];
        const float dz = z > 0.f ? dout[o] : dout[o] * slope;
        if (dgamma) dgamma[o] = dz * xn;
        if (dbeta) dbeta[o] = dz;
        s1 += dz * g1;
        s2 += dz * g1 * xn;
    }
    if (!out0) return;
    const float m1 = s1 / (float)C, m2 = s2 / (float)(C - 1);
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * N;
        const float xn = (x[o] - mean) * r, g1 = 1.0f + gamma[o];
        const float z = xn * g1 + beta[o];
        const float dz = z > 0.f ? dout[o] : dout[o] * slope;
        out0[o] = r * (dz * g1 - m1 - xn * m2);
    }
}

static bool pono_spade_reg_ok(int C, int N) {
    if (!(C % PS_CG == 0 && C / PS_CG <= 32 && N % 4 == 0 && (size_t)C * N * 4 < 0x7fffffffull)) return false;
    switch (C / PS_CG) {
        case 1: case 2: case 3: case 4: case 6: case 8: case 12: case 16: case 24: case 32: return true;
        default: return false;
    }
}

// amax_part (nullable): the register kernel leaves its per-workgroup maxima there and *wrote_part = true
template <bool BWD>
static int pono_spade_launch(const float* x, const float* gamma, const float* beta, const float* dout,
                             float* out0, float* dgamma, float* dbeta, int B, int C, int N, float eps,
                             float slope, hipStream_t s, float* amax = nullptr, bool* wrote_part = nullptr) {
    if (wrote_part) *wrote_part = false;
    const bool reg = pono_spade_reg_ok(C, N) &&
                     aligned16(x) && aligned16(gamma) && aligned16(beta) && (!BWD || aligned16(dout)) &&
                     (!out0 || aligned16(out0)) && (!dgamma || aligned16(dgamma)) && (!dbeta || aligned16(dbeta));
    if (reg) {
        if (wrote_part) *wrote_part = amax != nullptr;
        const dim3 grid((N + PS_POS - 1) / PS_POS, B);
#define COCOS_PS(NI)                                                                                      \
    case NI:                                                                                              \
        hipLaunchKernelGGL((pono_spade_reg_kernel<NI, BWD>), grid, dim3(256), 0, s, x, gamma, beta, dout, \
                           out0, dgamma, dbeta, C, N, eps, slope, amax);                                  \
        break;
        switch (C / PS_CG) {
            COCOS_PS(1) COCOS_PS(2) COCOS_PS(3) COCOS_PS(4) COCOS_PS(6) COCOS_PS(8) COCOS_PS(12) COCOS_PS(16)
            COCOS_PS(24) COCOS_PS(32)
            default:
                hipLaunchKernelGGL((pono_spade_generic_kernel<BWD>), dim3((N + 255) / 256, B), dim3(256), 0, s,
                                   x, gamma, beta, dout, out0, dgamma, dbeta, C, N, eps, slope);
        }
#undef COCOS_PS
    } else {
        hipLaunchKernelGGL((pono_spade_generic_kernel<BWD>), dim3((N + 255) / 256, B), dim3(256), 0, s, x,
                           gamma, beta, dout, out0, dgamma, dbeta, C, N, eps, slope);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// cells[j] = max(cells[j], max part[j][0..n)), j = blockIdx.x
__global__ __launch_bounds__(256) void pono_amax_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ cells) {
    __shared__ float redm[4];
    const float* p = part + (size_t)blockIdx.x * n;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, p[i]);
    m = wave_max_dpp(m);
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        if (m > cells[blockIdx.x] && m < INFINITY) cells[blockIdx.x] = m;
    }
}

}  // namespace cocos

extern "C" int cocos_pono_spade_fwd(const float* x, const float* gamma, const float* beta, float* y, int B,
                                    int C, int N, float eps, float slope, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && gamma && beta && y, COCOS_ERR_INVALID, "pono_spade_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && C >= 2 && N >= 1, COCOS_ERR_INVALID,
                  "pono_spade_fwd: bad dims B=%d C=%d N=%d (unbiased variance needs C >= 2)", B, C, N);
    return pono_spade_launch<false>(x, gamma, beta, nullptr, y, nullptr, nullptr, B, C, N, eps, slope,
                                    as_stream(stream));
}

extern "C" int cocos_pono_spade_bwd(const float* x, const float* gamma, const float* beta, const float* dy,
                                    float* dx, float* dgamma, float* dbeta, int B, int C, int N, float eps,
                                    float slope, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && gamma && beta && dy, COCOS_ERR_INVALID, "pono_spade_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && C >= 2 && N >= 1, COCOS_ERR_INVALID,
                  "pono_spade_bwd: bad dims B=%d C=%d N=%d (unbiased variance needs C >= 2)", B, C, N);
    return pono_spade_launch<true>(x, gamma, beta, dy, dx, dgamma, dbeta, B, C, N, eps, slope,
                                   as_stream(stream));
}

// Round 6: the same passes also leaving max|.| of what the next convolutions split into f16 planes (*cell = max(*cell, max|.|); cells
// holding finite values >= 0): the forward's y (the input of conv_0 / conv_1 / conv_s, architecture.py:88-95), the backward's
// dgamma and dbeta (the output gradients of SPADE's mlp_gamma / mlp_beta, normalization.py:121-127): amax2 = [max|dgamma|, max|dbeta|].
// amax_partials: cocos_pono_spade_amax_partials() floats (forward), twice that (backward) — one maximum per workgroup, reduced by a
// small kernel of the same call; shapes on the generic kernel take cocos_absmax_accumulate passes instead (same result).
extern "C" int cocos_pono_spade_amax_partials(int B, int C, int N) {
    if (B < 1 || C < 2 || N < 1) return 0;
    return ((N + cocos::PS_POS - 1) / cocos::PS_POS) * B;
}

extern "C" int cocos_pono_spade_fwd_amax(const float* x, const float* gamma, const float* beta, float* y, float* y_amax_inout_dev,
                                         float* amax_partials, int B, int C, int N, float eps, float slope, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && gamma && beta && y && y_amax_inout_dev && amax_partials, COCOS_ERR_INVALID, "pono_spade_fwd_amax: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && C >= 2 && N >= 1, COCOS_ERR_INVALID,
                  "pono_spade_fwd_amax: bad dims B=%d C=%d N=%d (unbiased variance needs C >= 2)", B, C, N);
    bool wrote = false;
    const int rc = pono_spade_launch<false>(x, gamma, beta, nullptr, y, nullptr, nullptr, B, C, N, eps, slope, as_stream(stream),
                                            amax_partials, &wrote);
    if (rc != COCOS_OK) return rc;
    if (!wrote) return cocos_absmax_accumulate(y, (long long)B * C * N, y_amax_inout_dev, stream);
    hipLaunchKernelGGL(pono_amax_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), amax_partials,
                       cocos_pono_spade_amax_partials(B, C, N), y_amax_inout_dev);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_pono_spade_bwd_amax(const float* x, const float* gamma, const float* beta, const float* dy, float* dx, float* dgamma,
                                         float* dbeta, float* amax2_inout_dev, float* amax_partials, int B, int C, int N, float eps,
                                         float slope, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && gamma && beta && dy && amax2_inout_dev && amax_partials, COCOS_ERR_INVALID, "pono_spade_bwd_amax: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && C >= 2 && N >= 1, COCOS_ERR_INVALID,
                  "pono_spade_bwd_amax: bad dims B=%d C=%d N=%d (unbiased variance needs C >= 2)", B, C, N);
    bool wrote = false;
    const int rc = pono_spade_launch<true>(x, gamma, beta, dy, dx, dgamma, dbeta, B, C, N, eps, slope, as_stream(stream), amax_partials,
                                           &wrote);
    if (rc != COCOS_OK) return rc;
    if (!wrote) {
        const long long n = (long long)B * C * N;
        int r2 = dgamma ? cocos_absmax_accumulate(dgamma, n, amax2_inout_dev, stream) : COCOS_OK;
        if (r2 == COCOS_OK && dbeta) r2 = cocos_absmax_accumulate(dbeta, n, amax2_inout_dev + 1, stream);
        return r2;
    }
    hipLaunchKernelGGL(pono_amax_finish_kernel, dim3(2), dim3(256), 0, as_stream(stream), amax_partials,
                       cocos_pono_spade_amax_partials(B, C, N), amax2_inout_dev);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
