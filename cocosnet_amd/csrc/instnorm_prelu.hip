// K13: InstanceNorm (+ residual) + PReLU in one pass, and its backward (gfx950).      [SURVEY §8(f) rank 4]
//
// Replaces, in the four ResidualBlocks in front of theta/phi (correspondence.py:13-36, :175-179):
//     y = prelu(InstanceNorm2d(x))              :29   (bn1 + the shared PReLU)
//     y = prelu(InstanceNorm2d(x) + residual)   :31-33 (bn2 + skip + the same PReLU)
// InstanceNorm2d(affine=False, eps=1e-5): per (sample, channel) plane, biased variance over the h*w positions.
// PyTorch runs this as a MIOpen/native instance-norm kernel + add + prelu (+ their autograd: 3 saved tensors);
// here one workgroup owns one plane, the plane lives in registers (x crosses HBM once), statistics are
// recomputed in the backward from x.  HBM-bound: 8 B/element forward (12 with the residual), 16-20 backward.
//   fwd: xn = (x - mean) / sqrt(var + eps);  z = xn (+ res);  y = z > 0 ? z : a*z          (a = the PReLU weight)
//   bwd: dz = dy * (z > 0 ? 1 : a);  dres = dz;  dx = r*(dz - mean(dz) - xn*mean(dz*xn));  da = sum_{z<0} dy*z
// Planes of up to 16384 positions take the register path (64 floats per thread); larger ones the streaming path.
#include <type_traits>

#include "common.h"

namespace cocos {

constexpr int INP_VPT = 16;   // float4 per thread in the register path (256 threads x 64 floats = 16384); planes up to 64 x 64 use 4
                              // (the arrays are sized by the template: with 16 the backward took 308 registers — one workgroup per CU,
                              //  a chain of load / reduce / reduce / store per plane with nothing to overlap it: 1.2 TB/s)

__device__ __forceinline__ float inp_block_sum(float v, float* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ double inp_block_sum_f64(double v, double* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// DA64 (round 6): the PReLU weight's gradient da = sum_{z <= 0} dy z — ONE number summed over every element of the layer, whose
// terms cancel (|sum| ~ sqrt(n) of sum |term| at n = 1.3e7) — is accumulated in fp64 from the products on: per element, per plane
// and (cocos_instnorm_prelu_bwd_f64) over the planes.  In fp32 it was 90x further from fp64 than the framework's on one probe.
template <bool BWD, bool REG, int VPT = INP_VPT, bool DA64 = false>
__global__ __launch_bounds__(256) void instnorm_prelu_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ dy, const float* __restrict__ aw,
                                                             float* __restrict__ out0 /* y | dx */,
                                                             float* __restrict__ dres, void* __restrict__ da_part,
                                                             int N, float eps) {
    __shared__ float red[4];
    __shared__ double red64[4];
    typedef typename std::conditional<DA64, double, float>::type da_t;
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * N;
    const float a = *aw;
    const float invn = 1.0f / (float)N;
    const bool vec = (N % 4 == 0);
    const int n4 = vec ? N / 4 : 0;

    // ---- statistics (two passes over the register-resident plane, or over memory when it does not fit) ----
    f32x4 v[REG ? VPT : 1];
    float s = 0.f;
    if (REG) {
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int q = u * 256 + tid;
            v[u] = (vec && q < n4) ? *reinterpret_cast<const f32x4*>(x + base + (size_t)q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (!vec)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (q * 4 + e < N) v[u][e] = x[base + q * 4 + e];
            s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        }
    } else {
        for (int i = tid; i < N; i += 256) s += x[base + i];
    }
    const float mean = inp_block_sum(s, red, tid) * invn;
    float ss = 0.f;
    if (REG) {
#pragma unroll
        for (int u = 0; u < VPT; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = (u * 256 + tid) * 4 + e < N;
                const float d = in ? v[u][e] - mean : 0.f;
                v[u][e] = d;
                ss += d * d;
            }
    } else {
        for (int i = tid; i < N; i += 256) { const float d = x[base + i] - mean; ss += d * d; }
    }
    const float r = 1.0f / sqrtf(inp_block_sum(ss, red, tid) * invn + eps);

    auto xn_at = [&](int u, int e, int i) { return REG ? v[u][e] * r : (x[base + i] - mean) * r; };

    if (!BWD) {
        auto emit = [&](int u, int e, int i) {
            float z = xn_at(u, e, i);
            if (res) z += res[base + i];
            out0[base + i] = z > 0.f ? z : a * z;
        };
        if (REG) {
#pragma unroll
            for (int u = 0; u < VPT; ++u) {
                const int q = u * 256 + tid;
                if (vec && q < n4) {
                    f32x4 z = v[u] * r;
                    if (res) z += *reinterpret_cast<const f32x4*>(res + base + (size_t)q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[e] = z[e] > 0.f ? z[e] : a * z[e];
                    *reinterpret_cast<f32x4*>(out0 + base + (size_t)q * 4) = z;
                } else if (!vec) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (q * 4 + e < N) emit(u, e, q * 4 + e);
                }
            }
        } else {
            for (int i = tid; i < N; i += 256) emit(0, 0, i);
        }
        return;
    }

    // ---- backward: pass 1 = sums of dz and dz*xn (+ da), pass 2 = dx --------------------------------------
    float s1 = 0.f, s2 = 0.f;
    da_t sa = 0;
    auto put_da = [&]() {
        if (DA64) {
            const double t = inp_block_sum_f64((double)sa, red64, tid);
            if (tid == 0 && da_part) static_cast<double*>(da_part)[blockIdx.x] = t;
        } else {
            const float t = inp_block_sum((float)sa, red, tid);
            if (tid == 0 && da_part) static_cast<float*>(da_part)[blockIdx.x] = t;
        }
    };
    auto dz_at = [&](int u, int e, int i, float& xn) {
        xn = xn_at(u, e, i);
        const float z = xn + (res ? res[base + i] : 0.f);
        const float g = dy[base + i];
        if (z <= 0.f) sa += (da_t)g * (da_t)z;
        return z > 0.f ? g : g * a;
    };
    if (REG && vec) {
        // whole 16-byte pieces: dy (and the residual) are read ONCE, dz stays in registers beside the centred plane, dx / dres
        // leave as 16-byte stores.  (The element-wise flavour below — 4-byte loads and stores at a 16-byte lane stride, dy read
        // twice — ran at 1.45 TB/s: 0.11 ms for the 8 x 407 planes of a ResidualBlock.)
        f32x4 d[VPT];
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int q = u * 256 + tid;
            d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < n4) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(dy + base + (size_t)q * 4);
                f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
                if (res) r4 = *reinterpret_cast<const f32x4*>(res + base + (size_t)q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xn = v[u][e] * r, z = xn + r4[e], g = g4[e];
                    if (z <= 0.f) sa += (da_t)g * (da_t)z;
                    const float dz = z > 0.f ? g : g * a;
                    d[u][e] = dz;
                    s1 += dz;
                    s2 += dz * xn;
                }
            }
        }
        const float m1 = inp_block_sum(s1, red, tid) * invn;
        const float m2 = inp_block_sum(s2, red, tid) * invn;
        put_da();
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int q = u * 256 + tid;
            if (q < n4) {
                if (dres) *reinterpret_cast<f32x4*>(dres + base + (size_t)q * 4) = d[u];
                if (out0) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = r * (d[u][e] - m1 - (v[u][e] * r) * m2);
                    *reinterpret_cast<f32x4*>(out0 + base + (size_t)q * 4) = o;
                }
            }
        }
        return;
    }
    // (the element-wise register path recomputes dz in pass 2 from dy — an L2 hit — instead of holding a second plane)
    if (REG) {
#pragma unroll
        for (int u = 0; u < VPT; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = (u * 256 + tid) * 4 + e;
                if (i < N) { float xn; const float dz = dz_at(u, e, i, xn); s1 += dz; s2 += dz * xn; }
            }
    } else {
        for (int i = tid; i < N; i += 256) { float xn; const float dz = dz_at(0, 0, i, xn); s1 += dz; s2 += dz * xn; }
    }
    const float m1 = inp_block_sum(s1, red, tid) * invn;
    const float m2 = inp_block_sum(s2, red, tid) * invn;
    put_da();
    float dummy = 0.f;
    auto fin = [&](int u, int e, int i) {
        float xn;
        const da_t sa_keep = sa;
        const float dz = dz_at(u, e, i, xn);
        sa = sa_keep;
        if (dres) dres[base + i] = dz;
        if (out0) out0[base + i] = r * (dz - m1 - xn * m2);
    };
    if (REG) {
#pragma unroll
        for (int u = 0; u < VPT; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = (u * 256 + tid) * 4 + e;
                if (i < N) fin(u, e, i);
            }
    } else {
        for (int i = tid; i < N; i += 256) fin(0, 0, i);
    }
    (void)dummy;
}

// *out = (float) sum of n doubles: one workgroup (n = planes of a layer: a few thousand)
__global__ __launch_bounds__(256) void sum_f64_to_f32_kernel(const double* __restrict__ p, int n, float* __restrict__ out) {
    __shared__ double red64[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += p[i];
    const double t = inp_block_sum_f64(v, red64, threadIdx.x);
    if (threadIdx.x == 0) *out = (float)t;
}

}  // namespace cocos

extern "C" int cocos_instnorm_prelu_fwd(const float* x, const float* residual, const float* prelu_weight, float* y,
                                        int planes, int N, float eps, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && prelu_weight && y, COCOS_ERR_INVALID, "instnorm_prelu_fwd: null pointer");
    COCOS_REQUIRE(planes >= 1 && N >= 1, COCOS_ERR_INVALID, "instnorm_prelu_fwd: bad dims planes=%d N=%d", planes, N);
    hipStream_t s = as_stream(stream);
    const bool reg = N <= 256 * 4 * INP_VPT && (N % 4 != 0 || (aligned16(x) && aligned16(y) && (!residual || aligned16(residual))));
    if (reg && N <= 256 * 4 * 4) hipLaunchKernelGGL((instnorm_prelu_kernel<false, true, 4>), dim3(planes), dim3(256), 0, s, x, residual, nullptr, prelu_weight, y, nullptr, nullptr, N, eps);
    else if (reg) hipLaunchKernelGGL((instnorm_prelu_kernel<false, true>), dim3(planes), dim3(256), 0, s, x, residual, nullptr, prelu_weight, y, nullptr, nullptr, N, eps);
    else     hipLaunchKernelGGL((instnorm_prelu_kernel<false, false>), dim3(planes), dim3(256), 0, s, x, residual, nullptr, prelu_weight, y, nullptr, nullptr, N, eps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_instnorm_prelu_bwd(const float* x, const float* residual, const float* prelu_weight, const float* dy,
                                        float* dx, float* dresidual, float* da_partials /* [planes] */, int planes,
                                        int N, float eps, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && prelu_weight && dy, COCOS_ERR_INVALID, "instnorm_prelu_bwd: null pointer");
    COCOS_REQUIRE(planes >= 1 && N >= 1, COCOS_ERR_INVALID, "instnorm_prelu_bwd: bad dims planes=%d N=%d", planes, N);
    hipStream_t s = as_stream(stream);
    const bool reg = N <= 256 * 4 * INP_VPT &&
                     (N % 4 != 0 || (aligned16(x) && aligned16(dy) && (!residual || aligned16(residual)) && (!dx || aligned16(dx)) &&
                                     (!dresidual || aligned16(dresidual))));
    if (reg && N <= 256 * 4 * 4) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true, 4>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, da_partials, N, eps);
    else if (reg) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, da_partials, N, eps);
    else     hipLaunchKernelGGL((instnorm_prelu_kernel<true, false>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, da_partials, N, eps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// The same backward with the PReLU weight's gradient accumulated in fp64 end to end: da_partials_f64 = workspace of `planes`
// doubles, *da_out = sum over every element of the layer with z <= 0 of dy * z (rounded to fp32 once, at the end).
extern "C" int cocos_instnorm_prelu_bwd_f64(const float* x, const float* residual, const float* prelu_weight, const float* dy,
                                            float* dx, float* dresidual, double* da_partials_f64, float* da_out, int planes, int N,
                                            float eps, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && prelu_weight && dy && da_partials_f64 && da_out, COCOS_ERR_INVALID, "instnorm_prelu_bwd_f64: null pointer");
    COCOS_REQUIRE(planes >= 1 && N >= 1, COCOS_ERR_INVALID, "instnorm_prelu_bwd_f64: bad dims planes=%d N=%d", planes, N);
    COCOS_REQUIRE((reinterpret_cast<uintptr_t>(da_partials_f64) & 7u) == 0, COCOS_ERR_INVALID, "instnorm_prelu_bwd_f64: workspace must be 8-byte aligned");
    hipStream_t s = as_stream(stream);
    const bool reg = N <= 256 * 4 * INP_VPT &&
                     (N % 4 != 0 || (aligned16(x) && aligned16(dy) && (!residual || aligned16(residual)) && (!dx || aligned16(dx)) &&
                                     (!dresidual || aligned16(dresidual))));
    void* dap = da_partials_f64;
    if (reg && N <= 256 * 4 * 4) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true, 4, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps);
    else if (reg) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true, INP_VPT, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps);
    else     hipLaunchKernelGGL((instnorm_prelu_kernel<true, false, INP_VPT, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps);
    COCOS_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(sum_f64_to_f32_kernel, dim3(1), dim3(256), 0, s, da_partials_f64, planes, da_out);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
