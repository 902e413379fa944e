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
                                                             int N, float eps, float* __restrict__ amax_part /* nullable: [planes] max|out0| of this plane */) {
    __shared__ float red[4];
    __shared__ float redm[4];
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

    // max|out0| as a by-product (round 6): the convolution / projection that reads this tensor next splits it into f16 planes scaled by
    // it.  One value per plane, reduced by amax_finish_kernel in the same entry point: a same-address atomicMax per wave (13 k of them
    // at 8 x 407 planes) serialises at ~10 ns each on the memory side — it made the forward 5x slower (0.033 -> 0.165 ms).
    float vmax = 0.f;
    auto put_amax = [&]() {
        if (amax_part) {
            vmax = wave_max_dpp(vmax);
            if ((tid & 63) == 0) redm[tid >> 6] = vmax;
            __syncthreads();
            if (tid == 0) amax_part[blockIdx.x] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        }
    };
    if (!BWD) {
        auto emit = [&](int u, int e, int i) {
            float z = xn_at(u, e, i);
            if (res) z += res[base + i];
            z = z > 0.f ? z : a * z;
            vmax = fmaxf(vmax, fabsf(z));
            out0[base + i] = z;
        };
        if (REG) {
#pragma unroll
            for (int u = 0; u < VPT; ++u) {
                const int q = u * 256 + tid;
                if (vec && q < n4) {
                    f32x4 z = v[u] * r;
                    if (res) z += *reinterpret_cast<const f32x4*>(res + base + (size_t)q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        z[e] = z[e] > 0.f ? z[e] : a * z[e];
                        vmax = fmaxf(vmax, fabsf(z[e]));
                    }
                    *reinterpret_cast<f32x4*>(out0 + base + (size_t)q * 4) = z;
                } else if (!vec) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (q * 4 + e < N) emit(u, e, q * 4 + e);
                }
            }
        } else {
            for (int i = tid; i < N; i += 256) emit(0, 0, i);
        }
        put_amax();
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
                    for (int e = 0; e < 4; ++e) {
                        o[e] = r * (d[u][e] - m1 - (v[u][e] * r) * m2);
                        vmax = fmaxf(vmax, fabsf(o[e]));
                    }
                    *reinterpret_cast<f32x4*>(out0 + base + (size_t)q * 4) = o;
                }
            }
        }
        put_amax();
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
        if (out0) {
            const float o = r * (dz - m1 - xn * m2);
            vmax = fmaxf(vmax, fabsf(o));
            out0[base + i] = o;
        }
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
    put_amax();
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

// cells[j] = max(cells[j], max part[j][0..n)) for j = blockIdx.x: the per-workgroup maxima of a producer kernel -> the caller's cell
__global__ __launch_bounds__(256) void amax_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ cells) {
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

static int inp_fwd_impl(const float* x, const float* residual, const float* prelu_weight, float* y, float* y_amax_dev, float* part, int planes, int N,
                        float eps, cocos_stream_t stream, const char* who) {
    using namespace cocos;
    COCOS_REQUIRE(x && prelu_weight && y, COCOS_ERR_INVALID, "%s: null pointer", who);
    COCOS_REQUIRE(planes >= 1 && N >= 1, COCOS_ERR_INVALID, "%s: bad dims planes=%d N=%d", who, planes, N);
    hipStream_t s = as_stream(stream);
    float* cell = part;
    const bool reg = N <= 256 * 4 * INP_VPT && (N % 4 != 0 || (aligned16(x) && aligned16(y) && (!residual || aligned16(residual))));
    if (reg && N <= 256 * 4 * 4) hipLaunchKernelGGL((instnorm_prelu_kernel<false, true, 4>), dim3(planes), dim3(256), 0, s, x, residual, nullptr, prelu_weight, y, nullptr, nullptr, N, eps, cell);
    else if (reg) hipLaunchKernelGGL((instnorm_prelu_kernel<false, true>), dim3(planes), dim3(256), 0, s, x, residual, nullptr, prelu_weight, y, nullptr, nullptr, N, eps, cell);
    else     hipLaunchKernelGGL((instnorm_prelu_kernel<false, false>), dim3(planes), dim3(256), 0, s, x, residual, nullptr, prelu_weight, y, nullptr, nullptr, N, eps, cell);
    COCOS_HIP_CHECK(hipGetLastError());
    if (part) {
        hipLaunchKernelGGL(amax_finish_kernel, dim3(1), dim3(256), 0, s, part, planes, y_amax_dev);
        COCOS_HIP_CHECK(hipGetLastError());
    }
    return COCOS_OK;
}

extern "C" int cocos_instnorm_prelu_fwd(const float* x, const float* residual, const float* prelu_weight, float* y,
                                        int planes, int N, float eps, cocos_stream_t stream) {
    return inp_fwd_impl(x, residual, prelu_weight, y, nullptr, nullptr, planes, N, eps, stream, "instnorm_prelu_fwd");
}

// ... also leaving max|y| in the caller's cell: *y_amax_inout = max(*y_amax_inout, max|y|) (a cell holding a finite value >= 0) — the
// scale source of the f16 split of the convolution / projection that reads y next (correspondence.py:21,:25, :272, :282).
extern "C" int cocos_instnorm_prelu_fwd_amax(const float* x, const float* residual, const float* prelu_weight, float* y,
                                             float* y_amax_inout_dev, float* amax_partials /* [planes] workspace */, int planes, int N,
                                             float eps, cocos_stream_t stream) {
    COCOS_REQUIRE(y_amax_inout_dev && amax_partials, COCOS_ERR_INVALID, "instnorm_prelu_fwd_amax: null pointer");
    return inp_fwd_impl(x, residual, prelu_weight, y, y_amax_inout_dev, amax_partials, planes, N, eps, stream, "instnorm_prelu_fwd_amax");
}

static int inp_bwd_impl(const float* x, const float* residual, const float* prelu_weight, const float* dy, float* dx, float* dresidual,
                        float* da_partials_f32, double* da_partials_f64, float* da_out, float* dx_amax_dev, float* part, int planes, int N,
                        float eps, cocos_stream_t stream, const char* who) {
    using namespace cocos;
    COCOS_REQUIRE(x && prelu_weight && dy, COCOS_ERR_INVALID, "%s: null pointer", who);
    COCOS_REQUIRE(planes >= 1 && N >= 1, COCOS_ERR_INVALID, "%s: bad dims planes=%d N=%d", who, planes, N);
    COCOS_REQUIRE((da_partials_f64 == nullptr) == (da_out == nullptr) && !(da_partials_f64 && da_partials_f32), COCOS_ERR_INVALID,
                  "%s: the fp64 partials and their sum go together (and exclude the fp32 partials)", who);
    COCOS_REQUIRE((reinterpret_cast<uintptr_t>(da_partials_f64) & 7u) == 0, COCOS_ERR_INVALID, "%s: workspace must be 8-byte aligned", who);
    COCOS_REQUIRE(!dx_amax_dev || dx, COCOS_ERR_INVALID, "%s: max|dx| without dx", who);
    hipStream_t s = as_stream(stream);
    float* cell = part;
    const bool reg = N <= 256 * 4 * INP_VPT &&
                     (N % 4 != 0 || (aligned16(x) && aligned16(dy) && (!residual || aligned16(residual)) && (!dx || aligned16(dx)) &&
                                     (!dresidual || aligned16(dresidual))));
    if (da_partials_f64) {
        void* dap = da_partials_f64;
        if (reg && N <= 256 * 4 * 4) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true, 4, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps, cell);
        else if (reg) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true, INP_VPT, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps, cell);
        else     hipLaunchKernelGGL((instnorm_prelu_kernel<true, false, INP_VPT, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps, cell);
        COCOS_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(sum_f64_to_f32_kernel, dim3(1), dim3(256), 0, s, da_partials_f64, planes, da_out);
    } else {
        void* dap = da_partials_f32;
        if (reg && N <= 256 * 4 * 4) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true, 4>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps, cell);
        else if (reg) hipLaunchKernelGGL((instnorm_prelu_kernel<true, true>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps, cell);
        else     hipLaunchKernelGGL((instnorm_prelu_kernel<true, false>), dim3(planes), dim3(256), 0, s, x, residual, dy, prelu_weight, dx, dresidual, dap, N, eps, cell);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    if (part) {
        hipLaunchKernelGGL(amax_finish_kernel, dim3(1), dim3(256), 0, s, part, planes, dx_amax_dev);
        COCOS_HIP_CHECK(hipGetLastError());
    }
    return COCOS_OK;
}

extern "C" int cocos_instnorm_prelu_bwd(const float* x, const float* residual, const float* prelu_weight, const float* dy,
                                        float* dx, float* dresidual, float* da_partials /* [planes] */, int planes,
                                        int N, float eps, cocos_stream_t stream) {
    return inp_bwd_impl(x, residual, prelu_weight, dy, dx, dresidual, da_partials, nullptr, nullptr, nullptr, nullptr, planes, N, eps, stream,
                        "instnorm_prelu_bwd");
}

// The same backward with the PReLU weight's gradient accumulated in fp64 end to end: da_partials_f64 = workspace of `planes`
// doubles, *da_out = sum over every element of the layer with z <= 0 of dy * z (rounded to fp32 once, at the end).
extern "C" int cocos_instnorm_prelu_bwd_f64(const float* x, const float* residual, const float* prelu_weight, const float* dy,
                                            float* dx, float* dresidual, double* da_partials_f64, float* da_out, int planes, int N,
                                            float eps, cocos_stream_t stream) {
    COCOS_REQUIRE(da_partials_f64 && da_out, COCOS_ERR_INVALID, "instnorm_prelu_bwd_f64: null pointer");
    return inp_bwd_impl(x, residual, prelu_weight, dy, dx, dresidual, nullptr, da_partials_f64, da_out, nullptr, nullptr, planes, N, eps, stream,
                        "instnorm_prelu_bwd_f64");
}

// Either backward (da_partials_f64 / da_out both null: no weight gradient) also leaving max|dx| in the caller's cell — dx is the output
// gradient of the convolution in front of the norm (correspondence.py:20-21, :24-25), whose backward splits it next.
extern "C" int cocos_instnorm_prelu_bwd_amax(const float* x, const float* residual, const float* prelu_weight, const float* dy,
                                             float* dx, float* dresidual, double* da_partials_f64, float* da_out, float* dx_amax_inout_dev,
                                             float* amax_partials /* [planes] workspace */, int planes, int N, float eps,
                                             cocos_stream_t stream) {
    COCOS_REQUIRE(dx && dx_amax_inout_dev && amax_partials, COCOS_ERR_INVALID, "instnorm_prelu_bwd_amax: null pointer");
    return inp_bwd_impl(x, residual, prelu_weight, dy, dx, dresidual, nullptr, da_partials_f64, da_out, dx_amax_inout_dev, amax_partials, planes,
                        N, eps, stream, "instnorm_prelu_bwd_amax");
}
