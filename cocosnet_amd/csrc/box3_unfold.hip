// match_kernel = 3 without the 9x blow-up (gfx950).
//
// The reference unfolds theta/phi into their zero-padded 3x3 neighbourhoods BEFORE centring and
// normalising (correspondence.py:276-280, :286-289), i.e. K = 256*9 = 2304 channels, 9x the
// memory and 9x the GEMM FLOPs of match_kernel 1.  With U_p the unfolded vector at position p:
//
//     <U_p, V_q> = sum_{d in 3x3} C_raw[p+d, q+d] * [p+d inside] * [q+d inside],   C_raw = theta_b^T phi_b
//
// (a 9-tap box filter ALONG THE DIAGONAL of the plain K = 256 correlation matrix), and centring +
// normalising after the unfold is a rank-1 correction and two per-position scales (PONO_C case):
//
//     f[p,q] = ( boxdiag(C_raw)[p,q] - K mu_p nu_q ) * a_p * b_q,
//     mu_p = mean(U_p), a_p = 1 / (||U_p - mu_p|| + eps)   (3x3 box sums of per-position channel sums /
//     sums of squares — tiny [B,1,h,w] maps the host computes), nu_q, b_q likewise for phi.
//
// Verified against the unfolded formulation to 6e-16 in fp64 (oracle test).  boxdiag
// is self-adjoint, so the backward uses the SAME kernel on G*a_p*b_q; the remaining gradients are
// row / column reductions of G and G*f (box3_bwd_reduce).  Both kernels are HBM/L2-bound streaming
// kernels.
#include "common.h"

namespace cocos {

// PRE = false (forward):  out = (boxdiag(in) - kc*mu_p*nu_q) * a_p * b_q * post
// PRE = true  (backward): out = boxdiag(in * a_p * b_q * post)      (weights taken at the SOURCE element)
//
// The operator walks DOWN THE DIAGONAL: the nine taps of an output are three runs of three consecutive
// elements of three diagonals (one per dy), and the output one step further down the diagonal (p+1, q+1) uses two
// of each run again.  A thread owns four neighbouring diagonals and walks BX_WR rows down them with a three-deep
// register window per dy: THREE 16-byte loads per four outputs instead of nine.  (The first version gave every
// output its nine taps straight from L2 — one workgroup per matrix row, XCD-aware so that the neighbour rows met
// in one L2 — and was L2-bandwidth-bound on those re-reads: 4.8 GB through L2 for a 0.5 GB matrix, 0.58 ms.)
// The validity of a tap belongs
// to the OUTPUT position ((py+dy, px+dx) and (qy+dy, qx+dx) inside the h x w grid), so masks are applied when a
// window element is used, not when it is loaded; whatever a load drags in from outside the matrix is masked too.
constexpr int BX_WR = 32;      // rows per workgroup walk

template <bool PRE>
__global__ __launch_bounds__(256) void box3_walk_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ mu, const float* __restrict__ nu,
                                                        const float* __restrict__ av, const float* __restrict__ bv,
                                                        int N, int h, int w, int nchunk, int nrb, float kc, float post,
                                                        float* __restrict__ out_amax) {
    __shared__ float amax_red[4];
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = vb % nchunk;
    const int rb = (vb / nchunk) % nrb;
    const int b = vb / (nchunk * nrb);
    const int p0 = rb * BX_WR;
    const int rows = min(BX_WR, N - p0);
    // diagonal of element e of this thread: q - (p - p0) = qs + e, qs in [-(BX_WR-1), N)
    const int qs = (chunk * 256 + (int)threadIdx.x) * 4 - (BX_WR - 1);
    if (qs >= N && !out_amax) return;
    float amax = 0.f;
    const __amdgpu_buffer_rsrc_t in_rs = make_rsrc(in + (size_t)b * N * N, (size_t)N * N * 4);
    const __amdgpu_buffer_rsrc_t out_rs = make_rsrc(out + (size_t)b * N * N, (size_t)N * N * 4);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(bv + (size_t)b * N, (size_t)N * 4);
    const __amdgpu_buffer_rsrc_t nu_rs = make_rsrc(nu + (size_t)b * N, (size_t)N * 4);
    const float* a_b = av + (size_t)b * N;
    const float* mu_b = mu + (size_t)b * N;

    // four consecutive floats starting at (possibly negative) float index `idx` of a buffer: elements in front of
    // the buffer read as 0, elements behind it too (descriptor); only the threads on the left edge take the slow path
    auto load4 = [&](__amdgpu_buffer_rsrc_t rs, long long idx) -> f32x4 {
        if (idx >= 0) return buf_load4(rs, (unsigned)(idx * 4));
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (idx + e >= 0) ? buf_load1(rs, (unsigned)((idx + e) * 4)) : 0.f;
        return v;
    };
    // one window element: row r, columns c .. c+3 (entries outside the matrix: zeros or garbage — both masked)
    auto load_diag = [&](int r, int c) -> f32x4 {
        const bool row_ok = (unsigned)r < (unsigned)N;
        f32x4 x = row_ok ? load4(in_rs, (long long)r * N + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (PRE) {   // weights of the SOURCE element
            const float as = row_ok ? a_b[r] * post : 0.f;
            x = x * load4(b_rs, c) * as;
        }
        return x;
    };
    // position of the first output of each element in the h x w grid (floor division: qs + e may be negative)
    int qx[4], qy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int q = qs + e;
        qy[e] = q >= 0 ? q / w : -((-q + w - 1) / w);
        qx[e] = q - qy[e] * w;
    }
    int py = p0 / w, px = p0 - py * w;

    // windows: win[dy][0..2] = diagonal elements at steps i-1, i, i+1 (rows p0+i+dy*w+{-1,0,1})
    // (a fourth, look-ahead element per diagonal was measured: slightly slower, more so with the PRE weights)
    f32x4 win[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int sh = (d - 1) * w;
        win[d][0] = load_diag(p0 + sh - 1, qs + sh - 1);
        win[d][1] = load_diag(p0 + sh, qs + sh);
        win[d][2] = load_diag(p0 + sh + 1, qs + sh + 1);
    }
    for (int i = 0; i < (qs < N ? rows : 0); ++i) {
        const int p = p0 + i, q0 = qs + i;
        f32x4 nxt[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {               // the elements the NEXT step needs, requested before the math
            const int sh = (d - 1) * w;
            nxt[d] = load_diag(p + sh + 2, q0 + sh + 2);
        }
        const bool rp[3] = {py > 0, true, py < h - 1};
        const bool cp[3] = {px > 0, true, px < w - 1};
        f32x4 acc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool rq[3] = {qy[e] > 0, true, qy[e] < h - 1};
            const bool cq[3] = {qx[e] > 0, true, qx[e] < w - 1};
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) t += (cp[k] && cq[k]) ? win[d][k][e] : 0.f;
                a += (rp[d] && rq[d]) ? t : 0.f;
            }
            acc[e] = a;
        }
        if (!PRE) {
            const float mp = mu_b[p] * kc, ap = a_b[p] * post;
            const f32x4 nq = load4(nu_rs, q0), bq = load4(b_rs, q0);
            acc = (acc - nq * mp) * bq * ap;
        }
        const unsigned ooff = (unsigned)(((long long)p * N + q0) * 4);
        if (q0 >= 0 && q0 + 3 < N) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), out_rs, (int)ooff, 0, 0);
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((unsigned)(q0 + e) < (unsigned)N) {
                    out[((size_t)b * N + p) * N + q0 + e] = acc[e];
                    amax = fmaxf(amax, fabsf(acc[e]));
                }
        }
        // one step down the diagonals
#pragma unroll
        for (int d = 0; d < 3; ++d) { win[d][0] = win[d][1]; win[d][1] = win[d][2]; win[d][2] = nxt[d]; }
        if (++px == w) { px = 0; ++py; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (++qx[e] == w) { qx[e] = 0; ++qy[e]; }
    }
    // max|out| as a by-product (the consumer of dc_raw, the K3 backward, scales its f16 split with it and would
    // otherwise read the whole matrix once more).  One value per workgroup into a partial array, no atomics: 5120
    // same-address atomics serialise into ~50 us at the memory side (measured: they cost what the saved pass cost);
    // the launcher folds the partials with one tiny max pass.
    if (out_amax) {
        const float wmax = wave_max_dpp(amax);
        if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = wmax;
        __syncthreads();
        if (threadIdx.x == 0)
            out_amax[blockIdx.x] = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
    }
}

// One pass over G (gradient w.r.t. the logits) and F (the forward's output) producing
//   row sums   r1[p] = sum_q G[p,q] * b_q * nu_q      r2[p] = sum_q G[p,q] * F[p,q]
//   col sums   c1[q] = sum_p G[p,q] * a_p * mu_p      c2[q] = sum_p G[p,q] * F[p,q]
// A workgroup takes BX_RB rows x 1024 columns (one float4 of columns per thread): column sums are
// thread-local over the rows, row sums are wave reductions parked in LDS (one barrier per
// workgroup, none per row).  Both come out as partials — rows [N][N/1024], columns [N/BX_RB][N] —
// and box3_finish_kernel adds them up: no atomics, deterministic.
constexpr int BX_RB = 16;

__global__ __launch_bounds__(256) void box3_bwd_reduce_kernel(
    const float* __restrict__ G, const float* __restrict__ F, const float* __restrict__ mu,
    const float* __restrict__ nu, const float* __restrict__ av, const float* __restrict__ bv,
    float* __restrict__ r1p, float* __restrict__ r2p, float* __restrict__ c1p, float* __restrict__ c2p,
    int N, int nchunk, int nrb) {
    __shared__ float red[BX_RB][4][2];
    const int b = blockIdx.z, rb = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = chunk * 1024 + tid * 4;
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(G + (size_t)b * N * N, (size_t)N * N * 4);
    const __amdgpu_buffer_rsrc_t f_rs = make_rsrc(F + (size_t)b * N * N, (size_t)N * N * 4);
    const float* mu_b = mu + (size_t)b * N;
    const float* a_b = av + (size_t)b * N;
    f32x4 wq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (q0 + e < N) wq[e] = bv[(size_t)b * N + q0 + e] * nu[(size_t)b * N + q0 + e];
    const bool vec = (N & 3) == 0;
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = 0; i < BX_RB; ++i) {
        const int p = rb * BX_RB + i;
        f32x4 g = {0.f, 0.f, 0.f, 0.f}, f = {0.f, 0.f, 0.f, 0.f};
        float amu = 0.f;
        if (p < N) {   // uniform
            amu = a_b[p] * mu_b[p];
            if (vec) {
                const unsigned off = q0 < N ? (unsigned)((size_t)p * N + q0) * 4u : kBufOob;
                g = buf_load4(g_rs, off);
                f = buf_load4(f_rs, off);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned off = q0 + e < N ? (unsigned)((size_t)p * N + q0 + e) * 4u : kBufOob;
                    g[e] = buf_load1(g_rs, off);
                    f[e] = buf_load1(f_rs, off);
                }
            }
        }
        c1 += g * amu;
        c2 += g * f;
        float s1 = g.x * wq.x + g.y * wq.y + g.z * wq.z + g.w * wq.w;
        float s2 = g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (lane == 0) { red[i][wave][0] = s1; red[i][wave][1] = s2; }
    }
    __syncthreads();
    if (tid < BX_RB) {
        const int p = rb * BX_RB + tid;
        if (p < N) {
            r1p[((size_t)b * N + p) * nchunk + chunk] = red[tid][0][0] + red[tid][1][0] + red[tid][2][0] + red[tid][3][0];
            r2p[((size_t)b * N + p) * nchunk + chunk] = red[tid][0][1] + red[tid][1][1] + red[tid][2][1] + red[tid][3][1];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (q0 + e < N) {
            c1p[((size_t)b * nrb + rb) * N + q0 + e] = c1[e];
            c2p[((size_t)b * nrb + rb) * N + q0 + e] = c2[e];
        }
}

// rows: r[p] = sum over column chunks; columns: c[q] = sum over row blocks
__global__ __launch_bounds__(256) void box3_finish_kernel(const float* __restrict__ r1p,
                                                          const float* __restrict__ r2p,
                                                          const float* __restrict__ c1p,
                                                          const float* __restrict__ c2p,
                                                          float* __restrict__ r1, float* __restrict__ r2,
                                                          float* __restrict__ c1, float* __restrict__ c2,
                                                          int N, int nchunk, int nrb) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nchunk; ++k) {
        s1 += r1p[((size_t)b * N + i) * nchunk + k];
        s2 += r2p[((size_t)b * N + i) * nchunk + k];
    }
    r1[(size_t)b * N + i] = s1;
    r2[(size_t)b * N + i] = s2;
    s1 = 0.f; s2 = 0.f;
    for (int r = 0; r < nrb; ++r) {
        s1 += c1p[((size_t)b * nrb + r) * N + i];
        s2 += c2p[((size_t)b * nrb + r) * N + i];
    }
    c1[(size_t)b * N + i] = s1;
    c2[(size_t)b * N + i] = s2;
}

}  // namespace cocos

extern "C" int cocos_box3_logits_fwd(const float* c_raw, const float* mu, const float* nu,
                                     const float* a, const float* b, float* f, int B, int h, int w,
                                     float k_unfolded, float scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(c_raw && mu && nu && a && b && f, COCOS_ERR_INVALID, "box3_logits_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && h >= 1 && w >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "box3_logits_fwd: bad dims B=%d h=%d w=%d", B, h, w);
    const long long N = (long long)h * w;
    COCOS_REQUIRE(N <= 65535 && N * N * 4 < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED,
                  "box3_logits_fwd: grid %dx%d too large (per-sample matrix must stay below 2 GiB)", h, w);
    const int nchunk = (int)((N + 1023) / 1024);
    COCOS_REQUIRE((long long)nchunk * N * B < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "box3_logits_fwd: grid too large");
    {
        const int nrb = (int)((N + BX_WR - 1) / BX_WR), nck = (int)((N + BX_WR - 1 + 1023) / 1024);
        COCOS_REQUIRE((long long)nck * nrb * B < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "box3_logits_fwd: grid too large");
        hipLaunchKernelGGL(box3_walk_kernel<false>, dim3((unsigned)(nck * nrb * B)), dim3(256), 0, as_stream(stream),
                           c_raw, f, mu, nu, a, b, (int)N, h, w, nck, nrb, k_unfolded, scale, (float*)nullptr);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

static size_t box3_walk_blocks(int B, size_t N) {
    return ((N + cocos::BX_WR - 1 + 1023) / 1024) * ((N + cocos::BX_WR - 1) / cocos::BX_WR) * (size_t)B;
}

extern "C" size_t cocos_box3_logits_bwd_workspace_bytes(int B, int h, int w) {
    const size_t N = (size_t)h * w;
    const size_t nrb = (N + cocos::BX_RB - 1) / cocos::BX_RB, nchunk = (N + 1023) / 1024;
    // row / column partial sums, + one partial maximum per workgroup of the box kernel (the _amax form)
    return (2 * (size_t)B * (nrb * N + N * nchunk) + box3_walk_blocks(B, N)) * sizeof(float);
}

extern "C" int cocos_absmax_accumulate(const float* x, long long n, float* inout_dev, cocos_stream_t stream);

static int box3_logits_bwd_impl(const float* g, const float* f, const float* mu, const float* nu, const float* a,
                                const float* b, float* dc_raw, float* r1, float* r2, float* c1, float* c2, void* ws,
                                size_t ws_bytes, int B, int h, int w, float scale, float* dc_amax_inout,
                                cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(g && f && mu && nu && a && b && dc_raw && r1 && r2 && c1 && c2, COCOS_ERR_INVALID,
                  "box3_logits_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && h >= 1 && w >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "box3_logits_bwd: bad dims B=%d h=%d w=%d", B, h, w);
    const long long N = (long long)h * w;
    COCOS_REQUIRE(N <= 65535 && N * N * 4 < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED,
                  "box3_logits_bwd: grid %dx%d too large", h, w);
    COCOS_REQUIRE(ws && ws_bytes >= cocos_box3_logits_bwd_workspace_bytes(B, h, w), COCOS_ERR_WORKSPACE,
                  "box3_logits_bwd: workspace too small (%zu bytes)", ws_bytes);
    hipStream_t s = as_stream(stream);
    const int nrb = (int)((N + BX_RB - 1) / BX_RB), nchunk = (int)((N + 1023) / 1024);
    float* c1p = static_cast<float*>(ws);
    float* c2p = c1p + (size_t)B * nrb * N;
    float* r1p = c2p + (size_t)B * nrb * N;
    float* r2p = r1p + (size_t)B * N * nchunk;
    float* amax_part = r2p + (size_t)B * N * nchunk;
    COCOS_REQUIRE((long long)nchunk * N * B < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "box3_logits_bwd: grid too large");
    {
        const int nrw = (int)((N + BX_WR - 1) / BX_WR), nck = (int)((N + BX_WR - 1 + 1023) / 1024);
        hipLaunchKernelGGL(box3_walk_kernel<true>, dim3((unsigned)(nck * nrw * B)), dim3(256), 0, s, g, dc_raw, mu, nu,
                           a, b, (int)N, h, w, nck, nrw, 0.f, scale, dc_amax_inout ? amax_part : (float*)nullptr);
        if (dc_amax_inout) {
            const int rc = cocos_absmax_accumulate(amax_part, (long long)nck * nrw * B, dc_amax_inout, stream);
            if (rc != COCOS_OK) return rc;
        }
    }
    hipLaunchKernelGGL(box3_bwd_reduce_kernel, dim3(nchunk, nrb, B), dim3(256), 0, s, g, f, mu, nu, a, b,
                       r1p, r2p, c1p, c2p, (int)N, nchunk, nrb);
    hipLaunchKernelGGL(box3_finish_kernel, dim3((unsigned)((N + 255) / 256), B), dim3(256), 0, s,
                       (const float*)r1p, (const float*)r2p, (const float*)c1p, (const float*)c2p, r1, r2,
                       c1, c2, (int)N, nchunk, nrb);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_box3_logits_bwd(const float* g, const float* f, const float* mu, const float* nu,
                                     const float* a, const float* b, float* dc_raw, float* r1, float* r2,
                                     float* c1, float* c2, void* ws, size_t ws_bytes, int B, int h, int w,
                                     float scale, cocos_stream_t stream) {
    return box3_logits_bwd_impl(g, f, mu, nu, a, b, dc_raw, r1, r2, c1, c2, ws, ws_bytes, B, h, w, scale, nullptr,
                                stream);
}

// Same, and on return *dc_amax_inout = max(*dc_amax_inout, max|dc_raw|) (cell must hold a finite value >= 0).
extern "C" int cocos_box3_logits_bwd_amax(const float* g, const float* f, const float* mu, const float* nu,
                                          const float* a, const float* b, float* dc_raw, float* r1, float* r2,
                                          float* c1, float* c2, void* ws, size_t ws_bytes, int B, int h, int w,
                                          float scale, float* dc_amax_inout, cocos_stream_t stream) {
    COCOS_REQUIRE(dc_amax_inout, COCOS_ERR_INVALID, "box3_logits_bwd_amax: null amax cell");
    return box3_logits_bwd_impl(g, f, mu, nu, a, b, dc_raw, r1, r2, c1, c2, ws, ws_bytes, B, h, w, scale,
                                dc_amax_inout, stream);
}
