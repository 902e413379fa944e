// K2 backward: gradients of  out = softmax_j(qn^T kn / T) . v^T  w.r.t. qn, kn (and v), gfx950.
//
// Replaces the autograd of correspondence.py:291-318 (softmax-backward, two bmm-backwards and the
// 1-1.5 GiB of saved [B,HW,HW] activations at B=8).  Flash-style: only the row log-sum-exp is
// saved by the forward; logits tiles are recomputed here with the same exact-fp32 MFMA.
//
// Math (per sample; i = query, j = key, P = softmax rows, D_i = sum_c dout[c,i] * out[c,i]):
//     dP[i,j] = sum_c dout[c,i] v[c,j]          dS[i,j] = P[i,j] * (dP[i,j] - D_i)
//     dqn[k,i] = 1/T * sum_j dS[i,j] kn[k,j]    dkn[k,j] = 1/T * sum_i dS[i,j] qn[k,i]
//     dv[c,j]  = sum_i P[i,j] dout[c,i]
//
// ONE kernel template does both sides.  A workgroup keeps 128 "resident" positions (32 per wave,
// their K-vector in registers as the MFMA B operand, their C-vector — dout or v — in LDS) and
// streams 32-position tiles of the other side through LDS:
//     side dq :  resident = queries, streamed = keys,    stats (lse, D) per lane
//     side dkv:  resident = keys,    streamed = queries, stats per accumulator register (LDS tile)
// Accumulator tiles are always [streamed (registers) x resident (lanes)], so P / dS leave the
// VALU already in the B-operand layout of the dX / dC MFMAs (acc_row_base in common.h) and the
// resident-side gradients accumulate in registers with no atomics; two kernels, deterministic.
#include "common.h"

namespace cocos {

constexpr int BWD_BR = 128;   // resident positions per workgroup
constexpr int BWD_LD = kTileLd;

__global__ __launch_bounds__(256) void corr_bwd_prep_kernel(const float* __restrict__ out,
                                                            const float* __restrict__ dout,
                                                            float* __restrict__ dvec, int Nq,
                                                            int Cv) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nq) return;
    const float* o = out + (size_t)b * Cv * Nq + i;
    const float* g = dout + (size_t)b * Cv * Nq + i;
    float acc = 0.f;
    for (int ch = 0; ch < Cv; ++ch) acc += o[(size_t)ch * Nq] * g[(size_t)ch * Nq];
    dvec[(size_t)b * Nq + i] = acc;
}

template <int KD, int CVB, bool STATS_RESIDENT, bool WITH_DC>
__global__ __launch_bounds__(256, 1) void corr_bwd_kernel(
    const float* __restrict__ xr,    // resident X [B,KD,R]
    const float* __restrict__ cr,    // resident C [B,Cv,R]   (dq: dout, dkv: v)
    const float* __restrict__ xs,    // streamed X [B,KD,S]
    const float* __restrict__ cs,    // streamed C [B,Cv,S]   (dq: v,    dkv: dout)
    const float* __restrict__ lse,   // [B,Nq]  (Nq = R for dq, S for dkv)
    const float* __restrict__ dvec,  // [B,Nq]
    float* __restrict__ dxr,         // out [B,KD,R]
    float* __restrict__ dcr,         // out [B,Cv,R] (WITH_DC)
    int B, int R, int S, int Cv, float scale_log2, float inv_t) {
    constexpr int CVP = CVB * 32;
    constexpr int KB = KD / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                          // [KD][LD]   streamed X tile
    float* ct = xt + KD * BWD_LD;              // [CVP][LD]  streamed C tile
    float* crs = ct + CVP * BWD_LD;            // [CVP][128] resident C
    float* st = crs + CVP * BWD_BR;            // [2][32]    streamed stats (lse*log2e, D)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;

    const int nrb = (R + BWD_BR - 1) / BWD_BR;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nrb, rb0 = (vb % nrb) * BWD_BR;
    const int r_lane = rb0 + wave * 32 + c;    // this lane's resident position

    const __amdgpu_buffer_rsrc_t xr_rs = make_rsrc(xr + (size_t)b * KD * R, (size_t)KD * R * 4);
    const __amdgpu_buffer_rsrc_t cr_rs = make_rsrc(cr + (size_t)b * Cv * R, (size_t)Cv * R * 4);
    const __amdgpu_buffer_rsrc_t xs_rs = make_rsrc(xs + (size_t)b * KD * S, (size_t)KD * S * 4);
    const __amdgpu_buffer_rsrc_t cs_rs = make_rsrc(cs + (size_t)b * Cv * S, (size_t)Cv * S * 4);
    const int nstat = STATS_RESIDENT ? R : S;
    const __amdgpu_buffer_rsrc_t lse_rs = make_rsrc(lse + (size_t)b * nstat, (size_t)nstat * 4);
    const __amdgpu_buffer_rsrc_t dv_rs = make_rsrc(dvec + (size_t)b * nstat, (size_t)nstat * 4);

    // ---- resident operands ------------------------------------------------------------------
    float xreg[KD / 2];
    {
        const unsigned off = r_lane < R ? (unsigned)(h * R + r_lane) * 4u : kBufOob;
#pragma unroll
        for (int kk = 0; kk < KD / 2; ++kk)
            xreg[kk] = buf_load1(xr_rs, off + (unsigned)(2 * kk * R) * 4u);
    }
    for (int idx = tid; idx < CVP * BWD_BR; idx += 256) {
        const int ch = idx >> 7, p = idx & 127;
        const unsigned off =
            (ch < Cv && rb0 + p < R) ? (unsigned)(ch * R + rb0 + p) * 4u : kBufOob;
        crs[idx] = buf_load1(cr_rs, off);
    }
    float lse2_lane = 0.f, d_lane = 0.f;
    if (STATS_RESIDENT) {
        // padded resident lanes: lse = +inf  ->  P = 0 everywhere in that column
        lse2_lane = r_lane < R ? buf_load1(lse_rs, (unsigned)r_lane * 4u) * kLog2e : INFINITY;
        d_lane = r_lane < R ? buf_load1(dv_rs, (unsigned)r_lane * 4u) : 0.f;
    }

    f32x16 dx[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[kb][r] = 0.f;
    f32x16 dc[WITH_DC ? CVB : 1];
#pragma unroll
    for (int cb = 0; cb < (WITH_DC ? CVB : 1); ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dc[cb][r] = 0.f;

    // ---- register staging of the next streamed tile -------------------------------------------
    TileRegs<KD> xsr;
    TileRegs<CVP> csr;
    float stat_r = 0.f;
    auto fetch = [&](int s0) {
        if (s0 + kTileCols <= S) {
            tile_fetch<KD, false>(xsr, xs_rs, KD, S, s0, tid);
            tile_fetch<CVP, false>(csr, cs_rs, Cv, S, s0, tid);
        } else {
            tile_fetch<KD, true>(xsr, xs_rs, KD, S, s0, tid);
            tile_fetch<CVP, true>(csr, cs_rs, Cv, S, s0, tid);
        }
        if (!STATS_RESIDENT && tid < 64) {
            const int sp = s0 + (tid & 31);
            if (tid < 32)   // padded streamed queries: lse = +inf  ->  P = 0 in that row
                stat_r = sp < S ? buf_load1(lse_rs, (unsigned)sp * 4u) * kLog2e : INFINITY;
            else
                stat_r = sp < S ? buf_load1(dv_rs, (unsigned)sp * 4u) : 0.f;
        }
    };

    const int ntiles = (S + kTileCols - 1) / kTileCols;
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        const int s0 = t * kTileCols;
        __syncthreads();
        tile_commit<KD>(xsr, xt, tid);
        tile_commit<CVP>(csr, ct, tid);
        if (!STATS_RESIDENT && tid < 64) st[tid] = stat_r;
        __syncthreads();   // also publishes `crs` on the first iteration
        if (t + 1 < ntiles) fetch(s0 + kTileCols);

        // ---- logits tile [streamed x resident] (recompute) ------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KD / 2; ++kk)
            s = mfma32(xt[(2 * kk + h) * BWD_LD + c], xreg[kk], s);

        // ---- dP tile = C_streamed^T . C_resident ------------------------------------------------
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
        for (int cc = 0; cc < CVP / 2; ++cc)
            dp = mfma32(ct[(2 * cc + h) * BWD_LD + c], crs[(2 * cc + h) * BWD_BR + wave * 32 + c],
                        dp);

        // ---- P = exp(S/T - lse),  dS = P * (dP - D) ---------------------------------------------
        const bool ragged = STATS_RESIDENT && (s0 + kTileCols > S);
        f32x16 p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int sl = acc_row_base(r) + 4 * h;   // streamed index inside the tile
            const float l2 = STATS_RESIDENT ? lse2_lane : st[sl];
            const float dd = STATS_RESIDENT ? d_lane : st[32 + sl];
            float pv = fast_exp2(s[r] * scale_log2 - l2);
            if (ragged && (s0 + sl >= S)) pv = 0.f;   // zero-filled keys past the end
            p[r] = pv;
            s[r] = pv * (dp[r] - dd);                 // s now holds dS
        }

        // ---- resident-side gradients ------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int sl = acc_row_base(r) + 4 * h;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                dx[kb] = mfma32(xt[(kb * 32 + c) * BWD_LD + sl], s[r], dx[kb]);
            if (WITH_DC) {
#pragma unroll
                for (int cb = 0; cb < CVB; ++cb)
                    dc[cb] = mfma32(ct[(cb * 32 + c) * BWD_LD + sl], p[r], dc[cb]);
            }
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------
    if (r_lane < R) {
        float* dx_b = dxr + (size_t)b * KD * R;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb * 32 + acc_row_base(r) + 4 * h;
                dx_b[(size_t)k * R + r_lane] = dx[kb][r] * inv_t;
            }
        if (WITH_DC) {
            float* dc_b = dcr + (size_t)b * Cv * R;
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = cb * 32 + acc_row_base(r) + 4 * h;
                    if (ch < Cv) dc_b[(size_t)ch * R + r_lane] = dc[cb][r];
                }
        }
    }
}

template <int KD, int CVB, bool STATS_RESIDENT, bool WITH_DC>
static int launch_bwd_side(const float* xr, const float* cr, const float* xs, const float* cs,
                           const float* lse, const float* dvec, float* dxr, float* dcr, int B,
                           int R, int S, int Cv, float inv_t, hipStream_t stream) {
    auto kern = corr_bwd_kernel<KD, CVB, STATS_RESIDENT, WITH_DC>;
    const size_t smem =
        ((size_t)(KD + CVB * 32) * BWD_LD + (size_t)CVB * 32 * BWD_BR + 64) * sizeof(float);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int nrb = (R + BWD_BR - 1) / BWD_BR;
    hipLaunchKernelGGL(kern, dim3(B * nrb), dim3(256), smem, stream, xr, cr, xs, cs, lse, dvec,
                       dxr, dcr, B, R, S, Cv, inv_t * kLog2e, inv_t);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

template <int CVB>
static int launch_bwd(const float* qn, const float* kn, const float* v, const float* lse,
                      const float* dout, const float* dvec, float* dqn, float* dkn, float* dv,
                      int B, int Nq, int Nk, int Cv, float inv_t, hipStream_t s) {
    int rc = COCOS_OK;
    if (dqn) {
        rc = launch_bwd_side<256, CVB, true, false>(qn, dout, kn, v, lse, dvec, dqn, nullptr, B,
                                                    Nq, Nk, Cv, inv_t, s);
        if (rc != COCOS_OK) return rc;
    }
    if (dv) {
        rc = launch_bwd_side<256, CVB, false, true>(kn, v, qn, dout, lse, dvec, dkn, dv, B, Nk,
                                                    Nq, Cv, inv_t, s);
    } else if (dkn) {
        rc = launch_bwd_side<256, CVB, false, false>(kn, v, qn, dout, lse, dvec, dkn, nullptr, B,
                                                     Nk, Nq, Cv, inv_t, s);
    }
    return rc;
}

}  // namespace cocos

extern "C" size_t cocos_corr_softmax_warp_bwd_workspace_bytes(int B, int K, int Nq, int Nk, int Cv) {
    (void)K; (void)Nk; (void)Cv;
    if (B < 1 || Nq < 1) return 0;
    return (size_t)B * Nq * sizeof(float);   // D_i = sum_c dout*out
}

extern "C" int cocos_corr_softmax_warp_bwd(const float* qn, const float* kn, const float* v,
                                           const float* out, const float* lse, const float* dout,
                                           float* dqn, float* dkn, float* dv, void* ws,
                                           size_t ws_bytes, int B, int K, int Nq, int Nk, int Cv,
                                           float inv_temperature, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && v && out && lse && dout, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd: null input pointer");
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd: bad dims B=%d Nq=%d Nk=%d Cv=%d", B, Nq, Nk, Cv);
    COCOS_REQUIRE(K == 256, COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_bwd: fused path needs K == 256 (got %d)", K);
    COCOS_REQUIRE(Cv <= 160, COCOS_ERR_UNSUPPORTED, "corr_softmax_warp_bwd: Cv=%d > 160", Cv);
    COCOS_REQUIRE((size_t)K * Nq * 4 < 0x7fffffffull && (size_t)K * Nk * 4 < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED, "corr_softmax_warp_bwd: per-sample tensor exceeds 2 GiB");
    COCOS_REQUIRE(!dv || dkn, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd: dv requires dkn (they come out of the same kernel)");
    COCOS_REQUIRE(ws && ws_bytes >= cocos_corr_softmax_warp_bwd_workspace_bytes(B, K, Nq, Nk, Cv),
                  COCOS_ERR_WORKSPACE, "corr_softmax_warp_bwd: workspace too small (%zu bytes)",
                  ws_bytes);
    hipStream_t s = as_stream(stream);
    float* dvec = static_cast<float*>(ws);
    hipLaunchKernelGGL(corr_bwd_prep_kernel, dim3((Nq + 255) / 256, B), dim3(256), 0, s, out, dout,
                       dvec, Nq, Cv);
    COCOS_HIP_CHECK(hipGetLastError());
    const int cvb = (Cv + 31) / 32;
    switch (cvb) {
        case 1: return launch_bwd<1>(qn, kn, v, lse, dout, dvec, dqn, dkn, dv, B, Nq, Nk, Cv, inv_temperature, s);
        case 2: return launch_bwd<2>(qn, kn, v, lse, dout, dvec, dqn, dkn, dv, B, Nq, Nk, Cv, inv_temperature, s);
        case 3: return launch_bwd<3>(qn, kn, v, lse, dout, dvec, dqn, dkn, dv, B, Nq, Nk, Cv, inv_temperature, s);
        case 4: return launch_bwd<4>(qn, kn, v, lse, dout, dvec, dqn, dkn, dv, B, Nq, Nk, Cv, inv_temperature, s);
        default: return launch_bwd<5>(qn, kn, v, lse, dout, dvec, dqn, dkn, dv, B, Nq, Nk, Cv, inv_temperature, s);
    }
}
