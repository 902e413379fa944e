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

// sched_group_barrier pins (per loop, so they can be ablated at compile time)
#ifndef COCOS_SGB_S
#define COCOS_SGB_S 1
#endif
#ifndef COCOS_SGB_DP
#define COCOS_SGB_DP 1
#endif
#ifndef COCOS_SGB_DX
#define COCOS_SGB_DX 1
#endif
#define SGB(en, mask, n) do { if (en) __builtin_amdgcn_sched_group_barrier(mask, n, 0); } while (0)

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

template <int KD, int CVB, bool STATS_RESIDENT, bool WITH_DC, bool STORE_DS>
__global__ __launch_bounds__(256, 1) void corr_bwd_kernel(
    const float* __restrict__ xr,    // resident X [B,KD,R]
    const float* __restrict__ cr,    // resident C [B,Cv,R]   (dq: dout, dkv: v)
    const float* __restrict__ xs,    // streamed X [B,KD,S]
    const float* __restrict__ cs,    // streamed C [B,Cv,S]   (dq: v,    dkv: dout)
    const float* __restrict__ lse,   // [B,Nq]  (Nq = R for dq, S for dkv)
    const float* __restrict__ dvec,  // [B,Nq]
    float* __restrict__ dxr,         // out [B,KD,R]
    float* __restrict__ dcr,         // out [B,Cv,R] (WITH_DC)
    float* __restrict__ dst,         // out [B,S,R]  (STORE_DS): dS^T / T, streamed-major
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

    // [S][R] matrices of this sample (dS^T out, saved logits in): lane offset = its column + its
    // half-wave's 4 rows; the tile / register part of the row index is wave-uniform (soffset)
    const __amdgpu_buffer_rsrc_t ds_rs = make_rsrc(STORE_DS ? dst + (size_t)b * S * R : nullptr,
                                                   STORE_DS ? (size_t)S * R * 4 : 0);
    const unsigned sr_lane_off = r_lane < R ? (unsigned)(4 * h * R + r_lane) * 4u : kBufOob;

    // ---- resident operands ------------------------------------------------------------------
    float xreg[KD / 2];
    {
        const unsigned off = r_lane < R ? (unsigned)(h * R + r_lane) * 4u : kBufOob;
#pragma unroll
        for (int kk = 0; kk < KD / 2; ++kk)
            xreg[kk] = buf_load1(xr_rs, off + (unsigned)(2 * kk * R) * 4u);
        // resident operand lives in the accumulator half of the register file (see the forward)
#pragma unroll
        for (int kk = 0; kk < KD / 2; ++kk) asm volatile("" : "+a"(xreg[kk]));
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
        // LDS operands are requested one batch ahead of the MFMAs that consume them and the
        // interleave is pinned with sched_group_barrier (see the forward kernel for the why).
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
            constexpr int NB = 8, NBATCH = KD / 2 / NB;
            const float* xl = xt + h * BWD_LD + c;
            float a[2][NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) a[0][u] = xl[(2 * u) * BWD_LD];
            SGB(COCOS_SGB_S, 0x100, NB / 2);
#pragma unroll
            for (int bt = 0; bt < NBATCH; ++bt) {
                if (bt + 1 < NBATCH) {
#pragma unroll
                    for (int u = 0; u < NB; ++u)
                        a[(bt + 1) & 1][u] = xl[(2 * ((bt + 1) * NB + u)) * BWD_LD];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) s = mfma32(a[bt & 1][u], xreg[bt * NB + u], s);
#pragma unroll
                for (int u = 0; u < NB / 2; ++u) {
                    SGB(COCOS_SGB_S, 0x008, 2);
                    SGB(COCOS_SGB_S, 0x100, 1);
                }
                if (COCOS_SGB_S) __builtin_amdgcn_sched_barrier(0);   // one region per batch
            }
        }

        // ---- dP tile = C_streamed^T . C_resident ------------------------------------------------
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
        {
            constexpr int NB = 4, NBATCH = CVP / 2 / NB;
            const float* cl = ct + h * BWD_LD + c;
            const float* rl = crs + h * BWD_BR + wave * 32 + c;
            float a[2][NB], bb[2][NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                a[0][u] = cl[(2 * u) * BWD_LD];
                bb[0][u] = rl[(2 * u) * BWD_BR];
            }
            SGB(COCOS_SGB_DP, 0x100, NB + NB / 2);
#pragma unroll
            for (int bt = 0; bt < NBATCH; ++bt) {
                if (bt + 1 < NBATCH) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        a[(bt + 1) & 1][u] = cl[(2 * ((bt + 1) * NB + u)) * BWD_LD];
                        bb[(bt + 1) & 1][u] = rl[(2 * ((bt + 1) * NB + u)) * BWD_BR];
                    }
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) dp = mfma32(a[bt & 1][u], bb[bt & 1][u], dp);
#pragma unroll
                for (int u = 0; u < NB / 2; ++u) {
                    SGB(COCOS_SGB_DP, 0x008, 2);
                    SGB(COCOS_SGB_DP, 0x100, 3);
                }
                if (COCOS_SGB_DP) __builtin_amdgcn_sched_barrier(0);   // one region per batch
            }
        }

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

        // ---- dS^T / T to HBM for the other side's GEMM (dq side only) -----------------------------
        if (STORE_DS) {
            const bool full = s0 + kTileCols <= S;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sp = s0 + acc_row_base(r);
                buf_store1s(ds_rs, s[r] * inv_t, (full || sp + 4 * h < S) ? sr_lane_off : kBufOob,
                            (unsigned)sp * (unsigned)R * 4u);
            }
        }

        // ---- resident-side gradients ------------------------------------------------------------
        {
            const float* xl = xt + c * BWD_LD + 4 * h;
            const float* cl = ct + c * BWD_LD + 4 * h;
            constexpr int NC = WITH_DC ? CVB : 0;
            float xa[2][KB], ca[2][NC > 0 ? NC : 1];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) xa[0][kb] = xl[kb * 32 * BWD_LD + acc_row_base(0)];
#pragma unroll
            for (int cb = 0; cb < NC; ++cb) ca[0][cb] = cl[cb * 32 * BWD_LD + acc_row_base(0)];
            if (COCOS_SGB_DX) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
                        xa[(r + 1) & 1][kb] = xl[kb * 32 * BWD_LD + acc_row_base(r + 1)];
#pragma unroll
                    for (int cb = 0; cb < NC; ++cb)
                        ca[(r + 1) & 1][cb] = cl[cb * 32 * BWD_LD + acc_row_base(r + 1)];
                }
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) dx[kb] = mfma32(xa[r & 1][kb], s[r], dx[kb]);
#pragma unroll
                for (int cb = 0; cb < NC; ++cb) dc[cb] = mfma32(ca[r & 1][cb], p[r], dc[cb]);
                // coarse pin (a 1:1 MFMA/read pattern over these 16 x (KB+NC) pairs sends hipcc's
                // group solver into minutes of compile time): half of this step's MFMAs, then ALL
                // reads of the next step, then the other half -> every read leads its use by >= 4 MFMAs
                SGB(COCOS_SGB_DX, 0x008, (KB + NC) / 2);
                SGB(COCOS_SGB_DX, 0x100, KB + NC);
                SGB(COCOS_SGB_DX, 0x008, (KB + NC) - (KB + NC) / 2);
                // one scheduling region per step keeps the group solver's work linear
                if (COCOS_SGB_DX) __builtin_amdgcn_sched_barrier(0);
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

template <int KD, int CVB, bool STATS_RESIDENT, bool WITH_DC, bool STORE_DS>
static int launch_bwd_side(const float* xr, const float* cr, const float* xs, const float* cs,
                           const float* lse, const float* dvec, float* dxr, float* dcr, float* dst,
                           int B, int R, int S, int Cv, float inv_t, hipStream_t stream) {
    auto kern = corr_bwd_kernel<KD, CVB, STATS_RESIDENT, WITH_DC, STORE_DS>;
    const size_t smem =
        ((size_t)(KD + CVB * 32) * BWD_LD + (size_t)CVB * 32 * BWD_BR + 64) * sizeof(float);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int nrb = (R + BWD_BR - 1) / BWD_BR;
    hipLaunchKernelGGL(kern, dim3(B * nrb), dim3(256), smem, stream, xr, cr, xs, cs, lse, dvec,
                       dxr, dcr, dst, B, R, S, Cv, inv_t * kLog2e, inv_t);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// Strategy for the key side.  Recomputing the logits a second time (flash-style dkv kernel) costs
// 2*HW^2*(K+Cv) extra fp32-MFMA FLOPs; on MI355X fp32 MFMA is only ~25 FLOP per HBM byte, so it is
// cheaper to let the query-side kernel write dS^T/T once (4 bytes per logit) and get dkn from a
// plain GEMM  dkn = qn . dS  that reads it back.  dv still needs P, which only the recompute
// kernel has.
int sgemm_dkn_from_ds(const float* qn, const float* dst, float* dkn, int B, int K, int Nq, int Nk,
                      hipStream_t s);

// query side with the forward's saved logits: corr_fused_bwd_saved.hip
int launch_bwd_query_saved(const float* kn, const float* v, const float* outp, const float* dout,
                           const float* lse, const float* lg, float* dqn, float* dst, int B, int Nq,
                           int Nk, int Cv, float inv_t, hipStream_t s);

template <int CVB>
static int launch_query_side(const float* qn, const float* kn, const float* v, const float* lse,
                             const float* dout, const float* dvec, float* dqn, float* dst, int B,
                             int Nq, int Nk, int Cv, float inv_t, hipStream_t s) {
    return dst ? launch_bwd_side<256, CVB, true, false, true>(qn, dout, kn, v, lse, dvec, dqn, nullptr,
                                                              dst, B, Nq, Nk, Cv, inv_t, s)
               : launch_bwd_side<256, CVB, true, false, false>(qn, dout, kn, v, lse, dvec, dqn,
                                                               nullptr, nullptr, B, Nq, Nk, Cv, inv_t, s);
}

template <int CVB>
static int launch_key_side(const float* qn, const float* kn, const float* v, const float* lse,
                           const float* dout, const float* dvec, float* dkn, float* dv, int B, int Nq,
                           int Nk, int Cv, float inv_t, hipStream_t s) {
    return dv ? launch_bwd_side<256, CVB, false, true, false>(kn, v, qn, dout, lse, dvec, dkn, dv,
                                                              nullptr, B, Nk, Nq, Cv, inv_t, s)
              : launch_bwd_side<256, CVB, false, false, false>(kn, v, qn, dout, lse, dvec, dkn, nullptr,
                                                               nullptr, B, Nk, Nq, Cv, inv_t, s);
}

#define COCOS_DISPATCH_CVB(cvb, FN, ...)                  \
    switch (cvb) {                                        \
        case 1: return FN<1>(__VA_ARGS__);                \
        case 2: return FN<2>(__VA_ARGS__);                \
        case 3: return FN<3>(__VA_ARGS__);                \
        case 4: return FN<4>(__VA_ARGS__);                \
        default: return FN<5>(__VA_ARGS__);               \
    }

static int check_dims(const char* who, int B, int K, int Nq, int Nk, int Cv) {
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "%s: bad dims B=%d Nq=%d Nk=%d Cv=%d", who, B, Nq, Nk, Cv);
    COCOS_REQUIRE(K == 256, COCOS_ERR_UNSUPPORTED, "%s: fused path needs K == 256 (got %d)", who, K);
    COCOS_REQUIRE(Cv <= 160, COCOS_ERR_UNSUPPORTED, "%s: Cv=%d > 160", who, Cv);
    COCOS_REQUIRE((size_t)K * Nq * 4 < 0x7fffffffull && (size_t)K * Nk * 4 < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED, "%s: per-sample tensor exceeds 2 GiB", who);
    return COCOS_OK;
}

}  // namespace cocos

extern "C" size_t cocos_corr_softmax_warp_bwd_workspace_bytes(int B, int K, int Nq, int Nk, int Cv) {
    (void)K; (void)Cv; (void)Nk;
    if (B < 1 || Nq < 1) return 0;
    return (size_t)B * Nq * sizeof(float);   // D_i = sum_c dout*out
}

extern "C" int cocos_corr_softmax_warp_bwd_prepare(const float* out, const float* dout, float* dvec,
                                                   int B, int Nq, int Cv, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(out && dout && dvec, COCOS_ERR_INVALID, "corr_softmax_warp_bwd_prepare: null pointer");
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Cv >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_prepare: bad dims B=%d Nq=%d Cv=%d", B, Nq, Cv);
    hipLaunchKernelGGL(corr_bwd_prep_kernel, dim3((Nq + 255) / 256, B), dim3(256), 0,
                       as_stream(stream), out, dout, dvec, Nq, Cv);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_corr_softmax_warp_bwd_query(const float* qn, const float* kn, const float* v,
                                                 const float* out, const float* lse,
                                                 const float* dout, const float* dvec,
                                                 const float* logits_t, float* dqn, float* ds_t, int B,
                                                 int K, int Nq, int Nk, int Cv, float inv_temperature,
                                                 cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && v && out && lse && dout && dqn, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query: null pointer");
    COCOS_REQUIRE(dvec || logits_t, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query: dvec (from _prepare) is required without logits_t");
    if (int rc = check_dims("corr_softmax_warp_bwd_query", B, K, Nq, Nk, Cv)) return rc;
    COCOS_REQUIRE((!ds_t && !logits_t) || (size_t)Nq * Nk * 4 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_bwd_query: per-sample [Nk,Nq] matrix exceeds 2 GiB; pass NULL");
    if (logits_t)
        return launch_bwd_query_saved(kn, v, out, dout, lse, logits_t, dqn, ds_t, B, Nq, Nk, Cv,
                                      inv_temperature, as_stream(stream));
    COCOS_DISPATCH_CVB((Cv + 31) / 32, launch_query_side, qn, kn, v, lse, dout, dvec, dqn, ds_t, B,
                       Nq, Nk, Cv, inv_temperature, as_stream(stream));
}

extern "C" int cocos_corr_softmax_warp_bwd_key(const float* qn, const float* kn, const float* v,
                                               const float* lse, const float* dout, const float* dvec,
                                               float* dkn, float* dv, int B, int K, int Nq, int Nk,
                                               int Cv, float inv_temperature, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && v && lse && dout && dvec && dkn, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_key: null pointer");
    if (int rc = check_dims("corr_softmax_warp_bwd_key", B, K, Nq, Nk, Cv)) return rc;
    COCOS_DISPATCH_CVB((Cv + 31) / 32, launch_key_side, qn, kn, v, lse, dout, dvec, dkn, dv, B, Nq,
                       Nk, Cv, inv_temperature, as_stream(stream));
}

extern "C" int cocos_corr_softmax_warp_bwd_key_from_ds(const float* qn, const float* ds_t, float* dkn,
                                                       int B, int K, int Nq, int Nk,
                                                       cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && ds_t && dkn, COCOS_ERR_INVALID, "corr_softmax_warp_bwd_key_from_ds: null pointer");
    COCOS_REQUIRE(B >= 1 && K >= 1 && Nq >= 1 && Nk >= 1, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_key_from_ds: bad dims B=%d K=%d Nq=%d Nk=%d", B, K, Nq, Nk);
    return sgemm_dkn_from_ds(qn, ds_t, dkn, B, K, Nq, Nk, as_stream(stream));
}

extern "C" int cocos_corr_softmax_warp_bwd(const float* qn, const float* kn, const float* v,
                                           const float* out, const float* lse, const float* dout,
                                           float* dqn, float* dkn, float* dv, void* ws,
                                           size_t ws_bytes, int B, int K, int Nq, int Nk, int Cv,
                                           float inv_temperature, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && v && out && lse && dout, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd: null input pointer");
    if (int rc = check_dims("corr_softmax_warp_bwd", B, K, Nq, Nk, Cv)) return rc;
    COCOS_REQUIRE(!dv || dkn, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd: dv requires dkn (they come out of the same kernel)");
    COCOS_REQUIRE(ws && ws_bytes >= cocos_corr_softmax_warp_bwd_workspace_bytes(B, K, Nq, Nk, Cv),
                  COCOS_ERR_WORKSPACE, "corr_softmax_warp_bwd: workspace too small (%zu bytes)",
                  ws_bytes);
    float* dvec = static_cast<float*>(ws);
    int rc = cocos_corr_softmax_warp_bwd_prepare(out, dout, dvec, B, Nq, Cv, stream);
    if (rc != COCOS_OK) return rc;
    if (dqn) {
        rc = cocos_corr_softmax_warp_bwd_query(qn, kn, v, out, lse, dout, dvec, nullptr, dqn, nullptr,
                                               B, K, Nq, Nk, Cv, inv_temperature, stream);
        if (rc != COCOS_OK) return rc;
    }
    if (dkn)
        rc = cocos_corr_softmax_warp_bwd_key(qn, kn, v, lse, dout, dvec, dkn, dv, B, K, Nq, Nk, Cv,
                                             inv_temperature, stream);
    return rc;
}
