// K2 backward, query side, when the forward saved its logits (training path) — gfx950.
//
// Same math as corr_fused_bwd.hip (autograd of correspondence.py:291-318 w.r.t. theta), but the
// logits tile is READ (logits_t written by the forward) instead of recomputed, which changes the
// resource picture completely: no resident K-slice is needed, so the accumulator half of the
// register file holds the wave's dO slice (the B operand of dP) and the dqn accumulators, LDS is
// free for a double-buffered ring, and the kernel becomes a two-GEMM pipeline
//
//     iteration t:   dS(t)  = P(t) * (dP(t) - D)                    VALU, short
//                    dqn   += K(t) . dS(t)          128 MFMAs   }   with, in the gaps between MFMAs:
//                                                               }   P(t+1) = exp2(S(t+1) - lse),
//                                                               }   the dS^T/T stores of tile t,
//                                                               }   the LDS commit of K(t+1), V(t+2)
//                    fetch K(t+2), V(t+3), S(t+2)   global -> registers, in flight for a whole tile
//                    dP(t+1) = V(t+1)^T . dO        Cv/2 MFMAs
//                    barrier                         (one per tile)
//
// D_i = sum_c dO[c,i] * out[c,i] is computed here from the wave's own dO slice (no prepare launch).
#include "common.h"

namespace cocos {

constexpr int BQS_LD = kTileLd;

// Optional phase timing (build with -DCOCOS_DEBUG_TIMING): shader-clock ticks spent by wave 0 of
// workgroup 0 in each phase of the tile loop, read back with cocos_debug_read_timing().
#ifdef COCOS_DEBUG_TIMING
__device__ long long g_phase_ticks[8];
#define PHASE_T(var) const long long var = __builtin_readcyclecounter()
#define PHASE_ADD(i, a, b) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_ticks[i] += (b) - (a); } while (0)
#else
#define PHASE_T(var) do {} while (0)
#define PHASE_ADD(i, a, b) do {} while (0)
#endif

template <int KD, int CVB, bool STORE_DS, bool RAGGED>
__global__ __launch_bounds__(256, 1) void corr_bwd_query_saved_kernel(
    const float* __restrict__ kn,    // [B,KD,Nk]
    const float* __restrict__ v,     // [B,Cv,Nk]
    const float* __restrict__ outp,  // [B,Cv,Nq]  forward output
    const float* __restrict__ dout,  // [B,Cv,Nq]
    const float* __restrict__ lse,   // [B,Nq]
    const float* __restrict__ lg,    // [B,Nk,Nq]  logits * log2(e)/T saved by the forward
    float* __restrict__ dqn,         // out [B,KD,Nq]
    float* __restrict__ dst,         // out [B,Nk,Nq] dS^T / T (STORE_DS)
    int B, int Nq, int Nk, int Cv, float inv_t) {
    constexpr int CVP = CVB * 32;
    constexpr int KB = KD / 32;
    constexpr int LD = BQS_LD;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                       // [2][KD][LD]   key tiles
    float* ct = smem + 2 * KD * LD;         // [2][CVP][LD]  V tiles

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;

    const int nqb = (Nq + 127) / 128;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, q0 = (vb % nqb) * 128;
    const int i_lane = q0 + wave * 32 + c;
    const bool live = i_lane < Nq;

    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kn + (size_t)b * KD * Nk, (size_t)KD * Nk * 4);
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(v + (size_t)b * Cv * Nk, (size_t)Cv * Nk * 4);
    const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(outp + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(dout + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    const __amdgpu_buffer_rsrc_t lg_rs = make_rsrc(lg + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const __amdgpu_buffer_rsrc_t ds_rs = make_rsrc(STORE_DS ? dst + (size_t)b * Nk * Nq : nullptr,
                                                   STORE_DS ? (size_t)Nk * Nq * 4 : 0);
    // [Nk][Nq] matrices: lane offset = its query column + its half-wave's 4 rows; the tile /
    // register part of the row index is wave-uniform and travels in the scalar offset
    const unsigned sr_lane_off = live ? (unsigned)(4 * h * Nq + i_lane) * 4u : kBufOob;

    // ---- resident: dO slice (B operand of dP, k = channel pair), D and lse of the lane's query ---
    float gd[CVP / 2];
    float d_lane;
    {
        // D is subtracted from dP, which is nearly equal to it wherever P is peaked: accumulate it
        // in fp64 (Cv/2 DFMAs per lane, once per kernel) so that the cancellation only sees dP's
        // own fp32 rounding, exactly like the reference's softmax backward
        double dacc = 0.0;
#pragma unroll
        for (int cc = 0; cc < CVP / 2; ++cc) {
            const int ch = 2 * cc + h;
            const unsigned off = (live && ch < Cv) ? (unsigned)(ch * Nq + i_lane) * 4u : kBufOob;
            gd[cc] = buf_load1(g_rs, off);
            dacc += (double)gd[cc] * (double)buf_load1(o_rs, off);
        }
        // the other half-wave holds the other channel parity
        const int lo = __shfl_xor((int)__double2loint(dacc), 32, 64);
        const int hi = __shfl_xor((int)__double2hiint(dacc), 32, 64);
        d_lane = (float)(dacc + __hiloint2double(hi, lo));
#pragma unroll
        for (int cc = 0; cc < CVP / 2; ++cc) asm volatile("" : "+a"(gd[cc]));
    }
    // padded lanes: lse = +inf -> P = 0 in that column
    const float lse2 = live ? lse[(size_t)b * Nq + i_lane] * kLog2e : INFINITY;

    f32x16 dx[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[kb][r] = 0.f;

    // ---- staging ---------------------------------------------------------------------------------
    TileRegs<KD> xs;
    TileRegs<CVP> cs;
    float sld[16];
    auto fetch_x = [&](int j0) {
        if (j0 + kTileCols <= Nk) tile_fetch<KD, false>(xs, k_rs, KD, Nk, j0, tid);
        else                      tile_fetch<KD, true>(xs, k_rs, KD, Nk, j0, tid);
    };
    auto fetch_c = [&](int j0) {
        if (j0 + kTileCols <= Nk) tile_fetch<CVP, false>(cs, v_rs, Cv, Nk, j0, tid);
        else                      tile_fetch<CVP, true>(cs, v_rs, Cv, Nk, j0, tid);
    };
    // rows past Nk read 0 (bounds check); their P is masked in prob()
    auto fetch_s_piece = [&](int r, int j0) {
        {
            const int jr = j0 + acc_row_base(r);
            // NB: the scalar offset is NOT covered by the descriptor's bounds check (only the
            // per-lane offset is), so rows that do not exist must be switched off explicitly —
            // including the whole look-ahead tiles past the end
            const bool ok = jr + 4 * h < Nk;
            // the scalar offset must stay provably wave-uniform (a lane-dependent select here makes
            // hipcc wrap every load in a waterfall loop: measured 4000 cycles per tile); rows that
            // do not exist are switched off through the per-lane offset instead
            sld[r] = buf_load1s(lg_rs, ok ? sr_lane_off : kBufOob, (unsigned)jr * (unsigned)Nq * 4u);
        }
    };
    auto fetch_s = [&](int j0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) fetch_s_piece(r, j0);
    };
    auto commit_piece = [&](const f32x4& x, float* tile, int u) {
        float* d = tile + (u * 32 + (tid >> 3)) * LD + (tid & 7) * 4;
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    };
    // P(t) from the staged logits; keys past Nk (zero-filled K/V rows) must not contribute
    auto prob = [&](int r, int j0) {
        float pv = fast_exp2(sld[r] - lse2);
        if (RAGGED && (j0 + acc_row_base(r) + 4 * h >= Nk)) pv = 0.f;
        return pv;
    };
    // dP tile = V_tile^T . dO  (A from LDS, B = resident dO), operands requested a batch ahead
    auto dp_tile = [&](const float* ctile, f32x16& dp) {
        constexpr int NB = 4, NBATCH = CVP / 2 / NB;
        const float* cl = ctile + h * LD + c;
        float a[2][NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) a[0][u] = cl[(2 * u) * LD];
        __builtin_amdgcn_sched_group_barrier(0x100, NB / 2, 0);
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
            if (bt + 1 < NBATCH) {
#pragma unroll
                for (int u = 0; u < NB; ++u) a[(bt + 1) & 1][u] = cl[(2 * ((bt + 1) * NB + u)) * LD];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) dp = mfma32(a[bt & 1][u], gd[bt * NB + u], dp);
#pragma unroll
            for (int u = 0; u < NB / 2; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int ntiles = (Nk + kTileCols - 1) / kTileCols;
    float* const xt0 = xt;
    float* const xt1 = xt + KD * LD;
    float* const ct0 = ct;
    float* const ct1 = ct + CVP * LD;

    // ---- prologue: K(0), V(0), V(1) resident; K(1), V(2), S(1) staged; dP(0), P(0) computed ------
    f32x16 dp, p;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
    {
        fetch_c(0);
        fetch_x(0);
        fetch_s(0);
        tile_commit<CVP>(cs, ct0, tid);
        tile_commit<KD>(xs, xt0, tid);
        fetch_c(kTileCols);
        fetch_x(kTileCols);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = prob(r, 0);
        fetch_s(kTileCols);
        __syncthreads();
        dp_tile(ct0, dp);
        tile_commit<CVP>(cs, ct1, tid);
        fetch_c(2 * kTileCols);
        __syncthreads();
    }

    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * kTileCols;
        float* const x_rd = (t & 1) ? xt1 : xt0;    // K(t)
        float* const x_wr = (t & 1) ? xt0 : xt1;    // <- K(t+1)
        float* const c_rd = (t & 1) ? ct0 : ct1;    // V(t+1)
        float* const c_wr = (t & 1) ? ct1 : ct0;    // <- V(t+2)

        PHASE_T(tp0);
        // dS(t) = P(t) * (dP(t) - D)
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] - d_lane);

        // ---- dqn += K(t) . dS(t)   (A = K tile read "transposed": 32 consecutive channels per lane
        //      group at one key, conflict-free thanks to the 33-float row stride) -------------------
        {
            const float* xl = x_rd + c * LD + 4 * h;
            float xa[2][KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) xa[0][kb] = xl[kb * 32 * LD + acc_row_base(0)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
                        xa[(r + 1) & 1][kb] = xl[kb * 32 * LD + acc_row_base(r + 1)];
                }
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) dx[kb] = mfma32(xa[r & 1][kb], ds[r], dx[kb]);
                // --- work hidden under this step's 8 MFMAs: one slice of everything, so that no
                //     phase ever issues a burst of memory instructions (a burst of ~30 loads stalls
                //     the in-order wave for 2-4k cycles while the L1 miss queue drains) ---
                p[r] = prob(r, j0 + kTileCols);                         // P(t+1)[r]
                fetch_s_piece(r, j0 + 2 * kTileCols);                   // S(t+2)[r] into the same register
                if (STORE_DS) {
                    const int jr = j0 + acc_row_base(r);
                    const bool ok = !RAGGED || (j0 + kTileCols <= Nk) || (jr + 4 * h < Nk);
                    buf_store1s(ds_rs, ds[r] * inv_t, ok ? sr_lane_off : kBufOob,
                                (unsigned)jr * (unsigned)Nq * 4u);
                }
                if (r < KB) {
                    commit_piece(xs.r[r], x_wr, r);                                        // K(t+1)
                    tile_fetch_piece<RAGGED>(xs.r[r], k_rs, r, KD, Nk, j0 + 2 * kTileCols, tid);
                } else if (r - KB < CVB) {
                    commit_piece(cs.r[r - KB], c_wr, r - KB);                              // V(t+2)
                    tile_fetch_piece<RAGGED>(cs.r[r - KB], v_rs, r - KB, Cv, Nk, j0 + 3 * kTileCols, tid);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, KB / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, KB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, KB - KB / 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        PHASE_T(tp1);
        PHASE_T(tp2);
        // ---- dP(t+1) -------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
        dp_tile(c_rd, dp);
        PHASE_T(tp3);
        __syncthreads();
        PHASE_T(tp4);
        PHASE_ADD(0, tp0, tp1); PHASE_ADD(1, tp1, tp2); PHASE_ADD(2, tp2, tp3); PHASE_ADD(3, tp3, tp4);
    }

    // ---- epilogue --------------------------------------------------------------------------------
    if (live) {
        float* dx_b = dqn + (size_t)b * KD * Nq;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb * 32 + acc_row_base(r) + 4 * h;
                dx_b[(size_t)k * Nq + i_lane] = dx[kb][r] * inv_t;
            }
    }
}

template <int CVB, bool STORE_DS, bool RAGGED>
static int launch_saved(const float* kn, const float* v, const float* outp, const float* dout,
                        const float* lse, const float* lg, float* dqn, float* dst, int B, int Nq,
                        int Nk, int Cv, float inv_t, hipStream_t s) {
    auto kern = corr_bwd_query_saved_kernel<256, CVB, STORE_DS, RAGGED>;
    const size_t smem = (size_t)2 * (256 + CVB * 32) * BQS_LD * sizeof(float);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int nqb = (Nq + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(B * nqb), dim3(256), smem, s, kn, v, outp, dout, lse, lg, dqn, dst,
                       B, Nq, Nk, Cv, inv_t);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

template <int CVB>
static int launch_saved_cvb(const float* kn, const float* v, const float* outp, const float* dout,
                            const float* lse, const float* lg, float* dqn, float* dst, int B, int Nq,
                            int Nk, int Cv, float inv_t, hipStream_t s) {
    // grids whose key count is a multiple of the 32-key tile (every power-of-two feature grid)
    // take the branch-free variant; ragged key counts pay a uniform test per fetched piece
    const bool ragged = (Nk % kTileCols) != 0;
#define COCOS_GO(DS, RG) launch_saved<CVB, DS, RG>(kn, v, outp, dout, lse, lg, dqn, dst, B, Nq, Nk, Cv, inv_t, s)
    if (dst) return ragged ? COCOS_GO(true, true) : COCOS_GO(true, false);
    return ragged ? COCOS_GO(false, true) : COCOS_GO(false, false);
#undef COCOS_GO
}

int launch_bwd_query_saved(const float* kn, const float* v, const float* outp, const float* dout,
                           const float* lse, const float* lg, float* dqn, float* dst, int B, int Nq,
                           int Nk, int Cv, float inv_t, hipStream_t s) {
    switch ((Cv + 31) / 32) {
        case 1: return launch_saved_cvb<1>(kn, v, outp, dout, lse, lg, dqn, dst, B, Nq, Nk, Cv, inv_t, s);
        case 2: return launch_saved_cvb<2>(kn, v, outp, dout, lse, lg, dqn, dst, B, Nq, Nk, Cv, inv_t, s);
        case 3: return launch_saved_cvb<3>(kn, v, outp, dout, lse, lg, dqn, dst, B, Nq, Nk, Cv, inv_t, s);
        case 4: return launch_saved_cvb<4>(kn, v, outp, dout, lse, lg, dqn, dst, B, Nq, Nk, Cv, inv_t, s);
        default: return launch_saved_cvb<5>(kn, v, outp, dout, lse, lg, dqn, dst, B, Nq, Nk, Cv, inv_t, s);
    }
}

}  // namespace cocos

#ifdef COCOS_DEBUG_TIMING
extern "C" int cocos_debug_read_timing(long long* host8, int reset) {
    using namespace cocos;
    COCOS_HIP_CHECK(hipDeviceSynchronize());
    COCOS_HIP_CHECK(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_phase_ticks), 8 * sizeof(long long)));
    if (reset) {
        long long z[8] = {0};
        COCOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), z, sizeof(z)));
    }
    return COCOS_OK;
}
#endif
