// Softmax + warp (and its backward) from MATERIALISED logits — gfx950.
//
// Used when the logits cannot come out of a plain K = 256 correlation inside the kernel — today the
// match_kernel = 3 path, whose logits are the diagonal box filter of box3_unfold.hip.  It replaces
// F.softmax (correspondence.py:307) + the P @ V products (:318, :334, ...) and their autograd with
// one streaming kernel each: P never reaches HBM, only the logits are read (forward) and the
// logit gradient written (backward).
//
// Logits are KEY-MAJOR, lg_t[b, j, i] = f[b, i, j] (already divided by the temperature): lanes run
// over queries i, accumulator registers over keys j — the same tile layout as the fused kernels, so
// the exponentiated registers feed the P.V MFMA (B operand) directly and a tile is loaded / stored
// as 16 coalesced 128-byte row segments per half-wave.
//
//   forward :  out[b,c,i] = sum_j softmax_j(lg_t[b,j,i]) v[b,c,j] ,  lse[b,i]
//   backward:  d lg_t[b,j,i] = P[i,j] * (dP[i,j] - D_i),  dP = v^T dout,  D_i = sum_c dout*out
#include "common.h"

namespace cocos {

constexpr int LSW_LD = kTileLd;
constexpr float kLswRescaleThr = 8.0f;

template <int CVB, bool BWD>
__global__ __launch_bounds__(256, 1) void logits_softmax_warp_kernel(
    const float* __restrict__ lg,     // [B,Nk,Nq]
    const float* __restrict__ v,      // [B,Cv,Nk]
    float* __restrict__ outp,         // fwd: out [B,Cv,Nq] (written) ; bwd: forward output (read)
    float* __restrict__ lse,          // fwd: written ; bwd: read
    const float* __restrict__ dout,   // bwd: [B,Cv,Nq]
    float* __restrict__ dlg,          // bwd: out [B,Nk,Nq]
    int B, int Nq, int Nk, int Cv) {
    constexpr int CVP = CVB * 32;
    constexpr int LD = LSW_LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][CVP][LD]  V tiles

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int nqb = (Nq + 127) / 128;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, q0 = (vb % nqb) * 128;
    const int i_lane = q0 + wave * 32 + c;
    const bool live = i_lane < Nq;

    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(v + (size_t)b * Cv * Nk, (size_t)Cv * Nk * 4);
    const __amdgpu_buffer_rsrc_t lg_rs = make_rsrc(lg + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const __amdgpu_buffer_rsrc_t dl_rs = make_rsrc(BWD ? dlg + (size_t)b * Nk * Nq : nullptr,
                                                   BWD ? (size_t)Nk * Nq * 4 : 0);
    const unsigned sr_lane_off = live ? (unsigned)(4 * h * Nq + i_lane) * 4u : kBufOob;

    // ---- backward only: resident dO slice (B operand of dP), D, lse ----------------------------------
    float gd[BWD ? CVP / 2 : 1];
    float d_lane = 0.f, lse2 = 0.f;
    if (BWD) {
        const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(outp + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
        const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(dout + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
        double dacc = 0.0;   // fp64: D is subtracted from the nearly equal dP where P is peaked
#pragma unroll
        for (int cc = 0; cc < (BWD ? CVP / 2 : 1); ++cc) {
            const int ch = 2 * cc + h;
            const unsigned off = (live && ch < Cv) ? (unsigned)(ch * Nq + i_lane) * 4u : kBufOob;
            gd[cc] = buf_load1(g_rs, off);
            dacc += (double)gd[cc] * (double)buf_load1(o_rs, off);
        }
        const int lo = __shfl_xor((int)__double2loint(dacc), 32, 64);
        const int hi = __shfl_xor((int)__double2hiint(dacc), 32, 64);
        d_lane = (float)(dacc + __hiloint2double(hi, lo));
#pragma unroll
        for (int cc = 0; cc < (BWD ? CVP / 2 : 1); ++cc) asm volatile("" : "+a"(gd[cc]));
        lse2 = live ? lse[(size_t)b * Nq + i_lane] * kLog2e : INFINITY;   // padded lanes: P = 0
    }

    f32x16 o[BWD ? 1 : CVB];
#pragma unroll
    for (int cb = 0; cb < (BWD ? 1 : CVB); ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    TileRegs<CVP> cs;
    float sld[16];
    auto fetch_s_piece = [&](int r, int j0) {
        const int jr = j0 + acc_row_base(r);
        // the scalar offset is not bounds-checked: rows that do not exist are switched off per lane
        sld[r] = buf_load1s(lg_rs, (jr + 4 * h < Nk) ? sr_lane_off : kBufOob, (unsigned)jr * (unsigned)Nq * 4u);
    };
    auto commit_piece = [&](const f32x4& x, float* tile, int u) {
        float* d = tile + (u * 32 + (tid >> 3)) * LD + (tid & 7) * 4;
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    };

    const int ntiles = (Nk + kTileCols - 1) / kTileCols;
    float* const ct0 = smem;
    float* const ct1 = smem + CVP * LD;

    // ---- prologue: V(0) resident, V(1) staged, S(0) in registers, S(1)... fetched per piece ----------
#pragma unroll
    for (int u = 0; u < CVB; ++u) tile_fetch_piece<true>(cs.r[u], v_rs, u, Cv, Nk, 0, tid);
#pragma unroll
    for (int r = 0; r < 16; ++r) fetch_s_piece(r, 0);
    tile_commit<CVP>(cs, ct0, tid);
#pragma unroll
    for (int u = 0; u < CVB; ++u) tile_fetch_piece<true>(cs.r[u], v_rs, u, Cv, Nk, kTileCols, tid);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * kTileCols;
        float* const c_rd = (t & 1) ? ct1 : ct0;    // V(t)
        float* const c_wr = (t & 1) ? ct0 : ct1;    // <- V(t+1)

        // S(t) out of the staging registers (log2 domain); the registers take S(t+1) right away
        f32x16 s;
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = sld[r] * kLog2e;
            if (j0 + acc_row_base(r) + 4 * h >= Nk) x = -INFINITY;   // keys past the end
            s[r] = x;
            tmax = fmaxf(tmax, x);
            fetch_s_piece(r, j0 + kTileCols);
        }

        if (!BWD) {
            // ---- online softmax with lazy rescale (see corr_fused_fwd.hip) ---------------------------
            tmax = fmaxf(tmax, swap_half(tmax));
            if (__any(tmax > m_run + kLswRescaleThr)) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = fast_exp2(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int cb = 0; cb < (BWD ? 1 : CVB); ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = fast_exp2(s[r] - m_run);
                psum += s[r];
            }
            l_run += psum;
            // ---- O^T += V^T . P^T ; V(t+1) goes to LDS and V(t+2) is requested under the MFMAs --------
            const float* vl = c_rd + c * LD + 4 * h;
            float va[2][CVB];
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb) va[0][cb] = vl[cb * 32 * LD + acc_row_base(0)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) {
#pragma unroll
                    for (int cb = 0; cb < CVB; ++cb)
                        va[(r + 1) & 1][cb] = vl[cb * 32 * LD + acc_row_base(r + 1)];
                }
#pragma unroll
                for (int cb = 0; cb < CVB; ++cb) o[cb] = mfma32(va[r & 1][cb], s[r], o[cb]);
                if (r < CVB) {
                    commit_piece(cs.r[r], c_wr, r);
                    tile_fetch_piece<true>(cs.r[r], v_rs, r, Cv, Nk, j0 + 2 * kTileCols, tid);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, (CVB + 1) / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, CVB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CVB / 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // ---- dP(t) = V(t)^T . dO, then d logits = P (dP - D) ---------------------------------------
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            constexpr int NB = 4, NBATCH = CVP / 2 / NB;
            const float* cl = c_rd + h * LD + c;
            float a[2][NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) a[0][u] = cl[(2 * u) * LD];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bt = 0; bt < NBATCH; ++bt) {
                if (bt + 1 < NBATCH) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) a[(bt + 1) & 1][u] = cl[(2 * ((bt + 1) * NB + u)) * LD];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) dp = mfma32(a[bt & 1][u], gd[bt * NB + u], dp);
                if (bt < CVB) {
                    commit_piece(cs.r[bt], c_wr, bt);
                    tile_fetch_piece<true>(cs.r[bt], v_rs, bt, Cv, Nk, j0 + 2 * kTileCols, tid);
                }
                if (bt < 16) s[bt] = fast_exp2(s[bt] - lse2);      // P(t)[bt], hidden under the MFMAs
#pragma unroll
                for (int u = 0; u < NB / 2; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = NBATCH; r < 16; ++r) s[r] = fast_exp2(s[r] - lse2);   // (only when Cv <= 96)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = j0 + acc_row_base(r);
                buf_store1s(dl_rs, s[r] * (dp[r] - d_lane), (jr + 4 * h < Nk) ? sr_lane_off : kBufOob,
                            (unsigned)jr * (unsigned)Nq * 4u);
            }
        }
        __syncthreads();
    }

    if (!BWD) {
        const float l_tot = l_run + swap_half(l_run);
        const float inv_l = 1.0f / l_tot;
        if (live) {
            float* out_b = outp + (size_t)b * Cv * Nq;
#pragma unroll
            for (int cb = 0; cb < (BWD ? 1 : CVB); ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = cb * 32 + acc_row_base(r) + 4 * h;
                    if (ch < Cv) out_b[(size_t)ch * Nq + i_lane] = o[cb][r] * inv_l;
                }
            if (h == 0) lse[(size_t)b * Nq + i_lane] = (m_run + log2f(l_tot)) * kLn2;
        }
    }
}

template <int CVB, bool BWD>
static int launch_lsw(const float* lg, const float* v, float* outp, float* lse, const float* dout,
                      float* dlg, int B, int Nq, int Nk, int Cv, hipStream_t s) {
    auto kern = logits_softmax_warp_kernel<CVB, BWD>;
    const size_t smem = (size_t)2 * CVB * 32 * LSW_LD * sizeof(float);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int nqb = (Nq + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(B * nqb), dim3(256), smem, s, lg, v, outp, lse, dout, dlg, B, Nq, Nk, Cv);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

template <bool BWD>
static int dispatch_lsw(const float* lg, const float* v, float* outp, float* lse, const float* dout,
                        float* dlg, int B, int Nq, int Nk, int Cv, hipStream_t s) {
    switch ((Cv + 31) / 32) {
        case 1: return launch_lsw<1, BWD>(lg, v, outp, lse, dout, dlg, B, Nq, Nk, Cv, s);
        case 2: return launch_lsw<2, BWD>(lg, v, outp, lse, dout, dlg, B, Nq, Nk, Cv, s);
        case 3: return launch_lsw<3, BWD>(lg, v, outp, lse, dout, dlg, B, Nq, Nk, Cv, s);
        case 4: return launch_lsw<4, BWD>(lg, v, outp, lse, dout, dlg, B, Nq, Nk, Cv, s);
        default: return launch_lsw<5, BWD>(lg, v, outp, lse, dout, dlg, B, Nq, Nk, Cv, s);
    }
}

static int check_lsw(const char* who, int B, int Nq, int Nk, int Cv) {
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1 && B <= 65535, COCOS_ERR_INVALID,
                  "%s: bad dims B=%d Nq=%d Nk=%d Cv=%d", who, B, Nq, Nk, Cv);
    COCOS_REQUIRE(Cv <= 160, COCOS_ERR_UNSUPPORTED, "%s: Cv=%d > 160", who, Cv);
    COCOS_REQUIRE((size_t)Nq * Nk * 4 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "%s: per-sample logits exceed 2 GiB", who);
    return COCOS_OK;
}

}  // namespace cocos

extern "C" int cocos_logits_softmax_warp_fwd(const float* logits_t, const float* v, float* out, float* lse,
                                             int B, int Nq, int Nk, int Cv, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(logits_t && v && out && lse, COCOS_ERR_INVALID, "logits_softmax_warp_fwd: null pointer");
    if (int rc = check_lsw("logits_softmax_warp_fwd", B, Nq, Nk, Cv)) return rc;
    return dispatch_lsw<false>(logits_t, v, out, lse, nullptr, nullptr, B, Nq, Nk, Cv, as_stream(stream));
}

extern "C" int cocos_logits_softmax_warp_bwd(const float* logits_t, const float* v, const float* out,
                                             const float* lse, const float* dout, float* dlogits_t, int B,
                                             int Nq, int Nk, int Cv, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(logits_t && v && out && lse && dout && dlogits_t, COCOS_ERR_INVALID,
                  "logits_softmax_warp_bwd: null pointer");
    if (int rc = check_lsw("logits_softmax_warp_bwd", B, Nq, Nk, Cv)) return rc;
    return dispatch_lsw<true>(logits_t, v, const_cast<float*>(out), const_cast<float*>(lse), dout,
                              dlogits_t, B, Nq, Nk, Cv, as_stream(stream));
}
