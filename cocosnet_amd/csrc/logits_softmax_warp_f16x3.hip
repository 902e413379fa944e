// K7, split-precision flavour: softmax + warp from MATERIALISED key-major logits, and its backward, with the
// P.V and V.dO products on v_mfma_f32_32x32x16_f16 (f16 hi/lo operands, three terms, fp32 accumulate) — gfx950.
//
// Same contract as logits_softmax_warp.hip (correspondence.py:307 + :318/:334 and their autograd; used by
// match_kernel 3, whose logits come from the box-filter kernel K6): these are corr_fused_fwd_f16x3.hip /
// corr_fused_bwd_f16x3.hip without their correlation halves — the S tile is loaded instead of computed, the
// backward writes dlogits in fp32 and has no dqn product.  With the matrix work cut 5x both kernels are
// HBM-bound on the logits matrix (4 B/logit forward, 8 B/logit backward).
//   forward:  out[b,c,i] = sum_j softmax_j(lg[b,j,i]) v[b,c,j];  lse [B,Nq]
//   backward: dlg[b,j,i] = P[i,j] * (sum_c dout[b,c,i] v[b,c,j] - sum_c dout[b,c,i] out[b,c,i])
// The HBM-streamed logits of the next tile are requested at the very start of a tile's MFMA loop (the whole
// tile to arrive), V tiles (L2-resident) one 8-/16-byte piece per MFMA step.
#include "common.h"

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int LW_VROW = 40;             // halfs per channel row of the forward V tile (32 permuted keys + pad)
constexpr float kLwRescaleThr = 6.0f;   // see corr_fused_fwd_f16x3.hip
constexpr float kLwPBias = 9.0f;

__device__ __forceinline__ f32x16 lw_mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// P -> f16 hi and lo' = 2^11 * (p - hi) (common.h: split_pair_rtz_lo_scaled): forward only — the backward's dlogits are
// fp32 and its P never becomes an MFMA operand
__device__ __forceinline__ void lw_split_pair(float a, float b, f16x2& hi, f16x2& lo) {
    unsigned h, l;
    split_pair_rtz_lo_scaled(a, b, h, l);
    hi = __builtin_bit_cast(f16x2, h);
    lo = __builtin_bit_cast(f16x2, l);
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int CVB, bool RAGGED>
__global__ __launch_bounds__(256, 1) void lsw_fwd_f16x3_kernel(const float* __restrict__ lg,          // [B,Nk,Nq]
                                                               const _Float16* __restrict__ vh,       // [B,Cv,Nk]
                                                               const _Float16* __restrict__ vl,
                                                               float* __restrict__ out, float* __restrict__ lse,
                                                               const float* __restrict__ v_scale,     // s_v or NULL
                                                               int B, int Nq, int Nk, int Cv) {
    constexpr int CVP = CVB * 32, VPLANE = CVP * LW_VROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lsw_smem[];   // 75 KB at CVB = 5: dynamic
    _Float16* const vt = reinterpret_cast<_Float16*>(lsw_smem);                // [2 buf][hi|lo|hi*2^-11][CVP][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int nqb = (Nq + 127) / 128;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, q0 = (vb % nqb) * 128;
    const int i_lane = q0 + wave * 32 + c;
    const bool live = i_lane < Nq;

    const size_t vbytes = (size_t)Cv * Nk * 2;
    const __amdgpu_buffer_rsrc_t vh_rs = make_rsrc(vh + (size_t)b * Cv * Nk, vbytes);
    const __amdgpu_buffer_rsrc_t vl_rs = make_rsrc(vl + (size_t)b * Cv * Nk, vbytes);
    const __amdgpu_buffer_rsrc_t lg_rs = make_rsrc(lg + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const unsigned lane_off = live ? (unsigned)(4 * h * Nq + i_lane) * 4u : kBufOob;

    f32x16 o[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    u32x2 vst[2][CVB];
    float sld[16];
    auto fetch_v_piece = [&](int i, int j0) {          // plane i & 1, chunk i >> 1
        const int pl = i & 1, u = i >> 1;
        const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
        unsigned off = (unsigned)(row * Nk + j0 + 4 * kq) * 2u;
        if (row >= Cv || j0 + 4 * kq >= Nk) off = kBufOob;             // Nk % 4 == 0 (launcher)
        vst[pl][u] = __builtin_amdgcn_raw_buffer_load_b64(pl ? vl_rs : vh_rs, (int)off, 0, 0);
    };
    auto commit_v_piece = [&](int i, int buf) {
        const int pl = i & 1, u = i >> 1;
        const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
        const int slot = 16 * (kq >> 2) + 8 * (kq & 1) + 4 * ((kq >> 1) & 1);   // see corr_fused_fwd_f16x3.hip
        *reinterpret_cast<u32x2*>(vt + (buf * 3 + pl) * VPLANE + row * LW_VROW + slot) = vst[pl][u];
        if (pl == 0)      // the 2^-11-scaled copy of the hi plane: A operand of the V_hi . P_lo' term
            *reinterpret_cast<u32x2*>(vt + (buf * 3 + 2) * VPLANE + row * LW_VROW + slot) =
                u32x2{pk_unshift_f16(vst[0][u].x), pk_unshift_f16(vst[0][u].y)};
    };
    auto fetch_s = [&](int j0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jr = j0 + acc_row_base(r);
            // rows that do not exist are switched off through the per-lane offset (the scalar offset is not
            // bounds-checked and must stay wave-uniform)
            sld[r] = buf_load1s(lg_rs, (jr + 4 * h < Nk) ? lane_off : kBufOob, (unsigned)jr * (unsigned)Nq * 4u);
        }
    };

    const int ntiles = (Nk + 31) / 32;
#pragma unroll
    for (int i = 0; i < 2 * CVB; ++i) fetch_v_piece(i, 0);
    fetch_s(0);
#pragma unroll
    for (int i = 0; i < 2 * CVB; ++i) commit_v_piece(i, 0);
#pragma unroll
    for (int i = 0; i < 2 * CVB; ++i) fetch_v_piece(i, 32);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 32, buf = t & 1;
        const bool ragged = RAGGED && (j0 + 32 > Nk);

        // ---- online softmax of the loaded tile (log2 domain) ------------------------------------------------------
        float p[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = sld[r] * kLog2e;
            if (ragged && (j0 + acc_row_base(r) + 4 * h >= Nk)) x = -INFINITY;
            p[r] = x;
            tmax = fmaxf(tmax, x);
        }
        fetch_s(j0 + 32);                              // HBM stream: the whole tile to arrive
        tmax = fmaxf(tmax, swap_half(tmax));
        if (__any(tmax > m_run + kLwRescaleThr)) {
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = fast_exp2(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb) asm volatile("" : "+a"(o[cb]));
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb) asm volatile("" : "+a"(o[cb]));
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = fast_exp2(p[r] - m_run + kLwPBias);
            psum += p[r];
        }
        l_run += psum;
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                f16x2 a, bq;
                lw_split_pair(p[8 * tt + j], p[8 * tt + j + 1], a, bq);
                ph[tt][j] = a[0]; ph[tt][j + 1] = a[1];
                pl[tt][j] = bq[0]; pl[tt][j + 1] = bq[1];
            }

        // ---- O^T += V . P, with the staged V pieces of tile t+1 / t+2 riding between the MFMAs ----------------------
        {
            const _Float16* vbase = vt + buf * 3 * VPLANE + c * LW_VROW + h * 8;
            constexpr int NS = 2 * CVB;
            f16x8 a_h[2], a_l[2], a_s[2];
            a_h[0] = *reinterpret_cast<const f16x8*>(vbase);
            a_l[0] = *reinterpret_cast<const f16x8*>(vbase + VPLANE);
            a_s[0] = *reinterpret_cast<const f16x8*>(vbase + 2 * VPLANE);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int tt = i / CVB, cb = i % CVB, cur = i & 1, nxt = cur ^ 1;
                if (i + 1 < NS) {
                    const int t2 = (i + 1) / CVB, c2 = (i + 1) % CVB;
                    a_h[nxt] = *reinterpret_cast<const f16x8*>(vbase + c2 * 32 * LW_VROW + t2 * 16);
                    a_l[nxt] = *reinterpret_cast<const f16x8*>(vbase + VPLANE + c2 * 32 * LW_VROW + t2 * 16);
                    a_s[nxt] = *reinterpret_cast<const f16x8*>(vbase + 2 * VPLANE + c2 * 32 * LW_VROW + t2 * 16);
                }
                o[cb] = lw_mfma(a_h[cur], ph[tt], o[cb]);
                o[cb] = lw_mfma(a_s[cur], pl[tt], o[cb]);
                o[cb] = lw_mfma(a_l[cur], ph[tt], o[cb]);
                commit_v_piece(i, buf ^ 1);
                fetch_v_piece(i, j0 + 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < CVB; ++cb) asm volatile("" : "+a"(o[cb]));
        __syncthreads();
    }

    const float l_tot = l_run + swap_half(l_run);
    const float inv_l = (v_scale ? 1.0f / *v_scale : 1.0f) / l_tot;     // the V planes hold s_v * v
    if (live) {
        float* out_b = out + (size_t)b * Cv * Nq;
#pragma unroll
        for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = cb * 32 + acc_row_base(r) + 4 * h;
                if (ch < Cv) out_b[(size_t)ch * Nq + i_lane] = o[cb][r] * inv_l;
            }
        if (h == 0) lse[(size_t)b * Nq + i_lane] = (m_run + log2f(l_tot) - kLwPBias) * kLn2;
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward (w.r.t. the logits)
// ---------------------------------------------------------------------------------------------------------
template <int CVB, bool RAGGED>
__global__ __launch_bounds__(256, 1) void lsw_bwd_f16x3_kernel(
    const float* __restrict__ lg,                                          // [B,Nk,Nq]
    const _Float16* __restrict__ vph, const _Float16* __restrict__ vpl,   // [B,Nk,CVP] position-major planes of v
    const _Float16* __restrict__ gph, const _Float16* __restrict__ gpl,   // [B,Nq,CVP] planes of s_o*dout
    const float* __restrict__ g_scale,                                     // s_o (device)
    const float* __restrict__ v_scale,                                     // s_v (device) or NULL: vph/vpl hold s_v * v
    const float* __restrict__ outp, const float* __restrict__ dout,        // [B,Cv,Nq] fp32 (for D)
    const float* __restrict__ lse, float* __restrict__ dlg,                // [B,Nq]; out [B,Nk,Nq]
    int B, int Nq, int Nk, int Cv) {
    constexpr int CVP = CVB * 32, CVS = CVP / 16, VROW = CVP + 8, VPLANE = 32 * VROW;
    __shared__ __attribute__((aligned(16))) _Float16 vt[2 * 2 * VPLANE];   // [2 buf][hi|lo][32 keys][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int nqb = (Nq + 127) / 128;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, q0 = (vb % nqb) * 128;
    const int i_lane = q0 + wave * 32 + c;
    const bool live = i_lane < Nq;

    const size_t vbytes = (size_t)Nk * CVP * 2, gbytes = (size_t)Nq * CVP * 2;
    const __amdgpu_buffer_rsrc_t vh_rs = make_rsrc(vph + (size_t)b * Nk * CVP, vbytes);
    const __amdgpu_buffer_rsrc_t vl_rs = make_rsrc(vpl + (size_t)b * Nk * CVP, vbytes);
    const __amdgpu_buffer_rsrc_t gh_rs = make_rsrc(gph + (size_t)b * Nq * CVP, gbytes);
    const __amdgpu_buffer_rsrc_t gl_rs = make_rsrc(gpl + (size_t)b * Nq * CVP, gbytes);
    const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(outp + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(dout + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    const __amdgpu_buffer_rsrc_t lg_rs = make_rsrc(lg + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const __amdgpu_buffer_rsrc_t dl_rs = make_rsrc(dlg + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const unsigned lane_off = live ? (unsigned)(4 * h * Nq + i_lane) * 4u : kBufOob;

    const float s_o = *g_scale * (v_scale ? *v_scale : 1.0f);      // scale of dP' = V' . dO'
    f16x8 goh[CVS], gol[CVS];
    {
        const unsigned off = live ? (unsigned)(i_lane * CVP + h * 8) * 2u : kBufOob;
#pragma unroll
        for (int u = 0; u < CVS; ++u) {
            goh[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(gh_rs, (int)(off + u * 32u), 0, 0));
            gol[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(gl_rs, (int)(off + u * 32u), 0, 0));
        }
#pragma unroll
        for (int u = 0; u < CVS; ++u) {
            asm volatile("" : "+a"(goh[u]));
            asm volatile("" : "+a"(gol[u]));
        }
    }
    float d_lane;
    {
        double dacc = 0.0;
        for (int ch = h; ch < Cv; ch += 2) {
            const unsigned off = live ? (unsigned)(ch * Nq + i_lane) * 4u : kBufOob;
            dacc += (double)buf_load1(g_rs, off) * (double)buf_load1(o_rs, off);
        }
        const int lo = __shfl_xor((int)__double2loint(dacc), 32, 64);
        const int hi = __shfl_xor((int)__double2hiint(dacc), 32, 64);
        d_lane = (float)((dacc + __hiloint2double(hi, lo)) * (double)s_o);
    }
    const float lse2 = live ? lse[(size_t)b * Nq + i_lane] * kLog2e : INFINITY;
    const float undo = 1.0f / s_o;

    constexpr int VCH = 32 * CVP / 8, VPT = (VCH + 255) / 256;
    u32x4 vst[2][VPT];
    float sld[16];
    auto fetch_v_piece = [&](int i, int j0) {
        const int pl = i & 1, u = i >> 1;
        const int g = u * 256 + tid, key = g / (CVP / 8), cc = g % (CVP / 8);
        vst[pl][u] = __builtin_amdgcn_raw_buffer_load_b128(
            pl ? vl_rs : vh_rs, (int)(g < VCH ? (unsigned)((j0 + key) * CVP + cc * 8) * 2u : kBufOob), 0, 0);
    };
    auto commit_v_piece = [&](int i, int buf) {
        const int pl = i & 1, u = i >> 1;
        const int g = u * 256 + tid, key = g / (CVP / 8), cc = g % (CVP / 8);
        if (g < VCH) *reinterpret_cast<u32x4*>(vt + (buf * 2 + pl) * VPLANE + key * VROW + cc * 8) = vst[pl][u];
    };
    auto fetch_s = [&](int j0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jr = j0 + acc_row_base(r);
            sld[r] = buf_load1s(lg_rs, (jr + 4 * h < Nk) ? lane_off : kBufOob, (unsigned)jr * (unsigned)Nq * 4u);
        }
    };

    const int ntiles = (Nk + 31) / 32;
#pragma unroll
    for (int i = 0; i < 2 * VPT; ++i) fetch_v_piece(i, 0);
    fetch_s(0);
#pragma unroll
    for (int i = 0; i < 2 * VPT; ++i) commit_v_piece(i, 0);
#pragma unroll
    for (int i = 0; i < 2 * VPT; ++i) fetch_v_piece(i, 32);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 32, buf = t & 1;
        // P of this tile first (frees the logits registers for the next tile's loads, which then have the whole
        // MFMA loop to arrive)
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pv = fast_exp2(sld[r] * kLog2e - lse2);
            if (RAGGED && (j0 + acc_row_base(r) + 4 * h >= Nk)) pv = 0.f;
            p[r] = pv;
        }
        fetch_s(j0 + 32);

        // ---- dP' = V(t) . dO' ---------------------------------------------------------------------------------------
        f32x16 dp0, dp1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dp0[r] = 0.f; dp1[r] = 0.f; }
        {
            const _Float16* vb0 = vt + buf * 2 * VPLANE + c * VROW + h * 8;
            f16x8 ah[2], al[2];
            ah[0] = *reinterpret_cast<const f16x8*>(vb0);
            al[0] = *reinterpret_cast<const f16x8*>(vb0 + VPLANE);
#pragma unroll
            for (int u = 0; u < CVS; ++u) {
                const int cur = u & 1, nxt = cur ^ 1;
                if (u + 1 < CVS) {
                    ah[nxt] = *reinterpret_cast<const f16x8*>(vb0 + (u + 1) * 16);
                    al[nxt] = *reinterpret_cast<const f16x8*>(vb0 + VPLANE + (u + 1) * 16);
                }
                dp0 = lw_mfma(ah[cur], goh[u], dp0);
                dp1 = lw_mfma(ah[cur], gol[u], dp1);
                dp1 = lw_mfma(al[cur], goh[u], dp1);
                if (u < 2 * VPT) {
                    commit_v_piece(u, buf ^ 1);
                    fetch_v_piece(u, j0 + 64);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_assert(2 * VPT <= CVS, "staging pieces must fit the dP steps");

        // ---- dlg = P * (dP - D), fp32, key-major (128-byte row segments per half-wave) ----------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jr = j0 + acc_row_base(r);
            const float val = p[r] * ((dp0[r] + dp1[r]) - d_lane) * undo;
            buf_store1s(dl_rs, val, (!RAGGED || jr + 4 * h < Nk) ? lane_off : kBufOob, (unsigned)jr * (unsigned)Nq * 4u);
        }
        __syncthreads();
    }
}

template <int CVB>
static int lsw_fwd_launch(const float* lg, const _Float16* vh, const _Float16* vl, float* out, float* lse,
                          const float* vs, int B, int Nq, int Nk, int Cv, hipStream_t s) {
    const int nqb = (Nq + 127) / 128;
    const size_t smem = (size_t)2 * 3 * CVB * 32 * LW_VROW * sizeof(_Float16);
    auto kern = (Nk % 32) ? lsw_fwd_f16x3_kernel<CVB, true> : lsw_fwd_f16x3_kernel<CVB, false>;
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(B * nqb), dim3(256), smem, s, lg, vh, vl, out, lse, vs, B, Nq, Nk, Cv);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

template <int CVB>
static int lsw_bwd_launch(const float* lg, const _Float16* vph, const _Float16* vpl, const _Float16* gph,
                          const _Float16* gpl, const float* gs, const float* vs, const float* outp, const float* dout,
                          const float* lse, float* dlg, int B, int Nq, int Nk, int Cv, hipStream_t s) {
    const int nqb = (Nq + 127) / 128;
    if (Nk % 32) hipLaunchKernelGGL((lsw_bwd_f16x3_kernel<CVB, true>), dim3(B * nqb), dim3(256), 0, s, lg, vph, vpl, gph, gpl, gs, vs, outp, dout, lse, dlg, B, Nq, Nk, Cv);
    else         hipLaunchKernelGGL((lsw_bwd_f16x3_kernel<CVB, false>), dim3(B * nqb), dim3(256), 0, s, lg, vph, vpl, gph, gpl, gs, vs, outp, dout, lse, dlg, B, Nq, Nk, Cv);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

}  // namespace cocos

extern "C" int cocos_logits_softmax_warp_fwd_f16x3(const float* logits_t, const void* vh, const void* vl, float* out,
                                                   float* lse, const float* v_scale_dev, int B, int Nq, int Nk, int Cv,
                                                   cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(logits_t && vh && vl && out && lse, COCOS_ERR_INVALID, "logits_softmax_warp_fwd_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1, COCOS_ERR_INVALID,
                  "logits_softmax_warp_fwd_f16x3: bad dims B=%d Nq=%d Nk=%d Cv=%d", B, Nq, Nk, Cv);
    COCOS_REQUIRE(Cv <= 160 && Nk % 4 == 0 && (size_t)Nq * Nk * 4 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "logits_softmax_warp_fwd_f16x3: needs Cv <= 160, Nk %% 4 == 0, logits < 2 GiB per sample");
    const _Float16 *a = static_cast<const _Float16*>(vh), *b2 = static_cast<const _Float16*>(vl);
    hipStream_t s = as_stream(stream);
    switch ((Cv + 31) / 32) {
        case 1: return lsw_fwd_launch<1>(logits_t, a, b2, out, lse, v_scale_dev, B, Nq, Nk, Cv, s);
        case 2: return lsw_fwd_launch<2>(logits_t, a, b2, out, lse, v_scale_dev, B, Nq, Nk, Cv, s);
        case 3: return lsw_fwd_launch<3>(logits_t, a, b2, out, lse, v_scale_dev, B, Nq, Nk, Cv, s);
        case 4: return lsw_fwd_launch<4>(logits_t, a, b2, out, lse, v_scale_dev, B, Nq, Nk, Cv, s);
        default: return lsw_fwd_launch<5>(logits_t, a, b2, out, lse, v_scale_dev, B, Nq, Nk, Cv, s);
    }
}

extern "C" int cocos_logits_softmax_warp_bwd_f16x3(const float* logits_t, const void* vph, const void* vpl,
                                                   const void* gph, const void* gpl, const float* g_scale_dev,
                                                   const float* v_scale_dev, const float* out, const float* dout,
                                                   const float* lse,
                                                   float* dlogits_t, int B, int Nq, int Nk, int Cv, int CvPad,
                                                   cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(logits_t && vph && vpl && gph && gpl && g_scale_dev && out && dout && lse && dlogits_t,
                  COCOS_ERR_INVALID, "logits_softmax_warp_bwd_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1, COCOS_ERR_INVALID,
                  "logits_softmax_warp_bwd_f16x3: bad dims B=%d Nq=%d Nk=%d Cv=%d", B, Nq, Nk, Cv);
    const int cvb = (Cv + 31) / 32;
    COCOS_REQUIRE(Cv <= 160 && CvPad == cvb * 32 && (size_t)Nq * Nk * 4 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "logits_softmax_warp_bwd_f16x3: needs Cv <= 160, CvPad = Cv rounded up to 32, logits < 2 GiB per sample");
    for (const void* p : {vph, vpl, gph, gpl})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "logits_softmax_warp_bwd_f16x3: planes must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
#define COCOS_ARGS                                                                                              \
    logits_t, static_cast<const _Float16*>(vph), static_cast<const _Float16*>(vpl), static_cast<const _Float16*>(gph), \
        static_cast<const _Float16*>(gpl), g_scale_dev, v_scale_dev, out, dout, lse, dlogits_t, B, Nq, Nk, Cv, s
    switch (cvb) {
        case 1: return lsw_bwd_launch<1>(COCOS_ARGS);
        case 2: return lsw_bwd_launch<2>(COCOS_ARGS);
        case 3: return lsw_bwd_launch<3>(COCOS_ARGS);
        case 4: return lsw_bwd_launch<4>(COCOS_ARGS);
        default: return lsw_bwd_launch<5>(COCOS_ARGS);
    }
#undef COCOS_ARGS
}
