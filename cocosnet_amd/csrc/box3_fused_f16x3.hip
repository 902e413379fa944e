// match_kernel = 3 (the reference's default, options/base_options.py:70) without unfolding and without a materialised
// logits matrix — K19 (box -> softmax -> warp, forward and backward) and K20 (box adjoint -> operand planes), gfx950.
//
// Replaces, for PONO_C on a 64-wide feature grid, the round-1/2 chain  K3 (C_raw) -> K6 (box filter, logits) -> K7
// (softmax + warp)  and its backward  K7' -> K6' -> K3'  (correspondence.py:276-280, :286-291, :304, :307, :318 and their
// autograd).  See box3_common.h for the decomposition; the data flow is
//
//   forward   T  = xbox(C_raw)                         cocos_box3_corr_xbox_f16x3 (hgemm_f16x3.hip, epilogue)
//             z[p,q] = scale * a_p * b_q * (T[p-w,q-w] + T[p,q] + T[p+w,q+w] - kc * mu_p * nu_q)
//             out = softmax_q(z) @ V                   K19 fwd: three 4 KB blocks of T per 32x32 tile, added in registers
//   backward  P from T again (no saved logits), L = dloss/dz = P * (V.dO - D),
//             G = L * scale * a_p * b_q  (= dloss / d(ybox T))  -> HBM, tile-blocked fp32              K19 bwd
//             d mu, d a (row sums of L) directly; d nu, d b (column sums) as per-workgroup partials
//             dC = xbox(ybox(G))  (the filter is self-adjoint) -> f16 hi/lo planes                     K20
//             d theta = dC . phi^T, d phi = dC^T . theta^T                                             hgemm (modes 3, 2)
//
// HBM traffic per logit: forward 4 B written (T) + 4 B read; backward 4 (T) + 4 (G) + 4 (G) + 4 (planes) + 2 x 4
// (planes) — against 16 B forward and 36 B backward for the materialised chain.
//
// mu / a (queries) and nu / b (keys) are the statistics of the unfolded, centred vectors (K12, unfold3_stats.hip):
// mean over the 9*256 entries and 1 / (norm + eps).  kc = 2304.
#include "box3_common.h"

// Ablation builds (debug only, results are WRONG): -DBX_ABLATE=<bits>  1: no column sums (butterfly + LDS atomics),
// 2: no G / P-plane stores, 4: only the centre T block is loaded (no y box), 8: no MFMAs, 16: key statistics not loaded,
// 32: no exp — tools/box3_ablate.sh times them.
#ifndef BX_ABLATE
#define BX_ABLATE 0
#endif

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BX_VROW = 40;             // halfs per channel row of the forward V tile (32 permuted keys + pad)
constexpr float kBxRescaleThr = 6.0f;   // lazy rescale of the running maximum (corr_fused_fwd_f16x3.hip)
constexpr float kBxPBias = 9.0f;
constexpr float kBxPPlaneScale = 16384.0f;
// Per-key statistics in LDS: the whole sample while it has <= 4096 keys (32 KB); beyond that (128-wide grids: 16384 keys)
// a ring of two chunks of BX_KCH keys — chunk c + 1 is staged by the whole workgroup during the first tile of chunk c, into
// the buffer that chunk c - 1 was read from (every wave passed that chunk's last barrier), and is first read 64 barriers later.
constexpr int BX_KCH = 2048, BX_TCH = BX_KCH / 32;

__device__ __forceinline__ f32x16 bx_mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// The three T blocks of one tile, in flight one tile ahead.
struct BxTile {
    f32x4 s[3][4];      // [dy + 1][g]: accumulator registers 4g..4g+3
};

// Geometry of a wave's query block, fixed for the kernel.
struct BxGeom {
    __amdgpu_buffer_rsrc_t t_rs;
    int qblk, py, tpr, tpr_log2, himg, nqblk;
    unsigned lane_off;      // lane * 16
    const float* kstat;     // LDS: [b_q (nk floats) | kc * nu_q * b_q (nk floats)] of this sample's keys (x 2 buffers when chunked)
    int nk;                 // keys per LDS buffer: Nk, or BX_KCH when chunked
    float* xt;              // TRANSPOSED T (see below): this wave's [32][BX_XT_ROW] floats of LDS; nullptr = T as stored
};

// Round 4: T = xbox(C) of the OTHER orientation is T transposed (the diagonal filter moves both indices alike), so the column
// pass of the cycle terms (correspondence.py:338,:351: softmax over the content positions) reads the row pass's T instead of
// running a second correlation GEMM: block (key tile kt, query block qb) of the transposed problem is block (qb, kt) of the
// stored tensor, transposed inside — the three y-box blocks are summed as stored and the 32 x 32 sum goes once through a
// per-wave LDS image (4 x ds_write_b128, 16 x ds_read_b32 per tile).  The backward writes its G the same way back, into the
// layout of the stored T, and ADDS it to what another pass has already written there (flags bit 1): one G, one box adjoint
// (K20) and one pair of GEMMs for both orientations.
constexpr int BX_XT_ROW = 36, BX_XT_FLOATS = 32 * BX_XT_ROW;
constexpr int BX_FLAG_TRANSPOSED = 1, BX_FLAG_ACCUMULATE = 2;

// x[4g + e] of lane (h, c) = element (8g + 4h + e, c) of a 32 x 32 matrix  ->  the same registers of the TRANSPOSED matrix
__device__ __forceinline__ void bx_transpose32(float (&x)[16], float* xt, int h, int c) {
    float* row = xt + c * BX_XT_ROW + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(row + 8 * g) = f32x4{x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
    __builtin_amdgcn_wave_barrier();      // one wave's LDS instructions execute in order: no workgroup barrier
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[4 * g + e] = xt[(8 * g + 4 * h + e) * BX_XT_ROW + c];
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void bx_fetch(BxTile& tl, const BxGeom& gm, int t, int ntiles) {
    const int tc = min(t, ntiles - 1);                  // look-ahead past the end re-reads the last tile
    const int ky = tc >> gm.tpr_log2;                   // (tpr = 2 or 4: a runtime division per tile was ~20 scalar instructions)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int dy = d - 1;
        if ((BX_ABLATE & 4) && d != 1) continue;
        const bool ok = (unsigned)(gm.py + dy) < (unsigned)gm.himg && (unsigned)(ky + dy) < (unsigned)gm.himg;
        const int kb = tc + dy * gm.tpr, qb = gm.qblk + dy * gm.tpr;
        const int blk = ok ? (gm.xt ? qb * gm.nqblk + kb : kb * gm.nqblk + qb) : 0;
        const unsigned voff = ok ? gm.lane_off : kBufOob;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            tl.s[d][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                gm.t_rs, (int)voff, (int)((unsigned)blk * 4096u + (unsigned)g * 1024u), 0));
    }
}

// The per-key statistics of the whole sample go to LDS once per workgroup (b_q and kn_q = kc * nu_q * b_q: 8 Nk bytes);
// every tile then reads its 2 x 16 values per lane as broadcast 16-byte LDS reads.  (Round-3 ablation: fetching them from
// global memory per tile — 8 buffer loads whose 32 lanes share an address — cost 0.06 ms of the 0.46 ms backward.)
// (`cap` = floats per array of the LDS buffer, `n` <= cap = keys staged)
__device__ __forceinline__ void bx_stage_stats(float* kstat, const float* __restrict__ b_k, const float* __restrict__ nu_k,
                                               int cap, int n, float kc, int tid) {
    for (int i = tid; i < n; i += 256) {
        const float bq = b_k[i];
        kstat[i] = bq;
        kstat[cap + i] = kc * nu_k[i] * bq;
    }
}
// chunked flavour: behind the barrier that ended tile t - 1; stages the chunk AFTER the one tile t opens
template <bool CHUNKED>
__device__ __forceinline__ void bx_stage_next_chunk(float* kstat, const float* __restrict__ b_k, const float* __restrict__ nu_k,
                                                    int t, int ntiles, float kc, int tid) {
    if (CHUNKED && (t & (BX_TCH - 1)) == 0 && t + BX_TCH < ntiles) {
        const int c1 = t / BX_TCH + 1;
        bx_stage_stats(kstat + (c1 & 1) * 2 * BX_KCH, b_k + c1 * BX_KCH, nu_k + c1 * BX_KCH, BX_KCH,
                       min(BX_KCH, ntiles * 32 - c1 * BX_KCH), kc, tid);
    }
}

// tt[r] = b_q * (T_sum) - mu_p * kc * nu_q * b_q  (the logit without its per-query factor scale * a_p); also returns
// b_q and kn_q per register for the backward.
template <bool CHUNKED>
__device__ __forceinline__ void bx_logits(const BxTile& tl, const BxGeom& gm, int t, int h, float mu_p, float (&tt)[16],
                                          float (&bq)[16], float (&kn)[16], int lane_c) {
    const float* ks = gm.kstat + (CHUNKED ? ((t / BX_TCH) & 1) * 2 * BX_KCH + (t & (BX_TCH - 1)) * 32 : t * 32);
    float ts[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            ts[4 * g + e] = (BX_ABLATE & 4) ? tl.s[1][g][e] : (tl.s[0][g][e] + tl.s[2][g][e]) + tl.s[1][g][e];
    if (gm.xt) bx_transpose32(ts, gm.xt, h, lane_c);      // (workgroup-uniform: one branch per tile)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(ks + 8 * g + 4 * h);
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(ks + gm.nk + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            bq[r] = b4[e];
            kn[r] = k4[e];
            tt[r] = __builtin_fmaf(bq[r], ts[r], -(mu_p * kn[r]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K19 forward
// ---------------------------------------------------------------------------------------------------------
// VLO0 (as in the K2 split kernels, corr_fused_fwd_f16x3.hip): only value block 0 has a non-zero f16 lo plane (one-hot label
// channels are exact in f16) — the V_lo . P_hi term, its fragment reads and its staging are skipped for the other blocks.
template <int CVB, bool VLO0, bool CHUNKED>
__device__ __forceinline__ void box3_sw_fwd_body(
    const float* __restrict__ T, const float* __restrict__ mu_q, const float* __restrict__ a_q,
    const float* __restrict__ nu_k, const float* __restrict__ b_k, const _Float16* __restrict__ vh,
    const _Float16* __restrict__ vl, float* __restrict__ out, float* __restrict__ lse,
    const float* __restrict__ v_scale, int B, int Nq, int Nk, int Cv, int himg, int wimg, float kc, float scale, int flags) {
    constexpr int CVP = CVB * 32, VPLANE = CVP * BX_VROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char bx_smem[];
    _Float16* const vt = reinterpret_cast<_Float16*>(bx_smem);      // [2 buf][hi|lo|hi*2^-11][CVP][VROW]
    float* const kstat = reinterpret_cast<float*>(vt + 2 * 3 * VPLANE);   // [b | kn][Nk]

    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform BY CONSTRUCTION for the compiler too (round 6): derived from threadIdx it lived in a VGPR, every T / G block offset
    // that depends on it (qblk, py, blk) was VGPR arithmetic, and every block load / store sat in a waterfall loop (readfirstlane /
    // compare / saveexec / branch: 12 loops per tile in K19, 56 in K20)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int nqb = Nq / 128;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, q0 = (vb % nqb) * 128;
    const int i_lane = q0 + wave * 32 + c;
    const float* const bk_b = b_k + (size_t)b * Nk;
    const float* const nu_b = nu_k + (size_t)b * Nk;
    bx_stage_stats(kstat, bk_b, nu_b, CHUNKED ? BX_KCH : Nk, CHUNKED ? BX_KCH : Nk, kc, tid);

    const size_t vbytes = (size_t)Cv * Nk * 2;
    const __amdgpu_buffer_rsrc_t vh_rs = make_rsrc(vh + (size_t)b * Cv * Nk, vbytes);
    const __amdgpu_buffer_rsrc_t vl_rs = make_rsrc(vl + (size_t)b * Cv * Nk, vbytes);
    BxGeom gm;
    gm.t_rs = make_rsrc(T + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    gm.kstat = kstat;
    gm.nk = CHUNKED ? BX_KCH : Nk;
    gm.tpr = wimg / 32;
    gm.tpr_log2 = gm.tpr == 4 ? 2 : 1;          // (64- or 128-wide grids: cocos_box3_fused_supported)
    gm.qblk = (q0 >> 5) + wave;
    gm.py = gm.qblk >> gm.tpr_log2;
    gm.himg = himg;
    gm.nqblk = Nq >> 5;
    gm.lane_off = (unsigned)lane * 16u;
    gm.xt = (flags & BX_FLAG_TRANSPOSED) ? kstat + (CHUNKED ? 4 * BX_KCH : 2 * Nk) + wave * BX_XT_FLOATS : nullptr;

    const float mu_p = mu_q[(size_t)b * Nq + i_lane];
    const float a2 = a_q[(size_t)b * Nq + i_lane] * scale * kLog2e;      // log2-domain factor of this query's logits (> 0)

    f32x16 o[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    u32x2 vst[2][CVB];
    // staging addresses: the tile-invariant part per lane once (round 6: `row >= Cv || j0 + 4 kq >= Nk` inside the loop compiled to two
    // EXEC-masked regions per piece — 20 s_and_saveexec / 30 s_or per tile), the tile as the buffer instruction's scalar offset.
    // Nk % 32 == 0 (whole tiles): a look-ahead tile past the end is the last tile again instead of an out-of-range test per lane.
    unsigned v_voff[CVB];
    int v_lds[CVB];
#pragma unroll
    for (int u = 0; u < CVB; ++u) {
        const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
        v_voff[u] = row < Cv ? (unsigned)(row * Nk + 4 * kq) * 2u : kBufOob;
        v_lds[u] = row * BX_VROW + 16 * (kq >> 2) + 8 * (kq & 1) + 4 * ((kq >> 1) & 1);   // permuted k-slots: see corr_fused_fwd_f16x3.hip
    }
    auto fetch_v_piece = [&](int i, int j0) {          // plane i & 1, chunk i >> 1 (= value block)
        const int pl = i & 1, u = i >> 1;
        if (VLO0 && pl == 1 && u >= 1) return;
        const int jc = min(j0, Nk - 32);
        vst[pl][u] = __builtin_amdgcn_raw_buffer_load_b64(pl ? vl_rs : vh_rs, (int)v_voff[u], (int)((unsigned)jc * 2u), 0);
    };
    auto commit_v_piece = [&](int i, int buf) {
        const int pl = i & 1, u = i >> 1;
        if (VLO0 && pl == 1 && u >= 1) return;
        _Float16* const d = vt + buf * 3 * VPLANE + v_lds[u];
        *reinterpret_cast<u32x2*>(d + pl * VPLANE) = vst[pl][u];
        if (pl == 0)
            *reinterpret_cast<u32x2*>(d + 2 * VPLANE) = u32x2{pk_unshift_f16(vst[0][u].x), pk_unshift_f16(vst[0][u].y)};
    };

    const int ntiles = Nk / 32;
    BxTile tl0, tl1;        // tiles t and t + 1 in flight: one tile of look-ahead did not cover the HBM latency of the T blocks
#pragma unroll
    for (int i = 0; i < 2 * CVB; ++i) fetch_v_piece(i, 0);
    bx_fetch(tl0, gm, 0, ntiles);
    bx_fetch(tl1, gm, 1, ntiles);
#pragma unroll
    for (int i = 0; i < 2 * CVB; ++i) commit_v_piece(i, 0);
#pragma unroll
    for (int i = 0; i < 2 * CVB; ++i) fetch_v_piece(i, 32);
    __syncthreads();

    auto do_tile = [&](int t, BxTile& tl) __attribute__((always_inline)) {
        const int j0 = t * 32, buf = t & 1;
        bx_stage_next_chunk<CHUNKED>(kstat, bk_b, nu_b, t, ntiles, kc, tid);
        float tt[16], bq[16], kn[16];
        bx_logits<CHUNKED>(tl, gm, t, h, mu_p, tt, bq, kn, c);
        bx_fetch(tl, gm, t + 2, ntiles);                  // two tiles ahead (round 4)
        float tmax = tt[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, tt[r]);
        tmax = fmaxf(tmax, swap_half(tmax)) * a2;
        if (__any(tmax > m_run + kBxRescaleThr)) {
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
        float p[16];
        float psum = 0.f;
        const float nmb = kBxPBias - m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = (BX_ABLATE & 32) ? tt[r] * 1e-3f + 1.0f : fast_exp2(__builtin_fmaf(tt[r], a2, nmb));
            psum += p[r];
        }
        l_run += psum;
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned hw, lw;
                split_pair_rtz_lo_scaled(p[8 * s + j], p[8 * s + j + 1], hw, lw);
                const f16x2 a = __builtin_bit_cast(f16x2, hw), bl = __builtin_bit_cast(f16x2, lw);
                ph[s][j] = a[0]; ph[s][j + 1] = a[1];
                pl[s][j] = bl[0]; pl[s][j + 1] = bl[1];
            }
        // ---- O^T += V . P, the staged V pieces of tile t+1 / t+2 riding between the MFMAs ---------------------------
        {
            const _Float16* vbase = vt + buf * 3 * VPLANE + c * BX_VROW + h * 8;
            constexpr int NS = 2 * CVB;
            f16x8 a_h[2], a_l[2], a_s[2];
            a_h[0] = *reinterpret_cast<const f16x8*>(vbase);
            a_l[0] = *reinterpret_cast<const f16x8*>(vbase + VPLANE);
            a_s[0] = *reinterpret_cast<const f16x8*>(vbase + 2 * VPLANE);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = i / CVB, cb = i % CVB, cur = i & 1, nxt = cur ^ 1;
                if (i + 1 < NS) {
                    const int s2 = (i + 1) / CVB, c2 = (i + 1) % CVB;
                    a_h[nxt] = *reinterpret_cast<const f16x8*>(vbase + c2 * 32 * BX_VROW + s2 * 16);
                    if (!VLO0 || c2 == 0)
                        a_l[nxt] = *reinterpret_cast<const f16x8*>(vbase + VPLANE + c2 * 32 * BX_VROW + s2 * 16);
                    a_s[nxt] = *reinterpret_cast<const f16x8*>(vbase + 2 * VPLANE + c2 * 32 * BX_VROW + s2 * 16);
                }
                if (!(BX_ABLATE & 8)) {
                    o[cb] = bx_mfma(a_h[cur], ph[s], o[cb]);
                    o[cb] = bx_mfma(a_s[cur], pl[s], o[cb]);      // (2^-11 V_hi) . (2^11 P_lo)
                    if (!VLO0 || cb == 0) o[cb] = bx_mfma(a_l[cur], ph[s], o[cb]);
                }
                commit_v_piece(i, buf ^ 1);
                fetch_v_piece(i, j0 + 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < CVB; ++cb) asm volatile("" : "+a"(o[cb]));
        __syncthreads();
    };
    {
        int t = 0;
        for (; t + 1 < ntiles; t += 2) {
            do_tile(t, tl0);
            do_tile(t + 1, tl1);
        }
        if (t < ntiles) do_tile(t, tl0);
    }

    const float l_tot = l_run + swap_half(l_run);
    const float inv_l = (v_scale ? 1.0f / *v_scale : 1.0f) / l_tot;
    float* out_b = out + (size_t)b * Cv * Nq;
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = cb * 32 + acc_row_base(r) + 4 * h;
            if (ch < Cv) out_b[(size_t)ch * Nq + i_lane] = o[cb][r] * inv_l;
        }
    if (h == 0) lse[(size_t)b * Nq + i_lane] = (m_run + log2f(l_tot) - kBxPBias) * kLn2;
}

// One launch holds both flavours of the body and picks one, workgroup-uniformly, from the device-side mask of V's lo plane
// (cocos_split_f16_chan_mask) as its first action.  DUAL = false: no mask / a single value block.
template <int CVB, bool DUAL, bool CHUNKED>
__global__ __launch_bounds__(256, 1) void box3_sw_fwd_kernel(
    const float* __restrict__ T, const float* __restrict__ mu_q, const float* __restrict__ a_q,
    const float* __restrict__ nu_k, const float* __restrict__ b_k, const _Float16* __restrict__ vh,
    const _Float16* __restrict__ vl, float* __restrict__ out, float* __restrict__ lse,
    const float* __restrict__ v_scale, const unsigned* __restrict__ v_lo_mask, int B, int Nq, int Nk, int Cv, int himg,
    int wimg, float kc, float scale, int flags) {
    if (DUAL && (__builtin_amdgcn_readfirstlane(*v_lo_mask) & ~1u) == 0u)
        box3_sw_fwd_body<CVB, DUAL, CHUNKED>(T, mu_q, a_q, nu_k, b_k, vh, vl, out, lse, v_scale, B, Nq, Nk, Cv, himg, wimg, kc, scale, flags);
    else
        box3_sw_fwd_body<CVB, false, CHUNKED>(T, mu_q, a_q, nu_k, b_k, vh, vl, out, lse, v_scale, B, Nq, Nk, Cv, himg, wimg, kc, scale, flags);
}

// ---------------------------------------------------------------------------------------------------------
// K19 backward
// ---------------------------------------------------------------------------------------------------------
// Column sums (over queries) of two 32 x 32 accumulator images per tile — d b and d nu are sums over ALL queries of a key.
// The four waves of a workgroup hold the same keys for four different query blocks: every wave parks its two images in
// its own LDS slot (8 ds_write_b128), and behind the tile's barrier every wave adds up ONE register group (4 of the 16
// registers) of the four slots (8 ds_read_b128) and reduces it over its 32 lanes: two reduce-scatter folds over lane bits
// 0 and 1 and an all-reduce over bits 2..4, all but the last on the DPP path.  (Round-3 history: every wave running the full
// 16-register butterfly for both quantities cost 0.083 ms of the 0.46 ms kernel; ds_add_f32 into one shared image — 32 LDS
// float atomics per wave and tile — cost 1.3 ms.)
// Exchange with lane c ^ MASK inside a row of 16 lanes, on the DPP path (no LDS crossbar, no address register): 1, 2 are
// quad permutations; 4 = row_half_mirror (c ^ 7) followed by the quad reversal (c ^ 3); 8 = a row rotation by 8.
template <int MASK>
__device__ __forceinline__ float bx_xchg(float v) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    if (MASK == 1) r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);            // quad_perm [1,0,3,2]
    else if (MASK == 2) r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
    else if (MASK == 4)
        r = __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false), 0x1B, 0xf, 0xf, false);
    else r = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);                     // row_ror:8
    return __builtin_bit_cast(float, r);
}
// x[0..3] = registers 4w..4w+3 of the summed image -> the column sum of register 4w + 2*b0 + b1 (b_k = bit k of the lane)
__device__ __forceinline__ float bx_colsum4(const float (&x)[4], int c) {
    const bool up0 = (c & 1) != 0, up1 = (c & 2) != 0;
    const float y0 = (up0 ? x[2] : x[0]) + bx_xchg<1>(up0 ? x[0] : x[2]);
    const float y1 = (up0 ? x[3] : x[1]) + bx_xchg<1>(up0 ? x[1] : x[3]);
    float z = (up1 ? y1 : y0) + bx_xchg<2>(up1 ? y0 : y1);
    z += bx_xchg<4>(z);
    z += bx_xchg<8>(z);
    return z + __shfl_xor(z, 16, 64);
}

#define COCOS_BXB_PARAMS \
    const float* __restrict__ T, const float* __restrict__ mu_q, const float* __restrict__ a_q,                        \
    const float* __restrict__ nu_k, const float* __restrict__ b_k, const _Float16* __restrict__ vph,                   \
    const _Float16* __restrict__ vpl, const _Float16* __restrict__ gph, const _Float16* __restrict__ gpl,              \
    const float* __restrict__ g_scale, const float* __restrict__ v_scale, const float* __restrict__ outp,              \
    const float* __restrict__ dout, const float* __restrict__ lse, float* __restrict__ G, float* __restrict__ dmu,     \
    float* __restrict__ da, float* __restrict__ colpart, float* __restrict__ gmax, _Float16* __restrict__ psh,         \
    _Float16* __restrict__ psl, int B, int Nq, int Nk, int Cv, int himg, int wimg, float kc, float scale,              \
    const float* __restrict__ d_pre, int flags
#define COCOS_BXB_ARGS \
    T, mu_q, a_q, nu_k, b_k, vph, vpl, gph, gpl, g_scale, v_scale, outp, dout, lse, G, dmu, da, colpart, gmax, psh, psl, B, Nq, \
    Nk, Cv, himg, wimg, kc, scale, d_pre, flags

template <int CVB, bool STORE_P, bool VLO0, bool CHUNKED>
__device__ __forceinline__ void box3_sw_bwd_body(COCOS_BXB_PARAMS) {
    constexpr int CVP = CVB * 32, CVS = CVP / 16, VROW = CVP + 8, VPLANE = 32 * VROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char bx_smem[];
    _Float16* const vt = reinterpret_cast<_Float16*>(bx_smem);          // [2 buf][hi|lo][32 keys][VROW]
    float* const colacc = reinterpret_cast<float*>(vt + 2 * 2 * VPLANE);   // [2 buf][4 waves][2 quantities][4 groups][64 lanes][4]
    float* const kstat = colacc + 2 * 4 * 2048;                            // [b | kn][Nk]

    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform BY CONSTRUCTION for the compiler too (round 6): derived from threadIdx it lived in a VGPR, every T / G block offset
    // that depends on it (qblk, py, blk) was VGPR arithmetic, and every block load / store sat in a waterfall loop (readfirstlane /
    // compare / saveexec / branch: 12 loops per tile in K19, 56 in K20)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int nqb = Nq / 128;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, wg = vb % nqb, q0 = wg * 128;
    const int i_lane = q0 + wave * 32 + c;
    const float* const bk_b = b_k + (size_t)b * Nk;
    const float* const nu_b = nu_k + (size_t)b * Nk;
    bx_stage_stats(kstat, bk_b, nu_b, CHUNKED ? BX_KCH : Nk, CHUNKED ? BX_KCH : Nk, kc, tid);

    const size_t vbytes = (size_t)Nk * CVP * 2, gbytes = (size_t)Nq * CVP * 2;
    const __amdgpu_buffer_rsrc_t vh_rs = make_rsrc(vph + (size_t)b * Nk * CVP, vbytes);
    const __amdgpu_buffer_rsrc_t vl_rs = make_rsrc(vpl + (size_t)b * Nk * CVP, vbytes);
    const __amdgpu_buffer_rsrc_t gh_rs = make_rsrc(gph + (size_t)b * Nq * CVP, gbytes);
    const __amdgpu_buffer_rsrc_t gl_rs = make_rsrc(gpl + (size_t)b * Nq * CVP, gbytes);
    const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(outp + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    const __amdgpu_buffer_rsrc_t go_rs = make_rsrc(dout + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    const __amdgpu_buffer_rsrc_t G_rs = make_rsrc(G + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const __amdgpu_buffer_rsrc_t ph_rs = make_rsrc(STORE_P ? psh + (size_t)b * Nk * Nq : nullptr, STORE_P ? (size_t)Nk * Nq * 2 : 0);
    const __amdgpu_buffer_rsrc_t pl_rs = make_rsrc(STORE_P ? psl + (size_t)b * Nk * Nq : nullptr, STORE_P ? (size_t)Nk * Nq * 2 : 0);
    BxGeom gm;
    gm.t_rs = make_rsrc(T + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    gm.kstat = kstat;
    gm.nk = CHUNKED ? BX_KCH : Nk;
    gm.tpr = wimg / 32;
    gm.tpr_log2 = gm.tpr == 4 ? 2 : 1;          // (64- or 128-wide grids: cocos_box3_fused_supported)
    gm.qblk = (q0 >> 5) + wave;
    gm.py = gm.qblk >> gm.tpr_log2;
    gm.himg = himg;
    gm.nqblk = Nq >> 5;
    gm.lane_off = (unsigned)lane * 16u;
    gm.xt = (flags & BX_FLAG_TRANSPOSED) ? kstat + (CHUNKED ? 4 * BX_KCH : 2 * Nk) + wave * BX_XT_FLOATS : nullptr;
    const bool accumulate = (flags & BX_FLAG_ACCUMULATE) != 0;

    const float mu_p = mu_q[(size_t)b * Nq + i_lane];
    const float a_p = a_q[(size_t)b * Nq + i_lane];
    const float a_n = a_p * scale;                     // natural-domain factor: z = tt * a_n
    const float a2 = a_n * kLog2e;

    const float s_o = *g_scale * (v_scale ? *v_scale : 1.0f);      // scale of dP' = V' . dO'
    f16x8 goh[CVS], gol[CVS];
    {
        const unsigned off = (unsigned)(i_lane * CVP + h * 8) * 2u;
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
    if (d_pre) {               // round 4: D from cocos_rowdot_f64 (a streaming kernel) instead of the serial loop below
        d_lane = d_pre[(size_t)b * Nq + i_lane] * s_o;
    } else {
        double dacc = 0.0;
        for (int ch = h; ch < Cv; ch += 2) {
            const unsigned off = (unsigned)(ch * Nq + i_lane) * 4u;
            dacc += (double)buf_load1(go_rs, off) * (double)buf_load1(o_rs, off);
        }
        const int lo = __shfl_xor((int)__double2loint(dacc), 32, 64);
        const int hi = __shfl_xor((int)__double2hiint(dacc), 32, 64);
        d_lane = (float)((dacc + __hiloint2double(hi, lo)) * (double)s_o);
    }
    const float undo = 1.0f / s_o;
    // (round 6: s_o is a power of two — the two operand scales are — so the 1 / s_o of L = P (dP' - D') / s_o rides in the exponent of
    //  P instead of a multiply per element; only the flavour that also stores the planes of P keeps the plain P)
    const float lse2 = lse[(size_t)b * Nq + i_lane] * kLog2e + (STORE_P ? 0.f : log2f(s_o));

    constexpr int VCH = 32 * CVP / 8, VPT = (VCH + 255) / 256;
    u32x4 vst[2][VPT];
    // staging addresses: per lane once, the tile in the scalar offset (see the forward); Nk % 32 == 0, look-ahead clamped to the last tile
    unsigned v_voff[2][VPT];
    int v_lds[VPT];
    bool v_ok[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int g = u * 256 + tid, key = g / (CVP / 8), cc = g % (CVP / 8);
        v_ok[u] = g < VCH;
        v_voff[0][u] = v_ok[u] ? (unsigned)(key * CVP + cc * 8) * 2u : kBufOob;
        // (VLO0: the lo plane of channels >= 32 is all zero and never read: not fetched — its LDS image stays whatever it was)
        v_voff[1][u] = (v_ok[u] && !(VLO0 && cc >= 4)) ? v_voff[0][u] : kBufOob;
        v_lds[u] = key * VROW + cc * 8;
    }
    auto fetch_v_piece = [&](int i, int j0) {
        const int pl = i & 1, u = i >> 1;
        const int jc = min(j0, Nk - 32);
        vst[pl][u] = __builtin_amdgcn_raw_buffer_load_b128(pl ? vl_rs : vh_rs, (int)v_voff[pl][u], (int)((unsigned)(jc * CVP) * 2u), 0);
    };
    auto commit_v_piece = [&](int i, int buf) {
        const int pl = i & 1, u = i >> 1;
        if (v_ok[u]) *reinterpret_cast<u32x4*>(vt + (buf * 2 + pl) * VPLANE + v_lds[u]) = vst[pl][u];
    };
    // behind the barrier that ended tile t: register group `wave` of the four waves' images, summed -> column sums -> colpart
    float* const cp_b = colpart + ((size_t)b * nqb + wg) * 2 * Nk;
    auto reduce_cols = [&](int t) {
        if (BX_ABLATE & 1) return;
        const float* a = colacc + (t & 1) * 4 * 2048 + wave * 256 + lane * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(a + q * 1024);
#pragma unroll
            for (int w = 1; w < 4; ++w) acc += *reinterpret_cast<const f32x4*>(a + w * 2048 + q * 1024);
            const float x[4] = {acc[0], acc[1], acc[2], acc[3]};
            const float z = bx_colsum4(x, c);
            if (c < 4) cp_b[(size_t)q * Nk + t * 32 + 8 * wave + 4 * h + 2 * (c & 1) + ((c >> 1) & 1)] = z;
        }
    };

    const int ntiles = Nk / 32;
    BxTile tl0, tl1;        // (two tiles of look-ahead, as in the forward)
#pragma unroll
    for (int i = 0; i < 2 * VPT; ++i) fetch_v_piece(i, 0);
    bx_fetch(tl0, gm, 0, ntiles);
    bx_fetch(tl1, gm, 1, ntiles);
#pragma unroll
    for (int i = 0; i < 2 * VPT; ++i) commit_v_piece(i, 0);
#pragma unroll
    for (int i = 0; i < 2 * VPT; ++i) fetch_v_piece(i, 32);
    __syncthreads();

    float r2 = 0.f, rm = 0.f, gabs = 0.f;
    auto do_tile = [&](int t, BxTile& tl) __attribute__((always_inline)) {
        const int j0 = t * 32, buf = t & 1;
        if (t > 0) reduce_cols(t - 1);
        bx_stage_next_chunk<CHUNKED>(kstat, bk_b, nu_b, t, ntiles, kc, tid);
        float tt[16], bq[16], kn[16];
        bx_logits<CHUNKED>(tl, gm, t, h, mu_p, tt, bq, kn, c);
        bx_fetch(tl, gm, t + 2, ntiles);
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = (BX_ABLATE & 32) ? tt[r] * 1e-3f : fast_exp2(__builtin_fmaf(tt[r], a2, -lse2));

        // ---- dP' = V(t) . dO' ---------------------------------------------------------------------------------------
        f32x16 dp0;          // ONE accumulator for the three terms (round 6, as corr_fused_fwd_f16x3.hip's QK1: the kernel is VALU-bound)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp0[r] = -d_lane;       // ... which starts at -D': L = P (dP' - D') without a subtraction per element
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
                    if (!VLO0 || u + 1 < 2) al[nxt] = *reinterpret_cast<const f16x8*>(vb0 + VPLANE + (u + 1) * 16);
                }
                if (!(BX_ABLATE & 8)) {
                    dp0 = bx_mfma(ah[cur], goh[u], dp0);
                    dp0 = bx_mfma(ah[cur], gol[u], dp0);
                    if (!VLO0 || u < 2) dp0 = bx_mfma(al[cur], goh[u], dp0);
                }
                if (u < 2 * VPT) {
                    commit_v_piece(u, buf ^ 1);
                    fetch_v_piece(u, j0 + 64);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_assert(2 * VPT <= CVS, "staging pieces must fit the dP steps");

        // ---- L = P (dP - D); G = L * a_n * b_q; row sums in registers, column sums through the butterfly -----------------
        float x1[16], x2[16], gv[16];
        f32x4 gout[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // La = L a_n (a_n = scale a_p > 0 folded once): G = La b_q, L z = La tt, L a_n mu_p = La mu_p, sum_q L kn_q = (sum_q La kn_q) / a_n
            const float dd = dp0[r];
            const float La = (STORE_P ? p[r] * undo : p[r]) * dd * a_n;
            gv[r] = La * bq[r];
            x1[r] = La * tt[r];
            x2[r] = La * mu_p;
            r2 += x1[r];
            rm = __builtin_fmaf(La, kn[r], rm);
        }
        // G leaves in the layout of the STORED T: transposed back when T was read transposed, and added to what another pass
        // has already written (one G for all passes over this T: one box adjoint + one pair of GEMMs)
        if (gm.xt) bx_transpose32(gv, gm.xt, h, c);
        const unsigned blk = (unsigned)((gm.xt ? gm.qblk * gm.nqblk + t : t * gm.nqblk + gm.qblk) * 4096);
        if (accumulate && !(BX_ABLATE & 2)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 old = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(G_rs, (int)gm.lane_off,
                                                                                                 (int)(blk + (unsigned)g * 1024u), 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[4 * g + e] += old[e];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gout[r >> 2][r & 3] = gv[r];
            gabs = fmaxf(gabs, fabsf(gv[r]));
        }
        if (!(BX_ABLATE & 2)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gout[g]), G_rs, (int)gm.lane_off,
                                                       (int)(blk + (unsigned)g * 1024u), 2);      // nt: a stream for K20
            // (a 16-byte store whose soffset is a REGISTER: LLVM models no write-after-read hazard on its data registers,
            //  gfx950 has one — corr_fused_fwd_f16x3.hip, round 4.  Keep the four quads untouched for a few cycles.)
            asm volatile("s_nop 4" : : "v"(gout[0]), "v"(gout[1]), "v"(gout[2]), "v"(gout[3]));
        }
        if (STORE_P && !(BX_ABLATE & 2)) {
            // planes of 2^14 P in the accumulator's own orientation: [Nq/32][Nk/32] blocks of 2 x [32 queries][16 keys]
            // (corr_fused_bwd_f16x3.hip, store_regs_blk; read by hgemm_f16x3 with b_blocked = 2)
            unsigned hw[8], lw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                split_pair_rtz(p[2 * j] * kBxPPlaneScale, p[2 * j + 1] * kBxPPlaneScale, hw[j], lw[j]);
            const unsigned blk = (unsigned)((gm.qblk * (Nk >> 5) + t) * 2048);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int m0 = 2 * pp, m1 = 2 * pp + 1;
                u32x4 xh, xl;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const auto sh = __builtin_amdgcn_permlane32_swap(hw[2 * m0 + i], hw[2 * m1 + i], false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(lw[2 * m0 + i], lw[2 * m1 + i], false, false);
                    xh[i] = sh[0]; xh[2 + i] = sh[1];
                    xl[i] = sl[0]; xl[2 + i] = sl[1];
                }
                const unsigned off = blk + (unsigned)(pp * 1024 + c * 32 + h * 16);
                __builtin_amdgcn_raw_buffer_store_b128(xh, ph_rs, (int)off, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(xl, pl_rs, (int)off, 0, 2);
            }
        }
        if (!(BX_ABLATE & 1)) {
            // (slot of tile t - 2 was read by every wave before it passed the barrier of tile t - 1)
            float* a = colacc + buf * 4 * 2048 + wave * 2048 + lane * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(a + g * 256) = f32x4{x1[4 * g], x1[4 * g + 1], x1[4 * g + 2], x1[4 * g + 3]};
                *reinterpret_cast<f32x4*>(a + 1024 + g * 256) = f32x4{x2[4 * g], x2[4 * g + 1], x2[4 * g + 2], x2[4 * g + 3]};
            }
        }
        __syncthreads();
    };
    {
        int t = 0;
        for (; t + 1 < ntiles; t += 2) {
            do_tile(t, tl0);
            do_tile(t + 1, tl1);
        }
        if (t < ntiles) do_tile(t, tl0);
    }
    reduce_cols(ntiles - 1);

    r2 += swap_half(r2);
    rm += swap_half(rm);
    if (h == 0) {
        da[(size_t)b * Nq + i_lane] = r2 / a_p;
        dmu[(size_t)b * Nq + i_lane] = -rm;          // (rm carries the factor a_n already)
    }
    gabs = wave_max_dpp(gabs);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(gmax), __builtin_bit_cast(unsigned, gabs));   // >= 0: ordered as integers
}

template <int CVB, bool STORE_P, bool DUAL, bool CHUNKED>
__global__ __launch_bounds__(256, 1) void box3_sw_bwd_kernel(COCOS_BXB_PARAMS, const unsigned* __restrict__ v_lo_mask) {
    if (DUAL && (__builtin_amdgcn_readfirstlane(*v_lo_mask) & ~1u) == 0u)
        box3_sw_bwd_body<CVB, STORE_P, DUAL, CHUNKED>(COCOS_BXB_ARGS);
    else
        box3_sw_bwd_body<CVB, STORE_P, false, CHUNKED>(COCOS_BXB_ARGS);
}

// d nu[b,q] = -kc * b_q * sum_wg colpart[b,wg,1,q];   d b[b,q] = sum_wg colpart[b,wg,0,q] / b_q
__global__ void box3_col_reduce_kernel(const float* __restrict__ colpart, const float* __restrict__ b_k,
                                       float* __restrict__ dnu, float* __restrict__ db, int B, int nwg, int Nk, float kc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Nk) return;
    const int b = i / Nk, q = i % Nk;
    float c2 = 0.f, cm = 0.f;
    for (int w = 0; w < nwg; ++w) {
        const float* p = colpart + ((size_t)b * nwg + w) * 2 * Nk;
        c2 += p[q];
        cm += p[Nk + q];
    }
    const float bq = b_k[i];
    dnu[i] = -kc * bq * cm;
    db[i] = c2 / bq;
}

// ---------------------------------------------------------------------------------------------------------
// K20: dC = xbox(ybox(G)) -> f16 hi/lo planes in the [Nq/32][Nk/32] blocks of 2 x [32 queries][16 keys] both
// correlation-gradient GEMMs read (hgemm_f16x3.hip modes 2 and 3).  One wave per (key image row, query image row).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void box3_adjoint_planes_kernel(const float* __restrict__ G,
                                                                     const float* __restrict__ gmax,
                                                                     _Float16* __restrict__ dch, _Float16* __restrict__ dcl,
                                                                     float* __restrict__ scale_out, int B, int Nq, int Nk,
                                                                     int himg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bx_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform BY CONSTRUCTION for the compiler too (round 6): derived from threadIdx it lived in a VGPR, every T / G block offset
    // that depends on it (qblk, py, blk) was VGPR arithmetic, and every block load / store sat in a waterfall loop (readfirstlane /
    // compare / saveexec / branch: 12 loops per tile in K19, 56 in K20)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    float* const img = reinterpret_cast<float*>(bx_smem) + wave * kXbFloats;
    xbox_zero_border(img, lane);

    // (the diagonal-major order of the 128-wide kernel below was measured here too: 0.21 -> 0.25 ms at cfg2' — a 64 x 64
    //  grid's G (67 MB per sample) is served by the memory-side cache either way, and row-major keeps the STORES sequential)
    // (round 6: consecutive workgroups on one XCD — a sample per XCD at B = 8, so that the three readers of a G block share an L2: PMC says
    //  784 MB are read for a 537 MB G — was neutral as well: 1.750-1.754 vs 1.752-1.754 ms for the step; the kernel is at the memory side's rate)
    const int grp = blockIdx.x * 4 + wave;                     // (b, ky, py)
    const int per = himg * himg;
    const int b = grp / per, ky = (grp % per) / himg, py = grp % himg;
    if (b >= B) return;
    const int nqblk = Nq >> 5;
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(G + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const __amdgpu_buffer_rsrc_t h_rs = make_rsrc(dch + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 2);
    const __amdgpu_buffer_rsrc_t l_rs = make_rsrc(dcl + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 2);

    // power-of-two scale: max|G| -> [2^6, 2^7), so |dC| <= 9 max|G| stays below 2^11
    float s = 1.0f;
    {
        const float m = *gmax;
        if (m > 0.f && m < INFINITY) {
            int e;
            frexpf(m, &e);
            s = ldexpf(1.0f, min(max(7 - e, -100), 100));
        }
        if (blockIdx.x == 0 && tid == 0) *scale_out = s;
    }

    f32x16 t[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) t[kt][qt][r] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int dy = d - 1;
        const bool ok = (unsigned)(py + dy) < (unsigned)himg && (unsigned)(ky + dy) < (unsigned)himg;
        f32x4 ld[2][2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int blk = ok ? ((ky + dy) * 2 + kt) * nqblk + (py + dy) * 2 + qt : 0;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    ld[kt][qt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        g_rs, (int)(ok ? (unsigned)lane * 16u : kBufOob), (int)((unsigned)blk * 4096u + (unsigned)g * 1024u), 0));
            }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[kt][qt][r] += ld[kt][qt][r >> 2][r & 3];
    }
    xbox_64x64(t, img, lane);

#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            unsigned hw[8], lw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) split_pair_rtz(t[kt][qt][2 * j] * s, t[kt][qt][2 * j + 1] * s, hw[j], lw[j]);
            const unsigned blk = (unsigned)(((py * 2 + qt) * (Nk >> 5) + ky * 2 + kt) * 2048);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int m0 = 2 * pp, m1 = 2 * pp + 1;
                u32x4 xh, xl;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const auto sh = __builtin_amdgcn_permlane32_swap(hw[2 * m0 + i], hw[2 * m1 + i], false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(lw[2 * m0 + i], lw[2 * m1 + i], false, false);
                    xh[i] = sh[0]; xh[2 + i] = sh[1];
                    xl[i] = sl[0]; xl[2 + i] = sl[1];
                }
                const unsigned off = blk + (unsigned)(pp * 1024 + c * 32 + h * 16);
                __builtin_amdgcn_raw_buffer_store_b128(xh, h_rs, (int)off, 0, 2);      // nt: streams for the two GEMMs
                __builtin_amdgcn_raw_buffer_store_b128(xl, l_rs, (int)off, 0, 2);
            }
        }
}

// K20 on a 128-wide grid: a (key image row, query image row) pair is 128 x 128 = FOUR 64 x 64 chunks; one workgroup per
// pair, wave = chunk (kq = key half, qq = query half).  Every chunk's image gets its inward-facing border cells from the
// three other waves (box3_common.h), with one workgroup barrier between the writes and the reads.
// (two workgroups per CU: 2 x 80 KB of LDS; the barriers of one hide behind the other's loads)
__global__ __launch_bounds__(256, 2) void box3_adjoint_planes_w128_kernel(const float* __restrict__ G,
                                                                          const float* __restrict__ gmax,
                                                                          _Float16* __restrict__ dch, _Float16* __restrict__ dcl,
                                                                          float* __restrict__ scale_out, int B, int Nq, int Nk,
                                                                          int himg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bx_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform BY CONSTRUCTION for the compiler too (round 6): derived from threadIdx it lived in a VGPR, every T / G block offset
    // that depends on it (qblk, py, blk) was VGPR arithmetic, and every block load / store sat in a waterfall loop (readfirstlane /
    // compare / saveexec / branch: 12 loops per tile in K19, 56 in K20)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    float* const img0 = reinterpret_cast<float*>(bx_smem);
    float* const img = img0 + wave * kXbFloats;
    xbox_zero_border(img, lane);

    // (b, ky, py): the grid is exactly B * himg * himg.  DIAGONAL-major order on XCD-contiguous virtual ids: block (r, s) of G is read
    // by the three row pairs (r + d, s + d), d = -1, 0, 1 — neighbours along a (wrapped) diagonal ky - py = const; enumerated that
    // way they run back to back on ONE XCD and the second and third read of a block are L2 hits (at HW = 16384 G is 1 GiB per
    // sample: row-major order re-read it from HBM)
    const int grp = xcd_remap(blockIdx.x, gridDim.x);
    const int per = himg * himg;
    const int b = grp / per, dg = (grp % per) / himg, py = grp % himg;
    const int ky = (py + dg) % himg;
    const int kq = wave >> 1, qq = wave & 1;
    constexpr int TPR = 4;                                     // 32-position tiles per image row
    const int nqblk = Nq >> 5;
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(G + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 4);
    const __amdgpu_buffer_rsrc_t h_rs = make_rsrc(dch + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 2);
    const __amdgpu_buffer_rsrc_t l_rs = make_rsrc(dcl + (size_t)b * Nk * Nq, (size_t)Nk * Nq * 2);

    float s = 1.0f;      // power-of-two scale: max|G| -> [2^6, 2^7), so |dC| <= 9 max|G| stays below 2^11
    {
        const float m = *gmax;
        if (m > 0.f && m < INFINITY) {
            int e;
            frexpf(m, &e);
            s = ldexpf(1.0f, min(max(7 - e, -100), 100));
        }
        if (blockIdx.x == 0 && tid == 0) *scale_out = s;
    }

    f32x16 t[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) t[kt][qt][r] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {                              // y box: three diagonal neighbours, whole tiles apart
        const int dy = d - 1;
        const bool ok = (unsigned)(py + dy) < (unsigned)himg && (unsigned)(ky + dy) < (unsigned)himg;
        f32x4 ld[2][2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int blk = ok ? ((ky + dy) * TPR + kq * 2 + kt) * nqblk + (py + dy) * TPR + qq * 2 + qt : 0;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    ld[kt][qt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        g_rs, (int)(ok ? (unsigned)lane * 16u : kBufOob), (int)((unsigned)blk * 4096u + (unsigned)g * 1024u), 0));
            }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[kt][qt][r] += ld[kt][qt][r >> 2][r & 3];
    }
    __syncthreads();                                            // every image's border is zero before the neighbours write into it
    xbox_write_chunk(t, img, lane);
    float* const img_k = img0 + (wave ^ 2) * kXbFloats;        // the other key half, same query half
    float* const img_q = img0 + (wave ^ 1) * kXbFloats;        // same key half, the other query half
    float* const img_d = img0 + (wave ^ 3) * kXbFloats;        // the diagonal chunk
    if (kq == 0) xbox_put_last_key(t, img_k, lane); else xbox_put_first_key(t, img_k, lane);
    if (qq == 0) xbox_put_last_query(t, img_q, lane); else xbox_put_first_query(t, img_q, lane);
    xbox_put_corner(t, img_d, lane, kq == 0, qq == 0);
    __syncthreads();
    xbox_add_diagonals(t, img, lane);

#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            unsigned hw[8], lw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) split_pair_rtz(t[kt][qt][2 * j] * s, t[kt][qt][2 * j + 1] * s, hw[j], lw[j]);
            const unsigned blk = (unsigned)(((py * TPR + qq * 2 + qt) * (Nk >> 5) + ky * TPR + kq * 2 + kt) * 2048);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int m0 = 2 * pp, m1 = 2 * pp + 1;
                u32x4 xh, xl;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const auto sh = __builtin_amdgcn_permlane32_swap(hw[2 * m0 + i], hw[2 * m1 + i], false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(lw[2 * m0 + i], lw[2 * m1 + i], false, false);
                    xh[i] = sh[0]; xh[2 + i] = sh[1];
                    xl[i] = sl[0]; xl[2 + i] = sl[1];
                }
                const unsigned off = blk + (unsigned)(pp * 1024 + c * 32 + h * 16);
                __builtin_amdgcn_raw_buffer_store_b128(xh, h_rs, (int)off, 0, 2);      // nt: streams for the two GEMMs
                __builtin_amdgcn_raw_buffer_store_b128(xl, l_rs, (int)off, 0, 2);
            }
        }
}

template <int CVB>
static int bx_fwd_launch(const float* T, const float* mu, const float* a, const float* nu, const float* bk,
                         const _Float16* vh, const _Float16* vl, float* out, float* lse, const float* vs,
                         const unsigned* mask, int B, int Nq, int Nk, int Cv, int himg, int wimg, float kc, float scale,
                         int flags, hipStream_t s) {
    const bool chunked = Nk > 2 * BX_KCH;
    const size_t smem = (size_t)2 * 3 * CVB * 32 * BX_VROW * sizeof(_Float16) + (size_t)(chunked ? 4 * BX_KCH : 2 * Nk) * sizeof(float) +
                        ((flags & BX_FLAG_TRANSPOSED) ? (size_t)4 * BX_XT_FLOATS * sizeof(float) : 0);
    auto kern = chunked ? ((CVB > 1 && mask) ? box3_sw_fwd_kernel<CVB, (CVB > 1), true> : box3_sw_fwd_kernel<CVB, false, true>)
                        : ((CVB > 1 && mask) ? box3_sw_fwd_kernel<CVB, (CVB > 1), false> : box3_sw_fwd_kernel<CVB, false, false>);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(B * (Nq / 128)), dim3(256), smem, s, T, mu, a, nu, bk, vh, vl, out, lse, vs, mask, B, Nq, Nk,
                       Cv, himg, wimg, kc, scale, flags);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

template <int CVB>
static int bx_bwd_launch(const float* T, const float* mu, const float* a, const float* nu, const float* bk,
                         const _Float16* vph, const _Float16* vpl, const _Float16* gph, const _Float16* gpl,
                         const float* gs, const float* vs, const float* outp, const float* dout, const float* lse, float* G,
                         float* dmu, float* da, float* colpart, float* gmax, _Float16* psh, _Float16* psl,
                         const unsigned* mask, int B, int Nq, int Nk, int Cv, int himg, int wimg, float kc, float scale,
                         const float* d_pre, int flags, hipStream_t s) {
    const bool chunked = Nk > 2 * BX_KCH;
    const size_t smem = (size_t)2 * 2 * 32 * (CVB * 32 + 8) * sizeof(_Float16) +
                        (size_t)(2 * 4 * 2048 + (chunked ? 4 * BX_KCH : 2 * Nk)) * sizeof(float) +
                        ((flags & BX_FLAG_TRANSPOSED) ? (size_t)4 * BX_XT_FLOATS * sizeof(float) : 0);
#define COCOS_BX_GO(SP, CH)                                                                                              \
    do {                                                                                                                 \
        auto kern = (CVB > 1 && mask) ? box3_sw_bwd_kernel<CVB, SP, (CVB > 1), CH> : box3_sw_bwd_kernel<CVB, SP, false, CH>; \
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(kern, dim3(B * (Nq / 128)), dim3(256), smem, s, T, mu, a, nu, bk, vph, vpl, gph, gpl, gs, vs, outp, \
                           dout, lse, G, dmu, da, colpart, gmax, psh, psl, B, Nq, Nk, Cv, himg, wimg, kc, scale, d_pre, flags, mask); \
    } while (0)
    if (chunked) { if (psh) COCOS_BX_GO(true, true); else COCOS_BX_GO(false, true); }
    else { if (psh) COCOS_BX_GO(true, false); else COCOS_BX_GO(false, false); }
#undef COCOS_BX_GO
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

static bool bx_shape_ok(int Nq, int Nk, int Cv, int himg, int wimg) {
    // (T / G of one sample below 2 GiB: 16384 x 16384 fp32 is 1 GiB)
    return himg >= 1 && (wimg == 64 || wimg == 128) && Nq == himg * wimg && Nk == Nq && Nq % 256 == 0 &&
           Cv >= 1 && Cv <= 160 && (size_t)Nq * Nk * 4 < 0x7fffffffull;
}

}  // namespace cocos

extern "C" int cocos_box3_fused_supported(int Nq, int Nk, int Cv, int grid_h, int grid_w) {
    return cocos::bx_shape_ok(Nq, Nk, Cv, grid_h, grid_w) ? 1 : 0;
}

extern "C" int cocos_box3_softmax_warp_fwd_f16x3(const float* t_blocked, const float* mu_q, const float* a_q,
                                                 const float* nu_k, const float* b_k, const void* vh, const void* vl,
                                                 float* out, float* lse, const float* v_scale_dev,
                                                 const unsigned* v_lo_mask_dev, int B, int Nq, int Nk,
                                                 int Cv, int grid_h, int grid_w, float k_unfolded, float scale, int flags,
                                                 cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(t_blocked && mu_q && a_q && nu_k && b_k && vh && vl && out && lse, COCOS_ERR_INVALID,
                  "box3_softmax_warp_fwd_f16x3: null pointer");
    COCOS_REQUIRE((flags & ~BX_FLAG_TRANSPOSED) == 0, COCOS_ERR_INVALID, "box3_softmax_warp_fwd_f16x3: flags=%d (bit 0: T transposed)", flags);
    COCOS_REQUIRE(B >= 1 && scale > 0.f, COCOS_ERR_INVALID, "box3_softmax_warp_fwd_f16x3: bad B=%d / scale", B);
    COCOS_REQUIRE(bx_shape_ok(Nq, Nk, Cv, grid_h, grid_w), COCOS_ERR_UNSUPPORTED,
                  "box3_softmax_warp_fwd_f16x3: needs a 64- or 128-wide grid with Nq == Nk == h*w, Nq %% 256 == 0, Cv <= 160 "
                  "(Nq=%d Nk=%d Cv=%d grid %dx%d)", Nq, Nk, Cv, grid_h, grid_w);
    COCOS_REQUIRE(aligned16(t_blocked) && aligned16(nu_k) && aligned16(b_k), COCOS_ERR_INVALID,
                  "box3_softmax_warp_fwd_f16x3: T / nu / b must be 16-byte aligned");
    for (const void* p : {vh, vl})
        COCOS_REQUIRE((reinterpret_cast<uintptr_t>(p) & 7u) == 0, COCOS_ERR_INVALID,
                      "box3_softmax_warp_fwd_f16x3: v planes must be 8-byte aligned");
    const _Float16 *a = static_cast<const _Float16*>(vh), *b2 = static_cast<const _Float16*>(vl);
    hipStream_t s = as_stream(stream);
#define COCOS_ARGS t_blocked, mu_q, a_q, nu_k, b_k, a, b2, out, lse, v_scale_dev, v_lo_mask_dev, B, Nq, Nk, Cv, grid_h, grid_w, k_unfolded, scale, flags, s
    switch ((Cv + 31) / 32) {
        case 1: return bx_fwd_launch<1>(COCOS_ARGS);
        case 2: return bx_fwd_launch<2>(COCOS_ARGS);
        case 3: return bx_fwd_launch<3>(COCOS_ARGS);
        case 4: return bx_fwd_launch<4>(COCOS_ARGS);
        default: return bx_fwd_launch<5>(COCOS_ARGS);
    }
#undef COCOS_ARGS
}

extern "C" size_t cocos_box3_softmax_warp_bwd_colpart_bytes(int B, int Nq, int Nk) {
    if (B < 1 || Nq < 128 || Nk < 1) return 0;
    return (size_t)B * (Nq / 128) * 2 * Nk * sizeof(float);
}

extern "C" int cocos_box3_softmax_warp_bwd_f16x3(
    const float* t_blocked, const float* mu_q, const float* a_q, const float* nu_k, const float* b_k, const void* vph,
    const void* vpl, const void* gph, const void* gpl, const float* g_scale_dev, const float* v_scale_dev, const float* out,
    const float* dout, const float* lse, float* g_blocked, float* dmu, float* da, float* dnu, float* db, void* colpart,
    float* gmax_dev, void* psh, void* psl, const unsigned* v_lo_mask_dev, int B, int Nq, int Nk, int Cv, int CvPad, int grid_h,
    int grid_w, float k_unfolded, float scale, const float* d_pre, int flags, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE((flags & ~(BX_FLAG_TRANSPOSED | BX_FLAG_ACCUMULATE)) == 0, COCOS_ERR_INVALID,
                  "box3_softmax_warp_bwd_f16x3: flags=%d (bit 0: T transposed, bit 1: add G to g_blocked)", flags);
    COCOS_REQUIRE(t_blocked && mu_q && a_q && nu_k && b_k && vph && vpl && gph && gpl && g_scale_dev && out && dout && lse &&
                      g_blocked && dmu && da && dnu && db && colpart && gmax_dev,
                  COCOS_ERR_INVALID, "box3_softmax_warp_bwd_f16x3: null pointer");
    COCOS_REQUIRE((psh == nullptr) == (psl == nullptr), COCOS_ERR_INVALID,
                  "box3_softmax_warp_bwd_f16x3: P plane pointers come as a hi/lo pair");
    COCOS_REQUIRE(B >= 1 && scale > 0.f, COCOS_ERR_INVALID, "box3_softmax_warp_bwd_f16x3: bad B=%d / scale", B);
    COCOS_REQUIRE(bx_shape_ok(Nq, Nk, Cv, grid_h, grid_w), COCOS_ERR_UNSUPPORTED,
                  "box3_softmax_warp_bwd_f16x3: needs a 64- or 128-wide grid with Nq == Nk == h*w, Nq %% 256 == 0, Cv <= 160 "
                  "(Nq=%d Nk=%d Cv=%d grid %dx%d)", Nq, Nk, Cv, grid_h, grid_w);
    const int cvb = (Cv + 31) / 32;
    COCOS_REQUIRE(CvPad == cvb * 32, COCOS_ERR_INVALID, "box3_softmax_warp_bwd_f16x3: CvPad=%d, expected %d", CvPad, cvb * 32);
    for (const void* p : {(const void*)t_blocked, (const void*)g_blocked, vph, vpl, gph, gpl, (const void*)nu_k, (const void*)b_k})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "box3_softmax_warp_bwd_f16x3: planes, T, G, nu, b must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
#define COCOS_ARGS                                                                                                       \
    t_blocked, mu_q, a_q, nu_k, b_k, static_cast<const _Float16*>(vph), static_cast<const _Float16*>(vpl),               \
        static_cast<const _Float16*>(gph), static_cast<const _Float16*>(gpl), g_scale_dev, v_scale_dev, out, dout, lse,   \
        g_blocked, dmu, da, static_cast<float*>(colpart), gmax_dev, static_cast<_Float16*>(psh),                         \
        static_cast<_Float16*>(psl), v_lo_mask_dev, B, Nq, Nk, Cv, grid_h, grid_w, k_unfolded, scale, d_pre, flags, s
    int rc;
    switch (cvb) {
        case 1: rc = bx_bwd_launch<1>(COCOS_ARGS); break;
        case 2: rc = bx_bwd_launch<2>(COCOS_ARGS); break;
        case 3: rc = bx_bwd_launch<3>(COCOS_ARGS); break;
        case 4: rc = bx_bwd_launch<4>(COCOS_ARGS); break;
        default: rc = bx_bwd_launch<5>(COCOS_ARGS); break;
    }
#undef COCOS_ARGS
    if (rc != COCOS_OK) return rc;
    const int n = B * Nk;
    hipLaunchKernelGGL(box3_col_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<const float*>(colpart),
                       b_k, dnu, db, B, Nq / 128, Nk, k_unfolded);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_box3_adjoint_planes_f16x3(const float* g_blocked, const float* gmax_dev, void* dc_hi, void* dc_lo,
                                               float* scale_out_dev, int B, int Nq, int Nk, int grid_h, int grid_w,
                                               cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(g_blocked && gmax_dev && dc_hi && dc_lo && scale_out_dev, COCOS_ERR_INVALID,
                  "box3_adjoint_planes_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && bx_shape_ok(Nq, Nk, 1, grid_h, grid_w), COCOS_ERR_UNSUPPORTED,
                  "box3_adjoint_planes_f16x3: needs a 64- or 128-wide grid with Nq == Nk == h*w, Nq %% 256 == 0 (Nq=%d Nk=%d grid %dx%d)",
                  Nq, Nk, grid_h, grid_w);
    for (const void* p : {(const void*)g_blocked, (const void*)dc_hi, (const void*)dc_lo})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "box3_adjoint_planes_f16x3: pointers must be 16-byte aligned");
    const long long groups = (long long)B * grid_h * grid_h;
    const size_t smem = (size_t)4 * kXbFloats * sizeof(float);
    COCOS_REQUIRE(groups <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "box3_adjoint_planes_f16x3: grid too large");
    if (grid_w == 128) {            // one workgroup per (key row, query row) pair, wave = 64 x 64 chunk
        auto kw = box3_adjoint_planes_w128_kernel;
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kw, dim3((unsigned)groups), dim3(256), smem, as_stream(stream), g_blocked, gmax_dev,
                           static_cast<_Float16*>(dc_hi), static_cast<_Float16*>(dc_lo), scale_out_dev, B, Nq, Nk, grid_h);
        COCOS_HIP_CHECK(hipGetLastError());
        return COCOS_OK;
    }
    auto kern = box3_adjoint_planes_kernel;
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3((unsigned)((groups + 3) / 4)), dim3(256), smem, as_stream(stream), g_blocked, gmax_dev,
                       static_cast<_Float16*>(dc_hi), static_cast<_Float16*>(dc_lo), scale_out_dev, B, Nq, Nk, grid_h);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
