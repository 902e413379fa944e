// K2 forward: fused correlation -> softmax over key positions -> warp, for gfx950 (MI355X).
//
// Replaces correspondence.py:281 (permute), :291 (matmul), :304 (/temperature), :307 (softmax),
// :318 (matmul with the exemplar) and the further P@V products at :334 — the [B,HW,HW] matrix
// is never written to HBM.
//
// Arithmetic: exact fp32.  The logits are cos/0.01, so a cosine error d becomes a 100*d relative
// error in every softmax weight; bf16/fp16 operands fail the 1e-3 parity bar (SURVEY.md §7 hard
// part 1), therefore the correlation runs on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate).
//
// Decomposition (one workgroup = 4 waves = 128 query positions, 1 wave per SIMD):
//   * each wave owns 32 query positions; its 32 x K query slice lives in registers as the MFMA
//     B operand for the whole kernel (K/2 = 128 registers);
//   * key tiles of 32 positions ([K][32] + the V tile [Cv][32]) stream through a double-buffered
//     LDS ring shared by the 4 waves; global loads run 2-3 tiles ahead in staging registers; the
//     logits of tile t+1 are computed WHILE tile t goes through the softmax (see "pipeline");
//   * the wave computes S^T (32 keys x 32 queries) = K_tile^T . Q  — swapped operands, so that
//     lane&31 indexes the QUERY: the softmax statistics (running max / sum) are one scalar per
//     lane and the key axis runs over the 16 accumulator registers (+ the other half-wave);
//   * the exponentiated accumulator registers are directly the B operand of the P.V MFMAs
//     (see acc_row_base in common.h), A = V^T read from LDS; O^T accumulates in registers with
//     lane&31 = query again, so the online-softmax rescale is a per-lane multiply.
#include "common.h"

namespace cocos {

constexpr int FWD_BQ = 128;          // query positions per workgroup
constexpr int FWD_BK = kTileCols;    // key positions per tile
constexpr int FWD_LD = kTileLd;      // LDS row stride (floats)
constexpr float kRescaleThr = 8.0f;  // log2 units: rescale only when a row max grows by > 2^8

template <int KD, int CVB, bool STORE_S, bool RAGGED>
__global__ __launch_bounds__(256, 1) void corr_softmax_warp_fwd_kernel(
    const float* __restrict__ qn, const float* __restrict__ kn, const float* __restrict__ v,
    float* __restrict__ out, float* __restrict__ lse,
    float* __restrict__ lg /* [B,Nk,Nq] scaled logits for the backward (STORE_S) */, int B, int Nq,
    int Nk, int Cv,
    float scale_log2 /* inv_temperature * log2(e) */) {
    constexpr int CVP = CVB * 32;
    static_assert(KD % 32 == 0, "K must be a multiple of 32");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* kt = smem;                        // [2][KD][FWD_LD]
    float* vt = smem + 2 * KD * FWD_LD;      // [2][CVP][FWD_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;

    const int nqb = (Nq + FWD_BQ - 1) / FWD_BQ;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, qb = vb % nqb;
    const int i_lane = qb * FWD_BQ + wave * 32 + c;   // this lane's query position

    const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(qn + (size_t)b * KD * Nq, (size_t)KD * Nq * 4);
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kn + (size_t)b * KD * Nk, (size_t)KD * Nk * 4);
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(v + (size_t)b * Cv * Nk, (size_t)Cv * Nk * 4);

    // training mode: the scaled logits (log2 domain) go to HBM once, key-major [Nk][Nq], so the
    // query-side backward reads them back instead of recomputing K/2 MFMAs per tile: on MI355X a
    // logit costs 512 fp32-MFMA FLOPs to recompute (3.3 ps at peak) but 8 bytes of HBM traffic to
    // store + reload (1.3 ps at 6 TB/s), and the stores hide under the MFMAs anyway.
    const __amdgpu_buffer_rsrc_t lg_rs = make_rsrc(STORE_S ? lg + (size_t)b * Nk * Nq : nullptr,
                                                   STORE_S ? (size_t)Nk * Nq * 4 : 0);
    const unsigned lg_lane_off = i_lane < Nq ? (unsigned)(4 * h * Nq + i_lane) * 4u : kBufOob;

    // ---- resident query slice: B operand  B[k = 2kk + h][col = query c] -------------------
    // (positions past Nq read 0 through the descriptor; their results are never stored)
    float qreg[KD / 2];
    {
        const unsigned q_off = i_lane < Nq ? (unsigned)(h * Nq + i_lane) * 4u : kBufOob;
#pragma unroll
        for (int kk = 0; kk < KD / 2; ++kk)
            qreg[kk] = buf_load1(q_rs, q_off + (unsigned)(2 * kk * Nq) * 4u);
        // park the slice in the accumulator half of the register file (MFMA B operands may be
        // AGPRs): frees 128 arch VGPRs so the LDS read-ahead below has registers to land in
#pragma unroll
        for (int kk = 0; kk < KD / 2; ++kk) asm volatile("" : "+a"(qreg[kk]));
    }

    f32x16 o[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m_run = -INFINITY;   // running max (log2 domain), per query lane
    float l_run = 0.f;         // running sum, partial over this half-wave's keys

    // ---- pipeline --------------------------------------------------------------------------------
    // LDS holds two key tiles and two V tiles.  Iteration t:
    //     S(t+1) = K(t+1)^T Q        (MFMA, reads kt[(t+1)&1])     \  one instruction stream: the VALU
    //     softmax of S(t) -> P(t)    (VALU, + logits store)         > work and the LDS commits ride in
    //     commit K(t+2) -> kt[t&1], V(t+1) -> vt[(t+1)&1]          /  the gaps between the MFMAs
    //     fetch  K(t+3), V(t+2)      (global -> staging registers, in flight until the next commit)
    //     O += V(t)^T P(t)           (MFMA, reads vt[t&1])
    //     barrier
    // so the matrix pipe never waits for the softmax, and there is ONE barrier per tile.  Tiles past
    // the end are fetched through the descriptors' bounds check (zeros) and multiplied for nothing
    // (one tile per workgroup) — cheaper than a second, non-overlapped code path for the tail.
    TileRegs<KD> ks;
    TileRegs<CVP> vs;
    auto fetch_k = [&](int j0) {
        if (j0 + FWD_BK <= Nk) tile_fetch<KD, false>(ks, k_rs, KD, Nk, j0, tid);
        else                   tile_fetch<KD, true>(ks, k_rs, KD, Nk, j0, tid);
    };
    auto fetch_v = [&](int j0) {
        if (j0 + FWD_BK <= Nk) tile_fetch<CVP, false>(vs, v_rs, Cv, Nk, j0, tid);
        else                   tile_fetch<CVP, true>(vs, v_rs, Cv, Nk, j0, tid);
    };
    // one float4 piece (32 rows) of a staged tile -> LDS
    auto commit_piece = [&](const f32x4& x, float* tile, int u) {
        float* d = tile + (u * 32 + (tid >> 3)) * FWD_LD + (tid & 7) * 4;
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    };

    constexpr int NB = 8, NBATCH = KD / 2 / NB;   // 16 batches of 8 MFMAs per S tile
    static_assert(NBATCH == 16, "hook schedule below is written for K = 256");
    // One batch of S^T += K_tile^T Q with its LDS operands requested a batch ahead (pinned).
    // `hook(bt)` injects the VALU / LDS-store work that should hide under this batch.
    auto qk_tile = [&](const float* ktile, f32x16& acc, auto&& hook, int bt_lo, int bt_hi,
                       float (&a)[2][NB]) {
        const float* kl = ktile + h * FWD_LD + c;
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
            if (bt < bt_lo || bt >= bt_hi) continue;
            if (bt + 1 < NBATCH) {
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    a[(bt + 1) & 1][u] = kl[(2 * ((bt + 1) * NB + u)) * FWD_LD];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) acc = mfma32(a[bt & 1][u], qreg[bt * NB + u], acc);
            hook(bt);
#pragma unroll
            for (int u = 0; u < NB / 2; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // one scheduling region per batch: hipcc's group solver is super-linear in region size
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto qk_lead = [&](const float* ktile, float (&a)[2][NB]) {
        const float* kl = ktile + h * FWD_LD + c;
#pragma unroll
        for (int u = 0; u < NB; ++u) a[0][u] = kl[(2 * u) * FWD_LD];
        __builtin_amdgcn_sched_group_barrier(0x100, NB / 2, 0);
    };

    const int ntiles = (Nk + FWD_BK - 1) / FWD_BK;
    float* const kt0 = kt;
    float* const kt1 = kt + KD * FWD_LD;
    float* const vt0 = vt;
    float* const vt1 = vt + CVP * FWD_LD;

    // ---- prologue: K(0), K(1), V(0) resident; K(2), V(1) staged; S(0) computed ------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
        fetch_k(0);
        tile_commit<KD>(ks, kt0, tid);
        fetch_k(FWD_BK);
        fetch_v(0);
        __syncthreads();
        float a[2][NB];
        qk_lead(kt0, a);
        qk_tile(kt0, s, [](int) {}, 0, NBATCH, a);
        tile_commit<KD>(ks, kt1, tid);
        tile_commit<CVP>(vs, vt0, tid);
        fetch_k(2 * FWD_BK);
        fetch_v(FWD_BK);
        __syncthreads();
    }

    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * FWD_BK;
        const bool ragged = RAGGED && (j0 + FWD_BK > Nk);
        float* const k_rd = (t & 1) ? kt0 : kt1;    // K(t+1)
        float* const k_wr = (t & 1) ? kt1 : kt0;    // <- K(t+2)
        float* const v_rd = (t & 1) ? vt1 : vt0;    // V(t)
        float* const v_wr = (t & 1) ? vt0 : vt1;    // <- V(t+1)

        f32x16 sn;
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[r] = 0.f;
        float tmax = -INFINITY, psum = 0.f;
        float a[2][NB];
        qk_lead(k_rd, a);

        // batches 0..3: scale + mask + row max of S(t), four accumulator registers per batch;
        // batches 3..15: one staged piece per batch goes to LDS (8 K pieces, then CVB V pieces)
        // ... and as soon as a staged piece is in LDS its registers take the next load (K three
        // tiles ahead, V two): memory instructions are issued one at a time under the MFMAs, never
        // as a burst (a burst of a dozen 1 KiB loads stalls the in-order wave for thousands of
        // cycles while the L1 miss queue drains)
        auto commit_hook = [&](int bt) {
            const int u = bt - 3;
            if (u >= 0 && u < KD / 32) {
                commit_piece(ks.r[u], k_wr, u);
                tile_fetch_piece<RAGGED>(ks.r[u], k_rs, u, KD, Nk, j0 + 3 * FWD_BK, tid);
            } else if (u >= KD / 32 && u - KD / 32 < CVB) {
                commit_piece(vs.r[u - KD / 32], v_wr, u - KD / 32);
                tile_fetch_piece<RAGGED>(vs.r[u - KD / 32], v_rs, u - KD / 32, Cv, Nk, j0 + 2 * FWD_BK, tid);
            }
        };
        qk_tile(k_rd, sn, [&](int bt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = bt * 4 + q;
                float x = s[r] * scale_log2;
                if (ragged && (j0 + acc_row_base(r) + 4 * h >= Nk)) x = -INFINITY;
                s[r] = x;
                tmax = fmaxf(tmax, x);
            }
            commit_hook(bt);
        }, 0, 4, a);

        tmax = fmaxf(tmax, swap_half(tmax));
        // Lazy rescale (guide T13): the running max only moves — and O, l are only rescaled —
        // when some row's tile max exceeds it by more than 2^kRescaleThr.  O lives in AGPRs, so an
        // unconditional rescale costs 2 register moves + 1 multiply per accumulator register per
        // tile; with the threshold the branch is taken a handful of times per row.  p is then
        // bounded by 2^kRescaleThr instead of 1, harmless in fp32.  The first tile always takes
        // the branch (m_run = -inf), every real tile holds >= 1 valid key so m_run becomes finite.
        if (__any(tmax > m_run + kRescaleThr)) {
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = fast_exp2(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
        }

        // batches 4..11: two registers per batch: logits store (training), exponentiate, row sum
        qk_tile(k_rd, sn, [&](int bt) {
            if (bt < 12) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = (bt - 4) * 2 + q;
                    if (STORE_S) {
                        const int jr = j0 + acc_row_base(r);
                        buf_store1s(lg_rs, s[r], (!ragged || jr + 4 * h < Nk) ? lg_lane_off : kBufOob,
                                    (unsigned)jr * (unsigned)Nq * 4u);
                    }
                    s[r] = fast_exp2(s[r] - m_run);
                    psum += s[r];
                }
            }
            commit_hook(bt);
        }, 4, NBATCH, a);
        l_run += psum;

        // ---- O^T += V^T . P^T : A = V^T[ch][key] from LDS, B = P^T (accumulator registers) --
        {
            const float* vl = v_rd + c * FWD_LD + 4 * h;
            float va[2][CVB];
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb) va[0][cb] = vl[cb * 32 * FWD_LD + acc_row_base(0)];
            __builtin_amdgcn_sched_group_barrier(0x100, CVB, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) {
#pragma unroll
                    for (int cb = 0; cb < CVB; ++cb)
                        va[(r + 1) & 1][cb] = vl[cb * 32 * FWD_LD + acc_row_base(r + 1)];
                }
#pragma unroll
                for (int cb = 0; cb < CVB; ++cb) o[cb] = mfma32(va[r & 1][cb], s[r], o[cb]);
#pragma unroll
                for (int cb = 0; cb < CVB; ++cb) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();   // K(t+2), V(t+1) visible; K(t+1), V(t) no longer read by anyone
        s = sn;
    }

    // ---- epilogue: normalise, store channel-major [B,Cv,Nq], store row LSE ------------------
    const float l_tot = l_run + swap_half(l_run);
    const float inv_l = 1.0f / l_tot;
    if (i_lane < Nq) {
        float* out_b = out + (size_t)b * Cv * Nq;
#pragma unroll
        for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = cb * 32 + acc_row_base(r) + 4 * h;
                if (ch < Cv) out_b[(size_t)ch * Nq + i_lane] = o[cb][r] * inv_l;
            }
        if (h == 0) lse[(size_t)b * Nq + i_lane] = (m_run + log2f(l_tot)) * kLn2;
    }
}

template <int KD, int CVB, bool STORE_S, bool RAGGED>
static int launch_fwd_k(const float* qn, const float* kn, const float* v, float* out, float* lse,
                        float* lg, int B, int Nq, int Nk, int Cv, float inv_t, hipStream_t stream) {
    auto kern = corr_softmax_warp_fwd_kernel<KD, CVB, STORE_S, RAGGED>;
    const size_t smem = (size_t)2 * (KD + CVB * 32) * FWD_LD * sizeof(float);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int nqb = (Nq + FWD_BQ - 1) / FWD_BQ;
    hipLaunchKernelGGL(kern, dim3(B * nqb), dim3(256), smem, stream, qn, kn, v, out, lse, lg, B, Nq,
                       Nk, Cv, inv_t * kLog2e);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

template <int KD, int CVB>
static int launch_fwd(const float* qn, const float* kn, const float* v, float* out, float* lse,
                      float* lg, int B, int Nq, int Nk, int Cv, float inv_t, hipStream_t stream) {
    // key counts that are a multiple of the 32-key tile (every power-of-two grid) get the variant
    // without any ragged-tile test
    const bool ragged = (Nk % FWD_BK) != 0;
#define COCOS_GO(ST, RG) launch_fwd_k<KD, CVB, ST, RG>(qn, kn, v, out, lse, lg, B, Nq, Nk, Cv, inv_t, stream)
    if (lg) return ragged ? COCOS_GO(true, true) : COCOS_GO(true, false);
    return ragged ? COCOS_GO(false, true) : COCOS_GO(false, false);
#undef COCOS_GO
}

}  // namespace cocos

extern "C" int cocos_corr_softmax_warp_fwd(const float* qn, const float* kn, const float* v,
                                           float* out, float* lse, float* logits_t, int B, int K,
                                           int Nq, int Nk, int Cv, float inv_temperature,
                                           cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && v && out && lse, COCOS_ERR_INVALID, "corr_softmax_warp_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1, COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd: bad dims B=%d Nq=%d Nk=%d Cv=%d", B, Nq, Nk, Cv);
    COCOS_REQUIRE(K == 256, COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_fwd: fused path needs K == 256 (got %d); use "
                  "cocos_corr_materialize + cocos_row_softmax_fwd", K);
    COCOS_REQUIRE(Cv <= 160, COCOS_ERR_UNSUPPORTED, "corr_softmax_warp_fwd: Cv=%d > 160", Cv);
    COCOS_REQUIRE((size_t)K * Nq * 4 < 0x7fffffffull && (size_t)K * Nk * 4 < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED, "corr_softmax_warp_fwd: per-sample tensor exceeds 2 GiB");
    COCOS_REQUIRE(!logits_t || (size_t)Nq * Nk * 4 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_fwd: per-sample logits exceed 2 GiB; pass logits_t = NULL");
    hipStream_t s = as_stream(stream);
    const int cvb = (Cv + 31) / 32;
    switch (cvb) {
        case 1: return launch_fwd<256, 1>(qn, kn, v, out, lse, logits_t, B, Nq, Nk, Cv, inv_temperature, s);
        case 2: return launch_fwd<256, 2>(qn, kn, v, out, lse, logits_t, B, Nq, Nk, Cv, inv_temperature, s);
        case 3: return launch_fwd<256, 3>(qn, kn, v, out, lse, logits_t, B, Nq, Nk, Cv, inv_temperature, s);
        case 4: return launch_fwd<256, 4>(qn, kn, v, out, lse, logits_t, B, Nq, Nk, Cv, inv_temperature, s);
        default: return launch_fwd<256, 5>(qn, kn, v, out, lse, logits_t, B, Nq, Nk, Cv, inv_temperature, s);
    }
}
