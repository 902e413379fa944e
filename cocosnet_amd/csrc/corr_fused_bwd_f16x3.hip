// K2 backward, query side, split-precision flavour (training path with saved logits) — gfx950.
//
// Same math as corr_fused_bwd_saved.hip (autograd of correspondence.py:291-318 w.r.t. theta, plus the
// dS^T matrix the key-side GEMM consumes), with both per-tile products on v_mfma_f32_32x32x16_f16 and
// every fp32 operand carried as f16 hi + lo (three MFMA terms per product, fp32 accumulate; see
// corr_fused_fwd_f16x3.hip for the arithmetic argument):
//     dP'(t)  = V'(t) . dO'                 A = V tile rows (keys) x 16 channels, B = resident dO' slice
//     dS''(t) = P(t) * (dP'(t) - D') * c     P from the saved logits, D' = s_o * s_v * sum_c dO*out (fp64)
//     dqn    += K(t) . dS''(t)               A = key tile rows (channels) x 16 permuted keys, B = dS''
// Scaled domains (all powers of two, exact, undone in the epilogue / by the GEMM):
//     dO' = s_o * dO   with s_o = 2^e from the tensor's |max| (device scalar written by cocos_split_f16_ex),
//     V'  = s_v * V    likewise (V has no a-priori magnitude in a general forward() call; NULL pointer = 1),
//     dS'' = s_o * s_v * ds_shift * dS  with ds_shift = the power of two that keeps the worst case
//            2 * CVP * 2^10 * max|V'| * inv_t  below 2^15 (computed here from the device-side max|v|; the product
//            s_o * s_v * ds_shift is written to `ds_scale_out` for the GEMM to undo),
//     K planes = k_scale * kn.
// The dS'' planes (f16 hi/lo, [B,Nk,Nq]) replace the fp32 dS^T/T matrix of the fp32 path: same bytes,
// already in the operand format of the f16x3 key-side GEMM (hgemm_f16x3.hip).
//
// Schedule (round 2).  A wave is alone on its SIMD (Q-side residents + accumulators = ~430 registers), so
// nothing but the instruction order overlaps the matrix pipe with the rest.  Round 1 ran, per 32-key tile,
// [30 MFMAs of dP'] -> [~150 VALU: exp, dS, f16 split; no MFMA] -> [48 MFMAs of dqn]; the VALU stretch and the
// pipeline drain around it left the matrix pipe idle for ~1/4 of the tile.  Now the tile loop is skewed by one
// stage: iteration t issues the dP' MFMAs of tile t, then the dqn MFMAs of tile t-1 with the VALU work of tile
// t sliced into their gaps (one accumulator register per MFMA step: the two are independent, so the in-order
// wave keeps the matrix pipe busy while the VALU runs).  Costs 32 more registers (dS'' operands and logits
// double-buffered, all statically named: the loop is unrolled by two).
// The saved logits arrive in the forward's private tile-blocked layout: four 16-byte loads per lane and tile
// (fully contiguous 1 KB per instruction) instead of sixteen 4-byte ones, requested a whole iteration ahead.
#include <type_traits>

#include "common.h"

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BQH_KD = 256;
constexpr int BQH_KROW = 40;     // halfs per channel row of the key tile (32 permuted keys + pad): 80 B

__device__ __forceinline__ f32x16 bq_mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 bq_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
__device__ __forceinline__ u32x4 bq_load16s(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, (int)soff, 0);
}

// Ablation builds (debug only, results are WRONG): -DCOCOS_ABLATE=<bits>  1: no tile staging, 2: no operand re-reads
// from LDS, 4: no exp/split arithmetic, 8: no dS'' transposition + plane stores, 16: no logits loads
// Projection of a FUSED key side ("accumulate dkn in the same kernel"; round 3): 1024: the 48 extra MFMAs
// per wave tile that dkn_tile[256 x 32 keys] += Q_wave . dS''^T would issue (on the dqn accumulators: no registers are left
// for a second 128-register accumulator set, which is the first obstacle); 2048: its output traffic — the wave's 256 x 32
// fp32 partial tile leaving as 128 atomic adds per lane and tile (32 KB per wave tile; with 8 + 2048 the dS'' stores it
// would replace are removed).  Not modelled and still owed by a real variant: the transposition of dS'' through LDS that
// the product needs (the contraction runs over QUERIES, which sit in the lanes of this kernel's tiles — an MFMA never
// contracts over the lane index), measured at +4.4 % of the kernel when the planes still left through it.
#ifndef COCOS_ABLATE
#define COCOS_ABLATE 0
#endif
// cache policy of the HWxHW streams (saved logits, dS'' / P planes): written once, read once by another kernel — `nt`
// (aux bit 1) keeps them from evicting the key/value tiles that the 32 workgroups of a sample share in their XCD's L2
#define COCOS_STREAM_AUX ((COCOS_ABLATE & 256) ? 0 : 2)
#ifdef COCOS_DEBUG_TIMING
__device__ long long g_phase_bq_h[8];
#define BPH_T(var) const long long var = __builtin_readcyclecounter()
#define BPH_ADD(i, a, b) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_bq_h[i] += (b) - (a); } while (0)
#else
#define BPH_T(var) do {} while (0)
#define BPH_ADD(i, a, b) do {} while (0)
#endif

constexpr float kPPlaneScale = 16384.0f;   // P in [0,1] -> planes of 2^14 * P (top of f16's range: floor 2^-38)

// one tile's dS'' (and optionally P) as MFMA B operands / packed store words
struct DsRegs {
    unsigned hw[8], lw[8];     // f16 hi / lo, two values per register: register j>>1, half j&1
};

#define COCOS_BQ_PARAMS \
    const _Float16* __restrict__ kch, const _Float16* __restrict__ kcl, const _Float16* __restrict__ vph,          \
    const _Float16* __restrict__ vpl, const _Float16* __restrict__ gph, const _Float16* __restrict__ gpl,          \
    const float* __restrict__ g_scale, const float* __restrict__ outp, const float* __restrict__ dout,             \
    const float* __restrict__ lse, const float* __restrict__ lg, float* __restrict__ dqn, _Float16* __restrict__ dsh, \
    _Float16* __restrict__ dsl, _Float16* __restrict__ psh, _Float16* __restrict__ psl, const float* __restrict__ v_amax, \
    const float* __restrict__ v_scale, float* __restrict__ ds_scale_out, const unsigned* __restrict__ v_lo_mask, int B, \
    int Nq, int Nk, int Cv, float inv_t, float k_scale, float q_scale, const float* __restrict__ rowstat, \
    const float* __restrict__ mtile, const float* __restrict__ d_pre, int kblocks
#define COCOS_BQ_ARGS \
    kch, kcl, vph, vpl, gph, gpl, g_scale, outp, dout, lse, lg, dqn, dsh, dsl, psh, psl, v_amax, v_scale, ds_scale_out, \
    v_lo_mask, B, Nq, Nk, Cv, inv_t, k_scale, q_scale, rowstat, mtile, d_pre, kblocks

// RAWM: the magnitude-free flavour (see corr_fused_fwd_f16x3.hip) — P = 2^((s_rel + (m_tile - m)) * scale - r) from the forward's
// RELATIVE saved logits, its per-tile reference m_tile and its per-row (m in raw-accumulator units, r = log2 l - bias) instead
// of from absolute logits and the row LSE: exact differences at any |logit|.
template <int CVB, bool STORE_DS, bool STORE_P, bool RAGGED, bool VLO0, bool BLK, bool RAWM = false, int KBA = BQH_KD / 32>
__device__ __forceinline__ void corr_bwd_query_f16x3_body(
    const _Float16* __restrict__ kch, const _Float16* __restrict__ kcl,   // [B,256,Nk] planes of k_scale*kn
    const _Float16* __restrict__ vph, const _Float16* __restrict__ vpl,   // [B,Nk,CVP] planes of s_v*v
    const _Float16* __restrict__ gph, const _Float16* __restrict__ gpl,   // [B,Nq,CVP] planes of s_o*dout
    const float* __restrict__ g_scale,                                     // s_o (device)
    const float* __restrict__ outp, const float* __restrict__ dout,        // [B,Cv,Nq] fp32 (for D)
    const float* __restrict__ lse, const float* __restrict__ lg,           // [B,Nq]; saved logits (blocked, see fwd)
    float* __restrict__ dqn,                                               // out [B,256,Nq]
    _Float16* __restrict__ dsh, _Float16* __restrict__ dsl,                // out [B,Nk,Nq] planes of dS''
    _Float16* __restrict__ psh, _Float16* __restrict__ psl,                // out [B,Nk,Nq] planes of 2^14 P (STORE_P)
    const float* __restrict__ v_amax,                                      // max|v| (device)
    const float* __restrict__ v_scale,                                     // s_v (device) or NULL
    float* __restrict__ ds_scale_out,                                      // out: s_o * s_v * ds_shift (device)
    const unsigned* __restrict__ v_lo_mask,                                // bit cb: value block cb has a non-zero lo plane (or NULL)
    int B, int Nq, int Nk, int Cv, float inv_t, float k_scale /* of the K planes */, float q_scale /* of the query planes the
    forward multiplied them with: raw logits = q_scale * k_scale * <q, k> */,
    const float* __restrict__ rowstat /* RAWM: [B][3][Nq] as written by the forward */,
    const float* __restrict__ mtile /* RAWM: [B][ntiles][2][Nq] as written by the forward */,
    const float* __restrict__ d_pre /* nullable: D = sum_c dout * out per query [B][Nq] (cocos_rowdot_f64) */,
    int /* kblocks: the kernel picks KBA from it */) {
    // KBA (RAWM): 32-channel blocks of k that hold non-zero channels — the others are the zero padding of the Attention block's
    // K = C/8 channels: their dqn MFMAs, fragment reads and key-tile fetches do not exist in the KBA = 1 / 2 / 4 instantiations (K <= 32 / 64 / 128)
    // (compile-time, like KST of the forward: run-time branches in the MFMA loop cost more than they save).
    static_assert(KBA >= 1 && KBA <= BQH_KD / 32 && (RAWM || KBA == BQH_KD / 32), "KBA < 8 belongs to the magnitude-free flavour");
    constexpr int CVP = CVB * 32;
    constexpr int CVS = CVP / 16;                     // k-steps of the dP product
    constexpr int KB = BQH_KD / 32;                   // channel blocks of dqn
    constexpr int VROW = CVP + 8;                     // halfs per key row of the V tile
    constexpr int VPLANE = 32 * VROW, KPLANE = BQH_KD * BQH_KROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const vt = reinterpret_cast<_Float16*>(smem_raw);   // [2 buf][hi|lo][32 keys][VROW]
    _Float16* const kt = vt + 2 * 2 * VPLANE;                      // [2 buf][hi|lo][256 ch][KROW]
    // per-wave transposition buffer for the dS'' stores: [hi|lo][32 keys][32 queries].  Rows of exactly 64 B: the
    // 16-byte read-back of a b128 lane group touches rows {0,3,5,6} / {1,2,4,7} (+8..), i.e. 16-dword windows at
    // dwords {0,48,16,32} of the 64-bank row — conflict-free (the 80-byte rows of round 1 were 2-way: 8.4 M conflict
    // cycles per launch); the 2-byte writes of one register go to two rows 4 apart from the two half-waves, which
    // are separate lane groups.
    constexpr int DSROW = 32, DSPLANE = 32 * DSROW;
    _Float16* const dstile = kt + 2 * 2 * KPLANE + (threadIdx.x >> 6) * 2 * DSPLANE;

    using std::false_type;
    using std::true_type;
    const int tid = threadIdx.x;
    // VLO0 (template): only value block 0 has a non-zero lo plane (see the forward kernel); the kernel below holds both
    // flavours of this body and picks one from the device-side mask.
    const std::integral_constant<bool, VLO0> vlo0_tag{};
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;

    const int nqb = (Nq + 127) / 128;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, q0 = (vb % nqb) * 128;
    const int i_lane = q0 + wave * 32 + c;
    const bool live = i_lane < Nq;
    const int ntiles = (Nk + 31) / 32;

    const size_t kbytes = (size_t)BQH_KD * Nk * 2, vbytes = (size_t)Nk * CVP * 2, gbytes = (size_t)Nq * CVP * 2;
    const __amdgpu_buffer_rsrc_t kh_rs = make_rsrc(kch + (size_t)b * BQH_KD * Nk, kbytes);
    const __amdgpu_buffer_rsrc_t kl_rs = make_rsrc(kcl + (size_t)b * BQH_KD * Nk, kbytes);
    const __amdgpu_buffer_rsrc_t vh_rs = make_rsrc(vph + (size_t)b * Nk * CVP, vbytes);
    const __amdgpu_buffer_rsrc_t vl_rs = make_rsrc(vpl + (size_t)b * Nk * CVP, vbytes);
    const __amdgpu_buffer_rsrc_t gh_rs = make_rsrc(gph + (size_t)b * Nq * CVP, gbytes);
    const __amdgpu_buffer_rsrc_t gl_rs = make_rsrc(gpl + (size_t)b * Nq * CVP, gbytes);
    const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(outp + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(dout + (size_t)b * Cv * Nq, (size_t)Cv * Nq * 4);
    // saved logits: per sample ntiles x nqblk blocks of 4 KB = [k = 0..3][lane][4 accumulator registers]
    const int nqblk = (Nq + 31) / 32;
    const size_t lg_bytes = (size_t)ntiles * nqblk * 4096;
    const __amdgpu_buffer_rsrc_t lg_rs = make_rsrc(reinterpret_cast<const char*>(lg) + (size_t)b * lg_bytes, lg_bytes);
    const unsigned lg_lane_off = live ? (unsigned)(((q0 >> 5) + wave) * 4096 + lane * 16) : kBufOob;
    const __amdgpu_buffer_rsrc_t dh_rs = make_rsrc(STORE_DS ? dsh + (size_t)b * Nk * Nq : nullptr,
                                                   STORE_DS ? (size_t)Nk * Nq * 2 : 0);
    const __amdgpu_buffer_rsrc_t dl_rs = make_rsrc(STORE_DS ? dsl + (size_t)b * Nk * Nq : nullptr,
                                                   STORE_DS ? (size_t)Nk * Nq * 2 : 0);
    const __amdgpu_buffer_rsrc_t ph_rs = make_rsrc(STORE_P ? psh + (size_t)b * Nk * Nq : nullptr,
                                                   STORE_P ? (size_t)Nk * Nq * 2 : 0);
    const __amdgpu_buffer_rsrc_t pl_rs = make_rsrc(STORE_P ? psl + (size_t)b * Nk * Nq : nullptr,
                                                   STORE_P ? (size_t)Nk * Nq * 2 : 0);

    const float s_v = v_scale ? *v_scale : 1.0f;
    const float s_ov = *g_scale * s_v;                 // scale of dP' = V' . dO'
    float ds_shift;
    {
        const float bound = 2.0f * CVP * 1024.0f * fmaxf(*v_amax * s_v, 1e-30f) * inv_t;   // >= |dP' - D'| * inv_t
        int e;
        frexpf(bound * (1.0f / 32768.0f), &e);                                          // 2^e > bound / 2^15
        ds_shift = ldexpf(1.0f, -min(max(e, -20), 60));
        if (ds_scale_out && blockIdx.x == 0 && tid == 0) *ds_scale_out = s_ov * ds_shift;
    }

    // ---- resident: dO' slice (B operand of dP'), D' and lse of the lane's query ---------------------
    f16x8 goh[CVS], gol[CVS];
    {
        const unsigned off = live ? (unsigned)(i_lane * CVP + h * 8) * 2u : kBufOob;
#pragma unroll
        for (int u = 0; u < CVS; ++u) {
            goh[u] = __builtin_bit_cast(f16x8, bq_load16(gh_rs, off + (unsigned)u * 32u));
            gol[u] = __builtin_bit_cast(f16x8, bq_load16(gl_rs, off + (unsigned)u * 32u));
        }
        // park the slice in the accumulator file (MFMA B operands may be AGPRs)
#pragma unroll
        for (int u = 0; u < CVS; ++u) {
            asm volatile("" : "+a"(goh[u]));
            asm volatile("" : "+a"(gol[u]));
        }
    }
    float d_lane;
    if (d_pre) {
        // round 4: D arrives precomputed (cocos_rowdot_f64: a streaming kernel on every CU) — the serial loop below, 2 x Cv / 2
        // dependent 4-byte loads per lane and an fp64 chain on a kernel that runs ONE wave per SIMD, was this kernel's prologue
        d_lane = live ? d_pre[(size_t)b * Nq + i_lane] * s_ov : 0.f;
    } else {
        // D in fp64 from the fp32 tensors (dP' - D' cancels wherever P is peaked); the two half-waves
        // take alternate channels
        double dacc = 0.0;
        for (int ch = h; ch < Cv; ch += 2) {
            const unsigned off = live ? (unsigned)(ch * Nq + i_lane) * 4u : kBufOob;
            dacc += (double)buf_load1(g_rs, off) * (double)buf_load1(o_rs, off);
        }
        const int lo = __shfl_xor((int)__double2loint(dacc), 32, 64);
        const int hi = __shfl_xor((int)__double2hiint(dacc), 32, 64);
        d_lane = (float)((dacc + __hiloint2double(hi, lo)) * (double)s_ov);
    }
    // P * cs in ONE exponential: the factor cs = inv_t * ds_shift of dS'' rides in the exponent,
    //   p_c = 2^(raw * scale_log2 - (lse * log2 e - log2 cs)),   dS'' = p_c * (dP' - D')
    // (raw = the forward's accumulator, k_scale^2 * cos; padded lanes: lse2c = +inf -> p_c = 0)
    const float cs = inv_t * ds_shift;
    const float scale_log2 = inv_t * kLog2e / (k_scale * q_scale);
    const float m_raw = (RAWM && live) ? rowstat[((size_t)b * 3 + 0) * Nq + i_lane] : 0.f;      // (hi, lo: an unevaluated sum)
    const float m_raw_lo = (RAWM && live) ? rowstat[((size_t)b * 3 + 1) * Nq + i_lane] : 0.f;
    const float nlse2c = !live ? -INFINITY
                         : RAWM ? log2f(cs) - rowstat[((size_t)b * 3 + 2) * Nq + i_lane] : log2f(cs) - lse[(size_t)b * Nq + i_lane] * kLog2e;
    const float p_from_pc = kPPlaneScale / cs;        // STORE_P: 2^14 * P = p_c * (2^14 / cs)

    f32x16 dx[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[kb][r] = 0.f;

    // ---- staging ---------------------------------------------------------------------------------------
    // V tile: 32 key rows x CVP*2 B per plane = 32*CVP/8 16-byte chunks; K tile: 256 channel rows x 64 B per
    // plane = 1024 16-byte chunks (8 keys each), 4 per thread.
    constexpr int VCH = 32 * CVP / 8;                 // chunks per V plane
    constexpr int VPT = (VCH + 255) / 256;
    u32x4 vst[2][VPT], kst[2][4];
    auto fetch_v = [&](int j0) {
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int g = u * 256 + tid, key = g / (CVP / 8), cc = g % (CVP / 8);
            // rows past Nk lie past the end of the buffer -> zeros
            const unsigned off = g < VCH ? (unsigned)((j0 + key) * CVP + cc * 8) * 2u : kBufOob;
            vst[0][u] = bq_load16(vh_rs, off);
            vst[1][u] = bq_load16(vl_rs, off);
        }
    };
    auto fetch_k = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = u * 256 + tid, row = g >> 2, k8 = g & 3;
            unsigned off = (unsigned)(row * Nk + j0 + 8 * k8) * 2u;
            if (j0 + 8 * k8 >= Nk) off = kBufOob;      // Nk % 8 == 0 (checked by the launcher)
            kst[0][u] = bq_load16(kh_rs, off);
            kst[1][u] = bq_load16(kl_rs, off);
        }
    };
    auto commit_v = [&](int buf) {
        _Float16* base = vt + buf * 2 * VPLANE;
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int g = u * 256 + tid, key = g / (CVP / 8), cc = g % (CVP / 8);
            if (g < VCH) {
                *reinterpret_cast<u32x4*>(base + key * VROW + cc * 8) = vst[0][u];
                *reinterpret_cast<u32x4*>(base + VPLANE + key * VROW + cc * 8) = vst[1][u];
            }
        }
    };
    // keys 8k8..8k8+3 and 8k8+4..8k8+7 -> k-slots of the dqn MFMA: the accumulator registers of dS''
    // are the B operand, register 8t+j of half-wave hh <-> key 16t + 8(j>>2) + 4hh + (j&3)
    auto commit_k = [&](int buf) {
        _Float16* base = kt + buf * 2 * KPLANE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = u * 256 + tid, row = g >> 2, k8 = g & 3;
            const int kq = 2 * k8;                                       // first 4-key group (kq even)
            const int slot0 = 16 * (kq >> 2) + 4 * ((kq >> 1) & 1);     // hh = 0
            _Float16* d = base + row * BQH_KROW;
            *reinterpret_cast<u32x2*>(d + slot0) = u32x2{kst[0][u].x, kst[0][u].y};
            *reinterpret_cast<u32x2*>(d + slot0 + 8) = u32x2{kst[0][u].z, kst[0][u].w};   // kq+1: hh = 1
            *reinterpret_cast<u32x2*>(d + KPLANE + slot0) = u32x2{kst[1][u].x, kst[1][u].y};
            *reinterpret_cast<u32x2*>(d + KPLANE + slot0 + 8) = u32x2{kst[1][u].z, kst[1][u].w};
        }
    };
    // Per-thread parts of the staging addresses do not depend on the tile: computed once; the tile-dependent
    // part travels in the scalar offset of the buffer instruction (non-ragged key counts only: the scalar
    // offset is not bounds-checked, so look-ahead tiles past the end are clamped to the last tile instead)
    unsigned v_voff[VPT], k_voff[4];
    int v_lds[VPT], k_lds[4];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int g = u * 256 + tid, key = g / (CVP / 8), cc = g % (CVP / 8);
        v_voff[u] = g < VCH ? (unsigned)(key * CVP + cc * 8) * 2u : kBufOob;
        v_lds[u] = key * VROW + cc * 8;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int g = u * 256 + tid, row = g >> 2, k8 = g & 3, kq = 2 * k8;
        // (RAWM: rows of zero-padding channels are not fetched — out-of-range lanes return the zeros that are there anyway)
        k_voff[u] = (row >= KBA * 32) ? kBufOob : (unsigned)(row * Nk + 8 * k8) * 2u;
        k_lds[u] = row * BQH_KROW + 16 * (kq >> 2) + 4 * ((kq >> 1) & 1);
    }
    // logits of tile tt (clamped to the last tile: look-ahead loads past the end are harmless re-reads),
    // register group k (accumulator registers 4k..4k+3): one contiguous 1 KB per wave-instruction
    // RAWM: m_tile(tt) - m of the lane's query (<= 0): added to the relative logits of tile tt
    const __amdgpu_buffer_rsrc_t mt_rs = make_rsrc(RAWM ? mtile + (size_t)b * ntiles * 2 * Nq : nullptr, RAWM ? (size_t)ntiles * 2 * Nq * 4 : 0);
    auto load_dm = [&](float& dm, int tt) {
        if (!RAWM) return;
        const int tc = min(tt, ntiles - 1);
        const float th = buf_load1(mt_rs, live ? (unsigned)((tc * 2 + 0) * Nq + i_lane) * 4u : kBufOob);
        const float tl = buf_load1(mt_rs, live ? (unsigned)((tc * 2 + 1) * Nq + i_lane) * 4u : kBufOob);
        dm = (th - m_raw) + (tl - m_raw_lo);              // both maxima are fp32 numbers a few ulps apart: exact differences
    };
    auto load_s = [&](f32x4& dst, int tt, int k) {
        if ((COCOS_ABLATE & 16) && tt > 1) return;
        const int tc = (COCOS_ABLATE & 128) ? (tt & 1) : min(tt, ntiles - 1);
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            lg_rs, (int)lg_lane_off, (int)((unsigned)(tc * nqblk) * 4096u + (unsigned)k * 1024u), COCOS_STREAM_AUX));
    };

    // one staged piece of the V tile (i < 2*VPT) or of the K tile (2*VPT <= i < 2*VPT + 8) to LDS, and the reload of
    // its register with the tile after: V(tv) -> vt[bufv], then V(tv + 1) requested; K(tk) -> kt[bufk], then K(tk + 1)
    auto stage_piece = [&](int i, int tv, int tk, auto piece_tag) __attribute__((always_inline)) {
        constexpr bool PLO0 = decltype(piece_tag)::value || (COCOS_ABLATE & 512);   // lo plane of value channels >= 32 is zero: not fetched
        if (COCOS_ABLATE & 1) return;
        _Float16* const vw = vt + (tv & 1) * 2 * VPLANE;
        _Float16* const kw = kt + (tk & 1) * 2 * KPLANE;
        if (i < 2 * VPT) {
            const int pl_ = i & 1, u = i >> 1;
            if (!RAGGED) {
                const int jn = (COCOS_ABLATE & 64) ? 32 : min((tv + 1) * 32, Nk - 32);
                if (u * 256 + 255 < VCH || u * 256 + tid < VCH)
                    *reinterpret_cast<u32x4*>(vw + pl_ * VPLANE + v_lds[u]) = vst[pl_][u];
                vst[pl_][u] = bq_load16s(pl_ ? vl_rs : vh_rs,
                                         (PLO0 && pl_ == 1 && (u * 256 + tid) % (CVP / 8) >= 4) ? kBufOob : v_voff[u],
                                         (unsigned)jn * (unsigned)(CVP * 2));
            } else {
                const int jn = (tv + 1) * 32;
                const int g = u * 256 + tid, key = g / (CVP / 8), cc = g % (CVP / 8);
                if (g < VCH) *reinterpret_cast<u32x4*>(vw + pl_ * VPLANE + key * VROW + cc * 8) = vst[pl_][u];
                vst[pl_][u] = bq_load16(pl_ ? vl_rs : vh_rs,
                                        (g < VCH && !(PLO0 && pl_ == 1 && cc >= 4)) ? (unsigned)((jn + key) * CVP + cc * 8) * 2u : kBufOob);
            }
        } else if (i - 2 * VPT < 8) {
            const int pl_ = (i - 2 * VPT) & 1, u = (i - 2 * VPT) >> 1;
            if (!RAGGED) {
                const int jn = (COCOS_ABLATE & 64) ? 32 : min((tk + 1) * 32, Nk - 32);
                _Float16* d = kw + pl_ * KPLANE + k_lds[u];
                *reinterpret_cast<u32x2*>(d) = u32x2{kst[pl_][u].x, kst[pl_][u].y};
                *reinterpret_cast<u32x2*>(d + 8) = u32x2{kst[pl_][u].z, kst[pl_][u].w};
                kst[pl_][u] = bq_load16s(pl_ ? kl_rs : kh_rs, k_voff[u], (unsigned)jn * 2u);
            } else {
                const int jn = (tk + 1) * 32;
                const int g = u * 256 + tid, row = g >> 2, k8 = g & 3;
                const int kq = 2 * k8, slot0 = 16 * (kq >> 2) + 4 * ((kq >> 1) & 1);
                _Float16* d = kw + pl_ * KPLANE + row * BQH_KROW;
                *reinterpret_cast<u32x2*>(d + slot0) = u32x2{kst[pl_][u].x, kst[pl_][u].y};
                *reinterpret_cast<u32x2*>(d + slot0 + 8) = u32x2{kst[pl_][u].z, kst[pl_][u].w};
                unsigned off = (unsigned)(row * Nk + jn + 8 * k8) * 2u;
                if (jn + 8 * k8 >= Nk) off = kBufOob;
                kst[pl_][u] = bq_load16(pl_ ? kl_rs : kh_rs, off);
            }
        }
    };
    static_assert(2 * VPT <= 4 * CVS, "value-tile pieces must fit the dP steps");

    // ---- dP'(t) = V'(t) . dO' : CVS steps of 3 MFMAs; the value-tile pieces of tile t+1 ride in the gaps.  The first
    //      fragments (vf_h, vf_l) were requested at the end of the previous iteration: the loop does not start cold ------
    f16x8 vf_h, vf_l;
    auto prefetch_v = [&](int t) {
        const _Float16* vb0 = vt + (t & 1) * 2 * VPLANE + c * VROW + h * 8;
        vf_h = *reinterpret_cast<const f16x8*>(vb0);
        vf_l = *reinterpret_cast<const f16x8*>(vb0 + VPLANE);
    };
    // VLO0 (uniform, see the forward kernel): only value block 0 has a non-zero lo plane — the V_lo * dO_hi term, its
    // fragment reads and the fetch of those lo channels are skipped for the other blocks (8 of 30 MFMAs, same result)
    auto phase_dp = [&](f32x16& dp0, int t) __attribute__((always_inline)) {
        constexpr bool SKIPLO = VLO0 || (COCOS_ABLATE & 512);
        // (round 6: the accumulator STARTS at -D' — the subtraction of dS'' = P (dP' - D') rides in the initialisation the chain needs
        //  anyway: 16 VALU instructions per tile fewer in a kernel that is bound by them)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp0[r] = -d_lane;
        const _Float16* vb0 = vt + (t & 1) * 2 * VPLANE + c * VROW + h * 8;
        f16x8 ah[2], al[2];
        ah[0] = vf_h;
        al[0] = vf_l;
        constexpr int NP = 2 * VPT;
        constexpr int PER = (NP + CVS - 1) / CVS;
#pragma unroll
        for (int u = 0; u < CVS; ++u) {
            const int cur = u & 1, nxt = cur ^ 1;
            if (!(COCOS_ABLATE & 2) && u + 1 < CVS) {
                ah[nxt] = *reinterpret_cast<const f16x8*>(vb0 + (u + 1) * 16);
                if (!SKIPLO || u + 1 < 2) al[nxt] = *reinterpret_cast<const f16x8*>(vb0 + VPLANE + (u + 1) * 16);
            }
            dp0 = bq_mfma(ah[cur], goh[u], dp0);
            dp0 = bq_mfma(ah[cur], gol[u], dp0);
            if (!SKIPLO || u < 2) dp0 = bq_mfma(al[cur], goh[u], dp0);
#pragma unroll
            for (int q = 0; q < PER; ++q)
                if (u * PER + q < NP) stage_piece(u * PER + q, t + 1, t, vlo0_tag);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // dS'' (and P) go to HBM as 16-byte row pieces of a [key][query] matrix, but each lane holds one QUERY column (16
    // keys, two per packed register): the wave transposes its 32x32 tile through its private LDS buffer.  Register j of
    // a lane = keys (2j', 2j'+1) of its query; neighbouring lanes exchange halves (one DPP move + one byte permute per
    // register, VALU work that hides under the MFMAs) so that every lane writes ONE 32-bit word = two neighbouring
    // queries of one key: 16 conflict-free ds_write_b32 per tile instead of 32 ds_write_b16 (measured 8 cycles each
    // beyond what the MFMAs hide — tools/probes/filler_cost.hip), then 4 b128 reads + 4 b128 global stores.
    unsigned* const ds32 = reinterpret_cast<unsigned*>(dstile) + (4 * h + (c & 1)) * (DSROW / 2) + (c >> 1);
    const unsigned pair_sel = (c & 1) ? 0x03020706u : 0x05040100u;
    auto stage_transpose = [&](int j, unsigned hw, unsigned lw) {
        const int row = acc_row_base(2 * j);                    // even lanes: key row(2j); odd lanes: the next one
        const unsigned xh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hw, 0xB1, 0xf, 0xf, false);   // lane ^ 1
        const unsigned xl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lw, 0xB1, 0xf, 0xf, false);
        ds32[row * (DSROW / 2)] = __builtin_amdgcn_perm(xh, hw, pair_sel);
        ds32[DSPLANE / 2 + row * (DSROW / 2)] = __builtin_amdgcn_perm(xl, lw, pair_sel);
    };

    // ---- VALU slice r of tile t: dS''[r] from logit register r; pairs are split to f16 hi/lo at odd r ----------
    auto valu_slice = [&](int r, int t, const f32x4 (&s)[4], float dm, const f32x16& dp0, float (&dsv)[2], float (&pv)[2],
                          DsRegs& out, DsRegs& pout) {
        float pc = (COCOS_ABLATE & 4) ? s[r >> 2][r & 3] * 1e-9f
                                      : fast_exp2(__builtin_fmaf(RAWM ? s[r >> 2][r & 3] + dm : s[r >> 2][r & 3], scale_log2, nlse2c));
        if (RAGGED && (t * 32 + acc_row_base(r) + 4 * h >= Nk)) pc = 0.f;
        dsv[r & 1] = pc * dp0[r];
        if (STORE_P) pv[r & 1] = pc * p_from_pc;
        if (r & 1) {
            split_pair_rtz(dsv[0], dsv[1], out.hw[r >> 1], out.lw[r >> 1]);
            if (STORE_P) split_pair_rtz(pv[0], pv[1], pout.hw[r >> 1], pout.lw[r >> 1]);
        }
        if (!BLK && STORE_DS && (r & 1) && !(COCOS_ABLATE & 8)) stage_transpose(r >> 1, out.hw[r >> 1], out.lw[r >> 1]);
    };
    // BLK: the planes are stored in the accumulator's own orientation, [query][key], as [Nq/32][Nk/32] blocks of
    // 2 x [32 queries][16 keys] halfs (2 KB; halves = keys 0..15 | 16..31 of the tile): lane (c, h) holds keys 8m + 4h .. +3 of query c in regs (2m, 2m+1); ONE
    // v_permlane32_swap per register pair of groups (m0, m1) = (0,1), (2,3) leaves lane h = 0 with the 8 consecutive keys of
    // group m0 and lane h = 1 with those of m1 (probed: tools/probes/permlane_probe.hip) — 16 bytes per lane, the wave's
    // tile is 2 KB contiguous and leaves as two stores per plane.  No LDS round trip; the key-side GEMM reads its B
    // fragments from this layout with ds_read_b64_tr_b16 (hgemm_f16x3.hip, b_blocked = 2).
    auto store_regs_blk = [&](int t, const DsRegs& d, __amdgpu_buffer_rsrc_t h_rs, __amdgpu_buffer_rsrc_t l_rs, int pp_lo = 0,
                              int pp_hi = 2) {
        if (COCOS_ABLATE & 8) return;
        const unsigned blk = (unsigned)((((q0 >> 5) + wave) * (Nk >> 5) + t) * 2048);     // bytes: block (q-block, key tile t)
#pragma unroll
        for (int pp = pp_lo; pp < pp_hi; ++pp) {         // group pairs (0,1) and (2,3)
            const int m0 = 2 * pp, m1 = 2 * pp + 1;         // lane h = 0 ends up with group m0, h = 1 with m1
            u32x4 xh, xl;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const auto sh = __builtin_amdgcn_permlane32_swap(d.hw[2 * m0 + i], d.hw[2 * m1 + i], false, false);
                const auto sl = __builtin_amdgcn_permlane32_swap(d.lw[2 * m0 + i], d.lw[2 * m1 + i], false, false);
                xh[i] = sh[0]; xh[2 + i] = sh[1];
                xl[i] = sl[0]; xl[2 + i] = sl[1];
            }
            // block = two 1 KB halves [pp][32 queries][16 keys]: each store instruction writes one half, contiguous
            unsigned off = blk + (unsigned)(pp * 1024 + c * 32 + h * 16);
            if (COCOS_ABLATE & 32) off = (unsigned)(lane * 16 + wave * 1024 + pp * 4096);
            __builtin_amdgcn_raw_buffer_store_b128(xh, h_rs, (int)off, 0, COCOS_STREAM_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(xl, l_rs, (int)off, 0, COCOS_STREAM_AUX);
        }
    };
    // !BLK: the wave's transposed 32x32 tile: LDS -> 16-byte row pieces of the row-major [Nk][Nq] planes
    auto store_planes = [&](int t, const DsRegs& d, __amdgpu_buffer_rsrc_t h_rs, __amdgpu_buffer_rsrc_t l_rs) {
        if (BLK) { store_regs_blk(t, d, h_rs, l_rs); return; }
        if (COCOS_ABLATE & 8) return;
        // (LDS instructions of one wave execute in order: the 2-byte writes above are visible here)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int key = pass * 16 + (lane >> 2), qc = (lane & 3) * 8;
            const u32x4 xh = *reinterpret_cast<const u32x4*>(dstile + key * DSROW + qc);
            const u32x4 xl = *reinterpret_cast<const u32x4*>(dstile + DSPLANE + key * DSROW + qc);
            const int row = t * 32 + key, col = q0 + wave * 32 + qc;        // Nq % 8 == 0: whole pieces
            unsigned off = (row < Nk && col < Nq) ? (unsigned)(row * Nq + col) * 2u : kBufOob;
            if (COCOS_ABLATE & 32) off = (unsigned)(lane * 16 + wave * 1024 + pass * 4096);
            __builtin_amdgcn_raw_buffer_store_b128(xh, h_rs, (int)off, 0, COCOS_STREAM_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(xl, l_rs, (int)off, 0, COCOS_STREAM_AUX);
        }
    };
    auto store_p_tile = [&](int t, const DsRegs& pr) {
        // same transposition for the P planes, through the same per-wave buffer (after the dS'' pieces have been
        // read back; one wave's LDS instructions execute in order)
        if (!BLK) {
#pragma unroll
            for (int j = 0; j < 8; ++j) stage_transpose(j, pr.hw[j], pr.lw[j]);
        }
        store_planes(t, pr, ph_rs, pl_rs);
    };

    // ---- dqn += K(t-1) . dS''(t-1) : 16 steps of 3 MFMAs (2 k-steps x 8 channel blocks), operands one step ahead.
    //      This phase starts right behind the tile's barrier (the key tile it reads was committed by all four waves
    //      during the previous iteration's pass through here), so its first fragments cannot be requested earlier:
    //      WITH_VALU runs the first LEAD slices of tile t's VALU work while they arrive, the other slices and, as the
    //      logit registers are consumed, the loads of tile t + 2's logits ride in the MFMA gaps.  STAGE_K: the key-tile
    //      pieces of tile t go to LDS (read from the next iteration on), one every other step ----------------------------
    constexpr int LEAD = 3;
#ifndef BQ_STORE_I0
#define BQ_STORE_I0 7        // dqn steps at which the two halves of the tile's planes are stored: slices 0..7 are done by step 4,
#define BQ_STORE_I1 15       // slices 8..15 by step 12 (tuned: see DESIGN 5.0)
#endif
    auto phase_dqn = [&](int t, const DsRegs& prev, auto with_valu, auto stage_k, f32x4 (&s)[4], float& dm, const f32x16& dp0,
                         DsRegs& cur, DsRegs& pcur) {
        constexpr bool WITH_VALU = decltype(with_valu)::value;
        constexpr bool STAGE_K = decltype(stage_k)::value;
        const _Float16* kb0 = kt + ((t - 1) & 1) * 2 * KPLANE + c * BQH_KROW + h * 8;
        f16x8 a_h[2], a_l[2];
        a_h[0] = *reinterpret_cast<const f16x8*>(kb0);
        a_l[0] = *reinterpret_cast<const f16x8*>(kb0 + KPLANE);
        float dsv[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
        const float dm_t = dm;                                           // (this tile's; `dm` is reloaded for tile t + 2 below)
        auto slice = [&](int r) {
            valu_slice(r, t, s, dm_t, dp0, dsv, pv, cur, pcur);
            if ((r & 3) == 3) load_s(s[r >> 2], t + 2, r >> 2);          // registers 4k..4k+3 are free again
            if (r == 15) load_dm(dm, t + 2);
        };
        if (WITH_VALU) {
#pragma unroll
            for (int r = 0; r < LEAD; ++r) slice(r);
        }
        f16x8 sh[2], sl[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            sh[tt] = __builtin_bit_cast(f16x8, u32x4{prev.hw[4 * tt], prev.hw[4 * tt + 1], prev.hw[4 * tt + 2], prev.hw[4 * tt + 3]});
            sl[tt] = __builtin_bit_cast(f16x8, u32x4{prev.lw[4 * tt], prev.lw[4 * tt + 1], prev.lw[4 * tt + 2], prev.lw[4 * tt + 3]});
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2 * KB; ++i) {                // i = tt * KB + kb
            const int tt = i / KB, kb = i % KB, cur_ = i & 1, nxt = cur_ ^ 1;
            if (!(COCOS_ABLATE & 2) && i + 1 < 2 * KB) {
                const int t2 = (i + 1) / KB, k2 = (i + 1) % KB;
                a_h[nxt] = *reinterpret_cast<const f16x8*>(kb0 + k2 * 32 * BQH_KROW + t2 * 16);
                a_l[nxt] = *reinterpret_cast<const f16x8*>(kb0 + KPLANE + k2 * 32 * BQH_KROW + t2 * 16);
            }
            if (kb < KBA) {      // (a compile-time fact after unrolling)
                dx[kb] = bq_mfma(a_h[cur_], sh[tt], dx[kb]);
                dx[kb] = bq_mfma(a_h[cur_], sl[tt], dx[kb]);
                dx[kb] = bq_mfma(a_l[cur_], sh[tt], dx[kb]);
            }
            if (WITH_VALU && i + LEAD < 16) slice(i + LEAD);
            // blocked planes leave from registers (no LDS round trip): each half of the tile as soon as its slices are done,
            // BETWEEN the MFMAs — at the end of the iteration the four 1 KB stores cost 0.04 ms of the 0.33 ms kernel even
            // with an L2-resident target (round-4 ablation), i.e. their issue, not HBM
            // (only when dS'' is the one plane set: with the P planes as well — cycle terms, the Attention block — four stores per
            //  position measured 1-2 % SLOWER than all eight at the end, same box, tools/bq_ab.py)
            if (WITH_VALU && BLK && !STORE_P && (i == BQ_STORE_I0 || i == BQ_STORE_I1)) {
                const int pp = i == BQ_STORE_I1 ? 1 : 0;
                if (STORE_DS) store_regs_blk(t, cur, dh_rs, dl_rs, pp, pp + 1);
            }
            if (STAGE_K && (i & 1) == 0) stage_piece(2 * VPT + (i >> 1), t + 1, t, false_type{});
            if (WITH_VALU && i == 2 * KB - 1) prefetch_v(t + 1);         // first fragments of the next iteration's dP'
            __builtin_amdgcn_sched_barrier(0);
        }
        if (COCOS_ABLATE & 1024) {
#pragma unroll
            for (int i = 0; i < 2 * KB; ++i) {
                dx[i % KB] = bq_mfma(a_h[i & 1], sh[i / KB], dx[i % KB]);
                dx[i % KB] = bq_mfma(a_h[i & 1], sl[i / KB], dx[i % KB]);
                dx[i % KB] = bq_mfma(a_l[i & 1], sh[i / KB], dx[i % KB]);
            }
        }
        if ((COCOS_ABLATE & 2048) && live) {
            float* base = dqn + (size_t)b * BQH_KD * Nq + (size_t)((t & 127) * 32) + c;      // a 256 x 32 tile somewhere in dqn[b]
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    atomicAdd(base + (size_t)(kb * 32 + acc_row_base(r) + 4 * h) * Nq, dx[kb][r]);
        }
        // the dqn accumulators live in the accumulator file for the whole kernel (without the pins hipcc
        // rotates them through other AGPR ranges: 64 v_accvgpr_mov per tile)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+a"(dx[kb]));
    };

    // ---- prologue: V(0), K(0) in LDS; V(1), K(1) staged in registers; logits of tiles 0 and 1 requested -----------
    f32x4 sA[4], sB[4];
    float dmA = 0.f, dmB = 0.f;
    fetch_v(0);
    fetch_k(0);
    load_dm(dmA, 0);
    load_dm(dmB, 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) load_s(sA[k], 0, k);
    commit_v(0);
    commit_k(0);
    fetch_v(RAGGED ? 32 : min(32, Nk - 32));
    fetch_k(RAGGED ? 32 : min(32, Nk - 32));
#pragma unroll
    for (int k = 0; k < 4; ++k) load_s(sB[k], 1, k);
    __syncthreads();
    prefetch_v(0);

    DsRegs dA, dB, pA, pB;
    f32x16 dp0;

    // ONE barrier per iteration, between the two MFMA loops.  Value tile t+1 is committed during dP'(t) into the buffer
    // dP'(t-1) read (every wave finished that before it passed the barrier of iteration t-1) and is read from the end of
    // iteration t on (prefetch_v), behind the barrier of iteration t.  Key tile t is committed during the dqn loop of
    // iteration t into the buffer the dqn loop of iteration t-1 read (every wave finished that before it reached this
    // iteration's barrier) and is read by the dqn loop of iteration t+1, behind that iteration's barrier.

    // iteration 0: dP'(0), then tile 0's VALU work on its own (there is no dqn product to hide it under yet)
    {
        phase_dp(dp0, 0);
        __syncthreads();
        float dsv[2], pv[2];
        const float dm0 = dmA;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            valu_slice(r, 0, sA, dm0, dp0, dsv, pv, dA, pA);
            if ((r & 3) == 3) load_s(sA[r >> 2], 2, r >> 2);
        }
        load_dm(dmA, 2);
        prefetch_v(1);
        if (STORE_DS) store_planes(0, dA, dh_rs, dl_rs);
        if (STORE_P) store_p_tile(0, pA);
    }
    // iterations 1 .. ntiles-1, two per trip so that every register set is named statically:
    //   odd t : logits sB -> dS'' dB, dqn of tile t-1 from dA;   even t: logits sA -> dA, dqn from dB
    auto iter = [&](int t, f32x4 (&s)[4], float& dm, const DsRegs& prev, DsRegs& cur, DsRegs& pcur) {
        BPH_T(tp0);
        phase_dp(dp0, t);
        BPH_T(tp1);
        __syncthreads();
        BPH_T(tp2);
        phase_dqn(t, prev, true_type{}, true_type{}, s, dm, dp0, cur, pcur);
        BPH_T(tp3);
        if (!BLK || STORE_P || BQ_STORE_I0 > 15) {      // (else: the blocked dS'' planes were stored inside phase_dqn)
            if (STORE_DS) store_planes(t, cur, dh_rs, dl_rs);
            if (STORE_P) store_p_tile(t, pcur);
        }
        BPH_T(tp4);
        BPH_ADD(0, tp0, tp1); BPH_ADD(3, tp1, tp2); BPH_ADD(1, tp2, tp3); BPH_ADD(2, tp3, tp4);
    };
    int t = 1;
    for (; t + 1 < ntiles; t += 2) {
        iter(t, sB, dmB, dA, dB, pB);
        iter(t + 1, sA, dmA, dB, dA, pA);
    }
    if (t < ntiles) {
        iter(t, sB, dmB, dA, dB, pB);
        __syncthreads();           // the last key tile was committed during that iteration
        phase_dqn(ntiles, dB, false_type{}, false_type{}, sA, dmA, dp0, dA, pA);
    } else {
        __syncthreads();
        phase_dqn(ntiles, dA, false_type{}, false_type{}, sB, dmB, dp0, dB, pB);
    }

    // ---- epilogue: undo the scales -----------------------------------------------------------------------
    if (live) {
        const float undo = 1.0f / (k_scale * s_ov * ds_shift);
        float* dx_b = dqn + (size_t)b * BQH_KD * Nq;
#pragma unroll
        for (int kb = 0; kb < KBA; ++kb)      // (KBA < 8: the rows of the padding channels are not written — dqn's real rows are a view)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb * 32 + acc_row_base(r) + 4 * h;
                dx_b[(size_t)k * Nq + i_lane] = dx[kb][r] * undo;
            }
    }
}

// The kernel: one launch holds both flavours of the body (V_lo terms of the value blocks >= 1 issued / skipped) and picks
// one, workgroup-uniformly, from the device-side mask as its first action (see corr_fused_fwd_f16x3.hip).  DUAL = false:
// no mask / a single value block — only the general flavour is compiled in.
template <int CVB, bool STORE_DS, bool STORE_P, bool RAGGED, bool DUAL, bool BLK>
__global__ __launch_bounds__(256, 1) void corr_bwd_query_f16x3_kernel(COCOS_BQ_PARAMS, const float* __restrict__ q_scale_dev,
                                                                      const float* __restrict__ k_scale_dev) {
    if (q_scale_dev) {           // operands with device-side scales (see the forward kernel)
        q_scale = *q_scale_dev;
        k_scale = *k_scale_dev;
    }
    if constexpr (DUAL) {
        if ((__builtin_amdgcn_readfirstlane(*v_lo_mask) & ~1u) == 0u)
            corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, true, BLK>(COCOS_BQ_ARGS);
        else
            corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, false, BLK>(COCOS_BQ_ARGS);
    } else {
        if (rowstat) {           // the magnitude-free flavour: (m, r) per row from the forward
            if constexpr (RAGGED || !BLK) {        // (odd shapes: the general instantiation only)
                corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, false, BLK, true>(COCOS_BQ_ARGS);
            } else {
                if (kblocks <= 1) corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, false, BLK, true, 1>(COCOS_BQ_ARGS);
                else if (kblocks <= 2) corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, false, BLK, true, 2>(COCOS_BQ_ARGS);
                else if (kblocks <= 4) corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, false, BLK, true, 4>(COCOS_BQ_ARGS);
                else corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, false, BLK, true>(COCOS_BQ_ARGS);
            }
        } else
            corr_bwd_query_f16x3_body<CVB, STORE_DS, STORE_P, RAGGED, false, BLK>(COCOS_BQ_ARGS);
    }
}

template <int CVB>
static int launch_bq_f16x3(const _Float16* kch, const _Float16* kcl, const _Float16* vph, const _Float16* vpl,
                           const _Float16* gph, const _Float16* gpl, const float* g_scale, const float* outp,
                           const float* dout, const float* lse, const float* lg, float* dqn, _Float16* dsh,
                           _Float16* dsl, _Float16* psh, _Float16* psl, const float* v_amax, const float* v_scale,
                           float* ds_scale_out, const unsigned* v_lo_mask, int B, int Nq, int Nk, int Cv, float inv_t,
                           float k_scale, int blocked, const float* qsd, const float* ksd, const float* rowstat,
                           const float* mtile, const float* d_pre, int kblocks, hipStream_t s) {
    const bool ragged = (Nk % 32) != 0, store = dsh != nullptr, storep = psh != nullptr;
    const size_t smem = ((size_t)2 * 2 * (32 * (CVB * 32 + 8) + BQH_KD * BQH_KROW) + 4 * 2 * 32 * 32) * sizeof(_Float16);
    const int nqb = (Nq + 127) / 128;
#define COCOS_GO(DS, RG)                                                                                     \
    do {                                                                                                     \
        if (storep) COCOS_GO2(DS, true, RG); else COCOS_GO2(DS, false, RG);                                  \
    } while (0)
#define COCOS_GO4(DS, SP, RG, VL, MASK, BL)                                                                  \
    do {                                                                                                     \
        auto kern = corr_bwd_query_f16x3_kernel<CVB, DS, SP, RG, VL, BL>;                                        \
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                             \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));        \
        hipLaunchKernelGGL(kern, dim3(B * nqb), dim3(256), smem, s, kch, kcl, vph, vpl, gph, gpl, g_scale,   \
                           outp, dout, lse, lg, dqn, dsh, dsl, psh, psl, v_amax, v_scale, ds_scale_out, MASK, B, Nq, \
                           Nk, Cv, inv_t, k_scale, k_scale, rowstat, mtile, d_pre, kblocks, qsd, ksd);       \
    } while (0)
    /* blocked planes ([query][key] blocks, see the kernel) only exist for whole tiles */
#define COCOS_GO3(DS, SP, RG, VL, MASK)                                                                      \
    do {                                                                                                     \
        if (!RG && blocked) COCOS_GO4(DS, SP, RG, VL, MASK, (!RG)); else COCOS_GO4(DS, SP, RG, VL, MASK, false); \
    } while (0)
    /* with a mask and more than one value block the kernel holds both flavours and picks one from the device-side mask
       (whole tiles only: the ragged two-flavour instantiations need a few dwords of scratch and are not built) */
#define COCOS_GO2(DS, SP, RG)                                                                                \
    do {                                                                                                     \
        if (CVB > 1 && !RG && v_lo_mask) COCOS_GO3(DS, SP, RG, (CVB > 1 && !RG), v_lo_mask);                 \
        else COCOS_GO3(DS, SP, RG, false, nullptr);                                                          \
    } while (0)
    if (store) { if (ragged) COCOS_GO(true, true); else COCOS_GO(true, false); }
    else       { if (ragged) COCOS_GO(false, true); else COCOS_GO(false, false); }
#undef COCOS_GO
#undef COCOS_GO2
#undef COCOS_GO3
#undef COCOS_GO4
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

}  // namespace cocos

#ifdef COCOS_DEBUG_TIMING
extern "C" int cocos_debug_read_timing_bwd_f16x3(long long* host8, int reset) {
    using namespace cocos;
    COCOS_HIP_CHECK(hipDeviceSynchronize());
    COCOS_HIP_CHECK(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_phase_bq_h), 8 * sizeof(long long)));
    if (reset) {
        long long z[8] = {0};
        COCOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_bq_h), z, sizeof(z)));
    }
    return COCOS_OK;
}
#endif

extern "C" int cocos_corr_softmax_warp_bwd_query_f16x3(
    const void* kch, const void* kcl, const void* vph, const void* vpl, const void* gph, const void* gpl,
    const float* g_scale_dev, const float* out, const float* dout, const float* lse, const void* saved_logits,
    float* dqn, void* dsh, void* dsl, void* psh, void* psl, const float* v_amax_dev, const float* v_scale_dev,
    float* ds_scale_out_dev, const unsigned* v_lo_mask_dev, int B, int K, int Nq, int Nk, int Cv, int CvPad,
    float inv_temperature, float k_scale, const float* q_scale_dev, const float* k_scale_dev, int planes_blocked,
    cocos_stream_t stream) {
    return cocos_corr_softmax_warp_bwd_query_f16x3_ex(kch, kcl, vph, vpl, gph, gpl, g_scale_dev, out, dout, lse, saved_logits, dqn, dsh,
                                                      dsl, psh, psl, v_amax_dev, v_scale_dev, ds_scale_out_dev, v_lo_mask_dev, B, K, Nq,
                                                      Nk, Cv, CvPad, inv_temperature, k_scale, q_scale_dev, k_scale_dev, planes_blocked,
                                                      nullptr, nullptr, nullptr, 0, stream);
}

namespace cocos {
// D[b][i] = sum_c a[b][c][i] * b[b][c][i] accumulated in fp64.  Workgroup = 128 queries (32 lanes of 16-byte loads) x 8 channel
// groups, partial sums folded through LDS: 40 MB at the benchmark shape, every CU busy.
__global__ __launch_bounds__(256) void rowdot_f64_kernel(const float* __restrict__ a, const float* __restrict__ bsrc,
                                                         float* __restrict__ d, int C, int N) {
    __shared__ double red[8][128];
    const int q4 = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int i0 = blockIdx.x * 128 + q4 * 4, b = blockIdx.y;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const float* pa = a + (size_t)b * C * N + i0;
    const float* pb = bsrc + (size_t)b * C * N + i0;
    if ((N & 3) == 0 && i0 < N) {
        for (int c = cg; c < C; c += 8) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(pa + (size_t)c * N), y = *reinterpret_cast<const f32x4*>(pb + (size_t)c * N);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (double)x[e] * (double)y[e];
        }
    } else {
        for (int c = cg; c < C; c += 8)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i0 + e < N) acc[e] += (double)pa[(size_t)c * N + e] * (double)pb[(size_t)c * N + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[cg][q4 * 4 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 128) {
        const int i = blockIdx.x * 128 + threadIdx.x;
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
        if (i < N) d[(size_t)b * N + i] = (float)t;
    }
}
}  // namespace cocos

extern "C" int cocos_rowdot_f64(const float* a, const float* b, float* d, int B, int C, int N, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(a && b && d, COCOS_ERR_INVALID, "rowdot_f64: null pointer");
    COCOS_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && N >= 1, COCOS_ERR_INVALID, "rowdot_f64: bad dims B=%d C=%d N=%d", B, C, N);
    COCOS_REQUIRE((N & 3) != 0 || (aligned16(a) && aligned16(b)), COCOS_ERR_INVALID, "rowdot_f64: a / b must be 16-byte aligned");
    hipLaunchKernelGGL(rowdot_f64_kernel, dim3((unsigned)((N + 127) / 128), (unsigned)B), dim3(256), 0, as_stream(stream), a, b, d, C, N);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_corr_softmax_warp_bwd_query_f16x3_ex(
    const void* kch, const void* kcl, const void* vph, const void* vpl, const void* gph, const void* gpl,
    const float* g_scale_dev, const float* out, const float* dout, const float* lse, const void* saved_logits,
    float* dqn, void* dsh, void* dsl, void* psh, void* psl, const float* v_amax_dev, const float* v_scale_dev,
    float* ds_scale_out_dev, const unsigned* v_lo_mask_dev, int B, int K, int Nq, int Nk, int Cv, int CvPad,
    float inv_temperature, float k_scale, const float* q_scale_dev, const float* k_scale_dev, int planes_blocked,
    const float* rowstat, const float* mtile, const float* d_pre, int k_active, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(k_active >= 0 && k_active <= K && (k_active == 0 || k_active == K || rowstat), COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query_f16x3: k_active=%d (channels >= k_active are zero in k) belongs to the magnitude-free flavour", k_active);
    const int kblocks = k_active ? (k_active + 31) / 32 : BQH_KD / 32;
    COCOS_REQUIRE((rowstat == nullptr) == (mtile == nullptr) && (!rowstat || (q_scale_dev && !v_lo_mask_dev)), COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query_f16x3: rowstat + mtile come as a pair and belong to the magnitude-free flavour "
                  "(device-side operand scales, no lo mask)");
    COCOS_REQUIRE(kch && kcl && vph && vpl && gph && gpl && g_scale_dev && out && dout && lse && saved_logits && dqn &&
                      v_amax_dev && ds_scale_out_dev,
                  COCOS_ERR_INVALID, "corr_softmax_warp_bwd_query_f16x3: null pointer");
    COCOS_REQUIRE((q_scale_dev == nullptr) == (k_scale_dev == nullptr), COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query_f16x3: the device-side operand scales come as a pair");
    COCOS_REQUIRE((dsh == nullptr) == (dsl == nullptr) && (psh == nullptr) == (psl == nullptr), COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query_f16x3: plane pointers come in hi/lo pairs");
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1 && k_scale > 0.f && inv_temperature > 0.f, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query_f16x3: bad dims B=%d Nq=%d Nk=%d Cv=%d", B, Nq, Nk, Cv);
    COCOS_REQUIRE(K == 256 && Cv <= 160 && Nk % 8 == 0 && ((dsh == nullptr && psh == nullptr) || Nq % 8 == 0),
                  COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_bwd_query_f16x3: needs K == 256, Cv <= 160, Nk %% 8 == 0 and (with dS planes) "
                  "Nq %% 8 == 0 (K=%d Cv=%d Nk=%d Nq=%d)", K, Cv, Nk, Nq);
    COCOS_REQUIRE(!planes_blocked || (Nk % 128 == 0 && Nq % 32 == 0), COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query_f16x3: blocked planes need Nk %% 128 == 0 and Nq %% 32 == 0 (Nk=%d Nq=%d)",
                  Nk, Nq);
    const int cvb = (Cv + 31) / 32;
    COCOS_REQUIRE(CvPad == cvb * 32, COCOS_ERR_INVALID,
                  "corr_softmax_warp_bwd_query_f16x3: CvPad=%d, expected %d (Cv rounded up to 32)", CvPad, cvb * 32);
    COCOS_REQUIRE(cocos_corr_softmax_warp_saved_logits_bytes(1, Nq, Nk) < 0x7fffffffull &&
                      (size_t)K * Nk * 2 < 0x7fffffffull && (size_t)K * Nq * 4 < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED, "corr_softmax_warp_bwd_query_f16x3: per-sample tensor exceeds 2 GiB");
    for (const void* p : {kch, kcl, vph, vpl, gph, gpl, saved_logits})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID,
                      "corr_softmax_warp_bwd_query_f16x3: planes and saved logits must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
#define COCOS_ARGS                                                                                              \
    static_cast<const _Float16*>(kch), static_cast<const _Float16*>(kcl), static_cast<const _Float16*>(vph),   \
        static_cast<const _Float16*>(vpl), static_cast<const _Float16*>(gph), static_cast<const _Float16*>(gpl), \
        g_scale_dev, out, dout, lse, static_cast<const float*>(saved_logits), dqn, static_cast<_Float16*>(dsh),  \
        static_cast<_Float16*>(dsl), static_cast<_Float16*>(psh), static_cast<_Float16*>(psl), v_amax_dev,      \
        v_scale_dev, ds_scale_out_dev, v_lo_mask_dev, B, Nq, Nk, Cv, inv_temperature, k_scale, planes_blocked, q_scale_dev, \
        k_scale_dev, rowstat, mtile, d_pre, kblocks, s
    switch (cvb) {
        case 1: return launch_bq_f16x3<1>(COCOS_ARGS);
        case 2: return launch_bq_f16x3<2>(COCOS_ARGS);
        case 3: return launch_bq_f16x3<3>(COCOS_ARGS);
        case 4: return launch_bq_f16x3<4>(COCOS_ARGS);
        default: return launch_bq_f16x3<5>(COCOS_ARGS);
    }
#undef COCOS_ARGS
}
