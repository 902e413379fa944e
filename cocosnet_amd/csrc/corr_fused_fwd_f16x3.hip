// K2 forward, split-precision flavour: fused correlation -> softmax -> warp on v_mfma_f32_32x32x16_f16
// with every fp32 operand carried as f16 hi + f16 lo and three MFMA terms per product (gfx950).
//
// Same contract as corr_fused_fwd.hip (correspondence.py:281,:291,:304,:307,:318,:334; fp32 in HBM at the
// API, fp32 accumulate, fp32 out) — the difference is where the FLOPs run.  On MI355X the fp32 MFMA
// (32x32x2, 64 cycles) tops out at 157 TFLOP/s nominal / 134 sustained, while the f16 MFMA (32x32x16, 32
// cycles) issues 16x the FLOPs per cycle: computing   a.b = a_hi.b_hi + a_hi.b_lo + a_lo.b_hi   (the dropped
// a_lo.b_lo term is 2^-22 relative) costs 3 instructions of 32 cycles per 16 k instead of 8 of 64 —
// 5.3x fewer matrix-pipe cycles at fp32-class accuracy (measured against the fp64 oracle in
// tests/test_gpu_parity.py; tools/probes/f16x3_rate.hip for the instruction rates on this box).
// Plain f16/bf16 operands would NOT do: the logits are cos/0.01, a 2^-11 operand error becomes a
// 5 % error in softmax weights.
//
// Operand planes are prepared once per tensor by cocos_split_f16 (split_f16.hip):
//   q, k : [B, N, 256] position-major, pre-scaled by 2^4 (unit-norm columns -> lo plane stays normal)
//   v    : [B, Cv, N]  channel-major
// Decomposition (one workgroup = 4 waves = 128 queries, 1 wave/SIMD, as in the fp32 kernel):
//   * S^T (32 keys x 32 queries) = K_tile . Q : A = key rows from LDS (one ds_read_b128 = 8 channels of
//     one key), B = the wave's query slice, register-resident for the whole kernel (hi+lo: 128 VGPRs);
//   * lane&31 = query, so the online-softmax statistics are per-lane scalars; the 16 accumulator
//     registers of a lane are 16 keys.  P is split to f16 hi/lo IN REGISTERS and is directly the B
//     operand of the P.V MFMAs: accumulator registers 8t..8t+7 are the 8 k-slots of step t — the k
//     order inside an MFMA is free as long as A uses the same one, so the V tile is written to LDS with
//     its keys permuted to match (commit_v) and no cross-lane movement is needed;
//   * O^T (Cv x 32 queries) accumulates in 16*CVB fp32 registers.  The LAST row of the padded V tile (channel
//     32*CVB - 1, always a padding channel: CVB = Cv/32 + 1) is a row of ones, so the softmax denominator
//     l = sum_j p_j comes out of the P.V MFMAs as one more output row instead of 16 VALU adds per tile — and
//     numerator and denominator see exactly the same hi+lo truncation of P.
//   * training: the raw accumulator (k_scale^2 * cos) of every tile is kept for the query-side backward in a
//     PRIVATE tile-blocked layout — [key tile][32-query block][k = 0..3][lane][4 registers] — so that a wave
//     stores (and the backward loads) its 32x32 tile as four fully contiguous 1 KB instructions instead of
//     sixteen 4-byte-per-lane ones (the store tail of round 1 was instruction-bound, not bandwidth-bound).
#include "common.h"
#include <type_traits>

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int SP_BQ = 128;               // queries per workgroup
constexpr int SP_BK = 32;                // keys per tile
constexpr int SP_KD = 256;               // channels
constexpr int SP_KROW = SP_KD + 8;       // halfs per key row in LDS: 528 B -> conflict-free b128 reads
constexpr int SP_VROW = 40;              // halfs per channel row of the V tile: 80 B -> conflict-free
// Lazy rescale threshold and the power-of-two bias of the exponent: p = 2^(x - m + kPBias) <= 2^(thr + bias) = 2^15
// sits at the TOP of f16's range, so that the truncation floor of the hi/lo split (2^-24 absolute, one-sided)
// is 2^-33 relative to a row maximum — thousands of tail keys cannot bias a row sum by more than ~1e-7.
// Numerator (O) and denominator (l) carry the same factor, which cancels in out = O / l.
// Round 6 measured the trade behind these two (thr + bias = 15 is f16's range; -DCOCOS_RESCALE_THR=<t>): threshold 8 / 9 / 10 / 12 make
// the forward 1.3 / 2.0 / 2.0 / 2.2 % faster (fewer 200-instruction rescales of O) and move the elementwise-relative floor of the soft
// label map up by 2^(t - 6): at 9 the entries at 1e-8 leave the 1e-3 band (tests/test_gpu_mk3_sizes.py).  Kept at 6 / 9.
#ifndef COCOS_RESCALE_THR
#define COCOS_RESCALE_THR 6.0f
#endif
constexpr float kSplitRescaleThr = COCOS_RESCALE_THR;
constexpr float kPBias = 15.0f - COCOS_RESCALE_THR;

__device__ __forceinline__ f32x16 mfma16h(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f16x8 buf_load_h8(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ u32x4 buf_load_u4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ u32x2 buf_load_u2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
}

__device__ __forceinline__ u32x4 buf_load_u4s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, (int)soff, 0);
}
__device__ __forceinline__ u32x2 buf_load_u2s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, (int)soff, 0);
}

// p (>= 0, <= 2^15) -> f16 hi (round toward zero, so lo >= 0) and f16 lo' = 2^11 * (p - hi), two values per
// instruction (common.h: split_pair_rtz_lo_scaled; the V_hi * P_lo term reads the 2^-11-scaled copy of V's hi plane)
__device__ __forceinline__ void split_pair(float a, float b, f16x2& hi, f16x2& lo) {
    unsigned h, l;
    split_pair_rtz_lo_scaled(a, b, h, l);
    hi = __builtin_bit_cast(f16x2, h);
    lo = __builtin_bit_cast(f16x2, l);
}

// Optional phase timing (build with COCOS_EXTRA_HIPFLAGS=-DCOCOS_DEBUG_TIMING): shader-clock ticks spent by
// wave 0 of workgroup 0 in each phase of the tile loop, read back with cocos_debug_read_timing_fwd_f16x3().
// Ablation builds (debug only, results are WRONG): -DCOCOS_ABLATE=<bits>  1: no tile staging in the QK loop,
// 2: no operand re-reads from LDS, 4: no softmax arithmetic, 8: no logits store, 512: V lo plane skipped for channel
// blocks >= 1 (what exactly-representable label channels could save) — tools/ablate_ms.sh times them.
#ifndef COCOS_ABLATE
#define COCOS_ABLATE 0
#endif
// cache policy of the HWxHW streams (saved logits, dS'' / P planes): written once, read once by another kernel — `nt`
// (aux bit 1) keeps them from evicting the key/value tiles that the 32 workgroups of a sample share in their XCD's L2
#define COCOS_STREAM_AUX ((COCOS_ABLATE & 256) ? 0 : 2)
#ifdef COCOS_DEBUG_TIMING
__device__ long long g_phase_fwd_h[8];
#define FPH_T(var) const long long var = __builtin_readcyclecounter()
#define FPH_ADD(i, a, b) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_fwd_h[i] += (b) - (a); } while (0)
#else
#define FPH_T(var) do {} while (0)
#define FPH_ADD(i, a, b) do {} while (0)
#endif

// RAWM (round 4; the flavour of operands WITHOUT an a-priori magnitude, ops.softmax_attention): the running maximum is kept in
// the units of the RAW accumulator and the S accumulators START at -m instead of 0, so that exp2 sees (s - m) * scale formed
// from an exact difference.  With m kept in the log2 domain (the flavour of the unit-norm operands: |logit| <= 144) the
// difference is taken between s * scale and a ROUNDED m * scale: at |logit| ~ 1e10 (a randomly initialised SPADE generator's
// Attention block: theta ~ 4e5) one ulp is ~700 in the exponent — p overflowed f16 or vanished and the output was NaN.
// Same instruction count per element; extra work only where m changes (rare) and in the training forward's logits store.
template <int CVB, bool STORE_S, bool RAGGED, bool VLO0, bool RAWM = false, int KST = SP_KD / 16>
__device__ __forceinline__ void corr_fwd_f16x3_body(
    const _Float16* __restrict__ qh, const _Float16* __restrict__ ql, const _Float16* __restrict__ kh,
    const _Float16* __restrict__ kl, const _Float16* __restrict__ vh, const _Float16* __restrict__ vl,
    float* __restrict__ out, float* __restrict__ lse, float* __restrict__ lg, const float* __restrict__ v_scale,
    const unsigned* __restrict__ v_lo_mask, int B, int Nq, int Nk, int Cv, float scale_log2 /* inv_temperature * log2(e) / (q_scale * k_scale) */,
    float* __restrict__ rowstat = nullptr /* RAWM: [B][3][Nq] = (m_hi, m_lo in raw units, log2 l - bias) for the backward; nullable */,
    float* __restrict__ mtile = nullptr /* RAWM + STORE_S: [B][Nk/32 tiles][2][Nq] = (m_hi, m_lo) when the tile's logits were stored */) {
    // KST (RAWM): 16-channel steps that hold non-zero channels — the Attention block's K = C/8 = 32 or 64 channels sit zero-padded
    // in 256-channel planes; the QK MFMAs, fragment reads and key-tile fetches of the all-padding steps do not exist in the
    // KST = 2 / 4 / 8 instantiations (K <= 32 / 64 / 128).  (A run-time step count was tried first: 48 scalar branches per tile cost more than the MFMAs
    // they skipped — QK phase 3250 cycles per tile against 2320 without them, tools/phase_timing_attention.py.)
    static_assert(KST >= 1 && KST <= SP_KD / 16 && (RAWM || KST == SP_KD / 16), "KST < 16 belongs to the magnitude-free flavour");
    constexpr int CVP = CVB * 32;
    constexpr int KPLANE = SP_BK * SP_KROW;          // halfs per K plane per buffer
    constexpr int VPLANE = CVP * SP_VROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const kt = reinterpret_cast<_Float16*>(smem_raw);   // [2 buf][hi|lo][32 keys][KROW]
    _Float16* const vt = kt + 2 * 2 * KPLANE;                      // [2 buf][hi|lo|hi*2^-11][CVP][VROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;

    const int nqb = (Nq + SP_BQ - 1) / SP_BQ;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / nqb, qb = vb % nqb;
    const int i_lane = qb * SP_BQ + wave * 32 + c;   // this lane's query position

    const size_t qbytes = (size_t)Nq * SP_KD * 2, kbytes = (size_t)Nk * SP_KD * 2, vbytes = (size_t)Cv * Nk * 2;
    const __amdgpu_buffer_rsrc_t qh_rs = make_rsrc(qh + (size_t)b * Nq * SP_KD, qbytes);
    const __amdgpu_buffer_rsrc_t ql_rs = make_rsrc(ql + (size_t)b * Nq * SP_KD, qbytes);
    const __amdgpu_buffer_rsrc_t kh_rs = make_rsrc(kh + (size_t)b * Nk * SP_KD, kbytes);
    const __amdgpu_buffer_rsrc_t kl_rs = make_rsrc(kl + (size_t)b * Nk * SP_KD, kbytes);
    const __amdgpu_buffer_rsrc_t vh_rs = make_rsrc(vh + (size_t)b * Cv * Nk, vbytes);
    const __amdgpu_buffer_rsrc_t vl_rs = make_rsrc(vl + (size_t)b * Cv * Nk, vbytes);
    // saved logits, tile-blocked (see header): per sample ntiles x nqblk blocks of 4 KB
    const int nqblk = (Nq + 31) / 32;
    const size_t lg_bytes = (size_t)((Nk + SP_BK - 1) / SP_BK) * nqblk * 4096;
    const __amdgpu_buffer_rsrc_t lg_rs = make_rsrc(STORE_S ? reinterpret_cast<const char*>(lg) + (size_t)b * lg_bytes : nullptr,
                                                   STORE_S ? lg_bytes : 0);
    const unsigned lg_lane_off = i_lane < Nq ? (unsigned)((qb * 4 + wave) * 4096 + lane * 16) : kBufOob;

    // ---- resident query slice: B operand of step s = channels 16s + 8h .. +7 of query c ------------
    f16x8 qhr[SP_KD / 16], qlr[SP_KD / 16];
    {
        const unsigned q_off = i_lane < Nq ? (unsigned)(i_lane * SP_KD + h * 8) * 2u : kBufOob;
#pragma unroll
        for (int s = 0; s < SP_KD / 16; ++s) {
            qhr[s] = buf_load_h8(qh_rs, q_off + (unsigned)s * 32u);
            qlr[s] = buf_load_h8(ql_rs, q_off + (unsigned)s * 32u);
        }
        // park the slice in the accumulator half of the register file (MFMA B operands may be AGPRs): the
        // arch VGPRs are needed for the staging registers and the LDS read-ahead
#pragma unroll
        for (int s = 0; s < SP_KD / 16; ++s) {
            asm volatile("" : "+a"(qhr[s]));
            asm volatile("" : "+a"(qlr[s]));
        }
    }

    f32x16 o[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m_run = RAWM ? 0.f : -INFINITY;             // RAWM: raw-accumulator units, set by tile 0 ...
    float m_lo = 0.f;                                  // ... as an unevaluated sum m_run + m_lo (see the rescale block)
    const float thr_raw = kSplitRescaleThr / scale_log2;      // (RAWM) the lazy-rescale threshold in raw units
    // the thread that stages the last row of the padded V tile writes ones into its hi plane (see header)
    const bool ones_thread = tid >= 248;
    const u32x2 kOnes2 = u32x2{0x3C003C00u, 0x3C003C00u};
    const u32x2 kOnesS2 = u32x2{0x10001000u, 0x10001000u};      // ... and 2^-11 into the scaled copy of the hi plane
    auto unshift2 = [](u32x2 v) { return u32x2{pk_unshift_f16(v.x), pk_unshift_f16(v.y)}; };

    // ---- staging: global -> registers (in flight under the MFMAs of the previous tile) -> LDS -------
    // K: 32 keys x 512 B per plane = 1024 16-byte chunks, 4 per thread; a key row is contiguous in HBM.
    // V: CVP rows x 64 B per plane = CVP*8 8-byte chunks (4 keys), CVB per thread.
    u32x4 kst[2][4];
    u32x2 vst[2][CVB];
    auto fetch_k = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = u * 256 + tid, key = g >> 5, cc = g & 31;
            // rows past Nk lie past the end of the buffer: the descriptor returns zeros
            const unsigned off = (unsigned)((j0 + key) * SP_KD + cc * 8) * 2u;
            kst[0][u] = buf_load_u4(kh_rs, off);
            kst[1][u] = buf_load_u4(kl_rs, off);
        }
    };
    auto fetch_v = [&](int j0) {
#pragma unroll
        for (int u = 0; u < CVB; ++u) {
            const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
            unsigned off = (unsigned)(row * Nk + j0 + 4 * kq) * 2u;
            if (row >= Cv || j0 + 4 * kq >= Nk) off = kBufOob;     // Nk % 4 == 0 (checked by the launcher)
            vst[0][u] = buf_load_u2(vh_rs, off);
            vst[1][u] = buf_load_u2(vl_rs, off);
        }
    };
    auto commit_k = [&](int buf) {
        _Float16* base = kt + buf * 2 * KPLANE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = u * 256 + tid, key = g >> 5, cc = g & 31;
            *reinterpret_cast<u32x4*>(base + key * SP_KROW + cc * 8) = kst[0][u];
            *reinterpret_cast<u32x4*>(base + KPLANE + key * SP_KROW + cc * 8) = kst[1][u];
        }
    };
    auto commit_v = [&](int buf) {
        _Float16* base = vt + buf * 3 * VPLANE;
#pragma unroll
        for (int u = 0; u < CVB; ++u) {
            const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
            // keys 4kq..4kq+3 -> k-slots of the P.V MFMA (see header): step kq>>2, half kq&1, quad (kq>>1)&1
            const int slot = 16 * (kq >> 2) + 8 * (kq & 1) + 4 * ((kq >> 1) & 1);
            const bool ones = u == CVB - 1 && ones_thread;
            *reinterpret_cast<u32x2*>(base + row * SP_VROW + slot) = ones ? kOnes2 : vst[0][u];
            *reinterpret_cast<u32x2*>(base + VPLANE + row * SP_VROW + slot) = vst[1][u];
            *reinterpret_cast<u32x2*>(base + 2 * VPLANE + row * SP_VROW + slot) = ones ? kOnesS2 : unshift2(vst[0][u]);
        }
    };

    // Tile-invariant parts of the staging addresses (computed once); the tile-dependent part travels in the
    // scalar offset of the buffer instruction — non-ragged key counts only: the scalar offset is not bounds-
    // checked, so look-ahead tiles past the end are clamped to the last tile instead
    unsigned k_voff[4], v_voff[CVB];
    int k_lds[4], v_lds[CVB];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int g = u * 256 + tid, key = g >> 5, cc = g & 31;
        // (RAWM: chunks that hold only zero-padding channels are not fetched at all — an out-of-range lane costs no memory
        //  access and returns the zeros that are there anyway: 8x less L2 traffic and footprint for the Attention block's K = 32)
        k_voff[u] = (cc * 8 >= KST * 16) ? kBufOob : (unsigned)(key * SP_KD + cc * 8) * 2u;
        k_lds[u] = key * SP_KROW + cc * 8;
    }
#pragma unroll
    for (int u = 0; u < CVB; ++u) {
        const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
        v_voff[u] = row < Cv ? (unsigned)(row * Nk + 4 * kq) * 2u : kBufOob;
        v_lds[u] = row * SP_VROW + 16 * (kq >> 2) + 8 * (kq & 1) + 4 * ((kq >> 1) & 1);
    }

    const int ntiles = (Nk + SP_BK - 1) / SP_BK;
    fetch_k(0);
    fetch_v(0);
    commit_k(0);
    commit_v(0);
    // tile 1: issued in the order in which the loop consumes its pieces (the s_waitcnt vmcnt the compiler
    // puts in front of each LDS commit is the worst case over both ways into the loop)
    auto fetch_piece = [&](int i, int j0) {
        if (i < 8) {
            const int pl_ = i & 1, u = i >> 1;
            const int g = u * 256 + tid, key = g >> 5, cc = g & 31;
            kst[pl_][u] = buf_load_u4(pl_ ? kl_rs : kh_rs, (unsigned)((j0 + key) * SP_KD + cc * 8) * 2u);
        } else if (i - 8 < 2 * CVB) {
            const int pl_ = (i - 8) & 1, u = (i - 8) >> 1;
            const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
            unsigned off = (unsigned)(row * Nk + j0 + 4 * kq) * 2u;
            if (row >= Cv || j0 + 4 * kq >= Nk) off = kBufOob;
            vst[pl_][u] = buf_load_u2(pl_ ? vl_rs : vh_rs, off);
        }
    };
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx) {
        fetch_piece(sidx, SP_BK);
        if (sidx < 2) fetch_piece(16 + sidx, SP_BK);
    }
    __syncthreads();

    // Operand rings (A fragments from LDS): requested RA - 1 steps ahead; with four waves on the LDS pipe a
    // ds_read_b128 takes several hundred cycles to come back, far more than the 96 cycles of one step's MFMAs.
    // The rings live ACROSS the tile loop: the first fragments of tile t+1's key tile are requested at the end of
    // tile t's P.V loop, so the QK loop never starts cold (round 1 / step 1 read them right after the barrier:
    // ~450 cycles of every tile's QK phase were that bubble — tools/ablate_fwd.sh, profiles/r02_ablation_fwd.txt).
    constexpr int RA = 4, NS = SP_KD / 16;
    f16x8 ah[RA], al[RA];
    auto prefetch_k = [&](int buf) {
        const _Float16* kb = kt + buf * 2 * KPLANE + c * SP_KROW + h * 8;
#pragma unroll
        for (int s = 0; s < (RA - 1 < KST ? RA - 1 : KST); ++s) {
            ah[s] = *reinterpret_cast<const f16x8*>(kb + s * 16);
            al[s] = *reinterpret_cast<const f16x8*>(kb + KPLANE + s * 16);
        }
    };
    prefetch_k(0);

    // One barrier per tile, BETWEEN the two MFMA loops.  Key tile t+1 is committed during QK(t) and read from QK(t+1)
    // on (and by the prefetch at the end of P.V(t)): every wave has passed this barrier in between.  Value tile t+1 is
    // committed during P.V(t) into the buffer P.V(t-1) read: every wave finished P.V(t-1) before it reached this
    // barrier in tile t; it is read in P.V(t+1), after the barrier of tile t+1.
    // VLO0 (template): only value block 0 has a non-zero lo plane (one-hot label channels are exact in f16: `v_lo_mask`,
    // cocos_f16_plane_block_mask) — the V_lo * P_hi term, its fragment reads and its staging are skipped for the other
    // blocks: same result, 8 of 30 P.V MFMAs fewer.  The kernel (below) holds both flavours of this body and picks one
    // from the mask before anything else (a choice further inside — per tile or per loop copy — cost registers: measured slower).
    const std::integral_constant<bool, VLO0> vlo0_tag{};
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * SP_BK, buf = t & 1;
        const bool ragged = RAGGED && (j0 + SP_BK > Nk);
        _Float16* const kw = kt + (buf ^ 1) * 2 * KPLANE;
        _Float16* const vw = vt + (buf ^ 1) * 3 * VPLANE;
        const int jn = j0 + 2 * SP_BK;
        // One staged piece of tile t+1 to the other LDS buffer, and its register immediately takes the load for tile
        // t+2 — memory instructions are never issued as a burst.  i < 8: key-tile pieces (plane i & 1, chunk i >> 1),
        // issued in the QK loop; 8 <= i < 8 + 2*CVB: value-tile pieces, issued in the P.V loop.
        auto piece = [&](int i, auto vlo0_piece_tag) __attribute__((always_inline)) {
            constexpr bool PLO0 = decltype(vlo0_piece_tag)::value;     // value blocks >= 1 have an all-zero lo plane: not staged
            if (COCOS_ABLATE & 1) return;
            if (!RAGGED) {
                const int jc = (COCOS_ABLATE & 64) ? SP_BK : min(jn, Nk - SP_BK);       // look-ahead past the end re-reads the last tile
                if (i < 8) {
                    const int pl_ = i & 1, u = i >> 1;
                    *reinterpret_cast<u32x4*>(kw + pl_ * KPLANE + k_lds[u]) = kst[pl_][u];
                    kst[pl_][u] = buf_load_u4s(pl_ ? kl_rs : kh_rs, k_voff[u], (unsigned)jc * (unsigned)(SP_KD * 2));
                } else if (i - 8 < 2 * CVB) {
                    const int pl_ = (i - 8) & 1, u = (i - 8) >> 1;
                    if ((PLO0 || (COCOS_ABLATE & 512)) && pl_ == 1 && u >= 1) return;
                    const bool ones = pl_ == 0 && u == CVB - 1 && ones_thread;
                    *reinterpret_cast<u32x2*>(vw + pl_ * VPLANE + v_lds[u]) = ones ? kOnes2 : vst[pl_][u];
                    if (pl_ == 0) *reinterpret_cast<u32x2*>(vw + 2 * VPLANE + v_lds[u]) = ones ? kOnesS2 : unshift2(vst[0][u]);
                    vst[pl_][u] = buf_load_u2s(pl_ ? vl_rs : vh_rs, v_voff[u], (unsigned)jc * 2u);
                }
                return;
            }
            if (i < 8) {
                const int pl_ = i & 1, u = i >> 1;
                const int g = u * 256 + tid, key = g >> 5, cc = g & 31;
                *reinterpret_cast<u32x4*>(kw + pl_ * KPLANE + key * SP_KROW + cc * 8) = kst[pl_][u];
                kst[pl_][u] = buf_load_u4(pl_ ? kl_rs : kh_rs, (unsigned)((jn + key) * SP_KD + cc * 8) * 2u);
            } else if (i - 8 < 2 * CVB) {
                const int pl_ = (i - 8) & 1, u = (i - 8) >> 1;
                if (PLO0 && pl_ == 1 && u >= 1) return;
                const int g = u * 256 + tid, row = g >> 3, kq = g & 7;
                const int slot = 16 * (kq >> 2) + 8 * (kq & 1) + 4 * ((kq >> 1) & 1);
                const bool ones = pl_ == 0 && u == CVB - 1 && ones_thread;
                *reinterpret_cast<u32x2*>(vw + pl_ * VPLANE + row * SP_VROW + slot) = ones ? kOnes2 : vst[pl_][u];
                if (pl_ == 0) *reinterpret_cast<u32x2*>(vw + 2 * VPLANE + row * SP_VROW + slot) = ones ? kOnesS2 : unshift2(vst[0][u]);
                unsigned off = (unsigned)(row * Nk + jn + 4 * kq) * 2u;
                if (row >= Cv || jn + 4 * kq >= Nk) off = kBufOob;
                vst[pl_][u] = buf_load_u2(pl_ ? vl_rs : vh_rs, off);
            }
        };

        FPH_T(tp0);
        // ---- S^T = K_tile . Q : 16 k-steps x 3 terms; three accumulators (one per product term) used round-robin, so
        //      that consecutive MFMAs never depend on each other and every gap can carry a filler; the key-tile pieces of
        //      tile t+1 ride in the gaps ------------------------------------------------------------------------------
        // QK1 (round 6): the three product terms accumulate into ONE register set — a dependent chain of v_mfma_f32_32x32x16_f16 issues at
        // the full rate on gfx950 (tools/probes/mfma_operand_rate.hip), and the kernel is bound by its VALU instructions, not by MFMA
        // dependencies: 32 accumulator reads and 16 packed adds fewer per tile, forward 0.311 -> 0.293 ms same box (tools/ab_libs.sh;
        // -DCOCOS_QK_THREE_ACC restores the round-2 form).  The magnitude-free flavour keeps its two chains (they START at -m_hi / -m_lo).
#ifdef COCOS_QK_THREE_ACC
        constexpr bool QK1 = false;
#else
        constexpr bool QK1 = !RAWM;
#endif
        f32x16 sa, sb, sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = RAWM ? -m_run : 0.f; sb[r] = RAWM ? -m_lo : 0.f; sc[r] = 0.f; }
        {
            const _Float16* kb = kt + buf * 2 * KPLANE + c * SP_KROW + h * 8;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int cur = s % RA;
                const bool live = s < KST;      // (a compile-time fact after unrolling)
                if (live) sa = mfma16h(ah[cur], qhr[s], sa);
                if (!(COCOS_ABLATE & 2) && s + RA - 1 < KST) ah[(s + RA - 1) % RA] = *reinterpret_cast<const f16x8*>(kb + (s + RA - 1) * 16);
                __builtin_amdgcn_sched_barrier(0);
                if (live) { if (QK1) sa = mfma16h(ah[cur], qlr[s], sa); else sb = mfma16h(ah[cur], qlr[s], sb); }
                if (!(COCOS_ABLATE & 2) && s + RA - 1 < KST) al[(s + RA - 1) % RA] = *reinterpret_cast<const f16x8*>(kb + KPLANE + (s + RA - 1) * 16);
                __builtin_amdgcn_sched_barrier(0);
                if (live) { if (QK1) sa = mfma16h(al[cur], qhr[s], sa); else sc = mfma16h(al[cur], qhr[s], sc); }
                if ((s & 1) == 0) piece(s >> 1, std::false_type{});   // the 8 key-tile pieces, every other step
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        FPH_T(tp1);
        __syncthreads();   // key tile t+1 visible to everyone; value buffer of tile t-1 free (see the loop header)
        FPH_T(tp2);
        // ---- the P.V loop's first value fragments: their latency hides under the softmax arithmetic ------------------
        constexpr int NSV = 2 * CVB;                          // P.V step i = tt * CVB + cb
        f16x8 a_h[RA], a_l[RA], a_s[RA];                      // hi, lo and hi * 2^-11 fragments of the value tile
        const _Float16* vbase = vt + buf * 3 * VPLANE + c * SP_VROW + h * 8;
#pragma unroll
        for (int i = 0; i < RA - 1 && i < NSV; ++i) {
            a_h[i] = *reinterpret_cast<const f16x8*>(vbase + (i % CVB) * 32 * SP_VROW + (i / CVB) * 16);
            a_l[i] = *reinterpret_cast<const f16x8*>(vbase + VPLANE + (i % CVB) * 32 * SP_VROW + (i / CVB) * 16);
            a_s[i] = *reinterpret_cast<const f16x8*>(vbase + 2 * VPLANE + (i % CVB) * 32 * SP_VROW + (i / CVB) * 16);
        }
        // ---- online softmax (log2 domain), lazy rescale as in the fp32 kernel -------------------------
        // The row maximum is taken on the raw accumulator (scale_log2 > 0) and scaled once; the exponent is one
        // fma per element: p = 2^(s * scale_log2 - (m - kPBias)).
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = QK1 ? sa[r] : (sa[r] + sb[r]) + sc[r];
        if (ragged) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + acc_row_base(r) + 4 * h >= Nk) s0[r] = -INFINITY;
        }
        float tmax = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s0[r]);
        tmax = fmaxf(tmax, swap_half(tmax));
        if (!RAWM) tmax *= scale_log2;
        if (RAWM ? (t == 0 || __any(tmax > thr_raw)) : __any(tmax > m_run + kSplitRescaleThr)) {
            float alpha;
            if (RAWM) {
                // s0 is relative to m = m_run + m_lo already; the row's new maximum moves m by EXACTLY delta >= 0 (tile 0: to the
                // tile's own maximum, whatever its sign): the element that sets the maximum must come out at exactly 0 — at
                // |logit| ~ 1e9 one ulp of a single fp32 m is ~2700 in the exponent, its p would overflow f16 or vanish.  m is
                // therefore carried as an unevaluated sum (two TwoSums per update), the accumulators of the next tiles start at
                // -m_run (S hi.hi chain) and -m_lo (hi.lo chain), and the current tile moves by delta itself.
                const float delta = t == 0 ? tmax : fmaxf(tmax, 0.f);
                alpha = t == 0 ? 1.0f : fast_exp2(-delta * scale_log2);         // (tile 0: O is still zero)
                const float u = m_lo + delta, ub = u - m_lo;
                const float e1 = (m_lo - (u - ub)) + (delta - ub);              // TwoSum(m_lo, delta) = u + e1
                const float hs = m_run + u, hb = hs - m_run;
                const float e2 = (m_run - (hs - hb)) + (u - hb);                // TwoSum(m_run, u) = hs + e2
                m_run = hs;
                m_lo = e2 + e1;
#pragma unroll
                for (int r = 0; r < 16; ++r) s0[r] -= delta;
            } else {
                const float m_new = fmaxf(m_run, tmax);
                alpha = fast_exp2(m_run - m_new);
                m_run = m_new;
            }
            // (the pins keep the AGPR -> VGPR copies of O inside this rarely taken branch: hipcc otherwise
            //  hoists all of them above it, i.e. into every tile).  The ones row of V makes l one of O's rows:
            //  it is rescaled with the rest.
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb) asm volatile("" : "+a"(o[cb]));
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
#pragma unroll
            for (int cb = 0; cb < CVB; ++cb) asm volatile("" : "+a"(o[cb]));
        }
        if (STORE_S && !(COCOS_ABLATE & 8)) {
            // the wave's 32x32 tile of raw logits: four contiguous 1 KB stores (registers 4k..4k+3 of every lane)
            const unsigned soff = (COCOS_ABLATE & 32) ? 0u : (unsigned)(t * nqblk) * 4096u;
            // RAWM: the saved logits are RELATIVE raw accumulators, s - m_tile, with m_tile (the row's running maximum when the
            // tile was stored, hi and lo part) in mtile[b][t][0..1][query]: the backward forms s_rel + (m_tile - m_final) — exact differences.  (Absolute
            // logits s_rel + m would be rounded at |m|: one ulp of 2^27 raw units is ~70 in the exponent at |logit| ~ 1e9, and
            // P of the row's own maximum came back as 2^+-69.)
            if (RAWM && mtile && h == 0 && i_lane < Nq) {
                mtile[(((size_t)b * ntiles + t) * 2 + 0) * Nq + i_lane] = m_run;
                mtile[(((size_t)b * ntiles + t) * 2 + 1) * Nq + i_lane] = m_lo;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4, f32x4{s0[4 * k], s0[4 * k + 1], s0[4 * k + 2], s0[4 * k + 3]}), lg_rs,
                    (int)lg_lane_off, (int)(soff + (unsigned)k * 1024u), COCOS_STREAM_AUX);
        }
        float p[16];
        const float nmb = RAWM ? kPBias : kPBias - m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = (COCOS_ABLATE & 4) ? s0[r] * 1e-3f + 1.0f : fast_exp2(__builtin_fmaf(s0[r], scale_log2, nmb));

        FPH_T(tp3);
        // P -> f16 hi/lo: registers 8t..8t+7 are the k-slots of P.V step t
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                f16x2 a, bq;
                split_pair(p[8 * tt + j], p[8 * tt + j + 1], a, bq);
                ph[tt][j] = a[0]; ph[tt][j + 1] = a[1];
                pl[tt][j] = bq[0]; pl[tt][j + 1] = bq[1];
            }

        FPH_T(tp4);
        // ---- O^T += V . P : A = V tile rows (channels) with permuted keys, B = P.  Riding in the gaps: the value-tile
        //      pieces of tile t+1 (one per step) and, in the last steps, the first key fragments of tile t+1 ----------------
        {
            constexpr bool SKIPLO = VLO0 || (COCOS_ABLATE & 512);
#pragma unroll
            for (int i = 0; i < NSV; ++i) {
                const int tt = i / CVB, cb = i % CVB, cur = i % RA, n = i + RA - 1;
                if (!(COCOS_ABLATE & 2) && n < NSV) {
                    a_h[n % RA] = *reinterpret_cast<const f16x8*>(vbase + (n % CVB) * 32 * SP_VROW + (n / CVB) * 16);
                    a_s[n % RA] = *reinterpret_cast<const f16x8*>(vbase + 2 * VPLANE + (n % CVB) * 32 * SP_VROW + (n / CVB) * 16);
                    if (!SKIPLO || (n % CVB) == 0)
                        a_l[n % RA] = *reinterpret_cast<const f16x8*>(vbase + VPLANE + (n % CVB) * 32 * SP_VROW + (n / CVB) * 16);
                }
                o[cb] = mfma16h(a_h[cur], ph[tt], o[cb]);
                o[cb] = mfma16h(a_s[cur], pl[tt], o[cb]);      // (2^-11 V_hi) . (2^11 P_lo)
                if (!SKIPLO || cb == 0) o[cb] = mfma16h(a_l[cur], ph[tt], o[cb]);
                piece(8 + i, vlo0_tag);
                if (i == NSV - 1) prefetch_k(buf ^ 1);      // (NSV = 2: both in the same step)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // O's home is the accumulator file: without this the allocator keeps it in arch VGPRs for the (rare)
        // rescale multiply and copies all 16*CVB registers to AGPRs and back around every QK loop
#pragma unroll
        for (int cb = 0; cb < CVB; ++cb) asm volatile("" : "+a"(o[cb]));
        FPH_T(tp5);
        FPH_ADD(0, tp0, tp1); FPH_ADD(1, tp1, tp2); FPH_ADD(2, tp2, tp3); FPH_ADD(3, tp3, tp4);
        FPH_ADD(4, tp4, tp5);
    }

    // ---- epilogue: normalise, store channel-major [B,Cv,Nq], store row LSE --------------------------
    // l = the ones row of O: channel 32*CVB - 1 = register 15 of the upper half-wave's lanes
    // (the exchange is executed by ALL lanes before the select: a cross-lane read under a divergent `h ? :` would read
    //  inactive source lanes, i.e. zeros)
    const float l_own = o[CVB - 1][15];
    const float l_other = swap_half(l_own);
    const float l_tot = h ? l_own : l_other;
    const float inv_l = (v_scale ? 1.0f / *v_scale : 1.0f) / l_tot;
    if (RAWM && rowstat && i_lane < Nq && h == 0) {
        rowstat[((size_t)b * 3 + 0) * Nq + i_lane] = m_run;
        rowstat[((size_t)b * 3 + 1) * Nq + i_lane] = m_lo;
        rowstat[((size_t)b * 3 + 2) * Nq + i_lane] = log2f(l_tot) - kPBias;
    }
    if (RAWM) m_run = (m_run + m_lo) * scale_log2;     // (for the lse below: the log2 domain)
    if (i_lane < Nq) {
        float* out_b = out + (size_t)b * Cv * Nq;
#pragma unroll
        for (int cb = 0; cb < CVB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = cb * 32 + acc_row_base(r) + 4 * h;
                if (ch < Cv) out_b[(size_t)ch * Nq + i_lane] = o[cb][r] * inv_l;
            }
        if (h == 0) lse[(size_t)b * Nq + i_lane] = (m_run + log2f(l_tot) - kPBias) * kLn2;
    }
}

// The kernel: one launch holds BOTH flavours of the body (V_lo terms of the value blocks >= 1 issued / skipped); the
// workgroup-uniform choice is the first thing it does — data on the device (cocos_f16_plane_block_mask), no host round
// trip, and no second launch whose workgroups only look at the mask and leave (that cost 5-8 us per call).
// DUAL = false: no mask / a single value block — only the general flavour is compiled in.
template <int CVB, bool STORE_S, bool RAGGED, bool DUAL>
__global__ __launch_bounds__(256, 1) void corr_fwd_f16x3_kernel(
    const _Float16* __restrict__ qh, const _Float16* __restrict__ ql, const _Float16* __restrict__ kh,
    const _Float16* __restrict__ kl, const _Float16* __restrict__ vh, const _Float16* __restrict__ vl,
    float* __restrict__ out, float* __restrict__ lse, float* __restrict__ lg, const float* __restrict__ v_scale,
    const unsigned* __restrict__ v_lo_mask, int B, int Nq, int Nk, int Cv, float scale_log2,
    const float* __restrict__ q_scale_dev, const float* __restrict__ k_scale_dev, float* __restrict__ rowstat,
    float* __restrict__ mtile, int ksteps) {
    // operands without an a-priori magnitude (ops.softmax_attention: the reference's Attention block feeds raw 1x1-conv
    // outputs): their planes carry device-side power-of-two scales; scale_log2 then arrives WITHOUT the 1 / (q_scale k_scale)
    if (q_scale_dev) scale_log2 = scale_log2 / (*q_scale_dev * *k_scale_dev);
    if constexpr (DUAL) {
        if ((__builtin_amdgcn_readfirstlane(*v_lo_mask) & ~1u) == 0u)
            corr_fwd_f16x3_body<CVB, STORE_S, RAGGED, true>(qh, ql, kh, kl, vh, vl, out, lse, lg, v_scale, v_lo_mask, B, Nq, Nk, Cv, scale_log2);
        else
            corr_fwd_f16x3_body<CVB, STORE_S, RAGGED, false>(qh, ql, kh, kl, vh, vl, out, lse, lg, v_scale, v_lo_mask, B, Nq, Nk, Cv, scale_log2);
    } else {
        if (q_scale_dev) {        // the magnitude-free flavour (host side: no lo mask together with device-side operand scales)
#define COCOS_RAWM_BODY(KST_) corr_fwd_f16x3_body<CVB, STORE_S, RAGGED, false, true, KST_>(qh, ql, kh, kl, vh, vl, out, lse, lg, v_scale, \
                                                                                           v_lo_mask, B, Nq, Nk, Cv, scale_log2, rowstat, mtile)
            if constexpr (RAGGED) {      // (odd key counts: the general instantiation only)
                COCOS_RAWM_BODY(SP_KD / 16);
            } else {
                if (ksteps <= 2) COCOS_RAWM_BODY(2);
                else if (ksteps <= 4) COCOS_RAWM_BODY(4);
                else if (ksteps <= 8) COCOS_RAWM_BODY(8);
                else COCOS_RAWM_BODY(SP_KD / 16);
            }
#undef COCOS_RAWM_BODY
        } else
            corr_fwd_f16x3_body<CVB, STORE_S, RAGGED, false>(qh, ql, kh, kl, vh, vl, out, lse, lg, v_scale, v_lo_mask, B, Nq, Nk, Cv, scale_log2);
    }
}

template <int CVB, bool STORE_S, bool RAGGED, bool VLO0>
static int launch_f16x3_k(const _Float16* qh, const _Float16* ql, const _Float16* kh, const _Float16* kl,
                          const _Float16* vh, const _Float16* vl, float* out, float* lse, float* lg,
                          const float* v_scale, const unsigned* v_lo_mask, int B, int Nq, int Nk, int Cv, float scale_log2,
                          const float* qsd, const float* ksd, float* rowstat, float* mtile, int ksteps, hipStream_t stream) {
    auto kern = corr_fwd_f16x3_kernel<CVB, STORE_S, RAGGED, VLO0>;
    const size_t smem = (size_t)2 * (2 * SP_BK * SP_KROW + 3 * CVB * 32 * SP_VROW) * sizeof(_Float16);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int nqb = (Nq + SP_BQ - 1) / SP_BQ;
    hipLaunchKernelGGL(kern, dim3(B * nqb), dim3(256), smem, stream, qh, ql, kh, kl, vh, vl, out, lse, lg, v_scale,
                       v_lo_mask, B, Nq, Nk, Cv, scale_log2, qsd, ksd, rowstat, mtile, ksteps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

}  // namespace cocos

#ifdef COCOS_DEBUG_TIMING
extern "C" int cocos_debug_read_timing_fwd_f16x3(long long* host8, int reset) {
    using namespace cocos;
    COCOS_HIP_CHECK(hipDeviceSynchronize());
    COCOS_HIP_CHECK(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_phase_fwd_h), 8 * sizeof(long long)));
    if (reset) {
        long long z[8] = {0};
        COCOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_fwd_h), z, sizeof(z)));
    }
    return COCOS_OK;
}
#endif

template <int CVB, bool ST, bool RG, typename... A>
static int cocos_go_both(const _Float16* a, const _Float16* b2, const _Float16* c2, const _Float16* d, const _Float16* e,
                         const _Float16* f, float* out, float* lse, float* lgp, const float* v_scale_dev,
                         const unsigned* v_lo_mask_dev, A... rest) {
    using namespace cocos;
    if (CVB > 1 && v_lo_mask_dev)
        return launch_f16x3_k<CVB, ST, RG, (CVB > 1)>(a, b2, c2, d, e, f, out, lse, lgp, v_scale_dev, v_lo_mask_dev, rest...);
    return launch_f16x3_k<CVB, ST, RG, false>(a, b2, c2, d, e, f, out, lse, lgp, v_scale_dev, nullptr, rest...);
}

extern "C" size_t cocos_corr_softmax_warp_saved_logits_bytes(int B, int Nq, int Nk) {
    if (B < 1 || Nq < 1 || Nk < 1) return 0;
    return (size_t)B * ((Nk + 31) / 32) * ((Nq + 31) / 32) * 4096;
}

extern "C" int cocos_corr_softmax_warp_fwd_f16x3(const void* qh, const void* ql, const void* kh,
                                                 const void* kl, const void* vh, const void* vl, float* out,
                                                 float* lse, void* saved_logits, const float* v_scale_dev,
                                                 const unsigned* v_lo_mask_dev, int B,
                                                 int K, int Nq, int Nk, int Cv, float inv_temperature,
                                                 float operand_scale, const float* q_scale_dev, const float* k_scale_dev,
                                                 cocos_stream_t stream) {
    return cocos_corr_softmax_warp_fwd_f16x3_ex(qh, ql, kh, kl, vh, vl, out, lse, saved_logits, v_scale_dev, v_lo_mask_dev, B, K, Nq,
                                                Nk, Cv, inv_temperature, operand_scale, q_scale_dev, k_scale_dev, nullptr, nullptr, 0, stream);
}

extern "C" int cocos_corr_softmax_warp_fwd_f16x3_ex(const void* qh, const void* ql, const void* kh,
                                                    const void* kl, const void* vh, const void* vl, float* out,
                                                    float* lse, void* saved_logits, const float* v_scale_dev,
                                                    const unsigned* v_lo_mask_dev, int B,
                                                    int K, int Nq, int Nk, int Cv, float inv_temperature,
                                                    float operand_scale, const float* q_scale_dev, const float* k_scale_dev,
                                                    float* rowstat_out, float* mtile_out, int k_active, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(k_active >= 0 && k_active <= K && (k_active == 0 || k_active == K || q_scale_dev), COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd_f16x3: k_active=%d (channels >= k_active are zero in q and k) belongs to the magnitude-free flavour", k_active);
    const int ksteps = k_active ? (k_active + 15) / 16 : SP_KD / 16;
    COCOS_REQUIRE(qh && ql && kh && kl && vh && vl && out && lse, COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd_f16x3: null pointer");
    COCOS_REQUIRE(!(q_scale_dev && v_lo_mask_dev), COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd_f16x3: device-side operand scales (the magnitude-free flavour) and a V lo mask are exclusive");
    COCOS_REQUIRE((!rowstat_out && !mtile_out) || q_scale_dev, COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd_f16x3: rowstat_out / mtile_out belong to the magnitude-free flavour (device-side operand scales)");
    COCOS_REQUIRE(!(q_scale_dev && saved_logits) || mtile_out, COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd_f16x3: the magnitude-free flavour saves RELATIVE logits: mtile_out is required with saved_logits");
    COCOS_REQUIRE(B >= 1 && Nq >= 1 && Nk >= 1 && Cv >= 1 && operand_scale > 0.f && inv_temperature > 0.f,
                  COCOS_ERR_INVALID, "corr_softmax_warp_fwd_f16x3: bad dims B=%d Nq=%d Nk=%d Cv=%d", B, Nq, Nk, Cv);
    COCOS_REQUIRE(K == 256, COCOS_ERR_UNSUPPORTED, "corr_softmax_warp_fwd_f16x3: needs K == 256 (got %d)", K);
    COCOS_REQUIRE(Cv <= 159, COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_fwd_f16x3: Cv=%d > 159 (one padding channel carries the row sums)", Cv);
    COCOS_REQUIRE(Nk % 4 == 0, COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_fwd_f16x3: Nk=%d must be a multiple of 4 (use the fp32 entry point)", Nk);
    COCOS_REQUIRE((size_t)K * Nq * 2 < 0x7fffffffull && (size_t)K * Nk * 2 < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED, "corr_softmax_warp_fwd_f16x3: per-sample tensor exceeds 2 GiB");
    COCOS_REQUIRE(!saved_logits || cocos_corr_softmax_warp_saved_logits_bytes(1, Nq, Nk) < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED,
                  "corr_softmax_warp_fwd_f16x3: per-sample logits exceed 2 GiB; pass saved_logits = NULL");
    for (const void* p : {qh, ql, kh, kl})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "corr_softmax_warp_fwd_f16x3: q/k planes must be 16-byte aligned");
    for (const void* p : {vh, vl})
        COCOS_REQUIRE((reinterpret_cast<uintptr_t>(p) & 7u) == 0, COCOS_ERR_INVALID,
                      "corr_softmax_warp_fwd_f16x3: v planes must be 8-byte aligned");
    COCOS_REQUIRE(!saved_logits || aligned16(saved_logits), COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd_f16x3: saved_logits must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    COCOS_REQUIRE((q_scale_dev == nullptr) == (k_scale_dev == nullptr), COCOS_ERR_INVALID,
                  "corr_softmax_warp_fwd_f16x3: the device-side operand scales come as a pair");
    const float scale_log2 = q_scale_dev ? inv_temperature * kLog2e : inv_temperature * kLog2e / (operand_scale * operand_scale);
    const bool ragged = (Nk % SP_BK) != 0;
    float* lgp = static_cast<float*>(saved_logits);
    const _Float16 *a = static_cast<const _Float16*>(qh), *b2 = static_cast<const _Float16*>(ql),
                   *c2 = static_cast<const _Float16*>(kh), *d = static_cast<const _Float16*>(kl),
                   *e = static_cast<const _Float16*>(vh), *f = static_cast<const _Float16*>(vl);
    // with a mask and more than one value block the kernel holds both flavours and picks one from the device-side mask
#define COCOS_GO(CVB, ST, RG) \
    cocos_go_both<CVB, ST, RG>(a, b2, c2, d, e, f, out, lse, lgp, v_scale_dev, v_lo_mask_dev, B, Nq, Nk, Cv, scale_log2, \
                               q_scale_dev, k_scale_dev, rowstat_out, mtile_out, ksteps, s)
#define COCOS_CVB(CVB)                                                           \
    case CVB:                                                                    \
        if (lgp) return ragged ? COCOS_GO(CVB, true, true) : COCOS_GO(CVB, true, false); \
        return ragged ? COCOS_GO(CVB, false, true) : COCOS_GO(CVB, false, false);
    switch (Cv / 32 + 1) {       // one more channel than Cv: the ones row (see the kernel header)
        COCOS_CVB(1) COCOS_CVB(2) COCOS_CVB(3) COCOS_CVB(4)
        default: COCOS_CVB(5)
    }
#undef COCOS_CVB
#undef COCOS_GO
}
