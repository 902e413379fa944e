// K23: the theta / phi 1x1 projection FUSED with centre + L2-normalise — K0 and K1 in one kernel (gfx950, round 6).
//
// Replaces correspondence.py:272 + :277-280 (theta) and :282 + :287-289 (phi) on the match_kernel-1 / PONO_C path:
//     theta = conv1x1(x) ;  theta -= theta.mean(dim=1) ;  theta /= (||theta||_2 over channels + eps)
// and writes what the split correlation kernels read — the f16 hi / lo OPERAND PLANES of 16 * theta_n, position-major
// [B,N,256] and (training) channel-major [B,256,N] — plus the row norms.  The fp32 projection never reaches HBM: round 5
// wrote it (33.5 MB per tensor), read it back in K1 and took a max|x| pass in between; per tensor the chain moved 167 MB in
// three launches, this kernel moves 53 (x) + 67 (planes) in one.
//
// Design (the round-2..5 streaming kernel proj_stream_f16x3.hip keeps the weight in 256 accumulator
// registers per lane, i.e. ONE wave per SIMD and a 5 us prologue per workgroup):
//   * a WAVE owns 32 positions x ALL 256 output channels (8 accumulator tiles = 128 registers): the per-position mean and
//     norm over the channels are an in-lane sum over 128 registers + one lane ^ 32 exchange — no cross-wave traffic, and
//     the accumulators ARE the K1 input;
//   * a workgroup = 4 waves = 128 positions; 64.5 KB of LDS and <= 256 registers, so TWO workgroups share a CU (two waves
//     per SIMD): one's loads / conversions / epilogue run under the other's MFMAs.  Both projections of a forward call go
//     into ONE launch (2 x 256 workgroups at the benchmark shape = exactly two per CU);
//   * the weight is prepared once per step in FRAGMENT ORDER (cocos_proj_weight_frag_planes: [k-step][row block][plane]
//     [lane][8 halfs], scaled to [2^9, 2^10)): a 16 KB stage is a linear LDS-DMA copy (buffer_load ... lds, no registers),
//     the MFMA A operand a conflict-free lane-linear ds_read_b128;
//   * x goes HBM -> LDS by LDS-DMA as well (16-byte pieces: 8 k rows x 32 positions per instruction = whole 128-byte lines),
//     two stages ahead; a lane picks its B operand (8 k of ITS position) with 8 ds_read_b32 and splits it in registers.
//     All memory instructions of the loop are LDS-DMA: the waits are counted by hand (vmcnt(2) + s_barrier per k-step,
//     conv_nhwc_bf16.hip's scheme), nothing drains;
//   * epilogue: bias, centre, norm, split (round to nearest, as split_f16.hip); position-major rows leave as 16-byte stores
//     (v_permlane32_swap pairs the half-waves' 4-channel groups), channel-major rows as 4-byte stores of two neighbouring
//     positions (one DPP exchange per 4 channels).
// Arithmetic identical to proj_stream_f16x3.hip: a.b ~= ah.bh + ah.bl + al.bh on v_mfma_f32_32x32x16_f16, fp32 accumulate.
#include "proj_frag.h"

namespace cocos {

typedef _Float16 pn_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* pn_lds_ptr;
typedef unsigned int pn_u32x2 __attribute__((ext_vector_type(2)));

constexpr int PN_XSLOT = 16 * 32 * 4;          // bytes of one x slot of a wave: [16 k][32 positions] fp32
constexpr int PN_XSLOTS = 3;                   // x stages in flight / being read, per wave
constexpr int PN_SMEM = 2 * PN_WSTAGE + 4 * PN_XSLOTS * PN_XSLOT;     // 32 + 24 KB

__device__ __forceinline__ float pn_scale_from_amax(const float* amax) {
    if (!amax) return 1.0f;
    const float a = *amax;
    if (!(a > 0.f) || !(a < INFINITY)) return 1.0f;
    int e;
    frexpf(a, &e);
    return ldexpf(1.0f, 10 - e);
}

struct PnProb {
    const float* x;          // [B][K][N]
    const void* wfrag;       // nst stages of PN_WSTAGE bytes (cocos_proj_weight_frag_planes)
    const float* w_scale;    // device cell: the power of two the weight planes were multiplied with
    const float* bias;       // [256] or null
    const float* x_amax;     // device cell: max|x| (null: 1)
    float* norm;             // [B][N]
    _Float16 *ph, *pl;       // position-major planes [B][N][256]
    _Float16 *ch, *cl;       // channel-major planes [B][256][N] (null: not written)
    float* sum2;             // RAW flavour: sum_c y^2 [B][N] (norm then holds sum_c y)
    float* y_scale;          // RAW flavour: device cell that receives the power of two the planes were multiplied with
};
struct PnArgs {
    PnProb p[2];
    int nprob, B, K, N, nst;
    int center;
    float eps, plane_scale;
};

// weight [256][K] fp32 -> fragment-ordered f16 hi / lo planes (proj_frag.h).  grid (nst, 8), 64 threads.
__global__ __launch_bounds__(64) void proj_weight_frag_kernel(const float* __restrict__ w, const float* __restrict__ w_amax,
                                                              unsigned char* __restrict__ out, float* __restrict__ w_scale, int K,
                                                              _Float16* __restrict__ t_hi, _Float16* __restrict__ t_lo) {
    const int s = blockIdx.x, blk = blockIdx.y, l = threadIdx.x;
    const float sc = pn_scale_from_amax(w_amax);
    if (s == 0 && blk == 0 && l == 0) *w_scale = sc;
    pf_weight_frag_item(w, sc, out, K, t_hi, t_lo, s, blk, l);
}

// Both fragment layouts (W for K23, W^T for K24) of up to two projections in ONE launch: the benchmark step ran four 5 us
// launches for them.  256 threads = 4 items of 64 lanes; items [0, nst * 8) of a problem are (stage, row block) pairs of W,
// the next PB_NST * 2 * PB_HB are those of W^T (absent when wtfrag is null).
struct PwProb {
    const float* w;
    const float* w_amax;
    unsigned char* wfrag;
    float* w_scale;
    _Float16 *t_hi, *t_lo;
    unsigned char* wtfrag;
};
struct PwArgs {
    PwProb p[2];
    int nprob, K, nst, items_per_prob;
};
__global__ __launch_bounds__(256) void proj_weight_prep_kernel(const PwArgs a) {
    const int l = threadIdx.x & 63;
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pi = item >= a.items_per_prob ? 1 : 0;
    item -= pi * a.items_per_prob;
    if (pi >= a.nprob || item >= a.items_per_prob) return;
    const PwProb P = pi ? a.p[1] : a.p[0];
    const float sc = pn_scale_from_amax(P.w_amax);
    if (item == 0 && l == 0) *P.w_scale = sc;
    const int nfrag = a.nst * 8;
    if (item < nfrag) {
        pf_weight_frag_item(P.w, sc, P.wfrag, a.K, P.t_hi, P.t_lo, item >> 3, item & 7, l);
    } else if (P.wtfrag) {
        const int j = item - nfrag;
        pf_weight_tfrag_item(P.w, sc, P.wtfrag, a.K, j / (2 * PB_HB), j % (2 * PB_HB), l);
    }
}

// RAW (K25, match_kernel 3): no centring / normalisation — the planes hold s * (W x + bias) itself, with the power of two s chosen
// from an a-priori bound on |y| (K max|w| max|x| + max|bias|: known before the first product, so planes and sums leave in the
// same pass; the bound is loose by a few binades, which a hi/lo pair of f16 absorbs — see the epilogue), and the per-position
// sums sum_c y, sum_c y^2 that K12's statistics are box sums of (unfold3_stats.hip) replace the norm.
template <bool WANT_CHAN, bool RAW = false>
__global__ __launch_bounds__(256, 2) void proj_norm_fwd_kernel(const PnArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char pn_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int tiles = a.N / 128, per_prob = a.B * tiles;
    const int vb = blockIdx.x;
    const int pi = vb >= per_prob ? 1 : 0;                        // (workgroup-uniform)
    const PnProb P = pi ? a.p[1] : a.p[0];
    const int rem = vb - pi * per_prob;
    const int b = rem / tiles, n0 = (rem - b * tiles) * 128 + wave * 32;     // this wave's 32 positions
    const int K = a.K, N = a.N, nst = a.nst;

    unsigned char* const wbuf = pn_smem;                                                  // [2][PN_WSTAGE]
    unsigned char* const xw = pn_smem + 2 * PN_WSTAGE + wave * (PN_XSLOTS * PN_XSLOT);    // this wave's x slots
    // x slots start as zeros: rows k >= K of the last stage are switched off in the DMA (whatever an out-of-range piece leaves in
    // LDS is multiplied with the weight planes' zero padding: it must be finite)
#pragma unroll
    for (int i = 0; i < PN_XSLOTS * PN_XSLOT / (64 * 16); ++i)
        *reinterpret_cast<u32x4*>(xw + (i * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(P.x + (size_t)b * K * N, (size_t)K * N * 4);
    const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(P.wfrag, (size_t)nst * PN_WSTAGE);
    const __amdgpu_buffer_rsrc_t none_rs = make_rsrc(P.wfrag, 0);        // every access out of range: the stages beyond the end
    // x pieces: instruction i of a stage = k rows 8i .. 8i+7 (lane >> 3) x positions 4 (lane & 7) .. + 3
    const int xk = lane >> 3;
    const unsigned x_voff = (unsigned)(xk * N + n0 + 4 * (lane & 7)) * 4u;
    const unsigned w_voff = (unsigned)(wave * 4096 + lane * 16);

    auto issue_w = [&](int s) {               // stage s -> wbuf[s & 1]; this wave's 4 KB of it
        const bool ok = s < nst;
        unsigned char* dst = wbuf + (s & 1) * PN_WSTAGE + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? w_rs : none_rs, (pn_lds_ptr)(dst + i * 1024), 16, (int)w_voff,
                                                     (int)((unsigned)s * PN_WSTAGE + i * 1024), 0, 0);      // (an instruction offset would move the LDS side too)
    };
    auto issue_x = [&](int s, int slot) {     // k-step s of this wave's positions -> slot
        const bool ok = s < nst;
        unsigned char* dst = xw + slot * PN_XSLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = 16 * s + 8 * i + xk;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? x_rs : none_rs, (pn_lds_ptr)(dst + i * 1024), 16,
                                                     (int)(k < K ? x_voff : kBufOob), (int)((unsigned)(16 * s + 8 * i) * N * 4u), 0, 0);
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const float sa = *P.w_scale, sb = pn_scale_from_amax(P.x_amax);

    // prologue: [x(0)] [W(0)] [x(1)] — from here on every k-step issues [W(s+1)] [x(s+2)], so that at the top of k-step s
    // "all but the last two instructions" = W(s) and x(s) have landed
    issue_x(0, 0);
    issue_w(0);
    issue_x(1, 1);
    int slot = 0;
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
        // this wave's pieces of W(s) and its x(s) are in LDS; behind the barrier so are the other waves' pieces of W(s), and
        // every wave is done reading W(s - 1)'s buffer (the one W(s + 1) goes to)
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_w(s + 1);
        int slot2 = slot + 2;
        if (slot2 >= PN_XSLOTS) slot2 -= PN_XSLOTS;
        issue_x(s + 2, slot2);                 // (that slot held x(s - 1): this wave's own reads of it are long complete)
        // ---- B operand: this lane's position, k = 16 s + 8 h .. + 7 ----
        const float* xs = reinterpret_cast<const float*>(xw + slot * PN_XSLOT) + (8 * h) * 32 + c;
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = xs[j * 32];
        unsigned bhw[4], blw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair_rn(xv[2 * q] * sb, xv[2 * q + 1] * sb, bhw[q], blw[q]);
        const pn_f16x8 bh = __builtin_bit_cast(pn_f16x8, u32x4{bhw[0], bhw[1], bhw[2], bhw[3]});
        const pn_f16x8 bl = __builtin_bit_cast(pn_f16x8, u32x4{blw[0], blw[1], blw[2], blw[3]});
        // ---- 8 row blocks x 3 terms, two accumulator chains at a time (dependent MFMAs are never neighbours) ----
        const unsigned char* wb = wbuf + (s & 1) * PN_WSTAGE + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const pn_f16x8 ah0 = *reinterpret_cast<const pn_f16x8*>(wb + (i * 2 + 0) * 1024);
            const pn_f16x8 al0 = *reinterpret_cast<const pn_f16x8*>(wb + (i * 2 + 1) * 1024);
            const pn_f16x8 ah1 = *reinterpret_cast<const pn_f16x8*>(wb + (i * 2 + 2) * 1024);
            const pn_f16x8 al1 = *reinterpret_cast<const pn_f16x8*>(wb + (i * 2 + 3) * 1024);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh, acc[i], 0, 0, 0);
            acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh, acc[i + 1], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl, acc[i], 0, 0, 0);
            acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl, acc[i + 1], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh, acc[i], 0, 0, 0);
            acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh, acc[i + 1], 0, 0, 0);
        }
        slot = slot + 1 == PN_XSLOTS ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the out-of-range pieces issued by the last two k-steps)

    // ---- epilogue: bias, centre, norm (this lane's position: 128 of the 256 channels here, 128 in lane ^ 32) ----
    const float oscale = 1.0f / (sa * sb);
    const __amdgpu_buffer_rsrc_t bias_rs = make_rsrc(P.bias, P.bias ? (size_t)PN_M * 4 : 0);
    float sum = 0.f, bmax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = buf_load4(bias_rs, (unsigned)(i * 32 + 8 * g + 4 * h) * 4u);      // (no bias: an empty descriptor -> zeros)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = __builtin_fmaf(acc[i][4 * g + e], oscale, b4[e]);
                acc[i][4 * g + e] = t;
                sum += t;
                if (RAW) bmax = fmaxf(bmax, fabsf(b4[e]));
            }
        }
    const float mean = (!RAW && a.center) ? (sum + swap_half(sum)) * (1.0f / (float)PN_M) : 0.f;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[i][r] - mean;
            acc[i][r] = d;
            ss = __builtin_fmaf(d, d, ss);
        }
    ss += swap_half(ss);
    float u, pscale;
    if (RAW) {
        // |y| <= K max|w| max|x| + max|bias|, with max|w| < 2^10 / sa and max|x| < 2^10 / sb (the scales put the maxima into
        // [2^9, 2^10)): every lane of every workgroup computes the same bound, hence the same power of two — no pass over y.
        // The bound lands in [2^13, 2^14); the real maximum sits a few binades below (random weights: ~2^5), where hi keeps its 11
        // bits and lo = the next 11 down to 2^-24: values 2^12 below the maximum still carry 22 bits.
        bmax = fmaxf(bmax, swap_half(bmax));
        const float bound = (1024.0f / sa) * (1024.0f / sb) * (float)K + bmax;
        int e = 0;
        if (bound > 0.f && bound < INFINITY) frexpf(bound, &e); else e = 14;
        pscale = ldexpf(1.0f, 14 - e);
        u = 1.0f;
        const float sum_t = sum + swap_half(sum);          // (the exchange is executed by all lanes, before the select)
        if (h == 0) {
            P.norm[(size_t)b * N + n0 + c] = sum_t;
            P.sum2[(size_t)b * N + n0 + c] = ss;
        }
        if (rem == 0 && tid == 0) *P.y_scale = pscale;
    } else {
        const float nrm = sqrtf(ss);
        u = 1.0f / (nrm + a.eps);
        pscale = a.plane_scale;
        if (h == 0) P.norm[(size_t)b * N + n0 + c] = nrm;
    }

    // ---- planes of plane_scale * y ----
    const size_t plane_b = (size_t)b * N * PN_M;
    const __amdgpu_buffer_rsrc_t ph_rs = make_rsrc(P.ph + plane_b, (size_t)N * PN_M * 2);
    const __amdgpu_buffer_rsrc_t pl_rs = make_rsrc(P.pl + plane_b, (size_t)N * PN_M * 2);
    const __amdgpu_buffer_rsrc_t ch_rs = make_rsrc(WANT_CHAN ? P.ch + plane_b : nullptr, WANT_CHAN ? (size_t)N * PN_M * 2 : 0);
    const __amdgpu_buffer_rsrc_t cl_rs = make_rsrc(WANT_CHAN ? P.cl + plane_b : nullptr, WANT_CHAN ? (size_t)N * PN_M * 2 : 0);
    const unsigned pos_voff = (unsigned)((n0 + c) * PN_M + (h ? 8 : 0)) * 2u;         // + (32 i + 8 g) * 2 (g even)
    const bool odd = (c & 1) != 0;
    const unsigned chan_voff = (unsigned)(((odd ? 2 : 0) + 4 * h) * N + n0 + (c & ~1)) * 2u;      // + (32 i + 8 g (+ 1)) * N * 2
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned hw[4][2], lw[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float y0 = (acc[i][4 * g + 2 * q] * u) * pscale, y1 = (acc[i][4 * g + 2 * q + 1] * u) * pscale;
                split_pair_rn(y0, y1, hw[g][q], lw[g][q]);
            }
        if (WANT_CHAN) {
            // channel-major: lane c holds channels ch .. ch+3 of position c.  Even lanes keep channels ch, ch+1 for positions (c, c+1),
            // odd lanes channels ch+2, ch+3 for positions (c-1, c): one neighbour exchange per plane and 4 channels
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const unsigned w01 = pl ? lw[g][0] : hw[g][0], w23 = pl ? lw[g][1] : hw[g][1];
                    const unsigned send = odd ? w01 : w23, keep = odd ? w23 : w01;
                    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xf, 0xf, false);   // lane ^ 1
                    const unsigned first = odd ? recv : keep, second = odd ? keep : recv;
                    const unsigned va = (first & 0xffffu) | (second << 16);
                    const unsigned vb2 = (first >> 16) | (second & 0xffff0000u);
                    const unsigned so = (unsigned)((32 * i + 8 * g) * N) * 2u;
                    __builtin_amdgcn_raw_buffer_store_b32(va, pl ? cl_rs : ch_rs, (int)chan_voff, (int)so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(vb2, pl ? cl_rs : ch_rs, (int)chan_voff, (int)(so + (unsigned)N * 2u), 0);
                }
        }
        // position-major: the half-waves hold alternating 4-channel groups of the same position; v_permlane32_swap turns groups
        // (g, g+1) into 8 consecutive channels per lane = one 16-byte store (guide T21)
#pragma unroll
        for (int g = 0; g < 4; g += 2)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                unsigned ax = pl ? lw[g][0] : hw[g][0], ay = pl ? lw[g][1] : hw[g][1];
                unsigned bx = pl ? lw[g + 1][0] : hw[g + 1][0], by = pl ? lw[g + 1][1] : hw[g + 1][1];
                const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                ax = rx[0]; bx = rx[1]; ay = ry[0]; by = ry[1];
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ax, ay, bx, by}, pl ? pl_rs : ph_rs, (int)pos_voff,
                                                       (int)((32 * i + 8 * g) * 2), 0);
            }
    }
}

}  // namespace cocos

extern "C" size_t cocos_proj_weight_frag_bytes(int K) {
    return K < 1 ? 0 : (size_t)((K + 15) / 16) * cocos::PN_WSTAGE;
}

// w [256][K] fp32 (+ device cell max|w|) -> fragment-ordered planes for cocos_proj_center_l2norm_planes_f16x3 and the scale
// they were multiplied with (*w_scale); t_hi / t_lo (nullable pair): the transposed planes [K][256] of the same scaled numbers.
extern "C" int cocos_proj_weight_frag_planes(const float* w, const float* w_amax, void* wfrag, float* w_scale, void* t_hi, void* t_lo,
                                             int M, int K, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(w && w_amax && wfrag && w_scale, COCOS_ERR_INVALID, "proj_weight_frag_planes: null pointer");
    COCOS_REQUIRE((t_hi == nullptr) == (t_lo == nullptr), COCOS_ERR_INVALID, "proj_weight_frag_planes: the transposed planes come as a hi/lo pair");
    COCOS_REQUIRE(M == PN_M && K >= 1 && K <= 4096, COCOS_ERR_UNSUPPORTED, "proj_weight_frag_planes: needs M == 256, K <= 4096 (M=%d K=%d)", M, K);
    COCOS_REQUIRE(aligned16(wfrag), COCOS_ERR_INVALID, "proj_weight_frag_planes: planes must be 16-byte aligned");
    hipLaunchKernelGGL(proj_weight_frag_kernel, dim3((K + 15) / 16, 8), dim3(64), 0, as_stream(stream), w, w_amax,
                       static_cast<unsigned char*>(wfrag), w_scale, K, static_cast<_Float16*>(t_hi), static_cast<_Float16*>(t_lo));
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// Both layouts for up to two projections of the same shape in one launch (see proj_weight_prep_kernel): per projection the weight
// [256][K], its max|w| cell, the K23 planes (cocos_proj_weight_frag_bytes(K)) + the scale cell, optionally the transposed
// row-major planes [K][256] (t_hi / t_lo) and optionally the K24 planes of W^T (cocos_proj_weight_tfrag_bytes(); needs K <= 448).
extern "C" int cocos_proj_weight_prep_pair(int nprob, const float* w0, const float* w_amax0, void* wfrag0, float* w_scale0, void* t_hi0,
                                           void* t_lo0, void* wtfrag0, const float* w1, const float* w_amax1, void* wfrag1,
                                           float* w_scale1, void* t_hi1, void* t_lo1, void* wtfrag1, int M, int K,
                                           cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(nprob == 1 || nprob == 2, COCOS_ERR_INVALID, "proj_weight_prep_pair: nprob = %d", nprob);
    COCOS_REQUIRE(w0 && w_amax0 && wfrag0 && w_scale0 && (nprob == 1 || (w1 && w_amax1 && wfrag1 && w_scale1)), COCOS_ERR_INVALID,
                  "proj_weight_prep_pair: null pointer");
    COCOS_REQUIRE((t_hi0 == nullptr) == (t_lo0 == nullptr) && (t_hi1 == nullptr) == (t_lo1 == nullptr), COCOS_ERR_INVALID,
                  "proj_weight_prep_pair: the transposed planes come as a hi/lo pair");
    COCOS_REQUIRE(M == PN_M && K >= 1 && K <= 4096, COCOS_ERR_UNSUPPORTED, "proj_weight_prep_pair: needs M == 256, K <= 4096 (M=%d K=%d)", M, K);
    const bool any_t = wtfrag0 || (nprob == 2 && wtfrag1);
    COCOS_REQUIRE(!any_t || K <= 2 * PB_HB * 32, COCOS_ERR_UNSUPPORTED, "proj_weight_prep_pair: the W^T planes need K <= 448 (K=%d)", K);
    for (const void* p : {(const void*)wfrag0, (const void*)wfrag1, (const void*)wtfrag0, (const void*)wtfrag1})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "proj_weight_prep_pair: planes must be 16-byte aligned");
    PwArgs a;
    a.p[0] = PwProb{w0, w_amax0, static_cast<unsigned char*>(wfrag0), w_scale0, static_cast<_Float16*>(t_hi0), static_cast<_Float16*>(t_lo0),
                    static_cast<unsigned char*>(wtfrag0)};
    a.p[1] = nprob == 2 ? PwProb{w1, w_amax1, static_cast<unsigned char*>(wfrag1), w_scale1, static_cast<_Float16*>(t_hi1),
                                 static_cast<_Float16*>(t_lo1), static_cast<unsigned char*>(wtfrag1)} : a.p[0];
    a.nprob = nprob; a.K = K; a.nst = (K + 15) / 16;
    a.items_per_prob = a.nst * 8 + (any_t ? PB_NST * 2 * PB_HB : 0);
    const int items = nprob * a.items_per_prob;
    hipLaunchKernelGGL(proj_weight_prep_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, as_stream(stream), a);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// One launch for up to two projections of the same shape (theta and phi of a forward call): x [B,K,N] fp32 -> row norms
// [B,N] and the operand planes of plane_scale * normalise(centre(W x + bias)): position-major [B,N,256] always, channel-major
// [B,256,N] when chan_hi / chan_lo are given (both problems alike).  center_over_channels: 1 (PONO_C) or 2 (no centring).
extern "C" int cocos_proj_center_l2norm_planes_f16x3(
    int nprob, const float* x0, const void* wfrag0, const float* w_scale0, const float* bias0, const float* x_amax0, float* norm0,
    void* pos_hi0, void* pos_lo0, void* chan_hi0, void* chan_lo0, const float* x1, const void* wfrag1, const float* w_scale1,
    const float* bias1, const float* x_amax1, float* norm1, void* pos_hi1, void* pos_lo1, void* chan_hi1, void* chan_lo1, int B,
    int K, int N, int center_over_channels, float eps, float plane_scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(nprob == 1 || nprob == 2, COCOS_ERR_INVALID, "proj_center_l2norm_planes: nprob = %d", nprob);
    COCOS_REQUIRE(x0 && wfrag0 && w_scale0 && norm0 && pos_hi0 && pos_lo0, COCOS_ERR_INVALID, "proj_center_l2norm_planes: null pointer");
    COCOS_REQUIRE(nprob == 1 || (x1 && wfrag1 && w_scale1 && norm1 && pos_hi1 && pos_lo1), COCOS_ERR_INVALID,
                  "proj_center_l2norm_planes: null pointer (second projection)");
    COCOS_REQUIRE((chan_hi0 == nullptr) == (chan_lo0 == nullptr) && (nprob == 1 || ((chan_hi1 == nullptr) == (chan_hi0 == nullptr) &&
                  (chan_lo1 == nullptr) == (chan_lo0 == nullptr))), COCOS_ERR_INVALID,
                  "proj_center_l2norm_planes: channel-major planes come as hi/lo pairs, for both projections or neither");
    COCOS_REQUIRE(B >= 1 && K >= 1 && K <= 4096 && N >= 128 && N % 128 == 0 && plane_scale > 0.f, COCOS_ERR_UNSUPPORTED,
                  "proj_center_l2norm_planes: needs N %% 128 == 0, K <= 4096 (B=%d K=%d N=%d)", B, K, N);
    COCOS_REQUIRE(center_over_channels == 1 || center_over_channels == 2, COCOS_ERR_UNSUPPORTED,
                  "proj_center_l2norm_planes: centring over channels (1) or none (2), got %d", center_over_channels);
    COCOS_REQUIRE((size_t)K * N * 4 < 0x7fffffffull && (long long)nprob * B * (N / 128) < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED,
                  "proj_center_l2norm_planes: one sample exceeds 2 GiB");
    for (const void* p : {(const void*)x0, wfrag0, (const void*)pos_hi0, (const void*)pos_lo0, (const void*)x1, wfrag1,
                          (const void*)pos_hi1, (const void*)pos_lo1, (const void*)bias0, (const void*)bias1})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "proj_center_l2norm_planes: pointers must be 16-byte aligned");
    PnArgs a;
    a.p[0] = PnProb{x0, wfrag0, w_scale0, bias0, x_amax0, norm0, static_cast<_Float16*>(pos_hi0), static_cast<_Float16*>(pos_lo0),
                    static_cast<_Float16*>(chan_hi0), static_cast<_Float16*>(chan_lo0), nullptr, nullptr};
    a.p[1] = nprob == 2 ? PnProb{x1, wfrag1, w_scale1, bias1, x_amax1, norm1, static_cast<_Float16*>(pos_hi1),
                                 static_cast<_Float16*>(pos_lo1), static_cast<_Float16*>(chan_hi1), static_cast<_Float16*>(chan_lo1),
                                 nullptr, nullptr}
                        : a.p[0];
    a.nprob = nprob; a.B = B; a.K = K; a.N = N; a.nst = (K + 15) / 16;
    a.center = center_over_channels == 1;
    a.eps = eps; a.plane_scale = plane_scale;
    const dim3 grid((unsigned)(nprob * B * (N / 128)));
    auto launch = [&](auto kern) -> int {
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PN_SMEM));
        hipLaunchKernelGGL(kern, grid, dim3(256), PN_SMEM, as_stream(stream), a);
        return COCOS_OK;
    };
    const int rc = chan_hi0 ? launch(proj_norm_fwd_kernel<true>) : launch(proj_norm_fwd_kernel<false>);
    if (rc != COCOS_OK) return rc;
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// K25 (round 6, match_kernel 3): the projection with NO normalisation — up to two projections of one shape per launch.  Per
// projection: x [B,K,N] fp32, wfrag / w_scale (cocos_proj_weight_frag_planes), bias [256] (nullable), x_amax (device cell) ->
//   sum1 [B,N] = sum_c y,  sum2 [B,N] = sum_c y^2   (y = W x + bias; cocos_unfold3_stats_finish_pair turns them into mu / a / nrm),
//   the planes of s * y, position-major [B,N,256] and (chan_hi / chan_lo, nullable pair) channel-major [B,256,N], and *y_scale = s,
//   the power of two chosen from the a-priori bound K max|w| max|x| + max|bias| (see the kernel).  The fp32 projection is not written.
extern "C" int cocos_proj_raw_planes_stats_f16x3(
    int nprob, const float* x0, const void* wfrag0, const float* w_scale0, const float* bias0, const float* x_amax0, float* sum1_0,
    float* sum2_0, float* y_scale0, void* pos_hi0, void* pos_lo0, void* chan_hi0, void* chan_lo0, const float* x1, const void* wfrag1,
    const float* w_scale1, const float* bias1, const float* x_amax1, float* sum1_1, float* sum2_1, float* y_scale1, void* pos_hi1,
    void* pos_lo1, void* chan_hi1, void* chan_lo1, int B, int K, int N, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(nprob == 1 || nprob == 2, COCOS_ERR_INVALID, "proj_raw_planes_stats: nprob = %d", nprob);
    COCOS_REQUIRE(x0 && wfrag0 && w_scale0 && x_amax0 && sum1_0 && sum2_0 && y_scale0 && pos_hi0 && pos_lo0, COCOS_ERR_INVALID,
                  "proj_raw_planes_stats: null pointer");
    COCOS_REQUIRE(nprob == 1 || (x1 && wfrag1 && w_scale1 && x_amax1 && sum1_1 && sum2_1 && y_scale1 && pos_hi1 && pos_lo1), COCOS_ERR_INVALID,
                  "proj_raw_planes_stats: null pointer (second projection)");
    COCOS_REQUIRE((chan_hi0 == nullptr) == (chan_lo0 == nullptr) && (nprob == 1 || ((chan_hi1 == nullptr) == (chan_hi0 == nullptr) &&
                  (chan_lo1 == nullptr) == (chan_lo0 == nullptr))), COCOS_ERR_INVALID,
                  "proj_raw_planes_stats: channel-major planes come as hi/lo pairs, for both projections or neither");
    COCOS_REQUIRE(B >= 1 && K >= 1 && K <= 4096 && N >= 128 && N % 128 == 0, COCOS_ERR_UNSUPPORTED,
                  "proj_raw_planes_stats: needs N %% 128 == 0, K <= 4096 (B=%d K=%d N=%d)", B, K, N);
    COCOS_REQUIRE((size_t)K * N * 4 < 0x7fffffffull && (long long)nprob * B * (N / 128) < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED,
                  "proj_raw_planes_stats: one sample exceeds 2 GiB");
    for (const void* p : {(const void*)x0, wfrag0, (const void*)pos_hi0, (const void*)pos_lo0, (const void*)x1, wfrag1,
                          (const void*)pos_hi1, (const void*)pos_lo1, (const void*)bias0, (const void*)bias1})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "proj_raw_planes_stats: pointers must be 16-byte aligned");
    PnArgs a;
    a.p[0] = PnProb{x0, wfrag0, w_scale0, bias0, x_amax0, sum1_0, static_cast<_Float16*>(pos_hi0), static_cast<_Float16*>(pos_lo0),
                    static_cast<_Float16*>(chan_hi0), static_cast<_Float16*>(chan_lo0), sum2_0, y_scale0};
    a.p[1] = nprob == 2 ? PnProb{x1, wfrag1, w_scale1, bias1, x_amax1, sum1_1, static_cast<_Float16*>(pos_hi1),
                                 static_cast<_Float16*>(pos_lo1), static_cast<_Float16*>(chan_hi1), static_cast<_Float16*>(chan_lo1),
                                 sum2_1, y_scale1}
                        : a.p[0];
    a.nprob = nprob; a.B = B; a.K = K; a.N = N; a.nst = (K + 15) / 16;
    a.center = 0;
    a.eps = 0.f; a.plane_scale = 1.0f;
    const dim3 grid((unsigned)(nprob * B * (N / 128)));
    auto launch = [&](auto kern) -> int {
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PN_SMEM));
        hipLaunchKernelGGL(kern, grid, dim3(256), PN_SMEM, as_stream(stream), a);
        return COCOS_OK;
    };
    const int rc = chan_hi0 ? launch(proj_norm_fwd_kernel<true, true>) : launch(proj_norm_fwd_kernel<false, true>);
    if (rc != COCOS_OK) return rc;
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
