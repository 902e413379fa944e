// K24: the INPUT GRADIENT of the theta / phi 1x1 projection fused with what sits between it and the correlation's gradient
// (gfx950, round 6) — autograd of correspondence.py:272-289 (match_kernel 1) resp. :272 + :276-280 (match_kernel 3):
//
//     d[ch, p]  = alpha_p * in1[ch, p] + beta_p * in2[ch, p] + gamma_p          (the gradient w.r.t. the fp32 projection)
//     dx[ci, p] = sum_ch W[ch, ci] * d[ch, p]                                    (dx = W^T d)
//
//   mode A (match_kernel 1, K1's backward fused): in1 = d qn from the K2 backward, in2 = y = the normalised projection, read back
//          from the position-major operand planes; the per-position coefficients come from a first sweep over the channels
//          (a = sum_ch in1 y, s1 = sum in1, s2 = sum y):  alpha = u = 1 / (nrm + eps),  beta = -a / nrm,  gamma = -(u s1 - (a / nrm) s2) / 256
//          — center_l2norm.hip's backward.  Round 5 ran cocos_center_l2norm_bwd_planes (100 MB) -> max|.| -> cocos_proj1x1_stream
//          (86 MB) per tensor; here d never reaches HBM.
//   mode B (match_kernel 3): in1 = the correlation GEMMs' gradient, in2 = the fp32 projection itself, alpha = 1, beta = 2 g2_p,
//          gamma = g1_p — K12's backward (unfold3_stats.hip: dx = g1 + 2 x g2) and autograd's addition of the two gradients, which
//          round 5 ran as three launches (apply 15 us + add 16 us + max|.| 11 us per tensor) in front of the projection's.
//
// The kernel also writes the coefficients [B][3][N] and max|d| (an upper bound in mode A) for the WEIGHT gradient, whose kernel
// (proj_dw_f16x3.hip) rebuilds d from in1 / in2 while it stages them: the fp32 d is neither written nor read.
//
// Structure = proj_norm_f16x3.hip's (K23): a wave owns 32 positions, a workgroup 128; operands reach LDS by LDS-DMA with counted
// waits, two workgroups per CU.  dx has up to 416 rows = 13 accumulator tiles: a workgroup takes HALF of them (7 tiles = 112
// registers; the two halves of a position tile run side by side on one XCD and share its L2).  Two sweeps over the 256
// channels: the first for the per-position coefficients and the per-position power-of-two scale of the f16 split (d has no
// a-priori magnitude: s_p from max_ch |d|, undone per lane in the epilogue — no max|.| pass over a tensor, no global scale), the
// second recomputes d chunk by chunk (L2 / Infinity-Cache hits) and multiplies.  Arithmetic as everywhere on the split path:
// a.b ~= ah.bh + ah.bl + al.bh on v_mfma_f32_32x32x16_f16, fp32 accumulate.
#include <algorithm>

#include "proj_frag.h"

// Debug builds (results of the ablations are WRONG; tools/proj_pair_bench.py, tools/k24_timeline.py):
//   -DCOCOS_K24_ABLATE=<bits>  1: no first sweep, 2: the weight stages are fetched once, 4: no dx stores, 8: no MFMAs
//   -DCOCOS_K24_TIMING         wall-clock stamps (100 MHz) of every workgroup at its start, after each sweep and at its end,
//                              read back with cocos_debug_k24_timing()
// What they showed at the bench shape (B = 8, Cin = 407, 64 x 64, both projections; 112 us on that box): sweep 1 ~21 us and sweep 2
// ~20 us per workgroup (1.3 us per 16-channel stage: the latency of a DMA two stages ahead), the store tail 4-9 us, two rounds of
// 512 resident workgroups; without the first sweep 89 us, without the weight stages 111 (they are L2 hits), without MFMAs 104.
// An eight-wave form (both row halves in one workgroup, operand slots shared and fetched once, three stages ahead, weight and
// operand DMAs issued by different waves so that each has its own vmcnt queue) was built and measured: the SAME 115 us — the
// duplicate fetches of the two halves were L2 hits all along; what the kernel waits for is the memory side: the operands of a
// round (67 MB) do not fit the 32 MB of L2, so BOTH sweeps stream them from HBM / Infinity Cache (268 MB read + 107 MB written in
// ~110 us).  Removing the first sweep needs its per-position sums from the kernels that produce d qn / d kn — not built.
#ifndef COCOS_K24_ABLATE
#define COCOS_K24_ABLATE 0
#endif

namespace cocos {

#ifdef COCOS_K24_TIMING
__device__ long long g_k24_t[2048][4];
#define K24_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_k24_t[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define K24_STAMP(i) do {} while (0)
#endif

typedef _Float16 pb_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* pb_lds_ptr;

constexpr int PB_WSTAGE = PB_HB * 2 * 1024;     // ... of which a workgroup stages its half: 14 KB
constexpr int PB_SLOT = 2048;                   // one operand slot of a wave: in1 [16 ch][32 pos] fp32, or in2 (fp32 the same; planes: hi | lo, [32 pos][16 ch] f16 each)
constexpr int PB_SLOTS = 3;
constexpr int PB_SMEM = 2 * PB_WSTAGE + 4 * 2 * PB_SLOTS * PB_SLOT;      // 28 + 48 KB

__device__ __forceinline__ float pb_pow2_scale(float amax) {      // max -> [2^9, 2^10)
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
    int e;
    frexpf(amax, &e);
    return ldexpf(1.0f, 10 - e);
}

struct PbProb {
    const float* in1;       // [B][256][N] fp32
    const void* in2a;       // mode A: position-major hi plane [B][N][256] f16; mode B: the projection, fp32 [B][256][N]
    const void* in2b;       // mode A: position-major lo plane
    const float* c1;        // mode A: nrm [B][N];  mode B: g1 [B][N]
    const float* c2;        // mode B: g2 [B][N]
    const void* wtfrag;     // PB_NST stages of PB_WSTAGE_ALL bytes (cocos_proj_weight_tfrag_planes)
    const float* w_scale;   // device cell
    float* dx;              // [B][Cin][N]
    float* coef;            // [B][3][N]: alpha, beta, gamma
    unsigned* amax;         // device cell: atomicMax of max|d| (as bits of a non-negative float)
    const float* in2_scale; // mode C: device cell with the power of two the in2 planes were multiplied with (null: a.inv_plane_scale)
};
struct PbArgs {
    PbProb p[2];
    int nprob, B, Cin, N;
    int center;
    float eps, inv_plane_scale;
};

// w [256][Cin] fp32 -> fragment-ordered planes of W^T for the kernel below (proj_frag.h).  grid (PB_NST, 2 * PB_HB), 64 threads.
__global__ __launch_bounds__(64) void proj_weight_tfrag_kernel(const float* __restrict__ w, const float* __restrict__ w_amax,
                                                               unsigned char* __restrict__ out, float* __restrict__ w_scale, int Cin) {
    const int s = blockIdx.x, blk = blockIdx.y, l = threadIdx.x;
    const float sc = pb_pow2_scale(*w_amax);
    if (s == 0 && blk == 0 && l == 0 && w_scale) *w_scale = sc;
    pf_weight_tfrag_item(w, sc, out, Cin, s, blk, l);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void proj_bwd_kernel(const PbArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char pb_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int N = a.N, Cin = a.Cin;
    K24_STAMP(0);
    const int tiles = N / 128, per_prob = a.B * tiles, total = a.nprob * per_prob;
    // the two row halves of a position tile: 8 workgroups apart in dispatch order = the same XCD, resident together
    const int vb = blockIdx.x;
    int half, tl;
    if ((total & 7) == 0) { const int j = vb >> 3; half = j & 1; tl = (j >> 1) * 8 + (vb & 7); }
    else { half = vb & 1; tl = vb >> 1; }
    const int pi = tl >= per_prob ? 1 : 0;
    const PbProb P = pi ? a.p[1] : a.p[0];
    const int rem = tl - pi * per_prob;
    const int b = rem / tiles, n0 = (rem - b * tiles) * 128 + wave * 32;

    unsigned char* const wbuf = pb_smem;                                                   // [2][PB_WSTAGE]
    unsigned char* const s1w = pb_smem + 2 * PB_WSTAGE + wave * (2 * PB_SLOTS * PB_SLOT);  // this wave's in1 slots
    unsigned char* const s2w = s1w + PB_SLOTS * PB_SLOT;                                   // ... and in2 slots

    // MODE 0 (A): planes + K1's coefficients; 1 (B): fp32 in2 + K12's coefficients; 2 (C, round 6 / K25): planes + K12's coefficients
    constexpr bool PLANES = MODE != 1;
    const float inv_plane_scale = (MODE == 2 && P.in2_scale) ? 1.0f / *P.in2_scale : a.inv_plane_scale;
    const size_t chan_bytes = (size_t)PB_K * N * 4;
    const __amdgpu_buffer_rsrc_t i1_rs = make_rsrc(P.in1 + (size_t)b * PB_K * N, chan_bytes);
    const __amdgpu_buffer_rsrc_t i2a_rs = PLANES ? make_rsrc(static_cast<const _Float16*>(P.in2a) + (size_t)b * N * PB_K, (size_t)N * PB_K * 2)
                                                    : make_rsrc(static_cast<const float*>(P.in2a) + (size_t)b * PB_K * N, chan_bytes);
    const __amdgpu_buffer_rsrc_t i2b_rs = PLANES ? make_rsrc(static_cast<const _Float16*>(P.in2b) + (size_t)b * N * PB_K, (size_t)N * PB_K * 2)
                                                    : i2a_rs;
    const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(P.wtfrag, (size_t)PB_NST * PB_WSTAGE_ALL);
    const __amdgpu_buffer_rsrc_t none_rs = make_rsrc(P.wtfrag, 0);
    // fp32 channel-major pieces: instruction i of a stage = channels 8i .. 8i+7 (lane >> 3) x positions 4 (lane & 7) .. + 3
    const unsigned f_voff = (unsigned)((lane >> 3) * N + n0 + 4 * (lane & 7)) * 4u;
    // position-major plane pieces (mode A): lane = (position lane >> 1, 8 channels lane & 1)
    const unsigned p_voff = (unsigned)((n0 + (lane >> 1)) * PB_K + 8 * (lane & 1)) * 2u;
    // this wave's share of a weight stage: KB pieces 4 wave .. 4 wave + 3 of the workgroup's 14 (the last wave repeats piece 13)
    unsigned w_soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w_soff[i] = (unsigned)(min(wave * 4 + i, 2 * PB_HB - 1) * 1024 + half * PB_WSTAGE);

    auto issue_w = [&](int s) {
        const bool ok = s < PB_NST && !((COCOS_K24_ABLATE & 2) && s > 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? w_rs : none_rs, (pb_lds_ptr)(wbuf + (s & 1) * PB_WSTAGE + (w_soff[i] - half * PB_WSTAGE)), 16,
                                                     (int)(lane * 16), (int)((unsigned)s * PB_WSTAGE_ALL + w_soff[i]), 0, 0);
    };
    auto issue_in = [&](int s, int slot) {       // 16 channels of in1 and in2 for this wave's positions: 4 instructions
        const bool ok = s < PB_NST;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? i1_rs : none_rs, (pb_lds_ptr)(s1w + slot * PB_SLOT + i * 1024), 16, (int)f_voff,
                                                     (int)((unsigned)(16 * s + 8 * i) * N * 4u), 0, 0);
        if (PLANES) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? i2a_rs : none_rs, (pb_lds_ptr)(s2w + slot * PB_SLOT), 16, (int)p_voff, (int)(32 * s), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? i2b_rs : none_rs, (pb_lds_ptr)(s2w + slot * PB_SLOT + 1024), 16, (int)p_voff, (int)(32 * s), 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? i2a_rs : none_rs, (pb_lds_ptr)(s2w + slot * PB_SLOT + i * 1024), 16, (int)f_voff,
                                                         (int)((unsigned)(16 * s + 8 * i) * N * 4u), 0, 0);
        }
    };
    // this lane's 8 values of a stage: channels 16 s + 8 h .. + 7 of position n0 + c
    auto read_in = [&](int slot, float (&v1)[8], float (&v2)[8]) {
        const float* q1 = reinterpret_cast<const float*>(s1w + slot * PB_SLOT) + (8 * h) * 32 + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) v1[j] = q1[j * 32];
        if (PLANES) {
            const pb_f16x8 yh = *reinterpret_cast<const pb_f16x8*>(s2w + slot * PB_SLOT + c * 32 + h * 16);
            const pb_f16x8 yl = *reinterpret_cast<const pb_f16x8*>(s2w + slot * PB_SLOT + 1024 + c * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) v2[j] = ((float)yh[j] + (float)yl[j]) * inv_plane_scale;
        } else {
            const float* q2 = reinterpret_cast<const float*>(s2w + slot * PB_SLOT) + (8 * h) * 32 + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) v2[j] = q2[j * 32];
        }
    };

    const size_t pos = (size_t)b * N + n0 + c;
    float alpha = 1.0f, beta = 0.f, gamma = 0.f;
    if (MODE != 0) { beta = 2.0f * P.c2[pos]; gamma = P.c1[pos]; }

    // ---------------- sweep 1: per-position sums / maximum over the 256 channels ----------------
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, mx = 0.f;
    issue_in(0, 0);
    issue_in(1, 1);
    int slot = 0;
#pragma unroll 1
    for (int s = 0; s < ((COCOS_K24_ABLATE & 1) ? 0 : PB_NST); ++s) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // in(s) landed; in(s + 1) may be in flight
        int slot2 = slot + 2;
        if (slot2 >= PB_SLOTS) slot2 -= PB_SLOTS;
        issue_in(s + 2, slot2);
        float v1[8], v2[8];
        read_in(slot, v1, v2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
                s0 = __builtin_fmaf(v1[j], v2[j], s0);
                s1 += v1[j];
                s2 += v2[j];
                mx = fmaxf(mx, fabsf(v1[j]));
            } else {
                mx = fmaxf(mx, fabsf(__builtin_fmaf(beta, v2[j], v1[j]) + gamma));
            }
        }
        slot = slot + 1 == PB_SLOTS ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    mx = fmaxf(mx, swap_half(mx));
    float dmax = mx;                       // max_ch |d| of this position (mode A: an upper bound)
    if (MODE == 0) {
        s0 += swap_half(s0); s1 += swap_half(s1); s2 += swap_half(s2);
        const float nrm = P.c1[pos];
        const float u = 1.0f / (nrm + a.eps);
        const float g = nrm > 0.f ? s0 / nrm : 0.f;
        const float m = a.center ? (u * s1 - g * s2) * (1.0f / (float)PB_K) : 0.f;
        alpha = u; beta = -g; gamma = -m;
        dmax = u * mx + fabsf(g) + fabsf(m);      // |y| <= 1: unit columns
    }
    const float sp = pb_pow2_scale(dmax);
    K24_STAMP(1);
    if (half == 0 && h == 0) {
        float* cf = P.coef + (size_t)b * 3 * N + n0 + c;
        cf[0] = alpha; cf[N] = beta; cf[2 * (size_t)N] = gamma;
    }
    if (half == 0) {
        const float wm = wave_max_dpp(dmax);
        if (lane == 0 && wm > 0.f && wm < INFINITY) atomicMax(P.amax, __float_as_uint(wm));
    }

    // ---------------- sweep 2: d chunk by chunk -> dx ----------------
    f32x16 acc[PB_HB];
#pragma unroll
    for (int i = 0; i < PB_HB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a_s = alpha * sp, b_s = beta * sp, g_s = gamma * sp;
    issue_in(0, 0);
    issue_w(0);
    issue_in(1, 1);
    slot = 0;
#pragma unroll 1
    for (int s = 0; s < PB_NST; ++s) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // W(s), in(s) landed; in(s + 1) in flight
        issue_w(s + 1);
        int slot2 = slot + 2;
        if (slot2 >= PB_SLOTS) slot2 -= PB_SLOTS;
        issue_in(s + 2, slot2);
        float v1[8], v2[8];
        read_in(slot, v1, v2);
        unsigned bhw[4], blw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d0 = __builtin_fmaf(a_s, v1[2 * q], __builtin_fmaf(b_s, v2[2 * q], g_s));
            const float d1 = __builtin_fmaf(a_s, v1[2 * q + 1], __builtin_fmaf(b_s, v2[2 * q + 1], g_s));
            split_pair_rn(d0, d1, bhw[q], blw[q]);
        }
        const pb_f16x8 bh = __builtin_bit_cast(pb_f16x8, u32x4{bhw[0], bhw[1], bhw[2], bhw[3]});
        const pb_f16x8 bl = __builtin_bit_cast(pb_f16x8, u32x4{blw[0], blw[1], blw[2], blw[3]});
        const unsigned char* wb = wbuf + (s & 1) * PB_WSTAGE + lane * 16;
#pragma unroll
        for (int i = 0; i < ((COCOS_K24_ABLATE & 8) ? 0 : PB_HB); i += 2) {
            const pb_f16x8 ah0 = *reinterpret_cast<const pb_f16x8*>(wb + (i * 2 + 0) * 1024);
            const pb_f16x8 al0 = *reinterpret_cast<const pb_f16x8*>(wb + (i * 2 + 1) * 1024);
            if (i + 1 < PB_HB) {
                const pb_f16x8 ah1 = *reinterpret_cast<const pb_f16x8*>(wb + (i * 2 + 2) * 1024);
                const pb_f16x8 al1 = *reinterpret_cast<const pb_f16x8*>(wb + (i * 2 + 3) * 1024);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh, acc[i], 0, 0, 0);
                acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh, acc[i + 1], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl, acc[i], 0, 0, 0);
                acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl, acc[i + 1], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh, acc[i], 0, 0, 0);
                acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh, acc[i + 1], 0, 0, 0);
            } else {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh, acc[i], 0, 0, 0);
            }
        }
        slot = slot + 1 == PB_SLOTS ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K24_STAMP(2);

    // ---------------- dx rows of this half: lane = position, register = row; 128-byte row segments ----------------
    const float oscale = 1.0f / (*P.w_scale * sp);
    const __amdgpu_buffer_rsrc_t dx_rs = make_rsrc(P.dx + (size_t)b * Cin * N, (size_t)Cin * N * 4);
    const int row0 = half * (PB_HB * 32) + 4 * h;
    const unsigned dx_voff = (unsigned)(row0 * N + n0 + c) * 4u;
#pragma unroll
    for (int i = 0; i < PB_HB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + i * 32 + acc_row_base(r);
            buf_store1s(dx_rs, acc[i][r] * oscale, (row < Cin && !(COCOS_K24_ABLATE & 4)) ? dx_voff : kBufOob, (unsigned)((i * 32 + acc_row_base(r)) * N) * 4u);
        }
#ifdef COCOS_K24_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    K24_STAMP(3);
}

}  // namespace cocos

#ifdef COCOS_K24_TIMING
extern "C" int cocos_debug_k24_timing(long long* host, int nblocks) {
    using namespace cocos;
    COCOS_HIP_CHECK(hipDeviceSynchronize());
    COCOS_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k24_t), (size_t)std::min(nblocks, 2048) * 4 * sizeof(long long)));
    return COCOS_OK;
}
#endif

extern "C" size_t cocos_proj_weight_tfrag_bytes(void) { return (size_t)cocos::PB_NST * cocos::PB_WSTAGE_ALL; }

// w [256][Cin] fp32 (Cin <= 448) + device cell max|w| -> the fragment-ordered planes of W^T for cocos_proj_bwd_input_f16x3
// (cocos_proj_weight_tfrag_bytes() bytes) and, when w_scale_dev is given, the power of two they were multiplied with.
extern "C" int cocos_proj_weight_tfrag_planes(const float* w, const float* w_amax_dev, void* wtfrag, float* w_scale_dev, int M, int Cin,
                                              cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(w && w_amax_dev && wtfrag, COCOS_ERR_INVALID, "proj_weight_tfrag_planes: null pointer");
    COCOS_REQUIRE(M == PB_K && Cin >= 1 && Cin <= 2 * PB_HB * 32, COCOS_ERR_UNSUPPORTED,
                  "proj_weight_tfrag_planes: needs M == 256 and Cin <= 448 (M=%d Cin=%d)", M, Cin);
    COCOS_REQUIRE(aligned16(wtfrag), COCOS_ERR_INVALID, "proj_weight_tfrag_planes: planes must be 16-byte aligned");
    hipLaunchKernelGGL(proj_weight_tfrag_kernel, dim3(PB_NST, 2 * PB_HB), dim3(64), 0, as_stream(stream), w, w_amax_dev,
                       static_cast<unsigned char*>(wtfrag), w_scale_dev, Cin);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_proj_bwd_input_supported(int Cin, int M, int N) {
    return M == cocos::PB_K && Cin >= 1 && Cin <= 2 * cocos::PB_HB * 32 && N >= 128 && N % 128 == 0 &&
           (size_t)std::max(Cin, M) * N * 4 < 0x7fffffffull;
}

// Up to two projections of the same shape in one launch.  mode 0 (A): in1 = d qn [B,256,N], in2a / in2b = the position-major
// hi / lo planes [B,N,256] of plane_scale * y, c1 = nrm [B,N] (c2 unused); mode 1 (B): in1 = the GEMMs' gradient, in2a = the fp32
// projection [B,256,N] (in2b unused), c1 = g1, c2 = g2 [B,N] (d = in1 + 2 g2 in2 + g1).  Outputs per projection: dx [B,Cin,N],
// coef [B,3,N] (alpha, beta, gamma), *amax = max(*amax, max|d|) (the cell must hold a finite value >= 0).
extern "C" int cocos_proj_bwd_input_f16x3(
    int mode, int nprob, const float* in1_0, const void* in2a_0, const void* in2b_0, const float* c1_0, const float* c2_0,
    const void* wtfrag0, const float* w_scale0, float* dx0, float* coef0, float* amax0, const float* in1_1, const void* in2a_1,
    const void* in2b_1, const float* c1_1, const float* c2_1, const void* wtfrag1, const float* w_scale1, float* dx1, float* coef1,
    float* amax1, int B, int Cin, int N, int center_over_channels, float eps, float plane_scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(mode == 0 || mode == 1, COCOS_ERR_INVALID, "proj_bwd_input: mode %d", mode);
    COCOS_REQUIRE(nprob == 1 || nprob == 2, COCOS_ERR_INVALID, "proj_bwd_input: nprob = %d", nprob);
    COCOS_REQUIRE(in1_0 && in2a_0 && c1_0 && wtfrag0 && w_scale0 && dx0 && coef0 && amax0 && (mode == 0 ? in2b_0 != nullptr : c2_0 != nullptr),
                  COCOS_ERR_INVALID, "proj_bwd_input: null pointer");
    COCOS_REQUIRE(nprob == 1 || (in1_1 && in2a_1 && c1_1 && wtfrag1 && w_scale1 && dx1 && coef1 && amax1 &&
                                 (mode == 0 ? in2b_1 != nullptr : c2_1 != nullptr)),
                  COCOS_ERR_INVALID, "proj_bwd_input: null pointer (second projection)");
    COCOS_REQUIRE(B >= 1 && cocos_proj_bwd_input_supported(Cin, PB_K, N) && plane_scale > 0.f, COCOS_ERR_UNSUPPORTED,
                  "proj_bwd_input: needs Cin <= 448, N %% 128 == 0 (B=%d Cin=%d N=%d)", B, Cin, N);
    COCOS_REQUIRE(mode == 1 || center_over_channels == 1 || center_over_channels == 2, COCOS_ERR_UNSUPPORTED,
                  "proj_bwd_input: centring over channels (1) or none (2), got %d", center_over_channels);
    COCOS_REQUIRE((long long)nprob * B * (N / 128) * 2 < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "proj_bwd_input: grid too large");
    for (const void* p : {(const void*)in1_0, in2a_0, in2b_0, wtfrag0, (const void*)in1_1, in2a_1, in2b_1, wtfrag1})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "proj_bwd_input: pointers must be 16-byte aligned");
    PbArgs a;
    a.p[0] = PbProb{in1_0, in2a_0, in2b_0, c1_0, c2_0, wtfrag0, w_scale0, dx0, coef0, reinterpret_cast<unsigned*>(amax0), nullptr};
    a.p[1] = nprob == 2 ? PbProb{in1_1, in2a_1, in2b_1, c1_1, c2_1, wtfrag1, w_scale1, dx1, coef1, reinterpret_cast<unsigned*>(amax1), nullptr}
                        : a.p[0];
    a.nprob = nprob; a.B = B; a.Cin = Cin; a.N = N;
    a.center = center_over_channels == 1;
    a.eps = eps; a.inv_plane_scale = 1.0f / plane_scale;
    const dim3 grid((unsigned)(nprob * B * (N / 128) * 2));
    auto launch = [&](auto kern) -> int {
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PB_SMEM));
        hipLaunchKernelGGL(kern, grid, dim3(256), PB_SMEM, as_stream(stream), a);
        return COCOS_OK;
    };
    const int rc = mode == 0 ? launch(proj_bwd_kernel<0>) : launch(proj_bwd_kernel<1>);
    if (rc != COCOS_OK) return rc;
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// Mode C (round 6, K25's backward): as mode 1 — d = in1 + 2 g2 y + g1 — with y read back from the position-major f16 hi / lo
// planes [B,N,256] of s * y that cocos_proj_raw_planes_stats_f16x3 wrote (in2_scale = its *y_scale cell): the fp32 projection
// does not exist on this path.  Up to two projections of one shape per launch; outputs as for cocos_proj_bwd_input_f16x3.
extern "C" int cocos_proj_bwd_input_planes_f16x3(
    int nprob, const float* in1_0, const void* y_hi0, const void* y_lo0, const float* y_scale0, const float* g1_0, const float* g2_0,
    const void* wtfrag0, const float* w_scale0, float* dx0, float* coef0, float* amax0, const float* in1_1, const void* y_hi1,
    const void* y_lo1, const float* y_scale1, const float* g1_1, const float* g2_1, const void* wtfrag1, const float* w_scale1, float* dx1,
    float* coef1, float* amax1, int B, int Cin, int N, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(nprob == 1 || nprob == 2, COCOS_ERR_INVALID, "proj_bwd_input_planes: nprob = %d", nprob);
    COCOS_REQUIRE(in1_0 && y_hi0 && y_lo0 && y_scale0 && g1_0 && g2_0 && wtfrag0 && w_scale0 && dx0 && coef0 && amax0, COCOS_ERR_INVALID,
                  "proj_bwd_input_planes: null pointer");
    COCOS_REQUIRE(nprob == 1 || (in1_1 && y_hi1 && y_lo1 && y_scale1 && g1_1 && g2_1 && wtfrag1 && w_scale1 && dx1 && coef1 && amax1),
                  COCOS_ERR_INVALID, "proj_bwd_input_planes: null pointer (second projection)");
    COCOS_REQUIRE(B >= 1 && cocos_proj_bwd_input_supported(Cin, PB_K, N), COCOS_ERR_UNSUPPORTED,
                  "proj_bwd_input_planes: needs Cin <= 448, N %% 128 == 0 (B=%d Cin=%d N=%d)", B, Cin, N);
    COCOS_REQUIRE((long long)nprob * B * (N / 128) * 2 < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "proj_bwd_input_planes: grid too large");
    for (const void* p : {(const void*)in1_0, y_hi0, y_lo0, wtfrag0, (const void*)in1_1, y_hi1, y_lo1, wtfrag1})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "proj_bwd_input_planes: pointers must be 16-byte aligned");
    PbArgs a;
    a.p[0] = PbProb{in1_0, y_hi0, y_lo0, g1_0, g2_0, wtfrag0, w_scale0, dx0, coef0, reinterpret_cast<unsigned*>(amax0), y_scale0};
    a.p[1] = nprob == 2 ? PbProb{in1_1, y_hi1, y_lo1, g1_1, g2_1, wtfrag1, w_scale1, dx1, coef1, reinterpret_cast<unsigned*>(amax1), y_scale1}
                        : a.p[0];
    a.nprob = nprob; a.B = B; a.Cin = Cin; a.N = N;
    a.center = 0;
    a.eps = 0.f; a.inv_plane_scale = 1.0f;
    const dim3 grid((unsigned)(nprob * B * (N / 128) * 2));
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PB_SMEM));
    hipLaunchKernelGGL(proj_bwd_kernel<2>, grid, dim3(256), PB_SMEM, as_stream(stream), a);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
