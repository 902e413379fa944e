// Fragment-ordered weight planes of the theta / phi 1x1 projections (correspondence.py:181-182 at :272, :282): the layouts that
// K23 (proj_norm_f16x3.hip: W, A operand of y = W x) and K24 (proj_bwd_f16x3.hip: W^T, A operand of dx = W^T d) stage by linear
// LDS-DMA copies.  The per-item bodies live here so that the per-layout entry points and the ONE-launch preparation of both
// layouts of both projections (cocos_proj_weight_prep_pair, proj_norm_f16x3.hip) run the same code.  gfx950.
#pragma once
#include "common.h"

namespace cocos {

constexpr int PN_M = 256;                      // output channels (= the correlation kernels' K)
constexpr int PN_WSTAGE = 8 * 2 * 1024;        // bytes of one stage of W: [row block 8][plane 2][lane 64][16 B], k-step of 16
constexpr int PB_K = 256;                       // channels of the projection's output = the contraction of dx
constexpr int PB_NST = PB_K / 16;               // 16-channel stages
constexpr int PB_HB = 7;                        // accumulator tiles (32 rows of dx) per workgroup: two halves cover 14 x 32 = 448 rows
constexpr int PB_WSTAGE_ALL = 2 * PB_HB * 2 * 1024;      // bytes of one stage of the transposed weight planes: [row block 14][plane 2][1 KB]

__device__ __forceinline__ float pf_pow2_scale(float amax) {      // max -> [2^9, 2^10)
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
    int e;
    frexpf(amax, &e);
    return ldexpf(1.0f, 10 - e);
}

// weight [256][K] fp32 -> fragment-ordered f16 hi / lo planes of sc * W.  Item (stage s, row block blk), 64 lanes: lane l owns
// W[blk*32 + (l & 31)][16 s + 8 (l >> 5) .. + 7].  t_hi / t_lo (nullable): the same numbers as TRANSPOSED row-major planes
// [K][256] — the A operand of dx = W^T dy in cocos_proj1x1_stream_f16x3.
__device__ __forceinline__ void pf_weight_frag_item(const float* __restrict__ w, float sc, unsigned char* __restrict__ out, int K,
                                                    _Float16* __restrict__ t_hi, _Float16* __restrict__ t_lo, int s, int blk, int l) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int row = blk * 32 + (l & 31), k0 = 16 * s + 8 * (l >> 5);
    unsigned hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = k0 + 2 * q;
        const float a = k < K ? w[(size_t)row * K + k] * sc : 0.f;
        const float b = k + 1 < K ? w[(size_t)row * K + k + 1] * sc : 0.f;
        split_pair_rn(a, b, hw[q], lw[q]);
        if (t_hi) {
            const h2 hh = __builtin_bit_cast(h2, hw[q]), ll = __builtin_bit_cast(h2, lw[q]);
            if (k < K) { t_hi[(size_t)k * PN_M + row] = hh[0]; t_lo[(size_t)k * PN_M + row] = ll[0]; }
            if (k + 1 < K) { t_hi[(size_t)(k + 1) * PN_M + row] = hh[1]; t_lo[(size_t)(k + 1) * PN_M + row] = ll[1]; }
        }
    }
    unsigned char* d = out + (size_t)s * PN_WSTAGE + (size_t)(blk * 2) * 1024 + l * 16;
    *reinterpret_cast<u32x4*>(d) = u32x4{hw[0], hw[1], hw[2], hw[3]};
    *reinterpret_cast<u32x4*>(d + 1024) = u32x4{lw[0], lw[1], lw[2], lw[3]};
}

// w [256][Cin] fp32 -> fragment-ordered planes of sc * W^T: lane l of item (stage s, row block blk) owns
// W[16 s + 8 (l >> 5) .. + 7][blk * 32 + (l & 31)] (A[i = input channel][k = output channel]), zero beyond Cin.
__device__ __forceinline__ void pf_weight_tfrag_item(const float* __restrict__ w, float sc, unsigned char* __restrict__ out, int Cin,
                                                     int s, int blk, int l) {
    const int ci = blk * 32 + (l & 31), k0 = 16 * s + 8 * (l >> 5);
    unsigned hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v0 = ci < Cin ? w[(size_t)(k0 + 2 * q) * Cin + ci] * sc : 0.f;
        const float v1 = ci < Cin ? w[(size_t)(k0 + 2 * q + 1) * Cin + ci] * sc : 0.f;
        split_pair_rn(v0, v1, hw[q], lw[q]);
    }
    unsigned char* d = out + (size_t)s * PB_WSTAGE_ALL + (size_t)(blk * 2) * 1024 + l * 16;
    *reinterpret_cast<u32x4*>(d) = u32x4{hw[0], hw[1], hw[2], hw[3]};
    *reinterpret_cast<u32x4*>(d + 1024) = u32x4{lw[0], lw[1], lw[2], lw[3]};
}

}  // namespace cocos
