// match_kernel = 3 without unfolding, round 3: pieces shared by the correlation GEMM's x-box epilogue
// (hgemm_f16x3.hip), the fused box -> softmax -> warp kernels and the box-adjoint kernel (box3_fused_f16x3.hip).
//
// The reference unfolds theta / phi to K = 2304 (F.unfold(k=3, padding=1), correspondence.py:276,:286) and multiplies
// the unfolded matrices (:291).  <U_p, V_q> over the 3x3 neighbourhoods is a 9-tap box filter of the K = 256
// correlation C along its DIAGONAL:  S[p,q] = sum_{d in 3x3} C[p+d, q+d]  (zero outside the grid) — K6 of round 1.
// That filter is separable:  S = ybox(xbox(C)),
//     xbox(C)[p,q] = C[p-1,q-1] + C[p,q] + C[p+1,q+1]      (same image row for p and for q)
//     ybox(T)[p,q] = T[p-w,q-w] + T[p,q] + T[p+w,q+w]      (w = grid width)
// and the two directions want different homes on this machine:
//   * x shifts move BOTH indices by one: inside a 32x32 MFMA tile that is one register row down and one lane across.
//     It is done ONCE per element where the tile is produced (the GEMM epilogue), through a per-wave LDS image whose
//     zero border is the grid's left / right edge — a wave's 128 x 64 sub-tile holds whole image rows in both directions
//     when w = 64, so no halo crosses a wave;
//   * y shifts move both indices by a whole row = by whole 32x32 tiles (w / 32 of them): T tiles of three diagonal
//     neighbours add up ELEMENT BY ELEMENT, same lane and register.  The consumer (softmax + warp) therefore never needs
//     the box-filtered matrix in memory: it loads three 4 KB blocks of T per tile (two of them L2 / MALL hits: the
//     neighbouring query rows load them within a few iterations) and adds them.
// Result: ONE HWxHW fp32 matrix (T) in HBM for the forward instead of three (C_raw, logits, and the logits again in the
// softmax pass), none for inference beyond T; the backward recomputes P from T.
//
// Tile-blocked layout of T and of dL/dTsum ("G"), shared with the K2 forward's saved logits (corr_fused_fwd_f16x3.hip):
//   [B][Nk/32 key tiles][Nq/32 query blocks] blocks of 4 KB = [g = 0..3][lane 0..63][4 floats]: the accumulator image
//   of v_mfma_f32_32x32x16_f16 with keys in the rows (registers 4g..4g+3 of a lane = keys 8g + 4*(lane>>5) + 0..3) and
//   queries in the columns (lane & 31).  One 32x32 tile = four contiguous 1 KB wave-stores / loads.
#pragma once
#include "common.h"

namespace cocos {

constexpr int kXbStride = 76;                      // floats per query row of the x-box image: 4 * odd -> b128 conflict-free
constexpr int kXbRows = 66;                        // 64 queries + a zero row on either side
constexpr int kXbFloats = kXbRows * kXbStride;     // per wave (20064 B)

// Zero the border of a wave's x-box image (once per kernel: data never overwrites it).
__device__ __forceinline__ void xbox_zero_border(float* img, int lane) {
    for (int i = lane; i < kXbStride; i += 64) {
        img[i] = 0.f;
        img[(kXbRows - 1) * kXbStride + i] = 0.f;
    }
    // columns 3 (key -1) and 68 (key 64) of rows 1..64
    img[(1 + lane) * kXbStride + 3] = 0.f;
    img[(1 + lane) * kXbStride + 68] = 0.f;
}

// The interior of a wave's image <- its 64 x 64 chunk t[kt][qt] (32 keys x 32 queries accumulator images; kt / qt = left /
// right half of the chunk's 64 keys / queries).
__device__ __forceinline__ void xbox_write_chunk(const f32x16 (&t)[2][2], float* img, int lane) {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(img + (qt * 32 + c + 1) * kXbStride + 4 + kt * 32 + 8 * g + 4 * h) =
                    f32x4{t[kt][qt][4 * g], t[kt][qt][4 * g + 1], t[kt][qt][4 * g + 2], t[kt][qt][4 * g + 3]};
}

// t[kt][qt][key, query] += img[key - 1, query - 1] + img[key + 1, query + 1]  (whatever the border cells hold: zeros at a
// grid edge, a neighbouring chunk's values inside an image row)
__device__ __forceinline__ void xbox_add_diagonals(f32x16 (&t)[2][2], const float* img, int lane) {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* dn = img + (qt * 32 + c) * kXbStride + 4 + kt * 32 + 8 * g + 4 * h;        // query - 1
                const float* up = img + (qt * 32 + c + 2) * kXbStride + 4 + kt * 32 + 8 * g + 4 * h;    // query + 1
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dn);      // keys m0 .. m0+3 of query - 1
                const float d0 = dn[-1];                                   // key m0 - 1
                const f32x4 u4 = *reinterpret_cast<const f32x4*>(up);      // keys m0 .. m0+3 of query + 1
                const float u5 = up[4];                                    // key m0 + 4
                t[kt][qt][4 * g + 0] += d0 + u4[1];
                t[kt][qt][4 * g + 1] += d4[0] + u4[2];
                t[kt][qt][4 * g + 2] += d4[1] + u4[3];
                t[kt][qt][4 * g + 3] += d4[2] + u5;
            }
}

// In place: t[kt][qt] <- xbox, for a chunk that holds WHOLE image rows in both directions (64-wide grid).  `img` = this
// wave's kXbFloats floats of LDS with a zeroed border.  One wave's LDS instructions execute in order, so the writes are
// visible to the reads that follow without a barrier.
__device__ __forceinline__ void xbox_64x64(f32x16 (&t)[2][2], float* img, int lane) {
    xbox_write_chunk(t, img, lane);
    __builtin_amdgcn_wave_barrier();
    xbox_add_diagonals(t, img, lane);
    __builtin_amdgcn_wave_barrier();
}

// ---- 128-wide grids (round 4): an image row is TWO 64-position chunks in either direction, so a chunk's image needs its
// neighbours' edge values in the border cells that face the inside of the row (the outward-facing ones stay zero):
//   column 3  = key - 1  of the chunk  (the last key of the chunk to the left),   column 68 = key 64 (first key of the right one)
//   row 0     = query - 1              (the last query of the chunk before),      row 65    = query 64
// and the four corner cells from the diagonal neighbours.  The helpers below write ONE edge of a chunk t[kt][qt] into a
// neighbour's image; the callers (the correlation GEMM's epilogue, where the key-direction neighbour is the same wave's other
// half, and K20, where all four chunks of a row pair are four waves) put a workgroup barrier between the writes and
// xbox_add_diagonals.
//
// one key of a chunk for every query of the chunk -> border column `col` of `dst` (rows 1..64; `dst` may be the writer's own
// image).  v0 / v1 = the lane's values for the two query tiles, held by the half-wave `hsel` (key 63 of a chunk = register 15 of
// key tile 1 in the UPPER half-wave, key 0 = register 0 of key tile 0 in the LOWER one).
__device__ __forceinline__ void xbox_put_key_column(float* dst, int col, float v0, float v1, int lane, int hsel) {
    const int h = lane >> 5, c = lane & 31;
    if (h == hsel) {
        dst[(c + 1) * kXbStride + col] = v0;
        dst[(32 + c + 1) * kXbStride + col] = v1;
    }
}
// the last key (63) of chunk t -> column 3 of `dst`;  the first key (0) of chunk t -> column 68
__device__ __forceinline__ void xbox_put_last_key(const f32x16 (&t)[2][2], float* dst, int lane) {
    xbox_put_key_column(dst, 3, t[1][0][15], t[1][1][15], lane, 1);
}
__device__ __forceinline__ void xbox_put_first_key(const f32x16 (&t)[2][2], float* dst, int lane) {
    xbox_put_key_column(dst, 68, t[0][0][0], t[0][1][0], lane, 0);
}
// the last query (63) of chunk t, keys 0..63 -> row 0 of `dst`, columns 4..67
__device__ __forceinline__ void xbox_put_last_query(const f32x16 (&t)[2][2], float* dst, int lane) {
    const int h = lane >> 5, c = lane & 31;
    if (c == 31)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(dst + 4 + kt * 32 + 8 * g + 4 * h) =
                    f32x4{t[kt][1][4 * g], t[kt][1][4 * g + 1], t[kt][1][4 * g + 2], t[kt][1][4 * g + 3]};
}
// the first query (0) of chunk t, keys 0..63 -> row 65 of `dst`
__device__ __forceinline__ void xbox_put_first_query(const f32x16 (&t)[2][2], float* dst, int lane) {
    const int h = lane >> 5, c = lane & 31;
    if (c == 0)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(dst + 65 * kXbStride + 4 + kt * 32 + 8 * g + 4 * h) =
                    f32x4{t[kt][0][4 * g], t[kt][0][4 * g + 1], t[kt][0][4 * g + 2], t[kt][0][4 * g + 3]};
}
// corner cells: element (key 63, query 63) of t -> dst[row 0][col 3];  (63, 0) -> [65][3];  (0, 63) -> [0][68];  (0, 0) -> [65][68]
// (v_q0 / v_q63: the lane's value of that key for query tile 0 / 1 — the same registers xbox_put_key_column takes)
__device__ __forceinline__ void xbox_put_corner_value(float* dst, int lane, bool last_key, bool last_query, float v_q0, float v_q63) {
    const int h = lane >> 5, c = lane & 31;
    if (h == (last_key ? 1 : 0) && c == (last_query ? 31 : 0))
        dst[(last_query ? 0 : 65) * kXbStride + (last_key ? 3 : 68)] = last_query ? v_q63 : v_q0;
}
__device__ __forceinline__ void xbox_put_corner(const f32x16 (&t)[2][2], float* dst, int lane, bool last_key, bool last_query) {
    xbox_put_corner_value(dst, lane, last_key, last_query, last_key ? t[1][0][15] : t[0][0][0], last_key ? t[1][1][15] : t[0][1][0]);
}
// one border column (3 or 68) of rows 0..65, or one border row's corner cells, back to zero (a cell that held a neighbour's
// value in the previous pass and faces the grid edge in this one)
__device__ __forceinline__ void xbox_zero_column(float* img, int col, int lane) {
    for (int r = lane; r < kXbRows; r += 64) img[r * kXbStride + col] = 0.f;
}

}  // namespace cocos
