// K0 weight gradient (and bias gradient) as a streaming reduction (gfx950):
//     dw[m,c] = sum_{b,n} dy[b,m,n] x[b,c,n]        db[m] = sum_{b,n} dy[b,m,n]
// for the theta / phi projections of correspondence.py:272,:282 (M = 256 output channels, C = 256 (+151) inputs,
// reduction over B*HW = 32 k positions).  Both operands are big (85 MB together), the result is tiny: the job is
// to read dy and x ONCE, at full memory speed, with all 256 CUs, and to keep the partial results small.
//
// The general split GEMM (sgemm_f16x3.hip, 256 x 128 tiles, split-K 8) reads and re-splits dy once per column
// tile (4x), and its bias gradient is a separate pass over dy.  Here a workgroup owns 128 rows of dy x ALL rows
// of x (accumulators: 64 x 224 per wave = 224 AGPRs) for a chunk of positions, so x is read by the two
// row-halves (the second one from L2: they run on the same XCD) and dy exactly once; db falls out of the dy
// staging (a running sum per staged row).  The 2 x S partial tiles go to a workspace and are summed by a second,
// bandwidth-bound kernel (S = 128 chunks at the reference's size: 53 MB, written and read once, mostly in the
// memory-side cache).
//
// Pipeline per workgroup: a k-step is 16 positions.  LDS holds k-steps t (being multiplied) and t+1 (committed
// meanwhile: fp32 -> f16 hi/lo planes, rows of 16 + 8 pad halfs = conflict-free 16-byte operand reads); two
// register stages hold t+2 / t+3 in flight, so a global load has two k-steps to arrive.  Staging pieces (one
// float4 = 4 positions of one row) are converted, committed and re-loaded one at a time between the MFMA triples.
// Arithmetic as everywhere on the split path: a.b ~= ah.bh + ah.bl + al.bh, v_mfma_f32_32x32x16_f16, fp32 accumulate.
#include <algorithm>

#include "common.h"

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int DW_BK = 16, DW_ROW = DW_BK + 8;      // positions per k-step; halfs per LDS row
constexpr int DW_MROWS = 128;                      // dy rows per workgroup

__device__ __forceinline__ float dw_scale_from_amax(const float* amax) {
    if (!amax) return 1.0f;
    const float a = *amax;
    if (!(a > 0.f) || !(a < INFINITY)) return 1.0f;
    int e;
    frexpf(a, &e);
    return ldexpf(1.0f, 10 - e);
}

__device__ __forceinline__ void dw_split4(const f32x4& x, float s, u32x2& hi, u32x2& lo) {
    const float a = x[0] * s, b = x[1] * s, c = x[2] * s, d = x[3] * s;
    const f16x2 h0 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    const f16x2 h1 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(c, d));
    const f16x2 l0 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a - (float)h0[0], b - (float)h0[1]));
    const f16x2 l1 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(c - (float)h1[0], d - (float)h1[1]));
    hi = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    lo = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}

// CBW = 32-column blocks (rows of x) per wave: the workgroup covers 2 * CBW * 32 rows of x
// DMODE (round 6, K24's companion): 0 — dy is the gradient itself; 1 / 2 — dy is REBUILT while it is staged,
//     dy[m, n] = alpha_n * in1[m, n] + beta_n * in2[m, n] + gamma_n        (coef [B][3][N] from cocos_proj_bwd_input_f16x3)
// with in1 = the `dy` argument and in2 = the fp32 projection [B,M,N] (1: match_kernel 3, K12's backward folded in) or the
// channel-major f16 hi / lo planes [B,M,N] of plane_scale * y (2: match_kernel 1, K1's backward folded in).  The fp32 gradient
// w.r.t. the projection is then neither written nor read anywhere.
struct DwAffine {
    const float* coef;        // [B][3][N]
    const void* in2a;         // DMODE 1: fp32 [B][M][N];  DMODE 2: hi plane [B][M][N] f16
    const void* in2b;         // DMODE 2: lo plane
    float inv_plane_scale;
    const float* plane_scale_dev;     // DMODE 2 (nullable): device cell with the planes' scale; overrides inv_plane_scale
};

// One problem of a launch (blockIdx.z picks it: theta's and phi's weight gradients share a launch — 2 x 128 workgroups with
// chunks twice as long instead of two launches of 256, i.e. half the partial tiles to write and to sum).
struct DwProb {
    const float* dy;
    const float* x;
    float* ws_dw;
    float* ws_db;
    const float* dy_amax;
    const float* x_amax;
    DwAffine af;
};
struct DwProbs {
    DwProb p[2];
};

// NW (round 6) = waves per workgroup.  With 4 a wave owns 64 x 224 of the tile = 224 accumulator registers, and the ~112 staging
// registers beside them do not fit the other half of the file: the allocator parked them in AGPRs and the loop carried 281
// v_accvgpr_read / _write / _mov per two k-steps next to its 84 MFMAs (a third of its instructions; SQ_INSTS_VALU 13.5 k per
// wave for 1.3 k MFMAs).  With 8 (two per SIMD) a wave owns 32 x 224 = 112 registers, everything else lives in VGPRs, and one
// wave's staging arithmetic runs under the other's MFMAs.
template <int CBW, int DMODE, int NW = 8>
__global__ __launch_bounds__(NW * 64, 1) void proj_dw_f16x3_kernel(const DwProbs pr, int M, int C, int N, int chunks_per_img, int chunk_len) {
    const bool second = blockIdx.z != 0;                         // (workgroup-uniform)
    const float* __restrict__ const dy = second ? pr.p[1].dy : pr.p[0].dy;
    const float* __restrict__ const x = second ? pr.p[1].x : pr.p[0].x;
    float* __restrict__ const ws_dw = second ? pr.p[1].ws_dw : pr.p[0].ws_dw;
    float* __restrict__ const ws_db = second ? pr.p[1].ws_db : pr.p[0].ws_db;
    const float* __restrict__ const dy_amax = second ? pr.p[1].dy_amax : pr.p[0].dy_amax;
    const float* __restrict__ const x_amax = second ? pr.p[1].x_amax : pr.p[0].x_amax;
    const DwAffine af = second ? pr.p[1].af : pr.p[0].af;
    constexpr int NT = NW * 64, PR = NT / 4;                      // threads; rows one piece index covers (a piece = 4 positions of a row)
    constexpr int RB = 8 / NW;                                    // 32-row blocks of dy per wave: 2 (four waves) or 1 (eight)
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr int XROWS = 2 * CBW * 32;                           // x rows staged per k-step
    constexpr int NPA = DW_MROWS / PR;                            // float4 pieces per thread and k-step: dy
    constexpr int NPB = (XROWS + PR - 1) / PR;                    //                                      x
    constexpr int XPAD = NPB * PR;                                // x rows of the LDS image (448 -> 512 with eight waves: the pad rows get zeros)
    constexpr int APLANE = DW_MROWS * DW_ROW, BPLANE = XPAD * DW_ROW;
    constexpr int BUF = 2 * (APLANE + BPLANE);                    // halfs per LDS buffer: A hi, A lo, B hi, B lo
    constexpr int NP = NPA + NPB;
    constexpr int SLOTS = RB * CBW;                               // MFMA triples per k-step and wave
    static_assert(NP <= SLOTS, "more staging pieces than MFMA steps to carry them");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const lds = reinterpret_cast<_Float16*>(smem_raw);  // [2 buf][A hi | A lo | B hi | B lo]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;                      // RB x 32 rows of dy, CBW x 32 rows of x
    // chunk (blockIdx.x) x row half (blockIdx.y); the halves of one chunk are 8k blocks apart = same XCD
    const int s_idx = blockIdx.x, mh = blockIdx.y;
    const int b = s_idx / chunks_per_img;
    const int n_beg = (s_idx - b * chunks_per_img) * chunk_len, n_end = min(N, n_beg + chunk_len);
    const int nsteps = (max(n_end - n_beg, 0) + DW_BK - 1) / DW_BK;
    const int m0 = mh * DW_MROWS;

    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(dy + (size_t)b * M * N, (size_t)M * N * 4);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(x + (size_t)b * C * N, (size_t)C * N * 4);
    const __amdgpu_buffer_rsrc_t a2_rs = DMODE == 1 ? make_rsrc(static_cast<const float*>(af.in2a) + (size_t)b * M * N, (size_t)M * N * 4)
                                       : DMODE == 2 ? make_rsrc(static_cast<const _Float16*>(af.in2a) + (size_t)b * M * N, (size_t)M * N * 2) : a_rs;
    const __amdgpu_buffer_rsrc_t a3_rs = DMODE == 2 ? make_rsrc(static_cast<const _Float16*>(af.in2b) + (size_t)b * M * N, (size_t)M * N * 2) : a_rs;
    const __amdgpu_buffer_rsrc_t cf_rs = make_rsrc(DMODE ? af.coef + (size_t)b * 3 * N : nullptr, DMODE ? (size_t)3 * N * 4 : 0);
    const float sa = dw_scale_from_amax(dy_amax), sb = dw_scale_from_amax(x_amax);
    const float inv_plane_scale = (DMODE == 2 && af.plane_scale_dev) ? 1.0f / *af.plane_scale_dev : af.inv_plane_scale;

    // ---- staging: piece p of this thread = 4 consecutive positions (kq) of one row -----------------------------
    // piece p covers row p*64 + (tid >> 2) of its operand (dy rows first, then x rows), positions kq*4 .. +3 of the
    // k-step with kq = tid & 3 for every piece: one LDS address register + immediates, one position test per step
    static_assert(DW_MROWS % PR == 0, "whole dy pieces only");
    const int kq = tid & 3, prow = tid >> 2;
    unsigned voff[NP];                       // per-lane byte offset of (row, kq*4) in its image; kBufOob: row absent
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const bool isA = p < NPA;
        const int grow = isA ? m0 + p * PR + prow : (p - NPA) * PR + prow;
        voff[p] = (grow < (isA ? M : C)) ? (unsigned)(grow * N + kq * 4) * 4u : kBufOob;
    }
    _Float16* const lds_t = lds + prow * DW_ROW + kq * 4;         // + buffer, operand, plane, p*64 rows: constants
    f32x4 st[2][NP];
    // DMODE != 0: the second operand of the affine rebuild of dy (fp32 piece, or hi | lo halves of a plane piece) and the three
    // coefficient quads of this thread's positions (the same for all of its pieces: kq is fixed), per register stage
    f32x4 st2[2][DMODE ? NPA : 1], cf[2][DMODE ? 3 : 1];
    float dbsum[NPA];
#pragma unroll
    for (int p = 0; p < NPA; ++p) dbsum[p] = 0.f;

    auto fetch_piece = [&](f32x4 (&sg)[NP], int p, int t, int stg = 0) {       // k-step t (beyond the chunk: zeros)
        const int n = n_beg + t * DW_BK;                          // uniform; + kq*4 per lane
        const bool ok = (t < nsteps) && (n + kq * 4 < n_end);
        sg[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            p < NPA ? a_rs : b_rs, (int)(ok ? voff[p] : kBufOob), n * 4, 0));
        if (DMODE && p < NPA) {
            if (DMODE == 1) {
                st2[stg][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a2_rs, (int)(ok ? voff[p] : kBufOob), n * 4, 0));
            } else {
                const u32x2 hw = __builtin_amdgcn_raw_buffer_load_b64(a2_rs, (int)(ok && voff[p] != kBufOob ? voff[p] / 2 : kBufOob), n * 2, 0);
                const u32x2 lw = __builtin_amdgcn_raw_buffer_load_b64(a3_rs, (int)(ok && voff[p] != kBufOob ? voff[p] / 2 : kBufOob), n * 2, 0);
                st2[stg][p] = __builtin_bit_cast(f32x4, u32x4{hw.x, hw.y, lw.x, lw.y});
            }
            if (p == NPA - 1) {      // (with the LAST dy piece: the earlier pieces of this stage are committed with the old quads first)
#pragma unroll
                for (int q = 0; q < 3; ++q)      // (beyond the chunk: zero coefficients -> the rebuilt dy is zero there, gamma included)
                    cf[stg][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        cf_rs, (int)(ok ? (unsigned)(q * N + kq * 4) * 4u : kBufOob), n * 4, 0));
            }
        }
    };
    auto commit_piece = [&](f32x4 (&sg)[NP], int p, int buf, int stg = 0) {
        if (DMODE && p < NPA) {
            f32x4 v2;
            if (DMODE == 1) v2 = st2[stg][p];
            else {
                typedef _Float16 dw_f16x4 __attribute__((ext_vector_type(4)));
                const u32x4 w = __builtin_bit_cast(u32x4, st2[stg][p]);
                const dw_f16x4 h4 = __builtin_bit_cast(dw_f16x4, u32x2{w.x, w.y}), l4 = __builtin_bit_cast(dw_f16x4, u32x2{w.z, w.w});
#pragma unroll
                for (int e = 0; e < 4; ++e) v2[e] = ((float)h4[e] + (float)l4[e]) * inv_plane_scale;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[p][e] = __builtin_fmaf(cf[stg][0][e], sg[p][e], __builtin_fmaf(cf[stg][1][e], v2[e], cf[stg][2][e]));
        }
        u32x2 hi, lo;
        dw_split4(sg[p], p < NPA ? sa : sb, hi, lo);
        if (p < NPA) dbsum[p] += (sg[p][0] + sg[p][1]) + (sg[p][2] + sg[p][3]);
        _Float16* d = lds_t + buf * BUF + (p < NPA ? p * PR * DW_ROW : 2 * APLANE + (p - NPA) * PR * DW_ROW);
        *reinterpret_cast<u32x2*>(d) = hi;
        *reinterpret_cast<u32x2*>(d + (p < NPA ? APLANE : BPLANE)) = lo;
    };

    f32x16 acc[RB][CBW];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;

    // one k-step: this wave's 64 x (CBW*32) block += A(64 x 16) . B(16 x CBW*32); hook(slot) after every MFMA triple
    auto kstep = [&](int buf, auto&& hook) {
        const _Float16* ab = lds + buf * BUF + (wm * RB * 32 + c) * DW_ROW + h * 8;
        const _Float16* bb = lds + buf * BUF + 2 * APLANE + (wn * CBW * 32 + c) * DW_ROW + h * 8;
        f16x8 ah[RB], al[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            ah[rb] = *reinterpret_cast<const f16x8*>(ab + rb * 32 * DW_ROW);
            al[rb] = *reinterpret_cast<const f16x8*>(ab + APLANE + rb * 32 * DW_ROW);
        }
        constexpr int RA = 2;
        f16x8 bh[RA], bl[RA];
#pragma unroll
        for (int i = 0; i < RA - 1 && i < CBW; ++i) {
            bh[i] = *reinterpret_cast<const f16x8*>(bb + i * 32 * DW_ROW);
            bl[i] = *reinterpret_cast<const f16x8*>(bb + BPLANE + i * 32 * DW_ROW);
        }
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            const int cur = cb % RA, nx = cb + RA - 1;
            if (nx < CBW) {
                bh[nx % RA] = *reinterpret_cast<const f16x8*>(bb + nx * 32 * DW_ROW);
                bl[nx % RA] = *reinterpret_cast<const f16x8*>(bb + BPLANE + nx * 32 * DW_ROW);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rb], bh[cur], acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rb], bl[cur], acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rb], bh[cur], acc[rb][cb], 0, 0, 0);
                hook(cb * RB + rb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };

    // prologue (loads in the order in which the loop consumes them: stage 0 = k-step 0, stage 1 = 1, stage 0 = 2)
#pragma unroll
    for (int p = 0; p < NP; ++p) fetch_piece(st[0], p, 0, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) fetch_piece(st[1], p, 1, 1);
#pragma unroll
    for (int p = 0; p < NP; ++p) commit_piece(st[0], p, 0, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) fetch_piece(st[0], p, 2, 0);
    __syncthreads();

    // k-step t: multiply LDS[t&1]; commit stage (t+1)&1 (= k-step t+1) to the other buffer, refill it with t+3
    for (int t = 0; t < nsteps; t += 2) {
        kstep(0, [&](int slot) {
            if (slot < NP) {
                commit_piece(st[1], slot, 1, 1);
                fetch_piece(st[1], slot, t + 3, 1);
            }
        });
        if (t + 1 < nsteps)
            kstep(1, [&](int slot) {
                if (slot < NP) {
                    commit_piece(st[0], slot, 0, 0);
                    fetch_piece(st[0], slot, t + 4, 0);
                }
            });
    }

    // ---- partial tile -> workspace [S][M][C] (lane = x row, register = dy row: 128-byte row segments) ----------
    // (rows padded to whole 128-byte lines: a 32-lane store then never leaves a line half written, which would cost
    //  a read of the line first — 407-float rows did, measured as 2x the algorithmic read traffic)
    const float oscale = 1.0f / (sa * sb);
    const int CP = (C + 31) / 32 * 32;
    float* wsb = ws_dw + (size_t)s_idx * M * CP;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            const int ci = (wn * CBW + cb) * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * RB * 32 + rb * 32 + acc_row_base(r) + 4 * h;
                if (m < M && ci < CP) wsb[(size_t)m * CP + ci] = acc[rb][cb][r] * oscale;
            }
        }
    // ---- bias gradient: a dy row was staged by 4 neighbouring lanes (kq = lane & 3) ----------------------------
    if (ws_db) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            float v = dbsum[p];
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
            const int row = m0 + p * PR + prow;
            if ((tid & 3) == 0 && row < M) ws_db[(size_t)s_idx * M + row] = v;
        }
    }
}

// dw[m][c] = sum_s ws[s][m][c]  (partial tiles of the kernel above, rows padded to CP floats); db likewise.
// Bandwidth-bound: a block takes 64 float4 columns, its four 64-thread groups each sum a quarter of the S partials
// (8 loads in flight), the quarters meet in LDS.
struct DwReduceProbs {
    const float* ws[2];
    float* out[2];
    const float* ws_db[2];
    float* db[2];
};
__global__ __launch_bounds__(256) void proj_dw_reduce_kernel(const DwReduceProbs rp, int S, int M, int C, int CP) {
    const bool second = blockIdx.y != 0;
    const float* __restrict__ const ws = second ? rp.ws[1] : rp.ws[0];
    float* __restrict__ const out = second ? rp.out[1] : rp.out[0];
    const float* __restrict__ const ws_db = second ? rp.ws_db[1] : rp.ws_db[0];
    float* __restrict__ const db = second ? rp.db[1] : rp.db[0];
    __shared__ __attribute__((aligned(16))) float red[3 * 64 * 4];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t n = (size_t)M * CP;                               // floats per partial tile
    const size_t i4 = ((size_t)blockIdx.x * 64 + col) * 4;         // padded index of this thread's float4
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i4 < n) {
        const int per = (S + 3) / 4, s0 = grp * per, s1 = min(S, s0 + per);
        f32x4 t[8];
        int s = s0;
        for (; s + 8 <= s1; s += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(ws + (size_t)(s + u) * n + i4);
#pragma unroll
            for (int u = 0; u < 8; ++u) a += t[u];
        }
        for (; s < s1; ++s) a += *reinterpret_cast<const f32x4*>(ws + (size_t)s * n + i4);
    }
    if (grp > 0) *reinterpret_cast<f32x4*>(red + ((grp - 1) * 64 + col) * 4) = a;
    __syncthreads();
    if (grp == 0 && i4 < n) {
#pragma unroll
        for (int g = 0; g < 3; ++g) a += *reinterpret_cast<const f32x4*>(red + (g * 64 + col) * 4);
        const int m = (int)(i4 / CP), c0 = (int)(i4 - (size_t)m * CP);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < C) out[(size_t)m * C + c0 + e] = a[e];
    }
    // db: row m is summed by block m (mod grid): its S partials one per thread, then a tree in LDS — as a serial
    // loop in one thread this was 128 dependent L2 round trips, the longest pole of the whole kernel
    if (db) {
        for (int m = blockIdx.x; m < M; m += gridDim.x) {
            float v = 0.f;
            for (int s = threadIdx.x; s < S; s += 256) v += ws_db[(size_t)s * M + m];
            __syncthreads();
            red[threadIdx.x] = v;
            __syncthreads();
#pragma unroll
            for (int w = 128; w >= 1; w >>= 1) {
                if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
                __syncthreads();
            }
            if (threadIdx.x == 0) db[m] = red[0];
        }
    }
}

static bool dw_plan(int B, int C, int M, int N, int* chunks_per_img, int* chunk_len, int* cbw, int nprob = 1) {
    if (M < 1 || M > 2 * DW_MROWS || C < 1 || C > 448 || N % 4 != 0 || B < 1) return false;
    *cbw = C <= 256 ? 4 : 7;
    const int mhalves = (M + DW_MROWS - 1) / DW_MROWS;
    // ~256 workgroups, a multiple of 8 chunks so that the row halves of a chunk share an XCD; >= 4 k-steps each
    int want = std::max(1, (256 / mhalves) / (B * nprob));
    const int max_chunks = std::max(1, N / (4 * DW_BK));
    want = std::min(want, max_chunks);
    int len = (N + want - 1) / want;
    len = (len + DW_BK - 1) / DW_BK * DW_BK;
    *chunk_len = len;
    *chunks_per_img = (N + len - 1) / len;
    return true;
}

}  // namespace cocos

// Workspace sizing: number of partial tiles S (0: shape not supported, use cocos_proj1x1_bwd_f16x3);
// ws_dw = S*M*roundup(C,32) floats, ws_db = S*M floats.
extern "C" int cocos_proj1x1_dw_partials_f16x3(int B, int C, int M, int N) {
    int cpi, len, cbw;
    if (!cocos::dw_plan(B, C, M, N, &cpi, &len, &cbw)) return 0;
    return B * cpi;
}

struct DwHostProb {
    const float *dy, *x;
    float *ws_dw, *ws_db, *dw, *db;
    const float *dy_amax, *x_amax;
    cocos::DwAffine af;
};

static int proj_dw_launch(int dmode, int nprob, const DwHostProb* hp, int B, int C, int M, int N, cocos_stream_t stream) {
    using namespace cocos;
    for (int i = 0; i < nprob; ++i) {
        COCOS_REQUIRE(hp[i].dy && hp[i].x && hp[i].ws_dw && hp[i].dw, COCOS_ERR_INVALID, "proj1x1_dw_f16x3: null pointer");
        COCOS_REQUIRE((hp[i].db == nullptr) == (hp[i].ws_db == nullptr), COCOS_ERR_INVALID, "proj1x1_dw_f16x3: db and ws_db go together");
        COCOS_REQUIRE(aligned16(hp[i].dy) && aligned16(hp[i].x) && aligned16(hp[i].ws_dw) && aligned16(hp[i].dw), COCOS_ERR_INVALID,
                      "proj1x1_dw_f16x3: pointers must be 16-byte aligned");
    }
    COCOS_REQUIRE(nprob == 1 || (hp[0].db == nullptr) == (hp[1].db == nullptr), COCOS_ERR_INVALID,
                  "proj1x1_dw_f16x3: the bias gradient for both projections of a pair or for neither");
    int cpi, len, cbw;
    COCOS_REQUIRE(dw_plan(B, C, M, N, &cpi, &len, &cbw, nprob), COCOS_ERR_UNSUPPORTED,
                  "proj1x1_dw_f16x3: needs M <= 256, C <= 448, N %% 4 == 0 (got M=%d C=%d N=%d): use "
                  "cocos_proj1x1_bwd_f16x3", M, C, N);
    COCOS_REQUIRE((size_t)M * N * 4 < 0x7fffffffull && (size_t)C * N * 4 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "proj1x1_dw_f16x3: one sample exceeds 2 GiB");
    hipStream_t s = as_stream(stream);
    const int S = B * cpi, mhalves = (M + DW_MROWS - 1) / DW_MROWS;
    const dim3 grid(S, mhalves, nprob);
    DwProbs pr;
    DwReduceProbs rp;
    for (int i = 0; i < 2; ++i) {
        const DwHostProb& h = hp[i < nprob ? i : 0];
        pr.p[i] = DwProb{h.dy, h.x, h.ws_dw, h.ws_db, h.dy_amax, h.x_amax, h.af};
        rp.ws[i] = h.ws_dw; rp.out[i] = h.dw; rp.ws_db[i] = h.ws_db; rp.db[i] = h.db;
    }
    constexpr int NW = 8, PRH = NW * 64 / 4;                  // (the kernel's NW / PR)
    auto launch = [&](auto kern, int xrows) -> int {
        const int xpad = (xrows + PRH - 1) / PRH * PRH;
        const size_t smem = (size_t)2 * 2 * (DW_MROWS + xpad) * DW_ROW * sizeof(_Float16);
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, s, pr, M, C, N, cpi, len);
        return COCOS_OK;
    };
    int rc;
    if (cbw == 4) rc = dmode == 0 ? launch(proj_dw_f16x3_kernel<4, 0>, 256) : dmode == 1 ? launch(proj_dw_f16x3_kernel<4, 1>, 256)
                                                                                         : launch(proj_dw_f16x3_kernel<4, 2>, 256);
    else rc = dmode == 0 ? launch(proj_dw_f16x3_kernel<7, 0>, 448) : dmode == 1 ? launch(proj_dw_f16x3_kernel<7, 1>, 448)
                                                                                : launch(proj_dw_f16x3_kernel<7, 2>, 448);
    if (rc != COCOS_OK) return rc;
    COCOS_HIP_CHECK(hipGetLastError());
    const int CP = (C + 31) / 32 * 32;
    const size_t n = (size_t)M * CP;
    hipLaunchKernelGGL(proj_dw_reduce_kernel, dim3((unsigned)((n + 255) / 256), nprob), dim3(256), 0, s, rp, S, M, C, CP);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_proj1x1_dw_f16x3(const float* dy, const float* x, float* ws_dw, float* ws_db, float* dw,
                                      float* db, int B, int C, int M, int N, const float* dy_amax,
                                      const float* x_amax, cocos_stream_t stream) {
    const DwHostProb hp{dy, x, ws_dw, ws_db, dw, db, dy_amax, x_amax, cocos::DwAffine{nullptr, nullptr, nullptr, 1.0f, nullptr}};
    return proj_dw_launch(0, 1, &hp, B, C, M, N, stream);
}

// The same reduction with dy REBUILT on the fly (round 6, the companion of cocos_proj_bwd_input_f16x3):
//     dy[b,m,n] = coef[b,0,n] * in1[b,m,n] + coef[b,1,n] * in2[b,m,n] + coef[b,2,n]
// mode 1: in2a = fp32 [B,M,N] (in2b unused);  mode 2: in2a / in2b = channel-major f16 hi / lo planes [B,M,N] of plane_scale * in2.
// dy_amax: device cell with max|dy| (or an upper bound) — the scale source of the f16 split, as for cocos_proj1x1_dw_f16x3.
static int dw_affine_check(int mode, const void* in2a, const void* in2b, const float* coef, float plane_scale, const float* dy_amax) {
    using namespace cocos;
    COCOS_REQUIRE(mode == 1 || mode == 2, COCOS_ERR_INVALID, "proj1x1_dw_affine_f16x3: mode %d", mode);
    COCOS_REQUIRE(in2a && coef && (mode == 1 || in2b) && plane_scale > 0.f && dy_amax, COCOS_ERR_INVALID, "proj1x1_dw_affine_f16x3: null pointer");
    COCOS_REQUIRE(aligned16(in2a) && aligned16(coef) && (mode == 1 || (reinterpret_cast<uintptr_t>(in2b) & 7u) == 0), COCOS_ERR_INVALID,
                  "proj1x1_dw_affine_f16x3: in2 / coef must be 16-byte aligned");
    return COCOS_OK;
}

extern "C" int cocos_proj1x1_dw_affine_f16x3(int mode, const float* in1, const void* in2a, const void* in2b, const float* coef,
                                             float plane_scale, const float* x, float* ws_dw, float* ws_db, float* dw, float* db, int B,
                                             int C, int M, int N, const float* dy_amax, const float* x_amax, cocos_stream_t stream) {
    using namespace cocos;
    const int rc = dw_affine_check(mode, in2a, in2b, coef, plane_scale, dy_amax);
    if (rc != COCOS_OK) return rc;
    const DwHostProb hp{in1, x, ws_dw, ws_db, dw, db, dy_amax, x_amax, DwAffine{coef, in2a, in2b, 1.0f / plane_scale, nullptr}};
    return proj_dw_launch(mode, 1, &hp, B, C, M, N, stream);
}

// Workspace sizing of the PAIR form below: partial tiles per projection.
extern "C" int cocos_proj1x1_dw_partials_pair_f16x3(int B, int C, int M, int N) {
    int cpi, len, cbw;
    if (!cocos::dw_plan(B, C, M, N, &cpi, &len, &cbw, 2)) return 0;
    return B * cpi;
}

// cocos_proj1x1_dw_affine_f16x3 for TWO projections of one shape (theta and phi) in one launch: half as many, twice as long
// position chunks per projection — the chip is filled by the pair, and half the partial tiles are written and summed.
// plane_scale_dev_i (mode 2, nullable): device cell holding the scale of projection i's planes (K25's *y_scale) instead of plane_scale.
extern "C" int cocos_proj1x1_dw_affine_pair_f16x3(
    int mode, float plane_scale, const float* in1_0, const void* in2a_0, const void* in2b_0, const float* coef0, const float* x0,
    float* ws_dw0, float* ws_db0, float* dw0, float* db0, const float* dy_amax0, const float* x_amax0, const float* in1_1,
    const void* in2a_1, const void* in2b_1, const float* coef1, const float* x1, float* ws_dw1, float* ws_db1, float* dw1, float* db1,
    const float* dy_amax1, const float* x_amax1, const float* plane_scale_dev0, const float* plane_scale_dev1, int B, int C, int M, int N,
    cocos_stream_t stream) {
    using namespace cocos;
    int rc = dw_affine_check(mode, in2a_0, in2b_0, coef0, plane_scale, dy_amax0);
    if (rc != COCOS_OK) return rc;
    rc = dw_affine_check(mode, in2a_1, in2b_1, coef1, plane_scale, dy_amax1);
    if (rc != COCOS_OK) return rc;
    const DwHostProb hp[2] = {
        {in1_0, x0, ws_dw0, ws_db0, dw0, db0, dy_amax0, x_amax0, DwAffine{coef0, in2a_0, in2b_0, 1.0f / plane_scale, plane_scale_dev0}},
        {in1_1, x1, ws_dw1, ws_db1, dw1, db1, dy_amax1, x_amax1, DwAffine{coef1, in2a_1, in2b_1, 1.0f / plane_scale, plane_scale_dev1}}};
    return proj_dw_launch(mode, 2, hp, B, C, M, N, stream);
}
