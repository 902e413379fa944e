// K22: the contextual loss without its [N, N] matrices (gfx950).
//
// SURVEY.md §8(f) rank 3 — `ContextualLoss_forward.forward` (models/networks/ContextualLoss.py:93-137), called three or four
// times per generator step on VGG features (pix2pix_model.py:196-203: N = 256 .. 1024 positions at 256^2, up to 4096 at 512^2,
// C = 128 .. 512 channels):
//     cos_ij = <Xn_i, Yn_j>,  d = 1 - cos,  d / (min_j d + eps),  w = exp((1 - .) / h),  A = w / sum_j w,  cx_i = max_j A_ij
// With m_i = max_j cos_ij and tau_i = 1 / (h (1 - m_i + eps)) a row is a softmax whose temperature depends on its own maximum:
//     cx_i = 1 / S_i,   S_i = sum_j e_ij,   e_ij = exp((cos_ij - m_i) tau_i)
// — a QK^T -> row-reduce op of the same class as the correspondence itself, but the temperature is only known once the row's
// maximum is: TWO sweeps over the keys (max, then sums), each a K = C GEMM on the f16 MFMA with split operands (three terms,
// fp32 accumulate: the arithmetic of hgemm_f16x3.hip).  The reference's formulation makes five passes over [B, N, N] and keeps
// four of them for autograd; round 2's K15 still read a materialised cosine matrix (and stopped at 4096 keys).  Here nothing
// N x N reaches HBM, forward or backward, for any N:
//
//   forward  (cf_kernel<0>): a workgroup owns 128 queries and walks the keys twice in tiles of 128; the cosine tile lives in the
//            accumulators (queries in the lanes, keys in the registers), so a lane's running maximum / argmax / sums are
//            its own row's — two half-wave exchanges and one LDS exchange between the two waves of a row at the end of a sweep.
//            Outputs per query: m, S, U = sum_j e_ij (cos_ij - m_i) (the backward's derivative through tau), argmax j*.
//   backward (cf_kernel<1>): G_ij = a_i e_ij (+ a term on column j* only, applied by the caller as a gather / scatter):
//            d Xn[:, i] = sum_j G_ij Yn[:, j]  and  d Yn[:, j] = sum_i G_ij Xn[:, i]  are both
//                out[ch][r] = beta_r * sum_c alpha_c exp2((S_rc - m) t) V[ch][c]
//            with (rows r, inner c) = (queries, keys) resp. (keys, queries) and the per-query (m, t) taken from the rows
//            resp. the inner index: per inner tile the workgroup recomputes the 128 x 128 cosine tile (GEMM 1, K = C), turns it
//            into f16 hi/lo planes of alpha e in LDS, and multiplies them with the inner side's channel-major planes (GEMM 2,
//            256 output channels per workgroup: C = 512 is two workgroups per row block, each recomputing GEMM 1).
//
// Layouts: position-major planes [B][Np][Kp] (k contiguous; Np % 128 == 0, Kp % 32 == 0, padding zero), channel-major value
// planes [B][Cv][Nip]; every plane set carries the power-of-two scale it was multiplied with in a device cell.
#include "common.h"

namespace cocos {

typedef _Float16 cf_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cf_u32x2 __attribute__((ext_vector_type(2)));

constexpr int CF_B = 128;                 // rows per workgroup and inner positions per tile
constexpr int CF_BK = 32, CF_ROW = CF_BK + 8;      // halfs per staged row: 80 B, 16-byte reads of 16 rows hit 16 distinct bank groups
constexpr int CF_PROW = CF_B + 8;         // halfs per row of the P image (272 B = 4 banks mod 64: same property)
constexpr int CF_CW = 256;                // output channels per workgroup (GEMM 2)
constexpr int CF_PLANE = CF_B * CF_ROW;   // one staged 128-row plane (halfs)
constexpr int CF_G1 = 4 * CF_PLANE;       // GEMM 1 stage: inner hi, inner lo, rows hi, rows lo  (40 KB)
constexpr int CF_VPLANE = CF_CW * CF_ROW;
constexpr int CF_G2 = 2 * CF_VPLANE;      // GEMM 2 stage: value hi, value lo  (40 KB)
constexpr int CF_PIMG = CF_B * CF_PROW;   // one P plane (halfs)
constexpr int CF_RED = 256;               // rows of the forward's exchange arrays (the widest workgroup: NJ = 4)
constexpr float CF_PSCALE = 1024.0f;      // P planes hold 2^10 alpha e (|alpha e| <= 1)
constexpr size_t CF_SMEM = (size_t)2 * CF_G1 * 2 + (size_t)2 * CF_PIMG * 2 + (size_t)(3 * CF_B + 4 * CF_RED) * 4;

struct CfArgs {
    const _Float16 *rh, *rl;      // rows side, position-major [B][Nrp][Kp]
    const _Float16 *ih, *il;      // inner side, position-major [B][Nip][Kp]
    const _Float16 *vh, *vl;      // inner side, channel-major [B][Cv][Nip]               (backward)
    const float *s_r, *s_i, *s_v; // device cells: the planes' power-of-two scales
    const float* mul;             // device cell multiplied into the result (nullable)      (backward)
    const float *m, *t;           // per QUERY: max cos and tau * log2(e)                   (backward; [B][Nq])
    const float* alpha;           // per inner position (nullable = 1)                      (backward; [B][Ni])
    const float* alpha_div;       // device cell: alpha is divided by it (nullable = 1)     (backward)
    const float* beta;            // per row (nullable = 1)                                 (backward; [B][Nr])
    const float* gat_src;         // [B][Cv][Ni] fp32 (nullable): out[ch][r] += gat_coef[r] * gat_src[ch][gat_idx[r]]   (backward)
    const int* gat_idx;           // [B][Nr]
    const float* gat_coef;        // [B][Nr]
    float* out;                   // backward: [B][Cv][Nr]
    float *m_out, *s_out, *u_out; // forward: [B][Nr]
    int* j_out;                   // forward: [B][Nr]
    int Nr, Ni, Nrp, Nip, Kp, Cv;
    float h, eps, host_scale;
};

// MODE 0: forward (rows = queries, two sweeps).  MODE 1: backward; STAT_ROWS: (m, t) belong to the rows (d Xn) or to the inner
// index (d Yn).  NJ: 32-row tiles per wave — a workgroup owns BR = 64 NJ rows (128, or 64 when 128-row blocks would leave most of
// the 256 CUs without a workgroup: get_ctx_loss's shapes are 8 samples x 256 .. 1024 positions).
template <int MODE, bool STAT_ROWS, int NJ>
__global__ __launch_bounds__(256, 1) void cf_kernel(const CfArgs a) {
    constexpr int BR = 64 * NJ, WR = 32 * NJ;       // rows per workgroup / per wave
    constexpr int RPLANE = BR * CF_ROW;             // one staged plane of the rows side (halfs)
    constexpr int G1 = 2 * CF_PLANE + 2 * RPLANE;   // GEMM 1 stage: inner hi, inner lo, rows hi, rows lo
    constexpr int STG = MODE == 0 ? G1 : (G1 > CF_G2 ? G1 : CF_G2);      // one staging buffer (the backward's two GEMMs share them)
    static_assert(MODE == 0 || NJ <= 2, "the backward's P image and second accumulator are sized for 128-row workgroups");
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    _Float16* const stage = reinterpret_cast<_Float16*>(cf_smem);                     // [2][STG]
    _Float16* const pimg = stage + 2 * STG;                                           // backward: [hi | lo][128 r][CF_PROW]
    float* const istat = reinterpret_cast<float*>(pimg + 2 * CF_PIMG);                // backward: [m | t | alpha][128] of the inner tile
    float* const red = MODE == 0 ? reinterpret_cast<float*>(stage + 2 * STG) : istat + 3 * CF_B;      // forward: [4][CF_RED] exchange between the waves

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, c = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;          // GEMM 1: inner half (M), row half (N)
    // XCD-aware order (common.h xcd_remap): the workgroups of ONE sample are consecutive virtual ids, i.e. they run on one XCD
    // and walk the sample's key planes in step.  (Measured neutral at B = 8, N = 4096, C = 512 — 1.12 ms either way: the kernel
    // is not bound by where its operands come from, see DESIGN 3.11 — kept because it is the cheaper traffic pattern.)
    const int nrb = (a.Nrp + BR - 1) / BR, nz = MODE == 1 ? (a.Cv + CF_CW - 1) / CF_CW : 1;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / (nrb * nz), rem = vb - b * (nrb * nz);
    const int r0 = (rem % nrb) * BR;
    const int ch0 = MODE == 1 ? (rem / nrb) * CF_CW : 0;

    const size_t rbytes = (size_t)a.Nrp * a.Kp * 2, ibytes = (size_t)a.Nip * a.Kp * 2;
    const __amdgpu_buffer_rsrc_t rh_rs = make_rsrc(a.rh + (size_t)b * a.Nrp * a.Kp, rbytes);
    const __amdgpu_buffer_rsrc_t rl_rs = make_rsrc(a.rl + (size_t)b * a.Nrp * a.Kp, rbytes);
    const __amdgpu_buffer_rsrc_t ih_rs = make_rsrc(a.ih + (size_t)b * a.Nip * a.Kp, ibytes);
    const __amdgpu_buffer_rsrc_t il_rs = make_rsrc(a.il + (size_t)b * a.Nip * a.Kp, ibytes);
    const size_t vbytes = MODE == 1 ? (size_t)a.Cv * a.Nip * 2 : 0;
    const __amdgpu_buffer_rsrc_t vh_rs = make_rsrc(MODE == 1 ? a.vh + (size_t)b * a.Cv * a.Nip : nullptr, vbytes);
    const __amdgpu_buffer_rsrc_t vl_rs = make_rsrc(MODE == 1 ? a.vl + (size_t)b * a.Cv * a.Nip : nullptr, vbytes);

    const float sscale = 1.0f / (*a.s_r * *a.s_i);
    const int nk = a.Kp / CF_BK, ntile = a.Nip / CF_B;

    // ---- GEMM 1 staging: 128 inner rows + BR rows, 4 chunks of 16 B each and plane: 2 resp. NJ chunks per thread and plane ----
    struct G1Regs { u32x4 ih[2], il[2], rh[NJ], rl[NJ]; };
    auto g1_fetch = [&](G1Regs& g, int c0, int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = u * 256 + tid, row = q >> 2, kc = q & 3;
            const unsigned io = (unsigned)((c0 + row) * a.Kp + k0 + kc * 8) * 2u;
            g.ih[u] = __builtin_amdgcn_raw_buffer_load_b128(ih_rs, (int)io, 0, 0);
            g.il[u] = __builtin_amdgcn_raw_buffer_load_b128(il_rs, (int)io, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int q = u * 256 + tid, row = q >> 2, kc = q & 3;
            const unsigned ro = (unsigned)((r0 + row) * a.Kp + k0 + kc * 8) * 2u;
            g.rh[u] = __builtin_amdgcn_raw_buffer_load_b128(rh_rs, (int)ro, 0, 0);
            g.rl[u] = __builtin_amdgcn_raw_buffer_load_b128(rl_rs, (int)ro, 0, 0);
        }
    };
    auto g1_commit = [&](const G1Regs& g, int buf) {
        _Float16* s = stage + buf * STG;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = u * 256 + tid, row = q >> 2, kc = q & 3;
            const int o = row * CF_ROW + kc * 8;
            *reinterpret_cast<u32x4*>(s + o) = g.ih[u];
            *reinterpret_cast<u32x4*>(s + CF_PLANE + o) = g.il[u];
        }
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int q = u * 256 + tid, row = q >> 2, kc = q & 3;
            const int o = row * CF_ROW + kc * 8;
            *reinterpret_cast<u32x4*>(s + 2 * CF_PLANE + o) = g.rh[u];
            *reinterpret_cast<u32x4*>(s + 2 * CF_PLANE + RPLANE + o) = g.rl[u];
        }
    };
    // the cosine tile of inner positions c0 .. c0 + 127: acc[i][j][g] = raw accumulator of
    //     inner  c0 + wm * 64 + i * 32 + acc_row_base(g) + 4 hh     x     row  r0 + wn * WR + j * 32 + c
    // `g` carries k-block 0 of this tile when `have0` (requested during the previous tile: its HBM / L2 latency is behind the
    // previous tile's epilogue instead of in front of this tile's first MFMA); `c0_next` >= 0: request the NEXT tile's k-block 0
    // during this tile's last k-step (forward; the backward does that inside GEMM 2, whose staging shares these buffers).
    auto gemm1 = [&](f32x16 (&acc)[2][NJ], int c0, G1Regs& g, bool have0, int c0_next) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
        if (!have0) g1_fetch(g, c0, 0);
        g1_commit(g, 0);
        __syncthreads();
        for (int kb = 0; kb < nk; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nk) g1_fetch(g, c0, (kb + 1) * CF_BK);
            else if (c0_next >= 0) g1_fetch(g, c0_next, 0);
            const _Float16* ap = stage + buf * STG + (wm * 64 + c) * CF_ROW + hh * 8;
            const _Float16* bp = stage + buf * STG + 2 * CF_PLANE + (wn * WR + c) * CF_ROW + hh * 8;
#pragma unroll
            for (int s = 0; s < CF_BK / 16; ++s) {
                cf_f16x8 bh[NJ], bl[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bh[j] = *reinterpret_cast<const cf_f16x8*>(bp + j * 32 * CF_ROW + s * 16);
                    bl[j] = *reinterpret_cast<const cf_f16x8*>(bp + RPLANE + j * 32 * CF_ROW + s * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const cf_f16x8 ah = *reinterpret_cast<const cf_f16x8*>(ap + i * 32 * CF_ROW + s * 16);
                    const cf_f16x8 al = *reinterpret_cast<const cf_f16x8*>(ap + CF_PLANE + i * 32 * CF_ROW + s * 16);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
            if (kb + 1 < nk) g1_commit(g, buf ^ 1);
            __syncthreads();          // stage[buf ^ 1] is complete; every wave is done with stage[buf]
        }
    };
    auto inner_of = [&](int c0, int i, int g) { return c0 + wm * 64 + i * 32 + acc_row_base(g) + 4 * hh; };

    if constexpr (MODE == 0) {
        // ---------------- forward: sweep 1 (max, argmax), sweep 2 (S, U) ------------------------------------------------
        float best[NJ];
        int arg[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { best[j] = -INFINITY; arg[j] = 0x7fffffff; }
        f32x16 acc[2][NJ];
        G1Regs gr;
        for (int tI = 0; tI < ntile; ++tI) {
            gemm1(acc, tI * CF_B, gr, tI > 0, tI + 1 < ntile ? (tI + 1) * CF_B : 0);     // (the last tile requests tile 0 again: sweep 2)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int ci = inner_of(tI * CF_B, i, g);
                        const float v = ci < a.Ni ? acc[i][j][g] * sscale : -INFINITY;
                        if (v > best[j]) { best[j] = v; arg[j] = ci; }       // (indices ascend within a lane: first maximum wins)
                    }
        }
        // a row's candidates: the two half-waves of this wave, and the wave with the other inner half
        auto pick = [](float& v, int& x, float ov, int ox) {
            if (ov > v || (ov == v && ox < x)) { v = ov; x = ox; }
        };
        float m[NJ];
        int jst[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            pick(best[j], arg[j], __shfl_xor(best[j], 32, 64), __shfl_xor(arg[j], 32, 64));
            if (hh == 0) {
                red[(wm * 2 + 0) * CF_RED + wn * WR + j * 32 + c] = best[j];
                red[(wm * 2 + 1) * CF_RED + wn * WR + j * 32 + c] = __builtin_bit_cast(float, arg[j]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int rr = wn * WR + j * 32 + c;
            m[j] = red[0 * CF_RED + rr];
            jst[j] = __builtin_bit_cast(int, red[1 * CF_RED + rr]);
            pick(m[j], jst[j], red[2 * CF_RED + rr], __builtin_bit_cast(int, red[3 * CF_RED + rr]));
        }
        __syncthreads();
        float t2[NJ], z[NJ], uu[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { t2[j] = kLog2e / (a.h * (1.0f - m[j] + a.eps)); z[j] = 0.f; uu[j] = 0.f; }
        for (int tI = 0; tI < ntile; ++tI) {
            // (the same instruction sequence as sweep 1: bit-identical cosines, e = 1 at the argmax)
            gemm1(acc, tI * CF_B, gr, true, tI + 1 < ntile ? (tI + 1) * CF_B : -1);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int ci = inner_of(tI * CF_B, i, g);
                        const float d = acc[i][j][g] * sscale - m[j];
                        const float e = ci < a.Ni ? fast_exp2(d * t2[j]) : 0.f;
                        z[j] += e;
                        uu[j] += e * d;
                    }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            z[j] += __shfl_xor(z[j], 32, 64);
            uu[j] += __shfl_xor(uu[j], 32, 64);
            if (hh == 0) {
                red[(wm * 2 + 0) * CF_RED + wn * WR + j * 32 + c] = z[j];
                red[(wm * 2 + 1) * CF_RED + wn * WR + j * 32 + c] = uu[j];
            }
        }
        __syncthreads();
        if (wm == 0 && hh == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int rr = wn * WR + j * 32 + c, r = r0 + rr;
                if (r < a.Nr) {
                    const size_t o = (size_t)b * a.Nr + r;
                    a.m_out[o] = m[j];
                    a.s_out[o] = red[0 * CF_RED + rr] + red[2 * CF_RED + rr];
                    a.u_out[o] = red[1 * CF_RED + rr] + red[3 * CF_RED + rr];
                    // (a row of NaN cosines never replaces the sentinel: a valid index keeps the backward's gather in bounds and lets
                    //  the NaNs propagate as NaNs, as the reference does — ADVICE r5)
                    a.j_out[o] = (unsigned)jst[j] < (unsigned)a.Ni ? jst[j] : 0;
                }
            }
        }
        return;
    } else {
        // ---------------- backward ---------------------------------------------------------------------------------------
        const int vm = wave >> 1, vn = wave & 1;     // GEMM 2: channel half (M, 128), row half (N, WR)
        f32x16 acc2[4][NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc2[i][j][g] = 0.f;
        float mr[NJ], tr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            mr[j] = tr[j] = 0.f;
            const int r = r0 + wn * WR + j * 32 + c;
            if (STAT_ROWS && r < a.Nr) { mr[j] = a.m[(size_t)b * a.Nr + r]; tr[j] = a.t[(size_t)b * a.Nr + r]; }
        }
        struct G2Regs { u32x4 vh[4], vl[4]; };
        auto g2_fetch = [&](G2Regs& g, int c0, int kb) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = u * 256 + tid, row = q >> 2, kc = q & 3;
                unsigned off = (unsigned)((ch0 + row) * a.Nip + c0 + kb * CF_BK + kc * 8) * 2u;
                if (ch0 + row >= a.Cv) off = kBufOob;
                g.vh[u] = __builtin_amdgcn_raw_buffer_load_b128(vh_rs, (int)off, 0, 0);
                g.vl[u] = __builtin_amdgcn_raw_buffer_load_b128(vl_rs, (int)off, 0, 0);
            }
        };
        auto g2_commit = [&](const G2Regs& g, int buf) {
            _Float16* s = stage + buf * STG;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = u * 256 + tid, row = q >> 2, kc = q & 3;
                *reinterpret_cast<u32x4*>(s + row * CF_ROW + kc * 8) = g.vh[u];
                *reinterpret_cast<u32x4*>(s + CF_VPLANE + row * CF_ROW + kc * 8) = g.vl[u];
            }
        };
        f32x16 acc[2][NJ];
        G1Regs gr;
        const float adiv = a.alpha_div ? (*a.alpha_div > 0.f ? 1.0f / *a.alpha_div : 0.f) : 1.0f;      // (all-zero d cx: nothing to scale)
        for (int tI = 0; tI < ntile; ++tI) {
            const int c0 = tI * CF_B;
            // the inner tile's per-position numbers (read after the barriers inside gemm1)
            if (tid < CF_B) {
                const int ci = c0 + tid;
                const bool ok = ci < a.Ni;
                const size_t o = (size_t)b * a.Ni + ci;
                if (!STAT_ROWS) {
                    istat[tid] = ok ? a.m[o] : 0.f;
                    istat[CF_B + tid] = ok ? a.t[o] : 0.f;
                }
                istat[2 * CF_B + tid] = ok ? (a.alpha ? a.alpha[o] * adiv : 1.0f) : 0.f;      // padding positions contribute nothing
            }
            gemm1(acc, c0, gr, tI > 0, -1);
            G2Regs g2;
            g2_fetch(g2, c0, 0);              // in flight under the exponentials
            // ---- P = 2^10 alpha_c exp2((S - m) t) as f16 hi / lo planes, [row][inner], inner contiguous ----
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = wm * 64 + i * 32 + 8 * q + 4 * hh;      // inner offset of registers 4q .. 4q+3
                        const f32x4 al4 = *reinterpret_cast<const f32x4*>(istat + 2 * CF_B + cl);
                        f32x4 m4, t4;
                        if (!STAT_ROWS) {
                            m4 = *reinterpret_cast<const f32x4*>(istat + cl);
                            t4 = *reinterpret_cast<const f32x4*>(istat + CF_B + cl);
                        }
                        float p[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float mm = STAT_ROWS ? mr[j] : m4[e], tt = STAT_ROWS ? tr[j] : t4[e];
                            p[e] = fast_exp2((acc[i][j][4 * q + e] * sscale - mm) * tt) * (al4[e] * CF_PSCALE);
                        }
                        unsigned h0, l0, h1, l1;
                        split_pair_rn(p[0], p[1], h0, l0);
                        split_pair_rn(p[2], p[3], h1, l1);
                        _Float16* dst = pimg + (wn * WR + j * 32 + c) * CF_PROW + cl;
                        *reinterpret_cast<cf_u32x2*>(dst) = cf_u32x2{h0, h1};
                        *reinterpret_cast<cf_u32x2*>(dst + CF_PIMG) = cf_u32x2{l0, l1};
                    }
            g2_commit(g2, 0);                 // (gemm1 ended with a barrier: the staging buffers are free)
            __syncthreads();                  // P and value block 0 are complete
            // ---- GEMM 2: out[ch][r] += V[ch][c] P[r][c] ----
            for (int kb = 0; kb < CF_B / CF_BK; ++kb) {
                const int buf = kb & 1;
                if (kb + 1 < CF_B / CF_BK) g2_fetch(g2, c0, kb + 1);
                else if (tI + 1 < ntile) g1_fetch(gr, c0 + CF_B, 0);      // the next tile's first GEMM-1 block (see gemm1)
                const _Float16* ap = stage + buf * STG + (vm * 128 + c) * CF_ROW + hh * 8;
                const _Float16* bp = pimg + (vn * WR + c) * CF_PROW + kb * CF_BK + hh * 8;
#pragma unroll
                for (int s = 0; s < CF_BK / 16; ++s) {
                    cf_f16x8 bh[NJ], bl[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        bh[j] = *reinterpret_cast<const cf_f16x8*>(bp + j * 32 * CF_PROW + s * 16);
                        bl[j] = *reinterpret_cast<const cf_f16x8*>(bp + CF_PIMG + j * 32 * CF_PROW + s * 16);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const cf_f16x8 ah = *reinterpret_cast<const cf_f16x8*>(ap + i * 32 * CF_ROW + s * 16);
                        const cf_f16x8 al = *reinterpret_cast<const cf_f16x8*>(ap + CF_VPLANE + i * 32 * CF_ROW + s * 16);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc2[i][j], 0, 0, 0);
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc2[i][j], 0, 0, 0);
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc2[i][j], 0, 0, 0);
                        }
                    }
                }
                if (kb + 1 < CF_B / CF_BK) g2_commit(g2, buf ^ 1);
                __syncthreads();
            }
        }
        const float gscale = a.host_scale * (a.mul ? *a.mul : 1.0f) / (*a.s_v * CF_PSCALE);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = r0 + vn * WR + j * 32 + c;
            const float bsc = (r < a.Nr && a.beta) ? a.beta[(size_t)b * a.Nr + r] * gscale : gscale;
            // the argmax column's extra term (through m_i and tau(m_i)): a gather from the inner side's fp32 values
            const bool gat = a.gat_src != nullptr && r < a.Nr;
            const float gco = gat ? a.gat_coef[(size_t)b * a.Nr + r] : 0.f;
            const float* gsrc = gat ? a.gat_src + (size_t)b * a.Cv * a.Ni + a.gat_idx[(size_t)b * a.Nr + r] : nullptr;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int ch = ch0 + vm * 128 + i * 32 + acc_row_base(g) + 4 * hh;
                    if (ch < a.Cv && r < a.Nr) {
                        float v = acc2[i][j][g] * bsc;
                        if (gat) v += gco * gsrc[(size_t)ch * a.Ni];
                        a.out[((size_t)b * a.Cv + ch) * a.Nr + r] = v;
                    }
                }
        }
    }
}

// The backward's per-query coefficients from the forward's statistics and d loss / d cx (one launch instead of ~10 framework
// element-wise ones):  dS = -dcx / S^2,  tau = 1 / (h (1 - m + eps)),  a = dS tau (coefficient of e_ij),  t2 = tau log2 e,
// extra = dS (h tau^2 U - tau S) (the argmax column's term),  *amax |= max |a| (cell zero on entry: non-negative floats order as integers).
__global__ __launch_bounds__(256) void cf_coeffs_kernel(const float* __restrict__ dcx, const float* __restrict__ S,
                                                        const float* __restrict__ U, const float* __restrict__ m, float* __restrict__ a_out,
                                                        float* __restrict__ t2_out, float* __restrict__ extra_out,
                                                        unsigned* __restrict__ amax, long long n, float h, float eps) {
    __shared__ float red[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float av = 0.f;
    if (i < n) {
        const float s = S[i], dS = -dcx[i] / (s * s), tau = 1.0f / (h * (1.0f - m[i] + eps));
        const float aa = dS * tau;
        a_out[i] = aa;
        t2_out[i] = tau * kLog2e;
        extra_out[i] = dS * (h * tau * tau * U[i] - tau * s);
        av = fabsf(aa);
    }
    av = wave_max_dpp(av);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = av;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

static int cf_check_planes(const char* what, int B, int Nr, int Ni, int Nrp, int Nip, int Kp) {
    COCOS_REQUIRE(B >= 1 && Nr >= 1 && Ni >= 1 && Nrp >= Nr && Nip >= Ni && Kp >= 32, COCOS_ERR_INVALID,
                  "%s: bad dims B=%d Nr=%d Ni=%d Nrp=%d Nip=%d Kp=%d", what, B, Nr, Ni, Nrp, Nip, Kp);
    COCOS_REQUIRE(Nrp % CF_B == 0 && Nip % CF_B == 0 && Kp % CF_BK == 0, COCOS_ERR_INVALID,
                  "%s: planes must be padded to 128 positions and 32 channels (Nrp=%d Nip=%d Kp=%d)", what, Nrp, Nip, Kp);
    COCOS_REQUIRE((size_t)Nrp * Kp * 2 < 0x7fffffffull && (size_t)Nip * Kp * 2 < 0x7fffffffull && B <= 65535, COCOS_ERR_UNSUPPORTED,
                  "%s: per-sample planes exceed 2 GiB (or B > 65535)", what);
    return COCOS_OK;
}

}  // namespace cocos

// Forward of the contextual loss's row statistics (see the header of this file).  xh, xl [B][Nqp][Kp], yh, yl [B][Nkp][Kp]:
// position-major f16 hi / lo planes of the normalised features times *x_scale_dev resp. *y_scale_dev (cocos_split_f16_ex),
// zero beyond Nq / Nk / the real channels.  Per query i < Nq:  m_out = max_j cos_ij,  s_out = S_i = sum_j exp((cos_ij - m_i) tau_i)
// with tau_i = 1 / (h (1 - m_i + eps)) (cx_i = 1 / S_i),  u_out = sum_j e_ij (cos_ij - m_i),  j_out = argmax_j cos_ij (first).
extern "C" int cocos_contextual_cx_fwd_f16x3(const void* xh, const void* xl, const void* yh, const void* yl,
                                             const float* x_scale_dev, const float* y_scale_dev, float* m_out, float* s_out,
                                             float* u_out, int* j_out, int B, int Nq, int Nk, int Nqp, int Nkp, int Kp, float h,
                                             float eps, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(xh && xl && yh && yl && x_scale_dev && y_scale_dev && m_out && s_out && u_out && j_out, COCOS_ERR_INVALID,
                  "contextual_cx_fwd_f16x3: null pointer");
    COCOS_REQUIRE(h > 0.f && eps > 0.f, COCOS_ERR_INVALID, "contextual_cx_fwd_f16x3: h=%g eps=%g", (double)h, (double)eps);
    if (int rc = cf_check_planes("contextual_cx_fwd_f16x3", B, Nq, Nk, Nqp, Nkp, Kp)) return rc;
    for (const void* p : {xh, xl, yh, yl}) COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "contextual_cx_fwd_f16x3: planes must be 16-byte aligned");
    CfArgs a{};
    a.rh = static_cast<const _Float16*>(xh); a.rl = static_cast<const _Float16*>(xl);
    a.ih = static_cast<const _Float16*>(yh); a.il = static_cast<const _Float16*>(yl);
    a.s_r = x_scale_dev; a.s_i = y_scale_dev;
    a.m_out = m_out; a.s_out = s_out; a.u_out = u_out; a.j_out = j_out;
    a.Nr = Nq; a.Ni = Nk; a.Nrp = Nqp; a.Nip = Nkp; a.Kp = Kp; a.h = h; a.eps = eps;
    // rows per workgroup: 256 when that still gives every CU a workgroup (the staged key block is re-used by twice the rows: the
    // kernel streams (128 + rows) x C operand elements per 128 x rows cosines from L2), 128, or 64 when 128-row blocks would leave
    // most of the chip without one
    const long long blocks128 = (long long)(Nqp / CF_B) * B;
    const int nj = blocks128 >= 512 ? 4 : blocks128 >= 256 ? 2 : 1;
    auto kern = nj == 4 ? cf_kernel<0, true, 4> : nj == 2 ? cf_kernel<0, true, 2> : cf_kernel<0, true, 1>;
    const size_t smem = (size_t)2 * (2 * CF_PLANE + 2 * 64 * nj * CF_ROW) * 2 + (size_t)4 * CF_RED * 4;
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const long long nblk = (long long)((Nqp + 64 * nj - 1) / (64 * nj)) * B;
    COCOS_REQUIRE(nblk <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "contextual_cx_fwd_f16x3: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, as_stream(stream), a);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// a, t2, extra [n] from dcx, S, U, m [n] (see cf_coeffs_kernel); *a_amax_dev (zero on entry) = max |a|.
extern "C" int cocos_contextual_cx_coeffs(const float* dcx, const float* S, const float* U, const float* m, float* a_out, float* t2_out,
                                          float* extra_out, float* a_amax_dev, long long n, float h, float eps, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(dcx && S && U && m && a_out && t2_out && extra_out && a_amax_dev, COCOS_ERR_INVALID, "contextual_cx_coeffs: null pointer");
    COCOS_REQUIRE(n >= 1 && (n + 255) / 256 <= 0x7fffffffLL && h > 0.f && eps > 0.f, COCOS_ERR_INVALID, "contextual_cx_coeffs: n=%lld", n);
    hipLaunchKernelGGL(cf_coeffs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), dcx, S, U, m, a_out, t2_out,
                       extra_out, reinterpret_cast<unsigned*>(a_amax_dev), n, h, eps);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// One side of the backward:  out[b][ch][r] = host_scale * mul * beta_r * sum_c (alpha_c / alpha_div) exp2((cos_rc - m) t) V[ch][c]
//                                            (+ gather_coef[r] * gather_src[ch][gather_idx[r]] when gather_src is given)  for r < Nr,
// ch < Cv, with (m, t) per QUERY — indexed by r when stats_on_rows (rows = queries: d Xn, V = Yn) and by c otherwise (rows = keys:
// d Yn, V = Xn).  rh, rl / ih, il: position-major planes of the rows / inner side (as above), vh, vl [B][Cv][Nip]: channel-major
// planes of the inner side's values times *v_scale_dev; alpha [B][Ni] (|alpha| <= 1), beta [B][Nr], mul: nullable = 1.
extern "C" int cocos_contextual_cx_bwd_f16x3(const void* rh, const void* rl, const void* ih, const void* il, const void* vh,
                                             const void* vl, const float* r_scale_dev, const float* i_scale_dev,
                                             const float* v_scale_dev, const float* mul_dev, const float* m, const float* t,
                                             const float* alpha, const float* alpha_div_dev, const float* beta,
                                             const float* gather_src, const int* gather_idx, const float* gather_coef, float* out,
                                             int B, int Nr, int Ni, int Nrp, int Nip, int Kp, int Cv, int stats_on_rows,
                                             float host_scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(rh && rl && ih && il && vh && vl && r_scale_dev && i_scale_dev && v_scale_dev && m && t && out, COCOS_ERR_INVALID,
                  "contextual_cx_bwd_f16x3: null pointer");
    if (int rc = cf_check_planes("contextual_cx_bwd_f16x3", B, Nr, Ni, Nrp, Nip, Kp)) return rc;
    COCOS_REQUIRE(Cv >= 1 && (size_t)Cv * Nip * 2 < 0x7fffffffull && (Cv + CF_CW - 1) / CF_CW <= 65535, COCOS_ERR_UNSUPPORTED,
                  "contextual_cx_bwd_f16x3: Cv=%d", Cv);
    for (const void* p : {rh, rl, ih, il, vh, vl}) COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "contextual_cx_bwd_f16x3: planes must be 16-byte aligned");
    CfArgs a{};
    a.rh = static_cast<const _Float16*>(rh); a.rl = static_cast<const _Float16*>(rl);
    a.ih = static_cast<const _Float16*>(ih); a.il = static_cast<const _Float16*>(il);
    a.vh = static_cast<const _Float16*>(vh); a.vl = static_cast<const _Float16*>(vl);
    a.s_r = r_scale_dev; a.s_i = i_scale_dev; a.s_v = v_scale_dev; a.mul = mul_dev;
    COCOS_REQUIRE(!gather_src || (gather_idx && gather_coef), COCOS_ERR_INVALID, "contextual_cx_bwd_f16x3: gather needs its index and coefficient");
    a.m = m; a.t = t; a.alpha = alpha; a.alpha_div = alpha_div_dev; a.beta = beta; a.out = out;
    a.gat_src = gather_src; a.gat_idx = gather_idx; a.gat_coef = gather_coef;
    a.Nr = Nr; a.Ni = Ni; a.Nrp = Nrp; a.Nip = Nip; a.Kp = Kp; a.Cv = Cv; a.host_scale = host_scale;
    const int nz = (Cv + CF_CW - 1) / CF_CW;
    const bool small = (long long)(Nrp / CF_B) * B * nz < 256;
    auto kern = small ? (stats_on_rows ? cf_kernel<1, true, 1> : cf_kernel<1, false, 1>)
                      : (stats_on_rows ? cf_kernel<1, true, 2> : cf_kernel<1, false, 2>);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_SMEM));
    const long long nblk = (long long)(Nrp / (small ? 64 : 128)) * B * nz;
    COCOS_REQUIRE(nblk <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "contextual_cx_bwd_f16x3: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), CF_SMEM, as_stream(stream), a);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
