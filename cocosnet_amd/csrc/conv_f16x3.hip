// K16: 2-D convolution as an implicit GEMM on v_mfma_f32_32x32x16_f16, fp32 in / fp32 out, every product carried as
// f16 hi + lo with three MFMA terms and fp32 accumulation (the arithmetic of corr_fused_fwd_f16x3.hip) — gfx950.
//
// the 3x3 convolutions in front of the path — `ResidualBlock` x4 (correspondence.py:13-36,
// :175-179: reflect-pad -> 3x3, 407 -> 407 channels on the 64x64 grid) and the adaptors (correspondence.py:150-173) — and,
// riding along, the k4 s2 convolutions of the PatchGAN (discriminator.py:92-115).  The features they produce are
// L2-normalised and multiplied by 1/T = 100 inside the softmax: plain f16/bf16 convolutions would break the 1e-3 parity
// of the path they feed, the fp32 MFMA runs at 1/16 of the f16 rate, hence the split.
//
// The contraction index is ordered by BLOCKS OF 32 CHANNELS, taps inside:  k = ((ci/32) * T + tap) * 32 + ci%32,
// tap = ky*KW + kx, T = KH*KW, channels zero-padded to Cp = 32*ceil(Cin/32).  A k-block of 32 is then 32 consecutive
// channels at ONE tap: its addresses are a per-thread constant plus a per-step scalar, its border masks are those of
// the tap — no per-element index arithmetic in the steady state — and the T k-blocks that follow each other read the
// SAME 32 channels shifted by a pixel or a row, so the window comes from L1/L2 after its first touch.  (Measured on the
// earlier orderings: with (ci,ky,kx) the integer work of the gather set the speed — a wave that owns a SIMD issues ~1
// instruction / 4-5 cycles, ~7 per 32-cycle MFMA; with taps outermost every tap pass streamed all channels of the
// XCD's images through its 4 MB L2 again.)
//
//   forward / input gradient (conv_fwd_kernel):   Y[b,co,oy,ox] = bias[co] + sum_k Wt[co,k] X[b,ci,oy*s+ky-p,ox*s+kx-p]
//       GEMM  C[M = Cout][N = B*OH*OW] over K = T*Cp;  A = weight planes (pre-split f16 hi/lo, k-block major
//       [K/32][Cout][32], zero for ci >= Cin), B = the im2col matrix, never materialised: each k-block of 32 channels x 128
//       positions is GATHERED from the fp32 input (zero padding = masked loads), split to hi/lo on the fly and written
//       to LDS position-contiguous; the MFMA B fragments (8 consecutive k for one position) come out of that image
//       through ds_read_b64_tr_b16, the LDS transpose read of gfx950 (lane i of a 16-lane group supplies the address of
//       piece (row i>>2, columns 4(i&3)..+3) of a 4 x 16 block and receives column i — probed in
//       tools/probes/tr16_probe.hip).  The input gradient of a stride-1 convolution is the same kernel on dY with the
//       flipped, transposed weights and padding K-1-p.
//   weight gradient (conv_wgrad_kernel):   dW[co,k] = sum_{b,oy,ox} dY[b,co,oy,ox] X[b,ci,oy*s+ky-p,ox*s+kx-p]
//       GEMM  C[M = Cout][N = T*Cp] over the B*OH*OW positions, split over position chunks (few output tiles, long
//       contraction); both operands are position-contiguous in memory, so both LDS images are k-contiguous and the
//       fragments are plain 16-byte reads.  A thread's rows (co, k) are fixed for the whole kernel; only the position
//       advances.
// Tile BM (128 | 256) x 128 x 32, 4 waves as 2 x 2, double-buffered LDS, staging work handed out between the MFMAs by a
// sched_group_barrier pipeline; ~2*M*N*K useful FLOPs, x3 issued.  Compile-time knobs below = measured experiments
// (DESIGN.md section 3.3), the defaults are what shipped.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifndef COCOS_CONV_SCHED_N
#define COCOS_CONV_SCHED_N 6    // non-MFMA instructions handed out per MFMA gap (4..8 measured equal; pinning the staging
#endif                          // pieces after each group of 6 MFMAs instead: 0.491 vs 0.470 ms on the 407-channel block)
#ifndef COCOS_CONV_WGRAD_PIPE
#define COCOS_CONV_WGRAD_PIPE 0  // weight gradient: 1 = the forward kernel's step shape (mid barrier, fragments half a step ahead, one stage):
                                // measured neutral (1.591 vs 1.582 ms fwd+bwd) and it spills, so 0 = barrier at the end, two stages
#endif
#ifndef COCOS_CONV_OCC2
#define COCOS_CONV_OCC2 1       // 128-row tiles: two workgroups per CU (2 waves per SIMD, 80 KB of LDS each)
#endif
#ifndef COCOS_CONV_STAGES
#define COCOS_CONV_STAGES 0     // 0: per tile shape (see conv_fwd_kernel); 1 | 2: forced (timing experiments)
#endif
#ifndef COCOS_CONV_ABLATE
#define COCOS_CONV_ABLATE 0     // timing experiments only (tools/build_conv_ablations.sh): 1 no gather loads, 2 no weight
#endif                          // loads, 4 no LDS commit, 8 no MFMA

namespace cocos {

// Optional phase timing (build with COCOS_EXTRA_HIPFLAGS=-DCOCOS_DEBUG_TIMING): shader-clock cycles spent by thread 0 of
// workgroup 0 in the two halves of a forward step and at its barrier — cocos_debug_read_timing_conv().
#ifdef COCOS_DEBUG_TIMING
__device__ long long g_phase_conv[8];
#define CPH_T(var) const long long var = __builtin_readcyclecounter()
#define CPH_ADD(i, a, b) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_conv[i] += (b) - (a); } while (0)
#else
#define CPH_T(var) do {} while (0)
#define CPH_ADD(i, a, b) do {} while (0)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int CV_BN = 128, CV_BK = 32;
constexpr int CV_AROW = CV_BK + 8;      // halfs per row of a k-contiguous image (80 B: conflict-free b128 reads)

// n / d for 0 <= n < 2^31, branch-free:  (umulhi(n, mul) + (add ? n : 0)) >> sh.
//   d not a power of two: sh = floor(log2 d), mul = ceil(2^(32+sh) / d), add = 0 — exact because the rounding excess of
//   the multiplier is < d < 2^(sh+1), so n * excess < 2^(32+sh);   d = 2^sh: mul = 0, add = 1.
struct Magic { unsigned mul, sh, add; };
__device__ __forceinline__ int cv_div(int n, Magic m) {
    return (int)((__umulhi((unsigned)n, m.mul) + (unsigned)n * m.add) >> m.sh);
}

struct ConvGeom {
    int Cin, H, W, OH, OW, KH, KW, stride, pad, padx, dil;   // pad = rows, padx = columns (equal except for the strided dgrad)
    // where output pixel (oy, ox) of plane (b, m) goes: Y[(b*M + m)*o_plane + o_off + oy*o_pitch + ox*o_cs]
    // (dense: o_plane = OH*OW, o_pitch = OW, o_cs = 1, o_off = 0; the parity classes of a strided input gradient scatter)
    int o_plane, o_pitch, o_cs, o_off;
    int Cp;                  // Cin rounded up to 32
    int Ktot;                // KH * KW * Cp
    int OWv;                 // forward kernel, stride 1: OW rounded up to 4 (virtual columns, computed and dropped) so
                             // that 4 consecutive GEMM columns are always 4 consecutive pixels of ONE row; else OW
    int Ntot;                // B * OH * OWv  (GEMM columns)
    int xelems;              // B * Cin * H * W
    Magic mT, mKW, mOHW, mOW;            // / (KH*KW), / KW, / (OH*OWv), / OWv
};

__device__ __forceinline__ float cv_scale_from_amax(const float* amax) {
    if (!amax) return 1.0f;
    const float a = *amax;
    if (!(a > 0.f) || !(a < INFINITY)) return 1.0f;
    int e;
    frexpf(a, &e);
    return ldexpf(1.0f, 10 - e);
}

// ONE-term flavour (BASELINE config 3's precision for the SPADE / PatchGAN convolutions: "bf16 MFMA"): operands are single
// bf16 planes (fp32 range: no amax pass, no power-of-two scales), one v_mfma_f32_32x32x16_bf16 per product instead of three
// f16 ones, fp32 accumulate.  8 mantissa bits per operand: NOT for anything upstream of the correlation (its features end
// up times 100 inside a softmax) — for the generator / discriminator stacks behind InstanceNorm / SPADE.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cv_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cv_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cv_pack_bf16(float a, float b) {     // round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(cv_f32x2{a, b}, cv_bf16x2));
}
__device__ __forceinline__ f32x16 cv_mfma_bf16(f16x8 a, f16x8 b, f32x16 c) {     // the planes are typed f16x8 in this file
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// x*s -> f16 hi (round to nearest) + f16 lo, four values
__device__ __forceinline__ void cv_split4(const float (&x)[4], float s, u32x2& hi, u32x2& lo) {
    unsigned h0, l0, h1, l1;
    split_pair_rn(x[0] * s, x[1] * s, h0, l0);
    split_pair_rn(x[2] * s, x[3] * s, h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}

// GEMM column n -> (image b, output pixel): element offset of X[b, 0, oy*s - p, ox*s - p] (may be negative) and the
// input coordinates of the window's corner.
struct Corner { int base, iy0, ix0; };
__device__ __forceinline__ Corner cv_corner(int n, const ConvGeom& g) {
    const int b = cv_div(n, g.mOHW);
    const int pos = n - b * g.OH * g.OWv;
    const int oy = cv_div(pos, g.mOW);
    const int ox = pos - oy * g.OWv;
    Corner c;
    c.iy0 = oy * g.stride - g.pad;
    c.ix0 = ox * g.stride - g.padx;
    c.base = (b * g.Cin * g.H + c.iy0) * g.W + c.ix0;
    return c;
}

__device__ __forceinline__ f32x4 buf_load4s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, (int)soff, 0));
}

// --------------------------------------------------------------------------------------------------------------------
// forward / input gradient
// --------------------------------------------------------------------------------------------------------------------
// Staging: two register stages; step t commits tile t+1 (fetched during step t-1) to the other LDS buffer under the
// first half of its MFMAs and fetches tile t+3 into the same registers under the second half, so every load has more
// than a full step to land.  The pieces are placed BETWEEN the MFMA groups by hand (sched_barrier): VALU, LDS writes
// and load issue then run in the shadow of the matrix pipe instead of after it.
// FAST4 (stride 1; rows padded to OWv): a thread's 4 positions are consecutive pixels of one row: ONE 16-byte load per piece
// (4-byte aligned) whose out-of-row neighbours are zeroed at commit by the tap's column masks.  Otherwise 4 element
// loads with the mask folded into the offset.
// The X descriptor starts `shift` = pad*W + pad elements BEFORE the tensor so that every per-thread offset constant is
// >= 0 (the step offset travels in the scalar offset, which the hardware does not range-check): nothing below X is ever
// dereferenced — a lane whose window corner lies outside the image is masked to the out-of-range offset.
template <int BM, int BN, bool FAST4, bool ONE = false>
__global__ __launch_bounds__(256, (BM == 128 && COCOS_CONV_OCC2) ? 2 : 1) void conv_fwd_kernel(const float* __restrict__ X, const _Float16* __restrict__ wh,
                                                          const _Float16* __restrict__ wl,
                                                          const float* __restrict__ w_scale, const float* __restrict__ x_amax,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int M,
                                                          ConvGeom g) {
    constexpr int MI = BM / 64, NJ = BN / 64;          // 32 x 32 blocks of a wave's tile (waves as 2 x 2)
    // register stages: ONE (a tile fetched under the second half of step t is committed under the first half of step
    // t+1); two measured slower once the fragments were double-buffered (0.455 vs 0.442 ms: register spills)
    constexpr int STAGES = COCOS_CONV_STAGES ? COCOS_CONV_STAGES : 1;
    constexpr int LPR = BN / 4;                        // lanes per k row of the gathered tile (4 positions each)
    constexpr int RPP = 256 / LPR;                     // k rows per pass of the 256 threads
    constexpr int NP = CV_BK / RPP;                    // gathered pieces per thread (4 | 8)
    constexpr int GPS = NP / MI;                       // gathered pieces per staging slot
    constexpr int GROW = BN + 32;                      // halfs per k row: 64 B more than a multiple of 256 B, so the four
                                                       // rows of a transpose read fall into different quarters of the banks
    constexpr int APLANE = BM * CV_AROW, GPLANE = CV_BK * GROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const at = reinterpret_cast<_Float16*>(smem_raw);   // [2 buf][hi|lo][BM][AROW]
    _Float16* const gt = at + 2 * 2 * APLANE;                      // [2 buf][hi|lo][32 k][GROW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int cg = tid % LPR, rowsub = tid / LPR;      // gather: column group (4 positions) and k row within a pass
    const int ntn = (g.Ntot + BN - 1) / BN;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (vb / ntn) * BM, n0 = (vb % ntn) * BN;          // consecutive ids: the position tiles of one row tile
    const int HW = g.H * g.W, ncb = g.Cp >> 5, nkb = g.KH * g.KW * ncb;
    const int shift = g.pad * g.W + g.padx;

    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(X - shift, ((size_t)g.xelems + shift) * 4);
    const size_t wbytes = (size_t)M * g.Ktot * 2;
    const __amdgpu_buffer_rsrc_t wh_rs = make_rsrc(wh, wbytes), wl_rs = make_rsrc(wl, wbytes);
    const float sx = ONE ? 1.0f : cv_scale_from_amax(x_amax);

    // the thread's columns: 4 consecutive GEMM columns n .. n+3
    constexpr int NC = FAST4 ? 1 : 4;
    Corner cr[NC];
    bool live[NC];
#pragma unroll
    for (int e = 0; e < NC; ++e) {
        const int n = n0 + 4 * cg + e;
        cr[e] = cv_corner(min(n, g.Ntot - 1), g);
        live[e] = n < g.Ntot;
    }
    // only windows at the very beginning / end of the tensor can see a 16-byte piece cross its ends (wave-uniform flag)
    const int maxoff = ((g.Cin - 1) * g.H + (g.KH - 1) * g.dil) * g.W + (g.KW - 1) * g.dil;
    const bool edge_tile = FAST4 && __builtin_amdgcn_ballot_w64(cr[0].base < 0 || cr[0].base + maxoff + 4 > g.xelems) != 0;
    // piece u = channel u*RPP + rowsub of the k-block, the thread's columns
    int rowlane[NP];
    unsigned vconst[NP][NC];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        rowlane[u] = u * RPP + rowsub;
#pragma unroll
        for (int e = 0; e < NC; ++e) vconst[u][e] = (unsigned)(cr[e].base + shift + rowlane[u] * HW) * 4u;
    }
    unsigned voffa[MI];                                // weight tile: 16-byte chunk idx of the BM x 32 tile
#pragma unroll
    for (int u = 0; u < MI; ++u) voffa[u] = (m0 + ((u * 256 + tid) >> 2) < M) ? (unsigned)(u * 256 + tid) * 16u : kBufOob;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Stage {
        u32x4 a[2][MI];
        float gv[NP][4];
        bool mk[4];        // FAST4: column e of the piece lies inside the image row (same for the 4 pieces of a step)
    };
    Stage st[STAGES];
    // per-step scalars of a fetch (set by fetch_begin)
    int f_cb = 0, f_ky = 0, f_kx = 0;
    unsigned f_soff = 0, f_soffa = 0;
    bool f_rowok[NC];
    auto fetch_begin = [&](Stage& S, int tt) {
        tt = min(tt, nkb - 1);                         // prefetches beyond the end re-read the last tile (never used)
        if (COCOS_CONV_ABLATE & 16) tt = 0;            // timing experiment: every step re-reads tile 0 (cache-hot loads)
        f_cb = cv_div(tt, g.mT);                       // k-block tt = 32 channels f_cb*32.. at tap tt % T
        const int tap = tt - f_cb * (g.KH * g.KW);
        f_ky = cv_div(tap, g.mKW);
        f_kx = (tap - f_ky * g.KW) * g.dil;            // from here on: the tap's offset in input pixels
        f_ky *= g.dil;
        f_soff = (unsigned)(f_cb * 32 * HW + f_ky * g.W + f_kx) * 4u;
        f_soffa = (unsigned)(tt * M + m0) * 64u;
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            f_rowok[e] = live[e] && (unsigned)(cr[e].iy0 + f_ky) < (unsigned)g.H;
            if (!FAST4) f_rowok[e] = f_rowok[e] && (unsigned)(cr[e].ix0 + f_kx) < (unsigned)g.W;
        }
        if (FAST4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) S.mk[e] = (unsigned)(cr[0].ix0 + f_kx + e) < (unsigned)g.W;
        }
    };
    auto fetch_a = [&](Stage& S, int u) {
        if (COCOS_CONV_ABLATE & 2) { S.a[0][u] = S.a[1][u] = u32x4{f_soffa, 0u, 0u, 0u}; return; }
        S.a[0][u] = __builtin_amdgcn_raw_buffer_load_b128(wh_rs, (int)voffa[u], (int)f_soffa, 0);
        if (!ONE) S.a[1][u] = __builtin_amdgcn_raw_buffer_load_b128(wl_rs, (int)voffa[u], (int)f_soffa, 0);
    };
    auto fetch_g = [&](Stage& S, int u, auto edge_tag) __attribute__((always_inline)) {
        if (COCOS_CONV_ABLATE & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) S.gv[u][e] = (float)(f_soff + u + e) * 1e-9f;
            return;
        }
        const bool chanok = rowlane[u] + f_cb * 32 < g.Cin;
        if (FAST4) {
            const bool ok = chanok && f_rowok[0];
            bool done = false;
            if constexpr (decltype(edge_tag)::value) {
                const int e0 = cr[0].base + (rowlane[u] + f_cb * 32) * HW + f_ky * g.W + f_kx;
                if (ok && (e0 < 0 || e0 + 4 > g.xelems)) {      // a handful of lanes of the first / last tile
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        S.gv[u][e] = buf_load1s(x_rs, S.mk[e] ? vconst[u][0] + 4u * e : kBufOob, f_soff);
                    done = true;
                }
            }
            if (!done) {
                const f32x4 w = buf_load4s(x_rs, ok ? vconst[u][0] : kBufOob, f_soff);
#pragma unroll
                for (int e = 0; e < 4; ++e) S.gv[u][e] = w[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                S.gv[u][e] = buf_load1s(x_rs, (chanok && f_rowok[e]) ? vconst[u][e] : kBufOob, f_soff);
        }
    };
    auto commit_a = [&](const Stage& S, int buf, int u) {
        if (COCOS_CONV_ABLATE & 4) return;
        _Float16* ab = at + buf * 2 * APLANE;
        const int idx = u * 256 + tid, row = idx >> 2, kc = idx & 3;
        *reinterpret_cast<u32x4*>(ab + row * CV_AROW + kc * 8) = S.a[0][u];
        if (!ONE) *reinterpret_cast<u32x4*>(ab + APLANE + row * CV_AROW + kc * 8) = S.a[1][u];
    };
    auto commit_g = [&](Stage& S, int buf, int u) {
        if (COCOS_CONV_ABLATE & 4) return;
        _Float16* gb = gt + buf * 2 * GPLANE;
        if (FAST4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) S.gv[u][e] = S.mk[e] ? S.gv[u][e] : 0.f;
        }
        const int kk = u * RPP + rowsub, col = 4 * cg;
        if (ONE) {
            *reinterpret_cast<u32x2*>(gb + kk * GROW + col) =
                u32x2{cv_pack_bf16(S.gv[u][0], S.gv[u][1]), cv_pack_bf16(S.gv[u][2], S.gv[u][3])};
            return;
        }
        u32x2 hi, lo;
        cv_split4(S.gv[u], sx, hi, lo);
        *reinterpret_cast<u32x2*>(gb + kk * GROW + col) = hi;
        *reinterpret_cast<u32x2*>(gb + GPLANE + kk * GROW + col) = lo;
    };
    auto fetch_all = [&](Stage& S, int tt) {
        fetch_begin(S, tt);
#pragma unroll
        for (int u = 0; u < MI; ++u) fetch_a(S, u);
#pragma unroll
        for (int u = 0; u < NP; ++u) fetch_g(S, u, std::true_type{});
    };
    auto commit_all = [&](Stage& S, int buf) {
#pragma unroll
        for (int u = 0; u < MI; ++u) commit_a(S, buf, u);
#pragma unroll
        for (int u = 0; u < NP; ++u) commit_g(S, buf, u);
    };

    fetch_all(st[0], 0);
    commit_all(st[0], 0);
    fetch_all(st[0], 1);
    if (STAGES == 2) fetch_all(st[1], 2);
    __syncthreads();

    // transpose-read addressing of the gathered image (see header): 16-lane group (lane >> 4) = (column half nb, k
    // half kg); lane i of the group supplies row (i >> 2), columns 4 (i & 3) .. +3 of its 4 x 16 block
    const int li = lane & 15, nb = (lane >> 4) & 1, kg = lane >> 5;
    const int tr_off = (8 * kg + (li >> 2)) * GROW + 16 * nb + 4 * (li & 3);

    // MFMA operand fragments of one 16-wide k half-step: F[0] = k 0..15, F[1] = k 16..31 of the tile
    f16x8 fah[2][MI], fal[2][MI], fbh[2][NJ], fbl[2][NJ];
    auto read_frags = [&](int buf, int s) __attribute__((always_inline)) {
        const _Float16* ab = at + buf * 2 * APLANE + (wm * (BM / 2) + c) * CV_AROW + h * 8;
        const _Float16* gb = gt + buf * 2 * GPLANE + wn * (32 * NJ) + tr_off;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const _Float16* p = gb + s * 16 * GROW + j * 32;
            const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
            const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * GROW));
            fbh[s][j] = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
            if (!ONE) {
                const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + GPLANE));
                const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + GPLANE + 4 * GROW));
                fbl[s][j] = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            fah[s][i] = *reinterpret_cast<const f16x8*>(ab + i * 32 * CV_AROW + s * 16);
            if (!ONE) fal[s][i] = *reinterpret_cast<const f16x8*>(ab + APLANE + i * 32 * CV_AROW + s * 16);
        }
    };
    // One step = one k-block of 32, ONE barrier, in its middle.  The operand fragments run half a step ahead of the MFMAs:
    //   first half : MFMAs of k 0..15 (fragments read during the previous step) | reads the k 16..31 fragments of this
    //                tile | commits the staged tile t+1 to the other LDS buffer (its last readers passed the previous
    //                barrier with their fragments already in registers)
    //   barrier    : tile t+1 is visible; nobody reads tile t from LDS any more
    //   second half: MFMAs of k 16..31 | reads the k 0..15 fragments of tile t+1 | fetches tile t+1+STAGES from memory
    // so neither the LDS reads nor the barrier leave the matrix pipe idle (the earlier shape — barrier, read, multiply —
    // paid the LDS latency of all four waves at once after every barrier).
    auto step = [&](int t, Stage& S, auto edge_tag) __attribute__((always_inline)) {
        const int buf = t & 1;
        CPH_T(ts0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            CPH_T(tsa);
            if (!(COCOS_CONV_ABLATE & 8)) {
                if (s == 0) read_frags(buf, 1); else read_frags(buf ^ 1, 0);
            }
            if (s == 1) fetch_begin(S, t + 1 + STAGES);
            // program order: the fragment reads, then per row block its MFMAs followed by a slice of the staging work
            // (first half: commits, second half: fetches); the pipeline below then hands the non-MFMA instructions out in
            // that order, a few per MFMA.  (Measured: all loads first, then reads, then MFMAs: 0.499 instead of 0.442 ms.)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (!(COCOS_CONV_ABLATE & 8)) {
                    // term-major: two MFMAs on the same accumulator are never neighbours (an instruction issued between
                    // two dependent MFMAs costs a ~43-cycle bubble on gfx950; between independent ones ~6)
                    if (ONE) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = cv_mfma_bf16(fah[s][i], fbh[s][j], acc[i][j]);
                    } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbh[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbl[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[s][i], fbh[s][j], acc[i][j], 0, 0, 0);
                    }
                }
                if (s == 0) {
                    commit_a(S, buf ^ 1, i);
#pragma unroll
                    for (int q = 0; q < GPS; ++q) commit_g(S, buf ^ 1, i * GPS + q);
                } else {
                    fetch_a(S, i);
#pragma unroll
                    for (int q = 0; q < GPS; ++q) fetch_g(S, i * GPS + q, edge_tag);
                }
            }
            if (!(COCOS_CONV_ABLATE & 8)) {
                // one MFMA, then up to COCOS_CONV_SCHED_N instructions of any other kind (a wave that owns its SIMD issues
                // about one instruction per 4-5 cycles: ~7 fit beside a 32-cycle MFMA)
#pragma unroll
                for (int q = 0; q < (ONE ? 1 : 3) * MI * NJ; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x010 | 0x080, ONE ? 3 * COCOS_CONV_SCHED_N : COCOS_CONV_SCHED_N, 0);
                }
            }
            CPH_T(tsb);
            CPH_ADD(s, tsa, tsb);
            if (s == 0) {
                CPH_T(ts1);
                __syncthreads();
                CPH_T(ts2);
                CPH_ADD(2, ts1, ts2);
            }
        }
        CPH_T(ts3);
        CPH_ADD(3, ts0, ts3);
        CPH_ADD(4, 0, 1);
    };
    auto run = [&](auto edge_tag) __attribute__((always_inline)) {
        read_frags(0, 0);
        int t = 0;
        for (; t + 1 < nkb; t += 2) {
            step(t, st[0], edge_tag);
            step(t + 1, st[STAGES - 1], edge_tag);
        }
        if (t < nkb) step(t, st[0], edge_tag);
    };
    // two copies of the loop: the common one has no branch in its body (one scheduling region per step)
    if (edge_tile) run(std::true_type{}); else run(std::false_type{});

    const float oscale = ONE ? 1.0f : 1.0f / ((w_scale ? *w_scale : 1.0f) * sx);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * (32 * NJ) + j * 32 + c;
        if (n >= g.Ntot) continue;
        const int b = cv_div(n, g.mOHW);
        const int pos = n - b * g.OH * g.OWv;
        const int oy = cv_div(pos, g.mOW);
        const int ox = pos - oy * g.OWv;
        if (ox >= g.OW) continue;                      // virtual column
        float* yb = Y + ((size_t)b * M) * g.o_plane + g.o_off + oy * g.o_pitch + ox * g.o_cs;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 32 + acc_row_base(r) + 4 * h;
                if (m < M) yb[(size_t)m * g.o_plane] = acc[i][j][r] * oscale + (bias ? bias[m] : 0.f);
            }
    }
}

// --------------------------------------------------------------------------------------------------------------------
// weight gradient: partial[slice][co][k] over the slice's positions, k = ((ci/32) * T + tap) * 32 + ci%32
// --------------------------------------------------------------------------------------------------------------------
template <int BM, bool FAST4, bool ONE = false>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                            const float* __restrict__ x_amax,
                                                            const float* __restrict__ g_amax, float* __restrict__ part,
                                                            int M, ConvGeom g, int nchunk /* positions per slice, % 32 == 0 */,
                                                            int ybytes) {
    constexpr int MI = BM / 64;
    constexpr int APT = BM / 32;                       // dY pieces per thread (2 per slot)
    constexpr int GPS = 4 / MI;
    constexpr int APLANE = BM * CV_AROW, BPLANE = CV_BN * CV_AROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const at = reinterpret_cast<_Float16*>(smem_raw);   // dY image [2 buf][hi|lo][BM co][32 n + pad]
    _Float16* const bt = at + 2 * 2 * APLANE;                      // gathered image [2 buf][hi|lo][128 k][32 n + pad]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntk = (g.Ktot + CV_BN - 1) / CV_BN, ntm = (M + BM - 1) / BM;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int slice = vb / (ntk * ntm), rem = vb % (ntk * ntm);
    const int m0 = (rem / ntk) * BM, k0 = (rem % ntk) * CV_BN;
    const int nbeg = slice * nchunk, nend = min(g.Ntot, nbeg + nchunk);
    const int shift = g.pad * g.W + g.padx;

    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(X - shift, ((size_t)g.xelems + shift) * 4);
    const __amdgpu_buffer_rsrc_t y_rs = make_rsrc(dY, (size_t)ybytes);
    const float sx = ONE ? 1.0f : cv_scale_from_amax(x_amax), sg = ONE ? 1.0f : cv_scale_from_amax(g_amax);
    const int ohw = g.OH * g.OW;
    // a 16-byte piece can cross the ends of the whole tensor only for windows in the first rows of the first image /
    // the last rows of the last one (conservative, workgroup-uniform): those slices run the loop copy with the check
    const bool edge_slice = FAST4 && (nbeg < (g.pad + 2) * g.OW || nend > g.Ntot - (g.pad + g.KH * g.dil + 2) * g.OW);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A (dY): BM rows (co) x 32 positions = BM*8 float4: piece u -> row u*32 + tid/8, 4 positions (tid & 7)*4
    // B (gather): 128 rows (k) x 32 positions = 1024 float4, 4 per thread: row u*32 + tid/8, the same 4 positions
    const int q4 = 4 * (tid & 7);
    // the thread's four k rows: tap and channel fixed for the whole kernel
    int r_ky[4], r_kx[4], r_off[4];
    bool r_ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 32 + (tid >> 3);
        const int cb = cv_div(k >> 5, g.mT), tap = (k >> 5) - cb * (g.KH * g.KW), ci = cb * 32 + (k & 31);
        r_ky[u] = cv_div(tap, g.mKW);
        r_kx[u] = (tap - r_ky[u] * g.KW) * g.dil;     // the tap's offset in input pixels
        r_ky[u] *= g.dil;
        r_ok[u] = k < g.Ktot && ci < g.Cin;
        r_off[u] = (ci * g.H + r_ky[u]) * g.W + r_kx[u] + shift;
    }
    unsigned y_row[APT];                                // co * OH*OW * 4 or out of range
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        const int co = m0 + u * 32 + (tid >> 3);
        y_row[u] = co < M ? (unsigned)(co * ohw) * 4u : kBufOob;
    }

    struct Stage { float a[APT][4]; float gv[4][4]; int xs[4]; };
    Stage st[COCOS_CONV_WGRAD_PIPE ? 1 : 2];
    constexpr int NC = FAST4 ? 1 : 4;
    Corner f_cr[NC];
    bool f_live[NC];
    unsigned f_y[NC];                                   // byte offset of dY[b, 0, pos]
    auto fetch_begin = [&](int np0) {
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            const int n = min(np0 + q4 + e, g.Ntot - 1);
            f_cr[e] = cv_corner(n, g);
            f_live[e] = np0 + q4 + e < nend;
            const int b = cv_div(n, g.mOHW);
            f_y[e] = (unsigned)(b * M * ohw + (n - b * ohw)) * 4u;
        }
    };
    auto fetch_g = [&](Stage& S, int u, auto edge_tag) __attribute__((always_inline)) {
        if (FAST4) {
            const bool ok = r_ok[u] && f_live[0] && (unsigned)(f_cr[0].iy0 + r_ky[u]) < (unsigned)g.H;
            const int e0s = f_cr[0].base + r_off[u];                    // element offset in the shifted descriptor
            S.xs[u] = f_cr[0].ix0 + r_kx[u];
            bool done = false;
            if constexpr (decltype(edge_tag)::value) {
                if (ok && (e0s - shift < 0 || e0s - shift + 4 > g.xelems)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        S.gv[u][e] = buf_load1(x_rs, (unsigned)(S.xs[u] + e) < (unsigned)g.W ? (unsigned)(e0s + e) * 4u : kBufOob);
                    done = true;
                }
            }
            if (!done) {
                const f32x4 w = buf_load4(x_rs, ok ? (unsigned)e0s * 4u : kBufOob);
#pragma unroll
                for (int e = 0; e < 4; ++e) S.gv[u][e] = w[e];
            }
        } else {
            S.xs[u] = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = r_ok[u] && f_live[e] && (unsigned)(f_cr[e].iy0 + r_ky[u]) < (unsigned)g.H &&
                                (unsigned)(f_cr[e].ix0 + r_kx[u]) < (unsigned)g.W;
                S.gv[u][e] = buf_load1(x_rs, ok ? (unsigned)(f_cr[e].base + r_off[u]) * 4u : kBufOob);
            }
        }
    };
    auto fetch_a = [&](Stage& S, int u) {
        if (FAST4) {      // OH*OW % 4 == 0: the four positions are 16 contiguous, aligned bytes of one image
            const f32x4 w = buf_load4(y_rs, (f_live[0] && y_row[u] != kBufOob) ? f_y[0] + y_row[u] : kBufOob);
#pragma unroll
            for (int e = 0; e < 4; ++e) S.a[u][e] = w[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                S.a[u][e] = buf_load1(y_rs, (f_live[e] && y_row[u] != kBufOob) ? f_y[e] + y_row[u] : kBufOob);
        }
    };
    auto commit_a = [&](Stage& S, int buf, int u) {
        _Float16* ab = at + buf * 2 * APLANE;
        const int row = u * 32 + (tid >> 3);
        if (ONE) {
            *reinterpret_cast<u32x2*>(ab + row * CV_AROW + q4) =
                u32x2{cv_pack_bf16(S.a[u][0], S.a[u][1]), cv_pack_bf16(S.a[u][2], S.a[u][3])};
            return;
        }
        u32x2 hi, lo;
        cv_split4(S.a[u], sg, hi, lo);
        *reinterpret_cast<u32x2*>(ab + row * CV_AROW + q4) = hi;
        *reinterpret_cast<u32x2*>(ab + APLANE + row * CV_AROW + q4) = lo;
    };
    auto commit_g = [&](Stage& S, int buf, int u) {
        _Float16* bb = bt + buf * 2 * BPLANE;
        if (FAST4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) S.gv[u][e] = (unsigned)(S.xs[u] + e) < (unsigned)g.W ? S.gv[u][e] : 0.f;
        }
        const int row = u * 32 + (tid >> 3);
        if (ONE) {
            *reinterpret_cast<u32x2*>(bb + row * CV_AROW + q4) =
                u32x2{cv_pack_bf16(S.gv[u][0], S.gv[u][1]), cv_pack_bf16(S.gv[u][2], S.gv[u][3])};
            return;
        }
        u32x2 hi, lo;
        cv_split4(S.gv[u], sx, hi, lo);
        *reinterpret_cast<u32x2*>(bb + row * CV_AROW + q4) = hi;
        *reinterpret_cast<u32x2*>(bb + BPLANE + row * CV_AROW + q4) = lo;
    };
    auto fetch_all = [&](Stage& S, int np0) {
        fetch_begin(np0);
#pragma unroll
        for (int u = 0; u < APT; ++u) fetch_a(S, u);
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch_g(S, u, std::true_type{});
    };
    auto commit_all = [&](Stage& S, int buf) {
#pragma unroll
        for (int u = 0; u < APT; ++u) commit_a(S, buf, u);
#pragma unroll
        for (int u = 0; u < 4; ++u) commit_g(S, buf, u);
    };

    const int nsteps = (max(nend - nbeg, 0) + CV_BK - 1) / CV_BK;
    constexpr int WSTAGES = COCOS_CONV_WGRAD_PIPE ? 1 : 2;
    // positions at or beyond nend are masked (f_live), so prefetches past the last step are harmless
    fetch_all(st[0], nbeg);
    commit_all(st[0], 0);
    fetch_all(st[0], nbeg + CV_BK);
    if (WSTAGES == 2) fetch_all(st[1], nbeg + 2 * CV_BK);
    __syncthreads();

    f16x8 fah[2][MI], fal[2][MI], fbh[2][2], fbl[2][2];
    auto read_frags = [&](int buf, int s) __attribute__((always_inline)) {
        const _Float16* ab = at + buf * 2 * APLANE + (wm * (BM / 2) + c) * CV_AROW + h * 8;
        const _Float16* bb = bt + buf * 2 * BPLANE + (wn * 64 + c) * CV_AROW + h * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fbh[s][j] = *reinterpret_cast<const f16x8*>(bb + j * 32 * CV_AROW + s * 16);
            if (!ONE) fbl[s][j] = *reinterpret_cast<const f16x8*>(bb + BPLANE + j * 32 * CV_AROW + s * 16);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            fah[s][i] = *reinterpret_cast<const f16x8*>(ab + i * 32 * CV_AROW + s * 16);
            if (!ONE) fal[s][i] = *reinterpret_cast<const f16x8*>(ab + APLANE + i * 32 * CV_AROW + s * 16);
        }
    };
    // same step shape as the forward kernel: one barrier in the middle, fragments read half a step ahead, the staged
    // tile committed under the first half and the next one fetched under the second
    auto step = [&](int t, Stage& S, auto edge_tag) __attribute__((always_inline)) {
        const int buf = t & 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (COCOS_CONV_WGRAD_PIPE) {
                if (s == 0) read_frags(buf, 1); else read_frags(buf ^ 1, 0);
            } else {
                read_frags(buf, s);
            }
            if (s == 1) fetch_begin(nbeg + (t + 1 + WSTAGES) * CV_BK);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (ONE) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = cv_mfma_bf16(fah[s][i], fbh[s][j], acc[i][j]);
                } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbh[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbl[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[s][i], fbh[s][j], acc[i][j], 0, 0, 0);
                }
                if (s == 0) {
                    commit_a(S, buf ^ 1, 2 * i);
                    commit_a(S, buf ^ 1, 2 * i + 1);
#pragma unroll
                    for (int q = 0; q < GPS; ++q) commit_g(S, buf ^ 1, i * GPS + q);
                } else {
                    fetch_a(S, 2 * i);
                    fetch_a(S, 2 * i + 1);
#pragma unroll
                    for (int q = 0; q < GPS; ++q) fetch_g(S, i * GPS + q, edge_tag);
                }
            }
#pragma unroll
            for (int q = 0; q < (ONE ? 2 : 6) * MI; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x010 | 0x080, ONE ? 3 * COCOS_CONV_SCHED_N : COCOS_CONV_SCHED_N, 0);
            }
            if (COCOS_CONV_WGRAD_PIPE ? s == 0 : s == 1) __syncthreads();
        }
    };
    auto run = [&](auto edge_tag) __attribute__((always_inline)) {
        if (COCOS_CONV_WGRAD_PIPE) read_frags(0, 0);
        int t = 0;
        for (; t + 1 < nsteps; t += 2) {
            step(t, st[0], edge_tag);
            step(t + 1, st[WSTAGES - 1], edge_tag);
        }
        if (t < nsteps) step(t, st[0], edge_tag);
    };
    if (edge_slice) run(std::true_type{}); else run(std::false_type{});

    const float oscale = ONE ? 1.0f : 1.0f / (sx * sg);
    float* pb = part + (size_t)slice * M * g.Ktot;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wn * 64 + j * 32 + c;
        if (k >= g.Ktot) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 32 + acc_row_base(r) + 4 * h;
                if (m < M) pb[(size_t)m * g.Ktot + k] = acc[i][j][r] * oscale;
            }
    }
}

static Magic cv_magic(int d) {
    unsigned sh = 0;
    while ((2u << sh) <= (unsigned)d) ++sh;                 // floor(log2 d)
    if ((1u << sh) == (unsigned)d) return Magic{0u, sh, 1u};
    const unsigned long long two = 1ull << (32 + sh);
    return Magic{(unsigned)((two + (unsigned)d - 1) / (unsigned)d), sh, 0u};
}

static int cv_geom(ConvGeom& g, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int dil, bool virt,
                   const char* who) {
    COCOS_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0 && dil >= 1,
                  COCOS_ERR_INVALID, "%s: bad geometry B=%d Cin=%d Cout=%d H=%d W=%d k=%dx%d stride=%d pad=%d dilation=%d", who, B,
                  Cin, Cout, H, W, KH, KW, stride, pad, dil);
    COCOS_REQUIRE(H + 2 * pad >= dil * (KH - 1) + 1 && W + 2 * pad >= dil * (KW - 1) + 1, COCOS_ERR_INVALID,
                  "%s: kernel larger than the padded input", who);
    g.Cin = Cin; g.H = H; g.W = W; g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.padx = pad; g.dil = dil;
    g.OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    g.OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    g.o_plane = g.OH * g.OW; g.o_pitch = g.OW; g.o_cs = 1; g.o_off = 0;
    g.Cp = (Cin + 31) / 32 * 32;
    g.OWv = virt ? (g.OW + 3) / 4 * 4 : g.OW;
    const long long ktot = (long long)KH * KW * g.Cp, ntot = (long long)B * g.OH * g.OWv;
    // 32-bit byte offsets everywhere (buffer descriptors): both tensors and the weight planes below 2 GiB
    COCOS_REQUIRE(ntot + 256 < (1ll << 30) && ktot + 256 < (1ll << 30), COCOS_ERR_UNSUPPORTED,
                  "%s: problem too large for the 32-bit index arithmetic (positions %lld, K %lld)", who, ntot, ktot);
    COCOS_REQUIRE(((long long)B * Cin * H * W + (long long)(g.Cp + 32) * H * W + (long long)pad * (W + 1)) * 4 < 0x7fffffffll &&
                      (long long)B * Cout * g.OH * g.OW * 4 < 0x7fffffffll && (long long)Cout * ktot * 4 < 0x7fffffffll,
                  COCOS_ERR_UNSUPPORTED, "%s: a tensor exceeds 2 GiB", who);
    g.Ktot = (int)ktot;
    g.Ntot = (int)ntot;
    g.xelems = B * Cin * H * W;
    g.mT = cv_magic(KH * KW); g.mKW = cv_magic(KW); g.mOHW = cv_magic(g.OH * g.OWv); g.mOW = cv_magic(g.OWv);
    return COCOS_OK;
}

}  // namespace cocos

extern "C" int cocos_conv2d_out_size(int in, int k, int stride, int pad, int dil) {
    return (stride >= 1 && dil >= 1 && in + 2 * pad >= dil * (k - 1) + 1) ? (in + 2 * pad - dil * (k - 1) - 1) / stride + 1 : 0;
}
extern "C" int cocos_conv2d_kdim(int Cin, int KH, int KW) { return KH * KW * ((Cin + 31) / 32 * 32); }

static int conv_fwd_launch(const cocos::ConvGeom& g, const float* x, const void* w_hi, const void* w_lo,
                           const float* w_scale_dev, const float* x_amax_dev, const float* bias, float* y, int Cout,
                           cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(aligned16(w_hi) && (!w_lo || aligned16(w_lo)), COCOS_ERR_INVALID,
                  "conv2d_fwd_f16x3: weight planes must be 16-byte aligned");
    const bool one = w_lo == nullptr;            // single bf16 plane: the one-term flavour
    const bool fast4 = g.stride == 1;
    // tile: BM = 256 rows for wide layers, BN = 128 positions.  The 256 x 256 tile (wave tile 128 x 128: each LDS operand
    // is re-read half as often per MFMA, all 256 accumulator registers in use, one register stage) exists and is
    // tested, but measured no faster on the 407-channel block (0.466 vs 0.470 ms) and slower on its input gradient
    // (fewer, longer workgroups): COCOS_CONV_BN=256 selects it for experiments.
    const char* force_bm = getenv("COCOS_CONV_BM");
    const int bm = (force_bm && atoi(force_bm) == 128) ? 128 : (Cout > 128 ? 256 : 128);
    const long long mt = (Cout + bm - 1) / bm;
    const char* force = getenv("COCOS_CONV_BN");
    const int bn = (force && atoi(force) == 256 && bm == 256) ? 256 : 128;
    const long long blocks = mt * ((g.Ntot + bn - 1) / bn);
    COCOS_REQUIRE(blocks <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "conv2d_fwd_f16x3: grid too large");
    hipStream_t s = as_stream(stream);
#define COCOS_GO(BMv, BNv, F4) do { if (one && BNv == 128) COCOS_GO1(BMv, 128, F4, true); else COCOS_GO1(BMv, BNv, F4, false); } while (0)
#define COCOS_GO1(BMv, BNv, F4, ONEv)                                                                              \
    do {                                                                                                           \
        auto kern = conv_fwd_kernel<BMv, BNv, F4, ONEv>;                                                           \
        const size_t smem = (size_t)2 * 2 * (BMv * CV_AROW + CV_BK * (BNv + 32)) * sizeof(_Float16);               \
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));              \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, s, x, static_cast<const _Float16*>(w_hi), \
                           static_cast<const _Float16*>(w_lo), w_scale_dev, x_amax_dev, bias, y, Cout, g);         \
    } while (0)
    if (bm == 256 && bn == 256 && !one) { if (fast4) COCOS_GO(256, 256, true); else COCOS_GO(256, 256, false); }
    else if (bm == 256)         { if (fast4) COCOS_GO(256, 128, true); else COCOS_GO(256, 128, false); }
    else                        { if (fast4) COCOS_GO(128, 128, true); else COCOS_GO(128, 128, false); }
#undef COCOS_GO
#undef COCOS_GO1
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_conv2d_fwd_f16x3(const float* x, const void* w_hi, const void* w_lo, const float* w_scale_dev,
                                      const float* x_amax_dev, const float* bias, float* y, int B, int Cin, int H, int W,
                                      int Cout, int KH, int KW, int stride, int pad, int dil, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && w_hi && y, COCOS_ERR_INVALID, "conv2d_fwd_f16x3: null pointer");
    ConvGeom g;
    if (int rc = cv_geom(g, B, Cin, H, W, Cout, KH, KW, stride, pad, dil, stride == 1, "conv2d_fwd_f16x3")) return rc;
    return conv_fwd_launch(g, x, w_hi, w_lo, w_scale_dev, x_amax_dev, bias, y, Cout, stream);
}

// One parity class of the input gradient of a STRIDED convolution (see cocos_hip.h): a stride-1 convolution of x
// ( = dy of the strided layer) with a JH x JW sub-kernel, separate row / column padding, an explicit output grid
// OHo x OWo (reads beyond x are zero) and a strided placement of the result inside y ( = dx of the strided layer).
extern "C" int cocos_conv2d_fwd_scatter_f16x3(const float* x, const void* w_hi, const void* w_lo, const float* w_scale_dev,
                                              const float* x_amax_dev, float* y, int B, int Cin, int H, int W, int Cout,
                                              int JH, int JW, int pad_y, int pad_x, int OHo, int OWo, long long y_plane,
                                              int y_pitch, int y_col_stride, long long y_offset, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && w_hi && y, COCOS_ERR_INVALID, "conv2d_fwd_scatter_f16x3: null pointer");
    COCOS_REQUIRE(pad_y >= 0 && pad_x >= 0 && OHo >= 1 && OWo >= 1 && y_plane >= 1 && y_pitch >= 1 && y_col_stride >= 1 &&
                      y_offset >= 0, COCOS_ERR_INVALID, "conv2d_fwd_scatter_f16x3: bad placement");
    ConvGeom g;
    // geometry of a stride-1 layer with the larger padding (bounds / 32-bit checks), then the explicit grid and placement
    const int pmax = pad_y > pad_x ? pad_y : pad_x;
    if (int rc = cv_geom(g, B, Cin, H + 0, W + 0, Cout, JH, JW, 1, pmax + (JH > JW ? JH : JW), 1, true, "conv2d_fwd_scatter_f16x3"))
        return rc;
    g.pad = pad_y; g.padx = pad_x; g.OH = OHo; g.OW = OWo; g.OWv = (OWo + 3) / 4 * 4;
    const long long ntot = (long long)B * OHo * g.OWv;
    COCOS_REQUIRE(ntot + 256 < (1ll << 30) && y_plane * (long long)B * Cout * 4 < 0x7fffffffll && y_plane < (1ll << 30) &&
                      y_offset + (long long)(OHo - 1) * y_pitch + (long long)(OWo - 1) * y_col_stride < y_plane,
                  COCOS_ERR_UNSUPPORTED, "conv2d_fwd_scatter_f16x3: output placement out of range");
    g.Ntot = (int)ntot;
    g.mOHW = cv_magic(OHo * g.OWv); g.mOW = cv_magic(g.OWv);
    g.o_plane = (int)y_plane; g.o_pitch = y_pitch; g.o_cs = y_col_stride; g.o_off = (int)y_offset;
    return conv_fwd_launch(g, x, w_hi, w_lo, w_scale_dev, x_amax_dev, nullptr, y, Cout, stream);
}

extern "C" int cocos_conv2d_wgrad_slices(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                                         int dil) {
    const int OH = cocos_conv2d_out_size(H, KH, stride, pad, dil), OW = cocos_conv2d_out_size(W, KW, stride, pad, dil);
    if (B < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1 || pad < 0 || OH < 1 || OW < 1) return 0;
    const long long ntot = (long long)B * OH * OW;
    const int bm = Cout > 128 ? 256 : 128;
    const long long tiles = (long long)((Cout + bm - 1) / bm) * ((cocos_conv2d_kdim(Cin, KH, KW) + 127) / 128);
    long long s = (1024 + tiles - 1) / tiles;             // ~4 workgroups per CU over the launch
    const long long maxs = (ntot + 255) / 256;            // at least 8 k-blocks of 32 positions per slice
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    return (int)s;
}

static int conv_wgrad_impl(const float* x, const float* dy, const float* x_amax_dev, const float* g_amax_dev, float* partials,
                           int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int dil, bool one,
                           cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && dy && partials, COCOS_ERR_INVALID, "conv2d_wgrad_f16x3: null pointer");
    ConvGeom g;
    if (int rc = cv_geom(g, B, Cin, H, W, Cout, KH, KW, stride, pad, dil, false, "conv2d_wgrad_f16x3")) return rc;
    const int nslices = cocos_conv2d_wgrad_slices(B, Cin, H, W, Cout, KH, KW, stride, pad, dil);
    const int nchunk = ((g.Ntot + nslices - 1) / nslices + CV_BK - 1) / CV_BK * CV_BK;
    const bool fast4 = stride == 1 && g.OW % 4 == 0 && aligned16(dy);
    const int bm = Cout > 128 ? 256 : 128;
    const long long blocks = (long long)nslices * ((Cout + bm - 1) / bm) * ((g.Ktot + CV_BN - 1) / CV_BN);
    COCOS_REQUIRE(blocks <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "conv2d_wgrad_f16x3: grid too large");
    const int ybytes = (int)((long long)B * Cout * g.OH * g.OW * 4);
    hipStream_t s = as_stream(stream);
#define COCOS_GO(BMv, F4) do { if (one) COCOS_GO1(BMv, F4, true); else COCOS_GO1(BMv, F4, false); } while (0)
#define COCOS_GO1(BMv, F4, ONEv)                                                                                   \
    do {                                                                                                           \
        auto kern = conv_wgrad_kernel<BMv, F4, ONEv>;                                                              \
        const size_t smem = (size_t)2 * 2 * (BMv + CV_BN) * CV_AROW * sizeof(_Float16);                            \
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));              \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, s, x, dy, x_amax_dev, g_amax_dev, partials, \
                           Cout, g, nchunk, ybytes);                                                               \
    } while (0)
    if (bm == 256) { if (fast4) COCOS_GO(256, true); else COCOS_GO(256, false); }
    else           { if (fast4) COCOS_GO(128, true); else COCOS_GO(128, false); }
#undef COCOS_GO
#undef COCOS_GO1
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_conv2d_wgrad_f16x3(const float* x, const float* dy, const float* x_amax_dev, const float* g_amax_dev,
                                        float* partials, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride,
                                        int pad, int dil, cocos_stream_t stream) {
    return conv_wgrad_impl(x, dy, x_amax_dev, g_amax_dev, partials, B, Cin, H, W, Cout, KH, KW, stride, pad, dil, false, stream);
}

// The one-term bf16 flavour of the weight gradient (no scales: bf16 has fp32's range).
extern "C" int cocos_conv2d_wgrad_bf16(const float* x, const float* dy, float* partials, int B, int Cin, int H, int W, int Cout,
                                       int KH, int KW, int stride, int pad, int dil, cocos_stream_t stream) {
    return conv_wgrad_impl(x, dy, nullptr, nullptr, partials, B, Cin, H, W, Cout, KH, KW, stride, pad, dil, true, stream);
}

#ifdef COCOS_DEBUG_TIMING
extern "C" int cocos_debug_read_timing_conv(long long* host8, int reset) {
    using namespace cocos;
    COCOS_HIP_CHECK(hipDeviceSynchronize());
    COCOS_HIP_CHECK(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_phase_conv), 8 * sizeof(long long)));
    if (reset) {
        long long z[8] = {0};
        COCOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_conv), z, sizeof(z)));
    }
    return COCOS_OK;
}
#endif

// --------------------------------------------------------------------------------------------------------------------
// host-side glue as two kernels (instead of ~8 framework launches per convolution and step)
// --------------------------------------------------------------------------------------------------------------------
namespace cocos {

// weight [Cout][Cin][KH][KW] fp32 -> the f16 hi/lo planes K16 reads ([K/32][M][32], k = ((c/32)*T + tap)*32 + c%32):
//   mode 0 (forward):         M = Cout rows, c = ci,  tap = (ky, kx) of the JH x JW = KH x KW kernel
//   mode 1 (input gradient):  M = Cin rows,  c = co,  tap (ky', kx') of the flipped sub-kernel:
//                             source tap (ry + s*(JH-1-ky'), rx + s*(JW-1-kx'))   (stride 1: ry = rx = 0, s = 1, J = K)
// scaled by the power of two from *amax_dev (written to *scale_out), zero for c beyond the channel count.
__global__ __launch_bounds__(256) void conv_weight_planes_kernel(const float* __restrict__ w, _Float16* __restrict__ hi,
                                                                 _Float16* __restrict__ lo, int Cout, int Cin, int KH, int KW,
                                                                 int mode, int JH, int JW, int ry, int rx, int s,
                                                                 const float* __restrict__ amax_dev, float* __restrict__ scale_out,
                                                                 int M, int C, int nkb) {
    const bool bf = (mode & 2) != 0;                                      // one bf16 plane (hi), no scale
    mode &= 1;
    const float scale = bf ? 1.0f : cv_scale_from_amax(amax_dev);
    const unsigned total2 = (unsigned)nkb * (unsigned)M * 16u;            // pairs of consecutive c: one 32-bit store per plane
    const int T = JH * JW;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total2; i += gridDim.x * 256u) {
        if (i == 0 && scale_out) *scale_out = scale;
        const int j = (int)(i & 15u) * 2;
        const unsigned r = i >> 4;
        const int m = (int)(r % (unsigned)M), kb = (int)(r / (unsigned)M);
        const int cb = kb / T, tap = kb - cb * T;
        const int kyp = tap / JW, kxp = tap - kyp * JW;
        const int ky = mode == 0 ? kyp : ry + s * (JH - 1 - kyp), kx = mode == 0 ? kxp : rx + s * (JW - 1 - kxp);
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = cb * 32 + j + e;
            const int co = mode == 0 ? m : c, ci = mode == 0 ? c : m;
            v[e] = c < C ? w[(((size_t)co * Cin + ci) * KH + ky) * KW + kx] * scale : 0.f;
        }
        if (bf) {
            reinterpret_cast<unsigned*>(hi)[i] = cv_pack_bf16(v[0], v[1]);
            continue;
        }
        unsigned h2, l2;
        split_pair_rn(v[0], v[1], h2, l2);
        reinterpret_cast<unsigned*>(hi)[i] = h2;
        reinterpret_cast<unsigned*>(lo)[i] = l2;
    }
}

// partial [S][Cout][K] (k as above) -> dw [Cout][Cin][KH][KW]: the sum over the S position slices and the re-ordering of
// k.  A workgroup owns 64 consecutive elements of the partials' own order (256-byte coalesced pieces) and splits the S
// slices over its four waves (a small layer has ~200 slices of 37 K elements: one thread per element walking all of them
// was 36 us of dependent loads on 144 workgroups); the four wave sums meet in LDS, the 4-byte writes scatter into dw.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S,
                                                                int Cout, int Cin, int T, int K) {
    __shared__ float red[4][64];
    const unsigned total = (unsigned)Cout * (unsigned)K;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (unsigned base = blockIdx.x * 64u; base < total; base += gridDim.x * 64u) {
        const unsigned i = base + lane;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (i < total) {
            const float* p = part + i;
            int sl = grp;
            for (; sl + 12 < S; sl += 16) {
                a0 += p[(size_t)sl * total];
                a1 += p[(size_t)(sl + 4) * total];
                a2 += p[(size_t)(sl + 8) * total];
                a3 += p[(size_t)(sl + 12) * total];
            }
            for (; sl < S; sl += 4) a0 += p[(size_t)sl * total];
        }
        red[grp][lane] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (grp == 0 && i < total) {
            const int co = (int)(i / (unsigned)K), k = (int)(i - (unsigned)co * (unsigned)K);
            const int kb = k >> 5, cb = kb / T, tap = kb - cb * T, ci = cb * 32 + (k & 31);
            if (ci < Cin) dw[((size_t)co * Cin + ci) * T + tap] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        }
        __syncthreads();
    }
}

}  // namespace cocos

extern "C" int cocos_conv2d_weight_planes(const float* w, void* hi, void* lo, int Cout, int Cin, int KH, int KW, int mode,
                                          int JH, int JW, int ry, int rx, int s, const float* amax_dev, float* scale_out_dev,
                                          cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(w && hi && (lo || (mode & 2)), COCOS_ERR_INVALID, "conv2d_weight_planes: null pointer");
    const int mode_full = mode;
    mode &= 1;
    COCOS_REQUIRE(mode_full >= 0 && mode_full <= 3, COCOS_ERR_INVALID, "conv2d_weight_planes: mode %d", mode_full);
    COCOS_REQUIRE(Cout >= 1 && Cin >= 1 && KH >= 1 && KW >= 1 && JH >= 1 && JW >= 1 && s >= 1 &&
                      ry >= 0 && rx >= 0 && ry + s * (JH - 1) < KH && rx + s * (JW - 1) < KW &&
                      (mode == 1 || (JH == KH && JW == KW && s == 1 && ry == 0 && rx == 0)),
                  COCOS_ERR_INVALID, "conv2d_weight_planes: bad arguments (k=%dx%d mode=%d J=%dx%d r=(%d,%d) s=%d)", KH, KW, mode,
                  JH, JW, ry, rx, s);
    const int M = mode == 0 ? Cout : Cin, C = mode == 0 ? Cin : Cout;
    const int nkb = JH * JW * ((C + 31) / 32);
    const size_t total = (size_t)nkb * M * 16;
    COCOS_REQUIRE(total < 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "conv2d_weight_planes: weight too large");
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(conv_weight_planes_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, as_stream(stream),
                       w, static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), Cout, Cin, KH, KW, mode_full, JH, JW, ry, rx, s,
                       amax_dev, scale_out_dev, M, C, nkb);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_conv2d_wgrad_reduce(const float* partials, float* dw, int S, int Cout, int Cin, int KH, int KW,
                                         cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(partials && dw, COCOS_ERR_INVALID, "conv2d_wgrad_reduce: null pointer");
    COCOS_REQUIRE(S >= 1 && Cout >= 1 && Cin >= 1 && KH >= 1 && KW >= 1, COCOS_ERR_INVALID, "conv2d_wgrad_reduce: bad dims");
    const size_t total = (size_t)Cout * cocos_conv2d_kdim(Cin, KH, KW);
    COCOS_REQUIRE(total < 0x7fffffffull, COCOS_ERR_UNSUPPORTED, "conv2d_wgrad_reduce: weight too large");
    const size_t blocks = (total + 63) / 64;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, as_stream(stream),
                       partials, dw, S, Cout, Cin, KH * KW, cocos_conv2d_kdim(Cin, KH, KW));
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
