// K16b: convolution (any stride; input gradient for stride 1) in ONE bf16 MFMA term on operands that are bf16 IN MEMORY — gfx950.
//
// The one-term flavour of conv_f16x3.hip (`COCOS_CONV=bf16`) halved the matrix work of K16 and then sat on the CU's vector
// memory path: its activations are fp32 NCHW in HBM, every k-block of 32 channels x 128 positions is gathered once PER TAP
// (9 x for a 3x3 kernel), converted in registers and committed to LDS by the wave that also issues the MFMAs
// (conv shapes of the feature producers, B = 8: 370-560 TFLOP/s, 15-22 % of the dense bf16 peak).  Here the producer-side
// work is done ONCE per tensor by a small streaming kernel and the GEMM's inner loop touches no VALU at all:
//
//   prep   x fp32 [B][C][H][W]  ->  xp bf16 [B][H+2p][W+2p][Cp]   (NHWC, channels zero-padded to Cp = 32 ceil(C/32), the
//          border written as zeros or mirrored: the GEMM never masks, never converts, a reflect-padded layer needs no
//          separate padding pass)                                     [conv_nhwc_prep_kernel]
//   GEMM   Y[co][n] = bias[co] + sum_k Wt[co][k] Xp[n][k],  k = ((ci/32) T + tap) 32 + ci%32 (the k order of K16: one k-block =
//          32 consecutive channels at one tap = 64 contiguous bytes of xp per position, 64 contiguous bytes of the weight
//          planes [K/32][Cout][32] per row).  Both tiles go HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`,
//          16 bytes per lane, no registers, no LDS write instructions), four stages deep, ONE raw s_barrier per k-step
//          and counted vmcnt waits (the CDNA guide's "3-buffer span" recipe); DMA pieces placed between the MFMAs.  An LDS-DMA writes lane-linear (wave base +
//          lane x 16), so the bank swizzle of the 64-byte rows is applied to the SOURCE chunk each lane fetches:
//          physical chunk = logical chunk ^ ((row >> 2) & 3), conflict-free `ds_read_b128` fragments.
//          Tile 256 | 128 (Cout) x 256 | 128 (positions) x 32; 256 x 256 with 8 waves as 2 x 4 (two per SIMD, wave tile 128 x 64),
//          the others with 4 waves as 2 x 2.                                     [conv_nhwc_bf16_kernel]
//   The input gradient of a stride-1 layer is the same GEMM on prep(dy, pad = d(K-1)-p, zeros) with the flipped,
//   transposed weight planes (cocos_conv2d_weight_planes mode 1|2).
//   The weight gradient contracts over positions, which are the STRIDED index of both NHWC operands: its fragments come out
//   of the same lane-linear LDS images through ds_read_b64_tr_b16.                [conv_nhwc_wgrad_bf16_kernel]
//
// Reference lines served: every stride-1 nn.Conv2d of the feature producers — ResidualBlock correspondence.py:13-36,
// adaptor layers :150-173, SPADE mlp convolutions normalization.py:118-127 — under `COCOS_CONV=bf16` (the precision of the
// reference's --amp run).  Layers with fewer than 128 output or 32 input channels, the input gradient of strided layers and weight
// gradients whose output rows are not whole k-steps of 32 positions stay on conv_f16x3.hip's kernels.
// Measured and dropped: the bias gradient taken inside the prep pass over dy (wave reductions + per-workgroup partials: the prep
// kernels of a module step +1.5 ms for 1.1 ms of cocos_channel_sum saved; with atomics instead — every CU on the same 13 cache
// lines — 3 -> 26 ms).
#include "common.h"

namespace cocos {

typedef __bf16 nb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 nb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float nb_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* nb_lds_ptr;
typedef short s16x4_nb __attribute__((ext_vector_type(4)));
typedef _Float16 nb_f16x8 __attribute__((ext_vector_type(8)));

constexpr int NB_BM = 256, NB_BK = 32, NB_STAGES = 4;     // 4 stages: a tile is requested 3 k-steps (~1.4 us) before it is read
constexpr int NB_ROW = NB_BK * 2;                       // bytes per LDS row (one position / one weight row, 32 k)

// s_waitcnt vmcnt(N) needs an immediate: the counted waits of the pipelines below pick among the few values that occur
#define NB_WAIT_BARRIER(ahead, per, lgkm)                                                                                  \
    do {                                                                                                                   \
        if ((ahead) >= 3) asm volatile("s_waitcnt vmcnt(%0)" lgkm "\n\ts_barrier" ::"n"(3 * (per)) : "memory");            \
        else if ((ahead) == 2) asm volatile("s_waitcnt vmcnt(%0)" lgkm "\n\ts_barrier" ::"n"(2 * (per)) : "memory");       \
        else if ((ahead) == 1) asm volatile("s_waitcnt vmcnt(%0)" lgkm "\n\ts_barrier" ::"n"(per) : "memory");             \
        else asm volatile("s_waitcnt vmcnt(0)" lgkm "\n\ts_barrier" ::: "memory");                                         \
    } while (0)

struct NhwcGeom {
    int B, Cp, Hp, Wp;        // padded NHWC input
    int OH, OW, OHW, Ntot;    // output grid, positions in all
    int Cout, T, KW, dil;     // rows of the weight planes, taps, kernel width, dilation
    int stride;
    int ncb, nsteps;          // channel blocks of 32, k-steps = ncb * T
    int ntiles;               // row tiles x column tiles of the launch's tile shape
    unsigned xp_bytes, w_bytes;
};

__device__ __forceinline__ unsigned nb_pack_bf16(float a, float b) {      // round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(nb_f32x2{a, b}, nb_bf16x2));
}

__device__ __forceinline__ int nb_reflect(int i, int n) {                  // torch's ReflectionPad2d index (pad < n)
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// ---------------------------------------------------------------------------------------------------------------------
// prep: fp32 NCHW -> bf16 NHWC with a border.  A workgroup owns 64 consecutive positions of one padded image (rows
// flattened: a 66-wide row does not strand 62 lanes) and walks the channels 64 at a time: wave g loads channels 16 g .. +15
// of the chunk for its lane's position (each load 256 bytes coalesced across the wave), the packed pairs cross through LDS
// ([position][64 channels], 144-byte rows: conflict-free 16-byte writes), and go out as 128 contiguous bytes per position.
// ---------------------------------------------------------------------------------------------------------------------
// SPLIT (K16c, the fp32-accurate flavour): x * s as f16 hi (round to nearest: common.h split_pair_rn) + f16 lo, s the power of two from *amax
// (cv's convention: |x s| < 2^10); two planes, the lo plane `plane_bytes` behind the hi plane.
__device__ __forceinline__ float nb_scale_from_amax(const float* amax) {
    if (!amax) return 1.0f;
    const float a = *amax;
    if (!(a > 0.f) || !(a < INFINITY)) return 1.0f;
    int e;
    frexpf(a, &e);
    return ldexpf(1.0f, 10 - e);
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void conv_nhwc_prep_kernel(const float* __restrict__ x, unsigned char* __restrict__ xp, int B, int C,
                                                             int H, int W, int Hp, int Wp, int Cp, int pad, int reflect,
                                                             const float* __restrict__ amax, size_t plane_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[(SPLIT ? 2 : 1) * 64 * 144];
    const float sc = SPLIT ? nb_scale_from_amax(amax) : 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npos = Hp * Wp, chunks = (npos + 63) >> 6;
    const int b = blockIdx.x / chunks, q0 = (blockIdx.x - b * chunks) * 64;
    const int q = q0 + lane;
    const int yp = q / Wp, xq = q - yp * Wp;
    int ys = yp - pad, xs = xq - pad;
    bool inside = q < npos;
    if (reflect) {
        ys = nb_reflect(ys, H);
        xs = nb_reflect(xs, W);
    } else {
        inside = inside && ys >= 0 && ys < H && xs >= 0 && xs < W;
    }
    const size_t plane = (size_t)H * W;
    const float* src = x + (size_t)b * C * plane + (inside ? (size_t)ys * W + xs : 0);
    // store side: position (tid >> 3) + 32 pass, 16-byte piece tid & 7 of the chunk's 128 bytes
    const int sp = threadIdx.x >> 3, so = threadIdx.x & 7;
    for (int c0 = 0; c0 < Cp; c0 += 64) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + wave * 16 + i;
            v[i] = (inside && c < C) ? src[(size_t)c * plane] : 0.f;
        }
        u32x4 o0, o1, l0, l1;
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned h, l;
                split_pair_rn(v[2 * i] * sc, v[2 * i + 1] * sc, h, l);
                o0[i] = h; l0[i] = l;
                split_pair_rn(v[8 + 2 * i] * sc, v[9 + 2 * i] * sc, h, l);
                o1[i] = h; l1[i] = l;
            }
        } else {
            o0 = u32x4{nb_pack_bf16(v[0], v[1]), nb_pack_bf16(v[2], v[3]), nb_pack_bf16(v[4], v[5]), nb_pack_bf16(v[6], v[7])};
            o1 = u32x4{nb_pack_bf16(v[8], v[9]), nb_pack_bf16(v[10], v[11]), nb_pack_bf16(v[12], v[13]), nb_pack_bf16(v[14], v[15])};
        }
        if (c0) __syncthreads();                           // the previous chunk has been read out
        *reinterpret_cast<u32x4*>(tile + lane * 144 + wave * 32) = o0;
        *reinterpret_cast<u32x4*>(tile + lane * 144 + wave * 32 + 16) = o1;
        if (SPLIT) {
            *reinterpret_cast<u32x4*>(tile + 64 * 144 + lane * 144 + wave * 32) = l0;
            *reinterpret_cast<u32x4*>(tile + 64 * 144 + lane * 144 + wave * 32 + 16) = l1;
        }
        __syncthreads();
        if (c0 + so * 8 < Cp) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int pl = sp + 32 * pass;
                if (q0 + pl < npos) {
                    unsigned char* dst = xp + (((size_t)b * npos + q0 + pl) * Cp + c0 + so * 8) * 2;
                    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(tile + pl * 144 + so * 16);
                    if (SPLIT)
                        *reinterpret_cast<u32x4*>(dst + plane_bytes) = *reinterpret_cast<const u32x4*>(tile + 64 * 144 + pl * 144 + so * 16);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMM (forward / input gradient)
// ---------------------------------------------------------------------------------------------------------------------
// k-step t = cb * T + tap reads weight rows at byte t * Cout * 64 and activations at ((ky*dil*Wp + kx*dil) * Cp + cb*32) * 2
// beyond the position's corner: both travel in the scalar offset of the buffer instruction, the per-lane offsets are loop
// constants.  Rows beyond Cout / positions beyond Ntot are clamped to the last valid one (computed, never stored).
//
// step t:   fragments k 16..31 of stage t  |  16 | 8 MFMAs (k 0..15)  |  vmcnt: stage t+1 landed; barrier: every wave is done
//           reading stage t  |  LDS-DMA of stage t+NB_STAGES into stage t's buffer  |  fragments k 0..15 of stage t+1  |  MFMAs (k 16..31)
//
// SK ("stream-K"): a layer whose tile count is a little above a multiple of the CU count (the input gradient of the
// reflect-padded 64x64 layers: 8 x 66 x 66 positions = 136.1 tiles of 256, x 2 row tiles = 274 workgroups on 256 CUs)
// pays a whole second round for 18 tiles.  With SK the grid is ONE workgroup per CU and workgroup w takes the k-step
// units [w U, (w+1) U) of the (tile, k-step) sequence: U >= one tile's steps, so a tile is cut at most once — its head
// (k-steps 0..) is the LAST thing workgroup w does, its tail the FIRST thing workgroup w+1 does.  The early finisher
// parks its partial tile in a workspace slot and raises a flag; the late one adds the parked tile to its own and stores.
// No atomics on the output (same-address-class fp32 atomics ran at ~140 G/s in the K2 experiments: 0.1 ms for this tensor).
// Round 6 measured the alternative schedule — whole tiles in k lock-step first (stream-K proper has every workgroup at another k offset:
// the 6 MB of weight planes are never hot in a 4 MB L2 at the slice a workgroup needs, PMC 1.36 GB read per launch against 97 MB for the
// plain launch), then the remaining tiles cut into W / rem k-chunks whose LAST arriver adds the other partials: correct (same tests), and
// SLOWER — 407 -> 407 forward + backward 1.133 -> 1.232 ms, 512 -> 512 1.39 -> 1.465: the last arriver streams 13 parked tiles (3.3 MB)
// through one CU.  Stream-K proper stays.
// TERMS = 3 (K16c): every operand is TWO f16 planes (hi, lo), every product three MFMAs (hi hi + hi lo + lo hi) — the arithmetic of
// conv_f16x3.hip on this kernel's data path; a stage holds four tiles (two stages of 64 KB at 256 x 256), the result is scaled
// back by 1 / (s_x s_w) in the epilogue.  The stage count follows: TERMS = 1 four stages, TERMS = 3 two.
struct NhwcSplit {
    float* ring;                 // fold mode (input gradient of a ReflectionPad2d(1) layer): the border ring, see the epilogue
    const void* w_lo;            // lo plane of the weights (same layout as the hi plane)
    unsigned xp_plane_bytes;     // the activations' lo plane lies this far behind their hi plane
    const float* x_amax;         // device cells: max|x| the activation planes were scaled by, the weight planes' scale
    const float* w_scale;
};

template <int BM, int BN, bool SK, int NW = 4, int TERMS = 1>
__global__ __launch_bounds__(NW * 64, 1) void conv_nhwc_bf16_kernel(const void* __restrict__ xp, const void* __restrict__ wpl,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                float* __restrict__ ws, int* __restrict__ flags, const NhwcGeom g,
                                                                const int units_per_wg, const NhwcSplit sp) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char nb_smem[];
    constexpr int NP = TERMS == 3 ? 2 : 1, NST = TERMS == 3 ? 2 : NB_STAGES;
    constexpr int A_PLANE = BM * NB_ROW, B_PLANE = BN * NB_ROW;
    constexpr int A_BYTES = NP * A_PLANE, B_BYTES = NP * B_PLANE, ST_BYTES = A_BYTES + B_BYTES;
    // NW waves as 2 (rows) x NW/2 (columns).  NW = 8: two waves per SIMD, 128 accumulator registers each — while one issues its DMA
    // pieces / fragment reads (~60 cycles per piece even between MFMAs) the other keeps the matrix pipe busy
    constexpr int WNW = NW / 2, NT = NW * 64;
    constexpr int NI = BM / 64, NJ = BN / (32 * WNW);   // 32x32 MFMA tiles of a wave: BM/2 rows x BN/WNW columns
    constexpr int NAI = BM / (16 * NW);                 // weight DMA instructions per wave and stage (16 rows each)
    constexpr int NBI = BN / (16 * NW);                 // activation DMA instructions per wave and stage (16 positions each)
    static_assert(NAI >= 1 && NBI >= 1 && NJ >= 1, "tile too small for the wave count");
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w / WNW, wn = w % WNW;
    const int ntm = (g.Cout + BM - 1) / BM;
    const int nsteps = g.nsteps, T = g.T;

    // (no arrays of __amdgpu_buffer_rsrc_t: hipcc then silently drops the kernel's HOST stub — the library fails to load)
    const __amdgpu_buffer_rsrc_t rA0 = make_rsrc(wpl, g.w_bytes), rB0 = make_rsrc(xp, g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rA1 = make_rsrc(NP == 2 ? sp.w_lo : wpl, g.w_bytes);
    const __amdgpu_buffer_rsrc_t rB1 = make_rsrc(static_cast<const unsigned char*>(xp) + (NP == 2 ? sp.xp_plane_bytes : 0u), g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, bias ? (size_t)g.Cout * 4 : 0);      // no bias: every read returns 0
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);   // the 16-byte piece of its row this lane fetches (source-side swizzle)
    const unsigned stepA = (unsigned)g.Cout * NB_ROW;
    // fragment addresses: row (lane & 31) of a 32-row MFMA tile, logical chunk kk*2 + (lane >> 5), swizzled by the row
    const int frow = lane & 31, fsw = (lane >> 2) & 3, fh = lane >> 5;
    const int fo0 = frow * NB_ROW + ((0 + fh) ^ fsw) * 16, fo1 = frow * NB_ROW + ((2 + fh) ^ fsw) * 16;
    const int fbaseA = wm * ((BM / 2) * NB_ROW), fbaseB = A_BYTES + wn * ((BN / WNW) * NB_ROW);

    int u0 = SK ? (int)blockIdx.x * units_per_wg : (int)blockIdx.x * nsteps;
    const int u1 = SK ? min(u0 + units_per_wg, g.ntiles * nsteps) : u0 + nsteps;
    if (SK && u0 >= u1) return;
#pragma unroll 1
    do {
        // (one tile per workgroup: consecutive tiles — the row tiles of one position tile, then its neighbour — go to ONE XCD's L2)
        const int tile = SK ? u0 / nsteps : xcd_remap((int)blockIdx.x, (int)gridDim.x);
        const int s0 = SK ? u0 - tile * nsteps : 0, s1 = SK ? min(nsteps, s0 + (u1 - u0)) : nsteps;
        const int mt = tile % ntm, nt = tile / ntm;
        const int m0 = mt * BM, n0 = nt * BN;
        unsigned voffA[NAI], voffB[NBI];
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            const int row = min(m0 + w * (BM / NW) + i * 16 + (lane >> 2), g.Cout - 1);
            voffA[i] = (unsigned)row * NB_ROW + chunk * 16;
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int n = min(n0 + w * (BN / NW) + i * 16 + (lane >> 2), g.Ntot - 1);
            const int b = n / g.OHW, rem = n - b * g.OHW, oy = rem / g.OW, ox = rem - oy * g.OW;
            voffB[i] = (unsigned)(((b * g.Hp + oy * g.stride) * g.Wp + ox * g.stride) * g.Cp) * 2u + chunk * 16;
        }
        auto issue = [&](int cb, int tap, int t, int buf) {
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
            const unsigned sA = (unsigned)t * stepA;
            const unsigned sB = (unsigned)(((ky * g.Wp + kx) * g.dil) * g.Cp + cb * 32) * 2u;
            unsigned char* st = nb_smem + buf * ST_BYTES;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < NAI; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? rA1 : rA0, (nb_lds_ptr)(st + p * A_PLANE + w * (NAI * 1024) + i * 1024), 16,
                                                             (int)voffA[i], (int)sA, 0, 0);
#pragma unroll
                for (int i = 0; i < NBI; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? rB1 : rB0, (nb_lds_ptr)(st + A_BYTES + p * B_PLANE + w * (NBI * 1024) + i * 1024),
                                                             16, (int)voffB[i], (int)sB, 0, 0);
            }
        };

        f32x16 acc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        u32x4 fa[2][NP][NI], fb[2][NP][NJ];
        auto frags = [&](int buf, int kk) {
            const unsigned char* st = nb_smem + buf * ST_BYTES + (kk ? fo1 : fo0);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[kk][p][i] = *reinterpret_cast<const u32x4*>(st + p * A_PLANE + fbaseA + i * (32 * NB_ROW));
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[kk][p][j] = *reinterpret_cast<const u32x4*>(st + p * B_PLANE + fbaseB + j * (32 * NB_ROW));
            }
        };
        auto mfmas = [&](int kk) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (TERMS == 1) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nb_bf16x8, fa[kk][0][i]),
                                                                            __builtin_bit_cast(nb_bf16x8, fb[kk][0][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const nb_f16x8 ah = __builtin_bit_cast(nb_f16x8, fa[kk][0][i]), al = __builtin_bit_cast(nb_f16x8, fa[kk][NP - 1][i]);
                        const nb_f16x8 bh = __builtin_bit_cast(nb_f16x8, fb[kk][0][j]), bl = __builtin_bit_cast(nb_f16x8, fb[kk][NP - 1][j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                    }
                }
        };

        // prologue: the first NB_STAGES k-steps in flight; the first landed and visible
        int it = s0, icb = s0 / T, itap = s0 - icb * T;      // the next k-step to issue
        const int nseg = s1 - s0;
#pragma unroll 1
        for (int u = 0; u < NST && it < s1; ++u, ++it) {
            issue(icb, itap, it, u);
            if (++itap == T) { itap = 0; ++icb; }
        }
        NB_WAIT_BARRIER(it - s0 - 1, NP * (NAI + NBI), "");   // all but the first of the issued stages may still be in flight
        frags(0, 0);

        int buf = 0, t = s0;
        // steady state (a stage to issue in every step): branch-free, and the DMA pieces / next fragments are handed out BETWEEN the
        // MFMAs of the second half — issued as a block after the barrier they cost 100-185 cycles each with the matrix pipe idle
        // (MI355X guide, "LDS-DMA piece issue cost"): that, not latency, was 1000 of a step's 2100 cycles
#pragma unroll 1
        for (; t + NST < s1; ++t) {
            frags(buf, 1);
            mfmas(0);
#pragma unroll
            for (int q = 0; q < (NI * NJ) / 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS, 0);
                if (q < NI + NJ) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
            }
            const int nbuf = buf == NST - 1 ? 0 : buf + 1;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NST - 2) * NP * (NAI + NBI)) : "memory");
            issue(icb, itap, it, buf);
            ++it;
            if (++itap == T) { itap = 0; ++icb; }
            frags(nbuf, 0);
            mfmas(1);
#pragma unroll
            for (int q = 0; q < (NI * NJ) / 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS, 0);
                if (q < NAI + NBI) __builtin_amdgcn_sched_group_barrier(0x010, NP, 0);
                if (q < NI + NJ) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
            }
            buf = nbuf;
        }
#pragma unroll 1
        for (; t < s1; ++t) {
            frags(buf, 1);
            mfmas(0);
            const int nbuf = buf == NST - 1 ? 0 : buf + 1;
            if (t + 1 < s1) {
                NB_WAIT_BARRIER(it - (t + 2), NP * (NAI + NBI), " lgkmcnt(0)");     // stage t+1 landed; the stages issued beyond it stay in flight
                if (it < s1) {
                    issue(icb, itap, it, buf);
                    ++it;
                    if (++itap == T) { itap = 0; ++icb; }
                }
                frags(nbuf, 0);
            }
            mfmas(1);
            buf = nbuf;
        }

        // epilogue: accumulator (i, j, r) is row m0 + wm*BM/2 + 32 i + (r&3) + 8 (r>>2) + 4 (lane>>5), position n0 + wn*BN/2 + 32 j + (lane&31)
        if (SK) {
            // the accumulators' home is the accumulator file, across the three-way epilogue too (hipcc otherwise spills all 256 of them
            // to scratch after the k-loop and reloads them in every branch: 0.5 MB per workgroup and segment)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
        }
        if (SK && s0 > 0) {
            // the tail of a cut tile (always this workgroup's first segment): park it, raise the flag
            // (buffer stores: ONE per-lane offset and a scalar offset per register — with flat stores the compiler made 256 addresses and
            //  spilled the accumulators around them: 1312 bytes of scratch per lane, ~0.4 GB of scratch traffic per launch by PMC)
            const __amdgpu_buffer_rsrc_t rs_slot = make_rsrc(ws + (size_t)blockIdx.x * (BM * BN), (size_t)BM * BN * 4);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        buf_store1s(rs_slot, acc[i][j][r], (unsigned)tid * 4u, (unsigned)(((i * NJ + j) * 16 + r) * NT) * 4u);
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + blockIdx.x, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // the head of a cut tile (always the last segment): the next workgroup parked the tail long ago.  Its partial is ADDED IN THE STORE
            // LOOP below (a buffer load per register through a descriptor that is empty for an uncut tile: zeros, no memory access) — as a
            // separate `acc += partial` pass in front of the stores it made hipcc move all 256 accumulators through scratch in every
            // segment (1312 bytes per lane; PMC: 438 MB read per launch for 56 MB of operands)
            const bool cut_head = SK && s1 < nsteps;
            const int partner = (int)blockIdx.x + 1;
            if (cut_head) {
                if (tid == 0) {
                    while (__hip_atomic_load(flags + partner, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
                }
                __syncthreads();
                (void)__hip_atomic_load(flags + partner, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);    // every wave acquires
            }
            const __amdgpu_buffer_rsrc_t rs_part = make_rsrc(ws + (cut_head ? (size_t)partner * (BM * BN) : (size_t)0), cut_head ? (size_t)BM * BN * 4 : (size_t)0);
            const int corow = m0 + wm * (BM / 2) + 4 * (lane >> 5);
            const float oscale = TERMS == 3 ? 1.0f / (nb_scale_from_amax(sp.x_amax) * (sp.w_scale ? *sp.w_scale : 1.0f)) : 1.0f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float bv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[r] = buf_load1(rbias, (unsigned)(corow + i * 32 + (r & 3) + 8 * (r >> 2)) * 4u);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = n0 + wn * (BN / WNW) + j * 32 + (lane & 31);
                    if (n >= g.Ntot) continue;
                    const int b = n / g.OHW, rem = n - b * g.OHW;
                    float* yp = y + (size_t)b * g.Cout * g.OHW + rem;
                    size_t cs = (size_t)g.OHW;               // elements between two output channels of this position
                    if (sp.ring) {
                        // the (H + 2) x (W + 2) gradient of a mirrored input: interior pixels go straight into dx [H][W], the border
                        // ring (top row, bottom row, left column, right column) into `ring`; conv_nhwc_fold_ring_kernel adds it back
                        const int oy = rem / g.OW, ox = rem - oy * g.OW, fH = g.OH - 2, fW = g.OW - 2, RL = 2 * g.OW + 2 * fH;
                        if (oy >= 1 && oy <= fH && ox >= 1 && ox <= fW) {
                            cs = (size_t)fH * fW;
                            yp = y + (size_t)b * g.Cout * cs + (size_t)(oy - 1) * fW + (ox - 1);
                        } else {
                            cs = (size_t)RL;
                            const int ri = oy == 0 ? ox : oy == fH + 1 ? g.OW + ox : ox == 0 ? 2 * g.OW + (oy - 1) : 2 * g.OW + fH + (oy - 1);
                            yp = sp.ring + (size_t)b * g.Cout * cs + ri;
                        }
                    }
                    float pv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        pv[r] = SK ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_part, (int)((unsigned)tid * 4u),
                                                                                                    (int)((unsigned)(((i * NJ + j) * 16 + r) * NT) * 4u), 2))
                                   : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = corow + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (co < g.Cout) yp[(size_t)co * cs] = (SK ? acc[i][j][r] + pv[r] : acc[i][j][r]) * oscale + bv[r];
                    }
                }
            }
            if (cut_head) {
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + partner, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
        }
        u0 += nseg;
        if (SK && u0 < u1) __syncthreads();      // every wave is done with the LDS stages before the next segment refills them
    } while (SK && u0 < u1);
}

// dx [H][W] += the mirrored border of the padded gradient (nn.ReflectionPad2d(1) backward): padded row 0 folds onto row 1, row H+1
// onto row H-2, likewise the columns (corners onto (1,1), (1,W-2), ...).  One thread per TARGET pixel (rows 1 / H-2, columns
// 1 / W-2: 2 W + 2 (H - 2) of them per plane) gathers its up to three ring elements: no atomics.  ring = [top W+2 | bottom W+2 |
// left H | right H] per plane, written by the GEMM epilogue above.
__global__ __launch_bounds__(256) void conv_nhwc_fold_ring_kernel(const float* __restrict__ ring, float* __restrict__ dx, int H, int W) {
    const int OW = W + 2, RL = 2 * OW + 2 * H, NTG = 2 * W + 2 * (H - 2);
    const float* rg = ring + (size_t)blockIdx.x * RL;
    float* o = dx + (size_t)blockIdx.x * H * W;
    for (int t = threadIdx.x; t < NTG; t += 256) {
        int y, x;
        if (t < W) { y = 1; x = t; }
        else if (t < 2 * W) { y = H - 2; x = t - W; }
        else {                                             // the two columns, rows other than 1 and H-2
            const int u = t - 2 * W, k = u >> 1;
            y = k == 0 ? 0 : k + 1;                        // 0, 2, 3, ..., H-3, H-1  (skipping 1 and H-2)
            if (y >= H - 2) y += 1;
            x = (u & 1) ? W - 2 : 1;
        }
        float acc = 0.f;
        // rows: padded row 0 -> y == 1, padded row H+1 -> y == H-2; the ring rows hold padded columns 0 .. W+1
        const float* rows[2] = {rg, rg + OW};
        const bool hit[2] = {y == 1, y == H - 2};
#pragma unroll
        for (int a = 0; a < 2; ++a)
            if (hit[a]) {
                acc += rows[a][x + 1];
                if (x == 1) acc += rows[a][0];
                if (x == W - 2) acc += rows[a][W + 1];
            }
        // columns: padded column 0 -> x == 1, padded column W+1 -> x == W-2; the ring columns hold padded rows 1 .. H
        if (x == 1) acc += rg[2 * OW + y];
        if (x == W - 2) acc += rg[2 * OW + H + y];
        o[(size_t)y * W + x] += acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient:  partial[s][co][k'] = sum over the slice's positions n of dY[n][co] Xp[n + tap][ci],  k' = ((ci/32) T + tap) 32 + ci%32
// ---------------------------------------------------------------------------------------------------------------------
// Both operands are NHWC: the contraction index (positions) is the strided one, so the MFMA fragments (8 consecutive
// positions of one channel) are transposing reads, ds_read_b64_tr_b16, of images the DMA lays out as
//     [n >> 2][column / 16][n & 3][16 columns]      (n = position inside the 32-position k-step, 128-byte 4 x 16 blocks)
// — a 16-lane group's four 32-byte row pieces are one contiguous 128 bytes, two groups 256: conflict-free.  One DMA
// instruction fills 8 such blocks (4 positions x 128 channels): per position its 16 lanes fetch 256 contiguous bytes.
// Requires OW % 32 == 0 (a k-step of 32 positions lies in one output row: its address is a scalar base + lane constants).
// dyp = dy as bf16 NHWC [B][OH + 2q][OW + 2q][Cop] (the input-gradient GEMM's operand, border q, or q = 0).
struct NhwcWgradSplit {
    unsigned xp_plane_bytes, dy_plane_bytes;     // lo planes behind the hi planes
    const float* x_amax;                         // device cells the two operands were scaled by
    const float* g_amax;
};

struct NhwcWgradGeom {
    int B, Cp, Hp, Wp;          // xp (the forward's operand)
    int Cop, Hq, Wq, q;         // dyp
    int OH, OW, Ntot;
    int Cout, T, KW, dil, nkb;  // nkb = T * Cp / 32 blocks of k'
    int stride;
    int steps_total, steps_per_slice;
    unsigned xp_bytes, dy_bytes;
};

template <int TERMS>
__global__ __launch_bounds__(256, 1) void conv_nhwc_wgrad_bf16_kernel(const void* __restrict__ xp, const void* __restrict__ dyp,
                                                                      float* __restrict__ partial, const NhwcWgradGeom g,
                                                                      const NhwcWgradSplit sp) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char nb_smem[];
    constexpr int NP = TERMS == 3 ? 2 : 1, NST = TERMS == 3 ? 2 : NB_STAGES;
    constexpr int PLANE = 256 * NB_ROW;                                  // 32 positions x 256 columns of one plane
    constexpr int A_BYTES = NP * PLANE, ST_BYTES = 2 * A_BYTES;          // both operands
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int ntm = (g.Cout + 255) / 256, ntn = (g.nkb * 32 + 255) / 256;
    // slice-major virtual ids, consecutive ids on one XCD: the workgroups that read one position slice share an L2 (PMC before:
    // 304 MB of HBM reads per launch for 58 MB of operands — every XCD fetched its own copy of every slice)
    const int vid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int mt = vid % ntm, ntile = (vid / ntm) % ntn, slice = vid / (ntm * ntn);
    const int m0 = mt * 256, k0 = ntile * 256;
    const int t_begin = slice * g.steps_per_slice, nsteps = min(g.steps_per_slice, g.steps_total - t_begin);

    const __amdgpu_buffer_rsrc_t rA0 = make_rsrc(dyp, g.dy_bytes), rB0 = make_rsrc(xp, g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rA1 = make_rsrc(static_cast<const unsigned char*>(dyp) + (NP == 2 ? sp.dy_plane_bytes : 0u), g.dy_bytes);
    const __amdgpu_buffer_rsrc_t rB1 = make_rsrc(static_cast<const unsigned char*>(xp) + (NP == 2 ? sp.xp_plane_bytes : 0u), g.xp_bytes);
    // DMA instruction id = w*4 + i: positions 4 (id >> 1) + ((lane >> 1) & 3), columns ((id & 1) * 8 + (lane >> 3)) * 16 + (lane & 1) * 8 .. +7
    unsigned voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = w * 4 + i, nl = 4 * (id >> 1) + ((lane >> 1) & 3), col = (((id & 1) * 8 + (lane >> 3)) * 2 + (lane & 1)) * 8;
        const int co = min(m0 + col, g.Cop - 8);
        voffA[i] = (unsigned)(nl * g.Cop + co) * 2u;
        const int kb = min((k0 + col) >> 5, g.nkb - 1), cb = kb / g.T, tap = kb - cb * g.T, ky = tap / g.KW, kx = tap - ky * g.KW;
        voffB[i] = (unsigned)(((ky * g.Wp + kx) * g.dil + nl * g.stride) * g.Cp + cb * 32 + (col & 31)) * 2u;
    }
    // scalar position of the next k-step to issue: image b, output row oy, first column ox
    int it = 0, ib, ioy, iox;
    {
        const int n = t_begin * 32, ohw = g.OH * g.OW;
        ib = n / ohw;
        const int rem = n - ib * ohw;
        ioy = rem / g.OW;
        iox = rem - ioy * g.OW;
    }
    auto issue = [&](int buf) {
        const unsigned sA = (unsigned)(((ib * g.Hq + ioy + g.q) * g.Wq + iox + g.q) * g.Cop) * 2u;
        const unsigned sB = (unsigned)(((ib * g.Hp + ioy * g.stride) * g.Wp + iox * g.stride) * g.Cp) * 2u;
        unsigned char* st = nb_smem + buf * ST_BYTES;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? rA1 : rA0, (nb_lds_ptr)(st + p * PLANE + (w * 4 + i) * 1024), 16, (int)voffA[i], (int)sA,
                                                         0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(p ? rB1 : rB0, (nb_lds_ptr)(st + A_BYTES + p * PLANE + (w * 4 + i) * 1024), 16, (int)voffB[i],
                                                         (int)sB, 0, 0);
        }
        ++it;
        iox += 32;
        if (iox >= g.OW) { iox = 0; if (++ioy == g.OH) { ioy = 0; ++ib; } }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing fragment reads: 16-lane group (nb = column half, kg = position half), lane li -> row li >> 2, columns 4 (li & 3)..+3
    const int li = lane & 15, nb = (lane >> 4) & 1, kg = lane >> 5;
    const int tr_off = (kg * 32 + nb) * 128 + (li >> 2) * 32 + (li & 3) * 8;
    const int fbaseA = wm * (8 * 128) + tr_off, fbaseB = A_BYTES + wn * (8 * 128) + tr_off;     // 128 columns = 8 blocks of 16

    u32x4 fa[2][NP][4], fb[2][NP][4];
    auto tr8 = [&](const unsigned char* p) {
        const s16x4_nb v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_nb __attribute__((address_space(3)))*)(p));
        const s16x4_nb v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_nb __attribute__((address_space(3)))*)(p + 2048));
        return __builtin_bit_cast(u32x4, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto frags = [&](int buf, int kk) {
        const unsigned char* st = nb_smem + buf * ST_BYTES + kk * 8192;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][p][i] = tr8(st + p * PLANE + fbaseA + i * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[kk][p][j] = tr8(st + p * PLANE + fbaseB + j * 256);
        }
    };
    auto mfmas = [&](int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (TERMS == 1) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nb_bf16x8, fa[kk][0][i]),
                                                                        __builtin_bit_cast(nb_bf16x8, fb[kk][0][j]), acc[i][j], 0, 0, 0);
                } else {
                    const nb_f16x8 ah = __builtin_bit_cast(nb_f16x8, fa[kk][0][i]), al = __builtin_bit_cast(nb_f16x8, fa[kk][NP - 1][i]);
                    const nb_f16x8 bh = __builtin_bit_cast(nb_f16x8, fb[kk][0][j]), bl = __builtin_bit_cast(nb_f16x8, fb[kk][NP - 1][j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                }
            }
    };

#pragma unroll 1
    for (int u = 0; u < NST && u < nsteps; ++u) issue(u);
    NB_WAIT_BARRIER(it - 1, 8 * NP, "");
    if (nsteps > 0) frags(0, 0);

    int buf = 0, t = 0;
    // steady state: DMA pieces and the next fragments between the MFMAs (see conv_nhwc_bf16_kernel)
#pragma unroll 1
    for (; t + NST < nsteps; ++t) {
        frags(buf, 1);
        mfmas(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 0);
        }
        const int nbuf = buf == NST - 1 ? 0 : buf + 1;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NST - 2) * 8 * NP) : "memory");
        issue(buf);
        frags(nbuf, 0);
        mfmas(1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 0);
        }
        buf = nbuf;
    }
#pragma unroll 1
    for (; t < nsteps; ++t) {
        frags(buf, 1);
        mfmas(0);
        const int nbuf = buf == NST - 1 ? 0 : buf + 1;
        if (t + 1 < nsteps) {
            NB_WAIT_BARRIER(it - (t + 2), 8 * NP, " lgkmcnt(0)");
            if (it < nsteps) issue(buf);
            frags(nbuf, 0);
        }
        mfmas(1);
        buf = nbuf;
    }

    const int Ktot = g.nkb * 32;
    const float oscale = TERMS == 3 ? 1.0f / (nb_scale_from_amax(sp.x_amax) * nb_scale_from_amax(sp.g_amax)) : 1.0f;
    float* out = partial + (size_t)slice * g.Cout * Ktot;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + wn * 128 + j * 32 + (lane & 31);
        if (k >= Ktot) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.Cout) out[(size_t)co * Ktot + k] = acc[i][j][r] * oscale;
            }
        }
    }
}

}  // namespace cocos

static int cocos_nhwc_prep_impl(const float* x, void* xp, const float* amax_dev, bool split, int B, int C, int H, int W, int pad,
                          int reflect, cocos_stream_t stream, const char* who) {
    using namespace cocos;
    COCOS_REQUIRE(x && xp, COCOS_ERR_INVALID, "%s: null pointer", who);
    COCOS_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && (!reflect || (pad < H && pad < W)), COCOS_ERR_INVALID,
                  "%s: bad arguments (B=%d C=%d H=%d W=%d pad=%d reflect=%d)", who, B, C, H, W, pad, reflect);
    const int Hp = H + 2 * pad, Wp = W + 2 * pad, Cp = (C + 31) / 32 * 32;
    const long long plane = (long long)B * Hp * Wp * Cp * 2;
    COCOS_REQUIRE(plane * (split ? 2 : 1) < 0x7fffffffLL && aligned16(xp), COCOS_ERR_UNSUPPORTED, "%s: tensor too large / xp unaligned", who);
    const long long blocks = (long long)B * (((long long)Hp * Wp + 63) / 64);
    COCOS_REQUIRE(blocks <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "%s: grid too large", who);
    if (split)
        hipLaunchKernelGGL(conv_nhwc_prep_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x,
                           static_cast<unsigned char*>(xp), B, C, H, W, Hp, Wp, Cp, pad, reflect ? 1 : 0, amax_dev, (size_t)plane);
    else
        hipLaunchKernelGGL(conv_nhwc_prep_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x,
                           static_cast<unsigned char*>(xp), B, C, H, W, Hp, Wp, Cp, pad, reflect ? 1 : 0, nullptr, (size_t)0);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_conv2d_nhwc_prep_bf16(const float* x, void* xp, int B, int C, int H, int W, int pad, int reflect,
                                           cocos_stream_t stream) {
    return cocos_nhwc_prep_impl(x, xp, nullptr, false, B, C, H, W, pad, reflect, stream, "conv2d_nhwc_prep_bf16");
}

// K16c: xp = TWO f16 planes [2][B][Hp][Wp][Cp] (hi, lo) of x * 2^k, k from *amax_dev (max|x|, nullable = 1.0)
extern "C" int cocos_conv2d_nhwc_prep_f16x3(const float* x, void* xp, const float* amax_dev, int B, int C, int H, int W, int pad,
                                            int reflect, cocos_stream_t stream) {
    return cocos_nhwc_prep_impl(x, xp, amax_dev, true, B, C, H, W, pad, reflect, stream, "conv2d_nhwc_prep_f16x3");
}

extern "C" int cocos_conv2d_nhwc_bf16_supported(int Cin, int Cout, int KH, int KW, int stride) {
    return (stride >= 1 && stride <= 4 && Cout >= 128 && Cin >= 32 && KH >= 1 && KW >= 1 && KH * KW <= 49) ? 1 : 0;
}

namespace {
constexpr long long kNhwcFlagBytes = 4096;                                  // 1024 flags (one per workgroup of an SK launch)
constexpr long long kNhwcSlotBytes = 256LL * 256 * 4;                        // one parked tile
int nhwc_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}
}  // namespace

// Bytes of the ZERO-INITIALISED scratch cocos_conv2d_nhwc_bf16 wants for its stream-K launches (flags + one parked tile per
// CU); the flags are left zero by every launch, so the caller clears the buffer once, when it allocates it, and hands the same
// buffer to every call ON ONE STREAM (calls on different streams need their own).
extern "C" long long cocos_conv2d_nhwc_bf16_workspace_bytes(void) {
    const int cus = nhwc_cu_count();
    return kNhwcFlagBytes + (long long)(cus < 1024 ? cus : 1024) * kNhwcSlotBytes;
}

static int cocos_nhwc_gemm_impl(const void* xp, const void* w_hi, const void* w_lo, const float* w_scale_dev, const float* x_amax_dev,
                          const float* bias, float* y, float* fold_ring, void* workspace, long long workspace_bytes, int B, int Cp,
                          int Hp, int Wp, int Cout, int KH, int KW, int dil, int stride, cocos_stream_t stream, const char* who) {
    using namespace cocos;
    const bool split = w_lo != nullptr;
    COCOS_REQUIRE(xp && w_hi && y, COCOS_ERR_INVALID, "%s: null pointer", who);
    COCOS_REQUIRE(B >= 1 && Cp >= 32 && Cp % 32 == 0 && Cout >= 1 && KH >= 1 && KW >= 1 && dil >= 1 && stride >= 1, COCOS_ERR_INVALID,
                  "%s: bad arguments (B=%d Cp=%d Cout=%d k=%dx%d dil=%d stride=%d)", who, B, Cp, Cout, KH, KW, dil, stride);
    NhwcGeom g;
    g.B = B; g.Cp = Cp; g.Hp = Hp; g.Wp = Wp; g.stride = stride;
    COCOS_REQUIRE(Hp > dil * (KH - 1) && Wp > dil * (KW - 1), COCOS_ERR_INVALID, "%s: kernel larger than the padded input", who);
    g.OH = (Hp - dil * (KH - 1) - 1) / stride + 1; g.OW = (Wp - dil * (KW - 1) - 1) / stride + 1;
    g.OHW = g.OH * g.OW;
    const long long ntot = (long long)B * g.OHW, xbytes = (long long)B * Hp * Wp * Cp * 2;
    g.Cout = Cout; g.T = KH * KW; g.KW = KW; g.dil = dil; g.ncb = Cp / 32; g.nsteps = g.ncb * g.T;
    const long long wbytes = (long long)g.nsteps * Cout * NB_ROW;
    COCOS_REQUIRE(ntot < 0x7fffffffLL && xbytes * (split ? 2 : 1) < 0x7fffffffLL && wbytes < 0x7fffffffLL &&
                      (long long)B * Cout * g.OHW < (1LL << 40), COCOS_ERR_UNSUPPORTED, "%s: tensor too large", who);
    COCOS_REQUIRE(aligned16(xp) && aligned16(w_hi) && (!split || aligned16(w_lo)), COCOS_ERR_UNSUPPORTED,
                  "%s: operands must be 16-byte aligned", who);
    g.Ntot = (int)ntot; g.xp_bytes = (unsigned)xbytes; g.w_bytes = (unsigned)wbytes;
    COCOS_REQUIRE(!fold_ring || (stride == 1 && g.OH >= 6 && g.OW >= 6), COCOS_ERR_INVALID, "%s: fold mode needs stride 1 and H, W >= 4", who);
    NhwcSplit sp{fold_ring, w_lo, (unsigned)xbytes, x_amax_dev, w_scale_dev};
    // tile: 256 rows unless the layer has at most 128 (a half-empty 256-row tile costs what a full one does); 256 columns
    // when that still gives the CUs something each
    const int cus = nhwc_cu_count() < 1024 ? nhwc_cu_count() : 1024;
    const int bm = Cout <= 128 ? 128 : 256;
    const long long ntm = (Cout + bm - 1) / bm;
    const int bn = (bm == 128 ? (ntot + 255) / 256 >= 100 : ntm * ((ntot + 255) / 256) >= 200) ? 256 : 128;
    const long long tiles = ntm * ((ntot + bn - 1) / bn);
    COCOS_REQUIRE(tiles * g.nsteps < 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "%s: too many tiles", who);
    g.ntiles = (int)tiles;
    // stream-K when the last round of one-tile workgroups would leave more than a fifth of the chip idle
    const long long rounds = (tiles + cus - 1) / cus;
    const bool sk = workspace && tiles > cus && tiles * 5 < rounds * cus * 4 &&
                    workspace_bytes >= kNhwcFlagBytes + (long long)cus * kNhwcSlotBytes && bm == 256 && bn == 256;
    // (measured, B = 8, 66 x 66 outputs: 407 -> 407 0.232 -> 0.204 ms, 512 -> 512 0.268 -> 0.250; nearly every tile is cut, so the
    //  parked partials are 128 MB of extra traffic — with 256 x 128 tiles, 273 of them, stream-K LOST: 0.089 -> 0.101 ms)
    hipStream_t s = as_stream(stream);
    int* flags = static_cast<int*>(workspace);
    float* slots = workspace ? reinterpret_cast<float*>(static_cast<char*>(workspace) + kNhwcFlagBytes) : nullptr;
    const int units = sk ? (int)((tiles * g.nsteps + cus - 1) / cus) : g.nsteps;
#define COCOS_NHWC_GO_W(BMv, BNv, SKv, NWv, TERMSv)                                                                      \
    do {                                                                                                                 \
        auto kern = conv_nhwc_bf16_kernel<BMv, BNv, SKv, NWv, TERMSv>;                                                   \
        const size_t smem = (size_t)(TERMSv == 3 ? 2 * 2 : NB_STAGES) * (BMv + BNv) * NB_ROW;                            \
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)smem));                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)(SKv ? cus : tiles)), dim3(NWv * 64), smem, s, xp, w_hi, bias, y, slots, flags, g, \
                           units, sp);                                                                                  \
    } while (0)
#define COCOS_NHWC_GO(BMv, BNv) do { if (split) COCOS_NHWC_GO_W(BMv, BNv, false, 4, 3); else COCOS_NHWC_GO_W(BMv, BNv, false, 4, 1); } while (0)
    if (bm == 128) {
        if (bn == 256) COCOS_NHWC_GO(128, 256); else COCOS_NHWC_GO(128, 128);
    } else if (bn == 256) {
        static const bool w8 = [] { const char* e = getenv("COCOS_CONV_NHWC_WAVES"); return !(e && e[0] == '4'); }();
        if (split && sk) COCOS_NHWC_GO_W(256, 256, true, 4, 3);
        else if (split) COCOS_NHWC_GO_W(256, 256, false, 4, 3);
        else if (sk) COCOS_NHWC_GO_W(256, 256, true, 4, 1);
        else if (w8) COCOS_NHWC_GO_W(256, 256, false, 8, 1);
        else COCOS_NHWC_GO_W(256, 256, false, 4, 1);
    } else {
        COCOS_NHWC_GO(256, 128);
    }
#undef COCOS_NHWC_GO
#undef COCOS_NHWC_GO_W
    if (fold_ring)
        hipLaunchKernelGGL(conv_nhwc_fold_ring_kernel, dim3((unsigned)(B * Cout)), dim3(256), 0, s, fold_ring, y, g.OH - 2, g.OW - 2);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// y fp32 [B][Cout][OH][OW] = bias + conv(xp), xp bf16 [B][Hp][Wp][Cp] already padded (cocos_conv2d_nhwc_prep_bf16),
// OH = (Hp - dil (KH-1) - 1) / stride + 1, OW likewise; w_planes = cocos_conv2d_weight_planes(mode | 2): bf16 [KH*KW*Cp/32][Cout][32].
// workspace (nullable): see cocos_conv2d_nhwc_bf16_workspace_bytes — without it every launch is one tile per workgroup.
extern "C" int cocos_conv2d_nhwc_bf16(const void* xp, const void* w_planes, const float* bias, float* y, float* fold_ring,
                                      void* workspace, long long workspace_bytes, int B, int Cp, int Hp, int Wp, int Cout, int KH, int KW,
                                      int dil, int stride, cocos_stream_t stream) {
    return cocos_nhwc_gemm_impl(xp, w_planes, nullptr, nullptr, nullptr, bias, y, fold_ring, workspace, workspace_bytes, B, Cp, Hp, Wp,
                                Cout, KH, KW, dil, stride, stream, "conv2d_nhwc_bf16");
}

// K16c: the same GEMM on f16 hi/lo planes with three MFMA terms per product (fp32-accurate): xp from cocos_conv2d_nhwc_prep_f16x3
// (scaled by the power of two of *x_amax_dev), w_hi / w_lo / *w_scale_dev from cocos_conv2d_weight_planes (mode 0 | 1).
extern "C" int cocos_conv2d_nhwc_f16x3(const void* xp, const void* w_hi, const void* w_lo, const float* w_scale_dev,
                                       const float* x_amax_dev, const float* bias, float* y, float* fold_ring, void* workspace,
                                       long long workspace_bytes, int B, int Cp, int Hp, int Wp, int Cout, int KH, int KW, int dil,
                                       int stride, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(w_lo && w_scale_dev, COCOS_ERR_INVALID, "conv2d_nhwc_f16x3: null pointer");
    return cocos_nhwc_gemm_impl(xp, w_hi, w_lo, w_scale_dev, x_amax_dev, bias, y, fold_ring, workspace, workspace_bytes, B, Cp, Hp, Wp,
                                Cout, KH, KW, dil, stride, stream, "conv2d_nhwc_f16x3");
}

extern "C" int cocos_conv2d_nhwc_wgrad_bf16_slices(int B, int OH, int OW, int Cp, int Cout, int KH, int KW) {
    if (B < 1 || OH < 1 || OW < 1 || Cp < 32 || Cout < 1 || OW % 32 != 0) return 0;
    const long long steps = (long long)B * OH * OW / 32;
    const long long tiles = (long long)((Cout + 255) / 256) * (((long long)KH * KW * Cp + 255) / 256);
    long long s = 256 / tiles;                               // ONE round of workgroups on the 256 CUs (a 257th doubles the time)
    if (s > steps / 8) s = steps / 8;                        // at least 8 k-steps per slice
    return (int)(s < 1 ? 1 : s);
}

static int cocos_nhwc_wgrad_impl(const void* xp, const void* dyp, const float* x_amax_dev, const float* g_amax_dev, bool split, float* partial,
                           int B, int Cp, int Hp, int Wp, int Cout, int q, int KH, int KW, int dil, int stride, cocos_stream_t stream,
                           const char* who) {
    using namespace cocos;
    COCOS_REQUIRE(xp && dyp && partial, COCOS_ERR_INVALID, "%s: null pointer", who);
    COCOS_REQUIRE(B >= 1 && Cp >= 32 && Cp % 32 == 0 && Cout >= 1 && KH >= 1 && KW >= 1 && dil >= 1 && q >= 0 && stride >= 1 &&
                      Hp > dil * (KH - 1) && Wp > dil * (KW - 1), COCOS_ERR_INVALID, "%s: bad arguments", who);
    NhwcWgradGeom g;
    g.B = B; g.Cp = Cp; g.Hp = Hp; g.Wp = Wp; g.stride = stride;
    g.OH = (Hp - dil * (KH - 1) - 1) / stride + 1; g.OW = (Wp - dil * (KW - 1) - 1) / stride + 1;
    COCOS_REQUIRE(g.OH >= 1 && g.OW >= 32 && g.OW % 32 == 0, COCOS_ERR_UNSUPPORTED,
                  "%s: output rows must be whole k-steps of 32 positions (OW=%d)", who, g.OW);
    g.Cop = (Cout + 31) / 32 * 32; g.q = q; g.Hq = g.OH + 2 * q; g.Wq = g.OW + 2 * q;
    g.Cout = Cout; g.T = KH * KW; g.KW = KW; g.dil = dil; g.nkb = g.T * (Cp / 32);
    const long long ntot = (long long)B * g.OH * g.OW, xbytes = (long long)B * Hp * Wp * Cp * 2,
                    ybytes = (long long)B * g.Hq * g.Wq * g.Cop * 2;
    COCOS_REQUIRE(ntot < 0x7fffffffLL && xbytes * (split ? 2 : 1) < 0x7fffffffLL && ybytes * (split ? 2 : 1) < 0x7fffffffLL,
                  COCOS_ERR_UNSUPPORTED, "%s: tensor too large", who);
    COCOS_REQUIRE(aligned16(xp) && aligned16(dyp), COCOS_ERR_UNSUPPORTED, "%s: operands must be 16-byte aligned", who);
    g.Ntot = (int)ntot; g.xp_bytes = (unsigned)xbytes; g.dy_bytes = (unsigned)ybytes;
    const int S = cocos_conv2d_nhwc_wgrad_bf16_slices(B, g.OH, g.OW, Cp, Cout, KH, KW);
    g.steps_total = (int)(ntot / 32);
    g.steps_per_slice = (g.steps_total + S - 1) / S;
    const int ntm = (Cout + 255) / 256, ntn = (g.nkb * 32 + 255) / 256;
    NhwcWgradSplit sp{(unsigned)xbytes, (unsigned)ybytes, x_amax_dev, g_amax_dev};
    if (split) {
        auto kern = conv_nhwc_wgrad_bf16_kernel<3>;
        const size_t smem = (size_t)2 * 2 * 2 * 256 * NB_ROW;
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3((unsigned)(ntm * ntn * S)), dim3(256), smem, as_stream(stream), xp, dyp, partial, g, sp);
    } else {
        auto kern = conv_nhwc_wgrad_bf16_kernel<1>;
        const size_t smem = (size_t)NB_STAGES * 2 * 256 * NB_ROW;
        COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3((unsigned)(ntm * ntn * S)), dim3(256), smem, as_stream(stream), xp, dyp, partial, g, sp);
    }
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// partial fp32 [S][Cout][KH*KW*Cp] (k' order of K16; summed and re-ordered by cocos_conv2d_wgrad_reduce), S = ..._slices(...);
// xp as in cocos_conv2d_nhwc_bf16; dyp = bf16 NHWC [B][OH+2q][OW+2q][Cop] (cocos_conv2d_nhwc_prep_bf16 of dy with pad q, zeros).
extern "C" int cocos_conv2d_nhwc_wgrad_bf16(const void* xp, const void* dyp, float* partial, int B, int Cp, int Hp, int Wp, int Cout,
                                            int q, int KH, int KW, int dil, int stride, cocos_stream_t stream) {
    return cocos_nhwc_wgrad_impl(xp, dyp, nullptr, nullptr, false, partial, B, Cp, Hp, Wp, Cout, q, KH, KW, dil, stride, stream,
                           "conv2d_nhwc_wgrad_bf16");
}

// K16c: both operands as f16 hi/lo plane pairs (cocos_conv2d_nhwc_prep_f16x3 with their max|.| cells), three terms per product
extern "C" int cocos_conv2d_nhwc_wgrad_f16x3(const void* xp, const void* dyp, const float* x_amax_dev, const float* g_amax_dev,
                                             float* partial, int B, int Cp, int Hp, int Wp, int Cout, int q, int KH, int KW, int dil,
                                             int stride, cocos_stream_t stream) {
    return cocos_nhwc_wgrad_impl(xp, dyp, x_amax_dev, g_amax_dev, true, partial, B, Cp, Hp, Wp, Cout, q, KH, KW, dil, stride, stream,
                           "conv2d_nhwc_wgrad_f16x3");
}
