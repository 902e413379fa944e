// K0 as a STREAMING kernel (gfx950): y[b,m,n] = sum_k A[m,k] x[b,k,n] (+ bias[m]) for the 1x1 projections of
// correspondence.py:272,:282 (theta / phi: 256 (+151 label channels with --maskmix) -> 256) and for their input
// gradient (A = W^T).
//
// The op moves 2 x 4 bytes per element of x / y and does ~2*256 FLOPs for it: it is HBM-bound by a factor of ~3
// even with the 3-MFMA split product, so the design is about touching x and y once and keeping the memory pipe
// busy from the first to the last cycle.  The general split GEMM (sgemm_f16x3.hip) launches one wave of 256
// workgroups that all load, then all multiply, then all store, and every workgroup re-reads and re-splits the
// whole fp32 weight (51-90 us for 85 MB of traffic); here
//   * the small operand A is split into f16 hi/lo planes ONCE by the caller and lives in the accumulator file
//     for the whole kernel (a wave owns 32*RB rows x all k: up to 256 AGPRs per lane; MFMA reads its A operand
//     straight from there) — no LDS traffic and no re-reads for A.  M is cut into slices of 128*RB rows
//     (blockIdx.y); the slices of one position tile run at the same time on the same XCD, so x comes from HBM
//     once and from that XCD's L2 for the other slice(s);
//   * a workgroup walks over tiles of 32*CB positions (all k rows) persistently; tile t+1 is committed to the
//     other LDS buffer and tile t+2 requested from HBM while tile t is multiplied;
//   * x is split into f16 hi/lo (scaled by the power of two from its device-side max|x|) on the way into LDS,
//     transposed to position-major rows there (8 k of one position = one 16-byte write, row order swizzled so
//     that the writes and the MFMA operand reads spread over the bank groups);
//   * the accumulator registers of a 32-position block leave as single 4-byte stores (128 B per row segment)
//     spread over the MFMA steps of the NEXT block — loads, stores and MFMAs of neighbouring tiles overlap
//     inside every workgroup;
//   * round 2: the MFMAs run on two accumulator chains in term-major order and a sched_group_barrier pipeline deals the
//     other ~450 instructions of a tile out between them (half commits at equal distances, loads behind the last
//     commit of their patch, the next tile's scalar arithmetic in a free slot); tiles beyond the end are switched off
//     through an empty buffer descriptor (a scalar select, not one v_cndmask per memory instruction), row tests are
//     compiled out when K = Kpad / M % (128 RB) = 0; threads beyond the last patch re-stage the first ones instead of
//     carrying a test.  33.4 -> 29.4 us (fwd), 34.0 -> 30.9 us (dx).  What is left is not the order of the
//     instructions (every arrangement of the same work lands within 2 %) but their sum: ablation builds
//     (tools/build_k0_ablations.sh) show the parts adding up — prologue 5 us, f16 split + LDS commit 6-8, stores 5-9,
//     loads 2, MFMAs 7-10 — on a wave that is alone on its SIMD (208-256 AGPRs of weights) and therefore cannot hide
//     anything behind another wave; PMC: issuing 49 % of its cycles, waiting for issue 30 %, on counters 21 %.
// Arithmetic identical to sgemm_f16x3.hip: a.b ~= ah.bh + ah.bl + al.bh on v_mfma_f32_32x32x16_f16, fp32 accumulate.
#include <algorithm>

#include "common.h"

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifndef PS_ABLATE
#define PS_ABLATE 0    // debug builds only (tools/build_k0_ablations.sh): 1 no x loads in the loop, 2 no y stores, 4 no f16 split / LDS commit, 8 no MFMAs
#endif
#ifndef PS_SCHED_N
#define PS_SCHED_N 5   // instructions of any other kind dealt out per MFMA (4 / 5 / 6 / 8 measured: 29.9 / 29.4 / 30.0 / 30.0 us)
#endif

__device__ __forceinline__ float ps_scale_from_amax(const float* amax) {
    if (!amax) return 1.0f;
    const float a = *amax;
    if (!(a > 0.f) || !(a < INFINITY)) return 1.0f;
    int e;
    frexpf(a, &e);
    return ldexpf(1.0f, 10 - e);
}

// LDS row of position n of a tile — rows are position-major [row][k]; the rotation by (n >> 4) spreads the 16
// lanes of one transposing write (positions 4 apart) and the 16 lanes of one operand read (consecutive positions)
// over the 16-byte bank groups (row stride = Kpad + 8 halfs = an odd number of groups)
__device__ __forceinline__ int ps_row(int n) {
    const int ng = n >> 2, j = n & 3, hi = ng >> 2, lo = ng & 3;
    return hi * 16 + lo * 4 + ((j + hi) & 3);
}

// RB = 32-row MFMA blocks per wave (a workgroup owns 128 * RB rows of A), KS = 16-wide k steps (Kpad = 16 * KS),
// CB = 32-position blocks per tile.  KFULL: K == Kpad (every staged row exists: the loads carry no row test);
// MFULL: M % (128 * RB) == 0 (every row of every workgroup exists: the stores carry no row test).
template <int RB, int KS, int CB, bool KFULL, bool MFULL>
__global__ __launch_bounds__(256, 1) void proj_stream_f16x3_kernel(
    const float* __restrict__ x, const _Float16* __restrict__ a_hi, const _Float16* __restrict__ a_lo,
    const float* __restrict__ a_scale, const float* __restrict__ bias, float* __restrict__ y, int M, int K, int N,
    int ntiles, const float* __restrict__ x_amax) {
    constexpr int KP = 16 * KS, XROW = KP + 8, NT = 32 * CB, PLANE = NT * XROW;
    constexpr int KG = KP / 8, NGRP = NT / 4;       // 8-row k groups x 4-position groups of a tile
    constexpr int NU = (KG * NGRP + 255) / 256;     // 8k x 4n patches per thread
    constexpr int SLOTS = KS * RB;                  // MFMA triples per 32-position block
    constexpr int TS = SLOTS * CB;                  // ... per tile: the staging of the next tiles is spread over them
    constexpr bool FULL = (KG * NGRP) % 256 == 0;   // every thread owns NU real patches
    static_assert(RB == 1 || RB == 2, "two accumulator chains: the row blocks (RB = 2) or the k-step parities (RB = 1)");
    static_assert(RB == 2 || KS % 2 == 0, "RB = 1 pairs the k steps");
    static_assert(SLOTS >= 16 * RB, "not enough MFMA steps to carry the stores");
    static_assert(TS >= 8 * NU + 2, "not enough MFMA steps to carry the staging");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const xt = reinterpret_cast<_Float16*>(smem_raw);      // [2 buf][hi|lo][NT rows][XROW]
    float* const bias_s = reinterpret_cast<float*>(xt + 2 * 2 * PLANE);   // [128 * RB] rows of this workgroup

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int tiles_per_img = N / NT;
    const int m_wave = (blockIdx.y * 4 + wave) * 32 * RB;            // first row of this wave
    const float sa = a_scale ? *a_scale : 1.0f, sb = ps_scale_from_amax(x_amax);
    const float oscale = 1.0f / (sa * sb);

    // ---- x staging: patch u of this thread = 8 k rows x 4 positions ------------------------------------------
    // (KG * NGRP not a multiple of 256: the threads beyond the last patch stage the first patches a second time —
    //  same data to the same LDS address, from L1 — so that no load, conversion or LDS write carries a test)
    f32x4 st[NU][8];
    unsigned x_voff[NU];
    int k_left[NU];                                  // rows of the patch that exist (k < K)
    _Float16* c_ptr[NU][4];                          // buffer 0, hi plane, row of position 4*ng + j, k = 8*kg
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int p = tid + 256 * u;
        if (!FULL && p >= KG * NGRP) p -= KG * NGRP;
        const int ng = p % NGRP, kg = p / NGRP;
        x_voff[u] = (unsigned)((kg * 8) * N + ng * 4) * 4u;
        k_left[u] = K - kg * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c_ptr[u][j] = xt + kg * 8 + ps_row(ng * 4 + j) * XROW;
    }
    static_assert(FULL || 2 * KG * NGRP >= 256 * NU, "the wrap-around covers at most one extra round");
    // a tile = image b, positions [n0, n0 + NT); resolved once per tile (one scalar division, in the MFMA shadow of
    // an earlier tile), not per memory op.  Slots: 0 = the tile whose last block is still leaving, 1 = the tile being
    // multiplied, 2 = the next one (already in registers), 3 = the one being requested.
    int tl_b[4], tl_n[4], tl_ok[4];
    auto tile_set = [&](int q, int T) __attribute__((always_inline)) {
        const int ok = T < ntiles, Tc = ok ? T : 0, b = Tc / tiles_per_img;
        tl_b[q] = b;
        tl_n[q] = (Tc - b * tiles_per_img) * NT;
        tl_ok[q] = ok;
    };
    // (the uniform part of the address travels in the scalar offset, which the descriptor's range check does not
    //  see: rows k >= K are switched off through the per-lane offset, tiles beyond the end through an EMPTY
    //  descriptor — a scalar select instead of a per-lane one in front of every load -> zeros)
    auto fetch_piece = [&](int i, int q) __attribute__((always_inline)) {   // i = u * 8 + kk; tile slot q
        const int u = i >> 3, kk = i & 7;
        if ((PS_ABLATE & 1) && q == 3) return;
        st[u][kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            make_rsrc(x + (size_t)tl_b[q] * K * N, tl_ok[q] ? (size_t)K * N * 4 : 0),
            (int)((KFULL || kk < k_left[u]) ? x_voff[u] : kBufOob), (kk * N + tl_n[q]) * 4, 0));
    };
    // one position of a patch (its 8 k values -> one 16-byte LDS write per plane), in two halves of 4 values so that the
    // conversion work can be dealt out evenly over the MFMA groups; the first half waits in four registers
    unsigned c_hw[2], c_lw[2];
    auto commit_half = [&](int i2, int buf) __attribute__((always_inline)) {   // i2 = (u * 4 + j) * 2 + half
        const int u = i2 >> 3, j = (i2 >> 1) & 3, half = i2 & 1;
        if (PS_ABLATE & 4) return;
        unsigned hw[2], lw[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kk = 4 * half + 2 * q;
            const float a = st[u][kk][j] * sb, b = st[u][kk + 1][j] * sb;
            const f16x2 hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
            const f16x2 ll = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a - (float)hh[0], b - (float)hh[1]));
            hw[q] = __builtin_bit_cast(unsigned, hh);
            lw[q] = __builtin_bit_cast(unsigned, ll);
        }
        if (half == 0) {
            c_hw[0] = hw[0]; c_hw[1] = hw[1]; c_lw[0] = lw[0]; c_lw[1] = lw[1];
        } else {
            _Float16* d = c_ptr[u][j] + buf * 2 * PLANE;
            *reinterpret_cast<u32x4*>(d) = u32x4{c_hw[0], c_hw[1], hw[0], hw[1]};
            *reinterpret_cast<u32x4*>(d + PLANE) = u32x4{c_lw[0], c_lw[1], lw[0], lw[1]};
        }
    };
    auto commit_sub = [&](int i, int buf) __attribute__((always_inline)) {   // i = u * 4 + j
        commit_half(2 * i, buf);
        commit_half(2 * i + 1, buf);
    };

    const int T0 = blockIdx.x, stride = gridDim.x;
    tile_set(1, T0);
    tile_set(2, T0 + stride);
    tile_set(3, T0 + 2 * stride);
    tl_b[0] = tl_b[1]; tl_n[0] = tl_n[1]; tl_ok[0] = 0;      // nothing to store while the very first block is multiplied
#pragma unroll
    for (int i = 0; i < NU * 8; ++i) fetch_piece(i, 1);

    // ---- A: this wave's 32*RB rows of both planes [M][KP], resident in the accumulator file ------------------
    // A fragment load straight from memory takes 16 bytes from each of 64 different rows: with 26-32 KB of rows
    // per wave and plane the 128-byte lines are evicted from L1 before their other seven pieces are asked for
    // (up to 8x over-fetch from L2).  The wave's rows are one contiguous
    // slab instead: it is read in whole lines, parked in the (still unused) x buffers — rows padded like the x
    // rows, which makes the fragment reads conflict-free — and picked up from there, one plane after the other.
    f16x8 ah[RB][KS], al[RB][KS];
    {
        constexpr int CPR = KP / 8;                              // 16-byte chunks per row
        constexpr int NCH = 32 * RB * CPR / 64;                  // chunks per lane
        static_assert(32 * RB * CPR % 64 == 0, "slab must divide among the lanes");
        static_assert(4 * 32 * RB * XROW <= 2 * 2 * PLANE, "the four slabs must fit into the x buffers");
        _Float16* const slab = xt + wave * 32 * RB * XROW;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(pl ? a_lo : a_hi, (size_t)M * KP * 2);
            u32x4 v[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((unsigned)m_wave * CPR + lane + 64 * i) * 16u), 0, 0);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int q = lane + 64 * i;
                *reinterpret_cast<u32x4*>(slab + (q / CPR) * XROW + (q % CPR) * 8) = v[i];
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8 f = *reinterpret_cast<const f16x8*>(slab + (rb * 32 + c) * XROW + ks * 16 + h * 8);
                    if (pl) al[rb][ks] = f; else ah[rb][ks] = f;
                }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (pl) asm volatile("" : "+a"(al[rb][ks]));
                    else    asm volatile("" : "+a"(ah[rb][ks]));
                }
        }
    }
    __syncthreads();                                             // slabs consumed: the buffers now belong to x
    if (tid < 128 * RB) {
        const int m = blockIdx.y * 128 * RB + tid;
        bias_s[tid] = (bias && m < M) ? bias[m] : 0.f;
    }
    const float* const bias_l = bias_s + wave * 32 * RB + 4 * h;      // + rb * 32 + acc_row_base(r)

#pragma unroll
    for (int i = 0; i < NU * 4; ++i) commit_sub(i, 0);
#pragma unroll
    for (int i = 0; i < NU * 8; ++i) fetch_piece(i, 2);
    __syncthreads();

    // ---- output: lane = position, register = row of A; one 4-byte store per register -------------------------
    const unsigned y_voff = (unsigned)((m_wave + 4 * h) * N + c) * 4u;
    const int m_lane = m_wave + 4 * h;
    auto store_one = [&](const f32x16 (&reg)[RB], int i, int q, int cb) __attribute__((always_inline)) {   // i = rb * 16 + r; tile slot q
        const int rb = i >> 4, r = i & 15;
        if (PS_ABLATE & 2) return;
        const bool live = MFULL || (m_lane + rb * 32 + acc_row_base(r) < M);
        buf_store1s(make_rsrc(y + (size_t)tl_b[q] * M * N, tl_ok[q] ? (size_t)M * N * 4 : 0), reg[rb][r],
                    live ? y_voff : kBufOob, (unsigned)((rb * 32 + acc_row_base(r)) * N + tl_n[q] + cb * 32) * 4u);
    };

    // One block: 32 positions (column block cb) x all rows of this wave.  The MFMAs run on TWO accumulator chains —
    // the two row blocks (RB = 2) or the even / odd k steps (RB = 1, summed in the hand-over) — in term-major order,
    // so that two MFMAs on the same accumulator are never neighbours: an instruction issued between two DEPENDENT
    // MFMAs costs a ~43-cycle bubble on gfx950, between independent ones it is free, and the wave (alone on its
    // SIMD) has ~500 other instructions per tile to issue in the shadow of the MFMAs.  A group = 6 MFMAs = 2 slots;
    // `hook(slot)` supplies the other work, the sched_group_barrier pipeline deals it out one MFMA : PS_SCHED_N.
    constexpr int KPG = RB == 2 ? 1 : 2;             // k steps per group
    constexpr int NGR = KS / KPG;                    // groups per block
    constexpr int RAG = RB == 2 ? 2 : 1;             // operand read-ahead in groups
    constexpr int NFB = (RAG + 1) * KPG;             // fragment buffers
    int rd_row[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) rd_row[cb] = ps_row(cb * 32 + c);
    auto block = [&](int buf, int cb, f32x16 (&acc)[2], auto&& hook) __attribute__((always_inline)) {
        const _Float16* bb = xt + buf * 2 * PLANE + rd_row[cb] * XROW + h * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        f16x8 bh[NFB], bl[NFB];
#pragma unroll
        for (int s = 0; s < RAG * KPG; ++s) {
            bh[s] = *reinterpret_cast<const f16x8*>(bb + s * 16);
            bl[s] = *reinterpret_cast<const f16x8*>(bb + PLANE + s * 16);
        }
#pragma unroll
        for (int g = 0; g < NGR; ++g) {
#pragma unroll
            for (int q = 0; q < KPG; ++q) {
                const int nx = (g + RAG) * KPG + q;
                if (nx < KS) {
                    bh[nx % NFB] = *reinterpret_cast<const f16x8*>(bb + nx * 16);
                    bl[nx % NFB] = *reinterpret_cast<const f16x8*>(bb + PLANE + nx * 16);
                }
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ks = RB == 2 ? g : 2 * g + j, rb = RB == 2 ? j : 0, fb = ks % NFB;
                    if (!(PS_ABLATE & 8))
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[rb][ks] : ah[rb][ks],
                                                                    term == 1 ? bl[fb] : bh[fb], acc[j], 0, 0, 0);
                }
            hook(2 * g);
            hook(2 * g + 1);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x010 | 0x080, PS_SCHED_N, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Staging of the next tiles, dealt out over the TS slots of a tile: the 8 * NU half commits (tile t+1, ~14
    // instructions each) at equal distances, the 8 loads of a patch (tile t+2) behind its last commit.
    constexpr int NH = 8 * NU;                                   // half commits per tile
    auto half_slot = [](int k) constexpr { return k * (TS - 2) / NH; };
    auto staging = [&](int ts, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NH; ++k)
            if (ts == half_slot(k)) commit_half(k, buf ^ 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int f0 = half_slot(8 * u + 7) + 1;             // first slot after the last commit of patch u
            const int f1 = u + 1 < NU ? half_slot(8 * (u + 1) + 7) : TS - 1;   // its loads are out by here
            const int span = f1 - f0 + 1 > 8 ? 8 : f1 - f0 + 1, per = (8 + span - 1) / span;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (ts == f0 + kk / per) fetch_piece(u * 8 + kk, 3);
        }
    };

    // `pend` = the finished block that is leaving through the stores carried by the block being multiplied
    f32x16 acc[2], pend[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pend[rb][r] = 0.f;
    int it = 0;
    for (int T = T0; T < ntiles; T += stride, ++it) {
        const int buf = it & 1;
        int nb = 0, nn = 0, nk = 0;                  // tile T + 3 * stride, resolved in a slot without staging work
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            block(buf, cb, acc, [&](int slot) __attribute__((always_inline)) {
                if (slot < 16 * RB) store_one(pend, slot, cb == 0 ? 0 : 1, (cb + CB - 1) % CB);
                staging(cb * SLOTS + slot, buf);
                if (cb * SLOTS + slot == TS - 2) {
                    const int T3 = T + 3 * stride;
                    nk = T3 < ntiles;
                    const int Tc = nk ? T3 : 0;
                    nb = Tc / tiles_per_img;
                    nn = (Tc - nb * tiles_per_img) * NT;
                }
            });
            // finished block -> final values (scale undone, bias added); it leaves during the next block
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pend[rb][r] = (RB == 2 ? acc[rb][r] : acc[0][r] + acc[1][r]) * oscale + bias_l[rb * 32 + acc_row_base(r)];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { tl_b[q] = tl_b[q + 1]; tl_n[q] = tl_n[q + 1]; tl_ok[q] = tl_ok[q + 1]; }
        tl_b[3] = nb; tl_n[3] = nn; tl_ok[3] = nk;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16 * RB; ++i) store_one(pend, i, 0, CB - 1);
}

template <int RB, int KS, int CB, bool KFULL, bool MFULL>
static int launch_proj_stream_k(const float* x, const void* a_hi, const void* a_lo, const float* a_scale_dev,
                                const float* bias, float* y, int B, int K, int M, int N, const float* x_amax,
                                hipStream_t s) {
    auto kern = proj_stream_f16x3_kernel<RB, KS, CB, KFULL, MFULL>;
    const size_t smem = (size_t)2 * 2 * (32 * CB) * (16 * KS + 8) * sizeof(_Float16) + 128 * RB * sizeof(float);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int slices = (M + 128 * RB - 1) / (128 * RB);
    const long long ntiles = (long long)B * (N / (32 * CB));
    // the slices of one tile sit 8k blocks apart in dispatch order = on the same XCD, resident at the same time
    int gx = std::max(8, (256 / slices) / 8 * 8);
    if (ntiles < gx) gx = (int)ntiles;
    hipLaunchKernelGGL(kern, dim3(gx, slices), dim3(256), smem, s, x, static_cast<const _Float16*>(a_hi),
                       static_cast<const _Float16*>(a_lo), a_scale_dev, bias, y, M, K, N, (int)ntiles, x_amax);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

template <int RB, int KS, int CB>
static int launch_proj_stream(const float* x, const void* a_hi, const void* a_lo, const float* a_scale_dev,
                              const float* bias, float* y, int B, int K, int M, int N, const float* x_amax,
                              hipStream_t s) {
    const bool kfull = K == 16 * KS, mfull = M % (128 * RB) == 0;
#define COCOS_PS(KF, MF) \
    return launch_proj_stream_k<RB, KS, CB, KF, MF>(x, a_hi, a_lo, a_scale_dev, bias, y, B, K, M, N, x_amax, s)
    if (kfull && mfull) COCOS_PS(true, true);
    if (kfull) COCOS_PS(true, false);
    if (mfull) COCOS_PS(false, true);
    COCOS_PS(false, false);
#undef COCOS_PS
}

}  // namespace cocos

extern "C" int cocos_proj1x1_stream_kpad(int K) {
    return K < 1 ? 0 : K <= 256 ? 256 : K <= 416 ? 416 : 0;
}

// y[b,m,n] = (sum_k A[m,k] x[b,k,n]) / (a_scale * x_scale) + bias[m]
// A as f16 planes [M][Kpad] (k contiguous, zero beyond K; Kpad = cocos_proj1x1_stream_kpad(K)) pre-multiplied by
// *a_scale_dev (cocos_split_f16_ex of the [K][M] matrix with transpose = 1, Cpad = Kpad); x fp32 [B,K,N].
extern "C" int cocos_proj1x1_stream_f16x3(const float* x, const void* a_hi, const void* a_lo,
                                          const float* a_scale_dev, const float* bias, float* y, int B, int K,
                                          int M, int N, const float* x_amax, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && a_hi && a_lo && y, COCOS_ERR_INVALID, "proj1x1_stream_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && N >= 1 && K >= 1 && M >= 1, COCOS_ERR_INVALID,
                  "proj1x1_stream_f16x3: bad dims B=%d K=%d M=%d N=%d", B, K, M, N);
    const int kpad = cocos_proj1x1_stream_kpad(K);
    COCOS_REQUIRE(kpad != 0 && N % 64 == 0, COCOS_ERR_UNSUPPORTED,
                  "proj1x1_stream_f16x3: needs K <= 416 and N %% 64 == 0 (got K=%d N=%d): use cocos_proj1x1_fwd_f16x3 "
                  "/ _bwd_f16x3", K, N);
    COCOS_REQUIRE(aligned16(x) && aligned16(a_hi) && aligned16(a_lo) && aligned16(y), COCOS_ERR_INVALID,
                  "proj1x1_stream_f16x3: pointers must be 16-byte aligned");
    COCOS_REQUIRE((size_t)K * N * 4 < 0x7fffffffull && (size_t)M * N * 4 < 0x7fffffffull &&
                      (long long)B * (N / 32) < 0x7fffffffLL && (M + 127) / 128 <= 65535,
                  COCOS_ERR_UNSUPPORTED, "proj1x1_stream_f16x3: one sample exceeds 2 GiB");
    hipStream_t s = as_stream(stream);
    if (kpad == 256) return launch_proj_stream<2, 16, 2>(x, a_hi, a_lo, a_scale_dev, bias, y, B, K, M, N, x_amax, s);
    return launch_proj_stream<1, 26, 1>(x, a_hi, a_lo, a_scale_dev, bias, y, B, K, M, N, x_amax, s);
}
