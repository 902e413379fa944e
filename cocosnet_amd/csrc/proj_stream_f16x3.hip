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
//     inside every workgroup.
// Arithmetic identical to sgemm_f16x3.hip: a.b ~= ah.bh + ah.bl + al.bh on v_mfma_f32_32x32x16_f16, fp32 accumulate.
#include <algorithm>

#include "common.h"

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

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
// CB = 32-position blocks per tile
template <int RB, int KS, int CB>
__global__ __launch_bounds__(256, 1) void proj_stream_f16x3_kernel(
    const float* __restrict__ x, const _Float16* __restrict__ a_hi, const _Float16* __restrict__ a_lo,
    const float* __restrict__ a_scale, const float* __restrict__ bias, float* __restrict__ y, int M, int K, int N,
    int ntiles, const float* __restrict__ x_amax) {
    constexpr int KP = 16 * KS, XROW = KP + 8, NT = 32 * CB, PLANE = NT * XROW;
    constexpr int KG = KP / 8, NGRP = NT / 4;       // 8-row k groups x 4-position groups of a tile
    constexpr int NU = (KG * NGRP + 255) / 256;     // 8k x 4n patches per thread
    constexpr int SLOTS = KS * RB;                  // MFMA triples per 32-position block
    constexpr bool FULL = (KG * NGRP) % 256 == 0;   // every thread owns NU real patches
    static_assert(SLOTS >= 16 * RB && SLOTS >= NU * 12, "not enough MFMA steps to carry the stores / the staging");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const xt = reinterpret_cast<_Float16*>(smem_raw);      // [2 buf][hi|lo][NT rows][XROW]
    float* const bias_s = reinterpret_cast<float*>(xt + 2 * 2 * PLANE);   // [128 * RB] rows of this workgroup

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    // threads without a real last patch (KG * NGRP not a multiple of 256) write its zeros here: no branch
    _Float16* const dump = reinterpret_cast<_Float16*>(bias_s + 128 * RB) + tid * 8;
    const int tiles_per_img = N / NT;
    const int m_wave = (blockIdx.y * 4 + wave) * 32 * RB;            // first row of this wave
    const float sa = a_scale ? *a_scale : 1.0f, sb = ps_scale_from_amax(x_amax);
    const float oscale = 1.0f / (sa * sb);

    // ---- x staging: patch u of this thread = 8 k rows x 4 positions ------------------------------------------
    f32x4 st[NU][8];
    unsigned x_voff[NU];
    int k_left[NU];                                  // rows of the patch that exist (k < K); <= 0: none
    _Float16* lds_w[NU];                             // buffer 0, hi plane, (row of position 4*ng, k = 8*kg)
    int ng_of[NU];
    bool has[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int p = tid + 256 * u, ng = p % NGRP, kg = p / NGRP;
        has[u] = FULL || kg < KG;
        x_voff[u] = (unsigned)((kg * 8) * N + ng * 4) * 4u;
        k_left[u] = has[u] ? K - kg * 8 : 0;
        ng_of[u] = ng;
        lds_w[u] = xt + kg * 8;
    }
    // a tile = image b, positions [n0, n0 + NT); resolved once per tile (scalar division), not per memory op
    struct Tile { int b, n0; bool ok; };
    auto tile_of = [&](int T) {
        Tile t;
        t.ok = T < ntiles;
        const int Tc = t.ok ? T : 0;
        t.b = Tc / tiles_per_img;
        t.n0 = (Tc - t.b * tiles_per_img) * NT;
        return t;
    };
    // (the uniform part of the address travels in the scalar offset, which the descriptor's range check does not
    //  see: rows k >= K and tiles beyond the end are switched off through the per-lane offset -> zeros)
    auto fetch_piece = [&](int i, const Tile& t) {   // i = u * 8 + kk
        const int u = i >> 3, kk = i & 7;
        st[u][kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            make_rsrc(x + (size_t)t.b * K * N, (size_t)K * N * 4), (int)((t.ok && kk < k_left[u]) ? x_voff[u] : kBufOob),
            (kk * N + t.n0) * 4, 0));
    };
    auto commit_sub = [&](int i, int buf) {          // i = u * 4 + j: position 4*ng + j of patch u, its 8 k values
        const int u = i >> 2, j = i & 3;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = st[u][2 * q][j] * sb, b = st[u][2 * q + 1][j] * sb;
            const f16x2 hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
            const f16x2 ll = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a - (float)hh[0], b - (float)hh[1]));
            hw[q] = __builtin_bit_cast(unsigned, hh);
            lw[q] = __builtin_bit_cast(unsigned, ll);
        }
        _Float16* d = lds_w[u] + buf * 2 * PLANE + ps_row(ng_of[u] * 4 + j) * XROW;
        _Float16* dl = d + PLANE;
        if (!FULL && !has[u]) d = dl = dump;
        *reinterpret_cast<u32x4*>(d) = u32x4{hw[0], hw[1], hw[2], hw[3]};
        *reinterpret_cast<u32x4*>(dl) = u32x4{lw[0], lw[1], lw[2], lw[3]};
    };

    const int T0 = blockIdx.x, stride = gridDim.x;
    {
        const Tile t0 = tile_of(T0);
#pragma unroll
        for (int i = 0; i < NU * 8; ++i) fetch_piece(i, t0);
    }

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
    {
        const Tile t1 = tile_of(T0 + stride);
#pragma unroll
        for (int i = 0; i < NU * 8; ++i) fetch_piece(i, t1);
    }
    __syncthreads();

    // ---- output: lane = position, register = row of A; one 4-byte store per register -------------------------
    const unsigned y_voff = (unsigned)((m_wave + 4 * h) * N + c) * 4u;
    const int m_lane = m_wave + 4 * h;
    auto store_one = [&](const f32x16 (&acc)[RB], int i, const Tile& t, int cb) {   // i = rb * 16 + r
        const int rb = i >> 4, r = i & 15;
        const bool live = t.ok && (m_lane + rb * 32 + acc_row_base(r) < M);
        buf_store1s(make_rsrc(y + (size_t)t.b * M * N, (size_t)M * N * 4), acc[rb][r], live ? y_voff : kBufOob, (unsigned)((rb * 32 + acc_row_base(r)) * N + t.n0 + cb * 32) * 4u);
    };

    // one block: 32 positions (column block cb) x all rows of this wave; `hook(slot)` after every MFMA triple
    int rd_row[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) rd_row[cb] = ps_row(cb * 32 + c);
    auto block = [&](int buf, int cb, f32x16 (&acc)[RB], auto&& hook) {
        const _Float16* bb = xt + buf * 2 * PLANE + rd_row[cb] * XROW + h * 8;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        constexpr int RA = 3;                        // operand read-ahead in k steps (deeper: measured neutral)
        f16x8 bh[RA], bl[RA];
#pragma unroll
        for (int s = 0; s < RA - 1; ++s) {
            bh[s] = *reinterpret_cast<const f16x8*>(bb + s * 16);
            bl[s] = *reinterpret_cast<const f16x8*>(bb + PLANE + s * 16);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = ks % RA, nx = ks + RA - 1;
            if (nx < KS) {
                bh[nx % RA] = *reinterpret_cast<const f16x8*>(bb + nx * 16);
                bl[nx % RA] = *reinterpret_cast<const f16x8*>(bb + PLANE + nx * 16);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rb][ks], bh[cur], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rb][ks], bl[cur], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rb][ks], bh[cur], acc[rb], 0, 0, 0);
                hook(ks * RB + rb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // `pend` = the finished block that is leaving through the stores carried by the block being multiplied
    f32x16 acc[RB], pend[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pend[rb][r] = 0.f;
    Tile prev = tile_of(T0);
    prev.ok = false;                                 // nothing to store while the very first block is multiplied
    int it = 0;
    for (int T = T0; T < ntiles; T += stride, ++it) {
        const int buf = it & 1;
        const Tile cur = tile_of(T), nxt2 = tile_of(T + 2 * stride);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            block(buf, cb, acc, [&](int slot) {
                if (slot < 16 * RB) store_one(pend, slot, cb == 0 ? prev : cur, (cb + CB - 1) % CB);
                if (cb == 0) {                       // staging of patch u: 4 commits (tile t+1), then its 8 re-loads (t+2)
                    const int u = slot / 12, q = slot % 12;
                    if (u < NU) {
                        if (q < 4) commit_sub(u * 4 + q, buf ^ 1);
                        else       fetch_piece(u * 8 + (q - 4), nxt2);
                    }
                }
            });
            // finished block -> final values (scale undone, bias added); it leaves during the next block
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pend[rb][r] = acc[rb][r] * oscale + bias_l[rb * 32 + acc_row_base(r)];
        }
        prev = cur;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16 * RB; ++i) store_one(pend, i, prev, CB - 1);
}

template <int RB, int KS, int CB>
static int launch_proj_stream(const float* x, const void* a_hi, const void* a_lo, const float* a_scale_dev,
                              const float* bias, float* y, int B, int K, int M, int N, const float* x_amax,
                              hipStream_t s) {
    auto kern = proj_stream_f16x3_kernel<RB, KS, CB>;
    const size_t smem = (size_t)2 * 2 * (32 * CB) * (16 * KS + 8) * sizeof(_Float16) + 128 * RB * sizeof(float) +
                        256 * 8 * sizeof(_Float16);
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
