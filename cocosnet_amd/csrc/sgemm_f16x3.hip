// fp32-in / fp32-out batched GEMM whose products run on v_mfma_f32_32x32x16_f16: the operands are split
// into f16 hi + lo ON THE FLY while a tile is committed to LDS (3 MFMA terms per product, fp32 accumulate —
// the arithmetic of corr_fused_fwd_f16x3.hip), so no operand planes ever exist in HBM and the traffic is
// that of the fp32 GEMM it replaces.  The split costs ~4 VALU ops per loaded element, paid once per element
// while each element meets >= 128 rows/columns of the other operand: it hides under the MFMAs.
//
//   C[b][m][n] = scale * sum_k A(b,m,k) * B(b,k,n) (+ row_bias[m]),     C row-major (n contiguous)
//   A_KC: A stored [m][k] (k contiguous)  else [k][m];   B_KC: B stored [n][k]  else [k][n]
// Operand magnitudes are arbitrary fp32, f16 is not: each operand is pre-scaled by a power of two derived
// from a device-side max|x| (a_amax / b_amax, e.g. from cocos_absmax; NULL = already O(1), scale 1) so that
// the largest element sits in [2^9, 2^10); the product of the two scales is divided out in the epilogue.
//
// Users: K0 (theta/phi 1x1 projections, correspondence.py:272,:282 and their autograd).
// Tile 256 (M) x 128 (N) x 32 (K), 4 waves as 2 x 2 (128 x 64 each = 4 x 2 MFMA tiles), LDS rows of
// 32 k-halfs + 8 pad (conflict-free 16-byte operand reads), double-buffered; ragged shapes (Cin = 407!)
// are handled by masked 16-byte loads as in sgemm_mfma.hip; split-K for reductions with few output tiles.
#include <algorithm>

#include "common.h"

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XG_BM = 256, XG_BN = 128, XG_BK = 32;
constexpr int XG_ROW = XG_BK + 8;   // halfs per LDS row

__device__ __forceinline__ float xg_scale_from_amax(const float* amax) {
    if (!amax) return 1.0f;
    const float a = *amax;
    if (!(a > 0.f) || !(a < INFINITY)) return 1.0f;
    int e;
    frexpf(a, &e);
    return ldexpf(1.0f, 10 - e);
}

// One float4 of an operand: 4 consecutive elements along the contiguous direction, starting at (mn, k).
// Entirely outside the matrix -> zeros through the descriptor; straddling the ragged end of its row -> masked;
// crossing the end of the buffer -> loaded 1-3 elements early and shifted (see sgemm_mfma.hip).
template <bool KC>
__device__ __forceinline__ f32x4 xg_piece(__amdgpu_buffer_rsrc_t rs, int mn, int k, int MN, int K, int kend) {
    const int total = MN * K;
    const int e0 = KC ? (mn * K + k) : (k * MN + mn);
    const int lim = KC ? kend : MN;
    const int pos = KC ? k : mn;
    const bool ok = (KC ? (mn < MN) : (k < kend)) && pos < lim;
    f32x4 v;
    // exactly ONE load per piece on every path (the launcher rejects operands with fewer than 4 elements):
    // the compiler can then count outstanding loads and wait for the right one instead of for all of them
    const int sh = max(e0 + 4 - total, 0);
    const f32x4 w = buf_load4(rs, ok ? (unsigned)(e0 - sh) * 4u : kBufOob);
    v[0] = sh == 0 ? w[0] : sh == 1 ? w[1] : sh == 2 ? w[2] : w[3];
    v[1] = sh == 0 ? w[1] : sh == 1 ? w[2] : w[3];
    v[2] = sh == 0 ? w[2] : w[3];
    v[3] = w[3];
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (pos + e >= lim) v[e] = 0.f;
    return v;
}

// x*s -> f16 hi (round toward zero) + f16 lo, two elements per instruction
__device__ __forceinline__ void xg_split4(const f32x4& x, float s, u32x2& hi, u32x2& lo) {
    const float a = x[0] * s, b = x[1] * s, c = x[2] * s, d = x[3] * s;
    const f16x2 h0 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    const f16x2 h1 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(c, d));
    const f16x2 l0 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a - (float)h0[0], b - (float)h0[1]));
    const f16x2 l1 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(c - (float)h1[0], d - (float)h1[1]));
    hi = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    lo = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}

// Staging of one operand slab (ROWS x 32 k): NP float4 per thread.
//   KC source:  idx = u*256 + tid -> row = idx >> 3, k = (idx & 7) * 4      (a row's 128 B are 8 lanes)
//   else:       k = (tid & 15) + 16 * (u & 1), mn = ((tid >> 4) + 16 * (u >> 1)) * 4
//               (a wave reads 64-byte row segments; its transposing 2-byte LDS writes hit 32 distinct dwords)
template <bool KC, int ROWS>
struct XgStage {
    static constexpr int NP = ROWS * XG_BK / 4 / 256;
    f32x4 r[NP];
    // FAST: the slab lies entirely inside the matrix — plain 16-byte loads, no masks.  The choice is a
    // template parameter (the k loop is split into an interior part and a tail), not a run-time test: a
    // branch around the loads makes hipcc fall back to s_waitcnt vmcnt(0) in front of every LDS commit.
    template <bool FAST>
    __device__ __forceinline__ void fetch(__amdgpu_buffer_rsrc_t rs, int mn0, int k0, int MN, int K, int kend, int tid) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            int mn, k;
            if (KC) { const int idx = u * 256 + tid; mn = mn0 + (idx >> 3); k = k0 + (idx & 7) * 4; }
            else    { k = k0 + (tid & 15) + 16 * (u & 1); mn = mn0 + ((tid >> 4) + 16 * (u >> 1)) * 4; }
            if (FAST) r[u] = buf_load4(rs, (unsigned)(KC ? mn * K + k : k * MN + mn) * 4u);
            else      r[u] = xg_piece<KC>(rs, mn, k, MN, K, kend);
        }
    }
    template <bool FAST>
    __device__ __forceinline__ void fetch_piece(int u, __amdgpu_buffer_rsrc_t rs, int mn0, int k0, int MN, int K,
                                                int kend, int tid) {
        int mn, k;
        if (KC) { const int idx = u * 256 + tid; mn = mn0 + (idx >> 3); k = k0 + (idx & 7) * 4; }
        else    { k = k0 + (tid & 15) + 16 * (u & 1); mn = mn0 + ((tid >> 4) + 16 * (u >> 1)) * 4; }
        if (FAST) r[u] = buf_load4(rs, (unsigned)(KC ? mn * K + k : k * MN + mn) * 4u);
        else      r[u] = xg_piece<KC>(rs, mn, k, MN, K, kend);
    }
    __device__ __forceinline__ void commit_piece(int u, _Float16* tile, float s, int tid) const {
        u32x2 hi, lo;
        xg_split4(r[u], s, hi, lo);
        if (KC) {
            const int idx = u * 256 + tid, row = idx >> 3, kq = idx & 7;
            _Float16* d = tile + row * XG_ROW + kq * 4;
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + ROWS * XG_ROW) = lo;
        } else {
            const int k = (tid & 15) + 16 * (u & 1), row = ((tid >> 4) + 16 * (u >> 1)) * 4;
            unsigned short* d = reinterpret_cast<unsigned short*>(tile) + row * XG_ROW + k;
            unsigned short* dl = d + ROWS * XG_ROW;
            d[0] = (unsigned short)hi.x;            d[XG_ROW] = (unsigned short)(hi.x >> 16);
            d[2 * XG_ROW] = (unsigned short)hi.y;   d[3 * XG_ROW] = (unsigned short)(hi.y >> 16);
            dl[0] = (unsigned short)lo.x;           dl[XG_ROW] = (unsigned short)(lo.x >> 16);
            dl[2 * XG_ROW] = (unsigned short)lo.y;  dl[3 * XG_ROW] = (unsigned short)(lo.y >> 16);
        }
    }
    // planes: hi at `tile`, lo at `tile + ROWS * XG_ROW`
    __device__ __forceinline__ void commit(_Float16* tile, float s, int tid) const {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            u32x2 hi, lo;
            xg_split4(r[u], s, hi, lo);
            if (KC) {
                const int idx = u * 256 + tid, row = idx >> 3, kq = idx & 7;
                _Float16* d = tile + row * XG_ROW + kq * 4;
                *reinterpret_cast<u32x2*>(d) = hi;
                *reinterpret_cast<u32x2*>(d + ROWS * XG_ROW) = lo;
            } else {
                const int k = (tid & 15) + 16 * (u & 1), row = ((tid >> 4) + 16 * (u >> 1)) * 4;
                unsigned short* d = reinterpret_cast<unsigned short*>(tile) + row * XG_ROW + k;
                unsigned short* dl = d + ROWS * XG_ROW;
                d[0] = (unsigned short)hi.x;            d[XG_ROW] = (unsigned short)(hi.x >> 16);
                d[2 * XG_ROW] = (unsigned short)hi.y;   d[3 * XG_ROW] = (unsigned short)(hi.y >> 16);
                dl[0] = (unsigned short)lo.x;           dl[XG_ROW] = (unsigned short)(lo.x >> 16);
                dl[2 * XG_ROW] = (unsigned short)lo.y;  dl[3 * XG_ROW] = (unsigned short)(lo.y >> 16);
            }
        }
    }
};

#ifdef COCOS_DEBUG_TIMING
__device__ long long g_phase_xg[8];
#define XPH_T(var) const long long var = __builtin_readcyclecounter()
#define XPH_ADD(i, a, b) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_phase_xg[i] += (b) - (a); } while (0)
#else
#define XPH_T(var) do {} while (0)
#define XPH_ADD(i, a, b) do {} while (0)
#endif

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 1) void sgemm_f16x3_kernel(const float* __restrict__ A,
                                                             const float* __restrict__ Bm,
                                                             float* __restrict__ C, int M, int N, int K,
                                                             size_t strideA, size_t strideB, size_t strideC,
                                                             float scale, const float* __restrict__ row_bias,
                                                             int ksplit, int kchunk,
                                                             const float* __restrict__ a_amax,
                                                             const float* __restrict__ b_amax, int gx, int gy) {
    constexpr int APLANE = XG_BM * XG_ROW, BPLANE = XG_BN * XG_ROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* const at = reinterpret_cast<_Float16*>(smem_raw);   // [2 buf][hi|lo][256][ROW]
    _Float16* const bt = at + 2 * 2 * APLANE;                      // [2 buf][hi|lo][128][ROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid, XCD-aware: consecutive virtual ids (the column tiles that share one A row tile, then the row
    // tiles that share the batch item / k slice) run on the same XCD and meet in its L2 — with the hardware's
    // round-robin placement every XCD fetched the shared operand from HBM for itself (dw of K0: dy read 4 times)
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = vb % gx, by = (vb / gx) % gy, bz = vb / (gx * gy);
    const int b = bz / ksplit;
    const int kbeg = (bz % ksplit) * kchunk, kend = min(K, kbeg + kchunk);
    const int m0 = by * XG_BM, n0 = bx * XG_BN;

    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(A + (size_t)b * strideA, (size_t)M * K * 4);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(Bm + (size_t)b * strideB, (size_t)N * K * 4);
    const float sa = xg_scale_from_amax(a_amax), sb = xg_scale_from_amax(b_amax);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Pipeline: LDS holds tile t (being multiplied) and tile t+1 (committed during step t); TWO register
    // stages hold tiles t+1 / t+2 resp. t+2 / t+3 in flight, so a global load has two full steps (~2 us) to
    // arrive — one step is not enough to cover HBM latency, and with K = 256..407 there are only 8-13 steps.
    XPH_T(xp0);
    XgStage<A_KC, XG_BM> sA0, sA1;
    XgStage<B_KC, XG_BN> sB0, sB1;
    const int nsteps = (max(kend - kbeg, 0) + XG_BK - 1) / XG_BK;
    // tiles 0 .. nfull-1 are full k-blocks; a workgroup whose rows/columns all exist may fetch them FAST
    const bool rows_in = (m0 + XG_BM <= M) && (n0 + XG_BN <= N);
    const int nfull = rows_in ? max(kend - kbeg, 0) / XG_BK : 0;

    auto compute = [&](int buf, auto&& mid) {
        const _Float16* ab = at + buf * 2 * APLANE + (wm * 128 + c) * XG_ROW + h * 8;
        const _Float16* bb = bt + buf * 2 * BPLANE + (wn * 64 + c) * XG_ROW + h * 8;
#pragma unroll
        for (int s = 0; s < XG_BK / 16; ++s) {
            f16x8 bvh[2], bvl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bvh[j] = *reinterpret_cast<const f16x8*>(bb + j * 32 * XG_ROW + s * 16);
                bvl[j] = *reinterpret_cast<const f16x8*>(bb + BPLANE + j * 32 * XG_ROW + s * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 avh = *reinterpret_cast<const f16x8*>(ab + i * 32 * XG_ROW + s * 16);
                const f16x8 avl = *reinterpret_cast<const f16x8*>(ab + APLANE + i * 32 * XG_ROW + s * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, bvh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, bvl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl, bvh[j], acc[i][j], 0, 0, 0);
                    mid((s * 4 + i) * 2 + j);          // one staged piece per MFMA triple (16 slots, 12 pieces)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
    };
    // step t: multiply LDS[t&1]; commit the stage that holds tile t+1 into the other buffer (released by the
    // barrier that ended step t-1) and refill it with tile t+3 (FAST when that tile is a full interior block)
    // (one piece = convert + LDS write of one staged float4, then its reload: the split's VALU work, the LDS
    //  writes and the global loads ride between the MFMAs instead of stalling the in-order wave as one block)
#define XG_STEP(FAST, T, SA, SB)                                                              \
    compute((T) & 1, [&](int slot) {                                                          \
        if (slot < 8) {                                                                       \
            SA.commit_piece(slot, at + (((T) & 1) ^ 1) * 2 * APLANE, sa, tid);                \
            SA.template fetch_piece<FAST>(slot, a_rs, m0, kbeg + ((T) + 3) * XG_BK, M, K, kend, tid); \
        } else if (slot < 12) {                                                               \
            SB.commit_piece(slot - 8, bt + (((T) & 1) ^ 1) * 2 * BPLANE, sb, tid);            \
            SB.template fetch_piece<FAST>(slot - 8, b_rs, n0, kbeg + ((T) + 3) * XG_BK, N, K, kend, tid); \
        }                                                                                     \
    })

    // Pipeline: LDS holds tile t (being multiplied) and tile t+1; TWO register stages hold tiles t+1 / t+2 in
    // flight, so a global load has two full steps to arrive (one step does not cover HBM latency, and with
    // K = 256..407 there are only 8-13 steps).
    if (nfull >= 3) {
        sA0.template fetch<true>(a_rs, m0, kbeg, M, K, kend, tid);
        sB0.template fetch<true>(b_rs, n0, kbeg, N, K, kend, tid);
        sA1.template fetch<true>(a_rs, m0, kbeg + XG_BK, M, K, kend, tid);
        sB1.template fetch<true>(b_rs, n0, kbeg + XG_BK, N, K, kend, tid);
        sA0.commit(at, sa, tid);
        sB0.commit(bt, sb, tid);
        sA0.template fetch<true>(a_rs, m0, kbeg + 2 * XG_BK, M, K, kend, tid);
        sB0.template fetch<true>(b_rs, n0, kbeg + 2 * XG_BK, N, K, kend, tid);
    } else {
        sA0.template fetch<false>(a_rs, m0, kbeg, M, K, kend, tid);
        sB0.template fetch<false>(b_rs, n0, kbeg, N, K, kend, tid);
        sA1.template fetch<false>(a_rs, m0, kbeg + XG_BK, M, K, kend, tid);
        sB1.template fetch<false>(b_rs, n0, kbeg + XG_BK, N, K, kend, tid);
        sA0.commit(at, sa, tid);
        sB0.commit(bt, sb, tid);
        sA0.template fetch<false>(a_rs, m0, kbeg + 2 * XG_BK, M, K, kend, tid);
        sB0.template fetch<false>(b_rs, n0, kbeg + 2 * XG_BK, N, K, kend, tid);
    }
    __syncthreads();
    // at the top of step t: LDS[t&1] = tile t; stage 1 (t even) / 0 (t odd) holds tile t+1, the other one t+2
    XPH_T(xp1);
    int t = 0;
    for (; t + 1 < nsteps && t + 4 < nfull; t += 2) {     // both steps refill with full interior tiles (t+3, t+4)
        XG_STEP(true, t, sA1, sB1);
        XG_STEP(true, t + 1, sA0, sB0);
    }
    for (; t < nsteps; t += 2) {                           // tail: masked fetches (also past-the-end tiles: zeros)
        XG_STEP(false, t, sA1, sB1);
        if (t + 1 < nsteps) XG_STEP(false, t + 1, sA0, sB0);
    }
#undef XG_STEP
    XPH_T(xp2);

    const float oscale = scale / (sa * sb);
    float* Cb = C + (size_t)bz * strideC;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + acc_row_base(r) + 4 * h;
                if (m < M && n < N) Cb[(size_t)m * N + n] = acc[i][j][r] * oscale + (row_bias ? row_bias[m] : 0.f);
            }
        }
    XPH_T(xp3);
    XPH_ADD(0, xp0, xp1); XPH_ADD(1, xp1, xp2); XPH_ADD(2, xp2, xp3);
}

template <bool A_KC, bool B_KC>
static int launch_gemm_f16x3(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, float scale,
                             hipStream_t s, bool shared_a, const float* row_bias, int ksplit,
                             const float* a_amax, const float* b_amax) {
    COCOS_REQUIRE((size_t)M * K * 4 < 0x7fffffffull && (size_t)N * K * 4 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "sgemm_f16x3: per-sample operand exceeds 2 GiB (M=%d N=%d K=%d)", M, N, K);
    COCOS_REQUIRE((long long)M * K >= 4 && (long long)N * K >= 4, COCOS_ERR_UNSUPPORTED,
                  "sgemm_f16x3: operands with fewer than 4 elements are not supported (M=%d N=%d K=%d): use the "
                  "fp32 entry point", M, N, K);
    const long long gx = (N + XG_BN - 1) / XG_BN, gy = (M + XG_BM - 1) / XG_BM, gz = (long long)batch * ksplit;
    COCOS_REQUIRE(gx * gy * gz <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "sgemm_f16x3: grid too large");
    auto kern = sgemm_f16x3_kernel<A_KC, B_KC>;
    const size_t smem = (size_t)2 * 2 * (XG_BM + XG_BN) * XG_ROW * sizeof(_Float16);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int kchunk = ((K + ksplit - 1) / ksplit + XG_BK - 1) / XG_BK * XG_BK;
    const dim3 grid((unsigned)(gx * gy * gz));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, A, Bm, C, M, N, K, shared_a ? (size_t)0 : (size_t)M * K,
                       (size_t)N * K, (size_t)M * N, scale, row_bias, ksplit, kchunk, a_amax, b_amax, (int)gx, (int)gy);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// max|x| of a tensor into *out (device), one pass; *out must be 0 on entry (cocos_absmax clears it)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n4, size_t n,
                                                     unsigned* __restrict__ out) {
    float m = 0.f;
    // 8 independent 16-byte loads in flight per thread and iteration (a dependent load -> max chain runs at
    // ~1.3 TB/s; this form is bandwidth-bound)
    constexpr int U = 8;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t idx = i + (size_t)u * stride;
            v[u] = idx < n4 ? *reinterpret_cast<const f32x4*>(x + idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    if (blockIdx.x == 0)
        for (size_t j = n4 * 4 + threadIdx.x; j < n; j += 256) m = fmaxf(m, fabsf(x[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup; non-negative floats order like their bit patterns
    if (threadIdx.x == 0)
        atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// max|x| of up to four tensors in ONE launch (round 6: the benchmark step took four of these passes in a row — the two feature
// tensors and the two projection weights — 5 + 14 us each with nothing to overlap).  Block ranges per segment; within a segment
// the same 8-loads-in-flight loop as above; each cell must hold a finite value >= 0 on entry (accumulate semantics).
struct AbsmaxSegs {
    const float* x[4];
    size_t n4[4], n[4];
    unsigned* out[4];
    int first_block[5];        // segment i owns blocks [first_block[i], first_block[i + 1])
};
__global__ __launch_bounds__(256) void absmax_multi_kernel(const AbsmaxSegs sg) {
    int seg = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) seg += (int)blockIdx.x >= sg.first_block[i] ? 1 : 0;
    const float* __restrict__ x = sg.x[seg];
    const size_t n4 = sg.n4[seg], n = sg.n[seg];
    const int bid = blockIdx.x - sg.first_block[seg], nb = sg.first_block[seg + 1] - sg.first_block[seg];
    float m = 0.f;
    constexpr int U = 8;
    const size_t stride = (size_t)nb * 256;
    for (size_t i = (size_t)bid * 256 + threadIdx.x; i < n4; i += stride * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t idx = i + (size_t)u * stride;
            v[u] = idx < n4 ? *reinterpret_cast<const f32x4*>(x + idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    if (bid == 0)
        for (size_t j = n4 * 4 + threadIdx.x; j < n; j += 256) m = fmaxf(m, fabsf(x[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(sg.out[seg], __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

}  // namespace cocos

#ifdef COCOS_DEBUG_TIMING
extern "C" int cocos_debug_read_timing_xg(long long* host8, int reset) {
    using namespace cocos;
    COCOS_HIP_CHECK(hipDeviceSynchronize());
    COCOS_HIP_CHECK(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_phase_xg), 8 * sizeof(long long)));
    if (reset) {
        long long z[8] = {0};
        COCOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_xg), z, sizeof(z)));
    }
    return COCOS_OK;
}
#endif

static int absmax_launch(const float* x, long long n, float* inout_dev, bool zero_first, hipStream_t s) {
    using namespace cocos;
    COCOS_REQUIRE(x && inout_dev && n >= 1, COCOS_ERR_INVALID, "absmax: bad arguments");
    if (zero_first) COCOS_HIP_CHECK(hipMemsetAsync(inout_dev, 0, sizeof(float), s));
    const size_t n4 = aligned16(x) ? (size_t)n / 4 : 0;
    const unsigned blocks = (unsigned)std::min<size_t>(512, (n4 + 256 * 8 - 1) / (256 * 8) + 1);   // one same-address atomic each
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, x, n4, (size_t)n,
                       reinterpret_cast<unsigned*>(inout_dev));
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_absmax(const float* x, long long n, float* out_dev, cocos_stream_t stream) {
    return absmax_launch(x, n, out_dev, true, cocos::as_stream(stream));
}

// *inout_dev = max(*inout_dev, max|x|): no memset in front of the kernel (a 5 us fill each on the hot path, which
// needs eight of these per step) — the caller hands in a cell that already holds a non-negative finite value,
// e.g. one of a pre-zeroed pool, or the maximum of another part of the same virtual tensor
extern "C" int cocos_absmax_accumulate(const float* x, long long n, float* inout_dev, cocos_stream_t stream) {
    return absmax_launch(x, n, inout_dev, false, cocos::as_stream(stream));
}

// *c_i = max(*c_i, max|x_i[0..n_i)|) for up to four tensors in one launch (x_i == NULL: slot unused).  Cells as for
// cocos_absmax_accumulate: finite, >= 0 on entry.
extern "C" int cocos_absmax4(const float* x0, long long n0, float* c0, const float* x1, long long n1, float* c1, const float* x2,
                             long long n2, float* c2, const float* x3, long long n3, float* c3, cocos_stream_t stream) {
    using namespace cocos;
    const float* xs[4] = {x0, x1, x2, x3};
    const long long ns[4] = {n0, n1, n2, n3};
    float* cs[4] = {c0, c1, c2, c3};
    AbsmaxSegs sg;
    int nseg = 0, blocks = 0;
    for (int i = 0; i < 4; ++i) {
        if (!xs[i]) continue;
        COCOS_REQUIRE(cs[i] && ns[i] >= 1, COCOS_ERR_INVALID, "absmax4: bad arguments for tensor %d", i);
        const size_t n4 = aligned16(xs[i]) ? (size_t)ns[i] / 4 : 0;
        sg.x[nseg] = xs[i]; sg.n4[nseg] = n4; sg.n[nseg] = (size_t)ns[i]; sg.out[nseg] = reinterpret_cast<unsigned*>(cs[i]);
        sg.first_block[nseg] = blocks;
        blocks += (int)std::min<size_t>(512, (n4 + 256 * 8 - 1) / (256 * 8) + 1);
        ++nseg;
    }
    COCOS_REQUIRE(nseg >= 1, COCOS_ERR_INVALID, "absmax4: no tensor");
    for (int i = nseg; i < 4; ++i) { sg.x[i] = sg.x[0]; sg.n4[i] = 0; sg.n[i] = 0; sg.out[i] = sg.out[0]; }
    for (int i = nseg; i <= 4; ++i) sg.first_block[i] = i == nseg ? blocks : 0x7fffffff;
    hipLaunchKernelGGL(absmax_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), sg);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// ---- K0 on the split-precision GEMM (same contract as cocos_proj1x1_fwd / _bwd in sgemm_mfma.hip) -------------
extern "C" int cocos_proj1x1_fwd_f16x3(const float* x, const float* w, const float* bias, float* y, int B,
                                       int Cin, int Cout, int N, const float* x_amax, const float* w_amax,
                                       cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && w && y, COCOS_ERR_INVALID, "proj1x1_fwd_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1 && N >= 1, COCOS_ERR_INVALID,
                  "proj1x1_fwd_f16x3: bad dims B=%d Cin=%d Cout=%d N=%d", B, Cin, Cout, N);
    // y[b,co,n] = sum_ci w[co,ci] x[b,ci,n] + bias[co]     A = w [m][k] (shared), B = x [k][n]
    return launch_gemm_f16x3<true, false>(w, x, y, B, Cout, N, Cin, 1.0f, as_stream(stream), true, bias, 1,
                                          w_amax, x_amax);
}

static int proj1x1_ksplit_f16x3(int B, int Cin, int Cout, int N) {
    const long long tiles = (long long)((Cin + cocos::XG_BN - 1) / cocos::XG_BN) *
                            ((Cout + cocos::XG_BM - 1) / cocos::XG_BM) * B;
    int sp = (int)((256 + tiles - 1) / tiles);
    const int max_sp = (N + 8 * cocos::XG_BK - 1) / (8 * cocos::XG_BK);   // at least 8 K steps per slice
    if (sp > max_sp) sp = max_sp;
    return sp < 1 ? 1 : sp;
}

extern "C" int cocos_proj1x1_bwd_partials_f16x3(int B, int Cin, int Cout, int N) {
    if (B < 1 || Cin < 1 || Cout < 1 || N < 1) return 0;
    return B * proj1x1_ksplit_f16x3(B, Cin, Cout, N);
}

extern "C" int cocos_proj1x1_bwd_f16x3(const float* x, const float* w, const float* dy, float* dx, float* dw_p,
                                       int B, int Cin, int Cout, int N, const float* x_amax,
                                       const float* w_amax, const float* dy_amax, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && w && dy, COCOS_ERR_INVALID, "proj1x1_bwd_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1 && N >= 1, COCOS_ERR_INVALID,
                  "proj1x1_bwd_f16x3: bad dims B=%d Cin=%d Cout=%d N=%d", B, Cin, Cout, N);
    hipStream_t s = as_stream(stream);
    int rc = COCOS_OK;
    // dx[b,ci,n] = sum_co w[co,ci] dy[b,co,n]          A = w [k=co][m=ci] (shared), B = dy [k][n]
    if (dx) rc = launch_gemm_f16x3<false, false>(w, dy, dx, B, Cin, N, Cout, 1.0f, s, true, nullptr, 1, w_amax, dy_amax);
    if (rc != COCOS_OK) return rc;
    // dw_p[p,co,ci] = sum_{n in slice} dy[b,co,n] x[b,ci,n]   A = dy [m][k=n], B = x [n=ci][k=n]; split-K partials
    if (dw_p) rc = launch_gemm_f16x3<true, true>(dy, x, dw_p, B, Cout, Cin, N, 1.0f, s, false, nullptr,
                                                 proj1x1_ksplit_f16x3(B, Cin, Cout, N), dy_amax, x_amax);
    return rc;
}

// ---- K3 on the split-precision GEMM (same contract as cocos_corr_materialize / _bwd in sgemm_mfma.hip) --------
// f[b,i,j] = scale * sum_k qn[b,k,i] kn[b,k,j]        C[m=i][n=j], A=[k][m], B=[k][n]
extern "C" int cocos_corr_materialize_f16x3(const float* qn, const float* kn, float* f, int B, int K, int Nq,
                                            int Nk, float scale, const float* q_amax, const float* k_amax,
                                            cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && f, COCOS_ERR_INVALID, "corr_materialize_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && K >= 1 && Nq >= 1 && Nk >= 1, COCOS_ERR_INVALID,
                  "corr_materialize_f16x3: bad dims B=%d K=%d Nq=%d Nk=%d", B, K, Nq, Nk);
    return launch_gemm_f16x3<false, false>(qn, kn, f, B, Nq, Nk, K, scale, as_stream(stream), false, nullptr, 1,
                                           q_amax, k_amax);
}

extern "C" int cocos_corr_materialize_bwd_f16x3(const float* qn, const float* kn, const float* df, float* dqn,
                                                float* dkn, int B, int K, int Nq, int Nk, float scale,
                                                const float* q_amax, const float* k_amax, const float* df_amax,
                                                cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && df, COCOS_ERR_INVALID, "corr_materialize_bwd_f16x3: null pointer");
    COCOS_REQUIRE(B >= 1 && K >= 1 && Nq >= 1 && Nk >= 1, COCOS_ERR_INVALID,
                  "corr_materialize_bwd_f16x3: bad dims B=%d K=%d Nq=%d Nk=%d", B, K, Nq, Nk);
    hipStream_t s = as_stream(stream);
    int rc = COCOS_OK;
    // dqn[k][i] = sum_j kn[k][j] df[i][j] : C[m=k][n=i], A = kn [m][kk=j] (k-contig), B = df [n=i][kk=j] (k-contig)
    if (dqn) rc = launch_gemm_f16x3<true, true>(kn, df, dqn, B, K, Nq, Nk, scale, s, false, nullptr, 1, k_amax, df_amax);
    if (rc != COCOS_OK) return rc;
    // dkn[k][j] = sum_i qn[k][i] df[i][j] : C[m=k][n=j], A = qn [m][kk=i] (k-contig), B = df [kk=i][n=j] (n-contig)
    if (dkn) rc = launch_gemm_f16x3<true, false>(qn, df, dkn, B, K, Nk, Nq, scale, s, false, nullptr, 1, q_amax, df_amax);
    return rc;
}
