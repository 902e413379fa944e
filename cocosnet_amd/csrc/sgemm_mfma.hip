// Batched exact-fp32 GEMM on v_mfma_f32_32x32x2_f32 for the MATERIALISED side of the path (gfx950):
//   K3  cocos_corr_materialize      f = scale * qn^T kn          correspondence.py:291,:304
//   K3b cocos_corr_materialize_bwd  autograd of :291
//   K5  cocos_warp_materialized_*   P @ V and its gradients      correspondence.py:318
// These run when the [B,HW,HW] matrix must exist in HBM: return_corr=True (:305-306), the
// WTA_scale branch (:300-303) and match_kernel != 1 (K = 256*mk^2).  The K == 256 training /
// inference path never comes here (see corr_fused_*.hip).
//
//   C[b][m][n] = scale * sum_k A(b,m,k) * B(b,k,n),     C row-major (n contiguous)
//   A_KC: A stored [m][k] (k contiguous)  else [k][m] (m contiguous)
//   B_KC: B stored [n][k] (k contiguous)  else [k][n] (n contiguous)
// Tile 128x128 per workgroup, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// registers); K advances 16 per step through one LDS buffer that is refilled from registers
// (loads for step t+1 are in flight under the 32 MFMAs of step t).  LDS images are always
// [k][m] / [k][n], so operand reads are lane-contiguous (conflict-free); k-contiguous sources are
// transposed by the staging write.
#include "common.h"

namespace cocos {

constexpr int GM = 128, GN = 128, GK = 32;
constexpr int GLD = 132;   // LDS row stride: 16-byte aligned rows, 2-way at worst on transposing writes

constexpr int GST = GK * 128 / 4 / 256;   // float4 per thread per operand slab

struct GemmStage {
    f32x4 r[GST];
};

// Fetch a [GK x 128] operand slab into registers.  `KC`: source is k-contiguous ([mn][k]).
// !EDGE: the slab is entirely inside the matrix — one 16-byte load per piece, nothing else.
// EDGE:  still one 16-byte load per piece (4-byte alignment is enough on gfx950) and no branches: a
//        piece entirely outside the matrix is switched off through the descriptor (offset out of range
//        -> zeros); a piece that straddles the ragged end of its row is masked element-wise after the
//        load; a piece that would cross the END OF THE BUFFER is loaded from 1-3 elements earlier and
//        shifted.  So Cin = 407 or Cv = 154 cost a few VALU ops per load, not 4 scalar loads.
template <bool KC, bool EDGE>
__device__ __forceinline__ void gemm_fetch(GemmStage& st, __amdgpu_buffer_rsrc_t rs, int mn0, int k0,
                                           int MN, int K, int kend, int tid) {
    // K = full reduction length (the row stride of k-contiguous sources); kend <= K = end of this
    // workgroup's slice of it (split-K)
    const int total = MN * K;
#pragma unroll
    for (int u = 0; u < GST; ++u) {
        const int idx = u * 256 + tid;
        int mn, k;
        if (KC) { mn = mn0 + idx / (GK / 4); k = k0 + (idx % (GK / 4)) * 4; }
        else    { k = k0 + (idx >> 5);  mn = mn0 + (idx & 31) * 4; }
        const int e0 = KC ? (mn * K + k) : (k * MN + mn);
        if (!EDGE) {
            st.r[u] = buf_load4(rs, (unsigned)e0 * 4u);
        } else {
            const int lim = KC ? kend : MN;         // extent of the contiguous direction
            const int pos = KC ? k : mn;            // first of the 4 contiguous elements
            const bool ok = (KC ? (mn < MN) : (k < kend)) && pos < lim;
            f32x4 v;
            if (total < 4) {                        // degenerate matrix (uniform branch): scalar loads
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = buf_load1(rs, (ok && pos + e < lim) ? (unsigned)(e0 + e) * 4u : kBufOob);
                st.r[u] = v;
                continue;
            }
            const int sh = max(e0 + 4 - total, 0);  // 1..3 only for the last piece of the buffer
            const f32x4 w = buf_load4(rs, ok ? (unsigned)(e0 - sh) * 4u : kBufOob);
            v[0] = sh == 0 ? w[0] : sh == 1 ? w[1] : sh == 2 ? w[2] : w[3];
            v[1] = sh == 0 ? w[1] : sh == 1 ? w[2] : w[3];
            v[2] = sh == 0 ? w[2] : w[3];
            v[3] = w[3];
#pragma unroll
            for (int e = 1; e < 4; ++e)
                if (pos + e >= lim) v[e] = 0.f;
            st.r[u] = v;
        }
    }
}

template <bool KC>
__device__ __forceinline__ void gemm_commit(const GemmStage& st, float* lds, int tid) {
#pragma unroll
    for (int u = 0; u < GST; ++u) {
        const int idx = u * 256 + tid;
        if (KC) {
            const int mn = idx / (GK / 4), k = (idx % (GK / 4)) * 4;
            lds[(k + 0) * GLD + mn] = st.r[u].x;
            lds[(k + 1) * GLD + mn] = st.r[u].y;
            lds[(k + 2) * GLD + mn] = st.r[u].z;
            lds[(k + 3) * GLD + mn] = st.r[u].w;
        } else {
            const int k = idx >> 5, mn = (idx & 31) * 4;
            *reinterpret_cast<f32x4*>(lds + k * GLD + mn) = st.r[u];
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 1) void sgemm_mfma_kernel(const float* __restrict__ A,
                                                            const float* __restrict__ Bm,
                                                            float* __restrict__ C, int M, int N,
                                                            int K, size_t strideA, size_t strideB,
                                                            size_t strideC, float scale,
                                                            const float* __restrict__ row_bias,
                                                            int ksplit, int kchunk) {
    __shared__ __attribute__((aligned(16))) float at[GK * GLD];
    __shared__ __attribute__((aligned(16))) float bt[GK * GLD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // split-K: blockIdx.z = sample * ksplit + slice; each slice writes its own C slab (partials)
    const int b = blockIdx.z / ksplit;
    const int kbeg = (blockIdx.z % ksplit) * kchunk, kend = min(K, kbeg + kchunk);
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;

    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(A + (size_t)b * strideA, (size_t)M * K * 4);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(Bm + (size_t)b * strideB, (size_t)N * K * 4);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool a_in = m0 + GM <= M, b_in = n0 + GN <= N;   // per operand: a ragged N does not slow A
    GemmStage sa, sb;
    auto fetch = [&](int k0) {
        const bool kfull = k0 + GK <= kend;
        if (a_in && kfull) gemm_fetch<A_KC, false>(sa, a_rs, m0, k0, M, K, kend, tid);
        else               gemm_fetch<A_KC, true>(sa, a_rs, m0, k0, M, K, kend, tid);
        if (b_in && kfull) gemm_fetch<B_KC, false>(sb, b_rs, n0, k0, N, K, kend, tid);
        else               gemm_fetch<B_KC, true>(sb, b_rs, n0, k0, N, K, kend, tid);
    };

    const int nsteps = (max(kend - kbeg, 0) + GK - 1) / GK;
    fetch(kbeg);
    for (int t = 0; t < nsteps; ++t) {
        __syncthreads();
        gemm_commit<A_KC>(sa, at, tid);
        gemm_commit<B_KC>(sb, bt, tid);
        __syncthreads();
        if (t + 1 < nsteps) fetch(kbeg + (t + 1) * GK);
        // operands of step kk+1 are requested while step kk multiplies (LDS latency ~100+ cycles
        // would otherwise sit between the 64-cycle MFMAs); interleave pinned, one region per step
        const float* al = at + h * GLD + wm * 64 + c;
        const float* bl = bt + h * GLD + wn * 64 + c;
        float av[2][2], bv[2][2];
        av[0][0] = al[0]; av[0][1] = al[32]; bv[0][0] = bl[0]; bv[0][1] = bl[32];
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < GK / 2) {
                av[nxt][0] = al[(2 * kk + 2) * GLD]; av[nxt][1] = al[(2 * kk + 2) * GLD + 32];
                bv[nxt][0] = bl[(2 * kk + 2) * GLD]; bv[nxt][1] = bl[(2 * kk + 2) * GLD + 32];
            }
            acc[0][0] = mfma32(av[cur][0], bv[cur][0], acc[0][0]);
            acc[0][1] = mfma32(av[cur][0], bv[cur][1], acc[0][1]);
            acc[1][0] = mfma32(av[cur][1], bv[cur][0], acc[1][0]);
            acc[1][1] = mfma32(av[cur][1], bv[cur][1], acc[1][1]);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    float* Cb = C + (size_t)blockIdx.z * strideC;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + acc_row_base(r) + 4 * h;
                if (m < M && n < N)
                    Cb[(size_t)m * N + n] = acc[i][j][r] * scale + (row_bias ? row_bias[m] : 0.f);
            }
        }
}

template <bool A_KC, bool B_KC>
static int launch_gemm(const float* A, const float* Bm, float* C, int batch, int M, int N, int K,
                       float scale, hipStream_t s, bool shared_a = false,
                       const float* row_bias = nullptr, int ksplit = 1) {
    COCOS_REQUIRE((size_t)M * K * 4 < 0x7fffffffull && (size_t)N * K * 4 < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED, "sgemm: per-sample operand exceeds 2 GiB (M=%d N=%d K=%d)",
                  M, N, K);
    COCOS_REQUIRE((long long)batch * ksplit <= 65535 && (M + GM - 1) / GM <= 65535, COCOS_ERR_UNSUPPORTED,
                  "sgemm: grid too large");
    // ksplit > 1: C holds batch*ksplit partial slabs [M,N]; slices are whole K steps
    const int kchunk = ((K + ksplit - 1) / ksplit + GK - 1) / GK * GK;
    const dim3 grid((N + GN - 1) / GN, (M + GM - 1) / GM, batch * ksplit);
    hipLaunchKernelGGL((sgemm_mfma_kernel<A_KC, B_KC>), grid, dim3(256), 0, s, A, Bm, C, M, N, K,
                       shared_a ? (size_t)0 : (size_t)M * K, (size_t)N * K, (size_t)M * N, scale, row_bias,
                       ksplit, kchunk);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

// dkn[b,k,j] = sum_i qn[b,k,i] * dst[b,j,i]   (dst = dS^T / T written by the dq kernel)
//   C[m=k][n=j], A = qn [m][kk=i] (k-contiguous), B = dst [n=j][kk=i] (k-contiguous)
int sgemm_dkn_from_ds(const float* qn, const float* dst, float* dkn, int B, int K, int Nq, int Nk,
                      hipStream_t s) {
    return launch_gemm<true, true>(qn, dst, dkn, B, K, Nk, Nq, 1.0f, s);
}

}  // namespace cocos

// f[b,i,j] = scale * sum_k qn[b,k,i] kn[b,k,j]        C[m=i][n=j], A=[k][m], B=[k][n]
extern "C" int cocos_corr_materialize(const float* qn, const float* kn, float* f, int B, int K,
                                      int Nq, int Nk, float scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && f, COCOS_ERR_INVALID, "corr_materialize: null pointer");
    COCOS_REQUIRE(B >= 1 && K >= 1 && Nq >= 1 && Nk >= 1, COCOS_ERR_INVALID,
                  "corr_materialize: bad dims B=%d K=%d Nq=%d Nk=%d", B, K, Nq, Nk);
    return launch_gemm<false, false>(qn, kn, f, B, Nq, Nk, K, scale, as_stream(stream));
}

extern "C" int cocos_corr_materialize_bwd(const float* qn, const float* kn, const float* df,
                                          float* dqn, float* dkn, int B, int K, int Nq, int Nk,
                                          float scale, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(qn && kn && df, COCOS_ERR_INVALID, "corr_materialize_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && K >= 1 && Nq >= 1 && Nk >= 1, COCOS_ERR_INVALID,
                  "corr_materialize_bwd: bad dims B=%d K=%d Nq=%d Nk=%d", B, K, Nq, Nk);
    hipStream_t s = as_stream(stream);
    int rc = COCOS_OK;
    // dqn[k][i] = sum_j kn[k][j] df[i][j] : C[m=k][n=i], A = kn [m][kk=j] (k-contig), B = df [n=i][kk=j] (k-contig)
    if (dqn) rc = launch_gemm<true, true>(kn, df, dqn, B, K, Nq, Nk, scale, s);
    if (rc != COCOS_OK) return rc;
    // dkn[k][j] = sum_i qn[k][i] df[i][j] : C[m=k][n=j], A = qn [m][kk=i] (k-contig), B = df [kk=i][n=j] (n-contig)
    if (dkn) rc = launch_gemm<true, false>(qn, df, dkn, B, K, Nk, Nq, scale, s);
    return rc;
}

// out[b,c,i] = sum_j v[b,c,j] p[b,i,j] : C[m=c][n=i], A = v [m][kk=j] (k-contig), B = p [n=i][kk=j] (k-contig)
extern "C" int cocos_warp_materialized_fwd(const float* p, const float* v, float* out, int B,
                                           int Nq, int Nk, int Cv, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(p && v && out, COCOS_ERR_INVALID, "warp_materialized_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && Cv >= 1 && Nq >= 1 && Nk >= 1, COCOS_ERR_INVALID,
                  "warp_materialized_fwd: bad dims B=%d Cv=%d Nq=%d Nk=%d", B, Cv, Nq, Nk);
    return launch_gemm<true, true>(v, p, out, B, Cv, Nq, Nk, 1.0f, as_stream(stream));
}

extern "C" int cocos_warp_materialized_bwd(const float* p, const float* v, const float* dout,
                                           float* dp, float* dv, int B, int Nq, int Nk, int Cv,
                                           cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(p && v && dout, COCOS_ERR_INVALID, "warp_materialized_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && Cv >= 1 && Nq >= 1 && Nk >= 1, COCOS_ERR_INVALID,
                  "warp_materialized_bwd: bad dims B=%d Cv=%d Nq=%d Nk=%d", B, Cv, Nq, Nk);
    hipStream_t s = as_stream(stream);
    int rc = COCOS_OK;
    // dp[i][j] = sum_c dout[c][i] v[c][j] : C[m=i][n=j], A = dout [kk=c][m=i], B = v [kk=c][n=j]
    if (dp) rc = launch_gemm<false, false>(dout, v, dp, B, Nq, Nk, Cv, 1.0f, s);
    if (rc != COCOS_OK) return rc;
    // dv[c][j] = sum_i dout[c][i] p[i][j] : C[m=c][n=j], A = dout [m=c][kk=i] (k-contig), B = p [kk=i][n=j]
    if (dv) rc = launch_gemm<true, false>(dout, p, dv, B, Cv, Nk, Nq, 1.0f, s);
    return rc;
}

// ---- theta / phi 1x1 projections on the same GEMM (correspondence.py:272, :282 and their autograd) ----
// y[b,co,n] = sum_ci w[co,ci] x[b,ci,n] + bias[co]          C[m=co][n], A = w [m][k] (shared), B = x [k][n]
extern "C" int cocos_proj1x1_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                                 int Cin, int Cout, int N, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && w && y, COCOS_ERR_INVALID, "proj1x1_fwd: null pointer");
    COCOS_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1 && N >= 1, COCOS_ERR_INVALID,
                  "proj1x1_fwd: bad dims B=%d Cin=%d Cout=%d N=%d", B, Cin, Cout, N);
    return launch_gemm<true, false>(w, x, y, B, Cout, N, Cin, 1.0f, as_stream(stream), true, bias);
}

// dx[b,ci,n] = sum_co w[co,ci] dy[b,co,n]                  C[m=ci][n], A = w [k=co][m=ci] (shared), B = dy [k][n]
// dw_p[p,co,ci] = sum_{n in slice} dy[b,co,n] x[b,ci,n]    C[m=co][n=ci], A = dy [m][k=n], B = x [n=ci][k=n]
//   The weight gradient is a [Cout,Cin] matrix reduced over B*N positions: 8 output tiles per sample
//   would leave 3/4 of the chip idle, so the reduction is split (split-K) until the launch has >= 512
//   workgroups; the caller adds the cocos_proj1x1_bwd_partials() slabs (and sums dy for the bias).
static int proj1x1_ksplit(int B, int Cin, int Cout, int N) {
    const long long tiles = (long long)((Cin + cocos::GN - 1) / cocos::GN) * ((Cout + cocos::GM - 1) / cocos::GM) * B;
    int sp = (int)((512 + tiles - 1) / tiles);
    const int max_sp = (N + 4 * cocos::GK - 1) / (4 * cocos::GK);      // at least 4 K steps per slice
    if (sp > max_sp) sp = max_sp;
    return sp < 1 ? 1 : sp;
}

extern "C" int cocos_proj1x1_bwd_partials(int B, int Cin, int Cout, int N) {
    if (B < 1 || Cin < 1 || Cout < 1 || N < 1) return 0;
    return B * proj1x1_ksplit(B, Cin, Cout, N);
}

extern "C" int cocos_proj1x1_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw_p,
                                 int B, int Cin, int Cout, int N, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(x && w && dy, COCOS_ERR_INVALID, "proj1x1_bwd: null pointer");
    COCOS_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1 && N >= 1, COCOS_ERR_INVALID,
                  "proj1x1_bwd: bad dims B=%d Cin=%d Cout=%d N=%d", B, Cin, Cout, N);
    hipStream_t s = as_stream(stream);
    int rc = COCOS_OK;
    if (dx) rc = launch_gemm<false, false>(w, dy, dx, B, Cin, N, Cout, 1.0f, s, true);
    if (rc != COCOS_OK) return rc;
    if (dw_p) rc = launch_gemm<true, true>(dy, x, dw_p, B, Cout, Cin, N, 1.0f, s, false, nullptr,
                                           proj1x1_ksplit(B, Cin, Cout, N));
    return rc;
}
