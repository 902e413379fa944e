// Split-precision batched GEMM on v_mfma_f32_32x32x16_f16 (gfx950): both operands arrive as f16 hi + lo
// planes, three MFMA terms per product, fp32 accumulate and output.
//
//   C[b][m][n] = host_scale / (*dev_scale * *dev_scale2) * sum_k A[b][m][k] * B[b][n][k]   (NULL scale = 1)
//   A planes [batch][M][K], B planes [batch][N][K]  (k contiguous), C fp32 [batch][M][N]
//
// Used as the key side of the K2 backward (autograd of correspondence.py:291 w.r.t. phi):
//   dkn[c][j] = sum_i qn[c][i] * dS[i][j]   with  A = planes of k_scale*qn [256][Nq],
//                                                 B = the dS'' planes [Nk][Nq] written by
//                                                     corr_bwd_query_f16x3_kernel (already scaled by s_o*ds_shift)
// At B=8, HW=4096 this is 68.7 GFLOP over a 512 MiB B operand that is read exactly once (the whole M = 256
// extent sits in one workgroup tile): ~0.13 ms of matrix pipe at the sustained f16 rate against ~0.1 ms of
// HBM — the fp32-MFMA version of the same product (sgemm_mfma.hip) is 0.56 ms.
//
// Tile 256 (M) x 128 (N) x 32 (K) per workgroup, 4 waves as 2 x 2, each wave 128 x 64 = 4 x 2 MFMA tiles
// (128 accumulator registers).  LDS rows are 32 k-halfs + 8 pad = 80 B: the 16-byte operand reads of 16
// consecutive rows fall into 16 distinct 4-bank groups (conflict-free).  Double-buffered LDS plus two register
// stages: the loads of k-blocks t+2 and t+3 are in flight while t multiplies.
#include "box3_common.h"

namespace cocos {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef HG_ABLATE
#define HG_ABLATE 0    // debug builds only (tools/xbox_ablate.sh): 1 no T stores, 2 no x box, 4 no MFMAs, 8 no loads inside the k loop, 16 row-major tile order
#endif
constexpr int HG_BM = 256, HG_BN = 128, HG_BK = 32;
constexpr int HG_ROW = HG_BK + 8;   // halfs per LDS row
// A tile (round 6): 64-byte rows, the 16-byte chunk of a row XOR-swizzled by (row >> 2) & 3.  With the padded 80-byte rows the staging
// writes (four lanes per row: 16 lanes = rows r .. r + 3) put rows r and r + 3 onto 12 common banks — PMC: 8.4 M LDS conflict cycles per
// launch, 9 % of its time; here a 16-lane group writes 256 contiguous bytes and the fragment reads of 16 consecutive rows still fall
// into 16 distinct 4-bank groups.  (Re-mapping the lanes instead — one chunk of 16 rows per 16 lanes — made the GLOBAL loads of the tile
// 16-byte pieces of 16 different rows: 0.175 -> 0.200 ms.)  -DHG_A_PADDED keeps the old rows.  The tile's LDS region keeps its size.
#ifdef HG_A_PADDED
constexpr int HG_AROW = HG_ROW;
__device__ __forceinline__ int hg_a_chunk(int row, int kc) { return kc; }
#else
constexpr int HG_AROW = HG_BK;
__device__ __forceinline__ int hg_a_chunk(int row, int kc) { return kc ^ ((row >> 2) & 3); }
#endif

// EXACT: M % 256 == 0, N % 128 == 0, K % 32 == 0 — no bounds tests at all.  With them hipcc wraps every staged load in
// an exec-mask region (s_and_saveexec / s_or per load: ~120 scalar instructions per k-step of 96 MFMAs).
// BSTREAM: the B operand is a stream that nothing re-reads (the dS'' / P planes of the K2 backward: 0.5 GB per launch):
// loaded `nt` so that it does not evict the A planes (4 MB per sample, re-read by every N tile) from the XCD's L2.
// EPI 1 (match_kernel 3, box3_common.h): C is the K = 256 correlation with KEYS in the rows (A = key planes) and queries
// in the columns; the epilogue applies the x-direction diagonal box filter to the wave's 128 x 64 sub-tile (= two image
// rows of keys x one image row of queries on a 64-wide grid) through a per-wave LDS image laid over the dead staging
// buffers, and stores the result in the tile-blocked layout the fused box -> softmax -> warp kernels read.
// EPI 2 (round 4): the same on a 128-wide grid — the sub-tile is ONE image row of keys x HALF an image row of queries;
// the halo between the two key chunks comes from the wave's own registers, the halo between the query halves from the
// neighbouring wave (wave ^ 1) through the LDS images, with workgroup barriers around the exchange.
// DUO (round 4, with EPI 1 / 2): TWO workgroups per CU.  The x-box GEMM has K = 256 — 8 k-steps — and a 128 KB tile to store:
// with one workgroup per CU the matrix pipe idles through every epilogue (the store of 537 MB per 8 samples is 0.13 ms of the
// 0.29 ms launch) and every prologue.  LDS is what held the occupancy at one: this flavour stages ONE k-block (61 KB; the four
// x-box images, 80 KB, lie over it) with a single register stage, commit -> barrier -> multiply -> barrier per step — the bubbles
// a single workgroup would see are the other workgroup's MFMA time, and one workgroup's epilogue drains while the other multiplies.
template <bool EXACT, int BMODE, int EPI = 0, bool DUO = false>
__global__ __launch_bounds__(256, DUO ? 2 : 1) void hgemm_f16x3_kernel(const _Float16* __restrict__ ah,
                                                             const _Float16* __restrict__ al,
                                                             const _Float16* __restrict__ bh,
                                                             const _Float16* __restrict__ bl,
                                                             float* __restrict__ C, int M, int N, int K,
                                                             float host_scale,
                                                             const float* __restrict__ dev_scale,
                                                             const float* __restrict__ dev_scale2) {
    // BMODE 3: the SAME blocks as mode 2 read the other way round — [N/32][K/32] blocks of 2 x [32 n][16 k] (k contiguous):
    // the box-adjoint kernel writes dC once in that layout and both correlation-gradient GEMMs read it (d phi: mode 2 with
    // n = keys, k = queries; d theta: mode 3 with n = queries, k = keys).
    // BMODE 0: B planes row-major [N][K]; 1: [N/128][K/32] blocks of [128 n][32 k] (k contiguous); 2: the orientation the
    // K2 query backward leaves its dS'' / P planes in — [K/32][N/32] blocks of 2 x [32 k][16 n] (n contiguous, 2 KB): staged
    // as they are into an LDS image [32 k][128 n + 32 pad] and read as MFMA fragments with ds_read_b64_tr_b16 (the
    // scheme of conv_f16x3.hip).  Modes 1 and 2 are streams nothing re-reads: loaded `nt`.
    constexpr bool BSTREAM = BMODE != 0;
    constexpr int HG_GROW = HG_BN + 32;             // halfs per k row of the mode-2 image (320 B = 64 B mod 256 B)
    constexpr int APLANE = HG_BM * HG_ROW, BPLANE = HG_BN * HG_ROW;     // (= 32 * HG_GROW: both images are 5120 halfs)
    static_assert(32 * HG_GROW == HG_BN * HG_ROW, "B plane size");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NBUF = DUO ? 1 : 2;
    _Float16* const at = reinterpret_cast<_Float16*>(smem_raw);   // [NBUF][hi|lo][256][ROW]
    _Float16* const bt = at + NBUF * 2 * APLANE;                   // [NBUF][hi|lo][128][ROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    const int ntn = (N + HG_BN - 1) / HG_BN, ntm = (M + HG_BM - 1) / HG_BM;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / (ntn * ntm), rem = vb % (ntn * ntm);
    int mi = rem / ntn, ni = rem % ntn;
    if (EPI != 0 && !(HG_ABLATE & 16) && (ntm & 7) == 0 && (ntn & 7) == 0) {
        // x-box GEMM (both operands are re-read: 16 x 32 tiles per sample at cfg2'): the ~64 tiles an XCD works on at a time
        // form an 8 x 8 block of the tile grid instead of two full rows — 3 MB of operand planes (8 key tiles + 8 query tiles)
        // instead of 4.5 MB, inside the XCD's 4 MB of L2 — and the blocks walk along n inside an m block, so the key planes of
        // an m block are fetched once (round 4; same-box A/B, -DHG_ABLATE=16: neutral at cfg2' — the 256 MB memory-side cache
        // was already serving the re-reads — and 1.23 -> 1.19 ms on the 64 x 128 tile grid of cfg5)
        const int blk = rem >> 6, w = rem & 63, nb = ntn >> 3;
        mi = (blk / nb) * 8 + (w >> 3);
        ni = (blk % nb) * 8 + (w & 7);
    }
    const int m0 = mi * HG_BM, n0 = ni * HG_BN;
    // ragged M (the Attention block's key-side GEMM has M = C/8 = 32..64 rows, its dv GEMM M = C/2): 32-row tiles of this wave
    // that hold a real row; the MFMAs of the others are skipped (wave-uniform) — such a GEMM is then bound by its B stream
    const int rows_live = EXACT ? 4 : max(0, min(4, (M - m0 - wm * 128 + 31) >> 5));

    const size_t abytes = (size_t)M * K * 2, bbytes = (size_t)N * K * 2;
    const __amdgpu_buffer_rsrc_t ah_rs = make_rsrc(ah + (size_t)b * M * K, abytes);
    const __amdgpu_buffer_rsrc_t al_rs = make_rsrc(al + (size_t)b * M * K, abytes);
    const __amdgpu_buffer_rsrc_t bh_rs = make_rsrc(bh + (size_t)b * N * K, bbytes);
    const __amdgpu_buffer_rsrc_t bl_rs = make_rsrc(bl + (size_t)b * N * K, bbytes);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: a row of a k-block is 64 B per plane = 4 chunks of 16 B (K % 8 == 0: chunks are whole).
    // TWO register stages: LDS holds k-blocks t and t+1, the stages hold t+1 / t+2 resp. t+2 / t+3 in flight — the
    // B planes stream from HBM and one step (~1.7 us) of look-ahead does not cover the latency.  No branch around
    // the loads, so the compiler's vmcnt waits are exact.
    // chunk g (0..511) of the B tile of a step: global offset in halfs and LDS position
    auto b_goff = [&](int g, int k0) -> unsigned {
        const int row = g >> 2, kc = g & 3;
        if (BMODE == 2)      // block = 2 x [32 k][16 n]: 8-n chunk kc of k row q sits in half kc >> 1
            return (unsigned)((((k0 >> 5) * (N >> 5) + (n0 >> 5) + (g >> 7)) * 1024) + (kc >> 1) * 512 + ((g & 127) >> 2) * 16 + (kc & 1) * 8);
        if (BMODE == 3)
            return (unsigned)((((n0 >> 5) + (row >> 5)) * (K >> 5) + (k0 >> 5)) * 1024 + (kc >> 1) * 512 + (row & 31) * 16 + (kc & 1) * 8);
        if (BMODE == 1) return (unsigned)(((n0 >> 7) * (K >> 5) + (k0 >> 5)) * 4096 + row * 32 + kc * 8);
        return (unsigned)((n0 + row) * K + k0 + kc * 8);
    };
    auto b_lds = [&](int g) -> int {
        if (BMODE == 2) return ((g & 127) >> 2) * HG_GROW + (g >> 7) * 32 + (g & 3) * 8;
        return (g >> 2) * HG_ROW + (g & 3) * 8;
    };
    struct Stage { u32x4 a[2][4], b[2][2]; };
    Stage st0, st1;
    // (loads are issued in the order in which the step loop consumes them: A h0 l0 h1 l1 ..., then B)
    auto fetch = [&](Stage& st, int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = u * 256 + tid, row = g >> 2, kc = g & 3;
            unsigned off = (unsigned)((m0 + row) * K + k0 + kc * 8) * 2u;
            if (!EXACT && (m0 + row >= M || k0 + kc * 8 >= K)) off = kBufOob;
            st.a[0][u] = __builtin_amdgcn_raw_buffer_load_b128(ah_rs, (int)off, 0, 0);
            st.a[1][u] = __builtin_amdgcn_raw_buffer_load_b128(al_rs, (int)off, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = u * 256 + tid, row = g >> 2, kc = g & 3;
            unsigned off = b_goff(g, k0) * 2u;
            if (!EXACT && BMODE != 2 && (n0 + row >= N || k0 + kc * 8 >= K)) off = kBufOob;
            if ((BMODE == 2 || BMODE == 3) && k0 >= K) off = kBufOob;       // look-ahead past the last k-block
            st.b[0][u] = __builtin_amdgcn_raw_buffer_load_b128(bh_rs, (int)off, 0, BSTREAM ? 2 : 0);
            st.b[1][u] = __builtin_amdgcn_raw_buffer_load_b128(bl_rs, (int)off, 0, BSTREAM ? 2 : 0);
        }
    };
    auto commit = [&](const Stage& st, int buf) {
        _Float16* ab = at + buf * 2 * APLANE;
        _Float16* bb = bt + buf * 2 * BPLANE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = u * 256 + tid, row = g >> 2, kc = g & 3;
            *reinterpret_cast<u32x4*>(ab + row * HG_AROW + hg_a_chunk(row, kc) * 8) = st.a[0][u];
            *reinterpret_cast<u32x4*>(ab + APLANE + row * HG_AROW + hg_a_chunk(row, kc) * 8) = st.a[1][u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = u * 256 + tid;
            *reinterpret_cast<u32x4*>(bb + b_lds(g)) = st.b[0][u];
            *reinterpret_cast<u32x4*>(bb + BPLANE + b_lds(g)) = st.b[1][u];
        }
    };

    const int nsteps = (K + HG_BK - 1) / HG_BK;
    fetch(st0, 0);
    if (!DUO) {
        fetch(st1, HG_BK);
        commit(st0, 0);
        fetch(st0, 2 * HG_BK);
        __syncthreads();
    }
    // at the top of step t: LDS[t&1] = k-block t; st1 (t even) / st0 (t odd) holds t+1, the other stage t+2
    // one staged 16-byte piece (plane pl, chunk u of operand A or B) -> the other LDS buffer, and its register
    // takes the load for k-block t+3: issued ONE piece at a time between MFMAs (12 pieces per step, one per
    // (s, i) pair ... ) so that LDS writes and global loads overlap the matrix pipe instead of stalling the
    // in-order wave as a block of 24 memory instructions
    auto piece = [&](Stage& st, int idx, int buf, int k0) {
        _Float16* ab = at + buf * 2 * APLANE;
        _Float16* bb = bt + buf * 2 * BPLANE;
        if (idx < 8) {                                   // A: plane idx & 1, chunk idx >> 1
            const int pl = idx & 1, u = idx >> 1;
            const int g = u * 256 + tid, row = g >> 2, kc = g & 3;
            *reinterpret_cast<u32x4*>(ab + pl * APLANE + row * HG_AROW + hg_a_chunk(row, kc) * 8) = st.a[pl][u];
            unsigned off = (unsigned)((m0 + row) * K + k0 + kc * 8) * 2u;
            if (!EXACT && (m0 + row >= M || k0 + kc * 8 >= K)) off = kBufOob;
            st.a[pl][u] = __builtin_amdgcn_raw_buffer_load_b128(pl ? al_rs : ah_rs, (int)off, 0, 0);
        } else {                                         // B: plane (idx - 8) & 1, chunk (idx - 8) >> 1
            const int pl = (idx - 8) & 1, u = (idx - 8) >> 1;
            const int g = u * 256 + tid, row = g >> 2, kc = g & 3;
            *reinterpret_cast<u32x4*>(bb + pl * BPLANE + b_lds(g)) = st.b[pl][u];
            unsigned off = b_goff(g, k0) * 2u;
            if (!EXACT && BMODE != 2 && (n0 + row >= N || k0 + kc * 8 >= K)) off = kBufOob;
            if ((BMODE == 2 || BMODE == 3) && k0 >= K) off = kBufOob;
            st.b[pl][u] = __builtin_amdgcn_raw_buffer_load_b128(pl ? bl_rs : bh_rs, (int)off, 0, BSTREAM ? 2 : 0);
        }
    };
    // mode 2: transpose-read addressing (lane i of a 16-lane group supplies row i>>2, columns 4(i&3)..+3 of its 4 x 16 block)
    const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * HG_GROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto step = [&](int t, Stage& st) {
        const int buf = DUO ? 0 : (t & 1);
        const _Float16* ab = at + buf * 2 * APLANE + (wm * 128 + c) * HG_AROW;      // (+ the swizzled chunk 2 s + h of the step)
        const _Float16* bb = bt + buf * 2 * BPLANE + (BMODE == 2 ? wn * 64 + tr_off : (wn * 64 + c) * HG_ROW + h * 8);
#pragma unroll
        for (int s = 0; s < HG_BK / 16; ++s) {
            f16x8 bvh[2], bvl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (BMODE == 2) {
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    const _Float16* p = bb + s * 16 * HG_GROW + j * 32;
                    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
                    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * HG_GROW));
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + BPLANE));
                    const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + BPLANE + 4 * HG_GROW));
                    bvh[j] = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    bvl[j] = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                } else {
                    bvh[j] = *reinterpret_cast<const f16x8*>(bb + j * 32 * HG_ROW + s * 16);
                    bvl[j] = *reinterpret_cast<const f16x8*>(bb + BPLANE + j * 32 * HG_ROW + s * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ach = hg_a_chunk(c, 2 * s + h) * 8;          // ((row >> 2) & 3 of row = wm*128 + c + 32 i is that of c)
                const f16x8 avh = *reinterpret_cast<const f16x8*>(ab + i * 32 * HG_AROW + ach);
                const f16x8 avl = *reinterpret_cast<const f16x8*>(ab + APLANE + i * 32 * HG_AROW + ach);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!(HG_ABLATE & 4) && (EXACT || i < rows_live)) {     // (ragged M: a wave-uniform branch around the MFMAs of all-padding row tiles)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, bvh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, bvl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl, bvh[j], acc[i][j], 0, 0, 0);
                    }
                    // 16 (s, i, j) slots per step, 12 pieces: k-block t+1 -> LDS[buf ^ 1] (released by the barrier
                    // that ended step t-1), k-block t+3 -> the freed registers
                    const int slot = (s * 4 + i) * 2 + j;
                    if (!DUO && slot < 12) piece(st, slot, buf ^ 1, (t + 3) * HG_BK);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
    };
    if (DUO) {
        for (int t = 0; t < nsteps; ++t) {
            commit(st0, 0);                                   // (every wave passed the barrier that ended step t - 1)
            if (!(HG_ABLATE & 8) && t + 1 < nsteps) fetch(st0, (t + 1) * HG_BK);  // in flight while this k-block multiplies
            __syncthreads();
            step(t, st0);                                     // (ends with a barrier)
        }
    } else {
        int t = 0;
        for (; t + 1 < nsteps; t += 2) {      // (no branch inside the pair: it would cost the exact vmcnt waits)
            step(t, st1);
            step(t + 1, st0);
        }
        if (t < nsteps) step(t, st1);
    }

    const float scale = host_scale / ((dev_scale ? *dev_scale : 1.0f) * (dev_scale2 ? *dev_scale2 : 1.0f));
    if (EPI == 1 || EPI == 2) {
        // (the last step() ended with a barrier: the staging buffers are dead)
        float* const img = reinterpret_cast<float*>(smem_raw) + wave * kXbFloats;
        xbox_zero_border(img, lane);
        const int nqblk = N >> 5;
        const __amdgpu_buffer_rsrc_t t_rs = make_rsrc(C + (size_t)b * M * N, (size_t)M * N * 4);
        float* const nbr = reinterpret_cast<float*>(smem_raw) + (wave ^ 1) * kXbFloats;     // EPI 2: the other half of my query row
        if (EPI == 2) __syncthreads();          // every image's border is zero before a neighbour writes into it
        // EPI 2: the RAW last key of chunk 0 (pass 0 replaces acc[0..1] by their filtered values before pass 1 needs it)
        const float k63_q0 = acc[1][0][15], k63_q1 = acc[1][1][15];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x16 (&t)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&acc[2 * hh]);
            if (HG_ABLATE & 2) {
            } else if (EPI == 1) {
                xbox_64x64(t, img, lane);
            } else {
                // 128-wide grid (box3_common.h): the wave's 128 keys are ONE image row (two chunks hh = 0, 1, handled one after
                // the other), its 64 queries HALF a row whose other half belongs to wave ^ 1.
                const f32x16 (&t1)[2][2] = *reinterpret_cast<const f32x16 (*)[2][2]>(&acc[2]);     // (raw in pass 0)
                xbox_write_chunk(t, img, lane);
                if (hh == 0) {
                    xbox_put_first_key(t1, img, lane);                       // key 64 of chunk 0 = key 0 of chunk 1
                } else {
                    xbox_put_key_column(img, 3, k63_q0, k63_q1, lane, 1);    // key -1 of chunk 1 = key 63 of chunk 0
                    xbox_zero_column(img, 68, lane);                         // chunk 1 ends at the grid's right edge
                }
                if (wn == 0) {                                               // my last query is the neighbour's query -1
                    xbox_put_last_query(t, nbr, lane);
                    if (hh == 0) xbox_put_corner(t1, nbr, lane, false, true);                          // (key 64, query -1)
                    else xbox_put_corner_value(nbr, lane, true, true, k63_q0, k63_q1);                // (key -1, query -1)
                } else {                                                     // my first query is the neighbour's query 64
                    xbox_put_first_query(t, nbr, lane);
                    if (hh == 0) xbox_put_corner(t1, nbr, lane, false, false);                         // (key 64, query 64)
                    else xbox_put_corner_value(nbr, lane, true, false, k63_q0, k63_q1);               // (key -1, query 64)
                }
                __syncthreads();
                xbox_add_diagonals(t, img, lane);
                __syncthreads();                                             // the next pass overwrites the images
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const unsigned blk = (unsigned)((((m0 + wm * 128 + (2 * hh + kt) * 32) >> 5) * nqblk + ((n0 + wn * 64 + qt * 32) >> 5)) * 4096);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (!(HG_ABLATE & 1) || t[kt][qt][4 * g] == 1234.5f)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(u32x4, f32x4{t[kt][qt][4 * g] * scale, t[kt][qt][4 * g + 1] * scale,
                                                            t[kt][qt][4 * g + 2] * scale, t[kt][qt][4 * g + 3] * scale}),
                            t_rs, (int)(blk + (unsigned)(g * 1024 + lane * 16)), 0, 2);     // nt: 537 MB written once, read by another
                                                                                             // kernel — keep the operand planes in L2
                }
        }
        return;
    }
    float* Cb = C + (size_t)b * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + acc_row_base(r) + 4 * h;
                if (EXACT || (m < M && n < N)) Cb[(size_t)m * N + n] = acc[i][j][r] * scale;
            }
        }
}

}  // namespace cocos

extern "C" int cocos_hgemm_f16x3(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo,
                                 float* c, int batch, int M, int N, int K, float host_scale,
                                 const float* dev_scale, const float* dev_scale2, int b_blocked,
                                 cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(a_hi && a_lo && b_hi && b_lo && c, COCOS_ERR_INVALID, "hgemm_f16x3: null pointer");
    COCOS_REQUIRE(batch >= 1 && M >= 1 && N >= 1 && K >= 1, COCOS_ERR_INVALID,
                  "hgemm_f16x3: bad dims batch=%d M=%d N=%d K=%d", batch, M, N, K);
    COCOS_REQUIRE(K % 8 == 0, COCOS_ERR_UNSUPPORTED, "hgemm_f16x3: K=%d must be a multiple of 8", K);
    COCOS_REQUIRE(!b_blocked || (N % 128 == 0 && K % 32 == 0), COCOS_ERR_INVALID,
                  "hgemm_f16x3: a blocked B operand needs N %% 128 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
    COCOS_REQUIRE((size_t)M * K * 2 < 0x7fffffffull && (size_t)N * K * 2 < 0x7fffffffull, COCOS_ERR_UNSUPPORTED,
                  "hgemm_f16x3: per-sample operand exceeds 2 GiB");
    for (const void* p : {a_hi, a_lo, b_hi, b_lo})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "hgemm_f16x3: planes must be 16-byte aligned");
    const long long blocks = (long long)batch * ((N + HG_BN - 1) / HG_BN) * ((M + HG_BM - 1) / HG_BM);
    COCOS_REQUIRE(blocks <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "hgemm_f16x3: grid too large");
    const size_t smem = (size_t)2 * 2 * (HG_BM + HG_BN) * HG_ROW * sizeof(_Float16);
    const bool exact = M % HG_BM == 0 && N % HG_BN == 0 && K % HG_BK == 0;
    COCOS_REQUIRE(b_blocked >= 0 && b_blocked <= 3 && (b_blocked < 2 || exact || (N % 128 == 0 && K % 32 == 0)),
                  COCOS_ERR_INVALID, "hgemm_f16x3: b_blocked=%d", b_blocked);
    auto kern = exact ? (b_blocked == 3 ? hgemm_f16x3_kernel<true, 3> : b_blocked == 2 ? hgemm_f16x3_kernel<true, 2>
                         : b_blocked ? hgemm_f16x3_kernel<true, 1> : hgemm_f16x3_kernel<true, 0>)
                      : (b_blocked == 3 ? hgemm_f16x3_kernel<false, 3> : b_blocked == 2 ? hgemm_f16x3_kernel<false, 2>
                         : b_blocked ? hgemm_f16x3_kernel<false, 1> : hgemm_f16x3_kernel<false, 0>);
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, as_stream(stream),
                       static_cast<const _Float16*>(a_hi), static_cast<const _Float16*>(a_lo),
                       static_cast<const _Float16*>(b_hi), static_cast<const _Float16*>(b_lo), c, M, N, K,
                       host_scale, dev_scale, dev_scale2);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}

extern "C" int cocos_box3_corr_xbox_f16x3(const void* k_hi, const void* k_lo, const void* q_hi, const void* q_lo,
                                          float* t_blocked, int batch, int Nk, int Nq, int K, int grid_w,
                                          const float* k_scale_dev, const float* q_scale_dev, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(k_hi && k_lo && q_hi && q_lo && t_blocked, COCOS_ERR_INVALID, "box3_corr_xbox_f16x3: null pointer");
    COCOS_REQUIRE(batch >= 1 && Nk >= 1 && Nq >= 1 && K >= 1, COCOS_ERR_INVALID,
                  "box3_corr_xbox_f16x3: bad dims batch=%d Nk=%d Nq=%d K=%d", batch, Nk, Nq, K);
    COCOS_REQUIRE((grid_w == 64 || grid_w == 128) && Nk % HG_BM == 0 && Nq % HG_BN == 0 && K % HG_BK == 0, COCOS_ERR_UNSUPPORTED,
                  "box3_corr_xbox_f16x3: needs a 64- or 128-wide grid, Nk %% 256 == 0, Nq %% 128 == 0, K %% 32 == 0 "
                  "(w=%d Nk=%d Nq=%d K=%d)", grid_w, Nk, Nq, K);
    COCOS_REQUIRE((size_t)Nk * K * 2 < 0x7fffffffull && (size_t)Nq * K * 2 < 0x7fffffffull &&
                      (size_t)Nk * Nq * 4 < 0x7fffffffull,
                  COCOS_ERR_UNSUPPORTED, "box3_corr_xbox_f16x3: per-sample tensor exceeds 2 GiB");
    for (const void* p : {k_hi, k_lo, q_hi, q_lo, (const void*)t_blocked})
        COCOS_REQUIRE(aligned16(p), COCOS_ERR_INVALID, "box3_corr_xbox_f16x3: pointers must be 16-byte aligned");
    const long long blocks = (long long)batch * (Nq / HG_BN) * (Nk / HG_BM);
    COCOS_REQUIRE(blocks <= 0x7fffffffLL, COCOS_ERR_UNSUPPORTED, "box3_corr_xbox_f16x3: grid too large");
    // two workgroups per CU (DUO): one staging buffer (61 KB) under the four x-box images (80 KB)
    constexpr size_t kStage1 = (size_t)2 * (HG_BM + HG_BN) * HG_ROW * sizeof(_Float16), kImages = (size_t)4 * kXbFloats * sizeof(float);
    const size_t smem = kStage1 > kImages ? kStage1 : kImages;
    static_assert(2 * (kStage1 > kImages ? kStage1 : kImages) <= 160 * 1024, "two workgroups of the x-box GEMM must fit one CU's LDS");
    auto kern = grid_w == 64 ? hgemm_f16x3_kernel<true, 0, 1, true> : hgemm_f16x3_kernel<true, 0, 2, true>;
    COCOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, as_stream(stream),
                       static_cast<const _Float16*>(k_hi), static_cast<const _Float16*>(k_lo),
                       static_cast<const _Float16*>(q_hi), static_cast<const _Float16*>(q_lo), t_blocked, Nk, Nq, K,
                       1.0f, k_scale_dev, q_scale_dev);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
