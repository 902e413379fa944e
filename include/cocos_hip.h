/*
 * cocos_hip.h — C ABI of the MI355X-native (gfx950) correspondence hot path.
 *
 * The reference (microsoft/CoCosNet) has no FFI of its own: its hot path is a chain of
 * ATen calls inside NoVGGCorrespondence.forward (models/networks/correspondence.py:271-372).
 * This header is the boundary we introduce underneath that Python class contract; each entry
 * point names the reference lines it replaces.  See INTEGRATION.md for the ctypes binding.
 *
 * Conventions (all entry points):
 *   - plain C, no torch types; pointers are DEVICE pointers owned by the caller, fp32,
 *     contiguous, 16-byte aligned; the library never allocates or frees device memory;
 *   - "positions" are the flattened h*w feature grid; every feature tensor is CHANNEL-MAJOR
 *     [B, C, positions] — exactly what `tensor.view(B, C, -1)` gives in the reference;
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no hidden device sync;
 *   - re-entrant / thread-safe (no global mutable state except a thread-local error string);
 *   - return 0 on success, a negative code on failure (never throws):
 *       -1 invalid argument, -2 unsupported shape, -3 HIP runtime error, -4 workspace too small.
 */
#ifndef COCOS_HIP_H
#define COCOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COCOS_OK 0
#define COCOS_ERR_INVALID (-1)
#define COCOS_ERR_UNSUPPORTED (-2)
#define COCOS_ERR_HIP (-3)
#define COCOS_ERR_WORKSPACE (-4)

typedef void* cocos_stream_t; /* hipStream_t */

/* ABI version: major*10000 + minor*100 + patch. */
int cocos_version(void);
/* Message for the last failing call on this thread ("" if none). */
const char* cocos_last_error_string(void);

/* ---------------------------------------------------------------------------------------
 * K1  centre + L2-normalise  (correspondence.py:277-280 for theta, :287-289 for phi)
 *   x      [B,K,N]  raw theta/phi (after view or F.unfold)
 *   y      [B,K,N]  out: (x - mean) / (||x - mean||_2 over K + eps)
 *   norm   [B,N]    out: ||x - mean||_2 per position (saved for backward)
 *   center_over_channels == 1  -> PONO_C: mean over K per position   (dim_mean = 1)
 *                        == 0  -> mean over N per channel            (dim_mean = -1);
 *                                 needs row_ws [B,K] floats of scratch.
 *                        == 2  -> no centring: y = x / (||x||_2 over K + eps), i.e.
 *                                 util.feature_normalize (util/util.py:31-34) as applied to the adaptor
 *                                 outputs at correspondence.py:247-248 (and :245 for the feature-pair loss)
 *   eps = sys.float_info.epsilon (2.220446049250313e-16) in the reference.
 * ------------------------------------------------------------------------------------- */
#define COCOS_CENTER_POSITIONS 0
#define COCOS_CENTER_CHANNELS 1
#define COCOS_CENTER_NONE 2
int cocos_center_l2norm_fwd(const float* x, float* y, float* norm, float* row_ws,
                            int B, int K, int N, int center_over_channels, float eps,
                            cocos_stream_t stream);
/* Backward of K1 (autograd of :277-280).  dy [B,K,N] -> dx [B,K,N].
 * col_ws [B,N] and row_ws [B,K] floats of scratch are required when center_over_channels == 0
 * (may be NULL otherwise). */
int cocos_center_l2norm_bwd(const float* y, const float* norm, const float* dy, float* dx,
                            float* col_ws, float* row_ws,
                            int B, int K, int N, int center_over_channels, float eps,
                            cocos_stream_t stream);
/* Same as cocos_center_l2norm_bwd, and on return *dx_amax_inout = max(*dx_amax_inout, max|dx|) (the cell must hold
 * a finite value >= 0, e.g. 0): the consumer of dx on the hot path — the backward of the theta/phi projection,
 * correspondence.py:272,:282 — needs it for the scale of its f16 split; produced while dx is written. */
int cocos_center_l2norm_bwd_amax(const float* y, const float* norm, const float* dy, float* dx, float* col_ws,
                                 float* row_ws, int B, int K, int N, int center_over_channels, float eps,
                                 float* dx_amax_inout, cocos_stream_t stream);

/* K1 for the split-precision correlation kernels (round 3): the forward writes the f16 hi/lo OPERAND PLANES of
 * plane_scale * y itself — position-major [B,N,256] (always) and channel-major [B,256,N] (nullable pair; the backward wants
 * them) — and no fp32 y: 12 B/element instead of 24 for K1 + two cocos_split_f16 launches per tensor.  K == 256,
 * N % 4 == 0, center_over_channels 1 (PONO_C) or 2 (none).  The planes are bit-identical to cocos_split_f16(y, scale).
 * The backward reads y back from the channel-major planes. */
int cocos_center_l2norm_fwd_planes(const float* x, float* norm, void* pos_hi, void* pos_lo, void* chan_hi /* nullable */,
                                   void* chan_lo /* nullable */, int B, int K, int N, int center_over_channels, float eps,
                                   float plane_scale, cocos_stream_t stream);
int cocos_center_l2norm_bwd_planes(const void* chan_hi, const void* chan_lo, const float* norm, const float* dy, float* dx,
                                   int B, int K, int N, int center_over_channels, float eps, float plane_scale,
                                   float* dx_amax_inout /* nullable */, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K2  fused correlation -> /temperature -> softmax over key positions -> warp
 *     (correspondence.py:281,291,304,307,318 and the extra P@V products at :334,:343-344,
 *      :351-372; the HWxHW matrix never reaches HBM)
 *   qn   [B,K,Nq]   normalised theta (query / content positions)
 *   kn   [B,K,Nk]   normalised phi   (key / exemplar positions)
 *   v    [B,Cv,Nk]  channels to warp (avg-pooled exemplar RGB, label maps, ... concatenated)
 *   out  [B,Cv,Nq]  out[b,c,i] = sum_j softmax_j(qn[:,i].kn[:,j] * inv_temperature) * v[b,c,j]
 *   lse  [B,Nq]     row log-sum-exp of the scaled logits (natural log), saved for backward
 *   logits_t [B,Nk,Nq] or NULL: when given (training), the scaled logits * log2(e) are also
 *                   written, key-major, for cocos_corr_softmax_warp_bwd_query to read back instead
 *                   of recomputing them (inference passes NULL: nothing HWxHW reaches HBM)
 * The column softmax `softmax(f^T)` of :338/:351 is this same call with qn/kn swapped.
 * Supported: K == 256 (match_kernel 1), 1 <= Cv <= 160, any Nq, Nk >= 1.
 * ------------------------------------------------------------------------------------- */
int cocos_corr_softmax_warp_fwd(const float* qn, const float* kn, const float* v,
                                float* out, float* lse, float* logits_t /* nullable */,
                                int B, int K, int Nq, int Nk, int Cv, float inv_temperature,
                                cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K2 forward, split-precision flavour (same reference lines, same outputs as the call above): the
 * matrix products run on v_mfma_f32_32x32x16_f16 with every fp32 operand carried as two f16 planes,
 * x*scale ~= hi + lo (22 mantissa bits), three MFMA terms per product, fp32 accumulation — fp32-class
 * accuracy at ~1/5 of the fp32-MFMA pipe time on gfx950.
 *   cocos_split_f16:  x [B,C,N] fp32 -> hi, lo (f16); transpose = 0: [B,C,N]; 1: [B,N,C]; `scale` must be
 *                     a power of two (exact).
 *   qh,ql [B,Nq,256]  kh,kl [B,Nk,256]  position-major planes of qn*operand_scale, kn*operand_scale
 *   vh,vl [B,Cv,Nk]   channel-major planes of v * s_v, s_v = *v_scale_dev (a power of two, e.g. the one
 *                     cocos_split_f16_ex derives from max|v|; NULL = 1): V has no a-priori magnitude in a general
 *                     forward() call, so its planes are normalised like every other operand and the epilogue
 *                     undoes the scale
 *   saved_logits      NULL (inference: nothing HWxHW reaches HBM) or cocos_corr_softmax_warp_saved_logits_bytes()
 *                     bytes, 16-byte aligned (training): the raw logits accumulator of every 32x32 tile in a
 *                     PRIVATE tile-blocked layout ([key tile][32-query block][k][lane][4]: four contiguous 1 KB
 *                     stores per wave and tile), read back only by cocos_corr_softmax_warp_bwd_query_f16x3
 *   Supported: K == 256, 1 <= Cv <= 159 (one padding channel of the V tile carries the row sums), Nk % 4 == 0
 *   (otherwise COCOS_ERR_UNSUPPORTED: use the fp32 call).
 * ------------------------------------------------------------------------------------- */
int cocos_split_f16(const float* x, void* hi, void* lo, int B, int C, int N, int transpose, float scale,
                    cocos_stream_t stream);
size_t cocos_corr_softmax_warp_saved_logits_bytes(int B, int Nq, int Nk);
int cocos_corr_softmax_warp_fwd_f16x3(const void* qh, const void* ql, const void* kh, const void* kl,
                                      const void* vh, const void* vl,
                                      float* out, float* lse, void* saved_logits /* nullable */,
                                      const float* v_scale_dev /* nullable */,
                                      const unsigned* v_lo_mask_dev /* nullable */,
                                      int B, int K, int Nq, int Nk, int Cv, float inv_temperature,
                                      float operand_scale,
                                      const float* q_scale_dev /* nullable */, const float* k_scale_dev /* nullable: operands
                                      without an a-priori magnitude (ops.softmax_attention) — planes scaled by device-side
                                      powers of two (cocos_split_f16_ex); given as a pair they replace operand_scale */,
                                      cocos_stream_t stream);
/* The same with one more output for the MAGNITUDE-FREE flavour (q_scale_dev / k_scale_dev given, v_lo_mask_dev NULL: operands that
 * are not unit-norm columns — Attention.forward, architecture.py:114-127, feeds raw 1x1-conv outputs; a randomly initialised SPADE
 * generator reaches |logit| ~ 1e10 there): in that flavour the running maximum is kept in the units of the raw accumulator and the
 * exponent is formed from an EXACT difference (round 4; with the maximum in the log2 domain one ulp of m was ~700 in the exponent
 * and the output NaN).  rowstat_out (nullable) [B][3][Nq]: per query (m as an unevaluated fp32 sum hi + lo in raw units, log2 l - bias) — what the backward of this
 * flavour takes instead of the row LSE (cocos_corr_softmax_warp_bwd_query_f16x3_ex).  In this flavour saved_logits holds RELATIVE
 * raw accumulators, s - m_tile, and mtile_out [B][ceil(Nk/32)][2][Nq] (required with saved_logits) the m_tile (hi, lo) they are relative to:
 * absolute logits of 1e9 would be rounded to +-70 in the exponent, differences of fp32 maxima are exact. */
int cocos_corr_softmax_warp_fwd_f16x3_ex(const void* qh, const void* ql, const void* kh, const void* kl,
                                         const void* vh, const void* vl, float* out, float* lse, void* saved_logits,
                                         const float* v_scale_dev, const unsigned* v_lo_mask_dev, int B, int K, int Nq, int Nk, int Cv,
                                         float inv_temperature, float operand_scale, const float* q_scale_dev,
                                         const float* k_scale_dev, float* rowstat_out /* nullable */, float* mtile_out /* nullable */,
                                         int k_active /* 0 = K; else: channels >= k_active of q and k are zero padding (the Attention
                                         block's C/8 channels in 256-channel planes): k_active <= 32 / <= 64 / <= 128 run instantiations without
                                         the QK steps, fragment reads and key fetches of the padding */,
                                         cocos_stream_t stream);
/* bit (c >> 5) of *mask_inout_dev |= (channel c of the channel-major f16 plane [B,C,N] has a non-zero element); the
 * cell must hold 0 (or an earlier partial mask) on entry; C <= 1024.  Run on the LO plane of V: value channels that
 * are exact in f16 (one-hot labels, masks) have an all-zero lo plane, and when every 32-channel block but the first
 * reports zero the K2 split kernels skip that plane's staging, fragment reads and MFMA term for those blocks
 * (v_lo_mask_dev of the two calls; NULL = no information, the general path). */
int cocos_f16_plane_block_mask(const void* plane, int B, int C, int N, unsigned* mask_inout_dev, cocos_stream_t stream);

/* Split-precision backward of K2 (same reference lines as cocos_corr_softmax_warp_bwd_query / _bwd_key_from_ds):
 *   cocos_split_f16_ex: as cocos_split_f16, plus (a) transposed rows padded with zero channels to Cpad halfs,
 *       (b) when amax_dev != NULL the scale is chosen on the device as the power of two that brings
 *       *amax_dev (= max|x|, e.g. torch's x.abs().amax()) into [2^9, 2^10), and written to *scale_out_dev.
 *   cocos_corr_softmax_warp_bwd_query_f16x3:
 *       kch,kcl [B,256,Nk]  channel-major planes of k_scale*kn
 *       vph,vpl [B,Nk,CvPad] position-major planes of s_v*v, s_v = *v_scale_dev (NULL = 1);
 *       gph,gpl [B,Nq,CvPad] of s_o*dout, s_o = *g_scale_dev
 *       out, dout [B,Cv,Nq] fp32 (for D = sum_c dout*out, fp64);  lse, saved_logits as left by the f16x3 forward
 *       -> dqn [B,256,Nq] fp32;  dsh,dsl [B,Nk,Nq] planes of dS'' = s_o*s_v*ds_shift * dS^T/T (both NULL: skipped);
 *          *ds_scale_out_dev = s_o*s_v*ds_shift (ds_shift is derived on the device from *v_amax_dev = max|v|);
 *          psh,psl [B,Nk,Nq] planes of 2^14 * P (both NULL: skipped) for the V gradient of the cycle terms:
 *          dv = hgemm(A = channel-major planes of s_o*dout [Cv][Nq], B = P planes, host_scale = 2^-14,
 *          dev_scale = g_scale_dev)
 *       Supported: K == 256, Cv <= 160, CvPad = Cv rounded up to 32, Nk % 8 == 0.
 *   planes_blocked != 0 (query kernel) <-> b_blocked = 2 (GEMM): the dS'' / P planes are stored in the orientation
 *       the accumulators of the query kernel have — [query][key] — as [Nq/32][Nk/32] blocks of 2 x [32 queries][16 keys]
 *       halfs (2 KB each, contiguous; the two halves are keys 0..15 and 16..31 of the tile): a wave's tile leaves its registers as two 16-byte stores per plane (no transposition),
 *       the GEMM stages four consecutive blocks per k-step and reads its B fragments with the transposing LDS read
 *       (needs Nk % 128 == 0, Nq % 32 == 0; both sides must agree).  planes_blocked = 0 <-> b_blocked = 0: row-major
 *       [Nk][Nq].  b_blocked = 1 (GEMM only): [N/128][K/32] blocks of [128 n][32 k] halfs, k contiguous.
 *   cocos_hgemm_f16x3: C[b][m][n] = host_scale / (*dev_scale * *dev_scale2) * sum_k A[b][m][k] B[b][n][k] on hi/lo planes
 *       (k contiguous, K % 8 == 0); the key side is  dkn = hgemm(A = planes of k_scale*qn [256][Nq],
 *       B = dS'' planes [Nk][Nq], host_scale = 1/k_scale, dev_scale = ds_scale_out_dev). */
int cocos_split_f16_ex(const float* x, void* hi, void* lo, int B, int C, int N, int Cpad, int transpose,
                       float scale, const float* amax_dev /* nullable */, float* scale_out_dev /* nullable */,
                       cocos_stream_t stream);
/* 2-D form with padded rows: x [rows][cols] -> hi, lo [rows][cols_pad], zero beyond cols (scale as in _ex): the
 * weight planes [M][Kpad] of cocos_proj1x1_stream_f16x3 straight from the weight matrix. */
int cocos_split_f16_rows(const float* x, void* hi, void* lo, int rows, int cols, int cols_pad, float scale,
                         const float* amax_dev, float* scale_out_dev, cocos_stream_t stream);
/* cocos_split_f16_ex(transpose = 1, device-side scales) for two tensors of one shape [B,C,N] in one launch (the K2 / K19 backward
 * splits d out and v back to back). */
int cocos_split_f16_transpose_pair(const float* x0, void* hi0, void* lo0, const float* amax0_dev, float* scale0_out_dev,
                                   const float* x1, void* hi1, void* lo1, const float* amax1_dev, float* scale1_out_dev, int B, int C,
                                   int N, int Cpad, cocos_stream_t stream);
int cocos_corr_softmax_warp_bwd_query_f16x3(
    const void* kch, const void* kcl, const void* vph, const void* vpl, const void* gph, const void* gpl,
    const float* g_scale_dev, const float* out, const float* dout, const float* lse, const void* saved_logits,
    float* dqn, void* dsh /* nullable */, void* dsl /* nullable */, void* psh /* nullable */,
    void* psl /* nullable */, const float* v_amax_dev, const float* v_scale_dev /* nullable */,
    float* ds_scale_out_dev, const unsigned* v_lo_mask_dev /* nullable: as in the forward call */,
    int B, int K, int Nq, int Nk, int Cv, int CvPad, float inv_temperature, float k_scale,
    const float* q_scale_dev /* nullable */, const float* k_scale_dev /* nullable: as in the forward call */,
    int planes_blocked, cocos_stream_t stream);
/* ... with `rowstat` [B][3][Nq] and `mtile` [B][ceil(Nk/32)][2][Nq] (both or neither) as written by
 * cocos_corr_softmax_warp_fwd_f16x3_ex: the magnitude-free flavour's backward (P from exact differences; `lse` is then not read for P). */
int cocos_corr_softmax_warp_bwd_query_f16x3_ex(
    const void* kch, const void* kcl, const void* vph, const void* vpl, const void* gph, const void* gpl,
    const float* g_scale_dev, const float* out, const float* dout, const float* lse, const void* saved_logits,
    float* dqn, void* dsh, void* dsl, void* psh, void* psl, const float* v_amax_dev, const float* v_scale_dev,
    float* ds_scale_out_dev, const unsigned* v_lo_mask_dev, int B, int K, int Nq, int Nk, int Cv, int CvPad,
    float inv_temperature, float k_scale, const float* q_scale_dev, const float* k_scale_dev, int planes_blocked,
    const float* rowstat /* nullable */, const float* mtile /* nullable */,
    const float* d_pre /* nullable: D[b][i] = sum_c dout * out from cocos_rowdot_f64 — the kernel then skips its own serial fp64 loop */,
    int k_active /* 0 = K; as in the forward call: the dqn MFMAs of all-padding channel blocks do not exist in the K <= 32 / 64 / 128
    instantiations, and the dqn rows of those blocks (channels >= 32 / 64 / 128) are NOT written */,
    cocos_stream_t stream);
/* d[b][i] = sum_c a[b][c][i] * b[b][c][i], fp64 accumulation, fp32 result: D of the softmax backward (autograd of
 * correspondence.py:307/:318: dS = P * (dP - D)) as a streaming kernel. */
int cocos_rowdot_f64(const float* a, const float* b, float* d, int B, int C, int N, cocos_stream_t stream);
int cocos_hgemm_f16x3(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* c,
                      int batch, int M, int N, int K, float host_scale, const float* dev_scale /* nullable */,
                      const float* dev_scale2 /* nullable */, int b_blocked, cocos_stream_t stream);

/* Backward of K2 (autograd of :291-318), flash-style: the logits are recomputed from qn/kn and `lse`.
 *   dout [B,Cv,Nq] -> dqn [B,K,Nq], dkn [B,K,Nk], dv [B,Cv,Nk]
 * It is exposed in stages so that the caller picks the strategy for the key side:
 *   prepare        dvec[b,i] = sum_c dout[b,c,i] * out[b,c,i]      (needed by `key`, and by `query`
 *                  when it has no saved logits; the saved-logits query kernel computes it itself)
 *   query          dqn; reads the forward's logits_t if given (else recomputes the logits);
 *                  optionally also writes ds_t [B,Nk,Nq] = (dS)^T / T
 *   key            dkn (and dv if non-NULL) with a SECOND recomputation of the logits
 *   key_from_ds    dkn = qn . dS as a plain fp32-MFMA GEMM over the ds_t written by `query`
 * `query(ds_t) + key_from_ds` executes 2*HW^2*(K+Cv) fewer MFMA FLOPs per sample than `query + key`
 * for 8 bytes of HBM traffic per logit — the faster choice on MI355X (fp32 MFMA is only ~25 FLOP per
 * HBM byte) whenever B*Nq*Nk*4 bytes of scratch are available and dv is not needed.
 * cocos_corr_softmax_warp_bwd is the all-in-one convenience (prepare + query + key; ws holds dvec:
 * cocos_corr_softmax_warp_bwd_workspace_bytes() = B*Nq*4 bytes; dqn / dkn / dv may be NULL). */
size_t cocos_corr_softmax_warp_bwd_workspace_bytes(int B, int K, int Nq, int Nk, int Cv);
int cocos_corr_softmax_warp_bwd_prepare(const float* out, const float* dout, float* dvec,
                                        int B, int Nq, int Cv, cocos_stream_t stream);
int cocos_corr_softmax_warp_bwd_query(const float* qn, const float* kn, const float* v,
                                      const float* out, const float* lse, const float* dout,
                                      const float* dvec /* nullable when logits_t is given */,
                                      const float* logits_t /* nullable: from the forward */,
                                      float* dqn, float* ds_t /* nullable */,
                                      int B, int K, int Nq, int Nk, int Cv, float inv_temperature,
                                      cocos_stream_t stream);
int cocos_corr_softmax_warp_bwd_key(const float* qn, const float* kn, const float* v,
                                    const float* lse, const float* dout, const float* dvec,
                                    float* dkn, float* dv /* nullable */,
                                    int B, int K, int Nq, int Nk, int Cv, float inv_temperature,
                                    cocos_stream_t stream);
int cocos_corr_softmax_warp_bwd_key_from_ds(const float* qn, const float* ds_t, float* dkn,
                                            int B, int K, int Nq, int Nk, cocos_stream_t stream);
int cocos_corr_softmax_warp_bwd(const float* qn, const float* kn, const float* v,
                                const float* out, const float* lse, const float* dout,
                                float* dqn, float* dkn, float* dv,
                                void* ws, size_t ws_bytes,
                                int B, int K, int Nq, int Nk, int Cv, float inv_temperature,
                                cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K3  materialised correlation  f[b,i,j] = scale * sum_k qn[b,k,i] * kn[b,k,j]
 *     (correspondence.py:291 + :304; needed by return_corr=True (:305-306), WTA_scale (:300-303)
 *      and match_kernel != 1 where K = 256*mk^2)
 *   f [B,Nq,Nk]; any K >= 1.
 * ------------------------------------------------------------------------------------- */
int cocos_corr_materialize(const float* qn, const float* kn, float* f,
                           int B, int K, int Nq, int Nk, float scale, cocos_stream_t stream);

/* K3b  gradients of K3 w.r.t. its operands (autograd of :291):
 *   dqn[b,k,i] = scale * sum_j df[b,i,j] kn[b,k,j] ;  dkn[b,k,j] = scale * sum_i df[b,i,j] qn[b,k,i]
 *   Either output may be NULL. */
int cocos_corr_materialize_bwd(const float* qn, const float* kn, const float* df,
                               float* dqn, float* dkn,
                               int B, int K, int Nq, int Nk, float scale, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K4  row softmax over the last dimension of a materialised matrix (correspondence.py:307)
 *     and its backward  ds = p * (dp - sum(p*dp)).  In-place (p == s, ds == dp) allowed.
 * ------------------------------------------------------------------------------------- */
int cocos_row_softmax_fwd(const float* s, float* p, int64_t rows, int cols, cocos_stream_t stream);
int cocos_row_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int cols,
                          cocos_stream_t stream);

/* K5  P @ V on a materialised P  (correspondence.py:318 on the fallback path)
 *   p [B,Nq,Nk], v [B,Cv,Nk] -> out [B,Cv,Nq];  and its two gradients:
 *   dp[b,i,j] = sum_c dout[b,c,i] v[b,c,j] ;  dv[b,c,j] = sum_i p[b,i,j] dout[b,c,i]. */
int cocos_warp_materialized_fwd(const float* p, const float* v, float* out,
                                int B, int Nq, int Nk, int Cv, cocos_stream_t stream);
int cocos_warp_materialized_bwd(const float* p, const float* v, const float* dout,
                                float* dp, float* dv,
                                int B, int Nq, int Nk, int Cv, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K0  theta / phi 1x1 projections (correspondence.py:272, :282: nn.Conv2d(Cl, 256, kernel_size=1))
 *   fwd: y[b,co,n] = sum_ci w[co,ci] x[b,ci,n] + bias[co]      x [B,Cin,N], w [Cout,Cin], bias [Cout] or NULL
 *   bwd: dx[b,ci,n] = sum_co w[co,ci] dy[b,co,n]  (dx may be NULL);
 *        dw_p[p,co,ci] = sum_{n in slice p} dy[b,co,n] x[b,ci,n]  partial sums (may be NULL):
 *        P = cocos_proj1x1_bwd_partials(B,Cin,Cout,N) slabs of [Cout,Cin] (sample x split-K slice, so the
 *        launch fills the chip); the caller adds the P slabs, and sums dy over (b,n) for the bias gradient.
 * ------------------------------------------------------------------------------------- */
int cocos_proj1x1_fwd(const float* x, const float* w, const float* bias, float* y,
                      int B, int Cin, int Cout, int N, cocos_stream_t stream);
int cocos_proj1x1_bwd_partials(int B, int Cin, int Cout, int N);   /* 0 on bad dims */
int cocos_proj1x1_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw_p,
                      int B, int Cin, int Cout, int N, cocos_stream_t stream);

/* K0 on the split-precision GEMM (sgemm_f16x3.hip): same contract, fp32 tensors in and out; the operands are
 * split into f16 hi/lo on the fly inside the kernel (3 MFMA terms per product, fp32 accumulate).  Each operand
 * is pre-scaled by a power of two derived from a device-side max|.| (x_amax, w_amax, dy_amax: 1-element device
 * arrays, e.g. from cocos_absmax; NULL = the operand is already O(1)).
 *   cocos_absmax: *out_dev = max|x[0..n)| (one pass; cleared by the call). */
int cocos_absmax(const float* x, long long n, float* out_dev, cocos_stream_t stream);
/* *inout_dev = max(*inout_dev, max|x|) without the memset cocos_absmax puts in front: the cell must hold a finite
 * value >= 0 (pre-zeroed pool; or the running maximum over the parts of a virtually concatenated tensor). */
int cocos_absmax_accumulate(const float* x, long long n, float* inout_dev, cocos_stream_t stream);
/* The same for up to four tensors in ONE launch (x_i == NULL: slot unused): the max|.| passes a step takes in a row — the two
 * feature tensors and the two projection weights in front of K23 / K0 — share a launch (csrc/sgemm_f16x3.hip). */
int cocos_absmax4(const float* x0, long long n0, float* c0, const float* x1, long long n1, float* c1, const float* x2,
                  long long n2, float* c2, const float* x3, long long n3, float* c3, cocos_stream_t stream);
int cocos_proj1x1_fwd_f16x3(const float* x, const float* w, const float* bias, float* y,
                            int B, int Cin, int Cout, int N, const float* x_amax, const float* w_amax,
                            cocos_stream_t stream);
int cocos_proj1x1_bwd_partials_f16x3(int B, int Cin, int Cout, int N);   /* 0 on bad dims */
/* K3 (cocos_corr_materialize / _bwd: correspondence.py:291 (+:304) and its autograd) on the same split GEMM;
 * q_amax, k_amax, df_amax: device-side max|.| of the operands (NULL = O(1)). */
int cocos_corr_materialize_f16x3(const float* qn, const float* kn, float* f, int B, int K, int Nq, int Nk,
                                 float scale, const float* q_amax, const float* k_amax, cocos_stream_t stream);
int cocos_corr_materialize_bwd_f16x3(const float* qn, const float* kn, const float* df, float* dqn, float* dkn,
                                     int B, int K, int Nq, int Nk, float scale, const float* q_amax,
                                     const float* k_amax, const float* df_amax, cocos_stream_t stream);
int cocos_proj1x1_bwd_f16x3(const float* x, const float* w, const float* dy, float* dx, float* dw_p,
                            int B, int Cin, int Cout, int N, const float* x_amax, const float* w_amax,
                            const float* dy_amax, cocos_stream_t stream);
/* K0 streaming form (correspondence.py:272,:282 at the reference's shapes: 256 or 256+151 channels -> 256):
 *     y[b,m,n] = (sum_k A[m,k] x[b,k,n]) / (*a_scale_dev * x_scale) + bias[m]
 * A = the weight (forward) or its transpose (input gradient) as f16 hi/lo planes [M][Kpad], zero beyond K, from
 * cocos_split_f16_ex(transpose = 1, Cpad = Kpad) of the k-major matrix [K][M] (its scale goes to a_scale_dev);
 * Kpad = cocos_proj1x1_stream_kpad(K) (0: K not supported).  x fp32 [B,K,N] is split in flight with the
 * power-of-two scale from x_amax (NULL = 1).  A stays in registers, x and y are touched once (HBM-bound).
 * COCOS_ERR_UNSUPPORTED unless K <= 416 and N % 64 == 0 (callers then use cocos_proj1x1_fwd_f16x3 / _bwd_f16x3). */
int cocos_proj1x1_stream_kpad(int K);
int cocos_proj1x1_stream_f16x3(const float* x, const void* a_hi, const void* a_lo, const float* a_scale_dev,
                               const float* bias, float* y, int B, int K, int M, int N, const float* x_amax,
                               cocos_stream_t stream);
/* K23 (round 6): K0 FUSED with K1 — the theta / phi 1x1 projection, centring over the channels and the L2 normalisation in one
 * kernel whose only products are the operand planes of the split correlation kernels (correspondence.py:272 + :277-280,
 * :282 + :287-289; csrc/proj_norm_f16x3.hip).  The fp32 projection never reaches HBM.
 *   cocos_proj_weight_frag_planes: w [256][K] fp32 (+ device cell max|w|) -> the weight's f16 hi / lo planes in the kernel's
 *       fragment order (cocos_proj_weight_frag_bytes(K) bytes: one 16 KB stage per 16 input channels, zero beyond K) and
 *       *w_scale = the power of two they were multiplied with; t_hi / t_lo (nullable pair): the same scaled numbers as transposed
 *       row-major planes [K][256] = the A operand of dx = W^T dy in cocos_proj1x1_stream_f16x3 (scale: *w_scale).  Once per step
 *       and projection.
 *   cocos_proj_center_l2norm_planes_f16x3: up to two projections of the same shape in one launch (theta and phi of a
 *       forward call fill the chip together: two workgroups per CU).  Per projection: x [B,K,N] fp32 (N % 128 == 0),
 *       x_amax (device cell max|x|, NULL = 1), bias [256] (nullable) -> norm [B,N] (the L2 norm of the centred column) and the
 *       planes of plane_scale * y, y = (W x + bias - mean) / (norm + eps): position-major [B,N,256] (always) and
 *       channel-major [B,256,N] (chan_hi / chan_lo, nullable pair — the backward's operands), same layouts and rounding
 *       as cocos_center_l2norm_fwd_planes.  center_over_channels: 1 (PONO_C) or 2 (no centring).
 *       The backward is cocos_center_l2norm_bwd_planes followed by K0's (cocos_proj1x1_stream_f16x3 with the transposed
 *       planes, cocos_proj1x1_dw_f16x3). */
size_t cocos_proj_weight_frag_bytes(int K);
int cocos_proj_weight_frag_planes(const float* w, const float* w_amax_dev, void* wfrag, float* w_scale_dev,
                                  void* t_hi /* nullable */, void* t_lo /* nullable */, int M, int K, cocos_stream_t stream);
int cocos_proj_center_l2norm_planes_f16x3(
    int nprob, const float* x0, const void* wfrag0, const float* w_scale0, const float* bias0, const float* x_amax0, float* norm0,
    void* pos_hi0, void* pos_lo0, void* chan_hi0, void* chan_lo0, const float* x1, const void* wfrag1, const float* w_scale1,
    const float* bias1, const float* x_amax1, float* norm1, void* pos_hi1, void* pos_lo1, void* chan_hi1, void* chan_lo1, int B,
    int K, int N, int center_over_channels, float eps, float plane_scale, cocos_stream_t stream);
/* K24 (round 6): the projection's INPUT gradient fused with what autograd puts between it and the correlation's gradient
 * (csrc/proj_bwd_f16x3.hip; autograd of correspondence.py:272-289 / :272 + :276-280):
 *     d[b,ch,n] = alpha[b,n] in1[b,ch,n] + beta[b,n] in2[b,ch,n] + gamma[b,n],      dx[b,ci,n] = sum_ch W[ch,ci] d[b,ch,n]
 *   mode 0 (match_kernel 1, K1's backward folded in): in1 = d qn [B,256,N], in2a / in2b = the position-major f16 hi / lo planes
 *       [B,N,256] of plane_scale * y (K23's / K1's), c1 = nrm [B,N]; alpha, beta, gamma are computed by a first sweep over the
 *       channels exactly as cocos_center_l2norm_bwd_planes does (center_over_channels 1 or 2);
 *   mode 1 (match_kernel 3, K12's backward and autograd's sum folded in): in1 = d theta from the correlation GEMMs, in2a = the
 *       fp32 projection [B,256,N], c1 = g1, c2 = g2 [B,N] (cocos_unfold3_stats_bwd's maps): alpha = 1, beta = 2 g2, gamma = g1.
 * Outputs per projection: dx [B,Cin,N] fp32; coef [B,3,N] = (alpha, beta, gamma); *amax = max(*amax, max|d|) (mode 0: an upper
 * bound; the cell must hold a finite value >= 0) — the inputs of cocos_proj1x1_dw_affine_f16x3, which rebuilds d while it
 * stages in1 / in2: the fp32 d is neither written nor read.  Up to two projections of one shape per launch; N % 128 == 0,
 * Cin <= 448 (cocos_proj_bwd_input_supported).  wtfrag: cocos_proj_weight_tfrag_planes (cocos_proj_weight_tfrag_bytes() bytes). */
size_t cocos_proj_weight_tfrag_bytes(void);
int cocos_proj_weight_tfrag_planes(const float* w, const float* w_amax_dev, void* wtfrag, float* w_scale_dev /* nullable */, int M,
                                   int Cin, cocos_stream_t stream);
/* cocos_proj_weight_frag_planes + cocos_proj_weight_tfrag_planes for up to two projections of one shape (theta and phi of a
 * forward call) in ONE launch: per projection w [256][K], its max|w| cell, wfrag + w_scale, t_hi / t_lo (nullable pair) and
 * wtfrag (nullable; needs K <= 448).  Same bytes as the single entry points write. */
int cocos_proj_weight_prep_pair(int nprob, const float* w0, const float* w_amax0, void* wfrag0, float* w_scale0, void* t_hi0,
                                void* t_lo0, void* wtfrag0, const float* w1, const float* w_amax1, void* wfrag1, float* w_scale1,
                                void* t_hi1, void* t_lo1, void* wtfrag1, int M, int K, cocos_stream_t stream);
int cocos_proj_bwd_input_supported(int Cin, int M, int N);
int cocos_proj_bwd_input_f16x3(
    int mode, int nprob, const float* in1_0, const void* in2a_0, const void* in2b_0, const float* c1_0, const float* c2_0,
    const void* wtfrag0, const float* w_scale0, float* dx0, float* coef0, float* amax0, const float* in1_1, const void* in2a_1,
    const void* in2b_1, const float* c1_1, const float* c2_1, const void* wtfrag1, const float* w_scale1, float* dx1, float* coef1,
    float* amax1, int B, int Cin, int N, int center_over_channels, float eps, float plane_scale, cocos_stream_t stream);
/* cocos_proj1x1_dw_f16x3 (below) with dy rebuilt on the fly: dy = coef[:,0] in1 + coef[:,1] in2 + coef[:,2].  mode 1: in2a = fp32
 * [B,M,N]; mode 2: in2a / in2b = CHANNEL-major f16 hi / lo planes [B,M,N] of plane_scale * in2.  dy_amax: max|dy| or a bound. */
int cocos_proj1x1_dw_affine_f16x3(int mode, const float* in1, const void* in2a, const void* in2b, const float* coef,
                                  float plane_scale, const float* x, float* ws_dw, float* ws_db, float* dw, float* db, int B,
                                  int C, int M, int N, const float* dy_amax, const float* x_amax, cocos_stream_t stream);
/* The same for TWO projections of one shape (theta and phi of a step) in one launch: half as many, twice as long position chunks
 * per projection (cocos_proj1x1_dw_partials_pair_f16x3 partial tiles each), one reduction launch for both. */
int cocos_proj1x1_dw_partials_pair_f16x3(int B, int C, int M, int N);   /* 0 on bad dims */
int cocos_proj1x1_dw_affine_pair_f16x3(
    int mode, float plane_scale, const float* in1_0, const void* in2a_0, const void* in2b_0, const float* coef0, const float* x0,
    float* ws_dw0, float* ws_db0, float* dw0, float* db0, const float* dy_amax0, const float* x_amax0, const float* in1_1,
    const void* in2a_1, const void* in2b_1, const float* coef1, const float* x1, float* ws_dw1, float* ws_db1, float* dw1, float* db1,
    const float* dy_amax1, const float* x_amax1, const float* plane_scale_dev0 /* nullable */, const float* plane_scale_dev1 /* nullable */,
    int B, int C, int M, int N, cocos_stream_t stream);   /* plane_scale_dev_i (mode 2): device cell with the planes' scale, overrides plane_scale */
/* K25 (round 6, match_kernel 3; csrc/proj_norm_f16x3.hip): the theta / phi projection with NO normalisation, straight to what the
 * match_kernel-3 family reads (correspondence.py:272, :282 + :276-280 / :286-289 without the unfold): per projection x [B,K,N] ->
 *   sum1 = sum_c y, sum2 = sum_c y^2 [B,N] (y = W x + bias) — cocos_unfold3_stats_finish_pair makes mu / a / nrm of both tensors from
 *       them in one launch (cocos_unfold3_stats_fwd's second half);
 *   the f16 hi / lo planes of s * y, position-major [B,N,256] (the x-box GEMM's operands) and channel-major [B,256,N] (nullable pair:
 *       the backward GEMMs' and the weight gradient's operands), *y_scale = s: a power of two from the a-priori bound
 *       K max|w| max|x| + max|bias| (no pass over y; the bound lands in [2^13, 2^14)).  The fp32 projection is never written.
 * Backward: cocos_unfold3_stats_bwd_maps_pair (g1, g2 of both tensors), cocos_proj_bwd_input_planes_f16x3 (K24 mode C: d = in1 +
 * 2 g2 y + g1 with y read back from the position-major planes, dx = W^T d), cocos_proj1x1_dw_affine_pair_f16x3 (mode 2, plane_scale_dev). */
int cocos_proj_raw_planes_stats_f16x3(
    int nprob, const float* x0, const void* wfrag0, const float* w_scale0, const float* bias0, const float* x_amax0, float* sum1_0,
    float* sum2_0, float* y_scale0, void* pos_hi0, void* pos_lo0, void* chan_hi0, void* chan_lo0, const float* x1, const void* wfrag1,
    const float* w_scale1, const float* bias1, const float* x_amax1, float* sum1_1, float* sum2_1, float* y_scale1, void* pos_hi1,
    void* pos_lo1, void* chan_hi1, void* chan_lo1, int B, int K, int N, cocos_stream_t stream);
int cocos_unfold3_stats_finish_pair(const float* s1_0, const float* s2_0, float* mu0, float* a0, float* nrm0, const float* s1_1,
                                    const float* s2_1, float* mu1, float* a1, float* nrm1, int B, int h, int w, float k_unfolded,
                                    float eps, cocos_stream_t stream);
int cocos_unfold3_stats_bwd_maps_pair(const float* mu0, const float* a0, const float* nrm0, const float* dmu0 /* nullable */,
                                      const float* da0 /* nullable */, float* ws0 /* 2*B*h*w */, const float* mu1, const float* a1,
                                      const float* nrm1, const float* dmu1 /* nullable */, const float* da1 /* nullable */,
                                      float* ws1, int B, int h, int w, float k_unfolded, cocos_stream_t stream);
int cocos_proj_bwd_input_planes_f16x3(
    int nprob, const float* in1_0, const void* y_hi0, const void* y_lo0, const float* y_scale0, const float* g1_0, const float* g2_0,
    const void* wtfrag0, const float* w_scale0, float* dx0, float* coef0, float* amax0, const float* in1_1, const void* y_hi1,
    const void* y_lo1, const float* y_scale1, const float* g1_1, const float* g2_1, const void* wtfrag1, const float* w_scale1, float* dx1,
    float* coef1, float* amax1, int B, int Cin, int N, cocos_stream_t stream);
/* K0 weight and bias gradient as one streaming reduction (autograd of correspondence.py:272,:282):
 *     dw[m,c] = sum_{b,n} dy[b,m,n] x[b,c,n]      db[m] = sum_{b,n} dy[b,m,n]   (db, ws_db: both or neither NULL)
 * dy [B,M,N], x [B,C,N] fp32, read once; f16x3 products with the power-of-two scales from dy_amax / x_amax.
 * S = cocos_proj1x1_dw_partials_f16x3(B, C, M, N) partial tiles go through the caller's workspace
 * (ws_dw: S*M*roundup(C,32) floats, ws_db: S*M floats) and are summed on the device; S = 0: shape not supported
 * (needs M <= 256, C <= 448, N % 4 == 0) -> cocos_proj1x1_bwd_f16x3. */
int cocos_proj1x1_dw_partials_f16x3(int B, int C, int M, int N);
int cocos_proj1x1_dw_f16x3(const float* dy, const float* x, float* ws_dw, float* ws_db, float* dw, float* db,
                           int B, int C, int M, int N, const float* dy_amax, const float* x_amax,
                           cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K6  match_kernel = 3 without unfolding (correspondence.py:276-280,:286-289 + :291 + :304, PONO_C):
 *     f[b,p,q] = scale * ( sum_{d in 3x3, p+d and q+d inside} c_raw[b,p+d,q+d] - k_unfolded*mu[b,p]*nu[b,q] )
 *                      * a[b,p] * b[b,q]
 *   c_raw [B,N,N] = theta_b^T phi_b (K = 256, cocos_corr_materialize), N = h*w;
 *   mu, a / nu, b [B,N]: mean and 1/(norm+eps) of the unfolded, centred theta / phi vectors
 *   (3x3 box sums of per-position channel sums — computed by the caller); k_unfolded = 256*9.
 *   bwd: g = dL/df ->  dc_raw = boxdiag(g * a_p * b_q * scale)  and the four reductions
 *        r1[p] = sum_q g b_q nu_q, r2[p] = sum_q g f, c1[q] = sum_p g a_p mu_p, c2[q] = sum_p g f
 *        from which dmu = -k*scale*a*r1, da = r2/a, dnu = -k*scale*b*c1, db = c2/b.
 * ------------------------------------------------------------------------------------- */
int cocos_box3_logits_fwd(const float* c_raw, const float* mu, const float* nu, const float* a,
                          const float* b, float* f, int B, int h, int w, float k_unfolded,
                          float scale, cocos_stream_t stream);
size_t cocos_box3_logits_bwd_workspace_bytes(int B, int h, int w);
int cocos_box3_logits_bwd(const float* g, const float* f, const float* mu, const float* nu,
                          const float* a, const float* b, float* dc_raw, float* r1, float* r2,
                          float* c1, float* c2, void* ws, size_t ws_bytes, int B, int h, int w,
                          float scale, cocos_stream_t stream);
/* Same, and on return *dc_amax_inout = max(*dc_amax_inout, max|dc_raw|) (the cell must hold a finite value >= 0): the
 * scale source of the K3 backward that consumes dc_raw, produced while it is written. */
int cocos_box3_logits_bwd_amax(const float* g, const float* f, const float* mu, const float* nu,
                               const float* a, const float* b, float* dc_raw, float* r1, float* r2,
                               float* c1, float* c2, void* ws, size_t ws_bytes, int B, int h, int w,
                               float scale, float* dc_amax_inout, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K7  softmax + warp from MATERIALISED, KEY-MAJOR logits (correspondence.py:307 + :318/:334/... and
 *     their autograd in one streaming kernel each; P never reaches HBM).  Used by match_kernel = 3.
 *   logits_t [B,Nk,Nq]: logits_t[b,j,i] = f[b,i,j] / temperature
 *   fwd: out[b,c,i] = sum_j softmax_j(logits_t[b,j,i]) v[b,c,j];  lse [B,Nq]
 *   bwd: dlogits_t[b,j,i] = P[i,j] * (sum_c dout[b,c,i] v[b,c,j] - sum_c dout[b,c,i] out[b,c,i])
 * ------------------------------------------------------------------------------------- */
int cocos_logits_softmax_warp_fwd(const float* logits_t, const float* v, float* out, float* lse,
                                  int B, int Nq, int Nk, int Cv, cocos_stream_t stream);
int cocos_logits_softmax_warp_bwd(const float* logits_t, const float* v, const float* out,
                                  const float* lse, const float* dout, float* dlogits_t,
                                  int B, int Nq, int Nk, int Cv, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K8  WTA_scale (correspondence.py:38-77, applied :300-303) fused with the /temperature of :304,
 *     on a materialised [rows, cols] correlation (rows = B*Nq):
 *   fwd: y = (x == rowmax(x) ? x : x*scale) * post_scale;  mask = one BIT per element (x == rowmax),
 *        64 columns per 64-bit word, cocos_wta_scale_mask_bytes(rows, cols) bytes (-1 on bad dims)
 *   bwd: dx = dy * post_scale * (mask ? 1 : 1e-4)   — the reference's hard-coded 1e-4 (:72)
 * ------------------------------------------------------------------------------------- */
long long cocos_wta_scale_mask_bytes(long long rows, int cols);
int cocos_wta_scale_fwd(const float* x, float* y, void* mask, long long rows, int cols,
                        float scale, float post_scale, cocos_stream_t stream);
int cocos_wta_scale_bwd(const float* dy, const void* mask, float* dx, long long rows, int cols,
                        float post_scale, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K9  PONO + SPADE modulation + LeakyReLU, fused (`--PONO` networks):
 *     PositionalNorm2d (normalization.py:63-68) -> SPADE.forward's `normalized * (1 + gamma) + beta`
 *     (normalization.py:148-151) -> the LeakyReLU(0.2) of SPADEResnetBlock.actvn (architecture.py:88-95)
 *   x, gamma, beta, y [B,C,N] fp32 (N = H*W);  C >= 2 (unbiased variance over channels)
 *   fwd: xn = (x - mean_C) / sqrt(var_C + eps);  z = xn*(1+gamma) + beta;  y = z > 0 ? z : slope*z
 *        (slope = 1: no activation, as for norm_s)
 *   bwd: dx, dgamma, dbeta from dy (any of the three may be NULL); statistics are recomputed from x.
 * ------------------------------------------------------------------------------------- */
int cocos_pono_spade_fwd(const float* x, const float* gamma, const float* beta, float* y,
                         int B, int C, int N, float eps, float slope, cocos_stream_t stream);
int cocos_pono_spade_bwd(const float* x, const float* gamma, const float* beta, const float* dy,
                         float* dx, float* dgamma, float* dbeta,
                         int B, int C, int N, float eps, float slope, cocos_stream_t stream);
/* Round 6: the same passes also leaving max|.| of the tensors the next convolutions split (*cell = max(*cell, max|.|); cells holding
 * finite values >= 0): the forward's y — the input of conv_0 / conv_1 / conv_s (architecture.py:88-95) — and the backward's dgamma /
 * dbeta — the output gradients of SPADE's mlp_gamma / mlp_beta (normalization.py:121-127): amax2 = [max|dgamma|, max|dbeta|].
 * amax_partials = workspace of cocos_pono_spade_amax_partials(B, C, N) floats (forward) resp. twice that (backward). */
int cocos_pono_spade_amax_partials(int B, int C, int N);
int cocos_pono_spade_fwd_amax(const float* x, const float* gamma, const float* beta, float* y, float* y_amax_inout_dev,
                              float* amax_partials, int B, int C, int N, float eps, float slope, cocos_stream_t stream);
int cocos_pono_spade_bwd_amax(const float* x, const float* gamma, const float* beta, const float* dy,
                              float* dx, float* dgamma, float* dbeta, float* amax2_inout_dev, float* amax_partials,
                              int B, int C, int N, float eps, float slope, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K11 nearest-neighbour up-sampling of the warped image (self.upsampling = nn.Upsample(scale_factor=down),
 *     correspondence.py:188, applied at :327) and its backward.  x [planes,h,w] -> y [planes,h*scale,w*scale]
 *     (planes = B*C);  bwd: dx[p,y,x] = sum of the scale x scale block of dy.  (w*scale) % 4 == 0.
 * ------------------------------------------------------------------------------------- */
int cocos_upsample_nearest_fwd(const float* x, float* y, int planes, int h, int w, int scale, cocos_stream_t stream);
int cocos_upsample_nearest_bwd(const float* dy, float* dx, int planes, int h, int w, int scale, cocos_stream_t stream);
/* The head of the first row pass (csrc/warp_head.hip; correspondence.py:327, :334 and their autograd).
 *   cocos_warp_head_fwd: y [B,Ci,h*down,w*down] = nearest up-sampling of channels [0,Ci) of o [B,C,h,w] ((w*down) % 4 == 0).
 *   cocos_warp_head_bwd: g_img [B,Ci,h*down,w*down] = d loss / d warp_out, g_mask [B,Cs,h,w] = d loss / d warp_mask,
 *       o [B,Ci+Cs,h,w] -> dout [B,Ci+Cs,h,w] (window sums | copy), drow [B,h*w] = sum_c dout * o accumulated in fp64 (the D of
 *       the softmax backward), *amax_inout = max(*amax_inout, max|dout|) — one pass instead of cocos_upsample_nearest_bwd +
 *       cocos_concat2_amax + cocos_rowdot_f64.  w % 4 == 0, tensors 16-byte aligned. */
int cocos_warp_head_fwd(const float* o, float* y, int B, int Ci, int C, int h, int w, int down, cocos_stream_t stream);
int cocos_warp_head_bwd(const float* g_img, const float* g_mask, const float* o, float* dout, float* drow, float* amax_inout_dev,
                        int B, int Ci, int Cs, int h, int w, int down, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K14  value tensor of the first row pass in one kernel (correspondence.py:314, :318-319, :331-334):
 *     out[b, 0:Ci]     = F.avg_pool2d(img, down)                          img [B,Ci,H,W]
 *     out[b, Ci:Ci+Cs] = F.interpolate(seg, scale_factor=1/down, 'nearest') = seg[.., y*down, x*down]
 *   out [B, Ci+Cs, H/down, W/down] (= torch.cat of the two); either part may be absent (count 0, NULL pointer).
 *   Forward only (the exemplar image and its label map are data).  H, W multiples of down. */
int cocos_warp_values(const float* img, const float* seg, float* out, int B, int Ci, int Cs, int H, int W,
                      int down, cocos_stream_t stream);
/* Same, and *amax_inout_dev = max(*amax_inout_dev, max|out|): the scale source of the K2 forward's f16 split of V comes out
 * of the kernel that writes V. */
int cocos_warp_values_amax(const float* img, const float* seg, float* out, int B, int Ci, int Cs, int H, int W,
                           int down, float* amax_inout_dev, cocos_stream_t stream);

/* Plane preparation folded into kernels that already touch the data (csrc/plane_prep.hip):
 *   cocos_concat2_amax: out[b] = [a[b] | b[b]] (torch.cat of two [B, *] fp32 tensors along dim 1; na, nb = elements per
 *       sample, multiples of 4) and *amax_inout_dev = max(., max|out|) — the backward of the row pass's output split, whose
 *       result is the `dout` of the K2 backward.
 *   cocos_split_f16_chan_mask: cocos_split_f16_ex(transpose = 0, amax_dev) + cocos_f16_plane_block_mask(lo) in one launch.
 *   cocos_proj_weight_planes: the f16 hi/lo planes of a 1x1-projection weight w [Cout][Cin] in both orientations —
 *       rows [Cout][KpadIn] (= cocos_split_f16_rows) and, when t_hi/t_lo are given, transposed [Cin][KpadOut]
 *       (= cocos_split_f16_ex(transpose = 1, Cpad = KpadOut)) — scaled by the power of two from *amax_dev. */
int cocos_concat2_amax(const float* a, const float* b, float* out, int B, long long na, long long nb,
                       float* amax_inout_dev /* nullable */, cocos_stream_t stream);
int cocos_split_f16_chan_mask(const float* x, void* hi, void* lo, int B, int C, int N, const float* amax_dev,
                              float* scale_out_dev /* nullable */, unsigned* mask_inout_dev /* nullable */,
                              cocos_stream_t stream);
int cocos_proj_weight_planes(const float* w, void* rows_hi, void* rows_lo, void* t_hi /* nullable */,
                             void* t_lo /* nullable */, int Cout, int Cin, int KpadIn, int KpadOut, const float* amax_dev,
                             float* scale_out_dev /* nullable */, cocos_stream_t stream);

/* Small reductions that used to be framework calls inside the autograd Functions:
 *   cocos_sum_leading: out[i] = sum_s x[s][i] over [S][n] partial tiles (weight-gradient partials of the general GEMMs);
 *   cocos_channel_sum: db[c] = sum_{b,n} dy[b][c][n] (the bias gradient of a convolution, dy.sum((0,2,3)));
 *   cocos_box3_stat_grads: dmu = -k s a r1, dnu = -k s b c1, da = r2 / a, db = c2 / b from K6's row / column sums. */
int cocos_sum_leading(const float* x, float* out, int S, long long n, cocos_stream_t stream);
int cocos_channel_sum_slices(int C, long long N);   /* S: `partials` of cocos_channel_sum holds S * C floats */
int cocos_channel_sum(const float* dy, float* db, float* partials, int B, int C, long long N, cocos_stream_t stream);
int cocos_box3_stat_grads(const float* r1, const float* r2, const float* c1, const float* c2, const float* a, const float* b,
                          float* dmu, float* dnu, float* da, float* db, long long n, float k_unfolded, float scale,
                          cocos_stream_t stream);

/* K7 on the f16 MFMA (same contract as cocos_logits_softmax_warp_fwd / _bwd; operand planes as for K2's split flavour):
 *   fwd: vh,vl [B,Cv,Nk] channel-major planes of s_v*v, s_v = *v_scale_dev (NULL = 1; undone in the epilogue); Nk % 4 == 0
 *   bwd: vph,vpl [B,Nk,CvPad] and gph,gpl [B,Nq,CvPad] position-major planes of s_v*v and of (*g_scale_dev)*dout
 *        (cocos_split_f16_ex), out/dout fp32 for D; writes dlogits_t fp32. */
int cocos_logits_softmax_warp_fwd_f16x3(const float* logits_t, const void* vh, const void* vl, float* out, float* lse,
                                        const float* v_scale_dev /* nullable */, int B, int Nq, int Nk, int Cv,
                                        cocos_stream_t stream);
int cocos_logits_softmax_warp_bwd_f16x3(const float* logits_t, const void* vph, const void* vpl, const void* gph,
                                        const void* gpl, const float* g_scale_dev, const float* v_scale_dev /* nullable */,
                                        const float* out, const float* dout, const float* lse, float* dlogits_t, int B,
                                        int Nq, int Nk, int Cv, int CvPad, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K15 row reductions of the contextual loss (ContextualLoss_forward.forward,
 *     models/networks/ContextualLoss.py:121-133, after its cosine-similarity matmul = cocos_corr_materialize):
 *       d = 1 - cos;  d_norm = d / (min_j d + eps);  w = exp((1 - d_norm) / h);  A = w / sum_j w;  cx[i] = max_j A[i,j]
 *     = 1 / sum_j exp((cos[i,j] - m_i) s_i),  m_i = max_j cos[i,j],  s_i = 1 / (h (1 - m_i + eps)).
 *   cosm [rows, cols] fp32 (rows = B * positions of X), 1 <= cols <= 4096; one read of the matrix forward, one read +
 *   one write backward (row statistics recomputed).  bwd: dcos[i,j] = dcx[i] * d cx[i] / d cos[i,j], including the
 *   path through the row maximum (first index on ties, like torch.min's gradient).
 * ------------------------------------------------------------------------------------- */
int cocos_contextual_rows_fwd(const float* cosm, float* cx, long long rows, int cols, float h, float eps,
                              cocos_stream_t stream);
int cocos_contextual_rows_bwd(const float* cosm, const float* dcx, float* dcos, long long rows, int cols, float h,
                              float eps, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K22 the contextual loss WITHOUT its [N, N] matrices (round 5; replaces cocos_corr_materialize + K15 inside
 *     ContextualLoss_forward.forward, models/networks/ContextualLoss.py:93-137, for any N and C):
 *       cx[i] = 1 / S_i,  S_i = sum_j exp((cos_ij - m_i) tau_i),  m_i = max_j cos_ij,  tau_i = 1 / (h (1 - m_i + eps))
 *     forward: one launch, a workgroup owns 128 queries and sweeps the keys twice (maximum, then sums); each sweep is a K = C
 *     GEMM on the f16 MFMA with split operands (three terms, fp32 accumulate), the cosine tile never leaves the registers.
 *     backward: G_ij = a_i e_ij (+ a term on column argmax_j only, which the caller applies as a gather / scatter-add):
 *       d Xn[:, i] = sum_j G_ij Yn[:, j],   d Yn[:, j] = sum_i G_ij Xn[:, i]
 *     — one launch per side; per 128 x 128 tile the cosines are recomputed, exponentiated, written to LDS as f16 hi / lo planes
 *     and multiplied with the other side's channel-major planes (256 output channels per workgroup).
 *   Planes: position-major [B][Np][Kp] f16 hi / lo of the normalised features times a power-of-two device scale
 *   (cocos_split_f16_ex), Np % 128 == 0, Kp % 32 == 0, zero beyond the real positions / channels; channel-major value planes
 *   [B][Cv][Nip].  All statistics fp32 [B][N].
 * ------------------------------------------------------------------------------------- */
int cocos_contextual_cx_fwd_f16x3(const void* xh, const void* xl, const void* yh, const void* yl, const float* x_scale_dev,
                                  const float* y_scale_dev, float* m_out /* max_j cos */, float* s_out /* S */,
                                  float* u_out /* sum_j e_ij (cos_ij - m_i) */, int* j_out /* argmax_j, first */, int B, int Nq,
                                  int Nk, int Nqp, int Nkp, int Kp, float h, float eps, cocos_stream_t stream);
/* out[b][ch][r] = host_scale * (*mul_dev) * beta[r] * sum_c (alpha[c] / *alpha_div_dev) exp2((cos_rc - m) t) V[ch][c]
 *                 (+ gather_coef[r] * gather_src[b][ch][gather_idx[r]]: the argmax column's term of d Xn, from the fp32 values);
 * (m, t = tau log2 e) per QUERY: indexed by r when stats_on_rows (rows = queries, inner = keys: d Xn), by c otherwise (rows = keys,
 * inner = queries: d Yn).  |alpha / alpha_div| <= 1.  alpha, alpha_div_dev, beta, mul_dev, gather_* nullable. */
int cocos_contextual_cx_bwd_f16x3(const void* rh, const void* rl, const void* ih, const void* il, const void* vh, const void* vl,
                                  const float* r_scale_dev, const float* i_scale_dev, const float* v_scale_dev,
                                  const float* mul_dev /* nullable */, const float* m, const float* t,
                                  const float* alpha /* nullable */, const float* alpha_div_dev /* nullable */,
                                  const float* beta /* nullable */, const float* gather_src /* nullable: [B][Cv][Ni] */,
                                  const int* gather_idx, const float* gather_coef, float* out, int B, int Nr, int Ni, int Nrp,
                                  int Nip, int Kp, int Cv, int stats_on_rows, float host_scale, cocos_stream_t stream);
/* The backward's per-query coefficients (n = B * Nq) from the forward's statistics and d loss / d cx:  a = dS tau with
 * dS = -dcx / S^2, tau = 1 / (h (1 - m + eps));  t2 = tau log2 e;  extra = dS (h tau^2 U - tau S);  *a_amax_dev = max |a| (zero on entry). */
int cocos_contextual_cx_coeffs(const float* dcx, const float* S, const float* U, const float* m, float* a_out, float* t2_out,
                               float* extra_out, float* a_amax_dev, long long n, float h, float eps, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K18 nn.ReflectionPad2d(pad) and its backward: the pad in front of the 3x3 convolutions of ResidualBlock
 *     (correspondence.py:19,:23), SPADE (normalization.py:118,:146) and SPADEResnetBlock (architecture.py:30).
 *   fwd: x [planes,H,W] -> y [planes,H+2pad,W+2pad];  bwd: dy -> dx as a gather (no atomics).  pad < H, W.
 * ------------------------------------------------------------------------------------- */
int cocos_reflect_pad2d_fwd(const float* x, float* y, long long planes, int H, int W, int pad, cocos_stream_t stream);
int cocos_reflect_pad2d_bwd(const float* dy, float* dx, long long planes, int H, int W, int pad, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K17 SPADE modulation + LeakyReLU behind any parameter-free norm (the non-PONO branch of SPADE, normalization.py
 *     :93-101 instance | syncbatch | batch, then :148 and architecture.py:88-95): xh = the normalised activations,
 *     gamma, beta, y and the gradients all hold n fp32 elements (same shape).
 *   fwd: y = leaky_relu(xh*(1+gamma) + beta, slope)       bwd: dxh, dgamma, dbeta (any may be NULL), z recomputed
 * ------------------------------------------------------------------------------------- */
int cocos_spade_modulate_fwd(const float* xh, const float* gamma, const float* beta, float* y, long long n, float slope,
                             cocos_stream_t stream);
int cocos_spade_modulate_bwd(const float* xh, const float* gamma, const float* beta, const float* dy, float* dxh,
                             float* dgamma, float* dbeta, long long n, float slope, cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K16 2-D convolution (cross-correlation, zero padding, like torch.nn.functional.conv2d with groups = 1, one stride /
 *     padding / dilation for both axes)
 *     as an implicit GEMM on the f16 MFMA with split operands (conv_f16x3.hip).  Replaces the nn.Conv2d calls of the
 *     networks in front of and behind the path: ResidualBlock (correspondence.py:13-36: ReflectionPad2d(1) stays a
 *     host-side pad, the 3x3 convolution is this kernel with pad = 0), the adaptor layers (correspondence.py:150-173)
 *     and the k4 s2 layers of the PatchGAN (discriminator.py:92-115).
 *   The contraction index runs over blocks of 32 channels, taps inside:  k = ((ci/32) * T + ky*KW + kx) * 32 + ci%32,
 *   T = KH*KW, channels zero-padded to Cp = 32*ceil(Cin/32), K = cocos_conv2d_kdim(Cin,KH,KW) = T*Cp (a k-block of 32
 *   is 32 channels at one tap; consecutive k-blocks re-read the same channels).
 *   fwd:   x [B,Cin,H,W] fp32, weight planes w_hi/w_lo [K/32][Cout][32] f16 (k-block major so that a tile's rows are
 *          contiguous; zero for ci >= Cin): cocos_split_f16_rows of the re-laid-out weight matrix, whose *scale_out
 *          goes to w_scale_dev (NULL = 1), x_amax_dev = max|x| (NULL: x is O(1)), bias [Cout] or NULL
 *          ->  y [B,Cout,OH,OW] fp32, OH = cocos_conv2d_out_size(H,KH,stride,pad,dilation) (0 when the kernel does not fit).
 *          The input gradient of a stride-1 convolution is the same call on dy with the planes of the flipped,
 *          transposed weight (roles of Cin and Cout swapped) and pad' = dilation*(K-1)-pad; of a strided one
 *          (dilation 1) stride^2 calls of cocos_conv2d_fwd_scatter_f16x3, one per parity class.
 *   wgrad: x, dy [B,Cout,OH,OW] (+ their max|.|, NULL = O(1))  ->  partials [S][Cout][K] fp32 (k as above; entries
 *          with ci >= Cin are written as zeros), S = cocos_conv2d_wgrad_slices(...) slices over the B*OH*OW
 *          positions; the caller sums them (and sums dy over (b,oy,ox) for the bias gradient).
 *   Every tensor (and Cout*K*4) must stay below 2 GiB (32-bit offsets).
 * ------------------------------------------------------------------------------------- */
int cocos_conv2d_out_size(int in, int k, int stride, int pad, int dilation);
int cocos_conv2d_kdim(int Cin, int KH, int KW);
int cocos_conv2d_fwd_f16x3(const float* x, const void* w_hi, const void* w_lo, const float* w_scale_dev /* nullable */,
                           const float* x_amax_dev /* nullable */, const float* bias /* nullable */, float* y,
                           int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int dilation,
                           cocos_stream_t stream);
/* One parity class of the input gradient of a STRIDED convolution.  For stride s, dx[y,x] only receives the taps with
 * ky = (y+p) mod s (mod s), kx likewise: the pixels of one class (ry,rx) form a grid on which the gradient is a
 * stride-1 convolution of dy with the JH x JW sub-kernel w[:, :, ry::s, rx::s] (flipped, channel roles swapped; planes
 * as for cocos_conv2d_fwd_f16x3).  This entry runs that convolution on x = dy [B,Cin,H,W] with separate row / column
 * padding, an explicit output grid OHo x OWo (reads beyond x are zero) and writes output pixel (oy,ox) of plane (b,m)
 * to y[(b*Cout + m)*y_plane + y_offset + oy*y_pitch + ox*y_col_stride]  (elements). */
int cocos_conv2d_fwd_scatter_f16x3(const float* x, const void* w_hi, const void* w_lo, const float* w_scale_dev,
                                   const float* x_amax_dev, float* y, int B, int Cin, int H, int W, int Cout, int JH, int JW,
                                   int pad_y, int pad_x, int OHo, int OWo, long long y_plane, int y_pitch, int y_col_stride,
                                   long long y_offset, cocos_stream_t stream);
/* The two pieces of host-side glue around the calls above as one launch each:
 *   cocos_conv2d_weight_planes: weight [Cout,Cin,KH,KW] fp32 -> the planes of cocos_conv2d_fwd_f16x3 (mode 0) or of the
 *       input-gradient calls (mode 1: channel roles swapped, sub-kernel w[:, :, ry::s, rx::s] of JH x JW taps, flipped;
 *       stride-1 layers: ry = rx = 0, s = 1, J = K), scaled by the power of two from *amax_dev (-> *scale_out_dev).
 *   cocos_conv2d_wgrad_reduce: partials [S][Cout][K] of cocos_conv2d_wgrad_f16x3 -> dw [Cout,Cin,KH,KW]. */
int cocos_conv2d_weight_planes(const float* w, void* hi, void* lo, int Cout, int Cin, int KH, int KW, int mode, int JH, int JW,
                               int ry, int rx, int s, const float* amax_dev /* nullable */,
                               float* scale_out_dev /* nullable */, cocos_stream_t stream);
int cocos_conv2d_wgrad_reduce(const float* partials, float* dw, int S, int Cout, int Cin, int KH, int KW,
                              cocos_stream_t stream);
int cocos_conv2d_wgrad_slices(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                              int dilation);   /* 0 on bad dims */
int cocos_conv2d_wgrad_f16x3(const float* x, const float* dy, const float* x_amax_dev /* nullable */,
                             const float* dy_amax_dev /* nullable */, float* partials, int B, int Cin, int H, int W,
                             int Cout, int KH, int KW, int stride, int pad, int dilation, cocos_stream_t stream);
/* One-term bf16 flavour of K16 (BASELINE config 3: "bf16 MFMA" for the SPADE generator / PatchGAN convolutions): single
 * bf16 operand planes, ONE v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate, no scales (bf16 has fp32's range).
 *   forward / input gradient: cocos_conv2d_fwd_f16x3 / _fwd_scatter_f16x3 with w_lo = NULL and w_hi = the bf16 plane written
 *       by cocos_conv2d_weight_planes(mode | 2) (lo may be NULL there; scale pointers ignored);
 *   weight gradient: cocos_conv2d_wgrad_bf16 (partials as for the f16x3 entry point).
 * 8 mantissa bits per operand: measured error vs fp64 in tests/test_gpu_conv.py; never used upstream of the correlation. */
int cocos_conv2d_wgrad_bf16(const float* x, const float* dy, float* partials, int B, int Cin, int H, int W, int Cout, int KH,
                            int KW, int stride, int pad, int dilation, cocos_stream_t stream);

/* K19 / K20 — match_kernel = 3 (the reference's DEFAULT, options/base_options.py:70) fused, round 3: replaces
 * F.unfold(k=3) + centre + normalise (correspondence.py:276-280, :286-289), the K = 2304 matmul (:291), /temperature (:304),
 * softmax (:307) and the warp matmul (:318 / :334) — and their autograd — for --PONO_C on a 64- or 128-wide feature grid, without a
 * [B,2304,HW] tensor, without the box-filtered logits matrix and without saved logits.  Decomposition (csrc/box3_common.h):
 *     S[p,q] = sum over the 3x3 offsets d of C[p+d, q+d]   (C = the K = 256 correlation; zero outside the grid)
 *            = ybox(xbox(C)),   xbox along a row (both indices +-1), ybox across rows (both indices +-w)
 *     z[p,q] = scale * a_p * b_q * (S[p,q] - kc * mu_p * nu_q),  kc = 9*256,
 *   mu / a (queries), nu / b (keys): mean and 1 / (norm + eps) of the unfolded, centred vectors (cocos_unfold3_stats_*).
 *   cocos_box3_corr_xbox_f16x3: T = xbox(C) from POSITION-major f16 hi/lo planes of the raw theta / phi ([B][N][K],
 *       scaled by *k_scale_dev / *q_scale_dev as written by cocos_split_f16_ex), keys in the rows.  T is TILE-BLOCKED:
 *       [B][Nk/32][Nq/32] blocks of 4 KB = [g][lane][4] — the accumulator image of a 32x32 MFMA tile (registers 4g..4g+3 of a
 *       lane = keys 8g + 4*(lane>>5) + 0..3, lane & 31 = query), B*Nk*Nq floats in all.
 *   cocos_box3_softmax_warp_fwd_f16x3: out[b,c,p] = sum_q softmax_q(z[p,q]) v[b,c,q], lse [B,Nq]; reads three blocks of T per
 *       tile (the y box) and nothing else of size HWxHW.  v planes channel-major [B,Cv,Nk] (cocos_split_f16_ex).
 *   cocos_box3_softmax_warp_bwd_f16x3: with L = dloss/dz = P * (V.dout - sum_c dout*out) it writes
 *       g_blocked = L * scale * a_p * b_q (= dloss / d ybox(T)[p,q], blocked like T), dmu, da [B,Nq], dnu, db [B,Nk] (column
 *       sums: per-workgroup partials in `colpart`, cocos_box3_softmax_warp_bwd_colpart_bytes, folded by a second launch),
 *       max|g| into *gmax_dev (a zeroed cell on entry; atomic max), and — when psh/psl are given (V is differentiated: the
 *       cycle terms) — the planes of 2^14 P in the [Nq/32][Nk/32] blocked orientation of cocos_hgemm_f16x3 mode 2.
 *       vph/vpl [B,Nk,CvPad], gph/gpl [B,Nq,CvPad]: position-major planes of s_v*v and s_o*dout as for K2 / K7.
 *   cocos_box3_adjoint_planes_f16x3: dC = xbox(ybox(g)) (the filter is self-adjoint) as f16 hi/lo planes scaled by the power of
 *       two written to *scale_out_dev (from *gmax_dev), in [Nq/32][Nk/32] blocks of 2 x [32 queries][16 keys]; then
 *           d theta_raw [256][Nq] = hgemm(A = channel-major planes of phi_raw,  B = dC planes, b_blocked = 3)
 *           d phi_raw   [256][Nk] = hgemm(A = channel-major planes of theta_raw, B = dC planes, b_blocked = 2)
 *       (b_blocked = 3: the same blocks read with n = queries, k = keys).
 *   cocos_box3_fused_supported: 1 when this family takes the shape (64- or 128-wide grid — every BASELINE configuration: 256^2 at
 *       down 4; 512^2 at down 4 and 256^2 at --warp_stride 2 are 128 x 128 — Nq == Nk == h*w, Nq % 256 == 0, Cv <= 160, one sample's
 *       T below 2 GiB); every other match_kernel-3 shape keeps the K3 -> K6 -> K7 chain.  Round 4: on 128-wide grids the x box
 *       exchanges its halo between the two 64-position chunks of an image row (GEMM epilogue: within a wave for the keys,
 *       between neighbouring waves for the queries; K20: four waves per row pair), the y box spans grid_w / 32 tiles, and the
 *       per-key statistics of samples with more than 4096 keys pass through LDS in a ring of 2048-key chunks.
 *   `flags` (round 4): bit 0 (COCOS_BOX3_T_TRANSPOSED) — t_blocked holds T of the OTHER orientation (keys and queries exchanged:
 *       xbox(C)^T = xbox(C^T)); the kernels read block (qb, kt) for (kt, qb) and transpose it in LDS, so the column pass of
 *       the cycle terms (correspondence.py:338,:351) shares the row pass's T — no second correlation GEMM — and the backward
 *       writes g_blocked in the layout of that stored T.  Bit 1 (COCOS_BOX3_G_ACCUMULATE, backward) — g_blocked already holds
 *       another pass's G over the same T: this pass's G is ADDED to it and *gmax_dev (zeroed on entry) receives max|sum|: one
 *       g tensor, one cocos_box3_adjoint_planes_f16x3 and one pair of GEMMs for all passes over a T. */
#define COCOS_BOX3_T_TRANSPOSED 1
#define COCOS_BOX3_G_ACCUMULATE 2
int cocos_box3_fused_supported(int Nq, int Nk, int Cv, int grid_h, int grid_w);
int cocos_box3_corr_xbox_f16x3(const void* k_hi, const void* k_lo, const void* q_hi, const void* q_lo, float* t_blocked,
                               int batch, int Nk, int Nq, int K, int grid_w, const float* k_scale_dev /* nullable */,
                               const float* q_scale_dev /* nullable */, cocos_stream_t stream);
int cocos_box3_softmax_warp_fwd_f16x3(const float* t_blocked, const float* mu_q, const float* a_q, const float* nu_k,
                                      const float* b_k, const void* vh, const void* vl, float* out, float* lse,
                                      const float* v_scale_dev /* nullable */,
                                      const unsigned* v_lo_mask_dev /* nullable: as for cocos_corr_softmax_warp_fwd_f16x3 */,
                                      int B, int Nq, int Nk, int Cv, int grid_h, int grid_w, float k_unfolded, float scale,
                                      int flags, cocos_stream_t stream);
size_t cocos_box3_softmax_warp_bwd_colpart_bytes(int B, int Nq, int Nk);
int cocos_box3_softmax_warp_bwd_f16x3(const float* t_blocked, const float* mu_q, const float* a_q, const float* nu_k,
                                      const float* b_k, const void* vph, const void* vpl, const void* gph, const void* gpl,
                                      const float* g_scale_dev, const float* v_scale_dev /* nullable */, const float* out,
                                      const float* dout, const float* lse, float* g_blocked, float* dmu, float* da, float* dnu,
                                      float* db, void* colpart, float* gmax_dev, void* psh /* nullable */,
                                      void* psl /* nullable */, const unsigned* v_lo_mask_dev /* nullable */, int B, int Nq,
                                      int Nk, int Cv, int CvPad, int grid_h, int grid_w, float k_unfolded, float scale,
                                      const float* d_pre /* nullable: D = sum_c dout * out [B][Nq] (cocos_rowdot_f64) */,
                                      int flags, cocos_stream_t stream);
int cocos_box3_adjoint_planes_f16x3(const float* g_blocked, const float* gmax_dev, void* dc_hi, void* dc_lo,
                                    float* scale_out_dev, int B, int Nq, int Nk, int grid_h, int grid_w,
                                    cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K12 statistics of the zero-padded 3x3-unfolded, centred feature vectors without unfolding (match_kernel 3 with
 *     PONO_C: correspondence.py:276-280 / :286-289), feeding K6:
 *   fwd: x [B,C,h,w] -> mu[b,p] = mean of the 9*C unfolded entries at p, nrm[b,p] = ||U_p - mu||_2,
 *        a[b,p] = 1/(nrm + eps);  k_unfolded = 9*C;  ws: 2*B*h*w floats of scratch
 *   bwd: dmu, da (either may be NULL) -> dx [B,C,h,w] (written, not accumulated);  ws: 2*B*h*w floats
 * ------------------------------------------------------------------------------------- */
int cocos_unfold3_stats_fwd(const float* x, float* mu, float* a, float* nrm, float* ws,
                            int B, int C, int h, int w, float k_unfolded, float eps, cocos_stream_t stream);
/* Same, and *amax_inout_dev = max(*amax_inout_dev, max|x|): the scale source of the f16 split of x (K19's correlation GEMM). */
int cocos_unfold3_stats_fwd_amax(const float* x, float* mu, float* a, float* nrm, float* ws, int B, int C, int h, int w,
                                 float k_unfolded, float eps, float* amax_inout_dev, cocos_stream_t stream);
int cocos_unfold3_stats_bwd(const float* x, const float* mu, const float* a, const float* nrm,
                            const float* dmu, const float* da, float* dx, float* ws,
                            int B, int C, int h, int w, float k_unfolded, cocos_stream_t stream);
/* ... only its two per-position maps (ws = [g1 | g2]): the consumer applies dx[c,p] = g1[p] + 2 x[c,p] g2[p]
 * (cocos_proj_bwd_input_f16x3 mode 1). */
int cocos_unfold3_stats_bwd_maps(const float* mu, const float* a, const float* nrm, const float* dmu /* nullable */,
                                 const float* da /* nullable */, float* ws /* 2*B*h*w */, int B, int h, int w, float k_unfolded,
                                 cocos_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K13 InstanceNorm2d(affine=False) (+ residual) + PReLU of the ResidualBlocks in front of theta/phi
 *     (correspondence.py:13-36: `prelu(bn1(conv1(..)))` at :29 and `prelu(bn2(conv2(..)) + x)` at :31-33):
 *   x, residual (nullable), y: [planes, N] with planes = B*C, N = h*w;  prelu_weight: 1 float (nn.PReLU())
 *   fwd: z = (x - mean)/sqrt(biased var + eps) (+ residual);  y = z > 0 ? z : a*z
 *   bwd: dx, dresidual (nullable, = dz), da_partials [planes] (nullable; the caller sums them)
 * ------------------------------------------------------------------------------------- */
int cocos_instnorm_prelu_fwd(const float* x, const float* residual, const float* prelu_weight, float* y,
                             int planes, int N, float eps, cocos_stream_t stream);
int cocos_instnorm_prelu_bwd(const float* x, const float* residual, const float* prelu_weight, const float* dy,
                             float* dx, float* dresidual, float* da_partials, int planes, int N, float eps,
                             cocos_stream_t stream);
/* The same with the PReLU weight's gradient accumulated in fp64 from the products to the last addition (round 6: da is one number
 * summed over every element of the layer with cancelling terms): da_partials_f64 = workspace of `planes` doubles, *da_out = the sum. */
int cocos_instnorm_prelu_bwd_f64(const float* x, const float* residual, const float* prelu_weight, const float* dy, float* dx,
                                 float* dresidual, double* da_partials_f64, float* da_out, int planes, int N, float eps,
                                 cocos_stream_t stream);
/* Round 6: the same two passes also leaving max|out| in a device cell (*cell = max(*cell, max|.|); the cell must hold a finite value
 * >= 0) — y is the input of the next convolution / projection (correspondence.py:21, :25, :272, :282) and dx the output gradient of the
 * convolution in front of the norm (:20, :24): both are split into f16 hi / lo planes scaled by that maximum, which used to cost a
 * pass over the tensor each (cocos_absmax).  amax_partials = workspace of `planes` floats: one maximum per plane, reduced into the cell
 * by a one-workgroup kernel of the same call (same-address atomics from 13 k waves serialise at ~10 ns each: measured 5x slower).
 * _bwd_amax: da_partials_f64 / da_out both NULL = no PReLU-weight gradient. */
int cocos_instnorm_prelu_fwd_amax(const float* x, const float* residual, const float* prelu_weight, float* y, float* y_amax_inout_dev,
                                  float* amax_partials, int planes, int N, float eps, cocos_stream_t stream);
int cocos_instnorm_prelu_bwd_amax(const float* x, const float* residual, const float* prelu_weight, const float* dy, float* dx,
                                  float* dresidual, double* da_partials_f64, float* da_out, float* dx_amax_inout_dev, float* amax_partials,
                                  int planes, int N, float eps, cocos_stream_t stream);

/* Debug: runs one v_mfma_f32_32x32x2_f32 with known operands and dumps the 64x16 accumulator
 * registers to out[64*16] so the host can verify the lane/register -> (row, col) map. */
int cocos_debug_mfma_probe(float* out, cocos_stream_t stream);

/* K16b (round 3): the stride-1 convolutions of the feature producers in ONE bf16 MFMA term on operands that are bf16 in
 * memory (csrc/conv_nhwc_bf16.hip) — replaces, under COCOS_CONV=bf16, the same reference lines as cocos_conv2d_fwd_f16x3 /
 * cocos_conv2d_wgrad_bf16 (ResidualBlock correspondence.py:13-36, adaptors :150-173, SPADE mlp normalization.py:118-127).
 *   cocos_conv2d_nhwc_prep_bf16:  x fp32 [B][C][H][W] -> xp bf16 [B][H+2 pad][W+2 pad][Cp], Cp = 32 ceil(C/32), channels
 *       beyond C zero, the border zero (reflect = 0) or mirrored as nn.ReflectionPad2d (reflect = 1).
 *   cocos_conv2d_nhwc_bf16:  y fp32 [B][Cout][OH][OW] = bias + conv(xp) with NO further padding,
 *       OH = (Hp - dil (KH-1) - 1) / stride + 1, OW likewise; w_planes = cocos_conv2d_weight_planes(mode | 2) ([KH*KW*Cp/32][Cout][32] bf16).
 *       The input gradient of a stride-1 layer is this call on prep(dy, dil (K-1) - pad, 0) with the mode-1 planes.
 *       fold_ring (input gradient of a layer behind nn.ReflectionPad2d(1), stride 1): y is then dx [B][Cout][OH-2][OW-2] — the
 *       interior of the (OH x OW) padded gradient is stored straight into it, the border ring goes to fold_ring
 *       ([B][Cout][2 OW + 2 (OH-2)] floats) and a second small launch adds it back mirrored: nn.ReflectionPad2d's backward without
 *       the padded tensor or a pass over it.
 *       workspace: zero-initialised scratch of cocos_conv2d_nhwc_bf16_workspace_bytes() bytes, one per stream, cleared once by
 *       the caller (every launch leaves its flags zero): with it, a layer whose tile count is just above a multiple of the CU
 *       count runs as ONE workgroup per CU over contiguous ranges of (tile, k-step) units ("stream-K"; a cut tile's two parts
 *       meet through the workspace) instead of paying a whole extra round; without it every workgroup owns one tile.
 *   cocos_conv2d_nhwc_wgrad_bf16:  partial fp32 [S][Cout][KH*KW*Cp] (summed / re-ordered by cocos_conv2d_wgrad_reduce),
 *       S = cocos_conv2d_nhwc_wgrad_bf16_slices(...) (0: shape not taken — OW must be a multiple of 32); dyp = prep(dy, q, 0).
 *   cocos_conv2d_nhwc_bf16_supported: 1 when the layer shape takes this path (>= 128 output rows, >= 32 input channels; strided layers: forward and weight
 *       gradient — their input gradient stays on cocos_conv2d_fwd_scatter_f16x3's parity classes). */
int cocos_conv2d_nhwc_bf16_supported(int Cin, int Cout, int KH, int KW, int stride);
/* K16c: the same three kernels for the fp32-accurate flavour — every operand as TWO f16 planes (hi, lo; the lo plane directly
 * behind the hi plane: [2][B][Hp][Wp][Cp]) of x * 2^k, k from the tensor's max|x| cell, three MFMA terms per product, results
 * scaled back in the epilogues: the arithmetic of cocos_conv2d_fwd_f16x3 / cocos_conv2d_wgrad_f16x3 (same reference lines) on
 * K16b's data path.  w_hi / w_lo / *w_scale_dev = cocos_conv2d_weight_planes(mode 0 | 1). */
int cocos_conv2d_nhwc_prep_f16x3(const float* x, void* xp, const float* amax_dev /* nullable: scale 1 */, int B, int C, int H, int W,
                                 int pad, int reflect, cocos_stream_t stream);
int cocos_conv2d_nhwc_f16x3(const void* xp, const void* w_hi, const void* w_lo, const float* w_scale_dev, const float* x_amax_dev,
                            const float* bias /* nullable */, float* y, float* fold_ring /* nullable */,
                            void* workspace /* nullable: as cocos_conv2d_nhwc_bf16 */, long long workspace_bytes, int B, int Cp, int Hp,
                            int Wp, int Cout, int KH, int KW, int dil, int stride, cocos_stream_t stream);
int cocos_conv2d_nhwc_wgrad_f16x3(const void* xp, const void* dyp, const float* x_amax_dev, const float* g_amax_dev, float* partial,
                                  int B, int Cp, int Hp, int Wp, int Cout, int q, int KH, int KW, int dil, int stride,
                                  cocos_stream_t stream);
int cocos_conv2d_nhwc_prep_bf16(const float* x, void* xp, int B, int C, int H, int W, int pad, int reflect, cocos_stream_t stream);
long long cocos_conv2d_nhwc_bf16_workspace_bytes(void);
int cocos_conv2d_nhwc_bf16(const void* xp, const void* w_planes, const float* bias /* nullable */, float* y,
                           float* fold_ring /* nullable */, void* workspace /* nullable */, long long workspace_bytes, int B, int Cp,
                           int Hp, int Wp, int Cout, int KH, int KW, int dil, int stride, cocos_stream_t stream);
int cocos_conv2d_nhwc_wgrad_bf16_slices(int B, int OH, int OW, int Cp, int Cout, int KH, int KW);
int cocos_conv2d_nhwc_wgrad_bf16(const void* xp, const void* dyp, float* partial, int B, int Cp, int Hp, int Wp, int Cout, int q,
                                 int KH, int KW, int dil, int stride, cocos_stream_t stream);

/* K21 (round 3): the weight of torch.nn.utils.spectral_norm — every convolution of the reference's generator / discriminator and of
 * netCorr's feature producers is wrapped in it (normalization.py:21-61, architecture.py:41-52).  W viewed as [R = out channels][K],
 * u [R] / v [K] the module's buffers.  fwd: (power_iteration: v <- normalize(W^T u), u <- normalize(W v), in place, ONE iteration)
 * sigma = u . (W v), wsn = W / sigma, *sigma_out = sigma; workspace: cocos_spectral_weight_workspace_floats(R, K) floats.
 * bwd: dW = G / sigma - (sum(G o W) / sigma^2) u v^T with the u, v sigma was taken with; workspace: 1024 floats. */
long long cocos_spectral_weight_workspace_floats(int R, int K);
int cocos_spectral_weight_fwd(const float* W, float* u, float* v, float* wsn, float* sigma_out,
                              float* amax_inout_dev /* nullable: max(*cell, max|wsn|), cell zeroed by the caller */, float* workspace,
                              int R, int K, float eps, int power_iteration, cocos_stream_t stream);
int cocos_spectral_weight_bwd(const float* G, const float* W, const float* u, const float* v, const float* sigma_dev, float* dW,
                              float* workspace, int R, int K, cocos_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* COCOS_HIP_H */
