// Version / error reporting of the C ABI plus a one-instruction MFMA layout probe.
#include <stdarg.h>

#include "common.h"

namespace cocos {

std::string& last_error() {
    static thread_local std::string e;
    return e;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

// One v_mfma_f32_32x32x2_f32 with A[i][k] = 1 + i + 100k and B[k][j] = (k == 0 ? 1 : 0) * ... chosen
// asymmetric so that a row/col swap or a wrong k-pairing is visible:
//   D[i][j] = A[i][0]*B[0][j] + A[i][1]*B[1][j],  B[0][j] = 1 + j,  B[1][j] = 1000 * (1 + j).
__global__ void mfma_probe_kernel(float* out) {
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, c = lane & 31;
    const float a = 1.0f + c + 100.0f * h;               // A[i = c][k = h]
    const float b = (h == 0 ? 1.0f : 1000.0f) * (1 + c); // B[k = h][j = c]
    f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    d = mfma32(a, b, d);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = d[r];
}

}  // namespace cocos

extern "C" int cocos_version(void) { return 620; /* 0.6.2 (round 6): cocos_instnorm_prelu_fwd_amax / _bwd_amax, cocos_pono_spade_fwd_amax / _bwd_amax / _amax_partials (K13 / K9 leave max|.| for the next convolution); 0.6.1: K24 / K25 — cocos_proj_bwd_input_f16x3 / _planes_f16x3, cocos_proj_raw_planes_stats_f16x3, cocos_proj1x1_dw_affine(_pair)_f16x3, pair / multi launches (cocos_absmax4, cocos_proj_weight_prep_pair, cocos_split_f16_transpose_pair, cocos_unfold3_stats_finish_pair / _bwd_maps_pair), cocos_warp_head_fwd / _bwd, cocos_instnorm_prelu_bwd_f64; 0.6.0: K23 — cocos_proj_center_l2norm_planes_f16x3 / cocos_proj_weight_frag_planes (K0 fused with K1); 0.5.2 (round 5): K22 — cocos_contextual_cx_fwd_f16x3 / _bwd_f16x3 / cocos_contextual_cx_coeffs (the contextual loss without its [N, N] matrices); the convolutions' operand planes are split round-to-nearest; 0.5.1 (round 4): `flags` on the K19 entry points (transposed T, accumulated G), `_ex` K2 entry points (magnitude-free flavour, k_active, d_pre), cocos_rowdot_f64, 128-wide box3 grids; 0.5.0 (round 3): K19/K20 box3 fused family, K1 planes, plane preparation, device-side operand scales, K16b NHWC bf16 convolution, cocos_channel_sum takes a partials buffer; 0.4.0 (round 2): K2 split kernels take a device-side V scale and a private saved-logits layout (signatures changed); K15 contextual rows; K16 convolution */ }

extern "C" const char* cocos_last_error_string(void) { return cocos::last_error().c_str(); }

extern "C" int cocos_debug_mfma_probe(float* out, cocos_stream_t stream) {
    using namespace cocos;
    COCOS_REQUIRE(out, COCOS_ERR_INVALID, "mfma_probe: null pointer");
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), out);
    COCOS_HIP_CHECK(hipGetLastError());
    return COCOS_OK;
}
