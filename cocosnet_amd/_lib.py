"""ctypes binding of libcocos_hip.so (the C ABI declared in include/cocos_hip.h).

No fallback: if the library is missing or a call fails, this raises — the product path never
silently degrades to PyTorch or CPU code (the oracle lives under oracle/ and is test-only).
"""
from __future__ import annotations

import ctypes
import os
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COCOS_LIB_PATH") or os.path.join(_PKG_DIR, "lib", "libcocos_hip.so")

_c_float_p = ctypes.c_void_p      # device pointers travel as integers
_stream_t = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/cocos_hip.h one to one
_SIGNATURES = {
    "cocos_version": (ctypes.c_int, []),
    "cocos_last_error_string": (ctypes.c_char_p, []),
    "cocos_center_l2norm_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_float, _stream_t]),
    "cocos_center_l2norm_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                                _c_float_p, _c_float_p,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_float, _stream_t]),
    "cocos_center_l2norm_bwd_amax": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 4
                                     + [ctypes.c_float, _c_float_p, _stream_t]),
    "cocos_center_l2norm_fwd_planes": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4
                                       + [ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_center_l2norm_bwd_planes": (ctypes.c_int, [ctypes.c_void_p] * 2 + [_c_float_p] * 3 + [ctypes.c_int] * 4
                                       + [ctypes.c_float, ctypes.c_float, _c_float_p, _stream_t]),
    "cocos_corr_softmax_warp_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                                    _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_float, _stream_t]),
    "cocos_corr_softmax_warp_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "cocos_corr_softmax_warp_bwd_prepare": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 3 + [_stream_t]),
    "cocos_corr_softmax_warp_bwd_query": (ctypes.c_int, [_c_float_p] * 10 + [ctypes.c_int] * 5
                                          + [ctypes.c_float, _stream_t]),
    "cocos_corr_softmax_warp_bwd_key": (ctypes.c_int, [_c_float_p] * 8 + [ctypes.c_int] * 5
                                        + [ctypes.c_float, _stream_t]),
    "cocos_corr_softmax_warp_bwd_key_from_ds": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 4
                                                + [_stream_t]),
    "cocos_corr_softmax_warp_bwd": (ctypes.c_int, [_c_float_p] * 9 + [ctypes.c_void_p,
                                                                      ctypes.c_size_t,
                                                                      ctypes.c_int, ctypes.c_int,
                                                                      ctypes.c_int, ctypes.c_int,
                                                                      ctypes.c_int, ctypes.c_float,
                                                                      _stream_t]),
    "cocos_corr_materialize": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_float, _stream_t]),
    "cocos_corr_materialize_bwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int, ctypes.c_int,
                                                                     ctypes.c_int, ctypes.c_int,
                                                                     ctypes.c_float, _stream_t]),
    "cocos_row_softmax_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int64, ctypes.c_int,
                                              _stream_t]),
    "cocos_row_softmax_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int64,
                                              ctypes.c_int, _stream_t]),
    "cocos_warp_materialized_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, _stream_t]),
    "cocos_warp_materialized_bwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int, ctypes.c_int,
                                                                      ctypes.c_int, ctypes.c_int,
                                                                      _stream_t]),
    "cocos_proj1x1_fwd": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_proj1x1_bwd_partials": (ctypes.c_int, [ctypes.c_int] * 4),
    "cocos_proj1x1_bwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_box3_logits_fwd": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 3
                              + [ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_box3_logits_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "cocos_box3_logits_bwd": (ctypes.c_int, [_c_float_p] * 11 + [ctypes.c_void_p, ctypes.c_size_t]
                              + [ctypes.c_int] * 3 + [ctypes.c_float, _stream_t]),
    "cocos_box3_logits_bwd_amax": (ctypes.c_int, [_c_float_p] * 11 + [ctypes.c_void_p, ctypes.c_size_t]
                                   + [ctypes.c_int] * 3 + [ctypes.c_float, _c_float_p, _stream_t]),
    "cocos_logits_softmax_warp_fwd": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_logits_softmax_warp_bwd": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_wta_scale_mask_bytes": (ctypes.c_longlong, [ctypes.c_longlong, ctypes.c_int]),
    "cocos_wta_scale_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_wta_scale_bwd": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, _c_float_p, ctypes.c_longlong, ctypes.c_int,
                                           ctypes.c_float, _stream_t]),
    "cocos_pono_spade_fwd": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float] * 2 + [_stream_t]),
    "cocos_pono_spade_bwd": (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int] * 3 + [ctypes.c_float] * 2 + [_stream_t]),
    "cocos_pono_spade_amax_partials": (ctypes.c_int, [ctypes.c_int] * 3),
    "cocos_pono_spade_fwd_amax": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_float] * 2 + [_stream_t]),
    "cocos_pono_spade_bwd_amax": (ctypes.c_int, [_c_float_p] * 9 + [ctypes.c_int] * 3 + [ctypes.c_float] * 2 + [_stream_t]),
    "cocos_split_f16": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4
                        + [ctypes.c_float, _stream_t]),
    "cocos_corr_softmax_warp_saved_logits_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "cocos_corr_softmax_warp_fwd_f16x3": (ctypes.c_int, [ctypes.c_void_p] * 6 + [_c_float_p] * 2 + [ctypes.c_void_p,
                                                                                                 _c_float_p, ctypes.c_void_p]
                                          + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_float, _c_float_p, _c_float_p,
                                                                  _stream_t]),
    "cocos_corr_softmax_warp_fwd_f16x3_ex": (ctypes.c_int, [ctypes.c_void_p] * 6 + [_c_float_p] * 2 + [ctypes.c_void_p,
                                                                                                    _c_float_p, ctypes.c_void_p]
                                             + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_float, _c_float_p, _c_float_p,
                                                                     _c_float_p, _c_float_p, ctypes.c_int, _stream_t]),
    "cocos_f16_plane_block_mask": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p, _stream_t]),
    "cocos_split_f16_ex": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5
                           + [ctypes.c_float, _c_float_p, _c_float_p, _stream_t]),
    "cocos_split_f16_rows": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3
                             + [ctypes.c_float, _c_float_p, _c_float_p, _stream_t]),
    "cocos_corr_softmax_warp_bwd_query_f16x3": (ctypes.c_int, [ctypes.c_void_p] * 6 + [_c_float_p] * 4
                                                + [ctypes.c_void_p, _c_float_p]
                                                + [ctypes.c_void_p] * 4 + [_c_float_p] * 3 + [ctypes.c_void_p]
                                                + [ctypes.c_int] * 6
                                                + [ctypes.c_float, ctypes.c_float, _c_float_p, _c_float_p, ctypes.c_int,
                                                   _stream_t]),
    "cocos_corr_softmax_warp_bwd_query_f16x3_ex": (ctypes.c_int, [ctypes.c_void_p] * 6 + [_c_float_p] * 4
                                                   + [ctypes.c_void_p, _c_float_p]
                                                   + [ctypes.c_void_p] * 4 + [_c_float_p] * 3 + [ctypes.c_void_p]
                                                   + [ctypes.c_int] * 6
                                                   + [ctypes.c_float, ctypes.c_float, _c_float_p, _c_float_p, ctypes.c_int,
                                                      _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, _stream_t]),
    "cocos_rowdot_f64": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 3 + [_stream_t]),
    "cocos_box3_fused_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "cocos_box3_corr_xbox_f16x3": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_float_p] + [ctypes.c_int] * 5
                                   + [_c_float_p, _c_float_p, _stream_t]),
    "cocos_box3_softmax_warp_fwd_f16x3": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_void_p] * 2 + [_c_float_p] * 3
                                          + [ctypes.c_void_p] + [ctypes.c_int] * 6
                                          + [ctypes.c_float, ctypes.c_float, ctypes.c_int, _stream_t]),
    "cocos_box3_softmax_warp_bwd_colpart_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "cocos_box3_softmax_warp_bwd_f16x3": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_void_p] * 4 + [_c_float_p] * 10
                                          + [ctypes.c_void_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
                                          + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_float, _c_float_p, ctypes.c_int, _stream_t]),
    "cocos_box3_adjoint_planes_f16x3": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p]
                                        + [ctypes.c_int] * 5 + [_stream_t]),
    "cocos_warp_values_amax": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 6 + [_c_float_p, _stream_t]),
    "cocos_concat2_amax": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong,
                                                              _c_float_p, _stream_t]),
    "cocos_split_f16_chan_mask": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3
                                  + [_c_float_p, _c_float_p, ctypes.c_void_p, _stream_t]),
    "cocos_proj_weight_planes": (ctypes.c_int, [_c_float_p] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4
                                 + [_c_float_p, _c_float_p, _stream_t]),
    "cocos_sum_leading": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_longlong, _stream_t]),
    "cocos_conv2d_nhwc_bf16_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "cocos_conv2d_nhwc_prep_bf16": (ctypes.c_int, [_c_float_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [_stream_t]),
    "cocos_conv2d_nhwc_bf16_workspace_bytes": (ctypes.c_longlong, []),
    "cocos_conv2d_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p,
                                               ctypes.c_longlong] + [ctypes.c_int] * 9 + [_stream_t]),
    "cocos_conv2d_nhwc_prep_f16x3": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 6 + [_stream_t]),
    "cocos_conv2d_nhwc_f16x3": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p,
                                                _c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_longlong] + [ctypes.c_int] * 9
                                + [_stream_t]),
    "cocos_conv2d_nhwc_wgrad_f16x3": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p]
                                      + [ctypes.c_int] * 10 + [_stream_t]),
    "cocos_conv2d_nhwc_wgrad_bf16_slices": (ctypes.c_int, [ctypes.c_int] * 7),
    "cocos_conv2d_nhwc_wgrad_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 10 + [_stream_t]),
    "cocos_spectral_weight_workspace_floats": (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int]),
    "cocos_spectral_weight_fwd": (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, _stream_t]),
    "cocos_spectral_weight_bwd": (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int, ctypes.c_int, _stream_t]),
    "cocos_channel_sum_slices": (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong]),
    "cocos_channel_sum": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, _stream_t]),
    "cocos_box3_stat_grads": (ctypes.c_int, [_c_float_p] * 10 + [ctypes.c_longlong, ctypes.c_float, ctypes.c_float,
                                                                  _stream_t]),
    "cocos_hgemm_f16x3": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_float_p] + [ctypes.c_int] * 4
                          + [ctypes.c_float, _c_float_p, _c_float_p, ctypes.c_int, _stream_t]),
    "cocos_absmax": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, _c_float_p, _stream_t]),
    "cocos_absmax_accumulate": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, _c_float_p, _stream_t]),
    "cocos_absmax4": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, _c_float_p] * 4 + [_stream_t]),
    "cocos_proj1x1_fwd_f16x3": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 4 + [_c_float_p] * 2 + [_stream_t]),
    "cocos_proj1x1_bwd_partials_f16x3": (ctypes.c_int, [ctypes.c_int] * 4),
    "cocos_proj1x1_stream_kpad": (ctypes.c_int, [ctypes.c_int]),
    "cocos_proj1x1_dw_partials_f16x3": (ctypes.c_int, [ctypes.c_int] * 4),
    "cocos_proj1x1_dw_f16x3": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 4 + [_c_float_p] * 2 + [_stream_t]),
    "cocos_proj1x1_stream_f16x3": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p,
                                                  _c_float_p] + [ctypes.c_int] * 4 + [_c_float_p, _stream_t]),
    "cocos_proj_weight_frag_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "cocos_proj_weight_frag_planes": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p,
                                                     ctypes.c_int, ctypes.c_int, _stream_t]),
    "cocos_proj_center_l2norm_planes_f16x3": (ctypes.c_int, [ctypes.c_int]
                                              + ([_c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p]
                                                 + [ctypes.c_void_p] * 4) * 2
                                              + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_proj_weight_tfrag_bytes": (ctypes.c_size_t, []),
    "cocos_proj_weight_tfrag_planes": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                                      _stream_t]),
    "cocos_proj_weight_prep_pair": (ctypes.c_int, [ctypes.c_int]
                                    + [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_int, _stream_t]),
    "cocos_proj_bwd_input_supported": (ctypes.c_int, [ctypes.c_int] * 3),
    "cocos_proj_bwd_input_f16x3": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]
                                   + ([_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p,
                                       _c_float_p, _c_float_p, _c_float_p]) * 2
                                   + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_proj1x1_dw_affine_f16x3": (ctypes.c_int, [ctypes.c_int, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, ctypes.c_float,
                                                     _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p] + [ctypes.c_int] * 4
                                      + [_c_float_p, _c_float_p, _stream_t]),
    "cocos_proj1x1_dw_partials_pair_f16x3": (ctypes.c_int, [ctypes.c_int] * 4),
    "cocos_proj1x1_dw_affine_pair_f16x3": (ctypes.c_int, [ctypes.c_int, ctypes.c_float]
                                           + [_c_float_p, ctypes.c_void_p, ctypes.c_void_p] + [_c_float_p] * 8
                                           + [_c_float_p, ctypes.c_void_p, ctypes.c_void_p] + [_c_float_p] * 8
                                           + [_c_float_p, _c_float_p] + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_proj_raw_planes_stats_f16x3": (ctypes.c_int, [ctypes.c_int]
                                          + ([_c_float_p, ctypes.c_void_p] + [_c_float_p] * 6 + [ctypes.c_void_p] * 4) * 2
                                          + [ctypes.c_int] * 3 + [_stream_t]),
    "cocos_unfold3_stats_finish_pair": (ctypes.c_int, [_c_float_p] * 10 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_unfold3_stats_bwd_maps_pair": (ctypes.c_int, [_c_float_p] * 12 + [ctypes.c_int] * 3 + [ctypes.c_float, _stream_t]),
    "cocos_proj_bwd_input_planes_f16x3": (ctypes.c_int, [ctypes.c_int]
                                          + ([_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p,
                                              _c_float_p, _c_float_p, _c_float_p, _c_float_p]) * 2
                                          + [ctypes.c_int] * 3 + [_stream_t]),
    "cocos_proj1x1_bwd_f16x3": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 4 + [_c_float_p] * 3 + [_stream_t]),
    "cocos_upsample_nearest_fwd": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_upsample_nearest_bwd": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_warp_head_fwd": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 6 + [_stream_t]),
    "cocos_warp_head_bwd": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 6 + [_stream_t]),
    "cocos_split_f16_transpose_pair": (ctypes.c_int, ([_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p]) * 2
                                       + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_warp_values": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 6 + [_stream_t]),
    "cocos_corr_materialize_f16x3": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_float]
                                     + [_c_float_p] * 2 + [_stream_t]),
    "cocos_corr_materialize_bwd_f16x3": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_float]
                                         + [_c_float_p] * 3 + [_stream_t]),
    "cocos_logits_softmax_warp_fwd_f16x3": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p,
                                                           _c_float_p, _c_float_p] + [ctypes.c_int] * 4 + [_stream_t]),
    "cocos_logits_softmax_warp_bwd_f16x3": (ctypes.c_int, [_c_float_p] + [ctypes.c_void_p] * 4 + [_c_float_p] * 6
                                            + [ctypes.c_int] * 5 + [_stream_t]),
    "cocos_unfold3_stats_fwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_float] * 2 + [_stream_t]),
    "cocos_unfold3_stats_fwd_amax": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 4
                                     + [ctypes.c_float, ctypes.c_float, _c_float_p, _stream_t]),
    "cocos_unfold3_stats_bwd": (ctypes.c_int, [_c_float_p] * 8 + [ctypes.c_int] * 4 + [ctypes.c_float, _stream_t]),
    "cocos_unfold3_stats_bwd_maps": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_float, _stream_t]),
    "cocos_instnorm_prelu_fwd": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_float, _stream_t]),
    "cocos_instnorm_prelu_bwd": (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int] * 2 + [ctypes.c_float, _stream_t]),
    "cocos_instnorm_prelu_bwd_f64": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_void_p, _c_float_p] + [ctypes.c_int] * 2
                                     + [ctypes.c_float, _stream_t]),
    "cocos_instnorm_prelu_fwd_amax": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 2 + [ctypes.c_float, _stream_t]),
    "cocos_instnorm_prelu_bwd_amax": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p]
                                      + [ctypes.c_int] * 2 + [ctypes.c_float, _stream_t]),
    "cocos_contextual_rows_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_float, _stream_t]),
    "cocos_contextual_rows_bwd": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_float,
                                                                   ctypes.c_float, _stream_t]),
    "cocos_contextual_cx_fwd_f16x3": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_float_p] * 5 + [ctypes.c_void_p] + [ctypes.c_int] * 6
                                      + [ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_contextual_cx_bwd_f16x3": (ctypes.c_int, [ctypes.c_void_p] * 6 + [_c_float_p] * 13 + [ctypes.c_int] * 8
                                      + [ctypes.c_float, _stream_t]),
    "cocos_contextual_cx_coeffs": (ctypes.c_int, [_c_float_p] * 8 + [ctypes.c_longlong, ctypes.c_float, ctypes.c_float, _stream_t]),
    "cocos_reflect_pad2d_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong] + [ctypes.c_int] * 3 + [_stream_t]),
    "cocos_reflect_pad2d_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong] + [ctypes.c_int] * 3 + [_stream_t]),
    "cocos_spade_modulate_fwd": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_longlong, ctypes.c_float, _stream_t]),
    "cocos_spade_modulate_bwd": (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_longlong, ctypes.c_float, _stream_t]),
    "cocos_conv2d_out_size": (ctypes.c_int, [ctypes.c_int] * 5),
    "cocos_conv2d_kdim": (ctypes.c_int, [ctypes.c_int] * 3),
    "cocos_conv2d_fwd_f16x3": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p,
                                              _c_float_p, _c_float_p] + [ctypes.c_int] * 10 + [_stream_t]),
    "cocos_conv2d_fwd_scatter_f16x3": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p,
                                                      _c_float_p] + [ctypes.c_int] * 11 + [ctypes.c_longlong, ctypes.c_int,
                                                      ctypes.c_int, ctypes.c_longlong, _stream_t]),
    "cocos_conv2d_weight_planes": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 10
                                   + [_c_float_p, _c_float_p, _stream_t]),
    "cocos_conv2d_wgrad_reduce": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 5 + [_stream_t]),
    "cocos_conv2d_wgrad_slices": (ctypes.c_int, [ctypes.c_int] * 10),
    "cocos_conv2d_wgrad_f16x3": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 10 + [_stream_t]),
    "cocos_conv2d_wgrad_bf16": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 10 + [_stream_t]),
    "cocos_debug_mfma_probe": (ctypes.c_int, [_c_float_p, _stream_t]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lock = threading.Lock()
_lib = None


class CocosHipError(RuntimeError):
    """A libcocos_hip.so entry point returned a negative status."""


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise CocosHipError(
                    f"{LIB_PATH} not found: build it with `python -m cocosnet_amd.build` "
                    "(there is no PyTorch/CPU fallback for the correspondence hot path)")
            # PyTorch-ROCm bundles its own libamdhip64; import it FIRST so that this library binds
            # to the same HIP runtime instance (two runtimes in one process = "no ROCm-capable
            # device" on the second one, and torch's streams/pointers would be foreign to it).
            import torch  # noqa: F401
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name)   # AttributeError -> header and library disagree
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def call(name: str, *args):
    """Invoke an int-returning entry point; raise CocosHipError with the library's message."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.cocos_last_error_string()
        err = CocosHipError(f"{name} failed with code {rc}: {msg.decode() if msg else ''}")
        err.code = rc                # COCOS_ERR_* of include/cocos_hip.h
        raise err
    return rc
