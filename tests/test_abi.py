"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950 without a GPU, loads,
and exports exactly what include/cocos_hip.h declares; the Python binding mirrors the header; the
product path refuses to run without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "cocos_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cocos_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for must in ("cocos_center_l2norm_fwd", "cocos_center_l2norm_bwd", "cocos_corr_softmax_warp_fwd",
                 "cocos_corr_softmax_warp_bwd", "cocos_corr_materialize", "cocos_row_softmax_fwd",
                 "cocos_row_softmax_bwd", "cocos_version", "cocos_last_error_string"):
        assert must in syms


def test_library_exports_every_declared_symbol(hip_lib):
    for name in _declared_symbols():
        assert hasattr(hip_lib, name), f"{name} declared in cocos_hip.h but not exported"


def test_binding_signature_table_matches_header(hip_lib):
    from cocosnet_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared_symbols()


def test_version_and_error_string(hip_lib):
    assert hip_lib.cocos_version() >= 100
    assert isinstance(hip_lib.cocos_last_error_string(), bytes)


def test_argument_validation_needs_no_gpu(hip_lib):
    """Null pointers / unsupported shapes are rejected before any HIP call is made."""
    rc = hip_lib.cocos_corr_softmax_warp_fwd(None, None, None, None, None, None, 1, 256, 4, 4, 3,
                                             ctypes.c_float(100.0), None)
    assert rc == -1 and b"null" in hip_lib.cocos_last_error_string()
    one = ctypes.c_void_p(16)
    rc = hip_lib.cocos_corr_softmax_warp_fwd(one, one, one, one, one, None, 1, 2304, 4, 4, 3,
                                             ctypes.c_float(100.0), None)
    assert rc == -2 and b"K == 256" in hip_lib.cocos_last_error_string()
    rc = hip_lib.cocos_corr_softmax_warp_fwd(one, one, one, one, one, None, 1, 256, 4, 4, 161,
                                             ctypes.c_float(100.0), None)
    assert rc == -2
    assert hip_lib.cocos_corr_softmax_warp_bwd_workspace_bytes(8, 256, 4096, 4096, 154) == 8 * 4096 * 4
    rc = hip_lib.cocos_row_softmax_fwd(one, one, 0, 16, None)
    assert rc == -1


def test_k0_streaming_planners_and_validation_need_no_gpu(hip_lib):
    """Host-side planning of the K0 streaming kernels: padded k extent of the register-resident weight planes,
    number of partial tiles of the weight-gradient reduction (0 = shape goes to the general split GEMM), and the
    argument checks that run before any HIP call."""
    kpad = hip_lib.cocos_proj1x1_stream_kpad
    assert [kpad(k) for k in (1, 64, 256, 257, 407, 416, 417, 0, -3)] == [256, 256, 256, 416, 416, 416, 0, 0, 0]
    parts = hip_lib.cocos_proj1x1_dw_partials_f16x3
    assert parts(8, 407, 256, 4096) == 128            # the benchmark shape: 16 chunks of 256 positions per image
    assert parts(8, 256, 256, 4096) == 128
    assert parts(1, 3, 2, 64) == 1                    # tiny grids: one chunk
    assert parts(5, 416, 130, 192) == 15              # >= 4 k-steps (64 positions) per chunk
    assert parts(2, 300, 407, 128) == 0               # more than 256 output channels
    assert parts(2, 449, 256, 128) == 0               # more than 448 input channels
    assert parts(1, 5, 3, 21) == 0                    # HW not a multiple of 4
    one = ctypes.c_void_p(16)
    f = ctypes.c_float
    rc = hip_lib.cocos_proj1x1_stream_f16x3(one, one, one, None, None, one, 2, 417, 256, 64, None, None)
    assert rc == -2 and b"K <= 416" in hip_lib.cocos_last_error_string()
    rc = hip_lib.cocos_proj1x1_stream_f16x3(one, one, one, None, None, one, 2, 256, 256, 100, None, None)
    assert rc == -2
    rc = hip_lib.cocos_proj1x1_stream_f16x3(None, one, one, None, None, one, 2, 256, 256, 64, None, None)
    assert rc == -1
    rc = hip_lib.cocos_proj1x1_dw_f16x3(one, one, one, None, one, one, 2, 256, 256, 64, None, None, None)
    assert rc == -1 and b"go together" in hip_lib.cocos_last_error_string()
    rc = hip_lib.cocos_proj1x1_dw_f16x3(one, one, one, None, one, None, 2, 256, 300, 64, None, None, None)
    assert rc == -2
    rc = hip_lib.cocos_warp_values(one, one, one, 1, 3, 5, 10, 10, 4, None)
    assert rc == -2 and b"multiple of down" in hip_lib.cocos_last_error_string()
    rc = hip_lib.cocos_warp_values(None, one, one, 1, 3, 5, 8, 8, 4, None)
    assert rc == -1
    rc = hip_lib.cocos_split_f16_rows(one, one, one, 4, 8, 7, f(1.0), None, None, None)
    assert rc == -1
    rc = hip_lib.cocos_absmax_accumulate(one, 0, one, None)
    assert rc == -1
    rc = hip_lib.cocos_center_l2norm_bwd_amax(one, one, one, one, None, None, 1, 16, 8, 1, f(1e-16), None, None)
    assert rc == -1 and b"amax" in hip_lib.cocos_last_error_string()
    assert hip_lib.cocos_box3_logits_bwd_workspace_bytes(8, 64, 64) > 2 * 8 * (256 * 4096 + 4096 * 4) * 4


def test_product_path_fails_loudly_on_cpu_tensors(hip_lib):
    from cocosnet_amd import _lib, ops
    x = torch.randn(1, 256, 8)
    with pytest.raises(_lib.CocosHipError, match="no CPU fallback"):
        ops.center_l2norm(x, True)
    with pytest.raises(_lib.CocosHipError):
        ops.corr_softmax_warp(x, x, torch.randn(1, 3, 8), 100.0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under cocosnet_amd/ may import it."""
    pkg = os.path.join(REPO, "cocosnet_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "corr_oracle" not in src, f
