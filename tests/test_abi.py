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
