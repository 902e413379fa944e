"""Loader for the READ-ONLY reference at /root/reference (test infrastructure, build container only).

The GPU box has no /root/reference: nothing under tests/ -m gpu, smoke() or bench.py may import
this module.  It exists to (a) pin oracle/corr_oracle.py against the reference's own code and
(b) generate the fixtures under tests/golden/ (see oracle/make_golden.py).

The reference has no tests and its hot path is pure PyTorch (models/networks/correspondence.py
:271-372), so "running the reference" means importing it.  Two third-party imports it needs are
absent here and are stubbed exactly as SURVEY.md §8(c) describes:
  * models.networks.sync_batchnorm (un-vendored git clone, README.md:28-34) -> BatchNorm subclasses
    (never instantiated under --PONO);
  * torchvision (only touched at import time by architecture.py:7 and ContextualLoss.py:13).
"""
from __future__ import annotations

import argparse
import contextlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "networks"))


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


_loaded = None


def load_reference():
    """Import the reference's `models.networks` package; returns the module (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not present (expected on the GPU box)")
    import torch.nn as nn

    sbn = types.ModuleType("models.networks.sync_batchnorm")
    sbn.SynchronizedBatchNorm2d = type("SynchronizedBatchNorm2d", (nn.BatchNorm2d,), {})
    sbn.SynchronizedBatchNorm1d = type("SynchronizedBatchNorm1d", (nn.BatchNorm1d,), {})
    sbn.DataParallelWithCallback = nn.DataParallel
    sys.modules["models.networks.sync_batchnorm"] = sbn
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        for n in ("Compose", "Lambda", "Normalize", "ToPILImage"):
            setattr(tvt, n, lambda *a, **k: None)
        tv.transforms = tvt
        tv.models = types.ModuleType("torchvision.models")
        tv.utils = types.ModuleType("torchvision.utils")
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                            "torchvision.models": tv.models, "torchvision.utils": tv.utils})
    sys.path.insert(0, REFERENCE_ROOT)
    with _cwd(REFERENCE_ROOT):   # util/util.py:22 loads ./util/color150.mat at import
        import models.networks as networks  # noqa: E402  (prints 'apex not found' — harmless)
    _loaded = networks
    return networks


def make_opt(**overrides) -> argparse.Namespace:
    """An `opt` Namespace with every field the correspondence path reads.

    Defaults = options/base_options.py defaults + the README ADE20k flag set
    (--use_attention --maskmix --warp_mask_losstype direct --PONO --PONO_C)."""
    opt = argparse.Namespace(
        semantic_nc=151, ngf=64, norm_E="spectralinstance", norm_G="spectralspadesyncbatch3x3",
        norm_D="spectralinstance", eqlr_sn=False, apex=False, PONO=True, PONO_C=True,
        adaptor_kernel=3, adaptor_se=False, adaptor_nonlocal=False, adaptor_res_deeper=False,
        dilation_conv=False, warp_stride=4, weight_domainC=0.0, domain_rela=False,
        use_coordconv=False, maskmix=True, warp_bilinear=False, mask_noise=False,
        noise_for_mask=False, isTrain=False, epoch=0, mask_epoch=-1, novgg_featpair=0.0,
        match_kernel=3, warp_patch=False, show_corr=False, warp_mask_losstype="direct",
        show_warpmask=False, warp_cycle_w=0.0, two_cycle=False, CBN_intype="warp_mask",
        use_attention=True, gpu_ids=[], init_type="xavier", init_variance=0.02, crop_size=256,
        aspect_ratio=1.0, netG="spade", video_like=False,
    )
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt


def build_reference_corr(opt, seed: int = 0):
    """`networks.define_Corr(opt)` of the reference (random xavier init, deterministic by seed)."""
    import torch

    networks = load_reference()
    torch.manual_seed(seed)
    with _cwd(REFERENCE_ROOT), contextlib.redirect_stdout(None):
        net = networks.define_Corr(opt)
    return net
