"""SPADE on the fused HIP kernel, for our own producers AND for the unmodified reference's translation generator.

SURVEY.md §8(f) rank 1: `PositionalNorm2d` (normalization.py:63-68) -> `SPADE.forward`'s modulation
`normalized * (1 + gamma) + beta` (:148-151) -> the `LeakyReLU(0.2)` that `SPADEResnetBlock.forward` applies right
after (architecture.py:88-95, `actvn` :107-108) are one HBM pass in K9 (`ops.pono_spade`, pono_spade.hip) instead of
~9 framework launches and their autograd.  `SPADEGenerator` (generator.py:35-45) uses 7 such blocks, each
`AdaptiveFeatureGenerator` 3 (:123-127).

`install_spade_into_reference(networks)` rebinds the two `forward` methods of the reference's own classes, so every
SPADE of an already-written training script (netG and netCorr alike) takes the fused path on the GPU; parameters,
sub-modules and `state_dict` are untouched (checkpoints load unchanged).  CPU tensors, non-fp32 tensors, the
`similarity_map` argument and non-PONO norms keep the reference's arithmetic (instance / batch / sync-batch statistics
are computed by their own modules, only the modulation + activation are fused where that helps).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops

LEAKY_SLOPE = 2e-1      # SPADEResnetBlock.actvn (architecture.py:107-108)


def _hip_ok(*ts) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


def modulate(x, gamma, beta, pono: bool, param_free_norm=None, slope: float = 1.0):
    """leaky_relu(norm(x) * (1 + gamma) + beta, slope): the tail of SPADE.forward (+ the block's activation).
    pono: PositionalNorm2d — fused with the modulation and the activation in K9 when x is CUDA fp32; otherwise the
    norm stays the module's own and the modulation + activation are K17."""
    if pono and _hip_ok(x, gamma, beta) and gamma.shape == x.shape and beta.shape == x.shape:
        return ops.pono_spade(x, gamma, beta, slope)
    if pono:
        mu = x.mean(dim=1, keepdim=True)                              # normalization.py:63-68
        normalized = (x - mu) / x.var(dim=1, keepdim=True).add(1e-5).sqrt()
    else:
        normalized = param_free_norm(x)                               # instance | batch | sync-batch (normalization.py:93-101)
        if _hip_ok(normalized, gamma, beta) and gamma.shape == normalized.shape and beta.shape == normalized.shape:
            return ops.spade_modulate(normalized, gamma, beta, slope)  # K17: modulation + activation in one pass
    y = normalized * (1 + gamma) + beta                               # normalization.py:148
    return y if slope == 1.0 else F.leaky_relu(y, slope)


def shared_activation(self, segmap):
    """`self.mlp_shared(segmap)` (normalization.py:139): conv + ReLU of the label map.  Its own function so that a caller can put
    another evaluation of the same piecewise-linear map in its place."""
    return self.mlp_shared(segmap)


def spade_forward(self, x, segmap, similarity_map=None, slope: float = 1.0):
    """Drop-in for `SPADE.forward(x, segmap, similarity_map=None)` (normalization.py:129-151); works on the reference's
    module instances (attributes param_free_norm, mlp_shared, pad, mlp_gamma, mlp_beta, pad_type).  `slope`: negative
    slope of the LeakyReLU the caller would apply next (1.0 = none)."""
    if segmap.shape[2:] != x.shape[2:]:      # (spade_resnet_block_forward resizes once for the block's two or three SPADEs)
        segmap = F.interpolate(segmap, size=x.size()[2:], mode="nearest")
    actv = shared_activation(self, segmap)
    if getattr(self, "pad_type", "nozero") != "zero":
        actv = self.pad(actv)
    gamma, beta = self.mlp_gamma(actv), self.mlp_beta(actv)
    if similarity_map is not None:
        similarity_map = F.interpolate(similarity_map, size=gamma.size()[2:], mode="nearest")
        gamma, beta = gamma * similarity_map, beta * similarity_map
    pnorm = getattr(self, "param_free_norm", None)
    pono = getattr(self, "pono", None)
    if pono is None:       # the reference stores the function itself
        pono = callable(pnorm) and getattr(pnorm, "__name__", "") == "PositionalNorm2d"
    return modulate(x, gamma, beta, bool(pono), pnorm, slope)


def spade_resnet_block_forward(self, x, seg1):
    """Drop-in for `SPADEResnetBlock.forward` (architecture.py:70-95): same sub-module calls in the same order, with
    the two `actvn(norm_k(...))` pairs as one fused call each."""
    # the reference resizes the label map inside every SPADE (normalization.py:133): the block's SPADEs all see x's grid (its
    # convolutions keep the size), so ONE nearest resize serves them — and the convolutions that read it find its max|.| from the
    # first one (per module step: 24 resize kernels and as many max|.| passes less)
    if seg1.shape[2:] != x.shape[2:]:
        seg1 = F.interpolate(seg1, size=x.size()[2:], mode="nearest")
    x_s = self.conv_s(self.norm_s(x, seg1)) if self.learned_shortcut else x            # :97-102
    pad = self.pad if getattr(self, "pad_type", "nozero") != "zero" else (lambda t: t)
    dx = self.conv_0(pad(self.norm_0(x, seg1, slope=LEAKY_SLOPE)))
    dx = self.conv_1(pad(self.norm_1(dx, seg1, slope=LEAKY_SLOPE)))
    if self.use_se:
        dx = self.se_layar(dx)
    return x_s + dx


def install_spade_into_reference(networks_module):
    """Rebind `SPADE.forward` and `SPADEResnetBlock.forward` of the reference (`models.networks.normalization` /
    `.architecture`) to the fused versions above.  Returns the two patched classes; `uninstall` restores them."""
    import importlib
    norm = importlib.import_module(networks_module.__name__ + ".normalization")
    arch = importlib.import_module(networks_module.__name__ + ".architecture")
    for cls, fn in ((norm.SPADE, spade_forward), (arch.SPADEResnetBlock, spade_resnet_block_forward)):
        if not hasattr(cls, "_cocos_reference_forward"):
            cls._cocos_reference_forward = cls.forward
        cls.forward = fn
    return norm.SPADE, arch.SPADEResnetBlock


def uninstall_spade_from_reference(networks_module):
    import importlib
    norm = importlib.import_module(networks_module.__name__ + ".normalization")
    arch = importlib.import_module(networks_module.__name__ + ".architecture")
    for cls in (norm.SPADE, arch.SPADEResnetBlock):
        if hasattr(cls, "_cocos_reference_forward"):
            cls.forward = cls._cocos_reference_forward
            del cls._cocos_reference_forward
