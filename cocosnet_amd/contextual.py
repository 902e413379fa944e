"""`ContextualLoss_forward` of the reference (models/networks/ContextualLoss.py:83-137) on the HIP kernels.

SURVEY.md §8(f) rank 3: the contextual loss is the other QK^T -> normalise -> reduce op on the training step's critical
path (pix2pix_model.py get_ctx_loss: VGG relu features of the generated image against the exemplar's).  Same class name,
constructor and `forward(X_features, Y_features, h=0.1, feature_centering=True)` signature; returns the per-sample loss.

    centring + x / (||x||_2 + eps)       K1 (`ops.center_l2norm`, mode 2 after the reference's own centring)   :108-116
    cos = X^T Y, d, d_norm, w, A, max_j A   K22 (`ops.contextual_cx`: the cosine tiles stay in the accumulators — a row-maximum sweep,
                                         then a sum sweep at the row's own temperature; its backward recomputes them)  :121-132
    CX = mean_i, loss = -log CX          host (B numbers)                                                      :132-133
Round 2's route (K3 `ops.corr_materialize` -> K15 `ops.contextual_rows` on the [B, N, N] cosine matrix, <= 4096 keys) is the
faster one while that matrix is small: `ROUTE = "auto"` picks per call (see below); no size limit and no framework fallback on a
GPU either way.  CPU tensors take the reference's formulation in torch (used by the CPU parity test only).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

_EPS = __import__("sys").float_info.epsilon      # util.feature_normalize (util/util.py:31-34)

#: "auto" (default, round 6): per call — the materialised route (round 2's K3 + K15 on the [B, N, N] cosine matrix) while that matrix is
#: small (B N^2 x 4 bytes <= MATERIALISED_MAX_BYTES and N <= 4096, K15's key limit), K22 above: `tools/contextual_bench.py` has the
#: materialised route FASTER wherever it applies (0.26 vs 0.39 ms at get_ctx_loss's B = 8, N = 1024, C = 512; 1.48 vs 2.79 ms at
#: N = 4096) and hungrier only where the matrix is large (1088 vs 321 MiB at N = 4096, B = 8 — 0.4 % of the 288 GB of one MI355X).
#: "fused": K22 always (nothing [N, N] in HBM, any N); "materialised": K3 + K15 always (raises beyond 4096 keys: no framework
#: fallback on a GPU).  A module attribute, not an environment switch.
ROUTE = "auto"
#: "auto" takes the materialised route up to this many bytes of cosine matrix (B = 8, N = 4096: 512 MiB)
MATERIALISED_MAX_BYTES = 1 << 30


def _route(B: int, N: int) -> str:
    if ROUTE != "auto":
        return ROUTE
    return "materialised" if (N <= 4096 and B * N * N * 4 <= MATERIALISED_MAX_BYTES) else "fused"


def _feature_normalize(x):
    if x.is_cuda and x.dtype == torch.float32:
        return ops.feature_normalize(x, _EPS)
    return x / (torch.norm(x, 2, 1, keepdim=True) + _EPS)


class ContextualLoss_forward(nn.Module):
    """Drop-in for the reference class; input feature maps [B, C, h, w] (or [B, C, N])."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    def forward(self, X_features, Y_features, h=0.1, feature_centering=True):
        B, C = X_features.shape[:2]
        if feature_centering:                                                                  # :108-114
            if getattr(self.opt, "PONO", False):
                mu = Y_features.mean(dim=1).unsqueeze(dim=1)
            else:
                mu = Y_features.reshape(B, C, -1).mean(dim=-1).reshape(B, C, *([1] * (Y_features.dim() - 2)))
            X_features, Y_features = X_features - mu, Y_features - mu
        Xn = _feature_normalize(X_features).reshape(B, C, -1)                                  # :115-116
        Yn = _feature_normalize(Y_features).reshape(B, C, -1)
        if Xn.is_cuda and Xn.dtype == torch.float32 and _route(B, Yn.shape[2]) == "materialised":
            if Yn.shape[2] > 4096:
                raise ValueError(f"contextual.ROUTE = 'materialised' supports at most 4096 key positions (got {Yn.shape[2]})")
            cx = ops.contextual_rows(ops.corr_materialize(Xn.contiguous(), Yn.contiguous(), 1.0), h, 1e-3)
            return -torch.log(cx.mean(dim=1))
        if Xn.is_cuda and Xn.dtype == torch.float32:
            # K22: cosine tiles -> row maximum -> per-row temperature -> sums, all in registers; any N, any C, nothing [N, N] in HBM
            cx = ops.contextual_cx(Xn.contiguous(), Yn.contiguous(), h, 1e-3)                  # :121-132
            return -torch.log(cx.mean(dim=1))                                                  # :132-133
        # CPU / non-fp32 tensors: the reference's formulation in torch (the CPU parity test and the fp64 arbiter of the GPU tests)
        d = 1 - torch.matmul(Xn.permute(0, 2, 1), Yn)
        d_norm = d / (torch.min(d, dim=-1, keepdim=True)[0] + 1e-3)
        w = torch.exp((1 - d_norm) / h)
        A = w / torch.sum(w, dim=-1, keepdim=True)
        return -torch.log(torch.mean(torch.max(A, dim=-1)[0], dim=1))


def install_contextual_loss_into_reference(models_module):
    """`models.networks.ContextualLoss.ContextualLoss_forward` -> this class (pix2pix_model.py builds it by name)."""
    import importlib
    mod = importlib.import_module(models_module.__name__ + ".ContextualLoss")
    mod.ContextualLoss_forward = ContextualLoss_forward
    return ContextualLoss_forward
