"""TEST INFRASTRUCTURE — fp64 restatement of the reference's contextual loss (models/networks/ContextualLoss.py:93-137,
`ContextualLoss_forward.forward`; util.feature_normalize: util/util.py:31-34) in torch, differentiable by autograd.

Only tests/ (and tools/ that print error tables) may import this file: the product (cocosnet_amd/contextual.py) computes the same
quantity on the HIP kernels (K1 + K22, or K3 + K15) and never touches it.  Pinned: tests/test_oracle_golden.py checks it against
tests/golden/contextual_*.npz — outputs and autograd gradients of the REFERENCE's own class (oracle/make_contextual_golden.py runs
/root/reference) — so that the at-size GPU tests (N = 4096 / 12288, where the reference itself would need five [N, N] fp32 matrices
per sample) have an arbiter that is not the product module.
"""
from __future__ import annotations

import sys

import torch

EPS = sys.float_info.epsilon          # util/util.py:31-34 feature_normalize: x / (||x||_2 over channels + sys.float_info.epsilon)


def feature_normalize(x: torch.Tensor) -> torch.Tensor:
    """util/util.py:31-34."""
    return x / (torch.norm(x, 2, 1, keepdim=True) + EPS)


def normalised_features(X: torch.Tensor, Y: torch.Tensor, pono: bool, feature_centering: bool = True):
    """ContextualLoss.py:103-116: centring by Y's mean (over channels with --PONO, over positions without) and normalisation;
    [B, C, ...] -> two [B, C, N] tensors."""
    B, C = X.shape[:2]
    if feature_centering:                                                                     # :108-114
        if pono:
            mu = Y.mean(dim=1).unsqueeze(dim=1)                                               # :110-111
        else:
            mu = Y.reshape(B, C, -1).mean(dim=-1).reshape(B, C, *([1] * (Y.dim() - 2)))       # :113-114
        X, Y = X - mu, Y - mu
    return feature_normalize(X).reshape(B, C, -1), feature_normalize(Y).reshape(B, C, -1)     # :115-116


def cx_rows(Xn: torch.Tensor, Yn: torch.Tensor, h: float = 0.1, eps: float = 1e-3) -> torch.Tensor:
    """ContextualLoss.py:124-134 up to the row maxima: max_j A_ij [B, N_x] of the normalised features [B, C, N]."""
    d = 1 - torch.matmul(Xn.permute(0, 2, 1), Yn)                                             # :125-126 cosine distance
    d_norm = d / (torch.min(d, dim=-1, keepdim=True)[0] + eps)                                # :130
    w = torch.exp((1 - d_norm) / h)                                                           # :133
    A = w / torch.sum(w, dim=-1, keepdim=True)                                                # :134
    return torch.max(A, dim=-1)[0]                                                            # :137 (inner)


def contextual_loss(X: torch.Tensor, Y: torch.Tensor, h: float = 0.1, pono: bool = True, feature_centering: bool = True) -> torch.Tensor:
    """The per-sample loss [B] (ContextualLoss.py:93-141).  Feed fp64 tensors for the arbiter's precision; any device."""
    Xn, Yn = normalised_features(X, Y, pono, feature_centering)
    return -torch.log(torch.mean(cx_rows(Xn, Yn, h), dim=1))                                  # :137-138
