"""TEST INFRASTRUCTURE (build container only): tests/golden/contextual_*.npz by RUNNING THE REFERENCE's own
`ContextualLoss_forward` (models/networks/ContextualLoss.py:83-137), imported read-only through oracle/ref_harness.py.

    python -m oracle.make_contextual_golden

Each fixture holds seeded feature maps X, Y [B, C, h, w] (fp32), the per-sample loss the reference class returns for them and the
gradient of its sum w.r.t. X as the reference's autograd gives it (Y is detached by the reference's caller, pix2pix_model.py:197-201).
The GPU parity test (tests/test_gpu_baseline_sizes.py) and the CPU test (tests/test_oracle_golden.py) read them; /root/reference is
not needed at test time.  Nothing is copied from the reference: only the numbers it produces are committed.
"""
from __future__ import annotations

import importlib
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import ref_harness as rh

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

#: name -> (B, C, h, w, PONO, h bandwidth, seed): VGG-like shapes scaled down; ragged positions / channels; both centrings
CASES = {
    "contextual_pono": (2, 64, 12, 12, True, 0.1, 11),
    "contextual_nopono": (2, 48, 9, 7, False, 0.1, 12),
    "contextual_h05": (1, 96, 16, 16, True, 0.5, 13),
}


def make_inputs(name):
    B, C, h, w, pono, bw, seed = CASES[name]
    g = torch.Generator().manual_seed(seed)
    Y = torch.randn(B, C, h, w, generator=g) + 0.3
    perm = torch.randperm(h * w, generator=g)
    X = 0.6 * Y.reshape(B, C, -1)[:, :, perm].reshape(B, C, h, w) + torch.randn(B, C, h, w, generator=g)
    return X, Y, pono, bw


def run_reference_case(name):
    rh.load_reference()
    ref_mod = importlib.import_module("models.networks.ContextualLoss")
    X, Y, pono, bw = make_inputs(name)
    x = X.clone().requires_grad_(True)
    loss = ref_mod.ContextualLoss_forward(SimpleNamespace(PONO=pono))(x, Y, h=bw)
    loss.sum().backward()
    return dict(X=X.numpy(), Y=Y.numpy(), pono=np.array(pono), h=np.array(bw, dtype=np.float64), loss=loss.detach().numpy(),
                dX=x.grad.numpy())


def main():
    for name in CASES:
        out = run_reference_case(name)
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **out)
        print(name, {k: getattr(v, "shape", v) for k, v in out.items()}, "loss", out["loss"])


if __name__ == "__main__":
    main()
