"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden ade_mk1    # some cases

For each case in oracle/golden_cases.py this imports /root/reference read-only
(oracle/ref_harness.py), builds the reference's own NoVGGCorrespondence, replaces the feature
producers (adaptors, residual blocks, theta/phi convs) by stubs that hand seeded tensors to the
hot path, calls the reference's forward() (correspondence.py:222-374) and stores its outputs —
and, where the case asks, the gradients its autograd gives w.r.t. theta_raw / phi_raw.
Nothing is copied from the reference; only the numbers it produces are committed.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn

from . import golden_cases as gc
from . import ref_harness as rh

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests",
                          "golden")


class _Const(nn.Module):
    """Stub producer: ignores its inputs and returns a fixed tensor."""

    def __init__(self, value):
        super().__init__()
        self.value = value

    def forward(self, *args, **kwargs):
        return self.value


def run_reference_case(name: str):
    c = gc.CASES[name]
    inp = gc.make_inputs(name)
    opt = rh.make_opt(semantic_nc=c["nc"], **c["opt"])
    net = rh.build_reference_corr(opt)
    net.train(bool(opt.isTrain))

    theta_raw = torch.from_numpy(inp.theta_raw).requires_grad_(True)
    phi_raw = torch.from_numpy(inp.phi_raw).requires_grad_(True)
    feat = torch.zeros(c["B"], 256, inp.fh, inp.fh)
    net.adaptive_model_seg = _Const(feat)
    net.adaptive_model_img = _Const(feat)
    net.layer = nn.Identity()
    net.theta = _Const(theta_raw)
    net.phi = _Const(phi_raw)

    t = torch.from_numpy
    res = net(t(inp.ref_img), t(inp.real_img), t(inp.seg_map), t(inp.ref_seg_map),
              **c.get("fwd", {}))
    outs = {"corr": res} if torch.is_tensor(res) else dict(res)
    outs = {k: v for k, v in outs.items() if torch.is_tensor(v)}

    store = {}
    if c.get("grads"):
        G = gc.grad_weights(name, {k: tuple(v.shape) for k, v in outs.items()})
        loss = sum((outs[k] * t(G[k])).sum() for k in outs)
        loss.backward()
        for key, g in (("theta_raw", theta_raw.grad), ("phi_raw", phi_raw.grad)):
            a = g.numpy().astype(np.float32)
            if c.get("sample", 0):     # big grids: a seeded sample of the gradient (golden_cases.grad_error)
                store["gsample__" + key] = a.reshape(-1)[gc.sample_index(name, "grad:" + key, a.size, c["sample"])]
            else:
                store["grad__" + key] = a
    n_sample = c.get("sample", 0)
    for k, v in outs.items():
        a = v.detach().numpy().astype(np.float32)
        store["shape__" + k] = np.asarray(a.shape, dtype=np.int64)
        if n_sample:
            idx = gc.sample_index(name, k, a.size, n_sample)
            store["sample__" + k] = a.reshape(-1)[idx]
            store["sum__" + k] = np.asarray(a.astype(np.float64).sum())
        else:
            store["out__" + k] = a
    return store


def main(argv):
    names = argv or sorted(gc.CASES)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name in names:
        store = run_reference_case(name)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **store)
        kb = os.path.getsize(path) / 1024
        print(f"{name}: {sorted(store)} -> {path} ({kb:.0f} KiB)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
