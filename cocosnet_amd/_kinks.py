"""Branch-pattern tape for the module's piecewise-linear sites (LeakyReLU / PReLU / ReLU after a norm, 2x2 max-pooling).

Why it exists: a parameter gradient UPSTREAM of `InstanceNorm -> LeakyReLU` is a discontinuous function of
the features — an element whose normalised value lies within rounding of zero takes the other branch in an fp32 evaluation than in
the fp64 arbiter, and ONE such element moves d weight by up to 2e-2 of its range, for any fp32 implementation.  To hold those
gradients to north_star's 1e-3 the comparison has to be made on the SAME branch pattern: the fp64 copy RECORDS its pattern at every
site, the fp32 arms REPLAY it (the site runs its production kernel with the activation's slope at 1 — the linear part — and the
recorded pattern is applied as a multiplier), and the number of elements whose own branch differed is counted per arm.

Only tests set `TAPE`; with `TAPE is None` (always, in production) every site is exactly its fused production call.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

#: the active tape (tests only); None = production
TAPE = None


class KinkTape:
    def __init__(self):
        self.masks: list[torch.Tensor] = []
        self.mode = "record"
        self.pos = 0
        self.flips: list[int] = []          # per site, replay mode: elements whose own branch differs from the recorded one
        self.elements = 0

    def rewind(self, mode: str):
        assert mode in ("record", "replay")
        self.mode, self.pos, self.flips, self.elements = mode, 0, [], 0
        if mode == "record":
            self.masks = []
        return self

    # ---- a LeakyReLU / PReLU / ReLU site: `lin` = the site's value at slope 1, `slope` a float or a (PReLU) tensor ----
    def act(self, lin: torch.Tensor, slope):
        own = lin > 0
        if self.mode == "record":
            self.masks.append(own)
            mask = own
        else:
            mask = self.masks[self.pos]
            self.pos += 1
            assert mask.shape == own.shape, (mask.shape, own.shape, self.pos)
            self.flips.append(int((mask != own).sum()))
            self.elements += own.numel()
        if torch.is_tensor(slope):
            a = slope.to(lin.dtype)
            a = a.reshape(1, -1, *([1] * (lin.dim() - 2))) if a.numel() > 1 else a.reshape(())
            return lin * torch.where(mask, torch.ones((), dtype=lin.dtype, device=lin.device), a)
        return lin * torch.where(mask, 1.0, float(slope)).to(lin.dtype)

    # ---- a max_pool2d(x, [2, 2]) site: the recorded ARGMAX is gathered --------------------------------------------------
    def pool2(self, x: torch.Tensor):
        _, own = F.max_pool2d(x, [2, 2], return_indices=True)
        if self.mode == "record":
            self.masks.append(own)
            idx = own
        else:
            idx = self.masks[self.pos]
            self.pos += 1
            self.flips.append(int((idx != own).sum()))
            self.elements += own.numel()
        B, C, H, W = x.shape
        return x.reshape(B, C, H * W).gather(2, idx.reshape(B, C, -1)).reshape(idx.shape)


def active() -> bool:
    return TAPE is not None
