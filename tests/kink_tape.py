"""Branch-pattern tape for the module's piecewise-linear sites (LeakyReLU / PReLU / ReLU after a norm, 2x2 max-pooling).

Why it exists: a parameter gradient UPSTREAM of `InstanceNorm -> LeakyReLU` is a discontinuous function of
the features — an element whose normalised value lies within rounding of zero takes the other branch in an fp32 evaluation than in
the fp64 arbiter, and ONE such element moves d weight by up to 2e-2 of its range, for any fp32 implementation.  To hold those
gradients to north_star's 1e-3 the comparison has to be made on the SAME branch pattern: the fp64 copy RECORDS its pattern at every
site, the fp32 arms REPLAY it (the site runs its production kernel with the activation's slope at 1 — the linear part — and the
recorded pattern is applied as a multiplier), and the number of elements whose own branch differed is counted per arm.

TEST INFRASTRUCTURE (round 6: moved out of the product package — VERDICT r5 weak 1b).  The product modules know nothing of
it: `install(tape)` monkeypatches the seven piecewise-linear sites (methods / functions of cocosnet_amd.producers, .spade,
.translation) with versions that run the site's production kernel at slope 1 and apply the tape on top, and restores them on exit.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

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


# ------------------------------------------------------------------------------------------------------------------------
# installation: the product's piecewise-linear sites, replaced for the duration of a `with install(tape):` block
# ------------------------------------------------------------------------------------------------------------------------
import contextlib


@contextlib.contextmanager
def install(tape):
    """Route every LeakyReLU / PReLU / ReLU / 2x2-max-pool site of the module scope through `tape` (None: nothing is patched)."""
    if tape is None:
        yield None
        return
    from torch import nn
    from cocosnet_amd import ops, producers as pr, spade as sp, translation as tl
    from cocosnet_amd.producers import reflect_conv

    saved = []

    def patch(obj, name, new):
        saved.append((obj, name, obj.__dict__[name] if name in obj.__dict__ else getattr(obj, name)))
        setattr(obj, name, new)

    # 1. Attention: the two max-pools (producers.Attention._pool)
    patch(pr.Attention, "_pool", staticmethod(tape.pool2))

    # 2. SPADEResnetBlock: norm -> LeakyReLU(0.2): the fused norm at slope 1, the recorded branches on top
    def norm_act(self, norm, x, seg):
        if self.slope != 1.0:
            return tape.act(norm(x, seg, slope=1.0), self.slope)
        return norm(x, seg, slope=self.slope)
    patch(pr.SPADEResnetBlock, "_norm_act", norm_act)

    # 3. AdaptiveFeatureGenerator: conv [+ InstanceNorm] + LeakyReLU
    orig_cna = pr.AdaptiveFeatureGenerator._conv_norm_act

    def conv_norm_act(self, layer, x, slope):
        if slope != 1.0:
            return tape.act(orig_cna(self, layer, x, 1.0), slope)
        return orig_cna(self, layer, x, slope)
    patch(pr.AdaptiveFeatureGenerator, "_conv_norm_act", conv_norm_act)

    # 4. ResidualBlock: K13 with a = 1 (its linear part), the recorded PReLU branches on top
    def residual_forward(self, x):
        if x.is_cuda and x.dtype == torch.float32:
            one = torch.ones_like(self.prelu.weight)
            y = tape.act(ops.instnorm_prelu(reflect_conv(self.padding1, self.conv1, x), None, one, self.bn1.eps), self.prelu.weight)
            return tape.act(ops.instnorm_prelu(reflect_conv(self.padding2, self.conv2, y), x, one, self.bn2.eps), self.prelu.weight)
        y = tape.act(self.bn1(self.conv1(self.padding1(x))), self.prelu.weight)
        return tape.act(self.bn2(self.conv2(self.padding2(y))) + x, self.prelu.weight)
    patch(pr.ResidualBlock, "forward", residual_forward)

    # 5. SPADE: the ReLU of mlp_shared (spade.shared_activation)
    orig_shared = sp.shared_activation

    def shared_activation(self, segmap):
        if isinstance(self.mlp_shared, nn.Sequential) and isinstance(self.mlp_shared[-1], nn.ReLU):
            actv = segmap
            for m in list(self.mlp_shared)[:-1]:
                actv = m(actv)
            return tape.act(actv, 0.0)
        return orig_shared(self, segmap)
    patch(sp, "shared_activation", shared_activation)

    # 6. SPADEGenerator: the LeakyReLU in front of conv_img
    patch(tl.SPADEGenerator, "_final_act", staticmethod(lambda x: tape.act(x, 2e-1)))

    # 7. PatchGAN: [conv (+ InstanceNorm)] + LeakyReLU(0.2)
    orig_cna_d = tl._ConvNormAct.forward

    def conv_norm_act_d(self, x):
        if not (len(self) == 2 and isinstance(self[1], nn.LeakyReLU)):
            return orig_cna_d(self, x)
        first = self[0]
        if (isinstance(first, nn.Sequential) and len(first) == 2 and type(first[1]) is nn.InstanceNorm2d and not first[1].affine
                and x.is_cuda and x.dtype == torch.float32 and ops.CONV_PRECISION != "torch"):
            y = first[0](x)
            if y.shape[2] * y.shape[3] <= 16384:      # K13 at slope 1 (its linear part), the recorded branches on top
                return tape.act(ops.instnorm_prelu(y, None, torch.ones(1, device=y.device), first[1].eps), self[1].negative_slope)
            return tape.act(first[1](y), self[1].negative_slope)
        return tape.act(first(x), self[1].negative_slope)
    patch(tl._ConvNormAct, "forward", conv_norm_act_d)

    try:
        yield tape
    finally:
        for obj, name, old in reversed(saved):
            setattr(obj, name, old)
