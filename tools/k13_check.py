"""K13 (ops.instnorm_prelu) and the framework's fp32 InstanceNorm + leaky_relu against fp64, forward and input gradient,
on planes of 16x16 .. 64x64, for well- and ill-conditioned planes (|mean| >> std)."""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from cocosnet_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
for (B, C, H) in [(2, 512, 16), (2, 256, 32), (2, 64, 64)]:
    for cond, mk in {"unit": lambda s: torch.randn(s, device="cuda", generator=g),
                     "mean5_std1e-2": lambda s: 5.0 + 1e-2 * torch.randn(s, device="cuda", generator=g),
                     "tiny_1e-4": lambda s: 1e-4 * torch.randn(s, device="cuda", generator=g),
                     "per_plane_scale": lambda s: torch.randn(s, device="cuda", generator=g) * torch.pow(10.0, -6 * torch.rand(s[0], s[1], 1, 1, device="cuda", generator=g))}.items():
        x = mk((B, C, H, H))
        dy = torch.randn(B, C, H, H, device="cuda", generator=g)
        for slope in (0.2, 1.0):
            w = torch.full((1,), slope, device="cuda")
            xa = x.clone().requires_grad_(True)
            ya = ops.instnorm_prelu(xa, None, w, 1e-5)
            ya.backward(dy)
            xb = x.clone().requires_grad_(True)
            yb = F.leaky_relu(F.instance_norm(xb, eps=1e-5), slope)
            yb.backward(dy)
            xc = x.double().requires_grad_(True)
            yc = F.leaky_relu(F.instance_norm(xc, eps=1e-5), slope)
            yc.backward(dy.double())
            print(json.dumps({"shape": [B, C, H, H], "x": cond, "slope": slope, "k13_y": rel(ya, yc), "k13_dx": rel(xa.grad, xc.grad),
                              "fw_y": rel(yb, yc), "fw_dx": rel(xb.grad, xc.grad)}), flush=True)
