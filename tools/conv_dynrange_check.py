"""Accuracy of K16's gradients against torch-fp64 when dy has a WIDE dynamic range (the gradient behind a softmax at
T = 0.01 spans many decades inside one tensor): per-tensor power-of-two scales put small elements into f16's subnormal
range.  Prints max-norm relative errors of dx / dw for log-normal dy of growing spread, gather kernels (Cin < 32) and
the NHWC kernels.   python tools/conv_dynrange_check.py"""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from cocosnet_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for name, (B, Cin, H, W, Cout, k, s, p) in {"adaptor_layer1": (2, 3, 64, 64, 64, 3, 1, 1), "adaptor_layer2": (2, 64, 64, 64, 128, 4, 2, 1),
                                            "nhwc_128": (2, 128, 32, 32, 128, 3, 1, 1)}.items():
    x = torch.rand(B, Cin, H, W, device="cuda", generator=g) * 2 - 1
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * 0.1
    for spread in (0.0, 2.0, 4.0, 6.0, 8.0):
        xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.conv2d(xx, ww, None, s, p, 1)
        dy = torch.randn(y.shape, device="cuda", generator=g) * torch.pow(10.0, -spread * torch.rand(y.shape, device="cuda", generator=g))
        y.backward(dy)
        x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
        F.conv2d(x64, w64, None, s, p).backward(dy.double())
        rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        print(json.dumps({"layer": name, "decades": spread, "dx": rel(xx.grad, x64.grad), "dw": rel(ww.grad, w64.grad)}), flush=True)
