"""Timing of K16 against the framework's convolution at the ResidualBlock shape: python tools/conv_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from cocosnet_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [(B, 407, 66, 66, 407, 3, 1, 0)]


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (b, ci, h, w, co, k, s, p) in shapes:
    x = torch.randn(b, ci, h, w, device="cuda", requires_grad=True)
    wt = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).requires_grad_(True)
    y = ops.conv2d(x, wt, None, s, p)
    go = torch.randn_like(y)
    flops = 2.0 * y.numel() * ci * k * k
    t_f = timeit(lambda: ops.conv2d(x.detach(), wt.detach(), None, s, p))
    t_fb = timeit(lambda: torch.autograd.grad(ops.conv2d(x, wt, None, s, p), (x, wt), go))
    r_f = timeit(lambda: F.conv2d(x.detach(), wt.detach(), None, s, p))
    r_fb = timeit(lambda: torch.autograd.grad(F.conv2d(x, wt, None, s, p), (x, wt), go))
    print(f"{(b, ci, h, w, co, k, s, p)}: K16 fwd {t_f:.3f} ms ({flops / t_f / 1e9:.0f} TFLOP/s) fwd+bwd {t_fb:.3f} ms "
          f"({3 * flops / t_fb / 1e9:.0f}) | torch fwd {r_f:.3f} ms fwd+bwd {r_fb:.3f} ms", flush=True)
