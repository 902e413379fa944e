"""K16 in its two flavours (f16x3 three-term split / bf16 one term) at the ResidualBlock shape and a PatchGAN layer, and the
whole NoVGGCorrespondence module (bench.py --scope netcorr step) with either: python tools/conv_flavours_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops


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


for name, (b, ci, h, w, co, k, s, p) in (("ResidualBlock 407->407 3x3 on 66x66 (B=8)", (8, 407, 66, 66, 407, 3, 1, 0)),
                                         ("PatchGAN 128->256 k4 s2 on 64x64 (B=8)", (8, 128, 64, 64, 256, 4, 2, 2))):
    x = torch.randn(b, ci, h, w, device="cuda", requires_grad=True)
    wt = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).requires_grad_(True)
    rec = {"shape": name}
    for prec in ("f16x3", "bf16"):
        ops.CONV_PRECISION = prec
        y = ops.conv2d(x, wt, None, s, p)
        go = torch.randn_like(y)
        flops = 2.0 * y.numel() * ci * k * k
        t_f = timeit(lambda: ops.conv2d(x.detach(), wt.detach(), None, s, p))
        t_fb = timeit(lambda: torch.autograd.grad(ops.conv2d(x, wt, None, s, p), (x, wt), go))
        rec[prec] = {"fwd_ms": round(t_f, 3), "fwd_alg_tflops": round(flops / t_f / 1e9, 1), "fwd_bwd_ms": round(t_fb, 3)}
    print(json.dumps(rec), flush=True)
