"""Time K0 (theta/phi 1x1 projections) on the fp32-MFMA GEMM against torch's conv2d (MIOpen / rocBLAS fp32).
Usage (GPU box): python tools/proj_bench.py [B Cin Cout h w]"""
import sys
import torch
import torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocosnet_amd import ops

B, Cin, Cout, h, w = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (8, 407, 256, 64, 64)
dev = torch.device("cuda:0")
x = torch.randn(B, Cin, h, w, device=dev, requires_grad=True)
wt = (torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05).requires_grad_(True)
b = torch.randn(Cout, device=dev, requires_grad=True)
g = torch.randn(B, Cout, h, w, device=dev)


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fwd_bwd(f):
    def run():
        x.grad = wt.grad = b.grad = None
        f(x, wt, b).backward(g)
    return run


NO_TORCH = os.environ.get("COCOS_BENCH_NO_TORCH") == "1"     # (MIOpen's first-call search costs seconds of GPU time)
with torch.no_grad():
    t_f_ours = bench(lambda: ops.proj1x1(x, wt, b))
    t_f_ref = float("nan") if NO_TORCH else bench(lambda: F.conv2d(x, wt, b))
t_ours = bench(fwd_bwd(ops.proj1x1))
t_ref = float("nan") if NO_TORCH else bench(fwd_bwd(F.conv2d))
gf = 2.0 * B * Cin * Cout * h * w / 1e9
print(f"proj1x1 B={B} Cin={Cin} Cout={Cout} {h}x{w}: fwd ours {t_f_ours*1e3:.0f} us ({gf/t_f_ours:.1f} TF/s... GF/ms) "
      f"torch {t_f_ref*1e3:.0f} us | fwd+bwd ours {t_ours*1e3:.0f} us torch {t_ref*1e3:.0f} us")
with ops.KernelTimer() as kt:
    for _ in range(10):
        fwd_bwd(ops.proj1x1)()
print({k: round(v["avg_ms"] * 1e3) for k, v in kt.summary().items()}, "us")
