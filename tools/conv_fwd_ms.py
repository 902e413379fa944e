"""forward-only timing of K16 at one shape (ablation runs): python tools/conv_fwd_ms.py B Cin H W Cout k s p"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
a = [int(v) for v in sys.argv[1:9]] if len(sys.argv) >= 9 else [8, 407, 66, 66, 407, 3, 1, 0]
b, ci, h, w, co, k, s, p = a
x = torch.randn(b, ci, h, w, device="cuda")
wt = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
f = lambda: ops.conv2d(x, wt, None, s, p)
for _ in range(3): y = f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"{t:.3f} ms  {2.0 * y.numel() * ci * k * k / t / 1e9:.0f} TFLOP/s")
