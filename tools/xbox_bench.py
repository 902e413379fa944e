"""Debug: time of the x-box correlation GEMM alone (cfg2' shape: B=8, 64x64 grid; cfg5: B=2, 128x128 grid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
out = {}
for name, B, fh, fw in (("cfg2'", 8, 64, 64), ("cfg5", 2, 128, 128)):
    g = torch.Generator(device="cuda").manual_seed(0)
    th = torch.randn(B, 256, fh, fw, device="cuda", generator=g)
    ph = torch.randn(B, 256, fh, fw, device="cuda", generator=g)
    with torch.no_grad():
        for _ in range(2): ops.box3_corr_xbox(th, ph)
        with ops.KernelTimer() as kt:
            for _ in range(5): ops.box3_corr_xbox(th, ph)
    out[name] = round(kt.summary()["box3_corr_xbox"]["avg_ms"], 4)
print(out)
