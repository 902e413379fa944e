"""Debug: the transposing f16 hi/lo split on the shapes of a benchmark step (event-bracketed averages)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
out = {}
for name, shape, cpad in (("theta [8,256,4096]", (8, 256, 4096), 256), ("dout [8,154,4096] -> 160", (8, 154, 4096), 160)):
    x = torch.randn(*shape, device="cuda", generator=g)
    am = ops.absmax(x)
    for _ in range(3): ops.split_f16(x, True, cpad=cpad, amax=am)
    with ops.KernelTimer() as kt:
        for _ in range(50): ops.split_f16(x, True, cpad=cpad, amax=am)
    out[name] = round(kt.summary()["split_f16"]["avg_ms"] * 1e3, 2)
print(out, "us")
