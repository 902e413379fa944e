"""Times K23 / K24 / the weight-gradient pair of the two projections at the bench shape (B = 8, Cin = 407, 64 x 64), one library
per run (COCOS_LIB_PATH selects ablation builds).  Usage (GPU box): python tools/proj_pair_bench.py [B Cin h w]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocosnet_amd import ops

B, Cin, h, w = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 407, 64, 64)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
leaves = [mk(B, Cin, h, w), mk(256, Cin, 1, 1) / Cin ** 0.5, mk(256) * 0.1, mk(B, Cin, h, w), mk(256, Cin, 1, 1) / Cin ** 0.5, mk(256) * 0.1]
for t in leaves:
    t.requires_grad_(True)
d1, d2 = mk(B, 256, h * w) * 1e-3, mk(B, 256, h * w) * 1e-3


def step():
    for t in leaves:
        t.grad = None
    planes = ops.OperandPlanes()
    qn, kn = ops.proj_center_l2norm_planes_pair(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), 1, planes)
    torch.autograd.backward([qn, kn], [d1, d2])


for _ in range(5):
    step()
torch.cuda.synchronize()
with ops.KernelTimer() as kt:
    for _ in range(30):
        step()
s = kt.summary()
print(os.environ.get("COCOS_LIB_PATH", "default"), {k: round(v["avg_ms"] * 1e3, 1) for k, v in s.items()}, "us")
