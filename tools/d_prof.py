"""Kernel breakdown of config 3's PatchGAN step (MultiscaleDiscriminator, B = 16, 256 x 256, forward + backward).
Usage (GPU box): rocprofv3 --kernel-trace --stats ... -- python tools/d_prof.py   |   python tools/d_prof.py (KernelTimer table)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocosnet_amd import ops, translation as tl

dev = torch.device("cuda:0")
opt = tl.celebahq_edge_train_options()
B, IMG = 16, 256
g = torch.Generator(device=dev).manual_seed(77)
seg = torch.rand(B, 15, IMG, IMG, device=dev, generator=g)
real = torch.rand(B, 3, IMG, IMG, device=dev, generator=g) * 2 - 1
torch.manual_seed(0)
D = tl.MultiscaleDiscriminator(opt).to(dev)
D.init_weights(opt.init_type, opt.init_variance)
D.train()
dp = list(D.parameters())
def d_step():
    for p in dp:
        p.grad = None
    res = D(torch.cat((seg, real), 1))[0]
    torch.autograd.backward([r[-1] for r in res], [torch.ones_like(r[-1]) for r in res])
for _ in range(3):
    d_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    d_step()
torch.cuda.synchronize()
print("D step ms", (time.perf_counter() - t0) / 10 * 1e3)
with ops.KernelTimer() as kt:
    d_step()
s = kt.summary()
print({k: (v["calls"], round(v["total_ms"], 3)) for k, v in sorted(s.items(), key=lambda kv: -kv[1]["total_ms"])})
print(D)
