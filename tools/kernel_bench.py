"""Run only the hot kernels at the benchmark shape (B=8, HW=4096, K=256, Cv=154) a few times.
Meant to sit under `rocprofv3 --pmc ...` (counter passes) or `--kernel-trace --stats`."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--cv", type=int, default=154)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
B, N, Cv = a.batch, a.n, a.cv
nrm = lambda x: (x - x.mean(1, keepdim=True)) / (x - x.mean(1, keepdim=True)).norm(dim=1, keepdim=True)
q = nrm(torch.randn(B, 256, N, device=dev, generator=g)).requires_grad_(True)
k = nrm(0.2 * q.detach() + torch.randn(B, 256, N, device=dev, generator=g)).requires_grad_(True)
v = torch.rand(B, Cv, N, device=dev, generator=g) * 2 - 1
go = torch.randn(B, Cv, N, device=dev, generator=g)
with ops.KernelTimer() as kt:
    for _ in range(a.iters):
        q.grad = None; k.grad = None
        ops.corr_softmax_warp(q, k, v, 100.0).backward(go)
    with torch.no_grad():
        for _ in range(a.iters):
            ops.corr_softmax_warp(q, k, v, 100.0)          # inference flavour (no logits store)
for tag, r in kt.summary().items():
    print(f"{tag:40s} calls {r['calls']:3d} avg {r['avg_ms']:.4f} ms")
