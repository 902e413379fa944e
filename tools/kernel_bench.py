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
def train_step():
    q.grad = None; k.grad = None
    ops.corr_softmax_warp(q, k, v, 100.0).backward(go)

def infer_step():
    with torch.no_grad():
        ops.corr_softmax_warp(q, k, v, 100.0)

for name, fn in (("train", train_step), ("infer", infer_step)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        for _ in range(a.iters):
            fn()
    for tag, r in kt.summary().items():
        fl = {"corr_softmax_warp_fwd": 2.0 * N * N * (256 + Cv) * B,
              "corr_softmax_warp_bwd_query": 2.0 * N * N * (256 + Cv) * B,
              "corr_softmax_warp_bwd_key_from_ds": 2.0 * N * N * 256 * B}.get(tag)
        extra = f"  {fl / r['avg_ms'] / 1e9:6.1f} TF alg ({fl / r['avg_ms'] / 1e9 / 157.3 * 100:4.1f} % of fp32 MFMA peak)" if fl else ""
        print(f"{name:6s} {tag:36s} calls {r['calls']:3d} avg {r['avg_ms']:.4f} ms{extra}")
