"""K13 (InstanceNorm + skip + PReLU) forward / backward at the module's plane sizes.  python tools/instnorm_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops


def timeit(f, n=20):
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


for name, shape, skip in (("ResidualBlock 8x407x64x64 + skip", (8, 407, 64, 64), True), ("same, no skip", (8, 407, 64, 64), False),
                          ("adaptor 8x256x128x128", (8, 256, 128, 128), False), ("adaptor 8x64x256x256", (8, 64, 256, 256), False)):
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    r = torch.randn(*shape, device="cuda", requires_grad=True) if skip else None
    w = torch.full((1,), 0.25, device="cuda", requires_grad=True)
    y = ops.instnorm_prelu(x, r, w)
    go = torch.randn_like(y)
    mb = x.numel() * 4 / 1e6
    with ops.KernelTimer() as kt:                      # HIP events around each ABI call: the Python side of autograd is ~0.15 ms per iteration
        for _ in range(10):
            torch.autograd.grad(ops.instnorm_prelu(x, r, w), (x, w) + ((r,) if skip else ()), go)
    k = {n: v["total_ms"] / 10 for n, v in kt.summary().items()}
    tf, tb = k["instnorm_prelu_fwd"], k["instnorm_prelu_bwd"]
    nf, nb = (3 if skip else 2), (5 if skip else 3)          # tensors moved: fwd x (+res) y; bwd x dy (+res) dx (+dres)
    print(json.dumps({"shape": name, "fwd_ms": round(tf, 4), "fwd_TBps": round(nf * mb / tf / 1e3, 2), "bwd_ms": round(tb, 4),
                      "bwd_TBps": round(nb * mb / tb / 1e3, 2), "other_calls_ms": {n: round(v, 4) for n, v in k.items() if "instnorm" not in n}}),
          flush=True)
