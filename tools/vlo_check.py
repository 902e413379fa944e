"""K2 split kernels with exactly-representable label channels (v_lo_mask = 1: lo plane of value blocks >= 1 skipped) vs a dense V."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from cocosnet_amd import ops
ops.PRECISION = "f16x3"
B, N = 8, 4096
g = torch.Generator(device="cuda").manual_seed(0)
nrm = lambda x: (x - x.mean(1, keepdim=True)) / (x - x.mean(1, keepdim=True)).norm(dim=1, keepdim=True)
q = nrm(torch.randn(B, 256, N, device="cuda", generator=g)).requires_grad_(True)
k = nrm(0.2 * q.detach() + torch.randn(B, 256, N, device="cuda", generator=g)).requires_grad_(True)
img = torch.rand(B, 3, N, device="cuda", generator=g) * 2 - 1
lab = torch.randint(0, 151, (B, 1, N), device="cuda", generator=g)
v_exact = torch.cat([img, torch.zeros(B, 151, N, device="cuda").scatter_(1, lab, 1.0)], 1).contiguous()
v_soft = torch.rand(B, 154, N, device="cuda", generator=g) * 2 - 1
go = torch.randn(B, 154, N, device="cuda", generator=g)
for name, v in (("exact labels", v_exact), ("dense V", v_soft), ("exact labels", v_exact), ("dense V", v_soft)):
    vh, vl, _ = ops.split_f16(v, False, amax=ops.absmax(v))
    print(name, "mask", bin(int(ops.f16_plane_block_mask(vl).view(torch.int32).item())))
    for _ in range(3):
        q.grad = None; k.grad = None
        ops.corr_softmax_warp(q, k, v, 100.0).backward(go)
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        for _ in range(20):
            q.grad = None; k.grad = None
            ops.corr_softmax_warp(q, k, v, 100.0).backward(go)
    s = kt.summary()
    print({t: round(r["avg_ms"], 4) for t, r in s.items() if "corr" in t})
