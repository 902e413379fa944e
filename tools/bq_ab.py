"""Debug: time of the K2 query backward (and the step) at four shapes — headline, cfg3 mk1 with the cycle term, cfg5 mk1, the netG Attention block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
from cocosnet_amd.hot_path import HotPathConfig as C, correspondence_hot_path
def run(name, B, size, nc, seg_float, cfg):
    g = torch.Generator(device="cuda").manual_seed(0)
    fh = size // cfg.down
    th = torch.randn(B, 256, fh, fh, device="cuda", generator=g).requires_grad_(True)
    ph = (0.3 * th.detach() + torch.randn(B, 256, fh, fh, device="cuda", generator=g)).requires_grad_(True)
    img = torch.rand(B, 3, size, size, device="cuda", generator=g) * 2 - 1
    if seg_float:
        seg = torch.rand(B, nc, size, size, device="cuda", generator=g)
    else:
        seg = torch.zeros(B, nc, size, size, device="cuda").scatter_(1, torch.randint(0, nc, (B, 1, size, size), device="cuda", generator=g), 1.0)
    cot = {}
    def step():
        th.grad = None; ph.grad = None
        o = correspondence_hot_path(th, ph, img, img, seg, seg, cfg)
        if not cot:
            cot.update({k: torch.randn(v.shape, device="cuda", generator=g) for k, v in o.items()})
        torch.autograd.backward([o[k] for k in sorted(o)], [cot[k] for k in sorted(o)])
    for _ in range(3): step()
    with ops.KernelTimer() as kt:
        for _ in range(6): step()
    print(name, round(kt.summary()["corr_softmax_warp_bwd_query"]["total_ms"] / 6, 4), end=" | ", flush=True)
run("cfg2", 8, 256, 151, False, C(match_kernel=1, PONO_C=True, warp_mask_losstype="direct", isTrain=True))
run("cfg3w", 16, 256, 15, True, C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_cycle_w=1.0, isTrain=True))
run("cfg5", 2, 512, 20, True, C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_patch=True, isTrain=True))
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(4, 32, 16384, device="cuda", generator=g).requires_grad_(True)
k = torch.randn(4, 32, 4096, device="cuda", generator=g).requires_grad_(True)
v = torch.randn(4, 128, 4096, device="cuda", generator=g).requires_grad_(True)
go = torch.randn(4, 128, 16384, device="cuda", generator=g)
def st():
    q.grad = k.grad = v.grad = None
    ops.softmax_attention(q, k, v, 1.0).backward(go)
for _ in range(2): st()
with ops.KernelTimer() as kt:
    for _ in range(5): st()
print("attn", round(kt.summary()["corr_softmax_warp_bwd_query"]["total_ms"] / 5, 4))
