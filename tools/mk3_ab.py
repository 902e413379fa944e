"""Debug: the two match_kernel-3 shapes that matter for K19 (cfg2' no cycle, cfg3 as written with the cycle term), K19 kernel times."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
from cocosnet_amd.hot_path import HotPathConfig as C, correspondence_hot_path
def run(name, B, nc, seg_float, cfg):
    g = torch.Generator(device="cuda").manual_seed(0)
    th = torch.randn(B, 256, 64, 64, device="cuda", generator=g).requires_grad_(True)
    ph = (0.3 * th.detach() + torch.randn(B, 256, 64, 64, device="cuda", generator=g)).requires_grad_(True)
    img = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    if seg_float:
        seg = torch.rand(B, nc, 256, 256, device="cuda", generator=g)
    else:
        seg = torch.zeros(B, nc, 256, 256, device="cuda").scatter_(1, torch.randint(0, nc, (B, 1, 256, 256), device="cuda", generator=g), 1.0)
    cot = {}
    def step():
        th.grad = None; ph.grad = None
        o = correspondence_hot_path(th, ph, img, img, seg, seg, cfg)
        if not cot:
            cot.update({k: torch.randn(v.shape, device="cuda", generator=g) for k, v in o.items()})
        torch.autograd.backward([o[k] for k in sorted(o)], [cot[k] for k in sorted(o)])
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    with ops.KernelTimer() as kt:
        for _ in range(3): step()
    ks = {k: round(v["total_ms"] / 3, 3) for k, v in kt.summary().items() if k.startswith("box3")}
    print(name, round(dt * 1e3, 3), ks, flush=True)
run("cfg2'", 8, 151, False, C(match_kernel=3, PONO_C=True, warp_mask_losstype="direct", isTrain=True))
run("cfg3w", 16, 15, True, C(match_kernel=3, PONO_C=True, warp_bilinear=True, warp_cycle_w=1.0, isTrain=True))
