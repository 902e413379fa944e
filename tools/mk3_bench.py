"""Time the hot path with match_kernel=3 (reference default): box-filter path vs explicit unfold."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
B, S, nc = 8, 256, 151
g = torch.Generator(device="cuda").manual_seed(0)
th = torch.randn(B, 256, 64, 64, device="cuda", generator=g).requires_grad_(True)
ph = (0.3 * th.detach() + torch.randn(B, 256, 64, 64, device="cuda", generator=g)).requires_grad_(True)
img = torch.rand(B, 3, S, S, device="cuda", generator=g) * 2 - 1
lab = torch.randint(0, nc, (B, 1, S, S), device="cuda", generator=g)
seg = torch.zeros(B, nc, S, S, device="cuda").scatter_(1, lab, 1.0)
def run(cfg):
    def step():
        th.grad = None; ph.grad = None
        o = correspondence_hot_path(th, ph, img, img, seg, seg, cfg)
        (o["warp_out"].sum() + o["warp_mask"].pow(2).sum()).backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with ops.KernelTimer() as kt:
        for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    return dt, kt.summary()
for name, cfg in (("mk3 box-filter path (PONO_C)", HotPathConfig(match_kernel=3, PONO_C=True, warp_mask_losstype="direct")),
                  ("mk3 explicit unfold path (no PONO_C)", HotPathConfig(match_kernel=3, PONO_C=False, warp_mask_losstype="direct")),
                  ("mk1 fused path", HotPathConfig(match_kernel=1, PONO_C=True, warp_mask_losstype="direct"))):
    dt, ks = run(cfg)
    print(f"{name}: {dt*1e3:.2f} ms/step  ({B/dt:.0f} images/s)")
    for k, v in ks.items(): print(f"      {k:36s} {v['avg_ms']:.3f} ms x{v['calls']//5}")
