"""Does a second workgroup per CU help K19?  The CVB = 1 instantiation (Cv <= 32: 47 KB of LDS, 240 registers) fits two workgroups
per CU; its grid is B * 32 workgroups on a 64 x 64 grid.  B = 8: one workgroup per CU; B = 16: two.  If time(B = 16) is well below
2 x time(B = 8), the kernel's skeleton (T loads, softmax VALU, V staging, barrier) is latency-bound and a second wave per SIMD
hides it; if it is ~2x, the skeleton is throughput-bound (L2 / LDS / issue) and re-tiling K19 for two waves per SIMD buys nothing."""
import sys
import torch
sys.path.insert(0, ".")
from cocosnet_amd import ops

def run(B, Cv, need_v):
    g = torch.Generator(device="cuda").manual_seed(B)
    th = torch.randn(B, 256, 64, 64, device="cuda", generator=g).requires_grad_(True)
    ph = (0.3 * th.detach() + torch.randn(B, 256, 64, 64, device="cuda", generator=g)).requires_grad_(True)
    v = torch.randn(B, Cv, 4096, device="cuda", generator=g).requires_grad_(need_v)
    from cocosnet_amd.hot_path import _unfold3_stats
    kc = 256.0 * 9
    def step():
        th.grad = ph.grad = None
        mu, a = _unfold3_stats(th, kc); nu, b = _unfold3_stats(ph, kc)
        sink = ops.Box3GradSink()
        t = ops.box3_corr_xbox(th, ph, sink)
        o = ops.box3_softmax_warp(t, mu, a, nu, b, v, 64, 64, kc, 100.0, sink=sink)
        o.backward(torch.ones_like(o))
    for _ in range(3): step()
    with ops.KernelTimer() as kt:
        for _ in range(5): step()
    s = kt.summary()
    return {k: round(v["total_ms"] / 5, 4) for k, v in s.items() if k.startswith("box3_softmax")}

for Cv in (3, 32):
    for B in (8, 16, 24, 32):
        print("Cv", Cv, "B", B, run(B, Cv, False), flush=True)
print("Cv 154 B 8", run(8, 154, False))
print("Cv 154 B 16", run(16, 154, False))
