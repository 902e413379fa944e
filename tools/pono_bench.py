"""K9 (PONO + SPADE modulation + LeakyReLU) against the torch chain it replaces; prints time, GB/s of
ALGORITHMIC traffic (fwd 16 B/elem, bwd 28 B/elem) and the fraction of 8 TB/s.
Usage (GPU box): python tools/pono_bench.py [B C H W]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from cocosnet_amd import ops
from cocosnet_amd.producers import positional_norm


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(B, C, H, W):
    dev = torch.device("cuda:0")
    x, ga, be = (torch.randn(B, C, H, W, device=dev, requires_grad=True) for _ in range(3))
    g = torch.randn(B, C, H, W, device=dev)
    ours = lambda: ops.pono_spade(x, ga, be, 0.2)
    ref = lambda: F.leaky_relu(positional_norm(x) * (1 + ga) + be, 0.2)

    def fb(f):
        def go():
            x.grad = ga.grad = be.grad = None
            f().backward(g)
        return go
    with torch.no_grad():
        tf, tfr = bench(ours), bench(ref)
    tb, tbr = bench(fb(ours)), bench(fb(ref))
    el = B * C * H * W
    with ops.KernelTimer() as kt:
        fb(ours)()
    k = {n: v["avg_ms"] for n, v in kt.summary().items()}
    print(f"[{B},{C},{H},{W}] fwd ours {tf*1e3:.0f} us torch {tfr*1e3:.0f} us | fwd+bwd ours {tb*1e3:.0f} us "
          f"torch {tbr*1e3:.0f} us | kernels fwd {k['pono_spade_fwd']*1e3:.0f} us = "
          f"{16*el/k['pono_spade_fwd']/1e6:.0f} GB/s ({16*el/k['pono_spade_fwd']/1e6/8000:.2f} of 8 TB/s), "
          f"bwd {k['pono_spade_bwd']*1e3:.0f} us = {28*el/k['pono_spade_bwd']/1e6:.0f} GB/s "
          f"({28*el/k['pono_spade_bwd']/1e6/8000:.2f})")


if len(sys.argv) > 4:
    run(*(int(a) for a in sys.argv[1:5]))
else:
    for shape in ((8, 512, 64, 64), (8, 256, 64, 64), (8, 128, 256, 256), (8, 1024, 16, 16)):
        run(*shape)
