"""K16b (conv_nhwc_bf16.hip) against the one-term gather kernels of conv_f16x3.hip at the module's big layer shapes: per-ABI-call
times (HIP events around each call) of forward and forward + backward.  python tools/conv_nhwc_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops

ops.CONV_PRECISION = os.environ.get("PREC", "bf16")          # PREC=f16x3: K16c against the split gather kernels
SHAPES = [("ResidualBlock 407->407 on 66x66", (8, 407, 66, 66, 407, 3, 0)),
          ("SPADE 128->512 on 66x66", (8, 128, 66, 66, 512, 3, 0)),
          ("512->512 on 66x66", (8, 512, 66, 66, 512, 3, 0)),
          ("adaptor 128->256 on 128x128 p1", (8, 128, 128, 128, 256, 3, 1)),
          ("256->256 on 66x66", (8, 256, 66, 66, 256, 3, 0))]
for name, (b, ci, h, w, co, k, p) in SHAPES:
    x = torch.randn(b, ci, h, w, device="cuda", requires_grad=True)
    wt = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).requires_grad_(True)
    bias = torch.randn(co, device="cuda", requires_grad=True)
    rec = {"shape": name}
    for nhwc in (False, True):
        ops.CONV_NHWC = ops.CONV_NHWC_F16X3 = nhwc
        y = ops.conv2d(x, wt, bias, 1, p)
        go = torch.randn_like(y)
        flops = 2.0 * y.numel() * ci * k * k
        for _ in range(3):
            torch.autograd.grad(ops.conv2d(x, wt, bias, 1, p), (x, wt, bias), go)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.autograd.grad(ops.conv2d(x, wt, bias, 1, p), (x, wt, bias), go)
        e1.record()
        torch.cuda.synchronize()
        with ops.KernelTimer() as kt:
            for _ in range(5):
                torch.autograd.grad(ops.conv2d(x, wt, bias, 1, p), (x, wt, bias), go)
        ks = {k_: round(v["total_ms"] / 5, 4) for k_, v in kt.summary().items()}
        rec["nhwc" if nhwc else "gather"] = {"fwd_bwd_ms": round(e0.elapsed_time(e1) / 10, 3), "alg_tflops_3gemms": round(3 * flops / (e0.elapsed_time(e1) / 10) / 1e9, 1), "calls_ms": ks}
    print(json.dumps(rec), flush=True)
