"""The K16b GEMM alone (cocos_conv2d_nhwc_bf16) at forward- and input-gradient-shaped launches, stream-K on / off.
python tools/conv_nhwc_gemm_bench.py"""
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


for name, (B, Cin, Hp, Wp, Cout) in (("407 fwd (66x66 -> 64x64: 256 tiles)", (8, 407, 66, 66, 407)),
                                     ("407 dx (68x68 -> 66x66: 274 tiles)", (8, 407, 68, 68, 407)),
                                     ("512 dx (274 tiles)", (8, 512, 68, 68, 512)),
                                     ("512->128 dx of the SPADE convs (128-row tiles)", (8, 512, 68, 68, 128)),
                                     ("128->512 fwd (K = 1152)", (8, 128, 66, 66, 512)),
                                     ("256 dx (273 tiles of 256x128)", (8, 256, 68, 68, 256))):
    x = torch.randn(B, Cin, Hp, Wp, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    xp = ops.conv_nhwc_prep(x, 0)
    planes, _, _ = ops._conv_weight_planes(w, None, 0)
    flops = 2.0 * B * (Hp - 2) * (Wp - 2) * Cout * Cin * 9
    rec = {"launch": name}
    for sk in (False, True):
        ops.CONV_NHWC_STREAMK = sk
        t = timeit(lambda: ops._conv_nhwc_call(xp, planes, None, Cout, 3, 3, 1))
        rec["stream_k" if sk else "tiles"] = {"ms": round(t, 4), "alg_tflops": round(flops / t / 1e9, 1)}
    rec["prep_ms"] = round(timeit(lambda: ops.conv_nhwc_prep(x, 0)), 4)
    print(json.dumps(rec), flush=True)
