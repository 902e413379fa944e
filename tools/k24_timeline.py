"""Per-workgroup timeline of K24 (debug library built with -DCOCOS_K24_TIMING): when each workgroup starts, how long its two sweeps
and its store tail take.  Usage (GPU box): COCOS_LIB_PATH=.../libcocos_hip_k24t.so python tools/k24_timeline.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocosnet_amd import ops, _lib

B, Cin, h, w = 8, 407, 64, 64
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
leaves = [mk(B, Cin, h, w), mk(256, Cin, 1, 1) / Cin ** 0.5, mk(256) * 0.1, mk(B, Cin, h, w), mk(256, Cin, 1, 1) / Cin ** 0.5, mk(256) * 0.1]
for t in leaves:
    t.requires_grad_(True)
d1, d2 = mk(B, 256, h * w) * 1e-3, mk(B, 256, h * w) * 1e-3
def step():
    for t in leaves:
        t.grad = None
    planes = ops.OperandPlanes()
    qn, kn = ops.proj_center_l2norm_planes_pair(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), 1, planes)
    torch.autograd.backward([qn, kn], [d1, d2])
for _ in range(4):
    step()
lib = _lib.load()
n = 2 * B * (h * w // 128) * 2
buf = (ctypes.c_longlong * (n * 4))()
fn = lib.cocos_debug_k24_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn(ctypes.cast(buf, ctypes.c_void_p), n)
t = np.array(buf[:], dtype=np.int64).reshape(n, 4).astype(np.float64) / 100.0      # us
t0 = t[:, 0].min()
print("workgroups", n, "kernel span %.1f us" % (t[:, 3].max() - t0))
start = t[:, 0] - t0
print("start us: p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f" % tuple(np.percentile(start, [0, 25, 50, 75, 100])))
for name, a, b in (("sweep1", 0, 1), ("sweep2", 1, 2), ("stores", 2, 3), ("whole", 0, 3)):
    d = t[:, b] - t[:, a]
    early, late = d[start < 5], d[start >= 5]
    print(f"{name}: first-round mean {early.mean():.1f} us (n={len(early)}), later mean {late.mean() if len(late) else float('nan'):.1f} us (n={len(late)}), min {d.min():.1f} max {d.max():.1f}")
