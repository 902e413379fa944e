"""Debug: per-phase shader-clock ticks of the split-precision forward (library built with
COCOS_EXTRA_HIPFLAGS=-DCOCOS_DEBUG_TIMING).  Usage: python tools/phase_timing_f16x3.py [Cv] [train]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops, _lib
lib = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else _lib.load()
ops.PRECISION = "f16x3"
B, N, Cv = 8, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 154
train = len(sys.argv) > 2
g = torch.Generator(device="cuda").manual_seed(0)
nrm = lambda x: (x - x.mean(1, keepdim=True)) / (x - x.mean(1, keepdim=True)).norm(dim=1, keepdim=True)
q = nrm(torch.randn(B, 256, N, device="cuda", generator=g)).requires_grad_(train)
k = nrm(0.2 * q.detach() + torch.randn(B, 256, N, device="cuda", generator=g))
v = torch.rand(B, Cv, N, device="cuda", generator=g) * 2 - 1
buf = (ctypes.c_longlong * 8)()
names = ["QK", "commit+fetch", "softmax", "split P", "PV", "barrier"]
for it in range(3):
    ops.corr_softmax_warp(q, k, v, 100.0)
    lib.cocos_debug_read_timing_fwd_f16x3(buf, 1)
    t = list(buf)[:6]
    print(" | ".join(f"{n} {x / 128:.0f}" for n, x in zip(names, t)), f"| total {sum(t) / 128:.0f} ticks/tile  (MFMA ideal: QK 1536, PV {32 * 6 * ((Cv + 31) // 32)})")

# ---- backward (query side) ----
q2 = q.detach().requires_grad_(True)
k2 = k.detach().requires_grad_(True)
go = torch.randn(B, Cv, N, device="cuda", generator=g)
for it in range(3):
    q2.grad = None; k2.grad = None
    ops.corr_softmax_warp(q2, k2, v, 100.0).backward(go)
    lib.cocos_debug_read_timing_bwd_f16x3(buf, 1)
    t = list(buf)[:4]
    # (round 2: iteration t = dP'(t) with the staging pieces | dqn(t-1) with tile t's VALU sliced in | plane stores |
    #  barrier; the first iteration is not instrumented: 127 per workgroup)
    print("bwd query: dP+staging %.0f | dqn+VALU %.0f | dS'' plane stores %.0f | barrier %.0f | total %.0f ticks/tile (MFMA ideal: dP %d, dQ 1536)"
          % (t[0] / 127, t[1] / 127, t[2] / 127, t[3] / 127, sum(t) / 127, 32 * 6 * ((Cv + 31) // 32)))
