"""Debug: per-phase shader-clock ticks of the saved-logits query backward (needs a library built
with COCOS_EXTRA_HIPFLAGS=-DCOCOS_DEBUG_TIMING)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops, _lib
lib = _lib.load()
B, N, Cv = 8, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 154
g = torch.Generator(device="cuda").manual_seed(0)
nrm = lambda x: (x - x.mean(1, keepdim=True)) / (x - x.mean(1, keepdim=True)).norm(dim=1, keepdim=True)
q = nrm(torch.randn(B, 256, N, device="cuda", generator=g)).requires_grad_(True)
k = nrm(0.2 * q.detach() + torch.randn(B, 256, N, device="cuda", generator=g)).requires_grad_(True)
v = torch.rand(B, Cv, N, device="cuda", generator=g) * 2 - 1
go = torch.randn(B, Cv, N, device="cuda", generator=g)
buf = (ctypes.c_longlong * 8)()
for it in range(3):
    q.grad = None; k.grad = None
    ops.corr_softmax_warp(q, k, v, 100.0).backward(go)
    lib.cocos_debug_read_timing(buf, 1)
    t = list(buf)[:4]
    tot = sum(t)
    print("ticks per tile: dS+dX loop %.0f | fetch issue %.0f | dP loop %.0f | barrier %.0f | total %.0f  (ideal MFMA: dX 8192, dP %d)"
          % (t[0] / 128, t[1] / 128, t[2] / 128, t[3] / 128, tot / 128, 64 * ((Cv + 31) // 32) * 16))
