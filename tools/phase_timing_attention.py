"""Debug: per-phase shader-clock ticks of the K2 split kernels in the magnitude-free (Attention) flavour at the netG shape
(library built with -DCOCOS_DEBUG_TIMING: COCOS_ABL_EXTRA=-DCOCOS_DEBUG_TIMING tools/build_ablations.sh 0;
COCOS_LIB_PATH=cocosnet_amd/lib/libcocos_hip_abl0.so python tools/phase_timing_attention.py)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops, _lib
lib = _lib.load()
B, K, Nq, Nk, Cv = 4, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 4096, 128
if len(sys.argv) > 2: B = int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(B, K, Nq, device="cuda", generator=g).requires_grad_(True)
k = torch.randn(B, K, Nk, device="cuda", generator=g).requires_grad_(True)
v = torch.randn(B, Cv, Nk, device="cuda", generator=g).requires_grad_(True)
go = torch.randn(B, Cv, Nq, device="cuda", generator=g)
buf = (ctypes.c_longlong * 8)()
nt = Nk // 32
for skip in (True,):
    for it in range(2):
        q.grad = k.grad = v.grad = None
        out = ops.softmax_attention(q, k, v, 1.0)
        lib.cocos_debug_read_timing_fwd_f16x3(buf, 1)
        t = list(buf)[:6]
        t8 = list(buf)[:8]
        out.backward(go)
        lib.cocos_debug_read_timing_bwd_f16x3(buf, 1)
        u = list(buf)[:4]
    print(f"skip={skip} fwd: " + " | ".join(f"{n} {x / nt:.0f}" for n, x in zip(["QK", "commit+fetch", "softmax", "split P", "PV", "barrier"], t)),
          f"| total {sum(t) / nt:.0f} ticks/tile")
    print(f"skip={skip} bwd query: dP+staging %.0f | dqn+VALU %.0f | plane stores %.0f | barrier %.0f | total %.0f ticks/tile"
          % tuple([x / (nt - 1) for x in u] + [sum(u) / (nt - 1)]))

# the unit-norm flavour (the correspondence itself) at the same geometry, for comparison
nrm = lambda x: (x - x.mean(1, keepdim=True)) / (x - x.mean(1, keepdim=True)).norm(dim=1, keepdim=True)
q2 = nrm(torch.randn(B, 256, Nq, device="cuda", generator=g)).requires_grad_(True)
k2 = nrm(0.2 * q2.detach()[:, :, :Nk] + torch.randn(B, 256, Nk, device="cuda", generator=g))
for it in range(2):
    ops.corr_softmax_warp(q2, k2, v.detach(), 100.0)
    lib.cocos_debug_read_timing_fwd_f16x3(buf, 1)
    t8 = list(buf)[:8]
print("unit-norm fwd: " + " | ".join(f"{n} {x / nt:.0f}" for n, x in zip(["QK", "commit+fetch", "softmax", "split P", "PV"], t8)),
      )
