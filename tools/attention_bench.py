"""ops.softmax_attention at the reference's Attention shapes (architecture.py:114-127): the fused K2 kernels (K zero-padded
to 256, device-side operand scales) against the materialised route (K3 -> K4 -> K5), forward + backward, peak memory."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops

def run(name, B, K, Nq, Nk, Cv, fused, steps=5):
    ops.ATTENTION_FUSED = fused
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(B, K, Nq, device="cuda", generator=g).requires_grad_(True)
    k = torch.randn(B, K, Nk, device="cuda", generator=g).requires_grad_(True)
    v = torch.randn(B, Cv, Nk, device="cuda", generator=g).requires_grad_(True)
    go = torch.randn(B, Cv, Nq, device="cuda", generator=g)
    def step():
        q.grad = k.grad = v.grad = None
        ops.softmax_attention(q, k, v, 1.0).backward(go)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    rec = {"shape": name, "route": "fused" if fused else "materialised", "ms_fwd_bwd": round(dt * 1e3, 3),
           "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
    print(json.dumps(rec), flush=True)
    del q, k, v, go
    torch.cuda.empty_cache()

for fused in (True, False):
    run("netG Attention(256) at 128x128: B=4, K=32, Nq=16384, Nk=4096, Cv=128", 4, 32, 16384, 4096, 128, fused)
    run("adaptor Attention(512) at 64x64: B=8, K=64, Nq=4096, Nk=1024, Cv=256", 8, 64, 4096, 1024, 256, fused)
# per-kernel breakdown of the fused route at the netG shape
ops.ATTENTION_FUSED = True
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(4, 32, 16384, device="cuda", generator=g).requires_grad_(True)
k = torch.randn(4, 32, 4096, device="cuda", generator=g).requires_grad_(True)
v = torch.randn(4, 128, 4096, device="cuda", generator=g).requires_grad_(True)
go = torch.randn(4, 128, 16384, device="cuda", generator=g)
def step():
    q.grad = k.grad = v.grad = None
    ops.softmax_attention(q, k, v, 1.0).backward(go)
def infer():
    with torch.no_grad():
        ops.softmax_attention(q, k, v, 1.0)
for fn, tag in ((step, "train"), (infer, "inference")):
    for _ in range(2):
        fn()
    with ops.KernelTimer() as kt:
        for _ in range(3):
            fn()
    print(json.dumps({"pass": tag, **{t: round(r["total_ms"] / 3, 3) for t, r in kt.summary().items()}}))
