"""First-light check of every HIP kernel against fp64 torch-CPU math (run on the GPU box).

Usage: python tools/gpu_first_light.py [--quick]   -> gpurun_out/first_light.json
Not a pytest file: it keeps going after a failure so one gpurun call reports on everything.
"""
import json, math, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops, _lib

OUT = {}
dev = "cuda"

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def record(name, fn):
    t0 = time.time()
    try:
        OUT[name] = fn()
    except Exception as e:
        OUT[name] = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
    OUT[name + "__s"] = round(time.time() - t0, 2)
    print(name, json.dumps(OUT[name])[:600], flush=True)

def probe():
    d = ops.mfma_probe().cpu()
    bad = 0
    for lane in range(64):
        h, c = lane >> 5, lane & 31
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * h
            j = c
            exp = (1 + i) * (1 + j) + (101 + i) * 1000.0 * (1 + j)
            if abs(d[lane, r].item() - exp) > 1e-3 * exp: bad += 1
    return {"mismatches": bad, "sample": d[33, :4].tolist()}

def make_qkv(B, Nq, Nk, Cv, seed=0, peaked=False):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, 256, Nq, generator=g, dtype=torch.float64)
    k = torch.randn(B, 256, Nk, generator=g, dtype=torch.float64)
    if peaked and Nq == Nk:
        k = q + 0.05 * torch.randn(B, 256, Nk, generator=g, dtype=torch.float64)
    def nrm(x):
        x = x - x.mean(1, keepdim=True)
        return x / (x.norm(dim=1, keepdim=True) + sys.float_info.epsilon)
    v = torch.rand(B, Cv, Nk, generator=g, dtype=torch.float64) * 2 - 1
    return nrm(q), nrm(k), v

def ref_attn(q, k, v, inv_t):
    S = torch.einsum("bki,bkj->bij", q, k) * inv_t
    P = torch.softmax(S, -1)
    return torch.einsum("bij,bcj->bci", P, v)

def fused_case(B, Nq, Nk, Cv, peaked=False, want_dv=True):
    def run():
        q, k, v = make_qkv(B, Nq, Nk, Cv, peaked=peaked)
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(want_dv)
        o_ref = ref_attn(q, k, v, 100.0)
        g = torch.randn(o_ref.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
        o_ref.backward(g)
        qd = q.detach().float().to(dev).requires_grad_(True)
        kd = k.detach().float().to(dev).requires_grad_(True)
        vd = v.detach().float().to(dev).requires_grad_(want_dv)
        o = ops.corr_softmax_warp(qd, kd, vd, 100.0)
        o.backward(g.float().to(dev))
        torch.cuda.synchronize()
        res = {"out": rel(o, o_ref), "dq": rel(qd.grad, q.grad), "dk": rel(kd.grad, k.grad)}
        if want_dv: res["dv"] = rel(vd.grad, v.grad)
        res["nan"] = bool(torch.isnan(o).any() or torch.isnan(qd.grad).any() or torch.isnan(kd.grad).any())
        return res
    return run

def center_case(B, K, N, pono_c):
    def run():
        g = torch.Generator().manual_seed(3)
        x = (torch.randn(B, K, N, generator=g, dtype=torch.float64) + 0.3).requires_grad_(True)
        xc = x - x.mean(dim=1 if pono_c else -1, keepdim=True)
        y = xc / (torch.norm(xc, 2, 1, keepdim=True) + sys.float_info.epsilon)
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(gy)
        xd = x.detach().float().to(dev).requires_grad_(True)
        yd = ops.center_l2norm(xd, pono_c)
        yd.backward(gy.float().to(dev))
        torch.cuda.synchronize()
        return {"y": rel(yd, y), "dx": rel(xd.grad, x.grad)}
    return run

def mat_case(B, K, Nq, Nk, Cv):
    def run():
        g = torch.Generator().manual_seed(5)
        q = torch.randn(B, K, Nq, generator=g, dtype=torch.float64).requires_grad_(True)
        k = torch.randn(B, K, Nk, generator=g, dtype=torch.float64).requires_grad_(True)
        v = torch.randn(B, Cv, Nk, generator=g, dtype=torch.float64).requires_grad_(True)
        f = torch.einsum("bki,bkj->bij", q, k) * 0.37
        p = torch.softmax(f, -1)
        o = torch.einsum("bij,bcj->bci", p, v)
        go = torch.randn(o.shape, generator=g, dtype=torch.float64)
        o.backward(go)
        qd = q.detach().float().to(dev).requires_grad_(True)
        kd = k.detach().float().to(dev).requires_grad_(True)
        vd = v.detach().float().to(dev).requires_grad_(True)
        fd = ops.corr_materialize(qd, kd, 0.37)
        pd = ops.row_softmax(fd)
        od = ops.warp_materialized(pd, vd)
        od.backward(go.float().to(dev))
        torch.cuda.synchronize()
        return {"f": rel(fd, f), "p": rel(pd, p), "o": rel(od, o), "dq": rel(qd.grad, q.grad),
                "dk": rel(kd.grad, k.grad), "dv": rel(vd.grad, v.grad)}
    return run

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def timing(B, N, Cv):
    def run():
        g = torch.Generator(device=dev).manual_seed(0)
        def nrm(x):
            x = x - x.mean(1, keepdim=True); return x / x.norm(dim=1, keepdim=True)
        q = nrm(torch.randn(B, 256, N, device=dev, generator=g)).requires_grad_(True)
        k = nrm(torch.randn(B, 256, N, device=dev, generator=g)).requires_grad_(True)
        v = (torch.rand(B, Cv, N, device=dev, generator=g) * 2 - 1)
        go = torch.randn(B, Cv, N, device=dev, generator=g)
        res = {}
        res["fused_fwd_ms"] = timeit(lambda: ops.corr_softmax_warp(q.detach(), k.detach(), v, 100.0))
        def fb():
            q.grad = None; k.grad = None
            ops.corr_softmax_warp(q, k, v, 100.0).backward(go)
        res["fused_fwd_bwd_ms"] = timeit(fb)
        fl_f = 2.0 * N * N * (256 + Cv) * B
        res["fused_fwd_tflops"] = fl_f / res["fused_fwd_ms"] / 1e9
        fl_fb = 2.0 * N * N * (3 * 256 + 2 * Cv) * B
        res["fused_fwd_bwd_tflops_alg"] = fl_fb / res["fused_fwd_bwd_ms"] / 1e9
        # stock PyTorch-ROCm formulation of the same math (the reference's op sequence)
        def stock(qq, kk):
            f = torch.matmul(qq.permute(0, 2, 1), kk) / 0.01
            p = torch.softmax(f, -1)
            return torch.matmul(p, v.permute(0, 2, 1)).permute(0, 2, 1)
        res["stock_fwd_ms"] = timeit(lambda: stock(q.detach(), k.detach()))
        def sfb():
            q.grad = None; k.grad = None
            stock(q, k).backward(go)
        res["stock_fwd_bwd_ms"] = timeit(sfb)
        o1 = ops.corr_softmax_warp(q.detach(), k.detach(), v, 100.0); o2 = stock(q.detach(), k.detach())
        res["fused_vs_stock_rel"] = rel(o1, o2)
        res["mat_ms"] = timeit(lambda: ops.corr_materialize(q.detach(), k.detach(), 100.0))
        f = ops.corr_materialize(q.detach(), k.detach(), 100.0)
        res["softmax_ms"] = timeit(lambda: ops.row_softmax(f))
        res["softmax_GBps"] = 2 * f.numel() * 4 / res["softmax_ms"] / 1e6
        x = torch.randn(B, 256, N, device=dev)
        res["center_ms"] = timeit(lambda: ops.center_l2norm(x, True))
        return res
    return run

if __name__ == "__main__":
    quick = "--quick" in sys.argv
    print("device", torch.cuda.get_device_name(0), "cpus", os.cpu_count(), flush=True)
    OUT["device"] = torch.cuda.get_device_name(0); OUT["cpus"] = os.cpu_count()
    OUT["version"] = _lib.load().cocos_version()
    record("mfma_probe", probe)
    record("center_ponoC_small", center_case(2, 256, 100, True))
    record("center_rowmean_small", center_case(2, 256, 100, False))
    record("center_ponoC_K2304", center_case(1, 2304, 300, True))
    record("fused_small_cv3", fused_case(2, 256, 256, 3))
    record("fused_ragged", fused_case(1, 200, 177, 5))
    record("fused_cv154", fused_case(1, 512, 512, 154))
    record("fused_cv40_peaked", fused_case(2, 384, 384, 40, peaked=True))
    record("fused_cv70", fused_case(1, 300, 260, 70))
    record("fused_cv100_nodv", fused_case(1, 256, 320, 100, want_dv=False))
    record("mat_small", mat_case(2, 256, 200, 300, 5))
    record("mat_K2304", mat_case(1, 2304, 130, 257, 3))
    record("mat_long_rows", mat_case(1, 64, 40, 5000, 2))
    if not quick:
        record("timing_b8_cv154", timing(8, 4096, 154))
        record("timing_b8_cv3", timing(8, 4096, 3))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(OUT, open("gpurun_out/first_light.json", "w"), indent=1)
    print("DONE")
