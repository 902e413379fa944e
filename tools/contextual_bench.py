"""K22 (fused contextual loss) against round 2's materialised route (K3 + K15) at get_ctx_loss's shapes (pix2pix_model.py:196-203),
forward + backward w.r.t. X (Y is detached by the caller), HIP-event timed.  usage: python tools/contextual_bench.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
from cocosnet_amd import ops  # noqa: E402


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    rows = []
    for B, C, N in [(8, 512, 256), (8, 512, 1024), (8, 256, 1024), (8, 128, 1024), (16, 512, 1024), (2, 512, 4096), (8, 512, 4096), (1, 64, 16384)]:
        g = torch.Generator(device="cuda").manual_seed(N + C)
        Y = torch.randn(B, C, N, device="cuda", generator=g)
        X = 0.6 * Y[:, :, torch.randperm(N, device="cuda", generator=g)] + torch.randn(B, C, N, device="cuda", generator=g)
        nrm = lambda t: t / (t.norm(dim=1, keepdim=True) + 2.2e-16)
        Xn, Yn = nrm(X).contiguous(), nrm(Y).contiguous()

        def fused():
            x = Xn.clone().requires_grad_(True)
            ops.contextual_cx(x, Yn, 0.1, 1e-3).sum().backward()

        def mat():
            x = Xn.clone().requires_grad_(True)
            ops.contextual_rows(ops.corr_materialize(x, Yn, 1.0), 0.1, 1e-3).sum().backward()
        row = {"B": B, "C": C, "N": N, "fused_ms": round(timed(fused), 4)}
        if N <= 4096:
            row["materialised_ms"] = round(timed(mat), 4)
        torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated(); fused(); torch.cuda.synchronize()
        row["fused_peak_MiB"] = (torch.cuda.max_memory_allocated() - base) >> 20
        if N <= 4096:
            torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated(); mat(); torch.cuda.synchronize()
            row["materialised_peak_MiB"] = (torch.cuda.max_memory_allocated() - base) >> 20
        with ops.KernelTimer() as kt:
            fused()
        row["fused_kernels_ms"] = {k: round(v["total_ms"], 4) for k, v in kt.summary().items() if k.startswith("contextual")}
        print(json.dumps(row), flush=True)
        rows.append(row)


if __name__ == "__main__":
    main()
