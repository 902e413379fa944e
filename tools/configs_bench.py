"""Hot-path timings (fwd+bwd from the theta/phi conv outputs on) for every BASELINE.json config shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path

def run(name, B, size, nc, seg_float, cfg, steps=10):
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    g = torch.Generator(device="cuda").manual_seed(0)
    fh = size // cfg.down
    th = torch.randn(B, 256, fh, fh, device="cuda", generator=g).requires_grad_(True)
    ph = (0.3 * th.detach() + torch.randn(B, 256, fh, fh, device="cuda", generator=g)).requires_grad_(True)
    img = torch.rand(B, 3, size, size, device="cuda", generator=g) * 2 - 1
    real = torch.rand(B, 3, size, size, device="cuda", generator=g) * 2 - 1
    if seg_float:
        seg = torch.rand(B, nc, size, size, device="cuda", generator=g)
    else:
        lab = torch.randint(0, nc, (B, 1, size, size), device="cuda", generator=g)
        seg = torch.zeros(B, nc, size, size, device="cuda").scatter_(1, lab, 1.0)
    cot = {}
    def step():   # the synthetic loss <out, G> is fed to autograd as its gradient G (no loss kernels in the timing: bench.py's rule)
        th.grad = None; ph.grad = None
        o = correspondence_hot_path(th, ph, img, real, seg, seg, cfg)
        if not cot:
            cot.update({k: torch.randn(v.shape, device="cuda", generator=g) for k, v in o.items()})
        torch.autograd.backward([o[k] for k in sorted(o)], [cot[k] for k in sorted(o)])
    for _ in range(3): step()
    # the step time WITHOUT per-call HIP events (each pair is two marker packets that serialise dispatch: bracketing all ~45
    # calls of a step inflates it by 5-8 %, DESIGN §5 finding 9); the per-kernel breakdown comes from separate steps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    with ops.KernelTimer() as kt:
        for _ in range(3): step()
    ks = {k: round(v["total_ms"] / 3, 3) for k, v in kt.summary().items()}
    mem = torch.cuda.max_memory_allocated() / 2**30
    rec = {"config": name, "B": B, "grid": f"{fh}x{fh}", "ms_per_step": round(dt * 1e3, 3),
           "images_per_s": round(B / dt, 1), "kernel_ms_per_step": ks, "peak_mem_GiB": round(mem, 2)}
    print(json.dumps(rec), flush=True)
    return rec

out = []
C = HotPathConfig
out.append(run("cfg2 ADE20k 256^2 B=8 mk1 direct mask (headline)", 8, 256, 151, False,
               C(match_kernel=1, PONO_C=True, warp_mask_losstype="direct", isTrain=True)))
out.append(run("cfg2' same, match_kernel 3 (reference default)", 8, 256, 151, False,
               C(match_kernel=3, PONO_C=True, warp_mask_losstype="direct", isTrain=True)))
out.append(run("cfg3+ CelebA-HQ edge 256^2 B=16 mk1 warp_cycle + two_cycle (superset of the README flags: three passes) bilinear", 16, 256, 15, True,
               C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_cycle_w=1.0, two_cycle=True, isTrain=True)))
out.append(run("cfg3' same, match_kernel 3", 16, 256, 15, True,
               C(match_kernel=3, PONO_C=True, warp_bilinear=True, warp_cycle_w=1.0, two_cycle=True, isTrain=True)))
# config 3 AS THE README RUNS IT (README.md:106): --warp_cycle_w 1 without --two_cycle (two passes: rows, columns), match_kernel 3
out.append(run("cfg3 as written: CelebA-HQ edge 256^2 B=16, warp_cycle_w 1 (no two_cycle), match_kernel 3", 16, 256, 15, True,
               C(match_kernel=3, PONO_C=True, warp_bilinear=True, warp_cycle_w=1.0, isTrain=True)))
out.append(run("cfg3 as written, match_kernel 1", 16, 256, 15, True,
               C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_cycle_w=1.0, isTrain=True)))
out.append(run("cfg5 DeepFashion 512^2 warp_patch, 128x128 grid, B=2 mk1", 2, 512, 20, True,
               C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_patch=True, isTrain=True)))
out.append(run("cfg5' 256^2 warp_stride 2 warp_patch, 128x128 grid, B=2 mk1", 2, 256, 20, True,
               C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_patch=True, down=2, isTrain=True)))
# the same two routes AS THE REFERENCE RUNS THEM: match_kernel 3 is the default of every README command (base_options.py:70)
out.append(run("cfg5 mk3 DeepFashion 512^2 warp_patch, 128x128 grid, B=2, match_kernel 3 (fused family, round 4)", 2, 512, 20, True,
               C(match_kernel=3, PONO_C=True, warp_bilinear=True, warp_patch=True, isTrain=True)))
out.append(run("cfg5' mk3 256^2 warp_stride 2 warp_patch, 128x128 grid, B=2, match_kernel 3", 2, 256, 20, True,
               C(match_kernel=3, PONO_C=True, warp_bilinear=True, warp_patch=True, down=2, isTrain=True)))
ops.BOX3_FUSED = False     # what round 3 ran for these shapes: K3 -> K6 -> K7, three 1 GiB matrices per sample and orientation
out.append(run("cfg5 mk3, round-3 route (materialised chain, COCOS_BOX3_FUSED=0)", 2, 512, 20, True,
               C(match_kernel=3, PONO_C=True, warp_bilinear=True, warp_patch=True, isTrain=True), steps=3))
ops.BOX3_FUSED = True
# ---- saved logits vs chunked recompute on the split flavour: the same three shapes, the route forced ----
def ab(name, *a, **k):
    saved = (ops.MAX_SAVED_LOGITS_BYTES, ops.RECOMPUTE_CHUNK_BYTES)
    try:
        for route, lim in (("saved logits", 64 << 30), ("chunked recompute (512 MiB per matrix)", 0)):
            ops.MAX_SAVED_LOGITS_BYTES = lim
            ops.RECOMPUTE_CHUNK_BYTES = 512 << 20
            out.append(run(f"A/B {name}: {route}", *a, **k))
    finally:
        ops.MAX_SAVED_LOGITS_BYTES, ops.RECOMPUTE_CHUNK_BYTES = saved
ab("cfg2 mk1 B=8", 8, 256, 151, False, C(match_kernel=1, PONO_C=True, warp_mask_losstype="direct", isTrain=True))
ab("cfg3 mk1 B=16 cycle", 16, 256, 15, True, C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_cycle_w=1.0, two_cycle=True, isTrain=True))
ab("cfg5 mk1 B=2 HW=16384", 2, 512, 20, True, C(match_kernel=1, PONO_C=True, warp_bilinear=True, warp_patch=True, isTrain=True))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/configs_bench.json", "w"), indent=1)
