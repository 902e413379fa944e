"""Which call sites take their own max|x| pass in one netCorr step (module scope), and how large the tensors are."""
import collections, sys, traceback
import torch
sys.path.insert(0, ".")
import bench
from cocosnet_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model, fwd = bench.make_step("netcorr", dev)
d = bench.build_inputs(dev, "netcorr")
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    o = fwd(d)
    torch.autograd.backward([o["warp_out"], o["warp_mask"]], [d["g_out"], d["g_mask"]])
for _ in range(3): step()
calls = collections.Counter(); size = collections.Counter()
orig = ops.absmax
def traced(x):
    fr = traceback.extract_stack(limit=6)
    site = " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(fr[:-1]))[:150]
    calls[site] += 1; size[site] += x.numel() * 4
    return orig(x)
ops.absmax = traced
step()
ops.absmax = orig
tot = sum(calls.values())
print("absmax calls per step:", tot, "MB:", sum(size.values()) >> 20)
for k, v in calls.most_common(14):
    print(f"{v:4d}  {size[k] >> 20:6d} MB  {k}")
with ops.KernelTimer() as kt:
    step()
s = kt.summary()
print({k: (v["calls"], round(v["total_ms"], 2)) for k, v in sorted(s.items(), key=lambda kv: -kv[1]["total_ms"])[:14]})
