"""Which framework ops (copies, adds, reductions) the netcorr step still launches and from where: torch.profiler with stacks,
aggregated by (op, innermost cocosnet_amd frame).  COCOS_CONV=bf16 python tools/netcorr_glue_trace.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda:0")
model, fwd_fn = bench.make_step("netcorr", dev)
d = bench.build_inputs(dev, "netcorr")


def step():
    for p in model.parameters():
        p.grad = None
    out = fwd_fn(d)
    torch.autograd.backward([out["warp_out"], out["warp_mask"]], [d["g_out"], d["g_mask"]])


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_stack_n=12):
    us = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if not e.key.startswith("aten::") or us <= 0:
        continue
    frame = next((f for f in e.stack if "cocosnet_amd" in f or "bench.py" in f), e.stack[0] if e.stack else "?")
    k = (e.key, frame.strip()[-110:])
    agg[k][0] += e.count
    agg[k][1] += us
tot = sum(v[1] for v in agg.values())
print(f"framework ops with GPU time of their own: {tot / 1e3:.2f} ms")
for (name, frame), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us / 1e3:7.3f} ms x{n:4d}  {name:32s} {frame}")
