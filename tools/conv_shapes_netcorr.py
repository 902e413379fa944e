"""Which convolutions the netcorr scope runs: shapes, calls per step and GPU time per call (events around each K16 launch
group: forward, input gradient + weight gradient).  python tools/conv_shapes_netcorr.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cocosnet_amd import ops

stats = collections.defaultdict(lambda: [0, 0.0])
orig_fwd, orig_bwd = ops._Conv2d.forward, ops._Conv2d.backward
pending = []


def timed(tag, key, fn, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn(*a)
    e1.record()
    pending.append((tag, key, e0, e1))
    return out


def fwd(ctx, x, weight, bias, stride, pad, *rest):
    key = (tuple(x.shape), tuple(weight.shape), stride, pad)
    ctx._key = key
    return timed("fwd", key, orig_fwd, ctx, x, weight, bias, stride, pad, *rest)


def bwd(ctx, dy):
    return timed("bwd", ctx._key, orig_bwd, ctx, dy)


ops._Conv2d.forward = staticmethod(fwd)
ops._Conv2d.backward = staticmethod(bwd)
dev = torch.device("cuda:0")
model, fwd_fn = bench.make_step("netcorr", dev)
d = bench.build_inputs(dev, "netcorr")
for it in range(4):
    pending.clear()
    for p in model.parameters():
        p.grad = None
    out = fwd_fn(d)
    torch.autograd.backward([out["warp_out"], out["warp_mask"]], [d["g_out"], d["g_mask"]])
    torch.cuda.synchronize()
for tag, key, e0, e1 in pending:
    s = stats[(tag, key)]
    s[0] += 1
    s[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in stats.values())
print(f"K16 time per step: {tot:.1f} ms")
for (tag, key), (n, ms) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"{tag} x{n:3d} {ms:7.2f} ms ({ms / n:6.3f} each)  x{key[0]} w{key[1]} s{key[2]} p{key[3]}")
