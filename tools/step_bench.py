"""The bench step itself (bench.HotPathStep on bench.build_inputs: projections + hot path, forward + backward), a few times — meant to sit
under `rocprofv3 --pmc ...` or `--kernel-trace --stats` so that EVERY kernel of the step gets counters (tools/kernel_bench.py runs the
three MFMA kernels only).  usage: python tools/step_bench.py [--iters 5] [--match-kernel 1|3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--match-kernel", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model, fwd = bench.make_step("hotpath", dev, a.match_kernel)
d = bench.build_inputs(dev, "hotpath")


def step():
    for p in model.parameters():
        p.grad = None
    d["cont_features"].grad = d["ref_features"].grad = None
    out = fwd(d)
    torch.autograd.backward([out["warp_out"], out["warp_mask"]], [d["g_out"], d["g_mask"]])


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    step()
e1.record()
torch.cuda.synchronize()
print(f"match_kernel {a.match_kernel}: {e0.elapsed_time(e1) / a.iters:.4f} ms per step over {a.iters} steps")
