import sys, torch
sys.path.insert(0, "/root/repo")
from cocosnet_amd import ops, _lib
B, h, w = 2, 128, 128
N = h * w
g = torch.randn(B * N * N, device="cuda")
gmax = ops.absmax(g)
dch = torch.empty(B * N * N, device="cuda", dtype=torch.float16); dcl = torch.empty_like(dch)
sc = torch.empty(1, device="cuda")
def run():
    _lib.call("cocos_box3_adjoint_planes_f16x3", g.data_ptr(), gmax.data_ptr(), dch.data_ptr(), dcl.data_ptr(), sc.data_ptr(), B, N, N, h, w, torch.cuda.current_stream().cuda_stream)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("K20 w128 B=2:", e0.elapsed_time(e1) / 10, "ms")
