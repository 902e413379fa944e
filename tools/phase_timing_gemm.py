"""Debug: prologue / k-loop / epilogue ticks of the split GEMM for the three K0 products (library built with
COCOS_EXTRA_HIPFLAGS=-DCOCOS_DEBUG_TIMING)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops, _lib
lib = _lib.load()
B, Cin, Cout, N = 8, 407, 256, 4096
x = torch.randn(B, Cin, 64, 64, device="cuda")
w = torch.randn(Cout, Cin, device="cuda") * 0.05
bias = torch.randn(Cout, device="cuda")
dy = torch.randn(B, Cout, 64, 64, device="cuda")
y = torch.empty(B, Cout, 64, 64, device="cuda")
dx = torch.empty_like(x)
parts = lib.cocos_proj1x1_bwd_partials_f16x3(B, Cin, Cout, N)
dw = torch.empty(parts, Cout, Cin, device="cuda")
st = torch.cuda.current_stream().cuda_stream
buf = (ctypes.c_longlong * 8)()
def show(name):
    lib.cocos_debug_read_timing_xg(buf, 1)
    t = list(buf)[:3]
    print(f"{name}: prologue {t[0]} | k loop {t[1]} | epilogue {t[2]} ticks (workgroup 0)")
for it in range(2):
    lib.cocos_debug_read_timing_xg(buf, 1)
    _lib.call("cocos_proj1x1_fwd_f16x3", x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Cin, Cout, N, 0, 0, st)
    show("fwd (13 steps)")
    _lib.call("cocos_proj1x1_bwd_f16x3", x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, B, Cin, Cout, N, 0, 0, 0, st)
    show("dx  (8 steps)")
    _lib.call("cocos_proj1x1_bwd_f16x3", x.data_ptr(), w.data_ptr(), dy.data_ptr(), 0, dw.data_ptr(), B, Cin, Cout, N, 0, 0, 0, st)
    show(f"dw  ({parts} partials)")
