"""Debug: shader-clock cycles per half step of the K16 forward kernel (library built with
COCOS_LIB_NAME=libcocos_hip_dbg.so COCOS_EXTRA_HIPFLAGS=-DCOCOS_DEBUG_TIMING python -m cocosnet_amd.build).
Usage: COCOS_LIB_PATH=.../libcocos_hip_dbg.so python tools/phase_timing_conv.py [B Cin H W Cout k s p]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops, _lib
lib = ctypes.CDLL(os.environ["COCOS_LIB_PATH"])
a = [int(v) for v in sys.argv[1:9]] if len(sys.argv) >= 9 else [8, 407, 66, 66, 407, 3, 1, 0]
b, ci, h, w, co, k, s, p = a
x = torch.randn(b, ci, h, w, device="cuda")
wt = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
buf = (ctypes.c_longlong * 8)()
for it in range(3):
    lib.cocos_debug_read_timing_conv(buf, 1)
    ops.conv2d(x, wt, None, s, p)
    lib.cocos_debug_read_timing_conv(buf, 1)
    t = list(buf)
    n = max(t[4], 1)
    print(f"steps {n}: half0 (commit) {t[0] / n:.0f} | half1 (fetch) {t[1] / n:.0f} | barrier {t[2] / n:.0f} | step {t[3] / n:.0f} cycles "
          f"(MFMA ideal 768 per half)")
