"""Accuracy of the two K2-forward flavours against the fp64 oracle on the same inputs (GPU box).
Prints max|out - ref| / max|ref| for diffuse and peaked logits."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from cocosnet_amd import ops
from oracle import corr_oracle as co

for peaked in (False, True):
    rs = np.random.RandomState(1)
    B, N, Cv = 1, 1024, 154
    q = rs.standard_normal((B, 256, N))
    k = q + 0.05 * rs.standard_normal((B, 256, N)) if peaked else rs.standard_normal((B, 256, N))
    v = rs.uniform(-1, 1, (B, Cv, N))
    qn, kn = co.center_l2norm(q, True), co.center_l2norm(k, True)
    ref = co.corr_softmax_warp(qn, kn, v, 100.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    for prec in ("fp32", "f16x3"):
        ops.PRECISION = prec
        out = ops.corr_softmax_warp(t(qn), t(kn), t(v), 100.0).double().cpu().numpy()
        print(f"peaked={peaked} {prec:6s} rel err {np.abs(out - ref).max() / np.abs(ref).max():.2e}")
