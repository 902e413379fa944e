"""Debug: five inference forwards of the K2 split kernel at B=8, Nq=Nk=4096, Cv=128 in one flavour (argv[1] = rawm | unit) —
the driver of tools/attention_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cocosnet_amd import ops
B, Nq, Nk, Cv = 8, 4096, 4096, 128
g = torch.Generator(device="cuda").manual_seed(0)
v = torch.randn(B, Cv, Nk, device="cuda", generator=g)
with torch.no_grad():
    if sys.argv[1] == "rawm":
        q = torch.randn(B, 32, Nq, device="cuda", generator=g)
        k = torch.randn(B, 32, Nk, device="cuda", generator=g)
        for _ in range(5):
            ops.softmax_attention(q, k, v, 1.0)
    else:
        nrm = lambda x: (x - x.mean(1, keepdim=True)) / (x - x.mean(1, keepdim=True)).norm(dim=1, keepdim=True)
        q = nrm(torch.randn(B, 256, Nq, device="cuda", generator=g))
        k = nrm(0.2 * q + torch.randn(B, 256, Nk, device="cuda", generator=g))
        for _ in range(5):
            ops.corr_softmax_warp(q, k, v, 100.0)
torch.cuda.synchronize()
