import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from cocosnet_amd import ops, _lib
from oracle import corr_oracle as co

B, Nq, Nk, Cv = 1, 64, 64, 3
rs = np.random.RandomState(0)
q, k = rs.standard_normal((B, 256, Nq)), rs.standard_normal((B, 256, Nk))
v = rs.uniform(-1, 1, (B, Cv, Nk))
qn, kn = co.center_l2norm(q, True), co.center_l2norm(k, True)
g = rs.standard_normal((B, Cv, Nq)) * 1e-3
# oracle pieces
s = np.einsum("bki,bkj->bij", qn, kn) * 100.0
p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
out = np.einsum("bij,bcj->bci", p, v)
dp = np.einsum("bci,bcj->bij", g, v)
D = (g * out).sum(1)[:, :, None]
ds = p * (dp - D) * 100.0            # [B,Nq,Nk]
dq_ref = np.einsum("bkj,bij->bki", kn, ds)
dk_ref = np.einsum("bki,bij->bkj", qn, ds)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
ops.PRECISION = "f16x3"
Q, K = t(qn).requires_grad_(True), t(kn).requires_grad_(True)
o = ops.corr_softmax_warp(Q, K, t(v), 100.0)
print("out err", float((o.detach().double().cpu() - torch.from_numpy(out)).abs().max()))
o.backward(t(g))
dq, dk = Q.grad.double().cpu().numpy(), K.grad.double().cpu().numpy()
print("dq max", np.abs(dq).max(), "ref", np.abs(dq_ref).max(), "err", np.abs(dq - dq_ref).max())
print("dk max", np.abs(dk).max(), "ref", np.abs(dk_ref).max(), "err", np.abs(dk - dk_ref).max())
num = (dq * dq_ref).sum() / (dq_ref * dq_ref).sum()
print("dq projection coefficient on ref:", num, " dk:", (dk * dk_ref).sum() / (dk_ref * dk_ref).sum())

# ---- manual replay of the split backward to look at the dS'' planes ----
st = torch.cuda.current_stream().cuda_stream
with torch.no_grad():
    qn_t, kn_t, v_t, g_t = t(qn), t(kn), t(v), t(g)
    o2 = ops.corr_softmax_warp(qn_t, kn_t, v_t, 100.0)
    lse_ref = np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)
    lse_t = t(lse_ref)
    lg_t = t((s * np.log2(np.e)).transpose(0, 2, 1))
    cvp = 32
    gph, gpl, gsc = ops.split_f16(g_t, True, cpad=cvp, amax=g_t.abs().amax().reshape(1))
    vph, vpl = ops.split_f16(v_t, True, cpad=cvp)
    kch, kcl = ops.split_f16(kn_t, False, 16.0)
    qch, qcl = ops.split_f16(qn_t, False, 16.0)
    dsh = torch.zeros((B, Nk, Nq), device="cuda", dtype=torch.float16)
    dsl = torch.zeros((B, Nk, Nq), device="cuda", dtype=torch.float16)
    dsc = torch.zeros(1, device="cuda")
    dqn = torch.empty_like(qn_t)
    vam = v_t.abs().amax().reshape(1)
    _lib.call("cocos_corr_softmax_warp_bwd_query_f16x3", kch.data_ptr(), kcl.data_ptr(), vph.data_ptr(), vpl.data_ptr(),
              gph.data_ptr(), gpl.data_ptr(), gsc.data_ptr(), o2.data_ptr(), g_t.data_ptr(), lse_t.data_ptr(), lg_t.data_ptr(),
              dqn.data_ptr(), dsh.data_ptr(), dsl.data_ptr(), 0, 0, vam.data_ptr(), dsc.data_ptr(), B, 256, Nq, Nk, Cv, cvp, 100.0, 16.0, 0, st)
    torch.cuda.synchronize()
    print("g_scale", float(gsc), "ds_scale", float(dsc))
    rec = (dsh.double() + dsl.double()).cpu().numpy() / float(dsc)
    ref = ds.transpose(0, 2, 1)
    print("ds planes: max", np.abs(rec).max(), "ref", np.abs(ref).max(), "err", np.abs(rec - ref).max())
    bad = np.abs(rec - ref) > 1e-3 * np.abs(ref).max()
    print("bad fraction", bad.mean(), "bad rows (keys)", np.unique(np.where(bad)[1])[:40], "bad cols (queries)", np.unique(np.where(bad)[2])[:40])
    dkn = torch.empty_like(kn_t)
    _lib.call("cocos_hgemm_f16x3", qch.data_ptr(), qcl.data_ptr(), dsh.data_ptr(), dsl.data_ptr(), dkn.data_ptr(), B, 256, Nk, Nq,
              1.0 / 16.0, dsc.data_ptr(), 0, 0, st)
    print("dk err (manual)", np.abs(dkn.double().cpu().numpy() - dk_ref).max())
    dk_from_rec = np.einsum("bki,bji->bkj", qn, rec)
    print("dk from reconstructed planes err", np.abs(dk_from_rec - dk_ref).max())
