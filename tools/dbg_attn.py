import sys, torch
sys.path.insert(0, "/root/repo")
from cocosnet_amd import ops
for (qmag, kmag, vmag) in [(40.0, 30.0, 5.0), (3e3, 2e3, 1.0), (4e5, 1.5e4, 1e4)]:
    torch.manual_seed(11)
    B, K, Nq, Nk, Cv = 2, 32, 512, 256, 70
    q = torch.randn(B, K, Nq, device="cuda") * qmag; k = torch.randn(B, K, Nk, device="cuda") * kmag
    v = (torch.rand(B, Cv, Nk, device="cuda") * 2 - 1) * vmag; g = torch.randn(B, Cv, Nq, device="cuda")
    qr, kr, vr = q.double().requires_grad_(True), k.double().requires_grad_(True), v.double().requires_grad_(True)
    lg = torch.bmm(qr.transpose(1, 2), kr)
    P = torch.softmax(lg, -1)
    ref = torch.bmm(vr, P.transpose(1, 2)); ref.backward(g.double())
    qq, kk, vv = q.clone().requires_grad_(True), k.clone().requires_grad_(True), v.clone().requires_grad_(True)
    o = ops.softmax_attention(qq, kk, vv, 1.0)
    fn = o.grad_fn
    rowstat, mtile = fn.rowstat.clone(), fn.mtile.clone()
    o.backward(g)
    nanc = lambda t: int(torch.isnan(t).sum())
    rel = lambda a, b: float(((a.double() - b).abs() * ~torch.isnan(a)).max() / b.abs().max())
    top2 = lg.detach().topk(2, -1).values
    print(f"mags {qmag} {kmag} {vmag}: max|logit| {float(lg.abs().max()):.3g} min gap {float((top2[...,0]-top2[...,1]).min()):.3g} | nan o {nanc(o)} dq {nanc(qq.grad)} dk {nanc(kk.grad)} dv {nanc(vv.grad)}"
          f" | err o {rel(o, ref.detach()):.2e} dv {rel(vv.grad, vr.grad):.2e} | max|dq| {float(qq.grad.nan_to_num().abs().max()):.3g} ref {float(qr.grad.abs().max()):.3g}"
          f" | rowstat r range {float(rowstat[:,1].min()):.3g}..{float(rowstat[:,1].max()):.3g} nan {nanc(rowstat)} mtile nan {nanc(mtile)}")
    # which rows of dv are wrong: the P the backward used, from dv with g = one-hot? compare argmax consistency
    am_ref = P.detach().argmax(-1)                                   # [B,Nq]
    dv_ref_onehot = torch.zeros(B, Cv, Nk, device="cuda", dtype=torch.float64).index_add_(2, am_ref[0], g[0:1].double().expand(B, -1, -1)) if False else None
