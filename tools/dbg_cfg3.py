import copy, sys, torch
import torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from cocosnet_amd import ops, translation as tl
opt = tl.celebahq_edge_train_options()
torch.manual_seed(0)
G = tl.SPADEGenerator(opt).cuda(); G.init_weights(opt.init_type, opt.init_variance); G.eval()
g = torch.Generator(device="cuda").manual_seed(21)
seg = torch.rand(2, 15, 256, 256, device="cuda", generator=g)
cbn = torch.cat((torch.rand(2, 3, 256, 256, device="cuda", generator=g) * 2 - 1, seg), 1)
cap = {}
h = G.attn.register_forward_hook(lambda m, i, o: cap.setdefault("x", i[0].detach()))
with torch.no_grad():
    G(seg, warp_out=cbn)
h.remove()
x = cap["x"]; a = G.attn; B, _, H, W = x.shape
with torch.no_grad():
    theta = a.theta(x).view(B, a.ch // 8, H * W)
    phi = F.max_pool2d(a.phi(x), [2, 2]).view(B, a.ch // 8, H * W // 4)
    gg = F.max_pool2d(a.g(x), [2, 2]).view(B, a.ch // 2, H * W // 4)
    print("x", float(x.abs().max()), "theta", float(theta.abs().max()), "phi", float(phi.abs().max()), "g", float(gg.abs().max()))
    lg = torch.bmm(theta[:1].double().transpose(1, 2), phi[:1].double())
    top2 = lg.topk(2, dim=-1).values
    print("logits max", float(lg.abs().max()), "min gap top1-top2", float((top2[..., 0] - top2[..., 1]).min()), "median gap", float((top2[..., 0] - top2[..., 1]).median()))
    ref = torch.bmm(gg[:1].double(), F.softmax(lg, -1).transpose(1, 2))
    att = ops.softmax_attention(theta.contiguous(), phi.contiguous(), gg.contiguous(), 1.0)
    print("att nan", bool(torch.isnan(att).any()), "inf", bool(torch.isinf(att).any()), "nan frac", float(torch.isnan(att).float().mean()))
    ok = ~torch.isnan(att[:1])
    print("err on finite", float(((att[:1].double() - ref).abs() * ok).max() / ref.abs().max()))
    for sc in (1e-1, 1e-2, 1e-3):
        att2 = ops.softmax_attention((theta * sc).contiguous(), (phi * sc).contiguous(), gg.contiguous(), 1.0)
        r2 = torch.bmm(gg[:1].double(), F.softmax(lg * sc * sc, -1).transpose(1, 2))
        print("scale", sc, "nan", bool(torch.isnan(att2).any()), "err", float((att2[:1].double() - r2).abs().max() / r2.abs().max()))
