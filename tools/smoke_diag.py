import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cocosnet_amd import ops
from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
from oracle import corr_oracle as co, torch_ref as tr
dev = torch.device("cuda", 0)
def run(order, fused=True, onehot=True, zero_mean=True):
    ops.BOX3_FUSED = fused
    rs = np.random.RandomState(0)
    B, nc, down, fh, fw = 2, 7, 4, 4, 64
    H, W = fh * down, fw * down
    theta = rs.standard_normal((B, 256, fh, fw)).astype(np.float32) + (0.0 if zero_mean else 0.15)
    phi = (0.3 * np.roll(theta, 3, axis=3) + rs.standard_normal(theta.shape)).astype(np.float32)
    ref_img = rs.uniform(-1, 1, (B, 3, H, W)).astype(np.float32)
    lab = rs.randint(0, nc, (B, H, W))
    ref_seg = (lab[:, None] == np.arange(nc)[None, :, None, None]).astype(np.float32)
    if not onehot: ref_seg = rs.uniform(0, 1, ref_seg.shape).astype(np.float32)
    g_out = rs.standard_normal((B, 3, H, W)).astype(np.float32)
    g_mask = rs.standard_normal((B, nc, fh, fw)).astype(np.float32)
    t = lambda a, g=False: torch.from_numpy(a).to(dev).requires_grad_(g)
    rel = lambda x, r: float(np.abs(x.detach().double().cpu().numpy() - r).max() / (np.abs(r).max() + 1e-30))
    res = []
    for mk in order:
        flags = dict(match_kernel=mk, PONO_C=True, down=down, warp_mask_losstype="direct", isTrain=True)
        th, ph = t(theta, True), t(phi, True)
        out = correspondence_hot_path(th, ph, t(ref_img), t(ref_img), t(ref_seg), t(ref_seg), HotPathConfig(**flags))
        torch.autograd.backward([out["warp_out"], out["warp_mask"]], [t(g_out), t(g_mask)])
        f64 = lambda a: a.astype(np.float64)
        ref, dth, dph = tr.forward_backward(f64(theta), f64(phi), f64(ref_img), f64(ref_img), f64(ref_seg), f64(ref_seg),
                                            co.default_opt(**flags), {"warp_out": f64(g_out), "warp_mask": f64(g_mask)})
        res.append((mk, rel(out["warp_mask"], ref["warp_mask"]), rel(th.grad, dth), rel(ph.grad, dph)))
    print("order", order, "fused", fused, "onehot", onehot, "zero_mean", zero_mean, ["mk%d out %.1e dth %.1e dph %.1e" % r for r in res], flush=True)
run((3,)); run((1, 3)); run((3,), fused=False); run((3,), onehot=False); run((3,), zero_mean=False); run((3,), onehot=False, zero_mean=False)
