"""Stage-by-stage check of the fused match_kernel-3 family (K19 / K20, box3_fused_f16x3.hip) on the GPU box: every kernel
against numpy fp64 on a small grid, so that a failing end-to-end test can be pinned to one stage.  Prints one line per stage."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocosnet_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def unblock(t, B, N):
    """tile-blocked [B][N/32][N/32][4][64][4] -> [B, key, query]"""
    a = t.detach().double().cpu().numpy().reshape(B, N // 32, N // 32, 4, 2, 32, 4)      # b, jt, qb, g, h, c, e
    a = a.transpose(0, 1, 3, 4, 6, 2, 5)                                                  # b, jt, g, h, e, qb, c
    return a.reshape(B, N, N)                                                              # key = jt*32 + 8g + 4h + e


def block(m):
    """[B, key, query] -> blocked flat (numpy)"""
    B, N, _ = m.shape
    a = m.reshape(B, N // 32, 4, 2, 4, N // 32, 32)                                       # b, jt, g, h, e, qb, c
    return np.ascontiguousarray(a.transpose(0, 1, 5, 2, 3, 6, 4)).reshape(-1)             # b, jt, qb, g, h, c, e


def box_np(c, h, w, xdir=True, ydir=True):
    """diagonal box of c [B, key, query] over the (h, w) grids of both indices"""
    B, N, _ = c.shape
    c6 = c.reshape(B, h, w, h, w)
    out = np.zeros_like(c6)
    for dy in ((-1, 0, 1) if ydir else (0,)):
        for dx in ((-1, 0, 1) if xdir else (0,)):
            ys, yd = slice(max(0, dy), h + min(0, dy)), slice(max(0, -dy), h + min(0, -dy))
            xs, xd = slice(max(0, dx), w + min(0, dx)), slice(max(0, -dx), w + min(0, -dx))
            out[:, yd, xd, yd, xd] += c6[:, ys, xs, ys, xs]
    return out.reshape(B, N, N)


def rel(x, ref):
    return float(np.abs(np.asarray(x, np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))


def main(h=4, w=64, B=2, Cv=37):
    N = h * w
    g = torch.Generator(device=DEV).manual_seed(1)
    th = torch.randn(B, 256, h, w, device=DEV, generator=g) + 0.1
    ph = 0.4 * th.flip(3) + torch.randn(B, 256, h, w, device=DEV, generator=g)
    f64 = lambda t: t.detach().double().cpu().numpy()
    # ---- stage 1: T = xbox(C_raw), keys in the rows
    t = ops.box3_corr_xbox(th, ph)
    c_raw = np.einsum("bcq,bck->bkq", f64(th).reshape(B, 256, N), f64(ph).reshape(B, 256, N))     # [B, key, query]
    print(f"stage 1 xbox GEMM      rel err {rel(unblock(t, B, N), box_np(c_raw, h, w, ydir=False)):.2e}")
    # ---- stage 2: forward
    kc = 2304.0
    from cocosnet_amd.hot_path import _unfold3_stats
    mu, a = _unfold3_stats(th, kc)
    nu, b = _unfold3_stats(ph, kc)
    v = torch.rand(B, Cv, N, device=DEV, generator=g) * 2 - 1
    out = ops.box3_softmax_warp(t, mu, a, nu, b, v, h, w, kc, 100.0)
    S = box_np(c_raw, h, w)                                                                        # [B, key, query]
    z = 100.0 * (S - kc * f64(nu)[:, :, None] * f64(mu)[:, None, :]) * f64(b)[:, :, None] * f64(a)[:, None, :]
    zq = z.transpose(0, 2, 1)                                                                      # [B, query, key]
    P = np.exp(zq - zq.max(-1, keepdims=True)); P /= P.sum(-1, keepdims=True)
    out_ref = np.einsum("bqk,bck->bcq", P, f64(v))
    print(f"stage 2 K19 forward    rel err {rel(f64(out), out_ref):.2e}")
    # ---- stage 3: backward kernel outputs
    go = torch.randn(B, Cv, N, device=DEV, generator=g)
    tt = t.detach().requires_grad_(True)
    leaves = [x.detach().requires_grad_(True) for x in (mu, a, nu, b)]
    vv = v.detach().requires_grad_(True)
    o2 = ops.box3_softmax_warp(tt, *leaves, vv, h, w, kc, 100.0)
    o2.backward(go)
    dP = np.einsum("bcq,bck->bqk", f64(go), f64(v))
    L = P * (dP - (P * dP).sum(-1, keepdims=True))                                                 # [B, q, k]
    A, Bk, MU, NU = f64(a)[:, :, None], f64(b)[:, None, :], f64(mu)[:, :, None], f64(nu)[:, None, :]
    G_ref = (L * 100.0 * A * Bk).transpose(0, 2, 1)                                                # [B, key, query]
    print(f"stage 3 K19 bwd G      rel err {rel(unblock(tt.grad, B, N), G_ref):.2e}")
    print(f"        da             rel err {rel(f64(leaves[1].grad), (L * zq).sum(-1) / f64(a)):.2e}")
    print(f"        dmu            rel err {rel(f64(leaves[0].grad), (L * (-100.0 * kc) * A * Bk * NU).sum(-1)):.2e}")
    print(f"        db             rel err {rel(f64(leaves[3].grad), (L * zq).sum(1) / f64(b)):.2e}")
    print(f"        dnu            rel err {rel(f64(leaves[2].grad), (L * (-100.0 * kc) * A * Bk * MU).sum(1)):.2e}")
    print(f"        dv             rel err {rel(f64(vv.grad), np.einsum('bcq,bqk->bck', f64(go), P)):.2e}")
    # ---- stage 4: K20 + the two GEMMs
    thg, phg = th.detach().requires_grad_(True), ph.detach().requires_grad_(True)
    t3 = ops.box3_corr_xbox(thg, phg)
    Gin = torch.from_numpy(block(G_ref)).float().to(DEV)
    t3.backward(Gin)
    dC = box_np(G_ref, h, w)                                                                       # [B, key, query]
    dth_ref = np.einsum("bkq,bck->bcq", dC, f64(ph).reshape(B, 256, N)).reshape(B, 256, h, w)
    dph_ref = np.einsum("bkq,bcq->bck", dC, f64(th).reshape(B, 256, N)).reshape(B, 256, h, w)
    print(f"stage 4 K20 + GEMMs    d theta rel err {rel(f64(thg.grad), dth_ref):.2e}   d phi rel err {rel(f64(phg.grad), dph_ref):.2e}")


if __name__ == "__main__":
    _lib.load()
    for hh in (4, 8):
        print(f"--- grid {hh} x 64")
        main(h=hh)
