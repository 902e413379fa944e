"""ORACLE (test infrastructure — never imported by the product path).

The same hot path as oracle/corr_oracle.py — /root/reference/models/networks/correspondence.py:272-372,
from the outputs of the theta / phi 1x1 convolutions on — restated on torch CPU tensors so that AUTOGRAD
provides the gradients of every flag combination (cycle terms, column softmax, patches, match_kernel 3)
at any size.  Two uses:

  * float64: the gradient checker of the GPU parity tests at BASELINE sizes (the numpy oracle has
    hand-derived backward pieces for the row pass only);
  * float32 with all host threads: bench.py's `cpu_baseline` — the reference's own op sequence (torch
    matmul / softmax / unfold ... on CPU) timed on the GPU box's host cores, next to the HIP number.

It materialises f [B,HW,HW] exactly like the reference does.  Pinned (tests/test_oracle_golden.py)
against the fixtures under tests/golden/ that were produced by running the reference itself, outputs
and autograd gradients.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import sys

import torch
import torch.nn.functional as F

EPS = sys.float_info.epsilon          # :279 / :288


def _unit_columns(x, pono_c):
    """:277-280 / :287-289 — centre over channels (PONO_C) or over positions, then x / (||x||_2 + eps)."""
    x = x - x.mean(dim=1 if pono_c else -1, keepdim=True)
    return x / (torch.norm(x, 2, 1, keepdim=True) + EPS)


class _WTA(torch.autograd.Function):
    """WTA_scale (:38-77): row maxima kept, the rest scaled; backward 1 at the maxima, a fixed 1e-4 elsewhere."""

    @staticmethod
    def forward(ctx, f, scale):
        keep = f == f.max(dim=-1, keepdim=True)[0]
        ctx.save_for_backward(keep)
        return torch.where(keep, f, f * scale)

    @staticmethod
    def backward(ctx, g):
        (keep,) = ctx.saved_tensors
        return g * torch.where(keep, torch.ones_like(g), torch.full_like(g, 1e-4)), None


def hot_path(theta_raw, phi_raw, ref_img, real_img, seg_map, ref_seg_map, opt, temperature=0.01,
             detach_flag=False, WTA_scale_weight=1, return_corr=False):
    """`coor_out` of NoVGGCorrespondence.forward from :272 on; torch tensors in, torch tensors out (same
    dtype/device as theta_raw).  `opt` needs match_kernel, PONO_C, down, warp_patch, warp_bilinear, isTrain,
    show_corr, warp_mask_losstype, show_warpmask, warp_cycle_w, two_cycle (corr_oracle.default_opt)."""
    B, C, fh, fw = theta_raw.shape
    H, W = ref_img.shape[2:]
    d, mk = opt.down, opt.match_kernel
    out = {}
    flat = lambda t: t.reshape(t.shape[0], t.shape[1], -1)
    rows = lambda p, x: torch.matmul(p, x.transpose(1, 2)).transpose(1, 2)     # P [B,n,m] @ x [B,c,m] -> [B,c,n]

    # :272-289
    if mk == 1:
        th, ph = flat(theta_raw), flat(phi_raw)
    else:
        th = F.unfold(theta_raw, kernel_size=mk, padding=mk // 2)
        ph = F.unfold(phi_raw, kernel_size=mk, padding=mk // 2)
    th, ph = _unit_columns(th, opt.PONO_C), _unit_columns(ph, opt.PONO_C)
    # :291-307
    f = torch.matmul(th.transpose(1, 2), ph)
    if detach_flag:
        f = f.detach()
    if WTA_scale_weight != 1:
        f = _WTA.apply(f, WTA_scale_weight)
    f = f / temperature
    if return_corr:
        return f
    p_row = F.softmax(f, dim=-1)
    # :309-327 (the reference folds to a hard-coded 256; the true size is used here, as in corr_oracle.fold)
    ref = F.unfold(ref_img, d, stride=d) if opt.warp_patch else flat(F.avg_pool2d(ref_img, d))
    y = rows(p_row, ref)
    y_img = F.fold(y, (H, W), d, stride=d) if opt.warp_patch else y.reshape(B, -1, fh, fw)
    up = lambda t, bil: t if opt.warp_patch else (
        F.interpolate(t, scale_factor=d, mode="bilinear", align_corners=False) if bil
        else F.interpolate(t, scale_factor=d, mode="nearest"))
    if (not opt.isTrain) and opt.show_corr:
        out["warp_out_bi"] = up(y_img, True)
    out["warp_out"] = up(y_img, opt.warp_bilinear)
    # :329-346
    p_col = None
    if opt.warp_mask_losstype == "direct" or opt.show_warpmask:
        rs = F.interpolate(ref_seg_map, scale_factor=1 / d, mode="nearest")
        out["warp_mask"] = rows(p_row, flat(rs)).reshape(B, -1, fh, fw)
    elif opt.warp_mask_losstype == "cycle":
        p_col = F.softmax(f.transpose(1, 2), dim=-1)
        sg = F.interpolate(seg_map, scale_factor=1 / d, mode="nearest")
        out["warp_mask"] = rows(p_row, rows(p_col, flat(sg))).reshape(B, -1, fh, fw)
    # :350-372
    if opt.warp_cycle_w > 0:
        if p_col is None:
            p_col = F.softmax(f.transpose(1, 2), dim=-1)
        if opt.warp_patch:
            out["warp_cycle"] = F.fold(rows(p_col, F.unfold(y_img, d, stride=d)), (H, W), d, stride=d)
        else:
            out["warp_cycle"] = rows(p_col, flat(y_img)).reshape(B, -1, fh, fw)
            if opt.two_cycle:
                i2r = rows(p_col, flat(F.avg_pool2d(real_img, d)))
                out["warp_i2r"] = i2r.reshape(B, -1, fh, fw)
                out["warp_i2r2i"] = rows(p_row, i2r).reshape(B, -1, fh, fw)
    return out


def forward_backward(theta_raw, phi_raw, ref_img, real_img, seg_map, ref_seg_map, opt, cotangents,
                     dtype=torch.float64, device=None, **fwd):
    """numpy (or torch) in -> (outputs dict, d theta_raw, d phi_raw) as numpy, for loss = sum_k <out_k, cotangents[k]>
    (keys missing from `cotangents` do not enter the loss).  `device`: where the framework's fp64 ops run (None = CPU);
    the GPU tests at BASELINE sizes pass "cuda" so that EVERY sample of a batch can be checked (a K = 2304 unfolded
    sample is 77 GFLOP in fp64: seconds on the host, milliseconds on the device) and cross-check one sample against the
    CPU run of the same code."""
    import numpy as np

    def t(a, g=False):
        if a is None:
            return None
        a = a.detach() if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=device or "cpu", dtype=dtype).clone().requires_grad_(g)
    th, ph = t(theta_raw, True), t(phi_raw, True)
    out = hot_path(th, ph, t(ref_img), t(real_img), t(seg_map), t(ref_seg_map), opt, **fwd)
    res = out if isinstance(out, dict) else {"corr": out}
    loss = sum((res[k] * t(g)).sum() for k, g in cotangents.items() if k in res)
    loss.backward()
    return ({k: v.detach().cpu().numpy() for k, v in res.items()}, th.grad.cpu().numpy(), ph.grad.cpu().numpy())
