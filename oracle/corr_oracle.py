"""ORACLE (test infrastructure — never imported by the product path).

CPU / numpy restatement of the reference's dense-correspondence hot path,
/root/reference/models/networks/correspondence.py:271-372 (+ WTA_scale :38-77), starting from
the outputs of the `theta` / `phi` 1x1 convolutions (:272, :282).  Each function cites the
reference lines it follows.  Default arithmetic is float64 (the high-precision arbiter between
the fp32 reference and the fp32 HIP kernels: T = 0.01 amplifies GEMM rounding 100x);
pass dtype=np.float32 to mimic the reference's precision (used as bench.py's CPU baseline).

Pinned (tests/test_oracle_golden.py) against fixtures in tests/golden/ that were produced by
RUNNING the reference's own code in the build container (oracle/make_golden.py): the reference
ships no tests or golden vectors for this path, so live execution is the pin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import sys
from types import SimpleNamespace

import numpy as np

EPS = sys.float_info.epsilon   # correspondence.py:279 `+ sys.float_info.epsilon`


# ------------------------------------------------------------------------------------------------
# small tensor helpers restating the torch ops the path calls
# ------------------------------------------------------------------------------------------------
def unfold(x, k, padding=0, stride=1):
    """F.unfold(x, k, padding=padding, stride=stride): [B,C,H,W] -> [B, C*k*k, L], channel index
    c*k*k + ky*k + kx, zero padding, L row-major over output positions (:276, :286, :311, :352)."""
    B, C, H, W = x.shape
    xp = np.zeros((B, C, H + 2 * padding, W + 2 * padding), dtype=x.dtype)
    xp[:, :, padding:padding + H, padding:padding + W] = x
    oh = (H + 2 * padding - k) // stride + 1
    ow = (W + 2 * padding - k) // stride + 1
    out = np.empty((B, C, k, k, oh, ow), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            out[:, :, ky, kx] = xp[:, :, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride]
    return out.reshape(B, C * k * k, oh * ow)


def fold(cols, out_hw, k, stride):
    """F.fold(cols, out_hw, k, stride=stride) for NON-overlapping patches (stride == k), :321,:357.
    The reference hard-codes out_hw = 256 there; callers pass the true image size."""
    assert stride == k
    H, W = out_hw
    B, CKK, L = cols.shape
    C = CKK // (k * k)
    oh, ow = H // k, W // k
    assert L == oh * ow
    x = cols.reshape(B, C, k, k, oh, ow).transpose(0, 1, 4, 2, 5, 3)
    return np.ascontiguousarray(x).reshape(B, C, H, W)


def avg_pool2d(x, k):
    """F.avg_pool2d(x, k) with stride k (:313, :365)."""
    B, C, H, W = x.shape
    return x[:, :, :H // k * k, :W // k * k].reshape(B, C, H // k, k, W // k, k).mean(axis=(3, 5))


def nearest_down(x, down):
    """F.interpolate(x, scale_factor=1/down, mode='nearest') == F.interpolate(x, size=(H/down,
    W/down), mode='nearest'): source index floor(dst * down)  (:258-259, :330, :339)."""
    return x[:, :, ::down, ::down][:, :, :x.shape[2] // down, :x.shape[3] // down]


def upsample_nearest(x, s):
    """nn.Upsample(scale_factor=s) (default mode 'nearest'), :188, :327."""
    return np.repeat(np.repeat(x, s, axis=2), s, axis=3)


def upsample_bilinear(x, s):
    """nn.Upsample(scale_factor=s, mode='bilinear') with align_corners=False (:184-186, :326)."""
    def axis_weights(n_in, n_out):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) / s - 0.5
        src = np.maximum(src, 0.0)
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        w1 = src - i0
        return i0, i1, (1.0 - w1), w1
    B, C, H, W = x.shape
    y0, y1, wy0, wy1 = axis_weights(H, H * s)
    x0, x1, wx0, wx1 = axis_weights(W, W * s)
    rows = x[:, :, y0, :] * wy0[None, None, :, None] + x[:, :, y1, :] * wy1[None, None, :, None]
    out = rows[:, :, :, x0] * wx0[None, None, None, :] + rows[:, :, :, x1] * wx1[None, None, None, :]
    return out.astype(x.dtype)


def softmax(x, axis=-1):
    """F.softmax (:307, :338, :351)."""
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


# ------------------------------------------------------------------------------------------------
# K1: centre + L2-normalise    (:277-280 / :287-289)
# ------------------------------------------------------------------------------------------------
def center_l2norm(x, pono_c):
    """x [B,K,N].  dim_mean = 1 if PONO_C else -1 (:277); norm over dim 1 + epsilon (:279)."""
    xc = x - x.mean(axis=1 if pono_c else -1, keepdims=True)
    nrm = np.sqrt((xc * xc).sum(axis=1, keepdims=True))
    return xc / (nrm + EPS)


def center_l2norm_bwd(x, dy, pono_c):
    """Hand-derived gradient of center_l2norm (what autograd does for :277-280)."""
    axis = 1 if pono_c else -1
    xc = x - x.mean(axis=axis, keepdims=True)
    nrm = np.sqrt((xc * xc).sum(axis=1, keepdims=True))
    u = 1.0 / (nrm + EPS)
    y = xc * u
    a = (dy * y).sum(axis=1, keepdims=True)
    g = np.where(nrm > 0, a / np.where(nrm > 0, nrm, 1.0), 0.0)
    dxc = u * dy - g * y
    return dxc - dxc.mean(axis=axis, keepdims=True)


# ------------------------------------------------------------------------------------------------
# K2: correlation -> /T -> softmax -> warp    (:281, :291, :304, :307, :318)
# ------------------------------------------------------------------------------------------------
def correlation(qn, kn):
    """f = matmul(theta.permute(0,2,1), phi)  (:281, :291): [B,K,Nq],[B,K,Nk] -> [B,Nq,Nk]."""
    return np.matmul(qn.transpose(0, 2, 1), kn)


def corr_softmax_warp(qn, kn, v, inv_t):
    """out[b,c,i] = sum_j softmax_j(f[b,i,j] * inv_t) v[b,c,j]; v channel-major [B,Cv,Nk]."""
    p = softmax(correlation(qn, kn) * inv_t, axis=-1)
    return np.matmul(p, v.transpose(0, 2, 1)).transpose(0, 2, 1)


def corr_softmax_warp_bwd(qn, kn, v, dout, inv_t):
    """Gradients of corr_softmax_warp w.r.t. (qn, kn, v) — autograd of :291-318 restated."""
    p = softmax(correlation(qn, kn) * inv_t, axis=-1)            # [B,Nq,Nk]
    dp = np.matmul(dout.transpose(0, 2, 1), v)                    # [B,Nq,Nk]
    ds = p * (dp - (p * dp).sum(axis=-1, keepdims=True)) * inv_t  # d/d f
    dqn = np.matmul(kn, ds.transpose(0, 2, 1))                    # [B,K,Nq]
    dkn = np.matmul(qn, ds)                                       # [B,K,Nk]
    dv = np.matmul(dout, p)                                       # [B,Cv,Nk]
    return dqn, dkn, dv


def wta_scale(f, scale):
    """WTA_scale.forward (:47-58): keep the row maxima, multiply everything else by `scale`."""
    mx = f.max(axis=-1, keepdims=True)
    return np.where(f == mx, f, f * scale)


def wta_scale_bwd(f, grad_out):
    """WTA_scale.backward (:61-77): grad * (1 at the row maxima, hard-coded 1e-4 elsewhere)."""
    mx = f.max(axis=-1, keepdims=True)
    return grad_out * np.where(f == mx, 1.0, 1e-4)


# ------------------------------------------------------------------------------------------------
# the whole hot path, :271-372
# ------------------------------------------------------------------------------------------------
def default_opt(**kw):
    o = dict(match_kernel=1, PONO_C=True, down=4, warp_patch=False, warp_bilinear=False,
             isTrain=False, show_corr=False, warp_mask_losstype="direct", show_warpmask=False,
             warp_cycle_w=0.0, two_cycle=False)
    o.update(kw)
    return SimpleNamespace(**o)


def hot_path_forward(theta_raw, phi_raw, ref_img, real_img, seg_map, ref_seg_map, opt,
                     temperature=0.01, WTA_scale_weight=1, return_corr=False, dtype=np.float64):
    """Restatement of NoVGGCorrespondence.forward from :272 (theta/phi conv outputs) to :374.

    theta_raw, phi_raw [B,256,h,w]; ref_img/real_img [B,3,H,W]; seg maps [B,nc,H,W].
    Returns the reference's `coor_out` dict (numpy arrays), or f_WTA/T if return_corr (:305-306).
    `detach_flag` (:292-293) only changes gradients and has no forward counterpart here."""
    cast = lambda a: None if a is None else np.asarray(a, dtype=dtype)
    theta_raw, phi_raw, ref_img, real_img = map(cast, (theta_raw, phi_raw, ref_img, real_img))
    seg_map, ref_seg_map = cast(seg_map), cast(ref_seg_map)
    B, C, fh, fw = theta_raw.shape
    H, W = ref_img.shape[2:]
    down = opt.down
    mk = opt.match_kernel
    out = {}

    # :272-289
    if mk == 1:
        theta = theta_raw.reshape(B, C, -1)
        phi = phi_raw.reshape(B, C, -1)
    else:
        theta = unfold(theta_raw, mk, padding=mk // 2)
        phi = unfold(phi_raw, mk, padding=mk // 2)
    theta = center_l2norm(theta, opt.PONO_C)
    phi = center_l2norm(phi, opt.PONO_C)

    # :291-307
    f = correlation(theta, phi)
    f_wta = f if WTA_scale_weight == 1 else wta_scale(f, WTA_scale_weight)
    f_wta = f_wta / temperature
    if return_corr:
        return f_wta
    p = softmax(f_wta, axis=-1)                       # f_div_C

    # :309-327
    if opt.warp_patch:
        ref = unfold(ref_img, down, stride=down)      # [B, 3*down^2, HW]
    else:
        ref = avg_pool2d(ref_img, down).reshape(B, ref_img.shape[1], -1)
    y = np.matmul(p, ref.transpose(0, 2, 1))          # [B,HW,ch]
    if opt.warp_patch:
        y_img = fold(y.transpose(0, 2, 1), (H, W), down, down)
    else:
        y_img = y.transpose(0, 2, 1).reshape(B, -1, fh, fw)
    if (not opt.isTrain) and opt.show_corr:
        out["warp_out_bi"] = y_img if opt.warp_patch else upsample_bilinear(y_img, down)
    if opt.warp_patch:
        out["warp_out"] = y_img
    else:
        out["warp_out"] = (upsample_bilinear if opt.warp_bilinear else upsample_nearest)(y_img, down)

    # :329-346
    if opt.warp_mask_losstype == "direct" or opt.show_warpmask:
        rs = nearest_down(ref_seg_map, down)
        ch = rs.shape[1]
        wm = np.matmul(p, rs.reshape(B, ch, -1).transpose(0, 2, 1))
        out["warp_mask"] = wm.transpose(0, 2, 1).reshape(B, ch, fh, fw)
    elif opt.warp_mask_losstype == "cycle":
        p_v = softmax(f_wta.transpose(0, 2, 1), axis=-1)
        sg = nearest_down(seg_map, down)
        ch = sg.shape[1]
        to_ref = np.matmul(p_v, sg.reshape(B, ch, -1).transpose(0, 2, 1))
        wm = np.matmul(p, to_ref)
        out["warp_mask"] = wm.transpose(0, 2, 1).reshape(B, ch, fh, fw)

    # :350-372
    if opt.warp_cycle_w > 0:
        p_v = softmax(f_wta.transpose(0, 2, 1), axis=-1)
        if opt.warp_patch:
            yy = unfold(y_img, down, stride=down).transpose(0, 2, 1)
            wc = np.matmul(p_v, yy).transpose(0, 2, 1)
            out["warp_cycle"] = fold(wc, (H, W), down, down)
        else:
            ch = y_img.shape[1]
            yy = y_img.reshape(B, ch, -1).transpose(0, 2, 1)
            wc = np.matmul(p_v, yy).transpose(0, 2, 1)
            out["warp_cycle"] = wc.reshape(B, ch, fh, fw)
            if opt.two_cycle:
                ri = avg_pool2d(real_img, down).reshape(B, ch, -1).transpose(0, 2, 1)
                i2r = np.matmul(p_v, ri).transpose(0, 2, 1).reshape(B, ch, fh, fw)
                i2r2i = np.matmul(p, i2r.reshape(B, ch, -1).transpose(0, 2, 1))
                out["warp_i2r"] = i2r
                out["warp_i2r2i"] = i2r2i.transpose(0, 2, 1).reshape(B, ch, fh, fw)
    return out
