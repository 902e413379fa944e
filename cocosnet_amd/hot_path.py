"""Host-side orchestration of the correspondence hot path (correspondence.py:271-372).

Takes the outputs of the `theta` / `phi` 1x1 convolutions and returns the reference's `coor_out`
dictionary.  Every correlation / softmax / warp product is a HIP kernel call (cocosnet_amd.ops);
the only PyTorch ops left are the tiny, shape-only ones the survey keeps on the stock backend:
avg-pool / nearest-resize of the 3..151-channel inputs, fold/unfold of patches and the final
x`down` up-sampling.

Pass structure on the fused path (K == 256).  The reference materialises f once and reuses it;
here each use is one fused launch that recomputes its logits tiles (f never reaches HBM), with all
the channel groups that share a softmax concatenated into a single V:
    R1  rows    softmax_j(f)    V = [exemplar rgb | ref_seg (direct mask)]        :318, :334
    C1  columns softmax_i(f^T)  V = [seg (cycle mask) | y (warp_cycle) | real]    :338-343, :351-367
        — the same kernel with theta/phi swapped (SURVEY.md §7 design notes)
    R2  rows    softmax_j(f)    V = [C1's mask | C1's i2r]                        :344, :369
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import ops


@dataclass
class HotPathConfig:
    """The `opt` fields the hot path reads, detached from argparse."""
    match_kernel: int = 3
    PONO_C: bool = False
    down: int = 4
    warp_patch: bool = False
    warp_bilinear: bool = False
    isTrain: bool = False
    show_corr: bool = False
    warp_mask_losstype: str = "none"
    show_warpmask: bool = False
    warp_cycle_w: float = 0.0
    two_cycle: bool = False

    @classmethod
    def from_opt(cls, opt, down=None):
        g = lambda k, d: getattr(opt, k, d)
        return cls(match_kernel=int(g("match_kernel", 3)), PONO_C=bool(g("PONO_C", False)),
                   down=int(down if down is not None else g("down", 4)),
                   warp_patch=bool(g("warp_patch", False)),
                   warp_bilinear=bool(g("warp_bilinear", False)), isTrain=bool(g("isTrain", False)),
                   show_corr=bool(g("show_corr", False)),
                   warp_mask_losstype=str(g("warp_mask_losstype", "none")),
                   show_warpmask=bool(g("show_warpmask", False)),
                   warp_cycle_w=float(g("warp_cycle_w", 0.0)), two_cycle=bool(g("two_cycle", False)))


class _SplitChannels(torch.autograd.Function):
    """x[:, :n], x[:, n:] whose backward is ONE concatenation of the two gradients.  Plain slicing makes autograd
    build two zero-filled full-size tensors, copy a slice into each and add them: five kernels and 100 MB of traffic
    for the 20 MB gradient of the row pass (37 us per step at the benchmark shape, now 12)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.shapes = (x.shape, int(n))
        return x[:, :n], x[:, n:]

    @staticmethod
    def backward(ctx, ga, gb):
        shape, n = ctx.shapes
        parts = []
        for g, c in ((ga, n), (gb, shape[1] - n)):
            parts.append(g if g is not None else torch.zeros((shape[0], c) + tuple(shape[2:]), device=(ga if ga is not None else gb).device,
                                                             dtype=(ga if ga is not None else gb).dtype))
        if ga is not None and gb is not None and ga.is_cuda and ga.dtype == torch.float32 and gb.dtype == torch.float32:
            return ops.concat_channels_amax(ga, gb), None      # one kernel, and max|.| for the K2 backward's f16 split
        return torch.cat(parts, dim=1), None


def _split_channels(x, n):
    return _SplitChannels.apply(x, n)


def _hip_fp32(t):
    return t is not None and t.is_cuda and t.dtype == torch.float32


def _flat(x):
    """[B,C,h,w] -> channel-major [B,C,h*w] (a view when contiguous)."""
    return x.reshape(x.shape[0], x.shape[1], -1)


def _upsample(y, down, bilinear):
    if bilinear:   # nn.Upsample(scale_factor, mode='bilinear'), align_corners=False (:184-186)
        return F.interpolate(y, scale_factor=down, mode="bilinear", align_corners=False)
    if y.is_cuda and y.dtype == torch.float32 and (y.shape[3] * down) % 4 == 0:
        return ops.upsample_nearest(y, down)                     # :188 (K11)
    return F.interpolate(y, scale_factor=down, mode="nearest")


class _Attention:
    """The two softmax directions of one correlation.  Three back ends:
       fused      K = 256 logits computed inside the kernels (match_kernel 1);
       streamed   logits materialised once per orientation, softmax + warp fused (match_kernel 3);
       generic    materialised logits -> row softmax -> GEMM (return_corr / WTA / everything else)."""

    def __init__(self, qn=None, kn=None, inv_t=1.0, f_scaled=None, logits_q=None, logits_k=None, boxed=None, planes=None):
        self.qn, self.kn, self.inv_t = qn, kn, inv_t
        self.fused = qn is not None
        self._boxed = boxed                          # match_kernel 3 fused (K19 / K20): a _BoxedCorr
        self._f = f_scaled
        self._lq, self._lk = logits_q, logits_k      # providers: query-major f / key-major f^T
        self._cache = {}
        # f16 operand planes of theta/phi, shared by the row / column / second row pass of THIS forward call
        # (per-call object: nothing is cached across calls or threads)
        self._planes = (planes if planes is not None else ops.OperandPlanes()) if self.fused else None

    def _get(self, name, fn):
        if name not in self._cache:
            self._cache[name] = fn()
        return self._cache[name]

    @property
    def f_scaled(self):
        return self._f if self._f is not None else self._get("lq", self._lq)

    def rows(self, v):
        """softmax over exemplar positions, then @ v   (f_div_C @ v, :307/:318)."""
        if self.fused:
            return ops.corr_softmax_warp(self.qn, self.kn, v, self.inv_t, self._planes)
        if self._boxed is not None:
            return self._boxed.rows(v)
        if self._lk is not None:
            return ops.logits_softmax_warp(self._get("lk", self._lk), v)
        return ops.warp_materialized(self._get("p_row", lambda: ops.row_softmax(self._f)), v)

    def cols(self, v):
        """softmax over content positions of f^T, then @ v   (f_div_C_v @ v, :338/:351)."""
        if self.fused:
            return ops.corr_softmax_warp(self.kn, self.qn, v, self.inv_t, self._planes)
        if self._boxed is not None:
            return self._boxed.cols(v)
        if self._lq is not None:   # f itself is the key-major logit matrix of the swapped problem
            return ops.logits_softmax_warp(self._get("lq", self._lq), v)
        return ops.warp_materialized(
            self._get("p_col", lambda: ops.row_softmax(self._f.transpose(1, 2).contiguous())), v)


class _BoxedCorr:
    """match_kernel 3, PONO_C, 64- or 128-wide grid: the statistics of the unfolded vectors (K12) and, per orientation that is
    actually used, T = xbox(C_raw) from the correlation GEMM's epilogue; every softmax + warp pass then reads T three
    blocks at a time (K19).  Nothing box-filtered and no logits matrix reaches HBM."""

    def __init__(self, theta_raw, phi_raw, inv_t):
        _, C, self.fh, self.fw = theta_raw.shape
        self.kc = float(C * 9)
        self._cache = {}
        self._raw_planes = None
        if isinstance(theta_raw, ops.LazyProj1x1) and ops.proj_raw_fused_ok(theta_raw, phi_raw):
            # round 6, K25: ONE launch from the features of both tensors to the operand planes of the raw projections and the sums
            # behind K12's statistics — the fp32 projections never exist; theta_raw / phi_raw are autograd handles from here on
            self._raw_planes = ops.Box3RawPlanes()
            (theta_raw, mu, a), (phi_raw, nu, b) = ops.proj_raw_planes_stats_pair(theta_raw, phi_raw, self.kc, self._raw_planes)
            self._cache["q"], self._cache["k"] = (mu, a), (nu, b)
        elif isinstance(theta_raw, ops.LazyProj1x1):
            # round 6: projection + K12 statistics as ONE autograd node per tensor (ops.proj_unfold3_stats): its backward folds K12's
            # backward and the sum of theta_raw's two gradients into the projection's input gradient (K24)
            if ops.PROJ_PRECISION == "f16x3":      # max|.| of both feature tensors and both weights in one launch
                ops.prefetch_amax([theta_raw.x, theta_raw.weight.reshape(C, -1), phi_raw.x, phi_raw.weight.reshape(C, -1)])
            theta_raw, mu, a = ops.proj_unfold3_stats(theta_raw, self.kc)
            phi_raw, nu, b = ops.proj_unfold3_stats(phi_raw, self.kc)
            self._cache["q"], self._cache["k"] = (mu, a), (nu, b)
        self.th, self.ph, self.inv_t = theta_raw, phi_raw, inv_t
        # round 4: ONE T for both orientations (xbox(C)^T = xbox(C^T): the column pass reads it transposed) and one gradient
        # buffer for all passes over it — no second correlation GEMM, one box adjoint and one pair of GEMMs in the backward
        self._sink = ops.Box3GradSink() if ops.BOX3_SHARE_T else None

    def _get(self, name, fn):
        if name not in self._cache:
            self._cache[name] = fn()
        return self._cache[name]

    def rows(self, v):
        mu, a = self._get("q", lambda: _unfold3_stats(self.th, self.kc))
        nu, b = self._get("k", lambda: _unfold3_stats(self.ph, self.kc))
        t = self._get("t_rows", lambda: ops.box3_corr_xbox(self.th, self.ph, self._sink, self._raw_planes))
        return ops.box3_softmax_warp(t, mu, a, nu, b, v, self.fh, self.fw, self.kc, self.inv_t, sink=self._sink)

    def cols(self, v):   # the same operator with the roles of theta and phi exchanged
        mu, a = self._get("q", lambda: _unfold3_stats(self.th, self.kc))
        nu, b = self._get("k", lambda: _unfold3_stats(self.ph, self.kc))
        if self._sink is not None:
            t = self._get("t_rows", lambda: ops.box3_corr_xbox(self.th, self.ph, self._sink, self._raw_planes))
            return ops.box3_softmax_warp(t, nu, b, mu, a, v, self.fh, self.fw, self.kc, self.inv_t, transposed=True, sink=self._sink)
        t = self._get("t_cols", lambda: ops.box3_corr_xbox(self.ph, self.th, None, self._raw_planes))
        return ops.box3_softmax_warp(t, nu, b, mu, a, v, self.fh, self.fw, self.kc, self.inv_t)


def _box_sum3(x):
    """zero-padded 3x3 box SUM of a [B,1,h,w] map."""
    return F.avg_pool2d(x, 3, stride=1, padding=1, divisor_override=1)


def _unfold3_stats(x_raw, k_unfolded):
    """mean and 1/(norm+eps) of the zero-padded 3x3-unfolded, centred vectors of x_raw [B,C,h,w]
    WITHOUT unfolding: box sums of the per-position channel sums / sums of squares (:276-280)."""
    if x_raw.is_cuda and x_raw.dtype == torch.float32:
        return ops.unfold3_stats(x_raw, k_unfolded)                  # K12: one pass over x, fwd and bwd
    s1 = x_raw.sum(dim=1, keepdim=True)
    s2 = (x_raw * x_raw).sum(dim=1, keepdim=True)
    mu = _box_sum3(s1) / k_unfolded
    nrm = torch.sqrt(torch.clamp(_box_sum3(s2) - k_unfolded * mu * mu, min=0.0))
    B = x_raw.shape[0]
    return mu.reshape(B, -1), (1.0 / (nrm + ops.NORM_EPS)).reshape(B, -1)


def _box3_logits(theta_raw, phi_raw, scale, transposed=False):
    """match_kernel 3, PONO_C: f (or f^T) * scale from the K = 256 correlation, no unfold (K6)."""
    B, C, fh, fw = theta_raw.shape
    kc = float(C * 9)
    mu, a = _unfold3_stats(theta_raw, kc)
    nu, b = _unfold3_stats(phi_raw, kc)
    if transposed:   # f^T[q,p]: the same operator with the roles of theta and phi exchanged
        c_raw = ops.corr_materialize(_flat(phi_raw), _flat(theta_raw), 1.0)
        return ops.box3_logits(c_raw, nu, mu, b, a, fh, fw, kc, scale)
    c_raw = ops.corr_materialize(_flat(theta_raw), _flat(phi_raw), 1.0)
    return ops.box3_logits(c_raw, mu, nu, a, b, fh, fw, kc, scale)


def _scaled_logits(theta_raw, phi_raw, cfg, inv_t, detach_flag, wta):
    """Materialised, query-major f_WTA / temperature (:272-304) for the generic back end."""
    mk = cfg.match_kernel
    wta_on = wta != 1
    pre_scaled = not (wta_on or detach_flag)
    if mk == 3 and cfg.PONO_C:
        f = _box3_logits(theta_raw, phi_raw, inv_t if pre_scaled else 1.0)
    else:
        if mk == 1:
            theta, phi = _flat(theta_raw), _flat(phi_raw)
        else:
            theta = F.unfold(theta_raw, kernel_size=mk, padding=mk // 2)
            phi = F.unfold(phi_raw, kernel_size=mk, padding=mk // 2)
        qn = ops.center_l2norm(theta, cfg.PONO_C)
        kn = ops.center_l2norm(phi, cfg.PONO_C)
        f = ops.corr_materialize(qn, kn, inv_t if pre_scaled else 1.0)      # :291 (+ :304)
    if pre_scaled:
        return f
    if detach_flag:                                                          # :292-293
        f = f.detach()
    if wta_on:
        return ops.wta_scale(f, wta, inv_t)                                  # :300-303 + :304 (K8)
    return f * inv_t                                                         # :304 (detached: no graph)


def correspondence_hot_path(theta_raw, phi_raw, ref_img, real_img, seg_map, ref_seg_map,
                            cfg: HotPathConfig, temperature=0.01, detach_flag=False,
                            WTA_scale_weight=1, return_corr=False):
    """correspondence.py:272-372 from the theta/phi conv outputs onward.

    theta_raw, phi_raw: [B,256,h,w] CUDA fp32.  Returns the `coor_out` dict of the reference
    (or the scaled correlation [B,HW,HW] when return_corr, :305-306)."""
    B, C, fh, fw = theta_raw.shape
    H, W = ref_img.shape[2], ref_img.shape[3]
    down, mk = cfg.down, cfg.match_kernel
    out = {}

    inv_t = 1.0 / temperature
    fused = (mk == 1) and (C == ops.FUSED_K) and WTA_scale_weight == 1 and not return_corr
    # round 6: theta / phi may arrive as LAZY projections (ops.LazyProj1x1: the 1x1 convolutions of :272 / :282 not run yet).  On the
    # fused match_kernel-1 / PONO_C path K23 then goes from the features straight to the correlation kernels' operand planes
    # (projection + centring + normalisation in one launch for both tensors: the fp32 projections never exist); every other
    # back end asks for the projections (K0) and continues as before.
    lazy = isinstance(theta_raw, ops.LazyProj1x1) and isinstance(phi_raw, ops.LazyProj1x1)
    if lazy != (isinstance(theta_raw, ops.LazyProj1x1) or isinstance(phi_raw, ops.LazyProj1x1)):
        raise TypeError("correspondence_hot_path: theta and phi must both be tensors or both be ops.LazyProj1x1")
    k23 = boxed_lazy = False
    if lazy:
        keep = torch.is_grad_enabled() and not detach_flag and (theta_raw.requires_grad or phi_raw.requires_grad)
        k23 = (fused and cfg.PONO_C and theta_raw.x.shape == phi_raw.x.shape and ops.proj_norm_fused_ok(theta_raw)
               and ops.proj_norm_fused_ok(phi_raw) and ops.corr_split_ok(B, C, fh * fw, fh * fw, 1, keep))
        # match_kernel 3, fused family: the projections stay lazy up to _BoxedCorr, which makes (theta_raw, mu, a) in one autograd node
        boxed_lazy = (mk == 3 and cfg.PONO_C and WTA_scale_weight == 1 and not return_corr and _hip_fp32(theta_raw)
                      and ops.PROJ_BWD_FUSED and ops.box3_fused_ok(B, C, fh, fw))
        if not (k23 or boxed_lazy):
            theta_raw, phi_raw = theta_raw.raw(), phi_raw.raw()
    if k23:
        th_l, ph_l = (theta_raw.detach(), phi_raw.detach()) if detach_flag else (theta_raw, phi_raw)   # :292-293 `f = f.detach()`
        planes = ops.OperandPlanes()
        qn, kn = ops.proj_center_l2norm_planes_pair(th_l, ph_l, 1, planes, want_chan=keep)
        attn = _Attention(qn=qn, kn=kn, inv_t=inv_t, planes=planes)
    elif fused:
        # :272-289 — flatten, centre, L2-normalise; the rest happens inside the fused kernels
        th_f, ph_f = _flat(theta_raw), _flat(phi_raw)
        if detach_flag:   # :292-293 `f = f.detach()`: nothing upstream of f receives a gradient
            th_f, ph_f = th_f.detach(), ph_f.detach()
        keep = torch.is_grad_enabled() and (th_f.requires_grad or ph_f.requires_grad)
        planes = ops.OperandPlanes()
        if (cfg.PONO_C and _hip_fp32(theta_raw)
                and ops.corr_split_ok(B, C, fh * fw, fh * fw, 1, keep)):
            # K1 writes the operand planes of the split kernels itself: fp32 qn / kn never exist (ops.center_l2norm_planes)
            # (channel-major planes of BOTH operands whenever either is differentiated: the K2 backward reads both)
            qn = ops.center_l2norm_planes(th_f, 1, planes, want_chan=keep)
            kn = ops.center_l2norm_planes(ph_f, 1, planes, want_chan=keep)
        else:
            qn = ops.center_l2norm(th_f, cfg.PONO_C)
            kn = ops.center_l2norm(ph_f, cfg.PONO_C)
        attn = _Attention(qn=qn, kn=kn, inv_t=inv_t, planes=planes)
    elif (mk == 3 and cfg.PONO_C and WTA_scale_weight == 1 and not return_corr and _hip_fp32(theta_raw)
          and ops.box3_fused_ok(B, C, fh, fw)):
        # the reference's default: 3x3 neighbourhoods, fused (round 3): x box in the correlation GEMM, y box + softmax + warp
        # in one kernel — see _BoxedCorr
        th, ph = (theta_raw.detach(), phi_raw.detach()) if detach_flag else (theta_raw, phi_raw)
        attn = _Attention(inv_t=inv_t, boxed=_BoxedCorr(th, ph, inv_t))
    elif mk == 3 and cfg.PONO_C and WTA_scale_weight == 1:
        # every other shape / the exact-fp32 flavour: logits = diagonal box filter of the K = 256
        # correlation, materialised once per orientation that is actually used, then streamed.
        th, ph = (theta_raw.detach(), phi_raw.detach()) if detach_flag else (theta_raw, phi_raw)
        attn = _Attention(inv_t=inv_t,
                          logits_q=lambda: _box3_logits(th, ph, inv_t, transposed=False),
                          logits_k=lambda: _box3_logits(th, ph, inv_t, transposed=True))
    else:
        attn = _Attention(inv_t=inv_t,
                          f_scaled=_scaled_logits(theta_raw, phi_raw, cfg, inv_t, detach_flag, WTA_scale_weight))
    if return_corr:
        return attn.f_scaled

    # ---- R1: exemplar colours (+ direct mask) through the row softmax  (:309-336) ----------------
    direct_mask = cfg.warp_mask_losstype == "direct" or cfg.show_warpmask
    fused_values = (direct_mask and not cfg.warp_patch and _hip_fp32(ref_img) and _hip_fp32(ref_seg_map)
                    and not (ref_img.requires_grad or ref_seg_map.requires_grad) and H % down == 0 and W % down == 0)
    if fused_values:
        n_ref = ref_img.shape[1]
        v1 = _flat(ops.warp_values(ref_img, ref_seg_map, down))       # pooled image + sampled mask in one kernel (K14)
    else:
        if cfg.warp_patch:
            ref = F.unfold(ref_img, down, stride=down)                    # [B, 3*down^2, HW]
        else:
            ref = _flat(F.avg_pool2d(ref_img, down))                      # [B, 3, HW]
        n_ref = ref.shape[1]
        v_r1 = [ref]
        if direct_mask:
            ref_seg = F.interpolate(ref_seg_map, scale_factor=1 / down, mode="nearest")
            v_r1.append(_flat(ref_seg))
        v1 = torch.cat(v_r1, dim=1) if len(v_r1) > 1 else ref
    o_r1 = attn.rows(v1)
    # round 6: when the row pass's output goes to the up-sampling and the direct mask and nowhere else, both come from one op whose
    # backward hands the K2 / K19 backward its d out, max|d out| and D in ONE kernel (ops.warp_head)
    head = (direct_mask and not cfg.warp_patch and not cfg.warp_bilinear and not (cfg.warp_cycle_w > 0)
            and not ((not cfg.isTrain) and cfg.show_corr) and ops.warp_head_ok(o_r1, n_ref, fh, fw, down))
    if head:
        out["warp_out"], out["warp_mask"] = ops.warp_head(o_r1, n_ref, fh, fw, down)
        y = y_img = None          # (only the cycle terms read them: excluded above)
    else:
        y, o_mask = _split_channels(o_r1, n_ref) if direct_mask else (o_r1, None)   # [B, ch, HW], [B, nc, HW]
        if cfg.warp_patch:
            y_img = F.fold(y, (H, W), down, stride=down)                  # reference hard-codes 256 (:321)
        else:
            y_img = y.reshape(B, n_ref, fh, fw)
        if (not cfg.isTrain) and cfg.show_corr:
            out["warp_out_bi"] = y_img if cfg.warp_patch else _upsample(y_img, down, True)
        out["warp_out"] = y_img if cfg.warp_patch else _upsample(y_img, down, cfg.warp_bilinear)
        if direct_mask:
            out["warp_mask"] = o_mask.reshape(B, -1, fh, fw)

    # ---- C1: everything that goes through the column softmax  (:337-343, :350-367) ---------------
    cycle_mask = (not direct_mask) and cfg.warp_mask_losstype == "cycle"
    want_cycle = cfg.warp_cycle_w > 0
    want_two = want_cycle and cfg.two_cycle and not cfg.warp_patch
    v_c1, names = [], []
    if cycle_mask:
        seg = F.interpolate(seg_map, scale_factor=1 / down, mode="nearest")
        v_c1.append(_flat(seg)); names.append("mask")
    if want_cycle:
        yy = F.unfold(y_img, down, stride=down) if cfg.warp_patch else y
        v_c1.append(yy); names.append("cycle")
    if want_two:
        v_c1.append(_flat(F.avg_pool2d(real_img, down))); names.append("i2r")
    if v_c1:
        o_c1 = attn.cols(torch.cat(v_c1, dim=1) if len(v_c1) > 1 else v_c1[0])
        parts = dict(zip(names, torch.split(o_c1, [t.shape[1] for t in v_c1], dim=1)))
        if want_cycle:
            wc = parts["cycle"]
            out["warp_cycle"] = (F.fold(wc, (H, W), down, stride=down) if cfg.warp_patch
                                 else wc.reshape(B, -1, fh, fw))
        if want_two:
            out["warp_i2r"] = parts["i2r"].reshape(B, -1, fh, fw)

        # ---- R2: second trip through the row softmax  (:344, :369) --------------------------------
        v_r2 = [parts[n] for n in ("mask", "i2r") if n in parts]
        if v_r2:
            o_r2 = attn.rows(torch.cat(v_r2, dim=1) if len(v_r2) > 1 else v_r2[0])
            pos = 0
            if cycle_mask:
                nm = parts["mask"].shape[1]
                out["warp_mask"] = o_r2[:, :nm].reshape(B, nm, fh, fw)
                pos = nm
            if want_two:
                out["warp_i2r2i"] = o_r2[:, pos:].reshape(B, -1, fh, fw)
    return out
