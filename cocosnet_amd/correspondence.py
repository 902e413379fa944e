"""Drop-in `NoVGGCorrespondence` — the reference's correspondence network with its hot path on HIP.

Boundary contract (reference models/networks/correspondence.py:148-374):
  * constructor `NoVGGCorrespondence(opt)` with the same side effects on `opt` (sets/deletes
    `opt.spade_ic`, sets `opt.down`; :154-166), same sub-module and parameter names, so
    `state_dict()` round-trips with the reference's `*_net_Corr.pth` checkpoints;
  * `forward(ref_img, real_img, seg_map, ref_seg_map, temperature=0.01, detach_flag=False,
    WTA_scale_weight=1, alpha=1, return_corr=False)` returning the same `coor_out` dict (or the
    scaled correlation tensor when `return_corr`), fresh autograd-tracked fp32 NCHW tensors;
  * `print_network()` / `init_weights(init_type, gain)` as `BaseNetwork` provides them
    (base_network.py:18-59), because `networks.create_network` calls both.

Everything up to `self.theta(...)` / `self.phi(...)` is stock PyTorch (cocosnet_amd.producers);
from there on (`correspondence.py:272-372`) it is `cocosnet_amd.hot_path` = HIP kernels.
To plug into the unmodified reference, see INTEGRATION.md (`install_into_reference()`).
"""
from __future__ import annotations

import argparse

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from . import ops
from .hot_path import HotPathConfig, correspondence_hot_path
from .producers import AdaptiveFeatureGenerator, ResidualBlock

_EPS = __import__("sys").float_info.epsilon
#: the four shared ResidualBlocks see content and exemplar features as ONE batch (False: two calls, as the reference writes it —
#: a test / A-B hook, no environment variable)
BATCH_SHARED_LAYERS = True


def feature_normalize(x):
    """util.feature_normalize (util/util.py:31-34): x / (||x||_2 over channels + epsilon)."""
    if x.is_cuda and x.dtype == torch.float32:      # K1 without the centring: one HBM pass instead of four
        return ops.feature_normalize(x, _EPS)
    return x / (torch.norm(x, 2, 1, keepdim=True) + _EPS)   # CPU / fp64: producer parity tests only (the fp64 arbiter of tests/)


class NetworkBase(nn.Module):
    """The two `BaseNetwork` services `create_network` relies on (base_network.py:18-59)."""

    def print_network(self):
        n = sum(p.numel() for p in self.parameters())
        print("Network [%s] was created. Total number of parameters: %.1f million. "
              "To see the architecture, do print(network)." % (type(self).__name__, n / 1e6))

    def init_weights(self, init_type="normal", gain=0.02):
        def init_one(m):
            cname = m.__class__.__name__
            if "BatchNorm2d" in cname:
                if getattr(m, "weight", None) is not None:
                    init.normal_(m.weight.data, 1.0, gain)
                if getattr(m, "bias", None) is not None:
                    init.constant_(m.bias.data, 0.0)
            elif hasattr(m, "weight") and ("Conv" in cname or "Linear" in cname):
                # NB (faithful to the reference): on spectral-normed layers `m.weight` is the
                # derived tensor, so their `weight_orig` keeps PyTorch's default initialisation
                w = m.weight.data
                if init_type == "normal":
                    init.normal_(w, 0.0, gain)
                elif init_type == "xavier":
                    init.xavier_normal_(w, gain=gain)
                elif init_type == "xavier_uniform":
                    init.xavier_uniform_(w, gain=1.0)
                elif init_type == "kaiming":
                    init.kaiming_normal_(w, a=0, mode="fan_in")
                elif init_type == "orthogonal":
                    init.orthogonal_(w, gain=gain)
                elif init_type == "none":
                    m.reset_parameters()
                else:
                    raise NotImplementedError(f"initialization method [{init_type}] is not implemented")
                if getattr(m, "bias", None) is not None:
                    init.constant_(m.bias.data, 0.0)
        self.apply(init_one)
        for child in self.children():
            if hasattr(child, "init_weights"):
                child.init_weights(init_type, gain)


class NoVGGCorrespondence(NetworkBase):
    def __init__(self, opt):
        self.opt = opt
        super().__init__()
        # the adaptors read opt.spade_ic at construction (:154-158)
        opt.spade_ic = opt.semantic_nc
        self.adaptive_model_seg = AdaptiveFeatureGenerator(opt)
        opt.spade_ic = 3
        self.adaptive_model_img = AdaptiveFeatureGenerator(opt)
        del opt.spade_ic
        if opt.weight_domainC > 0 and (not opt.domain_rela):
            raise NotImplementedError("DomainClassifier (--weight_domainC > 0) is outside the MI355X "
                                      "hot-path scope (SURVEY.md §2 row 3)")
        if "down" not in opt:
            opt.down = 4
        if opt.warp_stride == 2:
            opt.down = 2
        assert opt.down in (2, 4)
        self.down = opt.down
        self.feature_channel = 64
        self.in_channels = self.feature_channel * 4
        self.inter_channels = 256

        cl = self.in_channels + (opt.semantic_nc if opt.maskmix else 0) + (3 if opt.use_coordconv else 0)
        self.layer = nn.Sequential(*[ResidualBlock(cl, cl, kernel_size=3, padding=1, stride=1)
                                     for _ in range(4)])
        self.phi = nn.Conv2d(cl, self.inter_channels, kernel_size=1, stride=1, padding=0)
        self.theta = nn.Conv2d(cl, self.inter_channels, kernel_size=1, stride=1, padding=0)
        # parameter-free; kept as attributes because the reference has them (:184-188)
        self.upsampling_bi = nn.Upsample(scale_factor=opt.down, mode="bilinear")
        self.upsampling = (nn.Upsample(scale_factor=opt.down, mode="bilinear") if opt.warp_bilinear
                           else nn.Upsample(scale_factor=opt.down))
        self.zero_tensor = None

    @staticmethod
    def addcoords(x):
        """CoordConv channels x, y in [-1, 1] and their radius (:202-220)."""
        B, _, h, w = x.shape
        xs = torch.arange(w, dtype=x.dtype, device=x.device) / (w - 1) * 2 - 1
        ys = torch.arange(h, dtype=x.dtype, device=x.device) / (h - 1) * 2 - 1
        xx = xs.view(1, 1, 1, w).expand(B, 1, h, w)
        yy = ys.view(1, 1, h, 1).expand(B, 1, h, w)
        return torch.cat((x, xx, yy, torch.sqrt(xx ** 2 + yy ** 2)), dim=1)

    def project(self, ref_img, real_img, seg_map, ref_seg_map, coor_out=None, lazy=False):
        """Everything BEFORE the hot path (:239-272, :282): returns (theta_raw, phi_raw)."""
        opt = self.opt
        if opt.mask_noise:
            noise = torch.randn_like(seg_map, requires_grad=False) * 0.1
            noise[seg_map == 0] = 0
            seg_input = seg_map + noise
        else:
            seg_input = seg_map
        feat_seg = feature_normalize(self.adaptive_model_seg(seg_input, seg_input))
        feat_img = feature_normalize(self.adaptive_model_img(ref_img, ref_img))
        if opt.isTrain and opt.novgg_featpair > 0 and coor_out is not None:
            pair = feature_normalize(self.adaptive_model_img(real_img, real_img))
            coor_out["loss_novgg_featpair"] = F.l1_loss(feat_seg, pair) * opt.novgg_featpair
        if opt.use_coordconv:
            feat_seg, feat_img = self.addcoords(feat_seg), self.addcoords(feat_img)
        seg = F.interpolate(seg_map, size=feat_seg.shape[2:], mode="nearest")
        ref_seg = F.interpolate(ref_seg_map, size=feat_img.shape[2:], mode="nearest")
        if opt.maskmix:
            cont_in = torch.cat((feat_seg, seg), 1)
            if opt.noise_for_mask and ((not opt.isTrain) or (opt.isTrain and opt.epoch > opt.mask_epoch)):
                noise = torch.randn_like(ref_seg, requires_grad=False) * 0.01
                ref_in = torch.cat((feat_img, noise), 1)
            else:
                ref_in = torch.cat((feat_img, ref_seg), 1)
        else:
            cont_in, ref_in = feat_seg, feat_img
        if BATCH_SHARED_LAYERS and cont_in.is_cuda and cont_in.shape == ref_in.shape:
            # `self.layer` (four ResidualBlocks, shared weights) is applied to both streams (:258-266): every op in it is
            # per-sample (convolution, InstanceNorm2d, PReLU), so ONE pass over the concatenated batch gives the same values
            # with half the launches — one weight-plane preparation, one weight gradient and one split-K reduction per
            # convolution instead of two
            both = self.layer(torch.cat((cont_in, ref_in), 0))
            cont, ref = both[:cont_in.shape[0]], both[cont_in.shape[0]:]
        else:
            cont, ref = self.layer(cont_in), self.layer(ref_in)
        if cont.is_cuda and cont.dtype == torch.float32:   # :272 / :282 on K0 (same parameters: checkpoints are unaffected)
            if lazy:      # forward(): the hot path decides — K23 (projection + K1 fused, no fp32 theta / phi) or K0 (ops.LazyProj1x1)
                return (ops.LazyProj1x1(cont, self.theta.weight, self.theta.bias),
                        ops.LazyProj1x1(ref, self.phi.weight, self.phi.bias))
            return (ops.proj1x1(cont, self.theta.weight, self.theta.bias),
                    ops.proj1x1(ref, self.phi.weight, self.phi.bias))
        return self.theta(cont), self.phi(ref)   # CPU / fp64: producer parity tests only; the hot path needs a GPU and fp32

    def forward(self, ref_img, real_img, seg_map, ref_seg_map, temperature=0.01, detach_flag=False,
                WTA_scale_weight=1, alpha=1, return_corr=False):
        coor_out = {}
        theta_raw, phi_raw = self.project(ref_img, real_img, seg_map, ref_seg_map, coor_out, lazy=True)
        cfg = HotPathConfig.from_opt(self.opt, down=self.opt.down)
        res = correspondence_hot_path(theta_raw, phi_raw, ref_img, real_img, seg_map, ref_seg_map, cfg,
                                      temperature=temperature, detach_flag=detach_flag,
                                      WTA_scale_weight=WTA_scale_weight, return_corr=return_corr)
        if return_corr:
            return res
        coor_out.update(res)
        return coor_out


def base_options(**overrides) -> argparse.Namespace:
    """`opt` with the defaults of options/base_options.py for every field the network reads."""
    opt = argparse.Namespace(
        semantic_nc=151, ngf=64, norm_E="spectralinstance", norm_G="spectralspadesyncbatch3x3",
        eqlr_sn=False, apex=False, PONO=False, PONO_C=False, adaptor_kernel=3, adaptor_se=False,
        adaptor_nonlocal=False, adaptor_res_deeper=False, dilation_conv=False, warp_stride=4,
        weight_domainC=0.0, domain_rela=False, use_coordconv=False, maskmix=False, warp_bilinear=False,
        mask_noise=False, noise_for_mask=False, isTrain=False, epoch=0, mask_epoch=-1,
        novgg_featpair=0.0, match_kernel=3, warp_patch=False, show_corr=False,
        warp_mask_losstype="none", show_warpmask=False, warp_cycle_w=0.0, two_cycle=False,
        CBN_intype="warp_mask", use_attention=False, init_type="xavier", init_variance=0.02,
        crop_size=256, aspect_ratio=1.0, gpu_ids=[0])
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt


def ade20k_options(**overrides) -> argparse.Namespace:
    """README.md:45 flag set: --use_attention --maskmix --warp_mask_losstype direct --PONO --PONO_C
    (label_nc 150 + dontcare -> semantic_nc 151)."""
    o = dict(semantic_nc=151, use_attention=True, maskmix=True, warp_mask_losstype="direct", PONO=True,
             PONO_C=True)
    o.update(overrides)
    return base_options(**o)


def celebahq_edge_options(**overrides) -> argparse.Namespace:
    """README.md:62,106: --use_attention --maskmix --PONO --PONO_C --warp_bilinear --adaptor_kernel 4
    (+ training: --warp_cycle_w 1); label_nc 15, no dontcare."""
    o = dict(semantic_nc=15, use_attention=True, maskmix=True, PONO=True, PONO_C=True, warp_bilinear=True,
             adaptor_kernel=4)
    o.update(overrides)
    return base_options(**o)


def deepfashion_options(**overrides) -> argparse.Namespace:
    """README.md:69,115: --use_attention --PONO --PONO_C --warp_bilinear --no_flip --warp_patch
    --video_like --adaptor_kernel 4; label_nc 20."""
    o = dict(semantic_nc=20, use_attention=True, PONO=True, PONO_C=True, warp_bilinear=True, warp_patch=True,
             adaptor_kernel=4)
    o.update(overrides)
    return base_options(**o)


def install_into_reference(networks_module):
    """Make the unmodified reference build THIS network: `networks.define_Corr(opt)` looks up a
    class named `novggcorrespondence` in `models.networks.correspondence` and asserts it subclasses
    the reference's `BaseNetwork` (networks/__init__.py:18-26,76-78).  Returns the injected class."""
    import importlib
    ref_corr = importlib.import_module(networks_module.__name__ + ".correspondence")
    base = networks_module.BaseNetwork
    cls = type("NoVGGCorrespondence", (NoVGGCorrespondence, base), {"__doc__": NoVGGCorrespondence.__doc__})
    ref_corr.NoVGGCorrespondence = cls
    return cls
