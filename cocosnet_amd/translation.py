"""The translation generator and the PatchGAN discriminator of BASELINE config 3 on the HIP kernels of this package.

SURVEY.md §8(f) rows 1 and 4, BASELINE config 3 ("CelebA-HQ edge->face 256x256 batch 16, + HIP SPADE generator / PatchGAN conv
kernels"): `SPADEGenerator` (reference models/networks/generator.py:17-89: fc conv, seven `SPADEResnetBlock`s with nearest x2
up-sampling between them, the non-local `Attention` block at 128x128, `conv_img` + tanh) and `MultiscaleDiscriminator` /
`NLayerDiscriminator` (discriminator.py:14-177: k4 convolutions, spectral norm + InstanceNorm, LeakyReLU(0.2), the Attention
block in front of `model3`), built from the fused pieces of producers.py — K16 convolutions (reflect padding fused), K9
PONO-SPADE + LeakyReLU, K13 InstanceNorm + LeakyReLU, K21 spectral norm, K11 nearest up-sampling, the fused attention — with
the reference's sub-module and parameter names, so `*_net_G.pth` / `*_net_D.pth` checkpoints load unchanged
(tests/test_module_cpu.py compares the state_dict keys and shapes with the reference's own classes).

These two networks are NOT on the correspondence hot path: a reference user keeps the reference's classes and switches their
layers over with `spade.install_spade_into_reference` + `producers.use_hip_convs`; this module exists so that
config 3 "as written" can be built, measured (bench.py `config.context.config3`) and held to torch-fp64 on a GPU box, where
/root/reference is not present.  Supported flag space = the README commands (no --eqlr_sn, D_cam = 0).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .correspondence import NetworkBase
from .producers import Attention, Conv2d, SPADEResnetBlock, hip_spectral_norm, nonspade_norm_layer


def _up2(x):
    """nn.Upsample(scale_factor=2) (nearest): K11 for fp32 CUDA tensors."""
    if x.is_cuda and x.dtype == torch.float32 and (x.shape[3] * 2) % 4 == 0:
        return ops.upsample_nearest(x, 2)
    return F.interpolate(x, scale_factor=2, mode="nearest")


class SPADEGenerator(NetworkBase):
    """generator.py:17-89.  `forward(input, warp_out=None)`: `warp_out` (= CBN_in of pix2pix_model.py:313-320: the warped
    exemplar, optionally concatenated with the label map) is the conditioning of every SPADE layer."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        if getattr(opt, "eqlr_sn", False):
            raise NotImplementedError("--eqlr_sn (EqualLR) is not supported by the MI355X drop-in")
        nf = opt.ngf
        self.sw = opt.crop_size // 32                       # five x2 up-samplings (generator.py:52-58)
        self.sh = round(self.sw / opt.aspect_ratio)
        ic = (3 if "warp" in opt.CBN_intype else 0) + (opt.semantic_nc if "mask" in opt.CBN_intype else 0)
        self.fc = Conv2d(ic, 16 * nf, 3, padding=1)
        self.head_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_1 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.up_0 = SPADEResnetBlock(16 * nf, 8 * nf, opt)
        self.up_1 = SPADEResnetBlock(8 * nf, 4 * nf, opt)
        if opt.use_attention:
            self.attn = Attention(4 * nf, "spectral" in opt.norm_G)
        self.up_2 = SPADEResnetBlock(4 * nf, 2 * nf, opt)
        self.up_3 = SPADEResnetBlock(2 * nf, 1 * nf, opt)
        self.conv_img = Conv2d(nf, 3, 3, padding=1)
        self.up = nn.Upsample(scale_factor=2)               # parameter-free; kept because the reference has it

    @staticmethod
    def _final_act(x):
        return F.leaky_relu(x, 2e-1)          # generator.py:99

    def forward(self, input, warp_out=None):
        seg = input if warp_out is None else warp_out
        x = self.fc(F.interpolate(seg, size=(self.sh, self.sw)))
        x = self.head_0(x, seg)
        x = self.G_middle_0(_up2(x), seg)
        x = self.G_middle_1(x, seg)
        x = self.up_0(_up2(x), seg)
        x = self.up_1(_up2(x), seg)
        x = _up2(x)
        if self.opt.use_attention:
            x = self.attn(x)
        x = self.up_2(x, seg)
        x = self.up_3(_up2(x), seg)
        return torch.tanh(self.conv_img(self._final_act(x)))


class _ConvNormAct(nn.Sequential):
    """[conv (+ spectral norm) (+ InstanceNorm2d)] + LeakyReLU(0.2) of the PatchGAN (discriminator.py:92-115) with the same child
    indices as the reference's `nn.Sequential` (0 = the wrapped convolution, 1 = the activation).  When child 0 is
    Sequential(conv, InstanceNorm2d(affine=False)) and the tensor is fp32 CUDA, the norm and the activation are ONE pass (K13)."""

    def forward(self, x):
        first = self[0]
        if (isinstance(first, nn.Sequential) and len(first) == 2 and type(first[1]) is nn.InstanceNorm2d and not first[1].affine
                and x.is_cuda and x.dtype == torch.float32 and ops.CONV_PRECISION != "torch" and len(self) == 2):
            y = first[0](x)
            if y.shape[2] * y.shape[3] <= 16384:
                slope = getattr(self, "_slope", None)
                if slope is None or slope.device != y.device:
                    slope = self._slope = torch.full((1,), float(self[1].negative_slope), device=y.device)
                return ops.instnorm_prelu(y, None, slope, first[1].eps)
            y = first[1](y)
            return self[1](y)
        return super().forward(x)


class NLayerDiscriminator(NetworkBase):
    """discriminator.py:73-177 (D_cam = 0): model0..model{n_layers_D}, `attn` in front of model3, and the (unused, but
    checkpointed) `dec` branch.  Returns (list of intermediate outputs | last output, None) like the reference."""

    def __init__(self, opt, stage1: bool = False):
        super().__init__()
        self.opt = opt
        self.stage1 = stage1
        if getattr(opt, "D_cam", 0) > 0:
            raise NotImplementedError("--D_cam > 0 (CAM logits) is not supported by the MI355X drop-in")
        kw, padw = 4, 1
        nf = opt.ndf
        input_nc = opt.label_nc + opt.output_nc + (1 if opt.contain_dontcare_label else 0)
        norm_layer = nonspade_norm_layer(opt, opt.norm_D)
        use_attn = ((not stage1) and opt.use_attention) or (stage1 and getattr(opt, "use_attention_st1", False))
        sequence = [[Conv2d(input_nc, nf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, False)]]
        for n in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            if use_attn and n == opt.n_layers_D - 1:
                self.attn = Attention(nf_prev, "spectral" in opt.norm_D)
            if n == opt.n_layers_D - 1 and not stage1:
                dec, nc_dec = [], nf_prev
                for _ in range(opt.n_layers_D - 1):
                    dec += [nn.Upsample(scale_factor=2), norm_layer(Conv2d(nc_dec, nc_dec // 2, kernel_size=3, stride=1, padding=1)),
                            nn.LeakyReLU(0.2, False)]
                    nc_dec //= 2
                dec += [Conv2d(nc_dec, opt.semantic_nc, kernel_size=3, stride=1, padding=1)]
                self.dec = nn.Sequential(*dec)
            sequence += [[norm_layer(Conv2d(nf_prev, nf, kernel_size=kw, stride=stride, padding=padw)), nn.LeakyReLU(0.2, False)]]
        sequence += [[Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)]]
        for n, layers in enumerate(sequence):
            self.add_module("model" + str(n), _ConvNormAct(*layers) if len(layers) == 2 else nn.Sequential(*layers))
        self._use_attn = use_attn

    def forward(self, input):
        results = [input]
        for name, sub in self.named_children():
            if "model" not in name:
                continue
            x = results[-1]
            if name == "model3" and self._use_attn:
                x = self.attn(x)
            results.append(sub(x))
        return (results[1:] if not self.opt.no_ganFeat_loss else results[-1]), None


class MultiscaleDiscriminator(NetworkBase):
    """discriminator.py:14-69: `num_D` PatchGANs on an image pyramid (avg-pool k3 s2 between the scales)."""

    def __init__(self, opt, stage1: bool = False):
        super().__init__()
        self.opt = opt
        self.stage1 = stage1
        for i in range(opt.num_D):
            self.add_module("discriminator_%d" % i, NLayerDiscriminator(opt, stage1=stage1))

    def forward(self, input):
        result, cam_logits = [], []
        for _, D in self.named_children():
            out, cam = D(input)
            cam_logits.append(cam)
            result.append(out if not self.opt.no_ganFeat_loss else [out])
            input = F.avg_pool2d(input, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
        return result, [], cam_logits


def celebahq_edge_train_options(**overrides):
    """BASELINE config 3 as written (README.md:106): CelebA-HQ edge->face training flags, with the defaults of
    options/base_options.py / train_options.py for every field the generator and the discriminator read."""
    from .correspondence import celebahq_edge_options
    o = dict(isTrain=True, warp_cycle_w=1.0, ndf=64, norm_D="spectralinstance", n_layers_D=4, num_D=2, output_nc=3, label_nc=15,
             contain_dontcare_label=False, no_ganFeat_loss=False, D_cam=0.0, netD_subarch="n_layer", use_attention_st1=False,
             norm_G="spectralspadesyncbatch3x3", CBN_intype="warp_mask")
    o.update(overrides)
    return celebahq_edge_options(**o)
