"""Feature producers that feed the correspondence hot path.

These are the layers immediately BEFORE the path: the two domain adaptors
(`AdaptiveFeatureGenerator`, reference models/networks/generator.py:91-160, built from
`SPADEResnetBlock` architecture.py:19-95 and `SPADE` normalization.py:83-151) and the four
`ResidualBlock`s (correspondence.py:13-36).  They are re-implemented here only so that the drop-in
`NoVGGCorrespondence` is a standalone module whose `state_dict()` keys and shapes equal the
reference's (`*_net_Corr.pth` checkpoints load unchanged).  On a GPU in fp32 their convolutions run on K16
(conv_f16x3.hip: split-precision implicit GEMM) / K0 (1x1), the InstanceNorm+PReLU and PONO-SPADE arithmetic on K13 / K11.

Supported flag space = what the README commands use: spectral-norm convs (not --eqlr_sn), no apex,
PONO or instance/batch/sync-batch parameter-free norms.  Unsupported flags raise immediately.
"""
from __future__ import annotations

import os
import re

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm



_HIP_BACKENDS = ("f16x3", "bf16")


def conv_backend() -> str:
    """The ONE switch of the convolution flavour, `cocosnet_amd.ops.CONV_PRECISION` (env COCOS_CONV), read at call time:
    "f16x3" / "bf16": K16 in that arithmetic (bf16 is BASELINE config 3's precision for the generator / discriminator
    stacks — inside netCorr it is a timing reference only: everything there is upstream of the correlation); "torch": the
    framework's convolution, spectral norm and padding (A/B measurements).  (Round 3 kept a second, import-time copy
    here; setting one of the two mixed flavours — ADVICE r3.)"""
    from . import ops
    return ops.CONV_PRECISION


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose fp32 GPU forward/backward run on the HIP kernels: K16 (ops.conv2d) for kxk, K0 (ops.proj1x1) for
    1x1.  Same parameters, same state_dict keys, works under torch.nn.utils.spectral_norm (which only rewrites
    `.weight`).  Anything the kernels do not cover (groups, non-zero padding modes, rectangular stride / padding /
    dilation, other dtypes, CPU) takes the framework path."""

    def forward(self, input, reflect: int = 0):
        """`reflect` = r: this call is conv(ReflectionPad2d(r)(input)) — passed explicitly by reflect_conv() THROUGH the
        module call (forward pre-hooks such as spectral norm see only the positional input and still run)."""
        return self._conv_forward(input, self.weight, self.bias, int(reflect))

    def _conv_forward(self, input, weight, bias, reflect: int = 0):
        s, p, k, d = self.stride, self.padding, self.kernel_size, self.dilation
        hip = conv_backend() in _HIP_BACKENDS
        if reflect and not (hip and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4):
            input, reflect = nn.functional.pad(input, (reflect,) * 4, mode="reflect"), 0
        if (hip and input.is_cuda and input.dtype == torch.float32 and weight.dtype == torch.float32
                and input.dim() == 4 and self.groups == 1 and self.padding_mode == "zeros"
                and not isinstance(p, str) and s[0] == s[1] and p[0] == p[1] and d[0] == d[1]):
            from . import _lib, ops
            try:
                if k == (1, 1) and s[0] == 1 and p[0] == 0 and not reflect:
                    return ops.proj1x1(input, weight, bias)
                return ops.conv2d(input, weight, bias, s[0], p[0], d[0], reflect)
            except _lib.CocosHipError as e:
                # kernel-side limits (a tensor of 2 GiB or more, a grid too large for the 32-bit index arithmetic: the
                # discriminator / VGG at very large batch or resolution) are not part of the check above: they take the
                # framework path instead of failing the training step.  Anything else is a real error.
                if getattr(e, "code", 0) != -2:          # COCOS_ERR_UNSUPPORTED (include/cocos_hip.h)
                    raise
        if reflect:
            input = nn.functional.pad(input, (reflect,) * 4, mode="reflect")
        return super()._conv_forward(input, weight, bias)


#: False: spectral-normed layers keep the framework's compute_weight (test / A-B hook, no environment variable)
SPECTRAL_HIP = True

import importlib                                            # noqa: E402
_sn_mod = importlib.import_module("torch.nn.utils.spectral_norm")    # the MODULE (torch.nn.utils re-exports its function under the same name)


class _SpectralNormHIP(_sn_mod.SpectralNorm):
    """torch.nn.utils.spectral_norm's hook with `compute_weight` on K21 (ops.spectral_weight) for fp32 GPU weights: same buffers
    (weight_orig / weight_u / weight_v: checkpoints unchanged), same in-place update of u and v, same gradient — four launches
    forward and two backward instead of ~20 framework launches per layer and step."""

    def compute_weight(self, module, do_power_iteration):
        weight = getattr(module, self.name + "_orig")
        if (SPECTRAL_HIP and conv_backend() in _HIP_BACKENDS and weight.is_cuda and weight.dtype == torch.float32 and self.dim == 0
                and self.n_power_iterations == 1 and weight.is_contiguous() and weight.numel() // weight.shape[0] <= 16384):
            from . import ops
            return ops.spectral_weight(weight, getattr(module, self.name + "_u"), getattr(module, self.name + "_v"),
                                       do_power_iteration, self.eps)
        return super().compute_weight(module, do_power_iteration)


def hip_spectral_norm(module: nn.Module, name: str = "weight") -> nn.Module:
    """torch.nn.utils.spectral_norm(module), with the hook's weight computation on K21."""
    module = spectral_norm(module, name)
    for hook in module._forward_pre_hooks.values():
        if type(hook) is _sn_mod.SpectralNorm and hook.name == name:
            hook.__class__ = _SpectralNormHIP
    return module


def reflect_conv(pad: nn.Module, conv: nn.Module, x):
    """conv(pad(x)) for a ReflectionPad2d directly in front of one of this module's Conv2d layers.  The convolution is still
    CALLED as a module (spectral-norm and other forward hooks run); the padding is an explicit keyword of its forward, so
    that the bf16 / NHWC path can write the mirrored border while it prepares its operand (ops.conv2d(reflect=r)) instead
    of materialising the padded fp32 tensor first."""
    p = pad.padding
    if isinstance(conv, Conv2d) and len(set(p)) == 1 and p[0] > 0 and torch.is_tensor(x) and x.dim() == 4 and p[0] < min(x.shape[2:]):
        return conv(x, reflect=int(p[0]))
    return conv(pad(x))


class ReflectionPad2d(nn.ReflectionPad2d):
    """nn.ReflectionPad2d whose fp32 GPU forward / backward are K18 (one padding for all four sides); anything else takes
    the framework path.  No parameters: state_dict keys are untouched."""

    def forward(self, input):
        p = self.padding
        if (conv_backend() in _HIP_BACKENDS and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
                and len(set(p)) == 1 and 0 <= p[0] < min(input.shape[2:])):
            from . import ops
            return ops.reflect_pad2d(input, p[0])
        return super().forward(input)


def use_hip_convs(module: nn.Module) -> int:
    """Re-class every plain `nn.Conv2d` inside `module` (e.g. the reference's `NLayerDiscriminator`
    discriminator.py:92-115, its `SPADEGenerator`, `VGG19`) to `Conv2d` above, in place: parameters, buffers, hooks
    (spectral norm) and state_dict keys are untouched, only `_conv_forward` changes.  Returns the number of layers
    switched.  The inverse is `use_framework_convs`."""
    n = 0
    for m in module.modules():
        if type(m) is nn.Conv2d:
            m.__class__ = Conv2d
            n += 1
        elif type(m) is nn.ReflectionPad2d:     # rides along (not counted): K18
            m.__class__ = ReflectionPad2d
    return n


def use_framework_convs(module: nn.Module) -> int:
    n = 0
    for m in module.modules():
        if type(m) is Conv2d:
            m.__class__ = nn.Conv2d
            n += 1
        elif type(m) is ReflectionPad2d:
            m.__class__ = nn.ReflectionPad2d
    return n


def positional_norm(x, eps=1e-5):
    """PONO: normalise over the channel axis with the UNBIASED variance (normalization.py:63-68)."""
    mu = x.mean(dim=1, keepdim=True)
    sd = (x.var(dim=1, keepdim=True) + eps).sqrt()
    return (x - mu) / sd


def _param_free_norm(kind: str, channels: int):
    if kind == "instance":
        return nn.InstanceNorm2d(channels, affine=False)
    if kind == "batch":
        return nn.BatchNorm2d(channels, affine=False)
    if kind == "syncbatch":
        # the reference's Python-queue SynchronizedBatchNorm2d -> statistics all-reduced over the process group
        # (RCCL over xGMI under backend "nccl"; same buffer names, so checkpoints load) — cocosnet_amd/dist.py
        from .dist import SyncBatchNorm2d
        return SyncBatchNorm2d(channels, affine=False)
    raise ValueError(f"{kind} is not a recognised parameter-free norm for SPADE")


class SPADE(nn.Module):
    """norm(x) * (1 + gamma(seg)) + beta(seg); gamma/beta from a 2-layer conv net on the nearest-
    resized label map, reflection padded (normalization.py:83-151)."""

    def __init__(self, config_text: str, norm_nc: int, label_nc: int, pono: bool):
        super().__init__()
        m = re.search(r"spade(\D+)(\d)x\d", config_text)
        if m is None:
            raise ValueError(f"bad SPADE config {config_text!r}")
        ks = int(m.group(2))
        self.pono = bool(pono)
        if not self.pono:
            self.param_free_norm = _param_free_norm(m.group(1), norm_nc)
        hidden = 128
        self.mlp_shared = nn.Sequential(ReflectionPad2d(ks // 2),
                                        Conv2d(label_nc, hidden, ks), nn.ReLU())
        self.pad = ReflectionPad2d(ks // 2)
        self.mlp_gamma = Conv2d(hidden, norm_nc, ks)
        self.mlp_beta = Conv2d(hidden, norm_nc, ks)

    def forward(self, x, segmap, similarity_map=None, slope: float = 1.0):
        """`slope` != 1 folds the LeakyReLU that SPADEResnetBlock applies right after (architecture.py:88-95).
        Shared with the hook for the reference's own SPADE (cocosnet_amd/spade.py)."""
        from .spade import spade_forward
        return spade_forward(self, x, segmap, similarity_map, slope)


class SELayer(nn.Module):
    """Squeeze-and-excitation (architecture.py:183-198), only with --adaptor_se."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())

    def forward(self, x):
        w = self.fc(self.avg_pool(x).flatten(1))
        return x * w[:, :, None, None]


class Attention(nn.Module):
    """Non-local block (architecture.py:97-127), only with --adaptor_nonlocal."""

    def __init__(self, ch, use_sn):
        super().__init__()
        self.ch = ch
        wrap = spectral_norm if use_sn else (lambda m: m)
        self.theta = wrap(Conv2d(ch, ch // 8, 1, bias=False))
        self.phi = wrap(Conv2d(ch, ch // 8, 1, bias=False))
        self.g = wrap(Conv2d(ch, ch // 2, 1, bias=False))
        self.o = wrap(Conv2d(ch // 2, ch, 1, bias=False))
        self.gamma = nn.Parameter(torch.tensor(0.0), requires_grad=True)

    @staticmethod
    def _pool(t):
        return F.max_pool2d(t, [2, 2])

    def forward(self, x, y=None):
        B, _, H, W = x.shape
        theta = self.theta(x).view(B, self.ch // 8, H * W)
        phi = self._pool(self.phi(x)).view(B, self.ch // 8, H * W // 4)
        g = self._pool(self.g(x)).view(B, self.ch // 2, H * W // 4)
        if x.is_cuda and x.dtype == torch.float32:
            from . import ops            # QK^T -> softmax -> PV on the HIP kernels of the path (§8f rank 3)
            att = ops.softmax_attention(theta.contiguous(), phi.contiguous(), g.contiguous(), 1.0)
        else:
            beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
            att = torch.bmm(g, beta.transpose(1, 2))
        o = self.o(att.view(B, self.ch // 2, H, W))
        return self.gamma * o + x


def _spade_label_nc(opt):
    if "spade_ic" in opt:
        return opt.spade_ic
    t = opt.CBN_intype
    return (3 if "warp" in t else 0) + (opt.semantic_nc if "mask" in t else 0)


class SPADEResnetBlock(nn.Module):
    """Pre-activation residual block with SPADE norms and reflection-padded 3x3 convs
    (architecture.py:19-95)."""

    def __init__(self, fin, fout, opt, use_se=False, dilation=1):
        super().__init__()
        self.learned_shortcut = fin != fout
        fmid = min(fin, fout)
        self.use_se = use_se
        self.slope = 0.2                 # `actvn`: LeakyReLU(2e-1) (architecture.py:107-108); a plain attribute (tests smooth it)
        self.pad = ReflectionPad2d(dilation)
        self.conv_0 = Conv2d(fin, fmid, 3, padding=0, dilation=dilation)
        self.conv_1 = Conv2d(fmid, fout, 3, padding=0, dilation=dilation)
        if self.learned_shortcut:
            self.conv_s = Conv2d(fin, fout, 1, bias=False)
        if "spectral" in opt.norm_G:
            self.conv_0 = hip_spectral_norm(self.conv_0)
            self.conv_1 = hip_spectral_norm(self.conv_1)
            if self.learned_shortcut:
                self.conv_s = hip_spectral_norm(self.conv_s)
        cfg = opt.norm_G.replace("spectral", "")
        ic = _spade_label_nc(opt)
        self.norm_0 = SPADE(cfg, fin, ic, opt.PONO)
        self.norm_1 = SPADE(cfg, fmid, ic, opt.PONO)
        if self.learned_shortcut:
            self.norm_s = SPADE(cfg, fin, ic, opt.PONO)
        if use_se:
            self.se_layar = SELayer(fout)   # (sic) the reference's attribute name, kept for checkpoints

    def _norm_act(self, norm, x, seg):
        return norm(x, seg, slope=self.slope)

    def forward(self, x, seg):
        # ONE nearest resize of the label map for the block's two or three SPADEs (the reference resizes inside each,
        # normalization.py:133: same values; spade.spade_forward skips it when the grids already agree)
        if seg.shape[2:] != x.shape[2:]:
            seg = nn.functional.interpolate(seg, size=x.size()[2:], mode="nearest")
        x_s = self.conv_s(self.norm_s(x, seg)) if self.learned_shortcut else x
        dx = reflect_conv(self.pad, self.conv_0, self._norm_act(self.norm_0, x, seg))         # norm -> LeakyReLU(0.2) -> pad -> conv
        dx = reflect_conv(self.pad, self.conv_1, self._norm_act(self.norm_1, dx, seg))
        if self.use_se:
            dx = self.se_layar(dx)
        return x_s + dx


def nonspade_norm_layer(opt, norm_type):
    """`get_nonspade_norm_layer` (normalization.py:21-61): spectral norm on the conv, then an
    optional instance / batch / sync-batch norm (which also drops the now-useless conv bias)."""
    if getattr(opt, "eqlr_sn", False):
        raise NotImplementedError("--eqlr_sn (EqualLR) is not supported by the MI355X drop-in")
    if getattr(opt, "apex", False):
        raise NotImplementedError("--apex is not supported (no apex on ROCm here)")

    def wrap(layer):
        sub = norm_type
        if norm_type.startswith("spectral"):
            layer = hip_spectral_norm(layer)
            sub = norm_type[len("spectral"):]
        if sub in ("none", ""):
            return layer
        if getattr(layer, "bias", None) is not None:
            delattr(layer, "bias")
            layer.register_parameter("bias", None)
        ch = layer.out_channels
        if sub == "batch":
            norm = nn.BatchNorm2d(ch, affine=True)
        elif sub == "sync_batch":
            from .dist import SyncBatchNorm2d
            norm = SyncBatchNorm2d(ch, affine=True)
        elif sub == "instance":
            norm = nn.InstanceNorm2d(ch, affine=False)
        else:
            raise ValueError(f"normalization layer {sub} is not recognized")
        return nn.Sequential(layer, norm)
    return wrap


class AdaptiveFeatureGenerator(nn.Module):
    """Domain adaptor: 5 strided convs + 3 SPADE residual blocks -> [B, 4*ngf, H/down, W/down]
    (generator.py:91-160).  Reads opt.spade_ic, which NoVGGCorrespondence sets around construction."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nf = opt.ngf
        wrap = nonspade_norm_layer(opt, opt.norm_E)
        ak = opt.adaptor_kernel
        self.layer1 = wrap(Conv2d(opt.spade_ic, nf, 3, stride=1, padding=1))
        self.layer2 = wrap(Conv2d(nf, nf * 2, ak, stride=2, padding=1))
        self.layer3 = wrap(Conv2d(nf * 2, nf * 4, 3, stride=1, padding=1))
        if opt.warp_stride == 2:
            self.layer4 = wrap(Conv2d(nf * 4, nf * 8, 3, stride=1, padding=1))
        else:
            self.layer4 = wrap(Conv2d(nf * 4, nf * 8, ak, stride=2, padding=1))
        self.layer5 = wrap(Conv2d(nf * 8, nf * 8, 3, stride=1, padding=1))
        self.actvn = nn.LeakyReLU(0.2, False)
        self._slopes = {}          # constant one-element "PReLU weights" per device (not parameters, not in the state_dict)
        self.head_0 = SPADEResnetBlock(8 * nf, 8 * nf, opt, use_se=opt.adaptor_se)
        if opt.adaptor_nonlocal:
            self.attn = Attention(8 * nf, False)
        self.G_middle_0 = SPADEResnetBlock(8 * nf, 8 * nf, opt, use_se=opt.adaptor_se)
        self.G_middle_1 = SPADEResnetBlock(8 * nf, 4 * nf, opt, use_se=opt.adaptor_se)
        if opt.adaptor_res_deeper:
            self.deeper0 = SPADEResnetBlock(4 * nf, 4 * nf, opt)
            if opt.dilation_conv:
                self.deeper1 = SPADEResnetBlock(4 * nf, 4 * nf, opt, dilation=2)
                self.deeper2 = SPADEResnetBlock(4 * nf, 4 * nf, opt, dilation=4)
                self.degridding0 = wrap(Conv2d(nf * 4, nf * 4, 3, stride=1, padding=2, dilation=2))
                self.degridding1 = wrap(Conv2d(nf * 4, nf * 4, 3, stride=1, padding=1))
            else:
                self.deeper1 = SPADEResnetBlock(4 * nf, 4 * nf, opt)
                self.deeper2 = SPADEResnetBlock(4 * nf, 4 * nf, opt)

    def _conv_norm_act(self, layer, x, slope: float):
        """`layer` = conv [+ InstanceNorm2d] (nonspade_norm_layer), followed by LeakyReLU(slope) (slope 1.0: none).  The parameter-free
        InstanceNorm and the activation are K13 — one HBM pass forward, one backward — instead of the framework's batch-norm
        kernels + leaky_relu (generator.py:133-138 calls them as `layerK(self.actvn(x))`: same values, other grouping)."""
        if (isinstance(layer, nn.Sequential) and len(layer) == 2 and type(layer[1]) is nn.InstanceNorm2d and not layer[1].affine
                and not layer[1].track_running_stats and x.is_cuda and x.dtype == torch.float32 and conv_backend() in _HIP_BACKENDS):
            from . import ops
            y = layer[0](x)
            if y.shape[2] * y.shape[3] > 16384:
                # K13 keeps a plane in registers up to 128 x 128; beyond that its streaming flavour (one workgroup per plane, five
                # passes backward) measured SLOWER than the framework's two-kernel batch norm (2.1 vs 1.2 ms per step): not used
                y = layer[1](y)
                return y if slope == 1.0 else nn.functional.leaky_relu(y, slope)
            key = (y.device, float(slope))
            w = self._slopes.get(key)
            if w is None:
                w = self._slopes[key] = torch.full((1,), float(slope), device=y.device, dtype=torch.float32)
            return ops.instnorm_prelu(y, None, w, layer[1].eps)
        y = layer(x)
        return y if slope == 1.0 else nn.functional.leaky_relu(y, slope)

    def forward(self, x, seg):
        slope = self.actvn.negative_slope
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = self._conv_norm_act(layer, x, slope)
        x = self._conv_norm_act(self.layer5, x, 1.0)
        x = self.head_0(x, seg)
        if self.opt.adaptor_nonlocal:
            x = self.attn(x)
        x = self.G_middle_1(self.G_middle_0(x, seg), seg)
        if self.opt.adaptor_res_deeper:
            x = self.deeper2(self.deeper1(self.deeper0(x, seg), seg), seg)
            if self.opt.dilation_conv:
                x = self.degridding1(self.degridding0(x))
        return x


class ResidualBlock(nn.Module):
    """reflect-pad -> 3x3 conv -> InstanceNorm -> PReLU, twice, with a skip and ONE shared PReLU
    (correspondence.py:13-36)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, stride=1):
        super().__init__()
        self.padding1 = ReflectionPad2d(padding)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size, padding=0, stride=stride)
        self.bn1 = nn.InstanceNorm2d(out_channels)
        self.prelu = nn.PReLU()
        self.padding2 = ReflectionPad2d(padding)
        self.conv2 = Conv2d(in_channels, out_channels, kernel_size, padding=0, stride=stride)
        self.bn2 = nn.InstanceNorm2d(out_channels)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32:
            from . import ops            # K13: InstanceNorm (+ skip) + PReLU in one HBM pass each
            y = ops.instnorm_prelu(reflect_conv(self.padding1, self.conv1, x), None, self.prelu.weight, self.bn1.eps)
            return ops.instnorm_prelu(reflect_conv(self.padding2, self.conv2, y), x, self.prelu.weight, self.bn2.eps)
        y = self.prelu(self.bn1(self.conv1(self.padding1(x))))
        y = self.bn2(self.conv2(self.padding2(y)))
        return self.prelu(y + x)
