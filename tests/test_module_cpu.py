"""CPU checks of the drop-in NoVGGCorrespondence module: checkpoint compatibility (state_dict names
and shapes equal the reference's), the feature producers against the reference's own code (only
where /root/reference exists, i.e. in the build container), and injection into the reference's
`networks.define_Corr`.  The hot path itself needs a GPU and is covered by test_gpu_parity.py."""
import argparse
import json
import os

import pytest
import torch

from cocosnet_amd import correspondence as cc
from oracle import ref_harness as rh

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "netcorr_state_dict_keys.json")
needs_ref = pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present")


def _opt(flags):
    return cc.base_options(**flags)


@pytest.mark.parametrize("name", ["ade20k", "celebahq_edge", "deepfashion", "no_pono_syncbn",
                                  "all_adaptor_options"])
def test_state_dict_names_and_shapes_equal_the_reference(name):
    rec = json.load(open(GOLDEN))[name]
    net = cc.NoVGGCorrespondence(_opt(rec["flags"]))
    mine = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert mine == rec["state_dict"]


def test_constructor_side_effects_on_opt():
    opt = _opt(dict(semantic_nc=5, maskmix=True, PONO=True, warp_stride=2))
    assert "down" not in opt
    cc.NoVGGCorrespondence(opt)
    assert opt.down == 2 and "spade_ic" not in opt
    opt = _opt(dict(semantic_nc=5, PONO=True))
    cc.NoVGGCorrespondence(opt)
    assert opt.down == 4


def test_unsupported_flags_fail_loudly():
    with pytest.raises(NotImplementedError):
        cc.NoVGGCorrespondence(_opt(dict(semantic_nc=5, PONO=True, eqlr_sn=True)))
    with pytest.raises(NotImplementedError):
        cc.NoVGGCorrespondence(_opt(dict(semantic_nc=5, PONO=True, weight_domainC=1.0)))


def test_forward_without_gpu_raises_instead_of_falling_back():
    from cocosnet_amd._lib import CocosHipError
    opt = _opt(dict(semantic_nc=4, maskmix=True, PONO=True, PONO_C=True, match_kernel=1, ngf=64))
    net = cc.NoVGGCorrespondence(opt).eval()
    x = torch.rand(1, 3, 32, 32) * 2 - 1
    seg = torch.zeros(1, 4, 32, 32); seg[:, 0] = 1
    with torch.no_grad(), pytest.raises(CocosHipError, match="no CPU fallback"):
        net(x, x, seg, seg)


def test_hot_path_config_from_opt():
    from cocosnet_amd.hot_path import HotPathConfig
    opt = cc.celebahq_edge_options(warp_cycle_w=1.0, two_cycle=True, isTrain=True)
    cfg = HotPathConfig.from_opt(opt, down=4)
    assert (cfg.match_kernel, cfg.PONO_C, cfg.warp_bilinear, cfg.warp_cycle_w, cfg.two_cycle) == \
        (3, True, True, 1.0, True)
    assert HotPathConfig.from_opt(argparse.Namespace()).down == 4


@needs_ref
@pytest.mark.parametrize("flags", [
    dict(semantic_nc=6, maskmix=True, PONO=True, PONO_C=True, match_kernel=1),
    dict(semantic_nc=4, maskmix=False, PONO=True, PONO_C=True, adaptor_kernel=4, warp_bilinear=True),
    dict(semantic_nc=5, maskmix=True, PONO=False, use_coordconv=True, adaptor_se=True,
         adaptor_nonlocal=True, adaptor_res_deeper=True, dilation_conv=True, warp_stride=2),
])
def test_producers_match_the_reference_code(flags):
    """Load the reference's weights into the drop-in and compare theta_raw / phi_raw (the inputs of
    the hot path) computed by both on CPU."""
    ref = rh.build_reference_corr(rh.make_opt(**flags), seed=3).eval()
    mine = cc.NoVGGCorrespondence(_opt(flags)).eval()
    mine.load_state_dict(ref.state_dict(), strict=True)
    g = torch.Generator().manual_seed(0)
    size = 32
    img = torch.rand(2, 3, size, size, generator=g) * 2 - 1
    lab = torch.randint(0, flags["semantic_nc"], (2, 1, size, size), generator=g)
    seg = torch.zeros(2, flags["semantic_nc"], size, size).scatter_(1, lab, 1.0)
    ref_seg = seg.flip(0)
    got = {}
    hooks = [ref.theta.register_forward_hook(lambda m, i, o: got.__setitem__("theta", o.detach())),
             ref.phi.register_forward_hook(lambda m, i, o: got.__setitem__("phi", o.detach()))]
    with torch.no_grad():
        ref(img, img, seg, ref_seg)
        th, ph = mine.project(img, img, seg, ref_seg)
    for h in hooks:
        h.remove()
    for a, b in ((th, got["theta"]), (ph, got["phi"])):
        assert a.shape == b.shape
        assert float((a - b).abs().max() / b.abs().max()) < 1e-5


@needs_ref
def test_install_into_reference_define_corr():
    """The unmodified reference factory builds OUR class once it is injected."""
    networks = rh.load_reference()
    ref_corr = __import__("importlib").import_module("models.networks.correspondence")
    original = ref_corr.NoVGGCorrespondence
    try:
        cls = cc.install_into_reference(networks)
        net = rh.build_reference_corr(rh.make_opt(semantic_nc=5, match_kernel=1))
        assert isinstance(net, cls) and isinstance(net, cc.NoVGGCorrespondence)
        assert isinstance(net, networks.BaseNetwork)
    finally:
        ref_corr.NoVGGCorrespondence = original


@needs_ref
@pytest.mark.parametrize("pono", [True, False])
def test_install_spade_into_reference(pono, monkeypatch):
    """§8(f) rank 1 reaches the reference's OWN generator: after install_spade_into_reference(networks) the
    reference's SPADE / SPADEResnetBlock classes (netG's `SPADEGenerator` and the adaptors are built from them) run
    the fused forward.  On CPU it must reproduce the original arithmetic (outputs and gradients); with a CUDA-fp32
    input it must hand PositionalNorm + modulation + LeakyReLU to K9 as ONE call."""
    import importlib
    from cocosnet_amd import ops, spade
    networks = rh.load_reference()
    arch = importlib.import_module("models.networks.architecture")
    opt = rh.make_opt(semantic_nc=5, PONO=pono, norm_G="spectralspadeinstance3x3")
    opt.spade_ic = 5
    torch.manual_seed(0)
    blk = arch.SPADEResnetBlock(16, 8, opt).eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 8, 8, generator=g)
    seg = torch.rand(2, 5, 32, 32, generator=g)

    def run():
        xx = x.clone().requires_grad_(True)
        y = blk(xx, seg)
        y.pow(2).sum().backward()
        return y.detach(), xx.grad.clone(), blk.norm_0.mlp_gamma.weight.grad.clone()

    ref = run()
    blk.zero_grad()
    try:
        SP, RB = spade.install_spade_into_reference(networks)
        assert SP.forward is spade.spade_forward and RB.forward is spade.spade_resnet_block_forward
        got = run()
        for a, b in zip(got, ref):
            assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
        # the dispatch on a GPU tensor: one fused call per (norm, activation) pair, slope 0.2; 1.0 on the shortcut
        calls = []

        def fake_pono_spade(xx, gamma, beta, slope=1.0, eps=1e-5):
            calls.append(slope)
            mu = xx.mean(1, keepdim=True)
            y = (xx - mu) / xx.var(1, keepdim=True).add(eps).sqrt() * (1 + gamma) + beta
            return torch.nn.functional.leaky_relu(y, slope) if slope != 1.0 else y

        mod_calls = []

        def fake_spade_modulate(xh, gamma, beta, slope=1.0):      # K17's arithmetic
            mod_calls.append(slope)
            y = xh * (1 + gamma) + beta
            return torch.nn.functional.leaky_relu(y, slope) if slope != 1.0 else y

        monkeypatch.setattr(spade, "_hip_ok", lambda *ts: True)
        monkeypatch.setattr(ops, "pono_spade", fake_pono_spade)
        monkeypatch.setattr(ops, "spade_modulate", fake_spade_modulate)
        with torch.no_grad():
            y2 = blk(x, seg)
        assert torch.allclose(y2, ref[0], atol=1e-5, rtol=1e-5)
        if pono:
            assert sorted(calls) == [0.2, 0.2, 1.0] and mod_calls == []
        else:       # instance-norm SPADE: statistics stay with nn.InstanceNorm2d, modulation + activation go to K17
            assert calls == [] and sorted(mod_calls) == [0.2, 0.2, 1.0]
    finally:
        spade.uninstall_spade_from_reference(networks)
    assert not hasattr(arch.SPADEResnetBlock, "_cocos_reference_forward")
    y3 = blk(x, seg)
    assert torch.allclose(y3.detach(), ref[0], atol=1e-6)


@needs_ref
def test_contextual_loss_module_equals_the_reference_class_on_cpu():
    """cocosnet_amd.contextual.ContextualLoss_forward restates ContextualLoss.py:93-137: on CPU tensors (where it runs
    the same formulation in torch) it must agree with the reference class itself, PONO and non-PONO centring."""
    import importlib
    from types import SimpleNamespace
    from cocosnet_amd.contextual import ContextualLoss_forward, install_contextual_loss_into_reference
    rh.load_reference()
    ref_mod = importlib.import_module("models.networks.ContextualLoss")
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(2, 16, 6, 5, generator=g), torch.randn(2, 16, 6, 5, generator=g) + 0.2
    for pono in (True, False):
        opt = SimpleNamespace(PONO=pono)
        a = ContextualLoss_forward(opt)(X, Y, h=0.1)
        b = ref_mod.ContextualLoss_forward(opt)(X, Y, h=0.1)
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    original = ref_mod.ContextualLoss_forward
    try:
        assert install_contextual_loss_into_reference(importlib.import_module("models.networks")) is ContextualLoss_forward
        assert ref_mod.ContextualLoss_forward is ContextualLoss_forward
    finally:
        ref_mod.ContextualLoss_forward = original


def test_use_hip_convs_reclasses_in_place_and_keeps_the_cpu_result():
    """producers.use_hip_convs: same parameters / keys / spectral-norm hooks, CPU forward unchanged (no GPU: the
    framework path), and reversible."""
    from cocosnet_amd import producers
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.utils.spectral_norm(torch.nn.Conv2d(3, 8, 4, stride=2, padding=2)),
                              torch.nn.LeakyReLU(0.2), torch.nn.Conv2d(8, 1, 4, stride=1, padding=2))
    net.eval()
    x = torch.randn(2, 3, 16, 16)
    keys, y0 = list(net.state_dict().keys()), net(x)
    assert producers.use_hip_convs(net) == 2
    assert all(isinstance(m, producers.Conv2d) for m in net if isinstance(m, torch.nn.Conv2d))
    assert list(net.state_dict().keys()) == keys
    assert torch.equal(net(x), y0)
    assert producers.use_framework_convs(net) == 2 and type(net[0]) is torch.nn.Conv2d


@needs_ref
def test_use_hip_convs_on_the_reference_discriminator():
    """The reference's PatchGAN (discriminator.py:92-115) takes the re-classing: every k4 convolution becomes a
    producers.Conv2d and the multiscale forward still runs (CPU here; the GPU parity of the kernel itself is
    tests/test_gpu_conv.py)."""
    from cocosnet_amd import producers
    rh.load_reference()
    disc = __import__("importlib").import_module("models.networks.discriminator")
    opt = rh.make_opt(semantic_nc=5)
    for k, v in dict(ndf=8, n_layers_D=3, norm_D="spectralinstance", netD_subarch="n_layer", num_D=2, label_nc=4,
                     contain_dontcare_label=True, no_instance=True, no_ganFeat_loss=False, D_cam=0.0, output_nc=3,
                     use_attention=False, use_attention_st1=False, eqlr_sn=False).items():
        setattr(opt, k, v)
    try:
        net = disc.MultiscaleDiscriminator(opt)
    except Exception as e:     # option surface of the reference differs: not this test's business
        pytest.skip(f"reference discriminator not constructible with the stand-in options: {e}")
    n = producers.use_hip_convs(net)
    assert n >= 2 * 4
    x = torch.randn(1, opt.label_nc + 1 + 3, 32, 32)
    out = net(x)
    assert len(out) >= 1


# ------------------------------------------------------------------ BASELINE config 3: the translation generator and the PatchGAN
def _cfg3_opts():
    from cocosnet_amd import translation as tl
    mine = tl.celebahq_edge_train_options()
    ref = rh.make_opt(**{k: getattr(mine, k) for k in vars(mine) if k != "gpu_ids"})
    ref.gpu_ids = []
    return mine, ref


@needs_ref
def test_translation_generator_and_discriminator_state_dicts_equal_the_reference():
    """cocosnet_amd.translation.{SPADEGenerator, MultiscaleDiscriminator} (BASELINE config 3) against the reference's own
    classes (generator.py:17-89, discriminator.py:14-177), CelebA-HQ edge training flags: same state_dict keys and shapes —
    `*_net_G.pth` / `*_net_D.pth` load unchanged — and, with the reference's parameters loaded, the same fp32 CPU outputs."""
    import contextlib
    from cocosnet_amd import translation as tl
    mine, ref = _cfg3_opts()
    networks = rh.load_reference()
    import importlib
    gen = importlib.import_module("models.networks.generator")
    dis = importlib.import_module("models.networks.discriminator")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(None):
        rg, rd = gen.SPADEGenerator(ref).eval(), dis.MultiscaleDiscriminator(ref).eval()
    g, d = tl.SPADEGenerator(mine).eval(), tl.MultiscaleDiscriminator(mine).eval()
    for a, b in ((g, rg), (d, rd)):
        assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {k: tuple(v.shape) for k, v in b.state_dict().items()}
        a.load_state_dict(b.state_dict(), strict=True)
    gg = torch.Generator().manual_seed(3)
    seg = torch.rand(1, 15, 256, 256, generator=gg)
    cbn = torch.cat((torch.rand(1, 3, 256, 256, generator=gg) * 2 - 1, seg), 1)
    with torch.no_grad():
        y, yr = g(seg, warp_out=cbn), rg(seg, warp_out=cbn)
        assert float((y - yr).abs().max()) < 1e-5 * float(yr.abs().max())
        x = torch.cat((seg, y), 1)
        (o, _, _), (orf, _, _) = d(x), rd(x)
    assert len(o) == len(orf) == 2
    for a, b in zip(o, orf):
        assert len(a) == len(b) == 5
        for u, v in zip(a, b):
            assert u.shape == v.shape and float((u - v).abs().max()) <= 1e-5 * max(float(v.abs().max()), 1e-6)
