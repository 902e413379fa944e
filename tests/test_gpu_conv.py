"""K16 (conv_f16x3.hip) against torch's fp64 conv2d: forward, input gradient, weight gradient, bias gradient.
Shapes: the ResidualBlock convolution (correspondence.py:13-36: 3x3 after ReflectionPad2d(1)), the adaptor layers
(correspondence.py:150-173: k3 s1 p1, k4 s2 p1), the PatchGAN layers (discriminator.py:92-115: k4 s2 p2, odd outputs),
and ragged sizes that exercise every mask.  Tolerance 1e-5 of the output range (the split carries ~22 bits)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin, H, W, Cout, k, stride, pad
    (2, 407, 66, 66, 407, 3, 1, 0),      # ResidualBlock, B reduced
    (1, 64, 64, 64, 128, 3, 1, 1),       # zero padding, fast gather
    (2, 3, 33, 37, 16, 4, 2, 2),         # PatchGAN first layer shape class: odd outputs, element gather
    (1, 16, 18, 22, 40, 4, 2, 1),        # adaptor down-sampling
    (3, 5, 9, 7, 7, 3, 1, 1),            # everything ragged
    (1, 8, 12, 12, 130, 1, 1, 0),        # 1x1, Cout > 128 tile class
    (2, 20, 10, 16, 24, 3, 1, 2),        # pad > (k-1)/2: output larger than input
    (1, 4, 8, 8, 4, 5, 1, 2),            # 5x5
    (1, 2, 6, 6, 3, 3, 2, 0),            # stride 2, tiny
    (1, 3, 10, 11, 4, 1, 2, 0),          # 1x1 stride 2: three of the four parity classes of dx see no tap
    (1, 3, 13, 12, 4, 5, 3, 2),          # stride 3, 5x5
    (1, 3, 8, 8, 2, 2, 3, 0),            # kernel smaller than the stride
    (2, 64, 32, 32, 130, 3, 2, 1),       # adaptor down-sampling layer class (k3 s2 p1), wide
    (2, 12, 20, 24, 10, 3, 1, 2, 2),     # dilation 2 (the adaptors' de-gridding convolution: k3 p2 d2)
    (1, 6, 17, 13, 5, 3, 2, 3, 3),       # dilation 3, stride 2, ragged
]


def _ref(x, w, b, stride, pad, go, dil=1):
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    bd = None if b is None else b.double().requires_grad_(True)
    y = F.conv2d(xd, wd, bd, stride=stride, padding=pad, dilation=dil)
    y.backward(go.double())
    return y.detach(), xd.grad, wd.grad, None if b is None else bd.grad


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("with_bias", [True, False])
def test_conv2d_matches_fp64(case, with_bias):
    from cocosnet_amd import ops
    B, Cin, H, W, Cout, k, stride, pad = case[:8]
    dil = case[8] if len(case) > 8 else 1
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, device="cuda", generator=g).requires_grad_(True) if with_bias else None
    y = ops.conv2d(x, w, b, stride, pad, dil)
    go = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(go)
    yr, dxr, dwr, dbr = _ref(x.detach(), w.detach(), None if b is None else b.detach(), stride, pad, go, dil)
    assert y.shape == yr.shape

    def close(a, r, what):
        err = (a.double() - r).abs().max().item()
        tol = 1e-5 * max(r.abs().max().item(), 1e-30)
        assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e}"

    close(y, yr, "y")
    close(x.grad, dxr, "dx")
    close(w.grad, dwr, "dw")
    if with_bias:
        close(b.grad, dbr, "db")


@pytest.mark.parametrize("case", [(2, 407, 66, 66, 407, 3, 1, 0), (1, 130, 20, 24, 140, 3, 1, 1), (2, 40, 9, 11, 200, 3, 2, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv2d_wide_tile_matches_fp64(case, monkeypatch):
    """The 256 x 256 tile of the forward / input-gradient kernel (picked for wide layers on large grids), forced here
    through COCOS_CONV_BN on shapes that finish quickly — including ragged ones that leave most of a tile empty."""
    from cocosnet_amd import ops
    monkeypatch.setenv("COCOS_CONV_BN", "256")
    B, Cin, H, W, Cout, k, stride, pad = case
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, device="cuda", generator=g).requires_grad_(True)
    y = ops.conv2d(x, w, b, stride, pad)
    go = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(go)
    yr, dxr, dwr, dbr = _ref(x.detach(), w.detach(), b.detach(), stride, pad, go)
    for a, r, what in ((y, yr, "y"), (x.grad, dxr, "dx"), (w.grad, dwr, "dw"), (b.grad, dbr, "db")):
        err = (a.double() - r).abs().max().item()
        assert err <= 1e-5 * r.abs().max().item(), f"{what}: {err:.3e}"


def test_conv2d_operand_ranges():
    """Operands far from O(1): the device-side power-of-two scales keep the f16 planes in range."""
    from cocosnet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    for sx, sw in ((1e4, 1e-5), (1e-6, 1e3)):
        x = (torch.randn(2, 12, 16, 16, device="cuda", generator=g) * sx).requires_grad_(True)
        w = (torch.randn(9, 12, 3, 3, device="cuda", generator=g) * sw).requires_grad_(True)
        y = ops.conv2d(x, w, None, 1, 1)
        go = torch.randn(y.shape, device="cuda", generator=g)
        y.backward(go)
        yr, dxr, dwr, _ = _ref(x.detach(), w.detach(), None, 1, 1, go)
        for a, r in ((y, yr), (x.grad, dxr), (w.grad, dwr)):
            assert (a.double() - r).abs().max().item() <= 1e-5 * r.abs().max().item()


def test_conv2d_rejects_bad_arguments():
    from cocosnet_amd import ops
    x = torch.randn(1, 4, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        ops.conv2d(x, torch.randn(4, 5, 3, 3, device="cuda"))
    with pytest.raises(ValueError):
        ops.conv2d(torch.randn(1, 4, 2, 2, device="cuda"), torch.randn(4, 4, 3, 3, device="cuda"))
    with pytest.raises(Exception, match="no CPU fallback"):
        ops.conv2d(x.cpu(), torch.randn(4, 4, 3, 3))


@pytest.mark.parametrize("shape,slope", [((2, 16, 12, 12), 0.2), ((1, 5, 7, 9), 1.0), ((3, 64, 32, 32), 0.2)])
def test_spade_modulate_matches_torch_fp64(shape, slope):
    """K17: leaky_relu(xh*(1+gamma)+beta) and its three gradients (normalization.py:148 + architecture.py:88-95)."""
    from cocosnet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    xh, ga, be = (torch.randn(shape, device="cuda", generator=g).requires_grad_(True) for _ in range(3))
    y = ops.spade_modulate(xh, ga, be, slope)
    go = torch.randn(shape, device="cuda", generator=g)
    y.backward(go)
    xd, gd, bd = (t.detach().double().requires_grad_(True) for t in (xh, ga, be))
    yr = F.leaky_relu(xd * (1 + gd) + bd, slope)
    yr.backward(go.double())
    for a, r in ((y, yr), (xh.grad, xd.grad), (ga.grad, gd.grad), (be.grad, bd.grad)):
        assert (a.double() - r.detach()).abs().max().item() <= 1e-5 * max(r.abs().max().item(), 1.0)


def test_spade_forward_instance_norm_branch_uses_the_kernels():
    """spade.modulate with a parameter-free InstanceNorm (non-PONO SPADE): same result as the torch formulation."""
    from cocosnet_amd import spade
    g = torch.Generator(device="cuda").manual_seed(4)
    x, ga, be = (torch.randn(2, 8, 16, 16, device="cuda", generator=g) for _ in range(3))
    norm = torch.nn.InstanceNorm2d(8, affine=False)
    y = spade.modulate(x, ga, be, False, norm, 0.2)
    ref = F.leaky_relu(norm(x.double()) * (1 + ga.double()) + be.double(), 0.2)
    assert (y.double() - ref).abs().max().item() < 1e-5


def test_patchgan_stack_on_hip_convs_matches_the_framework():
    """A PatchGAN-shaped stack (discriminator.py:92-115: k4 s2 convolutions, spectral norm, InstanceNorm, LeakyReLU, a final
    k4 s1 convolution) re-classed with producers.use_hip_convs: forward and every gradient equal the framework's
    convolutions to fp32 round-off (the same module, `use_framework_convs` in between)."""
    from cocosnet_amd import producers
    torch.manual_seed(0)
    sn = torch.nn.utils.spectral_norm
    net = torch.nn.Sequential(
        torch.nn.Conv2d(7, 16, 4, stride=2, padding=1), torch.nn.LeakyReLU(0.2),
        sn(torch.nn.Conv2d(16, 32, 4, stride=2, padding=1)), torch.nn.InstanceNorm2d(32), torch.nn.LeakyReLU(0.2),
        sn(torch.nn.Conv2d(32, 64, 4, stride=1, padding=1)), torch.nn.InstanceNorm2d(64), torch.nn.LeakyReLU(0.2),
        torch.nn.Conv2d(64, 1, 4, stride=1, padding=1)).cuda().eval()       # eval: spectral norm's u is not advanced
    x = torch.randn(2, 7, 64, 48, device="cuda")

    def run():
        net.zero_grad()
        xx = x.clone().requires_grad_(True)
        y = net(xx)
        y.square().mean().backward()
        return [y.detach(), xx.grad] + [p.grad.clone() for p in net.parameters()]
    ref = run()
    assert producers.use_hip_convs(net) == 4
    got = run()
    producers.use_framework_convs(net)
    for a, r in zip(got, ref):
        assert (a - r).abs().max().item() <= 2e-4 * max(r.abs().max().item(), 1e-6), (a.shape, (a - r).abs().max().item())


def _module_e2e(monkeypatch, flags, size=64, B=2, smooth_adaptors=False, forced=False, seed=2):     # smooth_adaptors: ALL kinks
    """The whole drop-in module — adaptors with SPADE blocks, feature_normalize, four ResidualBlocks, theta / phi, AND the hot
    path behind them — in one graph: ours (every convolution / norm / correlation kernel on HIP, `arm` = f16x3 | bf16 | torch)
    against a torch-FP64 copy of the same module (same parameters and buffers; framework ops in double up to theta / phi, the
    oracle's restatement of correspondence.py:272-372 in double from there on).  Returns {arm: {name: relative error}} for
    features, outputs and parameter gradients of loss = sum_k <out_k, G_k>.
    forced: the fp64 copy RECORDS the branch of every LeakyReLU / PReLU / ReLU / max-pool element (tests/kink_tape.py) and the
    fp32 arms are evaluated ON THAT PATTERN (their kernels run with the slope at 1, the recorded pattern is the multiplier);
    errs[arm]["#flips"] = [elements whose own branch differed, elements]."""
    import contextlib
    import copy
    import kink_tape
    from cocosnet_amd import correspondence as cc
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    from oracle import torch_ref as tr
    opt = cc.base_options(**flags)
    torch.manual_seed(0)
    net = cc.NoVGGCorrespondence(opt).cuda()
    net.init_weights(opt.init_type, opt.init_variance)
    net.eval()                     # freezes the spectral-norm power iteration: both copies see the same W / sigma
    if smooth_adaptors:            # every kink of the module -> identity (see the test's docstring); all kernels still run
        from cocosnet_amd import producers
        net.adaptive_model_seg.actvn.negative_slope = net.adaptive_model_img.actvn.negative_slope = 1.0
        for m in net.modules():
            if isinstance(m, torch.nn.PReLU):
                m.weight.data.fill_(1.0)
            elif isinstance(m, producers.SPADEResnetBlock):
                m.slope = 1.0
            elif isinstance(m, producers.SPADE):
                m.mlp_shared[2] = torch.nn.Identity()
    g = torch.Generator(device="cuda").manual_seed(seed)
    nc = flags["semantic_nc"]
    img = torch.rand(B, 3, size, size, device="cuda", generator=g) * 2 - 1
    real = torch.rand(B, 3, size, size, device="cuda", generator=g) * 2 - 1
    lab = torch.randint(0, nc, (B, 1, size // 8, size // 8), device="cuda", generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    seg = torch.zeros(B, nc, size, size, device="cuda").scatter_(1, lab, 1.0)
    ref_seg = seg.flip(0).contiguous()
    probes = {"theta.weight": lambda n: n.theta.weight, "phi.bias": lambda n: n.phi.bias,
              "layer.0.conv1.weight": lambda n: n.layer[0].conv1.weight, "layer.3.conv2.weight": lambda n: n.layer[3].conv2.weight,
              "layer.0.prelu.weight": lambda n: n.layer[0].prelu.weight,
              "adaptive_model_img.layer1.0.weight_orig": lambda n: n.adaptive_model_img.layer1[0].weight_orig,
              "adaptive_model_img.layer4.0.weight_orig": lambda n: n.adaptive_model_img.layer4[0].weight_orig,
              "adaptive_model_seg.layer2.0.weight_orig": lambda n: n.adaptive_model_seg.layer2[0].weight_orig,
              "adaptive_model_seg.G_middle_1.conv_0.weight_orig": lambda n: n.adaptive_model_seg.G_middle_1.conv_0.weight_orig,
              "adaptive_model_seg.head_0.norm_0.mlp_gamma.weight": lambda n: n.adaptive_model_seg.head_0.norm_0.mlp_gamma.weight}
    cfg = HotPathConfig.from_opt(opt, down=opt.down)

    # ---- the fp64 arbiter -------------------------------------------------------------------------------------------------
    net64 = copy.deepcopy(net).double()
    d = lambda t: t.double()
    tape = kink_tape.KinkTape() if forced else None
    stack = contextlib.ExitStack()
    stack.enter_context(kink_tape.install(tape))       # (None: the product's own forward everywhere)
    th64, ph64 = net64.project(d(img), d(real), d(seg), d(ref_seg))
    out64 = tr.hot_path(th64, ph64, d(img), d(real), d(seg), d(ref_seg), cfg)
    keys = sorted(out64)
    G = {k: torch.randn(out64[k].shape, device="cuda", generator=g) for k in keys}
    torch.autograd.backward([out64[k] for k in keys], [d(G[k]) for k in keys])
    want = {"theta_raw": th64.detach(), "phi_raw": ph64.detach()}
    want.update({k: out64[k].detach() for k in keys})
    want.update({"d " + n: f(net64).grad.clone() for n, f in probes.items()})

    def arm(backend):
        monkeypatch.setattr(ops, "CONV_PRECISION", backend)
        net.zero_grad()
        if tape is not None:
            tape.rewind("replay")
        th, ph = net.project(img, real, seg, ref_seg)
        out = correspondence_hot_path(th, ph, img, real, seg, ref_seg, cfg)
        assert sorted(out) == keys
        torch.autograd.backward([out[k] for k in keys], [G[k] for k in keys])
        got = {"theta_raw": th.detach(), "phi_raw": ph.detach()}
        got.update({k: out[k].detach() for k in keys})
        got.update({"d " + n: f(net).grad.clone() for n, f in probes.items()})
        e = {k: float((got[k].double() - want[k]).abs().max() / (want[k].abs().max() + 1e-300)) for k in want}
        if tape is not None:
            assert tape.pos == len(tape.masks), (tape.pos, len(tape.masks))
            e["#flips"] = [sum(tape.flips), tape.elements]
        return e
    try:
        return {b: arm(b) for b in ("f16x3", "torch", "bf16")}
    finally:
        stack.close()


E2E_FLAGS = {
    # README.md:45 / :88 ADE20k: maskmix, direct mask, on match_kernel 1 and on the shipped default 3
    "ade20k_mk1": dict(semantic_nc=6, match_kernel=1, maskmix=True, PONO=True, PONO_C=True, warp_mask_losstype="direct", isTrain=True),
    "ade20k_mk3": dict(semantic_nc=6, match_kernel=3, maskmix=True, PONO=True, PONO_C=True, warp_mask_losstype="direct", isTrain=True),
    # README.md:106 CelebA-HQ edge training: adaptor_kernel 4, bilinear, cycle terms (both softmax directions)
    "celebaedge_mk1": dict(semantic_nc=5, match_kernel=1, maskmix=True, PONO=True, PONO_C=True, warp_bilinear=True, adaptor_kernel=4,
                           warp_cycle_w=1.0, isTrain=True),
}


def _is_kinked(probe: str) -> bool:
    """Parameter gradients that sit UPSTREAM of an InstanceNorm / PONO -> LeakyReLU / PReLU / ReLU site are discontinuous functions
    of the features: every probed `d ...` except theta's and phi's own (everything between those and the loss is smooth)."""
    return probe.startswith("d ") and probe not in ("d theta.weight", "d phi.bias")


@pytest.mark.parametrize("name", sorted(E2E_FLAGS))
def test_module_end_to_end_against_an_fp64_copy_of_itself(name, monkeypatch):
    """VERDICT r3 weak 1a / 1b.  project() + the hot path in ONE graph against torch-fp64, as the PRODUCT runs it (every site its
    fused kernel): features, outputs and the gradients of theta / phi within north_star's 1e-3 for the default flavour (f16x3
    convolutions); the framework-fp32 arm (`torch`: MIOpen convolutions, our norms and hot path) is the yardstick of what fp32
    arithmetic itself loses; the bf16 arm's figures are REPORTED (printed here, collected into profiles/ by tools/final_artifacts.sh).

    Every OTHER parameter gradient sits upstream of InstanceNorm -> LeakyReLU(0.2) / PReLU kinks: a normalised value within fp32
    rounding of zero takes the other branch than in fp64, and ONE such element (the twin test below counts 0-3 of 9.1 M per arm) moves
    d weight by 1e-3 .. 6e-2 of its range — for any fp32 implementation, the framework's included (round 5: with the operand split
    rounding to nearest instead of truncating, OTHER elements flip than in round 4 and probes that had passed at 1e-3 by luck no longer
    did, on the framework arm just as on ours).  Those gradients are therefore reported and sanity-bounded here, and HELD to 1e-3 in
    test_module_end_to_end_every_gradient_on_the_fp64_branch_pattern, where both evaluations take the same branches."""
    errs = _module_e2e(monkeypatch, E2E_FLAGS[name])
    import json
    print("E2E_FP64", name, json.dumps(errs))
    bad = {k: v for k, v in errs["f16x3"].items() if not _is_kinked(k) and not v < 1e-3}
    assert not bad, (bad, {k: errs["torch"][k] for k in bad})
    # the kinked probes are only required to be finite here: a magnitude bound cannot tell one unlucky element (6.3e-2 in round 5)
    # from a kernel that flips systematically more — that is asserted on the flip COUNTS per arm
    # (test_flip_counts_of_the_split_arm_stay_within_the_framework_arms) and on the fp64 branch pattern (the twin test below)
    import math
    assert all(math.isfinite(v) for k, v in errs["f16x3"].items() if _is_kinked(k)), errs["f16x3"]
    # the one-term flavour is not held to 1e-3 (it is not parity-qualified: DESIGN.md §3.6); it must be finite and sane
    assert all(v < 0.6 for v in errs["bf16"].values()), errs["bf16"]


@pytest.mark.parametrize("name", sorted(E2E_FLAGS))
def test_module_end_to_end_every_gradient_on_the_fp64_branch_pattern(name, monkeypatch):
    """the kink probes MEASURED instead of bounded at 0.1.  The fp64 copy records which branch every
    LeakyReLU / PReLU / ReLU element (and every 2x2 max-pool window) takes; the fp32 arms run their production kernels with the
    activation's slope at 1 and apply the recorded pattern (tests/kink_tape.py), so that the comparison is between two evaluations
    of the SAME piecewise-linear function.  On that footing EVERY probed gradient of the default flavour is within north_star's 1e-3
    — the adaptors' strided layers included, and celebaedge_mk1's layer1.0 (4.5e-3 on this arm, 9e-6 on the framework arm in round 4:
    ONE element of the f16x3 arm's own flips, see the printed counts).  The counts say how many elements each fp32 arm would have
    put on the other branch than fp64."""
    errs = _module_e2e(monkeypatch, E2E_FLAGS[name], forced=True)
    import json
    print("E2E_FP64_FORCED", name, json.dumps(errs))
    bad = {k: v for k, v in errs["f16x3"].items() if k != "#flips" and not v < 1e-3}
    assert not bad, (bad, {k: errs["torch"][k] for k in bad}, errs["f16x3"]["#flips"])


def test_flip_counts_of_the_split_arm_stay_within_the_framework_arms(monkeypatch):
    """over three input seeds, the number of piecewise-linear elements that the f16x3 arm puts on the other branch
    than fp64 must stay within what fp32 arithmetic itself does — f16x3 flips <= 2 x framework-fp32 flips + 4 per seed (1-3 of 9.1 M
    on both arms when this was written).  A kernel that flipped systematically more would pass any magnitude bound on the probes;
    it cannot pass this."""
    import json
    for seed in (2, 11, 23):
        errs = _module_e2e(monkeypatch, E2E_FLAGS["ade20k_mk1"], forced=True, seed=seed)
        ours, elements = errs["f16x3"]["#flips"]
        theirs, _ = errs["torch"]["#flips"]
        print("E2E_FLIPS", seed, json.dumps({"f16x3": ours, "torch": theirs, "bf16": errs["bf16"]["#flips"][0], "elements": elements}))
        assert ours <= 2 * theirs + 4, (seed, ours, theirs, elements)


def test_module_end_to_end_against_fp64_every_gradient_without_the_adaptor_kinks(monkeypatch):
    """The same comparison with EVERY kink of the module taken out in BOTH copies — the adaptors' LeakyReLU and the SPADE blocks'
    LeakyReLU at slope 1, the ResidualBlocks' PReLU at weight 1, SPADE's ReLU replaced by the identity; every kernel still runs
    (K13 / K9 with a = 1): no branch can be taken differently, and EVERY probed gradient is within 1e-3 of fp64 on the default
    flavour."""
    errs = _module_e2e(monkeypatch, E2E_FLAGS["ade20k_mk3"], smooth_adaptors=True)
    import json
    print("E2E_FP64_SMOOTH", json.dumps(errs))
    bad = {k: v for k, v in errs["f16x3"].items() if not v < 1e-3}
    assert not bad, bad


@pytest.mark.parametrize("shape,pad", [((2, 5, 8, 11), 1), ((1, 3, 4, 4), 3), ((2, 16, 64, 64), 1), ((1, 2, 7, 5), 2), ((1, 1, 3, 9), 0)])
def test_reflect_pad2d_matches_torch(shape, pad):
    """K18 vs nn.ReflectionPad2d: forward bit-exact (pure data movement), backward equal to autograd's (a gather here)."""
    from cocosnet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn(shape, device="cuda", generator=g).requires_grad_(True)
    y = ops.reflect_pad2d(x, pad)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.pad(xr, (pad,) * 4, mode="reflect") if pad else xr * 1.0
    assert torch.equal(y, yr)
    go = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(go)
    # arbiter of the backward: the framework's CPU kernel in fp64 (its GPU kernel under torch.use_deterministic_algorithms — the
    # COCOS_POISON_EMPTY=1 run of this suite — returns wrong sums where several mirrored pixels fold onto one: (4, 4) pad 3 off by 2.8)
    xc = x.detach().cpu().double().requires_grad_(True)
    (F.pad(xc, (pad,) * 4, mode="reflect") if pad else xc * 1.0).backward(go.cpu().double())
    assert (x.grad.cpu().double() - xc.grad).abs().max().item() <= 1e-6 * max(xc.grad.abs().max().item(), 1.0)
    with pytest.raises(ValueError):
        ops.reflect_pad2d(x.detach(), min(shape[2:]))


# ------------------------------------------------------------------ the one-term bf16 flavour (BASELINE config 3's precision)
BF16_CASES = [
    (2, 407, 66, 66, 407, 3, 1, 0),      # ResidualBlock shape (for the error figure only: that block stays on f16x3)
    (2, 64, 64, 64, 128, 3, 1, 1),       # SPADE mlp_shared class
    (2, 3, 33, 37, 16, 4, 2, 2),         # PatchGAN first layer: odd outputs, element gather, strided input gradient
    (2, 64, 32, 32, 130, 4, 2, 2),       # PatchGAN inner layer class
    (3, 5, 9, 7, 7, 3, 1, 1),            # everything ragged
    (2, 12, 20, 24, 10, 3, 1, 2, 2),     # dilation 2
    # K16b (conv_nhwc_bf16.hip: stride 1, >= 128 output channels, >= 32 input channels)
    (1, 40, 20, 36, 200, 3, 1, 1),       # ragged rows (OW = 36: its weight gradient stays on K16), row / position tails
    (2, 128, 34, 34, 512, 3, 1, 0),      # SPADE gamma/beta class, OW = 32: forward + both gradients on K16b, 128-column tiles
    (2, 32, 44, 44, 256, 3, 1, 2, 2),    # dilation 2
    (1, 96, 8, 64, 130, 1, 1, 0),        # 1x1 kernel, 3 k-steps
    (2, 32, 16, 32, 128, 1, 1, 0),       # ONE k-step (the pipeline's prologue is everything)
    (1, 64, 6, 64, 160, 5, 1, 2),        # 5x5, 50 k-steps
    (9, 256, 64, 64, 256, 3, 1, 1),      # 36864 positions: 256-column tiles, two slices of the weight gradient per tile
    (2, 64, 64, 128, 160, 3, 2, 1),      # stride 2 on K16b (forward + weight gradient; input gradient by parity classes), OW = 64
    (2, 32, 33, 37, 128, 4, 2, 2),       # stride 2, k4, ragged output (weight gradient on K16)
]


@pytest.mark.parametrize("case", BF16_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d_bf16_flavour_error_vs_fp64(case, monkeypatch):
    """COCOS_CONV=bf16: single bf16 operand planes, one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate (VERDICT r2
    missing 4).  bf16 keeps 8 mantissa bits: with K = Cin*k*k random products the error of a sum is ~2^-9 * sqrt(K) of
    a term, i.e. a few 1e-3 of the output range — asserted at 1.5e-2 (forward, input / weight / bias gradient), and the
    SAME inputs through the f16x3 flavour stay at 1e-5: the two flavours are different arithmetic, not different code paths."""
    from cocosnet_amd import ops
    B, Cin, H, W, Cout, k, stride, pad = case[:8]
    dil = case[8] if len(case) > 8 else 1
    g = torch.Generator(device="cuda").manual_seed(21)
    x0 = torch.randn(B, Cin, H, W, device="cuda", generator=g)
    w0 = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    b0 = torch.randn(Cout, device="cuda", generator=g)
    errs = {}
    for prec, tol in (("bf16", 1.5e-2), ("f16x3", 1e-5)):
        monkeypatch.setattr(ops, "CONV_PRECISION", prec)
        x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
        with ops.KernelTimer() as kt:
            y = ops.conv2d(x, w, b, stride, pad, dil)
            if prec == "bf16":
                go = torch.randn(y.shape, device="cuda", generator=g)
            y.backward(go)
        assert ("absmax" in kt.summary()) == (prec == "f16x3")          # bf16: no max|x| passes at all
        yr, dxr, dwr, dbr = _ref(x0, w0, b0, stride, pad, go, dil)
        for a, r, what in ((y, yr, "y"), (x.grad, dxr, "dx"), (w.grad, dwr, "dw"), (b.grad, dbr, "db")):
            e = (a.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
            errs[(prec, what)] = e
            assert e <= tol, f"{prec} {what}: {e:.3e}"
    assert errs[("bf16", "y")] > 1e-4          # it really is the low-precision arithmetic


@pytest.mark.parametrize("B,C,H,W,pad,reflect", [(2, 37, 9, 70, 1, True), (1, 64, 5, 5, 2, False), (3, 3, 8, 8, 0, False),
                                                 (1, 130, 4, 66, 3, True)])
def test_conv_nhwc_prep_is_pad_plus_bf16_plus_permute(B, C, H, W, pad, reflect):
    """cocos_conv2d_nhwc_prep_bf16 against the framework's pad -> bf16 (round to nearest even) -> NHWC, bit for bit."""
    from cocosnet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, C, H, W, device="cuda", generator=g) * 3
    xp = ops.conv_nhwc_prep(x, pad, reflect)
    Cp = (C + 31) // 32 * 32
    assert xp.shape == (B, H + 2 * pad, W + 2 * pad, Cp) and xp.dtype == torch.bfloat16
    ref = torch.nn.functional.pad(x, (pad,) * 4, mode="reflect" if reflect else "constant") if pad else x
    ref = torch.nn.functional.pad(ref.to(torch.bfloat16).permute(0, 2, 3, 1), (0, Cp - C))
    assert torch.equal(xp, ref)


@pytest.mark.parametrize("prec,tol", [("bf16", 2e-5), ("f16x3", 2e-6)])
def test_conv2d_bf16_nhwc_matches_the_gather_flavour(prec, tol, monkeypatch):
    """Same arithmetic (one bf16 term / three f16 terms, fp32 accumulate) through the two data paths: K16b / K16c (operands 16-bit NHWC
    in memory, LDS-DMA) against conv_f16x3.hip's kernels (fp32 NCHW gathered per tap) — differences are summation order only."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "CONV_PRECISION", prec)
    g = torch.Generator(device="cuda").manual_seed(8)
    x0 = torch.randn(2, 96, 34, 66, device="cuda", generator=g)
    w0 = torch.randn(288, 96, 3, 3, device="cuda", generator=g) / 30
    b0 = torch.randn(288, device="cuda", generator=g)
    go = torch.randn(2, 288, 32, 64, device="cuda", generator=g)
    res = {}
    for nhwc in (True, False):
        monkeypatch.setattr(ops, "CONV_NHWC", nhwc)
        monkeypatch.setattr(ops, "CONV_NHWC_F16X3", nhwc)
        x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
        with ops.KernelTimer() as kt:
            y = ops.conv2d(x, w, b, 1, 0, 1)
            y.backward(go)
        assert ("conv2d_nhwc_prep" in kt.summary()) == nhwc
        res[nhwc] = (y.detach(), x.grad, w.grad, b.grad)
    for a, r, what in zip(res[True], res[False], ("y", "dx", "dw", "db")):
        e = (a - r).abs().max().item() / r.abs().max().item()
        assert e <= tol, f"{what}: {e:.3e}"


@pytest.mark.parametrize("prec", ["bf16", "f16x3"])
def test_conv2d_reflect_fused_equals_pad_then_conv(prec, monkeypatch):
    """ops.conv2d(reflect=r) on the K16b path (mirrored border written by the operand preparation, input gradient folded back by
    K18's backward) against reflect_pad2d followed by the same layer; shapes the fused path does not take fall back to exactly that."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "CONV_PRECISION", prec)
    g = torch.Generator(device="cuda").manual_seed(9)
    # (Cin >= 128: the input gradient is a K16b / K16c GEMM too, and for r = 1 its epilogue folds the mirrored border back itself)
    for (B, Cin, H, W, Cout, r, fused) in ((2, 64, 32, 64, 192, 1, True), (1, 40, 12, 64, 128, 2, True), (2, 64, 20, 36, 192, 1, False),
                                           (2, 128, 16, 64, 192, 1, True), (1, 160, 9, 32, 130, 1, True), (1, 128, 4, 32, 128, 1, True),
                                           (1, 136, 12, 32, 128, 2, True)):
        k = 2 * r + 1
        x0 = torch.randn(B, Cin, H, W, device="cuda", generator=g)
        w0 = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
        b0 = torch.randn(Cout, device="cuda", generator=g)
        go = torch.randn(B, Cout, H, W, device="cuda", generator=g)
        res = []
        for mode in ("fused", "separate"):
            x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
            with ops.KernelTimer() as kt:
                y = ops.conv2d(x, w, b, 1, 0, 1, reflect=r) if mode == "fused" else ops.conv2d(ops.reflect_pad2d(x, r), w, b, 1, 0, 1)
                y.backward(go)
            if mode == "fused":
                assert ("reflect_pad2d_fwd" not in kt.summary()) == fused
                if fused and Cin >= 128:                   # fold mode: no pass over a padded gradient either (r = 1 only)
                    assert ("reflect_pad2d_bwd" not in kt.summary()) == (r == 1)
            res.append((y.detach(), x.grad, w.grad, b.grad))
        for a, ref, what in zip(res[0], res[1], ("y", "dx", "dw", "db")):
            e = (a - ref).abs().max().item() / ref.abs().max().item()
            assert e <= 2e-5, f"{(B, Cin, H, W, Cout, r)} {what}: {e:.3e}"


def test_reflect_conv_helper_keeps_module_semantics(monkeypatch):
    """producers.reflect_conv(pad, conv, x) == conv(pad(x)) — with spectral norm on the layer (its hook must still run), in both
    convolution flavours, and for a layer the fused path does not take."""
    from cocosnet_amd import ops, producers
    g = torch.Generator(device="cuda").manual_seed(10)
    x = torch.randn(2, 64, 32, 32, device="cuda", generator=g)
    for prec in ("bf16", "f16x3"):
        monkeypatch.setattr(ops, "CONV_PRECISION", prec)
        for cout in (160, 24):
            torch.manual_seed(0)
            conv = torch.nn.utils.spectral_norm(producers.Conv2d(64, cout, 3)).cuda()
            pad = producers.ReflectionPad2d(1)
            conv.eval()                                    # spectral norm: no power iteration, same weight for both calls
            a = producers.reflect_conv(pad, conv, x)
            b = conv(pad(x))
            assert a.shape == b.shape == (2, cout, 32, 32)
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
            c2 = conv(x, reflect=1)                        # the explicit keyword reflect_conv passes (no thread-local note any more)
            assert torch.equal(a, c2)


@pytest.mark.parametrize("prec", ["bf16", "f16x3"])
@pytest.mark.parametrize("Cout,tile", [(288, "256x256 (stream-K)"), (256, "256x128 (never stream-K: measured slower)")])
def test_conv2d_bf16_nhwc_stream_k_equals_one_tile_per_workgroup(Cout, tile, prec, monkeypatch):
    """8 x 66 x 66 output positions are 136.1 column tiles of 256 (272.3 of 128): with two row tiles 274 (273) workgroups for 256
    CUs.  With the workspace K16b runs such a layer as ONE workgroup per CU over contiguous (tile, k-step) ranges, cut tiles meeting
    through the parked partials; the result must be the one-tile-per-workgroup result up to fp32 summation order — on every repeat
    (the flags are left clean) and next to a layer that does not take the path."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "CONV_PRECISION", prec)
    g = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn(8, 32, 68, 68, device="cuda", generator=g)
    w = torch.randn(Cout, 32, 3, 3, device="cuda", generator=g) / 17
    b = torch.randn(Cout, device="cuda", generator=g)
    monkeypatch.setattr(ops, "CONV_NHWC_STREAMK", False)
    monkeypatch.setattr(ops, "CONV_NHWC_STREAMK_SPLIT", False)
    ref = ops.conv2d(x, w, b, 1, 0, 1)
    monkeypatch.setattr(ops, "CONV_NHWC_STREAMK", True)
    monkeypatch.setattr(ops, "CONV_NHWC_STREAMK_SPLIT", True)
    small = ops.conv2d(x[:1], w, b, 1, 0, 1)
    for rep in range(3):
        y = ops.conv2d(x, w, b, 1, 0, 1)
        e = (y - ref).abs().max().item() / ref.abs().max().item()
        assert e <= 2e-6, f"{tile} repeat {rep}: {e:.3e}"
        assert torch.equal(ops.conv2d(x[:1], w, b, 1, 0, 1), small)
    ws = ops._conv_nhwc_workspace(x.device, split=prec == "f16x3")
    assert ws is not None and int(ws[:1024].abs().sum()) == 0          # every flag back to zero
    yr = torch.nn.functional.conv2d(x.double(), w.double(), b.double())
    assert (y.double() - yr).abs().max().item() / yr.abs().max().item() <= (1.5e-2 if prec == "bf16" else 1e-5)


@pytest.mark.parametrize("shape", [(407, 407, 3, 3), (64, 151, 3, 3), (512, 128, 3, 3), (256, 256, 1, 1), (5, 3, 4, 4)])
def test_spectral_weight_matches_the_framework_hook(shape, monkeypatch):
    """K21 against torch.nn.utils.spectral_norm's own compute_weight on the same buffers: the normalised weight, the IN-PLACE update of
    u and v, the gradient through W / sigma (sigma = u . W v depends on W), training and eval mode, two forwards before one backward
    (the GAN pattern the framework clones u / v for)."""
    from cocosnet_amd import ops, producers
    monkeypatch.setattr(ops, "CONV_PRECISION", "f16x3")
    cout, cin, k, _ = shape
    res = {}
    for hip in (True, False):
        monkeypatch.setattr(producers, "SPECTRAL_HIP", hip)
        torch.manual_seed(3)
        conv = producers.hip_spectral_norm(torch.nn.Conv2d(cin, cout, k)).cuda()
        g = torch.Generator(device="cuda").manual_seed(4)
        x1 = torch.randn(2, cin, 8, 8, device="cuda", generator=g)
        x2 = torch.randn(2, cin, 8, 8, device="cuda", generator=g)
        conv.train()
        monkeypatch.setattr(ops, "CONV_PRECISION", "torch" if not hip else "f16x3")     # reference arm: everything on the framework
        y = conv(x1).square().mean() - conv(x2).square().mean()       # two forwards (two power iterations), then backward
        w_train = conv.weight.detach().clone()
        y.backward()
        conv.eval()
        with torch.no_grad():
            conv(x1)
        res[hip] = (w_train, conv.weight_u.clone(), conv.weight_v.clone(), conv.weight_orig.grad.clone(), conv.weight.detach().clone())
        monkeypatch.setattr(ops, "CONV_PRECISION", "f16x3")
    for a, r, what in zip(res[True], res[False], ("weight (train)", "u", "v", "d weight_orig", "weight (eval)")):
        e = (a - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        assert e <= (2e-4 if what == "d weight_orig" else 2e-5), f"{shape} {what}: {e:.3e}"


def _sweep_cases():
    """Seeded sweep of layer shapes around the K16b / K16c limits: output widths that are and are not whole k-steps of 32, channel
    counts next to the 32 / 128 / 256 boundaries, batch sizes that leave position tails, kernel sizes 1..5, strides and dilations."""
    import random
    rnd = random.Random(20250926)
    cases = []
    while len(cases) < 14:
        k = rnd.choice([1, 3, 3, 3, 4, 5])
        stride = rnd.choice([1, 1, 1, 2, 3])
        dil = rnd.choice([1, 1, 2]) if k > 1 else 1
        pad = rnd.choice([0, (dil * (k - 1)) // 2, dil * (k - 1)])
        cin = rnd.choice([32, 33, 47, 64, 96, 127, 128, 130, 200])
        cout = rnd.choice([128, 129, 160, 255, 256, 257, 300])
        B = rnd.choice([1, 2, 3, 5])
        ow = rnd.choice([32, 64, 96, 33, 40, 7])
        oh = rnd.choice([1, 3, 8, 17])
        H = (oh - 1) * stride + dil * (k - 1) + 1 - 2 * pad
        W = (ow - 1) * stride + dil * (k - 1) + 1 - 2 * pad
        if H < 1 or W < 1 or B * cin * H * W > 6e6:
            continue
        cases.append((B, cin, H, W, cout, k, stride, pad, dil))
    return cases


@pytest.mark.parametrize("prec,tol", [("bf16", 1.5e-2), ("f16x3", 1e-5)])
@pytest.mark.parametrize("case", _sweep_cases(), ids=lambda c: "x".join(map(str, c)))
def test_conv2d_nhwc_shape_sweep_vs_fp64(case, prec, tol, monkeypatch):
    from cocosnet_amd import ops
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    monkeypatch.setattr(ops, "CONV_PRECISION", prec)
    g = torch.Generator(device="cuda").manual_seed(31)
    x0 = torch.randn(B, Cin, H, W, device="cuda", generator=g)
    w0 = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    b0 = torch.randn(Cout, device="cuda", generator=g)
    x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
    y = ops.conv2d(x, w, b, stride, pad, dil)
    go = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(go)
    yr, dxr, dwr, dbr = _ref(x0, w0, b0, stride, pad, go, dil)
    for a, r, what in ((y, yr, "y"), (x.grad, dxr, "dx"), (w.grad, dwr, "dw"), (b.grad, dbr, "db")):
        e = (a.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        assert e <= tol, f"{prec} {what}: {e:.3e}"


# ------------------------------------------------------------------ BASELINE config 3: SPADE generator + PatchGAN on the HIP kernels
def _config3_vs_fp64(monkeypatch, forced, arms=("f16x3", "torch", "bf16")):
    """cocosnet_amd.translation.{SPADEGenerator, MultiscaleDiscriminator} with the CelebA-HQ edge training flags (README.md:106)
    at 256x256: forward AND parameter gradients of both networks against fp64 copies of the same modules evaluated by the framework.
    Arms: f16x3 (default: every convolution / SPADE / InstanceNorm / attention kernel on HIP), torch (the framework's fp32
    convolutions under the same norms — the yardstick of what fp32 arithmetic itself loses), bf16 (reported).  forced: every arm on
    the fp64 copy's branch pattern (tests/kink_tape.py); errs[arm]["#flips"] = [own branches that differed, elements]."""
    import copy
    import kink_tape
    from cocosnet_amd import ops, translation as tl
    opt = tl.celebahq_edge_train_options()
    torch.manual_seed(0)
    G = tl.SPADEGenerator(opt).cuda(); G.init_weights(opt.init_type, opt.init_variance); G.eval()
    D = tl.MultiscaleDiscriminator(opt).cuda(); D.init_weights(opt.init_type, opt.init_variance); D.eval()
    g = torch.Generator(device="cuda").manual_seed(21)
    B = 2
    seg = torch.rand(B, 15, 256, 256, device="cuda", generator=g)
    cbn = torch.cat((torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1, seg), 1)
    gy = torch.randn(B, 3, 256, 256, device="cuda", generator=g)
    gprobes = {"fc.weight": lambda n: n.fc.weight, "conv_img.weight": lambda n: n.conv_img.weight,
               "up_3.conv_1.weight_orig": lambda n: n.up_3.conv_1.weight_orig, "head_0.conv_0.weight_orig": lambda n: n.head_0.conv_0.weight_orig,
               "up_1.norm_0.mlp_gamma.weight": lambda n: n.up_1.norm_0.mlp_gamma.weight, "attn.theta.weight_orig": lambda n: n.attn.theta.weight_orig,
               "attn.gamma": lambda n: n.attn.gamma}
    dprobes = {"discriminator_0.model0.0.weight": lambda n: n.discriminator_0.model0[0].weight,
               "discriminator_0.model2.0.0.weight_orig": lambda n: n.discriminator_0.model2[0][0].weight_orig,
               "discriminator_1.model4.0.weight": lambda n: n.discriminator_1.model4[0].weight}

    def run(Gm, Dm, dt):
        Gm.zero_grad(); Dm.zero_grad()
        y = Gm(seg.to(dt), warp_out=cbn.to(dt))
        y.backward(gy.to(dt))
        res = Dm(torch.cat((seg.to(dt), y.detach()), 1))[0]
        torch.autograd.backward([r[-1] for r in res], [torch.ones_like(r[-1]) for r in res])
        out = {"fake_image": y.detach(), "D0_logits": res[0][-1].detach(), "D1_logits": res[1][-1].detach(), "D0_feat2": res[0][2].detach()}
        out.update({"d G." + k: f(Gm).grad.clone() for k, f in gprobes.items()})
        out.update({"d D." + k: f(Dm).grad.clone() for k, f in dprobes.items()})
        return out
    tape = kink_tape.KinkTape() if forced else None
    with kink_tape.install(tape):
        want = run(copy.deepcopy(G).double(), copy.deepcopy(D).double(), torch.float64)
        errs = {}
        for flavour in arms:
            monkeypatch.setattr(ops, "CONV_PRECISION", flavour)
            if tape is not None:
                tape.rewind("replay")
            got = run(G, D, torch.float32)
            errs[flavour] = {k: float((got[k].double() - want[k]).abs().max() / (want[k].abs().max() + 1e-300)) for k in want}
            if tape is not None:
                assert tape.pos == len(tape.masks), (tape.pos, len(tape.masks))
                errs[flavour]["#flips"] = [sum(tape.flips), tape.elements]
    return errs


def test_config3_generator_and_patchgan_against_fp64_copies(monkeypatch):
    """Config 3 as the product runs it.  Default flavour (f16x3): outputs within north_star's 1e-3; the gradients sit behind
    LeakyReLU / ReLU kinks after InstanceNorm / PONO, so they are printed next to the framework-fp32 arm's here and HELD to 1e-3
    in the twin test below, where both evaluations are on the fp64 copy's branch pattern.  bf16: reported."""
    import json
    errs = _config3_vs_fp64(monkeypatch, forced=False)
    print("CFG3_FP64", json.dumps(errs))
    e = errs["f16x3"]
    assert all(e[k] < 1e-3 for k in ("fake_image", "D0_logits", "D1_logits", "D0_feat2")), e
    grads = sorted(v for k, v in e.items() if k.startswith("d "))
    assert grads[-1] < 0.05, e            # (kinks: a sanity bound; the measured comparison is the twin test)
    assert all(v < 1.0 for v in errs["bf16"].values()), errs["bf16"]


def test_config3_every_gradient_on_the_fp64_branch_pattern(monkeypatch):
    """the same comparison with every fp32 arm evaluated on the fp64 copy's branch pattern (LeakyReLU after
    InstanceNorm / PONO-SPADE, SPADE's ReLU, the Attention blocks' max-pools): every probed gradient of the default flavour within
    1e-3, with the framework-fp32 arm beside it."""
    import json
    errs = _config3_vs_fp64(monkeypatch, forced=True, arms=("f16x3", "torch"))
    print("CFG3_FP64_FORCED", json.dumps(errs))
    bad = {k: v for k, v in errs["f16x3"].items() if k != "#flips" and not v < 1e-3}
    assert not bad, (bad, {k: errs["torch"][k] for k in bad}, errs["f16x3"]["#flips"])
