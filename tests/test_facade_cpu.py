"""the reference's OWN callers — `Pix2PixModel.forward(data, 'inference')` ->
`.inference()` (models/pix2pix_model.py:325-337), and `generate_fake`'s consumption of the dict (:303-323) — run with the
drop-in installed, on CPU, in the build container.

What this proves: the plumbing between the unmodified facade and the drop-in — `networks.define_Corr` discovery and the
BaseNetwork assertion (networks/__init__.py:18-26,76-78), construction order / `opt.spade_ic` side effects next to netG
(pix2pix_model.py:122-124), `util.load_network`, the positional call `netCorr(ref_image, None, input_semantics,
ref_semantics, alpha=...)` (:327), the `coor_out` keys, `CBN_intype` concatenation (:328-333), netG consuming `warp_out`,
the `{**generate_out, **coor_out}` merge (:336) — with bit-identical parameters in both models.

What it does NOT prove: kernel arithmetic.  There is no GPU here and the product has no CPU path, so IN THIS TEST ONLY
the module attribute `cocosnet_amd.correspondence.correspondence_hot_path` is replaced by the oracle's restatement of
the same lines (oracle/torch_ref.hot_path, pinned to the reference fixtures by tests/test_oracle_golden.py).  The kernels
behind the real function are held to the same fixtures by tests/test_gpu_parity.py on the GPU.
"""
import argparse
import contextlib
import importlib

import pytest
import torch

from oracle import ref_harness as rh
from oracle import torch_ref as tr

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present")


def _test_opt(train_fields=False, **over):
    """The reference's own parser defaults (options/test_options.py, or train_options.py when the training-only
    fields are needed, + networks.modify_commandline_options) with the README's ADE20k inference flags
    (README.md:45); CPU (`gpu_ids=[]`)."""
    networks = rh.load_reference()
    with rh._cwd(rh.REFERENCE_ROOT):
        if train_fields:
            to = importlib.import_module("options.train_options").TrainOptions()
        else:
            to = importlib.import_module("options.test_options").TestOptions()
        parser = to.initialize(argparse.ArgumentParser())
        parser = networks.modify_commandline_options(parser, bool(train_fields))
        opt, _ = parser.parse_known_args([])
    flags = dict(name="facade_test", dataset_mode="ade20k", gpu_ids=[], use_attention=True, maskmix=True,
                 warp_mask_losstype="direct", PONO=True, PONO_C=True, batchSize=1, isTrain=False, semantic_nc=151,
                 label_nc=150, contain_dontcare_label=True, no_instance=True, crop_size=256, load_size=256,
                 aspect_ratio=1.0)
    flags.update(over)
    for k, v in flags.items():
        setattr(opt, k, v)
    return opt


def _data(B, nc, seed=0, float_labels=False):
    g = torch.Generator().manual_seed(seed)
    if float_labels:      # celebahqedge / deepfashion style: float label maps straight into the networks
        lab = lambda: torch.rand(B, nc, 256, 256, generator=g)
    else:                 # 16x16 constant blocks of class ids (int64: pix2pix_model.py:172-187 scatter_ needs it)
        lab = lambda: torch.randint(0, nc, (B, 1, 16, 16), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3)
    return {"label": lab(), "label_ref": lab(), "image": torch.rand(B, 3, 256, 256, generator=g) * 2 - 1,
            "ref": torch.rand(B, 3, 256, 256, generator=g) * 2 - 1, "self_ref": torch.zeros(B), "path": ["x"] * B}


@contextlib.contextmanager
def _drop_in_installed():
    """install_into_reference for the duration of the block; the reference's class is put back afterwards (other
    tests in this process build the reference's own network)."""
    from cocosnet_amd import correspondence as cc
    networks = rh.load_reference()
    ref_corr = importlib.import_module(networks.__name__ + ".correspondence")
    original = ref_corr.NoVGGCorrespondence
    try:
        yield cc.install_into_reference(networks)
    finally:
        ref_corr.NoVGGCorrespondence = original


def _oracle_hot_path(theta_raw, phi_raw, ref_img, real_img, seg_map, ref_seg_map, cfg, temperature=0.01,
                     detach_flag=False, WTA_scale_weight=1, return_corr=False):
    """The oracle behind the product's hot-path signature (HotPathConfig has the fields torch_ref.hot_path reads)."""
    return tr.hot_path(theta_raw, phi_raw, ref_img, real_img, seg_map, ref_seg_map, cfg, temperature=temperature,
                       detach_flag=detach_flag, WTA_scale_weight=WTA_scale_weight, return_corr=return_corr)


def _build(opt, seed=0):
    p2p = importlib.import_module("models.pix2pix_model")
    torch.manual_seed(seed)
    with rh._cwd(rh.REFERENCE_ROOT), contextlib.redirect_stdout(None):
        return p2p.Pix2PixModel(opt).eval()


@pytest.mark.parametrize("flags", [
    dict(match_kernel=3),                                     # README.md:45 as written (match_kernel defaults to 3)
    dict(match_kernel=1, CBN_intype="warp"),                  # the other CBN_intype branch that reads coor_out
], ids=["ade20k_mk3_warp_mask", "ade20k_mk1_warp"])
def test_pix2pix_inference_through_the_drop_in_equals_the_reference(flags, monkeypatch):
    from cocosnet_amd import correspondence as cc
    rh.load_reference()
    opt = _test_opt(**flags)
    data = _data(1, 151)
    ref_model = _build(opt)
    with torch.no_grad():
        want = ref_model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}, mode="inference")

    with _drop_in_installed() as cls:
        ours = _build(opt)
        net = ours.net["netCorr"]
        assert isinstance(net, cls) and isinstance(net, cc.NoVGGCorrespondence)
        # identical parameters: the drop-in loads the reference's state_dict strictly (names and shapes are the contract)
        net.load_state_dict(ref_model.net["netCorr"].state_dict(), strict=True)
        ours.net["netG"].load_state_dict(ref_model.net["netG"].state_dict(), strict=True)
        monkeypatch.setattr(cc, "correspondence_hot_path", _oracle_hot_path)
        with torch.no_grad():
            got = ours({k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}, mode="inference")

    assert set(got) == set(want), (sorted(got), sorted(want))
    for k in want:
        assert got[k].shape == want[k].shape and got[k].dtype == want[k].dtype, k
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    # same fp32 op sequence up to the hot path, the oracle's restatement from there on: agreement to rounding
    for k in ("warp_out", "warp_mask", "fake_image", "input_semantics", "ref_semantics"):
        assert rel(got[k], want[k]) < 1e-4, (k, rel(got[k], want[k]))


def test_generate_fake_consumes_the_drop_in_dict(monkeypatch):
    """pix2pix_model.py:303-323 with the CelebA-HQ edge training flags (README.md:106): `generate_fake` is called on a
    facade whose VGG feature extractor is stubbed (models/vgg19_conv.pth is not shipped) — the dict merge must carry
    warp_cycle for the loss at :210-216, and gradients must reach netCorr's parameters through netG's consumption of
    warp_out."""
    from cocosnet_amd import correspondence as cc
    rh.load_reference()
    opt = _test_opt(train_fields=True, dataset_mode="celebahqedge", semantic_nc=15, label_nc=15, contain_dontcare_label=False,
                    warp_mask_losstype="none", warp_bilinear=True, adaptor_kernel=4, warp_cycle_w=1.0, match_kernel=1)
    data = _data(1, 15, float_labels=True)
    with _drop_in_installed():
        model = _build(opt)
        ref_like = _build(opt)       # only to show that a second build under the drop-in gives the same parameter names
        assert list(model.state_dict()) == list(ref_like.state_dict())
        monkeypatch.setattr(cc, "correspondence_hot_path", _oracle_hot_path)
        model.vggnet_fix = lambda img, layers, preprocess=True: [img.mean(dim=(1, 2, 3)) for _ in layers]
        opt.isTrain = True           # (the facade was BUILT with isTrain False: no netD, no vgg19_conv.pth needed)
        sem, ref_sem = data["label"].float(), data["label_ref"].float()
        out = model.generate_fake(sem, data["image"], ref_semantics=ref_sem, ref_image=data["ref"], self_ref=data["self_ref"])
        opt.isTrain = False
    for k in ("fake_image", "warp_out", "warp_cycle", "ref_features", "real_features"):
        assert k in out, sorted(out)
    assert out["warp_cycle"].shape == (1, 3, 64, 64) and out["fake_image"].shape == (1, 3, 256, 256)
    (out["fake_image"].mean() + out["warp_cycle"].mean()).backward()
    g = model.net["netCorr"].theta.weight.grad
    assert g is not None and float(g.abs().max()) > 0
