"""match_kernel = 3 — the reference's SHIPPED default (options/base_options.py:70; correspondence.py:276,286,291) — at
BASELINE.json's real sizes, through the C ABI, against torch-fp64 autograd of the reference's own formulation
(F.unfold -> centre -> normalise -> matmul -> /T -> softmax -> matmul: oracle/torch_ref.py, pinned to the reference's
fixtures by tests/test_oracle_golden.py incl. `ade_mk3`).

Round-2 VERDICT, missing 1 / weak 1: match_kernel 3 was only checked up to 33x31 grids.  Here:
  cfg2'  ADE20k 256^2, B = 8, 64x64 grid, Cv = 3 + 151 (direct mask), PONO_C         — ALL 8 samples
  cfg3'  CelebA-HQ edge, B = 16, warp_cycle + two_cycle (row AND column softmax of f) — samples 0, 5, 10, 15
outputs and d theta / d phi, both arithmetic flavours.  The oracle unfolds to K = 2304 in fp64 (77 GFLOP per sample
forward).  Round 4: the oracle's torch ops run in fp64 ON THE DEVICE (oracle/torch_ref.py is
device-agnostic), so every sample is affordable; sample 0 is additionally run on the host cores and the two runs of the
same code must agree to 1e-7 — the arbiter itself is cross-checked.  Computed once per configuration, shared by the flavours.

Also here (VERDICT "weak 1", tolerance kind): an ELEMENTWISE relative check — north_star's "1e-3 relative" read
literally — of `warp_mask` (the loss takes its log, pix2pix_model.py:276) on every entry above 1e-6.
"""
import numpy as np
import pytest
import torch

from oracle import corr_oracle as co
from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu

TOL = 2e-4
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no fallback)")


@pytest.fixture(params=["f16x3", "fp32"])
def precision(request, monkeypatch):
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", request.param)
    monkeypatch.setattr(ops, "PROJ_PRECISION", request.param)
    return request.param


def rel(x, ref, floor=1e-30):
    x = x.detach().double().cpu().numpy() if torch.is_tensor(x) else np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    return float(np.abs(x - ref).max() / (np.abs(ref).max() + floor))


f64 = lambda t: t.detach().double().cpu().numpy()


def _features(B, fh, g, corr=0.35):
    """theta/phi-like features with SPATIAL structure (3x3 neighbourhoods only discriminate when neighbouring
    positions differ): phi = a spatially permuted copy of theta (block-wise, so neighbourhoods survive) + noise."""
    th = torch.randn(B, 256, fh, fh, device=DEV, generator=g)
    blk = fh // 8
    perm = torch.randperm(64, device=DEV, generator=g)
    t = th.reshape(B, 256, 8, blk, 8, blk).permute(0, 1, 2, 4, 3, 5).reshape(B, 256, 64, blk, blk)
    t = t[:, :, perm].reshape(B, 256, 8, 8, blk, blk).permute(0, 1, 2, 4, 3, 5).reshape(B, 256, fh, fh)
    ph = corr * t + torch.randn(B, 256, fh, fh, device=DEV, generator=g)
    return th, ph


@pytest.fixture(scope="module")
def cfg2_mk3():
    """Inputs of cfg2' on the device + the fp64 oracle of samples 0 and 7."""
    B, S, d, nc = 8, 256, 4, 151
    fh = S // d
    g = torch.Generator(device=DEV).manual_seed(23)
    th, ph = _features(B, fh, g)
    ref_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    lab = torch.randint(0, nc, (B, 1, S // 16, S // 16), device=DEV, generator=g)
    lab = lab.repeat_interleave(16, 2).repeat_interleave(16, 3)
    ref_seg = torch.zeros(B, nc, S, S, device=DEV).scatter_(1, lab, 1.0)
    flags = dict(match_kernel=3, PONO_C=True, down=d, isTrain=True, warp_mask_losstype="direct")
    G = {"warp_out": torch.randn(B, 3, S, S, device=DEV, generator=g),
         "warp_mask": torch.randn(B, nc, fh, fh, device=DEV, generator=g)}
    oracle = {}
    for b in range(B):
        sl = slice(b, b + 1)
        oracle[b] = tr.forward_backward(th[sl], ph[sl], ref_img[sl], ref_img[sl], ref_seg[sl], ref_seg[sl],
                                        co.default_opt(**flags), {k: v[sl] for k, v in G.items()}, device=DEV)
    _cross_check_arbiter(oracle[0], tr.forward_backward(f64(th[:1]), f64(ph[:1]), f64(ref_img[:1]), f64(ref_img[:1]), f64(ref_seg[:1]),
                                                        f64(ref_seg[:1]), co.default_opt(**flags), {k: f64(v[:1]) for k, v in G.items()}))
    return dict(th=th, ph=ph, ref_img=ref_img, ref_seg=ref_seg, flags=flags, G=G, oracle=oracle)


def _cross_check_arbiter(on_device, on_host):
    """The fp64 oracle run on the device and on the host cores (same code, two BLAS libraries) must agree far below TOL."""
    (o1, a1, b1), (o2, a2, b2) = on_device, on_host
    for k in o2:
        assert rel(o1[k], o2[k]) < 1e-7, k
    assert rel(a1, a2) < 1e-7 and rel(b1, b2) < 1e-7


def test_config2_match_kernel3_b8_vs_fp64(cfg2_mk3, precision):
    """cfg2': the benchmark shape with the reference's default match_kernel — outputs and d theta / d phi of ALL 8
    samples (the K6 backward's scratch is size-dependent: cocos_box3_logits_bwd_workspace_bytes)."""
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    c = cfg2_mk3
    th, ph = c["th"].clone().requires_grad_(True), c["ph"].clone().requires_grad_(True)
    out = correspondence_hot_path(th, ph, c["ref_img"], c["ref_img"], c["ref_seg"], c["ref_seg"], HotPathConfig(**c["flags"]))
    assert set(out) == {"warp_out", "warp_mask"}
    torch.autograd.backward([out[k] for k in sorted(out)], [c["G"][k] for k in sorted(out)])
    for b, (outs, dth, dph) in c["oracle"].items():
        sl = slice(b, b + 1)
        for k in outs:
            assert rel(out[k][sl], outs[k]) < TOL, (b, k)
        assert rel(th.grad[sl], dth) < TOL, b
        assert rel(ph.grad[sl], dph) < TOL, b
    # inference: same outputs without the autograd graph (the path that must not keep anything it does not need)
    with torch.no_grad():
        o2 = correspondence_hot_path(c["th"], c["ph"], c["ref_img"], None, c["ref_seg"], c["ref_seg"],
                                     HotPathConfig(**{**c["flags"], "isTrain": False}))
    for k in out:
        assert rel(o2[k], f64(out[k])) < 1e-5, k


@pytest.fixture(scope="module")
def cfg3_mk3():
    B, S, d = 16, 256, 4
    fh = S // d
    g = torch.Generator(device=DEV).manual_seed(29)
    th, ph = _features(B, fh, g)
    ref_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    real_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    seg = torch.rand(B, 15, S, S, device=DEV, generator=g)
    ref_seg = torch.rand(B, 15, S, S, device=DEV, generator=g)
    flags = dict(match_kernel=3, PONO_C=True, down=d, warp_bilinear=True, isTrain=True, warp_mask_losstype="none",
                 warp_cycle_w=1.0, two_cycle=True)
    G = {"warp_out": torch.randn(B, 3, S, S, device=DEV, generator=g)}
    for k in ("warp_cycle", "warp_i2r", "warp_i2r2i"):
        G[k] = torch.randn(B, 3, fh, fh, device=DEV, generator=g)
    oracle = {}
    for b in range(B):         # every sample (the mk 1 twin checks all 16, this one checked 4)
        sl = slice(b, b + 1)
        oracle[b] = tr.forward_backward(th[sl], ph[sl], ref_img[sl], real_img[sl], seg[sl], ref_seg[sl],
                                        co.default_opt(**flags), {k: v[sl] for k, v in G.items()}, device=DEV)
    return dict(th=th, ph=ph, ref_img=ref_img, real_img=real_img, seg=seg, ref_seg=ref_seg, flags=flags, G=G,
                oracle=oracle)


def test_config3_match_kernel3_cycle_b16_all_16_samples(cfg3_mk3, precision):
    """cfg3': CelebA training flags with the default match_kernel — row and column softmax of the same box-filtered
    correlation, V differentiated (warp_cycle feeds warp_out back through the column pass)."""
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    c = cfg3_mk3
    th, ph = c["th"].clone().requires_grad_(True), c["ph"].clone().requires_grad_(True)
    out = correspondence_hot_path(th, ph, c["ref_img"], c["real_img"], c["seg"], c["ref_seg"], HotPathConfig(**c["flags"]))
    assert set(out) == {"warp_out", "warp_cycle", "warp_i2r", "warp_i2r2i"}
    torch.autograd.backward([out[k] for k in sorted(out)], [c["G"][k] for k in sorted(out)])
    for b, (outs, dth, dph) in c["oracle"].items():
        sl = slice(b, b + 1)
        for k in outs:
            assert rel(out[k][sl], outs[k]) < TOL, (b, k)
        assert rel(th.grad[sl], dth) < TOL, b
        assert rel(ph.grad[sl], dph) < TOL, b


# ------------------------------------------------------------------ elementwise relative accuracy of warp_mask
def _rel_hist(x, ref, floor):
    """elementwise |x - ref| / |ref| over the entries with |ref| > floor: (max, 99.9th percentile, count)."""
    x, ref = np.asarray(x, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    m = np.abs(ref) > floor
    e = np.abs(x[m] - ref[m]) / np.abs(ref[m])
    return float(e.max()), float(np.percentile(e, 99.9)), int(m.sum())


@pytest.mark.parametrize("mk", [1, 3])
def test_warp_mask_is_elementwise_relative_1e3(mk, precision):
    """north_star: "within 1e-3 relative fp32 tolerance".  Every other test normalises by the tensor's maximum; the
    soft label map `warp_mask` goes through log(. + 1e-10) in the loss (pix2pix_model.py:276), so its SMALL entries
    matter: every entry above 1e-6 must be within 1e-3 of fp64 ELEMENTWISE (it is a sum of non-negative terms
    P[i,j]*onehot[j,c], so no cancellation hides behind the requirement).  ADE20k flags, 64x64 grid, both
    match_kernels."""
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    B, S, d, nc = 2, 256, 4, 151
    fh = S // d
    g = torch.Generator(device=DEV).manual_seed(41 + mk)
    th, ph = _features(B, fh, g, corr=0.5)
    ref_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    lab = torch.randint(0, nc, (B, 1, S // 8, S // 8), device=DEV, generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    ref_seg = torch.zeros(B, nc, S, S, device=DEV).scatter_(1, lab, 1.0)
    flags = dict(match_kernel=mk, PONO_C=True, down=d, isTrain=False, warp_mask_losstype="direct")
    with torch.no_grad():
        out = correspondence_hot_path(th, ph, ref_img, None, ref_seg, ref_seg, HotPathConfig(**flags))
    with torch.no_grad():
        o64 = tr.hot_path(th[:1].double().cpu(), ph[:1].double().cpu(), ref_img[:1].double().cpu(), None,
                          ref_seg[:1].double().cpu(), ref_seg[:1].double().cpu(), co.default_opt(**flags))
    mx, p999, n = _rel_hist(f64(out["warp_mask"][:1]), o64["warp_mask"].numpy(), 1e-6)
    assert n > 1000, n
    assert mx < 1e-3, (mx, p999, n)
    # ... and every entry above 1e-9 within 1e-3 as well (the split flavour's P planes carry >= 11 bits down to 2^-39 of a
    # row maximum since round 3: lo plane scaled by 2^11, common.h split_pair_rtz_lo_scaled; before: 30 % off at 1e-9)
    mx8, _, _ = _rel_hist(f64(out["warp_mask"][:1]), o64["warp_mask"].numpy(), 1e-8)
    assert mx8 < 1e-3, mx8
    # (the floor: a probability below 2^-29 of its row maximum sits in the SUBNORMAL range of the hi plane; measured
    #  1.6e-3 at 1e-9 for match_kernel 3, 2e-4 for match_kernel 1 — profiles/r03_precision_check.jsonl)
    mx9, _, _ = _rel_hist(f64(out["warp_mask"][:1]), o64["warp_mask"].numpy(), 1e-9)
    assert mx9 < 5e-3, mx9
    # the log the loss takes, over ALL entries (also the ones below 1e-9, where only the 1e-10 of the loss keeps the
    # logarithm finite): per-pixel loss term log(mask + 1e-10) within 1e-3 of its own magnitude
    lref = np.log(o64["warp_mask"].numpy() + 1e-10)
    lg = np.abs(np.log(f64(out["warp_mask"][:1]).clip(0) + 1e-10) - lref) / np.maximum(np.abs(lref), 1.0)
    assert float(lg.max()) < 1e-3, float(lg.max())


# ------------------------------------------------------------------ the fused family (K19 / K20) on small 64-wide grids
@pytest.mark.parametrize("fh,flags", [
    (4, dict(warp_mask_losstype="direct")),
    (8, dict(warp_mask_losstype="cycle", warp_cycle_w=1.0, two_cycle=True)),
    (12, dict(warp_mask_losstype="none", warp_cycle_w=1.0, warp_bilinear=True)),
])
def test_box3_fused_family_on_small_grids_vs_fp64_and_vs_the_materialised_chain(fh, flags, monkeypatch):
    """fh x 64 grids (N = 256 .. 768): edge rows everywhere (fh = 4: two of four rows are border rows), row AND column
    softmax, V differentiated (cycle terms: P planes + GEMM), both against torch-fp64 autograd of the unfolded
    formulation and against the round-2 chain (K3 -> K6 -> K7, COCOS_BOX3_FUSED=0) on the same inputs."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    B, d, fw, nc = 2, 4, 64, 11
    g = torch.Generator(device=DEV).manual_seed(100 + fh)
    th = torch.randn(B, 256, fh, fw, device=DEV, generator=g) + 0.15
    ph = 0.4 * th.roll((1, 5), (2, 3)) + torch.randn(B, 256, fh, fw, device=DEV, generator=g) - 0.1
    H, W = fh * d, fw * d
    ref_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    real_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    seg = torch.rand(B, nc, H, W, device=DEV, generator=g)
    ref_seg = torch.rand(B, nc, H, W, device=DEV, generator=g)
    cfg = dict(match_kernel=3, PONO_C=True, down=d, isTrain=True, **flags)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "BOX3_FUSED", fused)
        t, p = th.clone().requires_grad_(True), ph.clone().requires_grad_(True)
        with ops.KernelTimer() as kt:
            out = correspondence_hot_path(t, p, ref_img, real_img, seg, ref_seg, HotPathConfig(**cfg))
            if fused:
                G = {k: torch.randn(v.shape, device=DEV, generator=g) for k, v in sorted(out.items())}
            torch.autograd.backward([out[k] for k in sorted(out)], [G[k] for k in sorted(out)])
        res[fused] = (out, t.grad, p.grad, set(kt.summary()))
    assert "box3_softmax_warp_fwd" in res[True][3] and "box3_adjoint_planes" in res[True][3]
    assert "box3_logits_fwd" in res[False][3] and "box3_softmax_warp_fwd" not in res[False][3]
    outs, dth, dph = tr.forward_backward(f64(th), f64(ph), f64(ref_img), f64(real_img), f64(seg), f64(ref_seg),
                                         co.default_opt(**cfg), {k: f64(v) for k, v in G.items()})
    for fused in (True, False):
        out, gt, gp, _ = res[fused]
        assert set(out) == set(outs)
        for k in outs:
            assert rel(out[k], outs[k]) < TOL, (fused, k)
        assert rel(gt, dth) < TOL, fused
        assert rel(gp, dph) < TOL, fused


# ------------------------------------------------------------------ round 4: the fused family on 128-wide grids (BASELINE config 5)
@pytest.mark.parametrize("fh,flags", [
    (2, dict(warp_mask_losstype="direct")),                                          # every row is a border row
    (4, dict(warp_mask_losstype="cycle", warp_cycle_w=1.0, two_cycle=True)),
    (6, dict(warp_patch=True, warp_cycle_w=1.0, warp_mask_losstype="none")),          # Cv = 48 patches, folded back
    (32, dict(warp_mask_losstype="direct")),       # N = 4096: the last size whose per-key statistics sit in LDS whole
    (40, dict(warp_mask_losstype="direct")),       # N = 5120: statistics in chunks of 2048 keys, the last one partial
])
def test_box3_fused_family_on_128_wide_grids_vs_fp64_and_vs_the_materialised_chain(fh, flags, monkeypatch):
    """fh x 128 grids (N = 256 .. 768): an image row is FOUR 32-position tiles, so the x box of the correlation GEMM's
    epilogue exchanges its halo between the two key chunks of a wave and between neighbouring waves, the y box shifts by
    four tiles, and K20 runs four waves per row pair (box3_common.h, round 4).  Against torch-fp64 autograd of the unfolded
    formulation and against the round-2 chain (COCOS_BOX3_FUSED=0) on the same inputs; row and column softmax, V
    differentiated."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    B, d, fw, nc = 2, 4, 128, 7
    g = torch.Generator(device=DEV).manual_seed(300 + fh)
    th = torch.randn(B, 256, fh, fw, device=DEV, generator=g) + 0.15
    ph = 0.4 * th.roll((1, 37), (2, 3)) + torch.randn(B, 256, fh, fw, device=DEV, generator=g) - 0.1
    H, W = fh * d, fw * d
    ref_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    real_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    seg = torch.rand(B, nc, H, W, device=DEV, generator=g)
    ref_seg = torch.rand(B, nc, H, W, device=DEV, generator=g)
    cfg = dict(match_kernel=3, PONO_C=True, down=d, isTrain=True, **flags)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "BOX3_FUSED", fused)
        t, p = th.clone().requires_grad_(True), ph.clone().requires_grad_(True)
        with ops.KernelTimer() as kt:
            out = correspondence_hot_path(t, p, ref_img, real_img, seg, ref_seg, HotPathConfig(**cfg))
            if fused:
                G = {k: torch.randn(v.shape, device=DEV, generator=g) for k, v in sorted(out.items())}
            torch.autograd.backward([out[k] for k in sorted(out)], [G[k] for k in sorted(out)])
        res[fused] = (out, t.grad, p.grad, set(kt.summary()))
    assert "box3_softmax_warp_fwd" in res[True][3] and "box3_adjoint_planes" in res[True][3]
    assert "box3_logits_fwd" in res[False][3] and "box3_softmax_warp_fwd" not in res[False][3]
    outs, dth, dph = tr.forward_backward(th, ph, ref_img, real_img, seg, ref_seg, co.default_opt(**cfg), G, device=DEV)
    errs = {}
    for fused in (True, False):
        out, gt, gp, _ = res[fused]
        assert set(out) == set(outs)
        errs[fused] = {**{k: rel(out[k], outs[k]) for k in outs}, "d theta": rel(gt, dth), "d phi": rel(gp, dph)}
    print("BOX3_128", fh, errs)
    # outputs at TOL; gradients at 5e-4 (north_star: 1e-3): at N = 5120 BOTH chains — two independent kernel families — sit at
    # the same 2.5e-4 from fp64: the floor is the fp32 evaluation of the statistics' gradients (sums of L * z with z up to
    # +-100 that cancel to first order), not a kernel
    for fused in (True, False):
        assert max(v for k, v in errs[fused].items() if not k.startswith("d ")) < TOL, (fused, errs)
        assert max(v for k, v in errs[fused].items() if k.startswith("d ")) < 5e-4, (fused, errs)


@pytest.mark.parametrize("fh,fw,flags,nc", [
    (8, 64, dict(warp_mask_losstype="cycle", warp_cycle_w=1.0, two_cycle=True), 9),     # rows x 3 (two of them via the sink), columns x 2
    (12, 64, dict(warp_mask_losstype="none", warp_cycle_w=1.0, warp_bilinear=True), 9), # README.md:106 (CelebA-HQ edge) flags
    (4, 128, dict(warp_patch=True, warp_cycle_w=1.0, warp_mask_losstype="none"), 9),    # 128-wide, Cv = 48
    (4, 64, dict(warp_mask_losstype="cycle"), 170),     # 170 label channels: every pass runs in two value chunks on the same sink
])
def test_column_pass_on_the_row_pass_T_and_one_shared_gradient(fh, fw, flags, nc, monkeypatch):
    """Round 4: xbox(C)^T = xbox(C^T), so the column pass of the cycle terms reads the row pass's T transposed
    (COCOS_BOX3_T_TRANSPOSED) and every pass adds its G into one buffer (COCOS_BOX3_G_ACCUMULATE): ONE correlation GEMM, ONE box
    adjoint and ONE pair of GEMMs per step whatever the number of passes.  Same outputs and gradients as with a T per orientation
    and autograd's sum of the G's (the round-3 arrangement, ops.BOX3_SHARE_T = False) — to fp32 summation order."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    B, d = 2, 4
    g = torch.Generator(device=DEV).manual_seed(700 + fh)
    th = torch.randn(B, 256, fh, fw, device=DEV, generator=g) + 0.15
    ph = 0.4 * th.roll((1, 5), (2, 3)) + torch.randn(B, 256, fh, fw, device=DEV, generator=g) - 0.1
    H, W = fh * d, fw * d
    ref_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    real_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    seg = torch.rand(B, nc, H, W, device=DEV, generator=g)
    ref_seg = torch.rand(B, nc, H, W, device=DEV, generator=g)
    cfg = dict(match_kernel=3, PONO_C=True, down=d, isTrain=True, **flags)
    res = {}
    for share in (True, False):
        monkeypatch.setattr(ops, "BOX3_SHARE_T", share)
        t, p = th.clone().requires_grad_(True), ph.clone().requires_grad_(True)
        with ops.KernelTimer() as kt:
            out = correspondence_hot_path(t, p, ref_img, real_img, seg, ref_seg, HotPathConfig(**cfg))
            if share:
                G = {k: torch.randn(v.shape, device=DEV, generator=g) for k, v in sorted(out.items())}
            torch.autograd.backward([out[k] for k in sorted(out)], [G[k] for k in sorted(out)])
        res[share] = (out, t.grad, p.grad, {k: v["calls"] for k, v in kt.summary().items()})
    calls = {s_: res[s_][3] for s_ in res}
    assert calls[True]["box3_corr_xbox"] == 1 and calls[False]["box3_corr_xbox"] == 2, calls
    assert calls[True]["box3_adjoint_planes"] == 1 and calls[False]["box3_adjoint_planes"] == 2, calls
    assert calls[True]["box3_corr_grad"] == 2 and calls[False]["box3_corr_grad"] == 4, calls
    assert calls[True]["box3_softmax_warp_fwd"] == calls[False]["box3_softmax_warp_fwd"] >= (4 if nc > 160 else 2)
    for k in res[True][0]:
        assert rel(res[True][0][k], f64(res[False][0][k])) < 1e-5, k
    # (gradients: both arrangements carry the fp32 noise of the statistics' gradients — sums of L * z that cancel to first
    #  order, 1e-4 of the gradient's range against fp64 in either — so they agree with each other to that, not to an ulp)
    assert rel(res[True][1], f64(res[False][1])) < 2e-4 and rel(res[True][2], f64(res[False][2])) < 2e-4
    outs, dth, dph = tr.forward_backward(th, ph, ref_img, real_img, seg, ref_seg, co.default_opt(**cfg), G, device=DEV)
    for k in outs:
        assert rel(res[True][0][k], outs[k]) < TOL, k
    assert rel(res[True][1], dth) < 5e-4 and rel(res[True][2], dph) < 5e-4


@pytest.mark.parametrize("route", ["512_patch48", "256_stride2_patch12"])
def test_config5_match_kernel3_hw16384_vs_fp64(route, monkeypatch):
    """BASELINE config 5 AS THE REFERENCE RUNS IT: DeepFashion flags (README.md:69,115: --warp_patch --warp_bilinear, no mask
    loss) on the DEFAULT match_kernel 3 (options/base_options.py:70), 128 x 128 grid, HW = 16384 — both routes of SURVEY §8d:
    512^2 input at down 4 (48 patch channels; the reference's F.fold hard-codes 256, the oracle folds to the true size) and
    256^2 input with --warp_stride 2 (12 patch channels).  B = 2, training graph; outputs and d theta / d phi of BOTH samples
    against torch-fp64 autograd of the unfolded K = 2304 formulation, evaluated on the device (16384^2 x 2304 in fp64: 1.2
    TFLOP per sample forward).  The fused family must be what runs: no materialised logits (the K3 -> K6 -> K7 chain would be
    three 1 GiB matrices per sample and orientation)."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    B, fh = 2, 128
    d = 4 if route == "512_patch48" else 2
    S = fh * d
    g = torch.Generator(device=DEV).manual_seed(77)
    th = torch.randn(B, 256, fh, fh, device=DEV, generator=g)
    blk = fh // 8
    perm = torch.randperm(64, device=DEV, generator=g)
    t = th.reshape(B, 256, 8, blk, 8, blk).permute(0, 1, 2, 4, 3, 5).reshape(B, 256, 64, blk, blk)
    t = t[:, :, perm].reshape(B, 256, 8, 8, blk, blk).permute(0, 1, 2, 4, 3, 5).reshape(B, 256, fh, fh)
    ph = 0.35 * t + torch.randn(B, 256, fh, fh, device=DEV, generator=g)
    ref_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    seg = torch.rand(B, 20, S, S, device=DEV, generator=g)
    flags = dict(match_kernel=3, PONO_C=True, down=d, warp_patch=True, warp_bilinear=True, isTrain=True, warp_mask_losstype="none")
    tq, pq = th.clone().requires_grad_(True), ph.clone().requires_grad_(True)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()             # (other tests' module fixtures live on the device: the step's own peak)
    with ops.KernelTimer() as kt:
        out = correspondence_hot_path(tq, pq, ref_img, ref_img, seg, seg, HotPathConfig(**flags))
        G = {"warp_out": torch.randn(out["warp_out"].shape, device=DEV, generator=g)}
        out["warp_out"].backward(G["warp_out"])
    tags = set(kt.summary())
    peak_gib = (torch.cuda.max_memory_allocated() - base) / 2 ** 30
    assert set(out) == {"warp_out"} and out["warp_out"].shape == (B, 3, S, S)
    assert {"box3_corr_xbox", "box3_softmax_warp_fwd", "box3_softmax_warp_bwd", "box3_adjoint_planes"} <= tags, tags
    assert not ({"box3_logits_fwd", "corr_materialize", "logits_softmax_warp_fwd"} & tags), tags
    print("CFG5_MK3", route, "peak GiB", round(peak_gib, 2))
    assert peak_gib < 4.6, peak_gib           # T + G: 2 x 1 GiB per sample + operands — the dC planes live in T's storage (round 5:
                                              # 5.1 -> 4.1 GiB measured; the materialised chain: > 20 GiB)
    for b in range(B):
        sl = slice(b, b + 1)
        outs, dth, dph = tr.forward_backward(th[sl], ph[sl], ref_img[sl], ref_img[sl], seg[sl], seg[sl], co.default_opt(**flags),
                                             {k: v[sl] for k, v in G.items()}, device=DEV)
        assert rel(out["warp_out"][sl], outs["warp_out"]) < TOL, b
        assert rel(tq.grad[sl], dth) < TOL, b
        assert rel(pq.grad[sl], dph) < TOL, b
        torch.cuda.empty_cache()


def test_t_storage_taken_over_by_the_planes_is_visible_to_autograd(monkeypatch):
    """when the dC planes take T's storage (ops.BOX3_ALIAS_T_BYTES; here forced for a small T), the kernels write
    through raw pointers — T's version counter is bumped by hand, so a SECOND backward over a retained graph (T is a saved input of
    every pass) raises instead of reading f16 planes as fp32 logits.  The first backward's gradients equal the un-aliased ones."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    monkeypatch.setattr(ops, "PROJ_PRECISION", "f16x3")
    B, S, d, nc = 1, 256, 4, 5
    fh = S // d
    g = torch.Generator(device=DEV).manual_seed(7)
    th0, ph0 = _features(B, fh, g, corr=0.5)
    ref_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    lab = torch.randint(0, nc, (B, 1, S // 8, S // 8), device=DEV, generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    seg = torch.zeros(B, nc, S, S, device=DEV).scatter_(1, lab, 1.0)
    cfg = HotPathConfig(match_kernel=3, PONO_C=True, down=d, isTrain=True, warp_mask_losstype="direct")
    go, gm = torch.randn(B, 3, S, S, device=DEV, generator=g), torch.randn(B, nc, fh, fh, device=DEV, generator=g)
    grads = {}
    for alias in (False, True):
        monkeypatch.setattr(ops, "BOX3_ALIAS_T_BYTES", 0 if alias else 1 << 62)
        th, ph = th0.clone().requires_grad_(True), ph0.clone().requires_grad_(True)
        out = correspondence_hot_path(th, ph, ref_img, None, seg, seg, cfg)
        torch.autograd.backward([out["warp_out"], out["warp_mask"]], [go, gm], retain_graph=True)
        grads[alias] = (th.grad.clone(), ph.grad.clone())
        if alias:
            with pytest.raises(RuntimeError, match="modified by an inplace operation"):
                torch.autograd.backward([out["warp_out"], out["warp_mask"]], [go, gm])
        else:
            torch.autograd.backward([out["warp_out"], out["warp_mask"]], [go, gm])       # un-aliased: a retained graph runs again
    for a, b in zip(grads[True], grads[False]):
        assert torch.equal(a, b)
