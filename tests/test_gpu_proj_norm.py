"""K23 — the theta / phi projection fused with centre + L2-normalise (proj_norm_f16x3.hip, correspondence.py:272-289) on the GPU
(`-m gpu`), against fp64 restatements:

  * the operand planes it writes (both orientations, both projections of the pair launch) against fp64 conv1x1 -> centre ->
    normalise (oracle.corr_oracle.center_l2norm), for the reference's channel counts (256, 256 + 151 labels), ragged ones
    (K not a multiple of the 16-wide k-step, K < 16) and 1 .. 32 position tiles per sample;
  * the same planes against the ones the unfused chain (K0 `proj1x1` -> K1 `center_l2norm_planes`) writes;
  * through `correspondence_hot_path`: lazy projections (K23) and eager ones (K0 + K1) give the same outputs and the same
    gradients w.r.t. features, weights and biases (the bench configuration's fp64 test runs both: test_gpu_baseline_sizes.py).
"""
import numpy as np
import pytest
import torch

from oracle import corr_oracle as co

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no fallback)")


def _planes_to_f64(hi, lo, scale):
    return (hi.double() + lo.double()).cpu().numpy() / scale


def _case(B, Cin, h, w, seed, bias=True, x_scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x1 = torch.randn(B, Cin, h, w, device=DEV, generator=g) * x_scale
    x2 = (0.3 * x1 + torch.randn(B, Cin, h, w, device=DEV, generator=g)) * x_scale
    w1 = torch.randn(256, Cin, 1, 1, device=DEV, generator=g) / Cin ** 0.5
    w2 = torch.randn(256, Cin, 1, 1, device=DEV, generator=g) / Cin ** 0.5
    b1 = torch.randn(256, device=DEV, generator=g) * 0.1 if bias else None
    b2 = torch.randn(256, device=DEV, generator=g) * 0.1 if bias else None
    return x1, w1, b1, x2, w2, b2


def _ref_qn(x, w, b, center=True):
    x64, w64 = x.double().cpu().numpy(), w.double().cpu().numpy().reshape(w.shape[0], -1)
    B, Cin = x64.shape[:2]
    th = np.einsum("mk,bkn->bmn", w64, x64.reshape(B, Cin, -1))
    if b is not None:
        th = th + b.double().cpu().numpy()[None, :, None]
    if not center:
        return th / (np.linalg.norm(th, axis=1, keepdims=True) + co.EPS)
    return co.center_l2norm(th, True)


@pytest.mark.parametrize("B,Cin,h,w,bias", [(2, 407, 16, 16, True), (1, 256, 8, 16, True), (3, 19, 8, 16, False), (1, 3, 16, 8, True),
                                            (2, 64, 64, 64, True), (1, 417, 4, 32, True)])
def test_k23_planes_match_fp64_projection_centring_normalisation(B, Cin, h, w, bias):
    from cocosnet_amd import ops
    x1, w1, b1, x2, w2, b2 = _case(B, Cin, h, w, seed=Cin + h)
    N = h * w
    planes = ops.OperandPlanes()
    with torch.no_grad():
        qn, kn = ops.proj_center_l2norm_planes_pair(ops.LazyProj1x1(x1, w1, b1), ops.LazyProj1x1(x2, w2, b2), 1, planes, want_chan=True)
    assert qn.shape == (B, 256, N) and kn.shape == (B, 256, N)
    S = ops.SPLIT_OPERAND_SCALE
    for hd, (x, wt, bb) in ((qn, (x1, w1, b1)), (kn, (x2, w2, b2))):
        ref = _ref_qn(x, wt, bb)                                            # [B,256,N], unit columns
        ph, pl = planes.get(hd, True, S)
        ch, cl = planes.get(hd, False, S)
        assert ph.shape == (B, N, 256) and ch.shape == (B, 256, N)
        got_pos = _planes_to_f64(ph, pl, S).transpose(0, 2, 1)
        got_chan = _planes_to_f64(ch, cl, S)
        # entries of unit-norm columns: |.| <= 1; hi + lo carries 22 bits, the projection's own fp32-class rounding ~1e-6
        assert np.abs(got_pos - ref).max() < 4e-6, np.abs(got_pos - ref).max()
        assert np.abs(got_chan - ref).max() < 4e-6, np.abs(got_chan - ref).max()
        assert np.array_equal(got_pos, got_chan)                            # the two orientations hold the same numbers
        # hi is the nearest f16 of the scaled value (|lo| <= half an ulp of hi)
        assert (np.abs(pl.float().cpu().numpy()) <= np.abs(ph.float().cpu().numpy()) * 2.0 ** -10 + 2.0 ** -24).all()


def test_k23_without_channel_planes_and_without_centring():
    """Inference (no channel-major planes) and center_over_channels = 2 (util.feature_normalize semantics)."""
    from cocosnet_amd import ops
    x1, w1, b1, x2, w2, b2 = _case(2, 407, 16, 16, seed=5)
    S = ops.SPLIT_OPERAND_SCALE
    for mode in (1, 2):
        planes = ops.OperandPlanes()
        with torch.no_grad():
            qn, kn = ops.proj_center_l2norm_planes_pair(ops.LazyProj1x1(x1, w1, b1), ops.LazyProj1x1(x2, w2, b2), mode, planes, want_chan=False)
        for hd, (x, wt, bb) in ((qn, (x1, w1, b1)), (kn, (x2, w2, b2))):
            ph, pl = planes.get(hd, True, S)
            ref = _ref_qn(x, wt, bb, center=(mode == 1))
            assert np.abs(_planes_to_f64(ph, pl, S).transpose(0, 2, 1) - ref).max() < 4e-6
            with pytest.raises(Exception):
                planes.get(hd, False, S)                                     # not written: asking for them is an error, not a re-split


@pytest.mark.parametrize("x_scale", [1.0, 3e4, 1e-5])
def test_k23_equals_the_unfused_chain_and_survives_operand_ranges(x_scale):
    """Same planes (to fp32 rounding of the projection) as K0 -> K1, also for features near f16's overflow / far below its normal
    range (the device-side power-of-two scale from max|x|)."""
    from cocosnet_amd import ops
    x1, w1, b1, x2, w2, b2 = _case(2, 407, 32, 32, seed=11, x_scale=x_scale)
    b1, b2 = b1 * x_scale, b2 * x_scale
    S = ops.SPLIT_OPERAND_SCALE
    planes = ops.OperandPlanes()
    with torch.no_grad():
        qn, kn = ops.proj_center_l2norm_planes_pair(ops.LazyProj1x1(x1, w1, b1), ops.LazyProj1x1(x2, w2, b2), 1, planes, want_chan=True)
        planes_u = ops.OperandPlanes()
        qu = ops.center_l2norm_planes(ops.proj1x1(x1, w1, b1).reshape(2, 256, -1), 1, planes_u, want_chan=True)
    a = _planes_to_f64(*planes.get(qn, True, S), S)
    u = _planes_to_f64(*planes_u.get(qu, True, S), S)
    ref = _ref_qn(x1, w1, b1).transpose(0, 2, 1)
    print("K23_VS_CHAIN", x_scale, "fused", np.abs(a - ref).max(), "chain", np.abs(u - ref).max(), "fused vs chain", np.abs(a - u).max())
    assert np.abs(a - ref).max() < 4e-6 and np.abs(a - u).max() < 6e-6


@pytest.mark.parametrize("mk", [1, 3])
def test_hot_path_with_lazy_projections_equals_eager_ones(mk):
    """correspondence_hot_path(LazyProj1x1, LazyProj1x1, ...) == correspondence_hot_path(proj1x1(...), proj1x1(...), ...): outputs and
    the gradients w.r.t. features, weights, biases.  match_kernel 1 takes K23, match_kernel 3 asks for the projections (K0)."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    B, Cin, fh, down, nc = 2, 64 + 7, 16, 4, 7
    fw = 64 if mk == 3 else 16
    x1, w1, b1, x2, w2, b2 = _case(B, Cin, fh, fw, seed=21 + mk)
    g = torch.Generator(device=DEV).manual_seed(99)
    H, W = fh * down, fw * down
    ref_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    lab = torch.randint(0, nc, (B, 1, H, W), device=DEV, generator=g)
    seg = torch.zeros(B, nc, H, W, device=DEV).scatter_(1, lab, 1.0)
    cfg = HotPathConfig(match_kernel=mk, PONO_C=True, down=down, warp_mask_losstype="direct", isTrain=True)
    g_out = torch.randn(B, 3, H, W, device=DEV, generator=g)
    g_mask = torch.randn(B, nc, fh, fw, device=DEV, generator=g)
    res = {}
    for name in ("lazy", "eager"):
        leaves = [t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]
        if name == "lazy":
            th, ph = ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:])
        else:
            th, ph = ops.proj1x1(*leaves[:3]), ops.proj1x1(*leaves[3:])
        with ops.KernelTimer() as kt:
            out = correspondence_hot_path(th, ph, ref_img, ref_img, seg, seg, cfg)
            torch.autograd.backward([out["warp_out"], out["warp_mask"]], [g_out, g_mask])
        tags = set(kt.summary())
        assert ("proj_center_l2norm_fwd" in tags) == (name == "lazy" and mk == 1), (name, mk, sorted(tags))
        res[name] = ([out["warp_out"].detach(), out["warp_mask"].detach()], [t.grad for t in leaves])
    relerr = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    for a, b in zip(res["lazy"][0] + res["lazy"][1], res["eager"][0] + res["eager"][1]):
        assert a.shape == b.shape and relerr(a, b) < 1e-4, relerr(a, b)


def test_hot_path_lazy_detach_flag_and_inference():
    """detach_flag (correspondence.py:292-293): nothing upstream of f receives a gradient, the lazy route included; no-grad
    inference writes no channel-major planes."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    x1, w1, b1, x2, w2, b2 = _case(1, 32, 16, 16, seed=31)
    ref_img = torch.rand(1, 3, 64, 64, device=DEV) * 2 - 1
    seg = torch.zeros(1, 5, 64, 64, device=DEV); seg[:, 2] = 1
    cfg = HotPathConfig(match_kernel=1, PONO_C=True, down=4, warp_mask_losstype="direct")
    leaves = [t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]
    out = correspondence_hot_path(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), ref_img, ref_img, seg, seg, cfg, detach_flag=True)
    assert not out["warp_out"].requires_grad
    with torch.no_grad():
        out2 = correspondence_hot_path(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), ref_img, ref_img, seg, seg, cfg)
    assert torch.equal(out["warp_out"], out2["warp_out"])
    with pytest.raises(TypeError):
        correspondence_hot_path(ops.LazyProj1x1(*leaves[:3]), ops.proj1x1(*leaves[3:]), ref_img, ref_img, seg, seg, cfg)


# ------------------------------------------------------------------ K24: the projections' backward (proj_bwd_f16x3.hip)
def _f64(t):
    return None if t is None else t.detach().double().cpu()


@pytest.mark.parametrize("B,Cin,h,w,bias", [(2, 407, 16, 16, True), (1, 256, 8, 16, True), (2, 19, 8, 16, False), (1, 448, 16, 8, True),
                                            (2, 64, 32, 32, True)])
@pytest.mark.parametrize("fused_bwd", [True, False])
def test_k24_mode0_gradients_of_the_fused_projection_match_fp64(B, Cin, h, w, bias, fused_bwd, monkeypatch):
    """d loss / d (x, weight, bias) through K23's handles — K1's backward + dx = W^T d in K24, d W / d b in the affine weight-gradient
    kernel — against torch fp64 autograd of conv1x1 -> centre -> normalise; `fused_bwd = False`: round 5's chain on the same handles."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PROJ_BWD_FUSED", fused_bwd)
    x1, w1, b1, x2, w2, b2 = _case(B, Cin, h, w, seed=3 * Cin + h)
    N = h * w
    g = torch.Generator(device=DEV).manual_seed(7)
    G1, G2 = torch.randn(B, 256, N, device=DEV, generator=g), torch.randn(B, 256, N, device=DEV, generator=g) * 1e-3
    leaves = [None if t is None else t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]
    planes = ops.OperandPlanes()
    with ops.KernelTimer() as kt:
        qn, kn = ops.proj_center_l2norm_planes_pair(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), 1, planes, want_chan=True)
        torch.autograd.backward([qn, kn], [G1, G2])
    assert ("proj_bwd_input" in kt.summary()) == fused_bwd, sorted(kt.summary())
    for i, (x, wt, bb, G) in enumerate(((x1, w1, b1, G1), (x2, w2, b2, G2))):
        xr, wr = _f64(x).requires_grad_(True), _f64(wt).requires_grad_(True)
        br = None if bb is None else _f64(bb).requires_grad_(True)
        th = torch.einsum("mk,bkn->bmn", wr.reshape(256, Cin), xr.reshape(B, Cin, N))
        if br is not None:
            th = th + br[None, :, None]
        thc = th - th.mean(dim=1, keepdim=True)
        y = thc / (thc.norm(dim=1, keepdim=True) + co.EPS)
        (y * _f64(G)).sum().backward()
        got = leaves[3 * i:3 * i + 3]
        rel = lambda a, r: float((a.detach().double().cpu() - r).abs().max() / (r.abs().max() + 1e-300))
        errs = (rel(got[0].grad, xr.grad), rel(got[1].grad, wr.grad), None if br is None else rel(got[2].grad, br.grad))
        print("K24_MODE0", (B, Cin, h, w), "fused" if fused_bwd else "chain", i, errs)
        assert errs[0] < 2e-5 and errs[1] < 2e-5 and (errs[2] is None or errs[2] < 2e-5), errs


@pytest.mark.parametrize("B,Cin,h,w", [(2, 407, 4, 64), (1, 256, 8, 64), (2, 33, 2, 64)])
@pytest.mark.parametrize("fused_bwd", [True, False])
def test_k24_mode1_projection_plus_unfold_statistics_gradients_match_fp64(B, Cin, h, w, fused_bwd, monkeypatch):
    """ops.proj_unfold3_stats (match_kernel 3): theta_raw, mu, a of a lazy projection as one autograd node; gradients of
    <theta_raw, G> + <mu, gm> + <a, ga> w.r.t. x, weight, bias against torch fp64 (hot_path's torch formulation of K12)."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import _unfold3_stats
    monkeypatch.setattr(ops, "PROJ_BWD_FUSED", fused_bwd)
    x1, w1, b1, _, _, _ = _case(B, Cin, h, w, seed=Cin)
    N, kc = h * w, 256.0 * 9
    g = torch.Generator(device=DEV).manual_seed(17)
    G = torch.randn(B, 256, h, w, device=DEV, generator=g)
    gm, ga = torch.randn(B, N, device=DEV, generator=g), torch.randn(B, N, device=DEV, generator=g) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (x1, w1, b1)]
    with ops.KernelTimer() as kt:
        th, mu, a = ops.proj_unfold3_stats(ops.LazyProj1x1(*leaves), kc)
        torch.autograd.backward([th, mu, a], [G, gm, ga])
    assert ("proj_bwd_input" in kt.summary()) == fused_bwd, sorted(kt.summary())
    xr, wr, br = _f64(x1).requires_grad_(True), _f64(w1).requires_grad_(True), _f64(b1).requires_grad_(True)
    thr = torch.einsum("mk,bkn->bmn", wr.reshape(256, Cin), xr.reshape(B, Cin, N)).reshape(B, 256, h, w) + br[None, :, None, None]
    mur, ar = _unfold3_stats(thr, kc)
    ((thr * _f64(G)).sum() + (mur * _f64(gm)).sum() + (ar * _f64(ga)).sum()).backward()
    rel = lambda t, r: float((t.detach().double().cpu() - r).abs().max() / (r.abs().max() + 1e-300))
    errs = (rel(th, thr.detach()), rel(mu, mur.detach()), rel(a, ar.detach()), rel(leaves[0].grad, xr.grad), rel(leaves[1].grad, wr.grad),
            rel(leaves[2].grad, br.grad))
    print("K24_MODE1", (B, Cin, h, w), "fused" if fused_bwd else "chain", errs)
    assert max(errs[:3]) < 5e-6 and max(errs[3:]) < 2e-5, errs


def test_k24_tiny_and_huge_gradients_keep_their_precision():
    """The per-position power-of-two scale of K24's f16 split: gradients of 1e-9 and of 1e+6 magnitude (no global max|.| pass)."""
    from cocosnet_amd import ops
    B, Cin, h, w = 1, 407, 8, 16
    x1, w1, b1, x2, w2, b2 = _case(B, Cin, h, w, seed=41)
    N = h * w
    g = torch.Generator(device=DEV).manual_seed(3)
    base = torch.randn(B, 256, N, device=DEV, generator=g)
    scale_p = torch.logspace(-9, 6, N, device=DEV).reshape(1, 1, N)          # every position its own magnitude
    for G in (base * 1e-9, base * 1e6, base * scale_p):
        leaves = [t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]
        planes = ops.OperandPlanes()
        qn, kn = ops.proj_center_l2norm_planes_pair(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), 1, planes, want_chan=True)
        torch.autograd.backward([qn, kn], [G, G])
        xr, wr, br = _f64(x1).requires_grad_(True), _f64(w1).requires_grad_(True), _f64(b1).requires_grad_(True)
        th = torch.einsum("mk,bkn->bmn", wr.reshape(256, Cin), xr.reshape(B, Cin, N)) + br[None, :, None]
        thc = th - th.mean(dim=1, keepdim=True)
        (thc / (thc.norm(dim=1, keepdim=True) + co.EPS) * _f64(G)).sum().backward()
        # d x column by column (each position has its own magnitude): relative to the column's own maximum
        got, ref = leaves[0].grad.double().cpu().reshape(B, Cin, N), xr.grad.reshape(B, Cin, N)
        col = ((got - ref).abs().amax(dim=1) / (ref.abs().amax(dim=1) + 1e-300)).max()
        assert float(col) < 3e-5, float(col)
        assert float((leaves[1].grad.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max()) < 3e-5


# ------------------------------------------------------------------ K25: match_kernel 3's projections without normalisation
@pytest.mark.parametrize("B,Cin,h,w,bias,x_scale", [(2, 407, 4, 64, True, 1.0), (1, 256, 8, 64, True, 3e4), (2, 33, 2, 64, False, 1e-5),
                                                    (1, 448, 16, 8, True, 1.0)])
def test_k25_planes_sums_and_statistics_match_fp64(B, Cin, h, w, bias, x_scale):
    """ops.proj_raw_planes_stats_pair: the planes of s * (W x + b) in both orientations (s = the device-side power of two from the
    a-priori bound), and (mu, a) / (nu, b) from the sums of its epilogue, against fp64 — for both projections of the pair launch."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import _unfold3_stats
    x1, w1, b1, x2, w2, b2 = _case(B, Cin, h, w, seed=5 * Cin + w, bias=bias, x_scale=x_scale)
    N, kc = h * w, 256.0 * 9
    leaves = [None if t is None else t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]
    holder = ops.Box3RawPlanes()
    assert ops.proj_raw_fused_ok(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]))
    (th, mu, a), (ph, nu, b) = ops.proj_raw_planes_stats_pair(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), kc, holder)
    for hd, st, (x, wt, bb) in ((th, (mu, a), (x1, w1, b1)), (ph, (nu, b), (x2, w2, b2))):
        y = torch.einsum("mk,bkn->bmn", _f64(wt).reshape(256, Cin), _f64(x).reshape(B, Cin, N))
        if bb is not None:
            y = y + _f64(bb)[None, :, None]
        phh, pll, chh, cll, sc = holder.get(hd)
        s = float(sc)
        assert s > 0 and np.log2(s) == round(np.log2(s))                         # a power of two
        ymax = float(y.abs().max())
        assert 2.0 ** 4 <= ymax * s < 2.0 ** 14, (ymax, s)                        # below the bound, not absurdly far below it
        pos = (phh.double() + pll.double()).cpu() / s
        chan = (chh.double() + cll.double()).cpu() / s
        assert float((pos - y.transpose(1, 2)).abs().max()) < 2.0 ** -20 * ymax
        assert float((chan - y).abs().max()) < 2.0 ** -20 * ymax
        mur, ar = _unfold3_stats(y.reshape(B, 256, h, w), kc)
        rel = lambda t, r: float((t.detach().double().cpu() - r).abs().max() / (r.abs().max() + 1e-300))
        assert rel(st[0], mur) < 5e-6 and rel(st[1], ar) < 5e-6, (rel(st[0], mur), rel(st[1], ar))


@pytest.mark.parametrize("B,Cin,h,w,bias", [(2, 407, 4, 64, True), (1, 256, 8, 64, True), (2, 33, 2, 64, False)])
def test_k25_gradients_match_fp64(B, Cin, h, w, bias):
    """Gradients of <theta_raw, G> + <mu, gm> + <a, ga> (and the same for phi) through the K25 pair node — K12's maps, K24 mode C
    (y read back from the planes), the pair weight gradient with the device-side plane scale — against torch fp64."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import _unfold3_stats
    x1, w1, b1, x2, w2, b2 = _case(B, Cin, h, w, seed=Cin + 1, bias=bias)
    N, kc = h * w, 256.0 * 9
    g = torch.Generator(device=DEV).manual_seed(19)
    Gs = [torch.randn(B, 256, h, w, device=DEV, generator=g), torch.randn(B, 256, h, w, device=DEV, generator=g) * 1e-3]
    gms = [torch.randn(B, N, device=DEV, generator=g) for _ in range(2)]
    gas = [torch.randn(B, N, device=DEV, generator=g) * 0.1 for _ in range(2)]
    leaves = [None if t is None else t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]
    holder = ops.Box3RawPlanes()
    with ops.KernelTimer() as kt:
        (th, mu, a), (ph, nu, b) = ops.proj_raw_planes_stats_pair(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), kc, holder)
        torch.autograd.backward([th, mu, a, ph, nu, b], [Gs[0], gms[0], gas[0], Gs[1], gms[1], gas[1]])
    assert "proj_bwd_input" in kt.summary(), sorted(kt.summary())
    for i, (x, wt, bb) in enumerate(((x1, w1, b1), (x2, w2, b2))):
        xr, wr = _f64(x).requires_grad_(True), _f64(wt).requires_grad_(True)
        br = None if bb is None else _f64(bb).requires_grad_(True)
        thr = torch.einsum("mk,bkn->bmn", wr.reshape(256, Cin), xr.reshape(B, Cin, N)).reshape(B, 256, h, w)
        if br is not None:
            thr = thr + br[None, :, None, None]
        mur, ar = _unfold3_stats(thr, kc)
        ((thr * _f64(Gs[i])).sum() + (mur * _f64(gms[i])).sum() + (ar * _f64(gas[i])).sum()).backward()
        got = leaves[3 * i:3 * i + 3]
        rel = lambda t, r: float((t.detach().double().cpu() - r).abs().max() / (r.abs().max() + 1e-300))
        errs = (rel(got[0].grad, xr.grad), rel(got[1].grad, wr.grad), None if br is None else rel(got[2].grad, br.grad))
        print("K25_GRAD", (B, Cin, h, w), i, errs)
        assert errs[0] < 2e-5 and errs[1] < 2e-5 and (errs[2] is None or errs[2] < 2e-5), errs


def test_k25_hot_path_equals_the_unfused_chain(monkeypatch):
    """match_kernel 3 through correspondence_hot_path with lazy projections: K25 (planes + sums in one launch, no fp32 projection)
    against COCOS_PROJ_RAW_FUSED=0 (K0 + K12 + separate splits) — outputs and all six gradients."""
    from cocosnet_amd import ops
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    B, Cin, fh, fw, down, nc = 2, 256 + 7, 4, 64, 4, 7
    x1, w1, b1, x2, w2, b2 = _case(B, Cin, fh, fw, seed=77)
    g = torch.Generator(device=DEV).manual_seed(5)
    H, W = fh * down, fw * down
    ref_img = torch.rand(B, 3, H, W, device=DEV, generator=g) * 2 - 1
    lab = torch.randint(0, nc, (B, 1, H, W), device=DEV, generator=g)
    seg = torch.zeros(B, nc, H, W, device=DEV).scatter_(1, lab, 1.0)
    cfg = HotPathConfig(match_kernel=3, PONO_C=True, down=down, warp_mask_losstype="direct", isTrain=True)
    g_out, g_mask = torch.randn(B, 3, H, W, device=DEV, generator=g), torch.randn(B, nc, fh, fw, device=DEV, generator=g)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "PROJ_RAW_FUSED", fused)
        leaves = [t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]
        with ops.KernelTimer() as kt:
            out = correspondence_hot_path(ops.LazyProj1x1(*leaves[:3]), ops.LazyProj1x1(*leaves[3:]), ref_img, ref_img, seg, seg, cfg)
            torch.autograd.backward([out["warp_out"], out["warp_mask"]], [g_out, g_mask])
        assert ("unfold3_stats_fwd" in kt.summary()) and (("proj1x1_fwd" in kt.summary())), sorted(kt.summary())
        res[fused] = [out["warp_out"].detach(), out["warp_mask"].detach()] + [t.grad for t in leaves]
    relerr = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    for a, b in zip(res[True], res[False]):
        assert a.shape == b.shape and relerr(a, b) < 1e-4, relerr(a, b)
