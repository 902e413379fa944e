"""GPU parity at BASELINE.json's REAL sizes (run with `-m gpu` on an MI355X), against fp64 oracles.

Round-1 VERDICT, "what's weak" 1-3: the configurations that are benchmarked must also be CHECKED —
  (i)   the exact bench.py step (B=8, HW=4096, Cl=407, Cv=154, gradients of theta/phi only, K0 -> K1 -> K2 and back),
        outputs and every gradient, both arithmetic flavours;
  (ii)  BASELINE config 3 (CelebA-HQ edge, B=16, warp_cycle + two_cycle, bilinear) — first and last sample;
  (iii) BASELINE config 5 (128x128 grid, HW=16384): forward and theta-gradient on sampled query rows, the key side
        against the independent exact-fp32 recompute kernels (the fallback above MAX_DS_WORKSPACE_BYTES), at size;
  (iv)  operand-range robustness of the split flavour: V scaled by 1e5 / 1e-6, a gradient with 1e6 dynamic range;
plus the boundary contract: two Python threads on two streams (the reference's DataParallelWithCallback threading
model) and a stand-in for the callers at pix2pix_model.py:303-337.

Checker: oracle/torch_ref.py (torch CPU autograd in float64, pinned to the reference's fixtures by
tests/test_oracle_golden.py) and oracle/corr_oracle.py.  Tolerances as in test_gpu_parity.py:
max|x - ref| / max|ref| < 2e-4 (north_star: 1e-3).
"""
import threading

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


def rel_elementwise(x, ref, floor_frac=1e-3):
    """(max over the elements with |ref| >= floor_frac * max|ref| of |x - ref| / |ref|,  ||x - ref||_2 / ||ref||_2): the
    two error measures that a max-norm ratio hides."""
    x = x.detach().double().cpu().numpy() if torch.is_tensor(x) else np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    big = np.abs(ref) >= floor_frac * np.abs(ref).max()
    return float((np.abs(x - ref)[big] / np.abs(ref)[big]).max()), float(np.linalg.norm(x - ref) / (np.linalg.norm(ref) + 1e-300))


def dev(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).requires_grad_(grad)


def nrm(x):
    return co.center_l2norm(x, True)


# ------------------------------------------------------------------ (i) the bench.py step itself
@pytest.fixture(scope="module")
def bench_oracle():
    """fp64 oracle of bench.py's step on bench.py's own inputs: theta/phi 1x1 convs -> hot path (ADE20k flags) ->
    loss <warp_out, G_out> + <warp_mask, G_mask>, sample by sample (the path is per-sample; the weight gradients
    are the sum over the batch).  Returns the module state, the inputs and every reference tensor."""
    import bench
    torch.manual_seed(0)
    device = torch.device(DEV)
    model = bench.HotPathStep(1).to(device)
    d = bench.build_inputs(device, "hotpath")
    opt = co.default_opt(match_kernel=1, PONO_C=True, down=bench.DOWN, warp_mask_losstype="direct")
    f64 = lambda t: t.detach().double().cpu().numpy()
    Wt, bt, Wp, bp = (f64(p) for p in (model.theta.weight, model.theta.bias, model.phi.weight, model.phi.bias))
    Wt, Wp = Wt.reshape(Wt.shape[0], -1), Wp.reshape(Wp.shape[0], -1)
    B = d["cont_features"].shape[0]
    fh = d["cont_features"].shape[2]
    ref = dict(warp_out=[], warp_mask=[], dcont=[], dref=[], dWt=0.0, dbt=0.0, dWp=0.0, dbp=0.0)
    for b in range(B):
        xc, xr = f64(d["cont_features"][b]).reshape(-1, fh * fh), f64(d["ref_features"][b]).reshape(-1, fh * fh)
        theta = (Wt @ xc + bt[:, None]).reshape(1, -1, fh, fh)            # :272
        phi = (Wp @ xr + bp[:, None]).reshape(1, -1, fh, fh)              # :282
        sl = slice(b, b + 1)
        outs, dth, dph = tr.forward_backward(
            theta, phi, f64(d["ref_img"][sl]), f64(d["real_img"][sl]), f64(d["seg"][sl]), f64(d["ref_seg"][sl]), opt,
            {"warp_out": f64(d["g_out"][sl]), "warp_mask": f64(d["g_mask"][sl])})
        dth, dph = dth.reshape(-1, fh * fh), dph.reshape(-1, fh * fh)
        ref["warp_out"].append(outs["warp_out"][0]); ref["warp_mask"].append(outs["warp_mask"][0])
        ref["dcont"].append((Wt.T @ dth).reshape(-1, fh, fh)); ref["dref"].append((Wp.T @ dph).reshape(-1, fh, fh))
        ref["dWt"] = ref["dWt"] + dth @ xc.T; ref["dbt"] = ref["dbt"] + dth.sum(1)
        ref["dWp"] = ref["dWp"] + dph @ xr.T; ref["dbp"] = ref["dbp"] + dph.sum(1)
    for k in ("warp_out", "warp_mask", "dcont", "dref"):
        ref[k] = np.stack(ref[k])
    return model, d, ref


def test_bench_configuration_forward_and_all_gradients_vs_fp64(bench_oracle, precision):
    """B=8, HW=4096, Cl=407, Cv=154: the step bench.py times, output for output and gradient for gradient (the
    theta/phi-only backward: saved logits -> query kernel in the blocked layout -> planes GEMM over all 8 samples,
    then K1 and K0 backward).  Round 6: the f16x3 flavour's forward goes through K23 (projection + K1 fused, lazy projections)."""
    from cocosnet_amd import ops
    with ops.KernelTimer() as kt:
        _bench_configuration_vs_fp64(bench_oracle, precision)
    assert ("proj_center_l2norm_fwd" in kt.summary()) == (precision == "f16x3"), sorted(kt.summary())


def test_bench_configuration_unfused_projection_chain_vs_fp64(bench_oracle, monkeypatch):
    """The same step with ops.PROJ_NORM_FUSED = False: K0 (`proj1x1`) -> K1 (`center_l2norm_planes`) in separate launches — the
    round-5 chain stays a tested route (it is what every non-fused back end and the match_kernel-3 path project with)."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PROJ_NORM_FUSED", False)
    with ops.KernelTimer() as kt:
        _bench_configuration_vs_fp64(bench_oracle, "f16x3")
    assert "proj_center_l2norm_fwd" not in kt.summary() and "proj1x1_fwd" in kt.summary(), sorted(kt.summary())


def _bench_configuration_vs_fp64(bench_oracle, precision):
    model, d, ref = bench_oracle
    for p in model.parameters():
        p.grad = None
    for k in ("cont_features", "ref_features"):
        d[k].grad = None
    out = model(d)
    torch.autograd.backward([out["warp_out"], out["warp_mask"]], [d["g_out"], d["g_mask"]])
    assert rel(out["warp_out"], ref["warp_out"]) < TOL
    assert rel(out["warp_mask"], ref["warp_mask"]) < TOL
    assert rel(d["cont_features"].grad, ref["dcont"]) < TOL
    assert rel(d["ref_features"].grad, ref["dref"]) < TOL
    W = model.theta.weight.shape
    assert rel(model.theta.weight.grad, ref["dWt"].reshape(W)) < TOL
    assert rel(model.phi.weight.grad, ref["dWp"].reshape(W)) < TOL
    assert rel(model.theta.bias.grad, ref["dbt"]) < TOL
    assert rel(model.phi.bias.grad, ref["dbp"]) < TOL
    # the same tensors element by element (every element above 1e-2 of the tensor's range: the absolute error is ~3e-6 of the
    # range, so at a floor of 1e-3 the ratio is 3e-3 by construction) and in the 2-norm (measured 2-3e-6)
    pairs = {"warp_out": (out["warp_out"], ref["warp_out"]), "warp_mask": (out["warp_mask"], ref["warp_mask"]),
             "d cont": (d["cont_features"].grad, ref["dcont"]), "d ref": (d["ref_features"].grad, ref["dref"]),
             "d W_theta": (model.theta.weight.grad, ref["dWt"].reshape(W)), "d W_phi": (model.phi.weight.grad, ref["dWp"].reshape(W))}
    ew = {k: rel_elementwise(a, b, 1e-2) for k, (a, b) in pairs.items()}
    print("BENCH_CFG_ELEMENTWISE", precision, ew)
    for k, (mx, l2) in ew.items():
        assert mx < 1e-3 and l2 < 2e-5, (k, mx, l2)


# ------------------------------------------------------------------ (ii) BASELINE config 3: CelebA-HQ edge, B = 16, cycle terms
def test_config3_celeba_cycle_b16_all_samples(precision):
    """README.md:106 flags (--warp_bilinear --warp_cycle_w 1) + two_cycle, 15 float label channels, B=16, 64x64 grid:
    R1 / C1 / R2 passes with V differentiated (P planes + second GEMM); outputs and d theta / d phi of ALL 16 samples
    (the split flavour; the exact-fp32 flavour: samples 0 and 15) against torch-fp64 autograd of the reference
    formulation ("config 3 checks 2 of 16 samples")."""
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    B, S, d = 16, 256, 4
    fh = S // d
    g = torch.Generator(device=DEV).manual_seed(3)
    th = torch.randn(B, 256, fh, fh, device=DEV, generator=g)
    perm = torch.randperm(fh * fh, device=DEV, generator=g)
    ph = (0.25 * th.reshape(B, 256, -1)[:, :, perm].reshape(B, 256, fh, fh)
          + torch.randn(B, 256, fh, fh, device=DEV, generator=g))
    ref_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    real_img = torch.rand(B, 3, S, S, device=DEV, generator=g) * 2 - 1
    seg = torch.rand(B, 15, S, S, device=DEV, generator=g)
    ref_seg = torch.rand(B, 15, S, S, device=DEV, generator=g)
    flags = dict(match_kernel=1, PONO_C=True, down=d, warp_bilinear=True, isTrain=True, warp_mask_losstype="none",
                 warp_cycle_w=1.0, two_cycle=True)
    th.requires_grad_(True); ph.requires_grad_(True)
    out = correspondence_hot_path(th, ph, ref_img, real_img, seg, ref_seg, HotPathConfig(**flags))
    assert set(out) == {"warp_out", "warp_cycle", "warp_i2r", "warp_i2r2i"}
    G = {k: torch.randn(v.shape, device=DEV, generator=g) for k, v in sorted(out.items())}
    torch.autograd.backward([out[k] for k in sorted(out)], [G[k] for k in sorted(out)])
    f64 = lambda t: t.detach().double().cpu().numpy()
    for b in (range(B) if precision == "f16x3" else (0, B - 1)):
        sl = slice(b, b + 1)
        outs, dth, dph = tr.forward_backward(f64(th[sl]), f64(ph[sl]), f64(ref_img[sl]), f64(real_img[sl]), f64(seg[sl]),
                                             f64(ref_seg[sl]), co.default_opt(**flags), {k: f64(G[k][sl]) for k in G})
        for k in outs:
            assert rel(out[k][sl], outs[k]) < TOL, (b, k)
        assert rel(th.grad[sl], dth) < TOL, b
        assert rel(ph.grad[sl], dph) < TOL, b


# ------------------------------------------------------------------ (iii) BASELINE config 5: 128x128 grid, HW = 16384
def _rows_oracle(qn, kn, v, g, idx, inv_t):
    """fp64 out[:, idx] and dqn[:, idx] of ONE sample: both depend on the sampled query rows only."""
    q = qn[:, idx]                                                   # [K, S]
    f = (q.T @ kn) * inv_t                                           # [S, Nk]
    p = co.softmax(f, axis=-1)
    out = (p @ v.T).T                                                # [Cv, S]
    dp = g[:, idx].T @ v                                             # [S, Nk]
    ds = p * (dp - (p * dp).sum(-1, keepdims=True)) * inv_t
    return out, kn @ ds.T                                            # [K, S]


def test_config5_hw16384_sampled_rows_and_key_side_vs_recompute(monkeypatch):
    """DeepFashion --warp_patch at 512x512: 128x128 grid, Cv = 48, logits = 1 GiB per sample (just below the 2^31
    per-sample guards).  Three routes through the same shape, all three in one test (the comparison IS between them):
      split_saved      f16x3, logits saved (ops.MAX_SAVED_LOGITS_BYTES raised): the chain the 64x64 grids take;
      split_recompute  f16x3, the bounded-memory route of round 4 (ops.MAX_SAVED_LOGITS_BYTES lowered to 1 GiB): LSE only,
                       logits recomputed per chunk of keys (ops._corr_bwd_recompute) — HWxHW scratch 2 x 1 GiB instead of
                       2 x 2 GiB (2 x 0.5 GiB with 512 MiB chunks);
      fp32_recompute   exact-fp32 flash-style kernels (the path above MAX_DS_WORKSPACE_BYTES of that flavour).
    Forward and d theta on sampled query rows vs fp64; d phi (a sum over ALL queries) between the routes."""
    from cocosnet_amd import ops
    B, N, Cv = 2, 16384, 48
    g = torch.Generator(device=DEV).manual_seed(5)
    q = torch.randn(B, 256, N, device=DEV, generator=g)
    k = 0.25 * q[:, :, torch.randperm(N, device=DEV, generator=g)] + torch.randn(B, 256, N, device=DEV, generator=g)
    q = q - q.mean(1, keepdim=True); q = q / q.norm(dim=1, keepdim=True)
    k = k - k.mean(1, keepdim=True); k = k / k.norm(dim=1, keepdim=True)
    v = torch.rand(B, Cv, N, device=DEV, generator=g) * 2 - 1
    go = torch.randn(B, Cv, N, device=DEV, generator=g)
    res, peak = {}, {}
    for name, prec, limit, saved in (("split_saved", "f16x3", 16 << 30, 16 << 30), ("split_recompute", "f16x3", 16 << 30, 1 << 30),
                                     ("fp32_recompute", "fp32", 0, 1 << 30)):
        monkeypatch.setattr(ops, "PRECISION", prec)
        monkeypatch.setattr(ops, "MAX_DS_WORKSPACE_BYTES", limit)
        monkeypatch.setattr(ops, "MAX_SAVED_LOGITS_BYTES", saved)
        qq, kk = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        with ops.KernelTimer() as kt:
            o = ops.corr_softmax_warp(qq, kk, v, 100.0)
            o.backward(go)
        tags = set(kt.summary())
        peak[name] = (torch.cuda.max_memory_allocated() - base) / 2 ** 30
        res[name] = (o.detach(), qq.grad, kk.grad, tags)
        del o
    print("CFG5_ROUTES peak GiB above the inputs:", {k_: round(v_, 2) for k_, v_ in peak.items()})
    assert "corr_softmax_warp_recompute" not in res["split_saved"][3] and "corr_softmax_warp_bwd_key_from_ds" in res["split_saved"][3]
    assert "corr_softmax_warp_recompute" in res["split_recompute"][3]      # chunked recompute on the split kernels
    assert "corr_softmax_warp_bwd_key" in res["fp32_recompute"][3]         # flash-style fp32 fallback
    assert peak["split_saved"] > 4.0 and peak["split_recompute"] < 3.0, peak    # 2 x 2 GiB of HWxHW scratch vs 2 x 1 GiB
    idx = np.arange(7, N, 509)
    f64 = lambda t: t.double().cpu().numpy()
    for b in range(B):
        o_ref, dq_ref = _rows_oracle(f64(q[b]), f64(k[b]), f64(v[b]), f64(go[b]), idx, 100.0)
        for name in res:
            assert rel(res[name][0][b][:, idx], o_ref) < TOL, (name, b)
            assert rel(res[name][1][b][:, idx], dq_ref) < TOL, (name, b)
    assert rel(res["split_saved"][2], f64(res["fp32_recompute"][2])) < TOL
    # the same dS'' per element; a chunk's key-side GEMM may split its reduction in two (ops._hgemm_planes): fp32 summation order
    assert rel(res["split_recompute"][2], f64(res["split_saved"][2])) < 3e-6
    assert rel(res["split_recompute"][1], f64(res["split_saved"][1])) < 1e-5     # d qn: summed over the chunks in another order
    assert rel(res["split_saved"][0], f64(res["fp32_recompute"][0])) < 1e-4


@pytest.mark.parametrize("with_dv", [False, True])
def test_chunked_recompute_backward_equals_the_saved_logits_chain(with_dv, monkeypatch):
    """ops._corr_bwd_recompute forced on a small shape with FOUR key chunks (one of them shorter), against the saved-logits
    chain on the same inputs — dqn / dkn / dv — and against fp64.  `with_dv`: V differentiated (P planes + second GEMM per chunk)."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    B, Nq, Nk, Cv = 2, 256, 896, 40
    g = torch.Generator(device=DEV).manual_seed(15)
    q = torch.randn(B, 256, Nq, device=DEV, generator=g)
    k = torch.randn(B, 256, Nk, device=DEV, generator=g)
    k[:, :, :Nq] += 0.3 * q
    q = q / q.norm(dim=1, keepdim=True); k = k / k.norm(dim=1, keepdim=True)
    v = torch.rand(B, Cv, Nk, device=DEV, generator=g) * 2 - 1
    go = torch.randn(B, Cv, Nq, device=DEV, generator=g)
    res = {}
    for name, saved, chunk in (("saved", 1 << 30, 512 << 20), ("recompute", 0, B * Nq * 4 * 256)):     # 896 keys in 4 chunks of 256 (the last: 128)
        monkeypatch.setattr(ops, "MAX_SAVED_LOGITS_BYTES", saved)
        monkeypatch.setattr(ops, "RECOMPUTE_CHUNK_BYTES", chunk)
        qq, kk, vv = q.clone().requires_grad_(True), k.clone().requires_grad_(True), v.clone().requires_grad_(with_dv)
        with ops.KernelTimer() as kt:
            o = ops.corr_softmax_warp(qq, kk, vv, 100.0)
            o.backward(go)
        tags = kt.summary()
        res[name] = (o.detach(), qq.grad, kk.grad, vv.grad, tags)
    assert res["recompute"][4]["corr_softmax_warp_recompute"]["calls"] == 4 and "corr_softmax_warp_recompute" not in res["saved"][4]
    f64 = lambda t: t.double().cpu().numpy()
    assert torch.equal(res["saved"][0], res["recompute"][0])
    assert rel(res["recompute"][1], f64(res["saved"][1])) < 1e-5
    assert rel(res["recompute"][2], f64(res["saved"][2])) < 1e-6
    if with_dv:
        assert rel(res["recompute"][3], f64(res["saved"][3])) < 1e-6
    o_ref = co.corr_softmax_warp(f64(q), f64(k), f64(v), 100.0)
    dq_ref, dk_ref, dv_ref = co.corr_softmax_warp_bwd(f64(q), f64(k), f64(v), f64(go), 100.0)
    assert rel(res["recompute"][0], o_ref) < TOL and rel(res["recompute"][1], dq_ref) < TOL and rel(res["recompute"][2], dk_ref) < TOL
    if with_dv:
        assert rel(res["recompute"][3], dv_ref) < TOL


def test_config5_warp_patch_512_through_the_hot_path():
    """The 512x512 --warp_patch route end to end (F.unfold / F.fold with the TRUE image size — the reference hard-codes
    256 at :321 and fails there): B=1, warp_cycle on, outputs + d theta / d phi vs torch-fp64 autograd."""
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    S, d = 512, 4
    fh = S // d
    g = torch.Generator(device=DEV).manual_seed(9)
    th = torch.randn(1, 256, fh, fh, device=DEV, generator=g)
    ph = 0.25 * th.flip(3) + torch.randn(1, 256, fh, fh, device=DEV, generator=g)
    ref_img = torch.rand(1, 3, S, S, device=DEV, generator=g) * 2 - 1
    seg = torch.rand(1, 20, S, S, device=DEV, generator=g)
    flags = dict(match_kernel=1, PONO_C=True, down=d, warp_patch=True, warp_bilinear=True, isTrain=True,
                 warp_mask_losstype="none", warp_cycle_w=1.0)
    th.requires_grad_(True); ph.requires_grad_(True)
    out = correspondence_hot_path(th, ph, ref_img, ref_img, seg, seg, HotPathConfig(**flags))
    assert out["warp_out"].shape == (1, 3, S, S) and out["warp_cycle"].shape == (1, 3, S, S)
    G = {k: torch.randn(v.shape, device=DEV, generator=g) for k, v in sorted(out.items())}
    torch.autograd.backward([out[k] for k in sorted(out)], [G[k] for k in sorted(out)])
    f64 = lambda t: t.detach().double().cpu().numpy()
    outs, dth, dph = tr.forward_backward(f64(th), f64(ph), f64(ref_img), f64(ref_img), f64(seg), f64(seg),
                                         co.default_opt(**flags), {k: f64(G[k]) for k in G})
    for kk in outs:
        assert rel(out[kk], outs[kk]) < TOL, kk
    assert rel(th.grad, dth) < TOL and rel(ph.grad, dph) < TOL


# ------------------------------------------------------------------ (iv) operand ranges of the split flavour
@pytest.mark.parametrize("vscale", [1e5, 1e-6, 1.0])
def test_split_flavour_is_scale_free_in_v(vscale, monkeypatch):
    """forward() is a general contract: V (and with it dP) may have any magnitude.  The split kernels normalise the
    V planes by a device-side power of two, so |V| ~ 1e5 (beyond f16's 65504) and ~ 1e-6 (f16 subnormals) give the
    same relative accuracy as |V| ~ 1: out, d theta, d phi, dV against fp64."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    rs = np.random.RandomState(17)
    B, Nq, Nk, Cv = 1, 512, 640, 37
    qn, kn = nrm(rs.standard_normal((B, 256, Nq))), nrm(rs.standard_normal((B, 256, Nk)))
    kn[:, :, :300] = nrm(qn[:, :, :300] + 0.08 * rs.standard_normal((B, 256, 300)))
    v = rs.uniform(-1, 1, (B, Cv, Nk)) * vscale
    g = rs.standard_normal((B, Cv, Nq))
    o_ref = co.corr_softmax_warp(qn, kn, v, 100.0)
    dq_ref, dk_ref, dv_ref = co.corr_softmax_warp_bwd(qn, kn, v, g, 100.0)
    for need_v in (False, True):
        q, k, vv = dev(qn, True), dev(kn, True), dev(v, need_v)
        o = ops.corr_softmax_warp(q, k, vv, 100.0)
        o.backward(dev(g))
        assert torch.isfinite(o).all()
        assert rel(o, o_ref) < TOL
        assert rel(q.grad, dq_ref) < TOL and rel(k.grad, dk_ref) < TOL
        if need_v:
            assert rel(vv.grad, dv_ref) < TOL


def test_split_flavour_gradient_with_1e6_dynamic_range(monkeypatch):
    """A gradient whose per-channel magnitudes span 1e6 within one tensor: one power-of-two scale protects the
    maximum; the small channels contribute below the fp32 rounding of the large ones, so the result still agrees with
    fp64 at the usual max-norm tolerance (and nothing overflows or flushes to a wrong value)."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    rs = np.random.RandomState(23)
    B, Nq, Nk, Cv = 1, 384, 512, 64
    qn, kn = nrm(rs.standard_normal((B, 256, Nq))), nrm(rs.standard_normal((B, 256, Nk)))
    v = rs.uniform(-1, 1, (B, Cv, Nk))
    g = rs.standard_normal((B, Cv, Nq)) * (10.0 ** rs.uniform(-6, 0, (1, Cv, 1)))
    dq_ref, dk_ref, _ = co.corr_softmax_warp_bwd(qn, kn, v, g, 100.0)
    q, k = dev(qn, True), dev(kn, True)
    ops.corr_softmax_warp(q, k, dev(v), 100.0).backward(dev(g))
    assert rel(q.grad, dq_ref) < TOL and rel(k.grad, dk_ref) < TOL
    # and with the small channels alone (the scale follows the tensor): full relative accuracy again
    small = np.argsort(np.abs(g).max(axis=(0, 2)))[: Cv // 2]
    g2 = np.zeros_like(g); g2[:, small] = g[:, small]
    dq2, dk2, _ = co.corr_softmax_warp_bwd(qn, kn, v, g2, 100.0)
    q, k = dev(qn, True), dev(kn, True)
    ops.corr_softmax_warp(q, k, dev(v), 100.0).backward(dev(g2))
    assert rel(q.grad, dq2) < TOL and rel(k.grad, dk2) < TOL


def test_rescale_branch_is_forced_and_exact(precision):
    """guide rule: a rare data-dependent branch needs its own test.  One key per query is spiked LATE in the key
    order (tiles 10 / 11 of 12) for two wave's worth of queries: their row maxima jump from ~25 to 100 — far more than
    the lazy-rescale threshold — after O and the running row sum (the ones row of V in the split flavour) have
    accumulated for ten tiles; the other waves of the same workgroups never take the branch."""
    from cocosnet_amd import ops
    rs = np.random.RandomState(31)
    B, Nq, Nk, Cv = 1, 256, 384, 35
    qn, kn = nrm(rs.standard_normal((B, 256, Nq))), nrm(rs.standard_normal((B, 256, Nk)))
    kn[0, :, 352:384] = qn[0, :, 0:32]          # queries 0..31 (workgroup 0, wave 0): exact match in key tile 11
    kn[0, :, 320:352] = qn[0, :, 128:160]       # queries 128..159 (workgroup 1, wave 0): in key tile 10
    v = rs.uniform(-1, 1, (B, Cv, Nk))
    g = rs.standard_normal((B, Cv, Nq))
    o_ref = co.corr_softmax_warp(qn, kn, v, 100.0)
    dq_ref, dk_ref, _ = co.corr_softmax_warp_bwd(qn, kn, v, g, 100.0)
    q, k = dev(qn, True), dev(kn, True)
    o = ops.corr_softmax_warp(q, k, dev(v), 100.0)
    o.backward(dev(g))
    assert rel(o, o_ref) < TOL
    assert rel(q.grad, dq_ref, floor=0.5) < TOL and rel(k.grad, dk_ref, floor=0.5) < TOL


# ------------------------------------------------------------------ boundary: threads + streams, callers
def test_two_threads_two_streams_match_single_threaded_results(precision):
    """SURVEY §8b threading contract: under DataParallelWithCallback each replica's forward runs in its own Python
    thread (here additionally on its own stream), backward in autograd's device thread.  Host-side scratch is
    per-thread / per-call, so two concurrent forward+backward passes give bitwise the single-threaded results."""
    from cocosnet_amd.hot_path import HotPathConfig, correspondence_hot_path
    cfg = HotPathConfig(match_kernel=1, PONO_C=True, down=4, warp_mask_losstype="direct")

    def make(seed):
        g = torch.Generator(device=DEV).manual_seed(seed)
        th = torch.randn(2, 256, 32, 32, device=DEV, generator=g)
        ph = 0.3 * th.flip(2) + torch.randn(2, 256, 32, 32, device=DEV, generator=g)
        img = torch.rand(2, 3, 128, 128, device=DEV, generator=g) * 2 - 1
        lab = torch.randint(0, 9, (2, 1, 128, 128), device=DEV, generator=g)
        seg = torch.zeros(2, 9, 128, 128, device=DEV).scatter_(1, lab, 1.0)
        return th, ph, img, seg, torch.randn(2, 3, 128, 128, device=DEV, generator=g)

    def run(inp, stream=None):
        th, ph, img, seg, gout = inp
        th, ph = th.clone().requires_grad_(True), ph.clone().requires_grad_(True)
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(3):
                th.grad = ph.grad = None
                out = correspondence_hot_path(th, ph, img, img, seg, seg, cfg)
                (out["warp_out"] * gout).sum().add(out["warp_mask"].pow(2).sum()).backward()
            if stream is not None:
                stream.synchronize()
        return out["warp_out"].detach().clone(), th.grad.clone(), ph.grad.clone()

    inputs = [make(1), make(2)]
    torch.cuda.synchronize()
    serial = [run(i) for i in inputs]
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def worker(i):
        try:
            results[i] = run(inputs[i], torch.cuda.Stream())
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    torch.cuda.synchronize()
    assert not errors, errors
    for a, b in zip(serial, results):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


class _StandInCaller:
    """What Pix2PixModel does with the correspondence network (pix2pix_model.py:303-337), without the rest of the
    model: pick CBN_in from coor_out per opt.CBN_intype, merge the dicts, keep the tensors alive for the D step."""

    def __init__(self, net, opt):
        self.net, self.opt, self.alpha = net, opt, 1

    def _cbn_in(self, coor_out, input_semantics):
        if self.opt.CBN_intype == "mask":
            return input_semantics
        if self.opt.CBN_intype == "warp":
            return coor_out["warp_out"]
        return torch.cat((coor_out["warp_out"], input_semantics), dim=1)          # 'warp_mask' (:317-318)

    def generate_fake(self, input_semantics, real_image, ref_semantics, ref_image):
        coor_out = self.net(ref_image, real_image, input_semantics, ref_semantics, alpha=self.alpha)      # :308
        fake = self._cbn_in(coor_out, input_semantics).mean(1, keepdim=True)     # stand-in for netG(...)
        return {**{"fake_image": fake}, **coor_out}                               # :322

    def inference(self, input_semantics, ref_semantics, ref_image):
        coor_out = self.net(ref_image, None, input_semantics, ref_semantics, alpha=self.alpha)           # :327
        fake = self._cbn_in(coor_out, input_semantics).mean(1, keepdim=True)
        return {**{"fake_image": fake}, **coor_out}                               # :336


def test_drop_in_behaves_for_the_reference_callers():
    """a16: the dict contract the callers rely on — `warp_out` concatenable with the label map (same H, W, device,
    dtype, contiguous NCHW), real_img=None at inference, `alpha` accepted, outputs usable after a second forward
    (the trainer keeps `out` across the D step), gradients reach the parameters through CBN_in."""
    from cocosnet_amd import correspondence as cc
    opt = cc.ade20k_options(semantic_nc=7, match_kernel=1, isTrain=True)
    torch.manual_seed(0)
    net = cc.NoVGGCorrespondence(opt).to(DEV)
    net.init_weights(opt.init_type, opt.init_variance)
    caller = _StandInCaller(net, opt)
    g = torch.Generator(device=DEV).manual_seed(4)
    img = torch.rand(2, 3, 64, 64, device=DEV, generator=g) * 2 - 1
    real = torch.rand(2, 3, 64, 64, device=DEV, generator=g) * 2 - 1
    lab = torch.randint(0, 7, (2, 1, 64, 64), device=DEV, generator=g)
    seg = torch.zeros(2, 7, 64, 64, device=DEV).scatter_(1, lab, 1.0)
    seg_before = seg.clone()
    out = caller.generate_fake(seg, real, seg.flip(0), img)
    assert {"fake_image", "warp_out", "warp_mask"} <= set(out)
    assert out["warp_out"].shape == (2, 3, 64, 64) and out["warp_out"].is_contiguous()
    assert out["warp_out"].dtype == torch.float32 and out["warp_out"].device == seg.device
    first = {k: v.detach().clone() for k, v in out.items()}
    out2 = caller.generate_fake(seg, real, seg.flip(0), img)                  # D step: a second forward
    for k in first:                                                            # the kept tensors were not overwritten
        assert torch.equal(out[k].detach(), first[k]), k
    loss = out["fake_image"].pow(2).sum() + torch.log(out["warp_mask"] + 1e-10).mul(seg[:, :, ::4, ::4]).sum()
    loss.backward()                                                            # :276 takes log(warp_mask + 1e-10)
    assert net.theta.weight.grad is not None and float(net.theta.weight.grad.abs().max()) > 0
    assert torch.equal(seg, seg_before)                                        # inputs are not mutated
    del out2
    net.eval()
    opt.isTrain = False
    with torch.no_grad():
        inf = caller.inference(seg, seg.flip(0), img)                          # real_img = None (:327)
    assert set(inf) >= {"fake_image", "warp_out", "warp_mask"} and torch.isfinite(inf["fake_image"]).all()
    for intype in ("mask", "warp"):
        opt.CBN_intype = intype
        with torch.no_grad():
            assert caller.inference(seg, seg.flip(0), img)["fake_image"].shape == (2, 1, 64, 64)


# ------------------------------------------------------------------ §8(f) rank 3: the same op class elsewhere in the model
@pytest.mark.parametrize("B,ch,H,W", [(2, 512, 32, 32), (1, 256, 64, 64), (2, 64, 16, 24)])
def test_attention_block_on_hip_matches_torch_fp64(B, ch, H, W):
    """`Attention.forward` (architecture.py:114-127: theta ch/8, max-pooled phi ch/8 and g ch/2, softmax over HW/4
    keys, gamma * o + x) through ops.softmax_attention — outputs and every gradient vs the torch formulation in fp64."""
    from cocosnet_amd.producers import Attention
    torch.manual_seed(0)
    att = Attention(ch, use_sn=False).to(DEV)
    with torch.no_grad():
        att.gamma.fill_(0.7)
        for m in (att.theta, att.phi):
            m.weight.mul_(3.0)                   # sharper rows than the default initialisation gives
    ref = Attention(ch, use_sn=False).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in att.state_dict().items()})
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, ch, H, W, device=DEV, generator=g)
    gy = torch.randn(B, ch, H, W, device=DEV, generator=g)
    xa = x.clone().requires_grad_(True)
    y = att(xa)
    y.backward(gy)
    xr = x.double().cpu().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gy.double().cpu())
    assert rel(y, yr.detach().numpy()) < TOL
    assert rel(xa.grad, xr.grad.numpy()) < TOL
    for (n, p), (_, q) in zip(att.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad.numpy()) < TOL, n


@pytest.mark.parametrize("B,C,h,w,pono", [(2, 512, 32, 32, True), (2, 256, 16, 16, False), (1, 64, 64, 64, True), (1, 48, 9, 7, False),
                                          (2, 128, 32, 32, True), (1, 300, 20, 13, False), (3, 512, 16, 16, True),
                                          (2, 3, 2, 3, False), (5, 33, 1, 130, True)])
def test_contextual_loss_forward_matches_the_reference_formula(B, C, h, w, pono):
    """`ContextualLoss_forward.forward` (ContextualLoss.py:93-137) on K1 + K22 against the reference's formulation
    in torch fp64: per-sample loss and the gradient w.r.t. the generated features X (the exemplar side Y is detached in
    the caller, pix2pix_model.py get_ctx_loss; checked here too).  The shapes of get_ctx_loss at 256^2 (relu5_1 16x16x512,
    relu4_1 32x32x512, pooled relu3_1 32x32x256, pooled relu2_1 32x32x128) and ragged ones (positions and channels padded to the
    kernel's 128 / 32 inside the op)."""
    from types import SimpleNamespace
    from cocosnet_amd.contextual import ContextualLoss_forward
    g = torch.Generator(device=DEV).manual_seed(C + h)
    Y = torch.randn(B, C, h, w, device=DEV, generator=g) + 0.3
    perm = torch.randperm(h * w, device=DEV, generator=g)
    X = (0.6 * Y.reshape(B, C, -1)[:, :, perm].reshape(B, C, h, w) + torch.randn(B, C, h, w, device=DEV, generator=g))
    from cocosnet_amd import contextual
    mod = ContextualLoss_forward(SimpleNamespace(PONO=pono))
    xa, ya = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
    contextual.ROUTE = "fused"            # K22 itself (the "auto" default would take K3 + K15 at these sizes)
    try:
        loss = mod(xa, ya, h=0.1)
        assert loss.shape == (B,)
        loss.sum().backward()
    finally:
        contextual.ROUTE = "auto"
    from oracle import contextual_ref as cr         # the arbiter: oracle/contextual_ref.py (ContextualLoss.py:93-137 in fp64), not the product
    xr, yr = X.double().cpu().requires_grad_(True), Y.double().cpu().requires_grad_(True)
    lr = cr.contextual_loss(xr, yr, h=0.1, pono=pono)
    lr.sum().backward()
    errs = (rel(loss, lr.detach().numpy()), rel(xa.grad, xr.grad.numpy()), rel(ya.grad, yr.grad.numpy()))
    print("CTX_FP64", (B, C, h, w, pono), errs)
    assert errs[0] < 2e-5 and errs[1] < 5e-5 and errs[2] < 5e-5, errs      # (measured 1e-8 .. 4e-6; VERDICT r3 weak 1c: was 1e-3 / 2e-3)
    # round 2's K3 + K15 on the materialised cosine matrix — what ROUTE = "auto" (the default) takes at these sizes: same contract
    from cocosnet_amd import ops
    assert contextual.ROUTE == "auto" and contextual._route(B, h * w) == "materialised"
    xm = X.clone().requires_grad_(True)
    with ops.KernelTimer() as kt:
        lm = mod(xm, Y, h=0.1)
        lm.sum().backward()
    assert "contextual_cx_fwd" not in kt.summary() and "contextual_rows_fwd" in kt.summary(), sorted(kt.summary())
    assert rel(lm, lr.detach().numpy()) < 2e-5 and rel(xm.grad, xr.grad.numpy()) < 5e-5


@pytest.mark.parametrize("name", ["contextual_pono", "contextual_nopono", "contextual_h05"])
@pytest.mark.parametrize("route", ["fused", "auto"])
def test_contextual_loss_matches_the_reference_generated_fixtures(name, route, monkeypatch):
    """tests/golden/contextual_*.npz: per-sample loss and d loss / d X of the REFERENCE's own ContextualLoss_forward (fp32 on CPU,
    oracle/make_contextual_golden.py) — the drop-in class on the GPU (K1 + K22) against them.  The fixtures are fp32 evaluations of a
    formulation whose exponent is (1 - d / (d_min + 1e-3)) / h: their own rounding is ~1e-5, which is the tolerance."""
    import os
    from types import SimpleNamespace
    from cocosnet_amd import contextual
    from cocosnet_amd.contextual import ContextualLoss_forward
    monkeypatch.setattr(contextual, "ROUTE", route)          # K22 itself, and whatever the default picks for the fixture's size
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    x = torch.from_numpy(f["X"]).to(DEV).requires_grad_(True)
    with_ops = __import__("cocosnet_amd.ops", fromlist=["ops"])
    with with_ops.KernelTimer() as kt:
        loss = ContextualLoss_forward(SimpleNamespace(PONO=bool(f["pono"])))(x, torch.from_numpy(f["Y"]).to(DEV), h=float(f["h"]))
        loss.sum().backward()
    if route == "fused":
        assert "contextual_cx_fwd" in kt.summary() and "contextual_cx_bwd" in kt.summary()
    assert rel(loss, f["loss"].astype(np.float64)) < 3e-5, name
    assert rel(x.grad, f["dX"].astype(np.float64)) < 5e-5, name


def _ctx_case(B, C, N, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    Y = torch.randn(B, C, N, device=DEV, generator=g) + 0.3
    perm = torch.randperm(N, device=DEV, generator=g)
    X = 0.6 * Y[:, :, perm] + torch.randn(B, C, N, device=DEV, generator=g)
    return X, Y


def test_contextual_loss_at_4096_positions_and_512_channels(monkeypatch):
    """VERDICT r4 item 4 ("done when"): N = 4096, C = 512 (relu4_1 at 512^2) — the per-sample loss and BOTH gradients against the
    reference's formulation in fp64 (on the device), with the framework's matmul and the round-2 materialised route (K3 + K15)
    poisoned: K22 is the only thing that may run.  Peak memory of the whole forward + backward (centring, normalisation, operand
    planes, both gradients) is a fixed number of FEATURE-sized tensors — measured 7.6, bounded at 10 — with no term in N^2 (at this
    shape a feature tensor is C / N = 1/8 of an [N, N] matrix; the round-2 route held two such matrices on top, the reference's
    formulation five; test_contextual_cx_beyond_the_old_4096_key_cap shows the same at N = 12288, where [N, N] is 576 MiB)."""
    from types import SimpleNamespace
    from cocosnet_amd import ops
    from cocosnet_amd.contextual import ContextualLoss_forward

    def poisoned(*a, **k):
        raise AssertionError("the contextual loss must not materialise a cosine matrix")
    monkeypatch.setattr(ops, "corr_materialize", poisoned)
    monkeypatch.setattr(ops, "contextual_rows", poisoned)
    from cocosnet_amd import contextual
    monkeypatch.setattr(contextual, "ROUTE", "fused")        # ("auto" takes K3 + K15 while the matrix is <= 1 GiB: 128 MiB here)
    B, C, N = 2, 512, 4096
    X, Y = _ctx_case(B, C, N, 77)
    X, Y = X.reshape(B, C, 64, 64), Y.reshape(B, C, 64, 64)
    mod = ContextualLoss_forward(SimpleNamespace(PONO=True))
    from oracle import contextual_ref as cr
    xr, yr = X.double().requires_grad_(True), Y.double().requires_grad_(True)
    lr = cr.contextual_loss(xr, yr, h=0.1, pono=True)      # oracle/contextual_ref.py in fp64, on the device (the checker, not the product)
    lr.sum().backward()
    want = (lr.detach().cpu().numpy(), xr.grad.cpu().numpy(), yr.grad.cpu().numpy())
    del xr, yr, lr
    torch.cuda.empty_cache()
    xa, ya = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with ops.KernelTimer() as kt:
        loss = mod(xa, ya, h=0.1)
        loss.sum().backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert {"contextual_cx_fwd", "contextual_cx_bwd"} <= set(kt.summary())
    errs = (rel(loss, want[0]), rel(xa.grad, want[1]), rel(ya.grad, want[2]))
    feat = B * C * N * 4
    print("CTX_FP64_4096", errs, "peak MiB", peak >> 20, "= %.1f feature tensors;" % (peak / feat), "[B,N,N] fp32 MiB", (B * N * N * 4) >> 20,
          {k: round(v["total_ms"], 3) for k, v in kt.summary().items()})
    assert errs[0] < 2e-5 and errs[1] < 5e-5 and errs[2] < 5e-5, errs
    if not torch.are_deterministic_algorithms_enabled():      # (the COCOS_POISON_EMPTY=1 run: deterministic framework ops take other workspaces)
        assert peak < 10 * feat, (peak, feat)


@pytest.mark.parametrize("B,C,Nq,Nk,h", [(2, 40, 200, 330, 0.1), (1, 64, 513, 129, 0.5), (3, 16, 64, 1000, 0.05)])
def test_contextual_cx_rectangular_and_other_bandwidths(B, C, Nq, Nk, h):
    """ops.contextual_cx on its own: Nq != Nk (the class always passes equal sizes), ragged against the 128-position tiles on
    either side, other bandwidths h; cx and BOTH gradients against the reference's formulation in fp64."""
    from cocosnet_amd import ops
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + Nq)
    nrm = lambda t: t / (t.norm(dim=1, keepdim=True) + 2.2e-16)
    Xn, Yn = nrm(torch.randn(B, C, Nq, device=DEV, generator=g)), nrm(torch.randn(B, C, Nk, device=DEV, generator=g))
    G = torch.randn(B, Nq, device=DEV, generator=g)
    from oracle import contextual_ref as cr
    xr, yr = Xn.double().requires_grad_(True), Yn.double().requires_grad_(True)
    cx_ref = cr.cx_rows(xr, yr, h, 1e-3)
    (cx_ref * G.double()).sum().backward()
    xa, ya = Xn.clone().requires_grad_(True), Yn.clone().requires_grad_(True)
    cx = ops.contextual_cx(xa, ya, h, 1e-3)
    (cx * G).sum().backward()
    assert rel(cx, cx_ref.detach().cpu().numpy()) < 2e-5
    assert rel(xa.grad, xr.grad.cpu().numpy()) < 5e-5 and rel(ya.grad, yr.grad.cpu().numpy()) < 5e-5


def test_contextual_cx_with_256_row_workgroups():
    """K22's forward takes 256-row workgroups when 128-row blocks would be two rounds of the chip (B N / 128 >= 512): B = 16,
    N = 4096, C = 32 — cx and d X against the reference's formulation in fp64 on the device."""
    from cocosnet_amd import ops
    B, C, N = 16, 32, 4096
    X, Y = _ctx_case(B, C, N, 15)
    nrm = lambda t: t / (t.norm(dim=1, keepdim=True) + 2.2e-16)
    Xn, Yn = nrm(X), nrm(Y)
    G = torch.randn(B, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    want = []
    for b in range(0, B, 4):        # (four samples at a time: [4, N, N] fp64 matrices)
        from oracle import contextual_ref as cr
        xr = Xn[b:b + 4].double().requires_grad_(True)
        cx_ref = cr.cx_rows(xr, Yn[b:b + 4].double(), 0.1, 1e-3)
        (cx_ref * G[b:b + 4].double()).sum().backward()
        want.append((cx_ref.detach().cpu().numpy(), xr.grad.cpu().numpy()))
        del cx_ref
    xa = Xn.clone().requires_grad_(True)
    cx = ops.contextual_cx(xa, Yn, 0.1, 1e-3)
    (cx * G).sum().backward()
    cx_w, dx_w = np.concatenate([w[0] for w in want]), np.concatenate([w[1] for w in want])
    assert rel(cx, cx_w) < 2e-5 and rel(xa.grad, dx_w) < 5e-5, (rel(cx, cx_w), rel(xa.grad, dx_w))


def test_contextual_cx_beyond_the_old_4096_key_cap():
    """Round 2's K15 stopped at 4096 keys and the class fell through to framework ops above.  N = 12288
    positions (a 96 x 128 feature map), C = 64: cx and the gradient w.r.t. X against fp64; extra memory of the op (planes,
    statistics, the gradient) stays three orders of magnitude below the 576 MiB of ONE [N, N] fp32 matrix."""
    from cocosnet_amd import ops
    B, C, N = 1, 64, 12288
    X, Y = _ctx_case(B, C, N, 5)
    nrm = lambda t: t / (t.norm(dim=1, keepdim=True) + 2.2e-16)
    Xn, Yn = nrm(X - Y.mean(1, keepdim=True)), nrm(Y - Y.mean(1, keepdim=True))
    from oracle import contextual_ref as cr
    xr = Xn.double().requires_grad_(True)
    cx_ref = cr.cx_rows(xr, Yn.double(), 0.1, 1e-3)
    G = torch.randn(B, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    (cx_ref * G.double()).sum().backward()
    want = (cx_ref.detach().cpu().numpy(), xr.grad.cpu().numpy())
    del cx_ref
    torch.cuda.empty_cache()
    xa = Xn.clone().requires_grad_(True)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    cx = ops.contextual_cx(xa, Yn, 0.1, 1e-3)
    (cx * G).sum().backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    errs = (rel(cx, want[0]), rel(xa.grad, want[1]))
    print("CTX_FP64_12288", errs, "peak MiB", peak >> 20)
    assert errs[0] < 2e-5 and errs[1] < 5e-5, errs
    assert peak < 64 << 20, peak


# ------------------------------------------------------------------ RCCL is at least initialised on the box
def test_rccl_world_size_one_smoke():
    """backend "nccl" IS RCCL on ROCm.  The driver owns the multi-GPU runs; here the same code path (process-group init
    from the environment, an all-reduce on a CUDA tensor, the bucketed gradient exchange with its hooks, Sync-BN
    statistics) runs once on a world of one, so that RCCL initialisation and the stream hand-over are exercised."""
    import os
    import socket
    import torch.distributed as dist
    from cocosnet_amd import dist as cdist
    from cocosnet_amd import trainer as ctr
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        if not dist.is_initialized():
            dist.init_process_group(backend="nccl", rank=0, world_size=1)
        t = torch.arange(8, device=DEV, dtype=torch.float32)
        dist.all_reduce(t)                                         # RCCL kernel on one rank: identity
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), cdist.SyncBatchNorm2d(4)).to(DEV)
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        buckets = ctr.attach_gradient_exchange(opt, bucket_bytes=64)
        ctr.broadcast_parameters(net)
        x = torch.randn(4, 3, 5, 5, device=DEV)
        ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4)).to(DEV)
        ref.load_state_dict(net.state_dict())
        opt.zero_grad()
        net(x).pow(2).mean().backward()
        ref(x).pow(2).mean().backward()
        opt.step()
        assert sum(buckets._launched) == len(buckets.buckets)      # every bucket left from a hook
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(a.grad, b.grad, atol=1e-5)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("labels_exact", [True, False])
def test_value_lo_plane_skipping_is_exact(labels_exact):
    """V = [3 image channels | one-hot labels]: the label blocks' f16 lo plane is all zero and the split kernels skip it
    (cocos_f16_plane_block_mask -> v_lo_mask); with soft labels the general path runs.  Both must match the fp64 oracle,
    forward and theta/phi gradients, and the mask must say what the data is."""
    from cocosnet_amd import ops
    from oracle import torch_ref as tr
    B, N, Cv = 2, 1024, 3 + 70
    g = torch.Generator(device="cuda").manual_seed(21)
    nrm = lambda x: (x - x.mean(1, keepdim=True)) / (x - x.mean(1, keepdim=True)).norm(dim=1, keepdim=True)
    q = nrm(torch.randn(B, 256, N, device="cuda", generator=g)).requires_grad_(True)
    k = nrm(0.3 * q.detach() + torch.randn(B, 256, N, device="cuda", generator=g)).requires_grad_(True)
    img = torch.rand(B, 3, N, device="cuda", generator=g) * 2 - 1
    lab = torch.randint(0, 70, (B, 1, N), device="cuda", generator=g)
    onehot = torch.zeros(B, 70, N, device="cuda").scatter_(1, lab, 1.0)
    if not labels_exact:
        onehot = onehot * 0.9 + 0.1 * torch.rand(B, 70, N, device="cuda", generator=g)
    v = torch.cat([img, onehot], 1).contiguous()
    # the mask itself
    vh, vl, _ = ops.split_f16(v, False, amax=ops.absmax(v))
    mask = int(ops.f16_plane_block_mask(vl).view(torch.int32).item())
    assert (mask & 1) == 1 and ((mask & ~1) == 0) == labels_exact
    prev = ops.PRECISION
    ops.PRECISION = "f16x3"
    try:
        out = ops.corr_softmax_warp(q, k, v, 100.0)
        go = torch.randn(out.shape, device="cuda", generator=g)
        out.backward(go)
    finally:
        ops.PRECISION = prev
    qd, kd = q.detach().double().requires_grad_(True), k.detach().double().requires_grad_(True)
    p = torch.softmax(torch.einsum("bkq,bkj->bqj", qd, kd) * 100.0, dim=2)
    ref = torch.einsum("bqj,bcj->bcq", p, v.double())
    ref.backward(go.double())
    for a, r, what in ((out, ref, "out"), (q.grad, qd.grad, "dq"), (k.grad, kd.grad, "dk")):
        err = (a.double() - r.detach()).abs().max().item()
        assert err <= 2e-4 * r.abs().max().item(), f"{what}: {err:.3e}"


def test_softmax_attention_takes_the_fused_kernels_for_small_k_and_any_operand_magnitude(monkeypatch):
    """VERDICT r2 missing 3 / ADVICE r2 (low): ops.softmax_attention with K = 32 (the reference's Attention block:
    architecture.py:114-127, K = ch/8) must not materialise [B,Nq,Nk]: it runs on the fused K2 kernels with K zero-padded
    to 256 and DEVICE-side operand scales — q, k of magnitude 300 (x16 would overflow f16) and 1e-3 give the same
    accuracy as O(1) operands.  Outputs and all three gradients vs fp64; the materialised route on the same inputs too."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    rs = np.random.RandomState(5)
    B, K, Nq, Nk, Cv = 2, 32, 1024, 512, 70
    for qmag, kmag in ((1.0, 1.0), (300.0, 0.02), (1e-3, 2e3)):
        q = rs.standard_normal((B, K, Nq)) * qmag
        k = rs.standard_normal((B, K, Nk)) * kmag
        v = rs.uniform(-1, 1, (B, Cv, Nk))
        g = rs.standard_normal((B, Cv, Nq))
        f = np.einsum("bci,bcj->bij", q, k)
        p = co.softmax(f)
        o_ref = np.einsum("bij,bcj->bci", p, v)
        dp = np.einsum("bci,bcj->bij", g, v)
        ds = p * (dp - (p * dp).sum(-1, keepdims=True))
        dq_ref, dk_ref, dv_ref = np.einsum("bij,bcj->bci", ds, k), np.einsum("bij,bci->bcj", ds, q), np.einsum("bci,bij->bcj", g, p)
        for fused in (True, False):
            monkeypatch.setattr(ops, "ATTENTION_FUSED", fused)
            qd, kd, vd = dev(q, True), dev(k, True), dev(v, True)
            with ops.KernelTimer() as kt:
                o = ops.softmax_attention(qd, kd, vd, 1.0)
                o.backward(dev(g))
            tags = set(kt.summary())
            assert ("corr_softmax_warp_fwd" in tags) == fused and ("row_softmax_fwd" in tags) == (not fused)
            assert torch.isfinite(o).all()
            assert rel(o, o_ref) < TOL, (qmag, fused)
            assert rel(qd.grad, dq_ref) < TOL and rel(kd.grad, dk_ref) < TOL and rel(vd.grad, dv_ref) < TOL, (qmag, fused)


@pytest.mark.parametrize("K,Nq,Nk", [(8, 1024, 256), (32, 2048, 384), (48, 1024, 256), (64, 1024, 128), (100, 512, 256),
                                     (128, 512, 256), (256, 512, 128), (32, 1000, 200), (64, 520, 36)])
def test_softmax_attention_channel_counts_and_split_reductions(K, Nq, Nk, monkeypatch):
    """Round 4: the fused attention op for every channel-count class — K <= 32 / K <= 64 (the kernels' instantiations without the
    matrix steps, fragment reads and fetches of the zero padding; d q's padding rows are never written), 64 < K < 256 (general
    instantiation on padded planes), K = 256 — on blocked shapes (Nk % 128 == 0, Nq % 32 == 0: the two backward GEMMs have
    too few output tiles at B = 1 and split their reduction, ops._hgemm_planes) and on ragged ones.  Outputs and all three
    gradients vs fp64."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    rs = np.random.RandomState(K + Nq)
    B, Cv = 1, 70
    q = rs.standard_normal((B, K, Nq)) * 1.5
    k = rs.standard_normal((B, K, Nk))
    v = rs.uniform(-1, 1, (B, Cv, Nk))
    g = rs.standard_normal((B, Cv, Nq))
    sc = 1.0 / np.sqrt(K)
    f = np.einsum("bci,bcj->bij", q, k) * sc
    p = co.softmax(f)
    o_ref = np.einsum("bij,bcj->bci", p, v)
    dp = np.einsum("bci,bcj->bij", g, v)
    ds = p * (dp - (p * dp).sum(-1, keepdims=True)) * sc
    dq_ref, dk_ref, dv_ref = np.einsum("bij,bcj->bci", ds, k), np.einsum("bij,bci->bcj", ds, q), np.einsum("bci,bij->bcj", g, p)
    qd, kd, vd = dev(q, True), dev(k, True), dev(v, True)
    with ops.KernelTimer() as kt:
        o = ops.softmax_attention(qd, kd, vd, float(sc))
        o.backward(dev(g))
    if Nk % 8 == 0:      # (else: the split backward does not take the shape — the materialised family, same contract)
        assert "corr_softmax_warp_fwd" in kt.summary() and "corr_softmax_warp_bwd_query" in kt.summary()
    assert qd.grad.shape == qd.shape and kd.grad.shape == kd.shape
    assert rel(o, o_ref) < TOL
    assert rel(qd.grad, dq_ref) < TOL and rel(kd.grad, dk_ref) < TOL and rel(vd.grad, dv_ref) < TOL


@pytest.mark.parametrize("K", [32, 128, 256])
def test_softmax_attention_with_only_v_differentiated(K, monkeypatch):
    """an Attention block whose theta / phi are frozen or detached differentiates V alone.  With K < 256 the
    forward used to run without saved logits and the backward fell through to the exact-fp32 kernels, which are K = 256 only
    ('fused path needs K == 256').  Now V's requires_grad keeps the logits: d v comes from the split chain's P planes; vs fp64."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    rs = np.random.RandomState(K)
    B, Nq, Nk, Cv = 2, 512, 256, 40
    q, k = rs.standard_normal((B, K, Nq)), rs.standard_normal((B, K, Nk))
    v, g = rs.uniform(-1, 1, (B, Cv, Nk)), rs.standard_normal((B, Cv, Nq))
    sc = 1.0 / np.sqrt(K)
    p = co.softmax(np.einsum("bci,bcj->bij", q, k) * sc)
    o_ref, dv_ref = np.einsum("bij,bcj->bci", p, v), np.einsum("bci,bij->bcj", g, p)
    qd, kd, vd = dev(q), dev(k), dev(v, True)
    o = ops.softmax_attention(qd, kd, vd, float(sc))
    o.backward(dev(g))
    assert rel(o, o_ref) < TOL and rel(vd.grad, dv_ref) < TOL
    assert qd.grad is None and kd.grad is None


@pytest.mark.parametrize("K", [32, 256])
def test_softmax_attention_chunked_recompute_backward_of_the_magnitude_free_flavour(K, monkeypatch):
    """the chunked RECOMPUTE backward (no saved logits: ops._corr_bwd_recompute) was only tested for unit-norm
    operands.  An Attention block whose logits exceed MAX_SAVED_LOGITS_BYTES takes it in the magnitude-free flavour: per-chunk
    reference maxima rebuilt by a chunk-local forward, combined with the full forward's row statistics, K-active rows of d q
    sliced.  MAX_SAVED_LOGITS_BYTES = 0 and 128-key chunks: against the saved-logits chain on the same inputs and against fp64."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    rs = np.random.RandomState(100 + K)
    B, Nq, Nk, Cv = 2, 512, 384, 40
    q, k = rs.standard_normal((B, K, Nq)) * 3.0, rs.standard_normal((B, K, Nk)) * 2.0
    v, g = rs.uniform(-1, 1, (B, Cv, Nk)), rs.standard_normal((B, Cv, Nq))
    sc = 1.0 / np.sqrt(K)
    p = co.softmax(np.einsum("bci,bcj->bij", q, k) * sc)
    dp = np.einsum("bci,bcj->bij", g, v)
    ds = p * (dp - (p * dp).sum(-1, keepdims=True)) * sc
    refs = (np.einsum("bij,bcj->bci", p, v), np.einsum("bij,bcj->bci", ds, k), np.einsum("bij,bci->bcj", ds, q), np.einsum("bci,bij->bcj", g, p))

    def run():
        qd, kd, vd = dev(q, True), dev(k, True), dev(v, True)
        with ops.KernelTimer() as kt:
            o = ops.softmax_attention(qd, kd, vd, float(sc))
            o.backward(dev(g))
        return (o.detach(), qd.grad, kd.grad, vd.grad), set(kt.summary())
    saved, tags = run()
    assert "corr_softmax_warp_recompute" not in tags
    monkeypatch.setattr(ops, "MAX_SAVED_LOGITS_BYTES", 0)
    monkeypatch.setattr(ops, "RECOMPUTE_CHUNK_BYTES", B * Nq * 128 * 4)
    chunked, tags = run()
    assert "corr_softmax_warp_recompute" in tags
    for a, b, r, what in zip(chunked, saved, refs, ("out", "dq", "dk", "dv")):
        assert rel(a, r) < TOL, what
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), what


@pytest.mark.parametrize("qmag,kmag,vmag", [(4e5, 1.5e4, 1e4), (3e3, 2e3, 1.0), (40.0, 30.0, 5.0)])
def test_softmax_attention_is_finite_and_exact_at_any_logit_magnitude(qmag, kmag, vmag, monkeypatch):
    """Round 4 (found by the BASELINE config 3 generator, cocosnet_amd/translation.py): a randomly initialised SPADEGenerator feeds
    its Attention block theta ~ 4e5, phi ~ 1.5e4 — logits of 1e10.  The fused kernels kept the running maximum in the log2 domain;
    one ulp of it was ~700 in the exponent and 75 % of the output was NaN (the fp32 reference is finite there: torch subtracts the
    row maximum exactly).  The magnitude-free flavour keeps the maximum in raw-accumulator units and forms exact differences:
    finite everywhere, and equal to fp64 wherever fp32 logits can decide the row (gaps far above the accumulation noise)."""
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    rs = np.random.RandomState(11)
    B, K, Nq, Nk, Cv = 2, 32, 512, 256, 70
    q = rs.standard_normal((B, K, Nq)) * qmag
    k = rs.standard_normal((B, K, Nk)) * kmag
    v = rs.uniform(-1, 1, (B, Cv, Nk)) * vmag
    g = rs.standard_normal((B, Cv, Nq))
    f = np.einsum("bci,bcj->bij", q, k)
    p = co.softmax(f)
    o_ref = np.einsum("bij,bcj->bci", p, v)
    dp = np.einsum("bci,bcj->bij", g, v)
    ds = p * (dp - (p * dp).sum(-1, keepdims=True))
    dq_ref, dk_ref, dv_ref = np.einsum("bij,bcj->bci", ds, k), np.einsum("bij,bci->bcj", ds, q), np.einsum("bci,bij->bcj", g, p)
    qd, kd, vd = dev(q, True), dev(k, True), dev(v, True)
    with ops.KernelTimer() as kt:
        o = ops.softmax_attention(qd, kd, vd, 1.0)
        o.backward(dev(g))
    assert "corr_softmax_warp_fwd" in kt.summary()
    for t in (o, qd.grad, kd.grad, vd.grad):
        assert torch.isfinite(t).all()
    assert rel(o, o_ref) < TOL and rel(vd.grad, dv_ref) < TOL
    if qmag * kmag > 1e4:
        # one-hot rows: d q = d k = 0 exactly in fp64.  In fp32 the row's own entry carries dS = P (dP - D) with dP - D the
        # ROUNDING residual of two equal numbers (~2^-20 |g| |v| Cv), times the other operand: bounded, not compared
        assert float(qd.grad.abs().max()) <= 1e-4 * kmag * vmag * Cv and float(kd.grad.abs().max()) <= 1e-4 * qmag * vmag * Cv * Nq / Nk
    else:
        assert rel(qd.grad, dq_ref) < 5 * TOL and rel(kd.grad, dk_ref) < 5 * TOL      # (rows with near-ties: fp32 logits decide)
