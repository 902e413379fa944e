"""Pin the oracle (oracle/corr_oracle.py) against the reference's own outputs (tests/golden/).

CPU only.  The fixtures were produced by running /root/reference's NoVGGCorrespondence.forward
(oracle/make_golden.py); inputs are regenerated from the case seed.  Tolerances: the reference is
fp32 and the oracle fp64, and T = 0.01 amplifies fp32 GEMM rounding 100x in the logits, so outputs
agree to ~1e-6..1e-4 relative; gradients (backward GEMMs over fp32 P) to ~1e-4.
"""
import numpy as np
import pytest

from oracle import corr_oracle as co
from oracle import golden_cases as gc

FWD_TOL = 2e-4    # max-abs error relative to the tensor's max magnitude


@pytest.mark.parametrize("name", sorted(gc.CASES))
def test_oracle_forward_matches_reference(name):
    c = gc.CASES[name]
    inp = gc.make_inputs(name)
    golden = gc.load_golden(name)
    opt = co.default_opt(**gc.hot_path_flags(name))
    res = co.hot_path_forward(inp.theta_raw, inp.phi_raw, inp.ref_img, inp.real_img, inp.seg_map,
                              inp.ref_seg_map, opt, **c.get("fwd", {}))
    if not isinstance(res, dict):
        res = {"corr": res}
    errs = gc.compare_with_golden(name, res, golden)
    assert errs and max(errs.values()) < FWD_TOL, errs


def test_golden_inputs_are_not_degenerate():
    """The fixtures must exercise the softmax: rows neither uniform nor all one-hot."""
    inp = gc.make_inputs("ade_mk1")
    q = co.center_l2norm(inp.theta_raw.reshape(2, 256, -1).astype(np.float64), True)
    k = co.center_l2norm(inp.phi_raw.reshape(2, 256, -1).astype(np.float64), True)
    p = co.softmax(co.correlation(q, k) * 100.0)
    row_max = p.max(-1)
    assert 0.05 < np.median(row_max) < 0.999 and (row_max < 0.9).mean() > 0.2


def _sum_pool(g, s):
    B, C, H, W = g.shape
    return g.reshape(B, C, H // s, s, W // s, s).sum(axis=(3, 5))


@pytest.mark.parametrize("name", ["ade_mk1", "temp_005", "noponoc_mk1"])
def test_oracle_backward_matches_reference_autograd(name):
    """Chain the oracle's hand-derived backward pieces and compare with the gradients the
    reference's autograd produced for loss = <warp_out, G1> + <warp_mask, G2>."""
    c = gc.CASES[name]
    inp = gc.make_inputs(name)
    golden = gc.load_golden(name)
    flags = gc.hot_path_flags(name)
    inv_t = 1.0 / c.get("fwd", {}).get("temperature", 0.01)
    B = c["B"]
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    theta = f64(inp.theta_raw).reshape(B, 256, -1)
    phi = f64(inp.phi_raw).reshape(B, 256, -1)
    qn, kn = co.center_l2norm(theta, flags["PONO_C"]), co.center_l2norm(phi, flags["PONO_C"])
    ref = co.avg_pool2d(f64(inp.ref_img), 4).reshape(B, 3, -1)
    seg = co.nearest_down(f64(inp.ref_seg_map), 4).reshape(B, c["nc"], -1)
    v = np.concatenate([ref, seg], axis=1)
    G = gc.grad_weights(name, {"warp_out": tuple(golden["shape__warp_out"]),
                               "warp_mask": tuple(golden["shape__warp_mask"])})
    d_y = _sum_pool(f64(G["warp_out"]), 4).reshape(B, 3, -1)          # nearest upsample backward
    d_m = f64(G["warp_mask"]).reshape(B, c["nc"], -1)
    dout = np.concatenate([d_y, d_m], axis=1)
    dqn, dkn, _ = co.corr_softmax_warp_bwd(qn, kn, v, dout, inv_t)
    dtheta = co.center_l2norm_bwd(theta, dqn, flags["PONO_C"]).reshape(inp.theta_raw.shape)
    dphi = co.center_l2norm_bwd(phi, dkn, flags["PONO_C"]).reshape(inp.phi_raw.shape)
    for got, key in ((dtheta, "grad__theta_raw"), (dphi, "grad__phi_raw")):
        ref_g = golden[key].astype(np.float64)
        err = np.abs(got - ref_g).max() / np.abs(ref_g).max()
        assert err < 1e-3, (key, err)


def test_oracle_backward_matches_finite_differences():
    """Independent of any autograd: central differences on a tiny problem (fp64)."""
    rs = np.random.RandomState(0)
    q = rs.standard_normal((1, 8, 5)); k = rs.standard_normal((1, 8, 6)); v = rs.standard_normal((1, 2, 6))
    g = rs.standard_normal((1, 2, 5))
    loss = lambda qq, kk, vv: (co.corr_softmax_warp(co.center_l2norm(qq, True), co.center_l2norm(kk, True),
                                                    vv, 7.0) * g).sum()
    qn, kn = co.center_l2norm(q, True), co.center_l2norm(k, True)
    dqn, dkn, dv = co.corr_softmax_warp_bwd(qn, kn, v, g, 7.0)
    dq, dk = co.center_l2norm_bwd(q, dqn, True), co.center_l2norm_bwd(k, dkn, True)
    h = 1e-6
    for arr, grad, which in ((q, dq, 0), (k, dk, 1), (v, dv, 2)):
        for idx in [(0, 1, 2), (0, 0, 0), (0, 1, 4)]:
            a0 = arr[idx]
            args = [q, k, v]
            arr[idx] = a0 + h; lp = loss(*args)
            arr[idx] = a0 - h; lm = loss(*args)
            arr[idx] = a0
            assert abs((lp - lm) / (2 * h) - grad[idx]) < 1e-6 * max(1.0, abs(grad[idx]))


def test_wta_scale_rule():
    f = np.array([[[1.0, 3.0, 2.0], [5.0, 5.0, -1.0]]])
    out = co.wta_scale(f, 0.5)
    assert np.allclose(out, [[[0.5, 3.0, 1.0], [5.0, 5.0, -0.5]]])
    gi = co.wta_scale_bwd(f, np.ones_like(f))
    assert np.allclose(gi, [[[1e-4, 1.0, 1e-4], [1.0, 1.0, 1e-4]]])


def test_unfold_fold_roundtrip_and_ordering():
    rs = np.random.RandomState(1)
    x = rs.standard_normal((2, 3, 8, 8))
    cols = co.unfold(x, 4, stride=4)
    assert cols.shape == (2, 48, 4)
    assert np.array_equal(co.fold(cols, (8, 8), 4, 4), x)
    # channel ordering c*k*k + ky*k + kx, zero padding (F.unfold semantics)
    u = co.unfold(x, 3, padding=1)
    assert u.shape == (2, 27, 64)
    assert u[0, 0 * 9 + 0 * 3 + 0, 0] == 0.0                      # top-left tap of position 0 is padding
    assert u[1, 2 * 9 + 1 * 3 + 1, 9] == x[1, 2, 1, 1]            # centre tap
    assert u[0, 1 * 9 + 2 * 3 + 2, 0] == x[0, 1, 1, 1]            # bottom-right tap of position 0


# ------------------------------------------------------------------ the torch-autograd oracle (oracle/torch_ref.py)
@pytest.mark.parametrize("name", sorted(gc.CASES))
def test_torch_oracle_matches_reference_outputs_and_autograd(name):
    """oracle/torch_ref.py (fp64, torch CPU autograd) against the reference's own outputs AND, for the cases that
    store them, the reference's autograd gradients of loss = sum_k <out_k, G_k> — every flag set, incl. the cycle
    terms / column softmax / patches / WTA whose backward the numpy oracle does not restate."""
    import torch
    from oracle import torch_ref as tr
    c = gc.CASES[name]
    inp = gc.make_inputs(name)
    golden = gc.load_golden(name)
    opt = co.default_opt(**gc.hot_path_flags(name))
    shapes = {f.split("__", 1)[1]: tuple(golden[f]) for f in golden.files if f.startswith("shape__")}
    G = gc.grad_weights(name, shapes) if c.get("grads") else {}
    outs, dth, dph = tr.forward_backward(inp.theta_raw, inp.phi_raw, inp.ref_img, inp.real_img, inp.seg_map,
                                         inp.ref_seg_map, opt, G if G else {k: np.zeros(s, np.float32) for k, s in shapes.items()},
                                         **c.get("fwd", {}))
    errs = gc.compare_with_golden(name, outs, golden)
    assert errs and max(errs.values()) < FWD_TOL, errs
    if c.get("grads"):
        for got, key in ((dth, "theta_raw"), (dph, "phi_raw")):
            err = gc.grad_error(name, key, got, golden)
            assert err < 1e-3, (key, err)


def test_torch_oracle_agrees_with_numpy_oracle_fp64():
    """Two independent restatements of the same lines: outputs to 1e-12, row-pass gradients to 1e-10."""
    from oracle import torch_ref as tr
    inp = gc.make_inputs("celeba_cycle")
    opt = co.default_opt(**gc.hot_path_flags("celeba_cycle"))
    ref = co.hot_path_forward(inp.theta_raw, inp.phi_raw, inp.ref_img, inp.real_img, inp.seg_map, inp.ref_seg_map, opt)
    outs, _, _ = tr.forward_backward(inp.theta_raw, inp.phi_raw, inp.ref_img, inp.real_img, inp.seg_map,
                                     inp.ref_seg_map, opt, {k: np.zeros_like(v) for k, v in ref.items()})
    assert set(outs) == set(ref)
    for k in ref:
        assert np.abs(outs[k] - ref[k]).max() < 1e-11, k


# ------------------------------------------------------------------ contextual loss: reference-generated fixtures
CTX_GOLDEN = ("contextual_pono", "contextual_nopono", "contextual_h05")


@pytest.mark.parametrize("name", CTX_GOLDEN)
def test_contextual_class_matches_the_reference_fixture_on_cpu(name):
    """tests/golden/contextual_*.npz are outputs and autograd gradients of the reference's own ContextualLoss_forward
    (oracle/make_contextual_golden.py).  The drop-in class on CPU tensors (the reference's formulation in torch: the fp64 arbiter
    of the GPU tests is this code path in double) reproduces them."""
    import os
    from types import SimpleNamespace
    import numpy as np
    import torch
    from cocosnet_amd.contextual import ContextualLoss_forward
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    x = torch.from_numpy(f["X"]).requires_grad_(True)
    loss = ContextualLoss_forward(SimpleNamespace(PONO=bool(f["pono"])))(x, torch.from_numpy(f["Y"]), h=float(f["h"]))
    loss.sum().backward()
    assert np.allclose(loss.detach().numpy(), f["loss"], rtol=2e-5, atol=1e-6)
    assert np.abs(x.grad.numpy() - f["dX"]).max() <= 2e-5 * np.abs(f["dX"]).max()


@pytest.mark.parametrize("name", CTX_GOLDEN)
def test_contextual_oracle_is_pinned_to_the_reference_fixtures(name):
    """oracle/contextual_ref.py (the fp64 arbiter of the at-size GPU tests of K22) against the reference-generated fixtures: the
    per-sample loss and d loss / d X — in fp64 on the fixtures' fp32 inputs, so the tolerance is the FIXTURES' own fp32 rounding."""
    import os
    import numpy as np
    import torch
    from oracle import contextual_ref as cr
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    x = torch.from_numpy(f["X"]).double().requires_grad_(True)
    loss = cr.contextual_loss(x, torch.from_numpy(f["Y"]).double(), h=float(f["h"]), pono=bool(f["pono"]))
    loss.sum().backward()
    assert np.allclose(loss.detach().numpy(), f["loss"], rtol=2e-5, atol=1e-6)
    assert np.abs(x.grad.numpy() - f["dX"]).max() <= 2e-5 * np.abs(f["dX"]).max()


def test_contextual_fixtures_regenerate_from_the_reference():
    """With /root/reference present (build container): running the reference again reproduces the committed fixtures bit for bit."""
    import os
    import numpy as np
    from oracle import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("reference tree not present (GPU box)")
    from oracle import make_contextual_golden as mk
    for name in CTX_GOLDEN:
        got = mk.run_reference_case(name)
        f = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
        for k in ("X", "Y", "loss", "dX"):
            assert np.array_equal(got[k], f[k]), (name, k)
