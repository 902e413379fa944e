"""Plane preparation folded into producer kernels (round 3; csrc/plane_prep.hip, center_l2norm.hip planes flavour,
warp_values.hip amax): the fused forms must give what the separate launches they replace gave — bit for bit where the
arithmetic is the same (concat, V split + mask, weight planes, K14), to the last fp32 bit of y for K1 (another summation order)."""
import numpy as np
import pytest
import torch

from cocosnet_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no fallback)")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, device=DEV, generator=g) * scale


@pytest.mark.parametrize("B,N,mode", [(2, 4096, 1), (1, 64, 1), (3, 260, 2), (1, 4, 1)])
def test_center_l2norm_planes_equal_split_of_the_fp32_kernel_and_backward_agrees(B, N, mode, monkeypatch):
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PROJ_PRECISION", "f16x3")
    x = _rand(B, 256, N, seed=N) + 0.3
    y = ops.center_l2norm(x, mode)
    planes = ops.OperandPlanes()
    xg = x.clone().requires_grad_(True)
    hdl = ops.center_l2norm_planes(xg, mode, planes)
    assert hdl.data_ptr() == xg.data_ptr()                               # a handle, no new memory
    for tr in (True, False):
        # (not bit-identical to split_f16(center_l2norm(x)): this kernel sums the 256 channels of a position in another
        #  order, so y may differ in its last fp32 bit — the planes must reproduce the fp32 result to 2^-21 of the row scale)
        hi, lo = planes.get(hdl, tr, ops.SPLIT_OPERAND_SCALE)
        rec = (hi.float() + lo.float()) / ops.SPLIT_OPERAND_SCALE
        want = y.transpose(1, 2) if tr else y
        assert rec.shape == want.shape
        assert float((rec - want).abs().max()) < 2.0 ** -21 * float(y.abs().max()) * 2, tr
        rh, _ = ops.split_f16(y, tr, ops.SPLIT_OPERAND_SCALE)
        assert float((hi != rh).float().mean()) < 1e-3                  # hi planes agree except where y sits on a rounding edge
    g = _rand(B, 256, N, seed=N + 1)
    hdl.backward(g)
    x2 = x.clone().requires_grad_(True)
    ops.center_l2norm(x2, mode).backward(g)
    err = float((xg.grad - x2.grad).abs().max() / x2.grad.abs().max())
    assert err < 2e-6, err                                                # y read back from 22-bit planes
    with torch.no_grad():
        xg.add_(0.0)                                                      # an in-place op on the tensor behind the handle
    with pytest.raises(ops._lib.CocosHipError):                           # planes-only operands cannot be re-made: loud
        planes.get(hdl, True, ops.SPLIT_OPERAND_SCALE)


def test_concat_channels_amax_is_torch_cat_plus_the_maximum(monkeypatch):
    from cocosnet_amd import ops
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    a, b = _rand(8, 3, 64, 64, seed=1), _rand(8, 151, 64, 64, seed=2, scale=3.0)
    b[5, 77, 13, 9] = -41.5
    out = ops.concat_channels_amax(a, b)
    assert torch.equal(out, torch.cat((a, b), 1))
    cell = ops._recall_amax(out)
    assert cell is not None and float(cell) == 41.5
    # shapes the kernel does not take fall back to the framework's cat
    c = ops.concat_channels_amax(_rand(2, 3, 5, 1, seed=3), _rand(2, 2, 5, 1, seed=4))
    assert c.shape == (2, 5, 5, 1)


@pytest.mark.parametrize("onehot", [True, False])
def test_split_f16_chan_mask_equals_split_plus_mask(onehot):
    from cocosnet_amd import ops
    B, C, N = 2, 154, 4096
    v = torch.rand(B, C, N, device=DEV) * 2 - 1
    if onehot:
        lab = torch.randint(0, 151, (B, 1, N), device=DEV)
        v[:, 3:] = torch.zeros(B, 151, N, device=DEV).scatter_(1, lab, 1.0)
    amax = ops.absmax(v)
    hi, lo, sc, mask = ops.split_f16_chan_mask(v, amax, True)
    rh, rl, rs = ops.split_f16(v, False, amax=amax)
    rm = ops.f16_plane_block_mask(rl)
    assert torch.equal(hi, rh) and torch.equal(lo, rl) and float(sc) == float(rs)
    assert int(mask.view(torch.int32)) == int(rm.view(torch.int32))
    assert (int(mask.view(torch.int32)) & ~1) == (0 if onehot else 0b11110)


@pytest.mark.parametrize("Cout,Cin", [(256, 407), (256, 256), (256, 271), (64, 40)])
def test_proj_weight_planes_equal_the_two_separate_splits(Cout, Cin):
    from cocosnet_amd import _lib, ops
    lib = _lib.load()
    kpi, kpo = lib.cocos_proj1x1_stream_kpad(Cin), lib.cocos_proj1x1_stream_kpad(Cout)
    if kpi == 0 or kpo == 0:
        kpi, kpo = (Cin + 15) // 16 * 16, (Cout + 15) // 16 * 16
    w = _rand(Cout, Cin, seed=Cin, scale=0.02)
    wa = ops.absmax(w)
    half = dict(device=DEV, dtype=torch.float16)
    rh, rl = torch.full((Cout, kpi), 7.0, **half), torch.full((Cout, kpi), 7.0, **half)
    th, tl = torch.full((Cin, kpo), 7.0, **half), torch.full((Cin, kpo), 7.0, **half)
    ws = torch.empty(1, device=DEV)
    ops._call("t", "cocos_proj_weight_planes", w.data_ptr(), rh.data_ptr(), rl.data_ptr(), th.data_ptr(), tl.data_ptr(),
              Cout, Cin, kpi, kpo, wa.data_ptr(), ws.data_ptr(), ops._stream())
    eh, el, es = torch.empty_like(rh), torch.empty_like(rl), torch.empty(1, device=DEV)
    ops._call("t", "cocos_split_f16_rows", w.data_ptr(), eh.data_ptr(), el.data_ptr(), Cout, Cin, kpi, 1.0, wa.data_ptr(),
              es.data_ptr(), ops._stream())
    xh, xl, xs = ops.split_f16(w.unsqueeze(0), transpose=True, cpad=kpo, amax=wa)
    assert torch.equal(rh, eh) and torch.equal(rl, el) and float(ws) == float(es) == float(xs)
    assert torch.equal(th, xh[0]) and torch.equal(tl, xl[0])


def test_warp_values_amax_matches_and_is_picked_up(monkeypatch):
    from cocosnet_amd import ops
    img = torch.rand(4, 3, 256, 256, device=DEV) * 2 - 1
    lab = torch.randint(0, 151, (4, 1, 256, 256), device=DEV)
    seg = torch.zeros(4, 151, 256, 256, device=DEV).scatter_(1, lab, 1.0)
    monkeypatch.setattr(ops, "PRECISION", "fp32")
    ref = ops.warp_values(img, seg, 4)
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    out = ops.warp_values(img, seg, 4)
    assert torch.equal(out, ref)
    cell = ops._recall_amax(out.reshape(4, 154, -1))
    assert cell is not None and float(cell) == float(ref.abs().max())


@pytest.mark.parametrize("B,C,shape", [(8, 64, (64, 64)), (2, 3, (37, 5)), (1, 256, (1, 1)), (4, 16, (128, 130)), (3, 1, (100003,))])
def test_channel_sum_is_the_bias_gradient_sum(B, C, shape):
    """cocos_channel_sum (sliced, two launches when the tensor is large) against an fp64 sum; an unaligned view takes the 4-byte path."""
    g = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(B, C, *shape, device=DEV, generator=g) + 0.25
    ref = dy.double().sum(dim=[0] + list(range(2, dy.dim())))
    n = dy[0, 0].numel() * B
    tol = 4e-7 * (n ** 0.5) * 4 + 1e-6
    assert (ops.channel_sum(dy).double() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    flat = torch.empty(dy.numel() + 1, device=DEV)
    odd = flat[1:].view_as(dy).copy_(dy)                       # data pointer 4 bytes past a 16-byte boundary
    assert (ops.channel_sum(odd).double() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))


def test_sum_leading_sums_partial_slices():
    g = torch.Generator(device=DEV).manual_seed(6)
    for S, n in ((7, 4096), (3, 1001), (1, 8)):
        x = torch.randn(S, n, device=DEV, generator=g)
        assert torch.allclose(ops.sum_leading(x), x.sum(0), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("B,Ci,Cs,h,w,down", [(8, 3, 151, 64, 64, 4), (2, 3, 7, 4, 64, 4), (1, 2, 5, 6, 8, 2), (3, 1, 1, 5, 12, 3)])
def test_warp_head_is_upsample_plus_view_and_its_backward_is_the_three_launch_chain(B, Ci, Cs, h, w, down, monkeypatch):
    """ops.warp_head (round 6): forward = nearest up-sampling of the image channels + a view of the mask channels; backward = the
    window sums | copy that autograd's upsample backward + concatenation gave (bit for bit: same additions in the same order for
    down = 4), with max|d o| and D = sum_c d o * o (fp64) left for the consumer exactly as absmax / cocos_rowdot_f64 give them."""
    import torch.nn.functional as F
    monkeypatch.setattr(ops, "PRECISION", "f16x3")
    N = h * w
    seen = {}

    class Tap(torch.autograd.Function):          # stands where the K2 / K19 backward stands: it receives d o itself
        @staticmethod
        def forward(ctx, x):
            ctx.save_for_backward(x)
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            x, = ctx.saved_tensors
            cell = ops._recall_amax(g)
            seen["amax"] = None if cell is None else float(cell)
            seen["d"] = ops._rowdot_cached(g, x)
            seen["left"] = ops._tls.known_rowdot
            return g

    leaf = _rand(B, Ci + Cs, N, seed=h).requires_grad_(True)
    o = Tap.apply(leaf)
    assert ops.warp_head_ok(o, Ci, h, w, down)
    y, m = ops.warp_head(o, Ci, h, w, down)
    assert torch.equal(y, F.interpolate(o[:, :Ci].reshape(B, Ci, h, w), scale_factor=down, mode="nearest"))
    assert m.data_ptr() == o[:, Ci:].data_ptr() and torch.equal(m, o[:, Ci:].reshape(B, Cs, h, w))
    gy, gm = _rand(*y.shape, seed=1, scale=0.3), _rand(*m.shape, seed=2, scale=2.0)
    torch.autograd.backward([y, m], [gy, gm])
    o2 = leaf.detach().clone().requires_grad_(True)
    y2 = F.interpolate(o2[:, :Ci].reshape(B, Ci, h, w), scale_factor=down, mode="nearest")
    torch.autograd.backward([y2, o2[:, Ci:].reshape(B, Cs, h, w)], [gy, gm])
    err = float((leaf.grad - o2.grad).abs().max() / o2.grad.abs().max())
    assert err < 1e-6, err
    assert torch.equal(leaf.grad[:, Ci:], gm.reshape(B, Cs, N))
    assert seen["amax"] == float(leaf.grad.abs().max())
    want = (leaf.grad.double() * leaf.detach().double()).sum(1)
    assert float((seen["d"].double() - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-30
    assert seen["left"] is None          # consumed


def test_split_pair_transposed_equals_two_single_splits():
    a, b = _rand(3, 154, 320, seed=5, scale=7.0), _rand(3, 154, 320, seed=6, scale=1e-3)
    aa, ba = ops.absmax(a), ops.absmax(b)
    (h0, l0, s0), (h1, l1, s1) = ops._split_pair_transposed(a, aa, b, ba, 160)
    for x, am, (h, l, s) in ((a, aa, (h0, l0, s0)), (b, ba, (h1, l1, s1))):
        rh, rl, rs = ops.split_f16(x, True, cpad=160, amax=am)
        assert torch.equal(h, rh) and torch.equal(l, rl) and float(s) == float(rs)


def test_absmax_many_and_weight_prep_pair_equal_the_single_launches():
    from cocosnet_amd import _lib
    lib = _lib.load()
    xs = [_rand(8, 407, 64, 64, seed=1, scale=3.0), _rand(256, 407, seed=2, scale=0.05), _rand(5, 7, seed=3), _rand(2, 407, 16, 16, seed=4),
          _rand(1, 3, seed=5, scale=100.0)]
    cells = ops.absmax_many(xs)
    for x, c in zip(xs, cells):
        assert float(c) == float(x.abs().max())
    Cin = 407
    ws = [xs[1], _rand(256, Cin, seed=9, scale=2.0)]
    was = ops.absmax_many(ws)
    single, pair, args = [], [], []
    u8 = dict(device=DEV, dtype=torch.uint8)
    for w2d, wa in zip(ws, was):
        f, t = torch.zeros(lib.cocos_proj_weight_frag_bytes(Cin), **u8), torch.zeros(lib.cocos_proj_weight_tfrag_bytes(), **u8)
        th, tl = torch.zeros((Cin, 256), device=DEV, dtype=torch.float16), torch.zeros((Cin, 256), device=DEV, dtype=torch.float16)
        sc = torch.zeros(1, device=DEV)
        ops._call("split_f16", "cocos_proj_weight_frag_planes", w2d.data_ptr(), wa.data_ptr(), f.data_ptr(), sc.data_ptr(), th.data_ptr(),
                  tl.data_ptr(), 256, Cin, ops._stream())
        ops._call("split_f16", "cocos_proj_weight_tfrag_planes", w2d.data_ptr(), wa.data_ptr(), t.data_ptr(), None, 256, Cin, ops._stream())
        single.append((f, t, th, tl, sc))
        f2, t2, th2, tl2, sc2 = torch.zeros_like(f), torch.zeros_like(t), torch.zeros_like(th), torch.zeros_like(tl), torch.zeros_like(sc)
        pair.append((f2, t2, th2, tl2, sc2))
        args += [w2d.data_ptr(), wa.data_ptr(), f2.data_ptr(), sc2.data_ptr(), th2.data_ptr(), tl2.data_ptr(), t2.data_ptr()]
    ops._call("split_f16", "cocos_proj_weight_prep_pair", 2, *args, 256, Cin, ops._stream())
    for a, b in zip(single, pair):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
