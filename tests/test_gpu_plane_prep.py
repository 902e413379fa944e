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
