"""max|.| left behind by the producer kernels (round 6): K13 (InstanceNorm + PReLU, correspondence.py:13-36) and K9 (PositionalNorm +
SPADE + LeakyReLU, normalization.py:63-68,:148-151) write the maximum of the tensors that the next convolution splits into f16
planes, so that the consumer does not take a pass of its own (cocos_absmax).  The cell must hold EXACTLY max|tensor| (the split
scale is a power of two derived from it: a smaller value overflows the hi plane, a larger one costs bits), the tensors themselves
must be bit-identical to the plain entry points', and the consumer must find the cell."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _cell(v=0.0):
    return torch.full((1,), v, device=DEV)


@pytest.mark.parametrize("shape,res", [((3, 5, 16, 16), False), ((2, 7, 64, 64), True), ((2, 3, 128, 128), True), ((1, 2, 7, 9), False),
                                       ((1, 2, 160, 160), True)])
def test_instnorm_prelu_leaves_the_maxima(shape, res, hip_lib):
    from cocosnet_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(*shape, device=DEV, generator=g) * 3 + 0.5
    r = torch.randn(*shape, device=DEV, generator=g) if res else None
    w = torch.tensor([0.25], device=DEV)
    dy = torch.randn(*shape, device=DEV, generator=g)
    planes, N = shape[0] * shape[1], shape[2] * shape[3]
    ptr = lambda t: 0 if t is None else t.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    c = _cell()
    _lib.call("cocos_instnorm_prelu_fwd", x.data_ptr(), ptr(r), w.data_ptr(), y0.data_ptr(), planes, N, 1e-5, s)
    part = torch.empty(planes, device=DEV)
    _lib.call("cocos_instnorm_prelu_fwd_amax", x.data_ptr(), ptr(r), w.data_ptr(), y1.data_ptr(), c.data_ptr(), part.data_ptr(), planes, N, 1e-5, s)
    assert torch.equal(y0, y1) and float(c) == float(y0.abs().max())
    dx0, dr0, dx1, dr1 = (torch.empty_like(x) for _ in range(4))
    dap0, dap1 = (torch.empty(planes, device=DEV, dtype=torch.float64) for _ in range(2))
    da0, da1 = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    c = _cell()
    _lib.call("cocos_instnorm_prelu_bwd_f64", x.data_ptr(), ptr(r), w.data_ptr(), dy.data_ptr(), dx0.data_ptr(), dr0.data_ptr(),
              dap0.data_ptr(), da0.data_ptr(), planes, N, 1e-5, s)
    _lib.call("cocos_instnorm_prelu_bwd_amax", x.data_ptr(), ptr(r), w.data_ptr(), dy.data_ptr(), dx1.data_ptr(), dr1.data_ptr(),
              dap1.data_ptr(), da1.data_ptr(), c.data_ptr(), part.data_ptr(), planes, N, 1e-5, s)
    assert torch.equal(dx0, dx1) and torch.equal(dr0, dr1) and torch.equal(da0, da1) and float(c) == float(dx0.abs().max())
    # without the weight gradient; a cell that already holds a larger value keeps it
    c = _cell(1e9)
    _lib.call("cocos_instnorm_prelu_bwd_amax", x.data_ptr(), ptr(r), w.data_ptr(), dy.data_ptr(), dx1.data_ptr(), 0, 0, 0, c.data_ptr(),
              part.data_ptr(), planes, N, 1e-5, s)
    assert torch.equal(dx0, dx1) and float(c) == 1e9


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (1, 256, 8, 12), (2, 96, 32, 32), (2, 40, 6, 10), (1, 33, 5, 5)])
def test_pono_spade_leaves_the_maxima(shape, hip_lib):
    from cocosnet_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x, ga, be, dy = (torch.randn(*shape, device=DEV, generator=g) for _ in range(4))
    B, C, N = shape[0], shape[1], shape[2] * shape[3]
    s = torch.cuda.current_stream().cuda_stream
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    c = _cell()
    _lib.call("cocos_pono_spade_fwd", x.data_ptr(), ga.data_ptr(), be.data_ptr(), y0.data_ptr(), B, C, N, 1e-5, 0.2, s)
    part = torch.empty(2 * _lib.load().cocos_pono_spade_amax_partials(B, C, N), device=DEV)
    _lib.call("cocos_pono_spade_fwd_amax", x.data_ptr(), ga.data_ptr(), be.data_ptr(), y1.data_ptr(), c.data_ptr(), part.data_ptr(), B, C, N, 1e-5, 0.2, s)
    assert torch.equal(y0, y1) and float(c) == float(y0.abs().max())
    a0 = [torch.empty_like(x) for _ in range(3)]
    a1 = [torch.empty_like(x) for _ in range(3)]
    c2 = torch.zeros(2, device=DEV)
    _lib.call("cocos_pono_spade_bwd", x.data_ptr(), ga.data_ptr(), be.data_ptr(), dy.data_ptr(), *(t.data_ptr() for t in a0), B, C, N, 1e-5, 0.2, s)
    _lib.call("cocos_pono_spade_bwd_amax", x.data_ptr(), ga.data_ptr(), be.data_ptr(), dy.data_ptr(), *(t.data_ptr() for t in a1), c2.data_ptr(),
              part.data_ptr(), B, C, N, 1e-5, 0.2, s)
    for u, v in zip(a0, a1):
        assert torch.equal(u, v)
    assert float(c2[0]) == float(a0[1].abs().max()) and float(c2[1]) == float(a0[2].abs().max())
    # only dgamma / dbeta wanted (SPADE on a constant input): dx absent
    c2.zero_()
    _lib.call("cocos_pono_spade_bwd_amax", x.data_ptr(), ga.data_ptr(), be.data_ptr(), dy.data_ptr(), 0, a1[1].data_ptr(), a1[2].data_ptr(),
              c2.data_ptr(), part.data_ptr(), B, C, N, 1e-5, 0.2, s)
    assert torch.equal(a0[1], a1[1]) and float(c2[0]) == float(a0[1].abs().max()) and float(c2[1]) == float(a0[2].abs().max())


def test_the_next_convolution_finds_the_cell(hip_lib, monkeypatch):
    """A ResidualBlock-shaped chain conv -> K13 -> conv, forward and backward: the only max|.| passes left are the ones over the leaves
    (the input, the weights) and over the incoming gradient."""
    from cocosnet_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(2, 128, 32, 32, device=DEV, generator=g).requires_grad_(True)
    w1 = (torch.randn(128, 128, 3, 3, device=DEV, generator=g) * 0.05).requires_grad_(True)
    w2 = (torch.randn(128, 128, 3, 3, device=DEV, generator=g) * 0.05).requires_grad_(True)
    a = torch.tensor([0.25], device=DEV, requires_grad=True)
    seen = []
    real = ops.absmax
    monkeypatch.setattr(ops, "absmax", lambda t: (seen.append(tuple(t.shape)), real(t))[1])
    y = ops.conv2d(ops.instnorm_prelu(ops.conv2d(x, w1, None, 1, 1), None, a), w2, None, 1, 1)
    n_fwd = len(seen)
    y.backward(torch.randn(y.shape, device=DEV, generator=g))
    assert n_fwd == 3 and sorted(seen[:3]) == sorted([tuple(x.shape), tuple(w1.shape), tuple(w2.shape)]), seen
    assert len(seen) == 4 and seen[3] == tuple(y.shape), seen      # the incoming gradient; K13's dx arrived with its cell
