"""is there a convolution flavour cheaper than f16x3 (three MFMA terms) that keeps BASELINE config 3's networks
(SPADE generator, PatchGAN: generator.py:17-89, discriminator.py:92-115) within north_star's 1e-3 of fp64?

Measured WITHOUT writing the kernels first: the candidate arithmetic is EMULATED on the framework's fp32 convolution by rounding
the operands the way the candidate kernel's operand preparation would (per-tensor power-of-two scale, max|x| -> [2^9, 2^10), round to
nearest) — forward, input gradient and weight gradient alike (a custom autograd Function around F.conv2d / torch.nn.grad).  fp32
accumulation of exact products of rounded operands is what the MFMA does, so the emulation's error is the candidate's error up to
the accumulation order.  All arms are evaluated on the fp64 copy's branch pattern (tests/kink_tape.py), so that the figures
are rounding, not LeakyReLU flips.  The real kernels' arms (f16x3, bf16) run beside the emulated ones as the calibration.

    arms:   f16x3 (real)   bf16 (real)   emu:bf16 (calibration of the emulation against the real bf16 arm)
            emu:f16        one f16 term: activations, gradients and weights rounded to 11 bits           (1/3 of the MFMA work)
            emu:f16x2a     two terms: activations / gradients hi + lo (22 bits), weights ONE f16 plane    (2/3)
            emu:f16x2w     two terms: weights hi + lo, activations / gradients ONE f16 plane              (2/3)
    scope:  which network's k > 1 convolutions take the flavour (the other network stays on f16x3).

usage: python tools/conv_flavour_table.py [out.json]
"""
import copy
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import kink_tape  # noqa: E402  (test infrastructure: tests/kink_tape.py)
from cocosnet_amd import ops, producers, translation as tl  # noqa: E402


def _scale(x):
    a = float(x.detach().abs().max())
    if not (a > 0) or a != a or a == float("inf"):
        return 1.0
    import math
    return 2.0 ** (10 - math.frexp(a)[1])


def q_f16(x):
    s = _scale(x)
    return (x * s).half().float() / s


def q_f16x2(x):          # hi + lo: 22 bits
    s = _scale(x)
    hi = (x * s).half().float()
    lo = (x * s - hi).half().float()
    return (hi + lo) / s


def q_bf16(x):
    return x.bfloat16().float()


FLAVOURS = {"emu:bf16": (q_bf16, q_bf16), "emu:f16": (q_f16, q_f16), "emu:f16x2a": (q_f16x2, q_f16), "emu:f16x2w": (q_f16, q_f16x2)}


class _QConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, dil, qa, qw):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, qa, qw, b is not None)
        return F.conv2d(qa(x), qw(w), b, stride, pad, dil)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil, qa, qw, has_b = ctx.cfg
        dyq = qa(dy)
        dx = torch.nn.grad.conv2d_input(x.shape, qw(w), dyq, stride=stride, padding=pad, dilation=dil) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(qa(x), w.shape, dyq, stride=stride, padding=pad, dilation=dil) if ctx.needs_input_grad[1] else None
        db = dy.sum((0, 2, 3)) if has_b and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None, None, None


def install(net, flavour):
    """k > 1 convolutions of `net` -> the emulated flavour (1x1 layers stay on K0 = f16x3, as in the real bf16 arm)."""
    qa, qw = FLAVOURS[flavour]
    for m in net.modules():
        if isinstance(m, producers.Conv2d) and m.kernel_size != (1, 1):
            def fwd(input, weight, bias, reflect=0, m=m):
                if reflect:
                    input = F.pad(input, (reflect,) * 4, mode="reflect")
                return _QConv.apply(input, weight, bias, m.stride, m.padding, m.dilation, qa, qw)
            m._conv_forward = fwd


def uninstall(net):
    for m in net.modules():
        if isinstance(m, producers.Conv2d) and "_conv_forward" in m.__dict__:
            del m.__dict__["_conv_forward"]


def main():
    opt = tl.celebahq_edge_train_options()
    torch.manual_seed(0)
    G = tl.SPADEGenerator(opt).cuda(); G.init_weights(opt.init_type, opt.init_variance); G.eval()
    D = tl.MultiscaleDiscriminator(opt).cuda(); D.init_weights(opt.init_type, opt.init_variance); D.eval()
    g = torch.Generator(device="cuda").manual_seed(21)
    B = 2
    seg = torch.rand(B, 15, 256, 256, device="cuda", generator=g)
    cbn = torch.cat((torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1, seg), 1)
    gy = torch.randn(B, 3, 256, 256, device="cuda", generator=g)
    fixed_fake = torch.tanh(torch.randn(B, 3, 256, 256, device="cuda", generator=g))     # D's input when G is not the subject

    def run(Gm, Dm, dt):
        Gm.zero_grad(); Dm.zero_grad()
        y = Gm(seg.to(dt), warp_out=cbn.to(dt))
        y.backward(gy.to(dt))
        res = Dm(torch.cat((seg.to(dt), fixed_fake.to(dt)), 1))[0]        # (a FIXED image: D's figures are D's own, not G's error fed forward)
        torch.autograd.backward([r[-1] for r in res], [torch.ones_like(r[-1]) for r in res])
        out = {"G fake_image": y.detach(), "D0 logits": res[0][-1].detach(), "D1 logits": res[1][-1].detach()}
        for n, p in Gm.named_parameters():
            if p.grad is not None and p.numel() > 64:
                out["dG " + n] = p.grad.clone()
        for n, p in Dm.named_parameters():
            if p.grad is not None and p.numel() > 64:
                out["dD " + n] = p.grad.clone()
        return out

    tape = kink_tape.KinkTape()
    tape_ctx = kink_tape.install(tape)
    tape_ctx.__enter__()
    want = run(copy.deepcopy(G).double(), copy.deepcopy(D).double(), torch.float64)
    table = {}

    def arm(name, backend, flavour=None, scope=""):
        ops.CONV_PRECISION = backend
        if flavour:
            if "G" in scope:
                install(G, flavour)
            if "D" in scope:
                install(D, flavour)
        tape.rewind("replay")
        try:
            got = run(G, D, torch.float32)
        finally:
            uninstall(G); uninstall(D)
        e = {k: float((got[k].double() - want[k]).abs().max() / (want[k].abs().max() + 1e-300)) for k in want}
        gG = [v for k, v in e.items() if k.startswith("dG ")]
        gD = [v for k, v in e.items() if k.startswith("dD ")]
        worstG = max((v, k) for k, v in e.items() if k.startswith("dG "))
        worstD = max((v, k) for k, v in e.items() if k.startswith("dD "))
        table[name] = {"G fake_image": e["G fake_image"], "D0 logits": e["D0 logits"], "D1 logits": e["D1 logits"],
                       "dG max": max(gG), "dG median": sorted(gG)[len(gG) // 2], "dG worst": worstG[1],
                       "dD max": max(gD), "dD median": sorted(gD)[len(gD) // 2], "dD worst": worstD[1],
                       "flips": [sum(tape.flips), tape.elements]}
        print(name, json.dumps(table[name]), flush=True)

    arm("f16x3 (real)", "f16x3")
    arm("torch fp32 (real)", "torch")
    arm("bf16 (real)", "bf16")
    for fl in FLAVOURS:
        arm(fl + " G+D", "f16x3", fl, "GD")
    tape_ctx.__exit__(None, None, None)
    ops.CONV_PRECISION = "f16x3"
    if len(sys.argv) > 1:
        json.dump(table, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
