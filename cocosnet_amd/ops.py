"""torch.autograd bindings of the HIP kernels (host side above the C ABI).

PyTorch here is plumbing only: it owns device memory, the stream and the autograd graph; every
piece of arithmetic on the path is a hand-written gfx950 kernel reached through ctypes.
All feature tensors are channel-major [B, C, positions] fp32 — what `x.view(B, C, -1)` gives in
the reference (models/networks/correspondence.py:274,284).
"""
from __future__ import annotations

import os
import sys
import threading
import weakref

import torch

from . import _lib

#: `sys.float_info.epsilon`, the constant the reference adds to the norm (correspondence.py:279,288)
NORM_EPS = sys.float_info.epsilon
#: widest V the fused kernel takes in one launch (5 blocks of 32 channels)
MAX_FUSED_CV = 160
#: ... of the split-precision forward: one padding channel of the V tile carries the softmax row sums
MAX_FUSED_SPLIT_CV = 159
#: largest dS^T scratch (bytes) the EXACT-FP32 backward may allocate to replace the second logits recompute by
#: a GEMM; above it (e.g. 128x128 grids at large batch) the flash-style key kernel is used
MAX_DS_WORKSPACE_BYTES = 16 << 30
#: split flavour: a training forward SAVES its logits (B*Nq*Nk*4 bytes, and the backward as much again for the dS'' planes)
#: only up to this size; above it the forward keeps the row LSE alone and the backward RECOMPUTES the logits, one chunk of keys
#: at a time (round 4: _corr_bwd_recompute — same kernels, HWxHW scratch bounded by RECOMPUTE_CHUNK_BYTES per matrix).
#: Measured A/B at the three BASELINE shapes (profiles/r04_configs_bench.json): the saved-logits chain is 24 % (cfg2), 33 % (cfg3)
#: and 45-68 % (cfg5) FASTER where it fits, so the default is speed first on a 288 GB part — 16 GiB, the cap of rounds 1-3, beyond
#: which this route now replaces the exact-fp32 flash kernels; COCOS_MAX_SAVED_LOGITS_BYTES=1073741824 puts BASELINE config 5
#: (1 GiB of logits PER SAMPLE) on the bounded-memory route: 4.3 -> 2.5 GiB peak.  Module attribute, read at call time.
MAX_SAVED_LOGITS_BYTES = int(os.environ.get("COCOS_MAX_SAVED_LOGITS_BYTES", 16 << 30))
#: ... the recomputed logits / dS'' / P planes of one key chunk (bytes per matrix).  1 GiB: at HW = 16384, B = 2 a chunk is 8192
#: keys — the key-side GEMM of a chunk still has 128 workgroups (512 MiB chunks: 64 workgroups, its time tripled)
RECOMPUTE_CHUNK_BYTES = int(os.environ.get("COCOS_RECOMPUTE_CHUNK_BYTES", 1 << 30))
#: split K2 backward: D = sum_c dout * out per query from a streaming kernel (cocos_rowdot_f64) instead of the query kernel's own
#: serial fp64 prologue; "0": the round-3 form (A/B runs)
BWD_D_PRECOMPUTED = True
#: channel count the fused kernels are specialised for (self.inter_channels, correspondence.py:170)
FUSED_K = 256
#: where the K2 forward's products run: "fp32" = v_mfma_f32_32x32x2_f32 (exact fp32 operands);
#: "f16x3" = v_mfma_f32_32x32x16_f16 on f16 hi+lo planes, 3 terms per product, fp32 accumulate
#: (fp32-class accuracy, ~1/5 of the matrix-pipe time).  Module attribute, read at call time.
PRECISION = os.environ.get("COCOS_PRECISION", "f16x3")
#: K0 (theta/phi 1x1 projections): "fp32" = the fp32-MFMA GEMM (on par with rocBLAS); "f16x3" = split products on
#: the f16 MFMA — the streaming kernels below at the reference's shapes, else the split GEMM of sgemm_f16x3.hip
#: (operands split on the fly, staged pieces converted/committed between MFMAs).
PROJ_PRECISION = os.environ.get("COCOS_PROJ_PRECISION", "f16x3")
# ---- switches below WITHOUT an environment variable are test / A-B hooks: plain module attributes, read at call time, every
# ---- non-default value covered by a GPU test (the env surface is COCOS_PRECISION, COCOS_PROJ_PRECISION,
# ---- COCOS_CONV, COCOS_MAX_SAVED_LOGITS_BYTES, COCOS_RECOMPUTE_CHUNK_BYTES and COCOS_LIB_PATH)
#: K2 split kernels: test V's f16 lo plane per 32-channel block and skip all-zero blocks (exact label / mask channels)
VALUE_LO_SKIP = True
#: K0 at the reference's own shapes (<= 416 input channels, HW % 64 == 0): y = W x and dx = W^T dy on the streaming
#: kernel (proj_stream_f16x3.hip: weight planes resident in the accumulator file, x / y touched once), dw + db as one
#: streaming reduction (proj_dw_f16x3.hip).  "0" = the general split GEMM everywhere (A/B, tests).
PROJ_STREAM = True
#: power-of-two pre-scale of the unit-norm operands before the f16 split (keeps the lo plane normal)
SPLIT_OPERAND_SCALE = 16.0


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Brackets every C-ABI call with HIP events on the stream the kernels are launched on (torch's
    current stream) and reports the average device time per entry point — bench.py's live source
    for `roofline.achieved`.  Use as a context manager; call summary() after a synchronize."""

    active = None

    def __init__(self, tags=None):
        """`tags`: only these entry-point tags are bracketed (every HIP event is a marker packet in the queue that
        serialises the dispatch around it: ~3.5 us each, 8 % of the benchmark step when all ~40 calls are
        bracketed); None = all."""
        self.events = {}
        self.tags = None if tags is None else frozenset(tags)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[tag] = {"calls": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms), "min_ms": min(ms)}
        return out


def _call(tag, name, *args):
    t = KernelTimer.active
    if t is None or (t.tags is not None and tag not in t.tags):
        return _lib.call(name, *args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    t.events.setdefault(tag, []).append((e0, e1))


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.CocosHipError(
            f"{name}: expected a CUDA/HIP tensor; the correspondence hot path has no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------
# K1  centre + L2-normalise            (correspondence.py:277-280, :287-289)
# ------------------------------------------------------------------------------------------
class _CenterL2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, center_over_channels: int, eps: float):
        x = _chk(x, "center_l2norm: x")
        B, K, N = x.shape
        y = torch.empty_like(x)
        norm = torch.empty((B, N), device=x.device, dtype=torch.float32)
        center_over_channels = int(center_over_channels)      # 0 positions, 1 channels (PONO_C), 2 none
        row_ws = None if center_over_channels else torch.empty((B, K), device=x.device,
                                                                dtype=torch.float32)
        _call("center_l2norm_fwd", "cocos_center_l2norm_fwd", x.data_ptr(), y.data_ptr(), norm.data_ptr(),
                  _ptr(row_ws), B, K, N, int(center_over_channels), float(eps), _stream())
        ctx.save_for_backward(y, norm)
        ctx.cfg = (center_over_channels, float(eps))
        return y

    @staticmethod
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        center_over_channels, eps = ctx.cfg
        dy = _chk(dy, "center_l2norm: dy")
        B, K, N = y.shape
        dx = torch.empty_like(y)
        col_ws = row_ws = None
        if not center_over_channels:
            col_ws = torch.empty((B, N), device=y.device, dtype=torch.float32)
            row_ws = torch.empty((B, K), device=y.device, dtype=torch.float32)
        # max|dx| as a by-product: K0's backward (the consumer on the theta/phi path) needs it.  Mode 2
        # (feature_normalize) feeds MIOpen convolutions: nobody would pick the value up
        if PROJ_PRECISION == "f16x3" and center_over_channels != 2:
            cell = _zero_cell(dx.device)
            _call("center_l2norm_bwd", "cocos_center_l2norm_bwd_amax", y.data_ptr(), norm.data_ptr(), dy.data_ptr(),
                  dx.data_ptr(), _ptr(col_ws), _ptr(row_ws), B, K, N, int(center_over_channels), eps,
                  cell.data_ptr(), _stream())
            _remember_amax(dx, cell)
        else:
            _call("center_l2norm_bwd", "cocos_center_l2norm_bwd", y.data_ptr(), norm.data_ptr(), dy.data_ptr(),
                  dx.data_ptr(), _ptr(col_ws), _ptr(row_ws), B, K, N, int(center_over_channels),
                  eps, _stream())
        return dx, None, None


def center_l2norm(x: torch.Tensor, center_over_channels, eps: float = NORM_EPS):
    """x [B,K,N] -> (x - mean) / (||x - mean||_2 over K + eps); mean over K (True / 1: PONO_C), over N
    (False / 0) or no centring at all (2)."""
    return _CenterL2Norm.apply(x, int(center_over_channels), eps)


def feature_normalize(x: torch.Tensor, eps: float = NORM_EPS):
    """util.feature_normalize (util/util.py:31-34): x [B,C,h,w] / (||x||_2 over C + eps) — K1 without
    the centring; used on the adaptor outputs right before the ResidualBlocks (correspondence.py:247-248)."""
    B, C = x.shape[:2]
    return _CenterL2Norm.apply(x.reshape(B, C, -1), 2, eps).reshape(x.shape)




class _CenterL2NormPlanes(torch.autograd.Function):
    """K1 for the split flavour: x [B,256,N] -> operand planes (handed to the caller's OperandPlanes) + norms.  The autograd
    OUTPUT is only a handle: a view of x (no new memory) whose VALUES are x's, not the normalised tensor's — the split
    correlation kernels never read it, they read the planes; its gradient (d qn from K2's backward) comes back here."""

    @staticmethod
    def forward(ctx, x, center_over_channels: int, eps: float, planes, want_chan: bool):
        x = _chk(x, "center_l2norm_planes: x")
        B, K, N = x.shape
        half = dict(device=x.device, dtype=torch.float16)
        norm = torch.empty((B, N), device=x.device, dtype=torch.float32)
        ph, pl = torch.empty((B, N, K), **half), torch.empty((B, N, K), **half)
        ch = cl = None
        if want_chan:
            ch, cl = torch.empty((B, K, N), **half), torch.empty((B, K, N), **half)
        _call("center_l2norm_fwd", "cocos_center_l2norm_fwd_planes", x.data_ptr(), norm.data_ptr(), ph.data_ptr(),
              pl.data_ptr(), _ptr(ch), _ptr(cl), B, K, N, int(center_over_channels), float(eps), SPLIT_OPERAND_SCALE, _stream())
        handle = x.view_as(x)
        planes.put(handle, True, SPLIT_OPERAND_SCALE, ph, pl)
        if want_chan:
            planes.put(handle, False, SPLIT_OPERAND_SCALE, ch, cl)
        ctx.save_for_backward(norm, ch, cl)
        ctx.cfg = (int(center_over_channels), float(eps))
        return handle

    @staticmethod
    def backward(ctx, dy):
        norm, ch, cl = ctx.saved_tensors
        mode, eps = ctx.cfg
        dy = _chk(dy, "center_l2norm_planes: dy")
        B, K, N = dy.shape
        if ch is None:
            raise _lib.CocosHipError("center_l2norm_planes: backward without the channel-major planes (forward ran without grad)")
        dx = torch.empty_like(dy)
        cell = _zero_cell(dx.device) if PROJ_PRECISION == "f16x3" else None
        _call("center_l2norm_bwd", "cocos_center_l2norm_bwd_planes", ch.data_ptr(), cl.data_ptr(), norm.data_ptr(),
              dy.data_ptr(), dx.data_ptr(), B, K, N, mode, eps, SPLIT_OPERAND_SCALE, _ptr(cell), _stream())
        if cell is not None:
            _remember_amax(dx, cell)
        return dx, None, None, None, None


def center_l2norm_planes(x: torch.Tensor, center_over_channels, planes: OperandPlanes, eps: float = NORM_EPS,
                         want_chan=None):
    """K1 whose only products are the operand planes of the split correlation kernels (registered in `planes`) — see
    _CenterL2NormPlanes.  Returns the autograd handle to pass to corr_softmax_warp(..., planes=planes) as qn / kn.  Only
    for callers that take the split path (corr_split_ok).

    `want_chan`: also write the channel-major planes.  The K2 backward reads them of BOTH operands as soon as EITHER
    theta or phi is differentiated (d qn contracts dS with kn's planes and vice versa), so a caller with two operands
    passes its `keep` (= either requires grad) for both; None = this tensor's own requires_grad (single-operand use)."""
    if want_chan is None:
        want_chan = torch.is_grad_enabled() and x.requires_grad
    return _CenterL2NormPlanes.apply(x, int(center_over_channels), eps, planes, bool(want_chan))


def corr_split_ok(B, K, Nq, Nk, Cv, keep: bool) -> bool:
    """True when corr_softmax_warp takes the split-precision kernels for this shape (forward and, when `keep`, backward):
    the condition under which qn / kn may exist as operand planes only."""
    chunk = min(Cv, MAX_FUSED_SPLIT_CV)
    return (PRECISION == "f16x3" and K == FUSED_K and Nk % 4 == 0 and Nq % 4 == 0
            and (not keep or (_split_bwd_ok(B, Nq, Nk, chunk) and _split_bwd_ok(B, Nk, Nq, chunk))))


# ------------------------------------------------------------------------------------------
# K2  fused correlation -> softmax -> warp     (correspondence.py:291,:304,:307,:318)
# ------------------------------------------------------------------------------------------
class OperandPlanes:
    """f16 hi/lo operand planes of the theta/phi tensors of ONE forward call, made on first use and shared by the up
    to three launches that read them (row pass, column pass, second row pass).  Owned by the caller (hot_path's
    _Attention), so there is no process-global cache: under the reference's DataParallelWithCallback every replica
    thread has its own object (SURVEY.md §8b threading contract)."""

    def __init__(self):
        self._planes = {}

    def put(self, x: torch.Tensor, transpose: bool, scale: float, hi: torch.Tensor, lo: torch.Tensor):
        """Planes a producer kernel wrote itself (K1's planes flavour): later get() calls for `x` return them."""
        self._planes[(id(x), bool(transpose), float(scale))] = (x, x._version, hi, lo, True)

    def get(self, x: torch.Tensor, transpose: bool, scale: float):
        key = (id(x), bool(transpose), float(scale))
        ent = self._planes.get(key)
        if ent is not None and ent[0] is x and len(ent) > 4:
            # producer-made planes: x is only a handle (its VALUES are not what the planes hold), they cannot be re-made
            if ent[1] != x._version:
                raise _lib.CocosHipError("OperandPlanes: the tensor behind producer-made operand planes was modified in place")
            return ent[2], ent[3]
        if ent is None and any(k[0] == id(x) and len(e) > 4 and e[0] is x for k, e in self._planes.items()):
            # x is a producer's HANDLE (its values are the raw features, not the normalised tensor): splitting it here
            # would hand the kernels planes of the wrong tensor — the producer must be asked for this orientation
            raise _lib.CocosHipError(
                f"OperandPlanes: {'position' if transpose else 'channel'}-major planes (scale {scale}) of a tensor that exists "
                "as producer-made planes only were not written by the producer (center_l2norm_planes(..., want_chan=True))")
        if ent is None or ent[0] is not x or ent[1] != x._version:
            hi, lo = split_f16(x, transpose, scale)
            ent = self._planes[key] = (x, x._version, hi, lo)     # holds x: its id cannot be recycled meanwhile
        return ent[2], ent[3]

    def get_scaled(self, x: torch.Tensor, transpose: bool):
        """Planes with a DEVICE-side power-of-two scale from max|x| (operands without an a-priori magnitude):
        (hi, lo, scale_tensor).  The scale depends on x only, so both orientations of x share it."""
        key = (id(x), bool(transpose), "amax")
        ent = self._planes.get(key)
        if ent is None or ent[0] is not x or ent[1] != x._version:
            am = self._planes.get((id(x), "amax_cell"))
            if am is None or am[0] is not x or am[1] != x._version:
                am = self._planes[(id(x), "amax_cell")] = (x, x._version, absmax(x))
            hi, lo, sc = split_f16(x, transpose, amax=am[2])
            ent = self._planes[key] = (x, x._version, hi, lo, sc)
        return ent[2], ent[3], ent[4]

    def has(self, x: torch.Tensor) -> bool:
        return any(k[0] == id(x) and e[0] is x for k, e in self._planes.items())

    def __len__(self):
        return len(self._planes)


def split_f16(x: torch.Tensor, transpose: bool, scale: float = 1.0, cpad=None, amax=None):
    """x [B,C,N] fp32 -> (hi, lo) f16 planes with x*scale ~= hi + lo; [B,N,C] when `transpose`.
    `cpad` (transpose only): rows padded with zero channels to cpad.  `amax`: a 1-element CUDA tensor
    holding max|x|; the scale is then chosen on the device (power of two, max -> [2^9, 2^10)) and the
    call returns (hi, lo, scale_tensor)."""
    x = _chk(x, "split_f16: x")
    B, C, N = x.shape
    if cpad is not None or amax is not None:
        cp = C if cpad is None else int(cpad)
        shape = (B, N, cp) if transpose else (B, C, N)
        hi = torch.empty(shape, device=x.device, dtype=torch.float16)
        lo = torch.empty(shape, device=x.device, dtype=torch.float16)
        sc = torch.empty(1, device=x.device, dtype=torch.float32) if amax is not None else None
        _call("split_f16", "cocos_split_f16_ex", x.data_ptr(), hi.data_ptr(), lo.data_ptr(), B, C, N, cp,
              int(bool(transpose)), float(scale), _ptr(amax), _ptr(sc), _stream())
        return (hi, lo, sc) if amax is not None else (hi, lo)
    shape = (B, N, C) if transpose else (B, C, N)
    hi = torch.empty(shape, device=x.device, dtype=torch.float16)
    lo = torch.empty(shape, device=x.device, dtype=torch.float16)
    _call("split_f16", "cocos_split_f16", x.data_ptr(), hi.data_ptr(), lo.data_ptr(), B, C, N, int(bool(transpose)),
          float(scale), _stream())
    return hi, lo


def split_f16_chan_mask(x: torch.Tensor, amax: torch.Tensor, want_mask: bool):
    """x [B,C,N] -> (hi, lo, scale_tensor, mask_cell | None): the channel-major planes of split_f16(x, False, amax=amax) and,
    when `want_mask`, f16_plane_block_mask(lo), in ONE launch (cocos_split_f16_chan_mask; N % 4 == 0, else two launches)."""
    x = _chk(x, "split_f16_chan_mask: x")
    B, C, N = x.shape
    if N % 4 != 0 or C > 1024 or x.data_ptr() % 16 != 0:
        hi, lo, sc = split_f16(x, False, amax=amax)
        return hi, lo, sc, (f16_plane_block_mask(lo) if want_mask else None)
    hi = torch.empty((B, C, N), device=x.device, dtype=torch.float16)
    lo = torch.empty((B, C, N), device=x.device, dtype=torch.float16)
    sc = torch.empty(1, device=x.device, dtype=torch.float32)
    mask = _zero_cell(x.device) if want_mask else None          # fp32 zero = integer zero
    _call("split_f16", "cocos_split_f16_chan_mask", x.data_ptr(), hi.data_ptr(), lo.data_ptr(), B, C, N, amax.data_ptr(),
          sc.data_ptr(), _ptr(mask), _stream())
    return hi, lo, sc, mask


def f16_plane_block_mask(plane: torch.Tensor) -> torch.Tensor:
    """plane [B,C,N] f16 (channel-major) -> 1-element int32 CUDA tensor whose bit (c >> 5) says "channel block c//32 has
    a non-zero element" (cocos_f16_plane_block_mask; no host sync)."""
    B, C, N = plane.shape
    cell = _zero_cell(plane.device)                      # fp32 zero = integer zero
    _call("f16_plane_block_mask", "cocos_f16_plane_block_mask", plane.data_ptr(), B, C, N, cell.data_ptr(), _stream())
    return cell


def _recompute_chunk(B, Nq, Nk):
    """Keys per chunk of the recompute backward: whole 128-key groups (blocked planes), RECOMPUTE_CHUNK_BYTES per matrix."""
    per_key = max(B * Nq * 4, 1)
    nchunks = max(1, -(-(per_key * Nk) // max(RECOMPUTE_CHUNK_BYTES, 1)))      # equal chunks: the last one is not a sliver
    kc = max(128, -(-Nk // nchunks) // 128 * 128 + (128 if (-(-Nk // nchunks)) % 128 else 0))
    return min(kc, (Nk + 127) // 128 * 128)


def _saves_logits(B, Nq, Nk) -> bool:
    """Split flavour: does a training forward of this shape save its logits (else: LSE only, chunked recompute)?"""
    return (B * Nq * Nk * 4 <= MAX_SAVED_LOGITS_BYTES
            and _lib.load().cocos_corr_softmax_warp_saved_logits_bytes(1, Nq, Nk) < 2 ** 31 - 1)


def _split_bwd_ok(B, Nq, Nk, Cv):
    """Shapes the split-precision K2 backward takes (cocos_corr_softmax_warp_bwd_query_f16x3 + the planes GEMM) — with saved
    logits, or by recomputing them chunk by chunk (any size whose per-sample chunk stays below 2 GiB)."""
    if not (Nk % 8 == 0 and Nq % 8 == 0 and Cv <= MAX_FUSED_CV):
        return False
    if _saves_logits(B, Nq, Nk):
        return True
    return Nk % 128 == 0 and Nq % 32 == 0 and Nq * _recompute_chunk(B, Nq, Nk) * 4 < 2 ** 31 - 1


GEMM_SPLIT_K_BELOW = 200      # workgroups: a backward GEMM with fewer tiles than this splits its reduction (256 CUs to fill)


def _hgemm_planes(timer, ah, al, bh, bl, out, M, n, Kred, host, dev1, dev2, bmode, st):
    """out[b] (M x n) = A[b] (M x Kred) . B[b] (Kred x n) on the split GEMM (cocos_hgemm_f16x3), A = channel-major planes [B,M,Kred].
    With [query][key]-blocked B planes (bmode 2) and too few 256 x 128 output tiles to fill the chip (the Attention block:
    M = C/8 rows, B = 4 samples -> 128 workgroups on 256 CUs, each streaming all of Kred) the reduction is cut into S
    contiguous slices that run as S * B "samples" (a slice of a blocked plane is contiguous, so only A is re-laid out) and the
    S partial products are summed in a fixed order: deterministic."""
    B = out.shape[0]
    tiles = B * ((M + 255) // 256) * ((n + 127) // 128)
    S = 1
    while bmode == 2 and tiles * S < GEMM_SPLIT_K_BELOW and S < 8 and Kred % (64 * S) == 0:
        S *= 2
    if S == 1:
        _call(timer, "cocos_hgemm_f16x3", ah.data_ptr(), al.data_ptr(), bh.data_ptr(), bl.data_ptr(), out.data_ptr(), B, M, n, Kred,
              host, _ptr(dev1), _ptr(dev2), bmode, st)
        return
    a2h = ah.view(B, M, S, Kred // S).transpose(1, 2).contiguous()
    a2l = al.view(B, M, S, Kred // S).transpose(1, 2).contiguous()
    part = torch.empty((B, S, M, n), device=out.device, dtype=torch.float32)
    _call(timer, "cocos_hgemm_f16x3", a2h.data_ptr(), a2l.data_ptr(), bh.data_ptr(), bl.data_ptr(), part.data_ptr(), B * S, M, n,
          Kred // S, host, _ptr(dev1), _ptr(dev2), bmode, st)
    torch.sum(part, dim=1, out=out)


def _key_gemm(qch, qcl, dsh, dsl, dk, n, Nq, k_active, qks, ds_scale, gemm_b, st):
    """dk[b] = (q_scale qn)[b] . dS''[b] on the split GEMM (K5's key side); qch / qcl / dk have the REAL channel count (the
    magnitude-free flavour's K < 256: M = K rows, see _hgemm_planes for what that means for the launch)."""
    K = dk.shape[1]
    assert qch.shape[1] == K and (not k_active or k_active == K)
    _hgemm_planes("corr_softmax_warp_bwd_key_from_ds", qch, qcl, dsh, dsl, dk, K, n, Nq, 1.0 if qks else 1.0 / SPLIT_OPERAND_SCALE,
                  ds_scale, qks[0] if qks else None, gemm_b, st)


class _CorrSoftmaxWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qn, kn, v, inv_temperature: float, keep_logits: bool, planes=None):
        qn, kn, v = _chk(qn, "qn"), _chk(kn, "kn"), _chk(v, "v")
        B, K, Nq = qn.shape
        Bk, Kk, Nk = kn.shape
        Bv, Cv, Nv = v.shape
        if (Bk, Kk) != (B, K) or (Bv, Nv) != (B, Nk):
            raise ValueError(f"corr_softmax_warp: shape mismatch qn{tuple(qn.shape)} "
                             f"kn{tuple(kn.shape)} v{tuple(v.shape)}")
        out = torch.empty((B, Cv, Nq), device=qn.device, dtype=torch.float32)
        lse = torch.empty((B, Nq), device=qn.device, dtype=torch.float32)
        # training: keep the logits for the query-side backward (cheaper than recomputing them, see
        # corr_fused_fwd.hip); inference never materialises anything HWxHW
        logits_t = None
        ctx.split = planes is not None
        ctx.v_amax = None
        ctx.v_lomask = None
        if planes is not None:       # split-precision flavour: same outputs, f16x3 matrix products
            qh, ql, kh, kl, vh, vl, v_scale, v_amax, v_lomask = planes[:9]
            qk_scales = planes[13] if len(planes) > 13 else None      # (q_scale, k_scale) device scalars, or None
            ctx.k_active = planes[14] if len(planes) > 14 else 0     # K < 256 (magnitude-free flavour): planes zero-padded to 256
            Kp = FUSED_K if ctx.k_active else K
            ctx.recompute = bool(keep_logits) and not _saves_logits(B, Nq, Nk)
            if keep_logits and not ctx.recompute:   # the forward's private tile-blocked layout (cocos_hip.h), opaque here
                nbytes = _lib.load().cocos_corr_softmax_warp_saved_logits_bytes(B, Nq, Nk)
                logits_t = torch.empty(nbytes // 4, device=qn.device, dtype=torch.float32)
            # magnitude-free flavour (device-side operand scales): per-row (m, r) for its backward (cocos_hip.h)
            ctx.rowstat = torch.empty((B, 3, Nq), device=qn.device, dtype=torch.float32) if (qk_scales and keep_logits) else None
            # ... and, with saved logits (which this flavour stores RELATIVE to the running maximum), that maximum per 32-key tile
            ctx.mtile = (torch.empty((B, (Nk + 31) // 32, 2, Nq), device=qn.device, dtype=torch.float32)
                         if (qk_scales and logits_t is not None) else None)
            _call("corr_softmax_warp_fwd", "cocos_corr_softmax_warp_fwd_f16x3_ex", qh.data_ptr(), ql.data_ptr(),
                  kh.data_ptr(), kl.data_ptr(), vh.data_ptr(), vl.data_ptr(), out.data_ptr(), lse.data_ptr(),
                  _ptr(logits_t), v_scale.data_ptr(), _ptr(v_lomask), B, Kp, Nq, Nk, Cv, float(inv_temperature),
                  SPLIT_OPERAND_SCALE, _ptr(qk_scales[0] if qk_scales else None), _ptr(qk_scales[1] if qk_scales else None),
                  _ptr(ctx.rowstat), _ptr(ctx.mtile), ctx.k_active, _stream())
            ctx.v_amax = v_amax
            ctx.v_lomask = v_lomask
            ctx.qk_scales = qk_scales
        else:
            keep = (keep_logits and B * Nq * Nk * 4 <= MAX_DS_WORKSPACE_BYTES and Nq * Nk * 4 < 2 ** 31 - 1)
            logits_t = torch.empty((B, Nk, Nq), device=qn.device, dtype=torch.float32) if keep else None
            _call("corr_softmax_warp_fwd", "cocos_corr_softmax_warp_fwd", qn.data_ptr(), kn.data_ptr(),
                  v.data_ptr(), out.data_ptr(), lse.data_ptr(), _ptr(logits_t), B, K, Nq, Nk, Cv,
                  float(inv_temperature), _stream())
        ctx.save_for_backward(qn, kn, v, out, lse)
        ctx.logits_t = logits_t
        ctx.inv_t = float(inv_temperature)
        # channel-major planes of k_scale*qn, k_scale*kn for the split-precision backward (when given); the recompute
        # backward also needs the position-major ones the forward multiplied
        recompute = planes is not None and getattr(ctx, "recompute", False)
        ctx.cplanes = planes[9:13] if (planes is not None and len(planes) > 9 and (logits_t is not None or recompute)) else None
        ctx.pplanes = planes[0:4] if recompute else None
        return out

    @staticmethod
    def backward(ctx, dout):
        qn, kn, v, out, lse = ctx.saved_tensors
        dout = _chk(dout, "dout")
        B, K, Nq = qn.shape
        Nk, Cv = kn.shape[2], v.shape[1]
        need_q, need_k, need_v = ctx.needs_input_grad[:3]
        dqn = torch.empty_like(qn) if need_q else None
        dkn = torch.empty_like(kn) if (need_k or need_v) else None
        dv = torch.empty_like(v) if need_v else None
        st = _stream()
        logits_t = ctx.logits_t
        if ctx.split and getattr(ctx, "recompute", False):
            return _CorrSoftmaxWarp._backward_recompute(ctx, qn, kn, v, out, lse, dout, dqn, dkn, dv, need_k)
        if ctx.split and logits_t is not None:
            # split-precision backward: everything on the f16 MFMA, fp32-class accuracy (see cocos_hip.h).  The
            # forward only takes this flavour (and saves its private logits layout) for shapes this branch takes.
            qch, qcl, kch, kcl = ctx.cplanes
            qks = ctx.qk_scales
            cvp = (Cv + 31) // 32 * 32
            g_amax, v_amax = _recall_amax(dout), ctx.v_amax   # max|v| was taken once, by the forward; max|dout| comes
            if g_amax is None:                                # with dout when concat_channels_amax produced it
                g_amax = absmax(dout)
            (gph, gpl, g_scale), (vph, vpl, v_scale) = _split_pair_transposed(dout, g_amax, v, v_amax, cvp)
            half = dict(device=qn.device, dtype=torch.float16)
            want_k = dkn is not None
            dsh = dsl = psh = psl = None
            if want_k:
                dsh, dsl = torch.empty((B, Nk, Nq), **half), torch.empty((B, Nk, Nq), **half)
            if dv is not None:      # cycle terms: V itself is differentiated, the key side needs P as well
                psh, psl = torch.empty((B, Nk, Nq), **half), torch.empty((B, Nk, Nq), **half)
            ds_scale = torch.empty(1, device=qn.device, dtype=torch.float32)
            blocked = int(Nk % 128 == 0 and Nq % 32 == 0)     # [query][key]-blocked dS'' / P planes (see cocos_hip.h)
            gemm_b = 2 * blocked                               # ... which the GEMM reads as b_blocked = 2
            ka = getattr(ctx, "k_active", 0)
            Kp = FUSED_K if ka else K
            if ka:                   # the kernel writes all 256 channel rows (zeros past K): dqn is the view of the real ones
                dqn_buf = torch.empty((B, Kp, Nq), device=qn.device, dtype=torch.float32)
                dqn = dqn_buf[:, :K] if dqn is not None else None
            elif dqn is None:        # the query kernel always accumulates dqn; scratch when nobody wants it
                dqn_buf = torch.empty_like(qn)
            else:
                dqn_buf = dqn
            d_pre = _rowdot_cached(dout, out) if BWD_D_PRECOMPUTED else None
            _call("corr_softmax_warp_bwd_query", "cocos_corr_softmax_warp_bwd_query_f16x3_ex", kch.data_ptr(),
                  kcl.data_ptr(), vph.data_ptr(), vpl.data_ptr(), gph.data_ptr(), gpl.data_ptr(),
                  g_scale.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), logits_t.data_ptr(),
                  dqn_buf.data_ptr(), _ptr(dsh), _ptr(dsl), _ptr(psh), _ptr(psl), v_amax.data_ptr(),
                  v_scale.data_ptr(), ds_scale.data_ptr(), _ptr(ctx.v_lomask), B, Kp, Nq, Nk, Cv, cvp, ctx.inv_t,
                  SPLIT_OPERAND_SCALE, _ptr(qks[0] if qks else None), _ptr(qks[1] if qks else None), blocked,
                  _ptr(getattr(ctx, "rowstat", None)), _ptr(getattr(ctx, "mtile", None)), _ptr(d_pre), ka, st)
            if want_k:    # A = the channel-major planes of q_scale * qn
                _key_gemm(qch, qcl, dsh, dsl, dkn, Nk, Nq, ka, qks, ds_scale, gemm_b, st)
            if dv is not None:      # dv[c,j] = sum_i dout[c,i] P[i,j]
                gch, gcl, _ = split_f16(dout, False, amax=g_amax)
                _hgemm_planes("corr_softmax_warp_bwd_dv", gch, gcl, psh, psl, dv, Cv, Nk, Nq, 1.0 / 16384.0, g_scale, None, gemm_b, st)
            return dqn, (dkn if need_k else None), dv, None, None, None
        # key side: GEMM over a materialised dS^T when it pays and fits (see cocos_hip.h), else the
        # flash-style kernel that recomputes the logits (always when dv is wanted: it needs P)
        ds_bytes = B * Nq * Nk * 4
        via_gemm = (dqn is not None and dkn is not None and dv is None
                    and ds_bytes <= MAX_DS_WORKSPACE_BYTES and Nq * Nk * 4 < 2 ** 31 - 1)
        key_recompute = dkn is not None and not via_gemm
        dvec = None
        if key_recompute or (dqn is not None and logits_t is None):
            dvec = torch.empty((B, Nq), device=qn.device, dtype=torch.float32)
            _call("corr_softmax_warp_bwd_prepare", "cocos_corr_softmax_warp_bwd_prepare",
                  out.data_ptr(), dout.data_ptr(), dvec.data_ptr(), B, Nq, Cv, st)
        dims = (B, K, Nq, Nk, Cv, ctx.inv_t, st)
        ds_t = torch.empty((B, Nk, Nq), device=qn.device, dtype=torch.float32) if via_gemm else None
        if dqn is not None:
            _call("corr_softmax_warp_bwd_query", "cocos_corr_softmax_warp_bwd_query", qn.data_ptr(),
                  kn.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(),
                  _ptr(dvec), _ptr(logits_t), _ptr(dqn), _ptr(ds_t), *dims)
        if via_gemm:
            _call("corr_softmax_warp_bwd_key_from_ds", "cocos_corr_softmax_warp_bwd_key_from_ds",
                  qn.data_ptr(), ds_t.data_ptr(), dkn.data_ptr(), B, K, Nq, Nk, st)
        elif dkn is not None:
            _call("corr_softmax_warp_bwd_key", "cocos_corr_softmax_warp_bwd_key", qn.data_ptr(),
                  kn.data_ptr(), v.data_ptr(), lse.data_ptr(), dout.data_ptr(), dvec.data_ptr(),
                  _ptr(dkn), _ptr(dv), *dims)
        return dqn, (dkn if need_k else None), dv, None, None, None


def _corr_bwd_recompute(ctx, qn, kn, v, out, lse, dout, dqn, dkn, dv, need_k):
    """Split-precision backward WITHOUT saved logits (the forward kept the row LSE only): the keys are walked in chunks of
    _recompute_chunk() keys; per chunk the forward kernel is run again on the resident operand planes just for its logits
    (one zero value channel: 6 instead of 30 P.V MFMAs per tile; its softmax output is discarded — the RAW logits it
    saves do not depend on it), then the query kernel (global LSE and D: a chunk's dS'' is exact) and the key-side GEMM(s)
    for that slice of d kn / d v; d qn accumulates over the chunks.  HWxHW scratch: RECOMPUTE_CHUNK_BYTES per matrix
    instead of B*Nq*Nk*4 — BASELINE config 5 (HW = 16384, B = 2): 2 x 0.5 GiB instead of 2 x 2 GiB.  Same kernels, same
    arithmetic per element as the saved-logits chain (bit-identical dS'')."""
    qh, ql, kh, kl = ctx.pplanes
    qch, qcl, kch, kcl = ctx.cplanes
    qks = ctx.qk_scales
    B, K, Nq = qn.shape
    Nk, Cv = kn.shape[2], v.shape[1]
    st = _stream()
    lib = _lib.load()
    cvp = (Cv + 31) // 32 * 32
    g_amax, v_amax = _recall_amax(dout), ctx.v_amax
    if g_amax is None:
        g_amax = absmax(dout)
    gph, gpl, g_scale = split_f16(dout, True, cpad=cvp, amax=g_amax)
    vph, vpl, v_scale = split_f16(v, True, cpad=cvp, amax=v_amax)
    gch = gcl = None
    if dv is not None:
        gch, gcl, _ = split_f16(dout, False, amax=g_amax)
    dev, half = qn.device, dict(device=qn.device, dtype=torch.float16)
    want_k = dkn is not None
    kc = _recompute_chunk(B, Nq, Nk)
    ds_scale = torch.empty(1, device=dev, dtype=torch.float32)
    d_pre = _rowdot(dout, out) if BWD_D_PRECOMPUTED else None
    ka = getattr(ctx, "k_active", 0)
    Kp = FUSED_K if ka else K        # (K < 256, magnitude-free flavour: the planes are zero-padded to the kernels' 256 channels)
    dq_acc = None
    # scratch shared by the chunks (sized for a full chunk)
    o1 = torch.empty((B, 1, Nq), device=dev, dtype=torch.float32)
    l1 = torch.empty((B, Nq), device=dev, dtype=torch.float32)
    for k0 in range(0, Nk, kc):
        n = min(kc, Nk - k0)
        khc, klc = kh[:, k0:k0 + n].contiguous(), kl[:, k0:k0 + n].contiguous()
        vz = torch.zeros((B, 1, n), **half)                  # one zero value channel (hi and lo plane alike)
        lg = torch.empty(lib.cocos_corr_softmax_warp_saved_logits_bytes(B, Nq, n) // 4, device=dev, dtype=torch.float32)
        rowstat = getattr(ctx, "rowstat", None)           # magnitude-free flavour: relative logits + their per-tile reference
        mt = torch.empty((B, (n + 31) // 32, 2, Nq), device=dev, dtype=torch.float32) if rowstat is not None else None
        _call("corr_softmax_warp_recompute", "cocos_corr_softmax_warp_fwd_f16x3_ex", qh.data_ptr(), ql.data_ptr(), khc.data_ptr(),
              klc.data_ptr(), vz.data_ptr(), vz.data_ptr(), o1.data_ptr(), l1.data_ptr(), lg.data_ptr(), None, None, B, Kp, Nq, n, 1,
              ctx.inv_t, SPLIT_OPERAND_SCALE, _ptr(qks[0] if qks else None), _ptr(qks[1] if qks else None), None, _ptr(mt), ka, st)
        kchc, kclc = kch[:, :, k0:k0 + n].contiguous(), kcl[:, :, k0:k0 + n].contiguous()
        vphc, vplc = vph[:, k0:k0 + n].contiguous(), vpl[:, k0:k0 + n].contiguous()
        dsh = dsl = psh = psl = None
        if want_k:
            dsh, dsl = torch.empty((B, n, Nq), **half), torch.empty((B, n, Nq), **half)
        if dv is not None:
            psh, psl = torch.empty((B, n, Nq), **half), torch.empty((B, n, Nq), **half)
        blocked = int(n % 128 == 0 and Nq % 32 == 0)
        dq_c = torch.empty((B, Kp, Nq), device=dev, dtype=torch.float32)
        _call("corr_softmax_warp_bwd_query", "cocos_corr_softmax_warp_bwd_query_f16x3_ex", kchc.data_ptr(), kclc.data_ptr(),
              vphc.data_ptr(), vplc.data_ptr(), gph.data_ptr(), gpl.data_ptr(), g_scale.data_ptr(), out.data_ptr(), dout.data_ptr(),
              lse.data_ptr(), lg.data_ptr(), dq_c.data_ptr(), _ptr(dsh), _ptr(dsl), _ptr(psh), _ptr(psl), v_amax.data_ptr(),
              v_scale.data_ptr(), ds_scale.data_ptr(), _ptr(ctx.v_lomask), B, Kp, Nq, n, Cv, cvp, ctx.inv_t, SPLIT_OPERAND_SCALE,
              _ptr(qks[0] if qks else None), _ptr(qks[1] if qks else None), blocked, _ptr(rowstat), _ptr(mt), _ptr(d_pre), ka, st)
        del lg
        if ka:                   # (only the real rows are written, see cocos_hip.h)
            dq_c = dq_c[:, :K]
        dq_acc = dq_c if dq_acc is None else dq_acc.add_(dq_c)
        if want_k:
            dk_c = dkn if n == Nk else torch.empty((B, K, n), device=dev, dtype=torch.float32)
            _key_gemm(qch, qcl, dsh, dsl, dk_c, n, Nq, ka, qks, ds_scale, 2 * blocked, st)
            if dk_c is not dkn:
                dkn[:, :, k0:k0 + n] = dk_c
        if dv is not None:
            dv_c = dv if n == Nk else torch.empty((B, Cv, n), device=dev, dtype=torch.float32)
            _hgemm_planes("corr_softmax_warp_bwd_dv", gch, gcl, psh, psl, dv_c, Cv, n, Nq, 1.0 / 16384.0, g_scale, None, 2 * blocked, st)
            if dv_c is not dv:
                dv[:, :, k0:k0 + n] = dv_c
    if dqn is not None:
        dqn = dq_acc
    return dqn, (dkn if need_k else None), dv, None, None, None


_CorrSoftmaxWarp._backward_recompute = staticmethod(_corr_bwd_recompute)


def _rowdot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """D[b,i] = sum_c a[b,c,i] b[b,c,i] (fp64 accumulation) -> [B,N]: the D of the softmax backward (cocos_rowdot_f64)."""
    B, C, N = a.shape
    d = torch.empty((B, N), device=a.device, dtype=torch.float32)
    _call("rowdot", "cocos_rowdot_f64", a.data_ptr(), b.data_ptr(), d.data_ptr(), B, C, N, _stream())
    return d


def _rowdot_cached(dout: torch.Tensor, out: torch.Tensor):
    """D of the softmax backward: the one the producer of `dout` left behind (warp_head's backward computes it in the pass that
    writes dout), else one cocos_rowdot_f64 pass."""
    ent, _tls.known_rowdot = _tls.known_rowdot, None
    if ent is not None:
        ref, version, out_ptr, d = ent
        src = ref()
        if (src is not None and src.data_ptr() == dout.data_ptr() and src.shape == dout.shape and dout._version == version
                and out.data_ptr() == out_ptr and dout.is_contiguous()):
            return d
    return _rowdot(dout, out)


def _split_pair_transposed(x0, amax0, x1, amax1, cpad):
    """split_f16(x0, True, cpad=cpad, amax=amax0) and the same of x1 (same shape) in ONE launch: ((hi, lo, scale), (hi, lo, scale))."""
    x0, x1 = _chk(x0, "split_f16: x"), _chk(x1, "split_f16: x")
    B, C, N = x0.shape
    if x1.shape != x0.shape or 2 * B > 65535:
        return split_f16(x0, True, cpad=cpad, amax=amax0), split_f16(x1, True, cpad=cpad, amax=amax1)
    half = dict(device=x0.device, dtype=torch.float16)
    res, args = [], []
    for x, am in ((x0, amax0), (x1, amax1)):
        hi, lo = torch.empty((B, N, cpad), **half), torch.empty((B, N, cpad), **half)
        sc = torch.empty(1, device=x0.device, dtype=torch.float32)
        res.append((hi, lo, sc))
        args += [x.data_ptr(), hi.data_ptr(), lo.data_ptr(), am.data_ptr(), sc.data_ptr()]
    _call("split_f16", "cocos_split_f16_transpose_pair", *args, B, C, N, int(cpad), _stream())
    return res[0], res[1]


def _wants_logits(qn, kn):
    # decided OUTSIDE the Function (inside it grad mode is always off): only a pass that will be
    # differentiated w.r.t. theta/phi saves the logits; inference never materialises anything HWxHW
    return torch.is_grad_enabled() and (qn.requires_grad or kn.requires_grad)


def corr_softmax_warp(qn, kn, v, inv_temperature: float, planes: OperandPlanes | None = None, operand_amax: bool = False,
                      precision: str | None = None):
    """out[b,c,i] = sum_j softmax_j(<qn[b,:,i], kn[b,:,j]> * inv_temperature) * v[b,c,j].

    qn [B,256,Nq], kn [B,256,Nk], v [B,Cv,Nk] -> [B,Cv,Nq].  Wider V is processed in chunks of
    159 channels (each chunk recomputes the logits; no materialisation).  `planes`: the caller's per-forward
    OperandPlanes (theta/phi planes shared by several launches); None = made for this call only.  `operand_amax`: qn / kn
    are NOT unit-norm columns (softmax_attention): their planes get device-side power-of-two scales from max|.| instead of
    the fixed SPLIT_OPERAND_SCALE (split flavour only; raises if this shape cannot take it).  `precision`: overrides the
    module-level PRECISION for this call (a per-call argument, not a global: threads do not see each other's choice).
    With operand_amax K may be < 256 (softmax_attention: K = C/8): the planes are zero-padded and the kernels' K = 32 / 64
    instantiations skip the matrix steps, fragment reads and fetches of the padding."""
    B, K, Nq = qn.shape
    Nk, Cv = kn.shape[2], v.shape[1]
    keep = _wants_logits(qn, kn)
    if operand_amax and not keep and torch.is_grad_enabled() and v.requires_grad:
        # only V is differentiated (frozen / detached theta, phi of an Attention block): the backward still needs P, and for K < 256
        # only the split chain (saved logits -> P planes -> the dv GEMM) exists — the exact-fp32 kernels are K = 256 only
        keep = True
    prec = PRECISION if precision is None else precision
    if prec not in ("fp32", "f16x3"):
        raise ValueError(f"cocosnet_amd.ops.PRECISION = {prec!r}: expected 'fp32' or 'f16x3'")
    chunk = min(Cv, MAX_FUSED_SPLIT_CV)
    # the split flavour saves its logits in a private layout only its own backward reads: a training pass takes it
    # only for shapes that backward takes too (otherwise the exact-fp32 kernels run, forward and backward)
    split = (prec == "f16x3" and (K == FUSED_K or (operand_amax and K < FUSED_K)) and Nk % 4 == 0
             and (not keep or _split_bwd_ok(B, Nq, Nk, chunk)))
    if planes is None:
        planes = OperandPlanes()
    if operand_amax and not split:
        raise _lib.CocosHipError("corr_softmax_warp: operand_amax needs the split-precision kernels for this shape")
    if not split and (planes.has(qn) or planes.has(kn)):
        raise _lib.CocosHipError("corr_softmax_warp: qn / kn exist as operand planes only (center_l2norm_planes) but this "
                                 "shape does not take the split-precision kernels")

    qk_made = []

    def qk_planes():
        # the magnitude-free flavour's q / k planes, once per call (the value chunks share them).  K <= 256 real channels:
        # the position-major planes the forward multiplies are written zero-padded to 256 by the split kernel itself; of the
        # channel-major ones (backward) q's keep the real K rows (the key-side GEMM's M) and k's are padded (the query
        # kernel stages 256-row tiles, fetching only the real rows)
        if not qk_made:
            qa, ka_ = absmax(qn), absmax(kn)
            qh, ql, qs = split_f16(qn, True, cpad=FUSED_K, amax=qa)
            kh, kl, ks = split_f16(kn, True, cpad=FUSED_K, amax=ka_)
            cpl = (None,) * 4
            if keep:
                qch, qcl, _ = split_f16(qn, False, amax=qa)
                kch, kcl, _ = split_f16(kn, False, amax=ka_)
                if K < FUSED_K:
                    pad = lambda t: torch.cat([t, t.new_zeros((B, FUSED_K - K, Nk))], dim=1)
                    kch, kcl = pad(kch), pad(kcl)
                cpl = (qch, qcl, kch, kcl)
            qk_made.append((qh, ql, kh, kl, *cpl, (qs, ks), K if K < FUSED_K else 0))
        return qk_made[0]

    def run(vv):
        pl = None
        if split:   # operand planes: theta/phi once per forward (shared through `planes`), V per launch
            with torch.no_grad():
                # V has no a-priori magnitude in a general forward() call: its planes are normalised by a
                # device-side power of two like every other operand (one max|v| pass, reused by the backward)
                v_amax = _recall_amax(vv)
                if v_amax is None:
                    v_amax = absmax(vv)
                # V planes, and which 32-channel blocks of V have a non-zero lo plane (one-hot labels / masks are exact in
                # f16: theirs is all zero and the kernels skip it) — the same launch, only when it can pay
                # (the magnitude-free flavour — operand_amax — takes no lo mask: its kernels are the single-flavour instantiations)
                vh, vl, v_scale, v_lomask = split_f16_chan_mask(vv, v_amax, VALUE_LO_SKIP and vv.shape[1] > 32 and not operand_amax)
                if operand_amax:
                    pl = (*qk_planes()[:4], vh, vl, v_scale, v_amax, v_lomask, *qk_planes()[4:])
                else:
                    pl = (*planes.get(qn, True, SPLIT_OPERAND_SCALE), *planes.get(kn, True, SPLIT_OPERAND_SCALE),
                          vh, vl, v_scale, v_amax, v_lomask)
                    if keep:   # the backward wants the channel-major planes as well
                        pl += (*planes.get(qn, False, SPLIT_OPERAND_SCALE), *planes.get(kn, False, SPLIT_OPERAND_SCALE))
        return _CorrSoftmaxWarp.apply(qn, kn, vv, inv_temperature, keep, pl)

    limit = MAX_FUSED_SPLIT_CV if split else MAX_FUSED_CV
    if Cv <= limit:
        return run(v)
    return torch.cat([run(v[:, c0:c0 + limit]) for c0 in range(0, Cv, limit)], dim=1)


# ------------------------------------------------------------------------------------------
# K3  materialised correlation          (correspondence.py:291 + :304; return_corr / WTA / mk != 1)
# ------------------------------------------------------------------------------------------
class _CorrMaterialize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qn, kn, scale: float):
        qn, kn = _chk(qn, "qn"), _chk(kn, "kn")
        B, K, Nq = qn.shape
        Nk = kn.shape[2]
        if kn.shape[:2] != (B, K):
            raise ValueError(f"corr_materialize: shape mismatch qn{tuple(qn.shape)} kn{tuple(kn.shape)}")
        f = torch.empty((B, Nq, Nk), device=qn.device, dtype=torch.float32)
        ctx.amax = None
        if PRECISION == "f16x3" and min(K * Nq, K * Nk) >= 4:
            ctx.amax = (absmax(qn), absmax(kn))
            if K % 8 == 0:
                # both operands are k-major [K][positions]: the split GEMM would transpose them through 2-byte LDS
                # writes; position-major planes + the planes GEMM do the same product in half the time
                qh, ql, qs = split_f16(qn, True, amax=ctx.amax[0])
                kh, kl, ks = split_f16(kn, True, amax=ctx.amax[1])
                _call("corr_materialize", "cocos_hgemm_f16x3", qh.data_ptr(), ql.data_ptr(), kh.data_ptr(), kl.data_ptr(),
                      f.data_ptr(), B, Nq, Nk, K, float(scale), qs.data_ptr(), ks.data_ptr(), 0, _stream())
            else:                                                   # split GEMM (sgemm_f16x3.hip)
                _call("corr_materialize", "cocos_corr_materialize_f16x3", qn.data_ptr(), kn.data_ptr(), f.data_ptr(), B,
                      K, Nq, Nk, float(scale), ctx.amax[0].data_ptr(), ctx.amax[1].data_ptr(), _stream())
        else:
            _call("corr_materialize", "cocos_corr_materialize", qn.data_ptr(), kn.data_ptr(), f.data_ptr(), B, K, Nq,
                  Nk, float(scale), _stream())
        ctx.save_for_backward(qn, kn)
        ctx.scale = float(scale)
        return f

    @staticmethod
    def backward(ctx, df):
        qn, kn = ctx.saved_tensors
        df = _chk(df, "df")
        B, K, Nq = qn.shape
        Nk = kn.shape[2]
        need_q, need_k = ctx.needs_input_grad[:2]
        dqn = torch.empty_like(qn) if need_q else None
        dkn = torch.empty_like(kn) if need_k else None
        if ctx.amax is not None and Nq * Nk >= 4:
            qa, ka = ctx.amax
            ga = _recall_amax(df)            # left by the kernel that wrote df (K6's backward), else one pass
            if ga is None:
                ga = absmax(df)
            _call("corr_materialize_bwd", "cocos_corr_materialize_bwd_f16x3", qn.data_ptr(), kn.data_ptr(),
                  df.data_ptr(), _ptr(dqn), _ptr(dkn), B, K, Nq, Nk, ctx.scale, qa.data_ptr(), ka.data_ptr(),
                  ga.data_ptr(), _stream())
        else:
            _call("corr_materialize_bwd", "cocos_corr_materialize_bwd", qn.data_ptr(), kn.data_ptr(), df.data_ptr(),
                  _ptr(dqn), _ptr(dkn), B, K, Nq, Nk, ctx.scale, _stream())
        return dqn, dkn, None


def corr_materialize(qn, kn, scale: float = 1.0):
    """f[b,i,j] = scale * <qn[b,:,i], kn[b,:,j]>   -> [B,Nq,Nk] (any K)."""
    return _CorrMaterialize.apply(qn, kn, scale)


# ------------------------------------------------------------------------------------------
# K4  row softmax on a materialised matrix   (correspondence.py:307)
# ------------------------------------------------------------------------------------------
class _RowSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s):
        s = _chk(s, "row_softmax: s")
        cols = s.shape[-1]
        rows = s.numel() // cols
        p = torch.empty_like(s)
        _call("row_softmax_fwd", "cocos_row_softmax_fwd", s.data_ptr(), p.data_ptr(), rows, cols, _stream())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        dp = _chk(dp, "row_softmax: dp")
        cols = p.shape[-1]
        rows = p.numel() // cols
        ds = torch.empty_like(p)
        _call("row_softmax_bwd", "cocos_row_softmax_bwd", p.data_ptr(), dp.data_ptr(), ds.data_ptr(), rows, cols,
                  _stream())
        return ds


def row_softmax(s):
    """softmax over the last dimension of a materialised matrix."""
    return _RowSoftmax.apply(s)


# ------------------------------------------------------------------------------------------
# K5  P @ V on a materialised P            (correspondence.py:318 on the fallback path)
# ------------------------------------------------------------------------------------------
class _WarpMaterialized(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, v):
        p, v = _chk(p, "p"), _chk(v, "v")
        B, Nq, Nk = p.shape
        Cv = v.shape[1]
        if v.shape[0] != B or v.shape[2] != Nk:
            raise ValueError(f"warp_materialized: shape mismatch p{tuple(p.shape)} v{tuple(v.shape)}")
        out = torch.empty((B, Cv, Nq), device=p.device, dtype=torch.float32)
        _call("warp_materialized_fwd", "cocos_warp_materialized_fwd", p.data_ptr(), v.data_ptr(), out.data_ptr(), B, Nq,
                  Nk, Cv, _stream())
        ctx.save_for_backward(p, v)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, v = ctx.saved_tensors
        dout = _chk(dout, "dout")
        B, Nq, Nk = p.shape
        Cv = v.shape[1]
        need_p, need_v = ctx.needs_input_grad
        dp = torch.empty_like(p) if need_p else None
        dv = torch.empty_like(v) if need_v else None
        _call("warp_materialized_bwd", "cocos_warp_materialized_bwd", p.data_ptr(), v.data_ptr(), dout.data_ptr(),
                  _ptr(dp), _ptr(dv), B, Nq, Nk, Cv, _stream())
        return dp, dv


def warp_materialized(p, v):
    """out[b,c,i] = sum_j p[b,i,j] v[b,c,j]   (p [B,Nq,Nk], v [B,Cv,Nk]) -> [B,Cv,Nq]."""
    return _WarpMaterialized.apply(p, v)


# ------------------------------------------------------------------------------------------
# K0  theta / phi 1x1 projections            (correspondence.py:272, :282)
# ------------------------------------------------------------------------------------------
class _ThreadState(threading.local):
    """Host-side scratch that must not be shared between threads (SURVEY.md §8b threading contract: under the
    reference's DataParallelWithCallback every replica's forward runs in its own Python thread, and autograd runs
    each device's backward in its own thread): the pool of pre-zeroed max|x| cells and the producer -> consumer
    max|x| table.  threading.local gives every thread its own instance; no locks on the hot path."""

    def __init__(self):
        self.zero_pool = {}      # (device, stream) -> [zeros tensor, next free cell]
        self.known_amax = {}     # (device, storage pointer) -> (the tensor, its version, amax cell)
        self.known_rowdot = None  # (weakref of dout, its version, data_ptr of out, D [B,N]) left by warp_head's backward


_tls = _ThreadState()
_KNOWN_AMAX_MAX = 8      # entries per thread that hold their tensor
_KNOWN_AMAX_WEAK_MAX = 48      # ... and entries that only watch theirs (see _remember_amax)


#: tensors at least this large are remembered through a weak reference only (see _remember_amax)
_AMAX_WEAK_BYTES = 32 << 20


def _remember_amax(t: torch.Tensor, cell: torch.Tensor, weak: bool = False):
    """A producer kernel computed max|t| while writing t: keep it for the consumer (autograd hands the gradient on
    as a view of the same storage; producer and consumer of one device's backward run in the same thread).  Small
    tensors are held by the entry itself, so their address cannot be recycled for other data while the entry exists;
    it is dropped when the consumer picks it up, or when younger ones arrive.  LARGE tensors (the HWxHW box-adjoint
    gradient of match_kernel 3: 0.5 GB at B = 8) are held through a WEAK reference: a strong one kept up to
    four of them alive across steps and, by raising the use count of a gradient, forced autograd's accumulation of
    several passes' gradients out of place.  A dead weak reference means the tensor was summed into another one or
    freed: the entry is void and the consumer takes its own max|.| pass — never a stale value."""
    table = _tls.known_amax
    # (weak = True: the entry must not keep `t` alive whatever its size — a convolution's INPUT remembered for a sibling convolution
    #  (ADVICE r5): a strong reference kept up to four dead activations alive across steps)
    big = weak or t.numel() * t.element_size() >= _AMAX_WEAK_BYTES
    # two quotas: entries that HOLD their tensor (small ones) stay few — they pin memory; weak entries pin nothing and may be many: the
    # label map that a SPADEResnetBlock's three SPADEs read is found by the third one although a dozen producer entries (K21's weights,
    # K9's outputs) came in between (with one quota of 8 it had been evicted: 24 max|.| passes per module step)
    n_strong = sum(1 for e in table.values() if not isinstance(e[0], weakref.ref))
    if big:
        for k in [k for k, e in table.items() if isinstance(e[0], weakref.ref) and e[0]() is None]:
            del table[k]                                     # (dead tensors first)
        while len(table) - n_strong >= _KNOWN_AMAX_WEAK_MAX:
            table.pop(next(k for k, e in table.items() if isinstance(e[0], weakref.ref)), None)
    else:
        while n_strong >= _KNOWN_AMAX_MAX:
            table.pop(next(k for k, e in table.items() if not isinstance(e[0], weakref.ref)), None)
            n_strong -= 1
    table[(t.device, t.untyped_storage().data_ptr())] = (weakref.ref(t) if big else t, t._version, cell)


def _recall_amax(t: torch.Tensor, consume: bool = True):
    table = _tls.known_amax
    key = (t.device, t.untyped_storage().data_ptr())
    ent = table.pop(key, None) if consume else table.get(key)
    if ent is None:
        return None
    src, version, cell = ent
    if isinstance(src, weakref.ref):
        src = src()
        if src is None:
            table.pop(key, None)
            return None
    if (t.numel() != src.numel() or t.storage_offset() != src.storage_offset() or not t.is_contiguous()
            or t._version != version or src._version != version):
        return None
    return cell


def _zero_cell(device) -> torch.Tensor:
    """A 1-element fp32 tensor holding 0, from a per-thread pool zeroed 4096 cells at a time on the current stream
    (one fill per ~500 steps instead of a 5 us memset in front of each of the max|x| passes of a step)."""
    pool = _tls.zero_pool
    key = (device, _stream())
    ent = pool.get(key)
    if ent is None or ent[1] >= ent[0].numel():
        ent = pool[key] = [torch.zeros(4096, device=device, dtype=torch.float32), 0]
    cell = ent[0][ent[1]:ent[1] + 1]
    ent[1] += 1
    return cell


def _zero_cells(device, n: int) -> torch.Tensor:
    """n CONSECUTIVE cells of the pool above (a kernel that leaves several maxima takes one pointer)."""
    pool = _tls.zero_pool
    key = (device, _stream())
    ent = pool.get(key)
    if ent is None or ent[1] + n > ent[0].numel():
        ent = pool[key] = [torch.zeros(4096, device=device, dtype=torch.float32), 0]
    cells = ent[0][ent[1]:ent[1] + n]
    ent[1] += n
    return cells


def absmax(x: torch.Tensor) -> torch.Tensor:
    """max|x| as a 1-element CUDA tensor (one pass, no host sync) — the scale source of the f16 splits."""
    x = _chk(x, "absmax: x")
    out = _zero_cell(x.device)
    _call("absmax", "cocos_absmax_accumulate", x.data_ptr(), x.numel(), out.data_ptr(), _stream())
    return out


def absmax_many(xs):
    """[max|x| cell for x in xs] with ONE launch per four tensors (cocos_absmax4) — the max|.| passes a step takes in a row (the two
    feature tensors and the two projection weights in front of K23 / K0) used to be a 5-14 us launch each."""
    xs = [_chk(x, "absmax_many: x") for x in xs]
    cells = [_zero_cell(x.device) for x in xs]
    for i in range(0, len(xs), 4):
        args = []
        for x, c in zip(xs[i:i + 4], cells[i:i + 4]):
            args += [x.data_ptr(), x.numel(), c.data_ptr()]
        args += [None, 0, None] * (4 - len(xs[i:i + 4]))
        _call("absmax", "cocos_absmax4", *args, _stream())
    return cells


def prefetch_amax(xs):
    """max|x| of the tensors in `xs` that no producer has left a value for, in ONE launch, remembered for their consumers (the two
    feature tensors and the two weights of a pair of projections: match_kernel 3's K0 calls picked them up one 5-14 us pass each)."""
    todo = [x for x in xs if x is not None and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and _recall_amax(x, consume=False) is None]
    if todo:
        for x, c in zip(todo, absmax_many(todo)):
            _remember_amax(x, c)


def sum_leading(x: torch.Tensor) -> torch.Tensor:
    """x.sum(0) for contiguous fp32 partial tiles [S, ...] (cocos_sum_leading)."""
    x = _chk(x, "sum_leading: x")
    out = torch.empty(x.shape[1:], device=x.device, dtype=torch.float32)
    _call("sum_leading", "cocos_sum_leading", x.data_ptr(), out.data_ptr(), x.shape[0], out.numel(), _stream())
    return out


def channel_sum(dy: torch.Tensor) -> torch.Tensor:
    """dy.sum((0, 2, 3)) of a contiguous fp32 [B,C,...] tensor: the bias gradient of a convolution (cocos_channel_sum)."""
    dy = _chk(dy, "channel_sum: dy")
    B, C = dy.shape[:2]
    N = dy.numel() // (B * C)
    S = _lib.load().cocos_channel_sum_slices(C, N)
    db = torch.empty(C, device=dy.device, dtype=torch.float32)
    part = torch.empty(S * C, device=dy.device, dtype=torch.float32) if S > 1 else None
    _call("channel_sum", "cocos_channel_sum", dy.data_ptr(), db.data_ptr(), part.data_ptr() if S > 1 else None, B, C, N, _stream())
    return db


def _proj1x1_forward(x, weight, bias, need_t: bool):
    """y = conv1x1(x, weight, bias) on K0 -> (y, x, w2, state): `state` = what the backward needs (split / stream flags, the max|.|
    cells, the transposed weight planes when `need_t`).  Shared by _Proj1x1 and _ProjUnfoldStats."""
    x = _chk(x, "proj1x1: x")
    w2 = _chk(weight.reshape(weight.shape[0], -1), "proj1x1: weight")
    B, Cin, h, w = x.shape
    Cout = w2.shape[0]
    if w2.shape[1] != Cin:
        raise ValueError(f"proj1x1: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
    bb = None if bias is None else _chk(bias, "proj1x1: bias")
    y = torch.empty((B, Cout, h, w), device=x.device, dtype=torch.float32)
    split = PROJ_PRECISION == "f16x3" and min(Cin * Cout, Cin * h * w, Cout * h * w) >= 4
    # the reference's shapes (<= 416 input channels, grid a multiple of 64 positions): weight planes resident in
    # the accumulator file, x and y streamed once (proj_stream_f16x3.hip)
    lib = _lib.load()
    stream = (split and PROJ_STREAM and (h * w) % 64 == 0 and lib.cocos_proj1x1_stream_kpad(Cin) != 0
              and lib.cocos_proj1x1_stream_kpad(Cout) != 0)
    st = dict(split=split, stream=stream, amax=None, t_planes=None, wshape=tuple(weight.shape), has_bias=bias is not None)
    if split:      # products on the f16 MFMA, operands split on the fly (sgemm_f16x3.hip)
        wa = _recall_amax(w2)      # left by K21 when the layer is spectral-normed (W / sigma) or by prefetch_amax, else one small pass
        xa = _recall_amax(x)
        xa, wa = (absmax(x) if xa is None else xa), (absmax(w2) if wa is None else wa)
        if stream:
            # A = W as planes [Cout][Kpad], rows zero-padded to whole MFMA k-steps
            kp = lib.cocos_proj1x1_stream_kpad(Cin)
            wh = torch.empty((Cout, kp), device=x.device, dtype=torch.float16)
            wl = torch.empty((Cout, kp), device=x.device, dtype=torch.float16)
            ws = torch.empty(1, device=x.device, dtype=torch.float32)
            # ... and, when the input gradient will be wanted, the transposed planes [Cin][Kpad(Cout)] of dx = W^T dy
            # in the same launch (they used to be a second split in the backward)
            th = tl = None
            if need_t:
                kpo = lib.cocos_proj1x1_stream_kpad(Cout)
                th = torch.empty((Cin, kpo), device=x.device, dtype=torch.float16)
                tl = torch.empty((Cin, kpo), device=x.device, dtype=torch.float16)
            _call("split_f16", "cocos_proj_weight_planes", w2.data_ptr(), wh.data_ptr(), wl.data_ptr(), _ptr(th), _ptr(tl),
                  Cout, Cin, kp, lib.cocos_proj1x1_stream_kpad(Cout), wa.data_ptr(), ws.data_ptr(), _stream())
            st["t_planes"] = (th, tl, ws) if th is not None else None
            _call("proj1x1_fwd", "cocos_proj1x1_stream_f16x3", x.data_ptr(), wh.data_ptr(), wl.data_ptr(),
                  ws.data_ptr(), _ptr(bb), y.data_ptr(), B, Cin, Cout, h * w, xa.data_ptr(), _stream())
        else:
            _call("proj1x1_fwd", "cocos_proj1x1_fwd_f16x3", x.data_ptr(), w2.data_ptr(), _ptr(bb), y.data_ptr(),
                  B, Cin, Cout, h * w, xa.data_ptr(), wa.data_ptr(), _stream())
        st["amax"] = (xa, wa)
    else:
        _call("proj1x1_fwd", "cocos_proj1x1_fwd", x.data_ptr(), w2.data_ptr(), _ptr(bb), y.data_ptr(), B, Cin,
              Cout, h * w, _stream())
    return y, x, w2, st


class _Proj1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        y, x, w2, st = _proj1x1_forward(x, weight, bias, ctx.needs_input_grad[0])
        ctx.split, ctx.stream, ctx.amax, ctx.t_planes = st["split"], st["stream"], st["amax"], st["t_planes"]
        ctx.save_for_backward(x, w2)
        ctx.wshape, ctx.has_bias = st["wshape"], st["has_bias"]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        return _proj1x1_backward(x, w2, dy, ctx.amax if ctx.split else None, getattr(ctx, "t_planes", None), ctx.split, ctx.stream,
                                 need_x, need_w, need_b and ctx.has_bias, ctx.wshape)


def _proj1x1_backward(x, w2, dy, amax, t_planes, split, stream, need_x, need_w, need_b, wshape):
    """Backward of y = conv1x1(x, w2) given dy: (dx, dw, db) — shared by _Proj1x1 and the fused projection + normalisation
    (_ProjCenterL2NormPlanesPair).  `amax` = (max|x|, max|w|) device cells of the split flavour; `t_planes` = the transposed weight
    planes (hi, lo, scale) when the forward already made them."""
    dy = _chk(dy, "proj1x1: dy")
    B, Cin, h, w = x.shape
    Cout = w2.shape[0]
    N = h * w
    lib = _lib.load()
    dx = torch.empty_like(x) if need_x else None
    dw = db = None
    if split:
        xa, wa = amax
        ga = _recall_amax(dy)            # left by the kernel that wrote dy (K1's backward), else one pass
        if ga is None:
            ga = absmax(dy)
        dx_gemm, dw_gemm = need_x, need_w
        if stream and need_x:     # dx = W^T dy, same streaming kernel with the transposed weight planes
            if t_planes is not None:
                th, tl, ts = t_planes
            else:
                th, tl, ts = split_f16(w2.unsqueeze(0), transpose=True, cpad=lib.cocos_proj1x1_stream_kpad(Cout),
                                       amax=wa)
            _call("proj1x1_bwd", "cocos_proj1x1_stream_f16x3", dy.data_ptr(), th.data_ptr(), tl.data_ptr(),
                  ts.data_ptr(), None, dx.data_ptr(), B, Cout, Cin, N, ga.data_ptr(), _stream())
            dx_gemm = False
        parts = lib.cocos_proj1x1_dw_partials_f16x3(B, Cin, Cout, N) if (need_w and PROJ_STREAM) else 0
        if parts:                     # dw (and db) in one pass over dy and x (proj_dw_f16x3.hip)
            ws = torch.empty((parts, Cout, (Cin + 31) // 32 * 32), device=x.device, dtype=torch.float32)
            wsb = torch.empty((parts, Cout), device=x.device, dtype=torch.float32) if need_b else None
            dw = torch.empty((Cout, Cin), device=x.device, dtype=torch.float32)
            db = torch.empty(Cout, device=x.device, dtype=torch.float32) if need_b else None
            _call("proj1x1_bwd", "cocos_proj1x1_dw_f16x3", dy.data_ptr(), x.data_ptr(), ws.data_ptr(), _ptr(wsb),
                  dw.data_ptr(), _ptr(db), B, Cin, Cout, N, ga.data_ptr(), xa.data_ptr(), _stream())
            dw_gemm = False
        if dx_gemm or dw_gemm:
            dwb = None
            if dw_gemm:
                dwb = torch.empty((lib.cocos_proj1x1_bwd_partials_f16x3(B, Cin, Cout, N), Cout, Cin),
                                  device=x.device, dtype=torch.float32)
            _call("proj1x1_bwd", "cocos_proj1x1_bwd_f16x3", x.data_ptr(), w2.data_ptr(), dy.data_ptr(),
                  dx.data_ptr() if dx_gemm else None, _ptr(dwb), B, Cin, Cout, N, xa.data_ptr(), wa.data_ptr(),
                  ga.data_ptr(), _stream())
            if dw_gemm:
                dw = sum_leading(dwb)                                   # [P,256,Cl] partials, small
    else:
        dwb = None
        if need_w:
            dwb = torch.empty((lib.cocos_proj1x1_bwd_partials(B, Cin, Cout, N), Cout, Cin), device=x.device,
                              dtype=torch.float32)
        _call("proj1x1_bwd", "cocos_proj1x1_bwd", x.data_ptr(), w2.data_ptr(), dy.data_ptr(), _ptr(dx),
              _ptr(dwb), B, Cin, Cout, N, _stream())
        if need_w:
            dw = sum_leading(dwb)
    if need_w:
        dw = dw.reshape(wshape)
    if need_b and db is None:
        db = channel_sum(dy)
    return dx, dw, db


def proj1x1(x, weight, bias=None):
    """nn.Conv2d(Cin, Cout, kernel_size=1): x [B,Cin,h,w], weight [Cout,Cin,1,1].  PROJ_PRECISION "f16x3" (default):
    the streaming split-precision kernels at the reference's shapes, else the split GEMM; "fp32": the fp32-MFMA GEMM."""
    return _Proj1x1.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------
# K23  K0 fused with K1: projection -> centre -> L2-normalise -> operand planes   (correspondence.py:272-289; proj_norm_f16x3.hip)
# ------------------------------------------------------------------------------------------
#: False: the fused match_kernel-1 path projects with K0 and normalises with K1 in separate launches (the round-5 chain; A/B runs)
PROJ_NORM_FUSED = os.environ.get("COCOS_PROJ_NORM_FUSED", "1") != "0"
#: False: the projections' backward runs as round 5's chain (K1 backward / K12 backward + autograd's add -> max|.| -> input gradient ->
#: weight gradient); True: K24 (cocos_proj_bwd_input_f16x3) folds everything in front of the input gradient into it and the weight
#: gradient rebuilds its dy on the fly (A/B runs)
PROJ_BWD_FUSED = os.environ.get("COCOS_PROJ_BWD_FUSED", "1") != "0"
#: False: theta's and phi's weight gradients run as two launches of cocos_proj1x1_dw_affine_f16x3 (A/B runs); True: one pair launch
PROJ_DW_PAIR = os.environ.get("COCOS_PROJ_DW_PAIR", "1") != "0"


class LazyProj1x1:
    """`conv1x1(x, weight, bias)` of the theta / phi projections (correspondence.py:272,:282), not computed yet.  The hot path asks
    for what its back end needs: `raw()` — the fp32 projection through K0 (computed once, cached) — or, on the fused
    match_kernel-1 / PONO_C path, nothing at all: `proj_center_l2norm_planes_pair` then produces the correlation kernels' operand
    planes straight from `x` (K23) and the projection never exists in HBM.  Quacks like the tensor it stands for where
    `correspondence_hot_path` looks (shape, is_cuda, dtype, requires_grad, detach)."""

    def __init__(self, x, weight, bias=None):
        if x.dim() != 4 or weight.shape[1] != x.shape[1]:
            raise ValueError(f"LazyProj1x1: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
        self.x, self.weight, self.bias = x, weight, bias
        self._raw = None

    @property
    def shape(self):
        return torch.Size((self.x.shape[0], self.weight.shape[0], self.x.shape[2], self.x.shape[3]))

    is_cuda = property(lambda self: self.x.is_cuda)
    dtype = property(lambda self: self.x.dtype)
    device = property(lambda self: self.x.device)

    @property
    def requires_grad(self):
        return any(t is not None and t.requires_grad for t in (self.x, self.weight, self.bias))

    def detach(self):
        return LazyProj1x1(self.x.detach(), self.weight.detach(), None if self.bias is None else self.bias.detach())

    def raw(self):
        if self._raw is None:
            self._raw = proj1x1(self.x, self.weight, self.bias)
        return self._raw


def proj_norm_fused_ok(p: LazyProj1x1) -> bool:
    """Shapes K23 takes: 256 output channels, whole 128-position tiles, fp32 on the GPU, the split flavour."""
    B, Cout, h, w = p.shape
    return (PROJ_NORM_FUSED and PROJ_PRECISION == "f16x3" and PRECISION == "f16x3" and p.is_cuda and p.dtype == torch.float32
            and Cout == FUSED_K and (h * w) % 128 == 0 and p.x.shape[1] <= 4096 and p.x.shape[1] * h * w * 4 < 2 ** 31 - 1)


class _ProjCenterL2NormPlanesPair(torch.autograd.Function):
    """K23 for the two projections of a forward call (theta, phi): ONE launch -> operand planes (handed to the caller's
    OperandPlanes) + row norms.  Outputs are HANDLES as in _CenterL2NormPlanes: [B,256,N] tensors whose memory is never written
    or read (the split correlation kernels read the planes); their gradients (d qn / d kn from K2's backward) come back here and
    go through K1's backward (cocos_center_l2norm_bwd_planes) and K0's (_proj1x1_backward)."""

    @staticmethod
    def forward(ctx, x1, w1, b1, x2, w2, b2, center_over_channels: int, eps: float, planes, want_chan: bool):
        lib = _lib.load()
        xs = [_chk(x1, "proj_center_l2norm_planes: x (theta)"), _chk(x2, "proj_center_l2norm_planes: x (phi)")]
        ws = [_chk(w1.reshape(w1.shape[0], -1), "proj_center_l2norm_planes: weight"),
              _chk(w2.reshape(w2.shape[0], -1), "proj_center_l2norm_planes: weight")]
        bs = [None if b is None else _chk(b, "proj_center_l2norm_planes: bias") for b in (b1, b2)]
        B, Cin, h, w = xs[0].shape
        N = h * w
        if xs[1].shape != xs[0].shape or ws[0].shape != (FUSED_K, Cin) or ws[1].shape != (FUSED_K, Cin):
            raise ValueError("proj_center_l2norm_planes_pair: the two projections must have the same shapes "
                             f"(x {tuple(xs[0].shape)} / {tuple(xs[1].shape)}, weights {tuple(ws[0].shape)} / {tuple(ws[1].shape)})")
        dev = xs[0].device
        half = dict(device=dev, dtype=torch.float16)
        args, keep, tplanes, tfrags = [], [], [], []
        # max|.| of the two feature tensors and the two weights: the ones no producer left behind (K21 leaves a spectral-normed
        # weight's, K13 / K9 epilogues the features') share ONE launch
        amax = [_recall_amax(xs[0], consume=False), _recall_amax(ws[0]), _recall_amax(xs[1], consume=False), _recall_amax(ws[1])]
        missing = [i for i, c in enumerate(amax) if c is None]
        if missing:
            for i, c in zip(missing, absmax_many([(xs[0], ws[0], xs[1], ws[1])[i] for i in missing])):
                amax[i] = c
        wprep = []
        for pi, (x, w2d, bb) in enumerate(zip(xs, ws, bs)):
            xa, wa = amax[2 * pi], amax[2 * pi + 1]
            wfrag = torch.empty(lib.cocos_proj_weight_frag_bytes(Cin), device=dev, dtype=torch.uint8)
            wsc = torch.empty(1, device=dev, dtype=torch.float32)
            # ... and, when the input gradient will be wanted, the transposed planes [Cin][256] of dx = W^T dy in the same launch
            th = tl = None
            if ctx.needs_input_grad[3 * len(keep)]:
                th, tl = torch.empty((Cin, FUSED_K), **half), torch.empty((Cin, FUSED_K), **half)
            tplanes.append((th, tl, wsc) if th is not None else None)
            # ... and W^T in K24's fragment order when the fused backward will run
            wtf = None
            if th is not None and PROJ_BWD_FUSED and lib.cocos_proj_bwd_input_supported(Cin, FUSED_K, N):
                wtf = torch.empty(lib.cocos_proj_weight_tfrag_bytes(), device=dev, dtype=torch.uint8)
            tfrags.append(wtf)
            wprep += [w2d.data_ptr(), wa.data_ptr(), wfrag.data_ptr(), wsc.data_ptr(), _ptr(th), _ptr(tl), _ptr(wtf)]
            norm = torch.empty((B, N), device=dev, dtype=torch.float32)
            ph, pl = torch.empty((B, N, FUSED_K), **half), torch.empty((B, N, FUSED_K), **half)
            ch = cl = None
            if want_chan:
                ch, cl = torch.empty((B, FUSED_K, N), **half), torch.empty((B, FUSED_K, N), **half)
            args += [x.data_ptr(), wfrag.data_ptr(), wsc.data_ptr(), _ptr(bb), xa.data_ptr(), norm.data_ptr(), ph.data_ptr(),
                     pl.data_ptr(), _ptr(ch), _ptr(cl)]
            keep.append((xa, wa, wfrag, wsc, norm, ph, pl, ch, cl))
        # every weight layout of both projections in one launch (they were four)
        _call("split_f16", "cocos_proj_weight_prep_pair", 2, *wprep, FUSED_K, Cin, _stream())
        _call("proj_center_l2norm_fwd", "cocos_proj_center_l2norm_planes_f16x3", 2, *args, B, Cin, N, int(center_over_channels),
              float(eps), SPLIT_OPERAND_SCALE, _stream())
        handles = []
        for (_xa, _wa, _wf, _ws, norm, ph, pl, ch, cl) in keep:
            hd = torch.empty((B, FUSED_K, N), device=dev, dtype=torch.float32)      # never written, never read: an autograd handle
            planes.put(hd, True, SPLIT_OPERAND_SCALE, ph, pl)
            if want_chan:
                planes.put(hd, False, SPLIT_OPERAND_SCALE, ch, cl)
            handles.append(hd)
        ctx.save_for_backward(xs[0], ws[0], xs[1], ws[1], keep[0][4], keep[1][4])
        ctx.chan = [(k[7], k[8]) for k in keep]
        ctx.amax = [(k[0], k[1]) for k in keep]
        ctx.t_planes = tplanes
        ctx.tfrags = tfrags
        ctx.pos = [(k[5], k[6]) for k in keep]          # position-major planes: K24 reads y back from them
        ctx.wsc = [k[3] for k in keep]
        ctx.cfg = (int(center_over_channels), float(eps))
        ctx.wshapes = (tuple(w1.shape), tuple(w2.shape))
        ctx.has_bias = (b1 is not None, b2 is not None)
        return handles[0], handles[1]

    @staticmethod
    def backward(ctx, d1, d2):
        x1, w1, x2, w2, n1, n2 = ctx.saved_tensors
        mode, eps = ctx.cfg
        needs = ctx.needs_input_grad
        if (d1 is not None and d2 is not None and needs[0] and needs[3] and all(t is not None for t in ctx.tfrags)
                and (needs[1] or not needs[2]) and (needs[4] or not needs[5])):
            # K24: K1's backward + the input gradient of BOTH projections in one launch; the weight gradients rebuild d on the fly
            d = [_chk(d1, "proj_center_l2norm_planes: d qn"), _chk(d2, "proj_center_l2norm_planes: d kn")]
            xs, w2s, norms = (x1, x2), (w1, w2), (n1, n2)
            B, K, N = d[0].shape
            Cin = x1.shape[1]
            dev = d[0].device
            dxs = [torch.empty_like(x1), torch.empty_like(x2)]
            coefs = [torch.empty((B, 3, N), device=dev, dtype=torch.float32) for _ in range(2)]
            cells = [_zero_cell(dev), _zero_cell(dev)]
            args = []
            for i in range(2):
                ph, pl = ctx.pos[i]
                args += [d[i].data_ptr(), ph.data_ptr(), pl.data_ptr(), norms[i].data_ptr(), None, ctx.tfrags[i].data_ptr(),
                         ctx.wsc[i].data_ptr(), dxs[i].data_ptr(), coefs[i].data_ptr(), cells[i].data_ptr()]
            _call("proj_bwd_input", "cocos_proj_bwd_input_f16x3", 0, 2, *args, B, Cin, N, mode, eps, SPLIT_OPERAND_SCALE, _stream())
            out = []
            nb = [needs[3 * i + 2] and ctx.has_bias[i] for i in range(2)]
            if needs[1] and needs[4] and nb[0] == nb[1] and PROJ_DW_PAIR:
                # both weight gradients in ONE launch (+ one reduction): half the partial tiles, the chip filled by the pair
                res = _proj1x1_dw_affine_pair(2, [(d[i], ctx.chan[i][0], ctx.chan[i][1], coefs[i], xs[i], cells[i], ctx.amax[i][0])
                                                  for i in range(2)], SPLIT_OPERAND_SCALE, nb[0])
                for i in range(2):
                    out += [dxs[i], res[i][0].reshape(ctx.wshapes[i]), res[i][1]]
                return (*out, None, None, None, None)
            for i in range(2):
                need_w, need_b = needs[3 * i + 1], nb[i]
                dw = db = None
                if need_w:
                    ch, cl = ctx.chan[i]
                    dw, db = _proj1x1_dw_affine(2, d[i], ch, cl, coefs[i], SPLIT_OPERAND_SCALE, xs[i], cells[i], ctx.amax[i][0], need_b)
                    dw = dw.reshape(ctx.wshapes[i])
                out += [dxs[i], dw, db]
            return (*out, None, None, None, None)
        out = []
        for i, (x, w2d, norm, dy) in enumerate(((x1, w1, n1, d1), (x2, w2, n2, d2))):
            need_x, need_w, need_b = ctx.needs_input_grad[3 * i:3 * i + 3]
            if dy is None or not (need_x or need_w or need_b):
                out += [None, None, None]
                continue
            ch, cl = ctx.chan[i]
            if ch is None:
                raise _lib.CocosHipError("proj_center_l2norm_planes: backward without the channel-major planes (forward ran without grad)")
            dy = _chk(dy, "proj_center_l2norm_planes: d qn")
            B, K, N = dy.shape
            dth = torch.empty((B, K, x.shape[2], x.shape[3]), device=dy.device, dtype=torch.float32)
            cell = _zero_cell(dy.device)
            _call("center_l2norm_bwd", "cocos_center_l2norm_bwd_planes", ch.data_ptr(), cl.data_ptr(), norm.data_ptr(),
                  dy.data_ptr(), dth.data_ptr(), B, K, N, mode, eps, SPLIT_OPERAND_SCALE, cell.data_ptr(), _stream())
            _remember_amax(dth, cell)
            lib = _lib.load()
            Cin = x.shape[1]
            stream = PROJ_STREAM and N % 64 == 0 and lib.cocos_proj1x1_stream_kpad(Cin) != 0 and lib.cocos_proj1x1_stream_kpad(K) != 0
            out += list(_proj1x1_backward(x, w2d, dth, ctx.amax[i], ctx.t_planes[i], True, stream, need_x, need_w,
                                          need_b and ctx.has_bias[i], ctx.wshapes[i]))
        return (*out, None, None, None, None)


def _proj1x1_dw_affine(mode, in1, in2a, in2b, coef, plane_scale, x, d_amax, x_amax, need_b):
    """(dw [Cout,Cin], db | None) of a 1x1 projection whose output gradient is alpha in1 + beta in2 + gamma (coef [B,3,N] from K24),
    rebuilt inside the weight-gradient kernel (cocos_proj1x1_dw_affine_f16x3).  in1 [B,Cout,N] fp32; mode 1: in2a fp32 [B,Cout,N];
    mode 2: in2a / in2b channel-major f16 planes."""
    lib = _lib.load()
    B, Cin = x.shape[:2]
    Cout, N = in1.shape[1], in1.shape[2]
    parts = lib.cocos_proj1x1_dw_partials_f16x3(B, Cin, Cout, N)
    if not parts:
        raise _lib.CocosHipError(f"proj1x1_dw_affine: shape not supported (Cin={Cin} Cout={Cout} N={N})")
    ws = torch.empty((parts, Cout, (Cin + 31) // 32 * 32), device=x.device, dtype=torch.float32)
    wsb = torch.empty((parts, Cout), device=x.device, dtype=torch.float32) if need_b else None
    dw = torch.empty((Cout, Cin), device=x.device, dtype=torch.float32)
    db = torch.empty(Cout, device=x.device, dtype=torch.float32) if need_b else None
    _call("proj1x1_bwd", "cocos_proj1x1_dw_affine_f16x3", mode, in1.data_ptr(), in2a.data_ptr(), _ptr(in2b), coef.data_ptr(),
          float(plane_scale), x.data_ptr(), ws.data_ptr(), _ptr(wsb), dw.data_ptr(), _ptr(db), B, Cin, Cout, N, d_amax.data_ptr(),
          x_amax.data_ptr(), _stream())
    return dw, db


def _proj1x1_dw_affine_pair(mode, probs, plane_scale, need_b, scale_cells=(None, None)):
    """_proj1x1_dw_affine for two projections of one shape in one launch: probs = [(in1, in2a, in2b, coef, x, d_amax, x_amax)] * 2
    -> [(dw, db | None)] * 2 (cocos_proj1x1_dw_affine_pair_f16x3)."""
    lib = _lib.load()
    x0 = probs[0][4]
    B, Cin = x0.shape[:2]
    Cout, N = probs[0][0].shape[1], probs[0][0].shape[2]
    parts = lib.cocos_proj1x1_dw_partials_pair_f16x3(B, Cin, Cout, N)
    if not parts or probs[1][4].shape != x0.shape:
        if any(c is not None for c in scale_cells):
            raise _lib.CocosHipError(f"proj1x1_dw_affine_pair: shape not supported (Cin={Cin} Cout={Cout} N={N})")
        return [_proj1x1_dw_affine(mode, in1, a, b, coef, plane_scale, x, da, xa, need_b) for (in1, a, b, coef, x, da, xa) in probs]
    res, args, hold = [], [], []
    f32 = dict(device=x0.device, dtype=torch.float32)
    for (in1, in2a, in2b, coef, x, d_amax, x_amax) in probs:
        ws = torch.empty((parts, Cout, (Cin + 31) // 32 * 32), **f32)
        wsb = torch.empty((parts, Cout), **f32) if need_b else None
        dw = torch.empty((Cout, Cin), **f32)
        db = torch.empty(Cout, **f32) if need_b else None
        res.append((dw, db))
        # the FIRST problem's partial tiles must outlive the loop: rebinding `ws` / `wsb` for the second problem handed their blocks back
        # to the allocator before the launch, and the second problem's 1 KB `db` was carved out of the first one's bias partials — the
        # reduction then read a partial row that the other half of the same launch was overwriting (caught by the full GPU suite only:
        # it takes an allocator state in which that block is the best fit)
        hold.append((ws, wsb))
        args += [in1.data_ptr(), in2a.data_ptr(), _ptr(in2b), coef.data_ptr(), x.data_ptr(), ws.data_ptr(), _ptr(wsb), dw.data_ptr(),
                 _ptr(db), d_amax.data_ptr(), x_amax.data_ptr()]
    _call("proj1x1_bwd", "cocos_proj1x1_dw_affine_pair_f16x3", mode, float(plane_scale), *args, _ptr(scale_cells[0]), _ptr(scale_cells[1]),
          B, Cin, Cout, N, _stream())
    del hold          # (released after the launch: stream order makes the reuse safe)
    return res


class _ProjUnfoldStats(torch.autograd.Function):
    """match_kernel 3 (round 6): theta_raw = conv1x1(x, w, b) AND the statistics (mu, a) of its 3x3-unfolded vectors (K0 + K12) as ONE
    autograd node, so that its backward sees both gradients of theta_raw at once — the one through the statistics (K12's
    backward: g1 + 2 theta_raw g2 per position) and the one from the correlation GEMMs — and folds K12's apply pass, autograd's
    addition and the max|.| pass into the projection's input gradient (K24 mode 1); the weight gradient rebuilds d on the fly."""

    @staticmethod
    def forward(ctx, x, weight, bias, k_unfolded: float, eps: float):
        lib = _lib.load()
        ctx.set_materialize_grads(False)      # (nrm is not differentiable: autograd would hand its backward a zero-filled [B,N] tensor per call)
        y, x, w2, st = _proj1x1_forward(x, weight, bias, ctx.needs_input_grad[0])
        B, C, h, w = y.shape
        N = h * w
        mk = lambda: torch.empty((B, N), device=y.device, dtype=torch.float32)
        mu, a, nrm = mk(), mk(), mk()
        ws = torch.empty(2 * B * N, device=y.device, dtype=torch.float32)
        cell = _zero_cell(y.device)          # max|theta_raw| as a by-product: the correlation GEMM of the fused family splits it next
        _call("unfold3_stats_fwd", "cocos_unfold3_stats_fwd_amax", y.data_ptr(), mu.data_ptr(), a.data_ptr(), nrm.data_ptr(),
              ws.data_ptr(), B, C, h, w, float(k_unfolded), float(eps), cell.data_ptr(), _stream())
        _remember_amax(y, cell)
        ctx.wtf = None
        if (ctx.needs_input_grad[0] and st["split"] and PROJ_BWD_FUSED and C == FUSED_K
                and lib.cocos_proj_bwd_input_supported(x.shape[1], C, N)):
            ctx.wtf = torch.empty(lib.cocos_proj_weight_tfrag_bytes(), device=y.device, dtype=torch.uint8)
            ctx.wtf_scale = torch.empty(1, device=y.device, dtype=torch.float32)
            _call("split_f16", "cocos_proj_weight_tfrag_planes", w2.data_ptr(), st["amax"][1].data_ptr(), ctx.wtf.data_ptr(),
                  ctx.wtf_scale.data_ptr(), C, x.shape[1], _stream())
        ctx.st = st
        ctx.kc = float(k_unfolded)
        ctx.save_for_backward(x, w2, y, mu, a, nrm)
        ctx.mark_non_differentiable(nrm)
        return y, mu, a, nrm

    @staticmethod
    def backward(ctx, dy, dmu, da, _dnrm):
        x, w2, y, mu, a, nrm = ctx.saved_tensors
        st = ctx.st
        B, C, h, w = y.shape
        N = h * w
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        need_b = need_b and st["has_bias"]
        dmu = None if dmu is None else _chk(dmu, "proj_unfold3_stats: dmu")
        da = None if da is None else _chk(da, "proj_unfold3_stats: da")
        if dy is not None and ctx.wtf is not None and need_x and (need_w or not need_b):
            dy = _chk(dy, "proj_unfold3_stats: d theta")
            maps = torch.empty(2 * B * N, device=y.device, dtype=torch.float32)
            _call("unfold3_stats_bwd", "cocos_unfold3_stats_bwd_maps", mu.data_ptr(), a.data_ptr(), nrm.data_ptr(), _ptr(dmu), _ptr(da),
                  maps.data_ptr(), B, h, w, ctx.kc, _stream())
            g1, g2 = maps[:B * N], maps[B * N:]
            dx = torch.empty_like(x)
            coef = torch.empty((B, 3, N), device=y.device, dtype=torch.float32)
            cell = _zero_cell(y.device)
            _call("proj_bwd_input", "cocos_proj_bwd_input_f16x3", 1, 1, dy.data_ptr(), y.data_ptr(), None, g1.data_ptr(), g2.data_ptr(),
                  ctx.wtf.data_ptr(), ctx.wtf_scale.data_ptr(), dx.data_ptr(), coef.data_ptr(), cell.data_ptr(),
                  None, None, None, None, None, None, None, None, None, None, B, x.shape[1], N, 1, 0.0, 1.0, _stream())
            dw = db = None
            if need_w:
                dw, db = _proj1x1_dw_affine(1, dy.reshape(B, C, N), y, None, coef, 1.0, x, cell, st["amax"][0], need_b)
                dw = dw.reshape(st["wshape"])
            return dx, dw, db, None, None
        # round 5's chain: K12's backward as a tensor, autograd-style addition, then the projection's backward
        dth = torch.empty_like(y)
        ws = torch.empty(2 * B * N, device=y.device, dtype=torch.float32)
        _call("unfold3_stats_bwd", "cocos_unfold3_stats_bwd", y.data_ptr(), mu.data_ptr(), a.data_ptr(), nrm.data_ptr(),
              _ptr(dmu), _ptr(da), dth.data_ptr(), ws.data_ptr(), B, C, h, w, ctx.kc, _stream())
        if dy is not None:
            dth = dth + dy
        return (*_proj1x1_backward(x, w2, dth, st["amax"], st["t_planes"], st["split"], st["stream"], need_x, need_w, need_b,
                                   st["wshape"]), None, None)


#: False: match_kernel 3 projects with K0 and takes the statistics / operand planes in separate launches (A/B runs)
PROJ_RAW_FUSED = os.environ.get("COCOS_PROJ_RAW_FUSED", "1") != "0"


class Box3RawPlanes:
    """What K25 leaves for the match_kernel-3 family about theta and phi, keyed by the autograd HANDLE that stands for each raw
    projection: (position-major hi, lo, channel-major hi | None, lo | None, scale cell).  Owned by hot_path's _BoxedCorr: one
    forward call, one thread."""

    def __init__(self):
        self._ent = {}

    def put(self, handle, ph, pl, ch, cl, scale):
        self._ent[id(handle)] = (handle, handle._version, ph, pl, ch, cl, scale)

    def get(self, handle):
        ent = self._ent.get(id(handle))
        if ent is None or ent[0] is not handle:
            return None
        if ent[1] != handle._version:
            raise _lib.CocosHipError("Box3RawPlanes: the tensor behind producer-made operand planes was modified in place")
        return ent[2:]


def proj_raw_fused_ok(theta: LazyProj1x1, phi: LazyProj1x1) -> bool:
    """Shapes / gradient patterns K25 and its backward take: both projections 256 channels from <= 448 of one shape, whole
    128-position tiles, and either nothing or everything (features, weights) differentiated."""
    if not (PROJ_RAW_FUSED and PROJ_BWD_FUSED and PROJ_PRECISION == "f16x3" and PRECISION == "f16x3"):
        return False
    B, Cout, h, w = theta.shape
    N = h * w
    if not (theta.is_cuda and theta.dtype == torch.float32 and theta.x.shape == phi.x.shape and phi.shape == theta.shape
            and Cout == FUSED_K and N % 128 == 0 and _lib.load().cocos_proj_bwd_input_supported(theta.x.shape[1], FUSED_K, N)
            and theta.x.shape[1] * N * 4 < 2 ** 31 - 1 and phi.is_cuda and phi.dtype == torch.float32):
        return False
    if not torch.is_grad_enabled():
        return True
    req = [t.requires_grad for t in (theta.x, theta.weight, phi.x, phi.weight)]
    return all(req) or not (any(req) or any(b is not None and b.requires_grad for b in (theta.bias, phi.bias)))


class _ProjRawPlanesStatsPair(torch.autograd.Function):
    """K25 for the two projections of a match_kernel-3 forward call: ONE launch from the features to the operand planes of the raw
    theta / phi (position-major for the x-box GEMM, channel-major for the backward GEMMs and the weight gradient) and to the
    per-position sums K12's statistics are box sums of; one more launch finishes (mu, a) / (nu, b).  Outputs: two HANDLES
    [B,256,h,w] (never written, never read: _Box3CorrXbox takes the planes from `holder`) and the statistics.  Backward: the
    handles' gradients (the correlation GEMMs') and the statistics' (K19's) meet here — K12's maps for both tensors (one launch),
    K24 mode C for both input gradients (one launch), both weight gradients (one launch + one reduction)."""

    @staticmethod
    def forward(ctx, x1, w1, b1, x2, w2, b2, k_unfolded: float, eps: float, holder):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        xs = [_chk(x1, "proj_raw_planes_stats: x (theta)"), _chk(x2, "proj_raw_planes_stats: x (phi)")]
        ws = [_chk(w1.reshape(w1.shape[0], -1), "proj_raw_planes_stats: weight"), _chk(w2.reshape(w2.shape[0], -1), "proj_raw_planes_stats: weight")]
        bs = [None if b is None else _chk(b, "proj_raw_planes_stats: bias") for b in (b1, b2)]
        B, Cin, h, w = xs[0].shape
        N = h * w
        dev = xs[0].device
        half, f32 = dict(device=dev, dtype=torch.float16), dict(device=dev, dtype=torch.float32)
        want_grad = any(ctx.needs_input_grad[:6])
        amax = [_recall_amax(xs[0], consume=False), _recall_amax(ws[0]), _recall_amax(xs[1], consume=False), _recall_amax(ws[1])]
        missing = [i for i, c in enumerate(amax) if c is None]
        if missing:
            for i, c in zip(missing, absmax_many([(xs[0], ws[0], xs[1], ws[1])[i] for i in missing])):
                amax[i] = c
        wprep, args, keep = [], [], []
        for pi in range(2):
            wfrag = torch.empty(lib.cocos_proj_weight_frag_bytes(Cin), device=dev, dtype=torch.uint8)
            wsc = torch.empty(1, **f32)
            wtf = torch.empty(lib.cocos_proj_weight_tfrag_bytes(), device=dev, dtype=torch.uint8) if want_grad else None
            wprep += [ws[pi].data_ptr(), amax[2 * pi + 1].data_ptr(), wfrag.data_ptr(), wsc.data_ptr(), None, None, _ptr(wtf)]
            sums = torch.empty((2, B, N), **f32)
            ysc = torch.empty(1, **f32)
            ph, pl = torch.empty((B, N, FUSED_K), **half), torch.empty((B, N, FUSED_K), **half)
            ch = cl = None
            if want_grad:
                ch, cl = torch.empty((B, FUSED_K, N), **half), torch.empty((B, FUSED_K, N), **half)
            args += [xs[pi].data_ptr(), wfrag.data_ptr(), wsc.data_ptr(), _ptr(bs[pi]), amax[2 * pi].data_ptr(), sums[0].data_ptr(),
                     sums[1].data_ptr(), ysc.data_ptr(), ph.data_ptr(), pl.data_ptr(), _ptr(ch), _ptr(cl)]
            keep.append((wfrag, wsc, wtf, sums, ysc, ph, pl, ch, cl))
        _call("split_f16", "cocos_proj_weight_prep_pair", 2, *wprep, FUSED_K, Cin, _stream())
        _call("proj1x1_fwd", "cocos_proj_raw_planes_stats_f16x3", 2, *args, B, Cin, N, _stream())
        stats = [tuple(torch.empty((B, N), **f32) for _ in range(3)) for _ in range(2)]      # (mu, a, nrm) per tensor
        fa = []
        for pi in range(2):
            fa += [keep[pi][3][0].data_ptr(), keep[pi][3][1].data_ptr()] + [t.data_ptr() for t in stats[pi]]
        _call("unfold3_stats_fwd", "cocos_unfold3_stats_finish_pair", *fa, B, h, w, float(k_unfolded), float(eps), _stream())
        handles = []
        for pi in range(2):
            hd = torch.empty((B, FUSED_K, h, w), **f32)      # never written, never read: an autograd handle
            (_wf, _ws, _wt, _s, ysc, ph, pl, ch, cl) = keep[pi]
            holder.put(hd, ph, pl, ch, cl, ysc)
            handles.append(hd)
        ctx.save_for_backward(xs[0], ws[0], xs[1], ws[1], *stats[0], *stats[1])
        ctx.keep = keep
        ctx.x_amax = (amax[0], amax[2])
        ctx.kc = float(k_unfolded)
        ctx.dims = (B, Cin, h, w)
        ctx.wshapes = (tuple(w1.shape), tuple(w2.shape))
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.mark_non_differentiable(stats[0][2], stats[1][2])
        return handles[0], stats[0][0], stats[0][1], handles[1], stats[1][0], stats[1][1], stats[0][2], stats[1][2]

    @staticmethod
    def backward(ctx, d1, dmu1, da1, d2, dmu2, da2, _n1, _n2):
        x1, w1, x2, w2, mu1, a1, nrm1, mu2, a2, nrm2 = ctx.saved_tensors
        B, Cin, h, w = ctx.dims
        N = h * w
        dev = x1.device
        f32 = dict(device=dev, dtype=torch.float32)
        needs = ctx.needs_input_grad
        if not (needs[0] and needs[1] and needs[3] and needs[4]):
            raise _lib.CocosHipError("proj_raw_planes_stats: backward needs the gradients of both feature tensors and both weights "
                                     "(proj_raw_fused_ok gates the forward)")
        ds = []
        for d in (d1, d2):      # (a projection whose correlation gradient is absent: only the statistics' path contributes)
            ds.append(torch.zeros((B, FUSED_K, N), **f32) if d is None else _chk(d, "proj_raw_planes_stats: d theta").reshape(B, FUSED_K, N))
        stats = ((mu1, a1, nrm1, dmu1, da1), (mu2, a2, nrm2, dmu2, da2))
        maps = [torch.empty(2 * B * N, **f32) for _ in range(2)]
        ma = []
        for pi in range(2):
            mu, a, nrm, dmu, da = stats[pi]
            ma += [mu.data_ptr(), a.data_ptr(), nrm.data_ptr(), _ptr(None if dmu is None else _chk(dmu, "dmu")),
                   _ptr(None if da is None else _chk(da, "da")), maps[pi].data_ptr()]
        _call("unfold3_stats_bwd", "cocos_unfold3_stats_bwd_maps_pair", *ma, B, h, w, ctx.kc, _stream())
        xs, dxs = (x1, x2), [torch.empty_like(x1), torch.empty_like(x2)]
        coefs = [torch.empty((B, 3, N), **f32) for _ in range(2)]
        cells = [_zero_cell(dev), _zero_cell(dev)]
        args = []
        for pi in range(2):
            (_wf, wsc, wtf, _s, ysc, ph, pl, _ch, _cl) = ctx.keep[pi]
            g1, g2 = maps[pi][:B * N], maps[pi][B * N:]
            args += [ds[pi].data_ptr(), ph.data_ptr(), pl.data_ptr(), ysc.data_ptr(), g1.data_ptr(), g2.data_ptr(), wtf.data_ptr(),
                     wsc.data_ptr(), dxs[pi].data_ptr(), coefs[pi].data_ptr(), cells[pi].data_ptr()]
        _call("proj_bwd_input", "cocos_proj_bwd_input_planes_f16x3", 2, *args, B, Cin, N, _stream())
        nb = [needs[2] and ctx.has_bias[0], needs[5] and ctx.has_bias[1]]
        probs = [(ds[pi], ctx.keep[pi][7], ctx.keep[pi][8], coefs[pi], xs[pi], cells[pi], ctx.x_amax[pi]) for pi in range(2)]
        scale_cells = (ctx.keep[0][4], ctx.keep[1][4])
        if nb[0] == nb[1]:
            res = _proj1x1_dw_affine_pair(2, probs, 1.0, nb[0], scale_cells)
        else:       # (a bias on one projection only: two pair-API launches of one problem each would need another entry point — both
            #  biases' gradients are computed and the unwanted one dropped)
            res = _proj1x1_dw_affine_pair(2, probs, 1.0, True, scale_cells)
            res = [(res[pi][0], res[pi][1] if nb[pi] else None) for pi in range(2)]
        return (dxs[0], res[0][0].reshape(ctx.wshapes[0]), res[0][1], dxs[1], res[1][0].reshape(ctx.wshapes[1]), res[1][1], None, None, None)


def proj_raw_planes_stats_pair(theta: LazyProj1x1, phi: LazyProj1x1, k_unfolded: float, holder: Box3RawPlanes, eps: float = NORM_EPS):
    """((theta handle, mu, a), (phi handle, nu, b)) of two lazy projections through K25: see _ProjRawPlanesStatsPair.  The handles go
    to box3_corr_xbox(..., raw_planes=holder).  Only for proj_raw_fused_ok() pairs."""
    th, mu, a, ph, nu, b, _, _ = _ProjRawPlanesStatsPair.apply(theta.x, theta.weight, theta.bias, phi.x, phi.weight, phi.bias,
                                                               float(k_unfolded), eps, holder)
    return (th, mu, a), (ph, nu, b)


def proj_unfold3_stats(p: LazyProj1x1, k_unfolded: float, eps: float = NORM_EPS):
    """(theta_raw [B,C,h,w], mu, a [B,h*w]) of a lazy projection: K0 + K12 as one autograd node (see _ProjUnfoldStats)."""
    y, mu, a, _ = _ProjUnfoldStats.apply(p.x, p.weight, p.bias, float(k_unfolded), eps)
    return y, mu, a


def proj_center_l2norm_planes_pair(theta: LazyProj1x1, phi: LazyProj1x1, center_over_channels, planes: OperandPlanes,
                                   eps: float = NORM_EPS, want_chan: bool = True):
    """(qn handle, kn handle) of the two lazy projections through K23: see _ProjCenterL2NormPlanesPair.  Pass the handles to
    corr_softmax_warp(..., planes=planes).  Only for proj_norm_fused_ok() shapes on the split path (corr_split_ok)."""
    return _ProjCenterL2NormPlanesPair.apply(theta.x, theta.weight, theta.bias, phi.x, phi.weight, phi.bias,
                                             int(center_over_channels), eps, planes, bool(want_chan))


# ------------------------------------------------------------------------------------------
# K16  nn.Conv2d (groups 1, dilation 1, zero padding) as a split-precision implicit GEMM   (conv_f16x3.hip)
#      ResidualBlock (correspondence.py:13-36), adaptor convolutions (:150-173), PatchGAN (discriminator.py:92-115)
# ------------------------------------------------------------------------------------------
#: K16's arithmetic: "f16x3" = f16 hi/lo split, three MFMA terms (fp32-class accuracy: everything upstream of the correlation);
#: "bf16" = single bf16 planes, one MFMA term (BASELINE config 3's precision for the generator / discriminator stacks behind
#: InstanceNorm / SPADE — ~3x less matrix work, no max|x| passes).  Module attribute, read at call time.
#: "torch" (A/B runs): producers.Conv2d / ReflectionPad2d / the spectral-norm hook keep the framework's kernels; a DIRECT
#: ops.conv2d call is an error in that setting.  This is the only copy of the switch (producers.conv_backend() reads it).
CONV_PRECISION = os.environ.get("COCOS_CONV", "f16x3")


def _conv_bf16() -> bool:
    if CONV_PRECISION == "torch":
        raise ValueError("cocosnet_amd.ops.conv2d called with CONV_PRECISION = 'torch': that setting routes the producers' "
                         "convolutions to the framework (producers.Conv2d); the HIP entry point needs 'f16x3' or 'bf16'")
    if CONV_PRECISION not in ("f16x3", "bf16"):
        raise ValueError(f"cocosnet_amd.ops.CONV_PRECISION = {CONV_PRECISION!r}: expected 'f16x3', 'bf16' or 'torch'")
    return CONV_PRECISION == "bf16"


def _conv_weight_planes(weight, amax, mode: int, JH=None, JW=None, ry=0, rx=0, s=1):
    """weight [Cout,Cin,KH,KW] -> K16's f16 hi/lo planes in ONE launch (cocos_conv2d_weight_planes): mode 0 = forward,
    mode 1 = input gradient (flipped (sub-)kernel w[:, :, ry::s, rx::s], channel roles swapped)."""
    Cout, Cin, KH, KW = weight.shape
    JH, JW = (KH if JH is None else JH), (KW if JW is None else JW)
    M, C = (Cout, Cin) if mode == 0 else (Cin, Cout)
    nkb = JH * JW * ((C + 31) // 32)
    wh = torch.empty((nkb, M, 32), device=weight.device, dtype=torch.float16)
    if amax is None:          # one bf16 plane (the bits live in an f16-typed tensor), no scale
        _call("split_f16", "cocos_conv2d_weight_planes", weight.data_ptr(), wh.data_ptr(), None, Cout, Cin, KH, KW, mode | 2,
              JH, JW, ry, rx, s, None, None, _stream())
        return wh, None, None
    wl = torch.empty_like(wh)
    ws = torch.empty(1, device=weight.device, dtype=torch.float32)
    _call("split_f16", "cocos_conv2d_weight_planes", weight.data_ptr(), wh.data_ptr(), wl.data_ptr(), Cout, Cin, KH, KW, mode,
          JH, JW, ry, rx, s, amax.data_ptr(), ws.data_ptr(), _stream())
    return wh, wl, ws


def _conv_fwd_call(x, wh, wl, ws, xa, bias, Cout, KH, KW, stride, pad, dil):
    B, Cin, H, W = x.shape
    lib = _lib.load()
    OH, OW = lib.cocos_conv2d_out_size(H, KH, stride, pad, dil), lib.cocos_conv2d_out_size(W, KW, stride, pad, dil)
    if OH < 1 or OW < 1:
        raise ValueError(f"conv2d: kernel {KH}x{KW} (dilation {dil}) does not fit input {tuple(x.shape)} with padding {pad}")
    y = torch.empty((B, Cout, OH, OW), device=x.device, dtype=torch.float32)
    _call("conv2d_fwd", "cocos_conv2d_fwd_f16x3", x.data_ptr(), wh.data_ptr(), _ptr(wl), _ptr(ws),
          _ptr(xa), _ptr(bias), y.data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, pad, dil, _stream())
    return y


#: "0": COCOS_CONV=bf16 keeps every layer on conv_f16x3.hip's one-term kernels (fp32 NCHW operands gathered per tap) — A/B runs
CONV_NHWC = True


def _conv_nhwc_ok(Cin, Cout, KH, KW, stride):
    return CONV_NHWC and bool(_lib.load().cocos_conv2d_nhwc_bf16_supported(Cin, Cout, KH, KW, stride))


#: "0": the fp32-accurate flavour keeps every layer on conv_f16x3.hip's gather kernels (A/B runs); default: K16c, the same
#: arithmetic on K16b's data path (f16 hi/lo planes NHWC in memory, LDS-DMA GEMMs)
CONV_NHWC_F16X3 = True


def conv_nhwc_prep_split(x: torch.Tensor, pad: int, reflect: bool, amax: torch.Tensor) -> torch.Tensor:
    """fp32 [B,C,H,W] -> f16 [2,B,H+2p,W+2p,Cp]: hi and lo plane of x * 2^k (k from the max|x| cell `amax`), NHWC, border zero or
    mirrored: the operand of K16c (cocos_conv2d_nhwc_prep_f16x3)."""
    x = _chk(x, "conv_nhwc_prep_split: x")
    B, C, H, W = x.shape
    xp = torch.empty((2, B, H + 2 * pad, W + 2 * pad, (C + 31) // 32 * 32), device=x.device, dtype=torch.float16)
    _call("conv2d_nhwc_prep", "cocos_conv2d_nhwc_prep_f16x3", x.data_ptr(), xp.data_ptr(), _ptr(amax), B, C, H, W, int(pad),
          int(bool(reflect)), _stream())
    return xp


def _nhwc_out(xp_shape, Cout, KH, KW, dil, stride, fold, device):
    """Output (and, in fold mode, the border ring) of a K16b / K16c GEMM: fold = the layer sits behind nn.ReflectionPad2d(1) and this
    is its input gradient — y is the UNPADDED dx, the mirrored border comes back through the ring."""
    B, Hp, Wp = xp_shape[-4], xp_shape[-3], xp_shape[-2]
    if Hp <= dil * (KH - 1) or Wp <= dil * (KW - 1):
        raise ValueError(f"conv2d: kernel {KH}x{KW} (dilation {dil}) does not fit the padded input {tuple(xp_shape)}")
    OH, OW = (Hp - dil * (KH - 1) - 1) // stride + 1, (Wp - dil * (KW - 1) - 1) // stride + 1
    if not fold:
        return torch.empty((B, Cout, OH, OW), device=device, dtype=torch.float32), None
    return (torch.empty((B, Cout, OH - 2, OW - 2), device=device, dtype=torch.float32),
            torch.empty((B, Cout, 2 * OW + 2 * (OH - 2)), device=device, dtype=torch.float32))


def _conv_nhwc_split_call(xp, wh, wl, ws, xa, bias, Cout, KH, KW, dil, stride=1, fold=False):
    _, B, Hp, Wp, Cp = xp.shape
    y, ring = _nhwc_out(xp.shape, Cout, KH, KW, dil, stride, fold, xp.device)
    wsp = _conv_nhwc_workspace(xp.device, split=True)
    _call("conv2d_fwd", "cocos_conv2d_nhwc_f16x3", xp.data_ptr(), wh.data_ptr(), wl.data_ptr(), ws.data_ptr(), _ptr(xa), _ptr(bias),
          y.data_ptr(), _ptr(ring), _ptr(wsp), wsp.numel() * 4 if wsp is not None else 0, B, Cp, Hp, Wp, Cout, KH, KW, dil, int(stride),
          _stream())
    return y


def conv_nhwc_prep(x: torch.Tensor, pad: int, reflect: bool = False) -> torch.Tensor:
    """fp32 [B,C,H,W] -> bf16 [B,H+2p,W+2p,Cp] (channels padded to 32, border zero or mirrored): the operand layout of K16b
    (cocos_conv2d_nhwc_prep_bf16)."""
    x = _chk(x, "conv_nhwc_prep: x")
    B, C, H, W = x.shape
    xp = torch.empty((B, H + 2 * pad, W + 2 * pad, (C + 31) // 32 * 32), device=x.device, dtype=torch.bfloat16)
    _call("conv2d_nhwc_prep", "cocos_conv2d_nhwc_prep_bf16", x.data_ptr(), xp.data_ptr(), B, C, H, W, int(pad), int(bool(reflect)),
          _stream())
    return xp


#: "1": stream-K launches for K16b layers whose tile count is just above a whole round of the CUs.  Off by default: once the
#: DMA pieces were interleaved with the MFMAs a tile got 20 % faster and the parked partials (nearly every tile is cut: 128 MB)
#: cost what the second round costs (407 -> 407 input gradient 0.191 vs 0.192 ms, 512 -> 512 0.224 vs 0.239).
CONV_NHWC_STREAMK = False
#: the three-term flavour (K16c) is the other way round: a tile takes 3x as long, the parked partials cost the same — stream-K on
#: (407 -> 407 input gradient on 66 x 66: two rounds of 0.3 ms against 1.07 rounds + 0.04 ms)
CONV_NHWC_STREAMK_SPLIT = True


def _conv_nhwc_workspace(device, split=False):
    """K16b / K16c's stream-K scratch: one zero-initialised buffer per (thread, device, stream), cleared when it is allocated (the
    kernel leaves its flags zero)."""
    if not (CONV_NHWC_STREAMK_SPLIT if split else CONV_NHWC_STREAMK):
        return None
    pool = getattr(_tls, "nhwc_ws", None)
    if pool is None:
        pool = _tls.nhwc_ws = {}
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = pool.get(key)
    if ws is None:
        ws = pool[key] = torch.zeros(_lib.load().cocos_conv2d_nhwc_bf16_workspace_bytes() // 4, device=device, dtype=torch.int32)
    return ws


def _conv_nhwc_call(xp, planes, bias, Cout, KH, KW, dil, stride=1, fold=False):
    B, Hp, Wp, Cp = xp.shape
    y, ring = _nhwc_out(xp.shape, Cout, KH, KW, dil, stride, fold, xp.device)
    ws = _conv_nhwc_workspace(xp.device)
    _call("conv2d_fwd", "cocos_conv2d_nhwc_bf16", xp.data_ptr(), planes.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(ring), _ptr(ws),
          ws.numel() * 4 if ws is not None else 0, B, Cp, Hp, Wp, Cout, KH, KW, dil, int(stride), _stream())
    return y


def _conv_dgrad_strided(xshape, weight, dy, ga, wa, s: int, p: int):
    """Input gradient of a stride-s convolution (dilation 1) as s*s stride-1 convolutions of dy.
    dx[y, x] only sees the taps with ky = (y + p) mod s (mod s): with y + p = s*u + ry and ky = ry + s*jy,
        dx[s*u + ry - p, s*v + rx - p] = sum_{co, jy, jx} w[co, ci, ry + s*jy, rx + s*jx] * dy[co, u - jy, v - jx],
    a correlation of dy with the flipped sub-kernel, evaluated for the u, v whose pixel lies inside the input and written
    straight into dx with stride s (cocos_conv2d_fwd_scatter_f16x3).  Returns None when a class needs negative padding."""
    B, Cin, H, W = xshape
    Cout, _, KH, KW = weight.shape
    OH, OW = dy.shape[2:]
    classes = []
    for ry in range(s):
        JH = len(range(ry, KH, s))
        u0 = max(0, -((ry - p) // s))                      # ceil((p - ry) / s), first u with s*u + ry - p >= 0
        U = (H - 1 + p - ry) // s + 1 - u0 if H - 1 + p - ry >= 0 else 0
        for rx in range(s):
            JW = len(range(rx, KW, s))
            v0 = max(0, -((rx - p) // s))
            V = (W - 1 + p - rx) // s + 1 - v0 if W - 1 + p - rx >= 0 else 0
            if U <= 0 or V <= 0:
                continue
            if JH == 0 or JW == 0:
                classes.append(None)                         # pixels no tap reaches: zeros
                continue
            py, px = JH - 1 - u0, JW - 1 - v0
            if py < 0 or px < 0:
                return None
            classes.append((ry, rx, JH, JW, u0, v0, U, V, py, px))
    dx = (torch.zeros if any(c is None for c in classes) else torch.empty)(xshape, device=dy.device, dtype=torch.float32)
    for c in classes:
        if c is None:
            continue
        ry, rx, JH, JW, u0, v0, U, V, py, px = c
        th, tl, ts = _conv_weight_planes(weight, wa, 1, JH, JW, ry, rx, s)
        off = (s * u0 + ry - p) * W + (s * v0 + rx - p)
        _call("conv2d_fwd", "cocos_conv2d_fwd_scatter_f16x3", dy.data_ptr(), th.data_ptr(), _ptr(tl), _ptr(ts),
              _ptr(ga), dx.data_ptr(), B, Cout, OH, OW, Cin, JH, JW, py, px, U, V, H * W, s * W, s, off, _stream())
    return dx


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride: int, pad: int, dil: int, reflect: int = 0):
        x = _chk(x, "conv2d: x")
        weight = _chk(weight, "conv2d: weight")
        if x.dim() != 4 or weight.dim() != 4 or weight.shape[1] != x.shape[1]:
            raise ValueError(f"conv2d: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
        if stride < 1 or pad < 0 or dil < 1:
            raise ValueError(f"conv2d: stride {stride} / padding {pad} / dilation {dil}")
        Cout, Cin, KH, KW = weight.shape
        bb = None if bias is None else _chk(bias, "conv2d: bias")
        if _conv_bf16():          # one-term flavour: no max|x| passes at all
            xa = wa = None
        else:
            xa = _recall_amax(x, consume=False)      # (not consumed: SPADE's gamma and beta convolutions read the SAME activation)
            if xa is None:
                xa = absmax(x)
                _remember_amax(x, xa, weak=True)     # ... and the second one finds what the first one measured
            wa = _recall_amax(weight)          # K21 leaves it for spectral-normed layers
            if wa is None:
                wa = absmax(weight)
        wh, wl, ws = _conv_weight_planes(weight, wa, 0)
        xp = None
        if wa is not None and CONV_NHWC_F16X3 and _conv_nhwc_ok(Cin, Cout, KH, KW, stride):
            # K16c: the split arithmetic on K16b's data path — hi/lo planes NHWC in memory, LDS-DMA GEMM, three terms
            xp = conv_nhwc_prep_split(x, reflect if reflect else pad, bool(reflect), xa)
            y = _conv_nhwc_split_call(xp, wh, wl, ws, xa, bb, Cout, KH, KW, dil, stride)
            if y.shape[3] % 32 != 0:
                xp = None
        elif wa is None and _conv_nhwc_ok(Cin, Cout, KH, KW, stride):
            # K16b: operands bf16 in memory (NHWC, border included), LDS-DMA GEMM — conv_nhwc_bf16.hip
            xp = conv_nhwc_prep(x, reflect if reflect else pad, bool(reflect))    # the border: zeros, or the layer's ReflectionPad2d
            y = _conv_nhwc_call(xp, wh, bb, Cout, KH, KW, dil, stride)
            if y.shape[3] % 32 != 0:
                xp = None               # its weight gradient takes whole rows of 32 positions: this layer's stays on K16
        else:
            y = _conv_fwd_call(x, wh, wl, ws, xa, bb, Cout, KH, KW, stride, pad, dil)
        if reflect and xp is None:
            raise _lib.CocosHipError("conv2d: reflect padding is only fused on the K16b path (ops.conv2d checks the shape first)")
        keep_xp = xp is not None and (ctx.needs_input_grad[1] or reflect)
        ctx.save_for_backward(weight, *((xp,) if keep_xp else (x,)))
        ctx.x_shape, ctx.has_xp, ctx.reflect = tuple(x.shape), keep_xp, int(reflect)
        ctx.cfg = (int(stride), int(pad), int(dil), bias is not None)
        ctx.amax = (xa, wa)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, saved = ctx.saved_tensors
        x, xp = (None, saved) if ctx.has_xp else (saved, None)
        stride, pad, dil, has_bias = ctx.cfg
        xa, wa = ctx.amax
        dy = _chk(dy, "conv2d: dy")
        B, Cin, H, W = ctx.x_shape
        reflect = ctx.reflect
        if reflect:                 # the layer the kernels see: pad 0 on the mirrored (H + 2r) x (W + 2r) input
            H, W = H + 2 * reflect, W + 2 * reflect
        Cout, _, KH, KW = weight.shape
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = dw = db = None
        bf = wa is None           # the flavour the forward ran
        ga = None
        if not bf:
            ga = _recall_amax(dy)
            if ga is None:
                ga = absmax(dy)
        q = dil * (KH - 1) - pad
        nhwc = bf or CONV_NHWC_F16X3
        dx_nhwc = need_x and nhwc and stride == 1 and q >= 0 and KW == KH and _conv_nhwc_ok(Cout, Cin, KH, KW, 1)
        dyp = None
        if dx_nhwc or (need_w and xp is not None):
            # dy as the NHWC operand (bf16, or f16 hi/lo planes), with the border the input gradient needs
            dyp = conv_nhwc_prep(dy, q if dx_nhwc else 0) if bf else conv_nhwc_prep_split(dy, q if dx_nhwc else 0, False, ga)
        fold = bool(dx_nhwc and reflect == 1 and H - 2 >= 4 and W - 2 >= 4)     # the mirrored border folded back by the GEMM itself
        if need_x:
            if dx_nhwc and not bf:
                th, tl, ts = _conv_weight_planes(weight, wa, 1)
                dx = _conv_nhwc_split_call(dyp, th, tl, ts, ga, None, Cin, KH, KW, dil, fold=fold)
            elif dx_nhwc:
                th, _, _ = _conv_weight_planes(weight, None, 1)
                dx = _conv_nhwc_call(dyp, th, None, Cin, KH, KW, dil, fold=fold)
            elif stride == 1 and q >= 0 and KW == KH:
                # dx = conv(dy, flipped weights with the channel roles swapped, padding d(K-1)-p): the same kernel
                th, tl, ts = _conv_weight_planes(weight, wa, 1)
                dx = _conv_fwd_call(dy, th, tl, ts, ga, None, Cin, KH, KW, 1, dil * (KH - 1) - pad, dil)
            elif stride > 1 and dil == 1 and (dx := _conv_dgrad_strided((B, Cin, H, W), weight, dy, ga, wa, stride, pad)) is not None:
                pass    # stride^2 parity classes, each a stride-1 convolution of dy scattered into dx (K16)
            else:       # whatever is left (dilated + strided, rectangular kernels with odd paddings): the framework
                dx = torch.nn.grad.conv2d_input((B, Cin, H, W), weight, dy, stride=stride, padding=pad, dilation=dil)
        if need_w and xp is not None:
            lib = _lib.load()
            Hp, Wp, Cp = xp.shape[-3:]
            S = lib.cocos_conv2d_nhwc_wgrad_bf16_slices(B, dy.shape[2], dy.shape[3], Cp, Cout, KH, KW)
            part = torch.empty((S, Cout, lib.cocos_conv2d_kdim(Cin, KH, KW)), device=dy.device, dtype=torch.float32)
            if bf:
                _call("conv2d_wgrad", "cocos_conv2d_nhwc_wgrad_bf16", xp.data_ptr(), dyp.data_ptr(), part.data_ptr(), B, Cp, Hp, Wp,
                      Cout, q if dx_nhwc else 0, KH, KW, dil, stride, _stream())
            else:
                _call("conv2d_wgrad", "cocos_conv2d_nhwc_wgrad_f16x3", xp.data_ptr(), dyp.data_ptr(), xa.data_ptr(), ga.data_ptr(),
                      part.data_ptr(), B, Cp, Hp, Wp, Cout, q if dx_nhwc else 0, KH, KW, dil, stride, _stream())
            dw = torch.empty_like(weight)
            _call("conv2d_wgrad", "cocos_conv2d_wgrad_reduce", part.data_ptr(), dw.data_ptr(), S, Cout, Cin, KH, KW, _stream())
        elif need_w:
            lib = _lib.load()
            S = lib.cocos_conv2d_wgrad_slices(B, Cin, H, W, Cout, KH, KW, stride, pad, dil)
            kdim = lib.cocos_conv2d_kdim(Cin, KH, KW)
            part = torch.empty((S, Cout, kdim), device=x.device, dtype=torch.float32)
            if bf:
                _call("conv2d_wgrad", "cocos_conv2d_wgrad_bf16", x.data_ptr(), dy.data_ptr(), part.data_ptr(), B, Cin, H, W,
                      Cout, KH, KW, stride, pad, dil, _stream())
            else:
                _call("conv2d_wgrad", "cocos_conv2d_wgrad_f16x3", x.data_ptr(), dy.data_ptr(), xa.data_ptr(), ga.data_ptr(),
                      part.data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, pad, dil, _stream())
            dw = torch.empty_like(weight)          # sum over the S slices + back to [Cout, Cin, KH, KW] in one pass
            _call("conv2d_wgrad", "cocos_conv2d_wgrad_reduce", part.data_ptr(), dw.data_ptr(), S, Cout, Cin, KH, KW, _stream())
        if need_b and has_bias:
            db = channel_sum(dy)
        if reflect and dx is not None and not fold:          # fold the mirrored border back (K18's backward gather)
            dxp, dx = dx, torch.empty(ctx.x_shape, device=dy.device, dtype=torch.float32)
            _call("reflect_pad2d_bwd", "cocos_reflect_pad2d_bwd", dxp.data_ptr(), dx.data_ptr(), B * Cin, ctx.x_shape[2], ctx.x_shape[3],
                  reflect, _stream())
        return dx, dw, db, None, None, None, None


def conv2d(x, weight, bias=None, stride: int = 1, padding: int = 0, dilation: int = 1, reflect: int = 0):
    """torch.nn.functional.conv2d(x, weight, bias, stride, padding, dilation) for groups = 1 and one stride / padding /
    dilation for both axes: fp32 in and out, products on the f16 MFMA with split operands (3 terms, fp32 accumulate) —
    conv_f16x3.hip — or, under COCOS_CONV=bf16, in one bf16 term (stride-1 layers with >= 128 output channels: conv_nhwc_bf16.hip).
    `reflect` = r > 0: the convolution of nn.ReflectionPad2d(r)(x) — on the K16b path the mirrored border is written by the
    operand preparation (no padded fp32 tensor), everywhere else it is ops.reflect_pad2d followed by the plain layer."""
    reflect = int(reflect)
    if reflect:
        Cout, Cin, KH, KW = weight.shape
        ow = x.shape[3] + 2 * reflect + 2 * int(padding) - int(dilation) * (KW - 1)
        fused = ((_conv_bf16() or CONV_NHWC_F16X3) and int(stride) == 1 and int(padding) == 0 and x.dim() == 4 and reflect < min(x.shape[2:])
                 and ow >= 32 and ow % 32 == 0 and _conv_nhwc_ok(Cin, Cout, KH, KW, 1))
        if not fused:
            x, reflect = reflect_pad2d(x, reflect), 0
    return _Conv2d.apply(x, weight, bias, int(stride), int(padding), int(dilation), reflect)


# ------------------------------------------------------------------------------------------
# K6  match_kernel = 3 logits from the K = 256 correlation (no unfold)   (correspondence.py:276-304)
# ------------------------------------------------------------------------------------------
class _Box3Logits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c_raw, mu, nu, a, b, h: int, w: int, k_unfolded: float, scale: float):
        c_raw = _chk(c_raw, "c_raw")
        mu, nu, a, b = (_chk(t, n) for t, n in ((mu, "mu"), (nu, "nu"), (a, "a"), (b, "b")))
        B, N, _ = c_raw.shape
        if N != h * w or any(t.shape != (B, N) for t in (mu, nu, a, b)):
            raise ValueError("box3_logits: shape mismatch")
        f = torch.empty_like(c_raw)
        _call("box3_logits_fwd", "cocos_box3_logits_fwd", c_raw.data_ptr(), mu.data_ptr(), nu.data_ptr(),
              a.data_ptr(), b.data_ptr(), f.data_ptr(), B, h, w, float(k_unfolded), float(scale), _stream())
        ctx.save_for_backward(f, mu, nu, a, b)
        ctx.cfg = (int(h), int(w), float(k_unfolded), float(scale))
        return f

    @staticmethod
    def backward(ctx, g):
        f, mu, nu, a, b = ctx.saved_tensors
        h, w, kc, scale = ctx.cfg
        g = _chk(g, "box3_logits: g")
        B, N, _ = f.shape
        dc = torch.empty_like(f)
        r1, r2, c1, c2 = (torch.empty((B, N), device=f.device, dtype=torch.float32) for _ in range(4))
        nbytes = _lib.load().cocos_box3_logits_bwd_workspace_bytes(B, h, w)
        ws = torch.empty((nbytes + 3) // 4, device=f.device, dtype=torch.float32)
        if PRECISION == "f16x3":          # max|dc| as a by-product: the K3 backward (the consumer) scales its split with it
            cell = _zero_cell(dc.device)
            _call("box3_logits_bwd", "cocos_box3_logits_bwd_amax", g.data_ptr(), f.data_ptr(), mu.data_ptr(),
                  nu.data_ptr(), a.data_ptr(), b.data_ptr(), dc.data_ptr(), r1.data_ptr(), r2.data_ptr(),
                  c1.data_ptr(), c2.data_ptr(), ws.data_ptr(), ws.numel() * 4, B, h, w, scale, cell.data_ptr(),
                  _stream())
            _remember_amax(dc, cell)
        else:
            _call("box3_logits_bwd", "cocos_box3_logits_bwd", g.data_ptr(), f.data_ptr(), mu.data_ptr(),
                  nu.data_ptr(), a.data_ptr(), b.data_ptr(), dc.data_ptr(), r1.data_ptr(), r2.data_ptr(),
                  c1.data_ptr(), c2.data_ptr(), ws.data_ptr(), ws.numel() * 4, B, h, w, scale, _stream())
        dmu, dnu, da, db = (torch.empty_like(r1) for _ in range(4))
        _call("box3_logits_bwd", "cocos_box3_stat_grads", r1.data_ptr(), r2.data_ptr(), c1.data_ptr(), c2.data_ptr(),
              a.data_ptr(), b.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), da.data_ptr(), db.data_ptr(), r1.numel(), kc, scale,
              _stream())
        return dc, dmu, dnu, da, db, None, None, None, None


def box3_logits(c_raw, mu, nu, a, b, h, w, k_unfolded, scale):
    """scale * (diagonal 3x3 box filter of c_raw - k mu nu^T) * a b^T  -> [B,N,N]; see cocos_hip.h K6."""
    return _Box3Logits.apply(c_raw, mu, nu, a, b, h, w, k_unfolded, scale)


# ------------------------------------------------------------------------------------------
# K19 / K20  match_kernel = 3 fused: x-box in the correlation GEMM's epilogue, y-box + softmax + warp in one kernel,
#            nothing box-filtered and no logits in HBM   (correspondence.py:276-291, :304, :307, :318; box3_fused_f16x3.hip)
# ------------------------------------------------------------------------------------------
#: "0": match_kernel 3 keeps the round-2 chain (K3 -> K6 -> K7) everywhere (A/B runs, tests of that chain)
BOX3_FUSED = True
#: the column pass of the cycle terms reads the row pass's T transposed and all passes share one gradient buffer (round 4)
BOX3_SHARE_T = True


#: T tensors of at least this many bytes lend their storage to the dC planes in the backward (see _Box3CorrXbox.backward): the
#: BASELINE config 5 class (128 x 128 grid: 1 GiB per sample).  Smaller ones keep a T that a retained graph may read again.
BOX3_ALIAS_T_BYTES = int(os.environ.get("COCOS_BOX3_ALIAS_T_BYTES", 2 << 30))


def box3_fused_ok(B, C, h, w, Cv=1):
    """Shapes the fused match_kernel-3 family takes (64- or 128-wide grid, whole 256-position tiles) on the split flavour."""
    return (BOX3_FUSED and PRECISION == "f16x3" and C == FUSED_K
            and bool(_lib.load().cocos_box3_fused_supported(h * w, h * w, min(Cv, MAX_FUSED_CV), h, w)))


class _Box3CorrXbox(torch.autograd.Function):
    """T = xbox(C_raw) with KEYS in the rows, tile-blocked (include/cocos_hip.h, K19).  Private contract with
    _Box3SoftmaxWarp: the gradient this node RECEIVES is G = dloss / d ybox(T) (same blocked layout), not dloss / dT — the
    y box of the adjoint is applied here, together with the x box, by K20."""

    @staticmethod
    def forward(ctx, q_raw, k_raw, sink=None, raw_planes=None):
        ctx.sink = sink
        q_raw, k_raw = _chk(q_raw, "box3_corr_xbox: q"), _chk(k_raw, "box3_corr_xbox: k")
        B, K, h, w = q_raw.shape
        N = h * w
        qp = kp = None
        if raw_planes is not None:        # K25 wrote the planes itself: q_raw / k_raw are handles (their memory is never read)
            qp, kp = raw_planes.get(q_raw), raw_planes.get(k_raw)
        ctx.chan = None
        if qp is not None and kp is not None:
            (qh, ql, qch, qcl, qs), (kh, kl, kch, kcl, ks) = qp, kp
            ctx.chan = ((qch, qcl, qs), (kch, kcl, ks))
            qa = ka = None
        else:
            if qp is not None or kp is not None:
                raise _lib.CocosHipError("box3_corr_xbox: producer-made planes for one operand only")
            qf, kf = q_raw.reshape(B, K, N), k_raw.reshape(B, K, N)
            qa, ka = _recall_amax(qf, consume=False), _recall_amax(kf, consume=False)   # left by K12 (both orientations read them)
            qa = absmax(qf) if qa is None else qa
            ka = absmax(kf) if ka is None else ka
            qh, ql, qs = split_f16(qf, True, amax=qa)            # position-major planes [B,N,K]
            kh, kl, ks = split_f16(kf, True, amax=ka)
        t = torch.empty(B * N * N, device=q_raw.device, dtype=torch.float32)
        _call("box3_corr_xbox", "cocos_box3_corr_xbox_f16x3", kh.data_ptr(), kl.data_ptr(), qh.data_ptr(), ql.data_ptr(),
              t.data_ptr(), B, N, N, K, w, ks.data_ptr(), qs.data_ptr(), _stream())
        ctx.save_for_backward(q_raw, k_raw)
        ctx.amax = (qa, ka)
        return t

    @staticmethod
    def backward(ctx, g):
        q_raw, k_raw = ctx.saved_tensors
        qa, ka = ctx.amax
        g = _chk(g, "box3_corr_xbox: g")
        B, K, h, w = q_raw.shape
        N = h * w
        gmax = _recall_amax(g)              # left by K19's backward; a sum of several passes' gradients: one pass over it
        if gmax is None:
            gmax = absmax(g)
        half = dict(device=g.device, dtype=torch.float16)
        t_dead = getattr(ctx.sink, "t_dead", None) if ctx.sink is not None else None
        if (t_dead is not None and t_dead.numel() == B * N * N and t_dead.dtype == torch.float32
                and B * N * N * 4 >= BOX3_ALIAS_T_BYTES):
            # round 5: T's last reader was the last pass's backward, which ran before this node — the dC planes
            # (two f16 planes = T's bytes) take T's storage instead of another B N^2 x 4 bytes (cfg5, B = 2: 5.1 -> 4.1 GiB peak).
            # The kernel writes through raw pointers, which autograd cannot see: the version counter of T is bumped by hand, so that
            # a second backward over a retained graph — T is a saved input of every pass — raises ("modified by an inplace
            # operation") instead of reading planes as logits.  Only for the large shapes (BOX3_ALIAS_T_BYTES).
            torch.autograd.graph.increment_version(t_dead)
            planes = t_dead.view(torch.float16)
            dch, dcl = planes[:B * N * N], planes[B * N * N:]
        else:
            dch, dcl = torch.empty(B * N * N, **half), torch.empty(B * N * N, **half)
        sc = torch.empty(1, device=g.device, dtype=torch.float32)
        _call("box3_adjoint_planes", "cocos_box3_adjoint_planes_f16x3", g.data_ptr(), gmax.data_ptr(), dch.data_ptr(),
              dcl.data_ptr(), sc.data_ptr(), B, N, N, h, w, _stream())
        dq = dk = None
        chan = ctx.chan
        if chan is not None and any(p is None for p in (chan[0][0], chan[1][0])):
            raise _lib.CocosHipError("box3_corr_xbox: backward without the channel-major planes (forward ran without grad)")
        if ctx.needs_input_grad[0]:         # d q_raw[c,P] = sum_Q dC[P,Q] k_raw[c,Q]
            ch, cl, cs = chan[1] if chan is not None else split_f16(k_raw.reshape(B, K, N), False, amax=ka)
            dq = torch.empty_like(q_raw)
            _call("box3_corr_grad", "cocos_hgemm_f16x3", ch.data_ptr(), cl.data_ptr(), dch.data_ptr(), dcl.data_ptr(),
                  dq.data_ptr(), B, K, N, N, 1.0, cs.data_ptr(), sc.data_ptr(), 3, _stream())
        if ctx.needs_input_grad[1]:         # d k_raw[c,Q] = sum_P dC[P,Q] q_raw[c,P]
            ch, cl, cs = chan[0] if chan is not None else split_f16(q_raw.reshape(B, K, N), False, amax=qa)
            dk = torch.empty_like(k_raw)
            _call("box3_corr_grad", "cocos_hgemm_f16x3", ch.data_ptr(), cl.data_ptr(), dch.data_ptr(), dcl.data_ptr(),
                  dk.data_ptr(), B, K, N, N, 1.0, cs.data_ptr(), sc.data_ptr(), 2, _stream())
        if ctx.sink is not None:
            ctx.sink.reset()      # (a second backward through the same graph starts a new G)
        return dq, dk, None, None


class Box3GradSink:
    """The ONE gradient buffer G of a T = box3_corr_xbox(...) that several box3_softmax_warp passes read (row pass, the column
    pass on the transposed T, a second row pass): the first pass's backward allocates G, writes it and hands it to autograd as
    d loss / d T; every later pass ADDS its G into the same buffer inside its kernel (COCOS_BOX3_G_ACCUMULATE) and returns no
    gradient for T — autograd never sums HWxHW tensors, and T's node runs ONE box adjoint + ONE pair of GEMMs for all passes.
    (T's node runs after all of its consumers, so the buffer is complete by then.)"""

    def __init__(self):
        self.buf = None
        self.gmax = None
        self.t_dead = None      # T as the LAST pass's backward saw it: nothing reads it after that (see _Box3CorrXbox.backward)

    def reset(self):
        self.buf = None
        self.gmax = None
        self.t_dead = None


def box3_corr_xbox(q_raw, k_raw, sink: Box3GradSink | None = None, raw_planes: Box3RawPlanes | None = None):
    """x-direction diagonal box filter of the K = 256 correlation of two raw [B,256,h,w] feature maps (64- or 128-wide grid),
    keys in the rows, in the tile-blocked layout of K19 (an opaque 1-D tensor of B*N*N floats).  `sink`: the gradient sink the
    box3_softmax_warp passes over this T share (see Box3GradSink)."""
    return _Box3CorrXbox.apply(q_raw, k_raw, sink, raw_planes)


class _Box3SoftmaxWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, mu, a, nu, b, v, h: int, w: int, kc: float, scale: float, transposed: bool = False, sink=None):
        ctx.transposed, ctx.sink = bool(transposed), sink
        t, v = _chk(t, "box3_softmax_warp: t"), _chk(v, "box3_softmax_warp: v")
        mu, a, nu, b = (_chk(x, n) for x, n in ((mu, "mu"), (a, "a"), (nu, "nu"), (b, "b")))
        B, N = mu.shape
        Cv = v.shape[1]
        if N != h * w or v.shape != (B, Cv, N) or t.numel() != B * N * N or any(x.shape != (B, N) for x in (a, nu, b)):
            raise ValueError("box3_softmax_warp: shape mismatch")
        v_amax = _recall_amax(v)
        if v_amax is None:
            v_amax = absmax(v)
        vh, vl, v_scale, v_lomask = split_f16_chan_mask(v, v_amax, VALUE_LO_SKIP and Cv > 32)
        out = torch.empty((B, Cv, N), device=v.device, dtype=torch.float32)
        lse = torch.empty((B, N), device=v.device, dtype=torch.float32)
        _call("box3_softmax_warp_fwd", "cocos_box3_softmax_warp_fwd_f16x3", t.data_ptr(), mu.data_ptr(), a.data_ptr(),
              nu.data_ptr(), b.data_ptr(), vh.data_ptr(), vl.data_ptr(), out.data_ptr(), lse.data_ptr(), v_scale.data_ptr(),
              _ptr(v_lomask), B, N, N, Cv, h, w, float(kc), float(scale), 1 if transposed else 0, _stream())
        ctx.v_lomask = v_lomask
        ctx.save_for_backward(t, mu, a, nu, b, v, out, lse)
        ctx.cfg = (int(h), int(w), float(kc), float(scale))
        ctx.v_amax = v_amax
        return out

    @staticmethod
    def backward(ctx, dout):
        t, mu, a, nu, b, v, out, lse = ctx.saved_tensors
        h, w, kc, scale = ctx.cfg
        dout = _chk(dout, "box3_softmax_warp: dout")
        B, N = mu.shape
        Cv = v.shape[1]
        cvp = (Cv + 31) // 32 * 32
        g_amax = _recall_amax(dout)          # left by the kernel that wrote dout (warp_head's backward, concat_channels_amax), else one pass
        if g_amax is None:
            g_amax = absmax(dout)
        d_pre = _rowdot_cached(dout, out) if BWD_D_PRECOMPUTED else None
        (gph, gpl, gs), (vph, vpl, v_scale) = _split_pair_transposed(dout, g_amax, v, ctx.v_amax, cvp)
        f32 = dict(device=v.device, dtype=torch.float32)
        sink, flags = ctx.sink, (1 if ctx.transposed else 0)
        if not ctx.needs_input_grad[0]:
            sink = None          # nobody differentiates T through this pass (autograd.grad(..., inputs=[v])): its G is scratch and must
                                 # not become — or be added into — the shared buffer
        if sink is not None and sink.buf is not None:      # another pass over this T has written its G: add ours to it
            g, g_ret = sink.buf, None
            gmax = sink.gmax = _zero_cell(v.device)        # ... and max|G| becomes that of the sum (a fresh zero cell)
            flags |= 2
        else:
            g = g_ret = torch.empty(B * N * N, **f32)
            gmax = _zero_cell(v.device)
            if sink is not None:
                sink.buf, sink.gmax = g, gmax
                # the buffer belongs to THIS backward: whatever happens to T's node (not reached, an exception on the way), the
                # engine drops it when the pass ends, so a later backward over a retained graph starts a new G
                torch.autograd.Variable._execution_engine.queue_callback(sink.reset)
        dmu, da, dnu, db = (torch.empty((B, N), **f32) for _ in range(4))
        colpart = torch.empty(_lib.load().cocos_box3_softmax_warp_bwd_colpart_bytes(B, N, N) // 4, **f32)
        need_v = ctx.needs_input_grad[5]
        psh = psl = None
        if need_v:        # cycle terms: V itself is differentiated -> the planes of 2^14 P for dv = dout . P
            psh = torch.empty(B * N * N, device=v.device, dtype=torch.float16)
            psl = torch.empty_like(psh)
        _call("box3_softmax_warp_bwd", "cocos_box3_softmax_warp_bwd_f16x3", t.data_ptr(), mu.data_ptr(), a.data_ptr(),
              nu.data_ptr(), b.data_ptr(), vph.data_ptr(), vpl.data_ptr(), gph.data_ptr(), gpl.data_ptr(), gs.data_ptr(),
              v_scale.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), g.data_ptr(), dmu.data_ptr(),
              da.data_ptr(), dnu.data_ptr(), db.data_ptr(), colpart.data_ptr(), gmax.data_ptr(), _ptr(psh), _ptr(psl),
              _ptr(ctx.v_lomask), B, N, N, Cv, cvp, h, w, kc, scale, _ptr(d_pre), flags, _stream())
        _remember_amax(g, gmax)      # (an accumulating pass replaces the cell remembered for the shared buffer)
        if sink is not None and t.numel() * 4 >= BOX3_ALIAS_T_BYTES:
            sink.t_dead = t          # (only when T's node will take the storage over: the reference keeps T alive until then)
        dv = None
        if need_v:
            gch, gcl, _ = split_f16(dout, False, amax=g_amax)
            dv = torch.empty_like(v)
            _call("box3_softmax_warp_bwd_dv", "cocos_hgemm_f16x3", gch.data_ptr(), gcl.data_ptr(), psh.data_ptr(),
                  psl.data_ptr(), dv.data_ptr(), B, Cv, N, N, 1.0 / 16384.0, gs.data_ptr(), None, 2, _stream())
        return g_ret, dmu, da, dnu, db, dv, None, None, None, None, None, None


def box3_softmax_warp(t, mu, a, nu, b, v, h, w, k_unfolded, scale, transposed: bool = False, sink: Box3GradSink | None = None):
    """out[b,c,p] = sum_q softmax_q(scale * a_p b_q (ybox(T)[p,q] - k mu_p nu_q)) v[b,c,q] with T from box3_corr_xbox —
    the whole match_kernel-3 attention of one orientation; v [B,Cv,N] in chunks of 160 channels.  `transposed`: T is the
    OTHER orientation's (mu / a then belong to T's keys, nu / b to its queries): the column pass on the row pass's T.
    `sink`: the Box3GradSink given to box3_corr_xbox — the passes then accumulate ONE gradient of T inside their kernels."""
    Cv = v.shape[1]
    if Cv <= MAX_FUSED_CV:
        return _Box3SoftmaxWarp.apply(t, mu, a, nu, b, v, h, w, k_unfolded, scale, transposed, sink)
    return torch.cat([_Box3SoftmaxWarp.apply(t, mu, a, nu, b, v[:, c0:c0 + MAX_FUSED_CV], h, w, k_unfolded, scale, transposed, sink)
                      for c0 in range(0, Cv, MAX_FUSED_CV)], dim=1)


# ------------------------------------------------------------------------------------------
# K12 statistics of the 3x3-unfolded vectors without unfolding   (correspondence.py:276-280, match_kernel 3)
# ------------------------------------------------------------------------------------------
class _Unfold3Stats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k_unfolded: float, eps: float):
        x = _chk(x, "unfold3_stats: x")
        B, C, h, w = x.shape
        mk = lambda: torch.empty((B, h * w), device=x.device, dtype=torch.float32)
        mu, a, nrm = mk(), mk(), mk()
        ws = torch.empty(2 * B * h * w, device=x.device, dtype=torch.float32)
        if PRECISION == "f16x3":      # max|x| as a by-product: the correlation GEMM of the fused family splits x next
            cell = _zero_cell(x.device)
            _call("unfold3_stats_fwd", "cocos_unfold3_stats_fwd_amax", x.data_ptr(), mu.data_ptr(), a.data_ptr(),
                  nrm.data_ptr(), ws.data_ptr(), B, C, h, w, float(k_unfolded), float(eps), cell.data_ptr(), _stream())
            _remember_amax(x, cell)
        else:
            _call("unfold3_stats_fwd", "cocos_unfold3_stats_fwd", x.data_ptr(), mu.data_ptr(), a.data_ptr(), nrm.data_ptr(),
                  ws.data_ptr(), B, C, h, w, float(k_unfolded), float(eps), _stream())
        ctx.save_for_backward(x, mu, a, nrm)
        ctx.kc = float(k_unfolded)
        ctx.mark_non_differentiable(nrm)
        return mu, a, nrm

    @staticmethod
    def backward(ctx, dmu, da, _dnrm):
        x, mu, a, nrm = ctx.saved_tensors
        B, C, h, w = x.shape
        dmu = None if dmu is None else _chk(dmu, "unfold3_stats: dmu")
        da = None if da is None else _chk(da, "unfold3_stats: da")
        dx = torch.empty_like(x)
        ws = torch.empty(2 * B * h * w, device=x.device, dtype=torch.float32)
        _call("unfold3_stats_bwd", "cocos_unfold3_stats_bwd", x.data_ptr(), mu.data_ptr(), a.data_ptr(), nrm.data_ptr(),
              _ptr(dmu), _ptr(da), dx.data_ptr(), ws.data_ptr(), B, C, h, w, ctx.kc, _stream())
        return dx, None, None


def unfold3_stats(x, k_unfolded: float, eps: float = NORM_EPS):
    """(mu, a) [B,h*w] of the zero-padded 3x3-unfolded, centred vectors of x [B,C,h,w]: mean over the 9*C entries and
    1 / (L2 norm of the centred vector + eps)."""
    mu, a, _ = _Unfold3Stats.apply(x, k_unfolded, eps)
    return mu, a


# ------------------------------------------------------------------------------------------
# K7  softmax + warp from materialised key-major logits   (correspondence.py:307 + :318 ...)
# ------------------------------------------------------------------------------------------
class _LogitsSoftmaxWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_t, v):
        logits_t, v = _chk(logits_t, "logits_t"), _chk(v, "v")
        B, Nk, Nq = logits_t.shape
        Cv = v.shape[1]
        if v.shape[0] != B or v.shape[2] != Nk:
            raise ValueError(f"logits_softmax_warp: shape mismatch {tuple(logits_t.shape)} {tuple(v.shape)}")
        out = torch.empty((B, Cv, Nq), device=v.device, dtype=torch.float32)
        lse = torch.empty((B, Nq), device=v.device, dtype=torch.float32)
        split = PRECISION == "f16x3" and Nk % 4 == 0 and Cv <= MAX_FUSED_CV and Nq * Nk * 4 < 2 ** 31 - 1
        ctx.v_amax = None
        if split:      # P.V on the f16 MFMA (logits_softmax_warp_f16x3.hip); V planes normalised on the device
            ctx.v_amax = _recall_amax(v)
            if ctx.v_amax is None:
                ctx.v_amax = absmax(v)
            vh, vl, v_scale = split_f16(v, False, amax=ctx.v_amax)
            _call("logits_softmax_warp_fwd", "cocos_logits_softmax_warp_fwd_f16x3", logits_t.data_ptr(), vh.data_ptr(),
                  vl.data_ptr(), out.data_ptr(), lse.data_ptr(), v_scale.data_ptr(), B, Nq, Nk, Cv, _stream())
        else:
            _call("logits_softmax_warp_fwd", "cocos_logits_softmax_warp_fwd", logits_t.data_ptr(), v.data_ptr(),
                  out.data_ptr(), lse.data_ptr(), B, Nq, Nk, Cv, _stream())
        ctx.split = split
        ctx.save_for_backward(logits_t, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        logits_t, v, out, lse = ctx.saved_tensors
        dout = _chk(dout, "dout")
        B, Nk, Nq = logits_t.shape
        Cv = v.shape[1]
        dlg = dv = None
        if ctx.needs_input_grad[0]:
            dlg = torch.empty_like(logits_t)
            if ctx.split:
                cvp = (Cv + 31) // 32 * 32
                gph, gpl, gs = split_f16(dout, True, cpad=cvp, amax=absmax(dout))
                vph, vpl, v_scale = split_f16(v, True, cpad=cvp, amax=ctx.v_amax)
                _call("logits_softmax_warp_bwd", "cocos_logits_softmax_warp_bwd_f16x3", logits_t.data_ptr(),
                      vph.data_ptr(), vpl.data_ptr(), gph.data_ptr(), gpl.data_ptr(), gs.data_ptr(), v_scale.data_ptr(),
                      out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dlg.data_ptr(), B, Nq, Nk, Cv, cvp, _stream())
            else:
                _call("logits_softmax_warp_bwd", "cocos_logits_softmax_warp_bwd", logits_t.data_ptr(), v.data_ptr(),
                      out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dlg.data_ptr(), B, Nq, Nk, Cv, _stream())
        if ctx.needs_input_grad[1]:
            # dv[c,j] = sum_i P[i,j] dout[c,i] (cycle terms only): P^T is rebuilt once and contracted with
            # the K5 GEMM; exp() here is the only non-HIP arithmetic, on a path no README command trains
            p_t = torch.exp(logits_t - lse[:, None, :])
            dv = torch.empty_like(v)
            _call("warp_materialized_fwd", "cocos_warp_materialized_fwd", p_t.data_ptr(), dout.data_ptr(),
                  dv.data_ptr(), B, Nk, Nq, Cv, _stream())
        return dlg, dv


def logits_softmax_warp(logits_t, v):
    """out[b,c,i] = sum_j softmax_j(logits_t[b,j,i]) v[b,c,j]; logits_t [B,Nk,Nq] key-major, v [B,Cv,Nk]."""
    Cv = v.shape[1]
    if Cv <= MAX_FUSED_CV:
        return _LogitsSoftmaxWarp.apply(logits_t, v)
    return torch.cat([_LogitsSoftmaxWarp.apply(logits_t, v[:, c0:c0 + MAX_FUSED_CV])
                      for c0 in range(0, Cv, MAX_FUSED_CV)], dim=1)


# ------------------------------------------------------------------------------------------
# K8  WTA_scale + /temperature on the materialised correlation   (correspondence.py:38-77, :300-304)
# ------------------------------------------------------------------------------------------
class _WTAScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, scale: float, post_scale: float):
        f = _chk(f, "wta_scale: f")
        cols = f.shape[-1]
        rows = f.numel() // cols
        nbytes = _lib.load().cocos_wta_scale_mask_bytes(rows, cols)
        if nbytes < 0:
            raise ValueError(f"wta_scale: bad shape {tuple(f.shape)}")
        mask = torch.empty(nbytes, device=f.device, dtype=torch.uint8)      # 1 bit per element
        y = torch.empty_like(f)
        _call("wta_scale_fwd", "cocos_wta_scale_fwd", f.data_ptr(), y.data_ptr(), mask.data_ptr(), rows, cols,
              float(scale), float(post_scale), _stream())
        ctx.save_for_backward(mask)
        ctx.post = float(post_scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _chk(dy, "wta_scale: dy")
        cols = dy.shape[-1]
        dx = torch.empty_like(dy)
        _call("wta_scale_bwd", "cocos_wta_scale_bwd", dy.data_ptr(), mask.data_ptr(), dx.data_ptr(),
              dy.numel() // cols, cols, ctx.post, _stream())
        return dx, None, None


def wta_scale(f, scale: float, post_scale: float = 1.0):
    """WTA_scale.apply(f, scale) * post_scale: each row's maxima kept, the rest multiplied by `scale`;
    backward = grad * post_scale * (1 at the maxima, the reference's fixed 1e-4 elsewhere)."""
    return _WTAScale.apply(f, scale, post_scale)


# ------------------------------------------------------------------------------------------
# K9  PONO + SPADE modulation + LeakyReLU   (normalization.py:63-68,:148-151; architecture.py:88-95)
# ------------------------------------------------------------------------------------------
PONO_EPS = 1e-5   # PositionalNorm2d's epsilon (normalization.py:63)


class _PonoSpade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, slope: float, eps: float):
        x, gamma, beta = _chk(x, "pono_spade: x"), _chk(gamma, "pono_spade: gamma"), _chk(beta, "pono_spade: beta")
        if gamma.shape != x.shape or beta.shape != x.shape:
            raise ValueError(f"pono_spade: x{tuple(x.shape)} gamma{tuple(gamma.shape)} beta{tuple(beta.shape)}")
        B, C = x.shape[:2]
        N = x.numel() // (B * C)
        y = torch.empty_like(x)
        if CONV_PRECISION == "f16x3":         # max|y| as a by-product: y is the input of the block's next convolution
            cell = _zero_cell(x.device)
            part = torch.empty(_lib.load().cocos_pono_spade_amax_partials(B, C, N), device=x.device, dtype=torch.float32)
            _call("pono_spade_fwd", "cocos_pono_spade_fwd_amax", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                  cell.data_ptr(), part.data_ptr(), B, C, N, float(eps), float(slope), _stream())
            _remember_amax(y, cell, weak=True)
        else:
            _call("pono_spade_fwd", "cocos_pono_spade_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                  y.data_ptr(), B, C, N, float(eps), float(slope), _stream())
        ctx.save_for_backward(x, gamma, beta)
        ctx.cfg = (float(slope), float(eps))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        slope, eps = ctx.cfg
        dy = _chk(dy, "pono_spade: dy")
        B, C = x.shape[:2]
        N = x.numel() // (B * C)
        need_x, need_g, need_b = ctx.needs_input_grad[:3]
        dx = torch.empty_like(x) if need_x else None
        dg = torch.empty_like(x) if need_g else None
        db = torch.empty_like(x) if need_b else None
        if (dg is not None or db is not None) and CONV_PRECISION == "f16x3":
            # max|dgamma| / max|dbeta| as by-products: the output gradients of SPADE's mlp_gamma / mlp_beta convolutions
            cells = _zero_cells(x.device, 2)
            part = torch.empty(2 * _lib.load().cocos_pono_spade_amax_partials(B, C, N), device=x.device, dtype=torch.float32)
            _call("pono_spade_bwd", "cocos_pono_spade_bwd_amax", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dy.data_ptr(), _ptr(dx),
                  _ptr(dg), _ptr(db), cells.data_ptr(), part.data_ptr(), B, C, N, eps, slope, _stream())
            if dg is not None:
                _remember_amax(dg, cells[0:1], weak=True)
            if db is not None:
                _remember_amax(db, cells[1:2], weak=True)
        else:
            _call("pono_spade_bwd", "cocos_pono_spade_bwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                  dy.data_ptr(), _ptr(dx), _ptr(dg), _ptr(db), B, C, N, eps, slope, _stream())
        return dx, dg, db, None, None


def pono_spade(x, gamma, beta, slope: float = 1.0, eps: float = PONO_EPS):
    """leaky_relu(PositionalNorm2d(x) * (1 + gamma) + beta, slope) for x, gamma, beta [B,C,H,W]."""
    return _PonoSpade.apply(x, gamma, beta, slope, eps)


class _ReflectPad2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad: int):
        x = _chk(x, "reflect_pad2d: x")
        if x.dim() != 4 or not (0 <= pad < min(x.shape[2:])):
            raise ValueError(f"reflect_pad2d: x{tuple(x.shape)} pad={pad} (pad must be smaller than H and W)")
        B, C, H, W = x.shape
        y = torch.empty((B, C, H + 2 * pad, W + 2 * pad), device=x.device, dtype=torch.float32)
        _call("reflect_pad2d_fwd", "cocos_reflect_pad2d_fwd", x.data_ptr(), y.data_ptr(), B * C, H, W, int(pad), _stream())
        ctx.cfg = (B, C, H, W, int(pad))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, pad = ctx.cfg
        dy = _chk(dy, "reflect_pad2d: dy")
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32)
        _call("reflect_pad2d_bwd", "cocos_reflect_pad2d_bwd", dy.data_ptr(), dx.data_ptr(), B * C, H, W, pad, _stream())
        return dx, None


def reflect_pad2d(x, pad: int):
    """nn.ReflectionPad2d(pad)(x) for x [B,C,H,W] (one padding for the four sides, pad < H, W): K18, backward as a gather."""
    return _ReflectPad2d.apply(x, int(pad))


class _SpadeModulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xh, gamma, beta, slope: float):
        xh, gamma, beta = (_chk(t, f"spade_modulate: {n}") for t, n in ((xh, "xh"), (gamma, "gamma"), (beta, "beta")))
        if gamma.shape != xh.shape or beta.shape != xh.shape:
            raise ValueError(f"spade_modulate: xh{tuple(xh.shape)} gamma{tuple(gamma.shape)} beta{tuple(beta.shape)}")
        y = torch.empty_like(xh)
        _call("spade_modulate_fwd", "cocos_spade_modulate_fwd", xh.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              y.data_ptr(), xh.numel(), float(slope), _stream())
        ctx.save_for_backward(xh, gamma, beta)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, gamma, beta = ctx.saved_tensors
        dy = _chk(dy, "spade_modulate: dy")
        need = ctx.needs_input_grad[:3]
        outs = [torch.empty_like(xh) if w else None for w in need]
        _call("spade_modulate_bwd", "cocos_spade_modulate_bwd", xh.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              dy.data_ptr(), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), xh.numel(), ctx.slope, _stream())
        return outs[0], outs[1], outs[2], None


def spade_modulate(xh, gamma, beta, slope: float = 1.0):
    """leaky_relu(xh * (1 + gamma) + beta, slope) where xh is ALREADY normalised (instance / batch / sync-batch norm):
    the non-PONO tail of SPADE.forward (normalization.py:148) + the block's activation, one pass each way (K17)."""
    return _SpadeModulate.apply(xh, gamma, beta, slope)


# ------------------------------------------------------------------------------------------
# K11 nearest-neighbour up-sampling of the warped image   (correspondence.py:188, :327)
# ------------------------------------------------------------------------------------------
class _UpsampleNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale: int):
        x = _chk(x, "upsample_nearest: x")
        B, C, h, w = x.shape
        y = torch.empty((B, C, h * scale, w * scale), device=x.device, dtype=torch.float32)
        _call("upsample_nearest_fwd", "cocos_upsample_nearest_fwd", x.data_ptr(), y.data_ptr(), B * C, h, w, int(scale),
              _stream())
        ctx.cfg = (B, C, h, w, int(scale))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, h, w, scale = ctx.cfg
        dy = _chk(dy, "upsample_nearest: dy")
        dx = torch.empty((B, C, h, w), device=dy.device, dtype=torch.float32)
        _call("upsample_nearest_bwd", "cocos_upsample_nearest_bwd", dy.data_ptr(), dx.data_ptr(), B * C, h, w, scale,
              _stream())
        return dx, None


def upsample_nearest(x, scale: int):
    """F.interpolate(x, scale_factor=scale, mode='nearest') for integer scales (x [B,C,h,w] fp32, (w*scale) % 4 == 0)."""
    return _UpsampleNearest.apply(x, scale)




class _WarpHead(torch.autograd.Function):
    """(warp_out, warp_mask) from the first row pass's output o [B,Ci+Cs,h*w] (correspondence.py:327, :334): nearest up-sampling
    of the Ci image channels and a VIEW of the Cs mask channels.  Backward: ONE kernel (cocos_warp_head_bwd) writes d o from
    the two loss gradients and leaves max|d o| and D = sum_c d o * o (fp64) for the K2 / K19 backward that consumes d o next —
    round 5 ran cocos_upsample_nearest_bwd, cocos_concat2_amax and cocos_rowdot_f64 for these."""

    @staticmethod
    def forward(ctx, o, n_img: int, h: int, w: int, down: int):
        o = _chk(o, "warp_head: o")
        B, C, N = o.shape
        y = torch.empty((B, n_img, h * down, w * down), device=o.device, dtype=torch.float32)
        _call("upsample_nearest_fwd", "cocos_warp_head_fwd", o.data_ptr(), y.data_ptr(), B, n_img, C, h, w, down, _stream())
        ctx.save_for_backward(o)
        ctx.cfg = (int(n_img), int(h), int(w), int(down))
        return y, o[:, n_img:].view(B, C - n_img, h, w)

    @staticmethod
    def backward(ctx, g_img, g_mask):
        o, = ctx.saved_tensors
        n_img, h, w, down = ctx.cfg
        B, C, N = o.shape
        if g_img is None:
            g_img = torch.zeros((B, n_img, h * down, w * down), device=o.device, dtype=torch.float32)
        if g_mask is None:
            g_mask = torch.zeros((B, C - n_img, h, w), device=o.device, dtype=torch.float32)
        g_img, g_mask = _chk(g_img, "warp_head: d warp_out"), _chk(g_mask, "warp_head: d warp_mask")
        dout = torch.empty_like(o)
        drow = torch.empty((B, N), device=o.device, dtype=torch.float32)
        cell = _zero_cell(o.device)
        _call("warp_head_bwd", "cocos_warp_head_bwd", g_img.data_ptr(), g_mask.data_ptr(), o.data_ptr(), dout.data_ptr(), drow.data_ptr(),
              cell.data_ptr(), B, n_img, C - n_img, h, w, down, _stream())
        if PRECISION == "f16x3":
            _remember_amax(dout, cell)
        _tls.known_rowdot = (weakref.ref(dout), dout._version, o.data_ptr(), drow)
        return dout, None, None, None, None


def warp_head_ok(o: torch.Tensor, n_img: int, h: int, w: int, down: int) -> bool:
    """Shapes cocos_warp_head_fwd / _bwd take: fp32 on the GPU, some image and some mask channels, grid width a multiple of 4."""
    return (o.is_cuda and o.dtype == torch.float32 and o.dim() == 3 and 0 < n_img < o.shape[1] and w % 4 == 0
            and o.shape[2] == h * w and o.shape[0] <= 65535)


def warp_head(o, n_img: int, h: int, w: int, down: int):
    """(warp_out [B,n_img,h*down,w*down], warp_mask [B,C-n_img,h,w]) of the row pass's output o [B,C,h*w]: see _WarpHead."""
    return _WarpHead.apply(o, int(n_img), int(h), int(w), int(down))


def warp_values(img, seg_map, down: int):
    """torch.cat((F.avg_pool2d(img, down), F.interpolate(seg_map, scale_factor=1/down, mode='nearest')), 1) in one
    kernel (K14; correspondence.py:314, :318-319, :331-334): img [B,Ci,H,W], seg_map [B,Cs,H,W] or None -> [B,Ci+Cs,
    H/down,W/down].  Forward only: the exemplar image and its label map are data (callers with gradients use torch)."""
    img = _chk(img, "warp_values: img")
    B, Ci, H, W = img.shape
    seg = None if seg_map is None else _chk(seg_map, "warp_values: seg_map")
    Cs = 0 if seg is None else seg.shape[1]
    if seg is not None and (seg.shape[0], seg.shape[2], seg.shape[3]) != (B, H, W):
        raise ValueError(f"warp_values: seg_map {tuple(seg.shape)} does not match img {tuple(img.shape)}")
    out = torch.empty((B, Ci + Cs, H // down, W // down), device=img.device, dtype=torch.float32)
    if PRECISION == "f16x3":      # max|V| as a by-product: the consumer (the K2 forward's f16 split of V) picks it up
        cell = _zero_cell(out.device)
        _call("warp_values", "cocos_warp_values_amax", img.data_ptr(), _ptr(seg), out.data_ptr(), B, Ci, Cs, H, W, int(down),
              cell.data_ptr(), _stream())
        _remember_amax(out, cell)
        return out
    _call("warp_values", "cocos_warp_values", img.data_ptr(), _ptr(seg), out.data_ptr(), B, Ci, Cs, H, W, int(down),
          _stream())
    return out


def concat_channels_amax(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat((a, b), dim=1) of two fp32 CUDA tensors whose per-sample sizes are multiples of 4, with max|result| left for
    the consumer (the f16 split of the K2 backward's dout) — one kernel instead of a framework copy + a max pass."""
    a, b = _chk(a, "concat_channels_amax: a"), _chk(b, "concat_channels_amax: b")
    B = a.shape[0]
    na, nb = a.numel() // B, b.numel() // B
    if b.shape[0] != B or a.shape[2:] != b.shape[2:] or na % 4 or nb % 4 or a.data_ptr() % 16 or b.data_ptr() % 16:
        return torch.cat((a, b), dim=1)
    out = torch.empty((B, a.shape[1] + b.shape[1]) + tuple(a.shape[2:]), device=a.device, dtype=torch.float32)
    cell = _zero_cell(out.device) if PRECISION == "f16x3" else None
    _call("concat2_amax", "cocos_concat2_amax", a.data_ptr(), b.data_ptr(), out.data_ptr(), B, na, nb, _ptr(cell), _stream())
    if cell is not None:
        _remember_amax(out, cell)
    return out


# ------------------------------------------------------------------------------------------
# K13 InstanceNorm (+ residual) + PReLU of the ResidualBlocks   (correspondence.py:13-36)
# ------------------------------------------------------------------------------------------
INSTNORM_EPS = 1e-5   # nn.InstanceNorm2d default


class _InstNormPReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, eps: float):
        x = _chk(x, "instnorm_prelu: x")
        res = None if residual is None else _chk(residual, "instnorm_prelu: residual")
        w = _chk(weight, "instnorm_prelu: weight")
        if w.numel() != 1:
            raise ValueError("instnorm_prelu: nn.PReLU() with a single parameter expected")
        if res is not None and res.shape != x.shape:
            raise ValueError(f"instnorm_prelu: residual {tuple(res.shape)} vs x {tuple(x.shape)}")
        B, C = x.shape[:2]
        N = x.numel() // (B * C)
        y = torch.empty_like(x)
        if CONV_PRECISION == "f16x3":
            # max|y| as a by-product: y is the next convolution's (or the projections') input, split with that scale
            cell = _zero_cell(x.device)
            part = torch.empty(B * C, device=x.device, dtype=torch.float32)       # one maximum per plane, reduced by the same call
            _call("instnorm_prelu_fwd", "cocos_instnorm_prelu_fwd_amax", x.data_ptr(), _ptr(res), w.data_ptr(), y.data_ptr(), cell.data_ptr(),
                  part.data_ptr(), B * C, N, float(eps), _stream())
            _remember_amax(y, cell, weak=True)
        else:
            _call("instnorm_prelu_fwd", "cocos_instnorm_prelu_fwd", x.data_ptr(), _ptr(res), w.data_ptr(), y.data_ptr(),
                  B * C, N, float(eps), _stream())
        ctx.save_for_backward(x, res, w)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, w = ctx.saved_tensors
        dy = _chk(dy, "instnorm_prelu: dy")
        B, C = x.shape[:2]
        N = x.numel() // (B * C)
        need_x, need_r, need_w = ctx.needs_input_grad[:3]
        dx = torch.empty_like(x) if need_x else None
        dr = torch.empty_like(x) if (need_r and res is not None) else None
        dap = dw = None
        if need_w:
            # the weight's gradient is ONE number summed over every element of the layer, with cancelling terms: fp64 from the
            # products to the last addition (cocos_instnorm_prelu_bwd_f64; in fp32 it was up to 90x further from fp64 than the framework's)
            dap = torch.empty(B * C, device=x.device, dtype=torch.float64)
            dw = torch.empty(1, device=x.device, dtype=torch.float32)
        if dx is not None and CONV_PRECISION == "f16x3":
            # max|dx| as a by-product: dx is the output gradient of the convolution in front of the norm, whose backward splits it next
            cell = _zero_cell(x.device)
            part = torch.empty(B * C, device=x.device, dtype=torch.float32)
            _call("instnorm_prelu_bwd", "cocos_instnorm_prelu_bwd_amax", x.data_ptr(), _ptr(res), w.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                  _ptr(dr), _ptr(dap), _ptr(dw), cell.data_ptr(), part.data_ptr(), B * C, N, ctx.eps, _stream())
            _remember_amax(dx, cell, weak=True)
        elif need_w:
            _call("instnorm_prelu_bwd", "cocos_instnorm_prelu_bwd_f64", x.data_ptr(), _ptr(res), w.data_ptr(), dy.data_ptr(),
                  _ptr(dx), _ptr(dr), dap.data_ptr(), dw.data_ptr(), B * C, N, ctx.eps, _stream())
        else:
            _call("instnorm_prelu_bwd", "cocos_instnorm_prelu_bwd", x.data_ptr(), _ptr(res), w.data_ptr(), dy.data_ptr(),
                  _ptr(dx), _ptr(dr), None, B * C, N, ctx.eps, _stream())
        return dx, dr, (dw.reshape(w.shape) if need_w else None), None


def instnorm_prelu(x, residual, weight, eps: float = INSTNORM_EPS):
    """prelu(InstanceNorm2d(x) [+ residual], weight) for x [B,C,h,w]: nn.InstanceNorm2d(affine=False) statistics per
    (sample, channel) plane, nn.PReLU() with one parameter."""
    return _InstNormPReLU.apply(x, residual, weight, eps)


# ------------------------------------------------------------------------------------------
# K21  torch.nn.utils.spectral_norm's weight: power iteration + W / sigma, and its backward   (spectral_norm.hip)
# ------------------------------------------------------------------------------------------
class _SpectralWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, u, v, power_iteration: bool, eps: float):
        w = _chk(weight, "spectral_weight: weight")
        R = w.shape[0]
        K = w.numel() // R
        if u.shape != (R,) or v.shape != (K,) or not (u.is_contiguous() and v.is_contiguous()):
            raise ValueError(f"spectral_weight: weight {tuple(w.shape)} with u {tuple(u.shape)} / v {tuple(v.shape)}")
        lib = _lib.load()
        wsn = torch.empty_like(w)
        sigma = torch.empty(1, device=w.device, dtype=torch.float32)
        ws = torch.empty(lib.cocos_spectral_weight_workspace_floats(R, K), device=w.device, dtype=torch.float32)
        # max|W / sigma| as a by-product: the convolution that consumes the weight splits it with it — only the f16x3
        # flavour does (bf16 planes carry no scale: the entries would pile up in the table and evict useful ones)
        cell = _zero_cell(w.device) if CONV_PRECISION != "bf16" else None
        _call("spectral_weight_fwd", "cocos_spectral_weight_fwd", w.data_ptr(), u.data_ptr(), v.data_ptr(), wsn.data_ptr(), sigma.data_ptr(),
              _ptr(cell), ws.data_ptr(), R, K, float(eps), int(bool(power_iteration)), _stream())
        if cell is not None:
            _remember_amax(wsn, cell)
        # the vectors sigma was taken with: copies, because the next forward (GAN training: D(real), D(fake)) updates the buffers in place
        ctx.save_for_backward(w, u.clone(), v.clone(), sigma)
        return wsn

    @staticmethod
    def backward(ctx, g):
        w, u, v, sigma = ctx.saved_tensors
        g = _chk(g, "spectral_weight: grad")
        R = w.shape[0]
        K = w.numel() // R
        dw = torch.empty_like(w)
        ws = torch.empty(1024, device=w.device, dtype=torch.float32)
        _call("spectral_weight_bwd", "cocos_spectral_weight_bwd", g.data_ptr(), w.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(),
              dw.data_ptr(), ws.data_ptr(), R, K, _stream())
        return dw, None, None, None, None


def spectral_weight(weight, u, v, power_iteration: bool, eps: float = 1e-12):
    """`SpectralNorm.compute_weight` of torch.nn.utils.spectral_norm (dim 0, one power iteration): u, v updated in place when
    `power_iteration`, returns weight / (u . W v) with the framework's gradient (through sigma as well) — K21."""
    return _SpectralWeight.apply(weight, u, v, bool(power_iteration), float(eps))


#: softmax_attention on the fused kernels for K < 256 as well (channels zero-padded to 256); "0": the materialised family
ATTENTION_FUSED = True


def softmax_attention(q, k, v, scale: float = 1.0):
    """out[b,c,i] = sum_j softmax_j(scale * <q[b,:,i], k[b,:,j]>) v[b,c,j] for any channel count K — the QK^T ->
    softmax -> PV op class of the path (`Attention.forward`, architecture.py:114-127, has K =
    ch/8 = 32..64, HW/4 keys and ch/2 value channels).

    K <= 256 on the split flavour: the FUSED kernels (K2) — nothing HWxHW is materialised in inference, training keeps the
    saved logits like the correspondence itself.  q / k are raw 1x1-conv outputs here, not unit-norm columns: their operand
    planes get device-side power-of-two scales from max|q|, max|k| (no fixed 2^4: ADVICE r2) and the kernels run in their
    magnitude-free flavour (exact exponent differences at any |logit|).  K < 256 sits zero-padded in the 256-channel planes
    the kernels are built for, and their K = 32 / K = 64 instantiations leave out the matrix steps, fragment reads and
    fetches of the padding; the two backward GEMMs (M = K resp. Cv rows: too few tiles for 256 CUs) split their reduction.
    netG's shape (B=4, K=32, 128x128 queries, 64x64 keys, Cv=128): 2.2 ms fwd+bwd against 6.3 ms materialised (round 3: 3.0;
    tools/attention_bench.py).  Everything else: the materialised family on the same MFMA GEMMs (K3 -> K4 -> K5)."""
    B, K, Nq = q.shape
    Nk = k.shape[2]
    keep = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
    if (ATTENTION_FUSED and K <= FUSED_K and q.is_cuda and q.dtype == torch.float32
            and corr_split_ok(B, FUSED_K, Nq, Nk, v.shape[1], keep)):
        return corr_softmax_warp(q, k, v, scale, operand_amax=True)
    if K == FUSED_K and q.is_cuda and q.dtype == torch.float32:
        # the split path is not available (COCOS_PRECISION=fp32, odd Nq / Nk): K = 256 still takes the FUSED exact-fp32
        # kernels, which materialise nothing either (this branch had been lost — a B x Nq x Nk matrix at 4096^2)
        return corr_softmax_warp(q, k, v, scale, precision="fp32")
    return warp_materialized(row_softmax(corr_materialize(q, k, scale)), v)


# ------------------------------------------------------------------------------------------
# K15 row reductions of the contextual loss   (ContextualLoss.py:121-133)
# ------------------------------------------------------------------------------------------
class _ContextualRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cosm, h: float, eps: float):
        cosm = _chk(cosm, "contextual_rows: cos")
        cols = cosm.shape[-1]
        rows = cosm.numel() // cols
        cx = torch.empty(cosm.shape[:-1], device=cosm.device, dtype=torch.float32)
        _call("contextual_rows_fwd", "cocos_contextual_rows_fwd", cosm.data_ptr(), cx.data_ptr(), rows, cols, float(h),
              float(eps), _stream())
        ctx.save_for_backward(cosm)
        ctx.cfg = (float(h), float(eps))
        return cx

    @staticmethod
    def backward(ctx, dcx):
        (cosm,) = ctx.saved_tensors
        h, eps = ctx.cfg
        dcx = _chk(dcx, "contextual_rows: dcx")
        cols = cosm.shape[-1]
        dcos = torch.empty_like(cosm)
        _call("contextual_rows_bwd", "cocos_contextual_rows_bwd", cosm.data_ptr(), dcx.data_ptr(), dcos.data_ptr(),
              cosm.numel() // cols, cols, h, eps, _stream())
        return dcos, None, None


def contextual_rows(cosm, h: float = 0.1, eps: float = 1e-3):
    """cx[..., i] = max_j A[..., i, j] of the contextual affinity A built from the cosine matrix cosm [..., N, M <= 4096]
    (d = 1 - cos, d / (min_j d + eps), exp((1 - .) / h), row-normalised): ContextualLoss.py:121-132 in one pass."""
    return _ContextualRows.apply(cosm, h, eps)


# ------------------------------------------------------------------------------------------
# K22 the contextual loss without its [N, N] matrices   (ContextualLoss.py:121-133; contextual_fused_f16x3.hip)
# ------------------------------------------------------------------------------------------
CX_TILE = 128     # positions per workgroup / inner tile of K22: the planes are padded to whole tiles


def _cx_pad_positions(x: torch.Tensor) -> torch.Tensor:
    n = x.shape[2]
    npad = (n + CX_TILE - 1) // CX_TILE * CX_TILE
    return x if npad == n else torch.nn.functional.pad(x, (0, npad - n))


class _ContextualCx(torch.autograd.Function):
    """cx[b,i] = max_j A_ij of the contextual affinity of two sets of normalised features xn [B,C,Nq], yn [B,C,Nk]:
    1 / sum_j exp((cos_ij - m_i) tau_i).  Forward: one launch of K22 (two sweeps over the keys).  Backward: one launch per side
    (the cosine tiles are recomputed) + the argmax column's term as a gather (d xn) / scatter-add (d yn).  Nothing [Nq, Nk]
    exists at any point; saved for the backward: the operand planes (the size of the features) and four numbers per query."""

    @staticmethod
    def forward(ctx, xn, yn, h: float, eps: float):
        xn, yn = _chk(xn, "contextual_cx: xn"), _chk(yn, "contextual_cx: yn")
        B, C, Nq = xn.shape
        if yn.dim() != 3 or yn.shape[:2] != (B, C):
            raise ValueError(f"contextual_cx: shape mismatch xn{tuple(xn.shape)} yn{tuple(yn.shape)}")
        Nk = yn.shape[2]
        kp = (C + 31) // 32 * 32
        xp, yp = _cx_pad_positions(xn), _cx_pad_positions(yn)
        xa, ya = absmax(xn), absmax(yn)
        xh, xl, xs = split_f16(xp, True, cpad=kp, amax=xa)          # [B, Nqp, Kp]
        yh, yl, ys = split_f16(yp, True, cpad=kp, amax=ya)
        f32 = dict(device=xn.device, dtype=torch.float32)
        m, S, U = (torch.empty((B, Nq), **f32) for _ in range(3))
        jstar = torch.empty((B, Nq), device=xn.device, dtype=torch.int32)
        _call("contextual_cx_fwd", "cocos_contextual_cx_fwd_f16x3", xh.data_ptr(), xl.data_ptr(), yh.data_ptr(), yl.data_ptr(),
              xs.data_ptr(), ys.data_ptr(), m.data_ptr(), S.data_ptr(), U.data_ptr(), jstar.data_ptr(), B, Nq, Nk, xp.shape[2],
              yp.shape[2], kp, float(h), float(eps), _stream())
        ctx.save_for_backward(xn, yn, m, S, U, jstar)
        ctx.planes = (xh, xl, xs, xa, yh, yl, ys, ya)
        ctx.cfg = (float(h), float(eps))
        return 1.0 / S

    @staticmethod
    def backward(ctx, dcx):
        xn, yn, m, S, U, jstar = ctx.saved_tensors
        xh, xl, xs, xa, yh, yl, ys, ya = ctx.planes
        h, eps = ctx.cfg
        dcx = _chk(dcx, "contextual_cx: dcx")
        B, C, Nq = xn.shape
        Nk = yn.shape[2]
        kp, nqp, nkp = xh.shape[2], xh.shape[1], yh.shape[1]
        need_x, need_y = ctx.needs_input_grad[:2]
        # per query: the coefficient of e_ij, the temperature and the argmax column's extra term — one small launch
        f32 = dict(device=xn.device, dtype=torch.float32)
        a, t2, extra = (torch.empty((B, Nq), **f32) for _ in range(3))
        a_amax = _zero_cell(xn.device)
        st = _stream()
        _call("contextual_cx_coeffs", "cocos_contextual_cx_coeffs", dcx.data_ptr(), S.data_ptr(), U.data_ptr(), m.data_ptr(), a.data_ptr(),
              t2.data_ptr(), extra.data_ptr(), a_amax.data_ptr(), B * Nq, h, eps, st)
        dx = dy = None
        if need_x:     # rows = queries, inner = keys, values = yn (channel-major), (m, t) and beta = a per row; the argmax column's
            vh, vl, vs = split_f16(_cx_pad_positions(yn), False, amax=ya)          # term is gathered from yn in the epilogue
            dx = torch.empty_like(xn)
            _call("contextual_cx_bwd", "cocos_contextual_cx_bwd_f16x3", xh.data_ptr(), xl.data_ptr(), yh.data_ptr(), yl.data_ptr(),
                  vh.data_ptr(), vl.data_ptr(), xs.data_ptr(), ys.data_ptr(), vs.data_ptr(), None, m.data_ptr(), t2.data_ptr(), None, None,
                  a.data_ptr(), yn.data_ptr(), jstar.data_ptr(), extra.data_ptr(), dx.data_ptr(), B, Nq, Nk, nqp, nkp, kp, C, 1, 1.0, st)
        if need_y:     # rows = keys, inner = queries, values = xn, (m, t) and alpha = a / max|a| per inner position
            vh, vl, vs = split_f16(_cx_pad_positions(xn), False, amax=xa)
            dy = torch.empty_like(yn)
            _call("contextual_cx_bwd", "cocos_contextual_cx_bwd_f16x3", yh.data_ptr(), yl.data_ptr(), xh.data_ptr(), xl.data_ptr(),
                  vh.data_ptr(), vl.data_ptr(), ys.data_ptr(), xs.data_ptr(), vs.data_ptr(), a_amax.data_ptr(), m.data_ptr(),
                  t2.data_ptr(), a.data_ptr(), a_amax.data_ptr(), None, None, None, None, dy.data_ptr(), B, Nk, Nq, nkp, nqp, kp, C, 0, 1.0, st)
            # the argmax column's term lands on key j*_i: a scatter-add over the queries (the exemplar side is detached by the
            # reference's caller, pix2pix_model.py:197-201: this branch is for completeness)
            dy.scatter_add_(2, jstar.long().unsqueeze(1).expand(B, C, Nq), extra.unsqueeze(1) * xn)
        return dx, dy, None, None


def contextual_cx(xn, yn, h: float = 0.1, eps: float = 1e-3):
    """cx [B,Nq] = max_j A_ij of the contextual affinity (ContextualLoss.py:121-132) of normalised features xn [B,C,Nq],
    yn [B,C,Nk] — any Nq, Nk, C; no [Nq, Nk] tensor in HBM, forward or backward (K22)."""
    return _ContextualCx.apply(xn, yn, float(h), float(eps))


def mfma_probe() -> torch.Tensor:
    """Debug: the 64x16 accumulator image of one v_mfma_f32_32x32x2_f32 (see api_common.hip)."""
    out = torch.empty((64, 16), device="cuda", dtype=torch.float32)
    _lib.call("cocos_debug_mfma_probe", out.data_ptr(), _stream())
    return out
