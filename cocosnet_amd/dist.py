"""Data-parallel exchange: one process per GPU, RCCL collectives over xGMI.

Replaces the reference's single-process `DataParallelWithCallback` (trainers/pix2pix_trainer.py
:23-29: replicate all parameters from GPU 0 on EVERY forward, gather outputs to GPU 0, reduce
gradients to GPU 0, optimiser on GPU 0 only) and the Python master/slave queue of its
Synchronized-BatchNorm (normalization.py:10,53,101,171).  Here every rank owns a full replica and its own
optimiser; the exchanges on the path are

  * ONE all-reduce of the gradients per optimiser step, issued bucket by bucket FROM autograd's
    post-accumulate hooks while backward is still running (`GradBuckets`): a bucket leaves as soon as its
    last gradient has been written.  Gradients live in flat fp32 buffers (`p.grad` is a view into its
    bucket), so nothing is copied in or out around the collective and the reduction is in place;
  * when SyncBN is live (the reference without `--PONO`): one small all-reduce of per-channel
    [sum x, sum x^2, count] in forward and one of [sum dy, sum dy*xhat] in backward (`SyncBatchNorm2d`).

Bucket sizing for MI355X: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring
all-reduce is bound by one link; RCCL reaches its plateau with messages of tens of MB.  netCorr
(59 M parameters, 237 MB) therefore goes out as a handful of 64 MiB buckets — large enough to be
bandwidth- rather than latency-bound, small enough that the first bucket can leave while autograd
is still producing the rest (buckets are filled in reverse parameter order = gradient-ready order).

backend "nccl" IS RCCL on ROCm; tests run the same code on "gloo" with world_size 2 on CPU.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist

DEFAULT_BUCKET_BYTES = 64 << 20


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).

    Returns (rank, local_rank, world_size); a no-op returning (0, 0, 1) when WORLD_SIZE <= 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 0, 1
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _live_world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class GradBuckets:
    """Flat fp32 gradient buckets over a fixed parameter list, all-reduced (averaged) in place and overlapped
    with backward.

        buckets = GradBuckets(model.parameters())      # p.grad becomes a view into its bucket
        for batch in data:
            buckets.zero_grad()                        # instead of p.grad = None
            loss(model(batch)).backward()              # full buckets leave from the hooks, during backward
            buckets.finish()                           # launch what is left, wait, average
            optimizer.step()

    Contract: ONE backward between zero_grad() and finish().  A second backward (gradient accumulation, two losses
    with retain_graph) would add into a buffer whose all-reduce has already left: its contribution would never be
    reduced and the ranks would silently diverge — the hook RAISES instead.  Accumulate explicitly:

            with buckets.accumulate():                 # hooks count nothing, nothing is launched
                loss_a.backward()
            loss_b.backward()                          # the last backward launches the buckets as usual

    A backward AFTER finish() and before the next zero_grad() (a GAN's generator step also writes the discriminator's
    gradients; its optimiser only ever zeroes them) is harmless and launches nothing.

    Stream ordering (backend "nccl" = RCCL).  A hook runs on the autograd thread with the stream of the forward op
    current; `dist.all_reduce(async_op=True)` makes RCCL's own stream wait for that stream at the launch point, so
    the collective sees the finished gradient.  `work.wait()` in finish() / zero_grad() does not block the host: it
    makes the CALLER's current stream wait for the collective, and the in-place division is enqueued behind it on
    that same stream — so finish() and the optimiser step that follows must run on the stream (or a stream ordered
    after the stream) backward ran on, which is the case for the reference's single-stream step functions.  After
    finish() returns no collective of this object is in flight (`in_flight() == 0`, tested on gloo).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = DEFAULT_BUCKET_BYTES,
                 overlap: bool = True, group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.overlap = bool(overlap)
        # reverse order: the last layers' gradients are ready first during backward
        order = list(reversed(self.params))
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        for p in order:
            if p.dtype != torch.float32:
                raise TypeError("GradBuckets: fp32 parameters expected (the reference trains in fp32)")
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        # flat storage; every p.grad is a view of it (autograd accumulates in place into an existing .grad)
        self._flat: List[torch.Tensor] = []
        self._bucket_of, self._offset_of = {}, {}
        for i, bucket in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=torch.float32, device=bucket[0].device)
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                self._bucket_of[p], self._offset_of[p] = i, off
                off += n
            self._flat.append(flat)
        self._pending = [len(b) for b in self.buckets]
        self._works: List = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._finished = False        # finish() has run since the last zero_grad(): later gradients are nobody's
        self._accumulating = False    # inside accumulate(): hooks neither count nor launch
        self._hooks = []
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))

    def nbytes(self) -> int:
        return sum(p.numel() * 4 for p in self.params)

    # -- per step ---------------------------------------------------------------------------------
    @torch.no_grad()
    def zero_grad(self) -> None:
        """One fill per bucket; re-attaches any p.grad that something replaced (e.g. `p.grad = None`)."""
        # a collective launched by a hook and never finish()ed (e.g. the discriminator's parameters receive gradients
        # during the GENERATOR step of a GAN; its optimiser only zeroes them) must be over before its buffer is reused
        for w in self._works:
            if w is not None:
                w.wait()
        for i, bucket in enumerate(self.buckets):
            flat = self._flat[i]
            flat.zero_()
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + off * 4:
                    p.grad = flat[off:off + n].view_as(p)
                off += n
        self._pending = [len(b) for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._finished = False

    def in_flight(self) -> int:
        """Collectives launched by this object that have not completed yet (0 after finish())."""
        return sum(1 for w in self._works if w is not None and not w.is_completed())

    def accumulate(self):
        """Context manager for every backward but the LAST of a step: gradients add up in the buckets, nothing is
        counted or launched (DistributedDataParallel's no_sync())."""
        outer = self

        class _Accumulate:
            def __enter__(self_inner):
                outer._accumulating = True
                return outer

            def __exit__(self_inner, *exc):
                outer._accumulating = False
                return False
        return _Accumulate()

    def _launch(self, i: int) -> None:
        if self._launched[i]:
            return
        self._launched[i] = True
        if _live_world(self.group) > 1:
            self._works[i] = dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _adopt(self, p: torch.nn.Parameter) -> None:
        """p.grad must be the view into its bucket; if someone set `p.grad = None` before backward, autograd installed
        a fresh tensor instead: move it in."""
        flat, off = self._flat[self._bucket_of[p]], self._offset_of[p]
        g = p.grad
        if g is not None and g.data_ptr() != flat.data_ptr() + off * 4:
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(g)
            p.grad = view

    def _on_grad_ready(self, p: torch.nn.Parameter) -> None:
        i = self._bucket_of[p]
        self._adopt(p)
        if self._accumulating or self._finished:
            return            # accumulate(): a later backward launches; after finish(): a gradient nobody steps with
        if self._launched[i]:
            raise RuntimeError(
                "GradBuckets: a second backward reached a gradient bucket whose all-reduce has already been launched "
                "(one backward per zero_grad()/finish() pair; wrap all but the last backward of a step in "
                "`with buckets.accumulate():`) - the new contribution would never be reduced")
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    @torch.no_grad()
    def finish(self, world_size: int | None = None) -> None:
        """Launch the buckets the hooks have not (parameters without a gradient this step, or overlap=False), wait
        for all of them and turn the sums into means.  grad <- mean over ranks."""
        world = world_size or _live_world(self.group)
        if self._finished:
            return                           # idempotent: the means are already in place
        self._finished = True
        if world <= 1:
            return
        for i in range(len(self.buckets)):
            if not self._launched[i]:        # not sent by a hook: adopt gradients that were installed behind our back
                for q in self.buckets[i]:
                    self._adopt(q)
            self._launch(i)
        for i, w in enumerate(self._works):
            if w is not None:
                w.wait()                     # nccl: the current stream waits for RCCL's; gloo: the host does
            self._flat[i].div_(world)        # ... and the division is ordered behind it on that stream

    # round-1 name: post-backward exchange in one call
    def all_reduce_(self, world_size: int | None = None, async_op: bool = True) -> None:
        self.finish(world_size)

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def shard_batch(global_batch: int, rank: int, world_size: int) -> tuple[int, int]:
    """[start, stop) of this rank's samples; mirrors base_options.py:197-199's divisibility rule."""
    if global_batch % world_size != 0:
        raise ValueError(f"batch size {global_batch} is not a multiple of {world_size} GPUs")
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


# ---------------------------------------------------------------------------------------------------
# Sync-BN statistics over the process group (the reference without --PONO: SynchronizedBatchNorm2d in
# SPADE's param_free_norm and in get_nonspade_norm_layer, normalization.py:53,101)
# ---------------------------------------------------------------------------------------------------
class _SyncBNFunction(torch.autograd.Function):
    """y = (x - mean_G) / sqrt(var_G + eps) * w + b with mean/var over the GLOBAL batch (all ranks): forward
    all-reduces per-channel [sum x, sum x^2, count], backward [sum dy, sum dy*xhat] — two small collectives per
    layer instead of the reference's Python master/slave queue + device-to-device copies."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        dims = [d for d in range(x.dim()) if d != 1]
        n_local = x.numel() // C
        stats = torch.empty(2 * C + 1, dtype=torch.float32, device=x.device)
        stats[:C] = x.sum(dims, dtype=torch.float32)
        stats[C:2 * C] = (x.float() * x.float()).sum(dims)
        stats[2 * C] = float(n_local)
        if _live_world(group) > 1:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        n = stats[2 * C]
        mean = stats[:C] / n
        var = (stats[C:2 * C] / n - mean * mean).clamp_min_(0.0)          # biased: what normalises
        invstd = torch.rsqrt(var + eps)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (x - mean.view(shape)) * invstd.view(shape)
        y = xhat
        if weight is not None:
            y = y * weight.view(shape)
        if bias is not None:
            y = y + bias.view(shape)
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.group, ctx.n = group, n
        ctx.mark_non_differentiable(mean, var, n)
        return y, mean, var, n

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dn):
        xhat, invstd, weight = ctx.saved_tensors
        C = xhat.shape[1]
        dims = [d for d in range(xhat.dim()) if d != 1]
        shape = [1, C] + [1] * (xhat.dim() - 2)
        red = torch.empty(2 * C, dtype=torch.float32, device=dy.device)
        red[:C] = dy.sum(dims)
        red[C:] = (dy * xhat).sum(dims)
        dw = red[C:].clone() if (weight is not None and ctx.needs_input_grad[1]) else None
        db = red[:C].clone() if ctx.needs_input_grad[2] else None
        if _live_world(ctx.group) > 1:
            dist.all_reduce(red, op=dist.ReduceOp.SUM, group=ctx.group)
        g = dy if weight is None else dy * weight.view(shape)
        mean_dy = (red[:C] / ctx.n) * (1.0 if weight is None else weight)
        mean_dyx = (red[C:] / ctx.n) * (1.0 if weight is None else weight)
        dx = (g - mean_dy.view(shape) - xhat * mean_dyx.view(shape)) * invstd.view(shape)
        return dx, dw, db, None, None


class SyncBatchNorm2d(torch.nn.modules.batchnorm._BatchNorm):
    """The reference's SynchronizedBatchNorm2d on torch.distributed: statistics over the global batch
    (RCCL all-reduce over xGMI under backend "nccl"), same parameters / buffers / momentum rule as
    nn.BatchNorm2d so the reference's checkpoints load unchanged (running_var is the UNBIASED estimate)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.group = group

    def _check_input_dim(self, x):
        if x.dim() < 2:
            raise ValueError(f"expected at least 2D input (got {x.dim()}D input)")

    def forward(self, x):
        self._check_input_dim(x)
        use_batch = self.training or not self.track_running_stats
        if not use_batch:
            return torch.nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                                                  False, 0.0, self.eps)
        y, mean, var, n = _SyncBNFunction.apply(x, self.weight, self.bias, self.eps, self.group)
        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (n / (n - 1).clamp_min(1.0))
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(unbiased, alpha=m)
        return y
