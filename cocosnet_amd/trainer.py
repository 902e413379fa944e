"""Trainer-level data parallelism: one process per GPU under the reference's own `Pix2PixTrainer`.

SURVEY.md §8(f) rank 2 / §8(e): the reference wraps `Pix2PixModel` in `DataParallelWithCallback`
(trainers/pix2pix_trainer.py:23-29): ONE process, parameters re-broadcast from GPU 0 on every forward, outputs gathered
and gradients reduced to GPU 0, optimiser on GPU 0.  Here every rank owns a replica and runs the unmodified step
functions (`run_generator_one_step` / `run_discriminator_one_step`, :52-74) on its own shard of the global batch; the
only exchange is the bucketed, hook-driven gradient all-reduce of `cocosnet_amd.dist.GradBuckets` (RCCL over xGMI under
backend "nccl"), attached to the two optimisers so that the step functions need no change:

    optimizer.zero_grad()  ->  one fill per flat gradient bucket (p.grad stays a view into its bucket)
    loss.backward()        ->  every full bucket leaves for the all-reduce from autograd's hooks, during backward
    optimizer.step()       ->  first waits for the buckets and turns sums into means, then the reference's Adam step

Sync-BN layers (the reference without --PONO) exchange their statistics through `cocosnet_amd.dist.SyncBatchNorm2d`.

Checkpoints: the replicas are identical, so `save()` writes on RANK 0 ONLY and every rank waits at a barrier — the
inherited `Pix2PixTrainer.save()` (:84-97 -> util.save_network, util/util.py:226-231: `net.cpu()`, `torch.save` to ONE
shared path, `net.cuda()`) run by every rank would race on the same files.  Data: `sampler(dataset)` returns the
`DistributedSampler` that gives each rank its own shard of every epoch (the reference's DataLoader has no notion of
ranks: data/__init__.py builds a plain shuffled loader, every process would draw the SAME samples).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .dist import DEFAULT_BUCKET_BYTES, GradBuckets, init_from_env, shard_batch


def broadcast_parameters(module_or_params, src: int = 0, group=None) -> None:
    """Identical replicas: rank `src`'s parameters and buffers to everyone (once, at start-up)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return
    if isinstance(module_or_params, torch.nn.Module):
        tensors = list(module_or_params.parameters()) + list(module_or_params.buffers())
    else:
        tensors = list(module_or_params)
    with torch.no_grad():
        for t in tensors:
            dist.broadcast(t.data, src, group=group)


def attach_gradient_exchange(optimizer: torch.optim.Optimizer, bucket_bytes: int = DEFAULT_BUCKET_BYTES,
                             overlap: bool = True, group=None) -> GradBuckets:
    """Make `optimizer` data-parallel in place: its `zero_grad()` re-arms the gradient buckets and its `step()` first
    completes the all-reduce (mean over ranks).  Returns the GradBuckets (also kept as `optimizer.grad_buckets`)."""
    params = [p for g in optimizer.param_groups for p in g["params"]]
    buckets = GradBuckets(params, bucket_bytes=bucket_bytes, overlap=overlap, group=group)
    inner_step = optimizer.step

    def zero_grad(set_to_none: bool = True):
        buckets.zero_grad()

    def step(closure=None):
        buckets.finish()
        return inner_step(closure) if closure is not None else inner_step()

    optimizer.zero_grad = zero_grad
    optimizer.step = step
    optimizer.grad_buckets = buckets
    return buckets


def make_distributed_trainer(trainer_cls):
    """`Pix2PixTrainer` -> a subclass for one-process-per-GPU runs (launch with torchrun: RANK / LOCAL_RANK / WORLD_SIZE).

    `opt.gpu_ids` becomes `[LOCAL_RANK]` (the reference then skips `DataParallelWithCallback`, :23-29), `opt.batchSize`
    is the PER-RANK batch (global batch = batchSize * world, as with the reference's per-GPU split, base_options.py
    :197-199), parameters are broadcast from rank 0 and both optimisers exchange gradients as described above."""

    class DistributedTrainer(trainer_cls):
        def __init__(self, opt, *args, backend=None, bucket_bytes=DEFAULT_BUCKET_BYTES, **kwargs):
            self.rank, self.local_rank, self.world_size = init_from_env(backend)
            if torch.cuda.is_available() and getattr(opt, "gpu_ids", None) not in (None, [], [-1]):
                opt.gpu_ids = [self.local_rank]
            super().__init__(opt, *args, **kwargs)
            model = getattr(self, "pix2pix_model_on_one_gpu", None)
            if isinstance(model, torch.nn.Module):
                broadcast_parameters(model)
            self.grad_buckets = {}
            self._bucket_bytes = bucket_bytes
            for name in ("optimizer_G", "optimizer_D"):
                optim = getattr(self, name, None)
                if optim is not None:
                    self.grad_buckets[name] = attach_gradient_exchange(optim, bucket_bytes=bucket_bytes)

        def shard(self, global_batch: int):
            """[start, stop) of this rank's samples of a global batch (for data loaders that index by sample)."""
            return shard_batch(global_batch, self.rank, self.world_size)

        def sampler(self, dataset, shuffle: bool = True, seed: int = 0, drop_last: bool = True):
            """The sampler to hand to `torch.utils.data.DataLoader(dataset, batch_size=opt.batchSize, sampler=...)`
            in place of the reference's `shuffle=` (data/__init__.py): disjoint shards of one common permutation per
            epoch (call `.set_epoch(epoch)` at the top of every epoch, as with DistributedDataParallel)."""
            from torch.utils.data.distributed import DistributedSampler
            return DistributedSampler(dataset, num_replicas=self.world_size, rank=self.rank, shuffle=shuffle,
                                      seed=seed, drop_last=drop_last)

        def save(self, epoch):
            """pix2pix_trainer.py:84-97 on rank 0 only (identical replicas; one writer per file), then a barrier so
            that no rank races ahead into a `continue_train` load or the next save.  The other ranks skip the
            `.cpu()` / `.cuda()` round trip of util.save_network altogether."""
            try:
                if self.rank == 0:
                    super().save(epoch)
            finally:
                if self.world_size > 1 and dist.is_initialized():
                    dist.barrier()
            # util.save_network moved rank 0's parameters (and their .grad) to the host and back: the gradient views
            # into the flat buckets are re-attached by the next zero_grad() (GradBuckets.zero_grad checks every view)

        def update_fixed_params(self):
            """pix2pix_trainer.py:125-139 REPLACES optimizer_G (netCorr unfrozen): move the gradient exchange to the
            new optimiser, or it would step on unreduced gradients."""
            old = self.grad_buckets.pop("optimizer_G", None)
            if old is not None:
                old.finish()
                old.close()
            super().update_fixed_params()
            self.grad_buckets["optimizer_G"] = attach_gradient_exchange(self.optimizer_G, bucket_bytes=self._bucket_bytes)

    DistributedTrainer.__name__ = "Distributed" + trainer_cls.__name__
    return DistributedTrainer
