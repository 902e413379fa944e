"""world_size-2 `gloo` tests of the data-parallel exchange (cocosnet_amd/dist.py) on CPU.
The same code runs over RCCL (`nccl`) on the GPU box; here it proves the bucketing / hook-driven overlap /
averaging / sharding logic and the Sync-BN statistics exchange: after finish() every rank holds the mean of the
per-rank gradients, which equals the gradient of the mean loss over the GLOBAL batch; SyncBatchNorm2d over two ranks
equals nn.BatchNorm2d over the concatenated batch (output, input gradient, running statistics)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(5, 7, 1), torch.nn.PReLU(), torch.nn.Conv2d(7, 3, 3, padding=1))


def _data(global_batch=4):
    g = torch.Generator().manual_seed(1)
    return torch.randn(global_batch, 5, 6, 6, generator=g), torch.randn(global_batch, 3, 6, 6, generator=g)


def _worker(rank, world, port, bucket_bytes, overlap, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cocosnet_amd import dist as cdist
    r, lr, w = cdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    x, y = _data()
    lo, hi = cdist.shard_batch(x.shape[0], rank, world)
    buckets = cdist.GradBuckets(model.parameters(), bucket_bytes=bucket_bytes, overlap=overlap)
    launched_during_backward = []
    for step in range(2):                       # second step: buffers are reused, zero_grad() re-arms the hooks
        buckets.zero_grad()
        if step == 1:
            list(model.parameters())[0].grad = None            # a caller that still does `p.grad = None`
        loss = ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean()
        loss.backward()
        launched_during_backward.append(sum(buckets._launched))
        buckets.finish()
    ret[rank] = ([p.grad.clone() for p in model.parameters()], launched_during_backward,
                 all(p.grad.data_ptr() >= buckets._flat[buckets._bucket_of[p]].data_ptr() for p in model.parameters()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,overlap", [(64, True), (1 << 20, True), (64, False)])
def test_gradient_allreduce_equals_global_batch_gradient(bucket_bytes, overlap):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, bucket_bytes, overlap, ret), nprocs=world, join=True)
    model = _model()
    x, y = _data()
    ((model(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in model.parameters()]
    n_buckets = None
    for rank in range(world):
        grads, launched, views = ret[rank]
        for a, b in zip(grads, ref):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)
        assert views                                            # gradients live in the flat buckets (no copy in/out)
        if overlap:                                             # every bucket left from a hook, i.e. DURING backward
            assert launched[0] > 0 and launched[0] == launched[1]
            n_buckets = launched[0]
        else:
            assert launched == [0, 0]
    if overlap and bucket_bytes == 64:
        assert n_buckets > 1


def test_bucket_layout_and_sharding():
    from cocosnet_amd import dist as cdist
    model = _model()
    b = cdist.GradBuckets(model.parameters(), bucket_bytes=64)
    assert sum(len(x) for x in b.buckets) == len(list(model.parameters()))
    assert b.buckets[0][0] is list(model.parameters())[-1]          # reverse (gradient-ready) order
    assert b.nbytes() == sum(p.numel() for p in model.parameters()) * 4
    # p.grad is a view into its bucket; single-process finish() is a no-op that leaves the sums alone
    x, y = _data()
    b.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    flat_sum = sum(float(f.abs().sum()) for f in b._flat)
    assert flat_sum > 0 and abs(flat_sum - sum(float(p.grad.abs().sum()) for p in model.parameters())) < 1e-4
    b.finish()
    assert cdist.shard_batch(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        cdist.shard_batch(10, 0, 4)
    assert cdist.init_from_env() == (0, 0, 1) or os.environ.get("WORLD_SIZE", "1") != "1"


# ---------------------------------------------------------------------------------- Sync-BN statistics
def _bn_data():
    g = torch.Generator().manual_seed(2)
    return torch.randn(6, 4, 5, 3, generator=g) * 2 + 0.5, torch.randn(6, 4, 5, 3, generator=g)


def _bn_worker(rank, world, port, affine, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cocosnet_amd import dist as cdist
    cdist.init_from_env("gloo")
    torch.manual_seed(3)
    bn = cdist.SyncBatchNorm2d(4, affine=affine)
    if affine:
        with torch.no_grad():
            bn.weight.copy_(torch.tensor([1.5, 0.5, -1.0, 2.0])); bn.bias.copy_(torch.tensor([0.1, -0.2, 0.3, 0.0]))
    x, g = _bn_data()
    lo, hi = (0, 2) if rank == 0 else (2, 6)            # UNEVEN shards: the count is part of the exchange
    xs = x[lo:hi].clone().requires_grad_(True)
    y = bn(xs)
    y.backward(g[lo:hi])
    wg = bn.weight.grad.clone() if affine else None
    bn.eval()
    ye = bn(x[:1])
    ret[rank] = (y.detach(), xs.grad, wg, bn.running_mean.clone(), bn.running_var.clone(), ye.detach(),
                 int(bn.num_batches_tracked))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("affine", [False, True])
def test_sync_batchnorm_equals_batchnorm_over_the_global_batch(affine):
    """The reference without --PONO normalises SPADE with SynchronizedBatchNorm2d (normalization.py:101): statistics
    over the GLOBAL batch.  Two ranks with 2 + 4 samples == nn.BatchNorm2d on all 6."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bn_worker, args=(world, port, affine, ret), nprocs=world, join=True)
    x, g = _bn_data()
    ref = torch.nn.BatchNorm2d(4, affine=affine)
    if affine:
        with torch.no_grad():
            ref.weight.copy_(torch.tensor([1.5, 0.5, -1.0, 2.0])); ref.bias.copy_(torch.tensor([0.1, -0.2, 0.3, 0.0]))
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g)
    ref.eval()
    for rank, (lo, hi) in enumerate(((0, 2), (2, 6))):
        y, dx, wg, rm, rv, ye, nb = ret[rank]
        assert torch.allclose(y, yr[lo:hi].detach(), atol=1e-5)
        assert torch.allclose(dx, xr.grad[lo:hi], atol=1e-5)
        assert torch.allclose(rm, ref.running_mean, atol=1e-6) and torch.allclose(rv, ref.running_var, atol=1e-5)
        assert torch.allclose(ye, ref(x[:1]).detach(), atol=1e-5) and nb == 1
    if affine:      # local sums; the gradient all-reduce (GradBuckets) adds them up
        assert torch.allclose(ret[0][2] + ret[1][2], ref.weight.grad, atol=1e-4)


def test_sync_batchnorm_single_process_and_state_dict_names():
    from cocosnet_amd import dist as cdist
    bn, ref = cdist.SyncBatchNorm2d(3), torch.nn.BatchNorm2d(3)
    assert set(bn.state_dict()) == set(ref.state_dict())           # reference checkpoints load unchanged
    x = torch.randn(4, 3, 2, 2)
    assert torch.allclose(bn(x), ref(x), atol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, atol=1e-6)


# ---------------------------------------------------------------------------------- trainer-level driver
class _ToyTrainer:
    """The shape of the reference's Pix2PixTrainer (trainers/pix2pix_trainer.py:20-74): a model, two optimisers, a
    generator step and a discriminator step that keeps the generator's output across the D step."""

    def __init__(self, opt):
        torch.manual_seed(opt.seed)
        self.G = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 3, padding=1), torch.nn.PReLU(), torch.nn.Conv2d(6, 3, 1))
        self.D = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.LeakyReLU(0.2), torch.nn.Conv2d(4, 1, 1))
        self.pix2pix_model_on_one_gpu = torch.nn.ModuleDict({"G": self.G, "D": self.D})
        self.optimizer_G = torch.optim.Adam(self.G.parameters(), lr=1e-2, betas=(0.0, 0.9), eps=1e-3)
        self.optimizer_D = torch.optim.Adam(self.D.parameters(), lr=2e-2, betas=(0.0, 0.9), eps=1e-3)
        self.opt_dir = getattr(opt, "checkpoints_dir", None)

    def run_generator_one_step(self, x, y):
        self.optimizer_G.zero_grad()
        self.out = self.G(x)
        g_loss = (self.out - y).abs().mean() - self.D(self.out).mean()
        g_loss.backward()
        self.optimizer_G.step()

    def run_discriminator_one_step(self, x, y):
        self.optimizer_D.zero_grad()
        d_loss = torch.relu(1 + self.D(self.out.detach())).mean() + torch.relu(1 - self.D(y)).mean()
        d_loss.backward()
        self.optimizer_D.step()

    def save(self, epoch):
        """The reference's save path in miniature (pix2pix_trainer.py:84-97 -> util/util.py:226-231): every caller
        writes the SAME file names."""
        os.makedirs(self.opt_dir, exist_ok=True)
        with open(os.path.join(self.opt_dir, "writers.log"), "a") as f:
            f.write("save\n")
        torch.save(self.pix2pix_model_on_one_gpu.state_dict(), os.path.join(self.opt_dir, f"{epoch}_net.pth"))
        torch.save({"G": self.optimizer_G.state_dict(), "D": self.optimizer_D.state_dict()},
                   os.path.join(self.opt_dir, "optimizer.pth"))

    def update_fixed_params(self):
        """pix2pix_trainer.py:125-139: a NEW optimizer_G object over the same parameters."""
        self.optimizer_G = torch.optim.Adam(self.G.parameters(), lr=1e-2, betas=(0.0, 0.9), eps=1e-3)


def _trainer_worker(rank, world, port, ret, ckpt_dir=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from types import SimpleNamespace
    from cocosnet_amd import trainer as ctr
    Dist = ctr.make_distributed_trainer(_ToyTrainer)
    tr_ = Dist(SimpleNamespace(seed=100 + rank, gpu_ids=[], checkpoints_dir=ckpt_dir), backend="gloo",
               bucket_bytes=64)                          # different seeds per rank: the broadcast must fix it
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(3, 4, 3, 5, 5, generator=g), torch.randn(3, 4, 3, 5, 5, generator=g)
    in_flight = []
    for it in range(3):
        lo, hi = tr_.shard(4)
        tr_.run_generator_one_step(X[it, lo:hi], Y[it, lo:hi])
        # after optimizer_G.step(): G's collectives are over.  (D's parameters also received gradients during this
        # backward; from the second iteration on their buckets are finished and launch nothing.)
        in_flight.append(tr_.grad_buckets["optimizer_G"].in_flight())
        tr_.run_discriminator_one_step(X[it, lo:hi], Y[it, lo:hi])
        in_flight.append(tr_.grad_buckets["optimizer_D"].in_flight() + tr_.grad_buckets["optimizer_G"].in_flight())
        if it == 1 and ckpt_dir is not None:
            tr_.save("latest")                           # EVERY rank calls it, as the reference's train.py would
            tr_.update_fixed_params()                    # optimizer_G is replaced: the exchange must follow it
    ret[rank] = ([p.detach().clone() for p in tr_.pix2pix_model_on_one_gpu.parameters()], in_flight,
                 hasattr(tr_.optimizer_G, "grad_buckets"))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_trainer_equals_single_process_on_the_global_batch():
    """Two ranks x batch 2 through the wrapped trainer == one process x batch 4 through the plain one (three G + D
    steps with Adam): identical replicas after the start-up broadcast, gradient means == global-batch gradients, the
    step functions themselves unchanged."""
    from types import SimpleNamespace
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, port, ret), nprocs=world, join=True)
    ref = _ToyTrainer(SimpleNamespace(seed=100))          # rank 0's initialisation
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(3, 4, 3, 5, 5, generator=g), torch.randn(3, 4, 3, 5, 5, generator=g)
    for it in range(3):
        ref.run_generator_one_step(X[it], Y[it])
        ref.run_discriminator_one_step(X[it], Y[it])
    want = [p.detach() for p in ref.pix2pix_model_on_one_gpu.parameters()]
    for rank in range(world):
        params, in_flight, _ = ret[rank]
        for a, b in zip(params, want):
            assert torch.allclose(a, b, atol=2e-5, rtol=1e-4)
        assert in_flight == [0] * 6                     # no collective left in flight across a G / D step boundary


def test_distributed_trainer_save_writes_once_and_follows_a_replaced_optimizer(tmp_path):
    """the inherited save() run by every rank races on ONE set of files.  Both ranks call save();
    exactly one writer ran, the checkpoint is rank 0's state after two iterations, and after update_fixed_params()
    (a NEW optimizer_G, pix2pix_trainer.py:125-139) the third iteration still exchanges gradients: the replicas end
    identical and equal to the single-process run that replaces its optimiser at the same point."""
    from types import SimpleNamespace
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ckpt = str(tmp_path / "ckpt")
    mp.spawn(_trainer_worker, args=(world, port, ret, ckpt), nprocs=world, join=True)
    assert open(os.path.join(ckpt, "writers.log")).read().count("save") == 1
    ref = _ToyTrainer(SimpleNamespace(seed=100))
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(3, 4, 3, 5, 5, generator=g), torch.randn(3, 4, 3, 5, 5, generator=g)
    for it in range(3):
        ref.run_generator_one_step(X[it], Y[it])
        ref.run_discriminator_one_step(X[it], Y[it])
        if it == 1:
            saved = {k: v.clone() for k, v in ref.pix2pix_model_on_one_gpu.state_dict().items()}
            ref.update_fixed_params()
    on_disk = torch.load(os.path.join(ckpt, "latest_net.pth"))
    for k, v in saved.items():
        assert torch.allclose(on_disk[k], v, atol=2e-5, rtol=1e-4), k
    want = [p.detach() for p in ref.pix2pix_model_on_one_gpu.parameters()]
    for rank in range(world):
        params, in_flight, attached = ret[rank]
        assert attached and in_flight == [0] * 6
        for a, b in zip(params, want):
            assert torch.allclose(a, b, atol=2e-5, rtol=1e-4)


def _accumulate_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cocosnet_amd import dist as cdist
    cdist.init_from_env("gloo")
    model = _model()
    x, y = _data(8)
    lo, hi = cdist.shard_batch(8, rank, world)
    mid = (lo + hi) // 2
    buckets = cdist.GradBuckets(model.parameters(), bucket_bytes=64)
    buckets.zero_grad()
    with buckets.accumulate():                           # micro-batch 1: nothing may leave
        (((model(x[lo:mid]) - y[lo:mid]) ** 2).mean() / 2).backward()
        launched_inside = sum(buckets._launched)
    (((model(x[mid:hi]) - y[mid:hi]) ** 2).mean() / 2).backward()
    buckets.finish()
    buckets.finish()                                     # idempotent: must not divide twice
    grads = [p.grad.clone() for p in model.parameters()]
    # the contract violation: a second backward WITHOUT accumulate() after the buckets have left
    buckets.zero_grad()
    ((model(x[lo:mid]) - y[lo:mid]) ** 2).mean().backward()
    try:
        ((model(x[mid:hi]) - y[mid:hi]) ** 2).mean().backward()
        raised = False
    except RuntimeError as e:
        raised = "second backward" in str(e)
    buckets.finish()
    ret[rank] = (grads, launched_inside, raised, buckets.in_flight())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_is_explicit_and_a_second_backward_raises():
    """two backwards between zero_grad() and finish() used to add into a bucket whose all-reduce had
    already left.  accumulate() defers the launch (two micro-batches per rank == the global-batch gradient); without it
    the hook raises."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_accumulate_worker, args=(world, port, ret), nprocs=world, join=True)
    model = _model()
    x, y = _data(8)
    ((model(x) - y) ** 2).mean().backward()
    for rank in range(world):
        grads, launched_inside, raised, in_flight = ret[rank]
        assert launched_inside == 0 and raised and in_flight == 0
        for a, p in zip(grads, model.parameters()):
            assert torch.allclose(a, p.grad, atol=1e-6, rtol=1e-5)
