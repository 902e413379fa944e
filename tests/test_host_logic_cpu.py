"""CPU checks of host-side plumbing that carries no arithmetic of its own but decides which device buffers the
kernels see: the autograd node that splits the row pass's output, the pool of pre-zeroed max|x| cells and the table
that hands a producer's max|x| to the consumer of the same storage."""
import torch

from cocosnet_amd import hot_path, ops


def test_split_channels_forward_views_and_single_cat_backward():
    x = torch.randn(2, 7, 5, dtype=torch.float64, requires_grad=True)
    a, b = hot_path._split_channels(x, 3)
    assert a.shape == (2, 3, 5) and b.shape == (2, 4, 5)
    assert torch.equal(a, x[:, :3]) and torch.equal(b, x[:, 3:])
    ga, gb = torch.randn_like(a), torch.randn_like(b)
    (a * ga).sum().backward(retain_graph=True)              # only one output used: the other half is zeros
    assert torch.equal(x.grad[:, :3], ga) and torch.count_nonzero(x.grad[:, 3:]) == 0
    x.grad = None
    ((a * ga).sum() + (b * gb).sum()).backward()
    assert torch.equal(x.grad, torch.cat((ga, gb), 1))
    # same gradients as plain slicing
    y = x.detach().clone().requires_grad_(True)
    ((y[:, :3] * ga).sum() + (y[:, 3:] * gb).sum()).backward()
    assert torch.equal(x.grad, y.grad)


def test_zero_cells_are_distinct_zeroed_and_refilled(monkeypatch):
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    monkeypatch.setattr(ops._tls, "zero_pool", {})
    dev = torch.device("cpu")
    cells = [ops._zero_cell(dev) for _ in range(4096 + 3)]    # crosses one refill of the pool
    assert all(c.shape == (1,) and float(c) == 0.0 for c in cells)
    ptrs = {c.data_ptr() for c in cells}
    assert len(ptrs) == len(cells)
    cells[0].fill_(5.0)                                        # cells are views: writing one leaves the others alone
    assert float(cells[1]) == 0.0


def test_amax_table_is_keyed_by_storage_version_and_consumed_once(monkeypatch):
    monkeypatch.setattr(ops._tls, "known_amax", {})
    t = torch.randn(2, 3, 4)
    cell = torch.tensor([1.5])
    ops._remember_amax(t, cell)
    view = t.reshape(2, 12)                                    # what autograd hands to the next node
    assert ops._recall_amax(view, consume=False) is cell
    assert ops._recall_amax(t[:, 1:]) is None                  # a different window of the storage: not it
    ops._remember_amax(t, cell)
    assert ops._recall_amax(view) is cell and ops._recall_amax(view) is None     # picked up once
    ops._remember_amax(t, cell)
    t.add_(1.0)                                                # modified in place after the producer ran
    assert ops._recall_amax(t) is None
    # bounded: a younger entry beyond the table's size evicts the oldest (and releases the tensor it held)
    xs = [torch.randn(4) for _ in range(ops._KNOWN_AMAX_MAX + 1)]
    for x in xs:
        ops._remember_amax(x, cell)
    assert ops._recall_amax(xs[0]) is None and all(ops._recall_amax(x) is cell for x in xs[1:])
    assert len(ops._tls.known_amax) == 0
    # entries that only WATCH their tensor have their own, larger quota: a dozen holding entries in between do not evict them, a dead
    # tensor's entry is void, and the quota itself is bounded
    keep = torch.randn(4)
    ops._remember_amax(keep, cell, weak=True)
    for x in xs:
        ops._remember_amax(x, cell)
    assert ops._recall_amax(keep, consume=False) is cell
    gone = torch.randn(4)
    ops._remember_amax(gone, cell, weak=True)
    key_gone = (gone.device, gone.untyped_storage().data_ptr())
    del gone
    many = [torch.randn(4) for _ in range(ops._KNOWN_AMAX_WEAK_MAX + 5)]
    for x in many:
        ops._remember_amax(x, cell, weak=True)
    table = ops._tls.known_amax
    weak_entries = [e for e in table.values() if not torch.is_tensor(e[0])]
    assert len(weak_entries) <= ops._KNOWN_AMAX_WEAK_MAX and all(e[0]() is not None for e in weak_entries)
    assert ops._recall_amax(many[-1]) is cell and ops._recall_amax(many[0]) is None      # the oldest watchers went first
    assert sum(1 for e in table.values() if torch.is_tensor(e[0])) <= ops._KNOWN_AMAX_MAX


def test_host_scratch_is_per_thread():
    """SURVEY §8b threading contract: the reference's DataParallelWithCallback runs every replica's forward in its own
    Python thread and autograd runs every device's backward in its own thread — the max|x| table and the zero-cell
    pool must not be shared between them (round-1 ADVICE: KeyError / StopIteration under concurrent backward)."""
    import threading
    t = torch.randn(8)
    cell = torch.tensor([2.0])
    ops._remember_amax(t, cell)
    seen, errors = {}, []

    def worker(i):
        try:
            seen[i] = ops._recall_amax(t, consume=False)          # another thread's entry is invisible here
            mine = torch.randn(4)
            for _ in range(2000):                                 # hammer the table: no shared dict to corrupt
                ops._remember_amax(mine, cell)
                assert ops._recall_amax(mine) is cell
        except Exception as e:                                    # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    assert not errors, errors
    assert all(v is None for v in seen.values())
    assert ops._recall_amax(t) is cell                            # the registering thread still finds its own


def test_operand_planes_are_per_call_objects(monkeypatch):
    """The f16 operand planes of theta/phi live in an object the caller owns (one per forward call) — not in a
    process-global cache keyed on id(tensor)."""
    calls = []

    def fake_split(x, transpose, scale=1.0, cpad=None, amax=None):
        calls.append((id(x), transpose, scale))
        return torch.zeros(1), torch.zeros(1)

    monkeypatch.setattr(ops, "split_f16", fake_split)
    x = torch.randn(1, 4, 3)
    pl = ops.OperandPlanes()
    a = pl.get(x, True, 16.0)
    b = pl.get(x, True, 16.0)
    assert a[0] is b[0] and len(calls) == 1                       # second request: same planes
    pl.get(x, False, 16.0)
    assert len(calls) == 2 and len(pl) == 2                       # other layout: its own planes
    x.add_(1.0)
    pl.get(x, True, 16.0)
    assert len(calls) == 3                                        # modified in place: planes rebuilt
    assert len(ops.OperandPlanes()) == 0                          # a new call starts empty
    assert not hasattr(ops, "_split_cache")


def test_bench_roofline_and_kernel_table_contract():
    """bench.py's JSON pieces (no GPU): per-kernel table from fake HIP-event times, the roofline object of the dominant
    kernel with `frac`, `frac_issued`, `traffic` from the committed PMC file, and the bounded CPU baseline's fields."""
    import bench
    kern = {"corr_softmax_warp_fwd": {"avg_ms": 0.29, "calls": 20, "total_ms": 5.8},
            "corr_softmax_warp_bwd_query": {"avg_ms": 0.35, "calls": 20, "total_ms": 7.0},
            "corr_softmax_warp_bwd_key_from_ds": {"avg_ms": 0.17, "calls": 20, "total_ms": 3.4}}
    tab = bench.kernel_table(kern, "f16x3")
    assert set(tab) == set(kern)
    alg = 2.0 * 4096 * 4096 * (256 + 154) * 8 / 0.35 / 1e9
    assert abs(tab["corr_softmax_warp_bwd_query"]["alg_tflops"] - alg) < 0.5
    # the key GEMM always issues three terms; forward / query backward 70 of 78 MFMAs when the label blocks' lo plane is skipped
    assert abs(tab["corr_softmax_warp_bwd_key_from_ds"]["issued_tflops"] / tab["corr_softmax_warp_bwd_key_from_ds"]["alg_tflops"] - 3.0) < 0.01
    ratio = tab["corr_softmax_warp_fwd"]["issued_tflops"] / tab["corr_softmax_warp_fwd"]["alg_tflops"]
    assert abs(ratio - (70 / 26 if bench.ops_value_lo_skip() else 3.0)) < 0.01
    roof = bench.roofline_of(tab, "f16x3")
    assert roof["kernel"] == "corr_softmax_warp_bwd_query" and roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["frac_issued"] <= roof["frac"] + 1e-6
    assert roof["traffic"] is not None and roof["traffic"] > 1e8          # bytes per launch from the committed PMC file (profiles/r03_pmc_f16x3.json)
    if roof["traffic"]:                                              # the HBM coordinate of the same kernel
        assert roof["hbm"]["unit"] == "GB/s" and roof["hbm"]["peak"] == bench.HBM_PEAK_GBS
        assert abs(roof["hbm"]["achieved"] - roof["traffic"] / 0.35e-3 / 1e9) < 1.0
        assert abs(roof["hbm"]["frac"] - roof["hbm"]["achieved"] / 8000.0) < 1e-3
    fp = bench.roofline_of(bench.kernel_table(kern, "fp32"), "fp32")
    assert fp["peak"] == bench.FP32_MFMA_PEAK_TFLOPS


def test_bench_cpu_baseline_is_bounded_and_labelled():
    import bench
    cb = bench.cpu_baseline(runs=1, batch=1, budget_s=5.0)
    assert cb["unit"] == "images/s" and cb["kind"] == "port" and cb["value"] > 0 and 1 <= cb["cores"] <= 32
    assert "all_cores" in cb and cb["all_cores"]["cores"] >= cb["cores"] and "sample" in cb
