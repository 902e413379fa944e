"""the REFERENCE's own code timed on CPU (build container only — /root/reference is not on the GPU box).

Test infrastructure.  Imports the unmodified reference through oracle/ref_harness.py and times, in fp32 with all host
threads, synthetic inputs of BASELINE config 1 (ADE20k flags, 256x256, batchSize 2):
  (ii)  netCorr = NoVGGCorrespondence: forward (inference) and forward + backward;
  (iii) the hot path alone, through the reference's own lines (correspondence.py:272-336 restated in
        oracle/torch_ref.py, which tests/test_oracle_golden.py pins to the reference's outputs).
  (i)   the facade Pix2PixModel(data, mode='inference') (netCorr + netG) with the README's ADE20k test flags; the
        literal `test.py --gpu_ids -1` is not runnable (torchvision absent, label_ref dtype defect
        pix2pix_model.py:172-187), so the options come from the reference's own parser defaults and the inputs are
        built here with an int64 label_ref.
Output: one JSON document (committed as profiles/r02_reference_cpu_baseline.json).
    python -m oracle.time_reference_cpu [--runs 10] [--warmup 3] [--batch 2]
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh   # noqa: E402


def timed(fn, runs, warmup):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return {"median_s": statistics.median(ts), "min_s": min(ts), "runs": runs}


def inputs(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    blocks = torch.randint(0, 151, (B, 1, 16, 16), generator=g)
    label = blocks.repeat_interleave(16, 2).repeat_interleave(16, 3)                  # 16x16 constant blocks
    seg = torch.zeros(B, 151, 256, 256).scatter_(1, label, 1.0)
    ref_blocks = torch.randint(0, 151, (B, 1, 16, 16), generator=g)
    ref_label = ref_blocks.repeat_interleave(16, 2).repeat_interleave(16, 3)
    ref_seg = torch.zeros(B, 151, 256, 256).scatter_(1, ref_label, 1.0)
    img = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
    ref_img = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
    return ref_img, img, seg, ref_seg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    B = args.batch
    out = {"what": "reference code on CPU (BASELINE.md section 3), fp32, synthetic inputs, random-init weights",
           "threads": torch.get_num_threads(), "torch": torch.__version__, "batch": B, "results": {}}
    ref_img, img, seg, ref_seg = inputs(B)
    for mk in (1, 3):
        opt = rh.make_opt(match_kernel=mk, isTrain=False)
        net = rh.build_reference_corr(opt).eval()
        def fwd():
            with torch.no_grad():
                return net(ref_img, img, seg, ref_seg)
        r = timed(fwd, args.runs, args.warmup)
        r["images_per_s"] = B / r["median_s"]
        out["results"][f"netCorr_forward_mk{mk}"] = r
        opt_t = rh.make_opt(match_kernel=mk, isTrain=True)
        net_t = rh.build_reference_corr(opt_t).train()
        def fwdbwd():
            for p in net_t.parameters():
                p.grad = None
            o = net_t(ref_img, img, seg, ref_seg)
            (o["warp_out"].sum() + o["warp_mask"].sum()).backward()
        r = timed(fwdbwd, max(3, args.runs // 2), 1)
        r["images_per_s"] = B / r["median_s"]
        out["results"][f"netCorr_forward_backward_mk{mk}"] = r
        print(json.dumps({k: v for k, v in out["results"].items() if f"mk{mk}" in k}), file=sys.stderr, flush=True)
    # (iii) hot path only, the reference's op sequence (bench.py's cpu_baseline leg: theta/phi 1x1 convs, centre + L2norm,
    # correlation, softmax, warp of [rgb | 151 labels], x4 up-sampling, and autograd backward)
    import bench
    cb = bench.cpu_baseline(runs=args.runs, batch=B)
    out["results"]["hot_path_forward_backward_mk1"] = cb
    # (i) the facade: Pix2PixModel(data, mode='inference') with the README's ADE20k test flags, B images, int64 label_ref
    try:
        import contextlib
        import importlib
        networks = rh.load_reference()
        with rh._cwd(rh.REFERENCE_ROOT):
            to = importlib.import_module("options.test_options")
            parser = to.TestOptions().initialize(argparse.ArgumentParser())
            parser = networks.modify_commandline_options(parser, False)
            opt, _ = parser.parse_known_args([])
            for k, v in dict(name="ade20k", dataset_mode="ade20k", gpu_ids=[], use_attention=True, maskmix=True,
                             warp_mask_losstype="direct", PONO=True, PONO_C=True, batchSize=B, isTrain=False,
                             semantic_nc=151, label_nc=150, contain_dontcare_label=True, no_instance=True, crop_size=256,
                             load_size=256, aspect_ratio=1.0, adaptor_nonlocal=True).items():
                setattr(opt, k, v)
            p2p = importlib.import_module("models.pix2pix_model")
            g = torch.Generator().manual_seed(0)
            blk = lambda: torch.randint(0, 151, (B, 1, 16, 16), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3)
            data = {"label": blk().long(), "label_ref": blk().long(), "image": torch.rand(B, 3, 256, 256, generator=g) * 2 - 1,
                    "ref": torch.rand(B, 3, 256, 256, generator=g) * 2 - 1, "self_ref": torch.zeros(B), "path": ["x"] * B}
            for mk in (1, 3):
                opt.match_kernel = mk
                torch.manual_seed(0)
                with contextlib.redirect_stdout(None):
                    model = p2p.Pix2PixModel(opt).eval()

                def run():
                    with torch.no_grad():
                        return model(data, mode="inference")
                r = timed(run, max(3, args.runs // 2), 1)
                r["images_per_s"] = B / r["median_s"]
                out["results"][f"pix2pix_inference_mk{mk}"] = r
    except Exception as e:       # noqa: BLE001
        out["results"]["pix2pix_inference"] = {"skipped": repr(e)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
