"""Golden-vector cases for the correspondence hot path (test infrastructure).

Shared by oracle/make_golden.py (which RUNS THE REFERENCE in the build container and stores its
outputs under tests/golden/) and by the tests (which regenerate the same seeded inputs and compare
the oracle / the HIP path with the stored reference outputs).  Inputs are never stored: they come
from `numpy.random.RandomState(seed)` (a frozen legacy generator, bit-stable across versions).

Every case drives NoVGGCorrespondence.forward (correspondence.py:222-374) with the feature
producers stubbed out, so `theta_raw` / `phi_raw` below are exactly what `self.theta(...)` /
`self.phi(...)` return at :272 / :282.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

# name -> dict(opt overrides for the reference flags, B, size (image H=W), nc (semantic_nc),
#              seg ('onehot' | 'float'), fwd (forward kwargs), grads (store d theta/phi),
#              sample (store only this many sampled output elements per tensor; 0 = everything))
CASES = {
    # README ADE20k flag set, match_kernel 1 = the north-star 64x64x256 shape in miniature
    "ade_mk1": dict(opt=dict(match_kernel=1), B=2, size=32, nc=5, seg="onehot", grads=True),
    # shipped default match_kernel=3 (K = 2304, zero-padded unfold, centring after unfold)
    "ade_mk3": dict(opt=dict(match_kernel=3), B=1, size=32, nc=5, seg="onehot", grads=True),
    # centring over HW instead of over channels (no --PONO_C)
    "noponoc_mk1": dict(opt=dict(match_kernel=1, PONO_C=False), B=1, size=32, nc=4, seg="onehot",
                        grads=True),
    "noponoc_mk3": dict(opt=dict(match_kernel=3, PONO_C=False), B=1, size=32, nc=4, seg="onehot"),
    # CelebA-HQ edge training flags: float label maps, bilinear upsample, warp-cycle + two-cycle
    "celeba_cycle": dict(opt=dict(match_kernel=1, warp_bilinear=True, warp_cycle_w=1.0,
                                  two_cycle=True, warp_mask_losstype="none", isTrain=True),
                         B=2, size=32, nc=15, seg="float", grads=True),
    # cycle-consistent mask loss (column softmax of the same logits)
    "mask_cycle": dict(opt=dict(match_kernel=1, warp_mask_losstype="cycle", isTrain=True),
                       B=1, size=32, nc=6, seg="onehot", grads=True),
    # B = 1 + show_corr (bilinear side output), inference flags
    "b1_showcorr": dict(opt=dict(match_kernel=1, show_corr=True), B=1, size=32, nc=3, seg="onehot"),
    # return_corr=True: the scaled correlation matrix itself
    "return_corr": dict(opt=dict(match_kernel=1), B=1, size=32, nc=3, seg="onehot",
                        fwd=dict(return_corr=True), grads=True),
    # WTA_scale custom autograd function (forward rule and the hard-coded 1e-4 backward)
    "wta_half": dict(opt=dict(match_kernel=1), B=1, size=32, nc=3, seg="onehot",
                     fwd=dict(WTA_scale_weight=0.5), grads=True),
    # temperature other than the default
    "temp_005": dict(opt=dict(match_kernel=1), B=1, size=32, nc=3, seg="onehot",
                     fwd=dict(temperature=0.05), grads=True),
    # --warp_stride 2: down = 2, 128-grid route of BASELINE config 5 in miniature (16x16 grid)
    "stride2": dict(opt=dict(match_kernel=1, warp_stride=2, warp_mask_losstype="none"), B=1,
                    size=32, nc=3, seg="float"),
    # ragged grid: 44x44 features (1936 positions — the size in the reference's own comments)
    "ragged_44": dict(opt=dict(match_kernel=1, warp_mask_losstype="none"), B=1, size=176, nc=3,
                      seg="float", sample=4096),
    # --warp_patch needs a 256x256 image (F.fold(y, 256, ...) at :321): 64x64 grid, HW = 4096
    "patch_256": dict(opt=dict(match_kernel=1, warp_patch=True, warp_cycle_w=1.0,
                               warp_mask_losstype="none", isTrain=True),
                      B=1, size=256, nc=3, seg="float", sample=4096),
    # ---- round 4: README flag sets that had no fixture ------------------------------------
    # CelebA-HQ mask-to-face TRAINING (README.md:97): `--warp_mask_losstype direct` together with `--warp_cycle_w 0.1`
    # and `--warp_bilinear`, on the shipped default match_kernel 3 and on match_kernel 1; label_nc = 19
    # (celebahq_dataset.py:25).  R1 carries [rgb | 19 labels], C1 the cycle term: both softmax directions, V of C1
    # differentiated, in ONE graph.
    "celebamask_train_mk3": dict(opt=dict(match_kernel=3, warp_bilinear=True, warp_cycle_w=0.1,
                                          warp_mask_losstype="direct", isTrain=True),
                                 B=2, size=32, nc=19, seg="onehot", grads=True, seed=2001),
    "celebamask_train_mk1": dict(opt=dict(match_kernel=1, warp_bilinear=True, warp_cycle_w=0.1,
                                          warp_mask_losstype="direct", isTrain=True),
                                 B=2, size=32, nc=19, seg="onehot", grads=True, seed=2002),
    # DeepFashion TRAINING (README.md:115): `--warp_patch` on the default match_kernel 3 (+ `--warp_bilinear`, no mask
    # loss); label_nc = 20 float maps (deepfashion_dataset.py:27).  256x256 image (F.fold(y, 256, ...) at :321) ->
    # 64x64 grid, V = 48 patch channels: the fused match_kernel-3 family at its real width.  Outputs AND gradients are
    # stored as seeded samples (the full gradients would be 8 MB).
    "fashion_patch_mk3": dict(opt=dict(match_kernel=3, warp_patch=True, warp_bilinear=True,
                                       warp_mask_losstype="none", isTrain=True),
                              B=1, size=256, nc=20, seg="float", sample=4096, grads=True, seed=2003),
}

# Seeds of the round-1 cases = 1000 + their rank in the sorted list of the round-1 names (frozen: adding a case must
# not move anyone else's seed); later cases carry an explicit `seed`.
_ROUND1_ORDER = ("ade_mk1", "ade_mk3", "b1_showcorr", "celeba_cycle", "mask_cycle", "noponoc_mk1", "noponoc_mk3",
                 "patch_256", "ragged_44", "return_corr", "stride2", "temp_005", "wta_half")


def case_seed(name: str) -> int:
    if "seed" in CASES[name]:
        return int(CASES[name]["seed"])
    return 1000 + _ROUND1_ORDER.index(name)


def make_inputs(name: str):
    """Seeded synthetic inputs of a case, float32, in the reference's tensor layouts."""
    c = CASES[name]
    rs = np.random.RandomState(case_seed(name))
    B, size, nc = c["B"], c["size"], c["nc"]
    down = 2 if c["opt"].get("warp_stride", 4) == 2 else 4
    fh = size // down
    theta_raw = rs.standard_normal((B, 256, fh, fh)).astype(np.float32)
    # exemplar features weakly correlated with the content features: the matching position gets a
    # logit of ~20 against a N(0, 6.25^2) background, so rows range from peaked to diffuse
    # (with phi = theta + small noise every row would be one-hot and the softmax untested)
    perm = rs.permutation(fh * fh)
    phi_raw = theta_raw.reshape(B, 256, -1)[:, :, perm].reshape(B, 256, fh, fh)
    phi_raw = (0.2 * phi_raw + rs.standard_normal(phi_raw.shape) + 0.1).astype(np.float32)
    ref_img = rs.uniform(-1, 1, (B, 3, size, size)).astype(np.float32)
    real_img = rs.uniform(-1, 1, (B, 3, size, size)).astype(np.float32)

    def seg():
        if c["seg"] == "float":
            return rs.uniform(0, 1, (B, nc, size, size)).astype(np.float32)
        blk = max(size // 8, 1)
        lab = rs.randint(0, nc, (B, size // blk, size // blk))
        lab = np.repeat(np.repeat(lab, blk, axis=1), blk, axis=2)
        return (lab[:, None] == np.arange(nc)[None, :, None, None]).astype(np.float32)

    seg_map, ref_seg_map = seg(), seg()
    return SimpleNamespace(theta_raw=theta_raw, phi_raw=phi_raw, ref_img=ref_img,
                           real_img=real_img, seg_map=seg_map, ref_seg_map=ref_seg_map,
                           down=down, fh=fh)


def grad_weights(name: str, out_shapes: dict):
    """Seeded cotangents G_k ~ N(0,1) for each output tensor; loss = sum_k <out_k, G_k>."""
    rs = np.random.RandomState(case_seed(name) + 7919)
    return {k: rs.standard_normal(out_shapes[k]).astype(np.float32) for k in sorted(out_shapes)}


def sample_index(name: str, key: str, numel: int, n: int):
    """Seeded flat indices used when only a sample of a big output tensor is stored."""
    rs = np.random.RandomState(case_seed(name) + 104729 + sum(map(ord, key)))
    return np.sort(rs.choice(numel, size=min(n, numel), replace=False))


def grad_error(name: str, key: str, got, golden) -> float:
    """max |got - ref| / max |ref| of a stored gradient (`key` = 'theta_raw' | 'phi_raw'); sampled cases store a
    seeded sample of the gradient (`gsample__*`) and compare that sample."""
    got = np.asarray(got, dtype=np.float64)
    if "grad__" + key in golden.files:
        ref = golden["grad__" + key].astype(np.float64)
        assert got.shape == ref.shape, (got.shape, ref.shape)
    else:
        idx = sample_index(name, "grad:" + key, got.size, CASES[name]["sample"])
        got, ref = got.reshape(-1)[idx], golden["gsample__" + key].astype(np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def hot_path_flags(name: str) -> dict:
    """The hot-path flags of a case in the form corr_oracle.default_opt / HotPathConfig take."""
    o = CASES[name]["opt"]
    return dict(match_kernel=o.get("match_kernel", 3), PONO_C=o.get("PONO_C", True),
                down=2 if o.get("warp_stride", 4) == 2 else 4,
                warp_patch=o.get("warp_patch", False), warp_bilinear=o.get("warp_bilinear", False),
                isTrain=o.get("isTrain", False), show_corr=o.get("show_corr", False),
                warp_mask_losstype=o.get("warp_mask_losstype", "direct"),
                show_warpmask=o.get("show_warpmask", False),
                warp_cycle_w=o.get("warp_cycle_w", 0.0), two_cycle=o.get("two_cycle", False))


def load_golden(name: str):
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests",
                        "golden", name + ".npz")
    return np.load(path)


def compare_with_golden(name: str, outputs: dict, golden) -> dict:
    """max |x - ref| / max |ref| per output tensor of a case (sampled cases compare the sample)."""
    errs = {}
    keys = [f.split("__", 1)[1] for f in golden.files if f.startswith("shape__")]
    for k in keys:
        if k not in outputs:
            errs[k] = float("inf")
            continue
        x = np.asarray(outputs[k], dtype=np.float64)
        if tuple(x.shape) != tuple(golden["shape__" + k]):
            errs[k] = float("inf")
            continue
        if "out__" + k in golden.files:
            ref = golden["out__" + k].astype(np.float64)
        else:
            idx = sample_index(name, k, x.size, CASES[name]["sample"])
            x, ref = x.reshape(-1)[idx], golden["sample__" + k].astype(np.float64)
        errs[k] = float(np.abs(x - ref).max() / (np.abs(ref).max() + 1e-30))
    for k in outputs:
        if k not in keys:
            errs["unexpected:" + k] = float("inf")
    return errs
