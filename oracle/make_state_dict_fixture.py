"""Record the parameter/buffer names and shapes of the REFERENCE's NoVGGCorrespondence for the
README flag sets -> tests/golden/netcorr_state_dict_keys.json (build container only).  The drop-in
module must reproduce them exactly so that `*_net_Corr.pth` checkpoints load."""
import json
import os

from . import ref_harness as rh

FLAG_SETS = {
    "ade20k": dict(semantic_nc=151, maskmix=True, PONO=True, PONO_C=True, use_attention=True),
    "celebahq_edge": dict(semantic_nc=15, maskmix=True, PONO=True, PONO_C=True, warp_bilinear=True,
                          adaptor_kernel=4),
    "deepfashion": dict(semantic_nc=20, maskmix=False, PONO=True, PONO_C=True, warp_bilinear=True,
                        warp_patch=True, adaptor_kernel=4),
    "no_pono_syncbn": dict(semantic_nc=12, maskmix=True, PONO=False, PONO_C=False),
    "all_adaptor_options": dict(semantic_nc=9, maskmix=True, PONO=True, PONO_C=True, adaptor_se=True,
                                adaptor_nonlocal=True, adaptor_res_deeper=True, dilation_conv=True,
                                use_coordconv=True, warp_stride=2),
}


def main():
    out = {}
    for name, flags in FLAG_SETS.items():
        net = rh.build_reference_corr(rh.make_opt(**flags))
        out[name] = {"flags": flags,
                     "state_dict": {k: list(v.shape) for k, v in net.state_dict().items()}}
        print(name, len(out[name]["state_dict"]), "entries")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "netcorr_state_dict_keys.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
