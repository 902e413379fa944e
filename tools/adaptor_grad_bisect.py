"""Which kernel loses accuracy in d adaptive_model_img.layer1.0.weight_orig (tests/test_gpu_conv.py e2e: 1.6e-3..4.5e-3 against
fp64 while the framework's fp32 arm has 8e-6)?  Runs the image adaptor alone against an fp64 copy with components switched."""
import copy
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from cocosnet_amd import correspondence as cc, ops, producers  # noqa: E402

opt = cc.base_options(semantic_nc=5, match_kernel=1, maskmix=True, PONO=True, PONO_C=True, adaptor_kernel=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
torch.manual_seed(0)
net = cc.NoVGGCorrespondence(opt).cuda()
net.init_weights(opt.init_type, opt.init_variance)
net.eval()
ad = net.adaptive_model_img
if len(sys.argv) > 2:          # second argument: the adaptor's LeakyReLU slope (1.0 = no kink: sign decisions cannot differ)
    ad.actvn.negative_slope = float(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(2)
img = torch.rand(2, 3, 64, 64, device="cuda", generator=g) * 2 - 1
ad64 = copy.deepcopy(ad).double()
names = ["layer1.0.weight_orig", "layer2.0.weight_orig", "layer3.0.weight_orig", "layer4.0.weight_orig", "layer5.0.weight_orig",
         "head_0.conv_0.weight_orig", "G_middle_1.conv_1.weight_orig"]
get = lambda m, n: dict(m.named_parameters())[n]


def grads(model, x, G):
    model.zero_grad()
    y = model(x, x)
    y.backward(G.to(y.dtype))
    return y.detach(), {n: get(model, n).grad.clone() for n in names}


y64, g64 = None, None
G = None
orig_cna = producers.AdaptiveFeatureGenerator._conv_norm_act


def framework_cna(self, layer, x, slope):
    y = layer(x)
    return y if slope == 1.0 else torch.nn.functional.leaky_relu(y, slope)


for label, setup in [("default", {}), ("default_again", {}), ("spectral_fw", {"spectral": False}), ("k13_fw", {"k13": False}), ("nhwc_off", {"nhwc": False}),
                     ("convs_torch", {"conv": "torch"})]:
    producers.SPECTRAL_HIP = setup.get("spectral", True)
    ops.CONV_NHWC_F16X3 = setup.get("nhwc", True)
    ops.CONV_PRECISION = setup.get("conv", "f16x3")
    producers.AdaptiveFeatureGenerator._conv_norm_act = orig_cna if setup.get("k13", True) else framework_cna
    if G is None:
        with torch.no_grad():
            G = torch.randn(ad(img, img).shape, device="cuda", generator=g)
        y64, g64 = grads(ad64, img.double(), G)
    y, gr = grads(ad, img, G)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    print(json.dumps({"arm": label, "y": rel(y, y64), **{n: rel(gr[n], g64[n]) for n in names}}), flush=True)
