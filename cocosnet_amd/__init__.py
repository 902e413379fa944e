"""cocosnet_amd — MI355X-native (gfx950) implementation of CoCosNet's dense-correspondence hot path.

Scope: feature correlation -> softmax over HW -> attention-weighted warp of
models/networks/correspondence.py:271-372, as hand-written HIP kernels behind a C ABI
(include/cocos_hip.h), exposed through a drop-in `NoVGGCorrespondence` module.
"""
__version__ = "0.1.0"
