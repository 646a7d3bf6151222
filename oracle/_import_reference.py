"""Import the upstream reference (read-only, /root/reference) with stubbed third-party deps.

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py and oracle/check_oracle_vs_reference.py
in the build container, where /root/reference exists.  Nothing in the product path, the -m gpu
tests, smoke() or bench.py imports this file: the reference cannot travel to the GPU box.

The reference's models/LiteISP.py imports sibling modules that upstream never published
(.cbam, .AWISP_utils, .AWISP_modules) plus thop; groupmix.py imports ipdb and timm.  None of these
is used by the hot path, so raising placeholders are enough to make the pure-torch files import
(SURVEY.md section 8c).
"""
import sys
sys.dont_write_bytecode = True
import types
import importlib

import torch

REF_ROOT = "/root/reference"


def _raiser(name):
    class _Missing(torch.nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError(f"{name}: not published upstream (stub)")
    _Missing.__name__ = name
    return _Missing


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _DropPath(torch.nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        assert self.p == 0.0 or not self.training
        return x


def install_stubs():
    pkg = types.ModuleType("models")
    pkg.__path__ = [REF_ROOT + "/models"]
    sys.modules["models"] = pkg
    _mod("models.cbam", CBAM=_raiser("CBAM"))
    _mod("models.AWISP_utils", DWT=_raiser("DWT"), IWT=_raiser("IWT"))
    _mod("models.AWISP_modules", **{n: _raiser(n) for n in (
        "shortcutblock", "GCIWTResUp", "GCWTResDown", "GCRDB", "ContextBlock2d", "SE_net",
        "PSPModule", "last_upsample")})
    _mod("thop", profile=lambda *a, **k: (0, 0), clever_format=lambda x, f: x)
    _mod("ipdb")
    _mod("timm")
    _mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    _mod("timm.models")
    # CompressAI (PyPI `compressai`, no version pinned upstream) is absent: NAME-ONLY stubs so that models/tcm.py imports
    # and its own classes (WMSA, Block, SwinBlock ...) can run; anything that instantiates a CompressAI layer raises.
    _mod("compressai")
    _mod("compressai.entropy_models", EntropyBottleneck=_raiser("EntropyBottleneck"), GaussianConditional=_raiser("GaussianConditional"))
    _mod("compressai.ans", BufferedRansEncoder=_raiser("BufferedRansEncoder"), RansDecoder=_raiser("RansDecoder"))
    _mod("compressai.models", CompressionModel=_raiser("CompressionModel"))
    _mod("compressai.layers", **{n: _raiser(n) for n in ("AttentionBlock", "ResidualBlock", "ResidualBlockUpsample",
                                                          "ResidualBlockWithStride", "conv3x3", "subpel_conv3x3")})
    # models/raw2bit.py additionally imports torchvision and more of CompressAI at module level (datasets, zoo, google models,
    # deconv/conv helpers, GDN, MaskedConv2d): name-only stubs again -- none of them is on the path the oracle restates.
    _mod("torchvision", transforms=types.ModuleType("torchvision.transforms"), models=types.ModuleType("torchvision.models"))
    _mod("torchvision.transforms")
    _mod("torchvision.models")
    _mod("compressai.datasets", ImageFolder=_raiser("ImageFolder"), Vimeo90kDataset=_raiser("Vimeo90kDataset"))
    _mod("compressai.zoo", models={})
    _mod("compressai.models.google", FactorizedPrior=_raiser("FactorizedPrior"), ScaleHyperprior=_raiser("ScaleHyperprior"),
         MeanScaleHyperprior=_raiser("MeanScaleHyperprior"))
    _mod("compressai.models.utils", deconv=_raiser("deconv"), conv=_raiser("conv"))
    for _n in ("GDN", "MaskedConv2d"):
        setattr(sys.modules["compressai.layers"], _n, _raiser(_n))
    _mod("timm.models.layers", DropPath=_DropPath, to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
         trunc_normal_=torch.nn.init.trunc_normal_)


def load(*names):
    """Return the requested reference modules, e.g. load('networks', 'LiteISP')."""
    install_stubs()
    return [importlib.import_module("models." + n) for n in names]
