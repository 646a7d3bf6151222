import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    sd = {}
    for k in z.files:
        v = z[k]
        if k.startswith("sd16."):      # bf16-exact weights stored as their upper 16 bits (halves the fixture)
            sd[k[5:]] = torch.from_numpy((v.astype(np.uint32) << 16).view(np.float32).copy())
        elif k.startswith("sd."):
            sd[k[3:]] = torch.from_numpy(v)
        elif v.dtype.kind in "US":
            out[k] = str(v)
        else:
            out[k] = torch.from_numpy(v)
    out["sd"] = sd
    return out


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


NET_NAMES = ("LiteISPNet", "LiteISPNet_GFM_LSC", "LiteISPNet_LSC", "LiteISPNet_GFM", "LiteISPNet_GFMresize",
             "ISPUNet_GFM_LSC", "ISPUNet_GFM", "ISPUNet_GFM_LFM", "ISPUNet_LSC", "ResUNet",
             "ISPUNet_GFM_crop", "ISPUNet_GFM_LSC1", "ISPUNet_GFM_LSC_noskip")


def net_name_of(fixture: str) -> str:
    """e2e_<Net>_<size> -> the net's class name."""
    import re
    m = re.match(r"e2e_([A-Za-z_0-9]+?)_(?:randn_)?\d+x\d+$", fixture)
    if m is None or m.group(1) not in NET_NAMES:
        raise KeyError(fixture)
    return m.group(1)


def sd_digest(sd) -> str:
    import hashlib
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].detach().cpu().float().numpy()).tobytes())
    return h.hexdigest()


_SEED0 = {}


def seed0_state_dict(name):
    """state_dict of the build's mirror module constructed under torch.manual_seed(0) (CPU, fp32)."""
    if name not in _SEED0:
        import realcamnet_amd as M
        torch.manual_seed(0)
        _SEED0[name] = {k: v.clone() for k, v in getattr(M, name)().eval().state_dict().items()}
    return _SEED0[name]


def rel_err(test, ref):
    ref = ref.double()
    return ((test.double() - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library; GPU tests fail (not skip) if it or the device is missing."""
    from realcamnet_amd import _lib
    lib = _lib.load()
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    return lib
