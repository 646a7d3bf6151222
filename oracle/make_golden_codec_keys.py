"""Generate tests/golden/codec_state_keys.npz (build container only):   python oracle/make_golden_codec_keys.py

The state_dict CONTRACT of the two codecs after `update()`: every key, shape and dtype that the reference's own classes
(models/tcm.py `TCM`, models/raw2bit.py `raw_compression_tcm_final`: their __init__, update() at tcm.py:430-435 and load_state_dict /
update_registered_buffers at tcm.py:91-137, 492-499) produce when run over restated CompressAI entropy models that carry
EntropyModel's buffers (`_offset`, `_quantized_cdf`, `_cdf_length`, LowerBound `bound`s, `scale_table`, `scale_bound`, `target`).
CompressAI is absent (unpinned upstream): the buffer set is its published one, restated.  Also stored: SHA-256 of the tables the oracle
builds for key-filled parameters, and the proof that the reference's own load_state_dict accepts such a checkpoint."""
import hashlib
import importlib
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import _import_reference as R
import entropy_oracle as E
import make_golden_tcm as G
from det_fill import det_fill_

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _LowerBound(torch.nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


def _entropy_model_buffers(mod, likelihood_bound=1e-9):
    mod.likelihood_lower_bound = _LowerBound(likelihood_bound)
    for name in ("_offset", "_quantized_cdf", "_cdf_length"):
        mod.register_buffer(name, torch.IntTensor())


class _EB(G._RestatedEntropyBottleneck):
    def __init__(self, channels, *a, **k):
        super().__init__(channels, *a, **k)
        _entropy_model_buffers(self)

    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        t = E.eb_update({"e." + k: v for k, v in self.state_dict().items()}, "e")
        self._offset, self._quantized_cdf, self._cdf_length = t["_offset"], t["_quantized_cdf"], t["_cdf_length"]
        return True


class _GC(G._RestatedGaussianConditional):
    def forward(self, inputs, scales, means=None):
        self_bound = float(self.scale_bound)
        outputs = torch.round(inputs - means) + means
        values = torch.abs(outputs - means)
        s = scales.clamp_min(self_bound)
        phi = lambda t: 0.5 * torch.erfc(-(2 ** -0.5) * t)
        return outputs, (phi((0.5 - values) / s) - phi((-0.5 - values) / s)).clamp_min(1e-9)

    def __init__(self, scale_table=None, scale_bound=0.11, tail_mass=1e-9):
        super().__init__(scale_table, scale_bound)
        del self.scale_bound                                    # the base restatement keeps it as a float; CompressAI registers a buffer
        _entropy_model_buffers(self)
        self.tail_mass = float(tail_mass)
        self.lower_bound_scale = _LowerBound(scale_bound)
        self.register_buffer("scale_table", torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = torch.as_tensor(sorted(float(s) for s in scale_table), dtype=torch.float32)
        t = E.gc_update(self.scale_table, self.tail_mass)
        self._offset, self._quantized_cdf, self._cdf_length = t["_offset"], t["_quantized_cdf"], t["_cdf_length"]
        return True


class _CompressionModel(torch.nn.Module):
    """compressai.models.CompressionModel: only what the reference calls through super() -- update() over the EntropyBottleneck
    children."""

    def __init__(self, *a, **k):
        super().__init__()

    def update(self, force=False):
        updated = False
        for m in self.children():
            if isinstance(m, _EB):
                updated |= m.update(force=force)
        return updated

    def load_state_dict(self, state_dict):
        # CompressionModel.load_state_dict: size the bottleneck's table buffers from the checkpoint (with the reference module's own
        # update_registered_buffers, models/tcm.py:91-137), then nn.Module.load_state_dict (strict)
        fn = sys.modules[type(self).__module__].update_registered_buffers
        fn(self.entropy_bottleneck, "entropy_bottleneck", ["_quantized_cdf", "_offset", "_cdf_length"], state_dict)
        torch.nn.Module.load_state_dict(self, state_dict)


def _load(module):
    R.install_stubs()
    L = sys.modules["compressai.layers"]
    L.AttentionBlock, L.ResidualBlock = G._RestatedAttentionBlock, G._RestatedResidualBlock
    L.ResidualBlockWithStride, L.ResidualBlockUpsample = G._RestatedResidualBlockWithStride, G._RestatedResidualBlockUpsample
    L.conv3x3, L.subpel_conv3x3, L.GDN = G._restated_conv3x3, G._restated_subpel_conv3x3, G._RestatedGDN
    sys.modules["compressai.models"].CompressionModel = _CompressionModel
    sys.modules["compressai.entropy_models"].EntropyBottleneck = _EB
    sys.modules["compressai.entropy_models"].GaussianConditional = _GC
    for m in ("models.tcm", "models.raw2bit"):
        sys.modules.pop(m, None)
    return importlib.import_module(module)


def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


def _contract(build):
    torch.manual_seed(0)
    m = build().eval()
    det_fill_(m.state_dict())
    assert m.update() is True                                   # the REFERENCE's update(): get_scale_table + tables
    sd = m.state_dict()
    torch.manual_seed(1)
    fresh = build().eval()                                      # table buffers empty: the reference's load_state_dict must resize them
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.gaussian_conditional._quantized_cdf, sd["gaussian_conditional._quantized_cdf"])
    keys = list(sd.keys())
    return {"keys": np.array(keys), "shapes": np.array(["x".join(map(str, sd[k].shape)) for k in keys]),
            "dtypes": np.array([str(sd[k].dtype) for k in keys]),
            "sha_gc_cdf": np.array(_sha(sd["gaussian_conditional._quantized_cdf"])), "sha_eb_cdf": np.array(_sha(sd["entropy_bottleneck._quantized_cdf"])),
            "sha_gc_offset": np.array(_sha(sd["gaussian_conditional._offset"])), "sha_eb_length": np.array(_sha(sd["entropy_bottleneck._cdf_length"]))}


def main():
    torch.set_num_threads(1)
    arrays = {"torch_version": np.array(torch.__version__),
              "reference": np.array("kepengxu/RealCamNet@2024-10-20 TCM / raw_compression_tcm_final __init__, update, load_state_dict "
                                    "(+ restated compressai classes with EntropyModel buffers; det_fill parameters)")}
    T = _load("models.tcm")
    for k, v in _contract(lambda: T.TCM(N=32, M=320, num_slices=5)).items():
        arrays["tcm." + k] = v
    RB = _load("models.raw2bit")
    for k, v in _contract(lambda: RB.raw_compression_tcm_final(N=32, M=320, num_slices=5)).items():
        arrays["raw." + k] = v
    path = os.path.join(OUT, "codec_state_keys.npz")
    np.savez_compressed(path, **arrays)
    print(f"codec_state_keys: {os.path.getsize(path) / 1024:.1f} KiB; TCM {len(arrays['tcm.keys'])} tensors, raw codec {len(arrays['raw.keys'])} tensors")


if __name__ == "__main__":
    main()
