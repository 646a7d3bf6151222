"""Generate tests/golden/raw2bit_*.npz by running the IMPORTED reference's models/raw2bit.py classes (build container only).

    python oracle/make_golden_raw2bit.py [--blocks | --gma | --forward | --forward-base]

CompressAI is absent: the reference module is imported over stubs (oracle/_import_reference.py) and the CompressAI classes it
instantiates are replaced by the restatements of oracle/make_golden_tcm.py, so the fixtures pin UPSTREAM'S OWN code
(ConvTransBlock_mzj, ResidualBlockWithCA, SpatialFeatureTransform, HybridConditionModule, raw_compression_tcm_final.__init__ /
.forward) and leave the CompressAI layers unpinned.  Fixtures are data only.  The full model is too large to ship its weights:
its parameters are oracle/det_fill.py's function of key and shape, re-created by the tests."""
import importlib
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import _import_reference as R
import make_golden_tcm as G
import raw2bit_oracle as RO
from det_fill import det_fill_

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "kepengxu/RealCamNet@2024-10-20 (+ restated compressai classes)"


def _load_reference():
    R.install_stubs()
    L = sys.modules["compressai.layers"]
    L.AttentionBlock, L.ResidualBlock = G._RestatedAttentionBlock, G._RestatedResidualBlock
    L.ResidualBlockWithStride, L.ResidualBlockUpsample = G._RestatedResidualBlockWithStride, G._RestatedResidualBlockUpsample
    L.conv3x3, L.subpel_conv3x3, L.GDN = G._restated_conv3x3, G._restated_subpel_conv3x3, G._RestatedGDN
    sys.modules["compressai.models"].CompressionModel = G._Dummy
    sys.modules["compressai.entropy_models"].EntropyBottleneck = G._RestatedEntropyBottleneck
    sys.modules["compressai.entropy_models"].GaussianConditional = G._RestatedGaussianConditional
    for m in ("models.tcm", "models.raw2bit"):
        sys.modules.pop(m, None)
    return importlib.import_module("models.raw2bit")


def _save(name, arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def blocks():
    torch.set_num_threads(1)
    T = _load_reference()
    g = torch.Generator().manual_seed(1122)
    for typ in ("W", "SW"):
        torch.manual_seed(0)
        m = T.ConvTransBlock_mzj(32, 32, 16, 8, 0.0, type=typ).eval()
        with torch.no_grad():
            G._perturb(m, g)
            x, cond = torch.randn(1, 64, 16, 24, generator=g), torch.randn(1, 32, 16, 24, generator=g)
            y, _ = m([x, cond])
            sd = m.state_dict()
            yo = RO.conv_trans_block_mzj(sd, "", x, cond, 32, 32, 16, 8, typ)
        assert (y - yo).abs().max() <= 1e-5 * y.abs().max(), typ
        arrays = {"x": x.numpy(), "cond": cond.numpy(), "y": y.numpy(), "type": np.array(typ), "torch_version": np.array(torch.__version__),
                  "reference": np.array(REF)}
        arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
        _save(f"raw2bit_convtrans_mzj_{typ}_n32_hd16_ws8", arrays)
    torch.manual_seed(0)
    m = T.HybridConditionModule(out_channels=32, init_mid_channels=16).eval()
    with torch.no_grad():
        G._perturb(m, g)
        x = torch.rand(1, 4, 32, 48, generator=g)
        ys = m(x)
        sd = m.state_dict()
        yo = RO.hybrid_condition_module(sd, "", x)
    for a, b in zip(ys, yo):
        assert (a - b).abs().max() <= 1e-5 * a.abs().max()
    arrays = {"x": x.numpy(), "cond_1": ys[0].numpy(), "cond_2": ys[1].numpy(), "cond_3": ys[2].numpy(), "torch_version": np.array(torch.__version__),
              "reference": np.array("kepengxu/RealCamNet@2024-10-20")}
    arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
    _save("raw2bit_hycond_c32", arrays)


def gma_blocks():
    """GMAAtten / ConvGMABlock (models/raw2bit.py:209-234, 330-355; the sizes of the file's own smoke lines :4362-4363, scaled) and RBU."""
    torch.set_num_threads(1)
    T = _load_reference()
    g = torch.Generator().manual_seed(3344)
    meta = {"torch_version": np.array(torch.__version__), "reference": np.array(REF)}
    cases = (("raw2bit_gmaatten_96_hd10_i80", lambda: T.GMAAtten(96, 96, 10, 0., 80), (1, 96, 16, 24), lambda sd, x: RO.gma_atten(sd, "", x, 10)),
             ("raw2bit_convgma_32_80_hd10", lambda: T.ConvGMABlock(32, 80, 10, drop_path=0.), (1, 112, 16, 24),
              lambda sd, x: RO.conv_gma_block(sd, "", x, 32, 80, 10)),
             ("raw2bit_convgma_16_40_hd5", lambda: T.ConvGMABlock(16, 40, 5, drop_path=0.), (2, 56, 8, 16),
              lambda sd, x: RO.conv_gma_block(sd, "", x, 16, 40, 5)),
             ("raw2bit_rbu_48_32", lambda: T.RBU(48, 32, 2), (2, 48, 8, 12), lambda sd, x: RO.rbu(sd, "", x)))
    for name, make, shape, orc in cases:
        torch.manual_seed(0)
        m = make().eval()
        with torch.no_grad():
            G._perturb(m, g)
            x = torch.randn(*shape, generator=g)
            y = m(x)
            sd = m.state_dict()
            yo = orc(sd, x)
        assert (y - yo).abs().max() <= 1e-5 * y.abs().max(), (name, (y - yo).abs().max())
        arrays = {"x": x.numpy(), "y": y.numpy(), **meta}
        arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
        _save(name, arrays)


def forward_base():
    """raw_compression_tcm (models/raw2bit.py:361-579) at N=32 on a 256 x 256 packed frame (the smallest whose latent is larger than the 8 x 8 window); det_fill parameters."""
    torch.set_num_threads(8)
    T = _load_reference()
    torch.manual_seed(0)
    n, slices = 32, 5
    model = T.raw_compression_tcm(N=n, M=320, num_slices=slices).eval()
    sd = model.state_dict()
    det_fill_(sd)
    g = torch.Generator().manual_seed(98)
    raw = torch.rand(1, 4, 256, 256, generator=g)
    cond = torch.rand(1, 4, 64, 64, generator=g)
    import liteisp_oracle as LO
    coord = LO.make_coord(1, 256, 256)
    flat = lambda o: {"x_hat": o["x_hat"], "lik_y": o["likelihoods"]["y"], "lik_z": o["likelihoods"]["z"], "means": o["para"]["means"],
                      "scales": o["para"]["scales"], "y": o["para"]["y"]}
    with torch.no_grad():
        out = flat(model([raw, cond, coord]))
        ours = flat(RO.raw_compression_tcm(sd, [raw, cond, coord], N=n, num_slices=slices))
    for k in out:
        assert (out[k] - ours[k]).abs().max() <= 1e-4 * max(out[k].abs().max().item(), 1e-6), (k, (out[k] - ours[k]).abs().max())
    arrays = {"raw": raw.numpy(), "cond": cond.numpy(), "N": np.array(n), "num_slices": np.array(slices),
              "n_keys": np.array(len(sd)), "torch_version": np.array(torch.__version__), "reference": np.array(REF + "; det_fill parameters")}
    arrays.update({"out." + k: v.numpy() for k, v in out.items() if k != "x_hat"})
    arrays["out.x_hat"] = out["x_hat"].numpy().astype(np.float16)
    _save("raw2bit_base_forward_n32", arrays)
    print({k: (tuple(v.shape), float(v.abs().mean())) for k, v in out.items()})


def _flat(o):
    return {"x_hat": o["x_hat"], "lik_y": o["likelihoods"]["y"], "lik_z": o["likelihoods"]["z"], "means": o["para"]["means"],
            "scales": o["para"]["scales"], "y": o["para"]["y"], "lft": o["lft"], "lsc": o["lsc"]}


def forward():
    torch.set_num_threads(8)
    T = _load_reference()
    torch.manual_seed(0)
    n, slices = 32, 5
    model = T.raw_compression_tcm_final(N=n, M=320, num_slices=slices).eval()
    sd = model.state_dict()
    det_fill_(sd)
    g = torch.Generator().manual_seed(97)
    raw = torch.rand(1, 4, 256, 256, generator=g)
    cond = torch.rand(1, 4, 64, 64, generator=g)
    import liteisp_oracle as LO
    coord = LO.make_coord(1, 256, 256)
    with torch.no_grad():
        out = _flat(model([raw, cond, coord]))
        ours = _flat(RO.raw_compression_tcm_final(sd, [raw, cond, coord], N=n, num_slices=slices))
    for k in out:
        assert (out[k] - ours[k]).abs().max() <= 1e-4 * max(out[k].abs().max().item(), 1e-6), (k, (out[k] - ours[k]).abs().max())
    # coord is not stored: liteisp_oracle.make_coord(1, 256, 256)
    arrays = {"raw": raw.numpy(), "cond": cond.numpy(), "N": np.array(n), "num_slices": np.array(slices),
              "n_keys": np.array(len(sd)), "torch_version": np.array(torch.__version__), "reference": np.array(REF + "; det_fill parameters")}
    arrays.update({"out." + k: v.numpy() for k, v in out.items() if k not in ("x_hat", "lsc")})
    arrays["out.lsc_s8"] = out["lsc"][:, :, ::8, ::8].contiguous().numpy()   # every 8th pixel of the 64 x 256 x 256 map
    arrays["out.x_hat"] = out["x_hat"].numpy().astype(np.float16)      # 3 x 512 x 512: stored at half precision (compared at 1e-2)
    _save("raw2bit_final_forward_n32", arrays)
    print({k: (tuple(v.shape), float(v.abs().mean())) for k, v in out.items()})


if __name__ == "__main__":
    if "--forward" in sys.argv:
        forward()
    elif "--forward-base" in sys.argv:
        forward_base()
    elif "--gma" in sys.argv:
        gma_blocks()
    else:
        blocks()
