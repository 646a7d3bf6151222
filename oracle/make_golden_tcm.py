"""Generate tests/golden/tcm_block_*.npz by running the IMPORTED reference's tcm.Block (build container only).

    python oracle/make_golden_tcm.py

Fixtures are data only (inputs, the block's small state_dict, expected outputs).  CompressAI is absent and stubbed by
name (oracle/_import_reference.py); tcm.Block / WMSA do not use it.  torch.set_num_threads(1); weights from
torch.manual_seed(0) + a bias/LN perturbation so that zero-initialised parameters are exercised too."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import _import_reference as R
import tcm_oracle as TO

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES = [  # name, C, head_dim, window, type, (b, h, w)
    ("tcm_block_W_ws8_c64_hd16", 64, 16, 8, "W", (2, 16, 24)),
    ("tcm_block_SW_ws8_c64_hd16", 64, 16, 8, "SW", (1, 16, 24)),
    ("tcm_block_W_ws4_c32_hd8", 32, 8, 4, "W", (2, 8, 12)),
    ("tcm_block_SW_ws4_c64_hd32", 64, 32, 4, "SW", (1, 12, 8)),
]


def main():
    torch.set_num_threads(1)
    (T,) = R.load("tcm")
    g = torch.Generator().manual_seed(1234)
    for name, c, hd, ws, typ, shape in CASES:
        torch.manual_seed(0)
        m = T.Block(c, c, hd, ws, 0.0, type=typ).eval()
        with torch.no_grad():
            for k, v in m.state_dict().items():          # exercise biases / LayerNorm affine / a livelier position bias
                if k.endswith("bias") or "ln" in k:
                    v.add_(torch.randn(v.shape, generator=g) * 0.1)
                if k.endswith("relative_position_params"):
                    v.add_(torch.randn(v.shape, generator=g) * 0.5)
            x = torch.randn(*shape, c, generator=g)
            y = m(x)
            sd = m.state_dict()
            yo = TO.block(sd, "", x, hd, ws, typ)
        assert (y - yo).abs().max() <= 1e-5 * y.abs().max(), name
        arrays = {"x": x.numpy(), "y": y.numpy(), "head_dim": np.array(hd), "window": np.array(ws), "type": np.array(typ),
                  "torch_version": np.array(torch.__version__), "reference": np.array("kepengxu/RealCamNet@2024-10-20")}
        arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, oracle max |diff| {(y - yo).abs().max().item():.2e}")


if __name__ == "__main__":
    main()
