"""Generate tests/golden/gma_*.npz from the IMPORTED reference GMA_Block (build container only).

    python oracle/make_golden_gma.py

Shapes follow the reference's own smoke test (models/raw2bit.py:4361-4367: dim 80 and dim 200 @ 32x32) plus a
ragged 24x40 case.  BatchNorm running statistics and all affine parameters are randomised (a fresh module has
mean 0 / var 1 / gamma 1 / beta 0, which would hide indexing mistakes); weights are stored in the fixture.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import _import_reference as R
import groupmix_oracle as GO

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    torch.set_num_threads(1)
    (G,) = R.load("groupmix")
    g = torch.Generator().manual_seed(99)
    for dim, hw in ((80, (32, 32)), (80, (24, 40)), (200, (32, 32))):
        torch.manual_seed(0)
        blk = G.GMA_Block(dim, 8).eval()
        with torch.no_grad():
            for name, t in blk.state_dict().items():
                if name.endswith("running_var"):
                    t.copy_(torch.rand(t.shape, generator=g) + 0.5)
                elif name.endswith("running_mean"):
                    t.copy_(torch.randn(t.shape, generator=g) * 0.2)
                elif ("norm" in name) and name.endswith(".weight"):
                    t.copy_(1 + 0.2 * torch.randn(t.shape, generator=g))
                elif name.endswith(".bias"):
                    t.copy_(0.1 * torch.randn(t.shape, generator=g))
        x = torch.randn(2, hw[0] * hw[1], dim, generator=g)
        with torch.no_grad():
            y = blk(x, hw)
            mine = GO.gma_block(blk.state_dict(), x, hw, 8)
        assert torch.equal(y, mine), (y - mine).abs().max()
        arrays = {"x": x.numpy(), "y": y.numpy(), "hw": np.array(hw), "torch_version": np.array(torch.__version__)}
        arrays.update({"sd." + k: v.numpy() for k, v in blk.state_dict().items()})
        path = os.path.join(OUT, f"gma_block_{dim}_{hw[0]}x{hw[1]}.npz")
        np.savez_compressed(path, **arrays)
        print(os.path.basename(path), f"{os.path.getsize(path) / 1024:.0f} KiB", "oracle == reference bitwise")


if __name__ == "__main__":
    main()
