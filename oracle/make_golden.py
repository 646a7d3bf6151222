"""Generate tests/golden/*.npz by running the IMPORTED reference (build container only).

    python oracle/make_golden.py

The reference (kepengxu/RealCamNet @ 2024-10-20, /root/reference) ships no known-answer vectors, so
these fixtures -- outputs of the reference's own modules on seeded inputs -- are what pins the
oracle (oracle/liteisp_oracle.py) and, through it, the HIP path.  Fixtures are data only: inputs,
(small) weights, expected outputs, and digests.  The reference source never leaves /root/reference.

Determinism: torch.set_num_threads(1); torch {version recorded}; weights torch.manual_seed(0) with
default init in the reference's construction order; inputs torch.Generator().manual_seed(1234).
End-to-end nets are too large to store (9-14 M parameters), so those fixtures store a digest of the
seed-0 state_dict instead; the build's mirror modules reproduce the same parameters from the seed.
"""
import hashlib
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import _import_reference as R
import liteisp_oracle as O

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def sd_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].detach().cpu().float().numpy()).tobytes())
    return h.hexdigest()


def save(name, **arrays):
    meta = {"torch_version": np.array(torch.__version__), "reference": np.array("kepengxu/RealCamNet@2024-10-20")}
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays, **meta)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sd_arrays(mod, prefix="sd."):
    return {prefix + k: v for k, v in mod.state_dict().items()}


def main():
    torch.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    N, L = R.load("networks", "LiteISP")
    g = torch.Generator().manual_seed(1234)

    def rnd(*shape, signed=False):
        t = torch.rand(*shape, generator=g)
        return t * 2 - 1 if signed else t

    with torch.no_grad():
        # ---- per-block fixtures (weights stored) -------------------------------------------------
        torch.manual_seed(0)
        m = N.DWTForward(16).eval(); x = rnd(2, 16, 12, 20, signed=True)
        save("block_dwt_forward", x=x, y=m(x), **sd_arrays(m))
        m = N.DWTInverse(64).eval(); x = rnd(2, 64, 6, 10, signed=True)
        save("block_dwt_inverse", x=x, y=m(x), **sd_arrays(m))

        torch.manual_seed(0)
        m = N.conv(16, 32, mode="C").eval(); x = rnd(2, 16, 11, 37, signed=True)   # ragged size on purpose
        save("block_conv3x3_16_32", x=x, y=m(x), **sd_arrays(m))
        m = N.conv(48, 48, mode="CRC").eval(); x = rnd(1, 48, 16, 40, signed=True)
        save("block_conv_crc_48", x=x, y=m(x), **sd_arrays(m))

        torch.manual_seed(0)
        m = N.CALayer(32, 16).eval(); x = rnd(2, 32, 9, 13, signed=True)
        save("block_calayer_32", x=x, y=m(x), **sd_arrays(m))
        m = N.RCABlock(32, 32).eval(); x = rnd(2, 32, 16, 24, signed=True)
        save("block_rcab_32", x=x, y=m(x), **sd_arrays(m))
        m = N.RCAGroup(32, 32, nb=4).eval(); x = rnd(1, 32, 16, 40, signed=True)
        save("block_rcag_32_nb4", x=x, y=m(x), **sd_arrays(m))
        m = N.RCAGroup(48, 48, nb=2).eval(); x = rnd(1, 48, 8, 32, signed=True)
        save("block_rcag_48_nb2", x=x, y=m(x), **sd_arrays(m))

        torch.manual_seed(0)
        m = L.Res_GFM(in_nc=48, chan=48, cond_c=32, out_nc=48, nf=48).eval()
        x = rnd(2, 48, 8, 24, signed=True); v = rnd(2, 32, signed=True)
        y, _ = m((x, v))
        save("block_res_gfm_48", x=x, v=v, y=y, **sd_arrays(m))
        m = L.Lens_Shading_Correction(in_channels=2, out_c=48, nf=48).eval(); x = O.make_coord(2, 16, 24)
        save("block_lsc_48", x=x, y=m(x), **sd_arrays(m))
        m = L.Color_Condition_GFM(in_channels=4, out_c=32).eval(); x = rnd(2, 4, 64, 48)
        save("block_color_condition", x=x, y=m(x).squeeze(3).squeeze(2), **sd_arrays(m))

        torch.manual_seed(0)
        ps = torch.nn.Sequential(N.conv(16, 64, mode="C"), torch.nn.PixelShuffle(2), N.conv(16, 3, mode="C")).eval()
        x = rnd(1, 16, 8, 24, signed=True)
        save("block_tail_16", x=x, y=ps(x), **sd_arrays(ps))

        raw, hw = L.pad_to_multiple_of_16(rnd(1, 4, 21, 35))
        save("block_pad16", x_shape=np.array([1, 4, 21, 35]), y=raw, hw=np.array(hw))

        # ---- end-to-end fixtures (weights reproduced from seed 0; digest stored) -------------------
        for name in ("LiteISPNet", "LiteISPNet_GFM_LSC", "ISPUNet_GFM_LSC"):
            torch.manual_seed(0)
            net = getattr(L, name)().eval()
            dig = sd_digest(net.state_dict())
            for hw_ in ((32, 32), (64, 64), (40, 72)):
                h, w = hw_
                raw = rnd(1, 4, h, w)
                cond = rnd(1, 4, 64, 64)
                coord = O.make_coord(1, h, w)
                y = net([raw, cond, coord])
                save(f"e2e_{name}_{h}x{w}", raw=raw, cond=cond, coord=coord, y=y, sd_digest=np.array(dig))
            # the reference smoke main's own input convention: randn (models/LiteISP.py:2670-2672)
            gn = torch.Generator().manual_seed(4321)
            raw = torch.randn(1, 4, 32, 32, generator=gn); cond = torch.randn(1, 4, 32, 32, generator=gn)
            coord = torch.randn(1, 2, 32, 32, generator=gn)
            save(f"e2e_{name}_randn_32x32", raw=raw, cond=cond, coord=coord, y=net([raw, cond, coord]), sd_digest=np.array(dig))

        # ---- added in round 4 (LAST, with their own generator, so that every earlier fixture regenerates bit-identically):
        # the channel-count-agnostic Haar pair (models/networks.py:9-47): one (4,1,2,2) tap set repeated over any C
        g4 = torch.Generator().manual_seed(2024)
        m = N.DWTForward_().eval()
        x = torch.rand(2, 24, 12, 20, generator=g4) * 2 - 1
        save("block_dwt_forward_anyc", x=x, y=m(x), **sd_arrays(m))
        m = N.DWTInverse_().eval()
        x = torch.rand(2, 32, 6, 10, generator=g4) * 2 - 1
        save("block_dwt_inverse_anyc", x=x, y=m(x), **sd_arrays(m))


if __name__ == "__main__":
    main()
