"""Generate tests/golden/e2e_<Net>_*.npz for the rest of the ISP family by running the IMPORTED reference
(build container only):   python oracle/make_golden_family.py

ResUNet, ISPUNet_GFM, ISPUNet_LSC (models/LiteISP.py:2038-2146, 963-1110, 1113-1225) and LiteISPNet_LSC, LiteISPNet_GFM,
LiteISPNet_GFMresize (:1710-1805, 1809-1920, 2414-2520), ISPUNet_GFM_LFM (:1535-1707).  Same conventions as oracle/make_golden.py: torch.set_num_threads(1),
weights = torch.manual_seed(0) default init in the reference's construction order (a SHA-256 of the state_dict is stored, the
weights are not), inputs from torch.Generator().manual_seed(2468).  At generation time the oracle restatement must equal the
reference output (checked here, max |diff| printed)."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import _import_reference as R
import liteisp_oracle as O
from make_golden import save, sd_arrays, sd_digest

NETS = ("ResUNet", "ISPUNet_GFM", "ISPUNet_LSC", "LiteISPNet_LSC", "LiteISPNet_GFM", "LiteISPNet_GFMresize", "ISPUNet_GFM_LFM")


def main():
    torch.set_num_threads(1)
    (L,) = R.load("LiteISP")
    g = torch.Generator().manual_seed(2468)
    with torch.no_grad():
        for name in NETS:
            torch.manual_seed(0)
            net = getattr(L, name)().eval()
            sd = net.state_dict()
            dig = sd_digest(sd)
            for (h, w) in ((32, 32), (40, 72)):
                raw = torch.rand(1, 4, h, w, generator=g)
                cond = torch.rand(1, 4, 64, 64, generator=g)
                coord = O.make_coord(1, h, w)
                y = net([raw, cond, coord])
                yo = O.FORWARDS[name](sd, [raw, cond, coord])
                err = (y - yo).abs().max().item()
                assert err <= 1e-5 * y.abs().max().item(), (name, err)
                save(f"e2e_{name}_{h}x{w}", raw=raw, cond=cond, coord=coord, y=y, sd_digest=np.array(dig), n_tensors=np.array(len(sd)))
                print(f"   oracle max |diff| {err:.2e}")

        # blocks of the GFM+LFM variant on their own (models/LiteISP.py:215-230, 293-321, 501-534, 601-620)
        def rnd(*shape):
            return torch.rand(*shape, generator=g) * 2 - 1

        torch.manual_seed(0)
        m = L.Res_GFM_LFM(cond_c=32, out_nc=64, nf=128).eval()
        x, v, cm = rnd(2, 64, 8, 24), rnd(2, 32), rnd(2, 32, 8, 24)
        y = m((x, v, cm))[0]
        save("block_res_gfm_lfm_64", x=x, v=v, cmap=cm, y=y, **sd_arrays(m))
        m = L.SFTLayer(cond_c=32, out_nc=32, nf=32).eval()
        x, cm = rnd(2, 32, 8, 24), rnd(2, 32, 8, 24)
        save("block_sftlayer_32", x=x, cmap=cm, y=m((x, cm)), **sd_arrays(m))
        m = L.GFMLayer(cond_c=32, out_nc=128, nf=256).eval()
        x, v = rnd(2, 128, 4, 8), rnd(2, 32)
        save("block_gfmlayer_128", x=x, v=v, y=m((x, v)), **sd_arrays(m))
        m = L.Color_Condition_GFM_LFM(in_channels=4, GFM_out_c=32, LFM_out_c=32).eval()
        gl, loc = torch.rand(2, 4, 64, 48, generator=g), torch.rand(2, 4, 16, 24, generator=g)
        vec, lfm = m(gl, loc)
        save("block_color_condition_gfm_lfm", x=gl, local=loc, y=vec.squeeze(3).squeeze(2), lfm=lfm, **sd_arrays(m))
        m = L.CB(4, 16, normalization=True).eval()
        x = torch.rand(2, 4, 32, 24, generator=g)
        save("block_cb_4_16", x=x, y=m(x), **sd_arrays(m))


if __name__ == "__main__":
    main()
