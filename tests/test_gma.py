"""GroupMix GMA_Block: oracle vs reference goldens (CPU), HIP path vs goldens (GPU)."""
import pytest
import torch

import groupmix_oracle as GO
import liteisp_oracle as O
import realcamnet_amd as M
from conftest import golden_names, load_golden, rel_err

GMA = golden_names("gma_block_")


@pytest.mark.parametrize("fixture", GMA)
def test_oracle_matches_reference(fixture):
    g = load_golden(fixture)
    hw = tuple(int(v) for v in g["hw"])
    with torch.no_grad():
        y = GO.gma_block(g["sd"], g["x"], hw, 8)
    assert (y - g["y"]).abs().max().item() <= 2e-5 * g["y"].abs().max().item()


@pytest.mark.parametrize("fixture", GMA)
def test_reference_state_dict_loads_strict(fixture):
    g = load_golden(fixture)
    blk = M.GMA_Block(g["x"].shape[-1], 8)
    assert list(blk.state_dict().keys()) == list(g["sd"].keys())
    blk.load_state_dict(g["sd"], strict=True)


def test_constraints_raise_like_upstream():
    with pytest.raises(NotImplementedError):
        M.GMA_Block(80, 8, drop_path_rate=0.1)
    blk = M.GMA_Block(80, 8).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(torch.zeros(1, 16, 80), (4, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", GMA)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gma_block_vs_reference_golden(hip, fixture, dt):
    """fp32: <= 5e-5 relative (1x1 convs on exact-f32 MFMA; softmax/k^T v in another summation order than ATen);
    bf16 storage: PSNR >= 40 dB vs the fp32 reference (5 bf16-rounded intermediates per token)."""
    g = load_golden(fixture)
    hw = tuple(int(v) for v in g["hw"])
    blk = M.GMA_Block(g["x"].shape[-1], 8)
    blk.load_state_dict(g["sd"], strict=True)
    blk = blk.to("cuda", dt).eval()
    with torch.no_grad():
        y = blk(g["x"].to("cuda", dt), hw)
        y2 = blk(g["x"].to("cuda", dt), hw)
    torch.cuda.synchronize()
    assert y.shape == g["y"].shape and y.dtype == dt
    assert torch.equal(y, y2)                                   # fixed-order reductions: run-to-run bitwise stable
    if dt == torch.float32:
        assert rel_err(y.float().cpu(), g["y"]) <= 5e-5
    else:
        assert O.psnr(y.float().cpu(), g["y"]) >= 40.0


@pytest.mark.gpu
def test_gma_long_sequence_matches_oracle(hip):
    """N = 96*160 = 15360 tokens: several reduction blocks per image -> exercises the online-softmax merge."""
    g = load_golden("gma_block_80_32x32")
    gen = torch.Generator().manual_seed(3)
    hw = (96, 160)
    x = torch.randn(1, hw[0] * hw[1], 80, generator=gen)
    x[0, 777, :] += 25.0                                          # one outlier token moves the running max mid-stream
    blk = M.GMA_Block(80, 8)
    blk.load_state_dict(g["sd"], strict=True)
    with torch.no_grad():
        ref = GO.gma_block(g["sd"], x, hw, 8)
        y = blk.to("cuda").eval()(x.to("cuda"), hw)
    assert rel_err(y.cpu(), ref) <= 5e-5


def test_gma_model_keeps_reference_base_parameters():
    """The cfg3 composition adds modules AFTER the reference's: its base parameters are the reference's seed-0 values."""
    from conftest import seed0_state_dict
    torch.manual_seed(0)
    net = M.LiteISPNet_GFM_LSC_GMA().eval()
    base = seed0_state_dict("LiteISPNet_GFM_LSC")
    sd = net.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in base.items())
    extra = [k for k in sd if k not in base]
    assert extra and all(k.startswith(("gma_in.", "gma.", "gma_out.")) for k in extra)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,floor", [(torch.float32, 95.0), (torch.bfloat16, 48.0)])
def test_gma_model_vs_oracle(hip, dt, floor):
    torch.manual_seed(0)
    net = M.LiteISPNet_GFM_LSC_GMA().eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    h, w = 44, 70
    mosaic = torch.rand(2, 1, 2 * h, 2 * w, generator=g)
    cond = torch.rand(2, 4, 32, 48, generator=g)
    coord = O.make_coord(2, h, w)
    with torch.no_grad():
        ref = O.run_padded("LiteISPNet_GFM_LSC_GMA", sd, O.bayer_unshuffle(mosaic), cond, coord)
        y = net.to("cuda", dt).forward_mosaic(mosaic.to("cuda", dt), cond.to("cuda", dt), coord.to("cuda", dt))
    assert y.shape == ref.shape
    assert O.psnr(y.float().cpu(), ref) >= floor
