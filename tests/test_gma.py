"""GroupMix GMA_Block: oracle vs reference goldens (CPU), HIP path vs goldens (GPU)."""
import pytest
import torch

import groupmix_oracle as GO
import liteisp_oracle as O
import realcamnet_amd as M
from conftest import golden_names, load_golden, rel_err
from realcamnet_amd import ops

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
    bf16 storage: PSNR >= 60 dB vs the fp32 reference (measured on MI355X: 68.2 - 69.5 dB on the three fixtures)."""
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
        p = O.psnr(y.float().cpu(), g["y"])
        _metric(f"gma_block bf16 {fixture}", p)
        assert p >= 60.0, p


def _metric(name, value):
    import json, os
    path = os.environ.get("RC_METRICS_OUT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({name: value}) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", [f for f in GMA if "_80_" in f])
def test_gma_fused_stages_vs_layer_by_layer_and_reference(hip, fixture):
    """dim 80, bf16: the per-token stages as two launches with register-resident activations (csrc/gma_fused.hip) against the
    layer-by-layer path (same rounding points; fp32 summation order and the erf evaluation differ) and the fp32 reference."""
    from realcamnet_amd import ops
    g = load_golden(fixture)
    hw = tuple(int(v) for v in g["hw"])
    blk = M.GMA_Block(80, 8)
    blk.load_state_dict(g["sd"], strict=True)
    blk = blk.to("cuda", torch.bfloat16).eval()
    x = g["x"].to("cuda", torch.bfloat16)
    old = ops.FUSE_GMA
    try:
        with torch.no_grad():
            ops.FUSE_GMA = True
            yf = blk(x, hw); yf2 = blk(x, hw)
            ops.FUSE_GMA = False
            yl = blk(x, hw)
    finally:
        ops.FUSE_GMA = old
    assert torch.equal(yf, yf2)
    pf, pl = O.psnr(yf.float().cpu(), g["y"]), O.psnr(yl.float().cpu(), g["y"])
    pfl = O.psnr(yf.float().cpu(), yl.float().cpu())
    _metric(f"gma_fused {fixture}", {"fused_vs_ref": pf, "layers_vs_ref": pl, "fused_vs_layers": pfl})
    assert pf >= 60.0 and pf >= pl - 1.0, (pf, pl)             # measured: 68.2 / 68.9 dB either way
    assert pfl >= 70.0, pfl                                     # measured: 80.7 / 81.6 dB between the two forms


@pytest.mark.gpu
def test_gma_fused_tail_with_output_conv_and_ragged_token_count(hip):
    """The cfg3 form: gma_out (80 -> 192) + residual folded into the block's last launch; 3 images of 37 x 29 = 1073 tokens
    (not a multiple of the 64-token wave tile, so the last tile of every image is partial)."""
    from realcamnet_amd import networks as N, ops
    torch.manual_seed(5)
    blk = M.GMA_Block(80, 8).to("cuda", torch.bfloat16).eval()
    conv = N.Conv2d(80, 192, 1, 1, 0).to("cuda", torch.bfloat16)
    gen = torch.Generator().manual_seed(6)
    a = torch.randn(3, 37, 29, 80, generator=gen).to("cuda", torch.bfloat16)
    d1 = torch.randn(3, 37, 29, 192, generator=gen).to("cuda", torch.bfloat16)
    old = ops.FUSE_GMA
    try:
        with torch.no_grad():
            ops.FUSE_GMA = True
            yf = blk._nhwc(a, post=(conv, d1))
            y0 = blk._nhwc(a)
            ops.FUSE_GMA = False
            yl = blk._nhwc(a, post=(conv, d1))
            y0l = blk._nhwc(a)
    finally:
        ops.FUSE_GMA = old
    assert yf.shape == (3, 37, 29, 192) and y0.shape == (3, 37, 29, 80)
    p1, p0 = O.psnr(yf.float().cpu(), yl.float().cpu()), O.psnr(y0.float().cpu(), y0l.float().cpu())
    _metric("gma_fused ragged", {"with_out_conv": p1, "block_only": p0})
    assert p1 >= 70.0 and p0 >= 70.0, (p1, p0)                  # measured: 82.5 / 86.7 dB
    # frames are independent: image 2 alone == image 2 of the batch (bitwise)
    with torch.no_grad():
        y2 = blk._nhwc(a[2:3], post=(conv, d1[2:3]))
    assert torch.equal(y2[0], yf[2])


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
@pytest.mark.parametrize("dt,floor", [(torch.float32, 100.0), (torch.bfloat16, 55.0)])      # measured: 139.0 / 61.8 dB
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
    p = O.psnr(y.float().cpu(), ref)
    _metric(f"gma_model {dt}", p)
    assert p >= floor, p


@pytest.mark.gpu
def test_kv_on_matrix_cores_equals_two_pass_form(hip):
    """rc_gma_aggregate's per-channel k maximum + rc_gma_kv_mfma against the two-pass VALU form (rc_gma_kv_planar) on the same aggregated
    planes: ragged token count (not a multiple of the 128-token tile), batch 2, and an outlier token that moves the maximum mid-stream.
    The MFMA form rounds exp(k - max) to bf16, so the bar is relative (1e-2 of the largest entry), and run-to-run bitwise equality."""
    R = torch.ops.realcam
    blk = M.GMA_Block(80, 8).to("cuda", torch.bfloat16).eval()
    g = torch.Generator().manual_seed(21)
    b, H, W = 2, 37, 53
    x = torch.randn(b, H, W, 80, generator=g)
    x[1, 20, 30] *= 25.0                                               # outlier token
    x = x.to("cuda", torch.bfloat16)
    with torch.no_grad():
        wq, bq = ops.packed_chain(blk.att.qkv)
        qkv = R.gma_ln_qkv(x, wq, bq, ops.f32_param(blk.norm1, "weight"), ops.f32_param(blk.norm1, "bias"), 1e-5)
        qkvp, loc, kmax = blk.att.aggregator._run_fused(qkv)
        k = qkvp[4:8].float().permute(1, 2, 3, 0, 4).reshape(b, H * W, 64)          # (B, tokens, segment * 16 + c)
        assert torch.equal(kmax, k.amax(dim=1))
        want = R.gma_kv(qkvp, 8, 8, float(blk.att.scale))
        got = R.gma_kv_mfma(qkvp, kmax, float(blk.att.scale))
        again = R.gma_kv_mfma(qkvp, kmax, float(blk.att.scale))
    assert got.shape == want.shape == (b, 8, 8, 8)
    assert torch.equal(got, again)
    assert (got - want).abs().max().item() <= 1e-2 * want.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("b,hw", [(2, (37, 53)), (1, (16, 32)), (3, (5, 7)), (1, (1, 1)), (2, (48, 96)), (1, (130, 201))])
@pytest.mark.parametrize("qkv_bias", [False, True])
def test_qkv_aggregate_in_one_launch_equals_the_two_launch_path(hip, b, hw, qkv_bias):
    """rc_gma_qkv_aggregate (LayerNorm1 + qkv + Aggregator in one launch, qkv kept on chip, the depth-wise windows as banded Toeplitz products on the
    matrix cores) against rc_gma_ln_qkv + rc_gma_aggregate.  Same K-step order, bias point and bf16 rounding points; the depth-wise sums are exact
    bf16 products added in fp32 in the matrix pipe's order instead of (dy, dx) fmaf order, so a value may land on the other side of a bf16 rounding
    boundary: the bar is >= 99.9 % of the values bit-equal and no value further than 2 bf16 ulps of the tensor's largest magnitude (measured at
    8 x 544 x 960: 3 685 of 8.0e8 values differ, largest 0.002; loc 155 of 6.7e7).  The pass-through segments, the per-channel maximum of k (a maximum of
    stored values) and run-to-run results are bitwise.  Ragged tiles, images smaller than a tile and than the 7x7 window, several tiles per block, an
    outlier token, with and without a qkv bias."""
    R = torch.ops.realcam
    torch.manual_seed(b * 1000 + hw[0])
    blk = M.GMA_Block(80, 8, qkv_bias=qkv_bias)
    with torch.no_grad():
        for p_ in blk.att.aggregator.parameters():
            p_.add_(0.1 * torch.randn_like(p_))
        for m_ in (blk.att.aggregator.norm0, blk.att.aggregator.norm1, blk.att.aggregator.norm2, blk.att.aggregator.norm3):
            m_.running_mean.normal_(0, 0.3); m_.running_var.uniform_(0.5, 2.0)
        blk.norm1.weight.normal_(1.0, 0.2); blk.norm1.bias.normal_(0, 0.2)
        if qkv_bias:
            blk.att.qkv.bias.normal_(0, 0.5)
    blk = blk.to("cuda", torch.bfloat16).eval()
    g = torch.Generator().manual_seed(21 + hw[1])
    x = torch.randn(b, *hw, 80, generator=g)
    x[b - 1, hw[0] // 2, hw[1] // 2] *= 25.0
    x = x.to("cuda", torch.bfloat16)
    with torch.no_grad():
        wq, bq = ops.packed_chain(blk.att.qkv)
        qkv = R.gma_ln_qkv(x, wq, bq, ops.f32_param(blk.norm1, "weight"), ops.f32_param(blk.norm1, "bias"), 1e-5)
        want = blk.att.aggregator._run_fused(qkv)
        got = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
        again = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
    torch.cuda.synchronize()
    for name, w_, g_, a_ in zip(("qkvp", "loc", "kmax"), want, got, again):
        assert w_.shape == g_.shape and w_.dtype == g_.dtype, name
        assert torch.equal(g_, a_), name
    qw, qg = want[0].float(), got[0].float()
    for seg in (0, 4, 8):                                             # pass-through groups of q, k, v: no depth-wise sum
        assert torch.equal(qw[seg], qg[seg]), seg
    for name, w_, g_ in (("qkvp", qw, qg), ("loc", want[1].float(), got[1].float())):
        diff = (w_ - g_).abs()
        assert (diff != 0).float().mean().item() <= 1e-3, (name, (diff != 0).float().mean().item())
        assert diff.max().item() <= 2.0 ** -7 * w_.abs().max().item(), (name, diff.max().item(), w_.abs().max().item())
    kw, kg = want[2], got[2]
    assert (kw - kg).abs().max().item() <= 2.0 ** -7 * kw.abs().max().item()
    k = got[0][4:8].float().permute(1, 2, 3, 0, 4).reshape(b, hw[0] * hw[1], 64)
    assert torch.equal(kg, k.amax(dim=1))                             # the maximum of the values this launch stored


@pytest.mark.gpu
def test_qkv_aggregate_is_bitwise_stable_under_load(hip):
    """rc_gma_qkv_aggregate keeps MFMA operands in registers that later loads recycle (Toeplitz fragments one segment ahead, A fragments per segment) and
    runs 16 waves a block with one barrier per segment: 24 launches on 4 x 272 x 480 tokens (510 tiles, two per block), every output compared with the
    first launch's.  (A sibling kernel written the same round, gma_in + ConvPosEnc in one launch, failed exactly this check -- single products of its
    1x1 convolution came out differently from run to run -- and was not shipped: profiles/r06_gma_stages.md.)"""
    torch.manual_seed(3)
    blk = M.GMA_Block(80, 8).to("cuda", torch.bfloat16).eval()
    x = torch.randn(4, 272, 480, 80, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        ref = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
        for i in range(24):
            out = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
            for name, r_, o_ in zip(("qkvp", "loc", "kmax"), ref, out):
                assert torch.equal(r_, o_), (i, name, int((r_.float() != o_.float()).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("c,hw", [(80, (37, 70)), (80, (16, 32)), (48, (130, 65)), (160, (9, 200))])
@pytest.mark.parametrize("identity", [True, False])
def test_depthwise3x3_segment_kernel_equals_the_general_one(hip, c, hw, identity):
    """rc_dwconv2d's bf16 3x3 single-rep case (ConvPosEnc) runs on the aggregator's 16-channel-segment core (csrc/gma_fused.hip,
    rc_debug_set("dw3_seg16")): same accumulation order as the general kernel -> the same bits, ragged tiles included; and both agree with
    an fp32 torch depth-wise convolution of the bf16 inputs to bf16 rounding."""
    g = torch.Generator().manual_seed(c + hw[0])
    x = torch.randn(2, *hw, c, generator=g).to("cuda", torch.bfloat16)
    w = (torch.randn(c, 1, 3, 3, generator=g) * 0.3).to("cuda")
    bias = torch.randn(c, generator=g).to("cuda")
    wT = ops.dw_taps(w)
    y1 = ops.dwconv2d(x, 0, (c,), 0, c, 3, wT, bias=bias, add_identity=identity)
    assert hip.rc_debug_set(b"dw3_seg16", 0) == 0
    try:
        y0 = ops.dwconv2d(x, 0, (c,), 0, c, 3, wT, bias=bias, add_identity=identity)
    finally:
        assert hip.rc_debug_set(b"dw3_seg16", 1) == 0
    assert torch.equal(y0, y1)
    xf = x.float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xf, w, bias, padding=1, groups=c) + (xf if identity else 0)
    assert rel_err(y1.float().permute(0, 3, 1, 2), ref) <= 6e-3
