"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the reference's golden
vectors.  Tolerances (stated here, reported as PSNR in bench.py):
  fp32: max|err| <= 2e-5 * max|ref| per block, end-to-end PSNR >= 100 dB
        (exact-f32 MFMA is an fmaf chain; ATen-CPU sums in another order -> ~1e-6 relative)
  bf16: bf16 storage / fp32 accumulate vs the fp32 CPU oracle: per block <= 3e-2 * max|ref|,
        end-to-end PSNR >= 55 dB (CPU bf16-vs-fp32 of the same net is ~61 dB)
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import torch.utils._python_dispatch

import liteisp_oracle as O
import realcamnet_amd as M
from realcamnet_amd import networks as N
from realcamnet_amd import ops
from conftest import net_name_of, golden_names, load_golden, rel_err, seed0_state_dict

pytestmark = pytest.mark.gpu

DEV = "cuda"
FP32_TOL, BF16_TOL = 2e-5, 3e-2
DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return FP32_TOL if dt == torch.float32 else BF16_TOL


def put(mod, sd, dt):
    mod.load_state_dict(sd, strict=True)
    return mod.to(device=DEV, dtype=dt).eval()


def run(mod, *xs, dt):
    with torch.no_grad():
        y = mod(*[x.to(DEV, dt) if isinstance(x, torch.Tensor) else x for x in xs])
    torch.cuda.synchronize()
    return y


def test_device_is_gfx950(hip):
    import ctypes
    buf = ctypes.create_string_buffer(64)
    assert hip.rc_device_arch(buf, 64) == 0
    assert buf.value.decode().startswith("gfx950"), buf.value


# ---- a1/a2 ingest ---------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
def test_bayer_unshuffle_bit_exact(hip, dt):
    g = torch.Generator().manual_seed(1)
    mosaic = torch.rand(3, 1, 42, 70, generator=g).to(dt)
    got = ops.bayer_unshuffle(mosaic.to(DEV), pad_to=16)
    torch.cuda.synchronize()
    ref, _ = O.pad_to_multiple(O.bayer_unshuffle(mosaic.float()), 16)
    assert got.shape == (3, 32, 48, 4)
    assert torch.equal(got.float().cpu().permute(0, 3, 1, 2), ref.to(dt).float())   # pure data movement: exact


def test_raw_ingest_vs_oracle(hip):
    """f4 ingest (rc_raw_ingest): black/white level + unshuffle + pad16 + bilinear cond resize in one launch, against the oracle
    (F.interpolate align_corners=False).  uint16 sensor counts and normalised floats; ragged size; up- and down-scaling cond."""
    g = torch.Generator().manual_seed(9)
    counts = torch.randint(60, 1024, (2, 1, 86, 140), generator=g, dtype=torch.int32).to(torch.uint16)
    want_p, want_c = O.raw_ingest(counts.to(torch.int32).float(), 64.0, 1023.0, cond_hw=(32, 48))
    want_p, _ = O.pad_to_multiple(want_p, 16)
    packed, cond = ops.raw_ingest(counts.to(DEV), dtype=torch.float32, pad_to=16, black_level=64.0, white_level=1023.0, cond_hw=(32, 48))
    torch.cuda.synchronize()
    assert packed.shape == (2, 48, 80, 4) and cond.shape == (2, 4, 32, 48)
    assert (packed.cpu().permute(0, 3, 1, 2) - want_p).abs().max() <= 1e-6
    assert (cond.cpu() - want_c).abs().max() <= 2e-6
    mosaic = torch.rand(1, 1, 40, 64, generator=g)
    want_p, want_c = O.raw_ingest(mosaic, cond_hw=(64, 96))            # up-scaling: taps clamp at the last row / column
    packed, cond = ops.raw_ingest(mosaic.to(DEV), pad_to=1, cond_hw=(64, 96))
    assert torch.equal(packed.cpu().permute(0, 3, 1, 2), want_p)        # black 0 / white 1: pure data movement
    assert (cond.cpu() - want_c).abs().max() <= 2e-6
    pb, cb = ops.raw_ingest(mosaic.to(DEV, torch.bfloat16), pad_to=1, cond_hw=(64, 96))
    assert pb.dtype == torch.bfloat16 and (cb.float().cpu() - want_c).abs().max() <= 1e-2
    # forward_mosaic(cond=None) == forward_mosaic(cond = the oracle's resized packed RAW) up to the resize's float noise
    net = net_on_gpu("LiteISPNet_GFM_LSC", torch.float32)
    mosaic = torch.rand(1, 1, 96, 128, generator=g)
    coord = O.make_coord(1, 48, 64)
    with torch.no_grad():
        y0 = net.forward_mosaic(mosaic.to(DEV), None, coord.to(DEV), cond_hw=(32, 32))
        ref = O.run_padded("LiteISPNet_GFM_LSC", seed0_state_dict("LiteISPNet_GFM_LSC"), O.bayer_unshuffle(mosaic),
                           O.raw_ingest(mosaic, cond_hw=(32, 32))[1], coord)
    assert O.psnr(y0.cpu(), ref) >= 100.0


def test_layout_roundtrip_and_pad(hip):
    x = torch.randn(2, 37, 13, 29)
    a = ops.to_nhwc(x.to(DEV), pad_hw=(16, 32))
    assert torch.equal(a[:, :13, :29].cpu(), x.permute(0, 2, 3, 1))
    assert a[:, 13:].abs().max() == 0 and a[:, :, 29:].abs().max() == 0
    assert torch.equal(ops.to_nchw(a, crop_hw=(13, 29)).cpu(), x)
    g = load_golden("block_pad16")
    x = torch.rand(*[int(v) for v in g["x_shape"]])
    yp, hw = M.LiteISP.pad_to_multiple_of_16(x.to(DEV))
    assert yp.shape == g["y"].shape and hw == tuple(int(v) for v in g["hw"])
    assert torch.equal(yp.cpu(), O.pad_to_multiple(x, 16)[0])


# ---- golden block fixtures (reference outputs) ------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
def test_dwt_blocks_vs_reference(hip, dt):
    g = load_golden("block_dwt_forward")
    y = run(put(N.DWTForward(16), g["sd"], dt), g["x"], dt=dt)
    assert rel_err(y.float().cpu(), g["y"]) <= (1e-6 if dt == torch.float32 else 1e-2)
    g = load_golden("block_dwt_inverse")
    y = run(put(N.DWTInverse(64), g["sd"], dt), g["x"], dt=dt)
    assert rel_err(y.float().cpu(), g["y"]) <= (1e-6 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_blocks_vs_reference(hip, dt):
    g = load_golden("block_conv3x3_16_32")                      # ragged 11x37 image, Cout=32 -> 2 cout tiles
    y = run(put(N.conv(16, 32, mode="C"), g["sd"], dt), g["x"], dt=dt)
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    g = load_golden("block_conv_crc_48")                        # conv-relu-conv at the net's 48 channels
    y = run(put(N.conv(48, 48, mode="CRC"), g["sd"], dt), g["x"], dt=dt)
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("schedule", ["early", "staged", "unfused"])
def test_channel_attention_blocks_vs_reference(hip, dt, schedule):
    """RCABlock / RCAGroup against the reference fixtures in all three schedules: `early` (default: gate of conv2's output computed ahead of conv2,
    applied in its epilogue), `staged` (gate + skip folded into the NEXT conv's input staging), `unfused` (rc_gate_residual pass)."""
    old = ops.FUSE_GATE, ops.EARLY_GATE
    ops.EARLY_GATE, ops.FUSE_GATE = schedule == "early", schedule != "unfused"
    try:
        g = load_golden("block_rcab_32")
        y = run(put(N.RCABlock(32, 32), g["sd"], dt), g["x"], dt=dt)
        assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
        g = load_golden("block_rcag_32_nb4")
        y = run(put(N.RCAGroup(32, 32, nb=4), g["sd"], dt), g["x"], dt=dt)
        assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
        g = load_golden("block_rcag_48_nb2")
        y = run(put(N.RCAGroup(48, 48, nb=2), g["sd"], dt), g["x"], dt=dt)
        assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    finally:
        ops.FUSE_GATE, ops.EARLY_GATE = old


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,H,W", [(48, 37, 70), (48, 1, 5), (32, 2, 2), (128, 24, 33), (64, 9, 1)])
def test_gate_ahead_equals_gate_of_the_conv_output(hip, dt, c, H, W):
    """rc_ca_gate_ahead (gate of conv2(t) from t's channel sums + border lines, by linearity of the mean) against rc_ca_gate on the channel sums
    conv2 itself emits: same MLP, two orders of summation -> fp32 rounding only (tolerance 2e-5 absolute on a sigmoid output); degenerate images
    (one row / one column / 2 x 2: border lines coincide) included."""
    g = torch.Generator().manual_seed(c + H * W)
    blk = N.RCABlock(c, c).to(DEV, dt).eval()
    with torch.no_grad():
        for p_ in blk.ca.parameters():
            p_.mul_(4.0)                                     # spread the gates away from 0.5
        x = torch.randn(2, H, W, c, generator=g).to(DEV, dt)
        t, sums_t = blk.res[0]._nhwc(x, act="relu", want_sums=True)
        r, sums_r = blk.res[2]._nhwc(t, want_sums=True)
        ref = ops.ca_gate(sums_r, H * W, blk.ca)
        ahead = ops.ca_gate_ahead(sums_t, t, blk.res[2], blk.ca)
    assert ahead.shape == ref.shape == (2, c)
    # bf16: conv1's channel sums S are taken from its fp32 accumulators BEFORE t is rounded, while the border lines E / corners K are read
    # from the rounded t that conv2 consumes -- so S, E, K do not describe exactly the same tensor and the gate is not the mean of the conv2 output
    # actually produced: O(2^-9) per term, largest relative to the total on TINY maps (the 1 x 5, 2 x 2 and 9 x 1 cases here; the codec's latents at
    # H * W ~ 64), where the border terms are most of the sum.  Stated tolerance on the sigmoid output: 5e-4 in bf16 (2e-5 in fp32: summation order only).
    assert (ahead - ref).abs().max().item() <= (2e-5 if dt == torch.float32 else 5e-4)
    assert ref.std().item() > 1e-3


@pytest.mark.parametrize("c", [32, 48, 64])
def test_wave_autonomous_kernels_equal_the_kernels_they_replace(hip, c):
    """Kernels 6 (32 / 48 channels) and 7 (64 channels) -- barrier-free waves with private halo strips, DESIGN 4.11 -- against the kernels they replace
    (`persist_auto` 0: kernel 2 / the general kernel): same unit map, same MFMA chain per pixel, so every operand form must match BIT FOR BIT on real-valued
    data, on ragged, multi-image and single-tile shapes (border strips take the bounds-checked load path; 130 x 260 has interior strips too).  The channel
    partial sums are split differently between the kernels (which tiles a wave carries): their totals agree to fp32 summation order."""
    lib = hip
    g = torch.Generator().manual_seed(7 * c)
    conv = N.Conv2d(c, c, 3, 1, 1).to(DEV, torch.bfloat16).eval()
    try:
        for (B, H, W) in ((1, 8, 32), (2, 23, 70), (3, 40, 97), (9, 17, 33), (1, 130, 260)):
            x = torch.randn(B, H, W, c, generator=g).to(DEV, torch.bfloat16)
            res = torch.randn(B, H, W, c, generator=g).to(DEV, torch.bfloat16)
            gate = torch.rand(B, c, generator=g).to(DEV)
            forms = {"plain": dict(), "relu": dict(act="relu"), "leaky": dict(act="leaky", slope=0.2), "sums": dict(want_sums=True),
                     "relu+sums": dict(act="relu", want_sums=True), "res": dict(residual=res), "gate+res": dict(residual=res, out_scale=gate)}
            if c == 64:
                forms["leaky+sums"] = dict(act="leaky", slope=0.01, want_sums=True)
            for name, kw in forms.items():
                outs = []
                for auto in (0, 2):
                    assert lib.rc_debug_set(b"persist_auto", auto) == 0
                    with torch.no_grad():
                        o = ops.conv2d(x, conv, **kw)
                    outs.append(o if isinstance(o, tuple) else (o,))
                torch.cuda.synchronize()
                assert torch.equal(outs[0][0], outs[1][0]), (c, B, H, W, name)
                if "sums" in name:
                    assert torch.allclose(outs[0][1].sum(1), outs[1][1].sum(1), rtol=1e-5, atol=1e-3), (c, B, H, W, name)
    finally:
        lib.rc_debug_set(b"persist_auto", 1)


def test_gma_in_and_conv_pos_enc_as_one_launch(hip):
    """rc_gma_in_cpe (ABI 14): the cfg3 net's gma_in (1x1 192 -> 80) + the block's ConvPosEnc in one launch.  Against the two launches on the same bf16 input: the 1x1 sums
    start from zero and add the bias last (rc_conv2d starts from the bias) and the depth-wise sums are formed in the matrix pipe's order, so equality is up to a few bf16 ulps of the
    intermediate on < 0.2 % of the values; run-to-run it must be BITWISE stable (the first build of this kernel was not: its 1x1 stage alternated two accumulators, each revisited
    one MFMA later -- DESIGN 4.7); ragged sizes; and the net's _refine_d1 must take it."""
    import realcamnet_amd.groupmix as G
    torch.manual_seed(3)
    blk = G.GMA_Block(80, 8).to(DEV, torch.bfloat16).eval()
    pre = N.Conv2d(192, 80, 1, 1, 0).to(DEV, torch.bfloat16).eval()
    assert ops.FUSE_GMA_ENTRY
    for (B, H, W) in ((1, 8, 32), (2, 23, 70), (1, 130, 260), (8, 136, 240)):
        d1 = torch.randn(B, H, W, 192, device=DEV).to(torch.bfloat16)
        with torch.no_grad():
            want = blk.cpe._nhwc(pre._nhwc(d1))
            got = blk._entry(d1, pre)
            again = [blk._entry(d1, pre) for _ in range(6)]
        torch.cuda.synchronize()
        assert got.shape == want.shape == (B, H, W, 80)
        diff = (got.float() - want.float()).abs()
        # a one-ulp difference of the intermediate `a` (|a| <~ 4: ulp 2^-6) reaches the output through the identity and nine taps: a few ulps of a, never more
        assert float(diff.max()) <= 2.0 ** -5 and rel_err(got.float().cpu(), want.float().cpu()) < 1e-2, (B, H, W, float(diff.max()))
        assert float((got != want).float().mean()) < 2e-3, (B, H, W, float((got != want).float().mean()))
        assert all(torch.equal(a_, got) for a_ in again), (B, H, W)
    d1 = torch.randn(2, 24, 64, 192, device=DEV).to(torch.bfloat16)
    post = N.Conv2d(80, 192, 1, 1, 0).to(DEV, torch.bfloat16).eval()
    outs = []
    for on in (True, False):
        ops.FUSE_GMA_ENTRY = on
        log = []
        try:
            with _OpLog(log), torch.no_grad():
                outs.append(blk._nhwc(d1, pre=pre, post=(post, d1)))
        finally:
            ops.FUSE_GMA_ENTRY = True
        assert sum(1 for n in log if "gma_in_cpe" in n) == (1 if on else 0), log
    assert rel_err(outs[0].float().cpu(), outs[1].float().cpu()) < 2e-2


class _OpLog(torch.utils._python_dispatch.TorchDispatchMode):
    """Names of the realcam:: ops dispatched inside the context."""

    def __init__(self, log):
        super().__init__()
        self.log = log

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name.startswith("realcam"):
            self.log.append(name)
        return func(*args, **(kwargs or {}))


@pytest.mark.parametrize("c", [48, 32])
def test_conv_then_haar_dwt_in_one_launch_equals_the_two_launches(hip, c):
    """RC_OUT_NHWC_DWT (ABI 14): conv [+ ReLU / LeakyReLU] -> networks.DWTForward as ONE launch of the wave-autonomous kernel -- the strip's outputs are rounded to
    bf16 as the NHWC store rounds them, parked in the wave's LDS strip and combined in dwt_forward_kernel's order with the reference's frozen taps
    (models/networks.py:224-235) -- must equal rc_conv2d + rc_dwt_forward BIT FOR BIT: single-tile, ragged (even) sizes whose last strips hang over the image,
    several images, and a map with interior strips; and the Sequential peephole (LiteISP `down1`: conv -> DWT, models/LiteISP.py:1950-1953) must take it."""
    g = torch.Generator().manual_seed(11 * c)
    conv = N.Conv2d(c, c, 3, 1, 1).to(DEV, torch.bfloat16).eval()
    dwt = N.DWTForward(c).to(DEV, torch.bfloat16).eval()
    assert ops.FUSE_DWT
    for (B, H, W) in ((1, 8, 32), (2, 22, 70), (3, 40, 98), (5, 18, 34), (1, 130, 260), (1, 2, 2)):
        x = torch.randn(B, H, W, c, generator=g).to(DEV, torch.bfloat16)
        for kw in (dict(), dict(act="relu"), dict(act="leaky", slope=0.1)):
            assert ops.conv_dwt_ok(x, conv, dwt, kw.get("act"), kw.get("slope", 0.0))
            with torch.no_grad():
                want = ops.dwt_forward(ops.conv2d(x, conv, **kw), dwt)
                got = ops.conv2d(x, conv, out_mode=ops.RC_OUT_NHWC_DWT, **kw)
            torch.cuda.synchronize()
            assert got.shape == (B, H // 2, W // 2, 4 * c) and torch.equal(got, want), (c, B, H, W, kw)
    # the executor: seq(conv, DWT) and seq(conv, ReLU, DWT) are one launch each; with the switch off, two -- the same bits
    x = torch.randn(2, 24, 64, c, generator=g).to(DEV, torch.bfloat16)
    for mods in ((conv, dwt), (conv, torch.nn.ReLU(), dwt)):
        sq = N.Sequential(*mods)
        outs = []
        for on in (True, False):
            ops.FUSE_DWT = on
            log = []
            try:
                with _OpLog(log), torch.no_grad():
                    outs.append(sq._nhwc(x))
            finally:
                ops.FUSE_DWT = True
            assert sum(1 for n in log if "haar_dwt" in n) == (0 if on else 1) and sum(1 for n in log if "conv2d" in n) == 1, log
        assert torch.equal(outs[0], outs[1])
    # + residual (an RCAGroup's closing conv + group skip in front of the DWT, LiteISP down2): kernel 6's residual mode, the same bits as the two launches
    for (B, H, W) in ((2, 22, 70), (1, 130, 260), (3, 8, 32)):
        x = torch.randn(B, H, W, c, generator=g).to(DEV, torch.bfloat16)
        res = torch.randn(B, H, W, c, generator=g).to(DEV, torch.bfloat16)
        assert ops.conv_dwt_ok(x, conv, dwt, residual=True) and not ops.conv_dwt_ok(x, conv, dwt, act="relu", residual=True)
        with torch.no_grad():
            want = ops.dwt_forward(ops.conv2d(x, conv, residual=res), dwt)
            got = ops.conv2d(x, conv, residual=res, out_mode=ops.RC_OUT_NHWC_DWT)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (c, B, H, W, "residual")
    grp = N.RCAGroup(c, c, nb=2).to(DEV, torch.bfloat16).eval()
    x = torch.randn(2, 24, 64, c, generator=g).to(DEV, torch.bfloat16)
    sq = N.Sequential(grp, dwt)
    outs = []
    for on in (True, False):
        ops.FUSE_DWT = on
        log = []
        try:
            with _OpLog(log), torch.no_grad():
                outs.append(sq._nhwc(x))
        finally:
            ops.FUSE_DWT = True
        assert sum(1 for n in log if "haar_dwt" in n) == (0 if on else 1), log
    assert torch.equal(outs[0], outs[1])
    # odd sizes and other layers have no such form: the peephole declines, the C ABI says so
    assert not ops.conv_dwt_ok(torch.zeros(1, 9, 32, c, device=DEV, dtype=torch.bfloat16), conv, dwt)
    assert not ops.conv_dwt_ok(torch.zeros(1, 8, 32, c, device=DEV), conv, dwt)
    c64 = N.Conv2d(64, 64, 3, 1, 1).to(DEV, torch.bfloat16).eval()
    with pytest.raises(RuntimeError, match="RC_OUT_NHWC_DWT"):
        ops.conv2d(torch.zeros(1, 8, 32, 64, device=DEV, dtype=torch.bfloat16), c64, out_mode=ops.RC_OUT_NHWC_DWT)
    bad = N.DWTForward(c).to(DEV, torch.bfloat16).eval()
    with torch.no_grad():
        bad.weight.mul_(2.0)
    assert not ops.conv_dwt_ok(x, conv, bad)


@pytest.mark.parametrize("cin,cout,shape", [(128, 128, (1, 135, 240)), (128, 128, (1, 68, 120)), (64, 128, (2, 40, 70)), (128, 256, (1, 33, 65)), (512, 128, (1, 17, 30))])
def test_fp32_small_maps_take_narrow_cout_tiles_with_the_same_bits(hip, cin, cout, shape):
    """fp32 3x3 layers on maps too small to fill the chip with 64-wide cout tiles run with 16-wide ones (rc_conv_desc.cout_tile, ops.small_map_cout_tile: cfg2's 128-channel
    levels at 1080p, B = 1).  The tile width changes which block computes a channel, not the order its products are summed in: BIT-identical to the automatic width, in every
    operand form the fp32 nets use, exact against F.conv2d on integer data, planar store included."""
    b, H, W = shape
    g = torch.Generator().manual_seed(cin + cout + H)
    conv = N.Conv2d(cin, cout, 3, 1, 1).to(DEV).eval()
    x = torch.randn(b, H, W, cin, generator=g).to(DEV)
    res = torch.randn(b, H, W, cout, generator=g).to(DEV)
    assert ops.SMALL_MAP_COUT_TILE and ops.small_map_cout_tile(x, conv, ops.RC_OUT_NHWC) == 16
    def run():
        with torch.no_grad():
            y, sums = ops.conv2d(x, conv, act="relu", want_sums=True)
            return (ops.conv2d(x, conv), y, sums.sum(1), ops.conv2d(x, conv, residual=res, act="leaky", slope=0.2), ops.conv2d(x, conv, out_mode=ops.RC_OUT_NCHW))
    narrow = run()
    ops.SMALL_MAP_COUT_TILE = False
    try:
        assert ops.small_map_cout_tile(x, conv, ops.RC_OUT_NHWC) == 0
        wide = run()
    finally:
        ops.SMALL_MAP_COUT_TILE = True
    for a, c in zip(narrow, wide):
        assert torch.equal(a, c)
    with torch.no_grad():
        conv.weight.copy_(torch.randint(-2, 3, conv.weight.shape, generator=g).float())
        conv.bias.copy_(torch.randint(-3, 4, (cout,), generator=g).float())
        ops.invalidate_caches(conv)
        xi = torch.randint(-3, 4, (b, cin, H, W), generator=g).float()
        ref = F.conv2d(xi, conv.weight.cpu(), conv.bias.cpu(), padding=1)
        y = ops.conv2d(ops.to_nhwc(xi.to(DEV)), conv, out_mode=ops.RC_OUT_NCHW)
    assert torch.equal(y.cpu(), ref)


def test_thin_stage_kernel_equals_kernel_4(hip):
    """Kernel 4b (one barrier per stage, weights and tile by LDS-DMA two stages ahead, dense LDS pixels) against kernel 4 (`thin` 0) on the
    layers it takes over: folded stride-2 3x3 (its structurally-zero taps skipped in both), 3x3 with a 16-wide cout tile over several chunks, 16-channel
    chunks, the un-folded 2x2 window.  Same work decomposition and accumulation order: BIT FOR BIT, on border-only, ragged and interior-tile shapes,
    several images, with activation / residual / channel sums."""
    lib = hip
    assert lib.rc_debug_set(b"lds_poison", 1) == 0            # every launch starts from LDS full of NaNs: nothing may depend on what an earlier kernel left there
    g = torch.Generator().manual_seed(411)
    cases = []
    for (cin, cout) in ((64, 64), (128, 128), (128, 192)):
        conv = N.Conv2d(cin, cout, 3, 2, 1).to(DEV, torch.bfloat16).eval()
        cases.append((f"fold {cin}->{cout}", cin, lambda x, conv=conv: (ops.conv_stride2(x, conv), ops.conv_stride2(x, conv, act="leaky", slope=0.1))))
    # (16, 128) / (16, 200): one chunk, several cout tiles;  (128, 128) .. (192, 48): form 2 (64- and 48-wide cout tiles over 32-channel chunks: `thin` 2)
    for (cin, cout) in ((128, 12), (64, 16), (112, 64), (96, 32), (16, 128), (144, 16), (48, 32), (16, 200), (128, 128), (192, 192), (128, 64), (192, 48), (32, 128)):
        conv = N.Conv2d(cin, cout, 3, 1, 1).to(DEV, torch.bfloat16).eval()
        def run(x, conv=conv, cout=cout):
            res = torch.randn(*x.shape[:3], cout, generator=torch.Generator().manual_seed(5)).to(DEV, torch.bfloat16)
            y, sums = ops.conv2d(x, conv, act="relu", want_sums=True)
            return (ops.conv2d(x, conv), y, sums.sum(1), ops.conv2d(x, conv, residual=res), ops.conv2d(x, conv, act="gelu"))      # GELU: the generic epilogue
        cases.append((f"3x3 {cin}->{cout}", cin, run))
    w2 = ops._ConvView((torch.randn(32, 64, 2, 2, generator=g) * 0.1).to(DEV, torch.bfloat16), torch.randn(32, generator=g).to(DEV, torch.bfloat16))
    cases.append(("2x2 64->32", 64, lambda x: (ops.conv2d(x, w2), ops.conv2d(x, w2, act="relu"))))
    try:
        for name, cin, fn in cases:
            for (B, H, W) in ((1, 9, 20), (2, 37, 71), (3, 64, 96), (1, 150, 230)):
                x = torch.randn(B, H, W, cin, generator=g).to(DEV, torch.bfloat16)
                outs = []
                for thin in (2, 0):
                    assert lib.rc_debug_set(b"thin", thin) == 0
                    with torch.no_grad():
                        outs.append(fn(x))
                torch.cuda.synchronize()
                for a, b in zip(*outs):
                    assert torch.equal(a, b), (name, B, H, W)
    finally:
        lib.rc_debug_set(b"thin", 2); lib.rc_debug_set(b"lds_poison", 0)


@pytest.mark.parametrize("c,B,H,W,auto", [(48, 2, 632, 256, 1), (48, 2, 632, 256, 0), (48, 3, 256, 1024, 1), (64, 2, 632, 256, 1), (32, 2, 256, 1024, 1)])
def test_compact_channel_sum_slots_equal_the_per_tile_layout(hip, c, B, H, W, auto):
    """The carried-sums kernels (2, 6, 7) write ONE partial-sum slot per (residue class of their tile walk, wave) instead of 4 per 8 x 32 tile
    (rc_conv_sum_slots; 2 048 instead of 32 640 per image at 4K, no zero stores per tile).  On integer data every partial sum is exact, so the totals of the
    two layouts are EQUAL; 632 rows = 79 tile rows: kernel 6's waves 4-7 have no strip in the last 16-row band, so some of their residue classes are empty in
    an image and must be zero-filled.  The gate computed from either layout is the same, and frame 1 of a batch gives the slots of frame 1 alone, bitwise."""
    lib = hip
    g = torch.Generator().manual_seed(c + H)
    conv = N.Conv2d(c, c, 3, 1, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randint(-2, 3, conv.weight.shape, generator=g).float() / 2)
        conv.bias.copy_(torch.randint(-2, 3, conv.bias.shape, generator=g).float())
    x = (torch.randint(-2, 3, (B, H, W, c), generator=g).float() / 2).to(DEV, torch.bfloat16)
    conv = conv.to(DEV, torch.bfloat16).eval()
    res = {}
    try:
        assert lib.rc_debug_set(b"persist_auto", auto) == 0
        for compact in (1, 0):
            assert lib.rc_debug_set(b"sums_compact", compact) == 0
            with torch.no_grad():
                y, sums = ops.conv2d(x, conv, act="relu", want_sums=True)
                y1, sums1 = ops.conv2d(x[1:2].contiguous(), conv, act="relu", want_sums=True)
            torch.cuda.synchronize()
            res[compact] = (y, sums, sums1)
    finally:
        lib.rc_debug_set(b"sums_compact", 1); lib.rc_debug_set(b"persist_auto", 1)
    legacy_n = lib.rc_conv_sum_tiles(H, W)
    assert res[0][1].shape[1] == legacy_n and res[1][1].shape[1] < legacy_n            # the compact layout really is in use, and smaller
    assert torch.equal(res[0][0], res[1][0])
    ref = res[1][0].float().sum(dim=(1, 2))
    assert torch.equal(res[1][1].sum(1), ref) and torch.equal(res[0][1].sum(1), ref)      # integers: exact, whatever the split
    assert torch.equal(res[1][2][0], res[1][1][1])                                        # frame 1 alone == frame 1 of the batch, slot for slot


@pytest.mark.parametrize("persist", [1, 2, 3, 0])
@pytest.mark.parametrize("dt,cin,H,W", [(torch.bfloat16, 48, 16, 40), (torch.bfloat16, 48, 9, 33), (torch.bfloat16, 128, 21, 70), (torch.float32, 32, 9, 33),
                                        (torch.bfloat16, 32, 21, 70), (torch.bfloat16, 64, 9, 33)])
def test_conv_out_scale_and_relu_sums_exact_on_integer_data(hip, persist, dt, cin, H, W):
    """The two epilogues of the early-gate RCAB on integer data, every launch form: conv + ReLU with channel sums (sums == the stored map's sums
    exactly), and conv * out_scale[b][c] + residual (power-of-two scales: exact) -- bit for bit against F.conv2d."""
    g = torch.Generator().manual_seed(cin * 7 + H)
    c = N.Conv2d(cin, cin, 3, 1, 1)
    with torch.no_grad():
        c.weight.copy_(torch.randint(-2, 3, c.weight.shape, generator=g).float() / 2)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-2, 3, (2, cin, H, W), generator=g).float() / 2
    res = torch.randint(-4, 5, (2, cin, H, W), generator=g).float() / 2
    scale = 2.0 ** torch.randint(-2, 2, (2, cin), generator=g).float()
    ref = F.conv2d(x, c.weight.detach(), c.bias.detach(), padding=1)
    rb = (lambda t: t.bfloat16().float()) if dt == torch.bfloat16 else (lambda t: t)
    c = c.to(DEV, dt).eval()
    a, r = ops.to_nhwc(x.to(DEV, dt)), ops.to_nhwc(res.to(DEV, dt))
    assert hip.rc_debug_set(b"persist", persist) == 0
    try:
        with torch.no_grad():
            y1, sums = ops.conv2d(a, c, act="relu", want_sums=True)
            y2 = ops.conv2d(a, c, out_scale=scale.to(DEV), residual=r)
    finally:
        hip.rc_debug_set(b"persist", 1)
    relu = ref.clamp_min(0)
    assert torch.equal(y1.float().cpu().permute(0, 3, 1, 2), rb(relu))
    assert torch.equal(sums.sum(1).cpu(), relu.sum((2, 3)))                  # small integers: every partial sum is exact in fp32
    assert torch.equal(y2.float().cpu().permute(0, 3, 1, 2), rb(ref * scale[:, :, None, None] + res))


@pytest.mark.parametrize("dt", DTYPES)
def test_standalone_calayer_vs_reference(hip, dt):
    """networks.CALayer.forward on its own (upstream models/networks.py:255-270): rc_channel_sums -> rc_ca_gate -> scale."""
    g = load_golden("block_calayer_32")
    mod = put(N.CALayer(32, 16), g["sd"], dt)
    y = run(mod, g["x"], dt=dt)
    y2 = run(mod, g["x"], dt=dt)
    assert torch.equal(y, y2)
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    # many slots per image (fixed-order fold), odd pixel count, vs the oracle
    x = torch.randn(2, 32, 97, 131, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = O.ca_layer({"ca." + k: v for k, v in g["sd"].items()}, "ca", x)
    assert rel_err(run(mod, x, dt=dt).float().cpu(), ref) <= tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_conditioning_blocks_vs_reference(hip, dt):
    g = load_golden("block_res_gfm_48")
    mod = put(M.LiteISP.Res_GFM(48, 48, 32, 48, 48), g["sd"], dt)
    with torch.no_grad():
        y, _ = mod((g["x"].to(DEV, dt), g["v"].to(DEV)))
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    g = load_golden("block_lsc_48")
    y = run(put(M.LiteISP.Lens_Shading_Correction(2, 48, 48), g["sd"], dt), g["x"], dt=dt)
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    g = load_golden("block_color_condition")
    mod = put(M.LiteISP.Color_Condition_GFM(4, 32), g["sd"], dt)
    with torch.no_grad():
        v = mod._vec(g["x"].to(DEV, dt))
    assert rel_err(v.cpu(), g["y"]) <= (1e-4 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("dt", DTYPES)
def test_gfm_lfm_blocks_vs_reference(hip, dt):
    """The global + local modulation blocks of ISPUNet_GFM_LFM (upstream models/LiteISP.py:215-230, 293-321, 501-534, 601-620) on the
    reference's own outputs."""
    L = M.LiteISP
    g = load_golden("block_res_gfm_lfm_64")
    mod = put(L.Res_GFM_LFM(cond_c=32, out_nc=64, nf=128), g["sd"], dt)
    with torch.no_grad():
        y, v, cm = mod((g["x"].to(DEV, dt), g["v"].to(DEV), g["cmap"].to(DEV, dt)))
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    g = load_golden("block_sftlayer_32")
    mod = put(L.SFTLayer(32, 32, 32), g["sd"], dt)
    with torch.no_grad():
        y = mod((g["x"].to(DEV, dt), g["cmap"].to(DEV, dt)))
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    g = load_golden("block_gfmlayer_128")
    mod = put(L.GFMLayer(32, 128, 256), g["sd"], dt)
    with torch.no_grad():
        y = mod((g["x"].to(DEV, dt), g["v"].to(DEV)))
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)
    g = load_golden("block_color_condition_gfm_lfm")
    mod = put(L.Color_Condition_GFM_LFM(4, 32, 32), g["sd"], dt)
    with torch.no_grad():
        vec, lfm = mod(g["x"].to(DEV, dt), g["local"].to(DEV, dt))
    assert vec.shape == (2, 32, 1, 1) and lfm.shape == g["lfm"].shape
    assert rel_err(vec.float().cpu().flatten(1), g["y"]) <= (1e-4 if dt == torch.float32 else 3e-2)
    assert rel_err(lfm.float().cpu(), g["lfm"]) <= tol(dt)
    g = load_golden("block_cb_4_16")
    mod = put(L.CB(4, 16, True), g["sd"], torch.float32)
    with torch.no_grad():
        y = mod(g["x"].to(DEV))
    assert rel_err(y.cpu(), g["y"]) <= 1e-4


@pytest.mark.parametrize("dt", DTYPES)
def test_tail_pixel_shuffle_vs_reference(hip, dt):
    g = load_golden("block_tail_16")
    tail = N.seq(N.conv(16, 64, mode="C"), torch.nn.PixelShuffle(2), N.conv(16, 3, mode="C"))
    tail = put(tail, g["sd"], dt)
    with torch.no_grad():
        a = ops.to_nhwc(g["x"].to(DEV, dt))
        t = tail[0]._nhwc(a, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)
        y = tail[2]._nhwc(t, out_mode=ops.RC_OUT_NCHW)
    assert rel_err(y.float().cpu(), g["y"]) <= tol(dt)


def test_tail_fold_vs_the_reference_tail_output(hip):
    """The reference's own tail output (block_tail_16: conv(16, 64) -> PixelShuffle(2) -> conv(16, 3), upstream models/LiteISP.py:2379-2383) against the
    FOLDED form (ops.tail_fold: one 5x5 convolution 16 -> 12 + the exact border ring) in fp32, where C = 16 has a 5x5 instantiation: the fold differs from
    the two convolutions by the fp32 rounding of the composed weights only."""
    g = load_golden("block_tail_16")
    tail = N.seq(N.conv(16, 64, mode="C"), torch.nn.PixelShuffle(2), N.conv(16, 3, mode="C"))
    tail = put(tail, g["sd"], torch.float32)
    with torch.no_grad():
        a = ops.to_nhwc(g["x"].to(DEV))
        assert ops.tail_fold_ok(a, tail[0], tail[2])
        y = ops.tail_fold(a, tail[0], tail[2])
        two = tail[2]._nhwc(tail[0]._nhwc(a, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2), out_mode=ops.RC_OUT_NCHW)
    assert y.shape == g["y"].shape
    assert rel_err(y.cpu(), g["y"]) <= tol(torch.float32)
    assert rel_err(y.cpu(), two.cpu()) <= tol(torch.float32)
    ring = torch.ones_like(y, dtype=torch.bool)
    ring[..., 1:-1, 1:-1] = False
    assert rel_err(y[ring].cpu(), two[ring].cpu()) <= 2e-6   # the outermost ring IS the two convolutions, recomputed on four strips (the side strips transposed: another fp32 summation order)


@pytest.mark.parametrize("dt", DTYPES)
def test_dwt_any_channel_count_pair_vs_reference(hip, dt):
    """networks.DWTForward_ / DWTInverse_ (upstream models/networks.py:9-47) on the reference's fixtures; the state_dict is the single (4,1,2,2) tap set."""
    g = load_golden("block_dwt_forward_anyc")
    fwd = put(N.DWTForward_(), g["sd"], dt)
    assert rel_err(run(fwd, g["x"], dt=dt).float().cpu(), g["y"]) <= tol(dt)
    gi = load_golden("block_dwt_inverse_anyc")
    inv = put(N.DWTInverse_(), gi["sd"], dt)
    assert rel_err(run(inv, gi["x"], dt=dt).float().cpu(), gi["y"]) <= tol(dt)
    assert list(fwd.state_dict()) == ["weight"] and tuple(fwd.weight.shape) == (4, 1, 2, 2)
    x = torch.randn(1, 16, 8, 12, generator=torch.Generator().manual_seed(1))
    back = run(inv, run(fwd, x, dt=torch.float32) if dt == torch.float32 else run(fwd, x, dt=dt), dt=dt)
    assert rel_err(back.float().cpu(), x) <= (1e-6 if dt == torch.float32 else 2e-2)


# ---- end to end ---------------------------------------------------------------------------------------
_NETS = {}


def net_on_gpu(name, dt):
    key = (name, dt)
    if key not in _NETS:
        net = getattr(M, name)()
        net.load_state_dict(seed0_state_dict(name), strict=True)
        _NETS[key] = net.to(device=DEV, dtype=dt).eval()
    return _NETS[key]


@pytest.mark.parametrize("fixture", golden_names("e2e_"))
@pytest.mark.parametrize("dt", DTYPES)
def test_end_to_end_vs_reference_golden(hip, fixture, dt):
    g = load_golden(fixture)
    name = net_name_of(fixture)
    net = net_on_gpu(name, dt)
    with torch.no_grad():
        y = net([g["raw"].to(DEV, dt), g["cond"].to(DEV, dt), g["coord"].to(DEV, dt)])
    torch.cuda.synchronize()
    assert y.shape == g["y"].shape and y.dtype == dt
    p = O.psnr(y.float().cpu(), g["y"])
    assert p >= (100.0 if dt == torch.float32 else 55.0), p        # fp32 >= 100 dB, bf16 >= 55 dB


@pytest.mark.parametrize("name", ["LiteISPNet", "LiteISPNet_GFM_LSC", "ISPUNet_GFM_LSC"])
def test_batch_and_ragged_mosaic_vs_oracle(hip, name):
    """B=3 frames of a 2*(43x61)... mosaic: exercises unshuffle+pad-to-16, partial tiles, batch indexing, crop."""
    g = torch.Generator().manual_seed(7)
    h, w = 44, 70                                   # packed size; pads to 48x80
    mosaic = torch.rand(3, 1, 2 * h, 2 * w, generator=g)
    cond = torch.rand(3, 4, 32, 48, generator=g)
    coord = O.make_coord(3, h, w)
    sd = seed0_state_dict(name)
    with torch.no_grad():
        ref = O.run_padded(name, sd, O.bayer_unshuffle(mosaic), cond, coord)
        y = net_on_gpu(name, torch.float32).forward_mosaic(mosaic.to(DEV), cond.to(DEV), coord.to(DEV))
    torch.cuda.synchronize()
    assert y.shape == ref.shape == (3, 3, 2 * h, 2 * w)
    assert O.psnr(y.cpu(), ref) >= 100.0
    # frames are independent: frame 1 alone gives the same bits as frame 1 inside the batch
    with torch.no_grad():
        y1 = net_on_gpu(name, torch.float32).forward_mosaic(mosaic[1:2].to(DEV), cond[1:2].to(DEV), coord[1:2].to(DEV))
    assert torch.equal(y1[0], y[1])


def test_run_to_run_bitwise_stable(hip):
    g = load_golden("e2e_LiteISPNet_GFM_LSC_64x64")
    net = net_on_gpu("LiteISPNet_GFM_LSC", torch.bfloat16)
    x = [g["raw"].to(DEV, torch.bfloat16), g["cond"].to(DEV, torch.bfloat16), g["coord"].to(DEV, torch.bfloat16)]
    with torch.no_grad():
        a = net(x).clone()
        b = net(x)
    assert torch.equal(a, b)


# ---- size-independent properties at a full 4K frame ----------------------------------------------------
def test_full_size_properties_4k(hip):
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    mosaic = torch.rand(1, 1, 2160, 3840, generator=g).to(DEV, dt)
    a = ops.bayer_unshuffle(mosaic, pad_to=16)
    assert a.shape == (1, 1088, 1920, 4)
    assert torch.equal(a[0, :1080].permute(2, 0, 1), F.pixel_unshuffle(mosaic, 2)[0])   # vs torch's own unshuffle, exact
    assert a[0, 1080:].abs().max() == 0
    # Haar analysis -> synthesis is the identity (taps are orthonormal): exact in fp32 up to rounding
    x = torch.rand(1, 1088, 1920, 48, generator=torch.Generator(device=DEV).manual_seed(5), device=DEV)
    fwd, inv = N.DWTForward(48).to(DEV), N.DWTInverse(192).to(DEV)
    back = inv._nhwc(fwd._nhwc(x))
    assert (back - x).abs().max().item() <= 1e-6
    # conv linearity at full size: conv(2x) - bias == 2*(conv(x) - bias)
    c = N.conv(48, 48, mode="C").to(DEV)
    with torch.no_grad():
        c.bias.zero_()
        y1 = c._nhwc(x)
        y2 = c._nhwc(x * 2)
    assert torch.equal(y2, y1 * 2)           # exact: scaling by 2 commutes with every fp32 rounding
    # a full 4K frame runs end to end, output finite and cropped to the mosaic size
    net = net_on_gpu("LiteISPNet_GFM_LSC", dt)
    with torch.no_grad():
        y = net.forward_mosaic(mosaic, torch.rand(1, 4, 256, 256, device=DEV).to(dt), O.make_coord(1, 1080, 1920).to(DEV, dt))
    torch.cuda.synchronize()
    assert y.shape == (1, 3, 2160, 3840) and torch.isfinite(y.float()).all()


def test_shape_errors_are_raised(hip):
    net = net_on_gpu("LiteISPNet", torch.float32)
    with pytest.raises(ValueError, match="multiples of 8"):
        net([torch.zeros(1, 4, 20, 24, device=DEV)])
    with pytest.raises(ValueError):
        ops.dwt_forward(torch.zeros(1, 5, 6, 8, device=DEV), N.DWTForward(8).to(DEV))
    with pytest.raises(ValueError):
        ops.conv2d(torch.zeros(1, 8, 8, 12, device=DEV), N.conv(16, 16, mode="C").to(DEV))


def test_persistent_kernel_equals_general_kernel(hip):
    """The weights-resident persistent conv (single chunk, single cout tile) and the general kernel are the
    same arithmetic in the same order: outputs, stored inputs and channel sums must match bit for bit."""
    g = torch.Generator().manual_seed(11)
    for dt in DTYPES:
        for (cin, cout, hw) in ((48, 48, (40, 100)), (48, 3, (24, 64)), (4, 48, (16, 40))):
            c = N.conv(cin, cout, mode="C").to(DEV, dt)
            x = torch.randn(3, hw[0], hw[1], cin, generator=g).to(DEV, dt)
            skip = torch.randn(3, hw[0], hw[1], cin, generator=g).to(DEV, dt)
            gate = torch.rand(3, cin, generator=g).to(DEV)
            outs = []
            for persist in (1, 2, 3, 0):
                assert hip.rc_debug_set(b"persist", persist) == 0
                if cout == 3:
                    r = (ops.conv2d(x, c, out_mode=ops.RC_OUT_NCHW, crop_hw=(hw[0] - 3, hw[1] - 5)),)
                else:
                    r = ops.conv2d(x, c, act="relu", gate=gate, skip=skip, store_input=True, want_sums=True)
                torch.cuda.synchronize()
                outs.append([t.clone() for t in r])
            hip.rc_debug_set(b"persist", 1)
            for other in outs[1:]:
                for a_, b_ in zip(outs[0], other):
                    assert torch.equal(a_, b_)


@pytest.mark.parametrize("dt", DTYPES)
def test_dwt_with_arbitrary_per_channel_taps(hip, dt):
    """The DWT taps are state_dict parameters; a checkpoint with non-Haar, per-channel taps must be honoured."""
    g = torch.Generator().manual_seed(5)
    fwd, inv = N.DWTForward(16), N.DWTInverse(64)
    with torch.no_grad():
        fwd.weight.copy_(torch.randn(fwd.weight.shape, generator=g))
        inv.weight.copy_(torch.randn(inv.weight.shape, generator=g))
    x = torch.randn(2, 16, 12, 20, generator=g)
    ref_f = O.dwt_forward({"w.weight": fwd.weight.detach()}, "w", x)
    ref_i = O.dwt_inverse({"w.weight": inv.weight.detach()}, "w", ref_f)
    t = 1e-6 if dt == torch.float32 else 2e-2
    yf = run(fwd.to(DEV, dt).eval(), x, dt=dt)
    assert rel_err(yf.float().cpu(), ref_f) <= t
    yi = run(inv.to(DEV, dt).eval(), ref_f, dt=dt)
    assert rel_err(yi.float().cpu(), ref_i) <= t


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("persist", [1, 2, 3, 0])
@pytest.mark.parametrize("shape", [(48, 48, 16, 40, 3), (48, 48, 9, 33, 3), (48, 3, 8, 32, 3), (4, 48, 24, 70, 3), (32, 32, 21, 70, 3), (32, 64, 9, 33, 3), (32, 3, 16, 40, 3),
                                   (64, 64, 8, 40, 3), (96, 48, 8, 32, 3), (48, 48, 8, 32, 1), (2, 48, 10, 34, 1)])
def test_conv_exact_on_small_integer_data(hip, dt, persist, shape):
    """Small-integer weights/inputs make every product and partial sum exactly representable, so the HIP conv
    must equal F.conv2d BIT FOR BIT in bf16 and fp32, on ragged multi-tile images too.  (Regression: an
    out-of-bounds sentinel that wrapped back into the image corrupted pixel (0,0) on ragged tiles.)"""
    cin, cout, h, w, ksz = shape
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    c = N.Conv2d(cin, cout, ksz, 1, ksz // 2)
    with torch.no_grad():
        c.weight.copy_(torch.randint(-2, 3, c.weight.shape, generator=g).float() / 2)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-2, 3, (2, cin, h, w), generator=g).float() / 2
    ref = F.conv2d(x, c.weight.detach(), c.bias.detach(), padding=ksz // 2)
    assert hip.rc_debug_set(b"persist", persist) == 0
    try:
        with torch.no_grad():
            y = c.to(DEV, dt)(x.to(DEV, dt)).float().cpu()
    finally:
        hip.rc_debug_set(b"persist", 1)
    assert torch.equal(y, ref)


def _pair_case(hip, H, W, b, mode, seed):
    """Run conv -> act -> conv fused (rc_conv_pair) and as two rc_conv2d launches on the same data."""
    g = torch.Generator().manual_seed(seed)
    dt = torch.bfloat16
    c1, c2 = N.Conv2d(48, 48, 3, 1, 1), N.Conv2d(48, 48, 3, 1, 1)
    for c in (c1, c2):
        c.weight.data = torch.randn(c.weight.shape, generator=g) * 0.05
        c.bias.data = torch.randn(48, generator=g) * 0.1
        c.to(DEV, dt).eval()
    x = (torch.randn(b, H, W, 48, generator=g)).to(DEV, dt)
    kw1, kw2 = {}, {}
    if mode == "gated":
        kw1 = dict(gate=torch.rand(b, 48, generator=g).to(DEV), skip=torch.randn(b, H, W, 48, generator=g).to(DEV, dt), store_input=True)
    if mode == "film":
        film = (torch.randn(b, 48, generator=g).to(DEV) * 0.5, torch.randn(b, 48, generator=g).to(DEV) * 0.5)
        fused = ops.conv_pair(x, c1, c2, act="leaky", slope=0.01, film=film, residual=x)
        t = ops.conv2d(x, c1, act="leaky", slope=0.01, film=film)
        return (fused,), (ops.conv2d(t, c2, residual=x),)
    sums = mode != "plain"
    fused = ops.conv_pair(x, c1, c2, act="relu", want_sums=sums, **kw1)
    t = ops.conv2d(x, c1, act="relu", **kw1)
    stored = None
    if isinstance(t, tuple):
        t, stored = t
    ref = ops.conv2d(t, c2, want_sums=sums)
    fused = fused if isinstance(fused, tuple) else (fused,)
    ref = ref if isinstance(ref, tuple) else (ref,)
    if stored is not None:
        ref = (ref[0], stored, *ref[1:])
    return fused, ref


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "sums", "gated", "film"])
@pytest.mark.parametrize("shape", [(5, 7, 1), (37, 70, 2), (64, 96, 1), (8, 32, 3), (90, 200, 1)])
def test_conv_pair_equals_two_convs(hip, mode, shape, impl):
    """The fused pair keeps the bf16 intermediate in LDS: same unit map, same accumulation order, same rounding
    points as two rc_conv2d launches, so the feature maps are bit-identical (ragged, multi-tile, border and
    interior tiles); the channel partial sums differ only in how they are split, so their totals agree."""
    H, W, b = shape
    assert hip.rc_debug_set(b"pair_impl", impl) == 0         # 1: weights in LDS (8 + 4 waves); 2: weights in registers, two teams (pair2)
    try:
        fused, ref = _pair_case(hip, H, W, b, mode, seed=H * 131 + W)
    finally:
        hip.rc_debug_set(b"pair_impl", 0)
    assert torch.equal(fused[0], ref[0])
    if mode == "gated":
        assert torch.equal(fused[1], ref[1])
    if mode in ("sums", "gated"):
        fs, rs = fused[-1].double().sum(1), ref[-1].double().sum(1)
        assert torch.allclose(fs, rs, rtol=1e-5, atol=1e-3)


@pytest.mark.gpu
def test_conv_pair_vs_fp32_reference(hip):
    """Against plain fp32 PyTorch on the CPU (bf16 tolerance 3e-2 of max|ref|, as for the single layers)."""
    g = torch.Generator().manual_seed(7)
    c1, c2 = N.Conv2d(48, 48, 3, 1, 1), N.Conv2d(48, 48, 3, 1, 1)
    x = torch.randn(2, 48, 40, 72, generator=g)
    with torch.no_grad():
        ref = F.conv2d(F.relu(F.conv2d(x, c1.weight, c1.bias, padding=1)), c2.weight, c2.bias, padding=1)
    c1.to(DEV, torch.bfloat16); c2.to(DEV, torch.bfloat16)
    out = ops.conv_pair(ops.to_nhwc(x.to(DEV, torch.bfloat16)), c1, c2, act="relu")
    assert rel_err(ops.to_nchw(out).float().cpu(), ref) < 3e-2


@pytest.mark.gpu
def test_conv_pair_rejects_other_shapes(hip):
    c = N.Conv2d(64, 64, 3, 1, 1).to(DEV, torch.bfloat16)
    x = torch.zeros(1, 8, 32, 64, device=DEV, dtype=torch.bfloat16)
    old, ops.FUSE_PAIR = ops.FUSE_PAIR, True
    try:
        assert not ops.conv_pair_ok(x, c, c)
    finally:
        ops.FUSE_PAIR = old
    with pytest.raises(RuntimeError, match="48 channels"):
        ops.conv_pair(x, c, c)


def test_end_to_end_with_fused_pairs_vs_reference_golden(hip):
    """The optional rc_conv_pair path (ops.FUSE_PAIR) through the whole flagship net: same golden, same bar."""
    fixture = [f for f in golden_names("e2e_LiteISPNet_GFM_LSC")][0]
    g = load_golden(fixture)
    net = net_on_gpu("LiteISPNet_GFM_LSC", torch.bfloat16)
    dt = torch.bfloat16
    old, ops.FUSE_PAIR = ops.FUSE_PAIR, True
    try:
        with torch.no_grad():
            y = net([g["raw"].to(DEV, dt), g["cond"].to(DEV, dt), g["coord"].to(DEV, dt)])
        torch.cuda.synchronize()
    finally:
        ops.FUSE_PAIR = old
    assert O.psnr(y.float().cpu(), g["y"]) >= 55.0


@pytest.mark.parametrize("persist", [1, 0])
@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("shape", [(128, 64, 16, 40), (192, 192, 9, 33), (512, 128, 8, 32), (192, 48, 24, 70), (128, 128, 37, 100)])
def test_multichunk_conv_exact_on_integer_data(hip, persist, gated, shape):
    """Cin = several 32-channel chunks (the 128/192/512-channel U-Net levels): the producer/consumer kernel that
    streams input and weight chunks through double-buffered LDS (persist=1) and the general kernel (persist=0) must
    both equal F.conv2d bit for bit on integer data (all sums < 256: exact in bf16), with and without the fused
    CALayer gate + skip input, on ragged multi-tile images."""
    cin, cout, h, w = shape
    g = torch.Generator().manual_seed(cin * 7 + cout)
    c = N.Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        sparse = (torch.rand(c.weight.shape, generator=g) < 96.0 / cin).float()     # keeps every sum below 256
        c.weight.copy_(torch.randint(-1, 2, c.weight.shape, generator=g).float() * sparse)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-1, 2, (2, cin, h, w), generator=g).float()
    kw, xin = {}, x
    if gated:   # x = r * gate + skip with gate in {0, 1, 2}: still small integers
        r = torch.randint(-1, 2, (2, cin, h, w), generator=g).float()
        gate = torch.randint(0, 2, (2, cin), generator=g).float()
        xin = r * gate[:, :, None, None] + x
    ref = F.conv2d(xin, c.weight.detach(), c.bias.detach(), padding=1)
    assert ref.abs().max() <= 256
    assert hip.rc_debug_set(b"persist", persist) == 0
    try:
        c = c.to(DEV, torch.bfloat16)
        with torch.no_grad():
            if gated:
                y, stored = ops.conv2d(ops.to_nhwc(r.to(DEV, torch.bfloat16)), c, gate=gate.to(DEV),
                                       skip=ops.to_nhwc(x.to(DEV, torch.bfloat16)), store_input=True)
                assert torch.equal(ops.to_nchw(stored).float().cpu(), xin)
            else:
                y = ops.conv2d(ops.to_nhwc(x.to(DEV, torch.bfloat16)), c)
            y = ops.to_nchw(y).float().cpu()
    finally:
        hip.rc_debug_set(b"persist", 1)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("act", [None, "relu", "leaky"])
@pytest.mark.parametrize("shape", [(1, 16, 32), (2, 37, 70), (1, 130, 200), (3, 48, 96)])
def test_tail_conv_with_lds_staged_pixel_shuffle_exact(hip, act, shape):
    """Kernel 5 (csrc/conv_kernel.hpp, rc_debug_set("pss", 1): the 48 -> 192 + PixelShuffle(2) layer with its output staged through LDS and
    stored by the loader waves; off by default, it ties with kernel 4): bit-exact against F.conv2d + pixel_shuffle on integer data -- one
    tile, ragged edges, several tiles per block, fused ReLU / LeakyReLU -- and bit-identical to the default kernel on real-valued data."""
    b, h, w = shape
    g = torch.Generator().manual_seed(h * 3 + w)
    c = N.Conv2d(48, 192, 3, 1, 1)
    with torch.no_grad():
        c.weight.copy_(torch.randint(-1, 2, c.weight.shape, generator=g).float())
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-1, 2, (b, 48, h, w), generator=g).float()
    ref = F.conv2d(x, c.weight.detach(), c.bias.detach(), padding=1)
    kw = {}
    if act == "relu":
        ref = ref.relu(); kw = dict(act="relu")
    elif act == "leaky":
        ref = torch.where(ref > 0, ref, ref * 0.5); kw = dict(act="leaky", slope=0.5)
    ref = F.pixel_shuffle(ref, 2)
    assert ref.abs().max() <= 512                                    # half-integers up to 512 are exact in bf16
    c = c.to(DEV, torch.bfloat16)
    xd = ops.to_nhwc(x.to(DEV, torch.bfloat16))
    xr = torch.randn(b, h, w, 48, generator=g).to(DEV, torch.bfloat16)
    with torch.no_grad():
        y0, r0 = (ops.conv2d(t, c, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2, **kw) for t in (xd, xr))
        assert hip.rc_debug_set(b"pss", 1) == 0
        try:
            y1, r1 = (ops.conv2d(t, c, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2, **kw) for t in (xd, xr))
        finally:
            assert hip.rc_debug_set(b"pss", 0) == 0
    assert torch.equal(ops.to_nchw(y1).float().cpu(), ref)
    assert torch.equal(y0, y1) and torch.equal(r0, r1)


@pytest.mark.parametrize("knob", [1, 2, 3])
@pytest.mark.parametrize("mode", ["plain", "gated", "res", "sums", "film", "ps"])
@pytest.mark.parametrize("shape", [(128, 64, 16, 40), (192, 192, 9, 33), (128, 128, 37, 100), (48, 192, 21, 70)])
def test_conv32_forms_exact_on_integer_data(hip, knob, mode, shape):
    """The 32x32x16 conv forms (csrc/conv32_kernel.hpp; rc_debug_set("conv32"): 1 = staged-output form on 16-channel chunks + the one-chunk
    48-channel form, 2 / 3 = two-barrier form with 4 / 8 compute waves) equal F.conv2d BIT FOR BIT on integer data in every operand form:
    gated input + materialised sum, residual, CALayer partial sums, FiLM + LeakyReLU, PixelShuffle store; ragged multi-tile images."""
    cin, cout, h, w = shape
    if cin == 48 and knob != 1:
        pytest.skip("the one-chunk 48-channel form has one variant")
    g = torch.Generator().manual_seed(cin * 7 + cout)
    c = N.Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        sparse = (torch.rand(c.weight.shape, generator=g) < 96.0 / cin).float()
        c.weight.copy_(torch.randint(-1, 2, c.weight.shape, generator=g).float() * sparse)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-1, 2, (2, cin, h, w), generator=g).float()
    xin, kw = x, {}
    if mode == "gated":
        r = torch.randint(-1, 2, (2, cin, h, w), generator=g).float()
        gate = torch.randint(0, 2, (2, cin), generator=g).float()
        xin = r * gate[:, :, None, None] + x
    ref = F.conv2d(xin, c.weight.detach(), c.bias.detach(), padding=1)
    if mode == "res":
        res = torch.randint(-2, 3, (2, cout, h, w), generator=g).float()
        ref = ref + res
    if mode == "film":
        fs = torch.randint(0, 2, (2, cout), generator=g).float(); ft = torch.randint(-1, 2, (2, cout), generator=g).float()
        ref = ref * fs[:, :, None, None] + ft[:, :, None, None] + ref
        ref = torch.where(ref > 0, ref, ref * 0.5)
        kw = dict(act="leaky", slope=0.5, film=(fs.to(DEV), ft.to(DEV)))
    if mode == "ps":
        ref = F.pixel_shuffle(ref, 2); kw = dict(out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)
    assert ref.abs().max() <= 256
    assert hip.rc_debug_set(b"conv32", knob) == 0
    try:
        c = c.to(DEV, torch.bfloat16)
        xd = ops.to_nhwc(x.to(DEV, torch.bfloat16))
        with torch.no_grad():
            if mode == "gated":
                y, stored = ops.conv2d(ops.to_nhwc(r.to(DEV, torch.bfloat16)), c, gate=gate.to(DEV), skip=xd, store_input=True)
                assert torch.equal(ops.to_nchw(stored).float().cpu(), xin)
            elif mode == "res":
                y = ops.conv2d(xd, c, residual=ops.to_nhwc(res.to(DEV, torch.bfloat16)))
            elif mode == "sums":
                y, sums = ops.conv2d(xd, c, want_sums=True)
                assert torch.equal(sums.float().reshape(2, -1, cout).sum(1).cpu(), ref.sum(dim=(2, 3)))
            else:
                y = ops.conv2d(xd, c, **kw)
            y = ops.to_nchw(y).float().cpu()
    finally:
        hip.rc_debug_set(b"conv32", 0)
    assert torch.equal(y, ref)



@pytest.mark.parametrize("shape", [(1, 5, 7), (2, 37, 70), (1, 64, 96)])
def test_lens_shading_chain_equals_layer_by_layer(hip, shape):
    """rc_pointwise_chain48 (all four 1x1 layers in one launch, activations in LDS) against the same module run as
    four rc_conv2d launches: same bf16 rounding points; layer 0 is plain FMAs instead of an MFMA, so allow 1 bf16
    ulp-scale differences (2e-2 of max|ref|), and check both against the fp32 CPU reference."""
    b, H, W = shape
    g = torch.Generator().manual_seed(H * 17 + W)
    lsc = M.LiteISP.Lens_Shading_Correction(in_channels=2, out_c=48, nf=48)
    x = torch.rand(b, 2, H, W, generator=g) * 2 - 1
    with torch.no_grad():
        ref = x
        for i, m in enumerate(lsc.model):
            ref = F.conv2d(ref, m.weight, m.bias) if isinstance(m, torch.nn.Conv2d) else F.leaky_relu(ref, m.negative_slope)
    lsc = lsc.to(DEV, torch.bfloat16).eval()
    xin = ops.to_nhwc(x.to(DEV, torch.bfloat16))
    outs = {}
    for fuse in (True, False):
        old, ops.FUSE_CHAIN = ops.FUSE_CHAIN, fuse
        try:
            with torch.no_grad():
                outs[fuse] = ops.to_nchw(lsc._nhwc(xin)).float().cpu()
        finally:
            ops.FUSE_CHAIN = old
    assert rel_err(outs[True], outs[False]) < 2e-2
    assert rel_err(outs[True], ref) < 3e-2 and rel_err(outs[False], ref) < 3e-2


@pytest.mark.parametrize("width", [32, 48, 64, 128])
@pytest.mark.parametrize("shape", [(1, 5, 7), (2, 37, 70), (3, 64, 96), (1, 16, 200)])
def test_lsc_chain_in_registers_vs_layer_by_layer(hip, shape, width):
    """rc_lsc_chain: the lens-shading chain with register-resident activations, alone and with the convolution it modulates folded in
    (head(raw) * (lsc(coord) + 1) in one launch), against the layer-by-layer launches (same bf16 rounding points, different MFMA k-order:
    bf16-ulp scale differences) and the fp32 CPU composition.  Borders, 64-pixel groups that straddle rows and images, batch > 1."""
    b, H, W = shape
    g = torch.Generator().manual_seed(H * 31 + W)
    torch.manual_seed(3)
    lsc = M.LiteISP.Lens_Shading_Correction(in_channels=2, out_c=width, nf=width)
    head = N.Conv2d(4, width, 3, 1, 1)
    raw, coord = torch.rand(b, 4, H, W, generator=g), torch.rand(b, 2, H, W, generator=g) * 2 - 1
    with torch.no_grad():
        t = coord
        for m in lsc.model:
            t = F.conv2d(t, m.weight, m.bias) if isinstance(m, torch.nn.Conv2d) else F.leaky_relu(t, m.negative_slope)
        ref = F.conv2d(raw, head.weight, head.bias, padding=1) * (t + 1)
    lsc, head = lsc.to(DEV, torch.bfloat16).eval(), head.to(DEV, torch.bfloat16).eval()
    a, c = ops.to_nhwc(raw.to(DEV, torch.bfloat16)), ops.to_nhwc(coord.to(DEV, torch.bfloat16))
    with torch.no_grad():
        chain = ops.lsc_chain(lsc, c)
        fused = ops.lsc_chain(lsc, c, head, a)
        old, ops.FUSE_CHAIN = ops.FUSE_CHAIN, False
        try:
            layered = lsc._nhwc(c)
            two = head._nhwc(a, mul_plus1=layered)
        finally:
            ops.FUSE_CHAIN = old
    assert chain is not None and chain.shape == two.shape == (b, H, W, width)
    f = lambda v: ops.to_nchw(v).float().cpu()
    assert rel_err(f(chain), f(layered)) < 2e-2 and rel_err(f(chain), t) < 3e-2
    if width == 128:        # no fused head at the codec's width (it spilled 29 registers, and the codec returns the lens-shading map itself): two launches
        assert fused is None
        return
    assert fused is not None and fused.shape == (b, H, W, width)
    assert rel_err(f(fused), f(two)) < 2e-2 and rel_err(f(fused), ref) < 3e-2


def test_bench_prints_one_contract_json_line(hip):
    """bench.py on a tiny workload: exactly one JSON line on stdout with the driver's contract keys, the roofline object
    and (with the CPU leg on) the cpu_baseline object."""
    import json, subprocess
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "1", "--height", "128", "--width", "192",
                        "--steps", "2", "--warmup", "1", "--model", "LiteISPNet_GFM_LSC"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "MP/s" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and "workload" in d["config"] and d["value"] > 0
    rf = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf)
    # what bounds what: the layer shape with the most time, with both of its fractions, and SURVEY 8(d)'s algorithmic FLOPs beside the executed ones
    assert set(("dominant_kernel", "dominant_ms_per_step", "dominant_tflops", "dominant_mfma_frac", "dominant_hbm_TBps", "dominant_hbm_frac_of_8TBps",
                "dominant_bound", "by_shape_top4", "flops_algorithmic", "frac_algorithmic_whole_step")) <= set(rf)
    assert rf["dominant_bound"] in ("hbm", "mfma") and 0 < rf["dominant_mfma_frac"] < 1 and 0 < rf["dominant_hbm_frac_of_8TBps"] < 1
    hp, wp = -(-64 // 16) * 16, -(-96 // 16) * 16                       # the 128 x 192 mosaic's packed RAW padded to multiples of 16
    assert rf["flops_algorithmic"] == 622084.0 * (2 * hp) * (2 * wp) and rf["flops_per_step"] < rf["flops_algorithmic"]      # the folded tail executes fewer
    assert all(not isinstance(v, (list, dict)) for k, v in rf.items() if k.startswith("dominant") or k in ("by_shape_top4", "flops_algorithmic"))   # flat: the driver's parser keeps them
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert d["psnr_db_vs_cpu_fp32"] >= 55.0


def test_bench_under_torchrun_gathers_over_rccl(hip):
    """One rank under torch.distributed.run: the process group is RCCL ("nccl"), every step's frames go through
    OverlappedGather (side stream, all_gather_into_tensor) and the gathered payload equals the local output."""
    import json, subprocess
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "2", "--height", "256", "--width", "384",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert d["n_gpus"] == 1 and "all_gather" in d["config"]["collective"] and d["value"] > 0


def test_bench_codec_leg_prints_one_contract_json_line(hip):
    """bench.py --model raw_compression_tcm_final on a 512x512 mosaic: same contract keys; the latent before rounding must agree
    with the CPU oracle (downstream of round() isolated flips are expected in bf16, so x_hat is only required to be sane)."""
    import json, subprocess
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "1", "--height", "512", "--width", "512", "--steps", "1",
                        "--warmup", "1", "--model", "raw_compression_tcm_final"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "MP/s" and d["value"] > 0 and "raw_compression_tcm_final" in d["config"]["workload"]
    assert d["cpu_baseline"]["kind"] == "port" and d["psnr_db_vs_cpu_fp32"]["y (latent, before rounding)"] >= 55.0
    assert d["psnr_db_vs_cpu_fp32"]["x_hat"] >= 41.0


def test_codec_forward_replayed_as_a_hip_graph_equals_the_eager_forward(hip):
    """realcamnet_amd.GraphedCall: raw_compression_tcm_final.forward_mosaic captured per input signature and replayed -- every tensor of the
    result dict bit-identical to the eager run, with the slice loop's two-stream forks captured as graph branches and with one stream; a second input through the
    same graph; a new shape gets its own capture."""
    import realcamnet_amd as M
    torch.manual_seed(0)
    net = M.raw2bit.raw_compression_tcm_final().eval().to(DEV, torch.bfloat16)
    g = torch.Generator().manual_seed(11)

    def inputs(h2, w2):
        return (torch.rand(1, 1, h2, w2, generator=g).to(DEV, torch.bfloat16), ops.make_coord(1, h2 // 2, w2 // 2, device=DEV, dtype=torch.bfloat16))

    def flat(d, pre=""):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from flat(v, pre + k + ".")
            else:
                yield pre + k, v

    fwd = lambda m, c: net.forward_mosaic(m, None, c)
    for fork in (True, False):
        ops.GRAPH_FORK = fork
        try:
            gf = M.GraphedCall(fwd)
            for h2, w2 in ((512, 768), (512, 768), (768, 512)):
                m, c = inputs(h2, w2)
                with torch.no_grad():
                    ref = {k: v.clone() for k, v in flat(fwd(m, c))}
                out = dict(flat(gf(m, c)))
                torch.cuda.synchronize()
                assert set(out) == set(ref)
                for k in ref:
                    assert torch.equal(out[k], ref[k]), (fork, k)
            assert len(gf._graphs) == 2
        finally:
            ops.GRAPH_FORK = True


def test_forward_is_hip_graph_capturable(hip):
    """Every op launches on the current stream, allocates through the caching allocator and never syncs with the host, so a
    whole forward can be captured in a HIP graph; the replay is bit-identical to the eager run."""
    net = net_on_gpu("LiteISPNet_GFM_LSC", torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    mosaic = torch.rand(2, 1, 96, 160, generator=g).to(DEV, torch.bfloat16)
    cond = torch.rand(2, 4, 32, 32, generator=g).to(DEV, torch.bfloat16)
    coord = O.make_coord(2, 48, 80).to(DEV, torch.bfloat16)

    def fwd():
        with torch.no_grad():
            return net.forward_mosaic(mosaic, cond, coord)
    for _ in range(2):
        y_eager = fwd()                                            # warm-up: packs weights, sets launch attributes
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y_graph = fwd()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_graph, y_eager)


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 192), (192, 320), (320, 128), (64, 3)])
@pytest.mark.parametrize("shape", [(1, 16, 24), (2, 37, 71), (1, 130, 200)])
def test_stride2_conv_reading_its_own_input_exact_on_integer_data(hip, cin, cout, shape):
    """conv3x3(stride 2) with 64 | channels: the 2x2-window kernel gathers the space-to-depth channels while staging (rc_conv_desc.src_h /
    src_w, no rc_space_to_depth2 launch).  Bit-exact against F.conv2d on small-integer data -- interior tiles, border tiles, odd source
    sizes (the phase-1 row / column beyond the edge reads zero) -- and bit-identical to the route through the map, with a fused activation."""
    b, H, W = shape
    g = torch.Generator().manual_seed(cin * 5 + cout + H)
    conv = N.Conv2d(cin, cout, 3, 2, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randint(-2, 3, conv.weight.shape, generator=g).float())
        conv.bias.copy_(torch.randint(-4, 5, (cout,), generator=g).float())
    x = torch.randint(-3, 4, (b, cin, H, W), generator=g).float()
    ref = F.conv2d(x, conv.weight, conv.bias, stride=2, padding=1)
    conv = conv.to(DEV, torch.bfloat16).eval()
    a = ops.to_nhwc(x.to(DEV, torch.bfloat16))
    assert ops.FOLD_STRIDE2
    with torch.no_grad():
        y = ops.conv_stride2(a, conv)
        ya = ops.conv_stride2(a, conv, act="leaky", slope=0.25)
        ops.FOLD_STRIDE2 = False
        try:
            y_map = ops.conv_stride2(a, conv)
            ya_map = ops.conv_stride2(a, conv, act="leaky", slope=0.25)
        finally:
            ops.FOLD_STRIDE2 = True
    assert torch.equal(ops.to_nchw(y).float().cpu(), ref.bfloat16().float())
    assert torch.equal(y, y_map) and torch.equal(ya, ya_map)
    xr = torch.randn(b, H, W, cin, generator=g).to(DEV, torch.bfloat16)                 # real-valued data: same accumulation order either way
    with torch.no_grad():
        y1 = ops.conv_stride2(xr, conv)
        ops.FOLD_STRIDE2 = False
        try:
            y2 = ops.conv_stride2(xr, conv)
        finally:
            ops.FOLD_STRIDE2 = True
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("cin,cout", [(16, 64), (128, 128), (32, 48), (128, 320), (80, 64)])
@pytest.mark.parametrize("shape", [(1, 16, 24), (2, 38, 70)])
def test_stride2_conv_as_2x2_window_exact_on_integer_data(hip, cin, cout, shape):
    """conv3x3(stride 2) in bf16 = rc_space_to_depth2 + rc_conv2d with ksize 2 (the {-1, 0}^2 window: 9 of 16 (tap, phase) blocks non-zero;
    single-chunk, multi-chunk and odd-width plans): bit-exact against F.conv2d on small-integer data, even and odd map sizes."""
    b, H, W = shape
    g = torch.Generator().manual_seed(cin * 7 + cout + H)
    conv = N.Conv2d(cin, cout, 3, 2, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randint(-2, 3, conv.weight.shape, generator=g).float())
        conv.bias.copy_(torch.randint(-4, 5, (cout,), generator=g).float())
    x = torch.randint(-3, 4, (b, cin, H, W), generator=g).float()
    ref = F.conv2d(x, conv.weight, conv.bias, stride=2, padding=1)
    conv = conv.to(DEV, torch.bfloat16).eval()
    with torch.no_grad():
        y = ops.to_nchw(conv._nhwc(ops.to_nhwc(x.to(DEV, torch.bfloat16))))
    assert y.shape == ref.shape
    assert torch.equal(y.float().cpu(), ref.bfloat16().float())     # integer sums are exact in the fp32 accumulators; one rounding to bf16
