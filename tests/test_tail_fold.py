"""The folded tail: conv 3x3 C -> 4C, PixelShuffle(2), conv 3x3 C -> 3 (upstream models/LiteISP.py:1996-2000, no activation in between) as ONE 5x5
convolution C -> 12 plus an exact border ring (include/realcam_hip.h, rc_tail_fold_weights / rc_tail_ring_*; realcamnet_amd/ops.py tail_fold).

CPU part: the host-side composition against torch's own convolutions in double, and the ring scheme (which strips, which rows / columns of their
results) emulated with F.conv2d -- it must reproduce the two-convolution result EXACTLY, crop included.  GPU part (-m gpu): the 5x5 kernel bit-exact
on integer data in every launch form and store mode, and ops.tail_fold against an fp64 reference with the ring checked separately."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from realcamnet_amd import _lib
from realcamnet_amd._lib import RC_BF16, RC_F32, RC_OUT_NCHW, RC_OUT_NHWC, RC_OUT_PIXEL_SHUFFLE2, RC_OUT_PIXEL_SHUFFLE2_NCHW


def _fold(w1, b1, w2, b2):
    lib = _lib.load()
    c, o = w1.shape[1], w2.shape[0]
    arrs = [None if t is None else np.ascontiguousarray(t.numpy(), np.float32) for t in (w1, b1, w2, b2)]
    wc, bc = np.empty((4 * o, c, 5, 5), np.float32), np.empty((4 * o,), np.float32)
    ptr = lambda a: None if a is None else a.ctypes.data
    assert lib.rc_tail_fold_weights(ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), ptr(arrs[3]), c, o, wc.ctypes.data, bc.ctypes.data) == 0
    return torch.from_numpy(wc), torch.from_numpy(bc)


def _two_step(x, w1, b1, w2, b2):
    return F.conv2d(F.pixel_shuffle(F.conv2d(x, w1, b1, padding=1), 2), w2, b2, padding=1)


@pytest.mark.parametrize("c,o,bias", [(48, 3, True), (16, 3, False), (8, 4, True)])
def test_folded_weights_equal_the_composition_away_from_the_border(c, o, bias):
    g = torch.Generator().manual_seed(c + o)
    w1, w2 = torch.randn(4 * c, c, 3, 3, generator=g) * 0.05, torch.randn(o, c, 3, 3, generator=g) * 0.05
    b1, b2 = (torch.randn(4 * c, generator=g), torch.randn(o, generator=g)) if bias else (None, None)
    wc, bc = _fold(w1, b1, w2, b2)
    x = torch.randn(2, c, 11, 14, generator=g, dtype=torch.float64)
    d = lambda t: None if t is None else t.double()
    ref = _two_step(x, d(w1), d(b1), d(w2), d(b2))
    one = F.pixel_shuffle(F.conv2d(x, wc.double(), bc.double(), padding=2), 2)
    assert (one - ref)[:, :, 1:-1, 1:-1].abs().max() <= 2e-6          # fp32 rounding of the folded weights
    assert (one - ref).abs().max() > 1e-3                             # ... and the outermost ring really differs: that is what the strips are for
    # each sub-pixel uses a 4x4 subset of the 5x5 taps
    assert float((wc == 0).double().mean()) == pytest.approx(0.36)


@pytest.mark.parametrize("H,W,crop", [(6, 9, None), (8, 8, (15, 16)), (5, 7, (10, 13)), (2, 2, None), (3, 2, (5, 4))])
def test_ring_scheme_reproduces_the_two_convolutions_exactly(H, W, crop):
    """ops.tail_fold's algorithm with F.conv2d standing in for the kernels: folded 5x5 everywhere, then output row 0 / 2H-1 from the two-conv result
    on x[:, 0:2] / x[:, H-2:H] (rows 0 / 3) and column 0 / 2W-1 from x[:, :, 0:2] / x[:, :, W-2:W] run transposed (rows 0 / 3 of that result), skipping what the crop removes."""
    g = torch.Generator().manual_seed(H * 100 + W)
    c, o, B = 8, 3, 2
    w1, w2 = torch.randn(4 * c, c, 3, 3, generator=g, dtype=torch.float64), torch.randn(o, c, 3, 3, generator=g, dtype=torch.float64)
    b1, b2 = torch.randn(4 * c, generator=g, dtype=torch.float64), torch.randn(o, generator=g, dtype=torch.float64)
    x = torch.randn(B, c, H, W, generator=g, dtype=torch.float64)
    oh, ow = crop if crop else (2 * H, 2 * W)
    ref = _two_step(x, w1, b1, w2, b2)[:, :, :oh, :ow]
    # the composition in double (the C function rounds to fp32; its equality with this is the previous test)
    wc = torch.zeros(4 * o, c, 5, 5, dtype=torch.float64); bc = b2.repeat_interleave(4).clone()
    for i in range(2):
        for j in range(2):
            for dy in range(3):
                for dx in range(3):
                    a, b = i + dy - 1, j + dx - 1
                    fy, ii, fx, jj = a // 2, a % 2, b // 2, b % 2
                    m = torch.arange(c) * 4 + 2 * ii + jj
                    rows = torch.arange(o) * 4 + 2 * i + j
                    wc[rows, :, fy + 1:fy + 4, fx + 1:fx + 4] += torch.einsum("oc,ckyx->okyx", w2[:, :, dy, dx], w1[m])
                    bc[rows] += w2[:, :, dy, dx] @ b1[m]
    out = F.pixel_shuffle(F.conv2d(x, wc, bc, padding=2), 2)[:, :, :oh, :ow].clone()
    rows = _two_step(torch.cat([x[:, :, 0:2], x[:, :, H - 2:H]], 0), w1, b1, w2, b2)           # (2B, o, 4, 2W)
    # the side strips run TRANSPOSED (2 x H images): ky <-> kx swapped weights, conv1's sub-pixel order 4c + 2i + j <-> 4c + 2j + i (ops._folded_tail)
    perm = (torch.arange(c)[:, None] * 4 + torch.tensor([0, 2, 1, 3])[None, :]).reshape(-1)
    w1t, b1t, w2t = w1.transpose(2, 3)[perm], b1[perm], w2.transpose(2, 3)
    cols_t = _two_step(torch.cat([x[:, :, :, 0:2], x[:, :, :, W - 2:W]], 0).transpose(2, 3), w1t, b1t, w2t, b2)     # (2B, o, 4, 2H)
    out[:, :, 0, :] = rows[:B, :, 0, :ow]
    if oh == 2 * H:
        out[:, :, oh - 1, :] = rows[B:, :, 3, :ow]
    y1 = oh - 1 if oh == 2 * H else oh            # the corner pixels belong to the row strips (one writer per pixel: rc_tail_ring_scatter)
    out[:, :, 1:y1, 0] = cols_t[:B, :, 0, 1:y1]
    if ow == 2 * W:
        out[:, :, 1:y1, ow - 1] = cols_t[B:, :, 3, 1:y1]
    assert (out - ref).abs().max() <= 1e-12


def test_5x5_plan_is_reported_and_bounded():
    lib = _lib.load()
    assert lib.rc_conv_packed_bytes(48, 12, 5, RC_BF16, RC_OUT_PIXEL_SHUFFLE2_NCHW) == 38 * 1024       # 25 + 13 MFMA steps of one 16-row tile
    assert lib.rc_conv_packed_cout(48, 12, 5, RC_BF16, RC_OUT_PIXEL_SHUFFLE2_NCHW) == 16
    assert lib.rc_conv_packed_bytes(48, 48, 5, RC_BF16, RC_OUT_NHWC) == 0                              # one cout tile only
    assert lib.rc_conv_packed_bytes(64, 12, 5, RC_BF16, RC_OUT_NHWC) == 2 * 25 * 1024                  # 64 channels: two 32-channel chunks of 25 steps
    assert lib.rc_conv_packed_bytes(32, 12, 5, RC_BF16, RC_OUT_NHWC) == 25 * 1024
    assert lib.rc_conv_packed_bytes(64, 12, 5, RC_F32, RC_OUT_NHWC) == 4 * 25 * 1024                   # fp32: 16-channel chunks
    assert lib.rc_conv_packed_bytes(40, 12, 5, RC_BF16, RC_OUT_NHWC) == 0 and lib.rc_conv_packed_bytes(24, 12, 5, RC_F32, RC_OUT_NHWC) == 0
    assert lib.rc_conv_packed_bytes(48, 10, 3, RC_BF16, RC_OUT_PIXEL_SHUFFLE2_NCHW) == 0               # pixel shuffle needs cout % 4 == 0
    assert lib.rc_tail_fold_weights(None, None, None, None, 48, 3, None, None) < 0
    assert lib.rc_tail_ring_gather(None, None, None, RC_BF16, 1, 8, 8, 48, None) < 0
    assert lib.rc_tail_ring_scatter(None, None, None, RC_BF16, 1, 3, 8, 8, 16, 16, None) < 0


# ---------------------------------------------------------------------------------------------------------------- GPU
DEV = "cuda"


@pytest.mark.gpu
@pytest.mark.parametrize("persist", [1, 2, 3, 0])
@pytest.mark.parametrize("cin,dt,cout,H,W", [(48, torch.bfloat16, 12, 16, 40), (48, torch.bfloat16, 16, 9, 33), (48, torch.bfloat16, 12, 37, 70),
                                             (48, torch.bfloat16, 4, 8, 32), (32, torch.bfloat16, 12, 21, 40), (64, torch.bfloat16, 12, 9, 70),
                                             (64, torch.float32, 12, 21, 40), (16, torch.float32, 8, 9, 33)])
def test_conv5x5_exact_on_small_integer_data(hip, persist, cin, dt, cout, H, W):
    """Integer-valued data: every product and partial sum is exact, so the 5x5 kernel must equal F.conv2d bit for bit -- in NHWC, NCHW and the
    pixel-shuffled planar store (cropped), in every launch form (persistent / producer-consumer / general) and instantiation (bf16 48- and 32-channel
    chunks, fp32 16-channel chunks)."""
    from realcamnet_amd import networks as N, ops
    g = torch.Generator().manual_seed(cout * 1000 + H)
    c = N.Conv2d(cin, cout, 5, 1, 2)
    with torch.no_grad():
        c.weight.copy_(torch.randint(-2, 3, c.weight.shape, generator=g).float() / 2)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-2, 3, (2, cin, H, W), generator=g).float() / 2
    ref = F.conv2d(x, c.weight.detach(), c.bias.detach(), padding=2)
    c = c.to(DEV, dt).eval()
    a = ops.to_nhwc(x.to(DEV, dt))
    assert hip.rc_debug_set(b"persist", persist) == 0
    try:
        with torch.no_grad():
            y_nhwc = ops.conv2d(a, c)
            y_nchw = ops.conv2d(a, c, out_mode=RC_OUT_NCHW, crop_hw=(H - 1, W - 3), out_dtype=torch.float32)
            y_ps = ops.conv2d(a, c, out_mode=RC_OUT_PIXEL_SHUFFLE2_NCHW, out_dtype=torch.float32)
            y_ps_crop = ops.conv2d(a, c, out_mode=RC_OUT_PIXEL_SHUFFLE2_NCHW, crop_hw=(2 * H - 3, 2 * W - 5))      # odd width: element stores
            y_ps_even = ops.conv2d(a, c, out_mode=RC_OUT_PIXEL_SHUFFLE2_NCHW, crop_hw=(2 * H - 1, 2 * W - 6))
    finally:
        hip.rc_debug_set(b"persist", 1)
    rb = (lambda t: t.bfloat16().float()) if dt == torch.bfloat16 else (lambda t: t)   # bf16 outputs: the exact fp32 sum rounded once (1200 terms reach |v| > 64)
    assert torch.equal(y_nhwc.float().cpu().permute(0, 3, 1, 2), rb(ref))
    assert torch.equal(y_nchw.cpu(), ref[:, :, :H - 1, :W - 3])
    ps = F.pixel_shuffle(ref, 2)
    assert torch.equal(y_ps.cpu(), ps)
    assert torch.equal(y_ps_crop.float().cpu(), rb(ps[:, :, :2 * H - 3, :2 * W - 5]))
    assert torch.equal(y_ps_even.float().cpu(), rb(ps[:, :, :2 * H - 1, :2 * W - 6]))


def _psnr(a, b):
    return float(10 * torch.log10((b.max() - b.min()) ** 2 / ((a - b) ** 2).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("C,dt", [(32, torch.bfloat16), (64, torch.bfloat16), (64, torch.float32)])
def test_folded_tail_of_the_other_widths_vs_fp64_reference(hip, C, dt):
    """The fold at the ISPUNet family's 32 channels, LiteISPNet's 64 (bf16) and in fp32 (cfg2): against conv -> PixelShuffle -> conv in double.
    fp32 tolerance: 2e-5 * max|ref| (reassociation only); bf16: as the 48-channel test."""
    from realcamnet_amd import networks as N, ops
    torch.manual_seed(2)
    tail = N.seq(N.conv(C, 4 * C, mode="C"), torch.nn.PixelShuffle(2), N.conv(C, 3, mode="C")).to(DEV, dt).eval()
    c1, c2 = tail[0], tail[2]
    x = torch.randn(2, 21, 70, C, generator=torch.Generator().manual_seed(C)).to(DEV, dt)
    crop = (41, 140)
    with torch.no_grad():
        assert ops.tail_fold_ok(x, c1, c2)
        one = ops.tail_fold(x, c1, c2, crop_hw=crop)
        two = c2._nhwc(c1._nhwc(x, out_mode=RC_OUT_PIXEL_SHUFFLE2), out_mode=RC_OUT_NCHW, crop_hw=crop)
    d = lambda t: t.detach().double().cpu()
    ref = _two_step(d(x).permute(0, 3, 1, 2), d(c1.weight), d(c1.bias), d(c2.weight), d(c2.bias))[:, :, :crop[0], :crop[1]]
    o, t = d(one), d(two)
    if dt == torch.float32:
        assert (o - ref).abs().max() <= 2e-5 * ref.abs().max() and (t - ref).abs().max() <= 2e-5 * ref.abs().max()
    else:
        assert _psnr(o, ref) >= 60.0 and _psnr(o, ref) >= _psnr(t, ref) - 1.0
        assert (o - ref)[:, :, 0].abs().max() <= 2 * (t - ref).abs().max() + 1e-6 and (o - ref)[:, :, :, 0].abs().max() <= 2 * (t - ref).abs().max() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("out_dtype", [None, torch.float32])
@pytest.mark.parametrize("B,H,W,crop", [(2, 37, 70, None), (1, 16, 32, (30, 64)), (2, 24, 40, (47, 79)), (1, 2, 2, None), (3, 8, 130, (16, 259))])
def test_folded_tail_vs_fp64_reference_ring_included(hip, B, H, W, crop, out_dtype):
    """ops.tail_fold against conv -> PixelShuffle -> conv in double.  Tolerance: the folded path rounds its weights to bf16 once and accumulates
    in fp32 (no bf16 intermediate map), so it must be at least as close as the two launches: PSNR >= 60 dB overall (bf16 output: ~70 dB is the
    output rounding itself), and the ring -- where an unrepaired fold is off by ~0.15 of the range -- within 2x the two launches' worst error."""
    from realcamnet_amd import networks as N, ops
    torch.manual_seed(1)
    tail = N.seq(N.conv(48, 192, mode="C"), torch.nn.PixelShuffle(2), N.conv(48, 3, mode="C")).to(DEV, torch.bfloat16).eval()
    c1, c2 = tail[0], tail[2]
    x = torch.randn(B, H, W, 48, generator=torch.Generator().manual_seed(H * W)).to(DEV, torch.bfloat16)
    with torch.no_grad():
        assert ops.tail_fold_ok(x, c1, c2)
        one = ops.tail_fold(x, c1, c2, crop_hw=crop, out_dtype=out_dtype)
        two = c2._nhwc(c1._nhwc(x, out_mode=RC_OUT_PIXEL_SHUFFLE2), out_mode=RC_OUT_NCHW, crop_hw=crop, out_dtype=out_dtype)
    assert one.shape == two.shape and one.dtype == two.dtype == (out_dtype or torch.bfloat16)
    d = lambda t: t.detach().double().cpu()
    ref = _two_step(d(x).permute(0, 3, 1, 2), d(c1.weight), d(c1.bias), d(c2.weight), d(c2.bias))
    if crop:
        ref = ref[:, :, :crop[0], :crop[1]]
    ring = torch.zeros_like(ref, dtype=torch.bool)
    ring[:, :, 0] = True; ring[:, :, :, 0] = True
    if not crop or crop[0] == 2 * H: ring[:, :, -1] = True
    if not crop or crop[1] == 2 * W: ring[:, :, :, -1] = True
    o, t = d(one), d(two)
    assert _psnr(o, ref) >= 60.0 and _psnr(o, ref) >= _psnr(t, ref) - 1.0
    assert (o - ref)[ring].abs().max() <= 2 * (t - ref).abs().max() + 1e-6
    if out_dtype == torch.float32:                      # without the output rounding the fold is clearly the closer of the two
        assert _psnr(o, ref) >= 65.0


@pytest.mark.gpu
def test_net_with_folded_tail_matches_the_module_list_form(hip):
    """LiteISPNet_GFM_LSC end to end with ops.FOLD_TAIL on / off on a ragged mosaic (pad + crop): same result within bf16 output rounding."""
    import realcamnet_amd as M
    from realcamnet_amd import ops
    from conftest import seed0_state_dict
    net = M.LiteISPNet_GFM_LSC()
    net.load_state_dict(seed0_state_dict("LiteISPNet_GFM_LSC"), strict=True)
    net = net.to(DEV, torch.bfloat16).eval()
    g = torch.Generator().manual_seed(5)
    mosaic = torch.rand(2, 1, 2 * 44, 2 * 72, generator=g).to(DEV, torch.bfloat16)
    cond = torch.rand(2, 4, 64, 64, generator=g).to(DEV, torch.bfloat16)
    coord = ops.make_coord(2, 44, 72, device=DEV, dtype=torch.bfloat16)
    outs = {}
    try:
        for flag in (True, False):
            ops.FOLD_TAIL = flag
            with torch.no_grad():
                outs[flag] = net.forward_mosaic(mosaic, cond, coord).float().cpu()
    finally:
        ops.FOLD_TAIL = True
    assert outs[True].shape == outs[False].shape == (2, 3, 88, 144)
    assert _psnr(outs[True], outs[False]) >= 60.0


@pytest.mark.gpu
@pytest.mark.parametrize("cin,dt", [(128, torch.bfloat16), (64, torch.bfloat16), (64, torch.float32)])
def test_3x3_conv_with_pixel_shuffled_planar_store_exact_on_integer_data(hip, cin, dt):
    """RC_OUT_PIXEL_SHUFFLE2_NCHW on an ordinary 3x3 convolution (the codecs' subpel_conv3x3(2N, 3, 2) writing x_hat: multi-chunk and general
    kernels): equals F.pixel_shuffle(F.conv2d(...)) bit for bit, and the separate rc_pixel_shuffle2_nchw pass it replaces."""
    from realcamnet_amd import networks as N, ops
    g = torch.Generator().manual_seed(cin)
    c = N.Conv2d(cin, 12, 3, 1, 1)
    with torch.no_grad():
        c.weight.copy_(torch.randint(-2, 3, c.weight.shape, generator=g).float() / 2)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-2, 3, (2, cin, 21, 40), generator=g).float() / 2
    ref = F.pixel_shuffle(F.conv2d(x, c.weight.detach(), c.bias.detach(), padding=1), 2)
    c = c.to(DEV, dt).eval()
    a = ops.to_nhwc(x.to(DEV, dt))
    with torch.no_grad():
        y = ops.conv2d(a, c, out_mode=RC_OUT_PIXEL_SHUFFLE2_NCHW)
        y2 = ops.pixel_shuffle2_nchw(ops.conv2d(a, c))
    rb = (lambda t: t.bfloat16().float()) if dt == torch.bfloat16 else (lambda t: t)
    assert torch.equal(y.float().cpu(), rb(ref)) and torch.equal(y, y2)
