"""CPU oracle for the RAW->sRGB hot path (LiteISP family).   *** TEST INFRASTRUCTURE ***

This file is a checker, never the thing measured or shipped.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it.  The product path (realcamnet_amd/) never does; it
raises if the HIP library is missing.

What it is: a functional fp32 PyTorch-CPU restatement of the reference algorithm.  Every function
takes the reference's *state_dict* (a plain ``{key: tensor}`` mapping -- the checkpoint contract,
SURVEY.md section 8b) plus NCHW tensors, and cites the reference lines it follows
(paths relative to the upstream repo kepengxu/RealCamNet @ 2024-10-20).

Pinning: oracle/make_golden.py imports the real reference (build container only) and writes
tests/golden/*.npz; tests/test_oracle_golden.py replays those fixtures through this file.  The
reference ships no known-answer vectors of its own, so the fixtures generated
from the imported reference are the pin.
"""
from __future__ import annotations

import math
from typing import Mapping, Sequence

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------
# a1 / a2  Bayer unshuffle, padding helpers
# ----------------------------------------------------------------------------------------------
def bayer_unshuffle(mosaic: torch.Tensor) -> torch.Tensor:
    """(B,1,2H,2W) mosaic -> packed (B,4,H,W); channel k=2i+j <- pixel (2y+i, 2x+j).

    Not in upstream code; defined by assets/networkarch.png ("Unpixel shuffle") and README.md:33-38.
    Equivalent to F.pixel_unshuffle(mosaic, 2).
    """
    b, one, h2, w2 = mosaic.shape
    assert one == 1 and h2 % 2 == 0 and w2 % 2 == 0
    planes = [mosaic[:, 0, i::2, j::2] for i in (0, 1) for j in (0, 1)]
    return torch.stack(planes, dim=1)


def raw_ingest(mosaic: torch.Tensor, black_level: float = 0.0, white_level: float = 1.0, cond_hw=(256, 256)):
    """The ingest step in front of the path ('Unpixel shuffle' + 'Resize' boxes of assets/networkarch.png,
    unpublished upstream): normalise (v - black) / (white - black), Bayer unshuffle, and cond = bilinear resize of the packed RAW
    (F.interpolate, align_corners=False) -- SURVEY.md 8d cfg3 'cond = bilinear resize of packed raw to 256x256'.
    Returns (packed (B,4,h,w), cond (B,4,*cond_hw))."""
    packed = bayer_unshuffle((mosaic.float() - black_level) * (1.0 / (white_level - black_level)))
    return packed, F.interpolate(packed, size=tuple(cond_hw), mode="bilinear", align_corners=False)


def pad_to_multiple(x: torch.Tensor, mult: int = 16):
    """Zero-pad bottom/right so H,W % mult == 0.  models/LiteISP.py:84-105 (mult=16 upstream)."""
    h, w = x.shape[-2:]
    ph, pw = (-h) % mult, (-w) % mult
    return F.pad(x, (0, pw, 0, ph), value=0.0), (h, w)


def remove_padding(y: torch.Tensor, orig_hw):
    """Crop the 2x output back to 2*orig size.  models/LiteISP.py:108-128."""
    h, w = orig_hw
    return y[..., : min(2 * h, y.shape[-2]), : min(2 * w, y.shape[-1])]


# ----------------------------------------------------------------------------------------------
# a3  conv, a8-a9 channel attention / RCAB / RCAGroup, a10 Haar DWT
# ----------------------------------------------------------------------------------------------
def conv(sd: SD, p: str, x: torch.Tensor, pad: int | None = None) -> torch.Tensor:
    """nn.Conv2d stride 1, zero padding k//2, + bias.  models/networks.py:146-160 (mode 'C')."""
    w = sd[p + ".weight"]
    b = sd.get(p + ".bias")
    if pad is None:
        pad = w.shape[-1] // 2
    return F.conv2d(x, w, b, stride=1, padding=pad)


def ca_layer(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Channel attention: x * sigmoid(W1 relu(W0 mean_hw(x)+b0)+b1).  models/networks.py:255-270."""
    y = x.mean(dim=(2, 3), keepdim=True)
    y = torch.relu(conv(sd, p + ".conv_du.0", y))
    y = torch.sigmoid(conv(sd, p + ".conv_du.2", y))
    return x * y


def rcab(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """x + CA(conv(relu(conv(x)))).  models/networks.py:296-311 (mode 'CRC')."""
    r = conv(sd, p + ".res.2", torch.relu(conv(sd, p + ".res.0", x)))
    return ca_layer(sd, p + ".ca", r) + x


def rcag(sd: SD, p: str, x: torch.Tensor, nb: int = 4) -> torch.Tensor:
    """nb RCABs + conv, plus group skip.  models/networks.py:317-335."""
    r = x
    for i in range(nb):
        r = rcab(sd, f"{p}.rg.{i}", r)
    return conv(sd, f"{p}.rg.{nb}", r) + x


def dwt_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Haar analysis as a frozen grouped 2x2 stride-2 conv; taps live in the state_dict.
    models/networks.py:224-235.  (B,C,h,w) -> (B,4C,h/2,w/2), channel 4c+k."""
    return F.conv2d(x, sd[p + ".weight"], None, stride=2, groups=x.shape[1])


def dwt_inverse(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Haar synthesis as grouped 2x2 stride-2 transposed conv.  models/networks.py:238-249."""
    return F.conv_transpose2d(x, sd[p + ".weight"], None, stride=2, groups=x.shape[1] // 4)


def dwt_forward_(weight: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """DWTForward_ (the channel-count-agnostic variant: ONE (4,1,2,2) tap set repeated over the input's channels).  models/networks.py:9-26."""
    c = x.shape[1]
    return F.conv2d(x, torch.cat([weight] * c, dim=0), None, stride=2, groups=c)


def dwt_inverse_(weight: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """DWTInverse_: grouped transposed conv with the same repeated taps.  models/networks.py:29-47."""
    c = x.shape[1] // 4
    return F.conv_transpose2d(x, torch.cat([weight] * c, dim=0), None, stride=2, groups=c)


def pixel_shuffle2(x: torch.Tensor) -> torch.Tensor:
    """nn.PixelShuffle(2): out[b,c,2y+i,2x+j] = in[b,4c+2i+j,y,x].  models/networks.py:201-202."""
    return F.pixel_shuffle(x, 2)


# ----------------------------------------------------------------------------------------------
# a5-a7  lens shading, colour prior, global feature modulation
# ----------------------------------------------------------------------------------------------
def lens_shading(sd: SD, p: str, coord: torch.Tensor) -> torch.Tensor:
    """4x Conv1x1 with LeakyReLU(0.1) between.  models/LiteISP.py:363-378."""
    h = coord
    for i in (0, 2, 4):
        h = F.leaky_relu(conv(sd, f"{p}.model.{i}", h), 0.1)
    return conv(sd, f"{p}.model.6", h)


def color_condition_gfm(sd: SD, p: str, cond: torch.Tensor) -> torch.Tensor:
    """Global colour prior vector.  models/LiteISP.py:345-361, color_block :23-30.

    5 x [Conv1x1 -> AvgPool(3,s2,p1,count_include_pad) -> LeakyReLU(0.2) -> InstanceNorm(affine)]
    (no norm on the 5th), Dropout (identity in eval), Conv1x1 -> out_c, global average pool.
    Returns (B, out_c) (the call site squeezes dims 2,3: models/LiteISP.py:2017).
    """
    h = cond
    for blk in range(5):
        i = 4 * blk
        h = conv(sd, f"{p}.model.{i}", h)
        h = F.avg_pool2d(h, 3, stride=2, padding=1, count_include_pad=True)
        h = F.leaky_relu(h, 0.2)
        if blk < 4:
            h = F.instance_norm(h, weight=sd[f"{p}.model.{i + 3}.weight"], bias=sd[f"{p}.model.{i + 3}.bias"],
                                use_input_stats=True, eps=1e-5)
    h = conv(sd, f"{p}.model.20", h)
    return h.mean(dim=(2, 3))


def _gfm_vec(sd: SD, p: str, which: str, v: torch.Tensor) -> torch.Tensor:
    h = F.leaky_relu(F.linear(v, sd[f"{p}.GFM_{which}_conv0.weight"], sd[f"{p}.GFM_{which}_conv0.bias"]), 0.1)
    return F.linear(h, sd[f"{p}.GFM_{which}_conv1.weight"], sd[f"{p}.GFM_{which}_conv1.bias"])


def res_gfm(sd: SD, p: str, x: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """GFT block: conv0 -> f*scale+shift+f -> LeakyReLU(0.01) -> conv1 + x.  models/LiteISP.py:537-559.
    (nn.LeakyReLU default slope 0.01, :548.)"""
    f = conv(sd, p + ".conv0", x)
    c = f.shape[1]
    s = _gfm_vec(sd, p, "scale", v).view(-1, c, 1, 1)
    t = _gfm_vec(sd, p, "shift", v).view(-1, c, 1, 1)
    f = f * s + t + f
    f = F.leaky_relu(f, 0.01)
    return conv(sd, p + ".conv1", f) + x


def color_condition_gfm_lfm(sd: SD, p: str, cond: torch.Tensor, raw: torch.Tensor):
    """Color_Condition_GFM_LFM.forward (models/LiteISP.py:501-534): six CB blocks (:215-230; conv1x1 -> AvgPool(3,2,1) -> LeakyReLU(0.2)
    -> InstanceNorm(affine), the sixth without norm), Dropout (identity in eval) -> Conv1x1 -> global mean = the vector; the local map
    is ONE 3x3 conv of the packed RAW (the 3-layer cond_first of :524-527 is overwritten at :529)."""
    h = cond
    i = 0
    while f"{p}.downblocks.{i}.conv.weight" in sd:
        q = f"{p}.downblocks.{i}"
        h = F.leaky_relu(F.avg_pool2d(conv(sd, q + ".conv", h), 3, stride=2, padding=1, count_include_pad=True), 0.2)
        if q + ".norm.weight" in sd:
            h = F.instance_norm(h, weight=sd[q + ".norm.weight"], bias=sd[q + ".norm.bias"], use_input_stats=True, eps=1e-5)
        i += 1
    return conv(sd, f"{p}.global_vector.1", h).mean(dim=(2, 3)), conv(sd, f"{p}.cond_first.0", raw)


def sft_layer(sd: SD, p: str, x: torch.Tensor, cmap: torch.Tensor) -> torch.Tensor:
    """SFTLayer.forward: x * (scale + 1) + shift, each a Conv1x1 -> LeakyReLU(0.1) -> Conv1x1 of the map.  models/LiteISP.py:293-305."""
    s = conv(sd, p + ".SFT_scale_conv1", F.leaky_relu(conv(sd, p + ".SFT_scale_conv0", cmap), 0.1))
    t = conv(sd, p + ".SFT_shift_conv1", F.leaky_relu(conv(sd, p + ".SFT_shift_conv0", cmap), 0.1))
    return x * (s + 1) + t


def res_gfm_lfm(sd: SD, p: str, x: torch.Tensor, v: torch.Tensor, cmap: torch.Tensor) -> torch.Tensor:
    """Res_GFM_LFM.forward: x + conv2(lfm(lrelu_0.1(conv1(gfm(x, v))), cmap)).  models/LiteISP.py:601-620; GFMLayer :308-321."""
    c = x.shape[1]
    s = _gfm_vec(sd, p + ".gfm", "scale", v).view(-1, c, 1, 1)
    t = _gfm_vec(sd, p + ".gfm", "shift", v).view(-1, c, 1, 1)
    f = F.leaky_relu(conv(sd, p + ".conv1", x * s + t + x), 0.1)
    return x + conv(sd, p + ".conv2", sft_layer(sd, p + ".lfm", f, cmap))


def _cond_net(sd: SD, p: str, t: torch.Tensor) -> torch.Tensor:
    """CondNet1..4 (models/LiteISP.py:1668-1676): Conv2d(k, stride k) layers, k in {1, 2}, LeakyReLU(0.1) between them."""
    idx = sorted(int(k.split(".")[1]) for k in sd if k.startswith(p + ".") and k.endswith(".weight"))
    for n, i in enumerate(idx):
        w = sd[f"{p}.{i}.weight"]
        t = F.conv2d(t, w, sd[f"{p}.{i}.bias"], stride=w.shape[-1])
        if n + 1 < len(idx):
            t = F.leaky_relu(t, 0.1)
    return t


# ----------------------------------------------------------------------------------------------
# a12  full nets
# ----------------------------------------------------------------------------------------------
def _unet_trunk(sd: SD, h: torch.Tensor, v: torch.Tensor | None, refine_d1=None) -> torch.Tensor:
    """Shared DWT U-Net body of LiteISPNet (models/LiteISP.py:2397-2409) and
    LiteISPNet_GFM_LSC (:2019-2032); `v` is the GFM vector or None."""
    def mod(i, t):
        return res_gfm(sd, f"encoder_modulation{i}", t, v) if v is not None else t

    h = mod(1, h)
    d1 = conv(sd, "down1.0", h)
    d1 = rcag(sd, "down1.1", d1)
    d1 = conv(sd, "down1.2", d1)
    d1 = dwt_forward(sd, "down1.3", d1)
    if refine_d1 is not None:
        d1 = refine_d1(d1)

    d2 = mod(2, d1)
    d2 = conv(sd, "down2.0", d2)
    d2 = rcag(sd, "down2.1", d2)
    d2 = dwt_forward(sd, "down2.2", d2)

    d3 = mod(3, d2)
    d3 = conv(sd, "down3.0", d3)
    d3 = rcag(sd, "down3.1", d3)
    d3 = dwt_forward(sd, "down3.2", d3)

    m = mod(4, d3)
    m = conv(sd, "middle.0", m)
    m = rcag(sd, "middle.1", m)
    m = rcag(sd, "middle.2", m)
    m = conv(sd, "middle.3", m) + d3

    u = dwt_inverse(sd, "up3.0", m)
    u = rcag(sd, "up3.1", u)
    u = conv(sd, "up3.2", u) + d2

    u = dwt_inverse(sd, "up2.0", u)
    u = rcag(sd, "up2.1", u)
    u = conv(sd, "up2.2", u) + d1

    u = dwt_inverse(sd, "up1.0", u)
    u = rcag(sd, "up1.1", u)
    u = conv(sd, "up1.2", u) + h

    t = conv(sd, "tail.0", u)
    t = pixel_shuffle2(t)
    return conv(sd, "tail.2", t)


def liteispnet(sd: SD, x: Sequence[torch.Tensor] | torch.Tensor) -> torch.Tensor:
    """LiteISPNet.forward -- reads only x[0].  models/LiteISP.py:2385-2412."""
    raw = x if isinstance(x, torch.Tensor) else x[0]
    return _unet_trunk(sd, conv(sd, "head", raw), None)


def liteispnet_gfm_lsc(sd: SD, x: Sequence[torch.Tensor]) -> torch.Tensor:
    """LiteISPNet_GFM_LSC.forward, x=[raw(B,4,H,W), cond(B,4,h,w), coord(B,2,H,W)].
    models/LiteISP.py:2002-2035."""
    raw, cond, coord = x
    h = conv(sd, "head", raw)
    h = h * (lens_shading(sd, "lsc", coord) + 1)
    v = color_condition_gfm(sd, "classifier", cond)
    return _unet_trunk(sd, h, v)


def liteispnet_gfm_lsc_gma(sd: SD, x: Sequence[torch.Tensor], heads: int = 8) -> torch.Tensor:
    """Build-defined cfg3 composition (realcamnet_amd.LiteISP.LiteISPNet_GFM_LSC_GMA): the flagship with one
    GMA_Block (models/groupmix.py:274-299) as a residual refinement of d1:  d1 += gma_out(GMA(gma_in(d1)))."""
    import groupmix_oracle as GO
    raw, cond, coord = x
    h = conv(sd, "head", raw)
    h = h * (lens_shading(sd, "lsc", coord) + 1)
    v = color_condition_gfm(sd, "classifier", cond)

    def refine(d1):
        t = conv(sd, "gma_in", d1)
        b, c, hh, ww = t.shape
        tok = GO.gma_block(sd, t.flatten(2).transpose(1, 2), (hh, ww), heads, p="gma.")
        return d1 + conv(sd, "gma_out", tok.transpose(1, 2).reshape(b, c, hh, ww))

    return _unet_trunk(sd, h, v, refine)


def _strided_unet(sd: SD, x: Sequence[torch.Tensor], cond_from_raw: bool = False, skips: bool = True) -> torch.Tensor:
    """Shared body of the ISPUNet / ResUNet family: ISPUNet_GFM_LSC (models/LiteISP.py:1340-1380),
    ISPUNet_GFM (:1076-1110), ISPUNet_LSC (:1196-1225), ResUNet (:2121-2146).  Conv2d(c, 2c, 2, 2) down-samplers (:1253),
    Conv1x1(c, 2c, bias=False) + PixelShuffle(2) up-samplers (:1292-1295), RCAGroups of 2 blocks (middle: 4).  Which of the
    optional parts exist (colour prior + Res_GFM modulation, lens shading) is read off the state_dict, like the count of
    Res_GFM blocks per level.  Round 3: ISPUNet_GFM_crop (:811-960: the colour prior reads x[0]), ISPUNet_GFM_LSC1 (:1382-1532: coord
    concatenated to the RAW in front of a 6-channel intro conv, seen from intro.weight's shape), ISPUNet_GFM_LSC_noskip (:2522-2652: no
    additive skips, no decoder modulation)."""
    raw = x[0]
    has_gfm, has_lsc = "classifier.model.0.weight" in sd, "lsc.model.0.weight" in sd
    has_lfm = "classifier.cond_first.0.weight" in sd     # ISPUNet_GFM_LFM (:1535-1707): vector AND map modulation, maps per level from CondNet1..4
    v = color_condition_gfm(sd, "classifier", raw if cond_from_raw else x[1]) if has_gfm else None
    lmap = {}
    if has_lfm:
        v, lfm = color_condition_gfm_lfm(sd, "classifier", x[1], raw)
        l1, l2, l4, l8 = (_cond_net(sd, f"CondNet{i}", lfm) for i in (1, 2, 3, 4))
        lmap = {"encoder_modulation1": l1, "encoder_modulation2": l2, "encoder_modulation3": l4, "middle_modulation": l8,
                "decoder_modulation3": l4, "decoder_modulation2": l2, "decoder_modulation1": l1}

    def gfm(p, t):
        if has_lfm:
            if f"{p}.conv1.weight" in sd:
                return res_gfm_lfm(sd, p, t, v, lmap[p])
            i = 0
            while f"{p}.{i}.conv1.weight" in sd:
                t = res_gfm_lfm(sd, f"{p}.{i}", t, v, lmap[p])
                i += 1
            return t
        if not has_gfm:
            return t
        if f"{p}.conv0.weight" in sd:                    # N.seq of ONE module is that module (models/networks.py:117-121)
            return res_gfm(sd, p, t, v)
        i = 0
        while f"{p}.{i}.conv0.weight" in sd:
            t = res_gfm(sd, f"{p}.{i}", t, v)
            i += 1
        return t

    def down(p, t):                                      # kernel 2, stride 2, no padding
        return F.conv2d(t, sd[p + ".weight"], sd.get(p + ".bias"), stride=2)

    def up(p, t):
        return pixel_shuffle2(F.conv2d(t, sd[p + ".0.weight"]))

    def enc(p, t, lead_conv):
        i = 0
        if lead_conv:
            t = conv(sd, f"{p}.0", t); i = 1
        t = rcag(sd, f"{p}.{i}", t, nb=2)
        return F.leaky_relu(conv(sd, f"{p}.{i + 1}", t), 0.1)

    intro = conv(sd, "intro", torch.cat([raw, x[2]], dim=1) if sd["intro.weight"].shape[1] == 6 else raw)     # :1497
    if has_lsc:
        intro = intro * (lens_shading(sd, "lsc", x[2]) + 1)
    d1 = down("down1", enc("encoder1", gfm("encoder_modulation1", intro), False))
    d2 = down("down2", enc("encoder2", gfm("encoder_modulation2", d1), False))
    d3 = down("down3", enc("encoder3", gfm("encoder_modulation3", d2), True))
    m = gfm("middle_modulation", d3)
    m = conv(sd, "middle.2", rcag(sd, "middle.1", conv(sd, "middle.0", m), nb=4))
    if skips:
        m = m + d3

    def dec(i, t, skip):
        t = conv(sd, f"decoder{i}.1", rcag(sd, f"decoder{i}.0", up(f"up{i}", t), nb=2))
        if not skips:                                    # :2646-2648
            return t
        return gfm(f"decoder_modulation{i}", t) + skip

    u = dec(1, dec(2, dec(3, m, d2), d1), intro)
    return conv(sd, "tail.2", pixel_shuffle2(conv(sd, "tail.0", u)))


ispunet_gfm_lsc = _strided_unet


def _dwt_unet(sd: SD, x: Sequence[torch.Tensor], cond_from_raw: bool = False) -> torch.Tensor:
    """The LiteISPNet family around _unet_trunk: LiteISPNet_LSC (models/LiteISP.py:1787-1805: head * (lsc + 1), no prior),
    LiteISPNet_GFM (:1894-1920: prior + Res_GFM, no lens shading; 64 channels, cond_c 64), LiteISPNet_GFMresize (:2490-2520: as
    _GFM but the colour prior reads the packed RAW x[0] itself)."""
    raw = x[0]
    h = conv(sd, "head", raw)
    if "lsc.model.0.weight" in sd:
        h = h * (lens_shading(sd, "lsc", x[2]) + 1)
    v = None
    if "classifier.model.0.weight" in sd:
        v = color_condition_gfm(sd, "classifier", raw if cond_from_raw else x[1])
    return _unet_trunk(sd, h, v)


def liteispnet_gfmresize(sd: SD, x: Sequence[torch.Tensor]) -> torch.Tensor:
    return _dwt_unet(sd, x, cond_from_raw=True)


def ispunet_gfm_crop(sd: SD, x: Sequence[torch.Tensor]) -> torch.Tensor:
    return _strided_unet(sd, x, cond_from_raw=True)


def ispunet_gfm_lsc_noskip(sd: SD, x: Sequence[torch.Tensor]) -> torch.Tensor:
    return _strided_unet(sd, x, skips=False)


FORWARDS = {"ISPUNet_GFM_crop": ispunet_gfm_crop, "ISPUNet_GFM_LSC1": _strided_unet, "ISPUNet_GFM_LSC_noskip": ispunet_gfm_lsc_noskip,
            "ISPUNet_GFM_LSC": ispunet_gfm_lsc, "LiteISPNet": liteispnet, "LiteISPNet_GFM_LSC": liteispnet_gfm_lsc, "LiteISPNet_GFM_LSC_GMA": liteispnet_gfm_lsc_gma,
            "ISPUNet_GFM": _strided_unet, "ISPUNet_LSC": _strided_unet, "ResUNet": _strided_unet, "ISPUNet_GFM_LFM": _strided_unet,
            "LiteISPNet_LSC": _dwt_unet, "LiteISPNet_GFM": _dwt_unet, "LiteISPNet_GFMresize": liteispnet_gfmresize}


# ----------------------------------------------------------------------------------------------
# helpers shared by tests / bench (input conventions of SURVEY.md section 8d)
# ----------------------------------------------------------------------------------------------
def make_coord(b: int, h: int, w: int) -> torch.Tensor:
    """Normalised meshgrid (B,2,H,W) in [-1,1]: channel 0 = y, channel 1 = x (build convention,
    upstream never published its coord generator)."""
    ys = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, 1, h, w)
    xs = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, 1, h, w)
    return torch.cat([ys, xs], dim=1).contiguous()


def psnr(test: torch.Tensor, ref: torch.Tensor) -> float:
    """PSNR with peak = range of the reference output."""
    ref = ref.double()
    mse = torch.mean((test.double() - ref) ** 2).item()
    peak = (ref.max() - ref.min()).item()
    if mse == 0.0:
        return float("inf")
    return 10.0 * math.log10(peak * peak / mse)


def run_padded(name: str, sd: SD, raw: torch.Tensor, cond=None, coord=None, mult: int = 16) -> torch.Tensor:
    """Oracle forward with the reference padding convention (pad packed RAW (+coord) bottom/right
    with zeros to a multiple of `mult`, crop the output to 2x the original size)."""
    rp, hw = pad_to_multiple(raw, mult)
    cp = pad_to_multiple(coord, mult)[0] if coord is not None else None
    out = FORWARDS[name](sd, [rp, cond, cp])
    return remove_padding(out, hw)
