"""HIP-backed mirror of the reference ISP nets on the RAW->sRGB path (upstream models/LiteISP.py).

Class names, constructor signatures, attribute names (=> state_dict keys) and forward() contracts
follow the reference:
    x = [raw (B,4,H,W), cond (B,4,h,w), coord (B,2,H,W)]  ->  sRGB (B,3,2H,2W)
`forward` is the drop-in entry; `forward_mosaic` adds the ingest step the paper's figure shows in
front of it (Bayer unshuffle + pad_to_multiple_of_16 + crop), which upstream never published.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import networks as N
from . import ops
from .groupmix import GMA_Block
from ._lib import RC_OUT_NCHW, RC_OUT_PIXEL_SHUFFLE2


def color_block(in_filters, out_filters, normalization=False):
    """Conv1x1 -> AvgPool(3,2,1) -> LeakyReLU(0.2) [-> InstanceNorm(affine)]; upstream LiteISP.py:23-30.
    The pooling/activation/norm modules are structural (they pin the Sequential indices and hold the
    affine parameters); execution is rc_color_block / rc_instance_stats."""
    layers = [N.Conv2d(in_filters, out_filters, 1, stride=1, padding=0),
              nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=True),
              nn.LeakyReLU(0.2)]
    if normalization:
        layers.append(nn.InstanceNorm2d(out_filters, affine=True))
    return layers


def pad_to_multiple_of_16(x):
    """Zero-pad NCHW x bottom/right to H,W % 16 == 0 (upstream LiteISP.py:84-105), on the device."""
    b, c, h, w = x.shape
    hp, wp = -(-h // 16) * 16, -(-w // 16) * 16
    return ops.to_nchw(ops.to_nhwc(x, pad_hw=(hp, wp))), (h, w)


def remove_padding(x_padded, original_size):
    """Crop the 2x output to 2*orig (upstream LiteISP.py:108-128)."""
    h, w = original_size
    return x_padded[:, :, :min(2 * h, x_padded.shape[-2]), :min(2 * w, x_padded.shape[-1])]


class Color_Condition_GFM(nn.Module):
    """Global colour prior (upstream LiteISP.py:345-361)."""

    def __init__(self, in_channels=4, out_c=32):
        super().__init__()
        self.model = nn.Sequential(
            *color_block(in_channels, 16, normalization=True),
            *color_block(16, 32, normalization=True),
            *color_block(32, 64, normalization=True),
            *color_block(64, 128, normalization=True),
            *color_block(128, 128),
            nn.Dropout(p=0.5),
            N.Conv2d(128, out_c, 1, stride=1, padding=0),
            nn.AdaptiveAvgPool2d(1),
        )

    def _vec(self, cond: torch.Tensor) -> torch.Tensor:
        """(B,4,h,w) NCHW cond image -> (B,out_c) fp32 vector; eval semantics (Dropout = identity)."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        m = self.model
        x, norm, stats = cond, None, None
        for blk in range(5):
            i = 4 * blk
            x = ops.color_block(x, m[i], norm, stats)
            if blk < 4:
                norm = m[i + 3]
                stats = ops.instance_stats(x, norm.eps)
            else:
                norm, stats = None, None
        return ops.color_head(x, m[20])

    def forward(self, img_input):
        v = self._vec(img_input)
        return v.to(img_input.dtype).view(v.shape[0], v.shape[1], 1, 1)


class Lens_Shading_Correction(N.HipModule):
    """4x Conv1x1 with LeakyReLU(0.1) between (upstream LiteISP.py:363-378)."""

    def __init__(self, in_channels=2, out_c=32, nf=32):
        super().__init__()
        self.model = N.Sequential(
            N.Conv2d(in_channels, nf, 1, 1),
            nn.LeakyReLU(negative_slope=0.1, inplace=True),
            N.Conv2d(nf, nf, 1, 1),
            nn.LeakyReLU(negative_slope=0.1, inplace=True),
            N.Conv2d(nf, nf, 1, 1),
            nn.LeakyReLU(negative_slope=0.1, inplace=True),
            N.Conv2d(nf, out_c, 1, 1),
        )

    def _nhwc(self, a):
        mods = list(self.model)
        convs, acts = mods[0::2], mods[1::2]
        if (all(isinstance(m, N.Conv2d) for m in convs) and all(isinstance(m, nn.LeakyReLU) for m in acts) and
                len(mods) == 2 * len(convs) - 1):
            slopes = [float(m.negative_slope) for m in acts]
            y = ops.lsc_chain(self, a)                          # all four layers in one launch, activations in registers
            if y is not None:
                return y
            if ops.pointwise_chain_ok(a, convs, slopes):       # the LDS-slab form (kept for rc_pointwise_chain48's callers)
                return ops.pointwise_chain(a, convs, slopes[0])
        return self.model._nhwc(a)


class Res_GFM(nn.Module):
    """GFT block (upstream LiteISP.py:537-559): conv0 -> f*scale+shift+f -> LeakyReLU(0.01) -> conv1 + x.
    Tuple in / tuple out so it chains inside Sequential like upstream."""

    def __init__(self, in_nc=32, chan=32, cond_c=32, out_nc=32, nf=64):
        super().__init__()
        self.conv0 = N.Conv2d(in_nc, chan, 3, 1, 1)
        self.conv1 = N.Conv2d(chan, chan, 3, 1, 1)
        self.GFM_scale_conv0 = nn.Linear(cond_c, nf)
        self.GFM_scale_conv1 = nn.Linear(nf, chan)
        self.GFM_shift_conv0 = nn.Linear(cond_c, nf)
        self.GFM_shift_conv1 = nn.Linear(nf, chan)
        self.out_nc = chan
        self.act = nn.LeakyReLU(inplace=True)

    def _nhwc(self, x):
        a, vec = x
        scale = ops.gfm_vector(vec, self.GFM_scale_conv0, self.GFM_scale_conv1)
        shift = ops.gfm_vector(vec, self.GFM_shift_conv0, self.GFM_shift_conv1)
        slope = float(self.act.negative_slope)
        if 0.0 <= slope <= 1.0 and ops.conv_pair_ok(a, self.conv0, self.conv1):
            return ops.conv_pair(a, self.conv0, self.conv1, act="leaky", slope=slope, film=(scale, shift), residual=a), vec
        f = self.conv0._nhwc(a, film=(scale, shift), act="leaky", slope=slope)
        return self.conv1._nhwc(f, residual=a), vec

    def forward(self, x):
        y, vec = self._nhwc((ops.to_nhwc(x[0]), x[1]))
        return ops.to_nchw(y), vec


class CB(nn.Module):
    """Conv1x1 -> AvgPool(3,2,1) -> LeakyReLU(0.2) [-> InstanceNorm(affine)] as a module (upstream LiteISP.py:215-230); executed
    inside Color_Condition_GFM_LFM through rc_color_block / rc_instance_stats like color_block()."""

    def __init__(self, in_filters, out_filters, normalization=False):
        super().__init__()
        self.conv = N.Conv2d(in_filters, out_filters, 1, stride=1, padding=0)
        self.pooling = nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=True)
        self.act = nn.LeakyReLU(0.2)
        self.normalization = normalization
        if normalization:
            self.norm = nn.InstanceNorm2d(out_filters, affine=True)

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        y = ops.color_block(x, self.conv)                       # fp32 NCHW like the colour branch
        return ops.instance_norm(y, self.norm) if self.normalization else y


class SFTLayer(nn.Module):
    """x0 * (scale(cond) + 1) + shift(cond), two 1x1-conv pairs on the condition MAP (upstream LiteISP.py:293-305)."""

    def __init__(self, cond_c=32, out_nc=64, nf=32):
        super().__init__()
        self.SFT_scale_conv0 = N.Conv2d(cond_c, nf, 1)
        self.SFT_scale_conv1 = N.Conv2d(nf, out_nc, 1)
        self.SFT_shift_conv0 = N.Conv2d(cond_c, nf, 1)
        self.SFT_shift_conv1 = N.Conv2d(nf, out_nc, 1)

    def _nhwc(self, x):
        fea, cond = x
        scale = self.SFT_scale_conv1._nhwc(self.SFT_scale_conv0._nhwc(cond, act="leaky", slope=0.1))
        shift = self.SFT_shift_conv1._nhwc(self.SFT_shift_conv0._nhwc(cond, act="leaky", slope=0.1))
        return ops.sft_apply(fea, scale, shift)                 # fea*scale + shift + fea

    def forward(self, x):
        return ops.to_nchw(self._nhwc((ops.to_nhwc(x[0]), ops.to_nhwc(x[1]))))


class GFMLayer(nn.Module):
    """x0 * scale(vec) + shift(vec) + x0 with two Linear pairs on the global vector (upstream LiteISP.py:308-321)."""

    def __init__(self, cond_c=32, out_nc=64, nf=32):
        super().__init__()
        self.GFM_scale_conv0 = nn.Linear(cond_c, nf)
        self.GFM_scale_conv1 = nn.Linear(nf, out_nc)
        self.GFM_shift_conv0 = nn.Linear(cond_c, nf)
        self.GFM_shift_conv1 = nn.Linear(nf, out_nc)
        self.out_nc = out_nc

    def _nhwc(self, x):
        fea, vec = x
        scale = ops.gfm_vector(vec, self.GFM_scale_conv0, self.GFM_scale_conv1)
        shift = ops.gfm_vector(vec, self.GFM_shift_conv0, self.GFM_shift_conv1)
        return ops.film_apply(fea, scale, shift)

    def forward(self, x):
        return ops.to_nchw(self._nhwc((ops.to_nhwc(x[0]), x[1])))


class Res_GFM_LFM(nn.Module):
    """Global (vector) + local (map) modulation block (upstream LiteISP.py:601-620): x0 + conv2(lfm(lrelu0.1(conv1(gfm(x0, vec))), map));
    triple in / triple out so it chains inside Sequential like upstream."""

    def __init__(self, cond_c=32, out_nc=32, nf=64):
        super().__init__()
        self.gfm = GFMLayer(cond_c=cond_c, out_nc=out_nc, nf=nf)
        self.conv1 = N.Conv2d(out_nc, out_nc, 3, 1, 1)
        self.lfm = SFTLayer(cond_c=cond_c, out_nc=out_nc, nf=out_nc)
        self.conv2 = N.Conv2d(out_nc, out_nc, 3, 1, 1)

    def _nhwc(self, x):
        a, vec, cmap = x
        f = self.conv1._nhwc(self.gfm._nhwc((a, vec)), act="leaky", slope=0.1)
        return self.conv2._nhwc(self.lfm._nhwc((f, cmap)), residual=a), vec, cmap

    def forward(self, x):
        y, vec, cmap = self._nhwc((ops.to_nhwc(x[0]), x[1], ops.to_nhwc(x[2])))
        return ops.to_nchw(y), vec, x[2]


class Color_Condition_GFM_LFM(nn.Module):
    """Global colour vector from the cond image + a local condition map from the RAW patch (upstream LiteISP.py:501-534)."""

    def __init__(self, in_channels=4, GFM_out_c=32, LFM_out_c=32):
        super().__init__()
        self.downblocks = nn.ModuleList([CB(in_channels, 16, True), CB(16, 32, True), CB(32, 64, True), CB(64, 128, True),
                                         CB(128, 256, True), CB(256, 384, False)])
        self.global_vector = nn.Sequential(nn.Dropout(p=0.8), N.Conv2d(384, GFM_out_c, 1, stride=1, padding=0), nn.AdaptiveAvgPool2d(1))
        # upstream builds a 3-layer cond_first and then overwrites the attribute with a single conv (:524-529); the RNG stream of the
        # discarded layers is consumed all the same, so they are constructed (and dropped) here too
        self.cond_first = nn.Sequential(N.Conv2d(in_channels, LFM_out_c, 3, 1, 1), nn.LeakyReLU(0.1, True), N.Conv2d(LFM_out_c, LFM_out_c, 1),
                                        nn.LeakyReLU(0.1, True), N.Conv2d(LFM_out_c, LFM_out_c, 1), nn.LeakyReLU(0.1, True))
        self.cond_first = nn.Sequential(N.Conv2d(in_channels, LFM_out_c, 3, 1, 1))

    def _run(self, global_raw, local_nhwc):
        """global_raw NCHW (B,4,h,w) -> vector (B,GFM_out_c) fp32; local_nhwc (B,H,W,4) -> local map NHWC (B,H,W,LFM_out_c)."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        x, norm, stats = ops._req(global_raw, "cond"), None, None
        for blk in self.downblocks:
            x = ops.color_block(x, blk.conv, norm, stats)
            if blk.normalization:
                norm, stats = blk.norm, ops.instance_stats(x, blk.norm.eps)
            else:
                norm, stats = None, None
        return ops.color_head(x, self.global_vector[1]), self.cond_first[0]._nhwc(local_nhwc)

    def forward(self, global_raw, local_patch):
        vec, lfm = self._run(global_raw, ops.to_nhwc(local_patch))
        return vec.to(global_raw.dtype).view(vec.shape[0], vec.shape[1], 1, 1), ops.to_nchw(lfm)


def _ingest(net, mosaic, cond, dt, pad_to, black_level, white_level, cond_hw):
    """Front end shared by every forward_mosaic: packed NHWC RAW (+ the cond image when the caller did not bring one)."""
    need_cond = hasattr(net, "classifier") and cond is None and not getattr(net, "cond_from_raw", False)
    if need_cond or black_level != 0.0 or white_level != 1.0 or mosaic.dtype == torch.uint16:
        a, c = ops.raw_ingest(mosaic, dtype=dt, pad_to=pad_to, black_level=black_level, white_level=white_level, cond_hw=cond_hw)
        return a, (c if need_cond else cond)
    return ops.bayer_unshuffle(mosaic, dtype=dt, pad_to=pad_to), cond


class _DwtUNet(nn.Module):
    """Shared trunk of the LiteISPNet family (upstream LiteISP.py:2019-2032, 2397-2409): Haar-DWT U-Net with RCAGroups; optional
    colour prior + Res_GFM modulation in front of each encoder level, optional lens-shading gain on the head."""

    output_dtype: Optional[torch.dtype] = None  # None: same as the parameters (reference semantics)
    cond_from_raw = False                       # LiteISPNet_GFMresize feeds the packed RAW itself to the colour prior

    def _build(self, ch_1, ch_2, ch_3, cond_c=None, lsc=False, nf=None, n_blocks=4):
        """Modules in the reference's construction order ([classifier], head, [lsc], [encoder_modulation i], down i, ..., tail),
        so torch.manual_seed(0) reproduces the reference's seed-0 parameters.  nf: Res_GFM hidden widths per level."""
        if cond_c is not None:
            self.classifier = Color_Condition_GFM(in_channels=4, out_c=cond_c)
        self.head = N.seq(N.conv(4, ch_1, mode='C'))
        if lsc:
            self.lsc = Lens_Shading_Correction(in_channels=2, out_c=ch_1, nf=ch_1)

        def gfm(i, c):
            if cond_c is not None:
                setattr(self, f"encoder_modulation{i}", N.seq(Res_GFM(in_nc=c, chan=c, cond_c=cond_c, out_nc=c, nf=nf[i - 1])))

        gfm(1, ch_1)
        self.down1 = N.seq(N.conv(ch_1, ch_1, mode='C'), N.RCAGroup(in_channels=ch_1, out_channels=ch_1, nb=n_blocks),
                           N.conv(ch_1, ch_1, mode='C'), N.DWTForward(ch_1))
        gfm(2, ch_1 * 4)
        self.down2 = N.seq(N.conv(ch_1 * 4, ch_1, mode='C'), N.RCAGroup(in_channels=ch_1, out_channels=ch_1, nb=n_blocks),
                           N.DWTForward(ch_1))
        gfm(3, ch_1 * 4)
        self.down3 = N.seq(N.conv(ch_1 * 4, ch_2, mode='C'), N.RCAGroup(in_channels=ch_2, out_channels=ch_2, nb=n_blocks),
                           N.DWTForward(ch_2))
        gfm(4, ch_2 * 4)
        self.middle = N.seq(N.conv(ch_2 * 4, ch_3, mode='C'), N.RCAGroup(in_channels=ch_3, out_channels=ch_3, nb=n_blocks),
                            N.RCAGroup(in_channels=ch_3, out_channels=ch_3, nb=n_blocks), N.conv(ch_3, ch_2 * 4, mode='C'))
        self.up3 = N.seq(N.DWTInverse(ch_2 * 4), N.RCAGroup(in_channels=ch_2, out_channels=ch_2, nb=n_blocks),
                         N.conv(ch_2, ch_1 * 4, mode='C'))
        self.up2 = N.seq(N.DWTInverse(ch_1 * 4), N.RCAGroup(in_channels=ch_1, out_channels=ch_1, nb=n_blocks),
                         N.conv(ch_1, ch_1 * 4, mode='C'))
        self.up1 = N.seq(N.DWTInverse(ch_1 * 4), N.RCAGroup(in_channels=ch_1, out_channels=ch_1, nb=n_blocks),
                         N.conv(ch_1, ch_1, mode='C'))
        self.tail = N.seq(N.conv(ch_1, ch_1 * 4, mode='C'), nn.PixelShuffle(upscale_factor=2), N.conv(ch_1, 3, mode='C'))

    def _act_dtype(self) -> torch.dtype:
        return self.head.weight.dtype

    def _refine_d1(self, d1):
        return d1

    def _check(self, raw):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        if raw.dim() != 4 or raw.shape[1] != 4:
            raise ValueError(f"raw must be (B,4,H,W), got {tuple(raw.shape)}")
        if raw.shape[2] % 8 or raw.shape[3] % 8:
            raise ValueError(f"packed RAW H,W must be multiples of 8 (3 Haar levels), got {raw.shape[2]}x{raw.shape[3]}; "
                             "use forward_mosaic()/pad_to_multiple_of_16 for other sizes")

    def _front(self, a, cond, coord_nhwc):
        """head(raw) [* (lsc(coord) + 1)] and the colour-prior vector (None without a classifier)."""
        def head():
            if not hasattr(self, "lsc"):
                return self.head._nhwc(a)
            h = ops.lsc_chain(self.lsc, coord_nhwc, self.head, a)       # h = head(raw) * (lsc + 1), one launch when 48-wide bf16
            return h if h is not None else self.head._nhwc(a, mul_plus1=self.lsc._nhwc(coord_nhwc))

        if not hasattr(self, "classifier"):
            return head(), None
        # the colour prior is a chain of ~10 tiny launches on the (small) cond image: on a side stream it runs under the head conv
        cond = ops._req(cond, "cond")
        vec, h = ops.fork_join(lambda: self.classifier._vec(cond), head, [cond])
        return h, vec

    def _trunk(self, h, vec, crop_hw=None):
        mod = (lambda i, t: getattr(self, f"encoder_modulation{i}")._nhwc((t, vec))[0]) if vec is not None else (lambda i, t: t)
        h = mod(1, h)
        d1 = self._refine_d1(self.down1._nhwc(h))
        d2 = self.down2._nhwc(mod(2, d1))
        d3 = self.down3._nhwc(mod(3, d2))
        m = self.middle._nhwc(mod(4, d3), residual=d3)
        u3 = self.up3._nhwc(m, residual=d2)
        u2 = self.up2._nhwc(u3, residual=d1)
        u1 = self.up1._nhwc(u2, residual=h)
        if ops.tail_fold_ok(u1, self.tail[0], self.tail[2]):          # no activation between the two tail convs: one folded 5x5 launch + border ring
            return ops.tail_fold(u1, self.tail[0], self.tail[2], crop_hw=crop_hw, out_dtype=self.output_dtype)
        t = self.tail[0]._nhwc(u1, out_mode=RC_OUT_PIXEL_SHUFFLE2)
        return self.tail[2]._nhwc(t, out_mode=RC_OUT_NCHW, crop_hw=crop_hw, out_dtype=self.output_dtype)

    def forward(self, x: Sequence[torch.Tensor]):
        """x = [raw (B,4,H,W), cond (B,4,h,w), coord (B,2,H,W)] (nets without a prior / lens shading ignore those entries, as
        upstream does) -> sRGB (B,3,2H,2W)."""
        raw = x if isinstance(x, torch.Tensor) else x[0]
        self._check(raw)
        dt = self._act_dtype()
        coord_nhwc = None
        if hasattr(self, "lsc"):
            coord = x[2]
            if coord.shape[0] != raw.shape[0] or coord.shape[1] != 2 or coord.shape[2:] != raw.shape[2:]:
                raise ValueError(f"coord must be (B,2,H,W) matching raw, got {tuple(coord.shape)}")
            coord_nhwc = ops.to_nhwc(coord, dtype=dt)
        cond = (raw if self.cond_from_raw else x[1]) if hasattr(self, "classifier") else None
        h, vec = self._front(ops.to_nhwc(raw, dtype=dt), cond, coord_nhwc)
        return self._trunk(h, vec)

    def forward_mosaic(self, mosaic, cond=None, coord=None, pad_to: int = 16, black_level: float = 0.0, white_level: float = 1.0,
                       cond_hw=(256, 256)):
        """Bayer mosaic (B,1,2h,2w), cond (B,4,hc,wc), coord (B,2,h,w) -> sRGB (B,3,2h,2w).
        RAW and coord are zero-padded bottom/right to a multiple of `pad_to` (reference convention,
        upstream LiteISP.py:84-105) and the output is cropped back.  cond=None on a net with a colour prior: the fused ingest
        kernel (ops.raw_ingest) also produces cond = bilinear resize of the normalised packed RAW to `cond_hw`."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        dt = self._act_dtype()
        a, cond = _ingest(self, mosaic, cond, dt, pad_to, black_level, white_level, cond_hw)
        b, hp, wp, _ = a.shape
        co = None
        if hasattr(self, "lsc"):
            if coord is None or coord.shape[-2:] != (mosaic.shape[-2] // 2, mosaic.shape[-1] // 2):
                raise ValueError("coord must be at packed resolution (h, w)")
            co = ops.to_nhwc(coord, dtype=dt, pad_hw=(hp, wp))
        if hasattr(self, "classifier") and self.cond_from_raw:
            cond = ops.to_nchw(a)                             # the padded packed RAW, as upstream's x[0]
        h, vec = self._front(a, cond, co)
        return self._trunk(h, vec, crop_hw=(mosaic.shape[-2], mosaic.shape[-1]))


class LiteISPNet(_DwtUNet):
    """upstream LiteISP.py:2322-2412 (reads only x[0])."""

    def __init__(self):
        super().__init__()
        self._build(64, 128, 128)


class LiteISPNet_GFM_LSC(_DwtUNet):
    """upstream LiteISP.py:1924-2035 -- the net the reference's __main__ builds."""

    def __init__(self):
        super().__init__()
        self._build(48, 128, 128, cond_c=32, lsc=True, nf=(48, 48, 48, 128))


class LiteISPNet_LSC(_DwtUNet):
    """upstream LiteISP.py:1710-1805: the 48-channel trunk with the lens-shading gain only (x[1] is ignored)."""

    def __init__(self):
        super().__init__()
        self._build(48, 128, 128, lsc=True)


class LiteISPNet_GFM(_DwtUNet):
    """upstream LiteISP.py:1809-1920: 64-channel trunk, colour prior (cond_c = 64) + Res_GFM modulation, no lens shading."""

    def __init__(self):
        super().__init__()
        self._build(64, 128, 128, cond_c=64, nf=(64, 64, 64, 128))


class LiteISPNet_GFMresize(_DwtUNet):
    """upstream LiteISP.py:2414-2520: as LiteISPNet_GFM with cond_c = 32, wider GFM MLPs, and the colour prior reading the packed
    RAW x[0] itself (:2496)."""
    cond_from_raw = True

    def __init__(self):
        super().__init__()
        self._build(64, 128, 128, cond_c=32, nf=(128, 256, 256, 512))


class _StridedUNet(nn.Module):
    """Shared body of the ISPUNet / ResUNet family (upstream LiteISP.py:963-1380, 2038-2146): widths chan,
    2chan, 4chan, 8chan; Conv2d(c, 2c, 2, 2) down-samplers; Conv1x1(c, 2c, bias=False) + PixelShuffle(2) up-samplers; RCAGroups of
    2 (middle: 4) blocks; optionally m_blocks Res_GFM per level on the way down AND up and the lens-shading gain on the intro.
    Modules are created in the reference's order ([classifier], intro, [lsc], [encoder_modulation1], encoder1, down1, ...), so the
    seed-0 parameters and the state_dict keys are identical."""

    output_dtype: Optional[torch.dtype] = None

    cond_from_raw = False        # ISPUNet_GFM_crop: the colour prior reads the packed RAW x[0] itself (upstream :928)
    coord_in_intro = False       # ISPUNet_GFM_LSC1: coord is concatenated to the RAW in front of a 6-channel intro conv (upstream :1497)
    skips = True                 # ISPUNet_GFM_LSC_noskip: no additive skips and no modulation on the way up (upstream :2626-2650)

    def _build(self, chan=32, cond_c=None, lsc=False, m_blocks=2, lsc_nf=None, intro_in=4, dec_gfm=True):
        n_blocks = 2
        if cond_c is not None:
            self.classifier = Color_Condition_GFM(in_channels=4, out_c=cond_c)

        def gfm(name, c):
            if cond_c is not None:
                setattr(self, name, N.seq(*[Res_GFM(in_nc=c, chan=c, cond_c=cond_c, out_nc=c, nf=c * 2) for _ in range(m_blocks)]))

        def lrelu():
            return nn.LeakyReLU(negative_slope=1e-1, inplace=True)

        self.intro = N.seq(N.Conv2d(intro_in, chan, 3, 1, 1))
        if lsc:
            self.lsc = Lens_Shading_Correction(in_channels=2, out_c=chan, nf=chan if lsc_nf is None else lsc_nf)
        gfm("encoder_modulation1", chan)
        self.encoder1 = N.seq(N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks), N.Conv2d(chan, chan, 3, 1, 1), lrelu())
        self.down1 = N.Conv2d(chan, chan * 2, 2, 2)
        chan = chan * 2
        gfm("encoder_modulation2", chan)
        self.encoder2 = N.seq(N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks), N.Conv2d(chan, chan, 3, 1, 1), lrelu())
        self.down2 = N.Conv2d(chan, chan * 2, 2, 2)
        chan = chan * 2
        gfm("encoder_modulation3", chan)
        self.encoder3 = N.seq(N.Conv2d(chan, chan, 3, 1, 1), N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks),
                              N.Conv2d(chan, chan, 3, 1, 1), lrelu())
        self.down3 = N.Conv2d(chan, chan * 2, 2, 2)
        chan = chan * 2
        gfm("middle_modulation", chan)
        self.middle = N.seq(N.Conv2d(chan, chan, 3, 1, 1), N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks * 2),
                            N.Conv2d(chan, chan, 3, 1, 1))
        for i in (3, 2, 1):
            setattr(self, f"up{i}", N.seq(N.Conv2d(chan, chan * 2, 1, bias=False), nn.PixelShuffle(2)))
            chan = chan // 2
            if dec_gfm:
                gfm(f"decoder_modulation{i}", chan)
            setattr(self, f"decoder{i}", N.seq(N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks), N.conv(chan, chan, mode='C')))
        self.tail = N.seq(N.conv(chan, chan * 4, mode='C'), nn.PixelShuffle(upscale_factor=2), N.conv(chan, 3, mode='C'))

    def _act_dtype(self) -> torch.dtype:
        return self.intro.weight.dtype

    def _run(self, a, cond, coord_nhwc, crop_hw=None):
        if self.coord_in_intro:                                          # intro(torch.cat([x[0], x[2]], dim=1)) as conv(raw) + conv(coord)
            w_raw, w_co = ops.split_conv_input_views(self.intro, (4, 2))
            intro = ops.conv2d(coord_nhwc, w_co, residual=ops.conv2d(a, w_raw))
        elif hasattr(self, "lsc"):
            intro = ops.lsc_chain(self.lsc, coord_nhwc, self.intro, a) if self.lsc.model[0].out_channels == self.intro.out_channels else None
            if intro is None:                                            # intro(raw) * (lsc + 1); lsc_chain fuses it for equal widths
                intro = self.intro._nhwc(a, mul_plus1=self.lsc._nhwc(coord_nhwc))
        else:
            intro = self.intro._nhwc(a)
        has_gfm = hasattr(self, "classifier")
        vec = self.classifier._vec(ops._req(cond, "cond")) if has_gfm else None

        def gfm(name, t):
            return getattr(self, name)._nhwc((t, vec))[0] if (has_gfm and hasattr(self, name)) else t

        def dec(i, t, skip):
            t = getattr(self, f"up{i}")._nhwc(t)
            if not self.skips:
                return getattr(self, f"decoder{i}")._nhwc(t)
            if has_gfm and hasattr(self, f"decoder_modulation{i}"):
                return ops.add(gfm(f"decoder_modulation{i}", getattr(self, f"decoder{i}")._nhwc(t)), skip)
            return getattr(self, f"decoder{i}")._nhwc(t, residual=skip)            # decoder's last conv takes the skip add

        d1 = self.down1._nhwc(self.encoder1._nhwc(gfm("encoder_modulation1", intro)))
        d2 = self.down2._nhwc(self.encoder2._nhwc(gfm("encoder_modulation2", d1)))
        d3 = self.down3._nhwc(self.encoder3._nhwc(gfm("encoder_modulation3", d2)))
        m = self.middle._nhwc(gfm("middle_modulation", d3), residual=d3 if self.skips else None)
        u1 = dec(1, dec(2, dec(3, m, d2), d1), intro)
        if ops.tail_fold_ok(u1, self.tail[0], self.tail[2]):          # no activation between the two tail convs: one folded 5x5 launch + border ring
            return ops.tail_fold(u1, self.tail[0], self.tail[2], crop_hw=crop_hw, out_dtype=self.output_dtype)
        t = self.tail[0]._nhwc(u1, out_mode=RC_OUT_PIXEL_SHUFFLE2)
        return self.tail[2]._nhwc(t, out_mode=RC_OUT_NCHW, crop_hw=crop_hw, out_dtype=self.output_dtype)

    def forward(self, x: Sequence[torch.Tensor]):
        raw = x if isinstance(x, torch.Tensor) else x[0]
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        if raw.dim() != 4 or raw.shape[1] != 4 or raw.shape[2] % 8 or raw.shape[3] % 8:
            raise ValueError(f"raw must be (B,4,H,W) with H,W multiples of 8 (three stride-2 levels), got {tuple(raw.shape)}")
        dt = self._act_dtype()
        co = None
        if hasattr(self, "lsc") or self.coord_in_intro:
            coord = x[2]
            if coord.shape[0] != raw.shape[0] or coord.shape[1] != 2 or coord.shape[2:] != raw.shape[2:]:
                raise ValueError(f"coord must be (B,2,H,W) matching raw, got {tuple(coord.shape)}")
            co = ops.to_nhwc(coord, dtype=dt)
        cond = (raw if self.cond_from_raw else x[1]) if hasattr(self, "classifier") else None
        return self._run(ops.to_nhwc(raw, dtype=dt), cond, co)

    def forward_mosaic(self, mosaic, cond=None, coord=None, pad_to: int = 16, black_level: float = 0.0, white_level: float = 1.0,
                       cond_hw=(256, 256)):
        """Bayer mosaic (B,1,2h,2w), cond, coord (B,2,h,w) -> sRGB (B,3,2h,2w) with the unshuffle / pad16 / crop front end
        (cond=None: see _DwtUNet.forward_mosaic)."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        dt = self._act_dtype()
        a, cond = _ingest(self, mosaic, cond, dt, pad_to, black_level, white_level, cond_hw)
        b, hp, wp, _ = a.shape
        co = None
        if hasattr(self, "lsc") or self.coord_in_intro:
            if coord is None or coord.shape[-2:] != (mosaic.shape[-2] // 2, mosaic.shape[-1] // 2):
                raise ValueError("coord must be at packed resolution (h, w)")
            co = ops.to_nhwc(coord, dtype=dt, pad_hw=(hp, wp))
        if hasattr(self, "classifier") and self.cond_from_raw:
            cond = ops.to_nchw(a)                             # the padded packed RAW, as upstream's x[0]
        return self._run(a, cond, co, crop_hw=(mosaic.shape[-2], mosaic.shape[-1]))


class ISPUNet_GFM_LSC(_StridedUNet):
    """upstream LiteISP.py:1228-1380: colour prior + Res_GFM on both sides + lens shading."""

    def __init__(self, cond_c=32, chan=32, m_blocks=2):
        super().__init__()
        self._build(chan=chan, cond_c=cond_c, lsc=True, m_blocks=m_blocks)


class ISPUNet_GFM(_StridedUNet):
    """upstream LiteISP.py:963-1110: ISPUNet_GFM_LSC without the lens-shading branch (x[2] is ignored)."""

    def __init__(self):
        super().__init__()
        self._build(chan=32, cond_c=32, lsc=False, m_blocks=2)


class ISPUNet_LSC(_StridedUNet):
    """upstream LiteISP.py:1113-1225: the strided U-Net with the lens-shading gain only (x[1] is ignored)."""

    def __init__(self):
        super().__init__()
        self._build(chan=32, cond_c=None, lsc=True)


class ResUNet(_StridedUNet):
    """upstream LiteISP.py:2038-2146: the bare strided U-Net (reads only x[0])."""

    def __init__(self):
        super().__init__()
        self._build(chan=32)


class ISPUNet_GFM_crop(_StridedUNet):
    """upstream LiteISP.py:811-960: ISPUNet_GFM at 64 channels, cond_c = 64, one Res_GFM per level, and the colour prior reading the
    packed RAW x[0] itself (:928) -- "its own crop as the global condition"."""
    cond_from_raw = True

    def __init__(self):
        super().__init__()
        self._build(chan=64, cond_c=64, lsc=False, m_blocks=1)


class ISPUNet_GFM_LSC1(_StridedUNet):
    """upstream LiteISP.py:1382-1532: the position code is concatenated to the RGGB planes (intro = Conv2d(6, 32, 3)) instead of going
    through a Lens_Shading_Correction branch."""
    coord_in_intro = True

    def __init__(self):
        super().__init__()
        self._build(chan=32, cond_c=32, lsc=False, m_blocks=2, intro_in=6)


class ISPUNet_GFM_LSC_noskip(_StridedUNet):
    """upstream LiteISP.py:2522-2652: ISPUNet_GFM_LSC with one Res_GFM per level on the way down only, no additive skips (middle,
    decoders) and a lens-shading MLP of hidden width lsc_c."""
    skips = False

    def __init__(self, cond_c=32, lsc_c=32):
        super().__init__()
        self._build(chan=32, cond_c=cond_c, lsc=True, m_blocks=1, lsc_nf=lsc_c, dec_gfm=False)


class ISPUNet_GFM_LFM(nn.Module):
    """upstream LiteISP.py:1535-1707: the strided U-Net with GLOBAL + LOCAL modulation -- Color_Condition_GFM_LFM gives a colour vector
    (from x[1]) and a condition map (3x3 conv of the packed RAW x[0]); CondNet1..4 bring the map to the four resolutions; every level's
    Res_GFM_LFM applies GFMLayer (vector) and SFTLayer (map).  Same construction order and attribute names as upstream."""

    output_dtype: Optional[torch.dtype] = None

    def __init__(self, cond_c=32, n_blocks=2, modulation_blocks=1, chan=32):
        super().__init__()
        self.cond_c = cond_c
        self.classifier = Color_Condition_GFM_LFM(in_channels=4, GFM_out_c=cond_c, LFM_out_c=cond_c)
        self.chan, self.n_blocks, self.modulation_blocks = chan, n_blocks, modulation_blocks

        def mod(c):
            return N.seq(*[Res_GFM_LFM(cond_c=cond_c, out_nc=c, nf=c * 2) for _ in range(modulation_blocks)])

        def lrelu():
            return nn.LeakyReLU(negative_slope=1e-1, inplace=True)

        self.intro = N.seq(N.Conv2d(4, chan, 3, 1, 1))
        self.encoder_modulation1 = mod(chan)
        self.encoder1 = N.seq(N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks), N.Conv2d(chan, chan, 3, 1, 1), lrelu())
        self.down1 = N.Conv2d(chan, chan * 2, 2, 2)
        chan = chan * 2
        self.encoder_modulation2 = mod(chan)
        self.encoder2 = N.seq(N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks), N.Conv2d(chan, chan, 3, 1, 1), lrelu())
        self.down2 = N.Conv2d(chan, chan * 2, 2, 2)
        chan = chan * 2
        self.encoder_modulation3 = mod(chan)
        self.encoder3 = N.seq(N.Conv2d(chan, chan, 3, 1, 1), N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks),
                              N.Conv2d(chan, chan, 3, 1, 1), lrelu())
        self.down3 = N.Conv2d(chan, chan * 2, 2, 2)
        chan = chan * 2
        self.middle_modulation = mod(chan)
        self.middle = N.seq(N.Conv2d(chan, chan, 3, 1, 1), N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks * 2),
                            N.Conv2d(chan, chan, 3, 1, 1))
        for i in (3, 2, 1):
            setattr(self, f"up{i}", N.seq(N.Conv2d(chan, chan * 2, 1, bias=False), nn.PixelShuffle(2)))
            chan = chan // 2
            setattr(self, f"decoder_modulation{i}", mod(chan))
            setattr(self, f"decoder{i}", N.seq(N.RCAGroup(in_channels=chan, out_channels=chan, nb=n_blocks), N.conv(chan, chan, mode='C')))
        self.tail = N.seq(N.conv(chan, chan * 4, mode='C'), nn.PixelShuffle(upscale_factor=2), N.conv(chan, 3, mode='C'))
        c = cond_c
        self.CondNet1 = N.Sequential(N.Conv2d(c, c, 1), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 1))
        self.CondNet2 = N.Sequential(N.Conv2d(c, c, 2, 2), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 1))
        self.CondNet3 = N.Sequential(N.Conv2d(c, c, 2, 2), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 2, 2), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 1))
        self.CondNet4 = N.Sequential(N.Conv2d(c, c, 2, 2), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 2, 2), nn.LeakyReLU(0.1, True),
                                     N.Conv2d(c, c, 2, 2), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 1))

    def _act_dtype(self) -> torch.dtype:
        return self.intro.weight.dtype

    def _run(self, a, cond, crop_hw=None):
        intro = self.intro._nhwc(a)
        vec, lfm = self.classifier._run(cond, a)
        l1, l2, l4, l8 = (getattr(self, f"CondNet{i}")._nhwc(lfm) for i in (1, 2, 3, 4))

        def mod(name, t, cmap):
            return getattr(self, name)._nhwc((t, vec, cmap))[0]

        d1 = self.down1._nhwc(self.encoder1._nhwc(mod("encoder_modulation1", intro, l1)))
        d2 = self.down2._nhwc(self.encoder2._nhwc(mod("encoder_modulation2", d1, l2)))
        d3 = self.down3._nhwc(self.encoder3._nhwc(mod("encoder_modulation3", d2, l4)))
        m = self.middle._nhwc(mod("middle_modulation", d3, l8), residual=d3)
        u3 = ops.add(mod("decoder_modulation3", self.decoder3._nhwc(self.up3._nhwc(m)), l4), d2)
        u2 = ops.add(mod("decoder_modulation2", self.decoder2._nhwc(self.up2._nhwc(u3)), l2), d1)
        u1 = ops.add(mod("decoder_modulation1", self.decoder1._nhwc(self.up1._nhwc(u2)), l1), intro)
        if ops.tail_fold_ok(u1, self.tail[0], self.tail[2]):          # no activation between the two tail convs: one folded 5x5 launch + border ring
            return ops.tail_fold(u1, self.tail[0], self.tail[2], crop_hw=crop_hw, out_dtype=self.output_dtype)
        t = self.tail[0]._nhwc(u1, out_mode=RC_OUT_PIXEL_SHUFFLE2)
        return self.tail[2]._nhwc(t, out_mode=RC_OUT_NCHW, crop_hw=crop_hw, out_dtype=self.output_dtype)

    def forward(self, x: Sequence[torch.Tensor]):
        raw, cond = x[0], x[1]
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        if raw.dim() != 4 or raw.shape[1] != 4 or raw.shape[2] % 8 or raw.shape[3] % 8:
            raise ValueError(f"raw must be (B,4,H,W) with H,W multiples of 8 (three stride-2 levels), got {tuple(raw.shape)}")
        return self._run(ops.to_nhwc(raw, dtype=self._act_dtype()), cond)

    def forward_mosaic(self, mosaic, cond=None, coord=None, pad_to: int = 16, black_level: float = 0.0, white_level: float = 1.0,
                       cond_hw=(256, 256)):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        a, cond = _ingest(self, mosaic, cond, self._act_dtype(), pad_to, black_level, white_level, cond_hw)
        return self._run(a, cond, crop_hw=(mosaic.shape[-2], mosaic.shape[-1]))


class LiteISPNet_GFM_LSC_GMA(LiteISPNet_GFM_LSC):
    """BUILD-DEFINED composition for BASELINE.json config 3 ("4K RAW->sRGB with GroupMix attention").

    Upstream defines GMA_Block (models/groupmix.py:274-299, copy at models/raw2bit.py:117-142) but wires it into
    no full model -- only the smoke test `test_gma` (models/raw2bit.py:4361-4367) instantiates it.  This class
    attaches one GMA_Block(dim=80, heads=8) as a residual refinement of the H/2-level feature d1 (192 ch):
        d1 <- d1 + gma_out( GMA_Block( gma_in(d1) ) ),   gma_in: Conv1x1 192->80, gma_out: Conv1x1 80->192
    i.e. N = (H/2)(W/2) tokens (522 240 at 4K), the placement SURVEY.md section 8d names.  Everything else is
    LiteISPNet_GFM_LSC unchanged (the extra modules are constructed last, so the base parameters keep the
    reference's seed-0 values).  oracle/liteisp_oracle.py restates the same composition for parity."""

    def __init__(self, gma_dim: int = 80, gma_heads: int = 8):
        super().__init__()
        c1 = self.down1[3].weight.shape[0]            # 4 * ch_1 = 192
        self.gma_in = N.Conv2d(c1, gma_dim, 1, 1, 0)
        self.gma = GMA_Block(gma_dim, gma_heads)
        self.gma_out = N.Conv2d(gma_dim, c1, 1, 1, 0)

    def _refine_d1(self, d1):
        return self.gma._nhwc(d1, pre=self.gma_in, post=(self.gma_out, d1))          # gma_in + ConvPosEnc in the block's first launch, gma_out(GMA(.)) + d1 in its last
