"""Host mirror of the RAW codec's own blocks and of `raw_compression_tcm` / `raw_compression_tcm_final` (SURVEY.md rows a18/a19;
upstream models/raw2bit.py:117-355, 361-727, 730-886, 1614-2027, 3181-3206): same class names, constructor signatures and attribute names, NCHW tensors at
the module boundary, NHWC inside, every op through librealcam_hip.so.

What is pinned and what is not is the same as in realcamnet_amd/tcm.py: upstream's own composition (these classes' forward
logic) is checked against fixtures produced by running the reference classes; the CompressAI layers underneath them
(`ResidualBlockWithStride`, `GDN`, `AttentionBlock`, entropy models ...) are restated and parity-unpinned.  `forward` (eval mode), `update`,
`compress` and `decompress` exist; training does not.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import networks as N
from . import ops
from . import tcm as T
from .LiteISP import Color_Condition_GFM, Lens_Shading_Correction, Res_GFM
from .groupmix import GMA_Block, Mlp
from .tcm import (Block, ConvTransBlock, EntropyBottleneck, GaussianConditional, ResidualBlock, ResidualBlockUpsample,
                  ResidualBlockWithStride, SWAtten, _slice_loop, conv1x1, conv3x3, slice_transform, subpel_conv3x3)


class CALayer(nn.Module):
    """Channel attention with bias-free Linear layers (upstream models/raw2bit.py:238-254).  Executed fused inside
    ResidualBlockWithCA: the producing conv emits the channel sums, rc_ca_gate the gate."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())


class ResidualBlockWithCA(nn.Module):
    """CA(conv3x3(LeakyReLU(conv3x3(x)))) + skip(x)   (upstream models/raw2bit.py:257-289)."""

    def __init__(self, in_ch: int, out_ch: int, redution=8):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.ca = CALayer(out_ch, redution)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def _nhwc(self, a):
        identity = a if self.skip is None else self.skip._nhwc(a)
        if ops.EARLY_GATE and ops.gate_ahead_ok(self.conv1, self.conv2, self.ca):
            # the gate of conv2's output from conv1's channel sums, ahead of conv2 (ops.ca_gate_ahead); conv2's epilogue writes conv2(t) * gate + skip
            t, sums = self.conv1._nhwc(a, act="leaky", slope=float(self.leaky_relu.negative_slope), want_sums=True)
            return self.conv2._nhwc(t, out_scale=ops.ca_gate_ahead(sums, t, self.conv2, self.ca), residual=identity)
        t = self.conv1._nhwc(a, act="leaky", slope=float(self.leaky_relu.negative_slope))
        r, sums = self.conv2._nhwc(t, want_sums=True)
        gate = ops.ca_gate_linear(sums, a.shape[1] * a.shape[2], self.ca.fc[0], self.ca.fc[2])
        return ops.gate_residual(r, gate, identity)

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class SpatialFeatureTransform(nn.Module):
    """x * scale(cond) + shift(cond) (+ x)   (upstream models/raw2bit.py:860-885)."""

    def __init__(self, cond_channels=64, n_features=64, ada_method='vanilla', residual=True):
        super().__init__()
        if ada_method != 'vanilla' or not residual:
            raise NotImplementedError("SpatialFeatureTransform: the 'vanilla', residual=True form is the one upstream instantiates")
        self.cond_scale = nn.Sequential(N.Conv2d(cond_channels, n_features, 3, stride=1, padding=1), nn.ReLU(inplace=True),
                                        N.Conv2d(n_features, n_features, 3, stride=1, padding=1))
        self.cond_shift = nn.Sequential(N.Conv2d(cond_channels, n_features, 3, stride=1, padding=1), nn.ReLU(inplace=True),
                                        N.Conv2d(n_features, n_features, 3, stride=1, padding=1))
        self.residual = residual

    def _nhwc(self, a, cond, identity=None):
        scale = self.cond_scale[2]._nhwc(self.cond_scale[0]._nhwc(cond, act="relu"))
        shift = self.cond_shift[2]._nhwc(self.cond_shift[0]._nhwc(cond, act="relu"))
        return ops.sft_apply(a, scale, shift, identity)

    def forward(self, x, cond):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x), ops.to_nhwc(cond)))


class ConvTransBlock_mzj(nn.Module):
    """ConvTransBlock whose conv branch is ResidualBlockWithCA followed by the spatial feature transform on the local RAW
    condition (upstream models/raw2bit.py:292-328).  Takes and returns the pair (x, cond) so it chains in nn.Sequential."""

    def __init__(self, conv_dim, trans_dim, head_dim, window_size, drop_path, type='W'):
        super().__init__()
        assert type in ['W', 'SW']
        self.conv_dim, self.trans_dim, self.head_dim, self.window_size, self.drop_path, self.type = \
            conv_dim, trans_dim, head_dim, window_size, drop_path, type
        self.num_head = trans_dim // head_dim
        self.trans_block = Block(trans_dim, trans_dim, head_dim, window_size, drop_path, type)
        self.conv1_1 = N.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv1_2 = N.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv_block = ResidualBlockWithCA(conv_dim, conv_dim, 8)
        self.spatial_transform = SpatialFeatureTransform(cond_channels=conv_dim, n_features=conv_dim)

    def _nhwc(self, xx):
        a, cond = xx
        va, vb = ops.split_conv_views(self.conv1_1, (self.conv_dim, self.trans_dim))     # torch.split(conv1_1(x)) without the slice copies
        conv_x, trans_x = ops.conv2d(a, va), ops.conv2d(a, vb)
        conv_x = self.spatial_transform._nhwc(self.conv_block._nhwc(conv_x), cond, identity=conv_x)
        trans_x = self.trans_block(trans_x)
        y = ops.cat_linear(conv_x, trans_x, self.conv1_2, residual=a)          # conv1_2(cat(conv_x, trans_x)) + x without the concatenated map
        return y if y is not None else self.conv1_2._nhwc(ops.channel_concat([conv_x, trans_x]), residual=a), cond

    def forward(self, xx):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        y, _ = self._nhwc((ops.to_nhwc(xx[0]), ops.to_nhwc(xx[1])))
        return ops.to_nchw(y), xx[1]


class GMABlock(nn.Module):
    """Two GroupMix blocks over the tokens of an NCHW map (upstream models/raw2bit.py:168-184; its GMA_Block, :117-143, is the
    composition of models/groupmix.py:274-299 re-declared over the imported ConvPosEnc / EfficientAtt -- the same mirror class serves)."""

    def __init__(self, input_dim, head_dim, drop_path) -> None:
        super().__init__()
        self.input_dim = input_dim
        self.num_head = input_dim // head_dim
        self.block_1 = GMA_Block(input_dim, self.num_head, drop_path_rate=drop_path)
        self.block_2 = GMA_Block(input_dim, self.num_head, drop_path_rate=drop_path)

    def _nhwc(self, a):
        return self.block_2._nhwc(self.block_1._nhwc(a))

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class GMAAtten(T.AttentionBlock):
    """SWAtten with the GroupMix pair as its non-local branch (upstream models/raw2bit.py:209-234)."""

    def __init__(self, input_dim, output_dim, head_dim, drop_path, inter_dim=192) -> None:
        if inter_dim is not None:
            super().__init__(inter_dim)
            self.num_head, self.head_dim = inter_dim // head_dim, head_dim
            self.non_local_block = GMABlock(inter_dim, head_dim, drop_path=drop_path)
            self.in_conv = conv1x1(input_dim, inter_dim)
            self.out_conv = conv1x1(inter_dim, output_dim)
        else:
            super().__init__(input_dim)
            self.num_head, self.head_dim = input_dim // head_dim, head_dim
            self.non_local_block = GMABlock(input_dim, self.num_head, drop_path=drop_path)      # (sic) upstream passes num_head as head_dim
            self.in_conv = self.out_conv = None

    def _nhwc(self, a):
        if self.in_conv is None:
            raise AttributeError("GMAAtten without inter_dim has no in_conv (same as upstream, raw2bit.py:225)")
        x = self.in_conv._nhwc(a)
        z = self.non_local_block._nhwc(x)
        return self.out_conv._nhwc(ops.sigmoid_gate_add(self._branch_a(x), self._branch_b(z), x))

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class ConvGMABlock(nn.Module):
    """ConvTransBlock with a GroupMix block on the transformer half (upstream models/raw2bit.py:330-355)."""

    def __init__(self, conv_dim, trans_dim, head_dim, drop_path=0.):
        super().__init__()
        self.conv_dim, self.trans_dim, self.head_dim, self.drop_path = conv_dim, trans_dim, head_dim, drop_path
        self.num_head = trans_dim // head_dim
        self.trans_block = GMA_Block(trans_dim, self.num_head, drop_path_rate=drop_path)
        self.conv1_1 = N.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv1_2 = N.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv_block = ResidualBlock(conv_dim, conv_dim)

    def _nhwc(self, a):
        va, vb = ops.split_conv_views(self.conv1_1, (self.conv_dim, self.trans_dim))     # torch.split(conv1_1(x)) without the slice copies
        conv_x, trans_x = ops.conv2d(a, va), ops.conv2d(a, vb)
        rb = self.conv_block._nhwc(conv_x)
        trans_x = self.trans_block._nhwc(trans_x)
        # conv1_2(cat(conv_block(conv_x) + conv_x, trans_x)) + x: the sum and the concatenation happen on the closing launch's operand load
        y = ops.cat_linear(rb, trans_x, self.conv1_2, residual=a, a_add=conv_x)
        return y if y is not None else self.conv1_2._nhwc(ops.channel_concat([ops.add(rb, conv_x), trans_x]), residual=a)

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class RBU(nn.Module):
    """ResidualBlockUpsample without the IGDN (upstream models/raw2bit.py:3181-3206): conv3x3(LeakyReLU(subpel(x))) + subpel'(x)."""

    def __init__(self, in_ch: int, out_ch: int, upsample: int = 2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def _nhwc(self, a):
        slope = float(self.leaky_relu.negative_slope)
        if (self.subpel_conv[0].out_channels // 4) % 16 == 0:
            t = self.subpel_conv[0]._nhwc(a, act="leaky", slope=slope, out_mode=N.RC_OUT_PIXEL_SHUFFLE2)
        else:
            t = ops.pixel_shuffle2(self.subpel_conv[0]._nhwc(a, act="leaky", slope=slope))
        return self.conv._nhwc(t, residual=self.upsample._nhwc(a))

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class HyCondModConvBlock(nn.Module):
    """conv + ReLU (upstream models/raw2bit.py:730-744; only the default 'relu' form is instantiated upstream)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, act='relu'):
        super().__init__()
        if act != 'relu':
            raise NotImplementedError("HyCondModConvBlock: act='relu' only")
        self.conv = N.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.act = nn.ReLU(inplace=True)

    def _nhwc(self, a):
        return self.conv._nhwc(a, act="relu")


class HyCondModEncBlock(nn.Module):
    """stride-2 conv block, conv block (upstream models/raw2bit.py:746-767)."""

    def __init__(self, in_channels, out_channels, downscale_method='stride'):
        super().__init__()
        if downscale_method != 'stride':
            raise NotImplementedError("HyCondModEncBlock: 'stride' only")
        self.down = HyCondModConvBlock(in_channels, out_channels, stride=2)
        self.conv = HyCondModConvBlock(out_channels, out_channels)

    def _nhwc(self, a):
        return self.conv._nhwc(self.down._nhwc(a))


class HyCondModDecBlock(nn.Module):
    """bilinear x2 (align_corners) -> conv block; cat([skip, up]) -> conv block (upstream models/raw2bit.py:781-813)."""

    def __init__(self, in_channels, out_channels, upscale_method='bilinear'):
        super().__init__()
        if upscale_method != 'bilinear':
            raise NotImplementedError("HyCondModDecBlock: 'bilinear' only")
        self.up = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True), HyCondModConvBlock(in_channels, out_channels))
        self.conv = HyCondModConvBlock(in_channels, out_channels)

    def _nhwc(self, a1, a2):
        up = self.up[1]._nhwc(ops.upsample_bilinear2(a1))
        return self.conv._nhwc(ops.channel_concat([a2, up]))


class HybridConditionModule(nn.Module):
    """Small U-Net on the packed RAW giving three local condition maps at 1/2, 1/4, 1/8 (upstream models/raw2bit.py:817-858)."""

    def __init__(self, in_channels=4, out_channels=64, init_mid_channels=16, down_method='stride', up_method='bilinear'):
        super().__init__()
        m = init_mid_channels
        self.in_conv = HyCondModConvBlock(in_channels, m)
        self.enc_1 = HyCondModEncBlock(m, m * 2, down_method)
        self.enc_2 = HyCondModEncBlock(m * 2, m * 4, down_method)
        self.enc_3 = HyCondModEncBlock(m * 4, m * 8, down_method)
        self.dec_1 = HyCondModDecBlock(m * 8, m * 4, up_method)
        self.dec_2 = HyCondModDecBlock(m * 4, m * 2, up_method)
        self.dec_3 = HyCondModDecBlock(m * 2, m, up_method)
        self.out_conv = HyCondModConvBlock(m, out_channels)
        c = out_channels
        self.CondNet1 = nn.Sequential(N.Conv2d(c, c, 3, 2, 1), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 1))
        self.CondNet2 = nn.Sequential(N.Conv2d(c, c, 3, 2, 1), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 3, 2, 1))
        self.CondNet3 = nn.Sequential(N.Conv2d(c, c, 3, 2, 1), nn.LeakyReLU(0.1, True), N.Conv2d(c, c, 3, 2, 1), nn.LeakyReLU(0.1, True),
                                      N.Conv2d(c, c, 3, 2, 1))

    @staticmethod
    def _cond(net, y, s2d=None):
        """s2d: the space-to-depth map of y, shared by the three CondNets (each starts with a stride-2 3x3 conv of the same 64-channel
        full-resolution map: three separate 2.3 GB re-layout passes at 4K x 4 otherwise)."""
        mods = list(net)
        i = 0
        while i < len(mods):
            kw = {}
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                kw = dict(act="leaky", slope=float(mods[i + 1].negative_slope))
            if i == 0 and s2d is not None and tuple(mods[0].stride) == (2, 2) and mods[0].kernel_size[0] == 3 and ops.is_stride2(mods[0]):
                y = ops.conv_stride2(y, mods[0], s2d=s2d, **kw)
            else:
                y = mods[i]._nhwc(y, **kw)
            i += 2 if kw else 1
        return y

    def _nhwc(self, a):
        x1 = self.in_conv._nhwc(a)
        x2 = self.enc_1._nhwc(x1)
        x3 = self.enc_2._nhwc(x2)
        x4 = self.enc_3._nhwc(x3)
        y = self.dec_1._nhwc(x4, x3)
        y = self.dec_2._nhwc(y, x2)
        y = self.dec_3._nhwc(y, x1)
        y = self.out_conv._nhwc(y)
        # one shared map only where the 2x2-window kernel cannot read y directly (ops.FOLD_STRIDE2: 64 | channels)
        s2d = None if (ops.FOLD_STRIDE2 and y.dtype == torch.bfloat16 and y.shape[-1] % 64 == 0) else ops.space_to_depth2(y)
        return [self._cond(self.CondNet1, y, s2d), self._cond(self.CondNet2, y, s2d), self._cond(self.CondNet3, y, s2d)]

    def forward(self, x):
        if x.shape[-1] % 8 or x.shape[-2] % 8:
            raise ValueError("HybridConditionModule: H and W must be multiples of 8 (three stride-2 stages with skip concatenation)")
        return [ops.to_nchw(t) for t in self._nhwc(ops.to_nhwc(x))]


class raw_compression_tcm_final(nn.Module):
    """The RAW codec (upstream models/raw2bit.py:1614-1855), likelihood path: x = [raw (B,4,H,W), cond (B,4,h,w), coord (B,2,H,W)]
    -> {"x_hat" (B,3,2H,2W), "y", "lft", "lsc", "likelihoods": {"y","z"}, "para": {"means","scales","y"}}; `update`, `compress`,
    `decompress` as upstream (GPU rANS coder, realcamnet_amd/bitstream.py).  Same attribute names and entropy-model buffers as
    upstream: a reference checkpoint loads with strict=True."""

    def __init__(self, config=[2, 2, 2, 2, 2, 2, 2], head_dim=[8, 16, 32, 32, 16, 8, 8], drop_path_rate=0, N=64, M=320, num_slices=5,
                 max_support_slices=5, **kwargs):
        super().__init__()
        if drop_path_rate != 0:
            raise NotImplementedError("inference path: drop_path_rate must be 0")
        self.config, self.head_dim, self.window_size = config, head_dim, 8
        self.num_slices, self.max_support_slices, self.M = num_slices, max_support_slices, M
        dim, ws = N, self.window_size
        n2 = 2 * N
        cond_c = 128
        self.classifier = Color_Condition_GFM(in_channels=4, out_c=cond_c)
        self.lsc = Lens_Shading_Correction(in_channels=2, out_c=n2, nf=n2)
        self.local_condition = HybridConditionModule(out_channels=N, init_mid_channels=16)
        self.conv_first = conv3x3(4, n2)
        self.conv_down = ResidualBlockWithStride(n2, n2, 2)

        def mzj(n, hd):
            return nn.Sequential(*[ConvTransBlock_mzj(dim, dim, hd, ws, 0, 'W' if not i % 2 else 'SW') for i in range(n)])

        def ctb(n, hd, w=ws):
            return [ConvTransBlock(dim, dim, hd, w, 0, 'W' if not i % 2 else 'SW') for i in range(n)]

        self.gfm1 = nn.Sequential(Res_GFM(in_nc=n2, chan=n2, cond_c=cond_c, nf=4 * N))
        self.m_down1 = mzj(config[0], head_dim[0])
        self.m_down1_down = ResidualBlockWithStride(n2, n2, stride=2)
        self.gfm2 = nn.Sequential(Res_GFM(in_nc=n2, chan=n2, cond_c=cond_c, nf=4 * N))
        self.m_down2 = mzj(config[1], head_dim[1])
        self.m_down2_down = ResidualBlockWithStride(n2, n2, stride=2)
        self.gfm3 = nn.Sequential(Res_GFM(in_nc=n2, chan=n2, cond_c=cond_c, nf=4 * N))
        self.m_down3 = mzj(config[2], head_dim[2])
        self.m_down3_down = conv3x3(n2, M, stride=2)
        self.g_s = N_seq(*[ResidualBlockUpsample(M, n2, 2)] + ctb(config[3], head_dim[3]) + [ResidualBlockUpsample(n2, n2, 2)] +
                         ctb(config[4], head_dim[4]) + [ResidualBlockUpsample(n2, n2, 2)] + ctb(config[5], head_dim[5]) +
                         [subpel_conv3x3(n2, n2, 2)] + [ResidualBlock(n2, n2), subpel_conv3x3(n2, 3, 2)])
        self.h_a = N_seq(*[ResidualBlockWithStride(320, n2, 2)] + ctb(config[0], 32, 4) + [conv3x3(n2, 192, stride=2)])
        self.h_mean_s = N_seq(*[ResidualBlockUpsample(192, n2, 2)] + ctb(config[3], 32, 4) + [subpel_conv3x3(n2, 320, 2)])
        self.h_scale_s = N_seq(*[ResidualBlockUpsample(192, n2, 2)] + ctb(config[3], 32, 4) + [subpel_conv3x3(n2, 320, 2)])
        width = lambda i, cap: 320 + (320 // num_slices) * min(i, cap)
        self.atten_mean = nn.ModuleList(nn.Sequential(SWAtten(width(i, 5), width(i, 5), 16, ws, 0, inter_dim=128)) for i in range(num_slices))
        self.atten_scale = nn.ModuleList(nn.Sequential(SWAtten(width(i, 5), width(i, 5), 16, ws, 0, inter_dim=128)) for i in range(num_slices))
        self.cc_mean_transforms = nn.ModuleList(slice_transform(width(i, 5), 320 // num_slices) for i in range(num_slices))
        self.cc_scale_transforms = nn.ModuleList(slice_transform(width(i, 5), 320 // num_slices) for i in range(num_slices))
        self.lrp_transforms = nn.ModuleList(slice_transform(width(i + 1, 6), 320 // num_slices) for i in range(num_slices))
        self.entropy_bottleneck = EntropyBottleneck(192)
        self.gaussian_conditional = GaussianConditional(None)

    def _act_dtype(self):
        return self.conv_first.weight.dtype

    update = T._codec_update
    load_state_dict = T._codec_load_state_dict

    def _latent(self, x):
        raw, cond, coord = x[0], x[1], x[2]
        dt = self._act_dtype()
        return self._analysis(ops.to_nhwc(raw, dtype=dt), cond, ops.to_nhwc(coord, dtype=dt))[0]

    def compress(self, x, fmt: str = "chunked", chunk: int = T.bitstream.DEFAULT_CHUNK, graph: bool = False):
        """upstream models/raw2bit.py:1876-1944: x = [raw, cond, coord] -> {"strings": [y_strings, z_strings], "shape"} (one string per
        image; fmt "chunked": GPU coder, "compressai": one CompressAI-layout stream per image).  graph (chunked only): the analysis transform and the
        slice loop replayed as ONE HIP graph captured per input shape (tcm._codec_compress_graphed): the same strings in less host time at low batch."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        if graph and fmt == "chunked":
            return T._codec_compress_graphed(self, lambda r, c, k: self._latent([r, c, k]), [ops._req(t, "x") for t in x[:3]], chunk)
        return T._codec_compress(self, self._latent(x), fmt, chunk)

    def decompress(self, strings, shape, fmt: str = "chunked"):
        """upstream models/raw2bit.py:1961-2027: -> {"x_hat": (B,3,2H,2W) clamped to [0, 1]}."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        return {"x_hat": T._codec_decompress(self, strings, shape, self._act_dtype(), fmt)}

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        raw, cond, coord = x[0], x[1], x[2]
        dt = self._act_dtype()
        return self._forward_nhwc(ops.to_nhwc(raw, dtype=dt), cond, ops.to_nhwc(coord, dtype=dt))

    def forward_mosaic(self, mosaic, cond, coord, pad_to: int = 128, black_level: float = 0.0, white_level: float = 1.0, cond_hw=(256, 256)):
        """Bayer mosaic (B,1,2h,2w), cond (B,4,hc,wc), coord (B,2,h,w) -> the same dict as forward().  The packed RAW and coord are
        zero-padded bottom/right to a multiple of `pad_to` (128: window 4 at 1/32 of the packed size), x_hat
        is NOT cropped (it is the decoder's output for the padded frame)."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        dt = self._act_dtype()
        from .LiteISP import _ingest
        a, cond = _ingest(self, mosaic, cond, dt, pad_to, black_level, white_level, cond_hw)    # cond=None: resized packed RAW
        if coord.shape[-2:] != (mosaic.shape[-2] // 2, mosaic.shape[-1] // 2):
            raise ValueError("coord must be at packed resolution (h, w)")
        return self._forward_nhwc(a, cond, ops.to_nhwc(coord, dtype=dt, pad_hw=(a.shape[1], a.shape[2])))

    def _analysis(self, a, cond, coord_nhwc):
        """packed RAW NHWC -> (latent y NHWC, local condition maps, lens-shading map): the encoder half of forward / compress."""
        lsc_fea = self.lsc._nhwc(coord_nhwc)
        vec = self.classifier._vec(ops._req(cond, "cond"))
        local = self.local_condition._nhwc(a)
        fea = self.conv_first._nhwc(a, mul_plus1=lsc_fea)                 # conv_first(raw) * (lsc + 1)
        fea = self.conv_down._nhwc(fea)
        for gfm, blocks, down, c in ((self.gfm1, self.m_down1, self.m_down1_down, local[0]), (self.gfm2, self.m_down2, self.m_down2_down, local[1]),
                                     (self.gfm3, self.m_down3, self.m_down3_down, local[2])):
            fea, _ = gfm[0]._nhwc((fea, vec))
            for blk in blocks:
                fea, _ = blk._nhwc((fea, c))
            fea = down._nhwc(fea)
        return fea, local, lsc_fea

    def _forward_nhwc(self, a, cond, coord_nhwc):
        fea, local, lsc_fea = self._analysis(a, cond, coord_nhwc)
        out = _slice_loop(self, fea)
        out.update({"y": out["para"]["y"], "lft": T.nchw_view(local[2]), "lsc": T.nchw_view(lsc_fea)})
        return out


class raw_compression_tcm(raw_compression_tcm_final):
    """The first RAW codec (upstream models/raw2bit.py:361-727): as the final one but without the local (hybrid) condition -- plain
    ConvTransBlocks in the analysis transform, colour prior of 64 channels -- and `forward` returns only x_hat / likelihoods / para
    (:491-579).  upstream's `compress` (:600) calls a `g_a` the class never builds; here it runs the analysis path of `forward` on
    x = [raw, cond, coord], like the final model's compress."""

    def __init__(self, config=[2, 2, 2, 2, 2, 2, 2], head_dim=[8, 16, 32, 32, 16, 8, 8], drop_path_rate=0, N=64, M=320, num_slices=5,
                 max_support_slices=5, **kwargs):
        nn.Module.__init__(self)
        if drop_path_rate != 0:
            raise NotImplementedError("inference path: drop_path_rate must be 0")
        self.config, self.head_dim, self.window_size = config, head_dim, 8
        self.num_slices, self.max_support_slices, self.M = num_slices, max_support_slices, M
        dim, ws = N, self.window_size
        n2 = 2 * N
        cond_c = 64
        self.classifier = Color_Condition_GFM(in_channels=4, out_c=cond_c)
        self.lsc = Lens_Shading_Correction(in_channels=2, out_c=n2, nf=n2)
        self.conv_first = conv3x3(4, n2)
        self.conv_down = ResidualBlockWithStride(n2, n2, 2)

        def ctb(n, hd, w=ws):
            return [ConvTransBlock(dim, dim, hd, w, 0, 'W' if not i % 2 else 'SW') for i in range(n)]

        self.gfm1 = nn.Sequential(Res_GFM(in_nc=n2, chan=n2, cond_c=cond_c, nf=4 * N))
        self.m_down1 = N_seq(*ctb(config[0], head_dim[0]) + [ResidualBlockWithStride(n2, n2, stride=2)])
        self.gfm2 = nn.Sequential(Res_GFM(in_nc=n2, chan=n2, cond_c=cond_c, nf=4 * N))
        self.m_down2 = N_seq(*ctb(config[1], head_dim[1]) + [ResidualBlockWithStride(n2, n2, stride=2)])
        self.gfm3 = nn.Sequential(Res_GFM(in_nc=n2, chan=n2, cond_c=cond_c, nf=4 * N))
        self.m_down3 = N_seq(*ctb(config[2], head_dim[2]) + [conv3x3(n2, M, stride=2)])
        self.g_s = N_seq(*[ResidualBlockUpsample(M, n2, 2)] + ctb(config[3], head_dim[3]) + [ResidualBlockUpsample(n2, n2, 2)] +
                         ctb(config[4], head_dim[4]) + [ResidualBlockUpsample(n2, n2, 2)] + ctb(config[5], head_dim[5]) +
                         [subpel_conv3x3(n2, n2, 2)] + [ResidualBlock(n2, n2), subpel_conv3x3(n2, 3, 2)])
        self.h_a = N_seq(*[ResidualBlockWithStride(320, n2, 2)] + ctb(config[0], 32, 4) + [conv3x3(n2, 192, stride=2)])
        self.h_mean_s = N_seq(*[ResidualBlockUpsample(192, n2, 2)] + ctb(config[3], 32, 4) + [subpel_conv3x3(n2, 320, 2)])
        self.h_scale_s = N_seq(*[ResidualBlockUpsample(192, n2, 2)] + ctb(config[3], 32, 4) + [subpel_conv3x3(n2, 320, 2)])
        width = lambda i, cap: 320 + (320 // num_slices) * min(i, cap)
        self.atten_mean = nn.ModuleList(nn.Sequential(SWAtten(width(i, 5), width(i, 5), 16, ws, 0, inter_dim=128)) for i in range(num_slices))
        self.atten_scale = nn.ModuleList(nn.Sequential(SWAtten(width(i, 5), width(i, 5), 16, ws, 0, inter_dim=128)) for i in range(num_slices))
        self.cc_mean_transforms = nn.ModuleList(slice_transform(width(i, 5), 320 // num_slices) for i in range(num_slices))
        self.cc_scale_transforms = nn.ModuleList(slice_transform(width(i, 5), 320 // num_slices) for i in range(num_slices))
        self.lrp_transforms = nn.ModuleList(slice_transform(width(i + 1, 6), 320 // num_slices) for i in range(num_slices))
        self.entropy_bottleneck = EntropyBottleneck(192)
        self.gaussian_conditional = GaussianConditional(None)

    def _analysis(self, a, cond, coord_nhwc):
        lsc_fea = self.lsc._nhwc(coord_nhwc)
        vec = self.classifier._vec(ops._req(cond, "cond"))
        fea = self.conv_down._nhwc(self.conv_first._nhwc(a, mul_plus1=lsc_fea))
        for gfm, stage in ((self.gfm1, self.m_down1), (self.gfm2, self.m_down2), (self.gfm3, self.m_down3)):
            fea = stage._nhwc(gfm[0]._nhwc((fea, vec))[0])
        return fea, None, lsc_fea

    def _forward_nhwc(self, a, cond, coord_nhwc):
        return _slice_loop(self, self._analysis(a, cond, coord_nhwc)[0])


def N_seq(*mods):
    return N.Sequential(*mods)
