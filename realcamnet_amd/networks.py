"""HIP-backed mirror of the reference block library (upstream models/networks.py).

Same factory names (`seq`, `conv`), class names, constructor signatures and sub-module attribute
names as the reference, so `state_dict()` keys are identical and reference checkpoints load with
`load_state_dict(strict=True)`.  The difference is what `forward` does: every module runs through
librealcam_hip.so on NHWC tensors (`_nhwc` methods); `forward(x)` keeps the reference's NCHW
signature by converting at the boundary.  Nothing here ever calls an ATen/MIOpen compute kernel.

Only the modes the RAW->sRGB path uses are built; the rest raise
NotImplementedError instead of silently running a non-native path.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from ._lib import RC_OUT_NHWC, RC_OUT_PIXEL_SHUFFLE2


def _is_act(m) -> bool:
    return isinstance(m, (nn.ReLU, nn.LeakyReLU)) or (isinstance(m, nn.GELU) and m.approximate == "none")


def _act_args(m):
    if isinstance(m, nn.ReLU):
        return dict(act="relu")
    if isinstance(m, nn.GELU):          # the codec's slice transforms (upstream models/tcm.py:398-425)
        return dict(act="gelu")
    return dict(act="leaky", slope=float(m.negative_slope))


class HipModule(nn.Module):
    """Base: NCHW `forward` wrapper around the NHWC `_nhwc` implementation."""

    def forward(self, x):
        a = ops.to_nhwc(x)
        return ops.to_nchw(self._nhwc(a))

    def _nhwc(self, a):  # pragma: no cover - abstract
        raise NotImplementedError


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose forward is the MFMA implicit-GEMM kernel (rc_conv2d).
    Replaces networks.conv mode 'C' (upstream models/networks.py:151-160)."""

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))

    def _nhwc(self, a, **fuse):
        if ops.is_down2x2(self):                      # ISPUNet family: Conv2d(c, 2c, 2, 2) (upstream LiteISP.py:1253)
            return ops.conv2x2s2(a, self, **fuse)
        if tuple(self.stride) == (2, 2) and ops.is_stride2(self):   # codec transforms: conv3x3 / conv1x1 with stride 2
            return ops.conv_stride2(a, self, **fuse)
        return ops.conv2d(a, self, **fuse)


class Sequential(nn.Sequential):
    """nn.Sequential with a fusing NHWC executor: Conv2d [+ ReLU/LeakyReLU] [+ PixelShuffle(2)] become
    one rc_conv2d launch; tuple-in/tuple-out children (Res_GFM) are chained as upstream does."""

    def forward(self, x):
        if isinstance(x, (tuple, list)):
            a = (ops.to_nhwc(x[0]), *x[1:])
            y = self._nhwc(a)
            return (ops.to_nchw(y[0]), *y[1:])
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))

    def _nhwc(self, a, residual=None):
        mods = list(self)
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            if isinstance(m, Conv2d):
                kw = {}
                j = i + 1
                if j < n and _is_act(mods[j]):
                    kw.update(_act_args(mods[j]))
                    j += 1
                shuffle_after = False
                if j < n and isinstance(mods[j], nn.PixelShuffle):
                    if mods[j].upscale_factor != 2:
                        raise NotImplementedError("PixelShuffle: only upscale_factor=2 is on the hot path")
                    if (m.out_channels // 4) % 16 == 0:
                        kw["out_mode"] = RC_OUT_PIXEL_SHUFFLE2
                    else:                             # narrow tails (the codec's subpel_conv3x3(2N, 3, 2)): separate shuffle
                        shuffle_after = True
                    j += 1
                if (j < n and isinstance(mods[j], DWTForward) and "out_mode" not in kw and not shuffle_after and
                        ops.conv_dwt_ok(a, m, mods[j], kw.get("act"), kw.get("slope", 0.0))):
                    kw["out_mode"] = ops.RC_OUT_NHWC_DWT          # conv [+ act] -> Haar DWT in one launch: the full-resolution map is never written
                    j += 1
                if j == n and residual is not None and kw.get("out_mode", RC_OUT_NHWC) == RC_OUT_NHWC and not shuffle_after:
                    kw["residual"], residual = residual, None
                a = m._nhwc(a, **kw)
                if shuffle_after:
                    a = ops.pixel_shuffle2(a)
                i = j
                continue
            if isinstance(m, RCAGroup) and i + 1 < n and isinstance(mods[i + 1], DWTForward) and not isinstance(a, (tuple, list)):
                a = m._nhwc(a, dwt=mods[i + 1])              # the group's closing conv + skip -> Haar DWT in one launch where the kernel has that form
                i += 2
                continue
            if isinstance(m, (nn.Dropout, nn.Identity)):
                i += 1
                continue
            if not hasattr(m, "_nhwc"):
                raise NotImplementedError(f"{type(m).__name__} has no HIP implementation on this path")
            a = m._nhwc(a)
            i += 1
        if residual is not None:
            raise NotImplementedError("trailing residual could not be fused (sequence does not end in a conv)")
        return a


def seq(*args):
    """Same contract as upstream `seq` (models/networks.py:117-128): one module is returned as is,
    lists nest into Sequential, OrderedDicts keep their keys."""
    if len(args) == 1:
        args = args[0]
    if isinstance(args, nn.Module):
        return args
    if isinstance(args, OrderedDict):
        return Sequential(OrderedDict((k, seq(v)) for k, v in args.items()))
    assert isinstance(args, (list, tuple))
    return Sequential(*[seq(i) for i in args])


_UNSUPPORTED = "XTBIiSP34UuMA"


def conv(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, output_padding=0, dilation=1,
         groups=1, bias=True, padding_mode='zeros', mode='CBR'):
    """Mode-string factory, upstream models/networks.py:146-221.  Built here: C, R/r, L/l, 2."""
    L = []
    for t in mode:
        if t == 'C':
            L.append(Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                            stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias,
                            padding_mode=padding_mode))
        elif t == 'R':
            L.append(nn.ReLU(inplace=True))
        elif t == 'r':
            L.append(nn.ReLU(inplace=False))
        elif t == 'L':
            L.append(nn.LeakyReLU(negative_slope=1e-1, inplace=True))
        elif t == 'l':
            L.append(nn.LeakyReLU(negative_slope=1e-1, inplace=False))
        elif t == '2':
            L.append(nn.PixelShuffle(upscale_factor=2))
        elif t in _UNSUPPORTED:
            raise NotImplementedError(f"conv mode '{t}' is not on the RAW->sRGB hot path (SURVEY.md section 8); no HIP kernel")
        else:
            raise NotImplementedError('Undefined type: {}'.format(t))
    return seq(*L)


_HAAR = [[[[0.5, 0.5], [0.5, 0.5]]],
         [[[0.5, 0.5], [-0.5, -0.5]]],
         [[[0.5, -0.5], [0.5, -0.5]]],
         [[[0.5, -0.5], [-0.5, 0.5]]]]


class DWTForward(nn.Conv2d):
    """Haar analysis as a frozen grouped conv whose taps live in the state_dict
    (upstream models/networks.py:224-235)."""

    def __init__(self, in_channels=64):
        super().__init__(in_channels, in_channels * 4, 2, 2, groups=in_channels, bias=False)
        w = torch.tensor(_HAAR, dtype=torch.get_default_dtype()).repeat(in_channels, 1, 1, 1)
        self.weight.data.copy_(w)
        self.requires_grad_(False)

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))

    def _nhwc(self, a):
        return ops.dwt_forward(a, self)


class DWTInverse(nn.ConvTranspose2d):
    """Haar synthesis as a frozen grouped transposed conv (upstream models/networks.py:238-249)."""

    def __init__(self, in_channels=64):
        super().__init__(in_channels, in_channels // 4, 2, 2, groups=in_channels // 4, bias=False)
        w = torch.tensor(_HAAR, dtype=torch.get_default_dtype()).repeat(in_channels // 4, 1, 1, 1)
        self.weight.data.copy_(w)
        self.requires_grad_(False)

    def forward(self, x, output_size=None):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))

    def _nhwc(self, a):
        return ops.dwt_inverse(a, self)


class DWTForward_(nn.Module):
    """Haar analysis for ANY channel count: one (4,1,2,2) tap set in the state_dict (the flipped taps upstream builds), repeated over the input's
    channels at call time (upstream models/networks.py:9-26)."""

    def __init__(self):
        super().__init__()
        ll, lh, hl, hh = [[0.5, 0.5], [0.5, 0.5]], [[-0.5, -0.5], [0.5, 0.5]], [[-0.5, 0.5], [-0.5, 0.5]], [[0.5, -0.5], [-0.5, 0.5]]
        flip = lambda f: [row[::-1] for row in f[::-1]]
        self.weight = nn.Parameter(torch.tensor([[flip(f)] for f in (ll, lh, hl, hh)], dtype=torch.get_default_dtype()), requires_grad=False)

    def _taps(self, c):
        return ops.host_cached(self, f"taps_x{c}", [self.weight], lambda w: w.repeat(c, 1, 1, 1))[0]

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))

    def _nhwc(self, a):
        return ops.dwt_forward(a, ops._ConvView(self._taps(a.shape[-1]), None))


class DWTInverse_(DWTForward_):
    """Haar synthesis for any channel count (upstream models/networks.py:29-47)."""

    def _nhwc(self, a):
        if a.shape[-1] % 4:
            raise ValueError("DWTInverse_: channels must be a multiple of 4")
        return ops.dwt_inverse(a, ops._ConvView(self._taps(a.shape[-1] // 4), None))


class CALayer(HipModule):
    """Channel attention (upstream models/networks.py:255-270).  The global mean needs the whole
    feature map, so inside RCABlock the producing conv emits per-tile channel sums and only the tiny
    gate MLP (rc_ca_gate) runs here."""

    def __init__(self, channel=64, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(
            Conv2d(channel, channel // reduction, 1, padding=0, bias=True),
            nn.ReLU(inplace=True),
            Conv2d(channel // reduction, channel, 1, padding=0, bias=True),
            nn.Sigmoid())

    def _nhwc(self, a):
        """x * sigmoid(conv_du(mean_HW(x))).  Inside RCABlock the producing conv emits the channel sums and the scale is
        folded into the next conv's staging; this stand-alone form reduces the map itself (rc_channel_sums)."""
        gate = ops.ca_gate(ops.channel_sums(a), a.shape[1] * a.shape[2], self)
        return ops.gate_residual(a, gate, None)


class ResBlock(HipModule):
    """x + conv(relu(conv(x))) (upstream models/networks.py:276-290)."""

    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True, mode='CRC'):
        super().__init__()
        assert in_channels == out_channels
        if mode[0] in ['R', 'L']:
            mode = mode[0].lower() + mode[1:]
        self.res = conv(in_channels, out_channels, kernel_size, stride, padding=padding, bias=bias, mode=mode)

    def _nhwc(self, a):
        return self.res._nhwc(a, residual=a)


class RCABlock(HipModule):
    """Residual channel-attention block (upstream models/networks.py:296-311)."""

    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True, mode='CRC',
                 reduction=16):
        super().__init__()
        assert in_channels == out_channels
        if mode[0] in ['R', 'L']:
            mode = mode[0].lower() + mode[1:]
        self.res = conv(in_channels, out_channels, kernel_size, stride, padding, bias=bias, mode=mode)
        self.ca = CALayer(out_channels, reduction)

    def _body(self, a, **first_conv_kw):
        """conv -> act -> conv of self.res; returns (res, gate [, stored_input])."""
        mods = list(self.res)
        if not (len(mods) == 3 and isinstance(mods[0], Conv2d) and _is_act(mods[1]) and isinstance(mods[2], Conv2d)):
            raise NotImplementedError("RCABlock: only mode 'CRC'/'CLC' is on the hot path")
        if isinstance(mods[1], nn.ReLU) and ops.conv_pair_ok(a, mods[0], mods[2]):
            # both convs in one launch, the ReLU'd intermediate never leaves LDS
            res = ops.conv_pair(a, mods[0], mods[2], act="relu", want_sums=True, **first_conv_kw)
            stored = res[1] if len(res) == 3 else None
            r, sums = res[0], res[-1]
        else:
            t = mods[0]._nhwc(a, **_act_args(mods[1]), **first_conv_kw)
            stored = None
            if isinstance(t, tuple):
                t, stored = t
            r, sums = mods[2]._nhwc(t, want_sums=True)
        gate = ops.ca_gate(sums, a.shape[1] * a.shape[2], self.ca)
        return r, gate, stored

    def _early(self, a) -> bool:
        mods = list(self.res)
        return (ops.EARLY_GATE and len(mods) == 3 and isinstance(mods[0], Conv2d) and _is_act(mods[1]) and isinstance(mods[2], Conv2d) and
                ops.gate_ahead_ok(mods[0], mods[2], self.ca) and           # stride 1, zero padding 1, C -> C, squeeze -> ReLU -> excite -> Sigmoid: else the closed form is wrong
                not (isinstance(mods[1], nn.ReLU) and ops.conv_pair_ok(a, mods[0], mods[2])))

    def _nhwc_early(self, a):
        """x + CA(conv2(t)), t = act(conv1(x)), as two launches + the small gate kernels: conv1 emits t's channel sums, the gate of conv2(t) follows
        from them ahead of conv2 (ops.ca_gate_ahead), conv2's epilogue writes conv2(t) * gate + x."""
        mods = list(self.res)
        t, sums = mods[0]._nhwc(a, **_act_args(mods[1]), want_sums=True)
        gate = ops.ca_gate_ahead(sums, t, mods[2], self.ca)
        return mods[2]._nhwc(t, out_scale=gate, residual=a)

    def _nhwc(self, a):
        if self._early(a):
            return self._nhwc_early(a)
        r, gate, _ = self._body(a)
        return ops.gate_residual(r, gate, a)


class RCAGroup(HipModule):
    """nb RCABlocks + conv + group skip (upstream models/networks.py:317-335)."""

    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True, mode='CRC',
                 reduction=16, nb=12):
        super().__init__()
        assert in_channels == out_channels
        if mode[0] in ['R', 'L']:
            mode = mode[0].lower() + mode[1:]
        RG = [RCABlock(in_channels, out_channels, kernel_size, stride, padding, bias, mode, reduction) for _ in range(nb)]
        RG.append(conv(out_channels, out_channels, mode='C'))
        self.rg = nn.Sequential(*RG)

    def _nhwc(self, a, dwt=None):
        """dwt: the networks.DWTForward that follows the group (LiteISP `down2` / `down3`): applied here, inside the closing conv's launch where that form exists."""
        if dwt is not None:
            fused = self._run(a, dwt)
            return fused if fused is not None else dwt._nhwc(self._run(a))
        return self._run(a)

    def _run(self, a, dwt=None):
        blocks = list(self.rg)
        last = blocks[-1]
        if not isinstance(last, Conv2d) or not all(isinstance(b, RCABlock) for b in blocks[:-1]):
            raise NotImplementedError("RCAGroup: unexpected layout")
        if not ops.FUSE_GATE or (len(blocks) > 1 and all(blk._early(a) for blk in blocks[:-1])):
            if dwt is not None and not ops.conv_dwt_ok(a, last, dwt, residual=True):
                return None
            y = a                                    # early-gate schedule: every block is two plain launches, nothing is carried between blocks
            for blk in blocks[:-1]:
                y = blk._nhwc(y)
            if dwt is not None:
                return last._nhwc(y, residual=a, out_mode=ops.RC_OUT_NHWC_DWT)
            return last._nhwc(y, residual=a)
        if dwt is not None:
            return None
        # fused: each block's "res*gate + skip" is formed inside the NEXT conv's input staging
        skip, r, gate = a, None, None
        for blk in blocks[:-1]:
            if r is None:
                r, gate, _ = blk._body(skip)
            else:
                r, gate, skip = blk._body(r, gate=gate, skip=skip, store_input=True)
        if r is None:
            return last._nhwc(a, residual=a)
        return last._nhwc(r, gate=gate, skip=skip, residual=a)
