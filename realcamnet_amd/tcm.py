"""Host mirror of the TCM transformer blocks: `WMSA` and `Block` of upstream models/tcm.py:139-236.

Same class names, constructor signatures and attribute names as upstream (`ln1`, `msa.embedding_layer`,
`msa.relative_position_params` (heads, 2ws-1, 2ws-1), `msa.linear`, `ln2`, `mlp.{0,2}`), so a reference state_dict loads
with strict=True.  Tensors are NHWC (b, h, w, c) exactly as upstream passes them, which is also the HIP path's layout.
The two Linear layers of WMSA and the MLP run through rc_conv2d as 1x1 convolutions (a point-wise op commutes with the
cyclic shift, so they run on the un-shifted map), LayerNorm through rc_layernorm, and the window attention core --
window partition, shift, relative-position bias, wrap mask, softmax, weighted sum -- is rc_window_attention.
ConvTransBlock, SWAtten and the slice transforms of the codec trunk (rows a18/a19) follow below; the CompressAI layers
they lean on (`ResidualBlock`, `AttentionBlock`) are not in the upstream tree and are restated from their published
definitions (parity unpinned for those layers).  The rest of the trunk (GDN stages, entropy models) is not built.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import bitstream
from . import networks as N
from . import networks as N_mod      # TCM.__init__ keeps upstream's parameter name `N`
from . import ops


class WMSA(nn.Module):
    """Window / shifted-window multi-head self-attention (upstream models/tcm.py:139-212)."""

    def __init__(self, input_dim, output_dim, head_dim, window_size, type):
        super().__init__()
        self.input_dim, self.output_dim, self.head_dim = input_dim, output_dim, head_dim
        self.scale = head_dim ** -0.5
        self.n_heads = input_dim // head_dim
        self.window_size = window_size
        self.type = type
        self.embedding_layer = nn.Linear(input_dim, 3 * input_dim, bias=True)
        rel = torch.zeros((2 * window_size - 1) * (2 * window_size - 1), self.n_heads)
        nn.init.trunc_normal_(rel, std=.02)
        self.relative_position_params = nn.Parameter(
            rel.view(2 * window_size - 1, 2 * window_size - 1, self.n_heads).transpose(1, 2).transpose(0, 1).contiguous())
        self.linear = nn.Linear(input_dim, output_dim)

    def _attend(self, t, residual=None, ln=None):
        """t NHWC (b,h,w,c) -> linear(attention(t)) [+ residual]; ln: the LayerNorm in front (applied here, fused into the embedding layer
        where the shape allows), None when t is already normalised."""
        t = ops._req(t, "WMSA input")
        b, h, w, c = t.shape
        ws = self.window_size
        if c != self.input_dim or h % ws or w % ws:
            raise ValueError(f"WMSA: expected (b, h, w, {self.input_dim}) with h, w multiples of {ws}, got {tuple(t.shape)}")
        planar = ln is not None and ops.planar_qkv_ok(t, ws)            # q / k / v segment-planar between the two launches (rc_window_attention_planar8: sector-sized reads)
        qkv = ops.ln_linear(t, ln, self.embedding_layer, planar8=planar) if ln is not None else None
        if qkv is None:
            planar = False
            qkv = ops.conv2d(t if ln is None else ops.layernorm(t, ln), self.embedding_layer)
        attend = torch.ops.realcam.window_attention_planar8 if planar else torch.ops.realcam.window_attention
        att = attend(qkv, ops.f32_param(self, "relative_position_params"), self.head_dim, ws, 0 if self.type == 'W' else ws // 2)
        return ops.conv2d(att, self.linear, residual=residual)

    def forward(self, x):
        return self._attend(x)


class Block(nn.Module):
    """x + WMSA(LN(x)); + MLP(LN(.))   (upstream models/tcm.py:214-236; DropPath is the identity in eval / at rate 0)."""

    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path, type='W', input_resolution=None):
        super().__init__()
        assert type in ['W', 'SW']
        self.input_dim, self.output_dim, self.type = input_dim, output_dim, type
        self.ln1 = nn.LayerNorm(input_dim)
        self.msa = WMSA(input_dim, input_dim, head_dim, window_size, self.type)
        self.drop_path = nn.Identity()
        self.ln2 = nn.LayerNorm(input_dim)
        self.mlp = nn.Sequential(nn.Linear(input_dim, 4 * input_dim), nn.GELU(), nn.Linear(4 * input_dim, output_dim))

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        x = ops._req(x, "Block input")
        x = self.msa._attend(x, residual=x, ln=self.ln1)
        y = ops.ln_mlp(x, self.ln2, self.mlp[0], self.mlp[2])             # bf16, width 32 / 64: one launch, activations in registers
        if y is not None:
            return y
        h = ops.conv2d(ops.layernorm(x, self.ln2), self.mlp[0], act="gelu")
        return ops.conv2d(h, self.mlp[2], residual=x)


class SwinBlock(nn.Module):
    """Block('W') then Block('SW') on an NCHW map (upstream models/tcm.py:299-312).  NCHW in / NCHW out as upstream; the
    two blocks run on the NHWC view.  Maps must be larger than the window in both dimensions (upstream's padding branch
    for smaller maps yields sizes the window partition rejects)."""

    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path) -> None:
        super().__init__()
        self.block_1 = Block(input_dim, output_dim, head_dim, window_size, drop_path, type='W')
        self.block_2 = Block(input_dim, output_dim, head_dim, window_size, drop_path, type='SW')
        self.window_size = window_size

    def forward(self, x):
        if x.size(-1) <= self.window_size or x.size(-2) <= self.window_size:
            raise ValueError("SwinBlock: the map must be larger than the window")
        t = self.block_2(self.block_1(ops.to_nhwc(x)))
        return ops.to_nchw(t)


class ResidualBlock(nn.Module):
    """CompressAI's `ResidualBlock(in_ch, out_ch)` (compressai.layers; PyPI package, NOT in the upstream tree, no version
    pinned -- SURVEY.md 8c "parity unpinned"), restated from its published definition:
        out = LeakyReLU(conv3x3(LeakyReLU(conv3x3(x)))) + (conv1x1(x) if in_ch != out_ch else x),   LeakyReLU slope 0.01
    with attributes conv1, conv2 (and skip) so a CompressAI state_dict loads."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.conv1 = N.Conv2d(in_ch, out_ch, 3, 1, 1)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = N.Conv2d(out_ch, out_ch, 3, 1, 1)
        self.skip = N.Conv2d(in_ch, out_ch, 1, 1, 0) if in_ch != out_ch else None

    def _nhwc(self, a):
        slope = float(self.leaky_relu.negative_slope)
        identity = a if self.skip is None else self.skip._nhwc(a)
        t = self.conv1._nhwc(a, act="leaky", slope=slope)
        return self.conv2._nhwc(t, act="leaky", slope=slope, residual=identity)

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class ConvTransBlock(nn.Module):
    """1x1 conv -> split -> [ResidualBlock(conv_x) + conv_x  ||  Block(trans_x)] -> concat -> 1x1 conv -> + x
    (upstream models/tcm.py:242-268; note the double residual on the conv branch, :262).  NCHW in / NCHW out."""

    def __init__(self, conv_dim, trans_dim, head_dim, window_size, drop_path, type='W'):
        super().__init__()
        assert type in ['W', 'SW']
        self.conv_dim, self.trans_dim, self.head_dim, self.window_size, self.drop_path, self.type = \
            conv_dim, trans_dim, head_dim, window_size, drop_path, type
        self.trans_block = Block(trans_dim, trans_dim, head_dim, window_size, drop_path, type)
        self.conv1_1 = N.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv1_2 = N.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv_block = ResidualBlock(conv_dim, conv_dim)

    def _nhwc(self, a):
        va, vb = ops.split_conv_views(self.conv1_1, (self.conv_dim, self.trans_dim))     # torch.split(conv1_1(x)) without the slice copies
        conv_x, trans_x = ops.conv2d(a, va), ops.conv2d(a, vb)
        rb = self.conv_block._nhwc(conv_x)
        trans_x = self.trans_block(trans_x)
        # conv1_2(cat(conv_block(conv_x) + conv_x, trans_x)) + x: the sum and the concatenation happen on the closing launch's operand load
        y = ops.cat_linear(rb, trans_x, self.conv1_2, residual=a, a_add=conv_x)
        return y if y is not None else self.conv1_2._nhwc(ops.channel_concat([ops.add(rb, conv_x), trans_x]), residual=a)

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


def conv1x1(in_ch: int, out_ch: int, stride: int = 1) -> nn.Module:
    """upstream models/tcm.py:29-31"""
    if stride not in (1, 2):
        raise NotImplementedError("conv1x1: stride 1 or 2")
    return N.Conv2d(in_ch, out_ch, kernel_size=1, stride=stride)


def conv3x3(in_ch: int, out_ch: int, stride: int = 1) -> nn.Module:
    """compressai.layers.conv3x3 (3x3, padding 1)"""
    if stride not in (1, 2):
        raise NotImplementedError("conv3x3: stride 1 or 2")
    return N.Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


def subpel_conv3x3(in_ch: int, out_ch: int, r: int = 1) -> nn.Module:
    """compressai.layers.subpel_conv3x3: conv3x3(in, out*r^2) + PixelShuffle(r) (one rc_conv2d launch for r = 2)."""
    if r != 2:
        raise NotImplementedError("subpel_conv3x3: only r = 2 is on this path")
    return N.Sequential(N.Conv2d(in_ch, out_ch * r ** 2, kernel_size=3, padding=1), nn.PixelShuffle(r))


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    """upstream models/tcm.py:130-137.  Only the stride-1 3x3 form the slice transforms use has a HIP kernel."""
    if stride != 1 or kernel_size not in (1, 3):
        raise NotImplementedError("conv: only kernel_size 1/3, stride 1 is on this path (the strided 5x5 stages of g_a/h_a are not built)")
    return N.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1, padding=kernel_size // 2)


def slice_transform(in_channels: int, out_channels: int) -> nn.Module:
    """One entry of TCM.cc_mean_transforms / cc_scale_transforms / lrp_transforms (upstream models/tcm.py:398-425):
    conv3x3(in, 224) -> GELU -> conv3x3(224, 128) -> GELU -> conv3x3(128, out), stride 1.  Same Sequential indices
    (0, 2, 4) as upstream, so the matching slice of a TCM state_dict loads.  Each conv + GELU is one rc_conv2d launch."""
    return N.Sequential(conv(in_channels, 224, stride=1, kernel_size=3), nn.GELU(),
                        conv(224, 128, stride=1, kernel_size=3), nn.GELU(),
                        conv(128, out_channels, stride=1, kernel_size=3))


class _ResidualUnit(nn.Module):
    """The residual unit inside CompressAI's AttentionBlock (restated, parity unpinned):
        relu(conv1x1(relu(conv3x3(relu(conv1x1(x))))) + x)   with N -> N/2 -> N/2 -> N channels."""

    def __init__(self, n: int):
        super().__init__()
        self.conv = nn.Sequential(conv1x1(n, n // 2), nn.ReLU(inplace=True), conv3x3(n // 2, n // 2), nn.ReLU(inplace=True),
                                  conv1x1(n // 2, n))
        self.relu = nn.ReLU(inplace=True)

    def _nhwc(self, a):
        t = self.conv[0]._nhwc(a, act="relu")
        t = self.conv[2]._nhwc(t, act="relu")
        return self.conv[4]._nhwc(t, act="relu_post", residual=a)


class AttentionBlock(nn.Module):
    """compressai.layers.AttentionBlock(N) (PyPI package, NOT in the upstream tree -- parity unpinned), restated from its
    published definition: conv_a = 3 residual units, conv_b = 3 residual units + conv1x1; out = a * sigmoid(b) + x.
    Attribute names (conv_a.{0,1,2}.conv.{0,2,4}, conv_b.{0,1,2}.conv.{0,2,4}, conv_b.3) match CompressAI's state_dict."""

    def __init__(self, N_: int):
        super().__init__()
        self.conv_a = nn.Sequential(_ResidualUnit(N_), _ResidualUnit(N_), _ResidualUnit(N_))
        self.conv_b = nn.Sequential(_ResidualUnit(N_), _ResidualUnit(N_), _ResidualUnit(N_), conv1x1(N_, N_))

    def _branch_a(self, a):
        for u in self.conv_a:
            a = u._nhwc(a)
        return a

    def _branch_b(self, b):
        for u in list(self.conv_b)[:3]:
            b = u._nhwc(b)
        return self.conv_b[3]._nhwc(b)

    def _nhwc(self, a):
        # conv_a || conv_b: two independent chains of 9 / 10 small launches (ops.fork_join: two streams outside graph capture)
        ya, yb = ops.fork_join(lambda: self._branch_a(a), lambda: self._branch_b(a), [a])
        return ops.sigmoid_gate_add(ya, yb, a)

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class SWAtten(AttentionBlock):
    """Window-attention gate of the codec's slice loop (upstream models/tcm.py:270-291): in_conv -> [conv_a(x),
    conv_b(SwinBlock(x))] -> a * sigmoid(b) + x -> out_conv.  NCHW in / NCHW out."""

    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path, inter_dim=192) -> None:
        if inter_dim is not None:
            super().__init__(inter_dim)
            self.non_local_block = SwinBlock(inter_dim, inter_dim, head_dim, window_size, drop_path)
            self.in_conv = conv1x1(input_dim, inter_dim)
            self.out_conv = conv1x1(inter_dim, output_dim)
        else:
            super().__init__(input_dim)
            self.non_local_block = SwinBlock(input_dim, input_dim, head_dim, window_size, drop_path)
            self.in_conv = self.out_conv = None

    def _nhwc(self, a):
        if self.in_conv is None:
            # upstream's inter_dim=None branch never sets in_conv and fails in forward (tcm.py:284); mirrored as an error
            raise AttributeError("SWAtten without inter_dim has no in_conv (same as upstream)")
        x = self.in_conv._nhwc(a)
        if x.shape[1] <= self.non_local_block.window_size or x.shape[2] <= self.non_local_block.window_size:
            raise ValueError("SWAtten: the map must be larger than the window")
        # conv_a(x) on a side stream beside [SwinBlock -> conv_b] (the longer chain) on this one: nested inside the slice loop's mean || scale fork
        ya, yb = ops.fork_join(lambda: self._branch_a(x), lambda: self._branch_b(self.non_local_block.block_2(self.non_local_block.block_1(x))), [x])
        out = ops.sigmoid_gate_add(ya, yb, x)
        return self.out_conv._nhwc(out)

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class _LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.tensor([float(bound)], dtype=torch.float32))     # (torch.Tensor([..]) ignores the device context)


class _NonNegativeParametrizer(nn.Module):
    """compressai.ops.parametrizers.NonNegativeParametrizer (restated): stored p -> max(p, bound)^2 - pedestal."""

    def __init__(self, minimum: float = 0.0, reparam_offset: float = 2 ** -18):
        super().__init__()
        pedestal = float(reparam_offset) ** 2
        self.register_buffer("pedestal", torch.tensor([pedestal], dtype=torch.float32))
        self.lower_bound = _LowerBound((float(minimum) + pedestal) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        return torch.max(x, self.lower_bound.bound.to(x.dtype)) ** 2 - self.pedestal.to(x.dtype)


class GDN(nn.Module):
    """compressai.layers.GDN (restated, parity unpinned): y = x * rsqrt(beta + gamma (*) x^2) over channels (a 1x1 convolution
    of x^2); inverse=True multiplies by sqrt(...) instead.  Parameters `beta` (C), `gamma` (C,C) are stored re-parametrised
    exactly as CompressAI stores them, with the same buffer names, so its state_dict loads.  Runs as rc_square -> rc_conv2d
    (1x1, weights = effective gamma, bias = effective beta, re-derived when the parameters change) -> rc_gdn_apply."""

    def __init__(self, in_channels: int, inverse: bool = False, beta_min: float = 1e-6, gamma_init: float = 0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = _NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = _NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def _effective(self):
        key = ops._key(self.beta, self.gamma)
        hit = getattr(self, "_eff", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                c = self.beta.numel()
                view = ops._ConvView(self.gamma_reparam(self.gamma.float()).reshape(c, c, 1, 1).contiguous(),
                                     self.beta_reparam(self.beta.float()).contiguous())
            hit = (key, view)
            object.__setattr__(self, "_eff", hit)
        return hit[1]

    def _nhwc(self, a, identity=None):
        if ops.FUSE_MLP and a.dtype == torch.bfloat16 and a.shape[-1] in (64, 128):        # one per-token launch (rc_gdn_chain)
            w, b = ops.packed_chain(self._effective())
            idn = ops._req(identity, "identity") if identity is not None else None
            return torch.ops.realcam.gdn_chain(ops._req(a, "GDN input"), idn, w, b, self.inverse)
        norm = ops.conv2d(ops.square(a), self._effective())
        return ops.gdn_apply(a, norm, self.inverse, identity)

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class ResidualBlockWithStride(nn.Module):
    """compressai.layers.ResidualBlockWithStride (restated, parity unpinned; call sites upstream models/tcm.py:336-339,361):
    GDN(conv3x3(LeakyReLU(conv3x3_s2(x)))) + conv1x1_s2(x)."""

    def __init__(self, in_ch: int, out_ch: int, stride: int = 2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.gdn = GDN(out_ch)
        self.skip = conv1x1(in_ch, out_ch, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def _nhwc(self, a):
        t = self.conv1._nhwc(a, act="leaky", slope=float(self.leaky_relu.negative_slope))
        t = self.conv2._nhwc(t)
        identity = a if self.skip is None else self.skip._nhwc(a)
        return self.gdn._nhwc(t, identity)

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


class ResidualBlockUpsample(nn.Module):
    """compressai.layers.ResidualBlockUpsample (restated, parity unpinned; call sites upstream models/tcm.py:347-353,364):
    IGDN(conv3x3(LeakyReLU(subpel_conv3x3(x)))) + subpel_conv3x3'(x)."""

    def __init__(self, in_ch: int, out_ch: int, upsample: int = 2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.igdn = GDN(out_ch, inverse=True)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def _nhwc(self, a):
        # LeakyReLU commutes with the pixel shuffle: fused into the conv's epilogue ahead of the shuffled store
        slope = float(self.leaky_relu.negative_slope)
        if (self.subpel_conv[0].out_channels // 4) % 16 == 0:
            t = self.subpel_conv[0]._nhwc(a, act="leaky", slope=slope, out_mode=N.RC_OUT_PIXEL_SHUFFLE2)
        else:
            t = ops.pixel_shuffle2(self.subpel_conv[0]._nhwc(a, act="leaky", slope=slope))
        t = self.conv._nhwc(t)
        return self.igdn._nhwc(t, self.upsample._nhwc(a))

    def forward(self, x):
        return ops.to_nchw(self._nhwc(ops.to_nhwc(x)))


_CODER_BUFFERS = ("_offset", "_quantized_cdf", "_cdf_length")


def _register_coder_buffers(mod, likelihood_bound: float) -> None:
    """The buffers compressai.entropy_models.EntropyModel.__init__ registers: empty int tables (filled by update() or a checkpoint)
    and the likelihood LowerBound -- so a reference checkpoint's keys all have a home (strict=True)."""
    mod.likelihood_lower_bound = _LowerBound(likelihood_bound)
    for name in _CODER_BUFFERS:
        mod.register_buffer(name, torch.IntTensor())


def _set_coder_buffers(mod, offset, cdf, length) -> None:
    dev = mod._offset.device
    mod._offset, mod._quantized_cdf, mod._cdf_length = offset.to(dev), cdf.to(dev), length.to(dev)
    mod.__dict__.pop("_coder_tables", None)


def _coder_tables(mod) -> "bitstream.Tables":
    dev = next(iter(mod.parameters()), mod._offset).device if any(True for _ in mod.parameters()) else mod._offset.device
    key = (mod._quantized_cdf.data_ptr(), mod._quantized_cdf._version, str(dev))
    hit = mod.__dict__.get("_coder_tables")
    if hit is None or hit[0] != key:
        hit = (key, bitstream.Tables(mod._quantized_cdf, mod._cdf_length, mod._offset, dev))
        mod.__dict__["_coder_tables"] = hit
    return hit[1]


def _resize_coder_buffers(mod, prefix: str, names, state_dict) -> None:
    """compressai.models.utils.update_registered_buffers(policy="resize_if_empty"): make the (empty) table buffers the checkpoint's
    size so that nn.Module.load_state_dict(strict=True) can copy them (upstream models/tcm.py:492-499)."""
    for name in names:
        key = f"{prefix}.{name}"
        if key in state_dict:
            cur = getattr(mod, name)
            new = state_dict[key]
            if cur.numel() == 0 or cur.shape != new.shape:
                setattr(mod, name, torch.empty(new.shape, dtype=cur.dtype, device=cur.device))
    mod.__dict__.pop("_coder_tables", None)


import inspect as _inspect
_APPLY_TAKES_RECURSE = "recurse" in _inspect.signature(nn.Module._apply).parameters      # torch >= 2.0; older: _apply(fn)


class _Fp32Masters:
    """fp32 master copies of the tensors the CODER depends on (CDF-index thresholds, medians, the density's parameters).  `.to(bfloat16)`
    rounds a module's floating-point parameters and buffers; thresholds and medians rounded on one side only make a stream that does not
    decode, and tables built before / after the cast differ.  The masters are taken the moment a cast would round an fp32 tensor (and
    from the checkpoint's own values in load_state_dict), follow device moves, and are what update() / the symbol kernels read -- so
    update() -> .to(bf16) and .to(bf16) -> update() give identical strings.  Never part of the state_dict."""
    _MASTER_NAMES: tuple = ()

    def _master(self, name: str) -> torch.Tensor:
        live = getattr(self, name)
        m = self.__dict__.get("_f32_masters", {}).get(name)
        if live.dtype == torch.float32 or m is None or m.shape != live.shape:
            return live.detach().float()
        m = m.to(live.device)
        # Is the live low-precision tensor still the master's rounding?  A value written after the cast is the truth and the master stale.  The comparison
        # costs a full-tensor compare and a host sync (and cannot run under HIP-graph capture), so its verdict is remembered per write generation of the live
        # tensor: `_version` moves on every in-place write (`.data` edits do not move it: ops.invalidate_caches(module) after those, as for packed weights).
        key = (live.data_ptr(), live._version, live.dtype, str(live.device))
        verdicts = self.__dict__.setdefault("_f32_verdict", {})
        hit = verdicts.get(name)
        if hit is None or hit[0] != key:
            hit = verdicts[name] = (key, bool(torch.equal(m.to(live.dtype), live.detach())))
        return m if hit[1] else live.detach().float()

    def _apply(self, fn, recurse=True):
        masters = self.__dict__.setdefault("_f32_masters", {})
        # snapshot first: nn.Module._apply swaps a Parameter's .data in place, so the "old" object would show the new dtype afterwards
        pre = {n: getattr(self, n).detach().clone() for n in self._MASTER_NAMES
               if getattr(self, n).dtype == torch.float32 and getattr(self, n).numel() > 0}
        # a master is authoritative only while the live low-precision tensor is still ITS rounding: a value written while the module was bf16
        # (optimizer step, p.data = ..., copy_) must survive the cast back, not be reverted to the stale master
        stale = set()
        for n, m in masters.items():
            live = getattr(self, n, None)
            if live is not None and live.dtype != torch.float32 and m.shape == live.shape and not torch.equal(m.to(live.device).to(live.dtype), live.detach()):
                stale.add(n)
        out = super()._apply(fn, recurse) if _APPLY_TAKES_RECURSE else super()._apply(fn)
        for n in self._MASTER_NAMES:
            new = getattr(self, n)
            if new.dtype == torch.float32:
                m = masters.pop(n, None)
                if m is not None and m.shape == new.shape and n not in stale:   # back to fp32 after a rounding cast: restore what the cast rounded, so
                    with torch.no_grad():                                       # .to(bf16).float() leaves the coder's tensors exact (tables stay identical)
                        new.copy_(m.to(new.device))
            elif n in pre:
                masters[n] = pre[n]                                   # this cast rounded an fp32 tensor: keep what it rounded
            elif n in stale:
                masters.pop(n, None)                                  # the live values were changed under the master: they are the truth now
            if n in masters:
                masters[n] = masters[n].to(new.device)
                if n in self._buffers:                                # a buffer (scale_table) simply stays fp32: the state_dict keeps exact values
                    self._buffers[n] = masters.pop(n)
        # a master that survived this cast IS the new live tensor's source (also across bf16 -> fp16, whose double rounding a value compare would misread)
        verdicts = self.__dict__.setdefault("_f32_verdict", {})
        verdicts.clear()
        for n, m in masters.items():
            live = getattr(self, n)
            if live.dtype != torch.float32 and m.shape == live.shape:
                verdicts[n] = ((live.data_ptr(), live._version, live.dtype, str(live.device)), True)
        self.__dict__.pop("_coder_tables", None)
        return out

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        masters = self.__dict__.setdefault("_f32_masters", {})
        for n in self._MASTER_NAMES:
            v = state_dict.get(prefix + n)
            if v is not None and v.is_floating_point() and getattr(self, n).dtype != torch.float32:
                masters[n] = v.detach().float().clone().to(getattr(self, n).device)   # the checkpoint's exact values, not their rounding
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class EntropyBottleneck(_Fp32Masters, nn.Module):
    """compressai.entropy_models.EntropyBottleneck(channels): parameters, the eval-mode likelihood path, the CDF tables (`update`) and
    `compress` / `decompress` over the GPU rANS coder (restated from its published definition, parity unpinned).  Parameter names
    `_matrix{i}`, `_bias{i}`, `_factor{i}`, `quantiles`, the `target` buffer and EntropyModel's table buffers follow CompressAI's
    classic layout."""
    _MASTER_NAMES = tuple(f"_matrix{i}" for i in range(5)) + tuple(f"_bias{i}" for i in range(5)) + tuple(f"_factor{i}" for i in range(4)) + ("quantiles",)

    def __init__(self, channels: int, tail_mass: float = 1e-9, init_scale: float = 10, filters=(3, 3, 3, 3), likelihood_bound: float = 1e-9):
        super().__init__()
        import math
        if tuple(filters) != (3, 3, 3, 3):
            raise NotImplementedError("EntropyBottleneck: the HIP kernel is built for filters=(3,3,3,3)")
        self.channels, self.filters, self.likelihood_bound = int(channels), (3, 3, 3, 3), float(likelihood_bound)
        f = (1, 3, 3, 3, 3, 1)
        scale = float(init_scale) ** (1 / 5)
        for i in range(5):
            init = math.log(math.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), init)))
            self.register_parameter(f"_bias{i}", nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < 4:
                self.register_parameter(f"_factor{i}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.tensor([-float(init_scale), 0.0, float(init_scale)]).repeat(channels, 1, 1))
        target = math.log(2 / float(tail_mass) - 1)
        self.register_buffer("target", torch.tensor([-target, 0.0, target]))
        _register_coder_buffers(self, likelihood_bound)

    def update(self, force: bool = False) -> bool:
        """EntropyBottleneck.update(): (re)build `_offset`, `_quantized_cdf`, `_cdf_length` from the density's parameters."""
        if self._offset.numel() > 0 and not force:
            return False
        _set_coder_buffers(self, *bitstream.bottleneck_tables(self))
        return True

    def _compress_nhwc(self, z, fmt="chunked", chunk=bitstream.DEFAULT_CHUNK):
        """z NHWC -> (one byte string per image, z_hat NHWC): symbols = round(z - median), index = channel."""
        b, h, w, c = z.shape
        med = self._master("quantiles")[:, 0, 1].contiguous()
        sym, idx, z_hat = torch.ops.realcam.eb_symbols(ops._req(z, "z"), None, med, b, h, w, z.dtype)
        tables = _coder_tables(self)
        return [bitstream.encode(sym[i], idx[i], tables, fmt, chunk) for i in range(b)], z_hat

    def _compress_nhwc_async(self, z, chunk=bitstream.DEFAULT_CHUNK):
        """_compress_nhwc(fmt="chunked") without the read-back: (pending containers, one per image; z_hat) -- bitstream.finish() turns them into strings."""
        b, h, w, c = z.shape
        med = self._master("quantiles")[:, 0, 1].contiguous()
        sym, idx, z_hat = torch.ops.realcam.eb_symbols(ops._req(z, "z"), None, med, b, h, w, z.dtype)
        tables = _coder_tables(self)
        return [bitstream.encode_async(sym[i], idx[i], tables, chunk) for i in range(b)], z_hat

    def _decompress_nhwc(self, strings, size, dtype, fmt="chunked", _decoders=None):
        h, w = size
        b, c = len(strings), self.channels
        dev = self.quantiles.device
        med = self._master("quantiles")[:, 0, 1].contiguous()
        tables = _coder_tables(self)
        idx = torch.arange(c, dtype=torch.int32, device=dev).view(c, 1).expand(c, h * w).contiguous()
        if _decoders is not None and fmt == "chunked":          # the codec's decompress: error flags looked at once, at its end
            _decoders.extend(bitstream.Decoder(s, tables, dev, fmt) for s in strings)
            sym = torch.stack([d.decode_async(idx).view(c, h * w) for d in _decoders[-b:]])
        else:
            sym = torch.stack([bitstream.Decoder(s, tables, dev, fmt).decode(idx).view(c, h * w) for s in strings])
        return torch.ops.realcam.eb_symbols(None, sym, med, b, h, w, dtype)[2]

    def compress(self, x, fmt="chunked"):
        """compressai EntropyBottleneck.compress(x NCHW) -> list of byte strings, one per image."""
        return self._compress_nhwc(ops.to_nhwc(x), fmt)[0]

    def decompress(self, strings, size, fmt="chunked"):
        """compressai EntropyBottleneck.decompress(strings, size) -> z_hat NCHW (the parameters' dtype)."""
        return ops.to_nchw(self._decompress_nhwc(strings, tuple(size), self.quantiles.dtype, fmt))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def _packed(self):
        ps = [getattr(self, f"_matrix{i}") for i in range(5)] + [getattr(self, f"_bias{i}") for i in range(5)] + \
             [getattr(self, f"_factor{i}") for i in range(4)] + [self.quantiles]
        from torch._subclasses.fake_tensor import FakeTensor
        if isinstance(self.quantiles, FakeTensor):                          # shape tracing: nothing to pack on the host
            q = self.quantiles
            return q.new_empty((self.channels, 58), dtype=torch.float32), q.new_empty((self.channels,), dtype=torch.float32)
        key = tuple((p._version, p.data_ptr()) for p in ps)
        hit = getattr(self, "_pk", None)
        if hit is None or hit[0] != key:
            import numpy as np
            g = lambda n: getattr(self, n).detach().float().cpu().numpy().astype(np.float64)
            softplus = lambda a: np.logaddexp(a, 0.0)
            cols = []
            for i in range(5):
                cols.append(softplus(g(f"_matrix{i}")).reshape(self.channels, -1))     # (out, in) row-major
                cols.append(g(f"_bias{i}").reshape(self.channels, -1))
                if i < 4:
                    cols.append(np.tanh(g(f"_factor{i}")).reshape(self.channels, -1))
            packed = np.concatenate(cols, axis=1).astype(np.float32)
            assert packed.shape == (self.channels, 58)
            dev = self.quantiles.device
            hit = (key, torch.from_numpy(packed).to(dev), self.quantiles.detach()[:, 0, 1].float().contiguous())
            object.__setattr__(self, "_pk", hit)
        return hit[1], hit[2]

    def _nhwc(self, z):
        """(z_hat, likelihood fp32) of an NHWC latent."""
        params, med = self._packed()
        return ops.entropy_bottleneck(z, params, med, self.likelihood_bound)

    def forward(self, x):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        z_hat, lik = self._nhwc(ops.to_nhwc(x))
        return ops.to_nchw(z_hat), ops.to_nchw(lik)


class GaussianConditional(_Fp32Masters, nn.Module):
    """compressai.entropy_models.GaussianConditional(None): the eval-mode likelihood path, the scale table / CDF tables
    (`update_scale_table`, `update`) and symbol preparation for the coder (restated, parity unpinned).  Buffers as CompressAI
    registers them: scale_table, scale_bound, lower_bound_scale.bound, likelihood_lower_bound.bound, _offset, _quantized_cdf,
    _cdf_length."""
    _MASTER_NAMES = ("scale_table",)

    def __init__(self, scale_table=None, scale_bound: float = 0.11, tail_mass: float = 1e-9, likelihood_bound: float = 1e-9):
        super().__init__()
        if scale_table is not None:
            raise NotImplementedError("GaussianConditional: construct with None and call update_scale_table(), as upstream does")
        self.scale_bound_value, self.likelihood_bound, self.tail_mass = float(scale_bound), float(likelihood_bound), float(tail_mass)
        _register_coder_buffers(self, likelihood_bound)
        self.lower_bound_scale = _LowerBound(scale_bound)
        self.register_buffer("scale_table", torch.Tensor())
        self.register_buffer("scale_bound", torch.tensor([float(scale_bound)], dtype=torch.float32))

    def update_scale_table(self, scale_table, force: bool = False) -> bool:
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = torch.as_tensor(sorted(float(s) for s in scale_table), dtype=torch.float32).to(self.scale_table.device)
        self.__dict__.get("_f32_masters", {}).pop("scale_table", None)      # always stored as fp32: the live buffer is the master
        self.update()
        return True

    def update(self) -> None:
        _set_coder_buffers(self, *bitstream.gaussian_tables(self._master("scale_table"), self.tail_mass))

    def _table(self, device):
        if self.scale_table.numel() == 0:
            raise RuntimeError("GaussianConditional has no scale table: call the model's update() first")
        return self._master("scale_table").to(device=device, dtype=torch.float32).contiguous()

    def _nhwc(self, y, scale, mu):
        return ops.gaussian_conditional(y, scale, mu, self.scale_bound_value, self.likelihood_bound)

    def forward(self, inputs, scales, means):
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        y_hat, lik = self._nhwc(ops.to_nhwc(inputs), ops.to_nhwc(scales), ops.to_nhwc(means))
        return ops.to_nchw(y_hat), ops.to_nchw(lik)


def _codec_update(self, scale_table=None, force: bool = False) -> bool:
    """upstream models/tcm.py:430-435 (TCM.update -> CompressionModel.update): scale table + Gaussian tables, then every
    EntropyBottleneck child's tables.  Must be called (or a checkpoint with tables loaded) before compress() / decompress()."""
    if scale_table is None:
        scale_table = bitstream.get_scale_table()
    updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
    for mod in self.modules():
        if isinstance(mod, EntropyBottleneck):
            updated |= mod.update(force=force)
    self.__dict__.pop("_graphs", None)               # captured compress graphs hold the old tables' addresses
    return updated


def _codec_load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
    """upstream models/tcm.py:492-499: size the entropy models' (empty) table buffers from the checkpoint, then load -- so a reference
    checkpoint (which carries `_quantized_cdf`, `_offset`, `_cdf_length`, `scale_table`) loads with strict=True."""
    _resize_coder_buffers(self.gaussian_conditional, "gaussian_conditional", _CODER_BUFFERS + ("scale_table",), state_dict)
    _resize_coder_buffers(self.entropy_bottleneck, "entropy_bottleneck", _CODER_BUFFERS, state_dict)
    return nn.Module.load_state_dict(self, state_dict, strict=strict, assign=assign)


class TCM(nn.Module):
    """The transforms of upstream's `TCM` codec (models/tcm.py:320-425), built in the same order under the same attribute
    names: g_a, g_s, h_a, h_mean_s, h_scale_s, atten_mean, atten_scale, cc_mean_transforms, cc_scale_transforms,
    lrp_transforms, entropy_bottleneck, gaussian_conditional; `forward` (the likelihood path, eval mode), `update`, `compress`,
    `decompress` (CDF tables + rANS on the GPU, realcamnet_amd/bitstream.py) and `load_state_dict` (a reference checkpoint, table
    buffers included, loads with strict=True).  NCHW at this boundary, NHWC inside."""

    def __init__(self, config=[2, 2, 2, 2, 2, 2], head_dim=[8, 16, 32, 32, 16, 8], drop_path_rate=0, N=64, M=320, num_slices=5,
                 max_support_slices=5, **kwargs):
        super().__init__()
        if drop_path_rate != 0:
            raise NotImplementedError("inference path: drop_path_rate must be 0")
        self.config, self.head_dim, self.window_size = config, head_dim, 8
        self.num_slices, self.max_support_slices, self.M = num_slices, max_support_slices, M
        dim = N_ = N

        def stage(n, hd, ws=self.window_size):
            return [ConvTransBlock(dim, dim, hd, ws, 0, 'W' if not i % 2 else 'SW') for i in range(n)]

        self.g_a = N_mod.Sequential(*[ResidualBlockWithStride(3, 2 * N_, 2)] + stage(config[0], head_dim[0]) + [ResidualBlockWithStride(2 * N_, 2 * N_, stride=2)] +
                                    stage(config[1], head_dim[1]) + [ResidualBlockWithStride(2 * N_, 2 * N_, stride=2)] +
                                    stage(config[2], head_dim[2]) + [conv3x3(2 * N_, M, stride=2)])
        self.g_s = N_mod.Sequential(*[ResidualBlockUpsample(M, 2 * N_, 2)] + stage(config[3], head_dim[3]) + [ResidualBlockUpsample(2 * N_, 2 * N_, 2)] +
                                    stage(config[4], head_dim[4]) + [ResidualBlockUpsample(2 * N_, 2 * N_, 2)] +
                                    stage(config[5], head_dim[5]) + [subpel_conv3x3(2 * N_, 3, 2)])
        self.h_a = N_mod.Sequential(*[ResidualBlockWithStride(320, 2 * N_, 2)] + stage(config[0], 32, 4) + [conv3x3(2 * N_, 192, stride=2)])
        self.h_mean_s = N_mod.Sequential(*[ResidualBlockUpsample(192, 2 * N_, 2)] + stage(config[3], 32, 4) + [subpel_conv3x3(2 * N_, 320, 2)])
        self.h_scale_s = N_mod.Sequential(*[ResidualBlockUpsample(192, 2 * N_, 2)] + stage(config[3], 32, 4) + [subpel_conv3x3(2 * N_, 320, 2)])
        width = lambda i, cap: 320 + (320 // num_slices) * min(i, cap)
        self.atten_mean = nn.ModuleList(nn.Sequential(SWAtten(width(i, 5), width(i, 5), 16, self.window_size, 0, inter_dim=128)) for i in range(num_slices))
        self.atten_scale = nn.ModuleList(nn.Sequential(SWAtten(width(i, 5), width(i, 5), 16, self.window_size, 0, inter_dim=128)) for i in range(num_slices))
        self.cc_mean_transforms = nn.ModuleList(slice_transform(width(i, 5), 320 // num_slices) for i in range(num_slices))
        self.cc_scale_transforms = nn.ModuleList(slice_transform(width(i, 5), 320 // num_slices) for i in range(num_slices))
        self.lrp_transforms = nn.ModuleList(slice_transform(width(i + 1, 6), 320 // num_slices) for i in range(num_slices))

        self.entropy_bottleneck = EntropyBottleneck(192)
        self.gaussian_conditional = GaussianConditional(None)

    def forward(self, x):
        """upstream models/tcm.py:437-486 (eval mode): x (B,3,H,W) -> {"x_hat", "likelihoods": {"y","z"}, "para": {"means","scales","y"}}.
        Every map stays NHWC between the first and the last line; likelihoods are fp32."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        return _slice_loop(self, self.g_a._nhwc(ops.to_nhwc(x, dtype=self._act_dtype())))

    def _act_dtype(self):
        return next(self.g_a.parameters()).dtype

    update = _codec_update
    load_state_dict = _codec_load_state_dict

    def compress(self, x, fmt: str = "chunked", chunk: int = bitstream.DEFAULT_CHUNK, graph: bool = False):
        """upstream models/tcm.py:511-570: x (B,3,H,W) -> {"strings": [y_strings, z_strings], "shape": z spatial size}; one string per
        image in each list.  fmt "chunked" (GPU coder) or "compressai" (one stream per image in CompressAI's layout, host coder).
        chunk: symbols per independent rANS stream of the chunked format -- a field of every container's header, so decompress() needs no argument;
        shorter chunks decode faster (more parallel lanes) and cost a 64-bit state flush + a 4-byte size each.
        graph (chunked only): replay the device half as a HIP graph captured per input shape (_codec_compress_graphed) -- same strings, less host time."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        if graph and fmt == "chunked":
            return _codec_compress_graphed(self, lambda t: self.g_a._nhwc(ops.to_nhwc(t, dtype=self._act_dtype())), [ops._req(x, "x")], chunk)
        return _codec_compress(self, self.g_a._nhwc(ops.to_nhwc(x, dtype=self._act_dtype())), fmt, chunk)

    def decompress(self, strings, shape, fmt: str = "chunked"):
        """upstream models/tcm.py:592-637: -> {"x_hat": (B,3,H,W) clamped to [0, 1]}."""
        if self.training:
            raise RuntimeError("realcamnet_amd is an inference path: call .eval() first")
        return {"x_hat": _codec_decompress(self, strings, shape, self._act_dtype(), fmt)}


CONTIGUOUS_OUTPUTS = False   # True: every tensor of the result dicts is materialised as a contiguous NCHW tensor, exactly like upstream's


def nchw_view(a):
    """(B,H,W,C) NHWC -> the same memory as a (B,C,H,W) tensor in torch's channels_last format: no copy, no launch.  The result dict's
    side outputs (likelihoods, means, scales, y, lft, lsc -- up to 2.3 GB at 4K) keep upstream's logical NCHW shapes, values and indexing
    this way; only x_hat is materialised planar.
    BOUNDARY NOTE (differs from upstream, which returns contiguous NCHW): these tensors satisfy `is_contiguous(memory_format=
    torch.channels_last)`, not plain `is_contiguous()`: `.view(...)` across the channel dimension raises (use `.reshape`), and consumers
    that need dense NCHW memory (`all_gather_into_tensor`, raw pointer / DLPack / NumPy hand-offs) must call `.contiguous()` -- or set
    `realcamnet_amd.tcm.CONTIGUOUS_OUTPUTS = True` to get upstream's layout everywhere at the cost of one transpose pass per tensor."""
    v = a.permute(0, 3, 1, 2)
    return v.contiguous() if CONTIGUOUS_OUTPUTS else v


def _fork_join(side_fn, main_fn, inputs):
    return ops.fork_join(side_fn, main_fn, inputs)


def _hyper_synthesis(m, z_hat):
    """(latent_scales, latent_means) = (h_scale_s(z_hat), h_mean_s(z_hat)): two independent transforms (models/tcm.py:449-450)."""
    return _fork_join(lambda: m.h_scale_s._nhwc(z_hat), lambda: m.h_mean_s._nhwc(z_hat), [z_hat])


def _slice_params(m, i, latent_means, latent_scales, y_hat_slices):
    """mean / scale of slice i from the hyper-prior maps and the already decoded slices (models/tcm.py:455-468)."""
    support = y_hat_slices if m.max_support_slices < 0 else y_hat_slices[:m.max_support_slices]

    def mean_branch():
        mean_support = m.atten_mean[i][0]._nhwc(ops.channel_concat([latent_means] + support))
        return mean_support, m.cc_mean_transforms[i]._nhwc(mean_support)

    def scale_branch():
        return m.cc_scale_transforms[i]._nhwc(m.atten_scale[i][0]._nhwc(ops.channel_concat([latent_scales] + support)))

    scale, (mean_support, mu) = _fork_join(scale_branch, mean_branch, [latent_scales] + support)
    return mean_support, mu, scale


def _refine(m, i, mean_support, y_hat_slice):
    """y_hat_slice + 0.5 tanh(lrp(cat(mean_support, y_hat_slice)))   (models/tcm.py:475-479)."""
    return ops.tanh_half_add(y_hat_slice, m.lrp_transforms[i]._nhwc(ops.channel_concat([mean_support, y_hat_slice])))


def _codec_compress_device(m, y, chunk):
    """The device half of compress(fmt="chunked"): h_a, the bottleneck's symbols, the hyper-synthesis and the slice loop, every container coded into its
    word arena -- NO host sync, so the whole of it can be enqueued ahead of the GPU or captured as a HIP graph.  -> (pending containers: z per image, then
    slice-major y per image; z's spatial size)."""
    gc = m.gaussian_conditional
    z = m.h_a._nhwc(y)
    z_pending, z_hat = m.entropy_bottleneck._compress_nhwc_async(z, chunk)
    latent_scales, latent_means = _hyper_synthesis(m, z_hat)
    if latent_means.shape[1:3] != y.shape[1:3]:
        raise NotImplementedError("latent size must be a multiple of 4; upstream crops here")
    b = y.shape[0]
    per = y.shape[-1] // m.num_slices
    tables, table = _coder_tables(gc), gc._table(y.device)
    y_hat_slices, pending = [], []
    for i in range(m.num_slices):
        mean_support, mu, scale = _slice_params(m, i, latent_means, latent_scales, y_hat_slices)
        sym, idx, y_hat_slice = torch.ops.realcam.gc_symbols(ops.channel_slice(y, i * per, per), mu, scale, table, gc.scale_bound_value)
        pending.extend(bitstream.encode_async(sym[k], idx[k], tables, chunk) for k in range(b))
        if i + 1 < m.num_slices:                     # (the last slice's refinement feeds nothing on the encoder side)
            y_hat_slices.append(_refine(m, i, mean_support, y_hat_slice))
    return z_pending + pending, tuple(z.shape[1:3])


def _codec_compress_finish(m, pending, shape, b):
    done = bitstream.finish(pending)                 # the one sync
    z_strings, done = done[:b], done[b:]
    return {"strings": [[b"".join(done[i * b + k] for i in range(m.num_slices)) for k in range(b)], z_strings], "shape": shape}


def _check_chunk(chunk):
    chunk = int(chunk)
    if not 1 <= chunk <= (1 << 24):
        raise ValueError("chunk: symbols per independent rANS stream, 1 .. 2^24")
    return chunk


def _codec_compress_graphed(m, latent_fn, inputs, chunk):
    """compress(fmt="chunked", graph=True): latent_fn(*inputs) -> y NHWC and _codec_compress_device captured as ONE HIP graph per input signature and chunk
    length (realcamnet_amd.graphs.GraphedCall; the capture lives on the module, `ops.invalidate_caches(module)` / update() drop it), replayed, then the
    usual single read-back.  At one 4K frame the host needs longer to enqueue the ~800 launches than the GPU to run them."""
    from .graphs import GraphedCall
    chunk = _check_chunk(chunk)
    graphs = m.__dict__.setdefault("_graphs", {})
    g = graphs.get(("compress", chunk))
    if g is None:
        g = graphs[("compress", chunk)] = GraphedCall(lambda *ts: _codec_compress_device(m, latent_fn(*ts), chunk))
    pending, shape = g(*inputs)
    return _codec_compress_finish(m, pending, shape, inputs[0].shape[0])


def _codec_compress(m, y, fmt, chunk=bitstream.DEFAULT_CHUNK):
    """Shared by TCM.compress and raw_compression_tcm_final.compress (models/tcm.py:515-570, raw2bit.py:1901-1944): y NHWC latent ->
    strings.  The symbols and CDF indexes of every slice are produced on the device (realcam::gc_symbols); "chunked": every slice
    becomes one container of GPU-coded chunk streams, an image's y string is the concatenation of its slices' containers -- the device half of
    every container is enqueued first, ONE sync (bitstream.finish) reads them all back;
    "compressai": the slices' symbols are concatenated and coded as ONE stream per image, as upstream's single BufferedRansEncoder."""
    chunk = _check_chunk(chunk)
    b = y.shape[0]
    if fmt == "chunked":
        pending, shape = _codec_compress_device(m, y, chunk)
        return _codec_compress_finish(m, pending, shape, b)
    gc = m.gaussian_conditional
    z = m.h_a._nhwc(y)
    z_strings, z_hat = m.entropy_bottleneck._compress_nhwc(z, fmt, chunk)
    latent_scales, latent_means = _hyper_synthesis(m, z_hat)
    if latent_means.shape[1:3] != y.shape[1:3]:
        raise NotImplementedError("latent size must be a multiple of 4; upstream crops here")
    per = y.shape[-1] // m.num_slices
    tables, table = _coder_tables(gc), gc._table(y.device)
    y_hat_slices, syms, idxs = [], [], []
    for i in range(m.num_slices):
        mean_support, mu, scale = _slice_params(m, i, latent_means, latent_scales, y_hat_slices)
        sym, idx, y_hat_slice = torch.ops.realcam.gc_symbols(ops.channel_slice(y, i * per, per), mu, scale, table, gc.scale_bound_value)
        syms.append(sym); idxs.append(idx)
        y_hat_slices.append(_refine(m, i, mean_support, y_hat_slice))
    sym, idx = torch.cat([t.reshape(b, -1) for t in syms], dim=1), torch.cat([t.reshape(b, -1) for t in idxs], dim=1)
    y_strings = [bitstream.encode(sym[k], idx[k], tables, fmt) for k in range(b)]
    return {"strings": [y_strings, z_strings], "shape": tuple(z.shape[1:3])}


def _codec_decompress(m, strings, shape, dtype, fmt):
    """Shared decompress (models/tcm.py:592-637, raw2bit.py:1961-2027): -> x_hat NCHW clamped to [0, 1]."""
    gc = m.gaussian_conditional
    y_strings, z_strings = strings
    z_decoders = []
    z_hat = m.entropy_bottleneck._decompress_nhwc(z_strings, tuple(shape), dtype, fmt, _decoders=z_decoders)
    latent_scales, latent_means = _hyper_synthesis(m, z_hat)
    b, dev = z_hat.shape[0], z_hat.device
    tables, table = _coder_tables(gc), gc._table(dev)
    decoders = [bitstream.Decoder(s, tables, dev, fmt) for s in y_strings]
    dec = (lambda d, ix: d.decode_async(ix)) if fmt == "chunked" else (lambda d, ix: d.decode(ix))      # chunked: the kernels' error flags are looked at ONCE, below
    y_hat_slices = []
    for i in range(m.num_slices):
        mean_support, mu, scale = _slice_params(m, i, latent_means, latent_scales, y_hat_slices)
        _, idx, _ = torch.ops.realcam.gc_symbols(None, None, scale, table, gc.scale_bound_value)
        sym = torch.stack([dec(decoders[k], idx[k]).view(idx.shape[1], idx.shape[2]) for k in range(b)])
        y_hat_slices.append(_refine(m, i, mean_support, torch.ops.realcam.gc_dequantize(sym, mu)))
    x_hat = _synthesis_nchw(m, ops.channel_concat(y_hat_slices)).clamp_(0, 1)
    for d in z_decoders + decoders:              # everything is enqueued: one look at the decode kernels' flags (a corrupt stream raises here, before x_hat is returned)
        d.check()
    return x_hat


def _synthesis_nchw(m, y_hat):
    """g_s(y_hat) as the NCHW image the module returns: when g_s ends in a narrow subpel_conv3x3 (conv -> PixelShuffle(2), 3 output
    channels) the shuffle writes NCHW directly instead of an NHWC shuffle followed by a 3-channel layout pass."""
    mods = list(m.g_s)
    last = mods[-1]
    if (isinstance(last, nn.Sequential) and len(last) == 2 and isinstance(last[0], N.Conv2d) and isinstance(last[1], nn.PixelShuffle) and
            last[1].upscale_factor == 2 and (last[0].out_channels // 4) % 16 != 0):
        a = y_hat
        for mod in mods[:-1]:
            a = mod._nhwc(a)
        if ops.FUSE_SHUFFLE_STORE:            # the conv's own store does the shuffle + planar layout (RC_OUT_PIXEL_SHUFFLE2_NCHW): no 12-channel map, no extra pass
            return last[0]._nhwc(a, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2_NCHW)
        return ops.pixel_shuffle2_nchw(last[0]._nhwc(a))
    return ops.to_nchw(m.g_s._nhwc(y_hat))


def _slice_loop(m, y):
    """h_a, entropy bottleneck, hyper-synthesis, the slice loop and g_s: identical in `TCM.forward` (models/tcm.py:439-486) and
    `raw_compression_tcm_final.forward` (models/raw2bit.py:1791-1846).  y NHWC; returns the NCHW result dict."""
    z = m.h_a._nhwc(y)
    z_hat, z_lik = m.entropy_bottleneck._nhwc(z)
    latent_scales, latent_means = _hyper_synthesis(m, z_hat)
    if latent_means.shape[1:3] != y.shape[1:3]:
        raise NotImplementedError("latent size must be a multiple of 4; upstream crops here")
    per = y.shape[-1] // m.num_slices
    y_hat_slices, y_lik, mu_list, scale_list = [], [], [], []
    for i in range(m.num_slices):
        y_slice = ops.channel_slice(y, i * per, per)
        mean_support, mu, scale = _slice_params(m, i, latent_means, latent_scales, y_hat_slices)
        y_hat_slice, lik = m.gaussian_conditional._nhwc(y_slice, scale, mu)
        y_hat_slices.append(_refine(m, i, mean_support, y_hat_slice))
        y_lik.append(lik); mu_list.append(mu); scale_list.append(scale)
    x_hat = _synthesis_nchw(m, ops.channel_concat(y_hat_slices))
    nchw = nchw_view
    return {"x_hat": x_hat, "likelihoods": {"y": nchw(ops.channel_concat(y_lik)), "z": nchw(z_lik)},
            "para": {"means": nchw(ops.channel_concat(mu_list)), "scales": nchw(ops.channel_concat(scale_list)), "y": nchw(y)}}
