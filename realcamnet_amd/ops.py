"""Host-side operator layer: thin, typed wrappers from torch tensors to the C ABI (include/realcam_hip.h).

PyTorch is used for device memory (caching allocator), the current HIP stream and nothing else.
Internal activations are NHWC tensors of shape (B, H, W, C), contiguous, fp32 or bf16.
There is NO fallback: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import os
import torch

from . import _lib
from . import torch_ops as _T  # noqa: F401  (registers torch.ops.realcam.*)
from ._lib import (RC_ACT_GELU, RC_ACT_LEAKY, RC_ACT_NONE, RC_ACT_RELU, RC_ACT_RELU_POST, RC_BF16, RC_F32, RC_OUT_NCHW, RC_OUT_NHWC, RC_OUT_NHWC_DWT,
                   RC_OUT_PIXEL_SHUFFLE2, RC_OUT_PIXEL_SHUFFLE2_NCHW, ConvDesc, ConvPairDesc, check)

_DT = {torch.float32: RC_F32, torch.bfloat16: RC_BF16}
_R = torch.ops.realcam          # every launch below goes through the dispatcher op registered in torch_ops.py


def _opt(t: torch.Tensor) -> Optional[torch.Tensor]:
    """An op's 'absent' output (an empty (0,) tensor) back to None."""
    return t if t.numel() else None

# Fold CALayer's "res*gate + skip" into the next conv's input staging (saves one full HBM pass per
# RCAB).  The unfused form (rc_gate_residual) is kept for A/B checks.
FUSE_GATE = True


def lib():
    return _lib.load()


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"realcamnet_amd: unsupported dtype {t.dtype} (fp32 / bf16 only)") from None


def _req(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path has no CPU fallback "
                           "(use oracle/ for CPU reference results)")
    return t if t.is_contiguous() else t.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------------------
# parameter caches (never part of the state_dict)
# --------------------------------------------------------------------------------------------------
def _key(*params) -> tuple:
    """Cache key of derived parameters: storage address + version counter + dtype + device.  NOTE: an in-place edit through
    `.data` (`w.data.copy_()`, EMA swaps) does not bump `_version`; call invalidate_caches(module) after such edits."""
    from torch._subclasses.fake_tensor import FakeTensor
    return tuple(((id(p) if isinstance(p, FakeTensor) else p.data_ptr()), p._version, p.dtype, str(p.device)) if p is not None else None
                 for p in params)


def invalidate_caches(module) -> None:
    """Drop every derived-parameter cache (packed MFMA weights, fp32 views, folded BatchNorm, CDF tables ...) below `module`.
    load_state_dict / optimizer steps / .to() are detected automatically; edits through `.data` are not."""
    for m in module.modules():
        for k in ("_rc_cache", "_pk", "_eff", "_coder_tables", "_f32_masters", "_graphs", "_f32_verdict"):
            m.__dict__.pop(k, None)


def _cache(mod) -> dict:
    c = mod.__dict__.get("_rc_cache")
    if c is None:
        c = {}
        mod.__dict__["_rc_cache"] = c
    return c


def f32_param(mod, name: str) -> torch.Tensor:
    """fp32 contiguous device view/copy of a (small) parameter, cached per parameter version."""
    p = getattr(mod, name)
    if p.dtype == torch.float32 and p.is_contiguous():
        return _req(p.detach(), name)
    c = _cache(mod)
    k = ("f32", name)
    hit = c.get(k)
    key = _key(p)
    if hit is None or hit[0] != key:
        hit = (key, _req(p.detach(), name).float().contiguous())
        c[k] = hit
    return hit[1]


class PackedConv:
    __slots__ = ("wpacked", "bias", "cin", "cout", "ksize", "dtype", "out_mode")


def packed_conv(mod, act_dtype: torch.dtype, out_mode: int, cout_tile: int = 0) -> PackedConv:
    """MFMA-fragment-ordered copy of a conv's weights (realcam::conv_pack_weights), cached on the module (one copy per cout tile width in use)."""
    w, b = mod.weight, mod.bias
    c = _cache(mod)
    k = ("conv", act_dtype, out_mode, cout_tile)
    key = (_key(w, b), _lib.knob(b"conv32"))      # the 32x32x16 layers' packed order depends on that knob (and on nothing else); mirrored host-side, no C call per conv
    hit = c.get(k)
    if hit is not None and hit[0] == key:
        return hit[1]
    if not w.is_cuda:
        raise RuntimeError("conv weights are not on a HIP device; move the module with .cuda() first")
    if w.dim() == 2:                       # nn.Linear over tokens == 1x1 convolution over NHWC pixels
        cout, cin = w.shape
        kh = kw = 1
    else:
        cout, cin, kh, kw = w.shape
    if kh != kw or kh not in (1, 2, 3, 5):                 # 2: the {-1, 0}^2 window of a stride-2 3x3 conv over its space-to-depth map; 5: the folded tail
        raise NotImplementedError(f"HIP conv supports 1x1 and 3x3 kernels, got {kh}x{kw}")
    wp, bp = _R.conv_pack_weights(w.detach(), b.detach() if b is not None else None, act_dtype, out_mode, cout_tile)
    pc = PackedConv()
    pc.wpacked = wp
    pc.bias = bp if b is not None else None
    pc.cin, pc.cout, pc.ksize, pc.dtype, pc.out_mode = cin, cout, kh, _DT[act_dtype], out_mode
    c[k] = (key, pc)
    return pc


def packed_chain(mod):
    """(packed bf16 MFMA fragments, packed fp32 bias) of an nn.Linear / 1x1 conv for the register-resident layer chains of
    csrc/gma_fused.hip (realcam::chain_pack_weights), cached on the module."""
    w, b = mod.weight, mod.bias
    c = _cache(mod)
    key = _key(w, b)
    hit = c.get("chain")
    if hit is not None and hit[0] == key:
        return hit[1]
    if not w.is_cuda:
        raise RuntimeError("weights are not on a HIP device; move the module with .cuda() first")
    if w.dim() == 4 and tuple(w.shape[2:]) != (1, 1):
        raise NotImplementedError("layer chains take Linear / 1x1 weights")
    pair = _R.chain_pack_weights(w.detach().reshape(w.shape[0], w.shape[1]), b.detach() if b is not None else None)
    c["chain"] = (key, pair)
    return pair


def packed_chain_natural(mod):
    """bf16 MFMA A fragments of an nn.Linear / 1x1 conv with the output rows in NATURAL channel order (16-row tile m = channels 16 m ..), for
    kernels that consume the accumulators by 16-channel segment (realcam::gma_qkv_aggregate); cached on the module."""
    w = mod.weight
    c = _cache(mod)
    key = _key(w)
    hit = c.get("chain_natural")
    if hit is not None and hit[0] == key:
        return hit[1]
    if not w.is_cuda:
        raise RuntimeError("weights are not on a HIP device; move the module with .cuda() first")
    packed = _R.chain_pack_weights_natural(w.detach().reshape(w.shape[0], w.shape[1]))
    c["chain_natural"] = (key, packed)
    return packed


# The per-token stages of GMA_Block (LayerNorm1 + qkv; attention read-out + proj + LayerNorm2 + MLP [+ the net's output conv]) as
# two launches with register-resident activations (csrc/gma_fused.hip) instead of nine layer-by-layer ones.  Built for dim 80, bf16.
FUSE_GMA = True
# LayerNorm1 + qkv + Aggregator as one launch (rc_gma_qkv_aggregate: qkv never reaches HBM) instead of rc_gma_ln_qkv + rc_gma_aggregate; same bits.
FUSE_GMA_FRONT = os.environ.get("RC_GMA_FRONT", "1") != "0"
FUSE_GMA_ENTRY = os.environ.get("RC_GMA_ENTRY", "1") != "0"    # gma_in (1x1 192 -> 80) + ConvPosEnc in one launch (realcam::gma_in_cpe); 0: two launches (A/B, tests)


class _ConvView:
    """A conv-shaped view (weight, bias) of another module's parameters, with its own pack cache."""
    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


def split_conv_views(mod, sizes):
    """Views of a convolution's output channels [0, s0), [s0, s0 + s1), ...: `torch.split(conv(x), sizes, dim=1)` (upstream models/tcm.py:261)
    as separate convolutions of the same input -- the input is read once more, the full-width result and its slice copies never exist."""
    cache = _cache(mod)
    key = _key(mod.weight, mod.bias)
    hit = cache.get("split_views")
    if hit is None or hit[0] != (key, tuple(sizes)):
        views, c0 = [], 0
        for n in sizes:
            views.append(_ConvView(mod.weight.detach()[c0:c0 + n], mod.bias.detach()[c0:c0 + n] if mod.bias is not None else None))
            c0 += n
        if c0 != mod.weight.shape[0]:
            raise ValueError("split_conv_views: sizes do not add up to the convolution's output channels")
        hit = ((key, tuple(sizes)), views)
        cache["split_views"] = hit
    return hit[1]


def split_conv_input_views(mod, sizes):
    """Views of a convolution's INPUT channels [0, s0), [s0, s0 + s1), ...: conv(torch.cat(parts, dim=1)) = sum_i conv_i(parts[i]) (only the
    first view carries the bias).  Lets `intro(torch.cat([raw, coord], 1))` (upstream models/LiteISP.py:1497) run without the 6-channel
    concatenated map -- whose 8-byte coord pixels are no whole 16-byte vectors for rc_channel_copy anyway."""
    cache = _cache(mod)
    key = _key(mod.weight, mod.bias)
    hit = cache.get("split_in_views")
    if hit is None or hit[0] != (key, tuple(sizes)):
        views, c0 = [], 0
        for i, n in enumerate(sizes):
            views.append(_ConvView(mod.weight.detach()[:, c0:c0 + n].contiguous(), mod.bias.detach() if (mod.bias is not None and i == 0) else None))
            c0 += n
        if c0 != mod.weight.shape[1]:
            raise ValueError("split_conv_input_views: sizes do not add up to the convolution's input channels")
        hit = ((key, tuple(sizes)), views)
        cache["split_in_views"] = hit
    return hit[1]


def is_down2x2(mod) -> bool:
    """nn.Conv2d(c, c2, kernel_size=2, stride=2) without padding: the ISPUNet family's down-sampler (LiteISP.py:1253)."""
    return (tuple(mod.kernel_size) == (2, 2) and tuple(mod.stride) == (2, 2) and tuple(mod.padding) == (0, 0) and
            tuple(mod.dilation) == (1, 1) and mod.groups == 1)


def is_stride2(mod) -> bool:
    """nn.Conv2d(k in {1,3}, stride=2, padding=k//2): CompressAI's conv3x3(stride=2) / conv1x1(stride=2) (models/tcm.py:336-345)."""
    k = mod.kernel_size[0]
    return (tuple(mod.kernel_size) in ((1, 1), (3, 3)) and tuple(mod.stride) == (2, 2) and tuple(mod.padding) == (k // 2, k // 2) and
            tuple(mod.dilation) == (1, 1) and mod.groups == 1 and getattr(mod, "padding_mode", "zeros") == "zeros")


def subsample2(x: torch.Tensor) -> torch.Tensor:
    """x[:, ::2, ::2, :] of an NHWC map as a new dense tensor (rc_subsample2)."""
    return _R.subsample2(_req(x, "subsample2 input"))


def upsample_bilinear2(x: torch.Tensor) -> torch.Tensor:
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) on an NHWC map (upstream models/raw2bit.py:790-793)."""
    return _R.upsample_bilinear2(_req(x, "upsample_bilinear2 input"))


def sft_apply(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, identity: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x*scale + shift + x (+ identity): SpatialFeatureTransform with residual=True (upstream models/raw2bit.py:877-885)."""
    x, scale, shift = _req(x, "x"), _req(scale, "scale"), _req(shift, "shift")
    if identity is not None:
        identity = _req(identity, "identity")
    for t in (scale, shift) + ((identity,) if identity is not None else ()):
        if t.shape != x.shape or t.dtype != x.dtype:
            raise ValueError("sft_apply: shape / dtype mismatch")
    return _R.sft_apply(x, scale, shift, identity)


def ca_gate_linear(sums: torch.Tensor, hw: int, fc0, fc1) -> torch.Tensor:
    """CALayer gate from channel partial sums for the bias-free nn.Linear form (upstream models/raw2bit.py:238-254)."""
    b, nt, c = sums.shape
    cr = fc0.weight.shape[0]
    z = _const(0.0, (max(c, cr),), sums)
    return _R.ca_gate(sums, int(hw), f32_param(fc0, "weight"), z, f32_param(fc1, "weight"), z)


def pixel_shuffle2_nchw(x: torch.Tensor) -> torch.Tensor:
    """nn.PixelShuffle(2) of an NHWC map (B,H,W,4c) written as NCHW (B,c,2H,2W): shuffle + module-boundary layout in one pass."""
    x = _req(x, "pixel_shuffle2_nchw input")
    if x.dim() != 4 or x.shape[-1] % 4:
        raise ValueError("pixel_shuffle2_nchw: (B,H,W,4c) expected")
    return _R.pixel_shuffle2_nchw(x)


def space_to_depth2(x: torch.Tensor) -> torch.Tensor:
    """(B,H,W,c) -> (B,ceil(H/2),ceil(W/2),4c), channel (2i+j)*c + k <- pixel (2y+i, 2x+j), zero beyond the edge."""
    return _R.space_to_depth2(_req(x, "space_to_depth2 input"))


def _stride2_view(mod) -> "_ConvView":
    """Weights of a kxk stride-2 conv re-indexed for the space-to-depth map: input pixel offset d in {-1, 0, +1} of the strided
    conv is (phase 1, offset -1), (phase 0, offset 0), (phase 1, offset 0) of the half-resolution map."""
    cache = _cache(mod)
    key = _key(mod.weight, mod.bias)
    hit = cache.get("stride2_s2d")
    if hit is None or hit[0] != key:
        w = mod.weight.detach().float().cpu()
        cout, c, k, _ = w.shape
        # bf16: the 3x3 case only touches map offsets {-1, 0}^2 -- a 2x2 window (rc_conv2d ksize 2): 16 (tap, phase) blocks of which 9 are
        # non-zero, instead of the 36 of the 3x3 embedding (fp32 and odd channel counts keep the 3x3 form)
        kk = 2 if (k == 3 and mod.weight.dtype == torch.bfloat16 and
                   _lib.load().rc_conv_packed_bytes(4 * c, cout, 2, _DT[torch.bfloat16], RC_OUT_NHWC) != 0) else k
        w3 = torch.zeros(cout, 4 * c, kk, kk)
        if k == 3:
            place = {-1: (1, -1), 0: (0, 0), 1: (1, 0)}
            for dy, (i, oy) in place.items():
                for dx, (j, ox) in place.items():
                    ph = 2 * i + j
                    w3[:, ph * c:(ph + 1) * c, oy + 1, ox + 1] = w[:, :, dy + 1, dx + 1]
        else:
            w3[:, :c, 0, 0] = w[:, :, 0, 0]
        view = _ConvView(w3.to(mod.weight.device, mod.weight.dtype), mod.bias.detach() if mod.bias is not None else None)
        hit = (key, view)
        cache["stride2_s2d"] = hit
    return hit[1]


# stride-2 3x3 convolutions with 64 | channels read their input directly (no rc_space_to_depth2 pass); False: always via the map
FOLD_STRIDE2 = True


def conv_stride2(x: torch.Tensor, mod, s2d: Optional[torch.Tensor] = None, **fuse):
    """kxk stride-2 padding-k//2 convolution (k in {1, 3}) as a stride-1 rc_conv2d at the OUTPUT resolution over the
    space-to-depth map (4c channels, re-indexed taps; bf16: a 2x2 window, 9 of 16 (tap, phase) blocks non-zero; fp32: the 3x3 embedding,
    9 of 36).  1x1 with a vectorisable channel count: sample (rc_subsample2), then convolve."""
    x = _req(x, "conv_stride2 input")
    if any(k not in ("act", "slope") for k in fuse):
        raise NotImplementedError("conv_stride2: only an activation can be fused")
    unit = 16 // x.element_size()
    if mod.kernel_size[0] == 1 and x.shape[-1] % unit == 0:
        cache = _cache(mod)
        view = cache.get("stride2_view")
        if view is None or view.weight is not mod.weight or view.bias is not mod.bias:
            view = cache["stride2_view"] = _ConvView(mod.weight, mod.bias)
        return conv2d(subsample2(x), view, **fuse)
    view = _stride2_view(mod)
    if FOLD_STRIDE2 and s2d is None and view.weight.shape[-1] == 2 and x.shape[-1] % 64 == 0:
        # the 2x2-window kernel gathers the space-to-depth channels itself (rc_conv_desc.src_h / src_w): no map is written or re-read
        pc = packed_conv(view, x.dtype, RC_OUT_NHWC)
        return _R.conv2d_fold2(x, pc.wpacked, pc.bias, pc.cout, _ACT[fuse.get("act")], float(fuse.get("slope", 0.0)))
    return conv2d(space_to_depth2(x) if s2d is None else s2d, view, **fuse)     # s2d: the caller's shared space-to-depth map of x


def entropy_bottleneck(z: torch.Tensor, params: torch.Tensor, medians: torch.Tensor, bound: float = 1e-9):
    """EntropyBottleneck likelihood path on an NHWC latent: returns (z_hat, likelihood fp32).  params (C,58) / medians (C) fp32
    device tensors as rc_entropy_bottleneck documents them."""
    z, params, medians = _req(z, "z"), _req(params, "params"), _req(medians, "medians")
    c = z.shape[-1]
    if tuple(params.shape) != (c, 58) or tuple(medians.shape) != (c,) or params.dtype != torch.float32 or medians.dtype != torch.float32:
        raise ValueError("entropy_bottleneck: params must be fp32 (C,58), medians fp32 (C)")
    return _R.entropy_bottleneck(z, params, medians, float(bound))


def gaussian_conditional(y: torch.Tensor, scale: torch.Tensor, mu: torch.Tensor, scale_bound: float = 0.11, bound: float = 1e-9):
    """GaussianConditional likelihood path: returns (y_hat = ste_round(y - mu) + mu, likelihood fp32)."""
    y, scale, mu = _req(y, "y"), _req(scale, "scale"), _req(mu, "mu")
    if y.shape != scale.shape or y.shape != mu.shape or y.dtype != scale.dtype or y.dtype != mu.dtype:
        raise ValueError("gaussian_conditional: shape / dtype mismatch")
    return _R.gaussian_conditional(y, scale, mu, float(scale_bound), float(bound))


def tanh_half_add(a: torch.Tensor, lrp: torch.Tensor) -> torch.Tensor:
    """a + 0.5 * tanh(lrp)  (upstream models/tcm.py:478-479)."""
    a, lrp = _req(a, "a"), _req(lrp, "lrp")
    if a.shape != lrp.shape or a.dtype != lrp.dtype:
        raise ValueError("tanh_half_add: shape / dtype mismatch")
    return _R.tanh_half_add(a, lrp)


def pixel_shuffle2(x: torch.Tensor) -> torch.Tensor:
    """nn.PixelShuffle(2) on an NHWC map (B,H,W,4c) -> (B,2H,2W,c), any c (rc_pixel_shuffle2)."""
    x = _req(x, "pixel_shuffle2 input")
    if x.shape[-1] % 4:
        raise ValueError("pixel_shuffle2: channels must be a multiple of 4")
    return _R.pixel_shuffle2(x)


def square(x: torch.Tensor) -> torch.Tensor:
    return _R.square(_req(x, "square input"))


def gdn_apply(x: torch.Tensor, norm: torch.Tensor, inverse: bool, identity: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x * rsqrt(norm) (GDN) or x * sqrt(norm) (inverse GDN), + identity."""
    x, norm = _req(x, "x"), _req(norm, "norm")
    if x.shape != norm.shape or x.dtype != norm.dtype:
        raise ValueError("gdn_apply: shape / dtype mismatch")
    if identity is not None:
        identity = _req(identity, "identity")
        if identity.shape != x.shape or identity.dtype != x.dtype:
            raise ValueError("gdn_apply: identity shape / dtype mismatch")
    return _R.gdn_apply(x, norm, bool(inverse), identity)


def conv2x2s2(x: torch.Tensor, mod, **fuse):
    """2x2 stride-2 convolution = space-to-depth (channel 4c + 2i + j <- pixel (2y+i, 2x+j), done by the Haar kernel with
    one-hot taps: exact) followed by a 1x1 MFMA convolution over the 4c channels.  The OIHW weight flattened over
    (c, i, j) already has that channel order, so the 1x1 weight is a reshape of the checkpoint tensor."""
    x = _req(x, "conv2x2s2 input")
    b, H, W, c = x.shape
    if H % 2 or W % 2:
        raise ValueError(f"stride-2 conv needs even H,W; got {H}x{W}")
    cout, cin = mod.weight.shape[:2]
    if cin != c:
        raise ValueError(f"conv expects {cin} input channels, got {c}")
    cache = _cache(mod)
    key = _key(mod.weight, mod.bias)
    hit = cache.get("down2x2")
    if hit is None or hit[0] != key:
        taps = torch.zeros((4 * c, 1, 2, 2), dtype=torch.float32)
        for k in range(4):
            taps[k::4, 0, k >> 1, k & 1] = 1.0
        view = _ConvView(mod.weight.detach().reshape(cout, 4 * c, 1, 1), mod.bias.detach() if mod.bias is not None else None)
        hit = (key, taps.to(x.device), view)
        cache["down2x2"] = hit
    _, taps, view = hit
    return conv2d(_R.haar_dwt(x, taps, False), view, **fuse)     # one-hot 2x2 taps, the same for every channel


def check_conv_module(mod) -> None:
    if isinstance(mod, _ConvView):
        return
    if tuple(mod.stride) != (1, 1) or tuple(mod.dilation) != (1, 1) or mod.groups != 1:
        raise NotImplementedError("HIP conv: stride 1, dilation 1, groups 1 only")
    k = mod.kernel_size[0]
    if tuple(mod.padding) != (k // 2, k // 2) or getattr(mod, "padding_mode", "zeros") != "zeros":
        raise NotImplementedError("HIP conv: 'same' zero padding only")


# --------------------------------------------------------------------------------------------------
# layout / ingest
# --------------------------------------------------------------------------------------------------
def to_nhwc(x: torch.Tensor, dtype: Optional[torch.dtype] = None, pad_hw: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """NCHW (B,C,h,w) -> NHWC (B,hp,wp,C), zero padded bottom/right, converted to `dtype`."""
    x = _req(x, "to_nhwc input")
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got shape {tuple(x.shape)}")
    hp, wp = pad_hw if pad_hw is not None else (x.shape[2], x.shape[3])
    dtype = dtype or x.dtype
    _dt(x), _DT[dtype]
    return _R.nchw_to_nhwc(x, dtype, int(hp), int(wp))


def to_nchw(a: torch.Tensor, dtype: Optional[torch.dtype] = None, crop_hw: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    a = _req(a, "to_nchw input")
    h, w = crop_hw if crop_hw is not None else (a.shape[1], a.shape[2])
    dtype = dtype or a.dtype
    _dt(a), _DT[dtype]
    return _R.nhwc_to_nchw(a, dtype, int(h), int(w))


def _mosaic3(mosaic: torch.Tensor) -> torch.Tensor:
    mosaic = _req(mosaic, "mosaic")
    if mosaic.dim() == 4:
        if mosaic.shape[1] != 1:
            raise ValueError("mosaic must have one channel")
        mosaic = mosaic[:, 0]
    if mosaic.dim() != 3 or mosaic.shape[1] % 2 or mosaic.shape[2] % 2:
        raise ValueError("mosaic must be (B,[1,]2h,2w) with even height/width")
    return mosaic


def bayer_unshuffle(mosaic: torch.Tensor, dtype: Optional[torch.dtype] = None, pad_to: int = 1) -> torch.Tensor:
    """(B,1,2h,2w) or (B,2h,2w) Bayer mosaic -> packed NHWC (B,hp,wp,4), zero padded to a multiple of
    `pad_to` (README.md:33-38 'Unpixel shuffle' + models/LiteISP.py:84-105)."""
    mosaic = _mosaic3(mosaic)
    dtype = dtype or mosaic.dtype
    _dt(mosaic), _DT[dtype]
    return _R.bayer_unshuffle(mosaic, dtype, int(pad_to))


def raw_ingest(mosaic: torch.Tensor, dtype: Optional[torch.dtype] = None, pad_to: int = 16, black_level: float = 0.0, white_level: float = 1.0,
               cond_hw: Tuple[int, int] = (256, 256)):
    """The RAW ingest step in front of the path (the 'Unpixel shuffle' and 'Resize' boxes of
    assets/networkarch.png), one launch: sensor mosaic (B,[1,]2h,2w) -> normalise (v - black) / (white - black) ->
    Bayer unshuffle -> zero-pad to `pad_to` -> packed NHWC (B,hp,wp,4); and cond (B,4,ch,cw) NCHW = bilinear resize
    (align_corners=False, as F.interpolate) of the un-padded normalised packed RAW.  Returns (packed, cond)."""
    mosaic = _mosaic3(mosaic)
    if not white_level > black_level:
        raise ValueError("white_level must exceed black_level")
    if mosaic.dtype == torch.uint16:
        if dtype is None:
            raise ValueError("raw_ingest: give the activation dtype for a uint16 mosaic")
    else:
        _dt(mosaic)
        dtype = dtype or mosaic.dtype
    _DT[dtype]
    return _R.raw_ingest(mosaic, dtype, int(pad_to), float(black_level), float(white_level), int(cond_hw[0]), int(cond_hw[1]))


def make_coord(b: int, h: int, w: int, device=None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Normalised pixel-coordinate map (B,2,h,w) in [-1,1], channel 0 = y, channel 1 = x: the lens-shading branch's input x[2]
    (upstream never published its generator; build convention).  Plain tensor construction, no kernel."""
    ys = torch.linspace(-1.0, 1.0, h, device=device).view(1, 1, h, 1).expand(b, 1, h, w)
    xs = torch.linspace(-1.0, 1.0, w, device=device).view(1, 1, 1, w).expand(b, 1, h, w)
    return torch.cat([ys, xs], dim=1).to(dtype).contiguous()


# --------------------------------------------------------------------------------------------------
# convolution and friends
# --------------------------------------------------------------------------------------------------
# "relu_post": ReLU applied after the residual add, relu(conv(x) + residual) (CompressAI ResidualUnit)
_ACT = {None: RC_ACT_NONE, "relu": RC_ACT_RELU, "leaky": RC_ACT_LEAKY, "gelu": RC_ACT_GELU, "relu_post": RC_ACT_RELU_POST}


# fp32 3x3 layers with 64 | cout on maps too small to fill the chip with 64-wide cout tiles take 16-wide ones (rc_conv_desc.cout_tile): the general kernel launches one
# block per (8 x 32 tile, cout tile), and cfg2's 128 -> 128 levels are 272 / 72 such blocks for 256 CUs at 1080p, B = 1 (52 TF/s against the 64 -> 64 level's 115)
SMALL_MAP_COUT_TILE = os.environ.get("RC_SMALL_MAP_COUT_TILE", "1") != "0"        # (the env switch is for A/B runs of bench.py)
_CU_COUNT = {}
_SMALL_MAP_FACTOR = int(os.environ.get("RC_SMALL_MAP_FACTOR", "2"))           # blocks of 64 couts below this many per CU -> 16-wide tiles


def small_map_cout_tile(x: torch.Tensor, mod, out_mode: int) -> int:
    w = mod.weight
    if not SMALL_MAP_COUT_TILE or x.dtype != torch.float32 or w.dim() != 4 or w.shape[-1] != 3 or w.shape[0] % 64 != 0 or w.shape[1] % 16 != 0:
        return 0
    if out_mode not in (RC_OUT_NHWC, RC_OUT_NCHW):
        return 0
    dev = x.device.index if x.device.index is not None else 0
    cus = _CU_COUNT.get(dev)
    if cus is None:
        try:
            cus = torch.cuda.get_device_properties(dev).multi_processor_count if x.is_cuda else 256
        except (RuntimeError, AssertionError):       # FakeTensor traces on a machine without a GPU
            cus = 256
        _CU_COUNT[dev] = cus
    b, H, W, _ = x.shape
    blocks = b * ((H + 7) // 8) * ((W + 31) // 32) * (w.shape[0] // 64)
    return 16 if blocks < _SMALL_MAP_FACTOR * cus else 0


# Winograd F(2x2, 3x3) for the stride-1 3x3 layers (rc_conv_desc.algo = 1, csrc/wino.hip): 2.25x fewer multiplications; fp32 results differ from the implicit GEMM's by
# rounding only.  RC_WINOGRAD=0 (or ops.WINOGRAD = False) keeps every layer on the implicit GEMM (A/B runs, tests).
WINOGRAD = os.environ.get("RC_WINOGRAD", "1") != "0"


def winograd_ok(x: torch.Tensor, mod, *, act=None, gate=None, skip=None, mul_plus1=None, out_mode: int = RC_OUT_NHWC, store_input: bool = False) -> bool:
    """Does this conv call take the Winograd form?  fp32 3x3 stride-1 layers with cin % 8 == 0 and cout % 16 == 0, a plain NHWC store and the
    epilogue forms the kernel has (bias, film, none / relu / leaky / relu_post, out_scale, residual, channel sums)."""
    w = mod.weight
    if not WINOGRAD or x.dtype != torch.float32 or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        return False
    if w.shape[1] % 8 != 0 or w.shape[0] % 64 != 0 or out_mode != RC_OUT_NHWC:       # the library also runs cout % 16 == 0 (narrower blocks: one wave per SIMD); the
        return False                                                                    # measured win is on the 64-cout-per-block form (tools/wino_probe.py)
    if gate is not None or skip is not None or mul_plus1 is not None or store_input or act == "gelu":
        return False
    return True


def packed_wino(mod, act_dtype: torch.dtype) -> PackedConv:
    """U = G g G^T of a 3x3 conv in the Winograd kernel's fragment order (realcam::wino_pack_weights) + its bias in natural order, cached on the module."""
    w, b = mod.weight, mod.bias
    c = _cache(mod)
    k = ("wino", act_dtype)
    key = _key(w, b)
    hit = c.get(k)
    if hit is not None and hit[0] == key:
        return hit[1]
    if not w.is_cuda:
        raise RuntimeError("conv weights are not on a HIP device; move the module with .cuda() first")
    pc = PackedConv()
    pc.wpacked = _R.wino_pack_weights(w.detach(), act_dtype)
    pc.bias = b.detach().float().contiguous() if b is not None else None
    pc.cout, pc.cin, pc.ksize, pc.dtype, pc.out_mode = w.shape[0], w.shape[1], 3, _DT[act_dtype], RC_OUT_NHWC
    c[k] = (key, pc)
    return pc


def conv2d(x: torch.Tensor, mod, *, act: Optional[str] = None, slope: float = 0.0,
           residual: Optional[torch.Tensor] = None, mul_plus1: Optional[torch.Tensor] = None,
           film: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
           gate: Optional[torch.Tensor] = None, skip: Optional[torch.Tensor] = None, store_input: bool = False,
           out_mode: int = RC_OUT_NHWC, want_sums: bool = False,
           crop_hw: Optional[Tuple[int, int]] = None, out_dtype: Optional[torch.dtype] = None,
           out_scale: Optional[torch.Tensor] = None):
    """KxK stride-1 'same' convolution of NHWC `x` with `mod`'s weights (an nn.Conv2d-shaped module).

    gate/skip: the conv input is x*gate[b,c] + skip (CALayer gate + RCAB skip); with store_input the
    combined tensor is also materialised and returned.
    out_scale: (B, cout) fp32, multiplies the result before the residual is added (the CALayer gate of this conv's own output, ca_gate_ahead).
    Returns out, or a tuple (out, [stored_input], [chan_sums]) when extras are requested.
    """
    if mod.weight.dim() == 4:
        check_conv_module(mod)
    x = _req(x, "conv input")
    b, H, W, cin = x.shape
    algo = 1 if mod.weight.dim() == 4 and winograd_ok(x, mod, act=act, gate=gate, skip=skip, mul_plus1=mul_plus1, out_mode=out_mode, store_input=store_input) else 0
    if algo:
        ct = 0
        pc = packed_wino(mod, x.dtype)
    else:
        ct = small_map_cout_tile(x, mod, out_mode)
        pc = packed_conv(mod, x.dtype, RC_OUT_NHWC if out_mode == RC_OUT_NHWC_DWT else out_mode, ct)      # conv -> DWT: the NHWC layer's packed weights
    if cin != pc.cin:
        raise ValueError(f"conv expects {pc.cin} input channels, got {cin}")
    if gate is not None:
        if skip is None:
            raise ValueError("gate needs skip")
        skip = _req(skip, "skip")
        gate = _req(gate, "gate")
        if skip.shape != x.shape or skip.dtype != x.dtype or gate.shape != (b, cin) or gate.dtype != torch.float32:
            raise ValueError("gate/skip shape or dtype mismatch")
    fs = ft = None
    if film is not None:
        fs, ft = (_req(t, "film") for t in film)
        if fs.shape != (b, pc.cout) or ft.shape != (b, pc.cout) or fs.dtype != torch.float32 or ft.dtype != torch.float32:
            raise ValueError("film tensors must be fp32 (B, cout)")
    if mul_plus1 is not None:
        mul_plus1 = _req(mul_plus1, "mul_plus1")
    if out_scale is not None:
        out_scale = _req(out_scale, "out_scale")
        if out_scale.shape != (b, pc.cout) or out_scale.dtype != torch.float32 or out_mode != RC_OUT_NHWC:
            raise ValueError("out_scale must be fp32 (B, cout) and needs the NHWC store")
    if residual is not None:
        residual = _req(residual, "residual")
    for name, t in (("mul_plus1", mul_plus1), ("residual", residual)):
        if t is not None and (t.shape != (b, H, W, pc.cout) or t.dtype != x.dtype):
            raise ValueError(f"{name} must be NHWC {(b, H, W, pc.cout)} {x.dtype}, got {tuple(t.shape)} {t.dtype}")
    planar = out_mode in (RC_OUT_NCHW, RC_OUT_PIXEL_SHUFFLE2_NCHW)
    ch, cw = (crop_hw if crop_hw is not None else (0, 0)) if planar else (0, 0)
    if out_dtype is not None:
        _DT[out_dtype]
    out, stored, sums = _R.conv2d(x, pc.wpacked, pc.bias, pc.cout, pc.ksize, _ACT[act], float(slope), residual, mul_plus1, fs, ft, gate, skip,
                                  bool(store_input), int(out_mode), bool(want_sums), int(ch), int(cw),
                                  out_dtype if planar else None, out_scale, ct, algo)
    extras = [t for t in (_opt(stored), _opt(sums)) if t is not None]
    return (out, *extras) if extras else out


FUSE_DWT = os.environ.get("RC_FUSE_DWT", "1") != "0"   # conv [+ act] -> DWTForward as one launch (RC_OUT_NHWC_DWT) where the kernel has that form; 0: two launches (A/B, tests)
_HAAR_TAPS = (0.5, 0.5, 0.5, 0.5, 0.5, 0.5, -0.5, -0.5, 0.5, -0.5, 0.5, -0.5, 0.5, -0.5, -0.5, 0.5)


def conv_dwt_ok(x: torch.Tensor, conv, dwt, act: Optional[str] = None, slope: float = 0.0, residual: bool = False) -> bool:
    """Can `conv [+ act | + residual] -> dwt` (networks.Conv2d, networks.DWTForward; upstream models/LiteISP.py:1950-1958: `down1`'s closing conv, `down2`'s RCAGroup
    = conv + group skip) run as ONE rc_conv2d launch with RC_OUT_NHWC_DWT?  The kernel form exists for the bf16 3x3 layers of the wave-autonomous kernel
    (cin == cout == 32 or 48) and computes with the reference's frozen Haar taps, so the module's taps must BE those (checked once per write of the tap tensor)."""
    if residual and act is not None:
        return False
    if not FUSE_DWT or x.dtype != torch.bfloat16 or x.dim() != 4 or conv.weight.dim() != 4:
        return False
    cout, cin, kh, kw = conv.weight.shape
    if not (kh == kw == 3 and cin == cout and cin in (32, 48) and x.shape[3] == cin and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0):
        return False
    if tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (1, 1) or tuple(conv.dilation) != (1, 1) or conv.groups != 1:
        return False
    if act not in (None, "relu", "leaky") or (act == "leaky" and not (0.0 <= slope <= 1.0)):
        return False
    w = getattr(dwt, "weight", None)
    if w is None or tuple(w.shape) != (4 * cout, 1, 2, 2):
        return False
    from torch._subclasses.fake_tensor import FakeTensor
    if isinstance(w, FakeTensor):                 # shape tracing: values are not there to look at; a DWTForward is built with these taps and frozen
        return True
    c = _cache(dwt)
    hit = c.get("haar_ok")
    if hit is None or hit[0] != _key(w):
        want = torch.tensor(_HAAR_TAPS, dtype=torch.float32).reshape(4, 1, 2, 2).repeat(cout, 1, 1, 1)
        hit = (_key(w), bool(torch.equal(w.detach().float().cpu(), want)))
        c["haar_ok"] = hit
    return hit[1]


FUSE_SHUFFLE_STORE = True    # narrow subpel tails (conv -> PixelShuffle(2), 3 output channels: the codecs' x_hat): shuffle + NCHW in the conv's store

# The tail conv(C -> 4C) -> PixelShuffle(2) -> conv(C -> 3) (upstream models/LiteISP.py:1996-2000) has no activation in between: it is ONE linear map,
# a 5x5 convolution C -> 12 whose channel 4o + 2i + j is colour o at sub-pixel (i, j) (rc_tail_fold_weights, composed once per checkpoint like weight
# packing).  4.6x fewer MACs and the 2H x 2W x C intermediate map never exists (4K x 8: 6.4 GB written + 7.7 GB read, 5.4 -> 0.9 ms).  The fold differs
# from the two convolutions on the outermost ring of output pixels only (zero padding of the SHUFFLED map); that ring is recomputed by the two original
# convolutions on four thin strips.  False: the two launches of the reference's module list.
FOLD_TAIL = True


def tail_fold_ok(x: torch.Tensor, conv1, conv2) -> bool:
    """conv(C -> 4C, 3x3) -> PixelShuffle(2) -> conv(C -> O, 3x3) with a 5x5 kernel instantiation for (C, 4 O, dtype): bf16 C = 48 k / 32 k, fp32 C = 16 k."""
    w1, w2 = conv1.weight, conv2.weight
    if not (FOLD_TAIL and x.dim() == 4 and x.dtype in _DT and x.shape[1] >= 2 and x.shape[2] >= 2 and w1.dim() == 4 and w2.dim() == 4):
        return False
    c = x.shape[-1]
    return (tuple(w1.shape) == (4 * c, c, 3, 3) and tuple(w2.shape[1:]) == (c, 3, 3) and 4 * w2.shape[0] <= 16 and
            _lib.load().rc_conv_packed_bytes(c, 4 * w2.shape[0], 5, _DT[x.dtype], RC_OUT_PIXEL_SHUFFLE2_NCHW) != 0)


def _folded_tail(conv1, conv2) -> "_ConvView":
    cache = _cache(conv2)
    key = (_key(conv1.weight, conv1.bias), _key(conv2.weight, conv2.bias))
    hit = cache.get("tail_fold")
    if hit is None or hit[0] != key:
        det = lambda t: None if t is None else t.detach()
        wc, bc = _R.tail_fold_weights(conv1.weight.detach(), det(conv1.bias), conv2.weight.detach(), det(conv2.bias))
        # the two side strips (H x 2 pixels) run TRANSPOSED (2 x H: a 2-pixel-wide image wastes 15/16 of every 8 x 32 tile): the same two convolutions
        # with ky <-> kx swapped and conv1's sub-pixel order 4c + 2i + j <-> 4c + 2j + i
        c = conv1.weight.shape[1]
        perm = (torch.arange(c, device=conv1.weight.device)[:, None] * 4 + torch.tensor([0, 2, 1, 3], device=conv1.weight.device)[None, :]).reshape(-1)
        v1t = _ConvView(conv1.weight.detach().transpose(2, 3)[perm].contiguous(), None if conv1.bias is None else conv1.bias.detach()[perm].contiguous())
        v2t = _ConvView(conv2.weight.detach().transpose(2, 3).contiguous(), det(conv2.bias))
        hit = cache["tail_fold"] = (key, _ConvView(wc, bc), v1t, v2t)
    return hit[1:]


def tail_fold(x: torch.Tensor, conv1, conv2, crop_hw: Optional[Tuple[int, int]] = None, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """conv2(PixelShuffle2(conv1(x))) of NHWC `x` (B,H,W,48) -> NCHW (B,O,2H,2W) cropped to crop_hw: one folded 5x5 launch + the exact border ring."""
    x = _req(x, "tail input")
    b, H, W, _ = x.shape
    odt = out_dtype or x.dtype

    folded, c1t, c2t = _folded_tail(conv1, conv2)

    def ring():      # six small launches, independent of the folded one until the scatter: they run beside it on the side stream
        rows, cols_t = _R.tail_ring_gather(x)              # (2B,2,W,C), and the side strips transposed: (2B,2,H,C)
        return [conv2d(conv2d(t, m1, out_mode=RC_OUT_PIXEL_SHUFFLE2), m2, out_mode=RC_OUT_NCHW, out_dtype=odt) for t, m1, m2 in ((rows, conv1, conv2), (cols_t, c1t, c2t))]

    strips, out = fork_join(ring, lambda: conv2d(x, folded, out_mode=RC_OUT_PIXEL_SHUFFLE2_NCHW, crop_hw=crop_hw, out_dtype=odt), [x])
    _R.tail_ring_scatter(out, strips[0], strips[1], H, W)
    return out


# conv -> act -> conv pairs (RCABlock.res, Res_GFM) as ONE launch with the intermediate in LDS (rc_conv_pair).
# Off by default: bit-identical to two launches and half their HBM traffic, but measured slower on MI355X in round 1
# (1.63 vs 1.51 ms per 4K x8 pair; gated 2.08 vs 2.03): the MFMA pipe sits at 53 % (phase 2 -- conv2 + stores + sums --
# takes 2.5x its MFMA time).  DESIGN.md section 4.3 has the phase breakdown.
FUSE_PAIR = False


def conv_pair_ok(x: torch.Tensor, m1, m2) -> bool:
    """rc_conv_pair is built for the flagship shape: two 3x3 48->48 convolutions on bf16 NHWC maps."""
    return (FUSE_PAIR and x.dtype == torch.bfloat16 and x.shape[-1] == 48 and
            all(m.weight.dim() == 4 and tuple(m.weight.shape) == (48, 48, 3, 3) for m in (m1, m2)))


def conv_pair(x: torch.Tensor, m1, m2, *, act: str = "relu", slope: float = 0.0,
              film: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
              gate: Optional[torch.Tensor] = None, skip: Optional[torch.Tensor] = None, store_input: bool = False,
              residual: Optional[torch.Tensor] = None, want_sums: bool = False):
    """out = conv(m2)(act(conv(m1)(x)))  (+ residual, + channel partial sums) in one launch; operands as conv2d.
    Returns out, or (out, [stored_input], [chan_sums])."""
    for m in (m1, m2):
        check_conv_module(m)
    x = _req(x, "conv_pair input")
    b, H, W, c = x.shape
    p1, p2 = packed_conv(m1, x.dtype, RC_OUT_NHWC), packed_conv(m2, x.dtype, RC_OUT_NHWC)
    if gate is not None:
        if skip is None:
            raise ValueError("gate needs skip")
        skip, gate = _req(skip, "skip"), _req(gate, "gate")
        if skip.shape != x.shape or skip.dtype != x.dtype or gate.shape != (b, c) or gate.dtype != torch.float32:
            raise ValueError("gate/skip shape or dtype mismatch")
    fs = ft = None
    if film is not None:
        fs, ft = (_req(t, "film") for t in film)
        if fs.shape != (b, c) or ft.shape != (b, c) or fs.dtype != torch.float32 or ft.dtype != torch.float32:
            raise ValueError("film tensors must be fp32 (B, C)")
    if residual is not None:
        residual = _req(residual, "residual")
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise ValueError("residual must match the input tensor")
    out, stored, sums = _R.conv_pair(x, p1.wpacked, p1.bias, p2.wpacked, p2.bias, _ACT[act], float(slope), fs, ft, gate, skip, bool(store_input),
                                     residual, bool(want_sums))
    extras = [t for t in (_opt(stored), _opt(sums)) if t is not None]
    return (out, *extras) if extras else out


FUSE_CHAIN = True   # Lens_Shading_Correction's four 1x1 convolutions as one launch (rc_pointwise_chain48)


def pointwise_chain_ok(x: torch.Tensor, convs, slopes) -> bool:
    if not (FUSE_CHAIN and x.dtype == torch.bfloat16 and 2 <= len(convs) <= 5 and x.shape[-1] <= 8):
        return False
    ok = all(tuple(m.weight.shape[2:]) == (1, 1) and m.weight.shape[0] == 48 for m in convs)
    ok = ok and convs[0].weight.shape[1] == x.shape[-1] and all(m.weight.shape[1] == 48 for m in convs[1:])
    return ok and len(set(slopes)) == 1 and 0.0 <= slopes[0] <= 1.0 and len(slopes) == len(convs) - 1


def pointwise_chain(x: torch.Tensor, convs, slope: float) -> torch.Tensor:
    """convs[0] (cin0 -> 48), LeakyReLU(slope), convs[1:] (48 -> 48) with LeakyReLU between, none after the last."""
    x = _req(x, "chain input")
    p0 = packed_conv(convs[0], x.dtype, RC_OUT_NHWC)
    packs = [packed_conv(m, x.dtype, RC_OUT_NHWC) for m in convs[1:]]
    return _R.pointwise_chain48(x, p0.wpacked, p0.bias, [p.wpacked for p in packs], [p.bias for p in packs], float(slope))


GRAPH_FORK = True            # keep the two-stream forks as graph branches under HIP-graph capture (False: one stream inside a capture)
BRANCH_STREAMS = True        # run independent small sub-graphs (the codec's mean / scale branches, the ISP nets' colour prior) on two HIP streams
_SIDE_STREAMS = {}


FORK_DEPTH = 1       # nesting depth of fork_join: 1 = one fork at a time (the slice loop's mean || scale); 3 also runs conv_a || conv_b inside each attention
                     # block on streams of their own -- measured +0.5 % at 8 frames, -3 % at 1 frame (host-bound): profiles/r05_small_things.md
_FORK_PATH = __import__("threading").local()     # where in a tree of nested fork_join calls the running code is: "" / "s" / "m" / "sm" ...


def fork_join(side_fn, main_fn, inputs):
    """(side_fn(), main_fn()) for two independent sub-graphs.  At the latent's size (1/16 of the packed frame) one launch covers about
    half of the 256 CUs, so the two branches are enqueued on two streams and overlap: side_fn on a side stream ordered behind the
    current stream, main_fn on the current stream, which then waits for the side stream.  `inputs`: tensors side_fn reads (the caching
    allocator is told about the second stream; the side branch's outputs likewise).  Calls NEST: a fork inside either branch of another gets a
    side stream of its own (one per position in the call tree), so the codec's slice loop runs mean || scale and, inside each, conv_a || conv_b
    as four concurrent chains of small launches.  One stream under fake tensors or BRANCH_STREAMS = False.
    Under HIP-graph capture the fork becomes two branches of the graph (the side stream joins the capture through wait_stream); the allocator's
    record_stream bookkeeping is skipped there: a capture's private pool is not reused across streams before the join, and every side tensor is
    consumed on the main stream only after it."""
    from torch._subclasses.fake_tensor import FakeTensor
    probe = inputs[0]
    if not BRANCH_STREAMS or not probe.is_cuda or isinstance(probe, FakeTensor):
        return side_fn(), main_fn()
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and not GRAPH_FORK:
        return side_fn(), main_fn()
    path = getattr(_FORK_PATH, "p", "")
    if len(path) >= FORK_DEPTH:                          # depth cap: 8 streams are plenty for 256 CUs (1 = round 4's single fork)
        return side_fn(), main_fn()
    main = torch.cuda.current_stream(probe.device)
    key = (probe.device.index, path)
    side = _SIDE_STREAMS.get(key)
    if side is None:
        side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=probe.device)
    side.wait_stream(main)
    if not capturing:
        for t in inputs:
            t.record_stream(side)
    try:
        _FORK_PATH.p = path + "s"
        with torch.cuda.stream(side):
            a = side_fn()
        _FORK_PATH.p = path + "m"
        b = main_fn()
    finally:
        _FORK_PATH.p = path
    main.wait_stream(side)
    if not capturing:
        for t in (a if isinstance(a, (tuple, list)) else (a,)):
            t.record_stream(main)
    return a, b



FUSE_MLP = True     # the codecs' transformer-block MLP (LayerNorm -> Linear -> GELU -> Linear -> + x) as one launch (rc_ln_mlp)


def ln_mlp(x: torch.Tensor, ln, fc1, fc2):
    """x + fc2(gelu(fc1(ln(x)))) in ONE launch (realcam::ln_mlp) for bf16 tokens of width 32 / 64 with a 4x hidden layer; None otherwise."""
    c = x.shape[-1]
    if not (FUSE_MLP and x.dtype == torch.bfloat16 and c in (32, 64) and tuple(fc1.weight.shape) == (4 * c, c) and tuple(fc2.weight.shape) == (c, 4 * c) and
            tuple(ln.normalized_shape) == (c,)):
        return None
    x = _req(x, "tokens")
    w1, b1 = packed_chain(fc1)
    w2, b2 = packed_chain(fc2)
    return _R.ln_mlp(x, f32_param(ln, "weight"), f32_param(ln, "bias"), float(ln.eps), w1, b1 if fc1.bias is not None else None, w2,
                     b2 if fc2.bias is not None else None)


PLANAR_QKV = os.environ.get("RC_PLANAR_QKV", "1") != "0"   # W-MSA: q / k / v written segment-planar by the embedding launch and read so by the attention launch (A/B, tests)


def planar_qkv_ok(x: torch.Tensor, window: int) -> bool:
    """Does rc_window_attention have its segment-planar form for this map (bf16, 8 x 8 windows, sizes inside 32-bit offsets)?"""
    b, h, w, c = x.shape
    return bool(PLANAR_QKV and x.dtype == torch.bfloat16 and _lib.load().rc_window_attention_planar8_ok(RC_BF16, b, h, w, c, window))


def ln_linear(x: torch.Tensor, ln, lin, planar8: bool = False):
    """lin(ln(x)) in ONE launch (realcam::ln_linear) for bf16 tokens of width 32 / 64 and an output width that is a multiple of 32; None otherwise.
    planar8: the result's MEMORY is [cout / 8 segments][tokens][8] (for torch.ops.realcam.window_attention_planar8 only; the shape stays (.., cout))."""
    c, cout = x.shape[-1], lin.weight.shape[0]
    if not (FUSE_MLP and x.dtype == torch.bfloat16 and c in (32, 64) and lin.weight.dim() == 2 and lin.weight.shape[1] == c and cout % 32 == 0 and
            cout <= 512 and tuple(ln.normalized_shape) == (c,)):
        return None
    w, b = packed_chain(lin)
    op = _R.ln_linear_planar8 if planar8 else _R.ln_linear
    return op(_req(x, "tokens"), f32_param(ln, "weight"), f32_param(ln, "bias"), float(ln.eps), w, b if lin.bias is not None else None, int(cout))


def cat_linear(a: torch.Tensor, b: torch.Tensor, conv, residual: Optional[torch.Tensor] = None, a_add: Optional[torch.Tensor] = None):
    """conv(torch.cat((a, b), channels)) + residual for a 1x1 conv, without writing the concatenated map (realcam::cat_linear): bf16, two
    equal halves of 32 / 64 channels, output width = input width.  None when the shapes are not of that form."""
    ca, cb = a.shape[-1], b.shape[-1]
    w = conv.weight
    if not (FUSE_MLP and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and ca == cb and ca + cb in (64, 128) and a.shape[:-1] == b.shape[:-1] and
            tuple(w.shape) == (ca + cb, ca + cb, 1, 1) and tuple(getattr(conv, "stride", (1, 1))) == (1, 1)):
        return None
    if residual is not None:
        residual = _req(residual, "residual")
        if residual.shape != (*a.shape[:-1], ca + cb) or residual.dtype != a.dtype:
            raise ValueError("cat_linear: residual shape / dtype mismatch")
    if a_add is not None:                                   # first half = a + a_add (rounded to bf16, as a separate add launch would store it)
        a_add = _req(a_add, "a_add")
        if a_add.shape != a.shape or a_add.dtype != a.dtype:
            raise ValueError("cat_linear: a_add shape / dtype mismatch")
    wp, bp = packed_chain(conv)
    return _R.cat_linear(_req(a, "a"), a_add, _req(b, "b"), residual, wp, bp if conv.bias is not None else None)


def lsc_chain(lsc, coord: torch.Tensor, head=None, raw: Optional[torch.Tensor] = None):
    """Lens_Shading_Correction as ONE launch with register-resident activations (realcam::lsc_chain), bf16, width 32 / 48 / 64 / 128:
    head is None -> lsc(coord);  else -> head(raw) * (lsc(coord) + 1)  (upstream models/LiteISP.py:2012-2014).  Returns None when the
    modules are not of that shape (the caller runs the layer-by-layer launches)."""
    import torch.nn as nn
    mods = list(lsc.model)
    convs, acts = mods[0::2], mods[1::2]
    if not (FUSE_CHAIN and coord.dtype == torch.bfloat16 and len(mods) == 2 * len(convs) - 1 and 2 <= len(convs) <= 5 and
            all(isinstance(m, nn.LeakyReLU) for m in acts) and all(hasattr(m, "weight") and tuple(m.weight.shape[2:]) == (1, 1) for m in convs)):
        return None
    slopes = {float(m.negative_slope) for m in acts}
    c = convs[0].weight.shape[0]
    if not (len(slopes) == 1 and 0.0 <= min(slopes) <= 1.0 and c in (32, 48, 64, 128) and convs[0].weight.shape[1] == coord.shape[-1] <= 4 and
            all(tuple(m.weight.shape[:2]) == (c, c) for m in convs[1:])):
        return None
    if head is not None:
        hw = head.weight
        if not (raw is not None and raw.dtype == torch.bfloat16 and tuple(hw.shape[2:]) == (3, 3) and hw.shape[0] == c and hw.shape[1] == raw.shape[-1] <= 4 and
                tuple(head.stride) == (1, 1) and tuple(head.padding) == (1, 1) and raw.shape[:3] == coord.shape[:3]):
            return None
        if c == 128:                   # no fused head at that width (rc_lsc_chain): the caller runs lsc and the head conv as two launches
            return None
        raw = _req(raw, "raw")
    coord = _req(coord, "coord")
    params = [p for m in convs for p in (m.weight, m.bias)] + ([head.weight, head.bias] if head is not None else [])
    cache = _cache(lsc)
    k = ("lsc", id(head))
    key = _key(*params)
    hit = cache.get(k)
    if hit is None or hit[0] != key:
        det = lambda t: None if t is None else t.detach()
        blob = _R.lsc_pack(det(convs[0].weight), det(convs[0].bias), [det(m.weight) for m in convs[1:]], [det(m.bias) for m in convs[1:]],
                           det(head.weight) if head is not None else None, det(head.bias) if head is not None else None)
        hit = cache[k] = (key, blob)
    return _R.lsc_chain(coord, hit[1], int(c), len(convs) - 1, slopes.pop(), raw if head is not None else None)


def ca_gate(sums: torch.Tensor, hw: int, ca) -> torch.Tensor:
    """CALayer gate (B,C) from the conv's channel partial sums.  models/networks.py:259-269."""
    c0, c1 = ca.conv_du[0], ca.conv_du[2]
    cr, c = c0.weight.shape[0], c0.weight.shape[1]
    return _R.ca_gate(sums, int(hw), f32_param(c0, "weight").reshape(cr, c), f32_param(c0, "bias"), f32_param(c1, "weight").reshape(c, cr),
                      f32_param(c1, "bias"))


# RCABlock as two plain launches: the CALayer gate of conv2's output is computed BEFORE conv2 runs (ca_gate_ahead: the mean of a convolution is
# linear in its input, so conv1's channel sums + t's four border lines give it), and conv2's epilogue writes x_new = conv2(t) * gate + x directly.
# Per block 5 map passes instead of 6 (the gated staging form read r and skip and wrote x and t); False: gate folded into the NEXT conv's staging.
EARLY_GATE = True


def gate_ahead_ok(conv1, conv2, ca) -> bool:
    """Is the closed form of ca_gate_ahead valid for this block?  It assumes conv2 = 3x3, C -> C, stride 1, zero padding 1, dilation 1, groups 1 fed by
    conv1's C channels, and a channel attention of the shape squeeze (1x1 conv / Linear) -> ReLU -> excite -> Sigmoid.  Anything else (another
    checkpoint-compatible conv2 configuration or activation) must take the schedule that reduces conv2's real output."""
    w = getattr(conv2, "weight", None)
    if w is None or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[0] != w.shape[1]:
        return False
    one = lambda v, k: tuple(v) == (k, k) if isinstance(v, (tuple, list)) else v == k
    if not (one(conv2.stride, 1) and one(conv2.padding, 1) and one(conv2.dilation, 1) and conv2.groups == 1 and getattr(conv2, "padding_mode", "zeros") == "zeros"):
        return False
    if getattr(conv1, "weight", None) is None or conv1.weight.shape[0] != w.shape[1]:
        return False
    seq = getattr(ca, "conv_du", None) or getattr(ca, "fc", None)
    if seq is None or len(seq) != 4 or not isinstance(seq[1], torch.nn.ReLU) or not isinstance(seq[3], torch.nn.Sigmoid):
        return False
    c0, c1 = seq[0], seq[2]
    ok_layer = lambda m: isinstance(m, torch.nn.Linear) or (isinstance(m, torch.nn.Conv2d) and tuple(m.weight.shape[2:]) == (1, 1) and m.groups == 1 and one(m.stride, 1))
    return ok_layer(c0) and ok_layer(c1) and c0.weight.shape[1] == w.shape[0] and c1.weight.shape[0] == w.shape[0] and c1.weight.shape[1] == c0.weight.shape[0]


def ca_gate_ahead(sums_t: torch.Tensor, t: torch.Tensor, conv2, ca) -> torch.Tensor:
    """CALayer gate (B,C) of conv2(t) from t's channel partial sums (emitted by the conv that produced t) and t itself (border lines only).
    `ca`: networks.CALayer (1x1 convs with bias, upstream models/networks.py:255-270) or raw2bit.CALayer (bias-free Linears, models/raw2bit.py:238-254)."""
    if hasattr(ca, "conv_du"):
        c0, c1 = ca.conv_du[0], ca.conv_du[2]
    else:
        c0, c1 = ca.fc[0], ca.fc[2]
    cr, c = c0.weight.shape[0], c0.weight.shape[1]
    if tuple(conv2.weight.shape) != (c, c, 3, 3):
        raise ValueError("ca_gate_ahead: conv2 must be a 3x3 convolution C -> C")
    z = _const(0.0, (max(c, cr),), sums_t)
    b0 = f32_param(c0, "bias") if c0.bias is not None else z
    b1 = f32_param(c1, "bias") if c1.bias is not None else z
    w2t, = host_cached(conv2, "w2_cin_tap_cout", [conv2.weight], lambda w: w.permute(1, 2, 3, 0))     # (C_in, 3, 3, C_out): coalesced over C_out
    gate, _ = _R.ca_gate_ahead(sums_t, _req(t, "t"), w2t, f32_param(conv2, "bias") if conv2.bias is not None else None,
                               f32_param(c0, "weight").reshape(cr, c), b0, f32_param(c1, "weight").reshape(c, cr), b1)
    return gate


def gate_residual(r: torch.Tensor, gate: torch.Tensor, x: Optional[torch.Tensor]) -> torch.Tensor:
    """r * gate[b, c] + x on NHWC maps; x None: r * gate (CALayer alone, models/networks.py:270)."""
    r, gate = _req(r, "r"), _req(gate, "gate")
    b, H, W, c = r.shape
    if x is not None:
        x = _req(x, "x")
        if x.shape != r.shape or x.dtype != r.dtype:
            raise ValueError("gate_residual: shape / dtype mismatch")
    if gate.shape != (b, c) or gate.dtype != torch.float32:
        raise ValueError("gate_residual: gate must be fp32 (B, C)")
    return _R.gate_residual(r, gate, x)


def film_apply(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """x*scale[b,c] + shift[b,c] + x on an NHWC map (GFMLayer, upstream models/LiteISP.py:317-320)."""
    x, scale, shift = _req(x, "x"), _req(scale, "scale"), _req(shift, "shift")
    b, c = x.shape[0], x.shape[-1]
    if scale.shape != (b, c) or shift.shape != (b, c) or scale.dtype != torch.float32 or shift.dtype != torch.float32:
        raise ValueError("film_apply: scale / shift must be fp32 (B, C)")
    return _R.film_apply(x, scale, shift)


def channel_sums(x: torch.Tensor) -> torch.Tensor:
    """Per-channel partial sums (B, slots, C) fp32 of an NHWC map, in the layout rc_ca_gate folds (AdaptiveAvgPool2d(1) of a
    CALayer that is not fed by a conv emitting them, models/networks.py:268)."""
    return _R.channel_sums(_req(x, "channel_sums input"))


def sigmoid_gate_add(a: torch.Tensor, b: torch.Tensor, identity: torch.Tensor) -> torch.Tensor:
    """a * sigmoid(b) + identity, element-wise (SWAtten / AttentionBlock, upstream models/tcm.py:287-288)."""
    a, b, identity = _req(a, "a"), _req(b, "b"), _req(identity, "identity")
    if a.shape != b.shape or a.shape != identity.shape or a.dtype != b.dtype or a.dtype != identity.dtype:
        raise ValueError("sigmoid_gate_add: shape / dtype mismatch")
    return _R.sigmoid_gate_add(a, b, identity)


def channel_slice(x: torch.Tensor, c0: int, n: int) -> torch.Tensor:
    """x[..., c0:c0+n] of an NHWC tensor as a new dense tensor (torch.split along channels, models/tcm.py:261)."""
    return _R.channel_slice(_req(x, "channel_slice input"), int(c0), int(n))


def channel_concat(parts) -> torch.Tensor:
    """torch.cat along the channel dim of NHWC tensors (models/tcm.py:265)."""
    parts = [_req(t, "channel_concat input") for t in parts]
    for t in parts:
        if t.shape[:-1] != parts[0].shape[:-1] or t.dtype != parts[0].dtype:
            raise ValueError("channel_concat: shape / dtype mismatch")
    return _R.channel_concat(parts)


_ONES = {}


def _const(value: float, shape, like: torch.Tensor) -> torch.Tensor:
    """A cached constant fp32 device tensor (never cached for fake tensors: they belong to their FakeTensorMode)."""
    from torch._subclasses.fake_tensor import FakeTensor
    if isinstance(like, FakeTensor):
        return torch.full(shape, value, dtype=torch.float32, device=like.device)
    key = (value, tuple(shape), str(like.device))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.full(shape, value, dtype=torch.float32, device=like.device)
    return t


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a + b on NHWC feature maps (the ISPUNet decoder skips, upstream LiteISP.py:1365-1373): rc_gate_residual with a
    unit gate."""
    a, b = _req(a, "a"), _req(b, "b")
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("add: shape / dtype mismatch")
    return gate_residual(a, _const(1.0, (a.shape[0], a.shape[-1]), a), b)




def dwt_forward(x: torch.Tensor, mod) -> torch.Tensor:
    x = _req(x, "dwt input")
    b, H, W, c = x.shape
    if H % 2 or W % 2:
        raise ValueError(f"DWTForward needs even H,W; got {H}x{W}")
    if mod.weight.shape[0] != 4 * c:
        raise ValueError("DWTForward channel mismatch")
    return _R.haar_dwt(x, f32_param(mod, "weight"), True)


def dwt_inverse(x: torch.Tensor, mod) -> torch.Tensor:
    x = _req(x, "idwt input")
    if mod.weight.shape[0] != x.shape[3]:
        raise ValueError("DWTInverse channel mismatch")
    return _R.haar_idwt(x, f32_param(mod, "weight"), True)


# --------------------------------------------------------------------------------------------------
# conditioning
# --------------------------------------------------------------------------------------------------
def color_block(x: torch.Tensor, conv, prev_norm=None, prev_stats=None) -> torch.Tensor:
    """Conv1x1 -> AvgPool(3,2,1) -> LeakyReLU(0.2) on NCHW x; applies the previous block's InstanceNorm
    (prev_norm affine params, prev_stats=(mean,rstd)) on load.  models/LiteISP.py:23-30."""
    x = _req(x, "color_block input")
    _dt(x)
    cout, cin = conv.weight.shape[0], conv.weight.shape[1]
    w, bias = f32_param(conv, "weight").reshape(cout, cin), f32_param(conv, "bias")
    if prev_norm is not None:
        mean, rstd = prev_stats
        return _R.color_block(x, w, bias, mean, rstd, f32_param(prev_norm, "weight"), f32_param(prev_norm, "bias"))
    return _R.color_block(x, w, bias, None, None, None, None)


def instance_stats(x: torch.Tensor, eps: float = 1e-5):
    return _R.instance_stats(x, float(eps))


def instance_norm(x: torch.Tensor, norm) -> torch.Tensor:
    """nn.InstanceNorm2d(affine=True) on an fp32 NCHW map (standalone CB block, upstream models/LiteISP.py:226-229)."""
    x = _req(x, "instance_norm input")
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("instance_norm: fp32 NCHW expected")
    mean, rstd = _R.instance_stats(x, float(norm.eps))
    return _R.instance_norm(x, mean, rstd, f32_param(norm, "weight"), f32_param(norm, "bias"))


def color_head(x: torch.Tensor, conv) -> torch.Tensor:
    cout, cin = conv.weight.shape[0], conv.weight.shape[1]
    return _R.color_head(x, f32_param(conv, "weight").reshape(cout, cin), f32_param(conv, "bias"))


def gfm_vector(vec: torch.Tensor, lin0, lin1) -> torch.Tensor:
    """lin1(leaky_relu(lin0(vec), 0.1)).  models/LiteISP.py:554-555."""
    vec = _req(vec, "gfm vector")
    if vec.dtype != torch.float32:
        vec = vec.float()
    return _R.gfm_vector(vec, f32_param(lin0, "weight"), f32_param(lin0, "bias"), f32_param(lin1, "weight"), f32_param(lin1, "bias"))


# --------------------------------------------------------------------------------------------------
# measurement hooks
# --------------------------------------------------------------------------------------------------
def prof_enable(on: bool) -> None:
    check(lib().rc_prof_enable(1 if on else 0), "rc_prof_enable")


class ProfRow(C.Structure):
    """Mirror of `struct rc_prof_row` (include/realcam_hip.h)."""
    _fields_ = [("cin", C.c_int), ("cout", C.c_int), ("ksize", C.c_int), ("launches", C.c_int), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


def prof_rows(max_rows: int = 64):
    """The profiled conv launches grouped by layer shape: [{cin, cout, ksize, launches, ms, flops, bytes}], most time first (rc_prof_collect_rows)."""
    rows = (ProfRow * max_rows)()
    n = C.c_int(0)
    check(lib().rc_prof_collect_rows(C.cast(rows, C.c_void_p), max_rows, C.byref(n)), "rc_prof_collect_rows")
    out = [{f: getattr(rows[i], f) for f, _ in ProfRow._fields_} for i in range(n.value)]
    return sorted(out, key=lambda r: -r["ms"])


def prof_collect():
    n, ms, fl = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
    check(lib().rc_prof_collect(C.byref(n), C.byref(ms), C.byref(fl)), "rc_prof_collect")
    return n.value, ms.value, fl.value


# --------------------------------------------------------------------------------------------------
# GroupMix attention pieces (upstream models/groupmix.py)
# --------------------------------------------------------------------------------------------------
def host_cached(mod, key: str, params, build):
    """Small derived fp32 tensors (folded BatchNorm, tap-major depth-wise weights ...) built on the host from
    module parameters once per parameter version and kept on the parameters' device."""
    c = _cache(mod)
    k = ("derived", key)
    ver = _key(*params)
    hit = c.get(k)
    if hit is None or hit[0] != ver:
        dev = params[0].device
        with torch.no_grad():
            vals = build(*[p.detach().float().cpu() for p in params])
        if isinstance(vals, torch.Tensor):
            vals = (vals,)
        hit = (ver, tuple(v.contiguous().to(dev) for v in vals))
        c[k] = hit
    return hit[1]


def dw_taps(weight: torch.Tensor, pad_to: Optional[int] = None) -> torch.Tensor:
    """(n,1,k,k) depth-wise weights -> tap-major (K*K, n) fp32, optionally zero-padded to a centred KxK."""
    n, _, k, _ = weight.shape
    if pad_to is not None and pad_to != k:
        r = (pad_to - k) // 2
        weight = torch.nn.functional.pad(weight, (r, r, r, r))
        k = pad_to
    return weight.reshape(n, k * k).t().contiguous()


def dwconv2d(x: torch.Tensor, x_c0: int, y_tail, y_c0: int, n_ch: int, ksize: int, wT: torch.Tensor,
             bias: Optional[torch.Tensor] = None, n_rep: int = 1, x_rep: int = 0, y_rep: int = 0, w_rep: int = 0,
             add_identity: bool = False, kvec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Depth-wise KxK conv over channel sub-ranges (rc_dwconv2d); allocates and returns y of shape (B,H,W,*y_tail), which the call
    must cover completely."""
    x = _req(x, "dwconv input")
    return _R.dwconv2d(x, int(x_c0), [int(v) for v in y_tail], int(y_c0), int(n_ch), int(ksize), wT, bias, int(n_rep), int(x_rep), int(y_rep),
                       int(w_rep), bool(add_identity), kvec)


def layernorm(x: torch.Tensor, norm) -> torch.Tensor:
    x = _req(x, "layernorm input")
    return _R.layernorm(x, f32_param(norm, "weight"), f32_param(norm, "bias"), float(norm.eps))
