"""PyTorch-ROCm custom ops `torch.ops.realcam.*` over the C ABI (include/realcam_hip.h).

SURVEY.md section 8(b): the drop-in boundary upstream sits behind is `nn.Module.forward`, so the extension exports its kernels
as dispatcher ops registered by schema.  Every op here is
    schema  : `realcam::<name>(...)` (TORCH_LIBRARY-style schema string, below)
    CUDA    : allocate the outputs through the caching allocator, enqueue ONE C-ABI launch (rc_<name>) on the current HIP
              stream, no host sync (HIP-graph capturable)
    Meta    : a fake-tensor kernel that only allocates the outputs, so FakeTensorMode / torch.compile tracing see shapes and
              dtypes without touching the library
There is no CPU kernel: a CPU tensor reaching one of these ops has no backend and raises (the wrappers in ops.py raise first,
with a message that names the oracle).  Layout: activations NHWC (B,H,W,C) contiguous, fp32 or bf16.  Absent optional outputs
are returned as empty (0,) tensors (an op's returns cannot be optional).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor

from . import _lib
from ._lib import RC_BF16, RC_F32, RC_OUT_NCHW, RC_OUT_NHWC, RC_OUT_NHWC_DWT, RC_OUT_PIXEL_SHUFFLE2, RC_OUT_PIXEL_SHUFFLE2_NCHW, ConvDesc, ConvPairDesc, check

_DT = {torch.float32: RC_F32, torch.bfloat16: RC_BF16}
_LIB = torch.library.Library("realcam", "DEF")
SCHEMAS = {}


def lib():
    return _lib.load()


def _dt(t: Tensor) -> int:
    return _DT[t.dtype]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _none(x: Tensor) -> Tensor:
    return x.new_empty((0,))


def define(schema: str, alloc, launch):
    """Register `realcam::<schema>`: alloc(*args) -> outputs (shared by the CUDA and the fake kernel), launch(outs, *args)
    enqueues the C-ABI call."""
    name = schema.split("(")[0]
    SCHEMAS[name] = schema
    _LIB.define(schema)

    def cuda_impl(*args):
        outs = alloc(*args)
        launch(outs, *args)
        return outs

    def fake_impl(*args):
        return alloc(*args)

    _LIB.impl(name, cuda_impl, "CUDA")
    torch.library.register_fake(f"realcam::{name}", fake_impl, lib=_LIB)


# ---- a1/a2: ingest and layout ---------------------------------------------------------------------------------------
def _bu_alloc(mosaic, dtype, pad_to):
    b, h2, w2 = mosaic.shape
    h, w = h2 // 2, w2 // 2
    return mosaic.new_empty((b, -(-h // pad_to) * pad_to, -(-w // pad_to) * pad_to, 4), dtype=dtype)


def _bu_launch(out, mosaic, dtype, pad_to):
    b, h2, w2 = mosaic.shape
    check(lib().rc_bayer_unshuffle(mosaic.data_ptr(), _dt(mosaic), out.data_ptr(), _DT[dtype], b, h2 // 2, w2 // 2, out.shape[1], out.shape[2],
                                   _stream()), "rc_bayer_unshuffle")


define("bayer_unshuffle(Tensor mosaic, ScalarType dtype, int pad_to) -> Tensor", _bu_alloc, _bu_launch)


def _ingest_alloc(mosaic, dtype, pad_to, black, white, cond_h, cond_w):
    b, h2, w2 = mosaic.shape
    h, w = h2 // 2, w2 // 2
    return (mosaic.new_empty((b, -(-h // pad_to) * pad_to, -(-w // pad_to) * pad_to, 4), dtype=dtype),
            mosaic.new_empty((b, 4, cond_h, cond_w), dtype=dtype))


def _ingest_launch(outs, mosaic, dtype, pad_to, black, white, cond_h, cond_w):
    packed, cond = outs
    b, h2, w2 = mosaic.shape
    in_dt = _lib.RC_U16 if mosaic.dtype == torch.uint16 else _dt(mosaic)
    check(lib().rc_raw_ingest(mosaic.data_ptr(), in_dt, packed.data_ptr(), cond.data_ptr(), _DT[dtype], b, h2 // 2, w2 // 2,
                              packed.shape[1], packed.shape[2], cond_h, cond_w, float(black), float(white), _stream()), "rc_raw_ingest")


define("raw_ingest(Tensor mosaic, ScalarType dtype, int pad_to, float black, float white, int cond_h, int cond_w) -> (Tensor, Tensor)",
       _ingest_alloc, _ingest_launch)


define("nchw_to_nhwc(Tensor x, ScalarType dtype, int hp, int wp) -> Tensor",
       lambda x, dtype, hp, wp: x.new_empty((x.shape[0], hp, wp, x.shape[1]), dtype=dtype),
       lambda out, x, dtype, hp, wp: check(lib().rc_nchw_to_nhwc(x.data_ptr(), _dt(x), out.data_ptr(), _DT[dtype], x.shape[0], x.shape[1],
                                                                 x.shape[2], x.shape[3], hp, wp, _stream()), "rc_nchw_to_nhwc"))

define("nhwc_to_nchw(Tensor a, ScalarType dtype, int h, int w) -> Tensor",
       lambda a, dtype, h, w: a.new_empty((a.shape[0], a.shape[3], h, w), dtype=dtype),
       lambda out, a, dtype, h, w: check(lib().rc_nhwc_to_nchw(a.data_ptr(), _dt(a), out.data_ptr(), _DT[dtype], a.shape[0], a.shape[3],
                                                               a.shape[1], a.shape[2], h, w, _stream()), "rc_nhwc_to_nchw"))


# ---- a3/a4: convolution -------------------------------------------------------------------------------------------------
def _wshape(weight):
    if weight.dim() == 2:
        return weight.shape[0], weight.shape[1], 1
    return weight.shape[0], weight.shape[1], weight.shape[2]


def _pack_alloc(weight, bias, act_dtype, out_mode, cout_tile=0):
    cout, cin, k = _wshape(weight)
    L = lib()
    dt = _DT[act_dtype]
    nbytes = L.rc_conv_packed_bytes_ct(cin, cout, k, dt, out_mode, cout_tile)
    if nbytes == 0:
        raise _lib.HipError(f"rc_conv_packed_bytes: {L.rc_last_error().decode()}")
    n_packed = L.rc_conv_packed_cout_ct(cin, cout, k, dt, out_mode, cout_tile)
    return (weight.new_empty((nbytes,), dtype=torch.uint8),
            weight.new_empty((n_packed if bias is not None else 0,), dtype=torch.float32))


def _pack_launch(outs, weight, bias, act_dtype, out_mode, cout_tile=0):
    # one-time host-side re-ordering into MFMA fragment order (rc_conv_pack_weights is a host function)
    wp, bp = outs
    cout, cin, k = _wshape(weight)
    L = lib()
    dt = _DT[act_dtype]
    w_host = np.ascontiguousarray(weight.detach().float().cpu().numpy())
    dst = np.empty(wp.numel(), dtype=np.uint8)
    check(L.rc_conv_pack_weights_ct(w_host.ctypes.data, cin, cout, k, dt, out_mode, cout_tile, dst.ctypes.data), "rc_conv_pack_weights")
    wp.copy_(torch.from_numpy(dst))
    if bias is not None:
        b_host = np.ascontiguousarray(bias.detach().float().cpu().numpy())
        bdst = np.zeros(bp.numel(), dtype=np.float32)
        check(L.rc_conv_pack_bias_ct(b_host.ctypes.data, cin, cout, k, dt, out_mode, cout_tile, bdst.ctypes.data), "rc_conv_pack_bias")
        bp.copy_(torch.from_numpy(bdst))


# cout_tile: cout tile width in channels (rc_conv_desc.cout_tile), 0 = automatic; the conv launch must be given the same value
define("conv_pack_weights(Tensor weight, Tensor? bias, ScalarType act_dtype, int out_mode, int cout_tile=0) -> (Tensor, Tensor)", _pack_alloc, _pack_launch)


def _wino_pack_alloc(weight, act_dtype):
    cout, cin, k = _wshape(weight)
    nbytes = lib().rc_wino_packed_bytes(cin, cout, _DT[act_dtype]) if k == 3 else 0
    if nbytes == 0:
        raise _lib.HipError(f"rc_wino_packed_bytes: {lib().rc_last_error().decode() if k == 3 else 'a 3x3 form'}")
    return weight.new_empty((nbytes,), dtype=torch.uint8)


def _wino_pack_launch(out, weight, act_dtype):
    cout, cin, _ = _wshape(weight)
    w_host = np.ascontiguousarray(weight.detach().float().cpu().numpy())
    dst = np.empty(out.numel(), dtype=np.uint8)
    check(lib().rc_wino_pack_weights(w_host.ctypes.data, cin, cout, _DT[act_dtype], dst.ctypes.data), "rc_wino_pack_weights")
    out.copy_(torch.from_numpy(dst))


# Winograd F(2x2, 3x3) form of a 3x3 conv (rc_conv_desc.algo = 1): U = G g G^T in MFMA fragment order; bias / film vectors stay in natural order
define("wino_pack_weights(Tensor weight, ScalarType act_dtype) -> Tensor", _wino_pack_alloc, _wino_pack_launch)


def _conv_alloc(x, wpacked, bias, cout, ksize, act, slope, residual, mul_plus1, film_scale, film_shift, gate, skip, store_input, out_mode,
                want_sums, crop_h, crop_w, out_dtype, out_scale=None, cout_tile=0, algo=0):
    b, H, W, _ = x.shape
    if out_mode == RC_OUT_NHWC:
        out = x.new_empty((b, H, W, cout))
    elif out_mode == RC_OUT_PIXEL_SHUFFLE2:
        out = x.new_empty((b, 2 * H, 2 * W, cout // 4))
    elif out_mode == RC_OUT_NHWC_DWT:                                  # conv -> Haar DWT in one launch: (H / 2, W / 2, 4 cout)
        out = x.new_empty((b, H // 2, W // 2, 4 * cout))
    elif out_mode == RC_OUT_PIXEL_SHUFFLE2_NCHW:
        out = x.new_empty((b, cout // 4, crop_h if crop_h > 0 else 2 * H, crop_w if crop_w > 0 else 2 * W), dtype=out_dtype or x.dtype)
    else:
        out = x.new_empty((b, cout, crop_h if crop_h > 0 else H, crop_w if crop_w > 0 else W), dtype=out_dtype or x.dtype)
    stored = torch.empty_like(x) if (store_input and gate is not None) else _none(x)
    sums = x.new_empty((0,), dtype=torch.float32)
    if want_sums:
        # how many partial-sum slots per image will THIS launch fill?  Asked of the launcher itself (rc_conv_sum_slots: the kernels that carry their sums
        # across tiles write grid x waves slots, the others 4 per 8 x 32 tile); only the NULL-ness of the pointers matters for the question.
        d = ConvDesc()
        d.batch, d.height, d.width, d.cin, d.cout, d.ksize, d.dtype = b, H, W, x.shape[-1], cout, ksize, _DT[x.dtype]
        # Non-NULL stand-ins that carry the REAL tensors' 16-byte alignment: the launcher's branch (vector vs scalar staging, fast vs generic epilogue) depends
        # on it, and a slot count asked about perfectly aligned pointers would not match a launch on a tensor at an odd storage offset.
        from torch._subclasses.fake_tensor import FakeTensor

        def stand_in(t):
            return 4096 + (0 if (t is None or isinstance(t, FakeTensor) or not t.is_cuda) else t.data_ptr() % 16)
        d.in0, d.wpacked, d.out, d.chan_sums = stand_in(x), stand_in(wpacked), stand_in(out), 4096
        if gate is not None:
            d.in1, d.in_gate = stand_in(skip), stand_in(gate)
            if store_input:
                d.in_store = 4096                                      # (allocated below: the caching allocator's blocks are 512-byte aligned)
        for name, t in (("bias", bias), ("film_scale", film_scale), ("film_shift", film_shift), ("mul_plus1", mul_plus1), ("residual", residual), ("out_scale", out_scale)):
            if t is not None:
                setattr(d, name, stand_in(t))
        d.act, d.act_slope, d.out_mode, d.out_dtype = act, float(slope), out_mode, _DT[out.dtype]
        d.cout_tile = cout_tile
        d.algo = algo
        if out_mode in (RC_OUT_NCHW, RC_OUT_PIXEL_SHUFFLE2_NCHW):
            d.out_h, d.out_w = out.shape[2], out.shape[3]
        if x.is_cuda and not isinstance(x, FakeTensor):                # the launcher sizes its grid by the CURRENT device's CU count: ask on the tensor's device
            with torch.cuda.device(x.device):
                n = lib().rc_conv_sum_slots(C.byref(d))
        else:
            n = lib().rc_conv_sum_slots(C.byref(d))
        if n <= 0:                                                     # an invalid description: the launch reports it (with its own message); allocate the per-tile count
            n = lib().rc_conv_sum_tiles(H, W)
        sums = x.new_empty((b, n, cout), dtype=torch.float32)
    return out, stored, sums


def _conv_launch(outs, x, wpacked, bias, cout, ksize, act, slope, residual, mul_plus1, film_scale, film_shift, gate, skip, store_input,
                 out_mode, want_sums, crop_h, crop_w, out_dtype, out_scale=None, cout_tile=0, algo=0):
    out, stored, sums = outs
    b, H, W, cin = x.shape
    d = ConvDesc()
    d.batch, d.height, d.width, d.cin, d.cout, d.ksize, d.dtype = b, H, W, cin, cout, ksize, _dt(x)
    d.in0 = x.data_ptr()
    if gate is not None:
        d.in1, d.in_gate = skip.data_ptr(), gate.data_ptr()
        if stored.numel():
            d.in_store = stored.data_ptr()
    d.wpacked, d.bias = wpacked.data_ptr(), _p(bias)
    d.film_scale, d.film_shift = _p(film_scale), _p(film_shift)
    d.act, d.act_slope = act, float(slope)
    d.mul_plus1, d.residual, d.out_scale = _p(mul_plus1), _p(residual), _p(out_scale)
    d.out, d.out_mode = out.data_ptr(), out_mode
    d.out_dtype = _dt(out)
    d.cout_tile = cout_tile
    d.algo = algo
    if out_mode in (RC_OUT_NCHW, RC_OUT_PIXEL_SHUFFLE2_NCHW):
        d.out_h, d.out_w = out.shape[2], out.shape[3]
    if want_sums:
        d.chan_sums = sums.data_ptr()
        d.chan_sums_slots = sums.shape[1]
    check(lib().rc_conv2d(C.byref(d), _stream()), "rc_conv2d")


define("conv2d(Tensor x, Tensor wpacked, Tensor? bias, int cout, int ksize, int act, float slope, Tensor? residual, Tensor? mul_plus1, "
       "Tensor? film_scale, Tensor? film_shift, Tensor? gate, Tensor? skip, bool store_input, int out_mode, bool want_sums, "
       "int crop_h, int crop_w, ScalarType? out_dtype, Tensor? out_scale=None, int cout_tile=0, int algo=0) -> (Tensor, Tensor, Tensor)", _conv_alloc, _conv_launch)


def _conv_fold2_launch(out, x, wpacked, bias, cout, act, slope):
    b, H, W, c = x.shape
    d = ConvDesc()
    d.batch, d.height, d.width, d.cin, d.cout, d.ksize, d.dtype = b, (H + 1) // 2, (W + 1) // 2, 4 * c, cout, 2, _dt(x)
    d.src_h, d.src_w = H, W
    d.in0, d.wpacked, d.bias = x.data_ptr(), wpacked.data_ptr(), _p(bias)
    d.act, d.act_slope = act, float(slope)
    d.out, d.out_mode, d.out_dtype = out.data_ptr(), RC_OUT_NHWC, _dt(out)
    check(lib().rc_conv2d(C.byref(d), _stream()), "rc_conv2d")


# 3x3 stride-2 convolution straight from its input (rc_conv_desc.src_h / src_w): the 2x2-window kernel gathers the space-to-depth channels itself
define("conv2d_fold2(Tensor x, Tensor wpacked, Tensor? bias, int cout, int act, float slope) -> Tensor",
       lambda x, wpacked, bias, cout, act, slope: x.new_empty((x.shape[0], (x.shape[1] + 1) // 2, (x.shape[2] + 1) // 2, cout)), _conv_fold2_launch)



# ---- a11 folded: the tail as one 5x5 convolution + its border ring (rc_tail_fold_weights / rc_tail_ring_*) -----------------------------
def _tail_fold_launch(outs, w1, b1, w2, b2):
    wc, bc = outs
    c, o = w1.shape[1], w2.shape[0]
    h = [None if t is None else np.ascontiguousarray(t.detach().float().cpu().numpy()) for t in (w1, b1, w2, b2)]
    wch, bch = np.empty(tuple(wc.shape), dtype=np.float32), np.empty(tuple(bc.shape), dtype=np.float32)
    check(lib().rc_tail_fold_weights(h[0].ctypes.data, None if h[1] is None else h[1].ctypes.data, h[2].ctypes.data,
                                     None if h[3] is None else h[3].ctypes.data, c, o, wch.ctypes.data, bch.ctypes.data), "rc_tail_fold_weights")
    wc.copy_(torch.from_numpy(wch)); bc.copy_(torch.from_numpy(bch))


# one-time host-side composition (like weight packing): conv2(PixelShuffle(conv1(x))) = conv5x5(x; wc, bc), fp32 results of a double accumulation
define("tail_fold_weights(Tensor w1, Tensor? b1, Tensor w2, Tensor? b2) -> (Tensor, Tensor)",
       lambda w1, b1, w2, b2: (w1.new_empty((4 * w2.shape[0], w1.shape[1], 5, 5), dtype=torch.float32), w1.new_empty((4 * w2.shape[0],), dtype=torch.float32)),
       _tail_fold_launch)

define("tail_ring_gather(Tensor x) -> (Tensor, Tensor)",
       lambda x: (x.new_empty((2 * x.shape[0], 2, x.shape[2], x.shape[3])), x.new_empty((2 * x.shape[0], 2, x.shape[1], x.shape[3]))),
       lambda outs, x: check(lib().rc_tail_ring_gather(x.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), _dt(x), x.shape[0], x.shape[1], x.shape[2],
                                                       x.shape[3], _stream()), "rc_tail_ring_gather"))

define("tail_ring_scatter(Tensor(a!) out, Tensor rows_out, Tensor cols_out, int H, int W) -> Tensor",
       lambda out, rows_out, cols_out, H, W: out.new_empty((0,)),
       lambda ret, out, rows_out, cols_out, H, W: check(lib().rc_tail_ring_scatter(rows_out.data_ptr(), cols_out.data_ptr(), out.data_ptr(), _dt(out),
                                                                                   out.shape[0], out.shape[1], H, W, out.shape[2], out.shape[3],
                                                                                   _stream()), "rc_tail_ring_scatter"))


def _pair_alloc(x, w1, b1, w2, b2, act, slope, film_scale, film_shift, gate, skip, store_input, residual, want_sums):
    b, H, W, c = x.shape
    stored = torch.empty_like(x) if (store_input and gate is not None) else _none(x)
    sums = x.new_empty((b, lib().rc_conv_pair_sum_slots(H, W), c), dtype=torch.float32) if want_sums else x.new_empty((0,), dtype=torch.float32)
    return torch.empty_like(x), stored, sums


def _pair_launch(outs, x, w1, b1, w2, b2, act, slope, film_scale, film_shift, gate, skip, store_input, residual, want_sums):
    out, stored, sums = outs
    b, H, W, c = x.shape
    d = ConvPairDesc()
    d.batch, d.height, d.width, d.channels, d.dtype = b, H, W, c, _dt(x)
    d.in0 = x.data_ptr()
    if gate is not None:
        d.in1, d.in_gate = skip.data_ptr(), gate.data_ptr()
        if stored.numel():
            d.in_store = stored.data_ptr()
    d.w1, d.b1, d.w2, d.b2 = w1.data_ptr(), _p(b1), w2.data_ptr(), _p(b2)
    d.film_scale, d.film_shift = _p(film_scale), _p(film_shift)
    d.act1, d.act1_slope = act, float(slope)
    d.residual, d.out = _p(residual), out.data_ptr()
    if want_sums:
        d.chan_sums = sums.data_ptr()
    check(lib().rc_conv_pair(C.byref(d), _stream()), "rc_conv_pair")


define("conv_pair(Tensor x, Tensor w1, Tensor? b1, Tensor w2, Tensor? b2, int act, float slope, Tensor? film_scale, Tensor? film_shift, "
       "Tensor? gate, Tensor? skip, bool store_input, Tensor? residual, bool want_sums) -> (Tensor, Tensor, Tensor)", _pair_alloc, _pair_launch)


def _chain_launch(out, x, w0, b0, wmid, bmid, slope):
    n = len(wmid)
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in wmid])
    bp = (C.c_void_p * n)(*[_p(b) for b in bmid])
    check(lib().rc_pointwise_chain48(x.data_ptr(), x.shape[-1], w0.data_ptr(), _p(b0), wp, bp, n, float(slope), out.data_ptr(), _dt(x),
                                     x.numel() // x.shape[-1], _stream()), "rc_pointwise_chain48")


define("pointwise_chain48(Tensor x, Tensor w0, Tensor? b0, Tensor[] wmid, Tensor?[] bmid, float slope) -> Tensor",
       lambda x, w0, b0, wmid, bmid, slope: x.new_empty((*x.shape[:-1], 48)), _chain_launch)


def _lsc_pack_alloc(w0, b0, wmid, bmid, whead, bhead):
    n = lib().rc_lsc_packed_bytes(w0.shape[0], len(wmid), 1 if whead is not None else 0) if not _is_fake(w0) else \
        _lsc_bytes(w0.shape[0], len(wmid), whead is not None)
    if n == 0:
        raise ValueError("lsc_pack: width must be 32, 48, 64 or 128 with at least one layer after the first")
    return w0.new_empty((n,), dtype=torch.uint8)


def _lsc_bytes(c, n_mid, head):
    mt, tb = c // 16, (c // 32) * 1024 + (512 if c % 32 else 0)
    return mt * 512 + n_mid * mt * tb + (mt * 1536 if head else 0) + (1 + n_mid + (1 if head else 0)) * mt * 64


def _is_fake(t):
    from torch._subclasses.fake_tensor import FakeTensor
    return isinstance(t, FakeTensor)


def _lsc_pack_launch(out, w0, b0, wmid, bmid, whead, bhead):
    host = lambda t: None if t is None else np.ascontiguousarray(t.detach().float().cpu().numpy())
    c, cin0, n = w0.shape[0], w0.shape[1], len(wmid)
    h_w0, h_b0, h_wh, h_bh = host(w0.reshape(c, cin0)), host(b0), host(whead), host(bhead)
    h_wm, h_bm = [host(w.reshape(c, c)) for w in wmid], [host(b) for b in bmid]
    ptr = lambda a: None if a is None else a.ctypes.data
    wp = (C.c_void_p * n)(*[ptr(a) for a in h_wm])
    bp = (C.c_void_p * n)(*[ptr(a) for a in h_bm])
    dst = np.zeros(out.numel(), dtype=np.uint8)
    check(lib().rc_lsc_pack(ptr(h_w0), ptr(h_b0), cin0, wp, bp, n, ptr(h_wh), ptr(h_bh), whead.shape[1] if whead is not None else 0, c,
                            dst.ctypes.data), "rc_lsc_pack")
    out.copy_(torch.from_numpy(dst))


define("lsc_pack(Tensor w0, Tensor? b0, Tensor[] wmid, Tensor?[] bmid, Tensor? whead, Tensor? bhead) -> Tensor", _lsc_pack_alloc, _lsc_pack_launch)

define("lsc_chain(Tensor x, Tensor blob, int c, int n_mid, float slope, Tensor? raw) -> Tensor",
       lambda x, blob, c, n_mid, slope, raw: x.new_empty((*x.shape[:-1], c)),
       lambda out, x, blob, c, n_mid, slope, raw: check(lib().rc_lsc_chain(x.data_ptr(), x.shape[-1], blob.data_ptr(), c, n_mid, float(slope), _p(raw),
                                                                           raw.shape[-1] if raw is not None else 0, out.data_ptr(), x.shape[0],
                                                                           x.shape[1], x.shape[2], _stream()), "rc_lsc_chain"))


# ---- a8: channel attention ------------------------------------------------------------------------------------------------
define("ca_gate(Tensor(a!) sums, int hw, Tensor w0, Tensor b0, Tensor w1, Tensor b1) -> Tensor",
       lambda sums, hw, w0, b0, w1, b1: sums.new_empty((sums.shape[0], sums.shape[2])),
       lambda out, sums, hw, w0, b0, w1, b1: check(lib().rc_ca_gate(sums.data_ptr(), sums.shape[0], sums.shape[1], sums.shape[2], w0.shape[0],
                                                                    1.0 / float(hw), w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                                                    out.data_ptr(), _stream()), "rc_ca_gate"))



def _gate_ahead_alloc(sums, t, w2t, b2, w0, b0, w1, b1):
    b, c = sums.shape[0], sums.shape[2]
    return sums.new_empty((b, c)), sums.new_empty((b * (4 * 8 + 4) * c,))      # gate, scratch (rc_ca_gate_ahead_scratch_floats)


def _gate_ahead_launch(outs, sums, t, w2t, b2, w0, b0, w1, b1):
    gate, scratch = outs
    b, n_tiles, c = sums.shape
    assert scratch.numel() >= lib().rc_ca_gate_ahead_scratch_floats(b, c)
    check(lib().rc_ca_gate_ahead(sums.data_ptr(), b, n_tiles, c, w0.shape[0], t.data_ptr(), _dt(t), t.shape[1], t.shape[2], w2t.data_ptr(), _p(b2),
                                 w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), scratch.data_ptr(), gate.data_ptr(), _stream()), "rc_ca_gate_ahead")


# CALayer's gate of conv2(t) from t's channel sums + border lines, before conv2 runs (the mean of a convolution is linear in its input)
define("ca_gate_ahead(Tensor(a!) sums, Tensor t, Tensor w2t, Tensor? b2, Tensor w0, Tensor b0, Tensor w1, Tensor b1) -> (Tensor, Tensor)",
       _gate_ahead_alloc, _gate_ahead_launch)

define("gate_residual(Tensor r, Tensor gate, Tensor? x) -> Tensor",
       lambda r, gate, x: torch.empty_like(r),
       lambda out, r, gate, x: check(lib().rc_gate_residual(r.data_ptr(), gate.data_ptr(), _p(x), out.data_ptr(), _dt(r), r.shape[0],
                                                            r.shape[1] * r.shape[2], r.shape[3], _stream()), "rc_gate_residual"))

define("channel_sums(Tensor x) -> Tensor",
       lambda x: x.new_empty((x.shape[0], lib().rc_channel_sums_slots(x.shape[1] * x.shape[2]), x.shape[3]), dtype=torch.float32),
       lambda out, x: check(lib().rc_channel_sums(x.data_ptr(), _dt(x), x.shape[0], x.shape[1] * x.shape[2], x.shape[3], out.data_ptr(),
                                                  _stream()), "rc_channel_sums"))


# ---- element-wise / resampling pieces of the codec (rows a18-a20) ----------------------------------------------------------
def _ew(name, schema_args, call):
    define(f"{name}({schema_args}) -> Tensor", lambda x, *a: torch.empty_like(x), call)


_ew("sigmoid_gate_add", "Tensor a, Tensor b, Tensor identity",
    lambda out, a, b, i: check(lib().rc_sigmoid_gate_add(a.data_ptr(), b.data_ptr(), i.data_ptr(), out.data_ptr(), _dt(a), a.numel(), _stream()),
                               "rc_sigmoid_gate_add"))
_ew("sft_apply", "Tensor x, Tensor scale, Tensor shift, Tensor? identity",
    lambda out, x, s, t, i: check(lib().rc_sft_apply(x.data_ptr(), s.data_ptr(), t.data_ptr(), _p(i), out.data_ptr(), _dt(x), x.numel(), _stream()),
                                  "rc_sft_apply"))
_ew("square", "Tensor x", lambda out, x: check(lib().rc_square(x.data_ptr(), out.data_ptr(), _dt(x), x.numel(), _stream()), "rc_square"))
_ew("gdn_apply", "Tensor x, Tensor norm, bool inverse, Tensor? identity",
    lambda out, x, n, inv, i: check(lib().rc_gdn_apply(x.data_ptr(), n.data_ptr(), _p(i), out.data_ptr(), _dt(x), int(bool(inv)), x.numel(),
                                                       _stream()), "rc_gdn_apply"))
_ew("tanh_half_add", "Tensor a, Tensor lrp",
    lambda out, a, l: check(lib().rc_tanh_half_add(a.data_ptr(), l.data_ptr(), out.data_ptr(), _dt(a), a.numel(), _stream()), "rc_tanh_half_add"))


def _resample(name, shape_fn, fn_name):
    def launch(out, x):
        b, H, W, c = x.shape
        cc = c // 4 if name == "pixel_shuffle2" else c
        check(getattr(lib(), fn_name)(x.data_ptr(), out.data_ptr(), _dt(x), b, H, W, cc, _stream()), fn_name)
    define(f"{name}(Tensor x) -> Tensor", lambda x: x.new_empty(shape_fn(*x.shape)), launch)


_resample("subsample2", lambda b, H, W, c: (b, (H + 1) // 2, (W + 1) // 2, c), "rc_subsample2")
_resample("upsample_bilinear2", lambda b, H, W, c: (b, 2 * H, 2 * W, c), "rc_upsample_bilinear2")
_resample("space_to_depth2", lambda b, H, W, c: (b, (H + 1) // 2, (W + 1) // 2, 4 * c), "rc_space_to_depth2")
_resample("pixel_shuffle2", lambda b, H, W, c: (b, 2 * H, 2 * W, c // 4), "rc_pixel_shuffle2")

define("entropy_bottleneck(Tensor z, Tensor params, Tensor medians, float bound) -> (Tensor, Tensor)",
       lambda z, p, m, bound: (torch.empty_like(z), z.new_empty(z.shape, dtype=torch.float32)),
       lambda outs, z, p, m, bound: check(lib().rc_entropy_bottleneck(z.data_ptr(), p.data_ptr(), m.data_ptr(), outs[0].data_ptr(),
                                                                      outs[1].data_ptr(), _dt(z), z.numel() // z.shape[-1], z.shape[-1],
                                                                      float(bound), _stream()), "rc_entropy_bottleneck"))

define("gaussian_conditional(Tensor y, Tensor scale, Tensor mu, float scale_bound, float bound) -> (Tensor, Tensor)",
       lambda y, s, m, sb, bound: (torch.empty_like(y), y.new_empty(y.shape, dtype=torch.float32)),
       lambda outs, y, s, m, sb, bound: check(lib().rc_gaussian_conditional(y.data_ptr(), s.data_ptr(), m.data_ptr(), outs[0].data_ptr(),
                                                                            outs[1].data_ptr(), _dt(y), y.numel(), float(sb), float(bound),
                                                                            _stream()), "rc_gaussian_conditional"))

define("channel_slice(Tensor x, int c0, int n) -> Tensor",
       lambda x, c0, n: x.new_empty((*x.shape[:-1], n)),
       lambda out, x, c0, n: check(lib().rc_channel_copy(x.data_ptr(), x.shape[-1], c0, out.data_ptr(), n, 0, n, x.numel() // x.shape[-1], _dt(x),
                                                         _stream()), "rc_channel_copy"))


def _concat_launch(out, parts):
    unit = 16 // out.element_size()
    if len(parts) <= 8 and all(t.shape[-1] % unit == 0 for t in parts):       # one launch for all parts
        ptrs = (C.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
        widths = (C.c_int * len(parts))(*[t.shape[-1] for t in parts])
        check(lib().rc_channel_concat(ptrs, widths, len(parts), out.data_ptr(), out.numel() // out.shape[-1], _dt(out), _stream()), "rc_channel_concat")
        return
    ctot, c0 = out.shape[-1], 0
    for t in parts:
        check(lib().rc_channel_copy(t.data_ptr(), t.shape[-1], 0, out.data_ptr(), ctot, c0, t.shape[-1], t.numel() // t.shape[-1], _dt(t),
                                    _stream()), "rc_channel_copy")
        c0 += t.shape[-1]


define("channel_concat(Tensor[] parts) -> Tensor",
       lambda parts: parts[0].new_empty((*parts[0].shape[:-1], sum(t.shape[-1] for t in parts))), _concat_launch)


# ---- a10: Haar DWT / inverse (taps live in the state_dict) -------------------------------------------------------------------
_UNIFORM = {}


def _taps_uniform(taps: Tensor) -> int:
    """1 if every channel carries the same 4 x (2x2) taps (the reference's Haar init): checked once per tensor version on the
    host, in the CUDA kernel's launcher only (a data-dependent branch has no place in the fake kernel)."""
    key = (taps.data_ptr(), taps._version)
    hit = _UNIFORM.get(key)
    if hit is None:
        w = taps.detach().float().reshape(-1, 4, 4)
        hit = int(bool((w == w[:1]).all().item()))
        if len(_UNIFORM) > 256:
            _UNIFORM.clear()
        _UNIFORM[key] = hit
    return hit


define("haar_dwt(Tensor x, Tensor taps, bool check_uniform) -> Tensor",
       lambda x, taps, cu: x.new_empty((x.shape[0], x.shape[1] // 2, x.shape[2] // 2, 4 * x.shape[3])),
       lambda out, x, taps, cu: check(lib().rc_dwt_forward(x.data_ptr(), out.data_ptr(), taps.data_ptr(), _taps_uniform(taps) if cu else 1, _dt(x),
                                                           x.shape[0], x.shape[1], x.shape[2], x.shape[3], _stream()), "rc_dwt_forward"))

define("haar_idwt(Tensor x, Tensor taps, bool check_uniform) -> Tensor",
       lambda x, taps, cu: x.new_empty((x.shape[0], 2 * x.shape[1], 2 * x.shape[2], x.shape[3] // 4)),
       lambda out, x, taps, cu: check(lib().rc_dwt_inverse(x.data_ptr(), out.data_ptr(), taps.data_ptr(), _taps_uniform(taps) if cu else 1, _dt(x),
                                                           x.shape[0], x.shape[1], x.shape[2], x.shape[3], _stream()), "rc_dwt_inverse"))


# ---- a5-a7: conditioning ------------------------------------------------------------------------------------------------------
def _cb_launch(out, x, w, b, mean, rstd, gamma, beta):
    bt, cin, h, wd = x.shape
    check(lib().rc_color_block(x.data_ptr(), _dt(x), out.data_ptr(), bt, cin, w.shape[0], h, wd, w.data_ptr(), b.data_ptr(), _p(mean), _p(rstd),
                               _p(gamma), _p(beta), _stream()), "rc_color_block")


define("color_block(Tensor x, Tensor w, Tensor b, Tensor? mean, Tensor? rstd, Tensor? gamma, Tensor? beta) -> Tensor",
       lambda x, w, b, mean, rstd, gamma, beta: x.new_empty((x.shape[0], w.shape[0], (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1),
                                                            dtype=torch.float32), _cb_launch)

define("instance_stats(Tensor x, float eps) -> (Tensor, Tensor)",
       lambda x, eps: (x.new_empty((x.shape[0], x.shape[1])), x.new_empty((x.shape[0], x.shape[1]))),
       lambda outs, x, eps: check(lib().rc_instance_stats(x.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), x.shape[0], x.shape[1],
                                                          x.shape[2] * x.shape[3], float(eps), _stream()), "rc_instance_stats"))

define("color_head(Tensor x, Tensor w, Tensor b) -> Tensor",
       lambda x, w, b: x.new_empty((x.shape[0], w.shape[0])),
       lambda out, x, w, b: check(lib().rc_color_head(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], w.shape[0], x.shape[2] * x.shape[3],
                                                      w.data_ptr(), b.data_ptr(), _stream()), "rc_color_head"))

define("gfm_vector(Tensor vec, Tensor w0, Tensor b0, Tensor w1, Tensor b1) -> Tensor",
       lambda vec, w0, b0, w1, b1: vec.new_empty((vec.shape[0], w1.shape[0])),
       lambda out, vec, w0, b0, w1, b1: check(lib().rc_gfm_vector(vec.data_ptr(), vec.shape[0], vec.shape[1], w0.shape[0], w1.shape[0],
                                                                  w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), out.data_ptr(),
                                                                  _stream()), "rc_gfm_vector"))


# ---- a14-a16: GroupMix pieces --------------------------------------------------------------------------------------------------
def _dw_launch(y, x, x_c0, y_tail, y_c0, n_ch, ksize, wT, bias, n_rep, x_rep, y_rep, w_rep, add_identity, kvec):
    b, H, W = x.shape[:3]
    xs = int(np.prod(x.shape[3:]))
    ys = int(np.prod(y.shape[3:]))
    check(lib().rc_dwconv2d(x.data_ptr(), xs, x_c0, y.data_ptr(), ys, y_c0, _dt(x), b, H, W, n_ch, ksize, wT.data_ptr(), wT.shape[1], _p(bias),
                            n_rep, x_rep, y_rep, w_rep, 1 if add_identity else 0, _p(kvec), _stream()), "rc_dwconv2d")


define("dwconv2d(Tensor x, int x_c0, int[] y_tail, int y_c0, int n_ch, int ksize, Tensor wT, Tensor? bias, int n_rep, int x_rep, int y_rep, "
       "int w_rep, bool add_identity, Tensor? kvec) -> Tensor",
       lambda x, x_c0, y_tail, *a: x.new_empty((*x.shape[:3], *y_tail)), _dw_launch)

define("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps) -> Tensor",
       lambda x, g, b, eps: torch.empty_like(x),
       lambda out, x, g, b, eps: check(lib().rc_layernorm(x.data_ptr(), out.data_ptr(), _dt(x), x.numel() // x.shape[-1], x.shape[-1], g.data_ptr(),
                                                          b.data_ptr(), float(eps), _stream()), "rc_layernorm"))


def _gpw_launch(outs, qkv, dwc, pw, scale, shift, pwl, ln_g, ln_b):
    qkvp, loc = outs
    b, H, W, c3 = qkv.shape
    c, seg = c3 // 3, c3 // 15
    es = qkv.element_size()
    check(lib().rc_gma_pointwise(qkv.data_ptr(), dwc.data_ptr(), dwc.data_ptr() + 3 * seg * es, 12 * seg, 4 * seg, 12 * seg, 4 * seg,
                                 qkvp.data_ptr(), loc.data_ptr(), _dt(qkv), b * H * W, c, pw.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                 pwl.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), _stream()), "rc_gma_pointwise")


define("gma_pointwise(Tensor qkv, Tensor dwc, Tensor pw, Tensor bn_scale, Tensor bn_shift, Tensor pwl, Tensor ln_g, Tensor ln_b) -> (Tensor, Tensor)",
       lambda qkv, dwc, *a: (qkv.new_empty((*qkv.shape[:3], 3, 4 * (qkv.shape[3] // 15))), qkv.new_empty((*qkv.shape[:3], qkv.shape[3] // 15))),
       _gpw_launch)


def _planar(qkvp) -> bool:
    """segment-planar (12, B, H, W, 16) from realcam::gma_aggregate, as opposed to token-major (B, H, W, 3, ct)"""
    return qkvp.shape[-1] == 16 and qkvp.shape[0] == 12 and qkvp.dim() == 5


def _gkv_launch(ktv, qkvp, heads, ch, scale):
    b, H, W = qkvp.shape[1:4] if _planar(qkvp) else qkvp.shape[:3]
    n = H * W
    scratch = torch.empty(lib().rc_gma_kv_scratch_bytes(b, n, heads, ch) // 4, dtype=torch.float32, device=qkvp.device)
    if _planar(qkvp):
        check(lib().rc_gma_kv_planar(qkvp.data_ptr(), b, n, heads, ch, float(scale), scratch.data_ptr(), ktv.data_ptr(), _stream()), "rc_gma_kv_planar")
    else:
        check(lib().rc_gma_kv(qkvp.data_ptr(), _dt(qkvp), b, n, heads, ch, float(scale), scratch.data_ptr(), ktv.data_ptr(), _stream()), "rc_gma_kv")


define("gma_kv(Tensor qkvp, int heads, int ch, float scale) -> Tensor",
       lambda qkvp, heads, ch, scale: qkvp.new_empty((qkvp.shape[1] if _planar(qkvp) else qkvp.shape[0], heads, ch, ch), dtype=torch.float32),
       _gkv_launch)

define("gma_apply(Tensor qkvp, Tensor convv, Tensor loc, Tensor ktv, int heads, int ch, int seg) -> Tensor",
       lambda qkvp, convv, loc, ktv, heads, ch, seg: qkvp.new_empty((*qkvp.shape[:3], heads * ch + seg)),
       lambda out, qkvp, convv, loc, ktv, heads, ch, seg: check(
           lib().rc_gma_apply(qkvp.data_ptr(), convv.data_ptr(), loc.data_ptr(), ktv.data_ptr(), out.data_ptr(), _dt(qkvp), qkvp.shape[0],
                              qkvp.shape[1] * qkvp.shape[2], heads, ch, seg, _stream()), "rc_gma_apply"))


# ---- a17: window attention ------------------------------------------------------------------------------------------------------
define("window_attention(Tensor qkv, Tensor rel_pos, int head_dim, int window, int shift) -> Tensor",
       lambda qkv, rel, hd, ws, sh: qkv.new_empty((*qkv.shape[:3], qkv.shape[3] // 3)),
       lambda out, qkv, rel, hd, ws, sh: check(lib().rc_window_attention(qkv.data_ptr(), rel.data_ptr(), out.data_ptr(), _dt(qkv), qkv.shape[0],
                                                                         qkv.shape[1], qkv.shape[2], qkv.shape[3] // 3, hd, ws, sh, _stream()),
                                               "rc_window_attention"))
# qkv SEGMENT-PLANAR ([3C / 8][B H W][8], the memory ln_linear_planar8 fills; the tensor keeps the (B,H,W,3C) shape as a size carrier only)
define("window_attention_planar8(Tensor qkv, Tensor rel_pos, int head_dim, int window, int shift) -> Tensor",
       lambda qkv, rel, hd, ws, sh: qkv.new_empty((*qkv.shape[:3], qkv.shape[3] // 3)),
       lambda out, qkv, rel, hd, ws, sh: check(lib().rc_window_attention_planar8(qkv.data_ptr(), rel.data_ptr(), out.data_ptr(), _dt(qkv), qkv.shape[0],
                                                                                 qkv.shape[1], qkv.shape[2], qkv.shape[3] // 3, hd, ws, sh, _stream()),
                                               "rc_window_attention_planar8"))


# ---- a15/a16 fused per-token stages of GMA_Block at dim 80 (csrc/gma_fused.hip) ---------------------------------------------------
def _cpack_alloc(weight, bias):
    cout, cin = weight.shape[0], weight.shape[1]
    nbytes = lib().rc_chain_packed_bytes(cin, cout)
    if nbytes == 0:
        raise _lib.HipError("rc_chain_packed_bytes: cin must be a multiple of 16")
    return weight.new_empty((nbytes,), dtype=torch.uint8), weight.new_empty((lib().rc_chain_packed_rows(cout),), dtype=torch.float32)


def _cpack_launch(outs, weight, bias):
    wp, bp = outs
    cout, cin = weight.shape[0], weight.shape[1]
    w_host = np.ascontiguousarray(weight.detach().float().reshape(cout, cin).cpu().numpy())
    dst = np.empty(wp.numel(), dtype=np.uint8)
    check(lib().rc_chain_pack_weights(w_host.ctypes.data, cin, cout, dst.ctypes.data), "rc_chain_pack_weights")
    wp.copy_(torch.from_numpy(dst))
    bdst = np.zeros(bp.numel(), dtype=np.float32)
    b_host = np.ascontiguousarray(bias.detach().float().cpu().numpy()) if bias is not None else None
    check(lib().rc_chain_pack_bias(b_host.ctypes.data if b_host is not None else None, cout, bdst.ctypes.data), "rc_chain_pack_bias")
    bp.copy_(torch.from_numpy(bdst))


define("chain_pack_weights(Tensor weight, Tensor? bias) -> (Tensor, Tensor)", _cpack_alloc, _cpack_launch)

define("gma_ln_qkv(Tensor x, Tensor wpacked, Tensor bias_packed, Tensor ln_gamma, Tensor ln_beta, float eps) -> Tensor",
       lambda x, wp, bp, g, b, eps: x.new_empty((15, *x.shape[:-1], 16)),         # planar by 16-channel segment
       lambda out, x, wp, bp, g, b, eps: check(lib().rc_gma_ln_qkv(x.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], wp.data_ptr(),
                                                                   bp.data_ptr(), g.data_ptr(), b.data_ptr(), float(eps), _stream()),
                                               "rc_gma_ln_qkv"))


def _tail_alloc(qkvp, convv, loc, x, ktv, w_proj, b_proj, ln_g, ln_b, eps, w_fc1, b_fc1, w_fc2, b_fc2, res, w_out, b_out):
    return x.new_empty(res.shape) if res is not None else torch.empty_like(x)


def _tail_launch(out, qkvp, convv, loc, x, ktv, w_proj, b_proj, ln_g, ln_b, eps, w_fc1, b_fc1, w_fc2, b_fc2, res, w_out, b_out):
    b = x.shape[0]
    n_tok = x.numel() // (b * x.shape[-1])
    frags = torch.empty((b, 8 * 1024), dtype=torch.uint8, device=x.device)
    check(lib().rc_gma_tail(qkvp.data_ptr(), convv.data_ptr(), loc.data_ptr(), x.data_ptr(), ktv.data_ptr(), frags.data_ptr(), b, n_tok,
                            w_proj.data_ptr(), b_proj.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), float(eps), w_fc1.data_ptr(),
                            b_fc1.data_ptr(), w_fc2.data_ptr(), b_fc2.data_ptr(), _p(res), _p(w_out), _p(b_out),
                            res.shape[-1] if res is not None else 0, out.data_ptr(), _stream()), "rc_gma_tail")


define("gma_tail(Tensor qkvp, Tensor convv, Tensor loc, Tensor x, Tensor ktv, Tensor w_proj, Tensor b_proj, Tensor ln_gamma, Tensor ln_beta, "
       "float eps, Tensor w_fc1, Tensor b_fc1, Tensor w_fc2, Tensor b_fc2, Tensor? res, Tensor? w_out, Tensor? b_out) -> Tensor",
       _tail_alloc, _tail_launch)


define("gma_aggregate(Tensor qkv, Tensor dw3, Tensor dw5, Tensor dw7, Tensor dwl, Tensor pw, Tensor pwl, Tensor bn_scale, Tensor bn_shift, "
       "Tensor ln_gamma, Tensor ln_beta) -> (Tensor, Tensor, Tensor)",
       lambda qkv, *a: (qkv.new_empty((12, *qkv.shape[1:4], 16)), qkv.new_empty((*qkv.shape[1:4], 16)),     # qkv: (15, B, H, W, 16)
                        qkv.new_empty((qkv.shape[1], 64), dtype=torch.float32)),                             # per-channel max of the aggregated k
       lambda outs, qkv, dw3, dw5, dw7, dwl, pw, pwl, sc, sh, lg, lb: check(
           lib().rc_gma_aggregate(qkv.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), qkv.shape[1], qkv.shape[2], qkv.shape[3], dw3.data_ptr(),
                                  dw5.data_ptr(), dw7.data_ptr(), dwl.data_ptr(), pw.data_ptr(), pwl.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                  lg.data_ptr(), lb.data_ptr(), outs[2].data_ptr(), _stream()), "rc_gma_aggregate"))


def _cpackn_launch(wp, weight):
    cout, cin = weight.shape[0], weight.shape[1]
    w_host = np.ascontiguousarray(weight.detach().float().reshape(cout, cin).cpu().numpy())
    dst = np.empty(wp.numel(), dtype=np.uint8)
    check(lib().rc_chain_pack_weights_natural(w_host.ctypes.data, cin, cout, dst.ctypes.data), "rc_chain_pack_weights_natural")
    wp.copy_(torch.from_numpy(dst))


define("chain_pack_weights_natural(Tensor weight) -> Tensor",
       lambda weight: weight.new_empty((lib().rc_chain_packed_bytes(weight.shape[1], weight.shape[0]),), dtype=torch.uint8), _cpackn_launch)

def _toep_launch(out, dw3, dw5, dw7, dwl):
    arrs = [np.ascontiguousarray(t.detach().float().cpu().numpy()) for t in (dw3, dw5, dw7, dwl)]
    dst = np.empty(out.numel(), dtype=np.uint8)
    check(lib().rc_gma_toeplitz_pack(*[a.ctypes.data for a in arrs], dst.ctypes.data), "rc_gma_toeplitz_pack")
    out.copy_(torch.from_numpy(dst))


define("gma_toeplitz_pack(Tensor dw3, Tensor dw5, Tensor dw7, Tensor dwl) -> Tensor",
       lambda dw3, *a: dw3.new_empty((lib().rc_gma_toeplitz_bytes(),), dtype=torch.uint8), _toep_launch)

def _dwtoep_launch(out, taps, ksize):
    t = np.ascontiguousarray(taps.detach().float().cpu().numpy())
    dst = np.empty(out.numel(), dtype=np.uint8)
    check(lib().rc_dw_toeplitz_pack(t.ctypes.data, int(ksize), t.shape[1], dst.ctypes.data), "rc_dw_toeplitz_pack")
    out.copy_(torch.from_numpy(dst))


define("dw_toeplitz_pack(Tensor taps, int ksize) -> Tensor",                       # taps: tap-major (K * K, n_ch) fp32 (ops.dw_taps)
       lambda taps, ksize: taps.new_empty((int(ksize) * taps.shape[1] * 1024,), dtype=torch.uint8), _dwtoep_launch)

define("gma_in_cpe(Tensor d1, Tensor w_in_natural, Tensor? b_in, Tensor toeplitz3, Tensor? b_cpe) -> Tensor",      # (B, H, W, 192) -> (B, H, W, 80)
       lambda d1, *a: d1.new_empty((*d1.shape[:3], 80)),
       lambda out, d1, w, bi, toep, bc: check(lib().rc_gma_in_cpe(d1.data_ptr(), w.data_ptr(), _p(bi), toep.data_ptr(), _p(bc), out.data_ptr(), d1.shape[0], d1.shape[1],
                                                                  d1.shape[2], _stream()), "rc_gma_in_cpe"))

define("gma_qkv_aggregate(Tensor x, Tensor wq_natural, Tensor? bq, Tensor ln1_gamma, Tensor ln1_beta, float eps, Tensor toeplitz, "
       "Tensor pw, Tensor pwl, Tensor bn_scale, Tensor bn_shift, Tensor ln_gamma, Tensor ln_beta) -> (Tensor, Tensor, Tensor)",
       lambda x, *a: (x.new_empty((12, *x.shape[:3], 16)), x.new_empty((*x.shape[:3], 16)),                  # x: (B, H, W, 80)
                      x.new_empty((x.shape[0], 64), dtype=torch.float32)),
       lambda outs, x, wq, bq, g1, b1, eps, toep, pw, pwl, sc, sh, lg, lb: check(
           lib().rc_gma_qkv_aggregate(x.data_ptr(), wq.data_ptr(), _p(bq), g1.data_ptr(), b1.data_ptr(), float(eps), outs[0].data_ptr(),
                                      outs[1].data_ptr(), x.shape[0], x.shape[1], x.shape[2], toep.data_ptr(), pw.data_ptr(), pwl.data_ptr(),
                                      sc.data_ptr(), sh.data_ptr(), lg.data_ptr(), lb.data_ptr(), outs[2].data_ptr(), _stream()),
           "rc_gma_qkv_aggregate"))


def _kvm_launch(ktv, qkvp, kmax, scale):
    b, n = qkvp.shape[1], qkvp.shape[2] * qkvp.shape[3]
    scratch = torch.empty(lib().rc_gma_kv_mfma_scratch_bytes(b, n) // 4, dtype=torch.float32, device=qkvp.device)
    check(lib().rc_gma_kv_mfma(qkvp.data_ptr(), b, n, float(scale), kmax.data_ptr(), scratch.data_ptr(), ktv.data_ptr(), _stream()), "rc_gma_kv_mfma")


define("gma_kv_mfma(Tensor qkvp, Tensor kmax, float scale) -> Tensor",
       lambda qkvp, kmax, scale: qkvp.new_empty((qkvp.shape[1], 8, 8, 8), dtype=torch.float32), _kvm_launch)

define("gma_crpe(Tensor qkvp, Tensor taps0, Tensor taps1, Tensor taps2, Tensor taps3, Tensor bias) -> Tensor",
       lambda qkvp, *a: qkvp.new_empty((4, *qkvp.shape[1:4], 16)),                 # segment-planar in and out
       lambda out, qkvp, t0, t1, t2, t3, bias: check(
           lib().rc_gma_crpe(qkvp.data_ptr(), out.data_ptr(), qkvp.shape[1], qkvp.shape[2], qkvp.shape[3], t0.data_ptr(), t1.data_ptr(),
                             t2.data_ptr(), t3.data_ptr(), bias.data_ptr(), _stream()), "rc_gma_crpe"))


# ---- f3: entropy coding (csrc/rans.hip) ------------------------------------------------------------------------------------------------
def _gcs_alloc(y, mu, scale, table, scale_bound):
    b, c = scale.shape[0], scale.shape[-1]
    hw = scale.numel() // (b * c)
    i32 = lambda: scale.new_empty((b, c, hw), dtype=torch.int32)
    return (i32() if y is not None else scale.new_empty((0,), dtype=torch.int32)), i32(), (torch.empty_like(y) if y is not None else _none(scale))


def _gcs_launch(outs, y, mu, scale, table, scale_bound):
    sym, idx, y_hat = outs
    b, c = scale.shape[0], scale.shape[-1]
    check(lib().rc_gc_symbols(_p(y), _p(mu), scale.data_ptr(), _dt(scale), b, scale.numel() // (b * c), c, table.data_ptr(), table.numel(),
                              float(scale_bound), _p(sym) if y is not None else None, idx.data_ptr(), _p(y_hat) if y is not None else None,
                              _stream()), "rc_gc_symbols")


define("gc_symbols(Tensor? y, Tensor? mu, Tensor scale, Tensor scale_table, float scale_bound) -> (Tensor, Tensor, Tensor)", _gcs_alloc, _gcs_launch)

define("gc_dequantize(Tensor symbols, Tensor mu) -> Tensor",
       lambda sym, mu: torch.empty_like(mu),
       lambda out, sym, mu: check(lib().rc_gc_dequantize(sym.data_ptr(), mu.data_ptr(), _dt(mu), mu.shape[0], mu.numel() // (mu.shape[0] * mu.shape[-1]),
                                                        mu.shape[-1], out.data_ptr(), _stream()), "rc_gc_dequantize"))


def _ebs_alloc(z, symbols, medians, b, h, w, dtype):
    c = medians.numel()
    return (symbols.new_empty((b, c, h * w)) if z is None else z.new_empty((b, c, h * w), dtype=torch.int32),
            medians.new_empty((b, c, h * w), dtype=torch.int32), medians.new_empty((b, h, w, c), dtype=dtype))


def _ebs_launch(outs, z, symbols, medians, b, h, w, dtype):
    sym, idx, z_hat = outs
    if z is None:
        sym.copy_(symbols.reshape(sym.shape))
    check(lib().rc_eb_symbols(_p(z), medians.data_ptr(), _DT[dtype], b, h * w, medians.numel(), 1 if z is not None else 0, sym.data_ptr(),
                              idx.data_ptr(), z_hat.data_ptr(), _stream()), "rc_eb_symbols")


define("eb_symbols(Tensor? z, Tensor? symbols, Tensor medians, int b, int h, int w, ScalarType dtype) -> (Tensor, Tensor, Tensor)", _ebs_alloc, _ebs_launch)


def _enc_alloc(sym, idx, cdf, sizes, offsets, chunk):
    n_chunks = -(-sym.numel() // chunk)
    return sym.new_empty((n_chunks, lib().rc_rans_chunk_words(chunk)), dtype=torch.int32), sym.new_empty((n_chunks,), dtype=torch.int32)


def _enc_launch(outs, sym, idx, cdf, sizes, offsets, chunk):
    # the prepared per-symbol operations live only for this call (the caching allocator keeps the block alive until the stream passes it)
    scratch = torch.empty((lib().rc_rans_encode_scratch_bytes(sym.numel(), chunk) + 7) // 8, dtype=torch.int64, device=sym.device)
    check(lib().rc_rans_encode_chunks(sym.data_ptr(), idx.data_ptr(), sym.numel(), chunk, cdf.data_ptr(), cdf.shape[1], cdf.shape[0], sizes.data_ptr(),
                                      offsets.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), scratch.data_ptr(), _stream()), "rc_rans_encode_chunks")


define("rans_encode_chunks(Tensor symbols, Tensor indexes, Tensor cdf, Tensor cdf_sizes, Tensor offsets, int chunk) -> (Tensor, Tensor)", _enc_alloc, _enc_launch)

define("rans_compact(Tensor words, Tensor nbytes, Tensor offsets, int chunk, int total_bytes) -> Tensor",
       lambda words, nbytes, offsets, chunk, total: words.new_empty((total,), dtype=torch.uint8),
       lambda out, words, nbytes, offsets, chunk, total: check(
           lib().rc_rans_compact(words.data_ptr(), chunk, nbytes.data_ptr(), offsets.data_ptr(), nbytes.numel(), out.data_ptr(), _stream()),
           "rc_rans_compact"))


def _dec_launch(out, stream, offsets, idx, cdf, sizes, cdf_offsets, chunk):
    err = torch.zeros(1, dtype=torch.int32, device=idx.device)
    check(lib().rc_rans_decode_chunks(stream.data_ptr(), stream.numel(), offsets.data_ptr(), idx.data_ptr(), idx.numel(), chunk, cdf.data_ptr(),
                                      cdf.shape[1], cdf.shape[0], sizes.data_ptr(), cdf_offsets.data_ptr(), out.data_ptr(), err.data_ptr(), _stream()),
          "rc_rans_decode_chunks")
    e = int(err.item())
    if e:
        raise _lib.HipError("rc_rans_decode_chunks: " + ("CDF index out of range" if e == 1 else "truncated or corrupt stream"))


define("rans_decode_chunks(Tensor stream, Tensor offsets, Tensor indexes, Tensor cdf, Tensor cdf_sizes, Tensor cdf_offsets, int chunk) -> Tensor",
       lambda stream, offsets, idx, *a: torch.empty_like(idx), _dec_launch)


def _dec_async_launch(outs, stream, offsets, idx, cdf, sizes, cdf_offsets, chunk):
    out, err = outs
    err.zero_()
    check(lib().rc_rans_decode_chunks(stream.data_ptr(), stream.numel(), offsets.data_ptr(), idx.data_ptr(), idx.numel(), chunk, cdf.data_ptr(),
                                      cdf.shape[1], cdf.shape[0], sizes.data_ptr(), cdf_offsets.data_ptr(), out.data_ptr(), err.data_ptr(), _stream()),
          "rc_rans_decode_chunks")


# the same launch without the host's look at the error flag: (symbols, err int32[1]) -- 0 ok, 1 CDF index out of range, 2 truncated / corrupt stream.  The
# caller checks `err` once, after everything that depends on the symbols has been enqueued (one sync per decompress() instead of one per slice and image)
define("rans_decode_chunks_async(Tensor stream, Tensor offsets, Tensor indexes, Tensor cdf, Tensor cdf_sizes, Tensor cdf_offsets, int chunk) -> (Tensor, Tensor)",
       lambda stream, offsets, idx, *a: (torch.empty_like(idx), idx.new_empty((1,), dtype=torch.int32)), _dec_async_launch)


define("film_apply(Tensor x, Tensor scale, Tensor shift) -> Tensor",
       lambda x, s, t: torch.empty_like(x),
       lambda out, x, s, t: check(lib().rc_film_apply(x.data_ptr(), s.data_ptr(), t.data_ptr(), out.data_ptr(), _dt(x), x.shape[0], x.shape[1] * x.shape[2],
                                                     x.shape[3], _stream()), "rc_film_apply"))

define("instance_norm(Tensor x, Tensor mean, Tensor rstd, Tensor gamma, Tensor beta) -> Tensor",
       lambda x, m, r, g, b: torch.empty_like(x),
       lambda out, x, m, r, g, b: check(lib().rc_instance_norm(x.data_ptr(), out.data_ptr(), m.data_ptr(), r.data_ptr(), g.data_ptr(), b.data_ptr(),
                                                               x.shape[0], x.shape[1], x.shape[2] * x.shape[3], _stream()), "rc_instance_norm"))


define("ln_mlp(Tensor x, Tensor ln_gamma, Tensor ln_beta, float eps, Tensor w_fc1, Tensor? b_fc1, Tensor w_fc2, Tensor? b_fc2) -> Tensor",
       lambda x, g, b, eps, w1, b1, w2, b2: torch.empty_like(x),
       lambda out, x, g, b, eps, w1, b1, w2, b2: check(lib().rc_ln_mlp(x.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], w1.data_ptr(),
                                                                       _p(b1), w2.data_ptr(), _p(b2), g.data_ptr(), b.data_ptr(), float(eps), _stream()),
                                                       "rc_ln_mlp"))

define("ln_linear(Tensor x, Tensor ln_gamma, Tensor ln_beta, float eps, Tensor w, Tensor? b, int cout) -> Tensor",
       lambda x, g, b, eps, w, bias, cout: x.new_empty((*x.shape[:-1], cout)),
       lambda out, x, g, b, eps, w, bias, cout: check(lib().rc_ln_linear(x.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], cout,
                                                                         w.data_ptr(), _p(bias), g.data_ptr(), b.data_ptr(), float(eps), _stream()),
                                                      "rc_ln_linear"))
define("ln_linear_planar8(Tensor x, Tensor ln_gamma, Tensor ln_beta, float eps, Tensor w, Tensor? b, int cout) -> Tensor",
       lambda x, g, b, eps, w, bias, cout: x.new_empty((*x.shape[:-1], cout)),
       lambda out, x, g, b, eps, w, bias, cout: check(lib().rc_ln_linear_planar8(x.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], cout,
                                                                                 w.data_ptr(), _p(bias), g.data_ptr(), b.data_ptr(), float(eps), _stream()),
                                                      "rc_ln_linear_planar8"))

define("pixel_shuffle2_nchw(Tensor x) -> Tensor",
       lambda x: x.new_empty((x.shape[0], x.shape[3] // 4, 2 * x.shape[1], 2 * x.shape[2])),
       lambda out, x: check(lib().rc_pixel_shuffle2_nchw(x.data_ptr(), out.data_ptr(), _dt(x), x.shape[0], x.shape[1], x.shape[2], x.shape[3] // 4,
                                                         _stream()), "rc_pixel_shuffle2_nchw"))

define("gdn_chain(Tensor x, Tensor? identity, Tensor gamma_packed, Tensor beta_packed, bool inverse) -> Tensor",
       lambda x, idn, w, b, inv: torch.empty_like(x),
       lambda out, x, idn, w, b, inv: check(lib().rc_gdn_chain(x.data_ptr(), _p(idn), out.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], w.data_ptr(),
                                                               b.data_ptr(), 1 if inv else 0, _stream()), "rc_gdn_chain"))

define("cat_linear(Tensor a, Tensor? a_add, Tensor b, Tensor? residual, Tensor w, Tensor? bias) -> Tensor",
       lambda a, a2, b, res, w, bias: a.new_empty((*a.shape[:-1], a.shape[-1] + b.shape[-1])),
       lambda out, a, a2, b, res, w, bias: check(lib().rc_cat_linear(a.data_ptr(), _p(a2), b.data_ptr(), _p(res), out.data_ptr(), a.numel() // a.shape[-1],
                                                                     a.shape[-1] + b.shape[-1], w.data_ptr(), _p(bias), _stream()), "rc_cat_linear"))
