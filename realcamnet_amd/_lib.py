"""ctypes binding of librealcam_hip.so (include/realcam_hip.h).

There is no CPU fallback: if the library is missing or does not export a declared symbol the import
fails loudly, and every op refuses non-CUDA tensors.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["RC_HIP_LIB"]) if os.environ.get("RC_HIP_LIB") else PKG / "librealcam_hip.so"   # override: kernel experiments only
HEADER = PKG.parent / "include" / "realcam_hip.h"

RC_F32, RC_BF16, RC_U16 = 0, 1, 2
RC_ACT_NONE, RC_ACT_RELU, RC_ACT_LEAKY, RC_ACT_GELU, RC_ACT_RELU_POST = 0, 1, 2, 3, 4
RC_OUT_NHWC, RC_OUT_PIXEL_SHUFFLE2, RC_OUT_NCHW, RC_OUT_PIXEL_SHUFFLE2_NCHW, RC_OUT_NHWC_DWT = 0, 1, 2, 3, 4
ABI_VERSION = 14


class ConvDesc(C.Structure):
    """Mirror of `struct rc_conv_desc` -- field order and types must match the header exactly."""
    _fields_ = [
        ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32),
        ("dtype", C.c_int32),
        ("in0", C.c_void_p), ("in1", C.c_void_p), ("in_gate", C.c_void_p), ("in_store", C.c_void_p),
        ("wpacked", C.c_void_p), ("bias", C.c_void_p),
        ("film_scale", C.c_void_p), ("film_shift", C.c_void_p),
        ("act", C.c_int32), ("act_slope", C.c_float),
        ("mul_plus1", C.c_void_p), ("residual", C.c_void_p),
        ("out", C.c_void_p), ("out_mode", C.c_int32), ("out_dtype", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32),
        ("chan_sums", C.c_void_p),
        ("src_h", C.c_int32), ("src_w", C.c_int32),
        ("out_scale", C.c_void_p),
        ("chan_sums_slots", C.c_int32), ("cout_tile", C.c_int32),
        ("algo", C.c_int32),
    ]


class ConvPairDesc(C.Structure):
    """Mirror of `struct rc_conv_pair_desc`."""
    _fields_ = [
        ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32),
        ("dtype", C.c_int32),
        ("in0", C.c_void_p), ("in1", C.c_void_p), ("in_gate", C.c_void_p), ("in_store", C.c_void_p),
        ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("film_scale", C.c_void_p), ("film_shift", C.c_void_p),
        ("act1", C.c_int32), ("act1_slope", C.c_float),
        ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("residual", C.c_void_p), ("out", C.c_void_p), ("chan_sums", C.c_void_p),
    ]


def declared_symbols() -> list[str]:
    """Every function the public header declares (used by the CPU-side ABI test)."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", text)))


_P, _I, _F, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SIGS = {
    "rc_abi_version": (C.c_int, []),
    "rc_last_error": (C.c_char_p, []),
    "rc_build_info": (C.c_char_p, []),
    "rc_device_arch": (C.c_int, [C.c_char_p, _SZ]),
    "rc_bayer_unshuffle": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rc_raw_ingest": (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "rc_nchw_to_nhwc": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "rc_nhwc_to_nchw": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "rc_conv_packed_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "rc_conv_pack_weights": (C.c_int, [_P, _I, _I, _I, _I, _I, _P]),
    "rc_conv_packed_cout": (C.c_int, [_I, _I, _I, _I, _I]),
    "rc_conv_pack_bias": (C.c_int, [_P, _I, _I, _I, _I, _I, _P]),
    "rc_conv_packed_bytes_ct": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "rc_wino_packed_bytes": (_SZ, [_I, _I, _I]),
    "rc_wino_pack_weights": (C.c_int, [_P, _I, _I, _I, _P]),
    "rc_conv_packed_cout_ct": (C.c_int, [_I, _I, _I, _I, _I, _I]),
    "rc_conv_pack_weights_ct": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _P]),
    "rc_conv_pack_bias_ct": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _P]),
    "rc_conv_sum_tiles": (C.c_int, [_I, _I]),
    "rc_conv_sum_slots": (C.c_int, [_P]),
    "rc_conv2d": (C.c_int, [C.POINTER(ConvDesc), _P]),
    "rc_conv_desc_size": (_SZ, []),
    "rc_tail_fold_weights": (C.c_int, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "rc_tail_ring_gather": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rc_tail_ring_scatter": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "rc_debug_stream_create_masked": (C.c_int, [_I, C.POINTER(C.c_void_p)]),
    "rc_debug_hbm_probe": (C.c_int, [_P, _P, C.c_size_t, _I, _I, _I, _I, _I, C.POINTER(C.c_double)]),
    "rc_debug_mfma_peak": (C.c_int, [_I, _I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rc_debug_mfma_peak32": (C.c_int, [_I, _I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rc_debug_poison_lds": (C.c_int, [C.c_uint, _P]),
    "rc_pointwise_chain48": (C.c_int, [_P, _I, _P, _P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _F, _P, _I, C.c_longlong, _P]),
    "rc_lsc_packed_bytes": (C.c_size_t, [_I, _I, _I]),
    "rc_lsc_pack": (C.c_int, [_P, _P, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _P, _P, _I, _I, _P]),
    "rc_lsc_chain": (C.c_int, [_P, _I, _P, _I, _I, _F, _P, _I, _P, _I, _I, _I, _P]),
    "rc_film_apply": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "rc_sigmoid_gate_add": (C.c_int, [_P, _P, _P, _P, _I, C.c_longlong, _P]),
    "rc_subsample2": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "rc_entropy_bottleneck": (C.c_int, [_P, _P, _P, _P, _P, _I, C.c_longlong, _I, C.c_float, _P]),
    "rc_gaussian_conditional": (C.c_int, [_P, _P, _P, _P, _P, _I, C.c_longlong, C.c_float, C.c_float, _P]),
    "rc_tanh_half_add": (C.c_int, [_P, _P, _P, _I, C.c_longlong, _P]),
    "rc_upsample_bilinear2": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "rc_sft_apply": (C.c_int, [_P, _P, _P, _P, _P, _I, C.c_longlong, _P]),
    "rc_space_to_depth2": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "rc_pixel_shuffle2": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "rc_pixel_shuffle2_nchw": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "rc_square": (C.c_int, [_P, _P, _I, C.c_longlong, _P]),
    "rc_gdn_apply": (C.c_int, [_P, _P, _P, _P, _I, _I, C.c_longlong, _P]),
    "rc_channel_copy": (C.c_int, [_P, _I, _I, _P, _I, _I, _I, C.c_longlong, _I, _P]),
    "rc_channel_concat": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), _I, _P, C.c_longlong, _I, _P]),
    "rc_window_attention": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "rc_window_attention_planar8": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "rc_window_attention_planar8_ok": (C.c_int, [_I, _I, _I, _I, _I, _I]),
    "rc_conv_pair": (C.c_int, [C.POINTER(ConvPairDesc), _P]),
    "rc_conv_pair_sum_slots": (C.c_int, [_I, _I]),
    "rc_conv_pair_desc_size": (_SZ, []),
    "rc_ca_gate": (C.c_int, [_P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "rc_ca_gate_ahead_scratch_floats": (_SZ, [_I, _I]),
    "rc_ca_gate_ahead": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rc_gate_residual": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "rc_channel_sums_slots": (C.c_int, [_I]),
    "rc_channel_sums": (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "rc_dwt_forward": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rc_dwt_inverse": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rc_color_block": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "rc_instance_stats": (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "rc_instance_norm": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rc_color_head": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "rc_gfm_vector": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "rc_dwconv2d": (C.c_int, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "rc_layernorm": (C.c_int, [_P, _P, _I, C.c_longlong, _I, _P, _P, _F, _P]),
    "rc_gma_pointwise": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, C.c_longlong, _I, _P, _P, _P, _P, _P, _P, _P]),
    "rc_chain_packed_bytes": (_SZ, [_I, _I]),
    "rc_chain_packed_rows": (C.c_int, [_I]),
    "rc_chain_pack_weights": (C.c_int, [_P, _I, _I, _P]),
    "rc_chain_pack_bias": (C.c_int, [_P, _I, _P]),
    "rc_gma_ln_qkv": (C.c_int, [_P, _P, C.c_longlong, _P, _P, _P, _P, _F, _P]),
    "rc_gma_tail": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "rc_gma_aggregate": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rc_chain_pack_weights_natural": (C.c_int, [_P, _I, _I, _P]),
    "rc_gma_toeplitz_bytes": (_SZ, []),
    "rc_gma_toeplitz_pack": (C.c_int, [_P, _P, _P, _P, _P]),
    "rc_dw_toeplitz_pack": (C.c_int, [_P, _I, _I, _P]),
    "rc_gma_in_cpe": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rc_gma_qkv_aggregate": (C.c_int, [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rc_cat_linear": (C.c_int, [_P, _P, _P, _P, _P, C.c_longlong, _I, _P, _P, _P]),
    "rc_gdn_chain": (C.c_int, [_P, _P, _P, C.c_longlong, _I, _P, _P, _I, _P]),
    "rc_ln_linear": (C.c_int, [_P, _P, C.c_longlong, _I, _I, _P, _P, _P, _P, _F, _P]),
    "rc_ln_linear_planar8": (C.c_int, [_P, _P, C.c_longlong, _I, _I, _P, _P, _P, _P, _F, _P]),
    "rc_ln_mlp": (C.c_int, [_P, _P, C.c_longlong, _I, _P, _P, _P, _P, _P, _P, _F, _P]),
    "rc_gma_kv_mfma_blocks": (C.c_int, [_I]),
    "rc_gma_kv_mfma_scratch_bytes": (C.c_size_t, [_I, _I]),
    "rc_gma_kv_mfma": (C.c_int, [_P, _I, _I, _F, _P, _P, _P, _P]),
    "rc_gma_crpe": (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "rc_gma_kv_blocks": (C.c_int, [_I]),
    "rc_gma_kv_scratch_bytes": (_SZ, [_I, _I, _I, _I]),
    "rc_gma_kv": (C.c_int, [_P, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "rc_gma_kv_planar": (C.c_int, [_P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "rc_gma_apply": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rc_pmf_to_quantized_cdf": (C.c_int, [_P, _I, _I, _P]),
    "rc_gc_symbols": (C.c_int, [_P, _P, _P, _I, _I, C.c_longlong, _I, _P, _I, _F, _P, _P, _P, _P]),
    "rc_gc_dequantize": (C.c_int, [_P, _P, _I, _I, C.c_longlong, _I, _P, _P]),
    "rc_eb_symbols": (C.c_int, [_P, _P, _I, _I, C.c_longlong, _I, _I, _P, _P, _P, _P]),
    "rc_rans_chunk_words": (C.c_int, [_I]),
    "rc_rans_encode_scratch_bytes": (_SZ, [C.c_longlong, _I]),
    "rc_rans_encode_chunks": (C.c_int, [_P, _P, C.c_longlong, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "rc_debug_rans_rcp_selftest": (C.c_longlong, [C.c_longlong, C.c_ulonglong]),
    "rc_rans_compact": (C.c_int, [_P, _I, _P, _P, C.c_longlong, _P, _P]),
    "rc_rans_decode_chunks": (C.c_int, [_P, C.c_longlong, _P, _P, C.c_longlong, _I, _P, _I, _I, _P, _P, _P, _P, _P]),
    "rc_rans_encode_host": (C.c_longlong, [_P, _P, C.c_longlong, _P, _I, _I, _P, _P, _P, C.c_longlong]),
    "rc_rans_decode_host": (C.c_int, [_P, C.c_longlong, _P, _P, C.c_longlong, _P, _I, _I, _P, _P, _P]),
    "rc_debug_set": (C.c_int, [C.c_char_p, _I]),
    "rc_debug_get": (C.c_int, [C.c_char_p]),
    "rc_debug_set_ptr": (C.c_int, [C.c_char_p, _P]),
    "rc_prof_enable": (C.c_int, [_I]),
    "rc_prof_collect": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rc_prof_collect_rows": (C.c_int, [_P, _I, C.POINTER(C.c_int)]),
}

_lib = None
_knobs = {}


def knob(key: bytes) -> int:
    """rc_debug_get(key), cached until the knob is set again (rc_debug_set through this binding drops the cached value)."""
    v = _knobs.get(key)
    if v is None:
        v = _knobs[key] = load().rc_debug_get(key)
    return v


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m realcamnet_amd.build` (hipcc, gfx950). "
            "There is no CPU / PyTorch fallback for the HIP path.")
    lib = C.CDLL(os.fspath(LIB_PATH))
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH.name} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.rc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {lib.rc_abi_version()} vs binding {ABI_VERSION}")
    # rc_debug_get is asked on every conv (pack-cache key): mirror the knobs here and invalidate the mirror whenever one is set through this binding
    raw_set = lib.rc_debug_set

    def _set(key, value):
        _knobs.pop(bytes(key), None)
        return raw_set(key, value)
    lib.rc_debug_set = _set
    _lib = lib
    # kernel experiments only: RC_DEBUG="key=value,key=value" applies rc_debug_set knobs at load (A/B runs of bench.py without code edits)
    for kv in filter(None, os.environ.get("RC_DEBUG", "").split(",")):
        k, _, v = kv.partition("=")
        if lib.rc_debug_set(k.strip().encode(), int(v or "1")) != 0:
            raise RuntimeError(f"RC_DEBUG: unknown knob {k!r}")
    return lib


class HipError(RuntimeError):
    pass


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().rc_last_error().decode(errors="replace")
        raise HipError(f"{what or 'librealcam_hip'} failed ({code}): {msg}")
