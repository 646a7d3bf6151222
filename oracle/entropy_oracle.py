"""Entropy-coding side of the codecs: CDF tables, symbol / index preparation and the rANS stream.   *** TEST INFRASTRUCTURE ***

CPU restatement (torch fp32 + the C coder in oracle/rans_oracle.c) of what the reference's `compress` / `decompress`
(models/tcm.py:511-570, 592-637; models/raw2bit.py:1876-1944, 1961-2027) delegate to CompressAI:
    EntropyBottleneck.update / compress / decompress,  GaussianConditional.update_scale_table / update / build_indexes / quantize /
    dequantize,  compressai.ans.BufferedRansEncoder / RansDecoder,  compressai._CXX.pmf_to_quantized_cdf,  get_scale_table.
CompressAI is absent from /root/reference and unpinned upstream: written from its published definitions, PARITY
UNPINNED against the package itself.  It pins the build's own tables and GPU coder (tests/test_bitstream.py, bit-exact).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import torch

import tcm_oracle as TO

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    """oracle/_build/librans_oracle.so, built on demand with gcc (oracle/Makefile)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "_build", "librans_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "rans_oracle.c")):
            subprocess.run(["make", "-C", HERE, "-s"], check=True)
        L = C.CDLL(path)
        L.ro_pmf_to_quantized_cdf.restype = C.c_int
        L.ro_encode_with_indexes.restype = C.c_long
        L.ro_decode_stream.restype = C.c_int
        _LIB = L
    return _LIB


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def pmf_to_quantized_cdf(pmf, precision: int = 16) -> np.ndarray:
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(pmf.size + 1, dtype=np.uint32)
    n = lib().ro_pmf_to_quantized_cdf(pmf.ctypes.data_as(C.c_void_p), C.c_int(pmf.size), C.c_int(precision), out.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise ValueError(f"pmf_to_quantized_cdf failed ({n})")
    return out.astype(np.int32)


def _pmf_to_cdf(pmf: torch.Tensor, tail_mass: torch.Tensor, pmf_length: torch.Tensor, max_length: int) -> torch.Tensor:
    """EntropyModel._pmf_to_cdf: per row, quantise [pmf[:len], tail_mass] and left-align in a (rows, max_length + 2) int table."""
    cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
    for i, p in enumerate(pmf):
        prob = torch.cat((p[: int(pmf_length[i])], tail_mass[i]), dim=0)
        q = pmf_to_quantized_cdf(prob.numpy())
        cdf[i, : q.size] = torch.from_numpy(q)
    return cdf


def eb_update(sd, p: str):
    """EntropyBottleneck.update(): tables from the factorised density's parameters (fp32)."""
    q = sd[p + ".quantiles"].float()
    medians = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
    offset = -minima
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max().item())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
    sdf = {k: v.float() for k, v in sd.items() if k.startswith(p + ".")}
    lower = TO.eb_logits_cumulative(sdf, p, samples - 0.5)
    upper = TO.eb_logits_cumulative(sdf, p, samples + 0.5)
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    return {"_offset": offset.int(), "_quantized_cdf": _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length), "_cdf_length": (pmf_length + 2).int()}


def get_scale_table(lo: float = 0.11, hi: float = 256.0, levels: int = 64) -> torch.Tensor:
    """compressai get_scale_table() as the reference's update() calls it (models/tcm.py:430-435)."""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


def gc_update(scale_table: torch.Tensor, tail_mass: float = 1e-9):
    """GaussianConditional.update() after update_scale_table(scale_table)."""
    from scipy.stats import norm
    scale_table = scale_table.float()
    multiplier = -float(norm.ppf(tail_mass / 2))
    pmf_center = torch.ceil(scale_table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(pmf_length.max().item())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    sc = scale_table.unsqueeze(1)
    phi = lambda t: 0.5 * torch.erfc(-(2 ** -0.5) * t)
    upper, lower = phi((0.5 - samples) / sc), phi((-0.5 - samples) / sc)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    return {"scale_table": scale_table, "_offset": (-pmf_center).int(), "_quantized_cdf": _pmf_to_cdf(pmf, tail, pmf_length, max_length),
            "_cdf_length": (pmf_length + 2).int()}


def gc_build_indexes(scales: torch.Tensor, scale_table: torch.Tensor, scale_bound: float = 0.11) -> torch.Tensor:
    scales = torch.clamp_min(scales.float(), scale_bound)                # LowerBound(scale_bound)
    idx = torch.full(scales.shape, len(scale_table) - 1, dtype=torch.int32)
    for s in scale_table[:-1]:
        idx -= (scales <= s).int()
    return idx


def quantize_symbols(x: torch.Tensor, means: torch.Tensor) -> torch.Tensor:
    return torch.round(x.float() - means.float()).int()


def encode_with_indexes(symbols, indexes, tables) -> bytes:
    """BufferedRansEncoder().encode_with_indexes(...); flush()  -> the stream."""
    sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    cdf, sizes, offs = _i32(tables["_quantized_cdf"]), _i32(tables["_cdf_length"]).reshape(-1), _i32(tables["_offset"]).reshape(-1)
    cap = 8 * sym.size + 64
    out = np.zeros(cap, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib().ro_encode_with_indexes(p(sym), p(idx), C.c_long(sym.size), p(cdf), C.c_int(cdf.shape[1]), C.c_int(cdf.shape[0]), p(sizes), p(offs),
                                     p(out), C.c_long(cap))
    if n < 0:
        raise ValueError(f"ro_encode_with_indexes failed ({n})")
    return out[:n].tobytes()


class Decoder:
    """RansDecoder: set_stream(), then decode_stream(indexes, ...) any number of times."""

    class _State(C.Structure):
        _fields_ = [("x", C.c_uint64), ("pos", C.c_long)]

    def __init__(self, stream: bytes):
        self.buf = np.frombuffer(stream, dtype=np.uint8).copy()
        self.st = Decoder._State()
        lib().ro_dec_init(self.buf.ctypes.data_as(C.c_void_p), C.byref(self.st))

    def decode_stream(self, indexes, tables) -> np.ndarray:
        idx = _i32(indexes).reshape(-1)
        cdf, sizes, offs = _i32(tables["_quantized_cdf"]), _i32(tables["_cdf_length"]).reshape(-1), _i32(tables["_offset"]).reshape(-1)
        out = np.zeros(idx.size, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib().ro_decode_stream(p(self.buf), C.byref(self.st), p(idx), C.c_long(idx.size), p(cdf), C.c_int(cdf.shape[1]), C.c_int(cdf.shape[0]),
                                    p(sizes), p(offs), p(out))
        if rc != 0:
            raise ValueError(f"ro_decode_stream failed ({rc})")
        return out
