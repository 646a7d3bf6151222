"""Entropy coding of the codecs' latents on the GPU: CDF tables (`update()`), symbol preparation and the rANS
streams behind `compress()` / `decompress()` (upstream models/tcm.py:430-435, 511-570, 592-637; models/raw2bit.py:1876-2027).

Upstream hands Python lists to CompressAI's C++ coder; CompressAI is not in the reference tree (no version pinned), so its published
algorithms are restated here (parity unpinned against the package; pinned bit-exactly against oracle/rans_oracle.c + entropy_oracle.py):
  * tables: EntropyBottleneck.update / GaussianConditional.update(_scale_table) in fp32 torch on the host (one-off, like weight packing),
    quantised by rc_pmf_to_quantized_cdf;
  * symbols + CDF indexes are produced on the device (realcam::gc_symbols / eb_symbols) and never become Python lists;
  * two wire formats: "chunked" (default) -- every `chunk` consecutive symbols are one complete rANS stream in CompressAI's layout,
    coded by one GPU lane each, behind a small header with the chunk sizes; "compressai" -- ONE stream over all symbols, byte-for-byte
    CompressAI's BufferedRansEncoder layout, coded on the host by the same primitives (a rANS state is a serial chain).
"""
from __future__ import annotations

import ctypes as C
import math
import struct
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

MAGIC = b"RCR1"
DEFAULT_CHUNK = 2048
_R = torch.ops.realcam


# ---- tables ---------------------------------------------------------------------------------------------------------------------------
def _quantize_rows(pmf: torch.Tensor, tail_mass: torch.Tensor, pmf_length: torch.Tensor, max_length: int) -> torch.Tensor:
    """EntropyModel._pmf_to_cdf: row i = quantised CDF of [pmf[i, :len_i], tail_mass_i], left-aligned in (rows, max_length + 2)."""
    L = _lib.load()
    cdf = np.zeros((pmf.shape[0], max_length + 2), dtype=np.int32)
    pmf, tail_mass = pmf.float().cpu().numpy(), tail_mass.float().cpu().numpy()
    for i in range(pmf.shape[0]):
        n = int(pmf_length[i])
        prob = np.ascontiguousarray(np.concatenate([pmf[i, :n], tail_mass[i].reshape(-1)[:1]]).astype(np.float32))
        row = np.zeros(n + 2, dtype=np.int32)
        check(L.rc_pmf_to_quantized_cdf(prob.ctypes.data, n + 1, 16, row.ctypes.data), "rc_pmf_to_quantized_cdf")
        cdf[i, : n + 2] = row
    return torch.from_numpy(cdf)


def get_scale_table(lo: float = 0.11, hi: float = 256.0, levels: int = 64) -> torch.Tensor:
    """The scale table the reference's update() builds (compressai get_scale_table, models/tcm.py:430-432)."""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


def gaussian_tables(scale_table: torch.Tensor, tail_mass: float = 1e-9):
    """GaussianConditional.update(): (offset, quantized_cdf, cdf_length) int32 for a sorted scale table."""
    scale_table = scale_table.detach().float().cpu()
    multiplier = -float(torch.special.ndtri(torch.tensor(tail_mass / 2, dtype=torch.float64)))
    pmf_center = torch.ceil(scale_table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(pmf_length.max())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    sc = scale_table.unsqueeze(1)
    phi = lambda t: 0.5 * torch.erfc(-(2 ** -0.5) * t)
    upper, lower = phi((0.5 - samples) / sc), phi((-0.5 - samples) / sc)
    return (-pmf_center).int(), _quantize_rows(upper - lower, 2 * lower[:, :1], pmf_length, max_length), (pmf_length + 2).int()


def bottleneck_tables(eb):
    """EntropyBottleneck.update(): tables of the factorised density from the module's parameters (fp32 on the host)."""
    g = (lambda n: eb._master(n).cpu()) if hasattr(eb, "_master") else (lambda n: getattr(eb, n).detach().float().cpu())   # fp32 masters (tcm._Fp32Masters)
    q = g("quantiles")
    mats = [torch.nn.functional.softplus(g(f"_matrix{i}")) for i in range(5)]
    biases = [g(f"_bias{i}") for i in range(5)]
    factors = [torch.tanh(g(f"_factor{i}")) for i in range(4)]

    def logits(x):
        for i in range(5):
            x = torch.matmul(mats[i], x) + biases[i]
            if i < 4:
                x = x + factors[i] * torch.tanh(x)
        return x

    medians = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
    lower, upper = logits(samples - 0.5), logits(samples + 0.5)
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    return (-minima).int(), _quantize_rows(pmf, tail, pmf_length, max_length), (pmf_length + 2).int()


class Tables:
    """Device copies of an entropy model's (quantized_cdf, cdf_length, offset), int32."""

    def __init__(self, cdf: torch.Tensor, sizes: torch.Tensor, offsets: torch.Tensor, device):
        if cdf.numel() == 0:
            raise RuntimeError("entropy model has no CDF tables: call update() (or load a checkpoint that carries them) first")
        self.cdf = cdf.to(device=device, dtype=torch.int32).contiguous()
        self.sizes = sizes.to(device=device, dtype=torch.int32).reshape(-1).contiguous()
        self.offsets = offsets.to(device=device, dtype=torch.int32).reshape(-1).contiguous()
        self._host = None

    def host(self):
        if self._host is None:
            self._host = tuple(np.ascontiguousarray(t.cpu().numpy()) for t in (self.cdf, self.sizes, self.offsets))
        return self._host


# ---- streams ----------------------------------------------------------------------------------------------------------------------------
def encode(symbols: torch.Tensor, indexes: torch.Tensor, tables: Tables, fmt: str = "chunked", chunk: int = DEFAULT_CHUNK) -> bytes:
    """int32 device tensors (n,) -> one byte string."""
    symbols, indexes = symbols.reshape(-1).contiguous(), indexes.reshape(-1).contiguous()
    n = symbols.numel()
    if fmt == "compressai":
        L = _lib.load()
        sym, idx = np.ascontiguousarray(symbols.cpu().numpy()), np.ascontiguousarray(indexes.cpu().numpy())
        cdf, sizes, offs = tables.host()
        out = np.empty(8 * n + 64, dtype=np.uint8)
        nb = L.rc_rans_encode_host(sym.ctypes.data, idx.ctypes.data, n, cdf.ctypes.data, cdf.shape[1], cdf.shape[0], sizes.ctypes.data,
                                   offs.ctypes.data, out.ctypes.data, out.size)
        if nb < 0:
            raise _lib.HipError(f"rc_rans_encode_host: {L.rc_last_error().decode()}")
        return out[:nb].tobytes()
    if fmt != "chunked":
        raise ValueError(f"unknown stream format {fmt!r}")
    return finish([encode_async(symbols, indexes, tables, chunk)])[0]


class PendingStream:
    """A "chunked" container whose chunk streams have been coded on the device but not yet read back (encode_async -> finish)."""
    __slots__ = ("words", "nbytes", "ends", "n", "chunk")


def encode_async(symbols: torch.Tensor, indexes: torch.Tensor, tables: Tables, chunk: int = DEFAULT_CHUNK) -> PendingStream:
    """The device half of encode(fmt="chunked"): every chunk coded into the word arena, byte counts and their running sum on the device.  No host sync:
    the codec enqueues the whole slice loop this way and calls finish() once (upstream models/tcm.py:515-570 hands every slice to the host coder in turn)."""
    symbols, indexes = symbols.reshape(-1).contiguous(), indexes.reshape(-1).contiguous()
    p = PendingStream()
    p.n, p.chunk = symbols.numel(), int(chunk)
    p.words, p.nbytes = _R.rans_encode_chunks(symbols, indexes, tables.cdf, tables.sizes, tables.offsets, p.chunk)
    p.ends = torch.cumsum(p.nbytes.to(torch.int64), 0)
    return p


def finish(pending: List[PendingStream]) -> List[bytes]:
    """The host half: ONE sync for all pending containers (their chunk byte counts in one transfer), then compaction + read-back of each payload."""
    if not pending:
        return []
    counts = torch.cat([p.nbytes for p in pending]).cpu().numpy()          # the sync
    out, pos, payloads = [], 0, []
    for p in pending:
        sizes = counts[pos:pos + p.nbytes.numel()]
        pos += p.nbytes.numel()
        if int(sizes.min()) < 0:
            raise _lib.HipError("rc_rans_encode_chunks: CDF index out of range")
        total = int(sizes.astype(np.int64).sum())
        payloads.append(_R.rans_compact(p.words, p.nbytes, (p.ends - p.nbytes).contiguous(), p.chunk, total))
        out.append((p, sizes.astype("<u4")))
    return [struct.pack("<4sIII", MAGIC, p.n, p.chunk, sizes.size) + sizes.tobytes() + pl.cpu().numpy().tobytes() for (p, sizes), pl in zip(out, payloads)]


class Decoder:
    """`RansDecoder` for one string: set the stream, then decode(indexes) any number of times ("compressai": the state carries over
    like decode_stream; "chunked": the string is a sequence of containers, one per decode() call)."""

    def __init__(self, stream: bytes, tables: Tables, device, fmt: str = "chunked"):
        self.fmt, self.tables, self.device = fmt, tables, device
        self.buf = np.frombuffer(stream, dtype=np.uint8)
        self.pos = 0
        self.state = (C.c_ulonglong * 2)(0, 0)
        self.errs: List[torch.Tensor] = []          # device error flags of decode_async launches not yet looked at (check())
        if fmt == "compressai":
            self.buf = np.ascontiguousarray(self.buf)

    def check(self) -> None:
        """Look at the error flags of every decode_async() since the last check (one sync)."""
        errs, self.errs = self.errs, []
        if errs:
            e = int(torch.cat(errs).max().item())
            if e:
                raise _lib.HipError("rc_rans_decode_chunks: " + ("CDF index out of range" if e == 1 else "truncated or corrupt stream"))

    def decode_async(self, indexes: torch.Tensor) -> torch.Tensor:
        """decode() without the host's look at the kernel's error flag ("chunked" only; the container is still validated on the host first): the symbols may be
        consumed by further launches at once, check() must be called before the result is trusted."""
        return self.decode(indexes, _async=True)

    def decode(self, indexes: torch.Tensor, _async: bool = False) -> torch.Tensor:
        indexes = indexes.reshape(-1).contiguous()
        n = indexes.numel()
        if self.fmt == "compressai":
            L = _lib.load()
            idx = np.ascontiguousarray(indexes.cpu().numpy())
            cdf, sizes, offs = self.tables.host()
            out = np.empty(n, dtype=np.int32)
            if self.buf.size < 8 or self.buf.size % 4:
                raise ValueError("corrupt rANS stream: not a whole number of 32-bit words / shorter than the flushed state")
            check(L.rc_rans_decode_host(self.buf.ctypes.data, self.buf.size, self.state, idx.ctypes.data, n, cdf.ctypes.data, cdf.shape[1],
                                        cdf.shape[0], sizes.ctypes.data, offs.ctypes.data, out.ctypes.data), "rc_rans_decode_host")
            return torch.from_numpy(out).to(self.device)
        # the container is validated BEFORE anything is uploaded: the kernels bound every read by these sizes
        if self.buf.size - self.pos < 16:
            raise ValueError("corrupt chunked rANS container: truncated header")
        magic, n_sym, chunk, n_chunks = struct.unpack_from("<4sIII", self.buf, self.pos)
        if magic != MAGIC or chunk < 1 or n_sym != n or n_chunks != -(-n // chunk):
            raise ValueError("corrupt or mismatched chunked rANS container")
        p = self.pos + 16
        if self.buf.size - p < 4 * n_chunks:
            raise ValueError("corrupt chunked rANS container: truncated size table")
        sizes = np.frombuffer(self.buf, dtype="<u4", count=n_chunks, offset=p).astype(np.int64)
        p += 4 * n_chunks
        total = int(sizes.sum())
        if int(sizes.min()) < 8 or np.any(sizes % 4) or p + total > self.buf.size:
            raise ValueError("corrupt chunked rANS container: a chunk is shorter than its flushed state / not word-sized, or the payload is truncated")
        payload = torch.from_numpy(self.buf[p:p + total].copy()).to(self.device)
        offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)).to(self.device)
        self.pos = p + total
        if _async:
            sym, err = _R.rans_decode_chunks_async(payload, offsets, indexes, self.tables.cdf, self.tables.sizes, self.tables.offsets, int(chunk))
            self.errs.append(err)
            return sym
        return _R.rans_decode_chunks(payload, offsets, indexes, self.tables.cdf, self.tables.sizes, self.tables.offsets, int(chunk))
