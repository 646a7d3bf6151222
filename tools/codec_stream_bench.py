#!/usr/bin/env python3
"""Time compress() / decompress() of raw_compression_tcm_final on the packed RAW of one 4K mosaic (4 x 1152 x 1920,
bf16, random-init weights after update()), both stream formats.   python tools/codec_stream_bench.py [--frames 1]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realcamnet_amd.raw2bit as RB
from realcamnet_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
H, W = 1152, 1920
m = RB.raw_compression_tcm_final().eval()
m.update()
m = m.to("cuda", torch.bfloat16)
x = [torch.rand(a.frames, 4, H, W, generator=g).to("cuda", torch.bfloat16), torch.rand(a.frames, 4, 256, 256, generator=g).to("cuda", torch.bfloat16),
     ops.make_coord(a.frames, H, W, "cuda", torch.bfloat16)]


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / a.steps * 1e3


with torch.no_grad():
    _, t_fwd = timed(lambda: m(x))
    for fmt in ("chunked", "compressai"):
        enc, t_enc = timed(lambda: m.compress(x, fmt=fmt))
        dec, t_dec = timed(lambda: m.decompress(enc["strings"], enc["shape"], fmt=fmt))
        nbytes = sum(len(s) for group in enc["strings"] for s in group)
        print(json.dumps({"format": fmt, "frames": a.frames, "packed_hw": [H, W], "forward_ms": round(t_fwd, 2), "compress_ms": round(t_enc, 2),
                          "decompress_ms": round(t_dec, 2), "stream_bytes": nbytes, "bpp_mosaic": round(8 * nbytes / (a.frames * 4 * H * W), 4),
                          "y_symbols": a.frames * 320 * (H // 16) * (W // 16)}))
        if fmt == "compressai":
            one_stream = nbytes
    # the chunk length is a field of the container: shorter chunks = more independent rANS streams = more decoder lanes, at a 64-bit flush + 4-byte size each
    for chunk in (4096, 2048, 1024, 512, 256):
        enc, t_enc = timed(lambda: m.compress(x, fmt="chunked", chunk=chunk))
        dec, t_dec = timed(lambda: m.decompress(enc["strings"], enc["shape"]))
        nbytes = sum(len(s) for group in enc["strings"] for s in group)
        print(json.dumps({"format": "chunked", "chunk": chunk, "compress_ms": round(t_enc, 2), "decompress_ms": round(t_dec, 2), "stream_bytes": nbytes,
                          "bytes_vs_one_stream": round(nbytes / one_stream, 4)}))
