#!/usr/bin/env python3
"""48 -> 48 3x3 at the level-0 size (8 x 1088 x 1920): the persistent kernel vs the producer/consumer (ws) form with its knock-outs
(conv_flags 1 no stores, 2 no MFMA, 4 no tile loads), plain and gated (+ materialised input).  What is the data-movement floor of the kernel's own
structure?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
L = ops.lib()
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, H, W = 8, 1088, 1920
x = torch.randn(B, H, W, 48, device=dev, dtype=bf)
r = torch.randn(B, H, W, 48, device=dev, dtype=bf)
gate = torch.rand(B, 48, device=dev)
c = N.Conv2d(48, 48, 3, 1, 1).to(dev, bf).eval()
dst = torch.empty_like(x)
print(f"torch copy 1.6 GB -> 1.6 GB: {timed(lambda: dst.copy_(x)):6.3f} ms")
with torch.no_grad():
    for persist in (1, 2):
        L.rc_debug_set(b"persist", persist)
        for flags in (0, 1, 2, 4, 6):
            L.rc_debug_set(b"conv_flags", flags)
            a = timed(lambda: c._nhwc(x, act="relu"))
            b = timed(lambda: c._nhwc(x, residual=r))
            g = timed(lambda: ops.conv2d(x, c, gate=gate, skip=r, store_input=True, act="relu"))
            print(f"persist {persist} flags {flags}:  plain+relu {a:6.3f}   +residual {b:6.3f}   gated+store {g:6.3f} ms")
L.rc_debug_set(b"persist", 1); L.rc_debug_set(b"conv_flags", 0)
