#!/usr/bin/env python3
"""Timing experiment (results are WRONG under conv_flags 32): does it pay to permute the cout order at pack time so that one store instruction's four lane
groups write 64 contiguous bytes of a pixel (today: four 16-byte pieces 24 / 32 bytes apart)?  48 -> 48 at level 0 and 192 -> 192 at level 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
L = ops.lib()
dev, bf = "cuda", torch.bfloat16


def timed(fn, n=10, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for (cin, cout, H, W) in ((48, 48, 1088, 1920), (48, 48, 544, 960), (192, 192, 544, 960), (128, 128, 272, 480)):
        c = N.Conv2d(cin, cout, 3, 1, 1).to(dev, bf).eval()
        x = torch.randn(8, H, W, cin, device=dev, dtype=bf)
        r = torch.randn(8, H, W, cout, device=dev, dtype=bf)
        for _ in range(20): c._nhwc(x, act="relu")
        row = []
        for flags in (0, 32, 0, 32):
            L.rc_debug_set(b"conv_flags", flags)
            row.append(f"flags {flags:2d}: relu {timed(lambda: c._nhwc(x, act='relu')):.3f}  +sums {timed(lambda: c._nhwc(x, act='relu', want_sums=True)[0]):.3f}  +res {timed(lambda: c._nhwc(x, residual=r)):.3f}")
        L.rc_debug_set(b"conv_flags", 0)
        print(f"{cin}->{cout} 8x{H}x{W}:  " + "   |   ".join(row))
