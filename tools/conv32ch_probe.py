#!/usr/bin/env python3
"""The ISPUNet family's 32 -> 32 (level 0) and 64 -> 64 (level 1) 3x3 layers: time of the forms the early-gate RCAB uses, against the HBM floor of their bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
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
    for (c, H, W) in ((32, 1088, 1920), (64, 544, 960), (128, 272, 480)):
        conv = N.Conv2d(c, c, 3, 1, 1).to(dev, bf).eval()
        x = torch.randn(8, H, W, c, device=dev, dtype=bf); r = torch.randn_like(x); g = torch.rand(8, c, device=dev)
        for _ in range(20): conv._nhwc(x, act="relu")
        gb = x.numel() * 2 / 1e9
        print(f"{c}->{c} 8x{H}x{W} ({gb:.2f} GB per map; copy-rate floor {2 * gb / 5.9:.3f} ms plain, {3 * gb / 5.9:.3f} ms with residual): "
              f"relu {timed(lambda: conv._nhwc(x, act='relu')):.3f}  relu+sums {timed(lambda: conv._nhwc(x, act='relu', want_sums=True)[0]):.3f}  "
              f"gate+res {timed(lambda: conv._nhwc(x, out_scale=g, residual=r)):.3f} ms")
