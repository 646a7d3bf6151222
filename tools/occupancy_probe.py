#!/usr/bin/env python3
"""How much do the HBM-paced 48 -> 48 layers owe to having TWO persistent blocks per CU?  conv_flags 64 pads the LDS so that one block fits (grid = #CUs)."""
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
    c = N.Conv2d(48, 48, 3, 1, 1).to(dev, bf).eval()
    x = torch.randn(8, 1088, 1920, 48, device=dev, dtype=bf); r = torch.randn_like(x); g = torch.rand(8, 48, device=dev)
    for _ in range(20): c._nhwc(x, act="relu")
    for flags in (0, 64, 0, 64):
        L.rc_debug_set(b"conv_flags", flags)
        print(f"conv_flags {flags:2d} ({'one' if flags else 'two'} block(s) per CU): relu {timed(lambda: c._nhwc(x, act='relu')):.3f}  relu+sums {timed(lambda: c._nhwc(x, act='relu', want_sums=True)[0]):.3f}  "
              f"gate+res {timed(lambda: c._nhwc(x, out_scale=g, residual=r)):.3f} ms")
    L.rc_debug_set(b"conv_flags", 0)
