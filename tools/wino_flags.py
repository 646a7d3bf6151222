#!/usr/bin/env python3
"""Knock-out timing of the Winograd kernel (rc_debug_set("conv_flags")): 1 no epilogue/stores, 2 no MFMA phase, 4 no loads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops, _lib
L = _lib.load()
cin, cout, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 64, 544, 960)
c = N.Conv2d(cin, cout, 3, 1, 1).to("cuda").eval()
x = torch.randn(1, H, W, cin, device="cuda")
def timed(fn, n=30, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    for f in (0, 1, 2, 4, 8, 5, 13, 3, 7, 15, 0):
        L.rc_debug_set(b"conv_flags", f)
        print(f"flags {f} ({'no-store ' if f & 1 else ''}{'no-mfma ' if f & 2 else ''}{'no-loads ' if f & 4 else ''}{'no-transform' if f & 8 else ''}): {timed(lambda: ops.conv2d(x, c, act='relu')):.1f} us", flush=True)
    L.rc_debug_set(b"conv_flags", 0)
