#!/usr/bin/env python3
"""Time the lens-shading chain forms at the cfg3 size (8 x 1088 x 1920 pixels, width 48, bf16) and the codec's (4 x 1152 x 1920, width 128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops, networks as N

dev, dt = "cuda", torch.bfloat16


def timeit(name, fn, n=20):
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:60s} {e0.elapsed_time(e1) / n * 1e3:9.1f} us")
    return out


for width, (B, H, W) in ((48, (8, 1088, 1920)), (128, (4, 1152, 1920))):
    torch.manual_seed(0)
    lsc = M.LiteISP.Lens_Shading_Correction(2, width, width).to(dev, dt).eval()
    head = N.Conv2d(4, width, 3, 1, 1).to(dev, dt).eval()
    coord = ops.make_coord(B, H, W, dev, dt)
    c = ops.to_nhwc(coord)
    a = torch.rand(B, H, W, 4, device=dev).to(dt)
    with torch.no_grad():
        timeit(f"w{width}: lsc_chain (registers)", lambda: ops.lsc_chain(lsc, c))
        timeit(f"w{width}: lsc_chain + head (registers, one launch)", lambda: ops.lsc_chain(lsc, c, head, a))
        if width == 48:
            timeit(f"w{width}: pointwise_chain48 (LDS slab)", lambda: ops.pointwise_chain(c, list(lsc.model)[0::2], 0.1))
        m = timeit(f"w{width}: layer by layer (4 x rc_conv2d)", lambda: lsc.model._nhwc(c))
        timeit(f"w{width}: head conv with mul_plus1", lambda: head._nhwc(a, mul_plus1=m))
