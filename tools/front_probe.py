#!/usr/bin/env python3
"""cfg3's front end: the colour-prior chain (side stream) against the lens-shading head (main stream) -- the two arms of _DwtUNet._front's fork."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
torch.manual_seed(0)
net = M.LiteISPNet_GFM_LSC_GMA().eval().to(dev, dt)
B = 8
g = torch.Generator(device=dev).manual_seed(1234)
cond = torch.rand(B, 4, 256, 256, generator=g, device=dev).to(dt)
a = torch.rand(B, 1088, 1920, 4, generator=g, device=dev).to(dt)
coord = ops.to_nhwc(ops.make_coord(B, 1088, 1920, device=dev, dtype=dt), dtype=dt)


def timed(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    print(f"colour prior chain (10 launches)     {timed(lambda: net.classifier._vec(cond)):7.1f} us")
    print(f"lens-shading head (one launch)       {timed(lambda: ops.lsc_chain(net.lsc, coord, net.head, a)):7.1f} us")
    print(f"_front (fork: both)                  {timed(lambda: net._front(a, cond, coord)):7.1f} us")
