#!/usr/bin/env python3
"""Time a conv layer with each cout tile width the kernels are instantiated for (rc_conv_desc.cout_tile): which small-map layers gain from more, narrower blocks?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.nn as N
from realcamnet_amd import ops, torch_ops
R = torch.ops.realcam
dev = "cuda"


def timed(fn, n=30, warm=20):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.05: fn()                  # clocks up: the first variant of a row used to read 5-40x slow on the tiny layers
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (dtype, B, H, W, cin, cout, ksize): the codec forward's distinct conv shapes at 8 frames (tools/codec_conv_breakdown.py 8), cfg3's small levels, cfg2's 128-channel level
shapes = [(torch.bfloat16, 8, 72, 120, 224, 128, 3), (torch.bfloat16, 8, 72, 120, 576, 224, 3), (torch.bfloat16, 8, 72, 120, 320, 224, 3), (torch.bfloat16, 8, 72, 120, 128, 64, 3),
          (torch.bfloat16, 8, 72, 120, 64, 64, 3), (torch.bfloat16, 8, 72, 120, 64, 128, 1), (torch.bfloat16, 8, 72, 120, 128, 64, 1), (torch.bfloat16, 8, 72, 120, 128, 128, 1),
          (torch.bfloat16, 8, 72, 120, 128, 512, 1), (torch.bfloat16, 8, 72, 120, 512, 128, 1), (torch.bfloat16, 8, 72, 120, 128, 384, 1), (torch.bfloat16, 8, 36, 60, 64, 64, 3),
          (torch.bfloat16, 8, 144, 240, 64, 64, 3), (torch.bfloat16, 8, 144, 240, 128, 128, 3), (torch.bfloat16, 8, 288, 480, 64, 64, 3), (torch.bfloat16, 8, 288, 480, 128, 64, 3),
          (torch.bfloat16, 8, 288, 480, 128, 128, 3), (torch.bfloat16, 8, 288, 480, 128, 64, 1), (torch.bfloat16, 8, 576, 960, 128, 64, 1), (torch.bfloat16, 8, 576, 960, 64, 64, 1),
          (torch.bfloat16, 8, 576, 960, 64, 32, 3), (torch.bfloat16, 8, 1152, 1920, 32, 16, 3), (torch.bfloat16, 8, 1152, 1920, 16, 64, 3),
          (torch.bfloat16, 8, 136, 240, 128, 128, 3), (torch.bfloat16, 8, 136, 240, 512, 512, 3), (torch.bfloat16, 8, 272, 480, 512, 128, 3), (torch.float32, 1, 135, 240, 128, 128, 3)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if str(s[4]) + "->" + str(s[5]) in sys.argv[1:]]
with torch.no_grad():
    for dt, B, H, W, cin, cout, k in shapes:
        conv = N.Conv2d(cin, cout, k, 1, k // 2).to(dev, dt).eval()
        x = torch.randn(B, H, W, cin, device=dev).to(dt)
        ref = None
        row = []
        for ct in (0, 16, 32, 48, 64):
            try:
                pc = ops.packed_conv(conv, dt, ops.RC_OUT_NHWC, ct)
                fn = lambda: R.conv2d(x, pc.wpacked, pc.bias, pc.cout, pc.ksize, 1, 0.0, None, None, None, None, None, None, False, 0, False, 0, 0, None, None, ct)[0]
                y = fn(); torch.cuda.synchronize()
                if ref is None: ref = y
                row.append(f"{ct}: {timed(fn):7.1f} us{'' if torch.equal(y, ref) else ' (DIFFERS)'}")
            except Exception as e:
                row.append(f"{ct}: n/a")
        print(f"{str(dt)[6:]:8s} {B}x{H}x{W} {cin:3d}->{cout:3d} k{k}   " + "   ".join(row), flush=True)
