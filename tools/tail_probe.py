#!/usr/bin/env python3
"""Knock-out timing of the two tail convolutions at the cfg3 size (8 x 1088 x 1920): 48 -> 192 + PixelShuffle(2), and 48 -> 3 planar at
2176 x 3840.  conv_flags: 1 no stores, 2 no MFMA, 4 no tile loads."""
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
c1 = N.Conv2d(48, 192, 3, 1, 1).to(dev, bf).eval()
c2 = N.Conv2d(48, 3, 3, 1, 1).to(dev, bf).eval()
with torch.no_grad():
    t = c1._nhwc(x, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)
    for flags in (0, 1, 2, 4, 3, 5, 6):
        L.rc_debug_set(b"conv_flags", flags)
        a = timed(lambda: c1._nhwc(x, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2))
        b = timed(lambda: c2._nhwc(t, out_mode=ops.RC_OUT_NCHW, out_dtype=bf))
        print(f"flags {flags}:  48->192 PS {a:6.3f} ms    48->3 NCHW {b:6.3f} ms")
    L.rc_debug_set(b"conv_flags", 0)
    outs = {}
    for pp in (0, 1):
        L.rc_debug_set(b"pss", pp)
        outs[pp] = c1._nhwc(x, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2).clone()
        print(f"pss {pp}: 48->192 PS {timed(lambda: c1._nhwc(x, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)):6.3f} ms")
    print("staged == direct store:", torch.equal(outs[0], outs[1]))
    for hw in ((37, 70), (16, 32), (130, 200)):                     # ragged edges
        xs = torch.randn(2, *hw, 48, device=dev, dtype=bf)
        r = []
        for pp in (0, 1):
            L.rc_debug_set(b"pss", pp)
            r.append(c1._nhwc(xs, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2).clone())
        print(hw, "equal:", torch.equal(r[0], r[1]))
    L.rc_debug_set(b"pss", 1)
    y = c1._nhwc(x)
    print(f"48->192 NHWC (no shuffle) at this size {timed(lambda: c1._nhwc(x)):6.3f} ms")
