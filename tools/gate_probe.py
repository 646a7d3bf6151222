#!/usr/bin/env python3
"""rc_ca_gate_ahead as ONE launch (last-block-done counter, rc_debug_set("gate_fused", 1)) against the two launches of round 4: bit-equality of the gate
(whichever block arrives last) and us per call at the cfg3 shapes; then the level-1 48 -> 192 (+ residual) layer and the tail ring's 48 -> 192 + PixelShuffle
on the multi-chunk kernel with its weights by LDS-DMA (spill-free) -- the A/B needs a rebuild, so only the times are printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops, _lib
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
L = _lib.load()


def timed(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


with torch.no_grad():
    for (B, H, W) in ((8, 1088, 1920), (8, 544, 960), (8, 272, 480), (3, 40, 72), (1, 8, 8)):
        blk = N.RCABlock(48, 48, 3, 1, 1, True, "CRC", 16).to(dev, bf).eval()
        x = torch.randn(B, H, W, 48, device=dev, dtype=bf)
        t, sums = blk.res[0]._nhwc(x, act="relu", want_sums=True)
        gates = []
        for v in (0, 1, 1, 1):
            assert L.rc_debug_set(b"gate_fused", v) == 0
            gates.append(ops.ca_gate_ahead(sums.clone(), t, blk.res[2], blk.ca))
        torch.cuda.synchronize()
        same = all(torch.equal(gates[0], g) for g in gates[1:])
        us = []
        for v in (0, 1):
            L.rc_debug_set(b"gate_fused", v)
            us.append(timed(lambda: ops.ca_gate_ahead(sums, t, blk.res[2], blk.ca)))
        print(f"{B}x{H}x{W}: gate bit-identical {same}   two launches {us[0]:.1f} us   one launch {us[1]:.1f} us")
    L.rc_debug_set(b"gate_fused", 1)
    # whole RCAGroup with both
    for (B, H, W) in ((8, 1088, 1920), (8, 544, 960)):
        rg = N.RCAGroup(48, 48, nb=4).to(dev, bf).eval()
        x = torch.randn(B, H, W, 48, device=dev, dtype=bf)
        ms = []
        for v in (0, 1):
            L.rc_debug_set(b"gate_fused", v)
            ms.append(timed(lambda: rg._nhwc(x), n=10, warm=10) / 1e3)
        print(f"{B}x{H}x{W} RCAGroup: two-launch gates {ms[0]:.3f} ms   one-launch gates {ms[1]:.3f} ms")
    L.rc_debug_set(b"gate_fused", 1)
    c = N.Conv2d(48, 192, 3, 1, 1).to(dev, bf).eval()
    x = torch.randn(8, 544, 960, 48, device=dev, dtype=bf)
    r = torch.randn(8, 544, 960, 192, device=dev, dtype=bf)
    print(f"48->192 at 8x544x960: plain {timed(lambda: c._nhwc(x), n=20, warm=30) / 1e3:.3f} ms   + residual {timed(lambda: c._nhwc(x, residual=r), n=20, warm=30) / 1e3:.3f} ms   "
          f"+ PixelShuffle {timed(lambda: c._nhwc(x, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2), n=20, warm=30) / 1e3:.3f} ms")
    xs = torch.randn(16, 2, 1920, 48, device=dev, dtype=bf)
    print(f"48->192 + PixelShuffle on the ring's row strips (16x2x1920): {timed(lambda: c._nhwc(xs, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2), n=50, warm=30):.1f} us")
