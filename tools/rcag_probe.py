#!/usr/bin/env python3
"""RCAGroup (4 RCAB + conv) at level 0 / level 1: the shipped schedule (gate + skip folded into the NEXT conv's staging: r, skip in; x, t out) against the
schedule an EARLY gate would allow (gate known before conv2 -> conv2's epilogue writes x_new = conv2(t) * g + x): timed here with the existing kernels as
proxies (conv + sums, conv + residual), before any kernel work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)


def timed(fn, n=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for (B, H, W) in ((8, 1088, 1920), (8, 544, 960)):
        rg = N.RCAGroup(48, 48, nb=4).to(dev, bf).eval()
        x = torch.randn(B, H, W, 48, device=dev, dtype=bf)
        for _ in range(5): rg._nhwc(x)
        t_ship = timed(lambda: rg._nhwc(x))
        blocks = list(rg.rg)

        def proxy():
            xk = x
            for blk in blocks[:-1]:
                c1, c2 = blk.res[0], blk.res[2]
                t, sums = c1._nhwc(xk, want_sums=True)
                ops.ca_gate(sums, H * W, blk.ca)
                xk = c2._nhwc(t, residual=xk)
            return blocks[-1]._nhwc(xk, residual=x)
        t_new = timed(proxy)
        c = blocks[0].res[0]
        t_plain = timed(lambda: c._nhwc(x, act="relu"))
        t_sums = timed(lambda: c._nhwc(x, want_sums=True))
        t_res = timed(lambda: c._nhwc(x, residual=x))
        print(f"{B}x{H}x{W}: RCAGroup shipped {t_ship:.3f} ms; early-gate schedule (proxy kernels) {t_new:.3f} ms  x{t_ship / t_new:.3f};  single layers: relu {t_plain:.3f}  +sums {t_sums:.3f}  +residual {t_res:.3f}")
