#!/usr/bin/env python3
"""The multi-chunk layers of cfg3 (192 -> 192 at H/2, 128 -> 128 at H/4 and H/8, 512 -> 512 at H/8, 192 -> 48, 48 -> 192 + residual) one by one: ms per launch
after 50 warm-up launches.  Used with RC_HIP_LIB=<alternative build> for A/B experiments on conv_mfma_wsm_kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)


def timed(fn, n=20, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    tot = 0.0
    for cin, cout, H, W, res, count in ((192, 192, 544, 960, False, 2), (192, 192, 544, 960, True, 2), (128, 128, 272, 480, False, 14), (128, 128, 136, 240, False, 9),
                                        (512, 512, 136, 240, False, 2), (192, 48, 544, 960, False, 1), (512, 128, 272, 480, False, 1), (128, 512, 136, 240, False, 1)):
        c = N.Conv2d(cin, cout, 3, 1, 1).to(dev, bf).eval()
        x = torch.randn(8, H, W, cin, device=dev, dtype=bf)
        r = torch.randn(8, H, W, cout, device=dev, dtype=bf) if res else None
        t = timed(lambda: c._nhwc(x, act="relu") if r is None else c._nhwc(x, residual=r))
        fl = 2 * 8 * H * W * cin * cout * 9
        tot += t * count
        print(f"{cin:4d} -> {cout:4d}  {H}x{W}{' +res' if res else '     '}: {t:.3f} ms  {fl / t / 1e9:.0f} TF/s   (x{count} per forward)")
    print(f"weighted total {tot:.2f} ms")
