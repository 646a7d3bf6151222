#!/usr/bin/env python3
"""fp32 3x3 layers of cfg2 (LiteISPNet, 1080p, B = 1) one by one: Winograd F(2x2,3x3) (rc_conv_desc.algo 1, csrc/wino.hip) against the implicit GEMM,
ms per launch after 50 warm-up launches, same box, alternating.  TF/s columns are ALGORITHMIC (2 * 9 * cin * cout per pixel) for both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
dev = "cuda"
torch.manual_seed(0)


def timed(fn, n=30, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with torch.no_grad():
    for cin, cout, H, W, form in ((64, 64, 544, 960, "relu"), (64, 64, 544, 960, "relu_sums"), (64, 64, 544, 960, "scale_res"), (64, 64, 272, 480, "relu"),
                                  (128, 128, 136, 240, "relu"), (128, 128, 68, 120, "relu"), (256, 64, 272, 480, "none"), (512, 128, 136, 240, "none"), (64, 256, 544, 960, "none"),
                                  (48, 48, 544, 960, "relu"), (192, 192, 272, 480, "relu")):
        c = N.Conv2d(cin, cout, 3, 1, 1).to(dev).eval()
        x = torch.randn(B, H, W, cin, device=dev)
        kw = dict(act="relu") if form.startswith("relu") else {}
        if form == "relu_sums":
            kw["want_sums"] = True
        if form == "scale_res":
            kw = dict(out_scale=torch.rand(B, cout, device=dev), residual=torch.randn(B, H, W, cout, device=dev))
        res = []
        for wino in (False, True, False, True):
            ops.WINOGRAD = wino
            res.append(timed(lambda: ops.conv2d(x, c, **kw)))
        ops.WINOGRAD = True
        fl = 2 * B * H * W * cin * cout * 9
        t0, t1 = min(res[0], res[2]), min(res[1], res[3])
        print(f"{cin:4d} -> {cout:4d}  {H}x{W} {form:10s}: direct {t0 * 1e3:7.1f} us ({fl / t0 / 1e9:5.1f} TF/s)   winograd {t1 * 1e3:7.1f} us ({fl / t1 / 1e9:5.1f} TF/s alg.)   x{t0 / t1:.2f}", flush=True)
