#!/usr/bin/env python3
"""One 3x3 layer launched N times (profiling target): conv_one.py cin cout H W B [n] [form]   (DT=bf16|f32; forms: relu, relu_sums, res, scale_res, none, dwt = conv -> Haar DWT in one launch, conv+dwt = the two launches)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
cin, cout, H, W, B = (int(v) for v in sys.argv[1:6])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 10
form = sys.argv[7] if len(sys.argv) > 7 else "relu"
dt = torch.float32 if os.environ.get("DT") == "f32" else torch.bfloat16
c = N.Conv2d(cin, cout, 3, 1, 1).to("cuda", dt).eval()
x = torch.randn(B, H, W, cin, device="cuda", dtype=dt)
kw = dict(act="relu") if form.startswith("relu") else {}
if form == "relu_sums": kw["want_sums"] = True
if form in ("res", "scale_res"): kw["residual"] = torch.randn(B, H, W, cout, device="cuda", dtype=dt)
if form == "scale_res": kw["out_scale"] = torch.rand(B, cout, device="cuda")
def timed(fn, n, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
if form == "dwt": kw["out_mode"] = ops.RC_OUT_NHWC_DWT
with torch.no_grad():
    if form == "conv+dwt":
        dwt = N.DWTForward(cout).to("cuda", dt).eval()
        t = timed(lambda: ops.dwt_forward(ops.conv2d(x, c), dwt), n)
    else:
        t = timed(lambda: ops.conv2d(x, c, **kw), n)
print(f"{cin}->{cout} {H}x{W} B={B} {form}: {t:.1f} us  lib={os.environ.get('RC_HIP_LIB', 'in-tree')}")
