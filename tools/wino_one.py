#!/usr/bin/env python3
"""One fp32 3x3 layer launched N times (profiling target): wino_one.py cin cout H W [n] [winograd 0/1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
cin, cout, H, W = (int(v) for v in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ops.WINOGRAD = (sys.argv[6] != "0") if len(sys.argv) > 6 else True
dt = torch.bfloat16 if os.environ.get("DT") == "bf16" else torch.float32
c = N.Conv2d(cin, cout, 3, 1, 1).to("cuda", dt).eval()
x = torch.randn(int(os.environ.get("B", "1")), H, W, cin, device="cuda", dtype=dt)
with torch.no_grad():
    for _ in range(n):
        ops.conv2d(x, c, act="relu")
torch.cuda.synchronize()
