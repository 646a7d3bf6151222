#!/usr/bin/env python3
"""Exact-arithmetic conv check: small-integer data so bf16/fp32 results must equal F.conv2d exactly."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from realcamnet_amd import networks as N, ops

def check(cin, cout, h, w, dt, persist, ksz=3, b=1):
    g = torch.Generator().manual_seed(0)
    c = N.Conv2d(cin, cout, ksz, 1, ksz // 2)
    with torch.no_grad():
        c.weight.copy_(torch.randint(-2, 3, c.weight.shape, generator=g).float() / 2)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    x = torch.randint(-2, 3, (b, cin, h, w), generator=g).float() / 2
    ref = F.conv2d(x, c.weight, c.bias, padding=ksz // 2)
    ops.lib().rc_debug_set(b"persist", persist)
    y = c.to("cuda", dt)(x.to("cuda", dt)).float().cpu()
    ops.lib().rc_debug_set(b"persist", 1)
    bad = (y != ref)
    msg = f"cin={cin} cout={cout} {h}x{w} k{ksz} {str(dt)[6:]} persist={persist}: bad={int(bad.sum())}/{bad.numel()}"
    if bad.any():
        idx = bad.nonzero()
        msg += f" chans={sorted(set(idx[:,1].tolist()))[:12]} rows={sorted(set(idx[:,2].tolist()))[:12]} cols={sorted(set(idx[:,3].tolist()))[:16]}"
        msg += f" maxabs={float((y-ref).abs().max()):.3g} finite={bool(torch.isfinite(y).all())}"
    print(msg)

for dt in (torch.bfloat16, torch.float32):
    for persist in (1, 0):
        check(48, 48, 8, 32, dt, persist)
        check(48, 48, 16, 40, dt, persist)
check(48, 48, 8, 32, torch.bfloat16, 1, ksz=1)
check(48, 16, 8, 32, torch.bfloat16, 1)
check(96, 48, 8, 32, torch.bfloat16, 0)
check(64, 64, 8, 32, torch.bfloat16, 0)
