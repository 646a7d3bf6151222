#!/usr/bin/env python3
"""Stress kernel 4b against kernel 4 on one layer shape, back to back with other launches and no host synchronisation: counts and locates mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.nn as N
from realcamnet_amd import ops, _lib
lib = _lib.load()
DEV = "cuda"
g = torch.Generator().manual_seed(3)
cin, cout, k = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 32, 2)))
B, H, W = (int(v) for v in (sys.argv[4:7] if len(sys.argv) > 6 else (1, 128, 128)))
if k == 2:
    conv = ops._ConvView((torch.randn(cout, cin, 2, 2, generator=g) * 0.1).to(DEV, torch.bfloat16), torch.randn(cout, generator=g).to(DEV, torch.bfloat16))
else:
    conv = N.Conv2d(cin, cout, 3, 1, 1).to(DEV, torch.bfloat16).eval()
other = N.Conv2d(64, 64, 3, 1, 1).to(DEV, torch.bfloat16).eval()
big = N.Conv2d(128, 128, 3, 1, 1).to(DEV, torch.bfloat16).eval()
xo = torch.randn(2, 200, 300, 64, generator=g).to(DEV, torch.bfloat16)
xb = torch.randn(1, 256, 256, 128, generator=g).to(DEV, torch.bfloat16)
x = torch.randn(B, H, W, cin, generator=g).to(DEV, torch.bfloat16)
with torch.no_grad():
    lib.rc_debug_set(b"thin", 0); ref = ops.conv2d(x, conv, act="leaky", slope=0.1); torch.cuda.synchronize()
    lib.rc_debug_set(b"thin", 2)
    outs = []
    for it in range(int(os.environ.get("ITERS", "300"))):
        if it % 3 == 0: ops.conv2d(xo, other)
        if it % 3 == 1: ops.conv2d(xb, big, act="relu")
        outs.append(ops.conv2d(x, conv, act="leaky", slope=0.1))
        if it % 7 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
bad = 0
for it, o in enumerate(outs):
    if not torch.equal(o, ref):
        bad += 1
        if bad <= 6:
            d = (o.float() - ref.float()).abs()
            idx = (d > 0).nonzero()
            ys, xs, cs = idx[:, 1], idx[:, 2], idx[:, 3]
            print(f"iter {it}: {idx.shape[0]} values differ, max {d.max().item():.4g}; rows {ys.min().item()}..{ys.max().item()} cols {xs.min().item()}..{xs.max().item()} ch {cs.min().item()}..{cs.max().item()}; "
                  f"tiles (16x32) {sorted(set((int(y) // 16, int(xx) // 32) for y, xx in zip(ys.tolist()[:4000], xs.tolist()[:4000])))[:12]} nan {torch.isnan(o.float()).sum().item()}")
print(f"{cin}->{cout} k{k} {B}x{H}x{W}: {bad} of {len(outs)} launches differ from kernel 4")
