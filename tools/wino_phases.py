#!/usr/bin/env python3
"""s_memtime stamps of the Winograd kernel's block 0, waves 0 and 1 (rc_debug_set_ptr("conv_phase_timing")): cycles per phase and stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops, _lib
L = _lib.load()
cin, cout, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 64, 544, 960)
c = N.Conv2d(cin, cout, 3, 1, 1).to("cuda").eval()
x = torch.randn(1, H, W, cin, device="cuda")
dbg = torch.zeros(1024, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(30): ops.conv2d(x, c, act="relu")
    torch.cuda.synchronize()
    L.rc_debug_set_ptr(b"conv_phase_timing", dbg.data_ptr())
    for _ in range(3): ops.conv2d(x, c, act="relu")
    torch.cuda.synchronize()
    L.rc_debug_set_ptr(b"conv_phase_timing", None)
d = dbg.cpu()[:512].view(2, 32, 8)
names = ["issue", "transform", "mfma0", "mfma1", "epilogue", "commit", "barrier"]
for w in range(2):
    print(f"wave {w}: stage total | " + " | ".join(names))
    for g in range(0, 24):
        t = d[w, g]
        nxt = d[w, g + 1, 0] if g + 1 < 32 else t[7]
        print(f"  g={g:2d} {int(nxt - t[0]):6d} | " + " | ".join(f"{int(t[i + 1] - t[i]):6d}" for i in range(7)))
    tot = (d[w, 1:24, 0] - d[w, 0:23, 0]).float().mean()
    print(f"  mean stage {tot:.0f} cycles; phases " + ", ".join(f"{names[i]} {(d[w, :24, i + 1] - d[w, :24, i]).float().mean():.0f}" for i in range(7)))
