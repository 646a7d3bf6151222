#!/usr/bin/env python3
"""Per-phase s_memtime cycle counts of the 32x32x16 conv (conv32_kernel.hpp): one compute wave and one loader wave of block 8.
usage: conv32_phases.py [cin cout h w b] [--ps] [--gated] [--res]   env: FLAGS=<conv_flags> V=<conv32 variant>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
nums = [int(v) for v in sys.argv[1:] if v.lstrip("-").isdigit()]
cin, cout, h, w, b = nums if len(nums) == 5 else (192, 192, 544, 960, 8)
flags, variant = int(os.environ.get("FLAGS", "0")), int(os.environ.get("V", "1"))
L = ops.lib()
L.rc_debug_set(b"conv32", variant)
c = N.Conv2d(cin, cout, 3, 1, 1).to("cuda", torch.bfloat16)
x = torch.rand(b, h, w, cin, device="cuda").to(torch.bfloat16)
kw = {}
if "--gated" in sys.argv: kw = dict(gate=torch.rand(b, cin, device="cuda"), skip=torch.rand_like(x), store_input=True)
if "--res" in sys.argv: kw = dict(residual=torch.rand(b, h, w, cout, device="cuda").to(torch.bfloat16))
if "--ps" in sys.argv: kw = dict(out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)
dbg = torch.zeros(1024, dtype=torch.int64, device="cuda")
L.rc_debug_set(b"conv_flags", flags)
for _ in range(40): ops.conv2d(x, c, **kw)
torch.cuda.synchronize()
L.rc_debug_set_ptr(b"conv_phase_timing", dbg.data_ptr())
t0 = time.perf_counter()
ops.conv2d(x, c, **kw); torch.cuda.synchronize()
dt = time.perf_counter() - t0
L.rc_debug_set_ptr(b"conv_phase_timing", None); L.rc_debug_set(b"conv_flags", 0); L.rc_debug_set(b"conv32", 4)
d = dbg.cpu()
cw = d[:480].view(60, 8)[6:54].float(); lw = d[512:752].view(60, 4)[6:54].float()
epi = cw[:, 4][cw[:, 4] > 0]
print(f"{cin}->{cout} {h}x{w}x{b} {sys.argv[6:] if len(nums)==5 else ''} V={variant} flags={flags}: wall {dt*1e3:.2f} ms")
print(f"  compute wave / stage: mfma-a {cw[:,0].mean():.0f}  bar1 {cw[:,1].mean():.0f}  mfma-b {cw[:,2].mean():.0f}  bar2(mid-chunk) {cw[:,3].mean():.0f}  "
      f"epilogue {epi.mean() if len(epi) else 0:.0f} (x{len(epi)}/{len(cw)})  bar-after-epi {cw[:,5][cw[:,4]>0].mean() if len(epi) else 0:.0f}   sum {cw[:, :6].sum(1).mean():.0f}")
if variant == 1 and cin != 48:
    print(f"  (staged form) compute: mfma {cw[:,0].mean():.0f}  epilogue->LDS {epi.mean() if len(epi) else 0:.0f} (x{len(epi)}/{len(cw)})  barrier {cw[:,5].mean():.0f}   "
          f"loader: commit+issue {lw[:,0].mean():.0f}  drain {lw[:,1].mean():.0f}  barrier {lw[:,2].mean():.0f}")
print(f"  loader wave / stage:  issue_a+commit_b {lw[:,0].mean():.0f}  bar1 {lw[:,1].mean():.0f}  issue_b+commit_a {lw[:,2].mean():.0f}  bar2 {lw[:,3].mean():.0f}   sum {lw.sum(1).mean():.0f}")
