#!/usr/bin/env python3
"""Single-layer conv probe for profiling: python tools/conv_probe.py --cin 48 --cout 48 --h 1088 --w 1920 --b 8 [--gated] [--iters 5]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=48); ap.add_argument("--cout", type=int, default=48)
ap.add_argument("--h", type=int, default=1088); ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--b", type=int, default=8); ap.add_argument("--k", type=int, default=3)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--gated", action="store_true"); ap.add_argument("--residual", action="store_true")
ap.add_argument("--sums", action="store_true"); ap.add_argument("--ps", action="store_true")
ap.add_argument("--persist", type=int, default=1); ap.add_argument("--flags", type=int, default=0); ap.add_argument("--act", default="relu")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
dev = "cuda"
c = N.Conv2d(a.cin, a.cout, a.k, 1, a.k // 2).to(dev, dt)
x = torch.rand(a.b, a.h, a.w, a.cin, device=dev).to(dt)
kw = {} if a.act == "none" else dict(act=a.act)
if a.gated:
    kw.update(gate=torch.rand(a.b, a.cin, device=dev), skip=torch.rand_like(x), store_input=True)
if a.residual:
    kw.update(residual=torch.rand(a.b, a.h, a.w, a.cout, device=dev).to(dt))
if a.sums:
    kw.update(want_sums=True)
if a.ps:
    kw.update(out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)
ops.lib().rc_debug_set(b"persist", a.persist); ops.lib().rc_debug_set(b"conv_flags", a.flags)
for _ in range(60):   # the GPU idles at low clocks: reach steady state before timing
    ops.conv2d(x, c, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    ops.conv2d(x, c, **kw)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / a.iters
fl = 2.0 * a.b * a.h * a.w * a.cin * a.cout * a.k * a.k
print(f"conv {a.cin}->{a.cout} k{a.k} {a.b}x{a.h}x{a.w} {a.dtype} gated={a.gated} res={a.residual} sums={a.sums} persist={a.persist} flags={a.flags} act={a.act}: "
      f"{t*1e3:.3f} ms  {fl/t/1e12:.1f} TF/s")
