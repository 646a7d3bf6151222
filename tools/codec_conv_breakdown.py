#!/usr/bin/env python3
"""Per-launch breakdown of realcam::conv2d inside one codec forward (4 frames of 4K): shape, time (HIP events), TF/s.  Finds the layers worth fusing."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
torch.manual_seed(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = M.raw2bit.raw_compression_tcm_final().eval().to("cuda", torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(1)
mosaic = torch.rand(frames, 1, 2160, 3840, generator=g, device="cuda").to(torch.bfloat16)
coord = ops.make_coord(frames, 1080, 1920, device="cuda", dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(3): net.forward_mosaic(mosaic, None, coord)
torch.cuda.synchronize()
recs = []
from torch.utils._python_dispatch import TorchDispatchMode
class T(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith("realcam."):
            return func(*args, **(kwargs or {}))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = func(*args, **(kwargs or {})); e1.record(); torch.cuda.synchronize()
        shp = tuple(args[0].shape) if hasattr(args[0], "shape") else ()
        extra = ""
        if "conv2d_fold2" in name:
            extra = f"cout={args[3]} k=s2 act={args[4]}"
        elif "conv2d" in name:
            extra = f"cout={args[3]} k={args[4]} act={args[5]} mode={args[14]}"
        recs.append((name.replace("realcam.", "").replace(".default", ""), shp, extra, e0.elapsed_time(e1)))
        return out
with torch.no_grad(), T():
    net.forward_mosaic(mosaic, None, coord)
tot = sum(r[3] for r in recs)
print(f"{len(recs)} launches, {tot:.1f} ms serialized")
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, t in recs:
    agg[(n, s, e)][0] += 1; agg[(n, s, e)][1] += t
for (n, s, e), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "45"))]:
    fl = ""
    if n == "conv2d_fold2":
        cout = int(e.split()[0][5:]); b, h, w, cin = s
        fl = f"{2.0*b*(h//2)*(w//2)*cin*cout*9/(t/c)/1e9:7.0f} TF/s"
    elif n == "conv2d":
        cout = int(e.split()[0][5:]); k = int(e.split()[1][2:]); b, h, w, cin = s
        fl = f"{2.0*b*h*w*cin*cout*k*k/(t/c)/1e9:7.0f} TF/s"
    print(f"{t:7.2f} ms  x{c:3d}  {t/c*1e3:8.1f} us  {n:22s} {str(s):26s} {e:32s} {fl}")
