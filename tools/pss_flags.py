import os, sys
sys.path.insert(0, "/root/repo")
import torch
from realcamnet_amd import networks as N, ops
L = ops.lib()
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, H, W = 8, 1088, 1920
x = torch.randn(B, H, W, 48, device=dev, dtype=bf)
c1 = N.Conv2d(48, 192, 3, 1, 1).to(dev, bf).eval()
with torch.no_grad():
    for pss in (0, 1):
        L.rc_debug_set(b"pss", pss)
        for flags in (0, 1, 8):
            L.rc_debug_set(b"conv_flags", flags)
            print(f"pss {pss} flags {flags}: {timed(lambda: c1._nhwc(x, out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)):6.3f} ms")
