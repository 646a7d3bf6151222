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
B, H, W = 8, 2176, 3840
t = torch.randn(B, H, W, 48, device=dev, dtype=bf)
with torch.no_grad():
    for persist in (1, 2, 3, 0):
        L.rc_debug_set(b"persist", persist)
        c2 = N.Conv2d(48, 3, 3, 1, 1).to(dev, bf).eval()
        print(f"persist {persist}: 48->3 NCHW {timed(lambda: c2._nhwc(t, out_mode=ops.RC_OUT_NCHW, out_dtype=bf)):6.3f} ms   NHWC-out(3ch n/a)")
    L.rc_debug_set(b"persist", 1)
    # how fast can this tensor be read at all?  a plain reduction-free pass: copy 6.4 GB -> discard via sum of a slice
    x = t.view(-1)
    print(f"torch sum over the 6.4 GB tensor: {timed(lambda: x.sum()):6.3f} ms")
    c48 = N.Conv2d(48, 48, 3, 1, 1).to(dev, bf).eval()
    print(f"48->48 at this size (reads 6.4, writes 6.4 GB): {timed(lambda: c48._nhwc(t)):6.3f} ms")
