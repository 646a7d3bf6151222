import sys
sys.path.insert(0, "/root/repo")
import torch
from realcamnet_amd import networks as N, ops
L = ops.lib()
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
c = N.Conv2d(48, 48, 3, 1, 1).to(dev, bf).eval()
with torch.no_grad():
    for (H, W) in ((1088, 1920), (544, 960), (272, 480)):
        x = torch.randn(8, H, W, 48, device=dev, dtype=bf); r = torch.randn_like(x)
        for rep in range(2):
            for flags in (0, 16):
                L.rc_debug_set(b"conv_flags", flags)
                print(f"{H}x{W} conv_flags {flags:2d}: plain+relu {timed(lambda: c._nhwc(x, act='relu')):7.4f}  +residual {timed(lambda: c._nhwc(x, residual=r)):7.4f}  +sums {timed(lambda: ops.conv2d(x, c, want_sums=True)[0]):7.4f} ms")
L.rc_debug_set(b"conv_flags", 0)
