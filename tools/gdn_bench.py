import sys; sys.path.insert(0, "/root/repo")
import torch, realcamnet_amd.tcm as T
from realcamnet_amd import ops
gdn = T.GDN(128).to("cuda", torch.bfloat16).eval()
x = torch.randn(4, 576, 960, 128, device="cuda").bfloat16(); idn = torch.randn_like(x)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print("gdn chain + identity: %.1f us" % t(lambda: gdn._nhwc(x, idn)))
    print("gdn chain: %.1f us" % t(lambda: gdn._nhwc(x)))
    ops.FUSE_MLP = False
    print("three launches + identity: %.1f us" % t(lambda: gdn._nhwc(x, idn)))
