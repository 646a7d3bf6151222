#!/usr/bin/env python3
"""s_memtime phase breakdown of the fused conv-pair kernel (wave 0 of one block) + steady-state timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
mode = sys.argv[1] if len(sys.argv) > 1 else "sums"
c1 = N.Conv2d(48, 48, 3, 1, 1).to("cuda", torch.bfloat16); c2 = N.Conv2d(48, 48, 3, 1, 1).to("cuda", torch.bfloat16)
x = torch.rand(8, 1088, 1920, 48, device="cuda").to(torch.bfloat16)
kw = dict(want_sums=True)
if mode == "gated":
    kw.update(gate=torch.rand(8, 48, device="cuda"), skip=torch.rand_like(x), store_input=True)
def run(): return ops.conv_pair(x, c1, c2, act="relu", **kw)
def run2():
    t = ops.conv2d(x, c1, act="relu", **{k: v for k, v in kw.items() if k != "want_sums"})
    t = t[0] if isinstance(t, tuple) else t
    return ops.conv2d(t, c2, want_sums=True)
for f, name in ((run, "fused"), (run2, "two launches")):
    for _ in range(40): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{mode} {name}: {e0.elapsed_time(e1)/100:.3f} ms")
dbg = torch.zeros(1024, dtype=torch.int64, device="cuda")
ops.lib().rc_debug_set_ptr(b"conv_phase_timing", dbg.data_ptr())
run(); torch.cuda.synchronize()
ops.lib().rc_debug_set_ptr(b"conv_phase_timing", None)
d = dbg.cpu()[:480].view(60, 8)[4:56].float().mean(0)
print("cycles/tile: mma1 %d  epi1 %d  barA %d  stage %d  mma2 %d  epi2 %d  barB %d  | sum %d" % (*d[:7].tolist(), d[:7].sum()))
